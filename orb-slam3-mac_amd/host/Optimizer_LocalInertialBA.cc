// Optimizer_LocalInertialBA.cc -- Optimizer::LocalInertialBA(KeyFrame*, bool*, Map*, bool bLarge, bool bRecInit) with the
// reference's signature (include/Optimizer.h:99) around the HIP solver.  The host parts of the reference function stay host code,
// restated here in the reference's order (src/Optimizer.cc): temporal window through mPrevKF :4588-4607, local map points
// :4609-4627, the fixed keyframe before the window :4629-4642, fixed covisible keyframes :4676-4697 (maxCovKF = 0 at :4645: no
// optimizable visual keyframes), locked fail check / erase / write-back :5090-5170.  What was the g2o block (:4702-5049) is flat
// packing in the insertion order of :4784-5034 plus ONE call of orbhip_inertial_ba_solve_batch.
// The EdgeInertial information (G2oTypes.cc:702-714: C(0:9,0:9) inverted through a float SVD, symmetrised, eigenvalues below
// 1e-12 clamped) is computed here in double from the float covariance by a cyclic Jacobi eigen-decomposition (pseudo-inverse
// with the same clamp); the random-walk informations (:4845-4863) by a 3x3 inverse.  pbStopFlag is handed to g2o only after
// optimize() in the reference (:5050-5051) and therefore has no effect; it has none here.
// Pinhole and KannalaBrandt8 cameras; keyframes of a two-camera rig (mpCamera2, mTrl, NLeft, mvKeysRight) get EdgeMono(1) edges for
// their right-camera observations (:5000-5031), with the reference's quirk that such an edge takes its inverse sigma^2 from the
// octave of the LEFT keypoint variable (kpUn, :5019-5020), a default-constructed keypoint (octave 0) when there is no left one.
#include "Optimizer.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include "hip_context.h"
#include <list>
#include <map>
#include <utility>
#include "../../include/orbhip.h"

namespace ORB_SLAM3 {

namespace {

inline orbhip_ctx *thread_ctx() { return hip::ThreadContext(); }      // one context per calling thread, GPU of hip::GetDevice() (hip_context.h)

// eigen-decomposition of a symmetric n x n matrix (n <= 9), cyclic Jacobi: A = V diag(w) V^T
void jacobi_eig(int n, double *A, double *V, double *w)
{
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[n * i + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) off += A[n * i + j] * A[n * i + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                if (A[n * p + q] == 0.0) continue;
                const double theta = (A[n * q + q] - A[n * p + p]) / (2 * A[n * p + q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) { const double a = A[n * k + p], b = A[n * k + q]; A[n * k + p] = c * a - s * b; A[n * k + q] = s * a + c * b; }
                for (int k = 0; k < n; k++) { const double a = A[n * p + k], b = A[n * q + k]; A[n * p + k] = c * a - s * b; A[n * q + k] = s * a + c * b; }
                for (int k = 0; k < n; k++) { const double a = V[n * k + p], b = V[n * k + q]; V[n * k + p] = c * a - s * b; V[n * k + q] = s * a + c * b; }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[n * i + i];
}

// information of EdgeInertial from the preintegration covariance (G2oTypes.cc:702-714)
void inertial_information(const cv::Mat &C, double *info81)
{
    double A[81], V[81], w[9];
    for (int r = 0; r < 9; r++) for (int c = 0; c < 9; c++) A[9 * r + c] = 0.5 * ((double)C.at<float>(r, c) + (double)C.at<float>(c, r));
    jacobi_eig(9, A, V, w);
    double wmax = 0;
    for (int i = 0; i < 9; i++) wmax = std::fmax(wmax, std::fabs(w[i]));
    for (int i = 0; i < 9; i++) {
        double inv = (w[i] > wmax * 1e-7 * 9) ? 1.0 / w[i] : 0.0;        // cv::invert(DECOMP_SVD) on floats drops singular values below FLT_EPSILON-scale
        if (inv < 1e-12) inv = 0.0;                                      // "if (eigs[i] < 1e-12) eigs[i] = 0"
        w[i] = inv;
    }
    for (int r = 0; r < 9; r++) for (int c = 0; c < 9; c++) {
        double s = 0;
        for (int k = 0; k < 9; k++) s += V[9 * r + k] * w[k] * V[9 * c + k];
        info81[9 * r + c] = s;
    }
}

void inv3_block(const cv::Mat &C, int o, double *out9)           // C.rowRange(o, o+3).colRange(o, o+3).inv(DECOMP_SVD)
{
    double A[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[3 * r + c] = (double)C.at<float>(o + r, o + c);
    const double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
    const double id = 1.0 / (A[0] * c0 + A[1] * c1 + A[2] * c2);
    out9[0] = c0 * id; out9[1] = (A[2] * A[7] - A[1] * A[8]) * id; out9[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    out9[3] = c1 * id; out9[4] = (A[0] * A[8] - A[2] * A[6]) * id; out9[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    out9[6] = c2 * id; out9[7] = (A[1] * A[6] - A[0] * A[7]) * id; out9[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    for (int i = 0; i < 9; i++) out9[i] = (double)(float)out9[i];      // InfoG(r,c) = cvInfoG.at<float>(r,c)
}

void put3x3(double *dst, const cv::Mat &m) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) dst[3 * r + c] = (double)m.at<float>(r, c); }
void put3(double *dst, const cv::Mat &m) { for (int r = 0; r < 3; r++) dst[r] = (double)m.at<float>(r); }

}  // namespace

void Optimizer::LocalInertialBA(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, bool bLarge, bool bRecInit)
{
    (void)pbStopFlag;
    Map *pCurrentMap = pKF->GetMap();
    int maxOpt = 10;
    if (bLarge) maxOpt = 25;                                                   // :4579-4585 (opt_it follows in the solver's parameters)
    const int Nd = std::min((int)pCurrentMap->KeyFramesInMap() - 2, maxOpt);

    std::vector<KeyFrame *> vpOptimizableKFs;
    vpOptimizableKFs.reserve(Nd > 0 ? Nd : 1);
    vpOptimizableKFs.push_back(pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    for (int i = 1; i < Nd; i++) {                                             // :4597-4606
        if (vpOptimizableKFs.back()->mPrevKF) {
            vpOptimizableKFs.push_back(vpOptimizableKFs.back()->mPrevKF);
            vpOptimizableKFs.back()->mnBALocalForKF = pKF->mnId;
        } else
            break;
    }
    int N = vpOptimizableKFs.size();

    std::list<MapPoint *> lLocalMapPoints;                                     // :4611-4627
    for (int i = 0; i < N; i++) {
        std::vector<MapPoint *> vpMPs = vpOptimizableKFs[i]->GetMapPointMatches();
        for (MapPoint *pMP : vpMPs)
            if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) { lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pKF->mnId; }
    }

    std::list<KeyFrame *> lFixedKeyFrames;                                     // :4630-4642
    if (vpOptimizableKFs.back()->mPrevKF) {
        lFixedKeyFrames.push_back(vpOptimizableKFs.back()->mPrevKF);
        vpOptimizableKFs.back()->mPrevKF->mnBAFixedForKF = pKF->mnId;
    } else {
        vpOptimizableKFs.back()->mnBALocalForKF = 0;
        vpOptimizableKFs.back()->mnBAFixedForKF = pKF->mnId;
        lFixedKeyFrames.push_back(vpOptimizableKFs.back());
        vpOptimizableKFs.pop_back();
    }
    // :4645-4673: maxCovKF = 0, the loop over the covisible keyframes breaks before its first iteration

    const size_t maxFixKF = 200;                                               // :4676-4697
    for (MapPoint *pMP : lLocalMapPoints) {
        std::map<KeyFrame *, std::tuple<int, int>> observations = pMP->GetObservations();
        for (auto &ob : observations) {
            KeyFrame *pKFi = ob.first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad()) { lFixedKeyFrames.push_back(pKFi); break; }
            }
        }
        if (lFixedKeyFrames.size() >= maxFixKF) break;
    }

    // ---- flat window: optimizable keyframes, then the fixed ones
    N = vpOptimizableKFs.size();
    std::vector<KeyFrame *> vKF(vpOptimizableKFs.begin(), vpOptimizableKFs.end());
    vKF.insert(vKF.end(), lFixedKeyFrames.begin(), lFixedKeyFrames.end());
    const int nKF = vKF.size();
    if (nKF == 0 || N == 0) return;
    std::map<KeyFrame *, int> kfIndex;
    std::vector<double> kf((size_t)ORBHIP_IBA_KF * nKF, 0.0);
    std::vector<uint8_t> fixed(nKF), imu(nKF);
    for (int i = 0; i < nKF; i++) {
        KeyFrame *pKFi = vKF[i];
        kfIndex[pKFi] = i;
        fixed[i] = i >= N;
        imu[i] = pKFi->bImu ? 1 : 0;
        double *s = &kf[(size_t)ORBHIP_IBA_KF * i];
        put3x3(s, pKFi->GetImuRotation()); put3(s + 9, pKFi->GetImuPosition());               // ImuCamPose(KeyFrame*), G2oTypes.cc:25-30
        if (pKFi->bImu) { put3(s + 12, pKFi->GetVelocity()); put3(s + 15, pKFi->GetGyroBias()); put3(s + 18, pKFi->GetAccBias()); }
    }
    orbhip_iba_window w = {};
    w.n_kf = nKF; w.kf_fixed = fixed.data(); w.kf_imu = imu.data();
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) w.Rcb[3 * r + c] = (double)pKF->mImuCalib.Tcb.at<float>(r, c); w.tcb[r] = (double)pKF->mImuCalib.Tcb.at<float>(r, 3); }
    w.fx = pKF->mpCamera->getParameter(0); w.fy = pKF->mpCamera->getParameter(1); w.cx = pKF->mpCamera->getParameter(2); w.cy = pKF->mpCamera->getParameter(3);
    w.bf = pKF->mbf;
    w.camera_model = pKF->mpCamera->GetType() == pKF->mpCamera->CAM_FISHEYE ? 1 : 0;
    for (int i = 0; i < 4; i++) w.kb[i] = w.camera_model ? (double)pKF->mpCamera->getParameter(4 + i) : 0.0;
    if (pKF->mpCamera2) {                                                      // ImuCamPose(KeyFrame*), G2oTypes.cc:57-67
        GeometricCamera *c2 = pKF->mpCamera2;
        w.has_cam2 = 1;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) w.Trl[4 * r + c] = (double)pKF->mTrl.at<float>(r, c);
        w.fx2 = c2->getParameter(0); w.fy2 = c2->getParameter(1); w.cx2 = c2->getParameter(2); w.cy2 = c2->getParameter(3);
        w.camera2_model = c2->GetType() == c2->CAM_FISHEYE ? 1 : 0;
        for (int i = 0; i < 4; i++) w.kb2[i] = w.camera2_model ? (double)c2->getParameter(4 + i) : 0.0;
    }

    // ---- inertial edges (:4784-4868)
    std::vector<int32_t> in1, in2;
    std::vector<double> preint, info, infog, infoa;
    std::vector<uint8_t> robust;
    for (int i = 0; i < N; i++) {
        KeyFrame *pKFi = vpOptimizableKFs[i];
        if (!pKFi->mPrevKF) { fprintf(stderr, "NOT INERTIAL LINK TO PREVIOUS FRAME!!!!\n"); continue; }
        if (!(pKFi->bImu && pKFi->mPrevKF->bImu && pKFi->mpImuPreintegrated)) { fprintf(stderr, "ERROR building inertial edge\n"); continue; }
        pKFi->mpImuPreintegrated->SetNewBias(pKFi->mPrevKF->GetImuBias());
        auto it1 = kfIndex.find(pKFi->mPrevKF);
        if (it1 == kfIndex.end()) { fprintf(stderr, "Error: inertial edge to a keyframe outside the window\n"); continue; }       // optimizer.vertex() == NULL, :4812-4816
        IMU::Preintegrated *pInt = pKFi->mpImuPreintegrated;
        in1.push_back(it1->second); in2.push_back(i);
        const size_t o = preint.size();
        preint.resize(o + ORBHIP_IBA_PREINT);
        double *p = &preint[o];
        p[0] = (double)pInt->dT;
        put3x3(p + 1, pInt->dR); put3(p + 10, pInt->dV); put3(p + 13, pInt->dP);
        put3x3(p + 16, pInt->JRg); put3x3(p + 25, pInt->JVg); put3x3(p + 34, pInt->JVa); put3x3(p + 43, pInt->JPg); put3x3(p + 52, pInt->JPa);
        p[61] = pInt->b.bwx; p[62] = pInt->b.bwy; p[63] = pInt->b.bwz; p[64] = pInt->b.bax; p[65] = pInt->b.bay; p[66] = pInt->b.baz;
        double I9[81];
        inertial_information(pInt->C, I9);
        const bool rk = (i == N - 1 || bRecInit);                              // :4828-4838
        if (i == N - 1) for (double &v : I9) v *= 1e-2;
        info.insert(info.end(), I9, I9 + 81);
        robust.push_back(rk ? 1 : 0);
        double G[9], A[9];
        inv3_block(pInt->C, 9, G); inv3_block(pInt->C, 12, A);
        infog.insert(infog.end(), G, G + 9); infoa.insert(infoa.end(), A, A + 9);
    }
    w.n_inertial = in1.size(); w.in_kf1 = in1.data(); w.in_kf2 = in2.data(); w.in_preint = preint.data(); w.in_info = info.data();
    w.in_info_g = infog.data(); w.in_info_a = infoa.data(); w.in_robust = robust.data();

    // ---- map points + visual edges (:4914-5034)
    std::vector<MapPoint *> vMP(lLocalMapPoints.begin(), lLocalMapPoints.end());
    std::vector<double> points(3 * vMP.size());
    std::vector<int32_t> eKF, ePoint;
    std::vector<double> obs, invS2;
    std::vector<uint8_t> stereo, close_;
    std::vector<std::pair<KeyFrame *, MapPoint *>> edgeOwner;
    for (size_t l = 0; l < vMP.size(); l++) {
        MapPoint *pMP = vMP[l];
        const cv::Mat Xw = pMP->GetWorldPos();
        for (int k = 0; k < 3; k++) points[3 * l + k] = (double)Xw.at<float>(k);
        const std::map<KeyFrame *, std::tuple<int, int>> observations = pMP->GetObservations();
        for (auto &ob : observations) {
            KeyFrame *pKFi = ob.first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) continue;
            if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;
            auto it = kfIndex.find(pKFi);
            if (it == kfIndex.end()) continue;
            const int leftIndex = std::get<0>(ob.second);
            cv::KeyPoint kpUn;                                                 // as in the reference: stays default (octave 0) without a left observation
            if (leftIndex != -1) {
                kpUn = pKFi->mvKeysUn[leftIndex];
                const float ur = pKFi->mvuRight[leftIndex];
                eKF.push_back(it->second); ePoint.push_back((int32_t)l);
                obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(ur);
                stereo.push_back(ur < 0 ? 0 : 1);                              // :4933 / :4966
                invS2.push_back((double)(pKFi->mvInvLevelSigma2[kpUn.octave] / 1.0f));       // uncertainty2 == 1 for Pinhole and KannalaBrandt8
                close_.push_back(pMP->mTrackDepth < 10.f ? 1 : 0);
                edgeOwner.push_back(std::make_pair(pKFi, pMP));
            }
            if (pKFi->mpCamera2) {                                             // :5000-5031
                int rightIndex = std::get<1>(ob.second);
                if (rightIndex != -1) {
                    rightIndex -= pKFi->NLeft;
                    const cv::KeyPoint &kp = pKFi->mvKeysRight[rightIndex];
                    eKF.push_back(it->second); ePoint.push_back((int32_t)l);
                    obs.push_back(kp.pt.x); obs.push_back(kp.pt.y); obs.push_back(-1.0);
                    stereo.push_back(2);
                    invS2.push_back((double)(pKFi->mvInvLevelSigma2[kpUn.octave] / 1.0f));
                    close_.push_back(pMP->mTrackDepth < 10.f ? 1 : 0);
                    edgeOwner.push_back(std::make_pair(pKFi, pMP));
                }
            }
        }
    }
    w.n_points = vMP.size(); w.n_edges = eKF.size(); w.edge_kf = eKF.data(); w.edge_point = ePoint.data(); w.edge_obs = obs.data();
    w.edge_stereo = stereo.data(); w.edge_inv_sigma2 = invS2.data(); w.edge_close = close_.data();

    orbhip_ctx *ctx = thread_ctx();
    if (!ctx) { fprintf(stderr, "orbhip: no HIP device (LocalInertialBA has no CPU fallback)\n"); abort(); }
    orbhip_iba_params prm;
    orbhip_iba_default_params(&prm, bLarge ? 1 : 0);
    std::vector<uint8_t> outlier(eKF.size() + 1);
    double *pk = kf.data(), *px = points.data();
    uint8_t *po = outlier.data();
    orbhip_iba_stats st;
    const int rc = orbhip_inertial_ba_solve_batch(ctx, &w, 1, &prm, &pk, &px, &po, &st);
    if (rc != ORBHIP_OK) { fprintf(stderr, "orbhip LocalInertialBA: %s\n", orbhip_last_error()); return; }

    std::vector<std::pair<KeyFrame *, MapPoint *>> vToErase;                   // :5056-5088
    for (size_t e = 0; e < eKF.size(); e++) {
        if (edgeOwner[e].second->isBad()) continue;
        if (outlier[e]) vToErase.push_back(edgeOwner[e]);
    }

    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
    if (st.failed) { fprintf(stderr, "FAIL LOCAL-INERTIAL BA!!!!\n"); return; }           // :5096-5100
    for (auto &er : vToErase) { er.first->EraseMapPointMatch(er.second); er.second->EraseObservation(er.first); }
    for (KeyFrame *pKFi : lFixedKeyFrames) pKFi->mnBAFixedForKF = 0;

    for (int i = 0; i < N; i++) {                                              // :5118-5140
        KeyFrame *pKFi = vpOptimizableKFs[i];
        const double *s = &kf[(size_t)ORBHIP_IBA_KF * i];
        cv::Mat Tcw = cv::Mat::eye(4, 4, CV_32F);                              // Rcw = Rcb Rwb^T, tcw = -Rcw twb + tcb
        for (int r = 0; r < 3; r++) {
            double t = w.tcb[r];
            for (int c = 0; c < 3; c++) {
                double a = 0;
                for (int k = 0; k < 3; k++) a += w.Rcb[3 * r + k] * s[3 * c + k];
                Tcw.at<float>(r, c) = (float)a;
                t -= a * s[9 + c];
            }
            Tcw.at<float>(r, 3) = (float)t;
        }
        pKFi->SetPose(Tcw);
        pKFi->mnBALocalForKF = 0;
        if (pKFi->bImu) {
            cv::Mat V(3, 1, CV_32F);
            for (int k = 0; k < 3; k++) V.at<float>(k) = (float)s[12 + k];
            pKFi->SetVelocity(V);
            pKFi->SetNewBias(IMU::Bias((float)s[18], (float)s[19], (float)s[20], (float)s[15], (float)s[16], (float)s[17]));
        }
    }
    for (size_t l = 0; l < vMP.size(); l++) {                                  // :5153-5160
        cv::Mat X(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) X.at<float>(k) = (float)points[3 * l + k];
        vMP[l]->SetWorldPos(X);
        vMP[l]->UpdateNormalAndDepth();
    }
    pMap->IncreaseChangeIndex();
}

}  // namespace ORB_SLAM3
