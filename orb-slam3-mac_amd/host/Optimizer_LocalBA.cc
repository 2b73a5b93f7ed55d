// Optimizer_LocalBA.cc -- Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*, int&) with the reference's signature
// (include/Optimizer.h:58) around the HIP solver.  The host parts of the reference function stay host code, restated here in
// the reference's order (src/Optimizer.cc): window selection :1703-1761, fixed keyframes :1763-1819 incl. the ">= 2 fixed"
// rule, abort check :2041-2043, outlier list :2126-2173, ">= 50 % outliers" bail-out :2177-2181, locked erase + write-back
// :2204-2343.  What was the g2o block (:1828-2039 graph build, :2045-2122 optimize(5) / optimize(10)) is SoA packing in
// exactly the insertion order of :1850-2034 plus ONE call of orbhip_ba_solve_batch (LM + Schur on the device).
// Not carried over: the "Too much distance" statistics of :2261-2313 (Verbose prints at VERBOSITY_DEBUG only) and the
// unreachable bRedrawError file dump (:2183-2200, :2209-2251: behind a `return`).
#include "Optimizer.h"
#include "optimizer_common.h"
#include "host_prof.h"
#include <cstdio>
#include "hip_context.h"
#include <list>
#include <unordered_map>
#include <tuple>
#include <map>
#include <utility>
#include "../../include/orbhip.h"

namespace ORB_SLAM3 {

using namespace optc;

void Optimizer::LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, int &num_fixedKF)
{
    hip::HostProf prof("LocalBundleAdjustment(KeyFrame*)");
    // Local KeyFrames: First Breath Search from Current Keyframe (:1703-1717)
    std::list<KeyFrame *> lLocalKeyFrames;
    lLocalKeyFrames.push_back(pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    Map *pCurrentMap = pKF->GetMap();

    const std::vector<KeyFrame *> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
    for (int i = 0, iend = vNeighKFs.size(); i < iend; i++) {
        KeyFrame *pKFi = vNeighKFs[i];
        pKFi->mnBALocalForKF = pKF->mnId;
        if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lLocalKeyFrames.push_back(pKFi);
    }

    // Local MapPoints seen in Local KeyFrames (:1719-1761)
    num_fixedKF = 0;
    std::list<MapPoint *> lLocalMapPoints;
    for (std::list<KeyFrame *>::iterator lit = lLocalKeyFrames.begin(), lend = lLocalKeyFrames.end(); lit != lend; lit++) {
        KeyFrame *pKFi = *lit;
        if (pKFi->mnId == pMap->GetInitKFid()) num_fixedKF = 1;
        std::vector<MapPoint *> vpMPs = pKFi->GetMapPointMatches();
        for (std::vector<MapPoint *>::iterator vit = vpMPs.begin(), vend = vpMPs.end(); vit != vend; vit++) {
            MapPoint *pMP = *vit;
            if (pMP)
                if (!pMP->isBad() && pMP->GetMap() == pCurrentMap)
                    if (pMP->mnBALocalForKF != pKF->mnId) {
                        lLocalMapPoints.push_back(pMP);
                        pMP->mnBALocalForKF = pKF->mnId;
                    }
        }
    }

    // Fixed Keyframes. Keyframes that see Local MapPoints but that are not Local Keyframes (:1763-1780).  MapPoint::GetObservations
    // returns a COPY of the point's std::map (under its mutex): the reference takes one here and a second one when it builds the
    // edges (:1929); this shim keeps the first snapshot, flattened, and builds the edges from it -- 2000 map copies instead of 4000
    std::list<KeyFrame *> lFixedCameras;
    std::vector<std::pair<KeyFrame *, std::tuple<int, int>>> obsFlat;
    std::vector<int> obsStart;
    obsFlat.reserve(lLocalMapPoints.size() * 8); obsStart.reserve(lLocalMapPoints.size() + 1);
    for (std::list<MapPoint *>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++) {
        std::map<KeyFrame *, std::tuple<int, int>> observations = (*lit)->GetObservations();
        obsStart.push_back((int)obsFlat.size());
        for (std::map<KeyFrame *, std::tuple<int, int>>::iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
            KeyFrame *pKFi = mit->first;
            obsFlat.push_back(*mit);
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) lFixedCameras.push_back(pKFi);
            }
        }
    }
    obsStart.push_back((int)obsFlat.size());
    num_fixedKF = lFixedCameras.size() + num_fixedKF;
    if (num_fixedKF < 2) {
        // "We set 2 KFs to fixed to avoid a degree of freedom in scale" (:1782-1817).  The reference reads pLowerKf / pSecondLowerKF
        // uninitialised when the window has no candidate; here a missing candidate is simply not fixed.
        std::list<KeyFrame *>::iterator lit = lLocalKeyFrames.begin();
        int lowerId = pKF->mnId;
        KeyFrame *pLowerKf = nullptr;
        int secondLowerId = pKF->mnId;
        KeyFrame *pSecondLowerKF = nullptr;
        for (; lit != lLocalKeyFrames.end(); lit++) {
            KeyFrame *pKFi = *lit;
            if (pKFi == pKF || pKFi->mnId == pMap->GetInitKFid()) continue;
            if ((int)pKFi->mnId < lowerId) { lowerId = pKFi->mnId; pLowerKf = pKFi; }
            else if ((int)pKFi->mnId < secondLowerId) { secondLowerId = pKFi->mnId; pSecondLowerKF = pKFi; }
        }
        if (pLowerKf) { lFixedCameras.push_back(pLowerKf); lLocalKeyFrames.remove(pLowerKf); num_fixedKF++; }
        if (num_fixedKF < 2 && pSecondLowerKF) { lFixedCameras.push_back(pSecondLowerKF); lLocalKeyFrames.remove(pSecondLowerKF); num_fixedKF++; }
    }

    // ---- SoA packing in the vertex / edge insertion order of :1850-2034 (was: the g2o graph) -------------------------------
    std::unordered_map<KeyFrame *, int> kfIndex;
    std::vector<KeyFrame *> vpKFs;
    std::vector<uint8_t> fixed;
    for (KeyFrame *pKFi : lLocalKeyFrames) { kfIndex[pKFi] = vpKFs.size(); vpKFs.push_back(pKFi); fixed.push_back(pKFi->mnId == pMap->GetInitKFid()); }   // :1850-1860
    for (KeyFrame *pKFi : lFixedCameras) { kfIndex[pKFi] = vpKFs.size(); vpKFs.push_back(pKFi); fixed.push_back(1); }                                         // :1864-1874
    const int nKF = vpKFs.size();
    std::vector<double> poses((size_t)7 * nKF);
    for (int i = 0; i < nKF; i++) toSE3Quat(vpKFs[i]->GetPose(), &poses[(size_t)7 * i]);

    std::vector<MapPoint *> vpMPs(lLocalMapPoints.begin(), lLocalMapPoints.end());
    const int nMP = vpMPs.size();
    std::vector<double> points((size_t)3 * nMP);
    std::vector<int32_t> ePose, ePoint; std::vector<double> obs, invS2; std::vector<uint8_t> eType;
    std::vector<KeyFrame *> vpEdgeKF; std::vector<MapPoint *> vpEdgeMP;
    ePose.reserve(obsFlat.size()); ePoint.reserve(obsFlat.size()); obs.reserve(3 * obsFlat.size()); invS2.reserve(obsFlat.size()); eType.reserve(obsFlat.size());
    vpEdgeKF.reserve(obsFlat.size()); vpEdgeMP.reserve(obsFlat.size());
    size_t nMonoEdges = 0, nStereoEdges = 0;
    KeyFrame *pRigKF = nullptr;
    for (int l = 0; l < nMP; l++) {
        MapPoint *pMP = vpMPs[l];
        const cv::Mat Xw = pMP->GetWorldPos();                                        // Converter::toVector3d, :1923
        for (int k = 0; k < 3; k++) points[(size_t)3 * l + k] = (double)Xw.at<float>(k);
        for (const std::pair<KeyFrame *, std::tuple<int, int>> *mit = obsFlat.data() + obsStart[l], *mend = obsFlat.data() + obsStart[l + 1]; mit != mend; mit++) {     // (:1929: in std::map order)
            KeyFrame *pKFi = mit->first;
            if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;             // :1937
            std::unordered_map<KeyFrame *, int>::iterator ki = kfIndex.find(pKFi);
            if (ki == kfIndex.end()) continue;                                        // (g2o: optimizer.vertex(id) == NULL -> addEdge refuses the edge)
            const int leftIndex = std::get<0>(mit->second);
            if (leftIndex != -1 && pKFi->mvuRight[leftIndex] < 0) {                   // Monocular observation (:1942-1968)
                const cv::KeyPoint &kpUn = pKFi->mvKeysUn[leftIndex];
                ePose.push_back(ki->second); ePoint.push_back(l); eType.push_back(0);
                obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(0.0);
                invS2.push_back((double)pKFi->mvInvLevelSigma2[kpUn.octave]);
                vpEdgeKF.push_back(pKFi); vpEdgeMP.push_back(pMP); nMonoEdges++;
            } else if (leftIndex != -1 && pKFi->mvuRight[leftIndex] >= 0) {           // Stereo observation (:1970-2000)
                const cv::KeyPoint &kpUn = pKFi->mvKeysUn[leftIndex];
                const float kp_ur = pKFi->mvuRight[leftIndex];
                ePose.push_back(ki->second); ePoint.push_back(l); eType.push_back(1);
                obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(kp_ur);
                invS2.push_back((double)pKFi->mvInvLevelSigma2[kpUn.octave]);
                vpEdgeKF.push_back(pKFi); vpEdgeMP.push_back(pMP); nStereoEdges++;
            }
            if (pKFi->mpCamera2) {                                                    // observation in the second camera (:2002-2032)
                int rightIndex = std::get<1>(mit->second);
                if (rightIndex != -1) {
                    rightIndex -= pKFi->NLeft;
                    const cv::KeyPoint kp = pKFi->mvKeysRight[rightIndex];
                    ePose.push_back(ki->second); ePoint.push_back(l); eType.push_back(2);
                    obs.push_back(kp.pt.x); obs.push_back(kp.pt.y); obs.push_back(0.0);
                    invS2.push_back((double)pKFi->mvInvLevelSigma2[kp.octave]);
                    vpEdgeKF.push_back(pKFi); vpEdgeMP.push_back(pMP);
                    pRigKF = pKFi;
                }
            }
        }
    }
    const int nE = ePose.size();

    if (pbStopFlag)
        if (*pbStopFlag) return;                                                      // :2041-2043

    orbhip_ba_graph g;
    memset(&g, 0, sizeof(g));
    g.n_poses = nKF; g.n_points = nMP; g.n_edges = nE;
    g.pose_fixed = fixed.data(); g.edge_pose = ePose.data(); g.edge_point = ePoint.data(); g.edge_obs = obs.data();
    g.edge_inv_sigma2 = invS2.data(); g.edge_stereo = eType.data();
    // the reference hands each edge its keyframe's calibration (e->pCamera = pKFi->mpCamera :1961, e->fx = pKFi->fx ... e->bf = pKFi->mbf
    // :1990-1994, mTrl / mpCamera2 :2021-2023): one camera table entry per distinct calibration of the window, an index per keyframe
    std::vector<orbhip_ba_camera> cams;
    std::vector<int32_t> poseCam;
    camera_table(vpKFs, cams, poseCam);
    if (cams.size() > 1) { g.n_cameras = (int32_t)cams.size(); g.cameras = cams.data(); g.pose_camera = poseCam.data(); }
    // (a single calibration -- every configuration the reference ships -- travels in the graph's own fields)
    camera_fields(pKF->mpCamera, g.fx, g.fy, g.cx, g.cy, g.camera_model, g.kb);
    g.bf = pKF->mbf;
    g.Trl[3] = 1.0;
    if (pRigKF) {
        trl_to_se3quat(pRigKF->mTrl, g.Trl);
        camera_fields(pRigKF->mpCamera2, g.fx2, g.fy2, g.cx2, g.cy2, g.camera2_model, g.kb2);
    }
    orbhip_ba_params p;
    orbhip_ba_default_params(&p);                                                     // optimize(5) + optimize(10), Huber sqrt(5.991) / sqrt(7.815)
    if (pMap->IsInertial()) p.user_lambda_init = 100.0;                               // :1837-1838
    p.no_discard = 1;                                                                 // the bail-out is decided below, with the reference's own count

    std::vector<uint8_t> outlier(nE ? nE : 1, 0);
    orbhip_ba_stats st;
    memset(&st, 0, sizeof(st));
    prof.mark();
    if (nE > 0 && nKF > 0 && nMP > 0) {
        orbhip_ctx *ctx = thread_ctx();
        double *pp = poses.data(), *px = points.data();
        uint8_t *po = outlier.data();
        const int rc = ctx ? orbhip_ba_solve_batch(ctx, &g, 1, &p, (volatile const uint8_t *)pbStopFlag, &pp, &px, &po, &st) : ORBHIP_E_NODEVICE;
        if (rc == ORBHIP_E_ABORTED) return;                                           // stop flag raised before the first iteration (:2041-2043)
        if (rc != ORBHIP_OK) {
            // the reference has no failure path here (g2o rejects a step it cannot solve and carries on): leave the map untouched
            fprintf(stderr, "LM-LBA: HIP solver failed (%d: %s), map left unchanged\n", rc, orbhip_last_error());
            return;
        }
    }

    prof.mark();
    // Check inlier observations (:2122-2173): edges whose chi2 exceeds the gate or whose depth is not positive
    std::vector<std::pair<KeyFrame *, MapPoint *>> vToErase;
    vToErase.reserve(nE);
    for (int pass = 0; pass < 3; pass++) {                                            // vpEdgesMono, then vpEdgesBody, then vpEdgesStereo
        const uint8_t type = pass == 0 ? 0 : pass == 1 ? 2 : 1;
        for (int e = 0; e < nE; e++) {
            if (eType[e] != type) continue;
            MapPoint *pMP = vpEdgeMP[e];
            if (pMP->isBad()) continue;
            if (outlier[e]) vToErase.push_back(std::make_pair(vpEdgeKF[e], pMP));
        }
    }
    if (vToErase.size() >= (nMonoEdges + nStereoEdges) * 0.5) {
        fprintf(stderr, "LM-LBA: ERROR IN THE OPTIMIZATION, MOST OF THE POINTS HAS BECOME OUTLIERS\n");   // :2177-2181
        return;
    }

    // Get Map Mutex (:2204)
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);

    if (!vToErase.empty()) {
        for (size_t i = 0; i < vToErase.size(); i++) {                                // :2220-2226
            KeyFrame *pKFi = vToErase[i].first;
            MapPoint *pMPi = vToErase[i].second;
            pKFi->EraseMapPointMatch(pMPi);
            pMPi->EraseObservation(pKFi);
        }
    }

    // Recover optimized data (:2254-2330)
    for (std::list<KeyFrame *>::iterator lit = lLocalKeyFrames.begin(), lend = lLocalKeyFrames.end(); lit != lend; lit++) {
        KeyFrame *pKFi = *lit;
        pKFi->SetPose(toCvMat(&poses[(size_t)7 * kfIndex[pKFi]]));
    }
    cv::Mat X(3, 1, CV_32F);                                                          // (SetWorldPos copies: one buffer serves every point)
    for (int l = 0; l < nMP; l++) {
        MapPoint *pMP = vpMPs[l];
        for (int k = 0; k < 3; k++) X.at<float>(k) = (float)points[(size_t)3 * l + k];
        pMP->SetWorldPos(X);
        pMP->UpdateNormalAndDepth();
    }

    pMap->IncreaseChangeIndex();                                                      // :2343
}

}  // namespace ORB_SLAM3
