// Optimizer_MergeBA.cc -- void Optimizer::LocalBundleAdjustment(KeyFrame *pMainKF, vector<KeyFrame*> vpAdjustKF, vector<KeyFrame*> vpFixedKF,
// bool *pbStopFlag) with the reference's signature (include/Optimizer.h:104, src/Optimizer.cc:6255-6911): the local BA of the map-merge
// welding window (LoopClosing::MergeLocal).  Host side restated in the reference's order: fixed keyframes and their map points
// (:6284-6331), adjustable keyframes and theirs (:6335-6369), one edge per observation that passes the filter of :6424, abort check
// (:6518-6520), erase list (:6592-6632), locked erase (:6635-6676), write-back of the adjustable keyframes and of every map point
// (:6713-6910).  What was g2o (optimize(5), the level-1 / no-robust-kernel second stage :6537-6588, optimize(10)) is ONE call of
// orbhip_ba_solve_batch under orbhip_ba_merge_params.
// Kept as the reference has them: an observation's keyframe must carry mnBALocalForMerge == pMainKF->mnId (:6424) -- which THIS function
// sets on the fixed keyframes only (:6292; the adjustable ones get mnBALocalForKF, :6342), so unless the caller marked them the window's
// edges all go to fixed keyframes and the adjustable poses come back unchanged (through the Converter round trip); vpMPs keeps the
// iteration order of the std::set<MapPoint*> of KeyFrame::GetMapPoints.  Not carried over: the Verbose prints, the statistics maps that
// only feed them, the bShowImages image dump (:6786-6880, behind a constant false).
#include "Optimizer.h"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <utility>
#include <vector>
#include "optimizer_common.h"

namespace ORB_SLAM3 {

using namespace optc;

void Optimizer::LocalBundleAdjustment(KeyFrame *pMainKF, std::vector<KeyFrame *> vpAdjustKF, std::vector<KeyFrame *> vpFixedKF, bool *pbStopFlag)
{
    std::vector<MapPoint *> vpMPs;
    long unsigned int maxKFid = 0;
    Map *pCurrentMap = pMainKF->GetMap();
    std::map<KeyFrame *, int> kfIndex;                            // optimizer.vertex(pKF->mnId) != NULL
    std::vector<KeyFrame *> vpKFs;
    std::vector<uint8_t> fixed;

    // Set fixed KeyFrame vertices (:6284-6331)
    for (KeyFrame *pKFi : vpFixedKF) {
        if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;
        pKFi->mnBALocalForMerge = pMainKF->mnId;
        if (!kfIndex.count(pKFi)) { kfIndex[pKFi] = vpKFs.size(); vpKFs.push_back(pKFi); fixed.push_back(1); }
        if (pKFi->mnId > maxKFid) maxKFid = pKFi->mnId;
        std::set<MapPoint *> spViewMPs = pKFi->GetMapPoints();
        for (MapPoint *pMPi : spViewMPs)
            if (pMPi)
                if (!pMPi->isBad() && pMPi->GetMap() == pCurrentMap)
                    if (pMPi->mnBALocalForMerge != pMainKF->mnId) { vpMPs.push_back(pMPi); pMPi->mnBALocalForMerge = pMainKF->mnId; }
    }
    // Set non fixed Keyframe vertices (:6335-6369)
    for (KeyFrame *pKFi : vpAdjustKF) {
        if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;
        pKFi->mnBALocalForKF = pMainKF->mnId;
        if (!kfIndex.count(pKFi)) { kfIndex[pKFi] = vpKFs.size(); vpKFs.push_back(pKFi); fixed.push_back(0); }   // (g2o refuses a second vertex with the same id)
        if (pKFi->mnId > maxKFid) maxKFid = pKFi->mnId;
        std::set<MapPoint *> spViewMPs = pKFi->GetMapPoints();
        for (MapPoint *pMPi : spViewMPs)
            if (pMPi)
                if (!pMPi->isBad() && pMPi->GetMap() == pCurrentMap)
                    if (pMPi->mnBALocalForMerge != pMainKF->mnId) { vpMPs.push_back(pMPi); pMPi->mnBALocalForMerge = pMainKF->mnId; }
    }
    const int nKF = vpKFs.size();
    std::vector<double> poses((size_t)7 * (nKF ? nKF : 1));
    for (int i = 0; i < nKF; i++) toSE3Quat(vpKFs[i]->GetPose(), &poses[(size_t)7 * i]);

    // Set MapPoint vertices and their edges (:6402-6497).  A map point without a single edge is an inactive vertex in g2o: it stays out of
    // the packed graph and keeps its position
    std::vector<MapPoint *> vpGraphMP;
    std::vector<double> points;
    std::vector<int32_t> ePose, ePoint; std::vector<double> obs, invS2; std::vector<uint8_t> eType;
    std::vector<KeyFrame *> vpEdgeKF; std::vector<MapPoint *> vpEdgeMP;
    for (unsigned int i = 0; i < vpMPs.size(); ++i) {
        MapPoint *pMPi = vpMPs[i];
        if (pMPi->isBad()) continue;
        const std::map<KeyFrame *, std::tuple<int, int>> observations = pMPi->GetObservations();
        const size_t e0 = ePose.size();
        const int l = vpGraphMP.size();
        for (std::map<KeyFrame *, std::tuple<int, int>>::const_iterator mit = observations.begin(); mit != observations.end(); mit++) {
            KeyFrame *pKF = mit->first;
            if (pKF->isBad() || pKF->mnId > maxKFid || pKF->mnBALocalForMerge != pMainKF->mnId || !pKF->GetMapPoint(std::get<0>(mit->second))) continue;   // :6424
            std::map<KeyFrame *, int>::iterator ki = kfIndex.find(pKF);
            if (ki == kfIndex.end()) continue;                    // optimizer.vertex(pKF->mnId) == NULL: addEdge refuses the edge
            const int idx = std::get<0>(mit->second);
            const cv::KeyPoint &kpUn = pKF->mvKeysUn[idx];
            ePose.push_back(ki->second); ePoint.push_back(l);
            if (pKF->mvuRight[idx] < 0) {                          // Monocular (:6434-6463)
                obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(0.0); eType.push_back(0);
            } else {                                              // RGBD or Stereo (:6465-6495)
                obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(pKF->mvuRight[idx]); eType.push_back(1);
            }
            invS2.push_back((double)pKF->mvInvLevelSigma2[kpUn.octave]);
            vpEdgeKF.push_back(pKF); vpEdgeMP.push_back(pMPi);
        }
        if (ePose.size() == e0) continue;
        vpGraphMP.push_back(pMPi);
        const cv::Mat Xw = pMPi->GetWorldPos();
        for (int k = 0; k < 3; k++) points.push_back((double)Xw.at<float>(k));
    }
    const int nE = ePose.size(), nMP = vpGraphMP.size();

    if (pbStopFlag)
        if (*pbStopFlag) return;                                  // :6518-6520

    std::vector<uint8_t> outlier(nE ? nE : 1, 0);
    if (nE > 0) {
        orbhip_ba_graph g;
        memset(&g, 0, sizeof(g));
        g.n_poses = nKF; g.n_points = nMP; g.n_edges = nE;
        g.pose_fixed = fixed.data(); g.edge_pose = ePose.data(); g.edge_point = ePoint.data(); g.edge_obs = obs.data();
        g.edge_inv_sigma2 = invS2.data(); g.edge_stereo = eType.data();
        camera_fields(vpKFs[0]->mpCamera, g.fx, g.fy, g.cx, g.cy, g.camera_model, g.kb);
        g.bf = vpKFs[0]->mbf;
        g.Trl[3] = 1.0;
        // every edge projects through its own keyframe's camera (src/Optimizer.cc:6423, :6452-6456): a table when the window mixes calibrations
        std::vector<orbhip_ba_camera> cams;
        std::vector<int32_t> poseCam;
        camera_table(vpKFs, cams, poseCam);
        if (cams.size() > 1) { g.n_cameras = (int32_t)cams.size(); g.cameras = cams.data(); g.pose_camera = poseCam.data(); }
        orbhip_ba_params p;
        orbhip_ba_merge_params(&p);                               // Huber sqrt(5.99) / sqrt(7.815), gates 5.991 / 7.815, first-pass outliers at level 1, no robust kernel in pass 2
        orbhip_ba_stats st;
        memset(&st, 0, sizeof(st));
        orbhip_ctx *ctx = thread_ctx();
        double *pp = poses.data(), *px = points.data();
        uint8_t *po = outlier.data();
        const int rc = ctx ? orbhip_ba_solve_batch(ctx, &g, 1, &p, (volatile const uint8_t *)pbStopFlag, &pp, &px, &po, &st) : ORBHIP_E_NODEVICE;
        if (getenv("ORBHIP_SHIM_DEBUG"))
            fprintf(stderr, "LBA (merge): %d keyframes, %d points, %d edges: iterations %d + %d, %d LM trials, %d outliers, chi2 %.9g -> %.9g\n", nKF, nMP, nE,
                    st.iterations_run[0], st.iterations_run[1], st.lm_trials, st.n_outliers, st.chi2_initial, st.chi2_final);
        if (rc == ORBHIP_E_ABORTED) return;
        if (rc != ORBHIP_OK) {
            fprintf(stderr, "LBA (merge): HIP solver failed (%d: %s), map left unchanged\n", rc, orbhip_last_error());
            return;
        }
    }

    // Check inlier observations (:6590-6632): monocular edges first, then stereo
    std::vector<std::pair<KeyFrame *, MapPoint *>> vToErase;
    vToErase.reserve(nE);
    for (int pass = 0; pass < 2; pass++)
        for (int e = 0; e < nE; e++) {
            if (eType[e] != pass) continue;
            MapPoint *pMP = vpEdgeMP[e];
            if (pMP->isBad()) continue;
            if (outlier[e]) vToErase.push_back(std::make_pair(vpEdgeKF[e], pMP));
        }

    // Get Map Mutex (:6635)
    std::unique_lock<std::mutex> lock(pMainKF->GetMap()->mMutexMapUpdate);
    if (!vToErase.empty())
        for (size_t i = 0; i < vToErase.size(); i++) {            // :6647-6653
            KeyFrame *pKFi = vToErase[i].first;
            MapPoint *pMPi = vToErase[i].second;
            pKFi->EraseMapPointMatch(pMPi);
            pMPi->EraseObservation(pKFi);
        }

    // Recover optimized data: Keyframes (:6713-6886)
    for (KeyFrame *pKFi : vpAdjustKF) {
        if (pKFi->isBad()) continue;
        std::map<KeyFrame *, int>::iterator ki = kfIndex.find(pKFi);
        if (ki == kfIndex.end()) continue;                        // (not in the current map: no vertex; the reference would dereference NULL here)
        pKFi->SetPose(toCvMat(&poses[(size_t)7 * ki->second]));
    }
    // Points (:6889-6899): every map point of the window, optimised or not
    std::map<MapPoint *, int> graphIndex;
    for (int l = 0; l < nMP; l++) graphIndex[vpGraphMP[l]] = l;
    for (MapPoint *pMPi : vpMPs) {
        if (pMPi->isBad()) continue;
        std::map<MapPoint *, int>::iterator gi = graphIndex.find(pMPi);
        if (gi != graphIndex.end()) {
            cv::Mat X(3, 1, CV_32F);
            for (int k = 0; k < 3; k++) X.at<float>(k) = (float)points[(size_t)3 * gi->second + k];
            pMPi->SetWorldPos(X);
        } else
            pMPi->SetWorldPos(pMPi->GetWorldPos());               // an inactive vertex keeps its estimate: toCvMat(toVector3d(x)) == x
        pMPi->UpdateNormalAndDepth();
    }
}

}  // namespace ORB_SLAM3
