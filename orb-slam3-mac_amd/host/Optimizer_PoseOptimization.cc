// Optimizer_PoseOptimization.cc -- int Optimizer::PoseOptimization(Frame *pFrame) with the reference's signature (include/Optimizer.h:62,
// src/Optimizer.cc:854-1168) around the HIP solver.  Tracking calls it after every matcher (src/Tracking.cc:1775, 1934, 1996, 2002, 2727,
// 2743).  Host side, in the reference's order: one unary edge per keypoint that holds a map point (:897-1037) -- monocular
// (EdgeSE3ProjectXYZOnlyPose), stereo (EdgeStereoSE3ProjectXYZOnlyPose, mvuRight >= 0), or on frames of a two-camera rig left-camera /
// right-camera observations (EdgeSE3ProjectXYZOnlyPoseToBody with mTrl) -- with mvbOutlier reset; "fewer than 3 correspondences: return 0"
// (:1040-1041).  What was g2o (4 rounds of optimize(10) from the frame's pose, the chi2 re-classification after each round with the
// "an outlier's error is recomputed at the current estimate" rule, the Huber kernel dropped for the last round, the < 10 edges break,
// :1043-1149) runs in ONE device call (k_pose_opt).  Then the mvbOutlier write-back, SetPose and the return value
// nInitialCorrespondences - nBad (:1152-1160).
#include "Optimizer.h"
#include <cstdio>
#include <mutex>
#include <vector>
#include "optimizer_common.h"
#include "host_prof.h"

namespace ORB_SLAM3 {

using namespace optc;

int Optimizer::PoseOptimization(Frame *pFrame)
{
    hip::HostProf prof("PoseOptimization(Frame*)");
    int nInitialCorrespondences = 0;
    const int N = pFrame->N;
    std::vector<double> Xw, obs, invS2;
    std::vector<uint8_t> right;
    std::vector<int> vnIndexEdge;                              // frame index of every edge, in creation order (vnIndexEdgeMono / Right / Stereo merged)
    Xw.reserve((size_t)3 * N); obs.reserve((size_t)3 * N); invS2.reserve(N); right.reserve(N); vnIndexEdge.reserve(N);
    bool anyRight = false;
    {
    // :894-895: the map points' positions are snapshotted under MapPoint::mGlobalMutex, which MapPoint::SetWorldPos takes too: the
    // write-back of a local BA / loop closing running on another thread cannot move points half-way through this loop.  Released
    // before the device call, as the reference releases it before optimizer.optimize (:1038)
    std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);
    for (int i = 0; i < N; i++) {                              // :897-1037
        MapPoint *pMP = pFrame->mvpMapPoints[i];
        if (!pMP) continue;
        cv::KeyPoint kpUn;
        float kp_ur = -1.f;
        uint8_t isRight = 0;
        if (!pFrame->mpCamera2) {                              // Conventional SLAM
            kpUn = pFrame->mvKeysUn[i];
            if (!(pFrame->mvuRight[i] < 0)) kp_ur = pFrame->mvuRight[i];          // Stereo observation
        } else {                                               // SLAM with respect a rigid body
            if (i < pFrame->Nleft) kpUn = pFrame->mvKeys[i];   // Left camera observation
            else { kpUn = pFrame->mvKeysRight[i - pFrame->Nleft]; isRight = 1; anyRight = true; }
        }
        nInitialCorrespondences++;
        pFrame->mvbOutlier[i] = false;
        const cv::Mat P = pMP->GetWorldPos();
        for (int k = 0; k < 3; k++) Xw.push_back((double)P.at<float>(k));
        obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(kp_ur);
        invS2.push_back((double)pFrame->mvInvLevelSigma2[kpUn.octave]);
        right.push_back(isRight);
        vnIndexEdge.push_back(i);
    }
    }
    if (nInitialCorrespondences < 3) return 0;                 // :1040-1041

    double pose[7];
    toSE3Quat(pFrame->mTcw, pose);
    // monocular edges project through pFrame->mpCamera, stereo edges through the frame's fx, fy, cx, cy, mbf (:944-948): one calibration
    double fx, fy, cx, cy, kb[4]; int32_t model;
    camera_fields(pFrame->mpCamera, fx, fy, cx, cy, model, kb);
    orbhip_camera2 cam2;
    const bool rig = pFrame->mpCamera2 != nullptr;
    if (rig) {
        trl_to_se3quat(pFrame->mTrl, cam2.Trl);
        camera_fields(pFrame->mpCamera2, cam2.fx, cam2.fy, cam2.cx, cam2.cy, cam2.camera_model, cam2.kb);
    }
    const int n = nInitialCorrespondences;
    std::vector<uint8_t> outlier(n, 0);
    int32_t nInliers = 0, stats[4] = {0, 0, 0, 0};
    orbhip_ctx *ctx = thread_ctx();
    prof.mark();
    const int rc = ctx ? orbhip_pose_optimization_host(ctx, Xw.data(), obs.data(), invS2.data(), n, fx, fy, cx, cy, (double)pFrame->mbf, model ? kb : nullptr,
                                                       rig ? &cam2 : nullptr, (rig && anyRight) ? right.data() : nullptr, pose, outlier.data(), &nInliers, stats)
                       : ORBHIP_E_NODEVICE;
    prof.mark();
    if (rc != ORBHIP_OK) {
        // the reference has no failure path: leave the frame's pose and flags as they are and report no inliers (Tracking treats the frame as lost)
        fprintf(stderr, "PoseOptimization: HIP solver failed (%d: %s)\n", rc, orbhip_last_error());
        return 0;
    }
    for (int e = 0; e < n; e++) pFrame->mvbOutlier[vnIndexEdge[e]] = outlier[e] != 0;      // :1062-1143 (state after the last round)
    // Recover optimized pose and return number of inliers (:1152-1160)
    pFrame->SetPose(toCvMat(pose));
    return nInliers;                                           // nInitialCorrespondences - nBad
}

}  // namespace ORB_SLAM3
