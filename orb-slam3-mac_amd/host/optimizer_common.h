// optimizer_common.h -- helpers shared by the Optimizer shims (Optimizer_LocalBA.cc, Optimizer_MergeBA.cc, Optimizer_PoseOptimization.cc,
// Optimizer_LocalInertialBA.cc): the Converter round trips between CV_32F poses and g2o::SE3Quat, camera parameter packing.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "slam_types.h"
#include "hip_context.h"
#include "../../include/orbhip.h"

namespace ORB_SLAM3 {
namespace optc {

// Converter::toSE3Quat (src/Converter.cc:34-44): float 4x4 -> double R, t -> g2o::SE3Quat(R, t), whose constructor
// (Thirdparty/g2o/g2o/types/se3quat.h:58-60) builds Eigen::Quaterniond(R) and normalizeRotation() (:280-285: w >= 0, unit norm).
inline void toSE3Quat(const cv::Mat &cvT, double *q7)
{
    double R[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = (double)cvT.at<float>(i, j);
    double q[4];                                        // x, y, z, w
    const double tr = R[0][0] + R[1][1] + R[2][2];
    if (tr > 0) {
        double s = std::sqrt(tr + 1.0);
        q[3] = 0.5 * s; s = 0.5 / s;
        q[0] = (R[2][1] - R[1][2]) * s; q[1] = (R[0][2] - R[2][0]) * s; q[2] = (R[1][0] - R[0][1]) * s;
    } else {
        int i = 0;
        if (R[1][1] > R[0][0]) i = 1;
        if (R[2][2] > R[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
        q[i] = 0.5 * s; s = 0.5 / s;
        q[3] = (R[k][j] - R[j][k]) * s; q[j] = (R[j][i] + R[i][j]) * s; q[k] = (R[k][i] + R[i][k]) * s;
    }
    if (q[3] < 0) for (double &v : q) v = -v;
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q7[i] = q[i] / n;
    for (int i = 0; i < 3; i++) q7[4 + i] = (double)cvT.at<float>(i, 3);
}

// Converter::toCvMat(g2o::SE3Quat) (src/Converter.cc:46-50, 60-68): to_homogeneous_matrix (Quaterniond::toRotationMatrix) -> CV_32F
inline cv::Mat toCvMat(const double *q7)
{
    const double x = q7[0], y = q7[1], z = q7[2], w = q7[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    const double R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
    cv::Mat m = cv::Mat::eye(4, 4, CV_32F);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) m.at<float>(i, j) = (float)R[i][j]; m.at<float>(i, 3) = (float)q7[4 + i]; }
    return m;
}

inline orbhip_ctx *thread_ctx() { return hip::ThreadContext(); }      // one context per calling thread, GPU of hip::GetDevice() (hip_context.h)

inline void camera_fields(GeometricCamera *cam, double &fx, double &fy, double &cx, double &cy, int32_t &model, double (&kb)[4])
{
    fx = cam->getParameter(0); fy = cam->getParameter(1); cx = cam->getParameter(2); cy = cam->getParameter(3);
    model = cam->GetType() == cam->CAM_FISHEYE ? 1 : 0;
    for (int i = 0; i < 4; i++) kb[i] = model ? (double)cam->getParameter(4 + i) : 0.0;
}


// mTrl (3x4 CV_32F) as g2o::SE3Quat: Converter::toSE3Quat reads rows 0..2 of its argument
inline void trl_to_se3quat(const cv::Mat &Trl, double *q7)
{
    cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) T.at<float>(i, j) = Trl.at<float>(i, j);
    toSE3Quat(T, q7);
}

// The reference hands every edge its own keyframe's calibration (Optimizer.cc:1961, 1990-1994, 2021-2023).  All keyframes of a map share
// one in every configuration the reference ships, but an Atlas window may mix cameras: the shims group the window's keyframes by
// calibration (same_calibration) and, when there is more than one group, pass the graph a camera table with an index per keyframe
// (orbhip_ba_graph::cameras / pose_camera, round 4; before that such a window was refused).
inline bool same_calibration(KeyFrame *a, KeyFrame *b)
{
    if (a->fx != b->fx || a->fy != b->fy || a->cx != b->cx || a->cy != b->cy || a->mbf != b->mbf) return false;
    if ((a->mpCamera2 == nullptr) != (b->mpCamera2 == nullptr)) return false;
    GeometricCamera *ca[2] = {a->mpCamera, a->mpCamera2}, *cb[2] = {b->mpCamera, b->mpCamera2};
    for (int k = 0; k < 2; k++) {
        if (!ca[k]) continue;
        if (ca[k] == cb[k]) continue;
        if (ca[k]->GetType() != cb[k]->GetType() || ca[k]->size() != cb[k]->size()) return false;
        for (size_t i = 0; i < ca[k]->size(); i++) if (ca[k]->getParameter(i) != cb[k]->getParameter(i)) return false;
    }
    if (a->mpCamera2)
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) if (a->mTrl.at<float>(i, j) != b->mTrl.at<float>(i, j)) return false;
    return true;
}


// one orbhip_ba_camera per distinct calibration among vpKFs, poseCam[i] = the entry keyframe i uses.  Monocular edges project through
// pKFi->mpCamera (getParameter), stereo edges through the keyframe's fx, fy, cx, cy, mbf members: the same numbers in every configuration
// of the reference (KeyFrame copies them from the same calibration, KeyFrame.cc:40-45); second-camera edges through mTrl / mpCamera2
inline void camera_table(const std::vector<KeyFrame *> &vpKFs, std::vector<orbhip_ba_camera> &cams, std::vector<int32_t> &poseCam)
{
    std::vector<KeyFrame *> rep;
    cams.clear(); poseCam.assign(vpKFs.size(), 0);
    for (size_t i = 0; i < vpKFs.size(); i++) {
        KeyFrame *kf = vpKFs[i];
        size_t c = 0;
        for (; c < rep.size(); c++) if (same_calibration(rep[c], kf)) break;
        if (c == rep.size()) {
            rep.push_back(kf);
            orbhip_ba_camera K;
            memset(&K, 0, sizeof(K));
            camera_fields(kf->mpCamera, K.fx, K.fy, K.cx, K.cy, K.camera_model, K.kb);
            K.bf = kf->mbf;
            K.Trl[3] = 1.0;
            if (kf->mpCamera2) {
                trl_to_se3quat(kf->mTrl, K.Trl);
                camera_fields(kf->mpCamera2, K.fx2, K.fy2, K.cx2, K.cy2, K.camera2_model, K.kb2);
            }
            cams.push_back(K);
        }
        poseCam[i] = (int32_t)c;
    }
}

}  // namespace optc
}  // namespace ORB_SLAM3
