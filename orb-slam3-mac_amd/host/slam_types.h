// slam_types.h -- minimal stand-ins for the reference's Frame / KeyFrame / MapPoint / Map / GeometricCamera, exposing ONLY the
// members and accessors that the hot-path callers' code touches (reference include/Frame.h, KeyFrame.h, MapPoint.h, Map.h,
// CameraModels/GeometricCamera.h -- same names, same types, same meaning), so that host/Optimizer_LocalBA.cc and
// host/ORBmatcher.cc compile and run in this image (no OpenCV, no Eigen, no reference build).  In a real integration define
// ORBHIP_WITH_ORBSLAM3 and the reference's own headers are included instead; nothing in the shims depends on anything that is
// not in the reference's classes.  These are plain containers: the pointer graph, not its maintenance (covisibility updates,
// culling ...), which stays the caller's.
#pragma once
#ifdef ORBHIP_WITH_ORBSLAM3
#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
#include "Map.h"
#include "CameraModels/GeometricCamera.h"
#else
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <vector>
#include "cvlite.h"

namespace ORB_SLAM3 {

class KeyFrame;
class Map;
}  // namespace ORB_SLAM3

// Thirdparty/DBoW2/DBoW2/FeatureVector.h:25-56: vocabulary node -> indices of the features below it
namespace DBoW2 {
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {
public:
    void addFeature(NodeId id, unsigned int i_feature) { (*this)[id].push_back(i_feature); }
};
}  // namespace DBoW2

namespace ORB_SLAM3 {

// include/ImuTypes.h (the members Optimizer.cc:4574-5187 and G2oTypes.cc:25-71, 693-715 read)
namespace IMU {
class Bias {
public:
    Bias() : bax(0), bay(0), baz(0), bwx(0), bwy(0), bwz(0) {}
    Bias(const float &b_acc_x, const float &b_acc_y, const float &b_acc_z, const float &b_ang_vel_x, const float &b_ang_vel_y, const float &b_ang_vel_z)
        : bax(b_acc_x), bay(b_acc_y), baz(b_acc_z), bwx(b_ang_vel_x), bwy(b_ang_vel_y), bwz(b_ang_vel_z) {}
    float bax, bay, baz, bwx, bwy, bwz;
};
class Calib {
public:
    cv::Mat Tcb, Tbc;
};
class Preintegrated {
public:
    Preintegrated() : dT(0) {}
    void SetNewBias(const Bias &bu_) { bu = bu_; }       // ImuTypes.cc:334-349 (db = bu - b is recomputed by the consumers here)
    float dT;
    cv::Mat C;                                           // 15 x 15 covariance
    Bias b;                                              // the bias the measurements were integrated with
    cv::Mat dR, dV, dP, JRg, JVg, JVa, JPg, JPa;
    Bias bu;
};
}  // namespace IMU

// include/CameraModels/GeometricCamera.h:36-104 (type tag + parameter vector; project(cv::Mat) as Pinhole.cpp:34-39 /
// KannalaBrandt8.cpp:52-69 compute it, in float)
class GeometricCamera {
public:
    GeometricCamera(const std::vector<float> &p, unsigned int type) : mvParameters(p), mnType(type) {}
    float getParameter(const int i) { return mvParameters[i]; }
    size_t size() { return mvParameters.size(); }
    unsigned int GetType() { return mnType; }
    const unsigned int CAM_PINHOLE = 0;
    const unsigned int CAM_FISHEYE = 1;
    cv::Point2f project(const cv::Mat &m3D)
    {
        const float *p = m3D.ptr<float>();
        const float x = m3D.cols == 1 ? m3D.at<float>(0) : p[0], y = m3D.cols == 1 ? m3D.at<float>(1) : p[1], z = m3D.cols == 1 ? m3D.at<float>(2) : p[2];
        if (mnType == 0) return cv::Point2f(mvParameters[0] * x / z + mvParameters[2], mvParameters[1] * y / z + mvParameters[3]);
        const float x2_plus_y2 = x * x + y * y;
        const float theta = atan2f(sqrtf(x2_plus_y2), z), psi = atan2f(y, x);
        const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
        const float r = theta + mvParameters[4] * theta3 + mvParameters[5] * theta5 + mvParameters[6] * theta7 + mvParameters[7] * theta9;
        return cv::Point2f(mvParameters[0] * r * cosf(psi) + mvParameters[2], mvParameters[1] * r * sinf(psi) + mvParameters[3]);
    }
protected:
    std::vector<float> mvParameters;
    unsigned int mnType;
};

// include/MapPoint.h (the members Optimizer.cc:1699-2344 and ORBmatcher.cc:48-218, 1965-2181 read or write)
class MapPoint {
public:
    MapPoint(long unsigned int id, const cv::Mat &Pos, Map *pMap) : mnId(id), mnBALocalForKF(0), mTrackProjX(0), mTrackProjY(0),
        mTrackDepth(0), mTrackDepthR(0), mTrackProjXR(0), mTrackProjYR(0), mbTrackInView(false), mbTrackInViewR(false),
        mnTrackScaleLevel(0), mnTrackScaleLevelR(-1), mTrackViewCos(1), mTrackViewCosR(1), mWorldPos(Pos.clone()), mpMap(pMap),
        mbBad(false), nObs(0), nNormalUpdates(0) {}
    void SetWorldPos(const cv::Mat &Pos) { mWorldPos = Pos.clone(); }
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    std::map<KeyFrame *, std::tuple<int, int>> GetObservations() { return mObservations; }
    int Observations() { return nObs; }
    void AddObservation(KeyFrame *pKF, int idxLeft, int idxRight = -1) { mObservations[pKF] = std::make_tuple(idxLeft, idxRight); nObs += (idxLeft != -1) + (idxRight != -1); }
    void EraseObservation(KeyFrame *pKF)
    {
        auto it = mObservations.find(pKF);
        if (it == mObservations.end()) return;
        nObs -= (std::get<0>(it->second) != -1) + (std::get<1>(it->second) != -1);
        mObservations.erase(it);
        if (nObs <= 2) mbBad = true;                    // MapPoint.cc:199-201 (SetBadFlag)
    }
    bool isBad() { return mbBad; }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    void UpdateNormalAndDepth() { nNormalUpdates++; }
    Map *GetMap() { return mpMap; }
    long unsigned int mnId;
    long unsigned int mnBALocalForKF;
    // Tracking's per-frame projection record (Frame::isInFrustum fills it; ORBmatcher.cc:57-79 reads it)
    float mTrackProjX, mTrackProjY, mTrackDepth, mTrackDepthR, mTrackProjXR, mTrackProjYR;
    bool mbTrackInView, mbTrackInViewR;
    int mnTrackScaleLevel, mnTrackScaleLevelR;
    float mTrackViewCos, mTrackViewCosR;
    cv::Mat mDescriptor;
    // stand-in state
    cv::Mat mWorldPos;
    std::map<KeyFrame *, std::tuple<int, int>> mObservations;
    Map *mpMap;
    bool mbBad;
    int nObs, nNormalUpdates;
};

// include/KeyFrame.h
class KeyFrame {
public:
    KeyFrame(long unsigned int id, Map *pMap, float fx_, float fy_, float cx_, float cy_, float mbf_, GeometricCamera *cam)
        : mnId(id), mnBALocalForKF(0), mnBAFixedForKF(0), fx(fx_), fy(fy_), cx(cx_), cy(cy_), mbf(mbf_), mpCamera(cam), mpCamera2(nullptr),
          NLeft(-1), mPrevKF(nullptr), mNextKF(nullptr), bImu(false), mpImuPreintegrated(nullptr), mpMap(pMap), mbBad(false) {}
    void SetPose(const cv::Mat &Tcw_) { Tcw = Tcw_.clone(); }
    cv::Mat GetPose() { return Tcw.clone(); }
    // src/KeyFrame.cc:161-172: Owb = Rwc tcb + Ow, Rwb = Rwc Rcb (float cv::Mat arithmetic)
    cv::Mat GetImuPosition()
    {
        cv::Mat o(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) {
            float a = 0;                                  // Ow = -Rwc tcw
            for (int k = 0; k < 3; k++) a += Tcw.at<float>(k, i) * (mImuCalib.Tcb.at<float>(k, 3) - Tcw.at<float>(k, 3));
            o.at<float>(i) = a;
        }
        return o;
    }
    cv::Mat GetImuRotation()
    {
        cv::Mat R(3, 3, CV_32F);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            float a = 0;
            for (int k = 0; k < 3; k++) a += Tcw.at<float>(k, i) * mImuCalib.Tcb.at<float>(k, j);
            R.at<float>(i, j) = a;
        }
        return R;
    }
    cv::Mat GetVelocity() { return Vw.clone(); }
    void SetVelocity(const cv::Mat &Vw_) { Vw = Vw_.clone(); }
    void SetNewBias(const IMU::Bias &b) { mImuBias = b; if (mpImuPreintegrated) mpImuPreintegrated->SetNewBias(b); }     // KeyFrame.cc:871-877
    IMU::Bias GetImuBias() { return mImuBias; }
    cv::Mat GetGyroBias() { cv::Mat m(3, 1, CV_32F); m.at<float>(0) = mImuBias.bwx; m.at<float>(1) = mImuBias.bwy; m.at<float>(2) = mImuBias.bwz; return m; }
    cv::Mat GetAccBias() { cv::Mat m(3, 1, CV_32F); m.at<float>(0) = mImuBias.bax; m.at<float>(1) = mImuBias.bay; m.at<float>(2) = mImuBias.baz; return m; }
    std::vector<KeyFrame *> GetVectorCovisibleKeyFrames() { return mvpOrderedConnectedKeyFrames; }
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    void EraseMapPointMatch(MapPoint *pMP) { for (auto &p : mvpMapPoints) if (p == pMP) p = nullptr; }
    bool isBad() { return mbBad; }
    Map *GetMap() { return mpMap; }
    long unsigned int mnId;
    long unsigned int mnBALocalForKF, mnBAFixedForKF;
    const float fx, fy, cx, cy, mbf;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    std::vector<float> mvInvLevelSigma2;
    GeometricCamera *mpCamera, *mpCamera2;
    cv::Mat mTrl;
    std::vector<cv::KeyPoint> mvKeysRight;
    int NLeft;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    // inertial members (include/KeyFrame.h:405-470)
    KeyFrame *mPrevKF, *mNextKF;
    bool bImu;
    IMU::Preintegrated *mpImuPreintegrated;
    IMU::Calib mImuCalib;
    // stand-in state
    cv::Mat Tcw, Vw;
    IMU::Bias mImuBias;
    std::vector<KeyFrame *> mvpOrderedConnectedKeyFrames;
    std::vector<MapPoint *> mvpMapPoints;
    Map *mpMap;
    bool mbBad;
};

// include/Map.h
class Map {
public:
    Map() : mnInitKFid(0), mbIsInertial(false), mnMapChange(0), nKeyFrames(0) {}
    long unsigned int KeyFramesInMap() { return nKeyFrames; }
    long unsigned int GetInitKFid() { return mnInitKFid; }
    bool IsInertial() { return mbIsInertial; }
    void IncreaseChangeIndex() { mnMapChange++; }
    std::mutex mMutexMapUpdate;
    long unsigned int mnInitKFid;
    bool mbIsInertial;
    int mnMapChange;
    long unsigned int nKeyFrames;
};

// include/Frame.h (the members ORBmatcher.cc:48-218, 710-825, 1965-2181 read or write)
class Frame {
public:
    Frame() : mbf(0), mb(0), N(0), mpCamera(nullptr), mpCamera2(nullptr), Nleft(-1), Nright(-1) {}
    float mbf, mb;
    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    std::vector<bool> mvbOutlier;
    DBoW2::FeatureVector mFeatVec;
    cv::Mat mTcw;
    std::vector<float> mvScaleFactors;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    GeometricCamera *mpCamera, *mpCamera2;
    int Nleft, Nright;
    std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
    cv::Mat mTrl;
};

}  // namespace ORB_SLAM3
#endif
