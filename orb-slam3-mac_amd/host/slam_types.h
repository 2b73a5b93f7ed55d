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
          NLeft(-1), mpMap(pMap), mbBad(false) {}
    void SetPose(const cv::Mat &Tcw_) { Tcw = Tcw_.clone(); }
    cv::Mat GetPose() { return Tcw.clone(); }
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
    // stand-in state
    cv::Mat Tcw;
    std::vector<KeyFrame *> mvpOrderedConnectedKeyFrames;
    std::vector<MapPoint *> mvpMapPoints;
    Map *mpMap;
    bool mbBad;
};

// include/Map.h
class Map {
public:
    Map() : mnInitKFid(0), mbIsInertial(false), mnMapChange(0) {}
    long unsigned int GetInitKFid() { return mnInitKFid; }
    bool IsInertial() { return mbIsInertial; }
    void IncreaseChangeIndex() { mnMapChange++; }
    std::mutex mMutexMapUpdate;
    long unsigned int mnInitKFid;
    bool mbIsInertial;
    int mnMapChange;
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
