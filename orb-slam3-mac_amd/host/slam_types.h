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
        return project(cv::Point3f(x, y, z));
    }
    cv::Point2f project(const cv::Point3f &p3D)
    {
        const float x = p3D.x, y = p3D.y, z = p3D.z;
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
        mbBad(false), nObs(0), nNormalUpdates(0), mfMinDistance(0), mfMaxDistance(0), mNormalVector(cv::Mat::zeros(3, 1, CV_32F)), mpReplaced(nullptr) {}
    // MapPoint.cc:116-127: SetWorldPos takes the class-wide mGlobalMutex (what Optimizer::PoseOptimization holds while it snapshots the
    // positions, Optimizer.cc:895) and then the point's own mMutexPos; GetWorldPos the latter only
    void SetWorldPos(const cv::Mat &Pos) { std::unique_lock<std::mutex> lock2(mGlobalMutex); std::unique_lock<std::mutex> lock(mMutexPos); mWorldPos = Pos.clone(); }
    cv::Mat GetWorldPos() { std::unique_lock<std::mutex> lock(mMutexPos); return mWorldPos.clone(); }
    static inline std::mutex mGlobalMutex;                            // include/MapPoint.h:226 (defined in MapPoint.cc:28 there)
    std::map<KeyFrame *, std::tuple<int, int>> GetObservations() { return mObservations; }
    int Observations() { return nObs; }
    // stand-in helper of the test programs (not a reference method): both indices of an observation at once
    void AddObservation(KeyFrame *pKF, int idxLeft, int idxRight) { mObservations[pKF] = std::make_tuple(idxLeft, idxRight); nObs += (idxLeft != -1) + (idxRight != -1); }
    inline void AddObservation(KeyFrame *pKF, int idx);               // src/MapPoint.cc:114-139 (defined below KeyFrame)
    std::tuple<int, int> GetIndexInKeyFrame(KeyFrame *pKF)             // MapPoint.cc:412-419
    {
        auto it = mObservations.find(pKF);
        return it != mObservations.end() ? it->second : std::tuple<int, int>(-1, -1);
    }
    bool IsInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) != 0; }          // MapPoint.cc:421-425
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }                   // MapPoint.cc:502-506
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }                   // MapPoint.cc:508-512
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    template <class T> int PredictScale(const float &currentDist, T *pKF)               // MapPoint.cc:514-546 (KeyFrame* and Frame* overloads)
    {
        const float ratio = mfMaxDistance / currentDist;
        int nScale = std::ceil(std::log(ratio) / pKF->mfLogScaleFactor);
        if (nScale < 0) nScale = 0;
        else if (nScale >= pKF->mnScaleLevels) nScale = pKF->mnScaleLevels - 1;
        return nScale;
    }
    // MapPoint.cc:233-300 moves the observations over and flags this point bad; the stand-in records the decision (the tests compare it)
    void Replace(MapPoint *pMP) { if (pMP != this) { mpReplaced = pMP; mbBad = true; } }
    MapPoint *GetReplaced() { return mpReplaced; }
    inline void EraseObservation(KeyFrame *pKF);                      // src/MapPoint.cc:168-201 (defined below KeyFrame)
    bool isBad() { return mbBad; }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    void UpdateNormalAndDepth() { nNormalUpdates++; }
    Map *GetMap() { return mpMap; }
    long unsigned int mnId;
    long unsigned int mnBALocalForKF;
    long unsigned int mnBALocalForMerge = 0;
    // Tracking's per-frame projection record (Frame::isInFrustum fills it; ORBmatcher.cc:57-79 reads it)
    float mTrackProjX, mTrackProjY, mTrackDepth, mTrackDepthR, mTrackProjXR, mTrackProjYR;
    bool mbTrackInView, mbTrackInViewR;
    int mnTrackScaleLevel, mnTrackScaleLevelR;
    float mTrackViewCos, mTrackViewCosR;
    cv::Mat mDescriptor;
    // stand-in state
    cv::Mat mWorldPos;
    std::mutex mMutexPos;
    std::map<KeyFrame *, std::tuple<int, int>> mObservations;
    Map *mpMap;
    bool mbBad;
    int nObs, nNormalUpdates;
    float mfMinDistance, mfMaxDistance;
    cv::Mat mNormalVector;
    MapPoint *mpReplaced;
};

// include/KeyFrame.h
class KeyFrame {
public:
    KeyFrame(long unsigned int id, Map *pMap, float fx_, float fy_, float cx_, float cy_, float mbf_, GeometricCamera *cam)
        : mnId(id), mnBALocalForKF(0), mnBAFixedForKF(0), fx(fx_), fy(fy_), cx(cx_), cy(cy_), mbf(mbf_), mpCamera(cam), mpCamera2(nullptr),
          NLeft(-1), mPrevKF(nullptr), mNextKF(nullptr), bImu(false), mpImuPreintegrated(nullptr), mpMap(pMap), mbBad(false) {}
    void SetPose(const cv::Mat &Tcw_) { Tcw = Tcw_.clone(); }
    cv::Mat GetPose() { return Tcw.clone(); }
    // src/KeyFrame.cc:131-160, 1176-1262: pose parts in the reference's float cv::Mat arithmetic (products accumulated in double,
    // one rounding per element, as cv::gemm does for CV_32F)
    cv::Mat GetRotation() { cv::Mat R(3, 3, CV_32F); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.at<float>(i, j) = Tcw.at<float>(i, j); return R; }
    cv::Mat GetTranslation() { cv::Mat t(3, 1, CV_32F); for (int i = 0; i < 3; i++) t.at<float>(i) = Tcw.at<float>(i, 3); return t; }
    cv::Mat GetCameraCenter()                               // Ow = -Rwc * tcw (KeyFrame.cc:113-118, SetPose)
    {
        cv::Mat o(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += (double)Tcw.at<float>(k, i) * (double)Tcw.at<float>(k, 3);
            o.at<float>(i) = (float)(-1.0 * a);
        }
        return o;
    }
    cv::Mat GetRightRotation()                              // Rrw = Rrl * Rlw, Rrl = mTlr.R^T (KeyFrame.cc:1243-1251)
    {
        cv::Mat R(3, 3, CV_32F);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += (double)mTlr.at<float>(k, i) * (double)Tcw.at<float>(k, j);
            R.at<float>(i, j) = (float)a;
        }
        return R;
    }
    cv::Mat GetRightTranslation()                           // trw = Rrl * tlw + trl, trl = -Rrl * tlr (KeyFrame.cc:1253-1262)
    {
        float trl[3];
        for (int i = 0; i < 3; i++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += (double)mTlr.at<float>(k, i) * (double)mTlr.at<float>(k, 3);
            trl[i] = (float)(-1.0 * a);
        }
        cv::Mat t(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += (double)mTlr.at<float>(k, i) * (double)Tcw.at<float>(k, 3);
            t.at<float>(i) = (float)(a + (double)trl[i]);
        }
        return t;
    }
    cv::Mat GetRightPose()                                  // Trw = [Rrw | trw], 3x4 (KeyFrame.cc:1176-1192: the same products)
    {
        const cv::Mat R = GetRightRotation(), t = GetRightTranslation();
        cv::Mat T(3, 4, CV_32F);
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T.at<float>(i, j) = R.at<float>(i, j); T.at<float>(i, 3) = t.at<float>(i); }
        return T;
    }
    cv::Mat GetRightCameraCenter()                          // twr = Rwl * tlr + twl (KeyFrame.cc:1232-1241)
    {
        const cv::Mat Ow = GetCameraCenter();
        cv::Mat t(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += (double)Tcw.at<float>(k, i) * (double)mTlr.at<float>(k, 3);
            t.at<float>(i) = (float)(a + (double)Ow.at<float>(i));
        }
        return t;
    }
    bool IsInImage(const float &x, const float &y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }    // KeyFrame.cc:816-819
    MapPoint *GetMapPoint(const size_t &idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint *pMP, const size_t &idx) { mvpMapPoints[idx] = pMP; }
    std::set<MapPoint *> GetMapPoints()                     // KeyFrame.cc:600-613
    {
        std::set<MapPoint *> s;
        for (size_t i = 0, iend = mvpMapPoints.size(); i < iend; i++) {
            if (!mvpMapPoints[i]) continue;
            MapPoint *pMP = mvpMapPoints[i];
            if (!pMP->isBad()) s.insert(pMP);
        }
        return s;
    }
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
    long unsigned int mnBALocalForMerge = 0;
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
    // members the keyframe-side matchers read (include/KeyFrame.h:360-400)
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys;
    std::vector<float> mvScaleFactors, mvLevelSigma2;
    int mnScaleLevels = 8;
    float mfLogScaleFactor = std::log(1.2f);
    int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
    cv::Mat mTlr;
    // stand-in state
    cv::Mat Tcw, Vw;
    IMU::Bias mImuBias;
    std::vector<KeyFrame *> mvpOrderedConnectedKeyFrames;
    std::vector<MapPoint *> mvpMapPoints;
    Map *mpMap;
    bool mbBad;
};

// src/MapPoint.cc:114-139
inline void MapPoint::AddObservation(KeyFrame *pKF, int idx)
{
    std::tuple<int, int> indexes = mObservations.count(pKF) ? mObservations[pKF] : std::tuple<int, int>(-1, -1);
    if (pKF->NLeft != -1 && idx >= pKF->NLeft) std::get<1>(indexes) = idx;
    else std::get<0>(indexes) = idx;
    mObservations[pKF] = indexes;
    if (!pKF->mpCamera2 && pKF->mvuRight[idx] >= 0) nObs += 2;
    else nObs++;
}

// src/MapPoint.cc:168-201 (SetBadFlag reduced to the flag)
inline void MapPoint::EraseObservation(KeyFrame *pKF)
{
    auto it = mObservations.find(pKF);
    if (it == mObservations.end()) return;
    const int leftIndex = std::get<0>(it->second), rightIndex = std::get<1>(it->second);
    if (leftIndex != -1) {
        if (!pKF->mpCamera2 && pKF->mvuRight[leftIndex] >= 0) nObs -= 2;
        else nObs--;
    }
    if (rightIndex != -1) nObs--;
    mObservations.erase(it);
    if (nObs <= 2) mbBad = true;
}

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
class ORBextractor;
class Frame {
public:
    Frame() : mbf(0), mb(0), N(0), mpORBextractorLeft(nullptr), mpORBextractorRight(nullptr), mpCamera(nullptr), mpCamera2(nullptr), Nleft(-1), Nright(-1),
              mnScaleLevels(8), mfLogScaleFactor(std::log(1.2f)) {}
    float mbf, mb;
    int N;
    // rectified stereo (include/Frame.h:153,239-248,296): the two extractors, the right image's features, and what ComputeStereoMatches fills
    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    cv::Mat mDescriptorsRight;
    std::vector<float> mvDepth;
    // Search a match for each keypoint in the left image to a keypoint in the right image.  If there is a match, depth is computed and the
    // right coordinate associated to the left keypoint is stored.  include/Frame.h:98-99, src/Frame.cc:802-980  (host/Frame.cc)
    void ComputeStereoMatches();
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
    int mnScaleLevels;
    float mfLogScaleFactor;
    std::vector<float> mvInvLevelSigma2;
    static float fx, fy, cx, cy;                          // include/Frame.h:212-217 (static calibration)
    void SetPose(cv::Mat Tcw) { mTcw = Tcw.clone(); }     // src/Frame.cc:352-356 (UpdatePoseMatrices: derived members only)
};

}  // namespace ORB_SLAM3
#endif
