// ORBmatcher.h -- signature-preserving host mirror of ORB_SLAM3::ORBmatcher (reference include/ORBmatcher.h:39-88) for the
// methods whose search runs in a HIP kernel behind the C ABI: Tracking calls them unchanged
//   SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)   src/Tracking.cc:3096  (TrackLocalMap)
//   SearchByProjection(Frame&, const Frame&, th, bMono)                                  src/Tracking.cc:2683  (TrackWithMotionModel)
//   SearchForInitialization(Frame&, Frame&, vbPrevMatched, vnMatches12, windowSize)      src/Tracking.cc:1506  (MonocularInitialization)
//   SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches)                                    src/Tracking.cc:1757  (TrackReferenceKeyFrame), :3290 (Relocalization)
//   DescriptorDistance(a, b)
// and, since round 3, every other public method of the class (ORBmatcher_keyframe.cc): LocalMapping and LoopClosing call them unchanged
//   SearchByProjection(Frame&, KeyFrame*, set<MapPoint*>&, th, ORBdist)                  src/Tracking.cc:2739, 2753  (Relocalization)
//   SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th, ratioHamming)            src/LoopClosing.cc
//   SearchByProjection(KeyFrame*, Scw, vpPoints, vpPointsKFs, vpMatched, vpMatchedKF, th, ratioHamming)
//   SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)                                       src/LoopClosing.cc:1005, 2284
//   SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse)         src/LocalMapping.cc:463
//   SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)                             src/LoopClosing.cc
//   Fuse(pKF, vpMapPoints, th, bRight)                                                   src/LocalMapping.cc:787-788, 816-817
//   Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)                                         src/LoopClosing.cc
// (SearchForTriangulation's second overload with vMatchedPoints, include/ORBmatcher.h:76-77, has no caller in the reference; it is mirrored all the same.)
// The per-point host geometry in front of each search (projection, frustum record, radius, level range) is kept as the
// reference writes it; the windowed best / second-best search with the claim rule, the ratio tests, the reprojection and epipolar gates
// and the rotation histogram run on the device.
#pragma once
#include <set>
#include <utility>
#include <vector>
#include "slam_types.h"
#include "../../include/orbhip.h"

namespace ORB_SLAM3 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // Computes the Hamming distance between two ORB descriptors (ORBmatcher.cc:2353-2369).
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);

    // Search matches between Frame keypoints and projected MapPoints. Returns number of matches.
    // Used to track the local map (Tracking)                                   include/ORBmatcher.h:49, src/ORBmatcher.cc:48-218
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3, const bool bFarPoints = false,
                           const float thFarPoints = 50.0f);

    // Project MapPoints tracked in last frame into the current frame and search matches.
    // Used to track from previous frame (Tracking)                             include/ORBmatcher.h:53, src/ORBmatcher.cc:1965-2181
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);

    // Project MapPoints seen in KeyFrame into the Frame and search matches.
    // Used in relocalisation (Tracking)                                        include/ORBmatcher.h:54, src/ORBmatcher.cc:2183-2305
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist);

    // Project MapPoints using a Similarity Transformation and search matches.
    // Used in loop detection (Loop Closing)                                    include/ORBmatcher.h:58, src/ORBmatcher.cc:477-591
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th,
                           float ratioHamming = 1.0);

    // Project MapPoints using a Similarity Transformation and search matches.
    // Used in Place Recognition (Loop Closing and Merging)                     include/ORBmatcher.h:62, src/ORBmatcher.cc:593-708
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, const std::vector<KeyFrame *> &vpPointsKFs,
                           std::vector<MapPoint *> &vpMatched, std::vector<KeyFrame *> &vpMatchedKF, int th, float ratioHamming = 1.0);

    // Search matches between MapPoints in a KeyFrame and ORB in a Frame. Brute force constrained to ORB that belong to the same vocabulary
    // node (at a certain level). Used in Relocalisation and Loop Detection     include/ORBmatcher.h:67-68, src/ORBmatcher.cc:273-475, 827-967
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12);

    // Matching for the Map Initialization (only used in the monocular case)    include/ORBmatcher.h:66, src/ORBmatcher.cc:710-825
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12,
                                int windowSize = 10);

    // Matching to triangulate new MapPoints. Check Epipolar Constraint.        include/ORBmatcher.h:74, src/ORBmatcher.cc:969-1210
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>> &vMatchedPairs,
                               const bool bOnlyStereo, const bool bCoarse = false);
    // ... returning the triangulated points as well (GeometricCamera::matchAndtriangulate)   include/ORBmatcher.h:76-77, src/ORBmatcher.cc:1212-1402
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>> &vMatchedPairs,
                               const bool bOnlyStereo, std::vector<cv::Mat> &vMatchedPoints);

    // Search matches between MapPoints seen in KF1 and KF2 transforming by a Sim3 [s12*R12|t12]
    // In the stereo and RGB-D case, s12=1                                      include/ORBmatcher.h:82, src/ORBmatcher.cc:1739-1963
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12,
                     const float th);

    // Project MapPoints into KeyFrame and search for duplicated MapPoints.     include/ORBmatcher.h:85, src/ORBmatcher.cc:1403-1613
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0, const bool bRight = false);

    // Project MapPoints into KeyFrame using a given Sim3 and search for duplicated MapPoints.  include/ORBmatcher.h:88, src/ORBmatcher.cc:1615-1737
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint);

    static const int TH_LOW;         // 50   ORBmatcher.cc:41
    static const int TH_HIGH;        // 100  ORBmatcher.cc:40
    static const int HISTO_LENGTH;   // 30   ORBmatcher.cc:42

protected:
    float RadiusByViewingCos(const float &viewCos);        // ORBmatcher.cc:220-226

    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace ORB_SLAM3
