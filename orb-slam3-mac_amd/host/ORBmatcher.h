// ORBmatcher.h -- signature-preserving host mirror of ORB_SLAM3::ORBmatcher (reference include/ORBmatcher.h:39-88) for the
// methods whose search runs in a HIP kernel behind the C ABI: Tracking calls them unchanged
//   SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)   src/Tracking.cc:3096  (TrackLocalMap)
//   SearchByProjection(Frame&, const Frame&, th, bMono)                                  src/Tracking.cc:2683  (TrackWithMotionModel)
//   SearchForInitialization(Frame&, Frame&, vbPrevMatched, vnMatches12, windowSize)      src/Tracking.cc:1506  (MonocularInitialization)
//   SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches)                                    src/Tracking.cc:1757  (TrackReferenceKeyFrame), :3290 (Relocalization)
//   DescriptorDistance(a, b)
// The per-point host geometry in front of each search (projection, frustum record, radius, level range) is kept as the
// reference writes it; the windowed best / second-best search with the claim rule, the ratio tests and the rotation
// histogram run on the device.  The remaining searches (keyframe-keyframe BoW, triangulation, Fuse, Sim3) have batched device entry points in
// include/orbhip.h and INTEGRATION.md shows their call sites; their class methods are not mirrored here.
#pragma once
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

    // Search matches between MapPoints in a KeyFrame and ORB in a Frame. Brute force constrained to ORB that belong to the same vocabulary
    // node (at a certain level). Used in Relocalisation and Loop Detection     include/ORBmatcher.h:62, src/ORBmatcher.cc:273-475
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);

    // Matching for the Map Initialization (only used in the monocular case)    include/ORBmatcher.h:66, src/ORBmatcher.cc:710-825
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12,
                                int windowSize = 10);

    static const int TH_LOW;         // 50   ORBmatcher.cc:41
    static const int TH_HIGH;        // 100  ORBmatcher.cc:40
    static const int HISTO_LENGTH;   // 30   ORBmatcher.cc:42

protected:
    float RadiusByViewingCos(const float &viewCos);        // ORBmatcher.cc:220-226

    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace ORB_SLAM3
