// ORBmatcher.h -- host mirror of the ORBmatcher entry points that have a HIP implementation
// (reference include/ORBmatcher.h:39-88).  DescriptorDistance is the same static function;
// SearchForInitialization keeps the reference's argument meaning but takes plain keypoint /
// descriptor arrays instead of Frame objects (Frame is the caller, out of scope: SURVEY 8b).
#pragma once
#include <vector>
#include "cvlite.h"
#ifdef ORBHIP_WITH_OPENCV
#include <opencv2/core.hpp>
#endif
#include "../../include/orbhip.h"

namespace ORB_SLAM3 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // Computes the Hamming distance between two ORB descriptors (ORBmatcher.cc:2353-2369).
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orbhip_descriptor_distance(a.ptr<uint8_t>(), b.ptr<uint8_t>()); }

    static const int TH_LOW = 50;        // ORBmatcher.cc:41
    static const int TH_HIGH = 100;      // ORBmatcher.cc:40
    static const int HISTO_LENGTH = 30;  // ORBmatcher.cc:42

    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace ORB_SLAM3
