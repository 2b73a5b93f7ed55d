// ORBextractor.h -- signature-preserving host mirror of ORB_SLAM3::ORBextractor
// (reference include/ORBextractor.h:44-112) on top of the orbhip C ABI.  Tracking / Frame
// (src/Frame.cc:410-417, src/Tracking.cc:206-212) compile against this unchanged: same ctor,
// same operator(), same getters, same public mvImagePyramid.  All arithmetic runs in the HIP
// kernels; this class only marshals.
#pragma once
#include <vector>
#include "cvlite.h"
#ifdef ORBHIP_WITH_OPENCV
#include <opencv2/core.hpp>
#endif
#include "../../include/orbhip.h"

namespace ORB_SLAM3 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };      // include/ORBextractor.h:47 (HARRIS_SCORE is unused there too)

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();

    // Compute the ORB features and descriptors on an image; mask is ignored (as in the reference).
    // Returns monoIndex; -1 if the image is empty (ORBextractor.cc:1072-1073).
    int operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints,
                   cv::OutputArray descriptors, std::vector<int> &vLappingArea);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // PUBLIC in the reference (include/ORBextractor.h:83), read by Frame::ComputeStereoMatches (src/Frame.cc:809,899,913,918)
    // right after operator() with no further call: operator() fills it itself -- level l is the w_l x h_l ROI at (19,19) of a
    // reflect-101 padded (w_l+38) x (h_l+38) parent, as ORBextractor.cc:1160-1173 builds it.  That costs one device-to-host
    // copy of the pyramid per call; callers that never read it (monocular / fisheye tracking, or stereo through
    // orbhip_compute_stereo_matches_device, which reads the device pyramids) switch it off with SetImagePyramidSync(false).
    std::vector<cv::Mat> mvImagePyramid;
    void SetImagePyramidSync(bool on) { syncPyramid_ = on; }
    void SyncImagePyramid();                       // explicit refresh (the opt-out case)

protected:
    int nfeatures; double scaleFactor; int nlevels; int iniThFAST; int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

private:
    orbhip_ctx *ctx_;
    orbhip_extractor *ext_;
    std::vector<std::vector<uint8_t>> padded_;     // backing store of mvImagePyramid (kept across calls)
    std::vector<orbhip_keypoint> kpStage_;         // staging of the C ABI's output rows (kept across calls, sized once per image size)
    std::vector<uint8_t> descStage_;
    int stageW_, stageH_, cap_;
    bool syncPyramid_;
    ORBextractor(const ORBextractor &);            // one instance = one device context (not copyable)
    ORBextractor &operator=(const ORBextractor &);
};

}  // namespace ORB_SLAM3
