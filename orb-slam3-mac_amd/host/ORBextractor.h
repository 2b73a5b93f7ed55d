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
#include "frame_cache.h"

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

    // PUBLIC in the reference (include/ORBextractor.h:83: std::vector<cv::Mat> mvImagePyramid), read by Frame::ComputeStereoMatches
    // (src/Frame.cc:809,899,913,918) as mvImagePyramid[level] right after operator() with no further call.  Level l is the
    // w_l x h_l ROI at (19,19) of a reflect-101 padded (w_l+38) x (h_l+38) parent, as ORBextractor.cc:1160-1173 builds it.
    // Round 4: the pyramid stays on the device until somebody reads it -- the member is a vector-like
    // object whose element access materialises the host copy of the LATEST extraction on first use (one device-to-host copy per level
    // into page-locked memory + the border synthesis), so monocular / fisheye tracking, which never reads it, pays nothing, and the
    // unchanged stereo caller gets exactly what it got before.  SetImagePyramidSync(true) restores the eager copy inside operator().
    class ImagePyramid {
    public:
        explicit ImagePyramid(ORBextractor *owner) : owner_(owner), stale_(false) {}
        size_t size() const { return v_.size(); }
        bool empty() const { return v_.empty(); }
        void resize(size_t n) { v_.resize(n); }
        cv::Mat &operator[](size_t i) { fresh(); return v_[i]; }
        const cv::Mat &operator[](size_t i) const { fresh(); return v_[i]; }
        cv::Mat &at(size_t i) { fresh(); return v_.at(i); }
        std::vector<cv::Mat>::iterator begin() { fresh(); return v_.begin(); }
        std::vector<cv::Mat>::iterator end() { fresh(); return v_.end(); }
        operator std::vector<cv::Mat> &() { fresh(); return v_; }              // code that wants the plain vector
    private:
        friend class ORBextractor;
        void fresh() const { if (stale_) { stale_ = false; owner_->SyncImagePyramid(); } }
        ORBextractor *owner_;
        mutable std::vector<cv::Mat> v_;
        mutable bool stale_;
    };
    ImagePyramid mvImagePyramid;
    void SetImagePyramidSync(bool on) { syncPyramid_ = on; }
    // (not in the reference) the device-side extractor behind this object: Frame::ComputeStereoMatches (host/Frame.cc) pairs the two
    // extractors' latest extractions where they lie, device pyramids included
    orbhip_extractor *DeviceExtractor() const { return ext_; }
    void SyncImagePyramid();                       // materialise the host pyramid of the latest extraction now

protected:
    int nfeatures; double scaleFactor; int nlevels; int iniThFAST; int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

private:
    orbhip_ctx *ctx_;
    orbhip_extractor *ext_;
    hip::ExtractorSlot *slot_;                     // frame_cache.h: the latest extraction stays on the device for the matchers
    std::vector<std::vector<uint8_t>> padded_;     // backing store of mvImagePyramid (kept across calls)
    int stageW_, stageH_;
    bool syncPyramid_;
    ORBextractor(const ORBextractor &);            // one instance = one device context (not copyable)
    ORBextractor &operator=(const ORBextractor &);
};

}  // namespace ORB_SLAM3
