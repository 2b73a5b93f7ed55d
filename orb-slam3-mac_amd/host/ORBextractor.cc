// ORBextractor.cc -- host marshalling for the signature-preserving ORBextractor (see header).
#include "ORBextractor.h"
#include <stdexcept>
#include <string>

namespace ORB_SLAM3 {

static void chk(int rc, const char *what)
{
    if (rc != ORBHIP_OK) throw std::runtime_error(std::string(what) + ": " + orbhip_last_error());
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST),
      ctx_(nullptr), ext_(nullptr)
{
    chk(orbhip_ctx_create(0, nullptr, &ctx_), "orbhip_ctx_create");        // fails loudly without a GPU
    chk(orbhip_extractor_create(ctx_, _nfeatures, _scaleFactor, _nlevels, _iniThFAST, _minThFAST, &ext_), "orbhip_extractor_create");
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    orbhip_extractor_table(ext_, 0, mvScaleFactor.data());
    orbhip_extractor_table(ext_, 1, mvInvScaleFactor.data());
    orbhip_extractor_table(ext_, 2, mvLevelSigma2.data());
    orbhip_extractor_table(ext_, 3, mvInvLevelSigma2.data());
    mnFeaturesPerLevel.resize(nlevels);
    orbhip_extractor_features_per_level(ext_, mnFeaturesPerLevel.data());
    umax.resize(16);
    orbhip_extractor_umax(ext_, umax.data());
    mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor()
{
    orbhip_extractor_destroy(ext_);
    orbhip_ctx_destroy(ctx_);
}

int ORBextractor::operator()(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &keypoints,
                             cv::OutputArray descriptors, std::vector<int> &vLappingArea)
{
#ifdef ORBHIP_WITH_OPENCV
    cv::Mat img = image.getMat();
    if (img.empty()) return -1;
#else
    const cv::Mat &img = image;
    if (img.empty()) return -1;
#endif
    chk(orbhip_extractor_reserve(ext_, img.cols, img.rows, 1), "orbhip_extractor_reserve");
    const int cap = orbhip_extractor_max_keypoints(ext_);
    std::vector<orbhip_keypoint> kp(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int32_t count = 0, mono = 0;
    int rc = orbhip_extract_batch_host(ext_, img.data, img.cols, img.rows, img.step, img.step * img.rows, 1, vLappingArea[0],
                                       vLappingArea[1], kp.data(), desc.data(), cap, &count, &mono);
    if (rc == ORBHIP_E_EMPTY) return -1;
    chk(rc, "orbhip_extract_batch_host");
    keypoints.resize(count);
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbhip_keypoint), "KeyPoint layout");
    if (count) memcpy((void *)keypoints.data(), kp.data(), sizeof(orbhip_keypoint) * count);
#ifdef ORBHIP_WITH_OPENCV
    if (count == 0) descriptors.release();
    else { descriptors.create(count, 32, CV_8U); memcpy(descriptors.getMat().data, desc.data(), (size_t)count * 32); }
#else
    if (count == 0) descriptors.release();
    else { descriptors.create(count, 32, cv::CV_8U); memcpy(descriptors.data, desc.data(), (size_t)count * 32); }
#endif
    return mono;
}

void ORBextractor::SyncImagePyramid()
{
    padded_.resize(nlevels);
    for (int l = 0; l < nlevels; l++) {
        int w = 0, h = 0;
        chk(orbhip_extractor_level_dims(ext_, l, &w, &h), "orbhip_extractor_level_dims");
        const int pw = w + 38, ph = h + 38;
        padded_[l].resize((size_t)pw * ph);
        chk(orbhip_extractor_get_pyramid_level(ext_, 0, l, 1, padded_[l].data(), pw), "orbhip_extractor_get_pyramid_level");
        // ROI view at (19,19) inside the reflect-101 padded parent, like ORBextractor.cc:1160
#ifdef ORBHIP_WITH_OPENCV
        mvImagePyramid[l] = cv::Mat(ph, pw, CV_8U, padded_[l].data(), pw)(cv::Rect(19, 19, w, h));
#else
        mvImagePyramid[l] = cv::Mat(h, w, cv::CV_8U, padded_[l].data() + (size_t)19 * pw + 19, pw);
#endif
    }
}

}  // namespace ORB_SLAM3
