// ORBextractor.cc -- host marshalling for the signature-preserving ORBextractor (see header).  No exceptions cross this
// boundary (the reference uses none) and the process is never aborted: a device that is missing at construction is reported on
// stderr (the reference's constructor cannot fail and there is no CPU path to fall back to) and every later operator() answers like
// the reference's only failure, the empty image: -1, no keypoints (ORBextractor.cc:1072-1073); so does a device error inside
// operator().  The GPU is hip::GetDevice() (hip_context.h: SetDevice(n) / ORBHIP_DEVICE / 0).
#include "ORBextractor.h"
#include "hip_context.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ORB_SLAM3 {

static void complain(int rc, const char *what)
{
    fprintf(stderr, "ORBextractor (HIP): %s failed: %d (%s) -- this build needs an MI355X, there is no CPU fallback; operator() will return -1\n", what, rc,
            orbhip_last_error());
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : mvImagePyramid(this), nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST),
      ctx_(nullptr), ext_(nullptr), slot_(nullptr), stageW_(0), stageH_(0), syncPyramid_(false)
{
    mvImagePyramid.resize(nlevels);
    int rc = orbhip_ctx_create(hip::GetDevice(), nullptr, &ctx_);           // one context (stream) per extractor instance: Frame.cc:109-110
    if (rc != ORBHIP_OK) { complain(rc, "orbhip_ctx_create"); ctx_ = nullptr; return; }
    rc = orbhip_extractor_create(ctx_, _nfeatures, _scaleFactor, _nlevels, _iniThFAST, _minThFAST, &ext_);
    if (rc != ORBHIP_OK) { complain(rc, "orbhip_extractor_create"); ext_ = nullptr; return; }
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    orbhip_extractor_table(ext_, 0, mvScaleFactor.data());
    orbhip_extractor_table(ext_, 1, mvInvScaleFactor.data());
    orbhip_extractor_table(ext_, 2, mvLevelSigma2.data());
    orbhip_extractor_table(ext_, 3, mvInvLevelSigma2.data());
    mnFeaturesPerLevel.resize(nlevels);
    orbhip_extractor_features_per_level(ext_, mnFeaturesPerLevel.data());
    umax.resize(16);
    orbhip_extractor_umax(ext_, umax.data());
    slot_ = hip::RegisterExtractor(ext_, hip::GetDevice());
}

ORBextractor::~ORBextractor()
{
    hip::UnregisterExtractor(slot_);                                         // waits for matcher calls that still read this extractor's arrays
    if (ext_) orbhip_extractor_destroy(ext_);
    if (ctx_) orbhip_ctx_destroy(ctx_);
}

int ORBextractor::operator()(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &keypoints,
                             cv::OutputArray descriptors, std::vector<int> &vLappingArea)
{
#ifdef ORBHIP_WITH_OPENCV
    cv::Mat img = image.getMat();
#else
    const cv::Mat &img = image;
#endif
    if (img.empty()) return -1;                                             // ORBextractor.cc:1072-1073
    if (!ext_) { keypoints.clear(); descriptors.release(); return -1; }     // no device (reported by the constructor)
    // the previous extraction's device arrays are about to be overwritten: matcher calls that read them (frame_cache.h) finish first
    std::unique_lock<std::shared_mutex> writer = hip::LockForExtraction(slot_);
    mvImagePyramid.stale_ = false;
    if (img.cols != stageW_ || img.rows != stageH_) {                       // device buffers + page-locked staging: once per image size
        const int rc = orbhip_extractor_reserve(ext_, img.cols, img.rows, 1);
        if (rc != ORBHIP_OK) { fprintf(stderr, "ORBextractor (HIP): reserve %dx%d: %d (%s)\n", img.cols, img.rows, rc, orbhip_last_error()); return -1; }
        stageW_ = img.cols; stageH_ = img.rows;
    }
    const orbhip_keypoint *kpView = nullptr; const uint8_t *descView = nullptr; const int32_t *countView = nullptr, *monoView = nullptr; int rowCap = 0;
    const int rc = orbhip_extract_batch_host_view(ext_, img.data, img.cols, img.rows, img.step, img.step * img.rows, 1, vLappingArea[0], vLappingArea[1],
                                                  &kpView, &descView, &rowCap, &countView, &monoView);
    if (rc != ORBHIP_OK) {
        if (rc != ORBHIP_E_EMPTY) fprintf(stderr, "ORBextractor (HIP): extract: %d (%s)\n", rc, orbhip_last_error());
        keypoints.clear(); descriptors.release();
        return -1;
    }
    const int count = countView[0], mono = monoView[0];
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbhip_keypoint), "KeyPoint layout");
    keypoints.resize(count);                                                // _keypoints = vector<cv::KeyPoint>(nkeypoints), :1100
    if (count) memcpy((void *)keypoints.data(), kpView, sizeof(orbhip_keypoint) * count);       // straight from the page-locked mirror
    if (count == 0) descriptors.release();                                  // :1090-1091
    else {
        descriptors.create(count, 32, CV_8U);                               // :1094
#ifdef ORBHIP_WITH_OPENCV
        memcpy(descriptors.getMat().data, descView, (size_t)count * 32);
#else
        memcpy(descriptors.data, descView, (size_t)count * 32);
#endif
    }
    writer.unlock();
    if (syncPyramid_) SyncImagePyramid();
    else mvImagePyramid.stale_ = true;                                      // materialised by the first mvImagePyramid[...] read (Frame.cc:809)
    return mono;
}

void ORBextractor::SyncImagePyramid()
{
    if (!ext_) return;
    mvImagePyramid.stale_ = false;
    padded_.resize(nlevels);
    std::vector<uint8_t *> lv(nlevels);
    std::vector<size_t> st(nlevels);
    std::vector<int> ws(nlevels), hs(nlevels);
    for (int l = 0; l < nlevels; l++) {
        if (orbhip_extractor_level_dims(ext_, l, &ws[l], &hs[l]) != ORBHIP_OK) return;          // nothing extracted yet
        const size_t need = (size_t)(ws[l] + 38) * (hs[l] + 38);
        if (padded_[l].size() != need) padded_[l].resize(need);
        lv[l] = padded_[l].data(); st[l] = (size_t)ws[l] + 38;
    }
    const int rc = orbhip_extractor_get_pyramid_padded(ext_, 0, lv.data(), st.data());
    if (rc != ORBHIP_OK) { fprintf(stderr, "ORBextractor (HIP): pyramid copy-out: %d (%s)\n", rc, orbhip_last_error()); return; }
    for (int l = 0; l < nlevels; l++) {
        const int pw = ws[l] + 38, ph = hs[l] + 38;
        // ROI view at (19,19) inside the reflect-101 padded parent, like ORBextractor.cc:1160
#ifdef ORBHIP_WITH_OPENCV
        mvImagePyramid.v_[l] = cv::Mat(ph, pw, CV_8U, padded_[l].data(), pw)(cv::Rect(19, 19, ws[l], hs[l]));
#else
        (void)ph;
        mvImagePyramid.v_[l] = cv::Mat(hs[l], ws[l], CV_8U, padded_[l].data() + (size_t)19 * pw + 19, pw);
#endif
    }
}

}  // namespace ORB_SLAM3
