// Frame.cc -- the Frame member functions of the hot path that run on the device: Frame::ComputeStereoMatches
// (reference src/Frame.cc:802-980, called by the rectified-stereo constructor at :130 right after the two ExtractORB threads, :109-112).
// Same signature, same members read and written; the association itself (row bands, descriptor distances, 11 x 11 SAD search on the
// left keypoint's pyramid level of BOTH images, parabola fit, median cut) is orbhip_compute_stereo_matches_* behind the C ABI, on the
// keypoints, descriptors and pyramids the two ORBextractor objects left on the device -- nothing is uploaded and the mvImagePyramid
// copy-back of the reference's host loop (:809, :899, :913, :918) is not needed on this path.
#include <cstdio>
#include "ORBextractor.h"
#include "slam_types.h"
#include "frame_cache.h"

namespace ORB_SLAM3 {

void Frame::ComputeStereoMatches()
{
    mvuRight = std::vector<float>(N, -1.0f);                                  // :804-805
    mvDepth = std::vector<float>(N, -1.0f);
    if (N == 0) return;
    if (!mpORBextractorLeft || !mpORBextractorRight) { fprintf(stderr, "Frame (HIP): ComputeStereoMatches: no extractors\n"); return; }
    orbhip_extractor *eL = mpORBextractorLeft->DeviceExtractor(), *eR = mpORBextractorRight->DeviceExtractor();
    // The kernels read the two extractors' LATEST extractions: this Frame's mvKeys / mDescriptors and mvKeysRight / mDescriptorsRight must be
    // those (they are when the constructor calls this; byte comparison with the extractors' page-locked mirrors, host/frame_cache.h).  The
    // shared locks keep the next operator() of either extractor out until the kernels have finished.
    hip::ResidentFrame rl = hip::FindResidentIn(eL, mvKeys.data(), mDescriptors.ptr<uint8_t>(), N);
    const bool right_empty = mvKeysRight.empty();                             // (a featureless right image: nothing can match)
    hip::ResidentFrame rr = right_empty ? hip::ResidentFrame() : hip::FindResidentIn(eR, mvKeysRight.data(), mDescriptorsRight.ptr<uint8_t>(), (int)mvKeysRight.size());
    if (!rl || !rl.d_kp || (!right_empty && (!rr || !rr.d_kp))) {
        fprintf(stderr, "Frame (HIP): ComputeStereoMatches: the frame's features are not the latest extractions of its two extractors (left %s, right %s)\n",
                rl && rl.d_kp ? "ok" : "no", right_empty || (rr && rr.d_kp) ? "ok" : "no");
        return;
    }
    if (right_empty) return;
    int32_t kept = 0;
    const int rc = orbhip_compute_stereo_matches_host(eL, eR, mb, mbf, mvuRight.data(), mvDepth.data(), N, &kept);
    if (rc != ORBHIP_OK) {
        fprintf(stderr, "Frame (HIP): ComputeStereoMatches: %d (%s)\n", rc, orbhip_last_error());
        mvuRight.assign(N, -1.0f); mvDepth.assign(N, -1.0f);
    }
}

}  // namespace ORB_SLAM3
