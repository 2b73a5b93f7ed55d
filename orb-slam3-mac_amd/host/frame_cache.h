// frame_cache.h -- which Frame's features are still on the device.
//
// ORBextractor::operator() leaves the keypoints and descriptors of the frame it has just extracted in its device result arrays and
// keeps a page-locked host mirror of them (orbhip_extractor_last_frame).  Tracking's next calls search exactly that frame:
// SearchByProjection(mCurrentFrame, mLastFrame, ...) (src/Tracking.cc:1911), SearchLocalPoints (:3083), SearchByBoW(pKF, mCurrentFrame,
// ...) (:1757) and SearchForInitialization(mInitialFrame, mCurrentFrame, ...) (:1506).  The ORBmatcher methods ask here whether the Frame
// they were handed IS such an extraction and, if so, give the kernels the device arrays instead of uploading the frame again.
//
// The reference's ORBextractor knows nothing of Frames (operator() runs inside the Frame constructor, src/Frame.cc:410-417) and a Frame's
// cv::Mat / vector buffers are re-used by the allocator from frame to frame, so neither mnId nor a pointer can identify the contents.
// The test is therefore the CONTENT: same count and byte-identical descriptors (memcmp against the mirror, ~1 us per 32 KB; a mismatch
// shows in the first bytes).  Keypoints are compared separately: with a distorted camera mvKeysUn differs from what the extractor wrote
// and is uploaded while the descriptors stay resident.  A hit can therefore never change a result -- the device arrays hold the very
// bytes that would have been uploaded.
//
// Lifetime: the extractor's next call overwrites the arrays.  Every extractor has a slot with a reader / writer lock: operator()
// holds it exclusively while it runs, a matcher call that uses the arrays holds it shared until its kernels have finished.
#pragma once
#include <mutex>
#include <shared_mutex>
#include "../../include/orbhip.h"

namespace ORB_SLAM3 {
namespace hip {

struct ExtractorSlot;

class ResidentFrame {
public:
    ResidentFrame() : d_kp(nullptr), d_desc(nullptr) {}
    ResidentFrame(ResidentFrame &&) = default;
    ResidentFrame &operator=(ResidentFrame &&) = default;
    const orbhip_keypoint *d_kp;     // nullptr: the frame's keypoints are not the extraction's (undistorted): upload them
    const uint8_t *d_desc;           // nullptr: not resident
    explicit operator bool() const { return d_desc != nullptr; }
private:
    std::shared_lock<std::shared_mutex> hold_;
    friend ResidentFrame FindResident(int, const void *, const uint8_t *, int);
    friend ResidentFrame FindResidentIn(orbhip_extractor *, const void *, const uint8_t *, int);
};

ExtractorSlot *RegisterExtractor(orbhip_extractor *ext, int device);
void UnregisterExtractor(ExtractorSlot *slot);
std::unique_lock<std::shared_mutex> LockForExtraction(ExtractorSlot *slot);
// kp: n cv::KeyPoint records (28 bytes each) or nullptr, desc: n x 32 bytes.  ORBHIP_FRAME_CACHE=0 switches the lookup off.
ResidentFrame FindResident(int device, const void *kp, const uint8_t *desc, int n);
// The same test against ONE extractor's latest extraction (Frame::ComputeStereoMatches needs exactly the left and the right extractor's,
// whatever else is resident; it has no upload path, so the ORBHIP_FRAME_CACHE switch does not apply).  Waits for a running extraction.
ResidentFrame FindResidentIn(orbhip_extractor *ext, const void *kp, const uint8_t *desc, int n);
void EnableFrameCache(bool on);          // run-time switch (measurements: host_latency.cc)

}  // namespace hip
}  // namespace ORB_SLAM3
