// frame_cache.cc -- see frame_cache.h
#include "frame_cache.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace ORB_SLAM3 {
namespace hip {

struct ExtractorSlot {
    orbhip_extractor *ext;
    int device;
    std::shared_mutex mu;
};

namespace {
std::mutex g_reg_mu;
std::vector<ExtractorSlot *> &registry() { static std::vector<ExtractorSlot *> r; return r; }
std::atomic<int> g_enabled{-1};
bool enabled()
{
    int e = g_enabled.load();
    if (e < 0) { e = !(std::getenv("ORBHIP_FRAME_CACHE") && std::atoi(std::getenv("ORBHIP_FRAME_CACHE")) == 0); g_enabled.store(e); }
    return e != 0;
}
}  // namespace

void EnableFrameCache(bool on) { g_enabled.store(on ? 1 : 0); }

ExtractorSlot *RegisterExtractor(orbhip_extractor *ext, int device)
{
    ExtractorSlot *s = new ExtractorSlot();
    s->ext = ext; s->device = device;
    std::lock_guard<std::mutex> g(g_reg_mu);
    registry().push_back(s);
    return s;
}

void UnregisterExtractor(ExtractorSlot *slot)
{
    if (!slot) return;
    {
        std::lock_guard<std::mutex> g(g_reg_mu);
        auto &r = registry();
        r.erase(std::remove(r.begin(), r.end(), slot), r.end());
    }
    { std::unique_lock<std::shared_mutex> wait(slot->mu); }          // matcher calls still reading the arrays finish first
    delete slot;
}

std::unique_lock<std::shared_mutex> LockForExtraction(ExtractorSlot *slot)
{
    return slot ? std::unique_lock<std::shared_mutex>(slot->mu) : std::unique_lock<std::shared_mutex>();
}

ResidentFrame FindResident(int device, const void *kp, const uint8_t *desc, int n)
{
    ResidentFrame res;
    if (!enabled() || !desc || n <= 0) return res;
    std::lock_guard<std::mutex> g(g_reg_mu);
    for (ExtractorSlot *s : registry()) {
        if (s->device != device) continue;
        std::shared_lock<std::shared_mutex> hold(s->mu, std::try_to_lock);
        if (!hold.owns_lock()) continue;                             // an extraction is running on it
        const orbhip_keypoint *dk = nullptr, *hk = nullptr; const uint8_t *dd = nullptr, *hd = nullptr; int32_t cnt = 0;
        if (orbhip_extractor_last_frame(s->ext, 0, &dk, &dd, &hk, &hd, &cnt, nullptr) != ORBHIP_OK) continue;
        if (cnt != n || std::memcmp(hd, desc, (size_t)n * 32) != 0) continue;
        res.d_desc = dd;
        res.d_kp = (kp && std::memcmp(hk, kp, (size_t)n * sizeof(orbhip_keypoint)) == 0) ? dk : nullptr;
        res.hold_ = std::move(hold);
        return res;
    }
    return res;
}

ResidentFrame FindResidentIn(orbhip_extractor *ext, const void *kp, const uint8_t *desc, int n)
{
    ResidentFrame res;
    if (!ext || !desc || n <= 0) return res;
    ExtractorSlot *slot = nullptr;
    {
        std::lock_guard<std::mutex> g(g_reg_mu);
        for (ExtractorSlot *s : registry()) if (s->ext == ext) { slot = s; break; }
    }
    if (!slot) return res;
    std::shared_lock<std::shared_mutex> hold(slot->mu);              // (the caller owns the extractor object: the slot outlives this call)
    const orbhip_keypoint *dk = nullptr, *hk = nullptr; const uint8_t *dd = nullptr, *hd = nullptr; int32_t cnt = 0;
    if (orbhip_extractor_last_frame(ext, 0, &dk, &dd, &hk, &hd, &cnt, nullptr) != ORBHIP_OK) return res;
    if (cnt != n || std::memcmp(hd, desc, (size_t)n * 32) != 0) return res;
    res.d_desc = dd;
    res.d_kp = (kp && std::memcmp(hk, kp, (size_t)n * sizeof(orbhip_keypoint)) == 0) ? dk : nullptr;
    res.hold_ = std::move(hold);
    return res;
}

}  // namespace hip
}  // namespace ORB_SLAM3
