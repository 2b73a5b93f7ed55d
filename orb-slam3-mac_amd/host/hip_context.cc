// hip_context.cc -- see hip_context.h
#include "hip_context.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace ORB_SLAM3 {
namespace hip {

namespace {
std::atomic<int> g_device{-1};          // -1 = not chosen yet
struct Ctx {
    orbhip_ctx *h; bool tried;
    Ctx() : h(nullptr), tried(false) {}
    ~Ctx() { if (h) orbhip_ctx_destroy(h); }
};
}  // namespace

void SetDevice(int device) { g_device.store(device < 0 ? 0 : device); }

int GetDevice()
{
    int d = g_device.load();
    if (d >= 0) return d;
    const char *ev = std::getenv("ORBHIP_DEVICE");
    d = ev ? std::atoi(ev) : 0;
    if (d < 0) d = 0;
    int expected = -1;
    g_device.compare_exchange_strong(expected, d);
    return g_device.load();
}

orbhip_ctx *ThreadContext()
{
    static thread_local Ctx c;
    if (!c.h && !c.tried) {
        c.tried = true;
        const int rc = orbhip_ctx_create(GetDevice(), nullptr, &c.h);
        if (rc != ORBHIP_OK) {
            c.h = nullptr;
            fprintf(stderr, "orbhip: no device context on GPU %d: %d (%s) -- this build needs an MI355X, there is no CPU fallback\n", GetDevice(), rc,
                    orbhip_last_error());
        }
    }
    return c.h;
}

}  // namespace hip
}  // namespace ORB_SLAM3
