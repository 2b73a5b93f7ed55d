// host_prof.h -- where a host-class call spends its wall-clock time (argument packing / the C-ABI call / unpacking), accumulated per
// method when ORBHIP_HOST_PROF is set (host_smoke latency prints it); otherwise one predictable branch per mark.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ORB_SLAM3 {
namespace hip {

struct HostProfEntry { const char *name; double ms[4]; long calls; };
inline HostProfEntry *host_prof_table() { static HostProfEntry t[16]; return t; }
inline bool host_prof_on() { static const bool on = std::getenv("ORBHIP_HOST_PROF") != nullptr; return on; }

// marks: begin -> mark(0) = packing done -> mark(1) = device call returned -> end = unpacking done
class HostProf {
public:
    explicit HostProf(const char *name) : e_(nullptr), k_(0)
    {
        if (!host_prof_on()) return;
        HostProfEntry *t = host_prof_table();
        for (int i = 0; i < 16; i++) {
            if (!t[i].name) t[i].name = name;
            if (t[i].name == name || !std::strcmp(t[i].name, name)) { e_ = &t[i]; break; }
        }
        t0_ = std::chrono::steady_clock::now();
    }
    void mark()
    {
        if (!e_) return;
        const auto t = std::chrono::steady_clock::now();
        if (k_ < 4) e_->ms[k_++] += std::chrono::duration<double, std::milli>(t - t0_).count();
        t0_ = t;
    }
    ~HostProf() { if (e_) { mark(); e_->calls++; } }
private:
    HostProfEntry *e_; int k_;
    std::chrono::steady_clock::time_point t0_;
};

inline void host_prof_reset() { std::memset(host_prof_table(), 0, sizeof(HostProfEntry) * 16); }
inline void host_prof_dump(FILE *f)
{
    HostProfEntry *t = host_prof_table();
    for (int i = 0; i < 16 && t[i].name; i++)
        if (t[i].calls)
            fprintf(f, "[host prof] %-44s %5ld calls: pack %.4f  device call %.4f  unpack %.4f  rest %.4f ms per call\n", t[i].name, t[i].calls,
                    t[i].ms[0] / t[i].calls, t[i].ms[1] / t[i].calls, t[i].ms[2] / t[i].calls, t[i].ms[3] / t[i].calls);
}

}  // namespace hip
}  // namespace ORB_SLAM3
