// orbhip_api.hip -- C-ABI implementation: context + ORB extractor (include/orbhip.h).
// Host logic mirrors ORBextractor's constructor / ComputePyramid bookkeeping
// (reference src/ORBextractor.cc:408-468, 1152-1177) and owns all device memory.
#include "orb_internal.h"
#include <cmath>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>
#include <set>
#include <utility>

static thread_local std::string g_last_error;
extern "C" const char *orbhip_last_error(void) { return g_last_error.c_str(); }
extern "C" const char *orbhip_version(void) { return "orbhip 0.1 (gfx950)"; }

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);           \
            return ORBHIP_E_HIP;                                                        \
        }                                                                               \
    } while (0)

int orb_lds_optin(const void *func, int device, size_t need)
{
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    const size_t total = 160 * 1024;
    std::lock_guard<std::mutex> lk(mu);
    if (!done.count({func, device})) {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, func) != hipSuccess) { g_last_error = "hipFuncGetAttributes"; return ORBHIP_E_HIP; }
        const int maxdyn = (int)(total - std::min(total, (size_t)fa.sharedSizeBytes));
        if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, maxdyn) != hipSuccess) { g_last_error = "hipFuncSetAttribute(MaxDynamicSharedMemorySize)"; return ORBHIP_E_HIP; }
        done.insert({func, device});
    }
    if (need > total) { g_last_error = "dynamic LDS request exceeds 160 KB"; return ORBHIP_E_BADARG; }
    return ORBHIP_OK;
}

struct orbhip_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    int32_t *d_status;      // sticky device-side error word (capacity overflows in matcher kernels)
    int32_t *status_redirect;
    void *scratch;          // grow-only device arena of the host-pointer convenience entry points (host_entry.hip)
    size_t scratch_bytes;
    void *work;             // grow-only device arena of the device entry points themselves (candidate lists of the low-latency matchers): separate from
    size_t work_bytes;      // `scratch`, which holds the host-form callers' staged inputs while those entry points run
    int n_cus;              // hipDeviceAttributeMultiprocessorCount of `device`: persistent grids are sized from it
    int ba_schur_mode;      // orbhip_ctx_set_ba_schur_mode
    void *ba_arena; size_t ba_arena_bytes; bool ba_arena_busy;     // cached device arena of one-shot local-BA solves (ba_kernels.hip)
    int *pinned_word;       // one page-locked word for such solves (the LM loop's active-graph counter)
    void *pinned;           // grow-only page-locked host arena of the host-pointer entry points: a call's inputs are gathered here and leave in ONE
    size_t pinned_bytes;    // host-to-device copy, its outputs come back in ONE device-to-host copy (host_entry.hip)
};

extern "C" int orbhip_ctx_create(int device, void *stream, orbhip_ctx **out)
{
    if (!out) return ORBHIP_E_BADARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0 || device < 0 || device >= n) {
        g_last_error = "no HIP device (this library has no CPU fallback)";
        return ORBHIP_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    orbhip_ctx *c = new orbhip_ctx();
    c->device = device;
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        hipError_t e2 = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e2 != hipSuccess) { delete c; g_last_error = hipGetErrorString(e2); return ORBHIP_E_HIP; }
        c->own_stream = true;
    }
    if (hipMalloc((void **)&c->d_status, sizeof(int32_t)) != hipSuccess || hipMemset(c->d_status, 0, sizeof(int32_t)) != hipSuccess) {
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c; g_last_error = "hipMalloc(status)"; return ORBHIP_E_HIP;
    }
    c->status_redirect = nullptr;
    c->scratch = nullptr; c->scratch_bytes = 0; c->work = nullptr; c->work_bytes = 0; c->ba_schur_mode = 0; c->n_cus = 0; c->pinned = nullptr; c->pinned_bytes = 0; c->ba_arena = nullptr; c->ba_arena_bytes = 0; c->ba_arena_busy = false; c->pinned_word = nullptr;
    if (hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->n_cus <= 0) c->n_cus = 256;
    *out = c;
    return ORBHIP_OK;
}
// Device arena of at least `bytes` (256-byte aligned), kept across calls; growing drains the stream first.
void *orbhip_ctx_scratch_internal(orbhip_ctx *c, size_t bytes)
{
    if (bytes <= c->scratch_bytes) return c->scratch;
    (void)hipStreamSynchronize(c->stream);
    if (c->scratch) (void)hipFree(c->scratch);
    c->scratch = nullptr; c->scratch_bytes = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 20);
    if (hipMalloc(&c->scratch, want) != hipSuccess) { g_last_error = "hipMalloc(scratch arena)"; return nullptr; }
    c->scratch_bytes = want;
    return c->scratch;
}
// Page-locked host arena of at least `bytes`, kept across calls (the host-pointer entry points are synchronous: nothing of an earlier call
// is in flight when the next one asks; growing drains the stream anyway).
void *orbhip_ctx_pinned_internal(orbhip_ctx *c, size_t bytes)
{
    if (bytes <= c->pinned_bytes) return c->pinned;
    (void)hipStreamSynchronize(c->stream);
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr; c->pinned_bytes = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 20);
    if (hipHostMalloc(&c->pinned, want, hipHostMallocDefault) != hipSuccess) { g_last_error = "hipHostMalloc(pinned staging arena)"; return nullptr; }
    c->pinned_bytes = want;
    return c->pinned;
}
// Cached device arena for one-shot BA solves (create + solve + download + destroy inside one call): nullptr when it is already lent
// out (the caller then allocates its own) or cannot be grown.
void *orbhip_ctx_ba_arena_acquire_internal(orbhip_ctx *c, size_t bytes)
{
    if (c->ba_arena_busy) return nullptr;
    if (bytes > c->ba_arena_bytes) {
        (void)hipStreamSynchronize(c->stream);
        if (c->ba_arena) (void)hipFree(c->ba_arena);
        c->ba_arena = nullptr; c->ba_arena_bytes = 0;
        const size_t want = bytes + bytes / 4;
        if (hipMalloc(&c->ba_arena, want) != hipSuccess) return nullptr;
        c->ba_arena_bytes = want;
    }
    c->ba_arena_busy = true;
    return c->ba_arena;
}
void orbhip_ctx_ba_arena_release_internal(orbhip_ctx *c) { c->ba_arena_busy = false; }
int *orbhip_ctx_pinned_word_internal(orbhip_ctx *c)
{
    if (!c->pinned_word && hipHostMalloc((void **)&c->pinned_word, 256, hipHostMallocDefault) != hipSuccess) c->pinned_word = nullptr;
    return c->pinned_word;
}
// Device work arena of at least `bytes`, kept across calls.  Growing waits for the stream (earlier kernels may still read the old arena).
void *orbhip_ctx_work_internal(orbhip_ctx *c, size_t bytes)
{
    if (bytes <= c->work_bytes) return c->work;
    (void)hipStreamSynchronize(c->stream);
    if (c->work) (void)hipFree(c->work);
    c->work = nullptr; c->work_bytes = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 20);
    if (hipMalloc(&c->work, want) != hipSuccess) { g_last_error = "hipMalloc(work arena)"; return nullptr; }
    c->work_bytes = want;
    return c->work;
}
void orbhip_set_last_error_internal(const char *msg) { g_last_error = msg; }
int32_t *orbhip_ctx_status_internal(orbhip_ctx *c);
extern "C" int orbhip_ctx_check_status(orbhip_ctx *c)
{
    if (!c) return ORBHIP_E_BADARG;
    int32_t st = 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(&st, orbhip_ctx_status_internal(c), sizeof(st), hipMemcpyDeviceToHost));
    if (st) { HIP_TRY(hipMemset(orbhip_ctx_status_internal(c), 0, sizeof(int32_t))); g_last_error = "device-side capacity exceeded in a matcher kernel"; }
    return st;
}
int32_t *orbhip_ctx_status_internal(orbhip_ctx *c) { return c->status_redirect ? c->status_redirect : c->d_status; }
// A host-pointer call (host_entry.hip) points the kernels' status word at a slot of its own blob for the duration of the call, so that the
// word travels with the call's single upload (zeroed) and single download instead of costing a copy of its own; nullptr restores
void orbhip_ctx_redirect_status_internal(orbhip_ctx *c, int32_t *p) { c->status_redirect = p; }
extern "C" void orbhip_ctx_destroy(orbhip_ctx *c)
{
    if (!c) return;
    (void)hipFree(c->d_status);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->work) (void)hipFree(c->work);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->ba_arena) (void)hipFree(c->ba_arena);
    if (c->pinned_word) (void)hipHostFree(c->pinned_word);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
extern "C" int orbhip_ctx_synchronize(orbhip_ctx *c)
{
    if (!c) return ORBHIP_E_BADARG;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ORBHIP_OK;
}
// Stream order across two contexts of one device: everything submitted to `other` so far completes before anything submitted to `c`
// after this call starts.  What a caller needs to run, say, the two matchers of a frame pair side by side on two contexts and to keep the
// next extraction from overwriting the arrays they still read.  No host synchronisation.
extern "C" int orbhip_ctx_wait_for(orbhip_ctx *c, orbhip_ctx *other)
{
    if (!c || !other) return ORBHIP_E_BADARG;
    if (c == other || c->stream == other->stream) return ORBHIP_OK;
    if (c->device != other->device) { g_last_error = "orbhip_ctx_wait_for: contexts of two devices"; return ORBHIP_E_BADARG; }
    HIP_TRY(hipSetDevice(c->device));
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev, other->stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, ev, 0));
    HIP_TRY(hipEventDestroy(ev));
    return ORBHIP_OK;
}
extern "C" void *orbhip_ctx_stream(orbhip_ctx *c) { return c ? (void *)c->stream : nullptr; }
hipStream_t orbhip_ctx_stream_internal(orbhip_ctx *c) { return c->stream; }
int orbhip_ctx_device_internal(orbhip_ctx *c) { return c->device; }
int orbhip_ctx_cus_internal(orbhip_ctx *c) { return c->n_cus; }
int orbhip_ctx_ba_schur_mode_internal(orbhip_ctx *c) { return c->ba_schur_mode; }
extern "C" int orbhip_ctx_set_ba_schur_mode(orbhip_ctx *c, int mode)
{
    if (!c || mode < 0 || mode > 2) return ORBHIP_E_BADARG;
    c->ba_schur_mode = mode;
    return ORBHIP_OK;
}

// ------------------------------------------------------------------------------------
static inline int cv_round_f(float v) { return (int)lrintf(v); }   // round-half-even
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

#define ORBHIP_PROF_SLOTS 32
struct orbhip_extractor {
    orbhip_ctx *ctx;
    int nfeatures, nlevels, ini_th, min_th;
    float scale_factor;
    float scale[ORB_MAX_LEVELS], inv_scale[ORB_MAX_LEVELS], sigma2[ORB_MAX_LEVELS], inv_sigma2[ORB_MAX_LEVELS];
    int per_level[ORB_MAX_LEVELS];
    int umax[ORB_HALF_PATCH + 1];
    int gauss_q8[7];
    // reserved geometry
    int width, height, max_batch;
    bool level0_owned;          // level-0 buffer allocated (host path / stride mismatch)
    OrbParams P;
    FastParams F;
    std::vector<void *> allocs;
    size_t bytes_reserved;
    int last_batch;
    bool profiling;
    hipEvent_t ev[ORBHIP_PROF_SLOTS][ORBHIP_STAGE_COUNT + 1];   // ring of per-call stage marks
    int ev_calls;               // extract calls recorded since the last stage_ms() query
    bool ev_created;
    float stage_ms[ORBHIP_STAGE_COUNT];
    uint8_t *d_level0;          // owned level-0 storage
    int32_t *d_stereo_sad;      // [max_batch][max_kp] scratch of orbhip_compute_stereo_matches_device (lazy)
    // graph mode (small batches are launch bound): the 19 launches of one extract call replayed as one hipGraph
    bool graph_mode, graph_valid;
    // k_blur (memory-wait bound) runs beside k_fast_cells / k_octree (issue / latency bound) on a second stream of the extractor
    hipStream_t aux; hipEvent_t ev_fork, ev_join, ev_pyr; bool aux_ok; int overlap;
    hipGraphExec_t graph_exec;
    int g_w, g_h, g_batch, g_lap0, g_lap1;
    size_t level0_frame_stride; int level0_pitch;
    // host-pointer path (ORBextractor::operator()): the four result arrays are slices of ONE device allocation (counts | mono | keypoints |
    // descriptors) with a page-locked mirror, so a call ends in one device-to-host copy; the input is gathered in a page-locked image
    uint8_t *d_outblob; size_t outblob_bytes, ob_kp, ob_desc;      // offsets of the keypoint / descriptor slices
    uint8_t *h_out; uint8_t *h_in; size_t h_in_bytes; int32_t *h_status;
    uint8_t *h_pyr; size_t h_pyr_bytes;                             // page-locked landing area of the lazy pyramid copy-out
    unsigned long long generation, view_generation;                 // extract calls so far; the call whose results h_out holds
};

extern "C" int orbhip_extractor_create(orbhip_ctx *ctx, int nfeatures, float scale_factor, int nlevels,
                                       int ini_th, int min_th, orbhip_extractor **out)
{
    if (!ctx || !out || nlevels < 1 || nlevels > ORB_MAX_LEVELS || nfeatures < 1 || !(scale_factor > 1.0f) || scale_factor > 2.0f)
        return ORBHIP_E_BADARG;     // k_resize stages the footprint of a 64x32 tile in LDS: scale factors up to 2
    orbhip_extractor *e = new orbhip_extractor();
    e->ctx = ctx; e->nfeatures = nfeatures; e->nlevels = nlevels; e->ini_th = ini_th; e->min_th = min_th;
    e->scale_factor = scale_factor;
    e->width = e->height = e->max_batch = 0; e->bytes_reserved = 0; e->last_batch = 0;
    e->profiling = false; e->ev_created = false; e->ev_calls = 0; e->d_level0 = nullptr; e->level0_owned = false;
    e->d_stereo_sad = nullptr;
    e->graph_mode = false; e->graph_valid = false; e->graph_exec = nullptr;
    e->aux = nullptr; e->ev_fork = e->ev_join = e->ev_pyr = nullptr; e->aux_ok = false; e->overlap = 1;
    e->d_outblob = nullptr; e->outblob_bytes = e->ob_kp = e->ob_desc = 0; e->h_out = e->h_in = e->h_pyr = nullptr; e->h_in_bytes = e->h_pyr_bytes = 0; e->h_status = nullptr;
    e->generation = 0; e->view_generation = ~0ull;
    memset(&e->P, 0, sizeof(e->P));
    memset(e->stage_ms, 0, sizeof(e->stage_ms));
    // scale tables, ORBextractor.cc:413-429
    e->scale[0] = 1.0f; e->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) { e->scale[i] = e->scale[i - 1] * scale_factor; e->sigma2[i] = e->scale[i] * e->scale[i]; }
    for (int i = 0; i < nlevels; i++) { e->inv_scale[i] = 1.0f / e->scale[i]; e->inv_sigma2[i] = 1.0f / e->sigma2[i]; }
    // per-level quota, ORBextractor.cc:433-445
    float factor = 1.0f / scale_factor;
    float desired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) { e->per_level[l] = cv_round_f(desired); sum += e->per_level[l]; desired *= factor; }
    e->per_level[nlevels - 1] = std::max(nfeatures - sum, 0);
    // umax, ORBextractor.cc:453-468
    int v, v0, vmax = (int)floor(ORB_HALF_PATCH * sqrtf(2.f) / 2 + 1), vmin = (int)ceil(ORB_HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = ORB_HALF_PATCH * ORB_HALF_PATCH;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = ORB_HALF_PATCH, v0 = 0; v >= vmin; --v) { while (e->umax[v0] == e->umax[v0 + 1]) ++v0; e->umax[v] = v0; ++v0; }
    // cv::getGaussianKernel(7, 2, CV_32F) -> q8
    {
        float cf[7]; double s = 0;
        for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)exp(-0.5 / 4.0 * x * x); s += cf[i]; }
        s = 1. / s;
        for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * s); e->gauss_q8[i] = cv_round_f(cf[i] * 256.f); }
    }
    *out = e;
    return ORBHIP_OK;
}

static void ext_free_all(orbhip_extractor *e)
{
    for (void *p : e->allocs) (void)hipFree(p);
    e->allocs.clear();
    e->bytes_reserved = 0; e->width = e->height = e->max_batch = 0; e->d_level0 = nullptr; e->d_stereo_sad = nullptr;
    if (e->h_out) (void)hipHostFree(e->h_out);
    if (e->h_in) (void)hipHostFree(e->h_in);
    if (e->h_pyr) (void)hipHostFree(e->h_pyr);
    if (e->h_status) (void)hipHostFree(e->h_status);
    e->h_out = e->h_in = e->h_pyr = nullptr; e->h_status = nullptr; e->h_in_bytes = e->h_pyr_bytes = 0; e->d_outblob = nullptr; e->view_generation = ~0ull;
    if (e->graph_valid) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_valid = false; }
}
static void ext_free_aux(orbhip_extractor *e)
{
    if (!e->aux_ok) return;
    (void)hipStreamSynchronize(e->aux);
    (void)hipEventDestroy(e->ev_fork); (void)hipEventDestroy(e->ev_join); (void)hipEventDestroy(e->ev_pyr); (void)hipStreamDestroy(e->aux);
    e->aux_ok = false;
}

extern "C" void orbhip_extractor_destroy(orbhip_extractor *e)
{
    if (!e) return;
    (void)hipStreamSynchronize(e->ctx->stream);
    ext_free_aux(e);
    ext_free_all(e);
    if (e->ev_created) for (auto &slot : e->ev) for (auto &ev : slot) (void)hipEventDestroy(ev);
    delete e;
}

extern "C" int orbhip_extractor_levels(const orbhip_extractor *e) { return e ? e->nlevels : ORBHIP_E_BADARG; }
extern "C" int orbhip_extractor_table(const orbhip_extractor *e, int which, float *out)
{
    if (!e || !out || which < 0 || which > 3) return ORBHIP_E_BADARG;
    const float *t = which == 0 ? e->scale : which == 1 ? e->inv_scale : which == 2 ? e->sigma2 : e->inv_sigma2;
    memcpy(out, t, sizeof(float) * e->nlevels);
    return ORBHIP_OK;
}
extern "C" int orbhip_extractor_features_per_level(const orbhip_extractor *e, int *out)
{
    if (!e || !out) return ORBHIP_E_BADARG;
    memcpy(out, e->per_level, sizeof(int) * e->nlevels);
    return ORBHIP_OK;
}
extern "C" int orbhip_extractor_umax(const orbhip_extractor *e, int *out16)
{
    if (!e || !out16) return ORBHIP_E_BADARG;
    memcpy(out16, e->umax, sizeof(int) * 16);
    return ORBHIP_OK;
}
extern "C" int orbhip_extractor_max_keypoints(const orbhip_extractor *e) { return e ? e->P.max_kp : ORBHIP_E_BADARG; }

template <typename T>
static int dev_alloc(orbhip_extractor *e, T **p, size_t count)
{
    void *q = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    hipError_t err = hipMalloc(&q, bytes);
    if (err != hipSuccess) { g_last_error = std::string("hipMalloc: ") + hipGetErrorString(err); return ORBHIP_E_HIP; }
    e->allocs.push_back(q);
    e->bytes_reserved += bytes;
    *p = (T *)q;
    return ORBHIP_OK;
}

// cv::resize INTER_LINEAR coefficient tables, host float math as OpenCV's.
static void resize_tables(int sn, int dn, bool horizontal, std::vector<int16_t> &ofs, std::vector<int16_t> &coef)
{
    double scale = 1. / ((double)dn / sn);
    ofs.resize(dn); coef.resize(2 * dn);
    for (int d = 0; d < dn; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (horizontal) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= sn - 1) { f = 0; s = sn - 1; }
        }
        int c0 = cv_round_f((1.f - f) * 2048), c1 = cv_round_f(f * 2048);
        ofs[d] = (int16_t)std::min(std::max(s, -32768), 32767);
        coef[2 * d] = (int16_t)std::min(std::max(c0, -32768), 32767);
        coef[2 * d + 1] = (int16_t)std::min(std::max(c1, -32768), 32767);
    }
}

extern "C" int orbhip_extractor_reserve(orbhip_extractor *e, int width, int height, int max_batch)
{
    if (!e || width <= 0 || height <= 0 || max_batch <= 0) return ORBHIP_E_BADARG;
    if (e->width == width && e->height == height && e->max_batch >= max_batch) return ORBHIP_OK;
    if (width > 4096 + 2 * ORB_MINB || height > 4096 + 2 * ORB_MINB) { g_last_error = "image larger than 12-bit key packing"; return ORBHIP_E_BADARG; }
    HIP_TRY(hipSetDevice(e->ctx->device));
    HIP_TRY(hipStreamSynchronize(e->ctx->stream));
    ext_free_all(e);
    OrbParams &P = e->P;
    memset(&P, 0, sizeof(P));
    P.nlevels = e->nlevels; P.ini_th = e->ini_th; P.min_th = e->min_th;
    memcpy(P.umax, e->umax, sizeof(P.umax));
    memcpy(P.gauss_q8, e->gauss_q8, sizeof(P.gauss_q8));
    int cells = 0, keys = 0, kps = 0, max_cell_cap = 0;
    for (int l = 0; l < e->nlevels; l++) {
        OrbLevel &L = P.lv[l];
        // ComputePyramid sizes, ORBextractor.cc:1156-1157
        L.w = cv_round_f((float)width * e->inv_scale[l]);
        L.h = cv_round_f((float)height * e->inv_scale[l]);
        // cell grid, ORBextractor.cc:771-781
        const float w_ = (float)((L.w - ORB_EDGE + 3) - ORB_MINB), h_ = (float)((L.h - ORB_EDGE + 3) - ORB_MINB);
        if (w_ < 30.f || h_ < 30.f) { g_last_error = "pyramid level smaller than one FAST cell"; return ORBHIP_E_BADARG; }
        L.ncols = (int)(w_ / 30.f); L.nrows = (int)(h_ / 30.f);
        L.wcell = (int)ceilf(w_ / L.ncols); L.hcell = (int)ceilf(h_ / L.nrows);
        if (L.wcell + 6 > 64 || L.hcell + 6 > 64) { g_last_error = "FAST cell exceeds the 64x64 LDS tile"; return ORBHIP_E_BADARG; }
        max_cell_cap = std::max(max_cell_cap, ((L.wcell + 1) / 2) * ((L.hcell + 1) / 2));
        L.cell_base = cells; cells += L.ncols * L.nrows;
        L.quota = e->per_level[l];
        // DistributeOctTree roots, ORBextractor.cc:541-543
        const int maxx = L.w - ORB_MINB, maxy = L.h - ORB_MINB;
        L.n_ini = (int)roundf((float)(maxx - ORB_MINB) / (maxy - ORB_MINB));
        if (L.n_ini < 1) { g_last_error = "nIni == 0 (image taller than 2x its width): undefined in the reference"; return ORBHIP_E_BADARG; }
        L.hx = (float)(maxx - ORB_MINB) / L.n_ini;
        L.scale = e->scale[l];
        L.size = (float)(int)(31 * e->scale[l]);              // ORBextractor.cc:862
        L.kp_base = kps; L.kp_cap = align_up(std::max(L.quota + 8, 4 * L.n_ini + 4), 4); kps += L.kp_cap;   // multiple of 4: k_orient_desc
    }
    // cell list capacity: 3x3 strict-greater NMS admits at most one keypoint per 2x2 block
    for (int l = 0; l < e->nlevels; l++) {
        OrbLevel &L = P.lv[l];
        L.cell_cap = max_cell_cap;
        L.key_base = keys;
        // candidates per level: bounded by the same 2x2 argument over the whole detection area
        L.key_cap = std::min(L.ncols * L.nrows * max_cell_cap, ((L.w - 2 * ORB_MINB + 1) / 2 + L.ncols) * ((L.h - 2 * ORB_MINB + 1) / 2 + L.nrows));
        L.key_cap = std::min(L.key_cap, 0xFFFFF);
        keys += align_up(L.key_cap, 4);
    }
    P.cells_per_frame = cells; P.keys_per_frame = keys; P.kps_per_frame = kps; P.max_kp = align_up(kps, 8);
    {   // k_fast_cells: compact parameter block + per-wave LDS geometry from the largest cell of this extractor
        FastParams &F = e->F;
        memset(&F, 0, sizeof(F));
        int mw = 0, mh = 0;
        bool small = true;
        for (int l = 0; l < e->nlevels; l++) {
            const OrbLevel &L = P.lv[l];
            FcLevel &f = F.lv[l];
            mw = std::max(mw, L.wcell); mh = std::max(mh, L.hcell);
            f.max_bx = L.w - ORB_MINB; f.max_by = L.h - ORB_MINB;
            f.ncols = L.ncols; f.nrows = L.nrows; f.wcell = L.wcell; f.hcell = L.hcell; f.cell_base = L.cell_base; f.cell_cap = L.cell_cap;
            const int npb = (L.wcell + 1) >> 1, nq = (npb + 5) >> 1;       // band pixel pairs per row; quads staged per row (pairs 0 .. npb+3)
            f.tpr = (npb + 1) >> 1; f.rpt = 64 / f.tpr;
            (void)nq;
            small = small && npb + 4 <= 24 && npb + 2 <= 24 && (L.hcell + 6) * 3 <= 64 * 2;       // <3, 24, 2>: 24 pairs per row, 128 chunk tasks
        }
        {   // k_fast_runs: a wave works on a RUN of cells of one cell row -- two while both fit the 32 pixel-pair columns of a half wave
            int runs = 0;
            bool ok = true;
            for (int l = 0; l < e->nlevels; l++) {
                const OrbLevel &L = P.lv[l];
                FcLevel &f = F.lv[l];
                f.rpc = 2 * L.wcell <= 64 ? 2 : 1;
                f.rpr = (L.ncols + f.rpc - 1) / f.rpc;
                f.run_base = runs; runs += f.rpr * L.nrows;
                ok = ok && L.wcell <= 64 && L.hcell + 6 <= 70 && L.wcell >= 8 && L.hcell >= 8;
            }
            const int hh = (mh + 1) / 2;
            F.runs_per_frame = runs;
            F.run_rows = mh + 6 + 7;                                               // + the rows a seven-row trip of the column walk may read past the band
            F.run_q0 = (hh * 32 + 7) & ~7;                                         // first queue: the upper half of the rows (u16 entries)
            F.run_dw = F.run_rows * 40 + (((mh + 2) * 40 + 2 * F.run_q0) + 1) / 2;   // pair tile + u16 score tile + u16 queues
            F.run_dw = (F.run_dw + 3) & ~3;
            // OPT-IN (ORBHIP_FAST_KERNEL=runs when the extractor reserves): bit-exact (every tests/test_gpu_orb.py test with it forced), but measured
            // SLOWER than k_fast_cells -- 1.72 ms (row-at-a-time ring) / 2.43 ms (seven-row trips) against 1.49 ms per 1024 VGA frames
            F.use_runs = ok && getenv("ORBHIP_FAST_KERNEL") && !strcmp(getenv("ORBHIP_FAST_KERNEL"), "runs") ? 1 : 0;
        }
        const int npb = (mw + 1) >> 1;
        F.small_cells = small ? 1 : 0;
        F.rows = mh + 6; F.srows = mh; F.qcap = (mh * npb + 1) & ~1;
        F.wave_dw = F.rows * (small ? 24 : 40) + ((F.srows + 2) * (small ? 24 : 32) + F.qcap + 1) / 2;      // pair tile (dwords) + u16 score tile + u16 queue
        F.wave_dw = (F.wave_dw + 3) & ~3;
        F.nlevels = e->nlevels; F.ini_th = e->ini_th; F.min_th = e->min_th; F.cells_per_frame = cells;
        for (int n = 1; n < 34; n++) F.div_magic[n] = (uint32_t)(65536 / n + 1);
    }
    P.bs_tiles[0] = 0;
    for (int l = 0; l < e->nlevels; l++) P.bs_tiles[l + 1] = P.bs_tiles[l] + ((P.lv[l].w + 63) / 64) * ((P.lv[l].h + 31) / 32);
    P.rows_min_batch = getenv("ORBHIP_ROWS_MIN_BATCH") ? atoi(getenv("ORBHIP_ROWS_MIN_BATCH")) : 16;     // measured: 1 frame 122 us (tiles) vs 158 us (rows), 32 frames 252 vs 222 us
    {   // k_blur_rows: lanes = 4-column chunks x bands of BLR_R rows; needs BLR_R + 4 rows for its reflected row indices and one interior chunk
        bool ok = !getenv("ORBHIP_BLUR_TILES");
        P.br_blocks[0] = 0;
        for (int l = 0; l < e->nlevels; l++) {
            const int w = P.lv[l].w, h = P.lv[l].h;
            if (h < BLR_R + 4 || w < 16) ok = false;
            P.br_blocks[l + 1] = P.br_blocks[l] + (((w + 3) / 4) * ((h + BLR_R - 1) / BLR_R) + 255) / 256;
        }
        if (!ok) P.br_blocks[e->nlevels] = 0;
    }
    P.cell_list_frame_stride = (size_t)cells * max_cell_cap;
    int rc;
    const size_t B = (size_t)max_batch;
    for (int l = 0; l < e->nlevels; l++) {
        OrbLevel &L = P.lv[l];
        L.img_pitch = align_up(L.w, 64); L.blur_pitch = L.img_pitch;
        L.img_frame_stride = (size_t)L.img_pitch * L.h; L.blur_frame_stride = L.img_frame_stride;
        if ((rc = dev_alloc(e, &L.img, B * L.img_frame_stride + 64))) return rc;
        if ((rc = dev_alloc(e, &L.blur, B * L.blur_frame_stride + 64))) return rc;
        if (l > 0) {
            std::vector<int16_t> xo, xa, yo, yb;
            resize_tables(P.lv[l - 1].w, L.w, true, xo, xa);
            resize_tables(P.lv[l - 1].h, L.h, false, yo, yb);
            int16_t *dxo, *dxa, *dyo, *dyb;
            if ((rc = dev_alloc(e, &dxo, xo.size()))) return rc;
            if ((rc = dev_alloc(e, &dxa, xa.size()))) return rc;
            if ((rc = dev_alloc(e, &dyo, yo.size()))) return rc;
            if ((rc = dev_alloc(e, &dyb, yb.size()))) return rc;
            HIP_TRY(hipMemcpy(dxo, xo.data(), xo.size() * 2, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(dxa, xa.data(), xa.size() * 2, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(dyo, yo.data(), yo.size() * 2, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(dyb, yb.data(), yb.size() * 2, hipMemcpyHostToDevice));
            L.xofs = dxo; L.xalpha = dxa; L.yofs = dyo; L.ybeta = dyb;
            // k_resize_rows: per 4 output columns the 8 source bytes they read (base clamped into the row), v_perm selectors that
            // put (S[sx], S[sx+1]) into the two halves of a dword, and the coefficient pairs; per output row the two clamped
            // source rows and the vertical coefficients pre-shifted for v_mul_hi_u32_u24
            const int sw = P.lv[l - 1].w, sh = P.lv[l - 1].h, nch = (L.w + 3) / 4;
            std::vector<uint32_t> xc((size_t)nch * 12, 0), yt((size_t)L.h * 4);
            bool fits = sw >= 8;
            const char *rm = getenv("ORBHIP_RESIZE_MODE");
            int mode = l > 1 || (sw % 4 == 0 && sw >= 12) ? 1 : 0;       // level 0 may alias the caller's images: never read past a row there
            if (rm && atoi(rm) == 0) mode = 0;
            for (int c = 0; c < nch && fits; c++) {
                const int base = std::min((int)xo[4 * c], sw - 8);
                // the aligned variant loads 12 bytes from base4 = base & ~3 and shifts them back with v_perm; level 0 may be the
                // caller's buffer, so there base4 is lowered where the third dword would leave the row (shift 4, width % 4 == 0)
                const int base4 = l > 1 ? (base & ~3) : std::min(base & ~3, sw - 12);
                xc[12 * c] = mode ? (uint32_t)base4 : (uint32_t)base;
                xc[12 * c + 9] = 0x03020100u + 0x01010101u * (uint32_t)(base - base4);
                for (int j = 0; j < 4; j++) {
                    const int col = std::min(4 * c + j, L.w - 1);
                    const int i0 = xo[col] - base, a0 = xa[2 * col], a1 = xa[2 * col + 1];
                    int i1 = i0 + 1;
                    if (i0 < 0 || i0 > 7 || a0 < 0 || a1 < 0) { fits = false; break; }
                    if (i1 > 7) { if (a1 != 0) { fits = false; break; } i1 = i0; }
                    xc[12 * c + 1 + j] = (uint32_t)i0 | 0x0c00u | ((uint32_t)i1 << 16) | 0x0c000000u;
                    xc[12 * c + 5 + j] = (uint32_t)a0 | ((uint32_t)a1 << 16);
                }
            }
            for (int dy = 0; dy < L.h; dy++) {
                const int r = yo[dy];
                yt[4 * dy] = (uint32_t)std::min(std::max(r, 0), sh - 1);
                yt[4 * dy + 1] = (uint32_t)std::min(std::max(r + 1, 0), sh - 1);
                if (yb[2 * dy] < 0 || yb[2 * dy + 1] < 0 || yb[2 * dy] > 2048 || yb[2 * dy + 1] > 2048) fits = false;
                yt[4 * dy + 2] = (uint32_t)yb[2 * dy] << 12;
                yt[4 * dy + 3] = (uint32_t)yb[2 * dy + 1] << 12;
            }
            L.xchunk = nullptr; L.ytab = nullptr;
            if (fits && !getenv("ORBHIP_RESIZE_TILES")) {
                uint32_t *dxc, *dyt;
                if ((rc = dev_alloc(e, &dxc, xc.size()))) return rc;
                if ((rc = dev_alloc(e, &dyt, yt.size()))) return rc;
                HIP_TRY(hipMemcpy(dxc, xc.data(), xc.size() * 4, hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(dyt, yt.data(), yt.size() * 4, hipMemcpyHostToDevice));
                L.xchunk = dxc; L.ytab = dyt;
                L.resize_mode = mode;
            }
        }
    }
    e->d_level0 = P.lv[0].img; e->level0_pitch = P.lv[0].img_pitch; e->level0_frame_stride = P.lv[0].img_frame_stride;
    {   // k_blur_mfma operand tables (orb_kernels.hip): horizontal band per (level, 32-column tile column, 16-column block) with the image
        // borders folded in and the kernel's chunk rule mirrored; vertical band per 16-row output block in the slot order the accumulator
        // layout of pass 1 dictates.  Not used (k_blur_rows stays) for levels narrower than 32 or shorter than 8.
        // Round 4: the DEFAULT for images of up to 320 K pixels (VGA) in batches of 128 frames or more -- where the blur runs beside
        // k_fast_cells and the pair is bound by the sum of their vector instructions, the matrix-core form needs ~6 per pixel against 14
        // (measured after k_fast_cells lost 9 % of its own: 3.67 against 3.73 ms per 1024-frame VGA step, two A/B pairs; in round 3 the
        // two were even).  Alone it is the slower kernel (0.93 against 0.75 ms), and at 1080p / 4K, where the blur is a smaller share of
        // a step that is bound elsewhere, 12 % slower in the step: k_blur_rows stays for small batches and big images.
        // ORBHIP_BLUR_MFMA=1: every batch that takes the row-streaming kernels, any size; =0: never.  Bit-exact either way.
        const char *bm_env = getenv("ORBHIP_BLUR_MFMA");
        const bool bm_force = bm_env && atoi(bm_env) == 1, bm_off = bm_env && atoi(bm_env) == 0;
        // (crossover measured in one call per pair, frames/s matrix cores / rows: 640x480 x 1024: 278 k / 270 k, x 128: 184.3 k / 181.4 k;
        //  752x480 x 1024: 247.8 k / 249.7 k; 1280x720 x 512: 98.9 k / 104.0 k; 1920x1080 x 512: 44.2 k / 50.1 k)
        bool ok = !bm_off && (bm_force || (size_t)width * height <= (size_t)320 * 1024) && P.br_blocks[e->nlevels] > 0;
        P.bm_min_batch = bm_force ? P.rows_min_batch : 128;
        int S = 0;
        for (int i = 0; i < 7; i++) { S += e->gauss_q8[i]; if (e->gauss_q8[i] < 0 || e->gauss_q8[i] > 63) ok = false; }     // two folded taps must fit int8
        if (255 * S > 65535) ok = false;                                                                                  // row sums must fit 16 bits
        P.bm_cols[0] = 0;
        for (int l = 0; l < e->nlevels; l++) {
            if (P.lv[l].w < 32 || P.lv[l].h < 8) ok = false;
            P.bm_cols[l + 1] = P.bm_cols[l] + (P.lv[l].w + 31) / 32;
        }
        auto refl = [](int x, int n) { return x < 0 ? -x : (x >= n ? 2 * n - 2 - x : x); };
        std::vector<int8_t> th((size_t)P.bm_cols[e->nlevels] * 2 * 64 * 16, 0), tv((size_t)4 * 64 * 16, 0);
        for (int l = 0; l < e->nlevels && ok; l++) {
            const int w = P.lv[l].w;
            for (int tx = 0; tx < (w + 31) / 32; tx++) {
                int cxs[4];
                for (int g = 0; g < 4; g++) { int cx = 32 * tx - 16 + 16 * g; cx = cx < 0 ? 0 : cx; if (cx + 16 > w) cx = w - 16; cxs[g] = cx; }
                for (int cb = 0; cb < 2; cb++)
                    for (int n = 0; n < 16; n++) {
                        const int xo = 32 * tx + 16 * cb + n;
                        if (xo >= w) continue;
                        for (int i = -3; i <= 3; i++) {
                            const int col = refl(xo + i, w);
                            int gp = -1;                                         // the first chunk that holds the column takes its tap
                            for (int g = 0; g < 4 && gp < 0; g++) if (col >= cxs[g] && col < cxs[g] + 16) gp = g;
                            if (gp < 0) { ok = false; break; }
                            int8_t &c = th[((((size_t)(P.bm_cols[l] + tx) * 2 + cb) * 64) + (size_t)(n + 16 * gp)) * 16 + (col - cxs[gp])];
                            c = (int8_t)(c + e->gauss_q8[i + 3]);
                        }
                    }
            }
        }
        for (int b = 0; b < 4; b++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 16; j++) {
                    const int n = lane & 15, g = lane >> 4, krow = 16 * (j >> 2) + 4 * g + (j & 3), d = krow - (16 * b + n);
                    tv[((size_t)b * 64 + lane) * 16 + j] = (int8_t)((d >= 0 && d <= 6) ? e->gauss_q8[d] : 0);
                }
        P.bm_th = nullptr; P.bm_tv = nullptr; P.bm_init = 128 * S;
        if (ok) {
            uint4 *dth, *dtv;
            if ((rc = dev_alloc(e, &dth, th.size() / 16))) return rc;
            if ((rc = dev_alloc(e, &dtv, tv.size() / 16))) return rc;
            HIP_TRY(hipMemcpy(dth, th.data(), th.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(dtv, tv.data(), tv.size(), hipMemcpyHostToDevice));
            P.bm_th = dth; P.bm_tv = dtv;
        } else P.bm_cols[e->nlevels] = 0;
    }
    if ((rc = dev_alloc(e, &P.cell_count, B * cells))) return rc;
    if ((rc = dev_alloc(e, &P.cell_list, B * P.cell_list_frame_stride))) return rc;
    if ((rc = dev_alloc(e, &P.keys, B * keys))) return rc;
    if ((rc = dev_alloc(e, &P.node_of, B * keys))) return rc;
    if ((rc = dev_alloc(e, &P.lvl_kp, B * kps))) return rc;
    if ((rc = dev_alloc(e, &P.lvl_perm, B * kps))) return rc;
    if ((rc = dev_alloc(e, &P.lvl_angle, B * kps))) return rc;
    if ((rc = dev_alloc(e, &P.lvl_desc, B * kps * 32))) return rc;
    if ((rc = dev_alloc(e, &P.lvl_count, B * e->nlevels))) return rc;
    if ((rc = dev_alloc(e, &P.lvl_ncand, B * e->nlevels))) return rc;
    if ((rc = dev_alloc(e, &P.status, 1))) return rc;
    {   // results: one allocation, [count B][mono B] | keypoints [B][max_kp] | descriptors [B][max_kp][32], slices 256-byte aligned
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        e->ob_kp = up(8 * B); e->ob_desc = e->ob_kp + up(sizeof(orbhip_keypoint) * B * P.max_kp);
        e->outblob_bytes = e->ob_desc + up((size_t)32 * B * P.max_kp);
        if ((rc = dev_alloc(e, &e->d_outblob, e->outblob_bytes))) return rc;
        P.out_count = reinterpret_cast<int32_t *>(e->d_outblob); P.out_mono = P.out_count + B;
        P.out_kp = reinterpret_cast<orbhip_keypoint *>(e->d_outblob + e->ob_kp); P.out_desc = e->d_outblob + e->ob_desc;
    }
    HIP_TRY(hipMemset(P.status, 0, sizeof(int32_t)));
    // the octree kernel keeps its node arrays in LDS: capacity = the largest node count any level can reach -- quota + 3 in the
    // subdivision loop, but the first pass splits every root unconditionally (up to 4 * nIni nodes; wide images, small budgets)
    P.oct_nc = orb_octree_nc(P);
    if (orb_octree_lds_bytes(P.oct_nc) > 150 * 1024) { g_last_error = "per-level quota (or nIni) too large for the LDS-resident octree"; return ORBHIP_E_BADARG; }
    if ((rc = orb_lds_optin(orb_octree_func(), e->ctx->device, orb_octree_lds_bytes(P.oct_nc)))) return rc;
    if ((rc = orb_lds_optin(orb_fast_cells_func(e->F.small_cells), e->ctx->device, sizeof(uint32_t) * (size_t)e->F.wave_dw))) return rc;
    if (e->F.use_runs && (rc = orb_lds_optin(orb_fast_runs_func((e->F.run_rows - 7) * 5 <= 256 ? 4 : 6), e->ctx->device, sizeof(uint32_t) * (size_t)e->F.run_dw))) return rc;
    e->width = width; e->height = height; e->max_batch = max_batch;
    return ORBHIP_OK;
}

// Frame::ComputeStereoMatches (reference src/Frame.cc:802-980) on the results of the latest extract call of a
// left and a right extractor (same geometry, same batch); frame f of `left` pairs with frame f of `right`.
extern "C" int orbhip_compute_stereo_matches_device(orbhip_extractor *left, orbhip_extractor *right, float mb, float mbf,
                                                    float *d_u_right, float *d_depth, int32_t *d_n_matches)
{
    if (!left || !right || !d_u_right || !d_depth || !(mb > 0) || !(mbf > 0)) return ORBHIP_E_BADARG;
    if (!left->max_batch || !right->max_batch || left->last_batch <= 0 || left->last_batch != right->last_batch ||
        left->width != right->width || left->height != right->height || left->nlevels != right->nlevels ||
        left->scale_factor != right->scale_factor || left->P.max_kp != right->P.max_kp || left->P.max_kp > 65535) {
        g_last_error = "stereo: extractors must share geometry, feature budget and batch, and have extracted";
        return ORBHIP_E_BADARG;
    }
    HIP_TRY(hipSetDevice(left->ctx->device));
    if (!left->d_stereo_sad) {
        int rc = dev_alloc(left, &left->d_stereo_sad, (size_t)left->max_batch * left->P.max_kp);
        if (rc) return rc;
    }
    if (right->ctx->stream != left->ctx->stream) {          // the right extractor's kernels must have finished
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev, right->ctx->stream));
        HIP_TRY(hipStreamWaitEvent(left->ctx->stream, ev, 0));
        HIP_TRY(hipEventDestroy(ev));
    }
    StereoArgs A;
    memset(&A, 0, sizeof(A));
    for (int l = 0; l < left->nlevels; l++) {
        const OrbLevel &a = left->P.lv[l], &b = right->P.lv[l];
        A.lv[l].imgL = a.img; A.lv[l].imgR = b.img; A.lv[l].fsL = a.img_frame_stride; A.lv[l].fsR = b.img_frame_stride;
        A.lv[l].pitchL = a.img_pitch; A.lv[l].pitchR = b.img_pitch; A.lv[l].wR = b.w;
        A.lv[l].scale = left->scale[l]; A.lv[l].inv_scale = left->inv_scale[l];
    }
    A.nlevels = left->nlevels; A.batch = left->last_batch; A.max_kp = left->P.max_kp; A.rows0 = left->height;
    A.kpL = left->P.out_kp; A.kpR = right->P.out_kp; A.descL = left->P.out_desc; A.descR = right->P.out_desc;
    A.nL = left->P.out_count; A.nR = right->P.out_count;
    A.mb = mb; A.mbf = mbf; A.u_right = d_u_right; A.depth = d_depth; A.sad = left->d_stereo_sad; A.n_kept = d_n_matches;
    orb_launch_stereo(A, left->ctx->stream);
    HIP_TRY(hipGetLastError());
    if (right->ctx->stream != left->ctx->stream) {
        // ... and the right extractor's NEXT extraction must not overwrite its pyramid / keypoints / descriptors while the stereo
        // kernels on the left stream still read them (write-after-read): its stream waits for them
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev, left->ctx->stream));
        HIP_TRY(hipStreamWaitEvent(right->ctx->stream, ev, 0));
        HIP_TRY(hipEventDestroy(ev));
    }
    return ORBHIP_OK;
}

// Host-pointer form for ONE stereo frame: what Frame::ComputeStereoMatches() of host/Frame.cc calls after the two extractors have run
// (src/Frame.cc:109-130).  mvuRight / mvDepth of frame 0 of the latest extractions, n = its left keypoint count.
extern "C" int orbhip_compute_stereo_matches_host(orbhip_extractor *left, orbhip_extractor *right, float mb, float mbf,
                                                  float *u_right_out, float *depth_out, int n, int32_t *n_matches_out)
{
    if (!left || !right || n < 0 || (n && (!u_right_out || !depth_out))) return ORBHIP_E_BADARG;
    if (n_matches_out) *n_matches_out = 0;
    for (int i = 0; i < n; i++) { u_right_out[i] = -1.0f; depth_out[i] = -1.0f; }      // Frame.cc:804-805
    if (n == 0) return ORBHIP_OK;
    if (left->last_batch <= 0 || n > left->P.max_kp) { g_last_error = "stereo: the left extractor holds no extraction of that many keypoints"; return ORBHIP_E_BADARG; }
    const size_t slots = (size_t)left->max_batch * left->P.max_kp;
    uint8_t *w = (uint8_t *)orbhip_ctx_work_internal(left->ctx, 8 * slots + 4 * (size_t)left->max_batch + 256);
    if (!w) return ORBHIP_E_HIP;
    float *d_ur = (float *)w, *d_dp = d_ur + slots;
    int32_t *d_nk = (int32_t *)(d_dp + slots);
    const int rc = orbhip_compute_stereo_matches_device(left, right, mb, mbf, d_ur, d_dp, d_nk);
    if (rc) return rc;
    uint8_t *h = (uint8_t *)orbhip_ctx_pinned_internal(left->ctx, 8 * (size_t)n + 16);
    if (!h) return ORBHIP_E_HIP;
    hipStream_t st = left->ctx->stream;
    HIP_TRY(hipMemcpyAsync(h, d_ur, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h + 4 * (size_t)n, d_dp, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h + 8 * (size_t)n, d_nk, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(u_right_out, h, 4 * (size_t)n); memcpy(depth_out, h + 4 * (size_t)n, 4 * (size_t)n);
    if (n_matches_out) memcpy(n_matches_out, h + 8 * (size_t)n, 4);
    return ORBHIP_OK;
}

// which kernel blurs a batch of that many frames: 0 = the LDS tile kernel, 1 = k_blur_rows, 2 = k_blur_mfma (bench.py labels its stage with it)
extern "C" int orbhip_extractor_blur_kernel(const orbhip_extractor *e, int batch)
{
    if (!e || !e->max_batch) return -1;
    const OrbParams &P = e->P;
    if (P.bm_cols[e->nlevels] > 0 && batch >= P.bm_min_batch) return 2;
    if (P.br_blocks[e->nlevels] > 0 && batch >= P.rows_min_batch) return 1;
    return 0;
}

extern "C" int orbhip_extractor_set_profiling(orbhip_extractor *e, int enable)
{
    if (!e) return ORBHIP_E_BADARG;
    if (enable && !e->ev_created) {
        for (auto &slot : e->ev) for (auto &ev : slot) HIP_TRY(hipEventCreate(&ev));
        e->ev_created = true;
    }
    e->profiling = enable != 0;
    e->ev_calls = 0;
    return ORBHIP_OK;
}

extern "C" int orbhip_extractor_stage_ms(orbhip_extractor *e, float *ms_out)
{
    if (!e || !ms_out || !e->ev_created || e->ev_calls <= 0) return ORBHIP_E_BADARG;
    // average over the (up to ORBHIP_PROF_SLOTS) most recent extract calls since the last query
    const int n = std::min(e->ev_calls, ORBHIP_PROF_SLOTS);
    for (int i = 0; i < ORBHIP_STAGE_COUNT; i++) ms_out[i] = 0.f;
    for (int c = 0; c < n; c++) {
        const int slot = (e->ev_calls - 1 - c) % ORBHIP_PROF_SLOTS;
        HIP_TRY(hipEventSynchronize(e->ev[slot][ORBHIP_STAGE_COUNT]));
        for (int i = 0; i < ORBHIP_STAGE_COUNT; i++) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, e->ev[slot][i], e->ev[slot][i + 1]));
            ms_out[i] += ms / n;
        }
    }
    e->ev_calls = 0;
    return ORBHIP_OK;
}

static int tune_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }      // tuning hook (development)

static int run_pipeline(orbhip_extractor *e, int batch, int lap0, int lap1)
{
    OrbParams &P = e->P;
    P.batch = batch; P.lap0 = lap0; P.lap1 = lap1; P.n_cus = e->ctx->n_cus;
    hipStream_t s = e->ctx->stream;
    const bool prof = e->profiling;
    const int slot = e->ev_calls % ORBHIP_PROF_SLOTS;
#define STAGE_MARK(i) do { if (prof) HIP_TRY(hipEventRecord(e->ev[slot][i], s)); } while (0)
    // fork: two streams inside one call.  Stage profiling measures the kernels one by one on one stream.
    const bool fork = e->overlap && !prof && batch >= ORB_OVERLAP_MIN_BATCH;
    if (fork && !e->aux_ok) {
        HIP_TRY(hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_pyr, hipEventDisableTiming));
        e->aux_ok = true;
    }
    FastParams &F = e->F;
    for (int l = 0; l < e->nlevels; l++) { F.lv[l].img = P.lv[l].img; F.lv[l].frame_stride = P.lv[l].img_frame_stride; F.lv[l].img_pitch = P.lv[l].img_pitch; }
    F.batch = batch; F.n_cus = e->ctx->n_cus; F.cell_count = P.cell_count; F.cell_list = P.cell_list; F.cell_list_frame_stride = P.cell_list_frame_stride; F.status = P.status;
    const bool rows = P.br_blocks[e->nlevels] > 0 && batch >= P.rows_min_batch;      // the row-streaming blur needs no LDS: k_fast_cells keeps its full grid
    // waves of k_fast_cells per CU while the blur runs beside it: 10 next to the LDS tile blur; 18 (of the 23 the LDS would hold) next to the
    // row-streaming / matrix-core kernels, which need wave slots and registers, not LDS -- measured in one call on one box (round 4, 1024 VGA
    // frames, matrix-core blur): 23 waves 3.79 ms per step, 22: 3.84, 20: 3.72, 18: 3.69, 16: 3.74, 14: 3.80; 1080p / 512 frames
    // (row-streaming blur): 49.2 k -> 50.7 k frames/s; VGA with the row-streaming blur: 3.82 either way
    const int fast_waves = fork ? tune_int("ORBHIP_TUNE_FAST_WAVES", rows ? 18 : 10) : 0;
    STAGE_MARK(ORBHIP_STAGE_PYRAMID);
    if (fork && e->nlevels > 1 && rows && tune_int("ORBHIP_TUNE_L0_EARLY", 0)) {
        // (off by default: measured 3.90 ms vs 3.84 ms per step)
        // Level 0 is the input itself: its FAST cells (a third of all pixels) start at once on the main stream, the pyramid is built
        // beside them on the second stream (k_resize_rows needs a few KB of LDS, k_fast_cells leaves half of the wave slots free and
        // ~40 % of the issue cycles idle); levels 1.. follow when the pyramid is there, the blur runs beside them.
        HIP_TRY(hipEventRecord(e->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(e->aux, e->ev_fork, 0));
        orb_launch_fast_cells(F, s, fast_waves, 0, 1);
        for (int l = 1; l < e->nlevels; l++) orb_launch_resize(P, l, e->aux);
        HIP_TRY(hipEventRecord(e->ev_pyr, e->aux));
        HIP_TRY(hipStreamWaitEvent(s, e->ev_pyr, 0));
        orb_launch_fast_cells(F, s, fast_waves, 1, e->nlevels);
        orb_launch_blur(P, e->aux, 5);
        HIP_TRY(hipEventRecord(e->ev_join, e->aux));
    } else {
        for (int l = 1; l < e->nlevels; l++) orb_launch_resize(P, l, s);
        STAGE_MARK(ORBHIP_STAGE_FAST_CELLS);
        if (fork) {                                         // the blur only needs the pyramid; it joins before the descriptors
            HIP_TRY(hipEventRecord(e->ev_fork, s));
            HIP_TRY(hipStreamWaitEvent(e->aux, e->ev_fork, 0));
        }
        orb_launch_fast_cells(F, s, fast_waves);
        if (fork) {                                         // after k_fast_cells on purpose: its waves take their LDS first, the blur fills the free wave slots
            if (!tune_int("ORBHIP_TUNE_SKIP_BLUR", 0))       // (timing experiment only: descriptors are wrong without the blur)
                orb_launch_blur(P, e->aux, tune_int("ORBHIP_TUNE_BLUR_WGS", 5));
            HIP_TRY(hipEventRecord(e->ev_join, e->aux));
        }
    }
    STAGE_MARK(ORBHIP_STAGE_BLUR);
    if (!fork) orb_launch_blur(P, s, 8);
    STAGE_MARK(ORBHIP_STAGE_OCTREE);
    orb_launch_octree(P, s);
    STAGE_MARK(ORBHIP_STAGE_DESC);
    if (fork) HIP_TRY(hipStreamWaitEvent(s, e->ev_join, 0));
    orb_launch_orient_desc(P, s);
    STAGE_MARK(ORBHIP_STAGE_ASSEMBLE);
    orb_launch_assemble(P, s);
    STAGE_MARK(ORBHIP_STAGE_COUNT);
#undef STAGE_MARK
    if (prof) e->ev_calls++;
    HIP_TRY(hipGetLastError());
    e->last_batch = batch;
    e->generation++;
    return ORBHIP_OK;
}

// Graph mode: capture the launches of run_pipeline once per (size, batch, lapping), then replay.  The caller has
// staged the input into the extractor's own level-0 buffer (fixed addresses).
static int run_pipeline_graph(orbhip_extractor *e, int width, int height, int batch, int lap0, int lap1)
{
    if (!(e->graph_valid && e->g_w == width && e->g_h == height && e->g_batch == batch && e->g_lap0 == lap0 && e->g_lap1 == lap1)) {
        if (e->graph_valid) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_valid = false; }
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(e->ctx->stream, hipStreamCaptureModeThreadLocal));
        const int rc = run_pipeline(e, batch, lap0, lap1);
        hipError_t ce = hipStreamEndCapture(e->ctx->stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ce != hipSuccess) { g_last_error = std::string("hipStreamEndCapture: ") + hipGetErrorString(ce); return ORBHIP_E_HIP; }
        hipError_t ie = hipGraphInstantiate(&e->graph_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) { g_last_error = std::string("hipGraphInstantiate: ") + hipGetErrorString(ie); return ORBHIP_E_HIP; }
        e->graph_valid = true; e->g_w = width; e->g_h = height; e->g_batch = batch; e->g_lap0 = lap0; e->g_lap1 = lap1;
    }
    HIP_TRY(hipGraphLaunch(e->graph_exec, e->ctx->stream));
    e->last_batch = batch;
    e->generation++;
    return ORBHIP_OK;
}

extern "C" int orbhip_extract_batch_device(orbhip_extractor *e, const uint8_t *d_images, int width, int height,
                                           size_t row_stride, size_t frame_stride, int batch, int lap0, int lap1)
{
    if (!e || batch <= 0 || row_stride < (size_t)width) return ORBHIP_E_BADARG;
    if (!d_images || width <= 0 || height <= 0) return ORBHIP_E_EMPTY;
    HIP_TRY(hipSetDevice(e->ctx->device));
    int rc = orbhip_extractor_reserve(e, width, height, batch);
    if (rc) return rc;
    // level 0 of the pyramid is the input itself (the reference copies it into a padded
    // buffer, ORBextractor.cc:1172; the bytes are identical): alias, do not copy -- unless the caller's
    // layout would break the kernels' aligned dword row accesses, then stage it once on the device.
    OrbLevel &L0 = e->P.lv[0];
    const bool aligned = ((uintptr_t)d_images % 4 == 0) && (row_stride % 4 == 0) && (frame_stride % 4 == 0);
    const bool use_graph = e->graph_mode && !e->profiling;
    if (aligned && !use_graph) {
        L0.img = const_cast<uint8_t *>(d_images);
        L0.img_pitch = (int)row_stride;
        L0.img_frame_stride = frame_stride;
    } else {                    // graph mode always stages: the captured kernels must see fixed addresses
        L0.img = e->d_level0; L0.img_pitch = e->level0_pitch; L0.img_frame_stride = e->level0_frame_stride;
        for (int f = 0; f < batch; f++)
            HIP_TRY(hipMemcpy2DAsync(L0.img + (size_t)f * L0.img_frame_stride, L0.img_pitch, d_images + (size_t)f * frame_stride,
                                     row_stride, width, height, hipMemcpyDeviceToDevice, e->ctx->stream));
    }
    return use_graph ? run_pipeline_graph(e, width, height, batch, lap0, lap1) : run_pipeline(e, batch, lap0, lap1);
}

// Small batches are launch bound (19 kernels per extract call): with graph mode on, the launches of one call are
// captured once per (size, batch, lapping) and replayed as a single hipGraph; the input is then always staged into
// the extractor's own level-0 buffer (one D2D copy) so that the captured kernels see fixed addresses.
extern "C" int orbhip_extractor_set_graph_mode(orbhip_extractor *e, int enable)
{
    if (!e) return ORBHIP_E_BADARG;
    e->graph_mode = enable != 0;
    if (!e->graph_mode && e->graph_valid) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_valid = false; }
    return ORBHIP_OK;
}

extern "C" int orbhip_extractor_results(orbhip_extractor *e, orbhip_keypoint **d_kp, uint8_t **d_desc,
                                        int32_t **d_count, int32_t **d_mono)
{
    if (!e || !e->max_batch) return ORBHIP_E_BADARG;
    if (d_kp) *d_kp = e->P.out_kp;
    if (d_desc) *d_desc = e->P.out_desc;
    if (d_count) *d_count = e->P.out_count;
    if (d_mono) *d_mono = e->P.out_mono;
    return ORBHIP_OK;
}

// Host-pointer extraction (what ORBextractor::operator() is): the image rows are gathered into a page-locked copy laid out like the
// device's level 0 and leave in ONE host-to-device copy; the results (counts, mono indices, keypoints, descriptors -- one device
// allocation) come back in ONE device-to-host copy into their page-locked mirror, the status word beside them, one synchronisation.
// round 3 issued one pageable 2-D copy per frame in, two synchronisations and 2 + 2 per-frame copies out.
static int extract_host_core(orbhip_extractor *e, const uint8_t *h_images, int width, int height, size_t row_stride, size_t frame_stride,
                             int batch, int lap0, int lap1)
{
    if (!e || batch <= 0) return ORBHIP_E_BADARG;
    if (!h_images || width <= 0 || height <= 0) return ORBHIP_E_EMPTY;      // ORBextractor.cc:1072-1073
    if (row_stride < (size_t)width) return ORBHIP_E_BADARG;
    HIP_TRY(hipSetDevice(e->ctx->device));
    int rc = orbhip_extractor_reserve(e, width, height, batch);
    if (rc) return rc;
    hipStream_t s = e->ctx->stream;
    OrbLevel &L0 = e->P.lv[0];
    L0.img = e->d_level0; L0.img_pitch = e->level0_pitch; L0.img_frame_stride = e->level0_frame_stride;
    const size_t in_bytes = (size_t)batch * L0.img_frame_stride;
    if (e->h_in_bytes < in_bytes) {
        if (e->h_in) (void)hipHostFree(e->h_in);
        e->h_in = nullptr; e->h_in_bytes = 0;
        HIP_TRY(hipHostMalloc((void **)&e->h_in, (size_t)e->max_batch * L0.img_frame_stride, hipHostMallocDefault));
        e->h_in_bytes = (size_t)e->max_batch * L0.img_frame_stride;
    }
    if (!e->h_out) HIP_TRY(hipHostMalloc((void **)&e->h_out, e->outblob_bytes, hipHostMallocDefault));
    if (!e->h_status) HIP_TRY(hipHostMalloc((void **)&e->h_status, 256, hipHostMallocDefault));
    for (int f = 0; f < batch; f++) {
        const uint8_t *src = h_images + (size_t)f * frame_stride;
        uint8_t *dst = e->h_in + (size_t)f * L0.img_frame_stride;
        if (row_stride == (size_t)L0.img_pitch) memcpy(dst, src, (size_t)L0.img_pitch * (height - 1) + width);
        else for (int y = 0; y < height; y++) memcpy(dst + (size_t)y * L0.img_pitch, src + (size_t)y * row_stride, width);
    }
    HIP_TRY(hipMemcpyAsync(L0.img, e->h_in, in_bytes, hipMemcpyHostToDevice, s));
    e->view_generation = ~0ull;
    rc = (e->graph_mode && !e->profiling) ? run_pipeline_graph(e, width, height, batch, lap0, lap1) : run_pipeline(e, batch, lap0, lap1);
    if (rc) return rc;
    const size_t B = (size_t)e->max_batch, mk = (size_t)e->P.max_kp;
    if ((size_t)batch == B) HIP_TRY(hipMemcpyAsync(e->h_out, e->d_outblob, e->outblob_bytes, hipMemcpyDeviceToHost, s));
    else {                                                                   // a smaller batch than reserved: the used rows of each slice
        HIP_TRY(hipMemcpyAsync(e->h_out, e->d_outblob, 8 * B, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(e->h_out + e->ob_kp, e->d_outblob + e->ob_kp, sizeof(orbhip_keypoint) * mk * batch, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(e->h_out + e->ob_desc, e->d_outblob + e->ob_desc, 32 * mk * batch, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipMemcpyAsync(e->h_status, e->P.status, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (*e->h_status) {
        g_last_error = "device-side list capacity exceeded";
        (void)hipMemsetAsync(e->P.status, 0, sizeof(int32_t), s);
        return *e->h_status;
    }
    e->view_generation = e->generation;
    return ORBHIP_OK;
}

extern "C" int orbhip_extract_batch_host(orbhip_extractor *e, const uint8_t *h_images, int width, int height,
                                         size_t row_stride, size_t frame_stride, int batch, int lap0, int lap1,
                                         orbhip_keypoint *kp_out, uint8_t *desc_out, int cap,
                                         int32_t *count_out, int32_t *mono_out)
{
    if (e && batch > 0 && h_images && width > 0 && height > 0 && (!kp_out || !desc_out || !count_out || !mono_out)) return ORBHIP_E_BADARG;
    const int rc = extract_host_core(e, h_images, width, height, row_stride, frame_stride, batch, lap0, lap1);
    if (rc) return rc;
    const int32_t *cnt = reinterpret_cast<const int32_t *>(e->h_out), *mono = cnt + e->max_batch;
    const size_t mk = (size_t)e->P.max_kp;
    for (int f = 0; f < batch; f++) {
        count_out[f] = cnt[f]; mono_out[f] = mono[f];
        if (cnt[f] > cap) return ORBHIP_E_CAPACITY;
        if (cnt[f] == 0) continue;
        memcpy(kp_out + (size_t)f * cap, e->h_out + e->ob_kp + sizeof(orbhip_keypoint) * mk * f, sizeof(orbhip_keypoint) * cnt[f]);
        memcpy(desc_out + (size_t)f * cap * 32, e->h_out + e->ob_desc + 32 * mk * f, (size_t)32 * cnt[f]);
    }
    return ORBHIP_OK;
}

extern "C" int orbhip_extract_batch_host_view(orbhip_extractor *e, const uint8_t *h_images, int width, int height, size_t row_stride,
                                              size_t frame_stride, int batch, int lap0, int lap1, const orbhip_keypoint **kp_view,
                                              const uint8_t **desc_view, int *row_capacity, const int32_t **count_view, const int32_t **mono_view)
{
    if (e && batch > 0 && h_images && width > 0 && height > 0 && (!kp_view || !desc_view || !row_capacity || !count_view || !mono_view)) return ORBHIP_E_BADARG;
    const int rc = extract_host_core(e, h_images, width, height, row_stride, frame_stride, batch, lap0, lap1);
    if (rc) return rc;
    *kp_view = reinterpret_cast<const orbhip_keypoint *>(e->h_out + e->ob_kp); *desc_view = e->h_out + e->ob_desc; *row_capacity = e->P.max_kp;
    *count_view = reinterpret_cast<const int32_t *>(e->h_out); *mono_view = *count_view + e->max_batch;
    return ORBHIP_OK;
}

extern "C" int orbhip_extractor_last_frame(orbhip_extractor *e, int frame, const orbhip_keypoint **d_kp, const uint8_t **d_desc,
                                           const orbhip_keypoint **h_kp_view, const uint8_t **h_desc_view, int32_t *count, unsigned long long *generation)
{
    if (!e || !e->max_batch || frame < 0 || frame >= e->last_batch || e->view_generation != e->generation) return ORBHIP_E_BADARG;
    const size_t mk = (size_t)e->P.max_kp;
    if (d_kp) *d_kp = e->P.out_kp + mk * frame;
    if (d_desc) *d_desc = e->P.out_desc + 32 * mk * frame;
    if (h_kp_view) *h_kp_view = reinterpret_cast<const orbhip_keypoint *>(e->h_out + e->ob_kp) + mk * frame;
    if (h_desc_view) *h_desc_view = e->h_out + e->ob_desc + 32 * mk * frame;
    if (count) *count = reinterpret_cast<const int32_t *>(e->h_out)[frame];
    if (generation) *generation = e->generation;
    return ORBHIP_OK;
}

// ------------------------------------------------------------------------------------ taps
extern "C" int orbhip_extractor_level_dims(const orbhip_extractor *e, int level, int *w, int *h)
{
    if (!e || level < 0 || level >= e->nlevels || !e->max_batch) return ORBHIP_E_BADARG;
    *w = e->P.lv[level].w; *h = e->P.lv[level].h;
    return ORBHIP_OK;
}

static inline int reflect101_host(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
    return p;
}

extern "C" int orbhip_extractor_get_pyramid_level(orbhip_extractor *e, int frame, int level, int padded,
                                                  uint8_t *h_out, size_t out_stride)
{
    if (!e || !h_out || level < 0 || level >= e->nlevels || frame < 0 || frame >= e->last_batch) return ORBHIP_E_BADARG;
    const OrbLevel &L = e->P.lv[level];
    HIP_TRY(hipSetDevice(e->ctx->device));
    HIP_TRY(hipStreamSynchronize(e->ctx->stream));
    if (!padded) {
        HIP_TRY(hipMemcpy2D(h_out, out_stride, L.img + (size_t)frame * L.img_frame_stride, L.img_pitch, L.w, L.h, hipMemcpyDeviceToHost));
        return ORBHIP_OK;
    }
    // mvImagePyramid's parent buffer: ROI + 19-px BORDER_REFLECT_101 (ORBextractor.cc:1167,1172),
    // synthesised lazily on copy-out; the extraction itself never reads the border.
    std::vector<uint8_t> tmp((size_t)L.w * L.h);
    HIP_TRY(hipMemcpy2D(tmp.data(), L.w, L.img + (size_t)frame * L.img_frame_stride, L.img_pitch, L.w, L.h, hipMemcpyDeviceToHost));
    const int pw = L.w + 2 * ORB_EDGE, ph = L.h + 2 * ORB_EDGE;
    for (int y = 0; y < ph; y++) {
        const uint8_t *srow = tmp.data() + (size_t)reflect101_host(y - ORB_EDGE, L.h) * L.w;
        uint8_t *drow = h_out + (size_t)y * out_stride;
        for (int x = 0; x < pw; x++) drow[x] = srow[reflect101_host(x - ORB_EDGE, L.w)];
    }
    return ORBHIP_OK;
}

extern "C" int orbhip_extractor_get_pyramid_padded(orbhip_extractor *e, int frame, uint8_t *const *levels_out, const size_t *strides)
{
    if (!e || !levels_out || !strides || frame < 0 || frame >= e->last_batch) return ORBHIP_E_BADARG;
    HIP_TRY(hipSetDevice(e->ctx->device));
    hipStream_t s = e->ctx->stream;
    // every level is one linear device-to-host copy (pitch x rows, page-locked landing area); rows are then laid into the caller's
    // padded parents on the host, where the border has to be synthesised anyway
    size_t total = 0;
    for (int l = 0; l < e->nlevels; l++) {
        const OrbLevel &L = e->P.lv[l];
        if (!levels_out[l] || strides[l] < (size_t)L.w + 2 * ORB_EDGE) return ORBHIP_E_BADARG;
        total += ((size_t)L.img_pitch * L.h + 255) & ~(size_t)255;
    }
    if (e->h_pyr_bytes < total) {
        if (e->h_pyr) (void)hipHostFree(e->h_pyr);
        e->h_pyr = nullptr; e->h_pyr_bytes = 0;
        HIP_TRY(hipHostMalloc((void **)&e->h_pyr, total, hipHostMallocDefault));
        e->h_pyr_bytes = total;
    }
    size_t off = 0;
    for (int l = 0; l < e->nlevels; l++) {
        const OrbLevel &L = e->P.lv[l];
        HIP_TRY(hipMemcpyAsync(e->h_pyr + off, L.img + (size_t)frame * L.img_frame_stride, (size_t)L.img_pitch * (L.h - 1) + L.w, hipMemcpyDeviceToHost, s));
        off += ((size_t)L.img_pitch * L.h + 255) & ~(size_t)255;
    }
    HIP_TRY(hipStreamSynchronize(s));
    off = 0;
    for (int l = 0; l < e->nlevels; l++) {
        const OrbLevel &L = e->P.lv[l];
        for (int y = 0; y < L.h; y++) memcpy(levels_out[l] + (size_t)(y + ORB_EDGE) * strides[l] + ORB_EDGE, e->h_pyr + off + (size_t)y * L.img_pitch, L.w);
        off += ((size_t)L.img_pitch * L.h + 255) & ~(size_t)255;
    }
    // the 19-px BORDER_REFLECT_101 frame (ORBextractor.cc:1167-1173), synthesised on the host: the extraction never reads it
    for (int l = 0; l < e->nlevels; l++) {
        const OrbLevel &L = e->P.lv[l];
        const size_t st = strides[l];
        uint8_t *base = levels_out[l];
        for (int y = ORB_EDGE; y < L.h + ORB_EDGE; y++) {
            uint8_t *row = base + (size_t)y * st;
            for (int x = 0; x < ORB_EDGE; x++) {
                row[x] = row[ORB_EDGE + reflect101_host(x - ORB_EDGE, L.w)];
                row[L.w + ORB_EDGE + x] = row[ORB_EDGE + reflect101_host(L.w + x, L.w)];
            }
        }
        for (int y = 0; y < ORB_EDGE; y++) {
            memcpy(base + (size_t)y * st, base + (size_t)(ORB_EDGE + reflect101_host(y - ORB_EDGE, L.h)) * st, (size_t)L.w + 2 * ORB_EDGE);
            memcpy(base + (size_t)(L.h + ORB_EDGE + y) * st, base + (size_t)(ORB_EDGE + reflect101_host(L.h + y, L.h)) * st, (size_t)L.w + 2 * ORB_EDGE);
        }
    }
    return ORBHIP_OK;
}

extern "C" int orbhip_extractor_get_blurred_level(orbhip_extractor *e, int frame, int level, uint8_t *h_out, size_t out_stride)
{
    if (!e || !h_out || level < 0 || level >= e->nlevels || frame < 0 || frame >= e->last_batch) return ORBHIP_E_BADARG;
    const OrbLevel &L = e->P.lv[level];
    HIP_TRY(hipSetDevice(e->ctx->device));
    HIP_TRY(hipStreamSynchronize(e->ctx->stream));
    HIP_TRY(hipMemcpy2D(h_out, out_stride, L.blur + (size_t)frame * L.blur_frame_stride, L.blur_pitch, L.w, L.h, hipMemcpyDeviceToHost));
    return ORBHIP_OK;
}

extern "C" int orbhip_extractor_get_fast_candidates(orbhip_extractor *e, int frame, int level, int32_t *xs, int32_t *ys,
                                                    int32_t *scores, int cap, int32_t *n_out)
{
    if (!e || level < 0 || level >= e->nlevels || frame < 0 || frame >= e->last_batch || !n_out) return ORBHIP_E_BADARG;
    const OrbParams &P = e->P;
    const OrbLevel &L = P.lv[level];
    HIP_TRY(hipSetDevice(e->ctx->device));
    HIP_TRY(hipStreamSynchronize(e->ctx->stream));
    int32_t n = 0;
    HIP_TRY(hipMemcpy(&n, P.lvl_ncand + frame * P.nlevels + level, sizeof(n), hipMemcpyDeviceToHost));
    *n_out = n;
    int m = std::min(n, cap);
    if (m > 0 && xs && ys && scores) {
        std::vector<uint32_t> k(m);
        HIP_TRY(hipMemcpy(k.data(), P.keys + (size_t)frame * P.keys_per_frame + L.key_base, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
        for (int i = 0; i < m; i++) { xs[i] = ORB_KEY_X(k[i]); ys[i] = ORB_KEY_Y(k[i]); scores[i] = ORB_KEY_S(k[i]); }
    }
    return ORBHIP_OK;
}

extern "C" int orbhip_extractor_get_level_keypoints(orbhip_extractor *e, int frame, int level, orbhip_keypoint *out,
                                                    int cap, int32_t *n_out)
{
    if (!e || level < 0 || level >= e->nlevels || frame < 0 || frame >= e->last_batch || !n_out) return ORBHIP_E_BADARG;
    const OrbParams &P = e->P;
    const OrbLevel &L = P.lv[level];
    HIP_TRY(hipSetDevice(e->ctx->device));
    HIP_TRY(hipStreamSynchronize(e->ctx->stream));
    int32_t n = 0;
    HIP_TRY(hipMemcpy(&n, P.lvl_count + frame * P.nlevels + level, sizeof(n), hipMemcpyDeviceToHost));
    *n_out = n;
    int m = std::min(n, cap);
    if (m > 0 && out) {
        std::vector<uint32_t> k(m); std::vector<float> a(m);
        const size_t off = (size_t)frame * P.kps_per_frame + L.kp_base;
        HIP_TRY(hipMemcpy(k.data(), P.lvl_kp + off, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(a.data(), P.lvl_angle + off, sizeof(float) * m, hipMemcpyDeviceToHost));
        for (int i = 0; i < m; i++) {
            out[i].x = (float)(ORB_KEY_X(k[i]) + ORB_MINB); out[i].y = (float)(ORB_KEY_Y(k[i]) + ORB_MINB);
            out[i].size = L.size; out[i].angle = a[i]; out[i].response = (float)ORB_KEY_S(k[i]);
            out[i].octave = level; out[i].class_id = -1;
        }
    }
    return ORBHIP_OK;
}
