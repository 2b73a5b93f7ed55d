// host_entry.hip -- host-pointer forms of the matcher / pose-optimisation entry points (include/orbhip.h): what an ORBmatcher or
// Optimizer method with the reference's signature needs for ONE frame.  No computation happens on the host.
//
// Round 4: a call used to issue 8-17 hipMemcpyAsync from PAGEABLE host vectors (each one staged and waited for by
// the runtime) and 2-4 copies back.  Now every call lays its arrays out in ONE blob -- inputs | in/outs | outputs | device-only buffers --
// that exists twice, in the context's device arena and in its page-locked host arena: the inputs are gathered into the host blob by
// memcpy and leave in ONE host-to-device copy, the kernels run on the device blob, in/outs + outputs come back in ONE device-to-host
// copy with the context's status word beside them, ONE synchronisation, and the results are scattered to the caller's arrays.
// The *_resident forms take the train side (keypoints + descriptors) as DEVICE pointers: the extractor's result arrays of the frame that
// was extracted a moment ago (orbhip_extractor_last_frame), so Tracking's matchers upload their queries only.
#include "orb_internal.h"
#include <cstdlib>
#include <cstring>
#include <vector>

hipStream_t orbhip_ctx_stream_internal(orbhip_ctx *c);
int orbhip_ctx_device_internal(orbhip_ctx *c);
void *orbhip_ctx_scratch_internal(orbhip_ctx *c, size_t bytes);
void *orbhip_ctx_pinned_internal(orbhip_ctx *c, size_t bytes);
int32_t *orbhip_ctx_status_internal(orbhip_ctx *c);
void orbhip_set_last_error_internal(const char *msg);
void orbhip_ctx_redirect_status_internal(orbhip_ctx *c, int32_t *p);

namespace {
inline size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
#define HTRY(e) do { if ((e) != hipSuccess) { orbhip_set_last_error_internal(#e); return ORBHIP_E_HIP; } } while (0)

struct HostCall {
    enum Kind { K_IN = 0, K_INOUT = 1, K_OUT = 2, K_BUF = 3 };
    struct Item { Kind k; const void *src; void *dst; size_t bytes, reserve, off; };
    orbhip_ctx *ctx; hipStream_t s;
    std::vector<Item> items;
    uint8_t *base, *host;                  // device blob, page-locked mirror
    size_t up_end, dl_start, dl_end, total;
    int32_t zero; int a_status; bool redirected;
    explicit HostCall(orbhip_ctx *c) : ctx(c), s(orbhip_ctx_stream_internal(c)), base(nullptr), host(nullptr), up_end(0), dl_start(0), dl_end(0), total(0), zero(0), a_status(-1), redirected(false) { items.reserve(32); }
    ~HostCall() { if (redirected) orbhip_ctx_redirect_status_internal(ctx, nullptr); }
    // `reserve` >= bytes: the array's size on the device (rows of a common stride); src == nullptr: an optional input that is absent
    int add(Kind k, const void *src, void *dst, size_t bytes, size_t reserve) { items.push_back({k, src, dst, bytes, reserve > bytes ? reserve : bytes, 0}); return (int)items.size() - 1; }
    int in(const void *src, size_t bytes, size_t reserve = 0) { return add(K_IN, src, nullptr, src ? bytes : 0, reserve); }
    int inout(void *p, size_t bytes, size_t reserve = 0) { return add(K_INOUT, p, p, bytes, reserve); }
    int out(void *dst, size_t bytes, size_t reserve = 0) { return add(K_OUT, nullptr, dst, bytes, reserve); }
    int buf(size_t bytes) { return add(K_BUF, nullptr, nullptr, 0, bytes); }
    int commit()
    {
        // the kernels' status word is an in/out slot of this call's blob: uploaded as 0 with the inputs, back with the outputs
        if (!pageable()) a_status = add(K_INOUT, &zero, nullptr, 4, 0);
        size_t off = 0;
        for (int k = K_IN; k <= K_BUF; k++) {
            if (k == K_INOUT) dl_start = off;
            for (Item &it : items) if (it.k == k) { it.off = off; off += al(it.reserve ? it.reserve : 1); }
            if (k == K_INOUT) up_end = off;
            if (k == K_OUT) dl_end = off;
        }
        total = off;
        if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) { orbhip_set_last_error_internal("hipSetDevice"); return ORBHIP_E_HIP; }
        base = (uint8_t *)orbhip_ctx_scratch_internal(ctx, total + 256);
        host = (uint8_t *)orbhip_ctx_pinned_internal(ctx, dl_end + 256);         // + the status word behind the outputs
        if (!base || !host) return ORBHIP_E_HIP;
        if (pageable()) {                                                        // round 3's transfer pattern, kept for A/B measurements
            for (const Item &it : items)
                if (it.k <= K_INOUT && it.src && it.bytes && hipMemcpyAsync(base + it.off, it.src, it.bytes, hipMemcpyHostToDevice, s) != hipSuccess) {
                    orbhip_set_last_error_internal("hipMemcpyAsync (host -> device staging)"); return ORBHIP_E_HIP;
                }
            return ORBHIP_OK;
        }
        for (const Item &it : items) if (it.k <= K_INOUT && it.src && it.bytes) memcpy(host + it.off, it.src, it.bytes);
        if (up_end && hipMemcpyAsync(base, host, up_end, hipMemcpyHostToDevice, s) != hipSuccess) { orbhip_set_last_error_internal("hipMemcpyAsync (host -> device blob)"); return ORBHIP_E_HIP; }
        orbhip_ctx_redirect_status_internal(ctx, ptr<int32_t>(a_status)); redirected = true;
        return ORBHIP_OK;
    }
    template <typename T> T *ptr(int i) const { return reinterpret_cast<T *>(base + items[i].off); }
    // one copy back, the status word beside it, one synchronisation; ORBHIP_E_CAPACITY etc. when a kernel flagged the frame
    // ORBHIP_HOST_PAGEABLE=1: one copy per array straight from / to the caller's pageable memory (what round 3 did)
    static bool pageable() { static const bool on = getenv("ORBHIP_HOST_PAGEABLE") && atoi(getenv("ORBHIP_HOST_PAGEABLE")) != 0; return on; }
    int finish()
    {
        int32_t *hst = reinterpret_cast<int32_t *>(host + dl_end);
        if (pageable()) {
            for (const Item &it : items)
                if ((it.k == K_INOUT || it.k == K_OUT) && it.dst && it.bytes) HTRY(hipMemcpyAsync(it.dst, base + it.off, it.bytes, hipMemcpyDeviceToHost, s));
            HTRY(hipStreamSynchronize(s));
            HTRY(hipMemcpy(hst, orbhip_ctx_status_internal(ctx), sizeof(int32_t), hipMemcpyDeviceToHost));
            if (*hst) {
                const int st = *hst;
                HTRY(hipMemset(orbhip_ctx_status_internal(ctx), 0, sizeof(int32_t)));
                orbhip_set_last_error_internal("device-side capacity exceeded in a matcher kernel");
                return st;
            }
            return ORBHIP_OK;
        }
        orbhip_ctx_redirect_status_internal(ctx, nullptr); redirected = false;
        HTRY(hipMemcpyAsync(host + dl_start, base + dl_start, dl_end - dl_start, hipMemcpyDeviceToHost, s));
        HTRY(hipStreamSynchronize(s));
        hst = reinterpret_cast<int32_t *>(host + items[a_status].off);
        if (*hst) {
            orbhip_set_last_error_internal("device-side capacity exceeded in a matcher kernel");
            return *hst;
        }
        for (const Item &it : items) if ((it.k == K_INOUT || it.k == K_OUT) && it.dst && it.bytes) memcpy(it.dst, host + it.off, it.bytes);
        return ORBHIP_OK;
    }
};
}  // namespace

// train side: host arrays (kp / desc) or, when d_kp_res / d_desc_res are given, arrays already on the device
static int sbp_host(orbhip_ctx *ctx, int mode, const orbhip_proj_query *q, const uint8_t *desc_q, int nq,
                    const orbhip_keypoint *kp, const uint8_t *desc, const orbhip_keypoint *d_kp_res, const uint8_t *d_desc_res,
                    const float *u_right, int n, int nleft, const int32_t *mirror,
                    float min_x, float min_y, float max_x, float max_y, int th_high, float nn_ratio,
                    int check_orientation, int32_t *train_match_inout, int32_t *nmatches_out)
{
    if (!ctx || nq < 0 || n < 0 || (nq && (!q || !desc_q)) || (n && ((!kp && !d_kp_res) || (!desc && !d_desc_res) || !train_match_inout)) || !nmatches_out ||
        (mode != 0 && mode != 1))
        return ORBHIP_E_BADARG;
    *nmatches_out = 0;
    if (nq == 0 || n == 0) return ORBHIP_OK;
    HostCall H(ctx);
    const int a_q = H.in(q, sizeof(orbhip_proj_query) * (size_t)nq), a_dq = H.in(desc_q, 32 * (size_t)nq);
    const int a_kp = H.in(d_kp_res ? nullptr : kp, sizeof(orbhip_keypoint) * (size_t)n), a_d = H.in(d_desc_res ? nullptr : desc, 32 * (size_t)n);
    const int a_ur = H.in(u_right, 4 * (size_t)n), a_mi = H.in(nleft >= 0 ? mirror : nullptr, 4 * (size_t)n);
    const int a_nq = H.in(&nq, 4), a_n = H.in(&n, 4), a_nl = H.in(&nleft, 4);
    const int a_tm = H.inout(train_match_inout, 4 * (size_t)n), a_nm = H.out(nmatches_out, 4);
    if (int rc = H.commit()) return rc;
    const orbhip_keypoint *dkp = d_kp_res ? d_kp_res : H.ptr<orbhip_keypoint>(a_kp);
    const uint8_t *dd = d_desc_res ? d_desc_res : H.ptr<uint8_t>(a_d);
    const float *dur = u_right ? H.ptr<float>(a_ur) : nullptr;
    int rc;
    if (nleft >= 0)
        rc = orbhip_search_by_projection_rig_device(ctx, mode, H.ptr<orbhip_proj_query>(a_q), H.ptr<uint8_t>(a_dq), H.ptr<int32_t>(a_nq), nq, dkp, dd, H.ptr<int32_t>(a_n),
                                                    H.ptr<int32_t>(a_nl), mirror ? H.ptr<int32_t>(a_mi) : nullptr, n, (size_t)n, 1, min_x, min_y, max_x, max_y, th_high,
                                                    nn_ratio, check_orientation, H.ptr<int32_t>(a_tm), H.ptr<int32_t>(a_nm));
    else if (mode == 0)
        rc = orbhip_search_by_projection_device(ctx, H.ptr<orbhip_proj_query>(a_q), H.ptr<uint8_t>(a_dq), H.ptr<int32_t>(a_nq), nq, dkp, dd, dur, H.ptr<int32_t>(a_n), n,
                                                (size_t)n, 1, min_x, min_y, max_x, max_y, th_high, check_orientation, H.ptr<int32_t>(a_tm), H.ptr<int32_t>(a_nm));
    else
        rc = orbhip_search_local_map_device(ctx, H.ptr<orbhip_proj_query>(a_q), H.ptr<uint8_t>(a_dq), H.ptr<int32_t>(a_nq), nq, dkp, dd, dur, H.ptr<int32_t>(a_n), n,
                                            (size_t)n, 1, min_x, min_y, max_x, max_y, th_high, nn_ratio, H.ptr<int32_t>(a_tm), H.ptr<int32_t>(a_nm));
    if (rc) return rc;
    return H.finish();                                   // ORBHIP_E_CAPACITY when the frame exceeds the kernel's limits
}

extern "C" int orbhip_search_by_projection_host(orbhip_ctx *ctx, int mode, const orbhip_proj_query *q, const uint8_t *desc_q, int nq,
                                                const orbhip_keypoint *kp, const uint8_t *desc, const float *u_right, int n,
                                                float min_x, float min_y, float max_x, float max_y, int th_high, float nn_ratio,
                                                int check_orientation, int32_t *train_match_inout, int32_t *nmatches_out)
{
    return sbp_host(ctx, mode, q, desc_q, nq, kp, desc, nullptr, nullptr, u_right, n, -1, nullptr, min_x, min_y, max_x, max_y, th_high, nn_ratio, check_orientation,
                    train_match_inout, nmatches_out);
}

extern "C" int orbhip_search_by_projection_host_resident(orbhip_ctx *ctx, int mode, const orbhip_proj_query *q, const uint8_t *desc_q, int nq,
                                                         const orbhip_keypoint *kp_host, const orbhip_keypoint *d_kp, const uint8_t *d_desc, const float *u_right, int n,
                                                         float min_x, float min_y, float max_x, float max_y, int th_high, float nn_ratio,
                                                         int check_orientation, int32_t *train_match_inout, int32_t *nmatches_out)
{
    if (n && (!d_desc || (!d_kp && !kp_host))) return ORBHIP_E_BADARG;
    return sbp_host(ctx, mode, q, desc_q, nq, d_kp ? nullptr : kp_host, nullptr, d_kp, d_desc, u_right, n, -1, nullptr, min_x, min_y, max_x, max_y, th_high, nn_ratio, check_orientation,
                    train_match_inout, nmatches_out);
}

extern "C" int orbhip_search_by_projection_rig_host(orbhip_ctx *ctx, int mode, const orbhip_proj_query *q, const uint8_t *desc_q, int nq,
                                                    const orbhip_keypoint *kp, const uint8_t *desc, int n, int nleft, const int32_t *mirror,
                                                    float min_x, float min_y, float max_x, float max_y, int th_high, float nn_ratio,
                                                    int check_orientation, int32_t *train_match_inout, int32_t *nmatches_out)
{
    if (nleft < 0 || nleft > n) return ORBHIP_E_BADARG;
    return sbp_host(ctx, mode, q, desc_q, nq, kp, desc, nullptr, nullptr, nullptr, n, nleft, mirror, min_x, min_y, max_x, max_y, th_high, nn_ratio, check_orientation,
                    train_match_inout, nmatches_out);
}

static int si_host(orbhip_ctx *ctx, const orbhip_keypoint *kpA, const uint8_t *descA, int nA, const orbhip_keypoint *kpB, const uint8_t *descB,
                   const orbhip_keypoint *d_kpB_res, const uint8_t *d_descB_res, int nB, float min_x, float min_y, float max_x, float max_y, int window_size,
                   float nn_ratio, int check_orientation, float *prev_matched_inout, int32_t *matches12_out, int32_t *nmatches_out)
{
    if (!ctx || nA < 0 || nB < 0 || (nA && (!kpA || !descA || !prev_matched_inout || !matches12_out)) || (nB && ((!kpB && !d_kpB_res) || (!descB && !d_descB_res))) ||
        !nmatches_out)
        return ORBHIP_E_BADARG;
    *nmatches_out = 0;
    for (int i = 0; i < nA; i++) matches12_out[i] = -1;                 // vnMatches12 = vector<int>(F1.mvKeysUn.size(), -1), ORBmatcher.cc:713
    if (nA == 0 || nB == 0) return ORBHIP_OK;
    // the device entry point takes both sides at one row stride; a resident B side sits in the extractor's arrays, so both sides are
    // addressed as single rows there (batch 1: the stride is never used)
    const int mx = nA > nB ? nA : nB;
    HostCall H(ctx);
    const int a_ka = H.in(kpA, sizeof(orbhip_keypoint) * (size_t)nA, sizeof(orbhip_keypoint) * (size_t)mx), a_da = H.in(descA, 32 * (size_t)nA, 32 * (size_t)mx);
    const int a_kb = H.in(d_kpB_res ? nullptr : kpB, sizeof(orbhip_keypoint) * (size_t)nB, d_kpB_res ? 0 : sizeof(orbhip_keypoint) * (size_t)mx);      // (absent: 0 bytes)
    const int a_db = H.in(d_descB_res ? nullptr : descB, 32 * (size_t)nB, d_descB_res ? 0 : 32 * (size_t)mx);
    const int a_na = H.in(&nA, 4), a_nb = H.in(&nB, 4);
    const int a_pm = H.inout(prev_matched_inout, 8 * (size_t)nA, 8 * (size_t)mx), a_m = H.out(matches12_out, 4 * (size_t)nA, 4 * (size_t)mx), a_nm = H.out(nmatches_out, 4);
    if (int rc = H.commit()) return rc;
    const int rc = orbhip_search_for_initialization_device(ctx, H.ptr<orbhip_keypoint>(a_ka), H.ptr<uint8_t>(a_da), H.ptr<int32_t>(a_na),
                                                           d_kpB_res ? d_kpB_res : H.ptr<orbhip_keypoint>(a_kb), d_descB_res ? d_descB_res : H.ptr<uint8_t>(a_db),
                                                           H.ptr<int32_t>(a_nb), 1, mx, (size_t)mx, min_x, min_y, max_x, max_y, window_size, nn_ratio, check_orientation,
                                                           H.ptr<float>(a_pm), H.ptr<int32_t>(a_m), H.ptr<int32_t>(a_nm));
    if (rc) return rc;
    return H.finish();
}

extern "C" int orbhip_search_for_initialization_host(orbhip_ctx *ctx, const orbhip_keypoint *kpA, const uint8_t *descA, int nA,
                                                     const orbhip_keypoint *kpB, const uint8_t *descB, int nB, float min_x, float min_y,
                                                     float max_x, float max_y, int window_size, float nn_ratio, int check_orientation,
                                                     float *prev_matched_inout, int32_t *matches12_out, int32_t *nmatches_out)
{
    return si_host(ctx, kpA, descA, nA, kpB, descB, nullptr, nullptr, nB, min_x, min_y, max_x, max_y, window_size, nn_ratio, check_orientation, prev_matched_inout,
                   matches12_out, nmatches_out);
}

extern "C" int orbhip_search_for_initialization_host_resident(orbhip_ctx *ctx, const orbhip_keypoint *kpA, const uint8_t *descA, int nA,
                                                              const orbhip_keypoint *kpB_host, const orbhip_keypoint *d_kpB, const uint8_t *d_descB, int nB, float min_x, float min_y,
                                                              float max_x, float max_y, int window_size, float nn_ratio, int check_orientation,
                                                              float *prev_matched_inout, int32_t *matches12_out, int32_t *nmatches_out)
{
    if (nB && (!d_descB || (!d_kpB && !kpB_host))) return ORBHIP_E_BADARG;
    return si_host(ctx, kpA, descA, nA, d_kpB ? nullptr : kpB_host, nullptr, d_kpB, d_descB, nB, min_x, min_y, max_x, max_y, window_size, nn_ratio, check_orientation, prev_matched_inout,
                   matches12_out, nmatches_out);
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) for ONE (keyframe, frame) pair from host memory: the two flattened
// FeatureVectors (node ids ascending, node_start [nnodes + 1], feature indices), the keyframe's "map point exists and is not bad"
// flags, keypoints (mvKeysUn / mvKeys, concatenated left | right for rig frames) and descriptors.  nleft < 0: F.Nleft == -1.
static int bow_host(orbhip_ctx *ctx,
        const int32_t *kf_node_ids, const int32_t *kf_node_start, const int32_t *kf_feat, int kf_nnodes, const uint8_t *kf_valid,
        const orbhip_keypoint *kf_kp, const uint8_t *kf_desc, int nK,
        const int32_t *f_node_ids, const int32_t *f_node_start, const int32_t *f_feat, int f_nnodes,
        const orbhip_keypoint *f_kp, const uint8_t *f_desc, const orbhip_keypoint *d_fkp_res, const uint8_t *d_fdesc_res, int nF, int nleft,
        float nn_ratio, int check_orientation, int32_t *match_f_out, int32_t *nmatches_out)
{
    if (!ctx || nK < 0 || nF < 0 || kf_nnodes < 0 || f_nnodes < 0 || !nmatches_out || (nF && !match_f_out) ||
        (kf_nnodes && (!kf_node_ids || !kf_node_start || !kf_feat || !kf_valid || !kf_kp || !kf_desc)) ||
        (f_nnodes && (!f_node_ids || !f_node_start || !f_feat || (!f_kp && !d_fkp_res) || (!f_desc && !d_fdesc_res))) || nleft > nF)
        return ORBHIP_E_BADARG;
    *nmatches_out = 0;
    for (int i = 0; i < nF; i++) match_f_out[i] = -1;                   // vpMapPointMatches = vector<MapPoint*>(F.N, NULL), ORBmatcher.cc:277
    if (nK == 0 || nF == 0 || kf_nnodes == 0 || f_nnodes == 0) return ORBHIP_OK;
    const int mn = kf_nnodes > f_nnodes ? kf_nnodes : f_nnodes, mx = nK > nF ? nK : nF;
    HostCall H(ctx);
    const int a_ki = H.in(kf_node_ids, 4 * (size_t)kf_nnodes, 4 * (size_t)mn), a_ks = H.in(kf_node_start, 4 * (size_t)(kf_nnodes + 1), 4 * (size_t)(mn + 1));
    const int a_kf = H.in(kf_feat, 4 * (size_t)kf_node_start[kf_nnodes], 4 * (size_t)mx);
    const int a_fi = H.in(f_node_ids, 4 * (size_t)f_nnodes, 4 * (size_t)mn), a_fs = H.in(f_node_start, 4 * (size_t)(f_nnodes + 1), 4 * (size_t)(mn + 1));
    const int a_ff = H.in(f_feat, 4 * (size_t)f_node_start[f_nnodes], 4 * (size_t)mx);
    const int a_kk = H.in(kf_kp, sizeof(orbhip_keypoint) * (size_t)nK, sizeof(orbhip_keypoint) * (size_t)mx), a_kd = H.in(kf_desc, 32 * (size_t)nK, 32 * (size_t)mx);
    const int a_fk = H.in(d_fkp_res ? nullptr : f_kp, sizeof(orbhip_keypoint) * (size_t)nF, d_fkp_res ? 0 : sizeof(orbhip_keypoint) * (size_t)mx);
    const int a_fd = H.in(d_fdesc_res ? nullptr : f_desc, 32 * (size_t)nF, d_fdesc_res ? 0 : 32 * (size_t)mx);
    const int a_va = H.in(kf_valid, (size_t)nK, (size_t)mx);
    const int a_kn = H.in(&kf_nnodes, 4), a_fn = H.in(&f_nnodes, 4), a_nF = H.in(&nF, 4), a_nl = H.in(&nleft, 4);
    const int a_m = H.out(match_f_out, 4 * (size_t)nF, 4 * (size_t)mx), a_nm = H.out(nmatches_out, 4);
    if (int rc = H.commit()) return rc;
    const orbhip_keypoint *dfk = d_fkp_res ? d_fkp_res : H.ptr<orbhip_keypoint>(a_fk);
    const uint8_t *dfd = d_fdesc_res ? d_fdesc_res : H.ptr<uint8_t>(a_fd);
    int rc;
    if (nleft >= 0)
        rc = orbhip_search_by_bow_rig_device(ctx, H.ptr<int32_t>(a_ki), H.ptr<int32_t>(a_ks), H.ptr<int32_t>(a_kf), H.ptr<int32_t>(a_kn), H.ptr<uint8_t>(a_va),
                                             H.ptr<orbhip_keypoint>(a_kk), H.ptr<uint8_t>(a_kd), H.ptr<int32_t>(a_fi), H.ptr<int32_t>(a_fs), H.ptr<int32_t>(a_ff),
                                             H.ptr<int32_t>(a_fn), dfk, dfd, H.ptr<int32_t>(a_nF), H.ptr<int32_t>(a_nl), 1, mn, mx, (size_t)mx, nn_ratio,
                                             check_orientation, H.ptr<int32_t>(a_m), H.ptr<int32_t>(a_nm));
    else
        rc = orbhip_search_by_bow_device(ctx, H.ptr<int32_t>(a_ki), H.ptr<int32_t>(a_ks), H.ptr<int32_t>(a_kf), H.ptr<int32_t>(a_kn), H.ptr<uint8_t>(a_va),
                                         H.ptr<orbhip_keypoint>(a_kk), H.ptr<uint8_t>(a_kd), H.ptr<int32_t>(a_fi), H.ptr<int32_t>(a_fs), H.ptr<int32_t>(a_ff),
                                         H.ptr<int32_t>(a_fn), dfk, dfd, H.ptr<int32_t>(a_nF), 1, mn, mx, (size_t)mx, nn_ratio, check_orientation,
                                         H.ptr<int32_t>(a_m), H.ptr<int32_t>(a_nm));
    if (rc) return rc;
    return H.finish();
}

extern "C" int orbhip_search_by_bow_host(orbhip_ctx *ctx,
        const int32_t *kf_node_ids, const int32_t *kf_node_start, const int32_t *kf_feat, int kf_nnodes, const uint8_t *kf_valid,
        const orbhip_keypoint *kf_kp, const uint8_t *kf_desc, int nK,
        const int32_t *f_node_ids, const int32_t *f_node_start, const int32_t *f_feat, int f_nnodes,
        const orbhip_keypoint *f_kp, const uint8_t *f_desc, int nF, int nleft,
        float nn_ratio, int check_orientation, int32_t *match_f_out, int32_t *nmatches_out)
{
    return bow_host(ctx, kf_node_ids, kf_node_start, kf_feat, kf_nnodes, kf_valid, kf_kp, kf_desc, nK, f_node_ids, f_node_start, f_feat, f_nnodes, f_kp, f_desc,
                    nullptr, nullptr, nF, nleft, nn_ratio, check_orientation, match_f_out, nmatches_out);
}

extern "C" int orbhip_search_by_bow_host_resident(orbhip_ctx *ctx,
        const int32_t *kf_node_ids, const int32_t *kf_node_start, const int32_t *kf_feat, int kf_nnodes, const uint8_t *kf_valid,
        const orbhip_keypoint *kf_kp, const uint8_t *kf_desc, int nK,
        const int32_t *f_node_ids, const int32_t *f_node_start, const int32_t *f_feat, int f_nnodes,
        const orbhip_keypoint *f_kp_host, const orbhip_keypoint *d_f_kp, const uint8_t *d_f_desc, int nF,
        float nn_ratio, int check_orientation, int32_t *match_f_out, int32_t *nmatches_out)
{
    if (f_nnodes && (!d_f_desc || (!d_f_kp && !f_kp_host))) return ORBHIP_E_BADARG;
    return bow_host(ctx, kf_node_ids, kf_node_start, kf_feat, kf_nnodes, kf_valid, kf_kp, kf_desc, nK, f_node_ids, f_node_start, f_feat, f_nnodes, d_f_kp ? nullptr : f_kp_host, nullptr,
                    d_f_kp, d_f_desc, nF, -1, nn_ratio, check_orientation, match_f_out, nmatches_out);
}

extern "C" int orbhip_search_for_triangulation_host(orbhip_ctx *ctx,
        const int32_t *nid1, const uint8_t *has_mp1, const orbhip_keypoint *kp1, const uint8_t *desc1, const float *u_right1, int n1,
        const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2,
        const uint8_t *has_mp2, const orbhip_keypoint *kp2, const uint8_t *desc2, const float *u_right2, int n2,
        const orbhip_tri_pair_general *pair, const float *level_sigma2_1, const float *scale_factors2, const float *level_sigma2_2,
        int nlevels, int check_orientation, int32_t *matches12_out, int32_t *nmatches_out)
{
    if (!ctx || n1 < 0 || n2 < 0 || nnodes2 < 0 || !pair || !nmatches_out || (n1 && (!nid1 || !has_mp1 || !kp1 || !desc1 || !matches12_out)) ||
        (n2 && (!has_mp2 || !kp2 || !desc2)) || (nnodes2 && (!node_ids2 || !node_start2 || !feat2)) || !level_sigma2_1 || !scale_factors2 || !level_sigma2_2)
        return ORBHIP_E_BADARG;
    *nmatches_out = 0;
    for (int i = 0; i < n1; i++) matches12_out[i] = -1;                 // vMatches12 = vector<int>(pKF1->N, -1), ORBmatcher.cc:1017
    if (n1 == 0 || n2 == 0 || nnodes2 == 0) return ORBHIP_OK;
    const int mx = n1 > n2 ? n1 : n2;
    HostCall H(ctx);
    const int a_nid = H.in(nid1, 4 * (size_t)n1), a_mp1 = H.in(has_mp1, n1), a_ur1 = H.in(u_right1, 4 * (size_t)n1);
    const int a_ids = H.in(node_ids2, 4 * (size_t)nnodes2), a_st = H.in(node_start2, 4 * (size_t)(nnodes2 + 1)), a_fe = H.in(feat2, 4 * (size_t)node_start2[nnodes2]);
    const int a_mp2 = H.in(has_mp2, n2), a_ur2 = H.in(u_right2, 4 * (size_t)n2);
    const int a_kp1 = H.in(kp1, sizeof(orbhip_keypoint) * (size_t)n1, sizeof(orbhip_keypoint) * (size_t)mx);      // rows of mx entries: one stride for both sides
    const int a_kp2 = H.in(kp2, sizeof(orbhip_keypoint) * (size_t)n2, sizeof(orbhip_keypoint) * (size_t)mx);
    const int a_d1 = H.in(desc1, 32 * (size_t)n1, 32 * (size_t)mx), a_d2 = H.in(desc2, 32 * (size_t)n2, 32 * (size_t)mx);
    const int a_pair = H.in(pair, sizeof(*pair)), a_n1 = H.in(&n1, 4), a_n2 = H.in(&n2, 4), a_nn = H.in(&nnodes2, 4);
    const int a_m = H.out(matches12_out, 4 * (size_t)n1, 4 * (size_t)mx), a_nm = H.out(nmatches_out, 4);
    if (int rc = H.commit()) return rc;
    const int rc = orbhip_search_for_triangulation_general_device(ctx, H.ptr<int32_t>(a_nid), H.ptr<uint8_t>(a_mp1), H.ptr<orbhip_keypoint>(a_kp1),
        H.ptr<uint8_t>(a_d1), u_right1 ? H.ptr<float>(a_ur1) : nullptr, H.ptr<int32_t>(a_n1), H.ptr<int32_t>(a_ids), H.ptr<int32_t>(a_st), H.ptr<int32_t>(a_fe),
        H.ptr<int32_t>(a_nn), H.ptr<uint8_t>(a_mp2), H.ptr<orbhip_keypoint>(a_kp2), H.ptr<uint8_t>(a_d2), u_right2 ? H.ptr<float>(a_ur2) : nullptr,
        H.ptr<int32_t>(a_n2), H.ptr<orbhip_tri_pair_general>(a_pair), 1, nnodes2, mx, (size_t)mx, level_sigma2_1, scale_factors2, level_sigma2_2, nlevels,
        check_orientation, H.ptr<int32_t>(a_m), H.ptr<int32_t>(a_nm));
    if (rc) return rc;
    return H.finish();
}

extern "C" int orbhip_match_and_triangulate_host(orbhip_ctx *ctx,
        const int32_t *nid1, const uint8_t *has_mp1, const orbhip_keypoint *kp1, const uint8_t *desc1, int n1,
        const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2,
        const uint8_t *has_mp2, const orbhip_keypoint *kp2, const uint8_t *desc2, int n2,
        const orbhip_tri_pair_general *pair, const orbhip_tri_pair_poses *poses, const float *level_sigma2_1, const float *level_sigma2_2,
        int nlevels, int check_orientation, int32_t *matches12_out, float *points12_out, int32_t *nmatches_out)
{
    if (!ctx || n1 < 0 || n2 < 0 || nnodes2 < 0 || !pair || !poses || !nmatches_out || (n1 && (!nid1 || !has_mp1 || !kp1 || !desc1 || !matches12_out || !points12_out)) ||
        (n2 && (!has_mp2 || !kp2 || !desc2)) || (nnodes2 && (!node_ids2 || !node_start2 || !feat2)) || !level_sigma2_1 || !level_sigma2_2)
        return ORBHIP_E_BADARG;
    *nmatches_out = 0;
    for (int i = 0; i < n1; i++) { matches12_out[i] = -1; points12_out[3 * i] = points12_out[3 * i + 1] = points12_out[3 * i + 2] = 0.f; }
    if (n1 == 0 || n2 == 0 || nnodes2 == 0) return ORBHIP_OK;
    const int mx = n1 > n2 ? n1 : n2;
    HostCall H(ctx);
    const int a_nid = H.in(nid1, 4 * (size_t)n1), a_mp1 = H.in(has_mp1, n1);
    const int a_ids = H.in(node_ids2, 4 * (size_t)nnodes2), a_st = H.in(node_start2, 4 * (size_t)(nnodes2 + 1)), a_fe = H.in(feat2, 4 * (size_t)node_start2[nnodes2]);
    const int a_mp2 = H.in(has_mp2, n2);
    const int a_kp1 = H.in(kp1, sizeof(orbhip_keypoint) * (size_t)n1, sizeof(orbhip_keypoint) * (size_t)mx);
    const int a_kp2 = H.in(kp2, sizeof(orbhip_keypoint) * (size_t)n2, sizeof(orbhip_keypoint) * (size_t)mx);
    const int a_d1 = H.in(desc1, 32 * (size_t)n1, 32 * (size_t)mx), a_d2 = H.in(desc2, 32 * (size_t)n2, 32 * (size_t)mx);
    const int a_pair = H.in(pair, sizeof(*pair)), a_poses = H.in(poses, sizeof(*poses)), a_n1 = H.in(&n1, 4), a_n2 = H.in(&n2, 4), a_nn = H.in(&nnodes2, 4);
    const int a_m = H.out(matches12_out, 4 * (size_t)n1, 4 * (size_t)mx), a_pt = H.out(points12_out, 12 * (size_t)n1, 12 * (size_t)mx), a_nm = H.out(nmatches_out, 4);
    if (int rc = H.commit()) return rc;
    const int rc = orbhip_match_and_triangulate_device(ctx, H.ptr<int32_t>(a_nid), H.ptr<uint8_t>(a_mp1), H.ptr<orbhip_keypoint>(a_kp1), H.ptr<uint8_t>(a_d1),
        H.ptr<int32_t>(a_n1), H.ptr<int32_t>(a_ids), H.ptr<int32_t>(a_st), H.ptr<int32_t>(a_fe), H.ptr<int32_t>(a_nn), H.ptr<uint8_t>(a_mp2),
        H.ptr<orbhip_keypoint>(a_kp2), H.ptr<uint8_t>(a_d2), H.ptr<int32_t>(a_n2), H.ptr<orbhip_tri_pair_general>(a_pair), H.ptr<orbhip_tri_pair_poses>(a_poses), 1,
        nnodes2, mx, (size_t)mx, level_sigma2_1, level_sigma2_2, nlevels, check_orientation, H.ptr<int32_t>(a_m), H.ptr<float>(a_pt), H.ptr<int32_t>(a_nm));
    if (rc) return rc;
    return H.finish();
}

extern "C" int orbhip_fuse_search_host(orbhip_ctx *ctx, const orbhip_proj_query *q, const uint8_t *desc_q, int nq, const orbhip_keypoint *kp,
                                       const uint8_t *desc, const float *u_right, int n, const float *inv_level_sigma2, int nlevels,
                                       float min_x, float min_y, float max_x, float max_y, int32_t *best_idx_out, int32_t *best_dist_out)
{
    if (!ctx || nq < 0 || n < 0 || (nq && (!q || !desc_q || !best_idx_out || !best_dist_out)) || (n && (!kp || !desc)) || !inv_level_sigma2 || nlevels <= 0)
        return ORBHIP_E_BADARG;
    for (int i = 0; i < nq; i++) { best_idx_out[i] = -1; best_dist_out[i] = 256; }
    if (nq == 0 || n == 0) return ORBHIP_OK;
    HostCall H(ctx);
    const int a_q = H.in(q, sizeof(orbhip_proj_query) * (size_t)nq), a_dq = H.in(desc_q, 32 * (size_t)nq), a_kp = H.in(kp, sizeof(orbhip_keypoint) * (size_t)n);
    const int a_d = H.in(desc, 32 * (size_t)n), a_ur = H.in(u_right, 4 * (size_t)n), a_nq = H.in(&nq, 4), a_n = H.in(&n, 4);
    const int a_bi = H.out(best_idx_out, 4 * (size_t)nq), a_bd = H.out(best_dist_out, 4 * (size_t)nq);
    if (int rc = H.commit()) return rc;
    const int rc = orbhip_fuse_search_device(ctx, H.ptr<orbhip_proj_query>(a_q), H.ptr<uint8_t>(a_dq), H.ptr<int32_t>(a_nq), nq, H.ptr<orbhip_keypoint>(a_kp),
                                             H.ptr<uint8_t>(a_d), u_right ? H.ptr<float>(a_ur) : nullptr, H.ptr<int32_t>(a_n), n, (size_t)n, 1, inv_level_sigma2,
                                             nlevels, min_x, min_y, max_x, max_y, H.ptr<int32_t>(a_bi), H.ptr<int32_t>(a_bd));
    if (rc) return rc;
    return H.finish();
}

extern "C" int orbhip_search_by_bow_kf_host(orbhip_ctx *ctx,
        const int32_t *node_ids1, const int32_t *node_start1, const int32_t *feat1, int nnodes1, const uint8_t *valid1,
        const orbhip_keypoint *kp1, const uint8_t *desc1, int n1,
        const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2, const uint8_t *valid2,
        const orbhip_keypoint *kp2, const uint8_t *desc2, int n2,
        float nn_ratio, int check_orientation, int32_t *matches12_out, int32_t *nmatches_out)
{
    if (!ctx || n1 < 0 || n2 < 0 || nnodes1 < 0 || nnodes2 < 0 || !nmatches_out || (n1 && !matches12_out) ||
        (nnodes1 && (!node_ids1 || !node_start1 || !feat1 || !valid1 || !kp1 || !desc1)) ||
        (nnodes2 && (!node_ids2 || !node_start2 || !feat2 || !valid2 || !kp2 || !desc2))) return ORBHIP_E_BADARG;
    *nmatches_out = 0;
    for (int i = 0; i < n1; i++) matches12_out[i] = -1;                 // vpMatches12 = vector<MapPoint*>(vpMapPoints1.size(), NULL), ORBmatcher.cc:839
    if (n1 == 0 || n2 == 0 || nnodes1 == 0 || nnodes2 == 0) return ORBHIP_OK;
    const int mn = nnodes1 > nnodes2 ? nnodes1 : nnodes2, mx = n1 > n2 ? n1 : n2;
    HostCall H(ctx);
    const int a_i1 = H.in(node_ids1, 4 * (size_t)nnodes1, 4 * (size_t)mn), a_s1 = H.in(node_start1, 4 * (size_t)(nnodes1 + 1), 4 * (size_t)(mn + 1));
    const int a_f1 = H.in(feat1, 4 * (size_t)node_start1[nnodes1], 4 * (size_t)mx), a_v1 = H.in(valid1, (size_t)n1, (size_t)mx);
    const int a_i2 = H.in(node_ids2, 4 * (size_t)nnodes2, 4 * (size_t)mn), a_s2 = H.in(node_start2, 4 * (size_t)(nnodes2 + 1), 4 * (size_t)(mn + 1));
    const int a_f2 = H.in(feat2, 4 * (size_t)node_start2[nnodes2], 4 * (size_t)mx), a_v2 = H.in(valid2, (size_t)n2, (size_t)mx);
    const int a_k1 = H.in(kp1, sizeof(orbhip_keypoint) * (size_t)n1, sizeof(orbhip_keypoint) * (size_t)mx), a_k2 = H.in(kp2, sizeof(orbhip_keypoint) * (size_t)n2, sizeof(orbhip_keypoint) * (size_t)mx);
    const int a_d1 = H.in(desc1, 32 * (size_t)n1, 32 * (size_t)mx), a_d2 = H.in(desc2, 32 * (size_t)n2, 32 * (size_t)mx);
    const int a_nn1 = H.in(&nnodes1, 4), a_nn2 = H.in(&nnodes2, 4), a_n1 = H.in(&n1, 4), a_n2 = H.in(&n2, 4);
    const int a_m = H.out(matches12_out, 4 * (size_t)n1, 4 * (size_t)mx), a_nm = H.out(nmatches_out, 4);
    if (int rc = H.commit()) return rc;
    const int rc = orbhip_search_by_bow_kf_device(ctx, H.ptr<int32_t>(a_i1), H.ptr<int32_t>(a_s1), H.ptr<int32_t>(a_f1), H.ptr<int32_t>(a_nn1), H.ptr<uint8_t>(a_v1),
        H.ptr<orbhip_keypoint>(a_k1), H.ptr<uint8_t>(a_d1), H.ptr<int32_t>(a_n1), H.ptr<int32_t>(a_i2), H.ptr<int32_t>(a_s2), H.ptr<int32_t>(a_f2), H.ptr<int32_t>(a_nn2),
        H.ptr<uint8_t>(a_v2), H.ptr<orbhip_keypoint>(a_k2), H.ptr<uint8_t>(a_d2), H.ptr<int32_t>(a_n2), 1, mn, mx, (size_t)mx, nn_ratio, check_orientation,
        H.ptr<int32_t>(a_m), H.ptr<int32_t>(a_nm));
    if (rc) return rc;
    return H.finish();
}

extern "C" int orbhip_pose_optimization_host(orbhip_ctx *ctx, const double *Xw, const double *obs, const double *inv_sigma2, int n,
                                             double fx, double fy, double cx, double cy, double bf, const double *kb8_k,
                                             const orbhip_camera2 *cam2, const uint8_t *right,
                                             double *pose_inout, uint8_t *outlier_out, int32_t *n_inliers_out, int32_t *stats_out)
{
    if (!ctx || n < 0 || !pose_inout || !n_inliers_out || (n && (!Xw || !obs || !inv_sigma2 || !outlier_out))) return ORBHIP_E_BADARG;
    *n_inliers_out = 0;
    if (stats_out) stats_out[0] = stats_out[1] = stats_out[2] = stats_out[3] = 0;
    if (n == 0) return ORBHIP_OK;
    HostCall H(ctx);
    const int a_x = H.in(Xw, 24 * (size_t)n), a_o = H.in(obs, 24 * (size_t)n), a_w = H.in(inv_sigma2, 8 * (size_t)n), a_n = H.in(&n, 4);
    const int a_r = H.in(right, (size_t)n), a_p = H.inout(pose_inout, 56), a_out = H.out(outlier_out, n), a_ni = H.out(n_inliers_out, 4), a_st = H.out(stats_out, stats_out ? 16 : 0, 16);
    if (int rc = H.commit()) return rc;
    const int rc = orbhip_pose_optimization_device(ctx, H.ptr<double>(a_x), H.ptr<double>(a_o), H.ptr<double>(a_w), H.ptr<int32_t>(a_n), 1, n, fx, fy, cx, cy, bf,
                                                   kb8_k, cam2, right ? H.ptr<uint8_t>(a_r) : nullptr, H.ptr<double>(a_p), H.ptr<uint8_t>(a_out), H.ptr<int32_t>(a_ni),
                                                   H.ptr<int32_t>(a_st));
    if (rc) return rc;
    return H.finish();
}
