// host_entry.hip -- host-pointer convenience forms of the matcher entry points (include/orbhip.h): what an ORBmatcher method
// with the reference's signature needs for ONE frame.  Upload into the context's grow-only device arena, run the same kernel
// as the batched device entry point, download, synchronise.  No computation happens on the host.
#include "orb_internal.h"
#include <cstring>
#include <vector>

hipStream_t orbhip_ctx_stream_internal(orbhip_ctx *c);
int orbhip_ctx_device_internal(orbhip_ctx *c);
void *orbhip_ctx_scratch_internal(orbhip_ctx *c, size_t bytes);
void orbhip_set_last_error_internal(const char *msg);

namespace {
struct Arena {
    uint8_t *base; size_t off, cap;
    template <typename T> T *take(size_t count) { off = (off + 255) & ~(size_t)255; T *p = reinterpret_cast<T *>(base + off); off += count * sizeof(T); return p; }
};
inline size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
#define HTRY(e) do { if ((e) != hipSuccess) { orbhip_set_last_error_internal(#e); return ORBHIP_E_HIP; } } while (0)
}  // namespace

static int sbp_host(orbhip_ctx *ctx, int mode, const orbhip_proj_query *q, const uint8_t *desc_q, int nq,
                    const orbhip_keypoint *kp, const uint8_t *desc, const float *u_right, int n, int nleft, const int32_t *mirror,
                    float min_x, float min_y, float max_x, float max_y, int th_high, float nn_ratio,
                    int check_orientation, int32_t *train_match_inout, int32_t *nmatches_out);

extern "C" int orbhip_search_by_projection_host(orbhip_ctx *ctx, int mode, const orbhip_proj_query *q, const uint8_t *desc_q, int nq,
                                                const orbhip_keypoint *kp, const uint8_t *desc, const float *u_right, int n,
                                                float min_x, float min_y, float max_x, float max_y, int th_high, float nn_ratio,
                                                int check_orientation, int32_t *train_match_inout, int32_t *nmatches_out)
{
    return sbp_host(ctx, mode, q, desc_q, nq, kp, desc, u_right, n, -1, nullptr, min_x, min_y, max_x, max_y, th_high, nn_ratio, check_orientation,
                    train_match_inout, nmatches_out);
}

extern "C" int orbhip_search_by_projection_rig_host(orbhip_ctx *ctx, int mode, const orbhip_proj_query *q, const uint8_t *desc_q, int nq,
                                                    const orbhip_keypoint *kp, const uint8_t *desc, int n, int nleft, const int32_t *mirror,
                                                    float min_x, float min_y, float max_x, float max_y, int th_high, float nn_ratio,
                                                    int check_orientation, int32_t *train_match_inout, int32_t *nmatches_out)
{
    if (nleft < 0 || nleft > n) return ORBHIP_E_BADARG;
    return sbp_host(ctx, mode, q, desc_q, nq, kp, desc, nullptr, n, nleft, mirror, min_x, min_y, max_x, max_y, th_high, nn_ratio, check_orientation,
                    train_match_inout, nmatches_out);
}

static int sbp_host(orbhip_ctx *ctx, int mode, const orbhip_proj_query *q, const uint8_t *desc_q, int nq,
                    const orbhip_keypoint *kp, const uint8_t *desc, const float *u_right, int n, int nleft, const int32_t *mirror,
                    float min_x, float min_y, float max_x, float max_y, int th_high, float nn_ratio,
                    int check_orientation, int32_t *train_match_inout, int32_t *nmatches_out)
{
    if (!ctx || nq < 0 || n < 0 || (nq && (!q || !desc_q)) || (n && (!kp || !desc || !train_match_inout)) || !nmatches_out || (mode != 0 && mode != 1))
        return ORBHIP_E_BADARG;
    *nmatches_out = 0;
    if (nq == 0 || n == 0) return ORBHIP_OK;
    HTRY(hipSetDevice(orbhip_ctx_device_internal(ctx)));
    hipStream_t s = orbhip_ctx_stream_internal(ctx);
    const size_t need = al(sizeof(orbhip_proj_query) * nq) + al(32 * (size_t)nq) + al(sizeof(orbhip_keypoint) * n) + al(32 * (size_t)n) +
                        al(4 * (size_t)n) * 3 + 5 * 256;
    Arena A = {(uint8_t *)orbhip_ctx_scratch_internal(ctx, need), 0, need};
    if (!A.base) return ORBHIP_E_HIP;
    orbhip_proj_query *dq = A.take<orbhip_proj_query>(nq); uint8_t *ddq = A.take<uint8_t>(32 * (size_t)nq);
    orbhip_keypoint *dkp = A.take<orbhip_keypoint>(n); uint8_t *dd = A.take<uint8_t>(32 * (size_t)n);
    float *dur = A.take<float>(n); int32_t *dtm = A.take<int32_t>(n), *dmi = A.take<int32_t>(n);
    int32_t *dnq = A.take<int32_t>(1), *dn = A.take<int32_t>(1), *dnm = A.take<int32_t>(1), *dnl = A.take<int32_t>(1);
    HTRY(hipMemcpyAsync(dq, q, sizeof(orbhip_proj_query) * nq, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(ddq, desc_q, 32 * (size_t)nq, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dkp, kp, sizeof(orbhip_keypoint) * n, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dd, desc, 32 * (size_t)n, hipMemcpyHostToDevice, s));
    if (u_right) HTRY(hipMemcpyAsync(dur, u_right, 4 * (size_t)n, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dtm, train_match_inout, 4 * (size_t)n, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dnq, &nq, 4, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dn, &n, 4, hipMemcpyHostToDevice, s));
    int rc;
    if (nleft >= 0) {
        HTRY(hipMemcpyAsync(dnl, &nleft, 4, hipMemcpyHostToDevice, s));
        if (mirror) HTRY(hipMemcpyAsync(dmi, mirror, 4 * (size_t)n, hipMemcpyHostToDevice, s));
        rc = orbhip_search_by_projection_rig_device(ctx, mode, dq, ddq, dnq, nq, dkp, dd, dn, dnl, mirror ? dmi : nullptr, n, (size_t)n, 1, min_x, min_y,
                                                    max_x, max_y, th_high, nn_ratio, check_orientation, dtm, dnm);
    } else if (mode == 0)
        rc = orbhip_search_by_projection_device(ctx, dq, ddq, dnq, nq, dkp, dd, u_right ? dur : nullptr, dn, n, (size_t)n, 1, min_x, min_y, max_x,
                                                max_y, th_high, check_orientation, dtm, dnm);
    else
        rc = orbhip_search_local_map_device(ctx, dq, ddq, dnq, nq, dkp, dd, u_right ? dur : nullptr, dn, n, (size_t)n, 1, min_x, min_y, max_x,
                                            max_y, th_high, nn_ratio, dtm, dnm);
    if (rc) return rc;
    HTRY(hipMemcpyAsync(train_match_inout, dtm, 4 * (size_t)n, hipMemcpyDeviceToHost, s));
    HTRY(hipMemcpyAsync(nmatches_out, dnm, 4, hipMemcpyDeviceToHost, s));
    return orbhip_ctx_check_status(ctx);               // synchronises; ORBHIP_E_CAPACITY when the frame exceeds the kernel's limits
}

extern "C" int orbhip_search_for_initialization_host(orbhip_ctx *ctx, const orbhip_keypoint *kpA, const uint8_t *descA, int nA,
                                                     const orbhip_keypoint *kpB, const uint8_t *descB, int nB, float min_x, float min_y,
                                                     float max_x, float max_y, int window_size, float nn_ratio, int check_orientation,
                                                     float *prev_matched_inout, int32_t *matches12_out, int32_t *nmatches_out)
{
    if (!ctx || nA < 0 || nB < 0 || (nA && (!kpA || !descA || !prev_matched_inout || !matches12_out)) || (nB && (!kpB || !descB)) || !nmatches_out)
        return ORBHIP_E_BADARG;
    *nmatches_out = 0;
    for (int i = 0; i < nA; i++) matches12_out[i] = -1;                 // vnMatches12 = vector<int>(F1.mvKeysUn.size(), -1), ORBmatcher.cc:713
    if (nA == 0 || nB == 0) return ORBHIP_OK;
    HTRY(hipSetDevice(orbhip_ctx_device_internal(ctx)));
    hipStream_t s = orbhip_ctx_stream_internal(ctx);
    const int mx = nA > nB ? nA : nB;
    const size_t need = 2 * (al(sizeof(orbhip_keypoint) * mx) + al(32 * (size_t)mx)) + al(8 * (size_t)mx) + al(4 * (size_t)mx) + 4 * 256;
    Arena A = {(uint8_t *)orbhip_ctx_scratch_internal(ctx, need), 0, need};
    if (!A.base) return ORBHIP_E_HIP;
    orbhip_keypoint *dka = A.take<orbhip_keypoint>(mx), *dkb = A.take<orbhip_keypoint>(mx);
    uint8_t *dda = A.take<uint8_t>(32 * (size_t)mx), *ddb = A.take<uint8_t>(32 * (size_t)mx);
    float *dpm = A.take<float>(2 * (size_t)mx); int32_t *dm12 = A.take<int32_t>(mx);
    int32_t *dna = A.take<int32_t>(1), *dnb = A.take<int32_t>(1), *dnm = A.take<int32_t>(1);
    HTRY(hipMemcpyAsync(dka, kpA, sizeof(orbhip_keypoint) * nA, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dda, descA, 32 * (size_t)nA, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dkb, kpB, sizeof(orbhip_keypoint) * nB, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(ddb, descB, 32 * (size_t)nB, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dpm, prev_matched_inout, 8 * (size_t)nA, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dna, &nA, 4, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dnb, &nB, 4, hipMemcpyHostToDevice, s));
    const int rc = orbhip_search_for_initialization_device(ctx, dka, dda, dna, dkb, ddb, dnb, 1, mx, (size_t)mx, min_x, min_y, max_x, max_y,
                                                           window_size, nn_ratio, check_orientation, dpm, dm12, dnm);
    if (rc) return rc;
    HTRY(hipMemcpyAsync(prev_matched_inout, dpm, 8 * (size_t)nA, hipMemcpyDeviceToHost, s));
    HTRY(hipMemcpyAsync(matches12_out, dm12, 4 * (size_t)nA, hipMemcpyDeviceToHost, s));
    HTRY(hipMemcpyAsync(nmatches_out, dnm, 4, hipMemcpyDeviceToHost, s));
    return orbhip_ctx_check_status(ctx);
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) for ONE (keyframe, frame) pair from host memory: the two flattened
// FeatureVectors (node ids ascending, node_start [nnodes + 1], feature indices), the keyframe's "map point exists and is not bad"
// flags, keypoints (mvKeysUn / mvKeys, concatenated left | right for rig frames) and descriptors.  nleft < 0: F.Nleft == -1.
extern "C" int orbhip_search_by_bow_host(orbhip_ctx *ctx,
        const int32_t *kf_node_ids, const int32_t *kf_node_start, const int32_t *kf_feat, int kf_nnodes, const uint8_t *kf_valid,
        const orbhip_keypoint *kf_kp, const uint8_t *kf_desc, int nK,
        const int32_t *f_node_ids, const int32_t *f_node_start, const int32_t *f_feat, int f_nnodes,
        const orbhip_keypoint *f_kp, const uint8_t *f_desc, int nF, int nleft,
        float nn_ratio, int check_orientation, int32_t *match_f_out, int32_t *nmatches_out)
{
    if (!ctx || nK < 0 || nF < 0 || kf_nnodes < 0 || f_nnodes < 0 || !nmatches_out || (nF && !match_f_out) ||
        (kf_nnodes && (!kf_node_ids || !kf_node_start || !kf_feat || !kf_valid || !kf_kp || !kf_desc)) ||
        (f_nnodes && (!f_node_ids || !f_node_start || !f_feat || !f_kp || !f_desc)) || nleft > nF)
        return ORBHIP_E_BADARG;
    *nmatches_out = 0;
    for (int i = 0; i < nF; i++) match_f_out[i] = -1;                   // vpMapPointMatches = vector<MapPoint*>(F.N, NULL), ORBmatcher.cc:277
    if (nK == 0 || nF == 0 || kf_nnodes == 0 || f_nnodes == 0) return ORBHIP_OK;
    HTRY(hipSetDevice(orbhip_ctx_device_internal(ctx)));
    hipStream_t s = orbhip_ctx_stream_internal(ctx);
    const int mn = kf_nnodes > f_nnodes ? kf_nnodes : f_nnodes, mx = nK > nF ? nK : nF;
    const size_t need = 2 * (al(4 * (size_t)mn) + al(4 * (size_t)(mn + 1)) + al(4 * (size_t)mx) + al(sizeof(orbhip_keypoint) * mx) + al(32 * (size_t)mx)) +
                        al(mx) + al(4 * (size_t)mx) + 8 * 256;
    Arena A = {(uint8_t *)orbhip_ctx_scratch_internal(ctx, need), 0, need};
    if (!A.base) return ORBHIP_E_HIP;
    int32_t *dki = A.take<int32_t>(mn), *dks = A.take<int32_t>(mn + 1), *dkf = A.take<int32_t>(mx);
    int32_t *dfi = A.take<int32_t>(mn), *dfs = A.take<int32_t>(mn + 1), *dff = A.take<int32_t>(mx);
    orbhip_keypoint *dkk = A.take<orbhip_keypoint>(mx), *dfk = A.take<orbhip_keypoint>(mx);
    uint8_t *dkd = A.take<uint8_t>(32 * (size_t)mx), *dfd = A.take<uint8_t>(32 * (size_t)mx), *dva = A.take<uint8_t>(mx);
    int32_t *dm = A.take<int32_t>(mx), *dkn = A.take<int32_t>(1), *dfn = A.take<int32_t>(1), *dnF = A.take<int32_t>(1), *dnl = A.take<int32_t>(1), *dnm = A.take<int32_t>(1);
    HTRY(hipMemcpyAsync(dki, kf_node_ids, 4 * (size_t)kf_nnodes, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dks, kf_node_start, 4 * (size_t)(kf_nnodes + 1), hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dkf, kf_feat, 4 * (size_t)kf_node_start[kf_nnodes], hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dfi, f_node_ids, 4 * (size_t)f_nnodes, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dfs, f_node_start, 4 * (size_t)(f_nnodes + 1), hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dff, f_feat, 4 * (size_t)f_node_start[f_nnodes], hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dkk, kf_kp, sizeof(orbhip_keypoint) * nK, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dfk, f_kp, sizeof(orbhip_keypoint) * nF, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dkd, kf_desc, 32 * (size_t)nK, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dfd, f_desc, 32 * (size_t)nF, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dva, kf_valid, nK, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dkn, &kf_nnodes, 4, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dfn, &f_nnodes, 4, hipMemcpyHostToDevice, s));
    HTRY(hipMemcpyAsync(dnF, &nF, 4, hipMemcpyHostToDevice, s));
    int rc;
    if (nleft >= 0) {
        HTRY(hipMemcpyAsync(dnl, &nleft, 4, hipMemcpyHostToDevice, s));
        rc = orbhip_search_by_bow_rig_device(ctx, dki, dks, dkf, dkn, dva, dkk, dkd, dfi, dfs, dff, dfn, dfk, dfd, dnF, dnl, 1, mn, mx, (size_t)mx, nn_ratio,
                                             check_orientation, dm, dnm);
    } else
        rc = orbhip_search_by_bow_device(ctx, dki, dks, dkf, dkn, dva, dkk, dkd, dfi, dfs, dff, dfn, dfk, dfd, dnF, 1, mn, mx, (size_t)mx, nn_ratio,
                                         check_orientation, dm, dnm);
    if (rc) return rc;
    HTRY(hipMemcpyAsync(match_f_out, dm, 4 * (size_t)nF, hipMemcpyDeviceToHost, s));
    HTRY(hipMemcpyAsync(nmatches_out, dnm, 4, hipMemcpyDeviceToHost, s));
    return orbhip_ctx_check_status(ctx);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Generic staging for the forms below: inputs / device-only buffers are declared first, then ONE arena of the summed size is taken
// (orbhip_ctx_scratch_internal may move the arena when it grows, so nothing is uploaded before the size is known).
namespace {
struct HostCall {
    orbhip_ctx *ctx; hipStream_t s;
    struct Item { const void *src; size_t bytes, off; };
    std::vector<Item> items; size_t total; uint8_t *base;
    explicit HostCall(orbhip_ctx *c) : ctx(c), s(orbhip_ctx_stream_internal(c)), total(0), base(nullptr) {}
    int in(const void *src, size_t bytes) { items.push_back({src, bytes, total}); total += al(bytes ? bytes : 1); return (int)items.size() - 1; }
    int buf(size_t bytes) { return in(nullptr, bytes); }
    int commit()
    {
        if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) { orbhip_set_last_error_internal("hipSetDevice"); return ORBHIP_E_HIP; }
        base = (uint8_t *)orbhip_ctx_scratch_internal(ctx, total + 256);
        if (!base) return ORBHIP_E_HIP;
        for (const Item &it : items)
            if (it.src && it.bytes && hipMemcpyAsync(base + it.off, it.src, it.bytes, hipMemcpyHostToDevice, s) != hipSuccess) {
                orbhip_set_last_error_internal("hipMemcpyAsync (host -> device staging)"); return ORBHIP_E_HIP;
            }
        return ORBHIP_OK;
    }
    template <typename T> T *ptr(int i) const { return reinterpret_cast<T *>(base + items[i].off); }
    int down(void *dst, int i, size_t bytes) const
    {
        if (!bytes) return ORBHIP_OK;
        if (hipMemcpyAsync(dst, base + items[i].off, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) { orbhip_set_last_error_internal("hipMemcpyAsync (device -> host)"); return ORBHIP_E_HIP; }
        return ORBHIP_OK;
    }
};
}  // namespace

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
    const int a_nid = H.in(nid1, 4 * (size_t)n1), a_mp1 = H.in(has_mp1, n1), a_ur1 = H.in(u_right1, u_right1 ? 4 * (size_t)n1 : 0);
    const int a_ids = H.in(node_ids2, 4 * (size_t)nnodes2), a_st = H.in(node_start2, 4 * (size_t)(nnodes2 + 1)), a_fe = H.in(feat2, 4 * (size_t)node_start2[nnodes2]);
    const int a_mp2 = H.in(has_mp2, n2), a_ur2 = H.in(u_right2, u_right2 ? 4 * (size_t)n2 : 0);
    const int a_kp1 = H.buf(sizeof(orbhip_keypoint) * (size_t)mx), a_kp2 = H.buf(sizeof(orbhip_keypoint) * (size_t)mx);      // rows of mx entries: one stride for both sides
    const int a_d1 = H.buf(32 * (size_t)mx), a_d2 = H.buf(32 * (size_t)mx);
    const int a_pair = H.in(pair, sizeof(*pair)), a_n1 = H.in(&n1, 4), a_n2 = H.in(&n2, 4), a_nn = H.in(&nnodes2, 4);
    const int a_m = H.buf(4 * (size_t)mx), a_nm = H.buf(4);
    if (int rc = H.commit()) return rc;
    HTRY(hipMemcpyAsync(H.ptr<void>(a_kp1), kp1, sizeof(orbhip_keypoint) * (size_t)n1, hipMemcpyHostToDevice, H.s));
    HTRY(hipMemcpyAsync(H.ptr<void>(a_kp2), kp2, sizeof(orbhip_keypoint) * (size_t)n2, hipMemcpyHostToDevice, H.s));
    HTRY(hipMemcpyAsync(H.ptr<void>(a_d1), desc1, 32 * (size_t)n1, hipMemcpyHostToDevice, H.s));
    HTRY(hipMemcpyAsync(H.ptr<void>(a_d2), desc2, 32 * (size_t)n2, hipMemcpyHostToDevice, H.s));
    const int rc = orbhip_search_for_triangulation_general_device(ctx, H.ptr<int32_t>(a_nid), H.ptr<uint8_t>(a_mp1), H.ptr<orbhip_keypoint>(a_kp1),
        H.ptr<uint8_t>(a_d1), u_right1 ? H.ptr<float>(a_ur1) : nullptr, H.ptr<int32_t>(a_n1), H.ptr<int32_t>(a_ids), H.ptr<int32_t>(a_st), H.ptr<int32_t>(a_fe),
        H.ptr<int32_t>(a_nn), H.ptr<uint8_t>(a_mp2), H.ptr<orbhip_keypoint>(a_kp2), H.ptr<uint8_t>(a_d2), u_right2 ? H.ptr<float>(a_ur2) : nullptr,
        H.ptr<int32_t>(a_n2), H.ptr<orbhip_tri_pair_general>(a_pair), 1, nnodes2, mx, (size_t)mx, level_sigma2_1, scale_factors2, level_sigma2_2, nlevels,
        check_orientation, H.ptr<int32_t>(a_m), H.ptr<int32_t>(a_nm));
    if (rc) return rc;
    if (int r2 = H.down(matches12_out, a_m, 4 * (size_t)n1)) return r2;
    if (int r2 = H.down(nmatches_out, a_nm, 4)) return r2;
    return orbhip_ctx_check_status(ctx);
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
    const int a_d = H.in(desc, 32 * (size_t)n), a_ur = H.in(u_right, u_right ? 4 * (size_t)n : 0), a_nq = H.in(&nq, 4), a_n = H.in(&n, 4);
    const int a_bi = H.buf(4 * (size_t)nq), a_bd = H.buf(4 * (size_t)nq);
    if (int rc = H.commit()) return rc;
    const int rc = orbhip_fuse_search_device(ctx, H.ptr<orbhip_proj_query>(a_q), H.ptr<uint8_t>(a_dq), H.ptr<int32_t>(a_nq), nq, H.ptr<orbhip_keypoint>(a_kp),
                                             H.ptr<uint8_t>(a_d), u_right ? H.ptr<float>(a_ur) : nullptr, H.ptr<int32_t>(a_n), n, (size_t)n, 1, inv_level_sigma2,
                                             nlevels, min_x, min_y, max_x, max_y, H.ptr<int32_t>(a_bi), H.ptr<int32_t>(a_bd));
    if (rc) return rc;
    if (int r2 = H.down(best_idx_out, a_bi, 4 * (size_t)nq)) return r2;
    if (int r2 = H.down(best_dist_out, a_bd, 4 * (size_t)nq)) return r2;
    return orbhip_ctx_check_status(ctx);
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
    const int a_i1 = H.buf(4 * (size_t)mn), a_s1 = H.buf(4 * (size_t)(mn + 1)), a_f1 = H.buf(4 * (size_t)mx), a_v1 = H.buf(mx);
    const int a_i2 = H.buf(4 * (size_t)mn), a_s2 = H.buf(4 * (size_t)(mn + 1)), a_f2 = H.buf(4 * (size_t)mx), a_v2 = H.buf(mx);
    const int a_k1 = H.buf(sizeof(orbhip_keypoint) * (size_t)mx), a_k2 = H.buf(sizeof(orbhip_keypoint) * (size_t)mx), a_d1 = H.buf(32 * (size_t)mx), a_d2 = H.buf(32 * (size_t)mx);
    const int a_nn1 = H.in(&nnodes1, 4), a_nn2 = H.in(&nnodes2, 4), a_n1 = H.in(&n1, 4), a_n2 = H.in(&n2, 4), a_m = H.buf(4 * (size_t)mx), a_nm = H.buf(4);
    if (int rc = H.commit()) return rc;
    struct Cp { int a; const void *src; size_t bytes; };
    const Cp cps[] = {{a_i1, node_ids1, 4 * (size_t)nnodes1}, {a_s1, node_start1, 4 * (size_t)(nnodes1 + 1)}, {a_f1, feat1, 4 * (size_t)node_start1[nnodes1]}, {a_v1, valid1, (size_t)n1},
                      {a_i2, node_ids2, 4 * (size_t)nnodes2}, {a_s2, node_start2, 4 * (size_t)(nnodes2 + 1)}, {a_f2, feat2, 4 * (size_t)node_start2[nnodes2]}, {a_v2, valid2, (size_t)n2},
                      {a_k1, kp1, sizeof(orbhip_keypoint) * (size_t)n1}, {a_k2, kp2, sizeof(orbhip_keypoint) * (size_t)n2}, {a_d1, desc1, 32 * (size_t)n1}, {a_d2, desc2, 32 * (size_t)n2}};
    for (const Cp &c : cps) if (c.bytes) HTRY(hipMemcpyAsync(H.ptr<void>(c.a), c.src, c.bytes, hipMemcpyHostToDevice, H.s));
    const int rc = orbhip_search_by_bow_kf_device(ctx, H.ptr<int32_t>(a_i1), H.ptr<int32_t>(a_s1), H.ptr<int32_t>(a_f1), H.ptr<int32_t>(a_nn1), H.ptr<uint8_t>(a_v1),
        H.ptr<orbhip_keypoint>(a_k1), H.ptr<uint8_t>(a_d1), H.ptr<int32_t>(a_n1), H.ptr<int32_t>(a_i2), H.ptr<int32_t>(a_s2), H.ptr<int32_t>(a_f2), H.ptr<int32_t>(a_nn2),
        H.ptr<uint8_t>(a_v2), H.ptr<orbhip_keypoint>(a_k2), H.ptr<uint8_t>(a_d2), H.ptr<int32_t>(a_n2), 1, mn, mx, (size_t)mx, nn_ratio, check_orientation,
        H.ptr<int32_t>(a_m), H.ptr<int32_t>(a_nm));
    if (rc) return rc;
    if (int r2 = H.down(matches12_out, a_m, 4 * (size_t)n1)) return r2;
    if (int r2 = H.down(nmatches_out, a_nm, 4)) return r2;
    return orbhip_ctx_check_status(ctx);
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
    const int a_r = H.in(right, right ? (size_t)n : 0), a_p = H.in(pose_inout, 56), a_out = H.buf(n), a_ni = H.buf(4), a_st = H.buf(16);
    if (int rc = H.commit()) return rc;
    const int rc = orbhip_pose_optimization_device(ctx, H.ptr<double>(a_x), H.ptr<double>(a_o), H.ptr<double>(a_w), H.ptr<int32_t>(a_n), 1, n, fx, fy, cx, cy, bf,
                                                   kb8_k, cam2, right ? H.ptr<uint8_t>(a_r) : nullptr, H.ptr<double>(a_p), H.ptr<uint8_t>(a_out), H.ptr<int32_t>(a_ni),
                                                   H.ptr<int32_t>(a_st));
    if (rc) return rc;
    if (int r2 = H.down(pose_inout, a_p, 56)) return r2;
    if (int r2 = H.down(outlier_out, a_out, n)) return r2;
    if (int r2 = H.down(n_inliers_out, a_ni, 4)) return r2;
    if (stats_out) if (int r2 = H.down(stats_out, a_st, 16)) return r2;
    return orbhip_ctx_check_status(ctx);
}
