// host_entry.hip -- host-pointer convenience forms of the matcher entry points (include/orbhip.h): what an ORBmatcher method
// with the reference's signature needs for ONE frame.  Upload into the context's grow-only device arena, run the same kernel
// as the batched device entry point, download, synchronise.  No computation happens on the host.
#include "orb_internal.h"
#include <cstring>

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
