// match_kernels.hip -- Hamming matching kernels (reference src/ORBmatcher.cc, src/Frame.cc).
//   M1  ORBmatcher::DescriptorDistance          ORBmatcher.cc:2353-2369
//   M2  all-pairs 2-NN + ratio                  Frame.cc:43,1146-1153 (cv::BFMatcher knnMatch k=2)
//   M3  Frame grid + GetFeaturesInArea          Frame.cc:377-408,645-726
//   M4  SearchForInitialization                 ORBmatcher.cc:710-825, ComputeThreeMaxima :2307-2348
// Integer popcount work: v_bcnt_u32_b32 on 8 dwords per pair; train descriptors are
// staged in LDS and read as wave-uniform (broadcast) 128-bit loads.
#include "orb_internal.h"
#include "wave_dpp.h"
#include <climits>
#include <cstring>
#include <string>

struct orbhip_ctx;
hipStream_t orbhip_ctx_stream_internal(orbhip_ctx *c);
void orbhip_set_last_error_internal(const char *msg);
int orbhip_ctx_device_internal(orbhip_ctx *c);
int32_t *orbhip_ctx_status_internal(orbhip_ctx *c);

// M1: host-callable scalar; same SWAR sequence as the reference (== sum of popcount32).
extern "C" int orbhip_descriptor_distance(const uint8_t *a32, const uint8_t *b32)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a32 + 4 * i, 4); memcpy(&pb, b32 + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

__device__ __forceinline__ int hamming256(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__device__ __forceinline__ int wave_incl_scan_i(int v) { return wave_scan_add_dpp(v); }      // DPP path (wave_dpp.h)

// ---------------------------------------------------------------------------- M2
// grid = (ceil(max_n/256), pairs); one query per thread; train tile of 256 descriptors in LDS.
#define BF_TILE 256
__global__ __launch_bounds__(256) void k_bf2nn(const uint8_t *descA, const int32_t *nA, size_t strideA,
                                               const uint8_t *descB, const int32_t *nB, size_t strideB,
                                               int max_n, double ratio, int32_t *idx2, int32_t *dist2, uint8_t *accept)
{
    __shared__ uint4 tile[BF_TILE * 2];
    const int pair = blockIdx.y, tid = threadIdx.x;
    const int na = nA[pair], nb = nB[pair];
    const int q = blockIdx.x * 256 + tid;
    if (blockIdx.x * 256 >= na) return;
    const uint4 *A = reinterpret_cast<const uint4 *>(descA + (size_t)pair * strideA);
    const uint4 *B = reinterpret_cast<const uint4 *>(descB + (size_t)pair * strideB);
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (q < na) { a0 = A[2 * q]; a1 = A[2 * q + 1]; }
    // best / second-best as keys (distance << 16 | train index): "first candidate wins on equal distance"
    // (strict < in the reference loop) is exactly the lexicographic order of the keys, so the running pair is
    // the two smallest keys -- three min/max per candidate instead of a compare-and-swap ladder
    uint32_t kb = 0xFFFFFFFFu, ks = 0xFFFFFFFFu;
    for (int t0 = 0; t0 < nb; t0 += BF_TILE) {
        const int tn = min(BF_TILE, nb - t0);
        __syncthreads();
        for (int i = tid; i < tn * 2; i += 256) tile[i] = B[2 * t0 + i];
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < tn; j++) {
            const uint32_t key = ((uint32_t)hamming256(a0, a1, tile[2 * j], tile[2 * j + 1]) << 16) | (uint32_t)(t0 + j);
            ks = min(ks, max(kb, key));
            kb = min(kb, key);
        }
    }
    if (q < na) {
        const int best = kb == 0xFFFFFFFFu ? INT_MAX : (int)(kb >> 16), second = ks == 0xFFFFFFFFu ? INT_MAX : (int)(ks >> 16);
        const int bi = kb == 0xFFFFFFFFu ? -1 : (int)(kb & 0xFFFFu), si = ks == 0xFFFFFFFFu ? -1 : (int)(ks & 0xFFFFu);
        const size_t o = ((size_t)pair * max_n + q) * 2;
        idx2[o] = bi; idx2[o + 1] = si; dist2[o] = best; dist2[o + 1] = second;
        // Frame.cc:1153: (*it).size() >= 2 && (*it)[0].distance < (*it)[1].distance * 0.7  (float < float*double)
        accept[(size_t)pair * max_n + q] = (si >= 0 && (double)(float)best < (double)(float)second * ratio) ? 1 : 0;
    }
}

// The same 2-NN search on the matrix cores.  With the query bits widened to -1 / +1 bytes (a' = 1 - 2a) and the train bits to 0 / 1 bytes (b),
// <a', b> = |b| - 2 <a, b>, so Hamming(a, b) = |a| + |b| - 2 <a, b> = |a| + <a', b>: v_mfma_i32_32x32x32_i8 started from C = |a| of the row
// leaves the Hamming distance in the accumulator (exact integers) and the key (distance << 16 | train index) is ONE v_lshl_add_u32 away.
// Descriptor matching is VALU-bound as xor + popcount (~21 vector instructions per pair and lane); here a 32 x 32 block of pairs costs
// 8 MFMAs plus 4 vector instructions per pair (accumulator read, key, v_med3 / v_min for the two smallest).  One workgroup = 8 waves
// x 32 queries of one pair of frames; the train side streams through LDS in tiles of 64 descriptors, widened once per workgroup
// (nibble * 0x00204081 & 0x01010101 puts 4 bits into 4 bytes), the next tile's fetch in flight behind this tile's MFMAs.
// A operand: lane l = (row l & 31, half l >> 5) holds bits [32 m + 16 h, +16) of its query for MFMA m; B likewise per train column, so
// element (h, j) of both operands is the same bit (the contraction index), whatever k the hardware assigns to it.  C: col = l & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).  Keys keep the reference's first-wins tie rule (cv::BFMatcher order).
#define BFM_ROWB 272                     // bytes per widened train descriptor in LDS (256 + 16: a 16-lane b128 read covers all banks once)
typedef int bfm_v4i __attribute__((ext_vector_type(4)));
typedef int bfm_v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ uint32_t bfm_widen4(uint32_t nib) { return __umul24(nib, 0x00204081u) & 0x01010101u; }      // full-rate 24-bit multiply (nib < 16, constant < 2^22)
__device__ __forceinline__ uint32_t bfm_widen4_pm(uint32_t nib)          // 4 bits -> 4 bytes: bit 1 -> -1, bit 0 -> +1 (once per query, not in the loop)
{
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) v |= (((nib >> k) & 1u) ? 0xFFu : 0x01u) << (8 * k);
    return v;
}
#define BFM_WAVES 8                      // waves (x 32 queries) per workgroup: the train tiles are widened once per workgroup
__global__ __launch_bounds__(64 * BFM_WAVES) void k_bf2nn_mfma(const uint8_t *descA, const int32_t *nA, size_t strideA,
                                                    const uint8_t *descB, const int32_t *nB, size_t strideB,
                                                    int max_n, double ratio, int32_t *idx2, int32_t *dist2, uint8_t *accept)
{
    __shared__ __attribute__((aligned(16))) uint8_t Bx[2 * 64 * BFM_ROWB];                 // two tile buffers of 64 x BFM_ROWB bytes (34 KB); at the end the merge area
    static_assert(BFM_WAVES * 2 * 8 * 64 * 4 <= 2 * 64 * BFM_ROWB, "the merge area (half of the rows at a time) fits the tile buffers");
    const int pair = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int na = nA[pair], nb = nB[pair];
    const int q0 = blockIdx.x * (32 * BFM_WAVES);
    if (q0 >= na) return;
    const uint32_t *A = reinterpret_cast<const uint32_t *>(descA + (size_t)pair * strideA);
    const uint32_t *B = reinterpret_cast<const uint32_t *>(descB + (size_t)pair * strideB);
    const int r = lane & 31, h = lane >> 5;
    // ---- the wave's 32 queries: operand fragments (8 MFMAs x 16 bytes of -1 / +1) and |a|
    const int qrow = q0 + 32 * wave + r;
    bfm_v4i af[8];
    int pa_row = 0;
    {
        uint32_t w[8];
#pragma unroll
        for (int m = 0; m < 8; m++) { w[m] = qrow < na ? A[(size_t)8 * qrow + m] : 0u; pa_row += __popc(w[m]); }
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const uint32_t hw = (w[m] >> (16 * h)) & 0xFFFFu;
            af[m] = (bfm_v4i){(int)bfm_widen4_pm(hw & 15u), (int)bfm_widen4_pm((hw >> 4) & 15u), (int)bfm_widen4_pm((hw >> 8) & 15u), (int)bfm_widen4_pm(hw >> 12)};
        }
    }
    // C input: |a| of the 16 rows this lane sees in the C layout (lanes 0..31 hold rows 0..31)
    bfm_v16i cinit;
#pragma unroll
    for (int g = 0; g < 16; g++) cinit[g] = __shfl(pa_row, (g & 3) + 8 * (g >> 2) + 4 * h, 64);
    uint32_t k1[16], k2[16];
#pragma unroll
    for (int g = 0; g < 16; g++) { k1[g] = 0xFFFFFFFFu; k2[g] = 0xFFFFFFFFu; }
    // ---- tiles of 64 train descriptors: thread t fetches dword t & 7 of descriptor t >> 3; the fetch of the NEXT tile is issued before
    //      this tile's MFMAs and widened into the other LDS buffer after them (its latency hides behind them)
    const int sc = tid >> 3, sm = tid & 7;
    auto fetch = [&](int t0, uint32_t &w0) { w0 = (t0 + sc < nb) ? B[(size_t)8 * (t0 + sc) + sm] : 0u; };
    auto widen = [&](int bufi, uint32_t w) {
        uint4 lo = make_uint4(bfm_widen4(w & 15u), bfm_widen4((w >> 4) & 15u), bfm_widen4((w >> 8) & 15u), bfm_widen4((w >> 12) & 15u));
        uint4 hi = make_uint4(bfm_widen4((w >> 16) & 15u), bfm_widen4((w >> 20) & 15u), bfm_widen4((w >> 24) & 15u), bfm_widen4(w >> 28));
        uint4 *dst = reinterpret_cast<uint4 *>(&Bx[bufi * 64 * BFM_ROWB + sc * BFM_ROWB + sm * 32]);
        dst[0] = lo; dst[1] = hi;
    };
    uint32_t nw0;
    fetch(0, nw0);
    widen(0, nw0);
    __syncthreads();
    int buf = 0;
    for (int t0 = 0; t0 < nb; t0 += 64, buf ^= 1) {
        const bool more = t0 + 64 < nb;
        if (more) fetch(t0 + 64, nw0);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if (t0 + 32 * half >= nb) break;                  // uniform
            bfm_v16i acc = cinit;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const bfm_v4i bf = *reinterpret_cast<const bfm_v4i *>(&Bx[buf * 64 * BFM_ROWB + (r + 32 * half) * BFM_ROWB + m * 32 + h * 16]);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[m], bf, acc, 0, 0, 0);
            }
            // acc = Hamming.  key = (acc << 16) + column (plain C: the compiler's hazard recogniser places the MFMA -> VALU wait states;
            // an inline-asm first reader would not get them); a column beyond the frame starts from 0x40000000, i.e. a distance no
            // descriptor can have (recognised at the end)
            const int col = t0 + 32 * half + r;
            const uint32_t base = col < nb ? (uint32_t)col : (0x40000000u | (uint32_t)col);
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const uint32_t key = ((uint32_t)acc[g] << 16) + base;
                asm("v_med3_u32 %0, %1, %2, %3" : "=v"(k2[g]) : "v"(k1[g]), "v"(key), "v"(k2[g]));      // k1 <= k2: the middle one is the new second best
                k1[g] = min(k1[g], key);
            }
        }
        if (more) widen(buf ^ 1, nw0);
        __syncthreads();
    }
    // ---- merge the 32 columns (lanes of one half) of every row: through LDS (the tile buffers are free now), one thread per row; rows
    //      0..15 (accumulator registers 0..7) first, then rows 16..31, so that the area is 32 KB and the kernel's LDS stays at the 34 KB of its
    //      tile buffers (64 KB until round 4: two workgroups filled 128 KB of a CU and no other kernel's workgroup could start beside them)
    uint32_t (*kout)[2][8][64] = reinterpret_cast<uint32_t (*)[2][8][64]>(&Bx[0]);           // [wave][k1 | k2][reg & 7][lane]
#pragma unroll
    for (int part = 0; part < 2; part++) {
        if (part) __syncthreads();
#pragma unroll
        for (int g = 0; g < 8; g++) { kout[wave][0][g][lane] = k1[8 * part + g]; kout[wave][1][g][lane] = k2[8 * part + g]; }
        __syncthreads();
        if (lane < 16) {
            const int row = 16 * part + lane, g = (row & 3) + 4 * ((row >> 3) & 1), hh = (row >> 2) & 1;
            uint32_t kb = 0xFFFFFFFFu, ks = 0xFFFFFFFFu;
            for (int c = 0; c < 32; c++) {
                const uint32_t a1 = kout[wave][0][g][32 * hh + c], a2 = kout[wave][1][g][32 * hh + c];
                ks = min(min(ks, a2), max(kb, a1));                                    // two smallest of {kb, ks, a1, a2} (a1 <= a2, kb <= ks)
                kb = min(kb, a1);
            }
            const int q = q0 + 32 * wave + row;
            if (q < na) {
                const bool hb = (kb >> 16) <= 256u, hs = (ks >> 16) <= 256u;            // a real descriptor distance (else: no such neighbour)
                const int best = hb ? (int)(kb >> 16) : INT_MAX, second = hs ? (int)(ks >> 16) : INT_MAX;
                const int bi = hb ? (int)(kb & 0xFFFFu) : -1, si = hs ? (int)(ks & 0xFFFFu) : -1;
                const size_t o = ((size_t)pair * max_n + q) * 2;
                idx2[o] = bi; idx2[o + 1] = si; dist2[o] = best; dist2[o + 1] = second;
                accept[(size_t)pair * max_n + q] = (si >= 0 && (double)(float)best < (double)(float)second * ratio) ? 1 : 0;
            }
        }
    }
}

extern "C" int orbhip_match_bf2nn_device(orbhip_ctx *ctx, const uint8_t *d_descA, const int32_t *d_nA, size_t strideA,
                                         const uint8_t *d_descB, const int32_t *d_nB, size_t strideB, int pairs,
                                         int max_n, double ratio, int32_t *d_idx2, int32_t *d_dist2, uint8_t *d_accept)
{
    if (!ctx || !d_descA || !d_descB || !d_nA || !d_nB || pairs <= 0 || max_n <= 0 || max_n > 65535 || !d_idx2 || !d_dist2 || !d_accept)
        return ORBHIP_E_BADARG;              // train indices ride in 16 bits of the 2-NN keys
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    if (max_n >= 64 && !getenv("ORBHIP_BF2NN_VALU")) {          // matrix-core form (the xor / popcount kernel stays for tiny frames and as a cross-check)
        dim3 grid((max_n + 32 * BFM_WAVES - 1) / (32 * BFM_WAVES), pairs);
        hipLaunchKernelGGL(k_bf2nn_mfma, grid, dim3(64 * BFM_WAVES), 0, orbhip_ctx_stream_internal(ctx), d_descA, d_nA, strideA, d_descB, d_nB,
                           strideB, max_n, ratio, d_idx2, d_dist2, d_accept);
    } else {
        dim3 grid((max_n + 255) / 256, pairs);
        hipLaunchKernelGGL(k_bf2nn, grid, dim3(256), 0, orbhip_ctx_stream_internal(ctx), d_descA, d_nA, strideA, d_descB, d_nB,
                           strideB, max_n, ratio, d_idx2, d_dist2, d_accept);
    }
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}

// ---------------------------------------------------------------------------- M3 + M4
// ORBmatcher::SearchForInitialization (ORBmatcher.cc:710-825) over the Frame grid
// (Frame.cc:377-408 AssignFeaturesToGrid, :716-726 PosInGrid, :645-714 GetFeaturesInArea).
// One wave per frame pair.  The F1 loop is inherently sequential (vMatchedDistance feeds
// later iterations, ORBmatcher.cc:749), but each iteration's best / second-best search over
// the window's grid cells is order-free except for the tie-break "first candidate in
// GetFeaturesInArea order wins" -- so the 64 lanes scan cells in parallel carrying
// (dist << 23 | visit order) keys and the wave reduces them.
#define SI_COLS 64            // FRAME_GRID_COLS (include/Frame.h:38)
#define SI_ROWS 48            // FRAME_GRID_ROWS (include/Frame.h:39)
#define SI_MAXN 8192          // keypoints per frame (the monocular-initialisation extractor runs 5 x nFeatures, Tracking.cc:210)
#define SI_CAP0 4096          // octave-0 keypoints per frame held in LDS
#define SI_RANKS 2048         // visit-order key: (cell visit index) * SI_RANKS + rank inside the cell  (3072 cells * 2048 < 2^23)
#define SI_TH_LOW 50          // ORBmatcher::TH_LOW  (ORBmatcher.cc:41)
#define SI_HISTO 30           // ORBmatcher::HISTO_LENGTH (ORBmatcher.cc:42)

// Only octave-0 keypoints take part (F1: ORBmatcher.cc:726-728; F2: GetFeaturesInArea(...,level1,level1)
// with level1 == 0), so both frames are first compacted to their octave-0 subsets in LDS.  The grid
// (Frame.cc:377-408) is kept implicitly: each F2 entry carries its cell (round(), Frame.cc:718-719) and
// its rank inside the cell; a keypoint that passes the |dx|<r,|dy|<r test always lies inside the
// floor/ceil cell range of Frame.cc:656-674 (round(a) is within [floor(b), ceil(c)] for b < a < c), so
// scanning the subset and ordering candidates by (cell visit index, rank) reproduces
// GetFeaturesInArea's list exactly.  The F1 loop stays sequential (vMatchedDistance, ORBmatcher.cc:749);
// per F1 point the lanes first collect candidates (LDS only), then evaluate one candidate per lane
// so all descriptor loads are in flight together.
// Register-resident sequential loop of SearchForInitialization (ORBmatcher.cc:726-790) for octave-0 subsets of at most 64 * NS points
// (level 0 holds ~0.22 x nFeatures keypoints: NS = 4 covers 1000 features, NS = 8 covers 2000).  Every lane keeps up to NS F2
// points -- position, grid cell + rank, descriptor, vMatchedDistance, vnMatches21 -- and NS F1 points in registers; F1 point t is
// broadcast from its lane with v_readlane, tests the window predicate on the F2 points directly and takes the Hamming distance:
// no candidate list, no LDS or global access and no barrier inside the chain (the general loop was 60 % candidate scan over
// LDS + 22 % descriptor fetches).  Same keys, same (best, second best), same update rules as the general loop.
template <int NS>
__device__ __forceinline__ int si_register_loop(int lane, int n0, int na0, const float *kx, const float *ky, const uint16_t *cellx, const uint16_t *celly,
                                                const uint16_t *cpos, const uint16_t *gidx, const uint16_t *aidx, uint16_t *bm,
                                                const uint4 *dA, const uint4 *dB, const float *prev, int32_t *m12,
                                                float min_x, float min_y, float inv_w, float inv_h, float r, float nn_ratio)
{
    int nmatches = 0;
    float fkx[NS], fky[NS]; uint32_t fpk[NS]; int fgi[NS], fmd[NS], fm21[NS]; uint4 fd0[NS], fd1[NS];
#pragma unroll
    for (int sl = 0; sl < NS; sl++) {
        const int li = sl * 64 + lane;
        const bool v = li < n0;
        const int lj = v ? li : 0;
        fkx[sl] = kx[lj]; fky[sl] = ky[lj];
        fpk[sl] = v ? ((uint32_t)cellx[lj] | ((uint32_t)celly[lj] << 8) | ((uint32_t)min((int)cpos[lj], SI_RANKS - 1) << 16)) : 0xFFFFu;   // cell (255, 255): never inside a window
        fgi[sl] = v ? (int)gidx[lj] : 0; fmd[sl] = INT_MAX; fm21[sl] = -1;      // (an empty slot must not index the descriptors with LDS garbage)
        fd0[sl] = dB[2 * fgi[sl]]; fd1[sl] = dB[2 * fgi[sl] + 1];
    }
    // the F1 points too: lane l holds points l, l + 64, ... (index, window centre, descriptor); the loop broadcasts point t
    // from its lane with v_readlane, so the sequential chain makes no memory access at all
    int qi[NS]; float qx[NS], qy[NS]; uint4 qd0[NS], qd1[NS];
#pragma unroll
    for (int sl = 0; sl < NS; sl++) {
        const int tq = sl * 64 + lane;
        qi[sl] = tq < na0 ? (int)aidx[tq] : 0;
        qx[sl] = prev[2 * qi[sl]]; qy[sl] = prev[2 * qi[sl] + 1];
        qd0[sl] = dA[2 * qi[sl]]; qd1[sl] = dA[2 * qi[sl] + 1];
    }
    for (int t = 0; t < na0; t++) {
        const int tsl = t >> 6, tln = t & 63;
        int i1 = 0; float x = 0, y = 0; uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
#define SI_RL(v) (uint32_t)__builtin_amdgcn_readlane((int)(v), tln)
#pragma unroll
        for (int sl = 0; sl < NS; sl++)
            if (tsl == sl) {                                            // uniform
                i1 = (int)SI_RL(qi[sl]); x = __uint_as_float(SI_RL(__float_as_uint(qx[sl]))); y = __uint_as_float(SI_RL(__float_as_uint(qy[sl])));
                a0 = make_uint4(SI_RL(qd0[sl].x), SI_RL(qd0[sl].y), SI_RL(qd0[sl].z), SI_RL(qd0[sl].w));
                a1 = make_uint4(SI_RL(qd1[sl].x), SI_RL(qd1[sl].y), SI_RL(qd1[sl].z), SI_RL(qd1[sl].w));
            }
#undef SI_RL
        int c0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, min_x), r), inv_w)); if (c0 < 0) c0 = 0;   // Frame.cc:656-674
        if (c0 >= SI_COLS) continue;
        int c1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, min_x), r), inv_w)); if (c1 > SI_COLS - 1) c1 = SI_COLS - 1;
        if (c1 < 0) continue;
        int r0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, min_y), r), inv_h)); if (r0 < 0) r0 = 0;
        if (r0 >= SI_ROWS) continue;
        int r1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, min_y), r), inv_h)); if (r1 > SI_ROWS - 1) r1 = SI_ROWS - 1;
        if (r1 < 0) continue;
        const int ncy = r1 - r0 + 1;
        // branch-free per slot: window predicate, distance, key (0xFFFFFFFF where the point is no candidate or already matched at
        // a distance <= this one, ORBmatcher.cc:749); per lane the smallest key and the distance of the runner-up
        uint32_t lk1 = 0xFFFFFFFFu, ld2 = 0xFFFFFFFFu; int best_sl = 0;
        const uint32_t cw = (uint32_t)(c1 - c0), rh = (uint32_t)(r1 - r0);
#pragma unroll
        for (int sl = 0; sl < NS; sl++) {
            if (sl * 64 >= n0) break;                                   // uniform
            const uint32_t dcx = (fpk[sl] & 255u) - (uint32_t)c0, dcy = ((fpk[sl] >> 8) & 255u) - (uint32_t)r0;
            const int dist = hamming256(a0, a1, fd0[sl], fd1[sl]);
            const bool c = dcx <= cw && dcy <= rh && fabsf(__fsub_rn(fkx[sl], x)) < r && fabsf(__fsub_rn(fky[sl], y)) < r && !(fmd[sl] <= dist);
            const uint32_t key = c ? ((uint32_t)dist << 23) | ((dcx * (uint32_t)ncy + dcy) * SI_RANKS + (fpk[sl] >> 16)) : 0xFFFFFFFFu;
            ld2 = min(ld2, max(lk1, key));                              // the larger of (best so far, new) is a runner-up
            best_sl = key < lk1 ? sl : best_sl;
            lk1 = min(lk1, key);
        }
        const uint32_t k1 = wave_min_u32_dpp(lk1);
        // second best over the wave: every lane's runner-up, and the best of every lane that is not the winner (keys are unique)
        const uint32_t k2 = wave_min_u32_dpp(lk1 == k1 ? ld2 : lk1);
        const int d2 = k2 == 0xFFFFFFFFu ? INT_MAX : (int)(k2 >> 23);
        if (k1 == 0xFFFFFFFFu) continue;                                // no candidate, or none below its vMatchedDistance
        const int best = (int)(k1 >> 23);
        if (!(best <= SI_TH_LOW && (float)best < __fmul_rn((float)d2, nn_ratio))) continue;   // ORBmatcher.cc:764-766
        const unsigned long long owner = __ballot(lk1 == k1);          // keys are unique: exactly one lane
        const int ol = __builtin_amdgcn_readfirstlane(__ffsll((long long)owner) - 1);
        const int bsl = __builtin_amdgcn_readlane(best_sl, ol);
        int old = -1, best_idx = 0;
#pragma unroll
        for (int sl = 0; sl < NS; sl++)
            if (bsl == sl) {                                            // uniform
                old = __builtin_amdgcn_readlane(fm21[sl], ol); best_idx = __builtin_amdgcn_readlane(fgi[sl], ol);
                if (lane == ol) { fm21[sl] = i1; fmd[sl] = best; }
            }
        if (old >= 0) nmatches--;                                       // ORBmatcher.cc:768-772
        nmatches++;
        if (lane == 0) {
            if (old >= 0) m12[old] = -1;
            m12[i1] = best_idx;
            bm[i1] = (uint16_t)best_idx;                                // its rotation-histogram entry is made after the loop
        }
    }
    return nmatches;
}

// Tail of SearchForInitialization shared by both forms: rotation consistency (ORBmatcher.cc:792-815, 2307-2348) and the vbPrevMatched
// update (:818-822).  bm[i1] = the F2 index an F1 keypoint was matched to at some time (0xFFFF: never); returns nmatches (lane 0's value counts).
__device__ __forceinline__ int si_tail(int lane, int n1, const orbhip_keypoint *kpA, const orbhip_keypoint *kpB, const uint16_t *bm, int8_t *bin_of,
                                       int *hist, int *s_keep, int32_t *m12, float *prev, int nmatches, int check_ori)
{
    const float factor = 1.0f / SI_HISTO;
    __syncthreads();
    // Every F1 point that was matched at some time has one histogram entry (ORBmatcher.cc:778-789), also when it was displaced later.
    if (check_ori) {
        for (int i1 = lane; i1 < n1; i1 += 64) {
            const int b = bm[i1];
            if (b == 0xFFFF) continue;
            float rot = __fsub_rn(kpA[i1].angle, kpB[b].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == SI_HISTO) bin = 0;
            atomicAdd(&hist[bin], 1); bin_of[i1] = (int8_t)bin;
        }
        __syncthreads();
        if (lane == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < SI_HISTO; i++) {
                const int sz = hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
                else if (sz > max3) { max3 = sz; ind3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
            s_keep[0] = ind1; s_keep[1] = ind2; s_keep[2] = ind3;
        }
        __syncthreads();
        int removed = 0;
        for (int i1 = lane; i1 < n1; i1 += 64) {
            const int b = bin_of[i1];
            if (b < 0 || b == s_keep[0] || b == s_keep[1] || b == s_keep[2]) continue;
            if (m12[i1] >= 0) { m12[i1] = -1; removed++; }
        }
        removed = wave_sum_dpp(removed);
        nmatches -= removed;                                                // lane 0's copy is the one written out
    }
    __syncthreads();
    for (int i1 = lane; i1 < n1; i1 += 64)                                  // ORBmatcher.cc:818-822
        if (m12[i1] >= 0) { prev[2 * i1] = kpB[m12[i1]].x; prev[2 * i1 + 1] = kpB[m12[i1]].y; }
    return nmatches;
}

#ifdef SI_PROF
__device__ long long g_si_prof[8];            // debug build only (EXTRA=-DSI_PROF): cycles of setup / candidate scan / distances + reduction / update / tail, iterations
#define SI_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = clock64(); g_si_prof[i] += t_ - t_prev; t_prev = t_; } } while (0)
extern "C" int orbhip_debug_si_prof(long long *out8, int reset)
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_si_prof), 64) != hipSuccess) return ORBHIP_E_HIP;
    if (reset) { long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_si_prof), z, 64) != hipSuccess) return ORBHIP_E_HIP; }
    return ORBHIP_OK;
}
#else
#define SI_T(i) do { } while (0)
#endif
__global__ __launch_bounds__(64) void k_search_init(const orbhip_keypoint *kpA_, const uint8_t *descA_, const int32_t *nA,
                                                    const orbhip_keypoint *kpB_, const uint8_t *descB_, const int32_t *nB,
                                                    int max_n, size_t kp_stride, float min_x, float min_y, float max_x, float max_y,
                                                    int window, float nn_ratio, int check_ori, int cap0, int maxn,
                                                    float *prev_, int32_t *m12_, int32_t *nmatches_, int32_t *status, const int32_t *redo_,
                                                    int32_t *redo_out)
{
    // redo_: only the flagged pairs are done (the others were finished by the replay form, or by the small-LDS launch of this kernel);
    // redo_out: this launch carves its LDS for cap0 octave-0 points only -- a pair with more is flagged there and left to the next launch
    if (redo_ && !redo_[blockIdx.x]) { if (redo_out && threadIdx.x == 0) redo_out[blockIdx.x] = 0; return; }
    // dynamic LDS, carved by the launcher's capacities: cap0 octave-0 entries per frame, maxn keypoints per frame
    extern __shared__ __attribute__((aligned(16))) uint8_t si_lds[];
    float *kx = reinterpret_cast<float *>(si_lds), *ky = kx + cap0;
    int *matched_dist = reinterpret_cast<int *>(ky + cap0);
    uint32_t *cand_key = reinterpret_cast<uint32_t *>(matched_dist + cap0);
    uint16_t *cellx = reinterpret_cast<uint16_t *>(cand_key + cap0), *celly = cellx + cap0, *cpos = celly + cap0, *gidx = cpos + cap0;
    int16_t *m21 = reinterpret_cast<int16_t *>(gidx + cap0);
    uint16_t *aidx = reinterpret_cast<uint16_t *>(m21 + cap0), *cand_li = aidx + cap0;
    uint16_t *bm = cand_li + cap0;                                  // [maxn] F2 index an F1 keypoint was matched to (0xFFFF = never)
    int8_t *bin_of = reinterpret_cast<int8_t *>(bm + maxn);
    __shared__ int hist[SI_HISTO];
    __shared__ int s_keep[3];
    const int pair = blockIdx.x, lane = threadIdx.x;
#ifdef SI_PROF
    long long t_prev = clock64();
#endif
    const unsigned long long lt_mask = (1ull << lane) - 1;
    const int n1 = nA[pair], n2 = nB[pair];
    const orbhip_keypoint *kpA = kpA_ + (size_t)pair * kp_stride, *kpB = kpB_ + (size_t)pair * kp_stride;
    const uint4 *dA = reinterpret_cast<const uint4 *>(descA_ + (size_t)pair * kp_stride * 32);
    const uint4 *dB = reinterpret_cast<const uint4 *>(descB_ + (size_t)pair * kp_stride * 32);
    float *prev = prev_ + (size_t)pair * max_n * 2;
    int32_t *m12 = m12_ + (size_t)pair * max_n;
    if (n1 > maxn || n2 > maxn) { if (lane == 0) { atomicExch(status, ORBHIP_E_CAPACITY); nmatches_[pair] = 0; } return; }
    const float inv_w = __fdiv_rn((float)SI_COLS, __fsub_rn(max_x, min_x));       // Frame.cc:334-335
    const float inv_h = __fdiv_rn((float)SI_ROWS, __fsub_rn(max_y, min_y));
    for (int i = lane; i < SI_HISTO; i += 64) hist[i] = 0;
    // ---- octave-0 subset of F2 that PosInGrid accepts, index order kept (= insertion order of the grid)
    int n0 = 0;
    for (int i0 = 0; i0 < n2; i0 += 64) {
        const int i = i0 + lane;
        bool in = false; int px = 0, py = 0; float fx = 0, fy = 0;
        if (i < n2) {
            const orbhip_keypoint k = kpB[i];
            fx = k.x; fy = k.y;
            px = (int)roundf(__fmul_rn(__fsub_rn(fx, min_x), inv_w));             // round(), Frame.cc:718-719
            py = (int)roundf(__fmul_rn(__fsub_rn(fy, min_y), inv_h));
            in = k.octave == 0 && px >= 0 && px < SI_COLS && py >= 0 && py < SI_ROWS;
        }
        const unsigned long long bal = __ballot(in);
        const int li = n0 + __popcll(bal & lt_mask);
        if (in && li < cap0) {
            kx[li] = fx; ky[li] = fy; cellx[li] = (uint16_t)px; celly[li] = (uint16_t)py; gidx[li] = (uint16_t)i;
            matched_dist[li] = INT_MAX; m21[li] = -1;
        }
        n0 += __popcll(bal);
    }
    // ---- octave-0 subset of F1 (ORBmatcher.cc:726-728)
    int na0 = 0;
    for (int i0 = 0; i0 < n1; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < n1 && kpA[i].octave == 0;
        const unsigned long long bal = __ballot(in);
        const int li = na0 + __popcll(bal & lt_mask);
        if (in && li < cap0) aidx[li] = (uint16_t)i;
        na0 += __popcll(bal);
        if (i < n1) { m12[i] = -1; bin_of[i] = -1; bm[i] = 0xFFFFu; }
    }
    if (redo_out) {
        if (lane == 0) redo_out[pair] = (n0 > cap0 || na0 > cap0) ? 1 : 0;
        if (n0 > cap0 || na0 > cap0) return;                                    // (m12 was reset, prev is untouched: the next launch starts over)
    }
    if (n0 > cap0 || na0 > cap0) { if (lane == 0) { atomicExch(status, ORBHIP_E_CAPACITY); nmatches_[pair] = 0; } return; }
    __syncthreads();
    for (int li = lane; li < n0; li += 64) {                                    // rank inside the grid cell
        const int cx = cellx[li], cy = celly[li];
        int rank = 0;
        for (int j = 0; j < li; j++) rank += (cellx[j] == cx && celly[j] == cy);
        cpos[li] = (uint16_t)rank;
    }
    __syncthreads();
    SI_T(0);
    // ---- sequential F1 loop
    int nmatches = 0;
    const float r = (float)window;
    if (n0 <= 4 * 64 && na0 <= 4 * 64)
        nmatches = si_register_loop<4>(lane, n0, na0, kx, ky, cellx, celly, cpos, gidx, aidx, bm, dA, dB, prev, m12, min_x, min_y, inv_w, inv_h, r, nn_ratio);
    else if (n0 <= 8 * 64 && na0 <= 8 * 64)
        nmatches = si_register_loop<8>(lane, n0, na0, kx, ky, cellx, celly, cpos, gidx, aidx, bm, dA, dB, prev, m12, min_x, min_y, inv_w, inv_h, r, nn_ratio);
    else {
    // the next F1 point's window centre and descriptor are fetched one iteration ahead
    int i1n = na0 > 0 ? (int)aidx[0] : 0;
    float xn = 0, yn = 0; uint4 pd0 = make_uint4(0, 0, 0, 0), pd1 = pd0;
    if (na0 > 0) { xn = prev[2 * i1n]; yn = prev[2 * i1n + 1]; pd0 = dA[2 * i1n]; pd1 = dA[2 * i1n + 1]; }
    for (int t = 0; t < na0; t++) {
        const int i1 = i1n;
        const float x = xn, y = yn;
        const uint4 a0 = pd0, a1 = pd1;
        if (t + 1 < na0) { i1n = aidx[t + 1]; xn = prev[2 * i1n]; yn = prev[2 * i1n + 1]; pd0 = dA[2 * i1n]; pd1 = dA[2 * i1n + 1]; }
        int c0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, min_x), r), inv_w)); if (c0 < 0) c0 = 0;   // Frame.cc:656-674
        if (c0 >= SI_COLS) continue;
        int c1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, min_x), r), inv_w)); if (c1 > SI_COLS - 1) c1 = SI_COLS - 1;
        if (c1 < 0) continue;
        int r0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, min_y), r), inv_h)); if (r0 < 0) r0 = 0;
        if (r0 >= SI_ROWS) continue;
        int r1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, min_y), r), inv_h)); if (r1 > SI_ROWS - 1) r1 = SI_ROWS - 1;
        if (r1 < 0) continue;
        const int ncy = r1 - r0 + 1;
        // phase A: collect candidates (GetFeaturesInArea's result set, any order, with their visit-order key)
        int ncand = 0;
        for (int l0 = 0; l0 < n0; l0 += 64) {
            const int li = l0 + lane;
            bool c = false; uint32_t key = 0;
            if (li < n0) {
                const int cx = cellx[li], cy = celly[li];
                c = cx >= c0 && cx <= c1 && cy >= r0 && cy <= r1 &&
                    fabsf(__fsub_rn(kx[li], x)) < r && fabsf(__fsub_rn(ky[li], y)) < r;
                key = (uint32_t)(((cx - c0) * ncy + (cy - r0)) * SI_RANKS + min((int)cpos[li], SI_RANKS - 1));
            }
            const unsigned long long bal = __ballot(c);
            if (c) { const int o = ncand + __popcll(bal & lt_mask); cand_li[o] = (uint16_t)li; cand_key[o] = key; }
            ncand += __popcll(bal);
        }
        SI_T(1);
        if (ncand == 0) continue;                                           // vIndices2.empty(), ORBmatcher.cc:732-733
        __syncthreads();
        // phase B: one candidate per lane
        uint32_t lk1 = 0xFFFFFFFFu; int d2 = INT_MAX, best_li = -1;
        for (int q0 = 0; q0 < ncand; q0 += 64) {
            const int q = q0 + lane;
            if (q < ncand) {
                const int li = cand_li[q];
                const int i2 = gidx[li];
                const int dist = hamming256(a0, a1, dB[2 * i2], dB[2 * i2 + 1]);
                if (!(matched_dist[li] <= dist)) {                          // ORBmatcher.cc:749
                    const uint32_t key = ((uint32_t)dist << 23) | cand_key[q];
                    if (key < lk1) { if (lk1 != 0xFFFFFFFFu) d2 = min(d2, (int)(lk1 >> 23)); lk1 = key; best_li = li; }
                    else d2 = min(d2, dist);
                }
            }
        }
        uint32_t k1 = lk1;
        // (best key, second-best distance) over the wave on the DPP path (scan order; the operation is associative and commutative)
#define SI_STEP(CTRL, RM) do { const uint32_t ok1 = (uint32_t)dpp_mov<CTRL, RM>(-1, (int)k1); const int od2 = dpp_mov<CTRL, RM>(INT_MAX, d2); \
            const uint32_t lose = max(k1, ok1); d2 = min(min(d2, od2), lose == 0xFFFFFFFFu ? INT_MAX : (int)(lose >> 23)); k1 = min(k1, ok1); } while (0)
        SI_STEP(ORB_DPP_ROW_SHR(1), 0xF); SI_STEP(ORB_DPP_ROW_SHR(2), 0xF); SI_STEP(ORB_DPP_ROW_SHR(4), 0xF); SI_STEP(ORB_DPP_ROW_SHR(8), 0xF);
        SI_STEP(ORB_DPP_ROW_BCAST15, 0xA); SI_STEP(ORB_DPP_ROW_BCAST31, 0xC);
#undef SI_STEP
        k1 = (uint32_t)__builtin_amdgcn_readlane((int)k1, 63); d2 = __builtin_amdgcn_readlane(d2, 63);
        __syncthreads();                                                    // cand_* reused by the next F1 point
        SI_T(2);
#ifdef SI_PROF
        if (blockIdx.x == 0 && threadIdx.x == 0) { g_si_prof[5] += 1; g_si_prof[6] += ncand; }
#endif
        if (k1 == 0xFFFFFFFFu) continue;                                    // bestDist stays INT_MAX > TH_LOW
        const int best = (int)(k1 >> 23);
        if (!(best <= SI_TH_LOW && (float)best < __fmul_rn((float)d2, nn_ratio))) continue;   // ORBmatcher.cc:764-766
        const unsigned long long owner = __ballot(lk1 == k1);              // keys are unique: exactly one lane
        const int bli = __builtin_amdgcn_readlane(best_li, __builtin_amdgcn_readfirstlane(__ffsll((long long)owner) - 1));
        if (lane == 0) {
            const int best_idx = gidx[bli];
            const int old = m21[bli];
            if (old >= 0) { m12[old] = -1; nmatches--; }                    // ORBmatcher.cc:768-772
            m12[i1] = best_idx; m21[bli] = (int16_t)i1; matched_dist[bli] = best; nmatches++;
            bm[i1] = (uint16_t)best_idx;                                    // its rotation-histogram entry is made after the loop
        }
        __syncthreads();
        SI_T(3);
    }
    }
    nmatches = si_tail(lane, n1, kpA, kpB, bm, bin_of, hist, s_keep, m12, prev, nmatches, check_ori);
    if (lane == 0) nmatches_[pair] = nmatches;
    SI_T(4);
}

// ---------------------------------------------------------------------------------------------------------------------
// Replay form of SearchForInitialization (round 3).  The F1 loop is sequential only through vMatchedDistance
// (ORBmatcher.cc:749): a candidate is skipped when its F2 point already holds a match at a distance <= this one.  Which F2 points lie
// in a query's window, their Hamming distances and GetFeaturesInArea's visiting order do not depend on it.  So:
//   k_si_prep        one wave per pair: the octave-0 subsets of both frames (the only points that take part) written out compactly --
//                    F2: position, grid cell + rank inside the cell, frame index, descriptor; F1: frame indices;
//   k_si_candidates  ONE WAVE PER F1 POINT over the whole chip: every F2 subset point is tested against the window, keys
//                    distance << 23 | visit order (the sequential kernels' key) are ranked by counting and the SIL_K smallest stored with
//                    their subset positions (transposed: lane = query of a 64-query trip);
//   k_si_replay      one wave per pair, 64 queries per trip: a lane's best / second best are the first two list entries NOT skipped by
//                    the current vMatchedDistance -- it only ever decreases, so a skipped entry stays skipped and a cursor only moves forward.
//                    Lanes speculate; an accepting lane publishes its F2 point (LDS atomicMin of the lane id); the prefix of lanes up to the
//                    first one whose best or second best an earlier lane of the round wants becomes final and applies its updates
//                    (distinct F2 points inside a prefix by construction), the rest look again.
// Same candidates, same keys, same update rule as k_search_init; both run in every test (si_form fixture).  A pair whose subsets exceed the
// work area, or a query whose truncated list runs dry before a best AND a second best were found, is flagged on the device and done by
// k_search_init, which returns at once for the others.
void *orbhip_ctx_work_internal(orbhip_ctx *c, size_t bytes);
#define SIL_K 64                   // list entries kept per query
#define SIL_BUF 512                // candidates a query may have before its pair falls back
#define SIL_CAP 1024               // octave-0 points per frame the replay form handles
struct SiWork {
    int cap0, chunks;
    float4 *rec;                   // [pairs][cap0]  F2 octave-0 subset: x, y, bits(cell x | cell y << 8 | rank << 16), bits(frame index)
    uint4 *desc;                   // [pairs][cap0][2]
    uint16_t *aidx;                // [pairs][cap0]  F1 octave-0 subset -> frame index
    int32_t *n0, *na0;             // [pairs]
    uint32_t *lkey;                // [pairs][chunks][SIL_K][64]
    uint16_t *lli;                 // [pairs][chunks][SIL_K][64]
    int32_t *count;                // [pairs][cap0]  candidates of the query (may exceed SIL_K; INT_MAX: more than SIL_BUF)
    int32_t *redo;                 // [pairs]
};

__global__ __launch_bounds__(64) void k_si_prep(const orbhip_keypoint *kpA_, const int32_t *nA, const orbhip_keypoint *kpB_, const uint8_t *descB_, const int32_t *nB,
                                                size_t kp_stride, float min_x, float min_y, float max_x, float max_y, int maxn, SiWork W)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sip_lds[];
    uint16_t *cellx = reinterpret_cast<uint16_t *>(sip_lds), *celly = cellx + W.cap0, *gidx = celly + W.cap0;
    const int pair = blockIdx.x, lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1;
    const int n1 = nA[pair], n2 = nB[pair];
    const orbhip_keypoint *kpA = kpA_ + (size_t)pair * kp_stride, *kpB = kpB_ + (size_t)pair * kp_stride;
    const uint4 *dB = reinterpret_cast<const uint4 *>(descB_ + (size_t)pair * kp_stride * 32);
    if (lane == 0) W.redo[pair] = 0;
    if (n1 > maxn || n2 > maxn) { if (lane == 0) W.redo[pair] = 1; return; }       // (k_search_init reports the capacity error)
    const float inv_w = __fdiv_rn((float)SI_COLS, __fsub_rn(max_x, min_x)), inv_h = __fdiv_rn((float)SI_ROWS, __fsub_rn(max_y, min_y));
    float4 *rec = W.rec + (size_t)pair * W.cap0;
    uint4 *desc = W.desc + (size_t)pair * W.cap0 * 2;
    uint16_t *aidx = W.aidx + (size_t)pair * W.cap0;
    int n0 = 0;                                                                    // F2: octave 0 and inside the grid, index order kept
    for (int i0 = 0; i0 < n2; i0 += 256) {                                         // four chunks of loads in flight (the loop is a chain of global round trips otherwise)
        float fxs[4], fys[4]; int ocs[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = min(i0 + 64 * u + lane, n2 - 1);
            fxs[u] = kpB[i].x; fys[u] = kpB[i].y; ocs[u] = kpB[i].octave;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + 64 * u + lane;
            const float fx = fxs[u], fy = fys[u];
            const int px = (int)roundf(__fmul_rn(__fsub_rn(fx, min_x), inv_w)), py = (int)roundf(__fmul_rn(__fsub_rn(fy, min_y), inv_h));      // Frame.cc:718-719
            const bool in = i < n2 && ocs[u] == 0 && px >= 0 && px < SI_COLS && py >= 0 && py < SI_ROWS;
            const unsigned long long bal = __ballot(in);
            const int li = n0 + __popcll(bal & lt_mask);
            if (in && li < W.cap0) {
                cellx[li] = (uint16_t)px; celly[li] = (uint16_t)py; gidx[li] = (uint16_t)i;
                rec[li] = make_float4(fx, fy, 0.0f, __uint_as_float((uint32_t)i));
                desc[2 * li] = dB[2 * i]; desc[2 * li + 1] = dB[2 * i + 1];
            }
            n0 += __popcll(bal);
        }
    }
    int na0 = 0;                                                                   // F1: octave 0 (ORBmatcher.cc:726-728)
    for (int i0 = 0; i0 < n1; i0 += 256) {
        int ocs[4];
#pragma unroll
        for (int u = 0; u < 4; u++) ocs[u] = kpA[min(i0 + 64 * u + lane, n1 - 1)].octave;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + 64 * u + lane;
            const bool in = i < n1 && ocs[u] == 0;
            const unsigned long long bal = __ballot(in);
            const int li = na0 + __popcll(bal & lt_mask);
            if (in && li < W.cap0) aidx[li] = (uint16_t)i;
            na0 += __popcll(bal);
        }
    }
    if (lane == 0) { W.n0[pair] = n0; W.na0[pair] = na0; }
    if (n0 > W.cap0 || na0 > W.cap0) { if (lane == 0) W.redo[pair] = 1; return; }
    __syncthreads();
    for (int li = lane; li < n0; li += 64) {                                       // rank inside the grid cell = position in the cell's vector
        const int cx = cellx[li], cy = celly[li];
        int rank = 0;
        for (int j = 0; j < li; j++) rank += (cellx[j] == cx && celly[j] == cy);
        reinterpret_cast<uint32_t *>(rec + li)[2] = (uint32_t)cx | ((uint32_t)cy << 8) | ((uint32_t)min(rank, SI_RANKS - 1) << 16);
    }
}

__global__ __launch_bounds__(256) void k_si_candidates(const uint8_t *descA_, size_t kp_stride, int max_n, const float *prev_, float min_x, float min_y,
                                                       float max_x, float max_y, int window, SiWork W)
{
    __shared__ uint32_t kbuf_all[4][SIL_BUF];
    __shared__ uint16_t lbuf_all[4][SIL_BUF];
    const int pair = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + w;
    if (W.redo[pair]) return;
    const int n0 = W.n0[pair], na0 = W.na0[pair];
    if (t >= na0) return;
    uint32_t *kbuf = kbuf_all[w]; uint16_t *lbuf = lbuf_all[w];
    const int i1 = W.aidx[(size_t)pair * W.cap0 + t];
    const uint4 *dA = reinterpret_cast<const uint4 *>(descA_ + ((size_t)pair * kp_stride + i1) * 32);
    const uint4 a0 = dA[0], a1 = dA[1];
    const float x = prev_[((size_t)pair * max_n + i1) * 2], y = prev_[((size_t)pair * max_n + i1) * 2 + 1], r = (float)window;
    const float inv_w = __fdiv_rn((float)SI_COLS, __fsub_rn(max_x, min_x)), inv_h = __fdiv_rn((float)SI_ROWS, __fsub_rn(max_y, min_y));
    int32_t *count = W.count + (size_t)pair * W.cap0;
    int c0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, min_x), r), inv_w)); if (c0 < 0) c0 = 0;                   // Frame.cc:656-674
    int c1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, min_x), r), inv_w)); if (c1 > SI_COLS - 1) c1 = SI_COLS - 1;
    int r0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, min_y), r), inv_h)); if (r0 < 0) r0 = 0;
    int r1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, min_y), r), inv_h)); if (r1 > SI_ROWS - 1) r1 = SI_ROWS - 1;
    if (c0 >= SI_COLS || c1 < 0 || r0 >= SI_ROWS || r1 < 0) { if (lane == 0) count[t] = 0; return; }
    const uint32_t cw = (uint32_t)(c1 - c0), rh = (uint32_t)(r1 - r0), ncy = rh + 1;
    const float4 *rec = W.rec + (size_t)pair * W.cap0;
    const uint4 *dC = W.desc + (size_t)pair * W.cap0 * 2;
    int total = 0;
    for (int l0 = 0; l0 < n0; l0 += 64) {
        const int li = l0 + lane;
        bool ok = li < n0;
        uint32_t key = 0xFFFFFFFFu;
        if (ok) {
            const float4 k = rec[li];
            const uint32_t bits = __float_as_uint(k.z);
            const uint32_t dcx = (bits & 255u) - (uint32_t)c0, dcy = ((bits >> 8) & 255u) - (uint32_t)r0;
            ok = dcx <= cw && dcy <= rh && fabsf(__fsub_rn(k.x, x)) < r && fabsf(__fsub_rn(k.y, y)) < r;
            if (ok) key = ((uint32_t)hamming256(a0, a1, dC[2 * li], dC[2 * li + 1]) << 23) | ((dcx * ncy + dcy) * SI_RANKS + (bits >> 16));
        }
        const unsigned long long m = __ballot(ok);
        const int before = __popcll(m & ((1ull << lane) - 1));
        if (ok && total + before < SIL_BUF) { kbuf[total + before] = key; lbuf[total + before] = (uint16_t)li; }
        total += __popcll(m);
    }
    if (total > SIL_BUF) { if (lane == 0) count[t] = 0x7FFFFFFF; return; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const size_t lo = (((size_t)pair * W.chunks + (t >> 6)) * SIL_K) * 64 + (t & 63);
    for (int e = lane; e < total; e += 64) {                                       // rank by counting (keys are unique)
        const uint32_t mine = kbuf[e];
        int rank = 0;
        for (int j = 0; j < total; j++) rank += kbuf[j] < mine;
        if (rank < SIL_K) { W.lkey[lo + (size_t)rank * 64] = mine; W.lli[lo + (size_t)rank * 64] = lbuf[e]; }
    }
    if (lane == 0) count[t] = total;
}

__global__ __launch_bounds__(64) void k_si_replay(const orbhip_keypoint *kpA_, const int32_t *nA, const orbhip_keypoint *kpB_, size_t kp_stride, int max_n,
                                                  float nn_ratio, int check_ori, int maxn, SiWork W, float *prev_, int32_t *m12_, int32_t *nmatches_)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sir_lds[];
    int *md = reinterpret_cast<int *>(sir_lds);                          // [cap0] vMatchedDistance of the F2 subset
    int *m21 = md + W.cap0;                                               // [cap0] vnMatches21
    int *owner = m21 + W.cap0;                                            // [cap0] lowest lane of this round that wants the point (64: none)
    uint32_t *tkey = reinterpret_cast<uint32_t *>(owner + W.cap0);        // [SIL_K][64] this trip's lists
    uint16_t *tli = reinterpret_cast<uint16_t *>(tkey + SIL_K * 64);      // [SIL_K][64]
    uint16_t *bm = tli + SIL_K * 64;                                      // [maxn]
    int8_t *bin_of = reinterpret_cast<int8_t *>(bm + maxn);               // [maxn]
    __shared__ int hist[SI_HISTO];
    __shared__ int s_keep[3];
    const int pair = blockIdx.x, lane = threadIdx.x;
    if (W.redo[pair]) return;
    const int n1 = nA[pair], n0 = W.n0[pair], na0 = W.na0[pair];
    const orbhip_keypoint *kpA = kpA_ + (size_t)pair * kp_stride, *kpB = kpB_ + (size_t)pair * kp_stride;
    float *prev = prev_ + (size_t)pair * max_n * 2;
    int32_t *m12 = m12_ + (size_t)pair * max_n;
    const int32_t *count = W.count + (size_t)pair * W.cap0;
    const uint16_t *aidx = W.aidx + (size_t)pair * W.cap0;
    const float4 *rec = W.rec + (size_t)pair * W.cap0;
    for (int i = lane; i < SI_HISTO; i += 64) hist[i] = 0;
    for (int i = lane; i < n0; i += 64) { md[i] = INT_MAX; m21[i] = -1; owner[i] = 64; }
    for (int i = lane; i < n1; i += 64) { m12[i] = -1; bin_of[i] = -1; bm[i] = 0xFFFFu; }
    __syncthreads();
    int nmatches = 0;
    bool give_up = false;
    for (int t0 = 0; t0 < na0 && !give_up; t0 += 64) {
        const int t = t0 + lane;
        const bool valid = t < na0;
        const int cnt_all = valid ? count[t] : 0;
        if (__ballot(cnt_all == 0x7FFFFFFF)) { give_up = true; break; }
        const int cnt = min(cnt_all, SIL_K);
        const int i1 = valid ? (int)aidx[t] : 0;
        // this trip's lists into LDS, coalesced (entry j of all 64 queries is one 256-byte row)
        const int maxcnt = wave_max_dpp(cnt);
        const size_t lo = (((size_t)pair * W.chunks + (t0 >> 6)) * SIL_K) * 64 + lane;
        for (int j = 0; j < maxcnt; j++) { tkey[j * 64 + lane] = W.lkey[lo + (size_t)j * 64]; tli[j * 64 + lane] = W.lli[lo + (size_t)j * 64]; }
        __syncthreads();
        int ptr = 0;
        bool fin = !valid || cnt == 0;
        while (true) {
            uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
            int l1 = 0, l2 = 0;
            bool accept = false, starved = false;
            if (!fin) {
                while (ptr < cnt) { const uint32_t k = tkey[ptr * 64 + lane]; const int li = tli[ptr * 64 + lane]; if (!(md[li] <= (int)(k >> 23))) { k1 = k; l1 = li; break; } ptr++; }   // ORBmatcher.cc:749
                if (k1 == 0xFFFFFFFFu) starved = cnt_all > SIL_K;                  // ran out of a truncated list
                else {
                    int p2 = ptr + 1;
                    while (p2 < cnt) { const uint32_t k = tkey[p2 * 64 + lane]; const int li = tli[p2 * 64 + lane]; if (!(md[li] <= (int)(k >> 23))) { k2 = k; l2 = li; break; } p2++; }
                    if (k2 == 0xFFFFFFFFu && cnt_all > SIL_K) starved = true;
                    const int best = (int)(k1 >> 23), d2 = k2 == 0xFFFFFFFFu ? INT_MAX : (int)(k2 >> 23);
                    accept = best <= SI_TH_LOW && (float)best < __fmul_rn((float)d2, nn_ratio);          // ORBmatcher.cc:764-766
                }
            }
            if (__ballot(starved)) { give_up = true; break; }
            if (!fin && accept) atomicMin(&owner[l1], lane);
            __syncthreads();
            bool conflict = false;
            if (!fin) {
                if (k1 != 0xFFFFFFFFu) conflict = owner[l1] < lane;
                if (k2 != 0xFFFFFFFFu) conflict = conflict || owner[l2] < lane;
            }
            const unsigned long long cm = __ballot(conflict);
            const int f = cm ? __ffsll((long long)cm) - 1 : 64;                    // lanes below f are final
            __syncthreads();
            if (!fin && accept) owner[l1] = 64;
            if (!fin && lane < f) {
                if (accept) {                                                      // ORBmatcher.cc:766-790
                    const int old = m21[l1];
                    const int best_idx = (int)__float_as_uint(rec[l1].w);
                    if (old >= 0) { m12[old] = -1; nmatches--; }
                    m12[i1] = best_idx; m21[l1] = i1; md[l1] = (int)(k1 >> 23); nmatches++;
                    bm[i1] = (uint16_t)best_idx;
                }
                fin = true;
            }
            __syncthreads();
            if (f == 64) break;
        }
    }
    if (give_up) { if (lane == 0) W.redo[pair] = 1; return; }                      // k_search_init starts over (it resets m12; prev is untouched so far)
    nmatches = wave_sum_dpp(nmatches);
    nmatches = si_tail(lane, n1, kpA, kpB, bm, bin_of, hist, s_keep, m12, prev, nmatches, check_ori);
    if (lane == 0) nmatches_[pair] = nmatches;
}

extern "C" int orbhip_search_for_initialization_device(orbhip_ctx *ctx,
        const orbhip_keypoint *d_kpA, const uint8_t *d_descA, const int32_t *d_nA,
        const orbhip_keypoint *d_kpB, const uint8_t *d_descB, const int32_t *d_nB,
        int pairs, int max_n, size_t frame_stride_kp, float min_x, float min_y, float max_x, float max_y,
        int window_size, float nn_ratio, int check_orientation,
        float *d_prev_matched, int32_t *d_matches12, int32_t *d_nmatches)
{
    if (!ctx || !d_kpA || !d_descA || !d_nA || !d_kpB || !d_descB || !d_nB || pairs <= 0 || max_n <= 0 ||
        !d_prev_matched || !d_matches12 || !d_nmatches || !(max_x > min_x) || !(max_y > min_y))
        return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    int32_t *d_status = orbhip_ctx_status_internal(ctx);    // frames over capacity set ORBHIP_E_CAPACITY
    // LDS is sized from the caller's row capacity: every keypoint of a frame may be octave 0
    const int maxn = max_n < SI_MAXN ? max_n : SI_MAXN, cap0 = max_n < SI_CAP0 ? max_n : SI_CAP0;
    const size_t lds = (size_t)cap0 * (4 * 4 + 7 * 2) + (size_t)maxn * 3;
    // Only octave-0 keypoints take part (~0.22 x nFeatures), but any keypoint MAY be one, so the full carve is 33 bytes per keypoint of
    // capacity: 36 KB per single-wave workgroup at 1100 -- four per CU, and no room beside another kernel's workgroups.  So the pairs are
    // first tried with LDS for ORBHIP_SI_SMALL_CAP0 (default 512) octave-0 points per frame; a pair with more is flagged on the device and
    // done by a second launch with the full carve, which returns at once for the others (0 switches the first launch off).
    const int small_env = getenv("ORBHIP_SI_SMALL_CAP0") ? atoi(getenv("ORBHIP_SI_SMALL_CAP0")) : 512;
    int cap0_small = small_env > 0 && small_env < cap0 ? (small_env + 7) & ~7 : 0;
    const size_t lds_small = (size_t)cap0_small * (4 * 4 + 7 * 2) + (size_t)maxn * 3;
    if (orb_lds_optin(reinterpret_cast<const void *>(k_search_init), orbhip_ctx_device_internal(ctx), lds)) return ORBHIP_E_HIP;
    hipStream_t st = orbhip_ctx_stream_internal(ctx);
    // Up to 512 pairs per call the replay form first (see k_si_replay); pairs it cannot finish are flagged on the device and done by the
    // sequential kernel, which returns at once for the others.  Measured (tools/si_sweep.py, VGA / 1000 features, ~220 octave-0 points per
    // frame; replay vs sequential, ms per call): 1 pair 0.110 / 0.231, 64: 0.136 / 0.240, 256: 0.188 / 0.244, 512: 0.250 / 0.263,
    // 1023: 0.370 / 0.310 -- from ~600 pairs on one register-resident wave per pair on every SIMD is the better use of the chip.
    // ORBHIP_SI_PARALLEL_MAX_PAIRS moves the switch (0: the sequential kernel alone; tests run both forms).
    const int par_max = getenv("ORBHIP_SI_PARALLEL_MAX_PAIRS") ? atoi(getenv("ORBHIP_SI_PARALLEL_MAX_PAIRS")) : 512;
    const int32_t *d_redo = nullptr;
    int32_t *d_redo2 = nullptr;
    {
        SiWork W;
        W.cap0 = ((max_n < SIL_CAP ? max_n : SIL_CAP) + 7) & ~7; W.chunks = (W.cap0 + 63) / 64;
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t P = (size_t)pairs;
        const size_t o_rec = 0, o_desc = o_rec + al(16 * P * W.cap0), o_aidx = o_desc + al(32 * P * W.cap0), o_n0 = o_aidx + al(2 * P * W.cap0),
                     o_na0 = o_n0 + al(4 * P), o_lkey = o_na0 + al(4 * P), o_lli = o_lkey + al(4 * P * W.chunks * SIL_K * 64),
                     o_count = o_lli + al(2 * P * W.chunks * SIL_K * 64), o_redo = o_count + al(4 * P * W.cap0), total = o_redo + al(4 * P);
        const bool replay = pairs <= par_max && total <= ((size_t)1 << 30);
        if (replay) cap0_small = 0;                                // behind the replay form the sequential kernel only sees the few pairs that one handed back: one launch
        uint8_t *wb = nullptr;
        if (replay || cap0_small) {                                // (one request: the small launch's flags sit behind the replay form's work area)
            wb = (uint8_t *)orbhip_ctx_work_internal(ctx, (replay ? total : 0) + (cap0_small ? al(4 * P) : 0));
            if (!wb) return ORBHIP_E_HIP;
            if (cap0_small) d_redo2 = (int32_t *)(wb + (replay ? total : 0));
        }
        if (replay) {
            W.rec = (float4 *)(wb + o_rec); W.desc = (uint4 *)(wb + o_desc); W.aidx = (uint16_t *)(wb + o_aidx); W.n0 = (int32_t *)(wb + o_n0);
            W.na0 = (int32_t *)(wb + o_na0); W.lkey = (uint32_t *)(wb + o_lkey); W.lli = (uint16_t *)(wb + o_lli); W.count = (int32_t *)(wb + o_count);
            W.redo = (int32_t *)(wb + o_redo);
            const size_t prep_lds = (size_t)W.cap0 * 6 + 16, rep_lds = (size_t)W.cap0 * 12 + (size_t)SIL_K * 64 * 6 + (size_t)maxn * 3 + 16;
            if (orb_lds_optin(reinterpret_cast<const void *>(k_si_prep), orbhip_ctx_device_internal(ctx), prep_lds) ||
                orb_lds_optin(reinterpret_cast<const void *>(k_si_replay), orbhip_ctx_device_internal(ctx), rep_lds)) return ORBHIP_E_HIP;
            hipLaunchKernelGGL(k_si_prep, dim3(pairs), dim3(64), prep_lds, st, d_kpA, d_nA, d_kpB, d_descB, d_nB, frame_stride_kp, min_x, min_y, max_x, max_y, maxn, W);
            hipLaunchKernelGGL(k_si_candidates, dim3((W.cap0 + 3) / 4, pairs), dim3(256), 0, st, d_descA, frame_stride_kp, max_n, d_prev_matched, min_x, min_y,
                               max_x, max_y, window_size, W);
            hipLaunchKernelGGL(k_si_replay, dim3(pairs), dim3(64), rep_lds, st, d_kpA, d_nA, d_kpB, frame_stride_kp, max_n, nn_ratio, check_orientation, maxn, W,
                               d_prev_matched, d_matches12, d_nmatches);
            d_redo = W.redo;
            if (getenv("ORBHIP_SI_DEBUG")) {                       // development: how many pairs the replay form handed back
                std::vector<int32_t> h(pairs);
                if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.data(), W.redo, 4 * (size_t)pairs, hipMemcpyDeviceToHost) == hipSuccess) {
                    int nredo = 0;
                    for (int v : h) nredo += v != 0;
                    fprintf(stderr, "[orbhip] SearchForInitialization: %d of %d pairs fall back to the sequential kernel\n", nredo, pairs);
                }
            }
        }
    }
    if (cap0_small) {
        hipLaunchKernelGGL(k_search_init, dim3(pairs), dim3(64), lds_small, st, d_kpA, d_descA, d_nA,
                           d_kpB, d_descB, d_nB, max_n, frame_stride_kp, min_x, min_y, max_x, max_y, window_size, nn_ratio,
                           check_orientation, cap0_small, maxn, d_prev_matched, d_matches12, d_nmatches, d_status, d_redo, d_redo2);
        d_redo = d_redo2;
    }
    hipLaunchKernelGGL(k_search_init, dim3(pairs), dim3(64), lds, st, d_kpA, d_descA, d_nA,
                       d_kpB, d_descB, d_nB, max_n, frame_stride_kp, min_x, min_y, max_x, max_y, window_size, nn_ratio,
                       check_orientation, cap0, maxn, d_prev_matched, d_matches12, d_nmatches, d_status, d_redo, (int32_t *)nullptr);
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}

// vbPrevMatched initialisation (Tracking.cc:1497-1499: mvbPrevMatched[i] = mvKeysUn[i].pt), batched.
__global__ void k_prev_matched_init(const orbhip_keypoint *kp, size_t kp_stride, int frames, int max_n, float *xy)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    if (i >= max_n || f >= frames) return;
    const orbhip_keypoint k = kp[(size_t)f * kp_stride + i];
    xy[((size_t)f * max_n + i) * 2] = k.x;
    xy[((size_t)f * max_n + i) * 2 + 1] = k.y;
}

extern "C" int orbhip_prev_matched_init_device(orbhip_ctx *ctx, const orbhip_keypoint *d_kp, size_t frame_stride_kp,
                                               int frames, int max_n, float *d_prev_matched)
{
    if (!ctx || !d_kp || !d_prev_matched || frames <= 0 || max_n <= 0) return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    hipLaunchKernelGGL(k_prev_matched_init, dim3((max_n + 255) / 256, frames), dim3(256), 0, orbhip_ctx_stream_internal(ctx),
                       d_kp, frame_stride_kp, frames, max_n, d_prev_matched);
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}

// ---------------------------------------------------------------------------- M3 + M4 (tracking)
// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), ORBmatcher.cc:1965-2181, Nleft == -1.
// One wave per frame pair.  The Frame grid (Frame.cc:377-408) is built in LDS as a CSR: cells are numbered
// ix*48+iy, so the cells (ix, r0..r1) that GetFeaturesInArea visits for one column are one contiguous run of
// the item list, and concatenating the runs of columns c0..c1 IS the function's visiting order (Frame.cc:
// 676-711).  Per query the lanes own one grid column each, a wave scan places the runs, and the candidates
// are then evaluated one per lane with key = distance << 12 | position ("first candidate wins", :2051-2055).
// The query loop is sequential: a keypoint claimed by a map point with observations drops out of later
// queries (:2037-2039).
#define SBP_CAP 2048          // keypoints / queries per frame of the replay form (its keys carry 11-bit positions and indices)
#define SBP_SEQ_CAP 8192      // ... of the sequential kernel, LDS permitting (17 B per keypoint + 3 B per query + the grid: see sbp_launch)
#define SBP_CAND_CAP 4096     // candidates of one query (12-bit position in its key)
struct OrbLevelSigma { float inv_sigma2[16]; };       // mvInvLevelSigma2, passed by value
#define SBP_CELLS (SI_COLS * SI_ROWS)
// Frame::AssignFeaturesToGrid (Frame.cc:377-408) as a CSR in LDS, built by one wave: cell by round() (PosInGrid, :716-726),
// insertion order = index order.  rank = number of earlier keypoints in the same cell = the cell's counter before this trip +
// the earlier lanes of the trip with the same cell.  cell_start must be zeroed by the caller; on return cell_start[c] is the
// first slot of cell c = ix*48+iy in items[], and kx / ky / oct hold the keypoints' coordinates and octaves.
// Rig frames (Nleft != -1, Frame.cc:395-405): keypoints nleft .. n-1 are the right camera's and fill mGridRight, here the cells
// SBP_CELLS .. 2*SBP_CELLS-1 of the same CSR (ncells = 2*SBP_CELLS); items hold frame-wide indices (i = right index + Nleft).
__device__ void sbp_build_grid(uint32_t *cell_start, float *kx, float *ky, uint8_t *oct, uint16_t *items, uint16_t *cell_of,
                               uint16_t *rank_of, const orbhip_keypoint *kp, int n, float min_x, float min_y, float inv_w, float inv_h, int lane,
                               int ncells = SI_COLS * SI_ROWS, int nleft = -1)
{
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        int c = 0xFFFF;
        if (i < n) {
            const orbhip_keypoint k = kp[i];
            kx[i] = k.x; ky[i] = k.y; oct[i] = (uint8_t)k.octave;
            const int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, min_x), inv_w));
            const int py = (int)roundf(__fmul_rn(__fsub_rn(k.y, min_y), inv_h));
            if (px >= 0 && px < SI_COLS && py >= 0 && py < SI_ROWS) c = px * SI_ROWS + py + ((nleft >= 0 && i >= nleft) ? SI_COLS * SI_ROWS : 0);
        }
        int intra = 0;
        for (int l = 0; l < 64; l++) {
            const int cl = __builtin_amdgcn_readlane(c, l);
            intra += (cl == c && l < lane);
        }
        if (i < n) {
            cell_of[i] = (uint16_t)c;
            if (c != 0xFFFF) rank_of[i] = (uint16_t)(cell_start[c + 1] + intra);
        }
        __syncthreads();                                          // every lane has read its counter
        if (i < n && c != 0xFFFF) atomicAdd(&cell_start[c + 1], 1u);
        __syncthreads();
    }
    {   // exclusive prefix over the cell counts (cell_start[c+1] holds count(c))
        uint32_t carry = 0;
        for (int c0 = 1; c0 <= ncells; c0 += 64) {
            const int c = c0 + lane;
            const int v = c <= ncells ? (int)cell_start[c] : 0;
            const int inc = wave_incl_scan_i(v);
            if (c <= ncells) cell_start[c] = carry + (uint32_t)inc;
            carry += (uint32_t)__builtin_amdgcn_readlane(inc, 63);
        }
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) { const int c = cell_of[i]; if (c != 0xFFFF) items[cell_start[c] + rank_of[i]] = (uint16_t)i; }
    __syncthreads();
}

// Register-resident sequential loop of SearchByProjection (both modes, frames of one camera, at most 64 * NS keypoints inside the
// grid).  The CSR grid orders the train keypoints by (grid column, row, rank) = GetFeaturesInArea's visit order; slot s of lane l
// holds the keypoint at CSR position 64 s + l -- position, cell, octave, holder state, uRight and descriptor in VGPRs.  A slot
// covers a contiguous range of grid columns, so a query only evaluates the 2..4 slots its window's columns intersect (uniform
// test), branch-free on all 64 lanes: cell-range + level + distance + holder + uRight gates, Hamming distance, key =
// distance << 12 | CSR position (the old candidate list's order).  No candidate list, no LDS or global access and no barrier
// in the chain (cycle counters, -DSBP_PROF: the list was 49 % of a query's 5200 cycles, the evaluation with its descriptor
// fetches 30 %).  Returns nmatches; writes qm[] and the final holder[] to LDS for the rotation check that follows.
template <int NS, bool UR>
__device__ __forceinline__ int sbp_register_loop(int lane, int n_items, int nq, const orbhip_proj_query *Q, const uint4 *dQ, const uint4 *dT,
                                                 const float *uright, const float *kx, const float *ky, const uint8_t *oct, const uint16_t *items,
                                                 const uint16_t *cell_of, int16_t *holder, int16_t *qm,
                                                 float min_x, float min_y, float inv_w, float inv_h, int th_high, int mode, float nn_ratio)
{
    float fkx[NS], fky[NS], fur[UR ? NS : 1]; uint32_t fpk[NS]; int fh[NS], fi[NS]; uint4 fd0[NS], fd1[NS];
    int slo[NS], shi[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int p = 64 * s + lane;
        const bool v = p < n_items;
        const int i2 = v ? (int)items[p] : 0;                                                          // (row 0 of the pair's arrays exists also for an empty frame)
        fi[s] = i2; fkx[s] = kx[i2]; fky[s] = ky[i2];
        const int cell = cell_of[i2], cx = (int)(((uint32_t)cell * 43691u) >> 21), cy = cell - SI_ROWS * cx;     // cell / 48, exact for cell < 3072
        fpk[s] = v ? (uint32_t)cx | ((uint32_t)cy << 7) | ((uint32_t)oct[i2] << 13) : 127u;          // column 127: in no window
        fh[s] = holder[i2]; if (UR) fur[s] = uright[i2];
        fd0[s] = dT[2 * i2]; fd1[s] = dT[2 * i2 + 1];
        const int lastl = min(63, n_items - 1 - 64 * s);                                              // uniform; < 0: empty slot
        slo[s] = lastl >= 0 ? __builtin_amdgcn_readlane(cx, 0) : 127;
        shi[s] = lastl >= 0 ? __builtin_amdgcn_readlane(cx, lastl) : -1;
    }
    int nmatches = 0;
    // the query records and descriptors are fetched two queries ahead
    orbhip_proj_query q1 = Q[0], q2 = Q[min(1, nq - 1)];
    uint4 a1_0 = dQ[0], a1_1 = dQ[1], a2_0 = dQ[2 * min(1, nq - 1)], a2_1 = dQ[2 * min(1, nq - 1) + 1];
    for (int t = 0; t < nq; t++) {
        const orbhip_proj_query qq = q1;
        const uint4 a0 = a1_0, a1 = a1_1;
        q1 = q2; a1_0 = a2_0; a1_1 = a2_1;
        { const int tn = min(t + 2, nq - 1); q2 = Q[tn]; a2_0 = dQ[2 * tn]; a2_1 = dQ[2 * tn + 1]; }
        const float x = qq.u, y = qq.v, r = qq.radius;
        int c0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, min_x), r), inv_w)); if (c0 < 0) c0 = 0;   // Frame.cc:656-674
        if (c0 >= SI_COLS) continue;
        int c1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, min_x), r), inv_w)); if (c1 > SI_COLS - 1) c1 = SI_COLS - 1;
        if (c1 < 0) continue;
        int r0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, min_y), r), inv_h)); if (r0 < 0) r0 = 0;
        if (r0 >= SI_ROWS) continue;
        int r1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, min_y), r), inv_h)); if (r1 > SI_ROWS - 1) r1 = SI_ROWS - 1;
        if (r1 < 0) continue;
        c0 = __builtin_amdgcn_readfirstlane(c0); c1 = __builtin_amdgcn_readfirstlane(c1);
        const bool check_lv = qq.min_level > 0 || qq.max_level >= 0;            // Frame.cc:676
        const uint32_t cw = (uint32_t)(c1 - c0), rh = (uint32_t)(r1 - r0);
        uint32_t lk1 = 0xFFFFFFFFu, lk2 = 0xFFFFFFFFu, pay1 = 0, pay2 = 0; int sl1 = 0;    // per lane: two smallest keys, their (index | octave << 16), the best's slot
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (shi[s] < c0 || slo[s] > c1) continue;                           // uniform: no keypoint of this slot in the window's columns
            const uint32_t dcx = (fpk[s] & 127u) - (uint32_t)c0, dcy = ((fpk[s] >> 7) & 63u) - (uint32_t)r0;
            const int o = (int)(fpk[s] >> 13), h = fh[s];
            // (bitwise &: no short-circuit branches -- every gate is a compare, the combination scalar logic)
            bool ok = (dcx <= cw) & (dcy <= rh);
            ok = ok & !((int)check_lv & ((int)(o < qq.min_level) | ((int)(qq.max_level >= 0) & (int)(o > qq.max_level))));   // Frame.cc:693-701
            ok = ok & (bool)((int)(fabsf(__fsub_rn(fkx[s], x)) < r) & (int)(fabsf(__fsub_rn(fky[s], y)) < r));   // Frame.cc:704-708
            ok = ok & !((h <= -2) | ((h >= 0) & ((h & 1) != 0)));                                       // ORBmatcher.cc:2037-2039 / 96-98
            if (UR) ok = ok & !((fur[s] > 0) & (fabsf(__fsub_rn(qq.ur, fur[s])) > r));                  // ORBmatcher.cc:2041-2047 / 100-105
            const int dist = hamming256(a0, a1, fd0[s], fd1[s]);
            const uint32_t key = ok ? ((uint32_t)dist << 12) | (uint32_t)(64 * s + lane) : 0xFFFFFFFFu;
            const uint32_t pay = (uint32_t)fi[s] | ((uint32_t)o << 16);
            const bool b1 = key < lk1, b2 = key < lk2;
            pay2 = b1 ? pay1 : (b2 ? pay : pay2); lk2 = b1 ? lk1 : (b2 ? key : lk2);
            pay1 = b1 ? pay : pay1; sl1 = b1 ? s : sl1; lk1 = b1 ? key : lk1;
        }
        const uint32_t key = wave_min_u32_dpp(lk1);
        if (key == 0xFFFFFFFFu) continue;
        bool accept = (int)(key >> 12) <= th_high && (int)(key >> 12) < 256;                         // ORBmatcher.cc:2030,2058 / 85,131
        const unsigned long long own1 = __ballot(lk1 == key);
        const int ol = __builtin_amdgcn_readfirstlane(__ffsll((long long)own1) - 1);
        const int bsl = __builtin_amdgcn_readlane(sl1, ol);
        if (accept && mode == 1) {
            // local-map variant (ORBmatcher.cc:131-137): ratio test against the second best when it is of the same octave; the second
            // smallest key is the minimum over the winner's runner-up and every other lane's best (keys are unique)
            const uint32_t c2 = lk1 == key ? lk2 : lk1, cp2 = lk1 == key ? pay2 : pay1;
            const uint32_t key2 = wave_min_u32_dpp(c2);
            const int best_lv = (int)((uint32_t)__builtin_amdgcn_readlane((int)pay1, ol) >> 16);
            int lv2 = -1, d2 = 256;
            if (key2 != 0xFFFFFFFFu && (int)(key2 >> 12) < 256) {
                d2 = (int)(key2 >> 12);
                const unsigned long long own2 = __ballot(c2 == key2);
                const int ol2 = __builtin_amdgcn_readfirstlane(__ffsll((long long)own2) - 1);
                lv2 = (int)((uint32_t)__builtin_amdgcn_readlane((int)cp2, ol2) >> 16);
            }
            if (best_lv == lv2 && (float)(int)(key >> 12) > __fmul_rn(nn_ratio, (float)d2)) accept = false;
        }
        if (accept) {
            const int best = (int)((uint32_t)__builtin_amdgcn_readlane((int)pay1, ol) & 0xFFFFu);
#pragma unroll
            for (int s = 0; s < NS; s++)
                if (bsl == s) { if (lane == ol) fh[s] = (t << 1) | (qq.has_obs & 1); }          // uniform slot test
            if (lane == 0) qm[t] = (int16_t)best;
            nmatches++;
        }
    }
    // the holder states go back to LDS for the rotation check and the final write-out
#pragma unroll
    for (int s = 0; s < NS; s++) if (64 * s + lane < n_items) holder[fi[s]] = (int16_t)fh[s];
    return nmatches;
}


#ifdef SBP_PROF
__device__ long long g_sbp_prof[8];           // debug build only (EXTRA=-DSBP_PROF): cycles of setup / candidate list / evaluation / update / tail of pair 0; queries, candidates
#define SBP_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = clock64(); g_sbp_prof[i] += t_ - t_prev; t_prev = t_; } } while (0)
extern "C" int orbhip_debug_sbp_prof(long long *out8, int reset)
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_sbp_prof), 64) != hipSuccess) return ORBHIP_E_HIP;
    if (reset) { long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sbp_prof), z, 64) != hipSuccess) return ORBHIP_E_HIP; }
    return ORBHIP_OK;
}
#else
#define SBP_T(i) do { } while (0)
#endif
template <bool DESC_LDS>
__global__ __launch_bounds__(64) void k_search_by_projection(const orbhip_proj_query *q_, const uint8_t *descq_, const int32_t *nq_, int max_q,
                                                             const orbhip_keypoint *kp_, const uint8_t *desc_, const float *uright_,
                                                             const int32_t *n_, int max_n, size_t kp_stride,
                                                             float min_x, float min_y, float max_x, float max_y,
                                                             int th_high, int check_ori, int mode, float nn_ratio, int cap_n, int cap_q,
                                                             int32_t *tm_, int32_t *nmatches_, int32_t *status,
                                                             const int32_t *nleft_, const int32_t *mirror_, int ncells, const int32_t *redo_)
{
    // dynamic LDS carved by the launcher's capacities (cap_n keypoints, cap_q queries per pair): small frames keep
    // four pairs per CU resident
    extern __shared__ __attribute__((aligned(16))) uint8_t sbp_lds[];
    uint32_t *cell_start = reinterpret_cast<uint32_t *>(sbp_lds);                   // [ncells + 1]: SBP_CELLS, twice that for rig frames
    float *kx = reinterpret_cast<float *>(cell_start + ncells + 1), *ky = kx + cap_n;
    int16_t *holder = reinterpret_cast<int16_t *>(ky + cap_n);                       // -1 free, -2 pre-held, else (query << 1 | has_obs)
    // (the candidate list of ONE query never needs more than 4096 entries -- its position travels in 12 bits of the key -- so keyframes of
    // up to ~7 600 keypoints fit: the 5 x nFeatures keypoints of a monocular map's first two keyframes, Tracking.cc:210)
    const int cap_c = cap_n < SBP_CAND_CAP ? cap_n : SBP_CAND_CAP;
    uint16_t *items = reinterpret_cast<uint16_t *>(holder + cap_n), *cand = items + cap_n, *cell_of = cand + cap_c, *rank_of = cell_of + cap_n;
    int16_t *qm = reinterpret_cast<int16_t *>(rank_of + cap_n);                      // query -> claimed keypoint
    int8_t *qbin = reinterpret_cast<int8_t *>(qm + cap_q);
    uint8_t *oct = reinterpret_cast<uint8_t *>(qbin + cap_q);
    // optional: the train descriptors too (32 B each) -- removes the one global round trip left in every query; used when
    // the launch is small enough that fewer resident pairs per CU do not matter
    uint4 *dlds = reinterpret_cast<uint4 *>(sbp_lds + ((sizeof(uint32_t) * ((size_t)ncells + 1) + (size_t)cap_n * 17 + (size_t)cap_c * 2 + (size_t)cap_q * 3 + 15) & ~(size_t)15));
    __shared__ int hist[SI_HISTO];
    __shared__ int s_keep[3];
    const int pair = blockIdx.x, lane = threadIdx.x;
    if (redo_ && !redo_[pair]) return;                       // the low-latency form (k_sbp_replay) has done this pair
#ifdef SBP_PROF
    long long t_prev = clock64();
#endif
    const unsigned long long lt_mask = (1ull << lane) - 1;
    const int n = n_[pair], nq = nq_[pair];
    // rig frames (Nleft != -1): keypoints [0, nleft) are the left camera's, [nleft, n) the right camera's; a query carries the camera
    // it searches in bit 1 of has_obs; mirror[i] = the same point's keypoint in the other camera (mvLeftToRightMatch / mvRightToLeftMatch
    // as frame-wide indices) or -1
    const int nleft = nleft_ ? nleft_[pair] : -1;
    const int32_t *mirror = mirror_ ? mirror_ + (size_t)pair * max_n : nullptr;
    const orbhip_proj_query *Q = q_ + (size_t)pair * max_q;
    const uint4 *dQ = reinterpret_cast<const uint4 *>(descq_ + (size_t)pair * max_q * 32);
    const orbhip_keypoint *kp = kp_ + (size_t)pair * kp_stride;
    const uint4 *dT = reinterpret_cast<const uint4 *>(desc_ + (size_t)pair * kp_stride * 32);
    const float *uright = uright_ ? uright_ + (size_t)pair * kp_stride : nullptr;
    int32_t *tm = tm_ + (size_t)pair * max_n;
    if (n > cap_n || nq > cap_q || n > max_n || nq > max_q) {
        if (lane == 0) { atomicExch(status, ORBHIP_E_CAPACITY); nmatches_[pair] = 0; }
        return;
    }
    const float inv_w = __fdiv_rn((float)SI_COLS, __fsub_rn(max_x, min_x));       // Frame.cc:334-335
    const float inv_h = __fdiv_rn((float)SI_ROWS, __fsub_rn(max_y, min_y));
    for (int i = lane; i < SI_HISTO; i += 64) hist[i] = 0;
    for (int c = lane; c <= ncells; c += 64) cell_start[c] = 0;
    for (int t = lane; t < nq; t += 64) { qm[t] = -1; qbin[t] = -1; }
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        holder[i] = tm[i] == -1 ? (int16_t)-1 : (int16_t)-2;
        if (DESC_LDS) { dlds[2 * i] = dT[2 * i]; dlds[2 * i + 1] = dT[2 * i + 1]; }
    }
    sbp_build_grid(cell_start, kx, ky, oct, items, cell_of, rank_of, kp, n, min_x, min_y, inv_w, inv_h, lane, ncells, nleft);
    SBP_T(0);
    // ---- sequential query loop (ORBmatcher.cc:1987-2088)
    int nmatches = 0;
    const float factor = 1.0f / SI_HISTO;
    const int n_items = (int)cell_start[ncells];
    if (nleft < 0 && !mirror && n_items <= 16 * 64 && nq > 0) {
        nmatches = uright ? sbp_register_loop<16, true>(lane, n_items, nq, Q, dQ, dT, uright, kx, ky, oct, items, cell_of, holder, qm, min_x, min_y, inv_w, inv_h,
                                                        th_high, mode, nn_ratio)
                          : sbp_register_loop<16, false>(lane, n_items, nq, Q, dQ, dT, uright, kx, ky, oct, items, cell_of, holder, qm, min_x, min_y, inv_w, inv_h,
                                                         th_high, mode, nn_ratio);
    } else {
    // the next query's record and descriptor are fetched one iteration ahead (their latency overlaps this query's work)
    orbhip_proj_query qn = Q[0];
    uint4 n0 = dQ[0], n1 = dQ[1];
    for (int t = 0; t < nq; t++) {
        const orbhip_proj_query qq = qn;
        const uint4 a0 = n0, a1 = n1;
        if (t + 1 < nq) { qn = Q[t + 1]; n0 = dQ[2 * t + 2]; n1 = dQ[2 * t + 3]; }
        const float x = qq.u, y = qq.v, r = qq.radius;
        int c0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, min_x), r), inv_w)); if (c0 < 0) c0 = 0;   // Frame.cc:656-674
        if (c0 >= SI_COLS) continue;
        int c1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, min_x), r), inv_w)); if (c1 > SI_COLS - 1) c1 = SI_COLS - 1;
        if (c1 < 0) continue;
        int r0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, min_y), r), inv_h)); if (r0 < 0) r0 = 0;
        if (r0 >= SI_ROWS) continue;
        int r1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, min_y), r), inv_h)); if (r1 > SI_ROWS - 1) r1 = SI_ROWS - 1;
        if (r1 < 0) continue;
        // lane = grid column c0+lane: its cells r0..r1 are one run of `items` (rig: the queried camera's half of the cells)
        const int cbase = (nleft >= 0 && (qq.has_obs & 2)) ? SBP_CELLS : 0;
        int start = 0, len = 0;
        if (c0 + lane <= c1) {
            start = (int)cell_start[cbase + (c0 + lane) * SI_ROWS + r0];
            len = (int)cell_start[cbase + (c0 + lane) * SI_ROWS + r1 + 1] - start;
        }
        const int inc = wave_scan_add_dpp(len);                      // DPP scans / reductions: no LDS round trips in the per-query chain
        const int off = inc - len, total = __builtin_amdgcn_readlane(inc, 63);
        if (total == 0) continue;
        if (total > cap_c) { if (lane == 0) atomicExch(status, ORBHIP_E_CAPACITY); continue; }      // > 4096 keypoints in one search window
        const int maxlen = wave_max_dpp(len);
        for (int j = 0; j < maxlen; j++) if (j < len) cand[off + j] = items[start + j];
        __syncthreads();
        SBP_T(1);
        const bool check_lv = qq.min_level > 0 || qq.max_level >= 0;            // Frame.cc:676
        uint32_t key = 0xFFFFFFFFu, key2 = 0xFFFFFFFFu;                         // the two smallest (distance << 12 | position)
        for (int k0 = 0; k0 < total; k0 += 64) {
            const int k = k0 + lane;
            if (k < total) {
                const int i2 = cand[k];
                // the descriptor is fetched together with the keypoint's other data (one latency instead of two on the chain)
                const uint4 t0 = DESC_LDS ? dlds[2 * i2] : dT[2 * i2], t1 = DESC_LDS ? dlds[2 * i2 + 1] : dT[2 * i2 + 1];
                const int o = oct[i2], h = holder[i2];
                bool ok = !(check_lv && (o < qq.min_level || (qq.max_level >= 0 && o > qq.max_level)));   // Frame.cc:693-701
                ok = ok && fabsf(__fsub_rn(kx[i2], x)) < r && fabsf(__fsub_rn(ky[i2], y)) < r;           // Frame.cc:704-708
                ok = ok && !(h <= -2 || (h >= 0 && (h & 1)));                                              // ORBmatcher.cc:2037-2039 / 96-98
                if (ok && uright) {
                    const float ur2 = uright[i2];
                    if (ur2 > 0 && fabsf(__fsub_rn(qq.ur, ur2)) > r) ok = false;                           // ORBmatcher.cc:2041-2047 / 100-105
                }
                if (ok) {
                    const int dist = hamming256(a0, a1, t0, t1);
                    const uint32_t kk = ((uint32_t)dist << 12) | (uint32_t)k;
                    key2 = min(key2, max(key, kk));
                    key = min(key, kk);
                }
            }
        }
        wave_min2_u32_dpp(key, key2);
        SBP_T(2);
#ifdef SBP_PROF
        if (blockIdx.x == 0 && threadIdx.x == 0) { g_sbp_prof[5] += 1; g_sbp_prof[6] += total; }
#endif
        bool accept = key != 0xFFFFFFFFu && (int)(key >> 12) <= th_high && (int)(key >> 12) < 256;       // ORBmatcher.cc:2030,2058 / 85,131
        if (accept && mode == 1) {
            // local-map variant (ORBmatcher.cc:131-137): ratio test against the second best of the same octave.
            // "second best" of the reference's scan == second smallest key (strict <, first candidate wins ties)
            const int best_lv = oct[cand[key & 0xFFFu]];
            int d2 = 256, lv2 = -1;
            if (key2 != 0xFFFFFFFFu && (int)(key2 >> 12) < 256) { d2 = (int)(key2 >> 12); lv2 = oct[cand[key2 & 0xFFFu]]; }
            if (best_lv == lv2 && (float)(int)(key >> 12) > __fmul_rn(nn_ratio, (float)d2)) accept = false;
        }
        if (accept) {
            if (lane == 0) {
                const int best = cand[key & 0xFFFu];
                holder[best] = (int16_t)((t << 1) | (qq.has_obs & 1));
                qm[t] = (int16_t)best;
            }
            nmatches++;
            if (mirror) {                                                       // also the stereo observation in the other camera (:142-146, :203-207)
                const int m = mirror[cand[key & 0xFFFu]];
                if (m >= 0) { if (lane == 0) holder[m] = (int16_t)((t << 1) | (qq.has_obs & 1)); nmatches++; }
            }
        }
        __syncthreads();                                                        // cand / holder reused by the next query
        SBP_T(3);
    }
    }
    __syncthreads();
    // ---- rotation consistency (ORBmatcher.cc:2156-2178): every histogram entry of a dropped bin clears its
    // keypoint (also when a later query re-claimed it) and counts once.  Bins (ORBmatcher.cc:2064-2084) are computed
    // here in parallel for all matches instead of inside the sequential loop (one dependent global read less per query).
    if (check_ori) {
        for (int t = lane; t < nq; t += 64) {
            const int best = qm[t];
            if (best < 0) continue;
            float rot = __fsub_rn(Q[t].angle, kp[best].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == SI_HISTO) bin = 0;
            atomicAdd(&hist[bin], 1); qbin[t] = (int8_t)bin;
        }
        __syncthreads();
        if (lane == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < SI_HISTO; i++) {
                const int sz = hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
                else if (sz > max3) { max3 = sz; ind3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
            s_keep[0] = ind1; s_keep[1] = ind2; s_keep[2] = ind3;
        }
        __syncthreads();
        int removed = 0;
        for (int t = lane; t < nq; t += 64) {
            const int b = qbin[t];
            if (b < 0 || b == s_keep[0] || b == s_keep[1] || b == s_keep[2]) continue;
            holder[qm[t]] = -1; removed++;
        }
        removed = wave_sum_dpp(removed);
        nmatches -= removed;
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) { const int h = holder[i]; tm[i] = h >= 0 ? (h >> 1) : h; }
    if (lane == 0) nmatches_[pair] = nmatches;
    SBP_T(4);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Low-latency form of SearchByProjection (both modes, single-camera frames) for calls with FEW frame pairs -- Tracking calls the
// matcher with one.  The register / LDS loops above walk the queries one after the other on ONE wave per pair (~1.2 us per query: 1.2 ms
// for a frame, whatever the batch); here the expensive part of a query -- window, gates, Hamming distances -- does not depend on the
// claim rule and is evaluated for ALL queries at once, over the whole chip:
//   k_sbp_prep        one wave per pair: the Frame grid as a CSR (sbp_build_grid), written out in CSR order: per position a record
//                     {x, y, cell | octave | pre-held, uRight}, the keypoint index, the descriptor, and the first position of every grid column;
//   k_sbp_candidates  one wave per query: the window's columns are ONE contiguous range of CSR positions; every position in it is gated
//                     (cell range, level, |dx|,|dy| < r, pre-held, uRight) and its Hamming distance taken; the keys
//                     distance << 22 | position << 11 | keypoint index are sorted (rank by counting, in LDS) and the SBPL_K smallest stored;
//   k_sbp_replay      one wave per pair replays the claim rule (ORBmatcher.cc:2037-2039 / :96-98) over the sorted lists, 64 queries per
//                     trip, one per lane: a lane's best (and, local-map mode, second best) is its first (two) list entries whose keypoint no
//                     EARLIER query with observations holds -- claims only ever block more keypoints, so a lane's cursor only moves forward.
//                     Inside a trip the lanes are speculative: every lane publishes its claim (LDS atomicMin of the lane id per keypoint), the
//                     prefix of lanes up to the first one whose best / second best was claimed by an earlier lane is final, the others look
//                     again.  A trip needs 1 + (number of conflicts) rounds of a few LDS operations.
// The result is the sequential loop's, bit for bit (same candidate sets, same key order, same claim rule); a query whose list was cut at
// SBPL_K entries and runs out of them makes its pair fall back to the sequential kernel (flag per pair, read on the device).
#define SBPL_K 32                  // list entries kept per query
#define SBPL_BUF 512               // candidates a query may have before the pair falls back
struct SbpWork {
    float4 *rec;                   // [pairs][cap_n]   CSR order: x, y, bits(cx | cy << 7 | octave << 13 | pre-held << 17), uRight
    uint4 *desc;                   // [pairs][cap_n][2]
    uint16_t *idx;                 // [pairs][cap_n]   keypoint index at a CSR position
    int32_t *col_start;            // [pairs][SI_COLS + 1]
    uint32_t *lists;               // [pairs][chunks][SBPL_K][64]   (transposed: lane = query inside a trip of 64)
    int32_t *count;                // [pairs][cap_q]   candidates of the query (may exceed SBPL_K; INT_MAX: more than SBPL_BUF)
    int32_t *redo;                 // [pairs]          1 = the sequential kernel must process this pair
    int cap_n, cap_q, chunks;
};

__global__ __launch_bounds__(64) void k_sbp_prep(const orbhip_keypoint *kp_, const uint8_t *desc_, const float *uright_, const int32_t *n_, int max_n,
                                                 size_t kp_stride, float min_x, float min_y, float max_x, float max_y, const int32_t *tm_, SbpWork W,
                                                 int32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sbp_lds[];
    uint32_t *cell_start = reinterpret_cast<uint32_t *>(sbp_lds);
    float *kx = reinterpret_cast<float *>(cell_start + SBP_CELLS + 1), *ky = kx + W.cap_n;
    uint16_t *items = reinterpret_cast<uint16_t *>(ky + W.cap_n), *cell_of = items + W.cap_n, *rank_of = cell_of + W.cap_n;
    uint8_t *oct = reinterpret_cast<uint8_t *>(rank_of + W.cap_n);
    const int pair = blockIdx.x, lane = threadIdx.x;
    const int n = n_[pair];
    int32_t *cs = W.col_start + (size_t)pair * (SI_COLS + 1);
    if (n > W.cap_n || n > max_n) {
        if (lane == 0) { atomicExch(status, ORBHIP_E_CAPACITY); W.redo[pair] = 1; }
        for (int c = lane; c <= SI_COLS; c += 64) cs[c] = 0;
        return;
    }
    const orbhip_keypoint *kp = kp_ + (size_t)pair * kp_stride;
    const uint4 *dT = reinterpret_cast<const uint4 *>(desc_ + (size_t)pair * kp_stride * 32);
    const float *uright = uright_ ? uright_ + (size_t)pair * kp_stride : nullptr;
    const int32_t *tm = tm_ + (size_t)pair * max_n;
    const float inv_w = __fdiv_rn((float)SI_COLS, __fsub_rn(max_x, min_x)), inv_h = __fdiv_rn((float)SI_ROWS, __fsub_rn(max_y, min_y));
    for (int c = lane; c <= SBP_CELLS; c += 64) cell_start[c] = 0;
    __syncthreads();
    sbp_build_grid(cell_start, kx, ky, oct, items, cell_of, rank_of, kp, n, min_x, min_y, inv_w, inv_h, lane);
    const int n_items = (int)cell_start[SBP_CELLS];
    float4 *rec = W.rec + (size_t)pair * W.cap_n;
    uint4 *dC = W.desc + (size_t)pair * W.cap_n * 2;
    uint16_t *idx = W.idx + (size_t)pair * W.cap_n;
    for (int p = lane; p < n_items; p += 64) {
        const int i = items[p];
        const int cell = cell_of[i], cx = (int)(((uint32_t)cell * 43691u) >> 21), cy = cell - SI_ROWS * cx;
        const uint32_t bits = (uint32_t)cx | ((uint32_t)cy << 7) | ((uint32_t)oct[i] << 13) | ((tm[i] != -1) ? (1u << 17) : 0u);
        rec[p] = make_float4(kx[i], ky[i], __uint_as_float(bits), uright ? uright[i] : -1.0f);
        idx[p] = (uint16_t)i;
        dC[2 * p] = dT[2 * i]; dC[2 * p + 1] = dT[2 * i + 1];
    }
    for (int c = lane; c <= SI_COLS; c += 64) cs[c] = (int32_t)cell_start[c * SI_ROWS];
    if (lane == 0) W.redo[pair] = 0;
}

__global__ __launch_bounds__(256) void k_sbp_candidates(const orbhip_proj_query *q_, const uint8_t *descq_, const int32_t *nq_, int max_q,
                                                        float min_x, float min_y, float max_x, float max_y, int use_ur, SbpWork W)
{
    __shared__ uint32_t kbuf_all[4][SBPL_BUF];
    const int pair = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + w;
    const int nq = nq_[pair];
    if (t >= nq || nq > W.cap_q || nq > max_q) return;
    uint32_t *kbuf = kbuf_all[w];
    const orbhip_proj_query qq = q_[(size_t)pair * max_q + t];
    const uint4 *dQ = reinterpret_cast<const uint4 *>(descq_ + ((size_t)pair * max_q + t) * 32);
    const uint4 a0 = dQ[0], a1 = dQ[1];
    const float inv_w = __fdiv_rn((float)SI_COLS, __fsub_rn(max_x, min_x)), inv_h = __fdiv_rn((float)SI_ROWS, __fsub_rn(max_y, min_y));
    const float x = qq.u, y = qq.v, r = qq.radius;
    int32_t *count = W.count + (size_t)pair * W.cap_q;
    int c0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, min_x), r), inv_w)); if (c0 < 0) c0 = 0;                   // Frame.cc:656-674
    int c1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, min_x), r), inv_w)); if (c1 > SI_COLS - 1) c1 = SI_COLS - 1;
    int r0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, min_y), r), inv_h)); if (r0 < 0) r0 = 0;
    int r1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, min_y), r), inv_h)); if (r1 > SI_ROWS - 1) r1 = SI_ROWS - 1;
    if (c0 >= SI_COLS || c1 < 0 || r0 >= SI_ROWS || r1 < 0) { if (lane == 0) count[t] = 0; return; }
    const int32_t *cs = W.col_start + (size_t)pair * (SI_COLS + 1);
    const int p0 = cs[c0], p1 = cs[c1 + 1];
    const float4 *rec = W.rec + (size_t)pair * W.cap_n;
    const uint4 *dC = W.desc + (size_t)pair * W.cap_n * 2;
    const uint16_t *idx = W.idx + (size_t)pair * W.cap_n;
    const bool check_lv = qq.min_level > 0 || qq.max_level >= 0;                                                     // Frame.cc:676
    const uint32_t cw = (uint32_t)(c1 - c0), rh = (uint32_t)(r1 - r0);
    int total = 0;
    for (int pb = p0; pb < p1; pb += 64) {
        const int p = pb + lane;
        bool ok = p < p1;
        uint32_t key = 0xFFFFFFFFu;
        if (ok) {
            const float4 k = rec[p];
            const uint32_t bits = __float_as_uint(k.z);
            const uint32_t dcx = (bits & 127u) - (uint32_t)c0, dcy = ((bits >> 7) & 63u) - (uint32_t)r0;
            const int o = (int)((bits >> 13) & 15u);
            ok = (dcx <= cw) & (dcy <= rh);
            ok = ok & !((int)check_lv & ((int)(o < qq.min_level) | ((int)(qq.max_level >= 0) & (int)(o > qq.max_level))));   // Frame.cc:693-701
            ok = ok & (bool)((int)(fabsf(__fsub_rn(k.x, x)) < r) & (int)(fabsf(__fsub_rn(k.y, y)) < r));                    // Frame.cc:704-708
            ok = ok & !((bits >> 17) & 1u);                                                                                  // holds a map point already (:2037 / :96 on entry)
            if (use_ur) ok = ok & !((k.w > 0) & (fabsf(__fsub_rn(qq.ur, k.w)) > r));                                        // ORBmatcher.cc:2041-2047 / 100-105
            if (ok) {
                const int dist = hamming256(a0, a1, dC[2 * p], dC[2 * p + 1]);
                key = ((uint32_t)dist << 22) | ((uint32_t)p << 11) | (uint32_t)idx[p];
            }
        }
        const unsigned long long m = __ballot(ok);
        const int before = __popcll(m & ((1ull << lane) - 1));
        if (ok && total + before < SBPL_BUF) kbuf[total + before] = key;
        total += __popcll(m);
    }
    if (total > SBPL_BUF) { if (lane == 0) count[t] = 0x7FFFFFFF; return; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // rank by counting (keys are unique): the SBPL_K smallest go to the list in order
    uint32_t *L = W.lists + (((size_t)pair * W.chunks + (t >> 6)) * SBPL_K) * 64 + (t & 63);
    for (int e = lane; e < total; e += 64) {
        const uint32_t mine = kbuf[e];
        int rank = 0;
        for (int j = 0; j < total; j++) rank += kbuf[j] < mine;
        if (rank < SBPL_K) L[(size_t)rank * 64] = mine;
    }
    if (lane == 0) count[t] = total;
}

__global__ __launch_bounds__(64) void k_sbp_replay(const orbhip_proj_query *q_, const int32_t *nq_, int max_q, const orbhip_keypoint *kp_,
                                                   const int32_t *n_, int max_n, size_t kp_stride, int th_high, int check_ori, int mode, float nn_ratio,
                                                   SbpWork W, int32_t *tm_, int32_t *nmatches_)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sbp_lds[];
    int32_t *holder = reinterpret_cast<int32_t *>(sbp_lds);                         // [cap_n]  -1 free, -2 pre-held, else query << 1 | has_obs
    int32_t *owner = holder + W.cap_n;                                               // [cap_n]  lowest lane of this round that claims the keypoint (64: none)
    int16_t *qm = reinterpret_cast<int16_t *>(owner + W.cap_n);                      // [cap_q]
    uint8_t *oct = reinterpret_cast<uint8_t *>(qm + W.cap_q);                        // [cap_n]
    int8_t *qbin = reinterpret_cast<int8_t *>(oct + W.cap_n);                        // [cap_q]
    __shared__ int hist[SI_HISTO];
    __shared__ int s_keep[3];
    const int pair = blockIdx.x, lane = threadIdx.x;
    if (W.redo[pair]) return;                                                        // (capacity: the sequential kernel reports it)
    const int n = n_[pair], nq = nq_[pair];
    if (nq > W.cap_q || nq > max_q) { if (lane == 0) W.redo[pair] = 1; return; }
    const orbhip_proj_query *Q = q_ + (size_t)pair * max_q;
    const orbhip_keypoint *kp = kp_ + (size_t)pair * kp_stride;
    int32_t *tm = tm_ + (size_t)pair * max_n;
    const int32_t *count = W.count + (size_t)pair * W.cap_q;
    for (int i = lane; i < SI_HISTO; i += 64) hist[i] = 0;
    for (int i = lane; i < n; i += 64) { holder[i] = tm[i] == -1 ? -1 : -2; owner[i] = 64; oct[i] = (uint8_t)kp[i].octave; }
    for (int t = lane; t < nq; t += 64) { qm[t] = -1; qbin[t] = -1; }
    __syncthreads();
    int nmatches = 0;
    bool give_up = false;
    for (int t0 = 0; t0 < nq && !give_up; t0 += 64) {
        const int t = t0 + lane;
        const bool valid = t < nq;
        const uint32_t *L = W.lists + (((size_t)pair * W.chunks + (t0 >> 6)) * SBPL_K) * 64 + lane;
        const int cnt_all = valid ? count[t] : 0;
        const int cnt = min(cnt_all, SBPL_K);
        const int ho = valid ? (Q[t].has_obs & 1) : 0;
        uint32_t kr0 = 0xFFFFFFFFu, kr1 = 0xFFFFFFFFu, kr2 = 0xFFFFFFFFu, kr3 = 0xFFFFFFFFu;      // the head of the list in registers (one batch of loads)
        if (cnt > 0) kr0 = L[0];
        if (cnt > 1) kr1 = L[64];
        if (cnt > 2) kr2 = L[128];
        if (cnt > 3) kr3 = L[192];
        if (__ballot(cnt_all == 0x7FFFFFFF)) { give_up = true; break; }
        int ptr = 0;
        bool fin = !valid || cnt == 0;
#define SBPL_ENTRY(p) ((p) == 0 ? kr0 : (p) == 1 ? kr1 : (p) == 2 ? kr2 : (p) == 3 ? kr3 : L[(size_t)(p) * 64])
#define SBPL_BLOCKED(key) (holder[(key) & 0x7FFu] >= 0 && (holder[(key) & 0x7FFu] & 1))
        while (true) {
            uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
            bool accept = false, starved = false;
            if (!fin) {
                while (ptr < cnt) { const uint32_t k = SBPL_ENTRY(ptr); if (!SBPL_BLOCKED(k)) { k1 = k; break; } ptr++; }
                if (k1 == 0xFFFFFFFFu) starved = cnt_all > SBPL_K;                    // ran out of a truncated list
                else {
                    const int d1 = (int)(k1 >> 22);
                    accept = d1 <= th_high && d1 < 256;                                // ORBmatcher.cc:2058 / :131
                    if (accept && mode == 1) {                                         // ratio test against the second best of the same octave (:131-137)
                        int p2 = ptr + 1;
                        while (p2 < cnt) { const uint32_t k = SBPL_ENTRY(p2); if (!SBPL_BLOCKED(k)) { k2 = k; break; } p2++; }
                        if (k2 == 0xFFFFFFFFu) starved = cnt_all > SBPL_K;
                        else {
                            const int d2 = (int)(k2 >> 22);
                            if (oct[k1 & 0x7FFu] == oct[k2 & 0x7FFu] && (float)d1 > __fmul_rn(nn_ratio, (float)d2)) accept = false;
                        }
                    }
                }
            }
            if (__ballot(starved)) { give_up = true; break; }
            // speculative claims of this round: the lowest lane that wants a keypoint (only claims of points WITH observations block others)
            const bool claims = !fin && accept && ho;
            if (claims) atomicMin(&owner[k1 & 0x7FFu], lane);
            __syncthreads();
            bool conflict = false;
            if (!fin) {
                if (k1 != 0xFFFFFFFFu) conflict = owner[k1 & 0x7FFu] < lane;
                if (k2 != 0xFFFFFFFFu) conflict = conflict || owner[k2 & 0x7FFu] < lane;
            }
            const unsigned long long cm = __ballot(conflict);
            const int f = cm ? __ffsll((long long)cm) - 1 : 64;                        // lanes below f are final
            __syncthreads();
            if (claims) owner[k1 & 0x7FFu] = 64;
            if (!fin && lane < f) {
                if (accept) {
                    // several final lanes may take one keypoint when the earlier ones carry no observations: the LAST query keeps it (:2061 / :140)
                    atomicMax(&holder[k1 & 0x7FFu], (t << 1) | ho);
                    qm[t] = (int16_t)(k1 & 0x7FFu);
                    nmatches++;
                }
                fin = true;
            }
            __syncthreads();
            if (f == 64) break;
        }
#undef SBPL_ENTRY
#undef SBPL_BLOCKED
    }
    if (give_up) { if (lane == 0) W.redo[pair] = 1; return; }                          // nothing written yet: the sequential kernel does this pair
    nmatches = wave_sum_dpp(nmatches);
    __syncthreads();
    const float factor = 1.0f / SI_HISTO;
    if (check_ori) {                                                                   // rotation consistency, as in k_search_by_projection
        for (int t = lane; t < nq; t += 64) {
            const int best = qm[t];
            if (best < 0) continue;
            float rot = __fsub_rn(Q[t].angle, kp[best].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == SI_HISTO) bin = 0;
            atomicAdd(&hist[bin], 1); qbin[t] = (int8_t)bin;
        }
        __syncthreads();
        if (lane == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < SI_HISTO; i++) {
                const int sz = hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
                else if (sz > max3) { max3 = sz; ind3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
            s_keep[0] = ind1; s_keep[1] = ind2; s_keep[2] = ind3;
        }
        __syncthreads();
        int removed = 0;
        for (int t = lane; t < nq; t += 64) {
            const int b = qbin[t];
            if (b < 0 || b == s_keep[0] || b == s_keep[1] || b == s_keep[2]) continue;
            holder[qm[t]] = -1; removed++;
        }
        nmatches -= wave_sum_dpp(removed);
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) { const int h = holder[i]; tm[i] = h >= 0 ? (h >> 1) : h; }
    if (lane == 0) nmatches_[pair] = nmatches;
}

// LDS of k_search_by_projection for the given row capacities (keypoints / queries per pair)
static size_t sbp_lds_bytes(int cap_n, int cap_q, int ncells)
{
    const int cap_c = cap_n < SBP_CAND_CAP ? cap_n : SBP_CAND_CAP;
    return sizeof(uint32_t) * ((size_t)ncells + 1) + (size_t)cap_n * (4 + 4 + 2 + 2 + 2 + 2 + 1) + (size_t)cap_c * 2 + (size_t)cap_q * (2 + 1) + 16;
}
void *orbhip_ctx_work_internal(orbhip_ctx *c, size_t bytes);
static int sbp_launch(orbhip_ctx *ctx, const orbhip_proj_query *d_q, const uint8_t *d_desc_q, const int32_t *d_nq, int max_q,
                      const orbhip_keypoint *d_kp, const uint8_t *d_desc, const float *d_u_right, const int32_t *d_n, int max_n,
                      size_t frame_stride_kp, int pairs, float min_x, float min_y, float max_x, float max_y, int th_high,
                      int check_orientation, int mode, float nn_ratio, int32_t *d_train_match, int32_t *d_nmatches,
                      const int32_t *d_nleft = nullptr, const int32_t *d_mirror = nullptr)
{
    // Frames / keyframes beyond the replay form's 2048 keypoints or queries (the two first keyframes of a monocular map carry 5 x nFeatures
    // keypoints, Tracking.cc:210; a loop's map points can be thousands of queries) take the sequential kernel alone, whose arrays are
    // sized by what LDS holds: 17 B per keypoint + 2 B per candidate slot + 3 B per query + the grid (round 4; round 3 refused > 2048)
    const bool big = max_n > SBP_CAP || max_q > SBP_CAP;
    const int lim = big ? SBP_SEQ_CAP : SBP_CAP;
    const int cap_n = ((max_n < lim ? max_n : lim) + 7) & ~7, cap_q = ((max_q < lim ? max_q : lim) + 7) & ~7;
    const int ncells = d_nleft ? 2 * SBP_CELLS : SBP_CELLS;
    const size_t base = sbp_lds_bytes(cap_n, cap_q, ncells), with_desc = base + (size_t)cap_n * 32;
    if (base > 160 * 1024 - 512) { orbhip_set_last_error_internal("SearchByProjection: frame too large for the LDS-resident grid (17 B per keypoint + 3 B per query <= ~145 KB)"); return ORBHIP_E_CAPACITY; }
    // descriptors in LDS when at most two rounds of workgroups are needed anyway (<= 2 pairs per CU resident is enough)
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipStream_t st = orbhip_ctx_stream_internal(ctx);
    // The replay form first (candidate lists for all queries at once, then the claim rule replayed over them); a pair it cannot finish
    // (a truncated list ran dry, > SBPL_BUF candidates) is flagged and done by the sequential kernel below, which returns at once for the
    // others.  Measured (tools/sbp_sweep.py, ~1000 queries x ~1000 keypoints per pair): 1 pair 0.15 ms against 1.18 ms, 1023 pairs 0.69 ms
    // against 1.25 ms (last frame); 0.24 / 1.22 ms and 0.80 / 1.32 ms (local map).  Its work arena (lists + CSR copies, ~190 KB per pair
    // at 1000 keypoints) is capped at 1 GiB; beyond that, for rig frames, and when ORBHIP_SBP_PARALLEL_MAX_PAIRS says so, the sequential
    // kernel runs alone.
    const int par_max = getenv("ORBHIP_SBP_PARALLEL_MAX_PAIRS") ? atoi(getenv("ORBHIP_SBP_PARALLEL_MAX_PAIRS")) : (1 << 20);
    const size_t work_per_pair = (size_t)cap_n * (16 + 32 + 2) + 4 * (SI_COLS + 1) + 4 * (size_t)((cap_q + 63) / 64) * SBPL_K * 64 + 4 * (size_t)cap_q + 1024;
    const int32_t *d_redo = nullptr;
    if (!big && !d_nleft && !d_mirror && pairs <= par_max && (size_t)pairs * work_per_pair <= ((size_t)1 << 30)) {
        SbpWork W;
        W.cap_n = cap_n; W.cap_q = cap_q; W.chunks = (cap_q + 63) / 64;
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t o_rec = 0, o_desc = o_rec + al(sizeof(float4) * (size_t)pairs * cap_n), o_idx = o_desc + al(32 * (size_t)pairs * cap_n),
                     o_cs = o_idx + al(2 * (size_t)pairs * cap_n), o_lists = o_cs + al(4 * (size_t)pairs * (SI_COLS + 1)),
                     o_count = o_lists + al(4 * (size_t)pairs * W.chunks * SBPL_K * 64), o_redo = o_count + al(4 * (size_t)pairs * cap_q),
                     total = o_redo + al(4 * (size_t)pairs);
        uint8_t *wb = (uint8_t *)orbhip_ctx_work_internal(ctx, total);
        if (!wb) return ORBHIP_E_HIP;
        W.rec = (float4 *)(wb + o_rec); W.desc = (uint4 *)(wb + o_desc); W.idx = (uint16_t *)(wb + o_idx); W.col_start = (int32_t *)(wb + o_cs);
        W.lists = (uint32_t *)(wb + o_lists); W.count = (int32_t *)(wb + o_count); W.redo = (int32_t *)(wb + o_redo);
        const size_t prep_lds = sizeof(uint32_t) * (SBP_CELLS + 1) + (size_t)cap_n * (4 + 4 + 2 + 2 + 2 + 1) + 16;
        const size_t rep_lds = (size_t)cap_n * (4 + 4 + 1) + (size_t)cap_q * (2 + 1) + 16;
        if (orb_lds_optin(reinterpret_cast<const void *>(k_sbp_prep), orbhip_ctx_device_internal(ctx), prep_lds) ||
            orb_lds_optin(reinterpret_cast<const void *>(k_sbp_replay), orbhip_ctx_device_internal(ctx), rep_lds)) return ORBHIP_E_HIP;
        hipLaunchKernelGGL(k_sbp_prep, dim3(pairs), dim3(64), prep_lds, st, d_kp, d_desc, d_u_right, d_n, max_n, frame_stride_kp, min_x, min_y, max_x, max_y,
                           d_train_match, W, orbhip_ctx_status_internal(ctx));
        hipLaunchKernelGGL(k_sbp_candidates, dim3((cap_q + 3) / 4, pairs), dim3(256), 0, st, d_q, d_desc_q, d_nq, max_q, min_x, min_y, max_x, max_y,
                           d_u_right ? 1 : 0, W);
        hipLaunchKernelGGL(k_sbp_replay, dim3(pairs), dim3(64), rep_lds, st, d_q, d_nq, max_q, d_kp, d_n, max_n, frame_stride_kp, th_high, check_orientation,
                           mode, nn_ratio, W, d_train_match, d_nmatches);
        d_redo = W.redo;
    }
    const bool desc_lds = with_desc <= 150 * 1024 && (size_t)pairs * with_desc <= (size_t)cus * 150 * 1024;
    const size_t lds = desc_lds ? with_desc : base;
    auto kern = desc_lds ? k_search_by_projection<true> : k_search_by_projection<false>;
    if (orb_lds_optin(reinterpret_cast<const void *>(kern), orbhip_ctx_device_internal(ctx), lds)) return ORBHIP_E_HIP;
    hipLaunchKernelGGL(kern, dim3(pairs), dim3(64), lds, st, d_q, d_desc_q, d_nq,
                       max_q, d_kp, d_desc, d_u_right, d_n, max_n, frame_stride_kp, min_x, min_y, max_x, max_y, th_high,
                       check_orientation, mode, nn_ratio, cap_n, cap_q, d_train_match, d_nmatches, orbhip_ctx_status_internal(ctx), d_nleft, d_mirror, ncells,
                       d_redo);
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}

extern "C" int orbhip_search_by_projection_rig_device(orbhip_ctx *ctx, int mode, const orbhip_proj_query *d_q, const uint8_t *d_desc_q,
                                                      const int32_t *d_nq, int max_q, const orbhip_keypoint *d_kp, const uint8_t *d_desc,
                                                      const int32_t *d_n, const int32_t *d_nleft, const int32_t *d_mirror, int max_n,
                                                      size_t frame_stride_kp, int pairs, float min_x, float min_y, float max_x, float max_y,
                                                      int th_high, float nn_ratio, int check_orientation, int32_t *d_train_match,
                                                      int32_t *d_nmatches)
{
    if (!ctx || (mode != 0 && mode != 1) || !d_q || !d_desc_q || !d_nq || !d_kp || !d_desc || !d_n || !d_nleft || pairs <= 0 || max_n <= 0 ||
        max_q <= 0 || !d_train_match || !d_nmatches || !(max_x > min_x) || !(max_y > min_y))
        return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    return sbp_launch(ctx, d_q, d_desc_q, d_nq, max_q, d_kp, d_desc, nullptr, d_n, max_n, frame_stride_kp, pairs, min_x, min_y, max_x, max_y,
                      th_high, mode == 0 ? check_orientation : 0, mode, nn_ratio, d_train_match, d_nmatches, d_nleft, mode == 1 ? d_mirror : nullptr);
}

extern "C" int orbhip_search_by_projection_device(orbhip_ctx *ctx, const orbhip_proj_query *d_q, const uint8_t *d_desc_q,
                                                  const int32_t *d_nq, int max_q, const orbhip_keypoint *d_kp,
                                                  const uint8_t *d_desc, const float *d_u_right, const int32_t *d_n, int max_n,
                                                  size_t frame_stride_kp, int pairs, float min_x, float min_y, float max_x,
                                                  float max_y, int th_high, int check_orientation, int32_t *d_train_match,
                                                  int32_t *d_nmatches)
{
    if (!ctx || !d_q || !d_desc_q || !d_nq || !d_kp || !d_desc || !d_n || pairs <= 0 || max_n <= 0 || max_q <= 0 ||
        !d_train_match || !d_nmatches || !(max_x > min_x) || !(max_y > min_y))
        return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    return sbp_launch(ctx, d_q, d_desc_q, d_nq, max_q, d_kp, d_desc, d_u_right, d_n, max_n, frame_stride_kp, pairs, min_x, min_y,
                      max_x, max_y, th_high, check_orientation, 0, 0.0f, d_train_match, d_nmatches);
}

// ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, ...), ORBmatcher.cc:48-218 (Nleft == -1)
extern "C" int orbhip_search_local_map_device(orbhip_ctx *ctx, const orbhip_proj_query *d_q, const uint8_t *d_desc_q,
                                              const int32_t *d_nq, int max_q, const orbhip_keypoint *d_kp,
                                              const uint8_t *d_desc, const float *d_u_right, const int32_t *d_n, int max_n,
                                              size_t frame_stride_kp, int pairs, float min_x, float min_y, float max_x,
                                              float max_y, int th_high, float nn_ratio, int32_t *d_train_match,
                                              int32_t *d_nmatches)
{
    if (!ctx || !d_q || !d_desc_q || !d_nq || !d_kp || !d_desc || !d_n || pairs <= 0 || max_n <= 0 || max_q <= 0 ||
        !d_train_match || !d_nmatches || !(max_x > min_x) || !(max_y > min_y))
        return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    return sbp_launch(ctx, d_q, d_desc_q, d_nq, max_q, d_kp, d_desc, d_u_right, d_n, max_n, frame_stride_kp, pairs, min_x, min_y,
                      max_x, max_y, th_high, 0, 1, nn_ratio, d_train_match, d_nmatches);
}

// ---------------------------------------------------------------------------- N3: distinctive descriptor
// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:327-403), batched over map points: one wave per point.
// The n x n distance matrix lives in LDS (u16); row i's median = sorted row [int(0.5*(n-1))] is found without
// sorting as the smallest v with #{j : D[i][j] <= v} >= k+1 (binary search over the 257 possible distances).
__global__ __launch_bounds__(64) void k_distinctive(const uint8_t *desc_, const int32_t *n_, int max_n, int32_t *best_idx, uint8_t *best_desc)
{
    extern __shared__ uint16_t dd[];               // [n][n]
    __shared__ uint32_t s_best;
    const int p = blockIdx.x, lane = threadIdx.x;
    const int n = min(n_[p], max_n);
    const uint4 *D = reinterpret_cast<const uint4 *>(desc_ + (size_t)p * max_n * 32);
    if (lane == 0) s_best = 0xFFFFFFFFu;
    if (n <= 0) { if (lane == 0) best_idx[p] = 0; return; }
    for (int i = lane; i < n; i += 64) {
        const uint4 a0 = D[2 * i], a1 = D[2 * i + 1];
        for (int j = i + 1; j < n; j++) {                                   // MapPoint.cc:369-378
            const int d = hamming256(a0, a1, D[2 * j], D[2 * j + 1]);
            dd[i * n + j] = (uint16_t)d; dd[j * n + i] = (uint16_t)d;
        }
        dd[i * n + i] = 0;
    }
    __syncthreads();
    const int k = (int)(0.5 * (n - 1));                                     // MapPoint.cc:387
    for (int i = lane; i < n; i += 64) {
        int lo = 0, hi = 256;                                               // smallest v with count(<= v) >= k+1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
            for (int j = 0; j < n; j++) c += dd[i * n + j] <= mid;
            if (c >= k + 1) hi = mid; else lo = mid + 1;
        }
        atomicMin(&s_best, ((uint32_t)lo << 16) | (uint32_t)i);            // least median, first index on ties (:389-393)
    }
    __syncthreads();
    const int bi = (int)(s_best & 0xFFFFu);
    if (lane == 0) best_idx[p] = bi;
    if (best_desc && lane < 8) reinterpret_cast<uint32_t *>(best_desc + (size_t)p * 32)[lane] = reinterpret_cast<const uint32_t *>(D + 2 * bi)[lane];
}

extern "C" int orbhip_distinctive_descriptors_device(orbhip_ctx *ctx, const uint8_t *d_desc, const int32_t *d_n, int points, int max_n,
                                                     int32_t *d_best_idx, uint8_t *d_best_desc)
{
    if (!ctx || !d_desc || !d_n || points <= 0 || max_n <= 0 || max_n > 256 || !d_best_idx) return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    const size_t lds = sizeof(uint16_t) * (size_t)max_n * max_n;
    if (orb_lds_optin(reinterpret_cast<const void *>(k_distinctive), orbhip_ctx_device_internal(ctx), lds)) return ORBHIP_E_HIP;
    hipLaunchKernelGGL(k_distinctive, dim3(points), dim3(64), lds, orbhip_ctx_stream_internal(ctx), d_desc, d_n, max_n, d_best_idx, d_best_desc);
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}

// ---------------------------------------------------------------------------- N3: BoW tree descent
// DBoW2 TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (TemplatedVocabulary.h:1218-1260),
// one thread per feature; the vocabulary is a flat CSR tree resident in HBM (node descriptors 32 B each).
__global__ __launch_bounds__(256) void k_bow_transform(const uint8_t *desc, const int32_t *n_, int frames, int max_n, size_t frame_stride,
                                                       const uint8_t *node_desc, const int32_t *child_start, const int32_t *child_ids,
                                                       const int32_t *node_word, const double *node_weight, int L, int levelsup,
                                                       int32_t *word_id, double *weight, int32_t *nid)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_[f] || i >= max_n) return;
    const uint4 *a = reinterpret_cast<const uint4 *>(desc + ((size_t)f * frame_stride + i) * 32);
    const uint4 a0 = a[0], a1 = a[1];
    const uint4 *nd = reinterpret_cast<const uint4 *>(node_desc);
    const int nid_level = L - levelsup;
    int out_nid = 0, final_id = 0, level = 0;
    int c0 = child_start[0], c1 = child_start[1];
    do {
        ++level;
        final_id = child_ids[c0];
        int best = hamming256(a0, a1, nd[2 * final_id], nd[2 * final_id + 1]);
        for (int c = c0 + 1; c < c1; c++) {
            const int id = child_ids[c];
            const int d = hamming256(a0, a1, nd[2 * id], nd[2 * id + 1]);
            if (d < best) { best = d; final_id = id; }
        }
        if (level == nid_level) out_nid = final_id;
        c0 = child_start[final_id]; c1 = child_start[final_id + 1];
    } while (c1 > c0);
    const size_t o = (size_t)f * max_n + i;
    word_id[o] = node_word[final_id]; weight[o] = node_weight[final_id]; nid[o] = out_nid;
}

extern "C" int orbhip_bow_transform_device(orbhip_ctx *ctx, const uint8_t *d_desc, const int32_t *d_n, int frames, int max_n,
                                           size_t frame_stride, const uint8_t *d_node_desc, const int32_t *d_child_start,
                                           const int32_t *d_child_ids, const int32_t *d_node_word, const double *d_node_weight,
                                           int L, int levelsup, int32_t *d_word_id, double *d_weight, int32_t *d_nid)
{
    if (!ctx || !d_desc || !d_n || frames <= 0 || max_n <= 0 || !d_node_desc || !d_child_start || !d_child_ids || !d_node_word ||
        !d_node_weight || L <= 0 || !d_word_id || !d_weight || !d_nid) return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    hipLaunchKernelGGL(k_bow_transform, dim3((max_n + 255) / 256, frames), dim3(256), 0, orbhip_ctx_stream_internal(ctx), d_desc, d_n,
                       frames, max_n, frame_stride, d_node_desc, d_child_start, d_child_ids, d_node_word, d_node_weight, L, levelsup,
                       d_word_id, d_weight, d_nid);
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}

// ---------------------------------------------------------------------------- Fuse (search part)
// ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight) (ORBmatcher.cc:1403-1613, NLeft == -1): the window search of :1499-1570 for
// every projected map point.  The queries do not depend on each other (the Replace / AddObservation bookkeeping of :1572-1595
// is the caller's), so one lane owns one query: it walks the grid columns of its window in GetFeaturesInArea order against
// LDS-resident keypoints AND descriptors (no dependent global gathers inside the lane-serial loop).
// BIG (round 4): keyframes of more than 2900 keypoints (up to 8192: the first two keyframes of a monocular map carry 5 x nFeatures,
// Tracking.cc:210) keep only the grid and the keypoint positions in LDS; descriptors and uRight are read from global memory (L2).
template <bool BIG>
__global__ __launch_bounds__(64) void k_fuse_search(const orbhip_proj_query *q_, const uint8_t *descq_, const int32_t *nq_, int max_q,
                                                    const orbhip_keypoint *kp_, const uint8_t *desc_, const float *uright_,
                                                    const int32_t *n_, int max_n, size_t kp_stride, OrbLevelSigma sig,
                                                    float min_x, float min_y, float max_x, float max_y, int cap_n,
                                                    int32_t *best_idx_, int32_t *best_dist_, int32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sbp_lds[];
    uint4 *dlds = reinterpret_cast<uint4 *>(sbp_lds);                                 // [cap_n][2]   (BIG: absent)
    uint32_t *cell_start = reinterpret_cast<uint32_t *>(dlds + (BIG ? 0 : 2 * (size_t)cap_n));   // [SBP_CELLS + 1]
    float *kx = reinterpret_cast<float *>(cell_start + SBP_CELLS + 1), *ky = kx + cap_n, *ur = ky + cap_n;      // (BIG: ur absent)
    uint16_t *items = reinterpret_cast<uint16_t *>(BIG ? ky + cap_n : ur + cap_n), *cell_of = items + cap_n, *rank_of = cell_of + cap_n;
    uint8_t *oct = reinterpret_cast<uint8_t *>(rank_of + cap_n);
    const int pair = blockIdx.x, lane = threadIdx.x;
    const int n = n_[pair], nq = nq_[pair];
    const orbhip_proj_query *Q = q_ + (size_t)pair * max_q;
    const uint4 *dQ = reinterpret_cast<const uint4 *>(descq_ + (size_t)pair * max_q * 32);
    const orbhip_keypoint *kp = kp_ + (size_t)pair * kp_stride;
    const uint4 *dT = reinterpret_cast<const uint4 *>(desc_ + (size_t)pair * kp_stride * 32);
    const float *uright = uright_ ? uright_ + (size_t)pair * kp_stride : nullptr;
    int32_t *bi = best_idx_ + (size_t)pair * max_q, *bd = best_dist_ + (size_t)pair * max_q;
    if (n > cap_n || n > max_n || nq > max_q) {
        if (lane == 0) atomicExch(status, ORBHIP_E_CAPACITY);
        for (int t = lane; t < min(nq, max_q); t += 64) { bi[t] = -1; bd[t] = 256; }
        return;
    }
    const float inv_w = __fdiv_rn((float)SI_COLS, __fsub_rn(max_x, min_x));       // Frame.cc:334-335
    const float inv_h = __fdiv_rn((float)SI_ROWS, __fsub_rn(max_y, min_y));
    for (int c = lane; c <= SBP_CELLS; c += 64) cell_start[c] = 0;
    if (!BIG) for (int i = lane; i < n; i += 64) { dlds[2 * i] = dT[2 * i]; dlds[2 * i + 1] = dT[2 * i + 1]; ur[i] = uright ? uright[i] : -1.0f; }
    __syncthreads();
    sbp_build_grid(cell_start, kx, ky, oct, items, cell_of, rank_of, kp, n, min_x, min_y, inv_w, inv_h, lane);
    for (int T = 0; T < nq; T += 64) {
        const int t = T + lane;
        if (t >= nq) continue;
        const orbhip_proj_query qq = Q[t];
        const float x = qq.u, y = qq.v, r = qq.radius;
        int best = 256, besti = -1;
        int c0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, min_x), r), inv_w)); if (c0 < 0) c0 = 0;   // Frame.cc:656-674
        int c1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, min_x), r), inv_w)); if (c1 > SI_COLS - 1) c1 = SI_COLS - 1;
        int r0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, min_y), r), inv_h)); if (r0 < 0) r0 = 0;
        int r1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, min_y), r), inv_h)); if (r1 > SI_ROWS - 1) r1 = SI_ROWS - 1;
        if (c0 < SI_COLS && c1 >= 0 && r0 < SI_ROWS && r1 >= 0) {
            const uint4 a0 = dQ[2 * t], a1 = dQ[2 * t + 1];
            for (int cx = c0; cx <= c1; cx++) {
                const int k1 = (int)cell_start[cx * SI_ROWS + r1 + 1];
                for (int k = (int)cell_start[cx * SI_ROWS + r0]; k < k1; k++) {
                    const int idx = items[k];
                    const float dx = __fsub_rn(kx[idx], x), dy = __fsub_rn(ky[idx], y);
                    if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;                                   // Frame.cc:704-708
                    const int lv = oct[idx];
                    if (lv < qq.min_level || lv > qq.max_level) continue;                              // ORBmatcher.cc:1527-1528
                    const float ex = __fsub_rn(x, kx[idx]), ey = __fsub_rn(y, ky[idx]);
                    const float kr = BIG ? (uright ? uright[idx] : -1.0f) : ur[idx];
                    if (kr >= 0) {                                                                      // ORBmatcher.cc:1530-1545
                        const float er = __fsub_rn(qq.ur, kr);
                        const float e2 = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(er, er));
                        if ((double)__fmul_rn(e2, sig.inv_sigma2[lv]) > 7.8) continue;
                    } else {                                                                            // ORBmatcher.cc:1546-1556
                        const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                        if ((double)__fmul_rn(e2, sig.inv_sigma2[lv]) > 5.99) continue;
                    }
                    const int dist = BIG ? hamming256(a0, a1, dT[2 * idx], dT[2 * idx + 1]) : hamming256(a0, a1, dlds[2 * idx], dlds[2 * idx + 1]);
                    if (dist < best) { best = dist; besti = idx; }                                     // ORBmatcher.cc:1564-1568
                }
            }
        }
        bi[t] = besti; bd[t] = best;
    }
}

extern "C" int orbhip_fuse_search_device(orbhip_ctx *ctx, const orbhip_proj_query *d_q, const uint8_t *d_desc_q, const int32_t *d_nq,
                                         int max_q, const orbhip_keypoint *d_kp, const uint8_t *d_desc, const float *d_u_right,
                                         const int32_t *d_n, int max_n, size_t frame_stride_kp, int pairs,
                                         const float *inv_level_sigma2, int nlevels, float min_x, float min_y, float max_x, float max_y,
                                         int32_t *d_best_idx, int32_t *d_best_dist)
{
    if (!ctx || !d_q || !d_desc_q || !d_nq || !d_kp || !d_desc || !d_n || pairs <= 0 || max_n <= 0 || max_q <= 0 || !inv_level_sigma2 ||
        nlevels < 1 || nlevels > 16 || !d_best_idx || !d_best_dist || !(max_x > min_x) || !(max_y > min_y)) return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    OrbLevelSigma sig;
    for (int l = 0; l < 16; l++) sig.inv_sigma2[l] = l < nlevels ? inv_level_sigma2[l] : 0.0f;
    // up to 2900 keypoints: keypoints AND descriptors of a keyframe live in LDS (160 KB); beyond (to 8192): positions and grid only
    const bool big = max_n > 2900;
    const int cap_n = ((max_n < (big ? 8192 : 2900) ? max_n : (big ? 8192 : 2900)) + 7) & ~7;
    const size_t lds = (size_t)cap_n * (big ? (4 + 4 + 2 + 2 + 2 + 1) : (32 + 4 + 4 + 4 + 2 + 2 + 2 + 1)) + sizeof(uint32_t) * (SBP_CELLS + 1) + 16;
    if (lds > 160 * 1024 - 512) return ORBHIP_E_BADARG;
    const void *fn = big ? reinterpret_cast<const void *>(k_fuse_search<true>) : reinterpret_cast<const void *>(k_fuse_search<false>);
    if (orb_lds_optin(fn, orbhip_ctx_device_internal(ctx), lds)) return ORBHIP_E_HIP;
    if (big) hipLaunchKernelGGL(k_fuse_search<true>, dim3(pairs), dim3(64), lds, orbhip_ctx_stream_internal(ctx), d_q, d_desc_q, d_nq, max_q, d_kp, d_desc,
                                d_u_right, d_n, max_n, frame_stride_kp, sig, min_x, min_y, max_x, max_y, cap_n, d_best_idx, d_best_dist,
                                orbhip_ctx_status_internal(ctx));
    else hipLaunchKernelGGL(k_fuse_search<false>, dim3(pairs), dim3(64), lds, orbhip_ctx_stream_internal(ctx), d_q, d_desc_q, d_nq, max_q, d_kp, d_desc,
                            d_u_right, d_n, max_n, frame_stride_kp, sig, min_x, min_y, max_x, max_y, cap_n, d_best_idx, d_best_dist,
                            orbhip_ctx_status_internal(ctx));
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}

// ---------------------------------------------------------------------------- SearchByBoW (KeyFrame, Frame)
// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:273-475, F.Nleft == -1).  A Frame feature belongs to exactly
// one vocabulary node, so the "already matched" rule (:321-322) only couples KF features of the SAME node: nodes are
// independent.  One wave per (keyframe, frame) pair, one lane per shared node (binary search of the frame's sorted node list),
// the node's KF features in order, the frame's descriptors and match slots LDS-resident.
struct BowSide { const int32_t *node_ids, *node_start, *feat, *nnodes; const orbhip_keypoint *kp; const uint8_t *desc; };
// KF_MODE: SearchByBoW(KeyFrame*, KeyFrame*) (ORBmatcher.cc:827-967): side F is the second keyframe with its own validity
// flags, the distance test is strict (:909) and the result is indexed by the first keyframe's feature (vpMatches12).
#define BOW_BIG_NODE 32        // frame features under one node from which the wave works on the node together
// BIG (round 4): frames / keyframes of more than 4096 features (to 16384) read the F side's descriptors from global memory
template <bool KF_MODE, bool BIG = false>
__global__ __launch_bounds__(64) void k_search_by_bow(BowSide K, const uint8_t *kf_valid_, const int32_t *nK_, BowSide F, const uint8_t *f_valid_,
                                                      const int32_t *nF_, int max_nodes, int max_n,
                                                      size_t kp_stride, float nn_ratio, int check_ori, int cap_n,
                                                      int32_t *match_f_, int32_t *nmatches_, int32_t *status, const int32_t *nleft_)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t bow_lds[];
    uint4 *dlds = reinterpret_cast<uint4 *>(bow_lds);                       // [cap_n][2] frame descriptors (BIG: absent)
    int16_t *mf = reinterpret_cast<int16_t *>(dlds + (BIG ? 0 : 2 * (size_t)cap_n));    // [cap_n] KF feature matched to frame feature j, -1 free, -2 invalid
    int16_t *inv = mf + cap_n;                                              // [cap_n] (KF_MODE) match of KF1 feature i
    int8_t *fbin = reinterpret_cast<int8_t *>(inv + (KF_MODE ? cap_n : 0)); // [cap_n] rotation bin of that match
    __shared__ int hist[SI_HISTO];
    __shared__ int s_keep[3];
    const int pair = blockIdx.x, lane = threadIdx.x;
    const int nF = nF_[pair], nk = K.nnodes[pair], nf = F.nnodes[pair];
    int32_t *match_f = match_f_ + (size_t)pair * max_n;
    const int nK = KF_MODE ? nK_[pair] : 0;
    // rig frames (F.Nleft != -1, ORBmatcher.cc:338-359): frame features [0, nleft) are the left camera's, the rest the right camera's;
    // each keyframe feature keeps a best / second best per camera
    const int nleft = (!KF_MODE && nleft_) ? nleft_[pair] : -1;
    if (nF > cap_n || nF > max_n || nK > cap_n || nK > max_n || nk > max_nodes || nf > max_nodes) {
        if (lane == 0) { atomicExch(status, ORBHIP_E_CAPACITY); nmatches_[pair] = 0; }
        return;
    }
    const uint8_t *fvalid = KF_MODE ? f_valid_ + (size_t)pair * max_n : nullptr;
    const int32_t *kids = K.node_ids + (size_t)pair * max_nodes, *kst = K.node_start + (size_t)pair * (max_nodes + 1), *kfe = K.feat + (size_t)pair * max_n;
    const int32_t *fids = F.node_ids + (size_t)pair * max_nodes, *fst = F.node_start + (size_t)pair * (max_nodes + 1), *ffe = F.feat + (size_t)pair * max_n;
    const uint8_t *kvalid = kf_valid_ + (size_t)pair * max_n;
    const orbhip_keypoint *kkp = K.kp + (size_t)pair * kp_stride, *fkp = F.kp + (size_t)pair * kp_stride;
    const uint4 *dK = reinterpret_cast<const uint4 *>(K.desc + (size_t)pair * kp_stride * 32);
    const uint4 *dF = reinterpret_cast<const uint4 *>(F.desc + (size_t)pair * kp_stride * 32);
    for (int i = lane; i < SI_HISTO; i += 64) hist[i] = 0;
    for (int j = lane; j < nF; j += 64) {
        if (!BIG) { dlds[2 * j] = dF[2 * j]; dlds[2 * j + 1] = dF[2 * j + 1]; }
        fbin[j] = -1;
        mf[j] = (KF_MODE && !fvalid[j]) ? -2 : -1;                          // :887-891
    }
    if (KF_MODE) for (int i = lane; i < nK; i += 64) inv[i] = -1;
    __syncthreads();
    const float factor = 1.0f / SI_HISTO;
    int mine = 0;
    for (int a0 = 0; a0 < nk; a0 += 64) {
        const int a = a0 + lane;
        if (a >= nk) continue;
        const int nid = kids[a];
        int lo = 0, hi = nf;                                                 // lower_bound of nid in the frame's node list (:435-442)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (fids[mid] < nid) lo = mid + 1; else hi = mid; }
        if (lo >= nf || fids[lo] != nid) continue;
        const int f0 = fst[lo], f1 = fst[lo + 1];
        if (nleft < 0 && f1 - f0 >= BOW_BIG_NODE) continue;                 // big node: the whole wave works on it below
        for (int ik = kst[a]; ik < kst[a + 1]; ik++) {
            const int ri = kfe[ik];
            if (!kvalid[ri]) continue;                                       // :297-302
            const uint4 a0v = dK[2 * ri], a1v = dK[2 * ri + 1];
            int b1 = 256, b2 = 256, bi = -1, b1r = 256, b2r = 256, bir = -1;
            for (int jf = f0; jf < f1; jf++) {                               // :317-360
                const int rj = ffe[jf];
                if (mf[rj] != -1) continue;
                const int dist = (BIG ? hamming256(a0v, a1v, dF[2 * rj], dF[2 * rj + 1]) : hamming256(a0v, a1v, dlds[2 * rj], dlds[2 * rj + 1]));
                if (nleft < 0 || rj < nleft) {
                    if (dist < b1) { b2 = b1; b1 = dist; bi = rj; }
                    else if (dist < b2) b2 = dist;
                } else {
                    if (dist < b1r) { b2r = b1r; b1r = dist; bir = rj; }
                    else if (dist < b2r) b2r = dist;
                }
            }
            auto take = [&](int j) {
                mf[j] = (int16_t)ri;
                mine++;
                if (check_ori) {                                             // :376-388, :406-421
                    float rot = __fsub_rn(kkp[ri].angle, fkp[j].angle);
                    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                    int bin = (int)roundf(__fmul_rn(rot, factor));
                    if (bin == SI_HISTO) bin = 0;
                    atomicAdd(&hist[bin], 1); fbin[j] = (int8_t)bin;
                }
            };
            if (KF_MODE ? b1 < SI_TH_LOW : b1 <= SI_TH_LOW) {                // :362 / :909
                if ((float)b1 < __fmul_rn(nn_ratio, (float)b2)) take(bi);     // :364-391 / :911
                // the right camera's best: inside the left test's TH_LOW branch, no ratio test ("|| true", :393-396)
                if (nleft >= 0 && b1r <= SI_TH_LOW) take(bir);
            }
        }
    }
    // ---- big nodes (coarse vocabularies, relocalisation with levelsup high: hundreds of features under one node): one node at a time, the
    // keyframe features in order (a frame feature claimed by an earlier one is skipped, :317-321), the 64 lanes over the node's frame
    // features; best = smallest (distance << 16 | position) -- the scan's strict "<" keeps the first of equal distances --, second best =
    // distance of the second smallest key.  Nodes are independent of each other (their frame features are disjoint), so doing these
    // after the lane-parallel pass changes nothing.
    if (nleft < 0) {
        for (int a = 0; a < nk; a++) {
            const int nid = kids[a];
            int lo = 0, hi = nf;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (fids[mid] < nid) lo = mid + 1; else hi = mid; }
            if (lo >= nf || fids[lo] != nid) continue;
            const int f0 = fst[lo], f1 = fst[lo + 1];
            if (f1 - f0 < BOW_BIG_NODE) continue;
            __syncthreads();
            // this lane's frame features of the node (positions lane, lane + 64, ...): indices kept in registers for the whole node
            int rjs[8];
#pragma unroll
            for (int u = 0; u < 8; u++) rjs[u] = f0 + lane + 64 * u < f1 ? ffe[f0 + lane + 64 * u] : -1;
            // the keyframe feature (index, validity, descriptor) is fetched two features ahead: the chain below would otherwise start with
            // two dependent global round trips per feature
            const int ik0 = kst[a], ik1 = kst[a + 1];
            int rq[2]; bool vq[2]; uint4 dq0[2], dq1[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                rq[u] = kfe[min(ik0 + u, ik1 - 1)]; vq[u] = kvalid[rq[u]] != 0; dq0[u] = dK[2 * rq[u]]; dq1[u] = dK[2 * rq[u] + 1];
            }
            for (int ik = ik0; ik < ik1; ik++) {
                const int ri = rq[0]; const bool vk = vq[0];
                const uint4 a0v = dq0[0], a1v = dq1[0];
                rq[0] = rq[1]; vq[0] = vq[1]; dq0[0] = dq0[1]; dq1[0] = dq1[1];
                rq[1] = kfe[min(ik + 2, ik1 - 1)]; vq[1] = kvalid[rq[1]] != 0; dq0[1] = dK[2 * rq[1]]; dq1[1] = dK[2 * rq[1] + 1];
                if (!vk) continue;
                uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (f0 + 64 * u >= f1) break;                           // uniform
                    const int rj = rjs[u];
                    if (rj < 0 || mf[rj] != -1) continue;
                    const uint32_t key = ((uint32_t)(BIG ? hamming256(a0v, a1v, dF[2 * rj], dF[2 * rj + 1]) : hamming256(a0v, a1v, dlds[2 * rj], dlds[2 * rj + 1])) << 16) | (uint32_t)(lane + 64 * u);
                    k2 = min(k2, max(k1, key)); k1 = min(k1, key);
                }
                for (int jf = f0 + 512 + lane; jf < f1; jf += 64) {           // (nodes of more than 512 frame features)
                    const int rj = ffe[jf];
                    if (mf[rj] != -1) continue;
                    const uint32_t key = ((uint32_t)(BIG ? hamming256(a0v, a1v, dF[2 * rj], dF[2 * rj + 1]) : hamming256(a0v, a1v, dlds[2 * rj], dlds[2 * rj + 1])) << 16) | (uint32_t)(jf - f0);
                    k2 = min(k2, max(k1, key)); k1 = min(k1, key);
                }
                wave_min2_u32_dpp(k1, k2);
                const int b1 = k1 == 0xFFFFFFFFu ? 256 : (int)(k1 >> 16), b2 = k2 == 0xFFFFFFFFu ? 256 : (int)(k2 >> 16);
                if ((KF_MODE ? b1 < SI_TH_LOW : b1 <= SI_TH_LOW) && (float)b1 < __fmul_rn(nn_ratio, (float)b2)) {
                    const int j = ffe[f0 + (int)(k1 & 0xFFFFu)];
                    if (lane == 0) {
                        mf[j] = (int16_t)ri;
                        mine++;
                        if (check_ori) {
                            float rot = __fsub_rn(kkp[ri].angle, fkp[j].angle);
                            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                            int bin = (int)roundf(__fmul_rn(rot, factor));
                            if (bin == SI_HISTO) bin = 0;
                            atomicAdd(&hist[bin], 1); fbin[j] = (int8_t)bin;
                        }
                    }
                    __syncthreads();                                         // mf[j] is read by every lane for the next keyframe feature
                }
            }
        }
    }
    mine = wave_sum_dpp(mine);
    __syncthreads();
    int removed = 0;
    if (check_ori) {                                                         // :445-470
        if (lane == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < SI_HISTO; i++) {
                const int sz = hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
                else if (sz > max3) { max3 = sz; ind3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
            s_keep[0] = ind1; s_keep[1] = ind2; s_keep[2] = ind3;
        }
        __syncthreads();
        for (int j = lane; j < nF; j += 64) {
            const int b = fbin[j];
            if (b < 0 || b == s_keep[0] || b == s_keep[1] || b == s_keep[2]) continue;
            mf[j] = -1; removed++;
        }
        removed = wave_sum_dpp(removed);
    }
    __syncthreads();
    if (KF_MODE) {
        for (int j = lane; j < nF; j += 64) if (mf[j] >= 0) inv[mf[j]] = (int16_t)j;
        __syncthreads();
        for (int i = lane; i < nK; i += 64) match_f[i] = inv[i];
    } else {
        for (int j = lane; j < nF; j += 64) match_f[j] = mf[j];
    }
    if (lane == 0) nmatches_[pair] = mine - removed;
}

static int bow_launch(orbhip_ctx *ctx, bool kf_mode, const BowSide &K, const uint8_t *d_kf_valid, const int32_t *d_nK, const BowSide &F,
                      const uint8_t *d_f_valid, const int32_t *d_nF, int pairs, int max_nodes, int max_n, size_t frame_stride_kp, float nn_ratio,
                      int check_orientation, int32_t *d_match, int32_t *d_nmatches, const int32_t *d_nleft = nullptr)
{
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    const bool big = max_n > 4096;                               // (the match slots are int16: 16384 features at most)
    const int lim = big ? 16384 : 4096;
    const int cap_n = ((max_n < lim ? max_n : lim) + 7) & ~7;
    const size_t lds = (size_t)cap_n * ((big ? 0 : 32) + 2 + 1 + (kf_mode ? 2 : 0)) + 16;
    {
        const void *fn = kf_mode ? (big ? reinterpret_cast<const void *>(k_search_by_bow<true, true>) : reinterpret_cast<const void *>(k_search_by_bow<true, false>))
                                 : (big ? reinterpret_cast<const void *>(k_search_by_bow<false, true>) : reinterpret_cast<const void *>(k_search_by_bow<false, false>));
        if (orb_lds_optin(fn, orbhip_ctx_device_internal(ctx), lds)) return ORBHIP_E_HIP;
    }
#define BOW_LAUNCH(KF, BG, NL) hipLaunchKernelGGL((k_search_by_bow<KF, BG>), dim3(pairs), dim3(64), lds, orbhip_ctx_stream_internal(ctx), K, d_kf_valid, d_nK, F, d_f_valid, d_nF, \
                           max_nodes, max_n, frame_stride_kp, nn_ratio, check_orientation, cap_n, d_match, d_nmatches, orbhip_ctx_status_internal(ctx), NL)
    if (kf_mode) { if (big) BOW_LAUNCH(true, true, nullptr); else BOW_LAUNCH(true, false, nullptr); }
    else { if (big) BOW_LAUNCH(false, true, d_nleft); else BOW_LAUNCH(false, false, d_nleft); }
#undef BOW_LAUNCH
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}

extern "C" int orbhip_search_by_bow_device(orbhip_ctx *ctx,
        const int32_t *d_kf_node_ids, const int32_t *d_kf_node_start, const int32_t *d_kf_feat, const int32_t *d_kf_nnodes,
        const uint8_t *d_kf_valid, const orbhip_keypoint *d_kf_kp, const uint8_t *d_kf_desc,
        const int32_t *d_f_node_ids, const int32_t *d_f_node_start, const int32_t *d_f_feat, const int32_t *d_f_nnodes,
        const orbhip_keypoint *d_f_kp, const uint8_t *d_f_desc, const int32_t *d_nF,
        int pairs, int max_nodes, int max_n, size_t frame_stride_kp, float nn_ratio, int check_orientation,
        int32_t *d_match_f, int32_t *d_nmatches)
{
    if (!ctx || !d_kf_node_ids || !d_kf_node_start || !d_kf_feat || !d_kf_nnodes || !d_kf_valid || !d_kf_kp || !d_kf_desc || !d_f_node_ids ||
        !d_f_node_start || !d_f_feat || !d_f_nnodes || !d_f_kp || !d_f_desc || !d_nF || pairs <= 0 || max_nodes <= 0 || max_n <= 0 ||
        !d_match_f || !d_nmatches) return ORBHIP_E_BADARG;
    BowSide K = {d_kf_node_ids, d_kf_node_start, d_kf_feat, d_kf_nnodes, d_kf_kp, d_kf_desc};
    BowSide F = {d_f_node_ids, d_f_node_start, d_f_feat, d_f_nnodes, d_f_kp, d_f_desc};
    return bow_launch(ctx, false, K, d_kf_valid, nullptr, F, nullptr, d_nF, pairs, max_nodes, max_n, frame_stride_kp, nn_ratio, check_orientation,
                      d_match_f, d_nmatches);
}

extern "C" int orbhip_search_by_bow_rig_device(orbhip_ctx *ctx,
        const int32_t *d_kf_node_ids, const int32_t *d_kf_node_start, const int32_t *d_kf_feat, const int32_t *d_kf_nnodes,
        const uint8_t *d_kf_valid, const orbhip_keypoint *d_kf_kp, const uint8_t *d_kf_desc,
        const int32_t *d_f_node_ids, const int32_t *d_f_node_start, const int32_t *d_f_feat, const int32_t *d_f_nnodes,
        const orbhip_keypoint *d_f_kp, const uint8_t *d_f_desc, const int32_t *d_nF, const int32_t *d_nleft,
        int pairs, int max_nodes, int max_n, size_t frame_stride_kp, float nn_ratio, int check_orientation,
        int32_t *d_match_f, int32_t *d_nmatches)
{
    if (!ctx || !d_kf_node_ids || !d_kf_node_start || !d_kf_feat || !d_kf_nnodes || !d_kf_valid || !d_kf_kp || !d_kf_desc || !d_f_node_ids ||
        !d_f_node_start || !d_f_feat || !d_f_nnodes || !d_f_kp || !d_f_desc || !d_nF || !d_nleft || pairs <= 0 || max_nodes <= 0 || max_n <= 0 ||
        !d_match_f || !d_nmatches) return ORBHIP_E_BADARG;
    BowSide K = {d_kf_node_ids, d_kf_node_start, d_kf_feat, d_kf_nnodes, d_kf_kp, d_kf_desc};
    BowSide F = {d_f_node_ids, d_f_node_start, d_f_feat, d_f_nnodes, d_f_kp, d_f_desc};
    return bow_launch(ctx, false, K, d_kf_valid, nullptr, F, nullptr, d_nF, pairs, max_nodes, max_n, frame_stride_kp, nn_ratio, check_orientation,
                      d_match_f, d_nmatches, d_nleft);
}

extern "C" int orbhip_search_by_bow_kf_device(orbhip_ctx *ctx,
        const int32_t *d_node_ids1, const int32_t *d_node_start1, const int32_t *d_feat1, const int32_t *d_nnodes1,
        const uint8_t *d_valid1, const orbhip_keypoint *d_kp1, const uint8_t *d_desc1, const int32_t *d_n1,
        const int32_t *d_node_ids2, const int32_t *d_node_start2, const int32_t *d_feat2, const int32_t *d_nnodes2,
        const uint8_t *d_valid2, const orbhip_keypoint *d_kp2, const uint8_t *d_desc2, const int32_t *d_n2,
        int pairs, int max_nodes, int max_n, size_t frame_stride_kp, float nn_ratio, int check_orientation,
        int32_t *d_matches12, int32_t *d_nmatches)
{
    if (!ctx || !d_node_ids1 || !d_node_start1 || !d_feat1 || !d_nnodes1 || !d_valid1 || !d_kp1 || !d_desc1 || !d_n1 || !d_node_ids2 ||
        !d_node_start2 || !d_feat2 || !d_nnodes2 || !d_valid2 || !d_kp2 || !d_desc2 || !d_n2 || pairs <= 0 || max_nodes <= 0 || max_n <= 0 ||
        !d_matches12 || !d_nmatches) return ORBHIP_E_BADARG;
    BowSide K = {d_node_ids1, d_node_start1, d_feat1, d_nnodes1, d_kp1, d_desc1};
    BowSide F = {d_node_ids2, d_node_start2, d_feat2, d_nnodes2, d_kp2, d_desc2};
    return bow_launch(ctx, true, K, d_valid1, d_n1, F, d_valid2, d_n2, pairs, max_nodes, max_n, frame_stride_kp, nn_ratio, check_orientation,
                      d_matches12, d_nmatches);
}

// ---------------------------------------------------------------------------- SearchForTriangulation
// ORBmatcher::SearchForTriangulation (ORBmatcher.cc:969-1210; Pinhole, mpCamera2 == 0) -- the matcher of
// LocalMapping::CreateNewMapPoints.  This fork never sets vbMatched2, so every KF1 keypoint is independent: one block per
// keyframe pair, one thread per KF1 keypoint, KF2's descriptors and flags LDS-resident.
struct TriSide { const int32_t *node_ids, *node_start, *feat, *nnodes; };
struct TriLevels { float scale[16], sigma2[16]; };
#define TRI_THREADS 256
__global__ __launch_bounds__(TRI_THREADS) void k_search_triangulation(const int32_t *nid1_, const uint8_t *mp1_, const orbhip_keypoint *kp1_,
        const uint8_t *desc1_, const float *ur1_, const int32_t *n1_, TriSide S2, const uint8_t *mp2_, const orbhip_keypoint *kp2_,
        const uint8_t *desc2_, const float *ur2_, const int32_t *n2_, const orbhip_tri_pair *geom_, int max_nodes, int max_n,
        size_t kp_stride, TriLevels lv, int check_ori, int cap_n, int32_t *matches12_, int32_t *nmatches_, int32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tri_lds[];
    uint4 *dlds = reinterpret_cast<uint4 *>(tri_lds);                        // [cap_n][2] KF2 descriptors
    uint8_t *flag2 = reinterpret_cast<uint8_t *>(dlds + 2 * (size_t)cap_n);  // [cap_n] bit0: has a map point, bit1: stereo
    int8_t *bin1 = reinterpret_cast<int8_t *>(flag2 + cap_n);                // [cap_n] rotation bin of KF1 keypoint i's match
    __shared__ int hist[SI_HISTO];
    __shared__ int s_keep[3];
    __shared__ int s_cnt;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int n1 = n1_[pair], n2 = n2_[pair], nn2 = S2.nnodes[pair];
    int32_t *matches12 = matches12_ + (size_t)pair * max_n;
    if (n1 > cap_n || n2 > cap_n || n1 > max_n || n2 > max_n || nn2 > max_nodes) {
        if (tid == 0) { atomicExch(status, ORBHIP_E_CAPACITY); nmatches_[pair] = 0; }
        return;
    }
    const int32_t *nid1 = nid1_ + (size_t)pair * max_n;
    const uint8_t *mp1 = mp1_ + (size_t)pair * max_n, *mp2 = mp2_ + (size_t)pair * max_n;
    const float *ur1 = ur1_ ? ur1_ + (size_t)pair * max_n : nullptr, *ur2 = ur2_ ? ur2_ + (size_t)pair * max_n : nullptr;
    const int32_t *ids2 = S2.node_ids + (size_t)pair * max_nodes, *st2 = S2.node_start + (size_t)pair * (max_nodes + 1), *fe2 = S2.feat + (size_t)pair * max_n;
    const orbhip_keypoint *kp1 = kp1_ + (size_t)pair * kp_stride, *kp2 = kp2_ + (size_t)pair * kp_stride;
    const uint4 *d1 = reinterpret_cast<const uint4 *>(desc1_ + (size_t)pair * kp_stride * 32);
    const uint4 *d2 = reinterpret_cast<const uint4 *>(desc2_ + (size_t)pair * kp_stride * 32);
    const orbhip_tri_pair g = geom_[pair];
    for (int i = tid; i < SI_HISTO; i += TRI_THREADS) hist[i] = 0;
    if (tid == 0) s_cnt = 0;
    for (int j = tid; j < n2; j += TRI_THREADS) {
        dlds[2 * j] = d2[2 * j]; dlds[2 * j + 1] = d2[2 * j + 1];
        flag2[j] = (uint8_t)((mp2[j] ? 1 : 0) | ((ur2 && ur2[j] >= 0.0f) ? 2 : 0));
    }
    __syncthreads();
    const float factor = 1.0f / SI_HISTO;
    int mine = 0;
    for (int idx1 = tid; idx1 < n1; idx1 += TRI_THREADS) {
        int best_idx = -1;
        bin1[idx1] = -1;
        const bool st1 = ur1 && ur1[idx1] >= 0.0f;
        if (!mp1[idx1] && !(g.only_stereo && !st1)) {                        // :1039-1048
            const int nid = nid1[idx1];
            int lo = 0, hi = nn2;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (ids2[mid] < nid) lo = mid + 1; else hi = mid; }
            if (lo < nn2 && ids2[lo] == nid) {
                const uint4 a0 = d1[2 * idx1], a1 = d1[2 * idx1 + 1];
                const float x1 = kp1[idx1].x, y1 = kp1[idx1].y;
                // epipolar line in the second image, Pinhole.cpp:130-132
                const float la = __fadd_rn(__fadd_rn(__fmul_rn(x1, g.F12[0]), __fmul_rn(y1, g.F12[3])), g.F12[6]);
                const float lb = __fadd_rn(__fadd_rn(__fmul_rn(x1, g.F12[1]), __fmul_rn(y1, g.F12[4])), g.F12[7]);
                const float lc = __fadd_rn(__fadd_rn(__fmul_rn(x1, g.F12[2]), __fmul_rn(y1, g.F12[5])), g.F12[8]);
                const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
                int best = SI_TH_LOW;
                for (int j = st2[lo]; j < st2[lo + 1]; j++) {                // :1062-1143
                    const int idx2 = fe2[j];
                    const int fl = flag2[idx2];
                    if ((fl & 1) || (g.only_stereo && !(fl & 2))) continue;
                    const int dist = hamming256(a0, a1, dlds[2 * idx2], dlds[2 * idx2 + 1]);
                    if (dist > best) continue;                               // :1073 (best <= TH_LOW always)
                    const orbhip_keypoint k2 = kp2[idx2];
                    if (!st1 && !(fl & 2)) {                                 // :1083-1091
                        const float ex = __fsub_rn(g.ep_x, k2.x), ey = __fsub_rn(g.ep_y, k2.y);
                        if (__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)) < __fmul_rn(100.0f, lv.scale[k2.octave & 15])) continue;
                    }
                    bool ok = g.coarse != 0;
                    if (!ok && den != 0.0f) {                                // Pinhole.cpp:134-143
                        const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, k2.x), __fmul_rn(lb, k2.y)), lc);
                        const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
                        ok = (double)dsqr < 3.84 * (double)lv.sigma2[k2.octave & 15];
                    }
                    if (ok) { best_idx = idx2; best = dist; }
                }
            }
        }
        if (best_idx >= 0) {
            mine++;
            if (check_ori) {                                                 // :1154-1164
                float rot = __fsub_rn(kp1[idx1].angle, kp2[best_idx].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int bin = (int)roundf(__fmul_rn(rot, factor));
                if (bin == SI_HISTO) bin = 0;
                atomicAdd(&hist[bin], 1); bin1[idx1] = (int8_t)bin;
            }
        }
        matches12[idx1] = best_idx;
    }
    __syncthreads();
    if (check_ori) {                                                         // :1171-1189
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < SI_HISTO; i++) {
                const int sz = hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
                else if (sz > max3) { max3 = sz; ind3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
            s_keep[0] = ind1; s_keep[1] = ind2; s_keep[2] = ind3;
        }
        __syncthreads();
        for (int i = tid; i < n1; i += TRI_THREADS) {
            const int b = bin1[i];
            if (b < 0 || b == s_keep[0] || b == s_keep[1] || b == s_keep[2]) continue;
            matches12[i] = -1; mine--;
        }
    }
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (tid == 0) nmatches_[pair] = s_cnt;
}

extern "C" int orbhip_search_for_triangulation_device(orbhip_ctx *ctx,
        const int32_t *d_nid1, const uint8_t *d_has_mp1, const orbhip_keypoint *d_kp1, const uint8_t *d_desc1, const float *d_u_right1,
        const int32_t *d_n1,
        const int32_t *d_node_ids2, const int32_t *d_node_start2, const int32_t *d_feat2, const int32_t *d_nnodes2,
        const uint8_t *d_has_mp2, const orbhip_keypoint *d_kp2, const uint8_t *d_desc2, const float *d_u_right2, const int32_t *d_n2,
        const orbhip_tri_pair *d_pair, int pairs, int max_nodes, int max_n, size_t frame_stride_kp,
        const float *scale_factors, const float *level_sigma2, int nlevels, int check_orientation,
        int32_t *d_matches12, int32_t *d_nmatches)
{
    if (!ctx || !d_nid1 || !d_has_mp1 || !d_kp1 || !d_desc1 || !d_n1 || !d_node_ids2 || !d_node_start2 || !d_feat2 || !d_nnodes2 ||
        !d_has_mp2 || !d_kp2 || !d_desc2 || !d_n2 || !d_pair || pairs <= 0 || max_nodes <= 0 || max_n <= 0 || !scale_factors ||
        !level_sigma2 || nlevels <= 0 || nlevels > 16 || !d_matches12 || !d_nmatches) return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    TriLevels lv;
    for (int l = 0; l < 16; l++) { lv.scale[l] = l < nlevels ? scale_factors[l] : 0.0f; lv.sigma2[l] = l < nlevels ? level_sigma2[l] : 0.0f; }
    const int cap_n = ((max_n < 4096 ? max_n : 4096) + 15) & ~15;
    const size_t lds = (size_t)cap_n * (32 + 1 + 1) + 16;
    if (orb_lds_optin(reinterpret_cast<const void *>(k_search_triangulation), orbhip_ctx_device_internal(ctx), lds)) return ORBHIP_E_HIP;
    TriSide S2 = {d_node_ids2, d_node_start2, d_feat2, d_nnodes2};
    hipLaunchKernelGGL(k_search_triangulation, dim3(pairs), dim3(TRI_THREADS), lds, orbhip_ctx_stream_internal(ctx), d_nid1, d_has_mp1, d_kp1,
                       d_desc1, d_u_right1, d_n1, S2, d_has_mp2, d_kp2, d_desc2, d_u_right2, d_n2, d_pair, max_nodes, max_n, frame_stride_kp, lv,
                       check_orientation, cap_n, d_matches12, d_nmatches, orbhip_ctx_status_internal(ctx));
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}

// ---------------------------------------------------------------------------- Frame glue: UndistortKeyPoints, AssignFeaturesToGrid
// Frame::UndistortKeyPoints (Frame.cc:738-771): cv::undistortPoints(pt, K, dist, R = I, P = K) of OpenCV 3.4.1 (cvUndistortPoints:
// 5 fixed-point iterations of the inverse Brown model in double, result rounded to float); one thread per keypoint.
struct UndistortArgs { double fx, fy, cx, cy, ifx, ify, k[5]; int copy_only; };
__global__ __launch_bounds__(256) void k_undistort(const orbhip_keypoint *kp_, const int32_t *n_, int max_n, size_t kp_stride, UndistortArgs A,
                                                   orbhip_keypoint *out_)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_[f] || i >= max_n) return;
    orbhip_keypoint k = kp_[(size_t)f * kp_stride + i];
    if (!A.copy_only) {
        double x = k.x, y = k.y;
        x = (x - A.cx) * A.ifx; y = (y - A.cy) * A.ify;
        const double x0 = x, y0 = y;
#pragma unroll 1
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = 1.0 / (1 + ((A.k[4] * r2 + A.k[1]) * r2 + A.k[0]) * r2);
            const double dX = 2 * A.k[2] * x * y + A.k[3] * (r2 + 2 * x * x);
            const double dY = A.k[2] * (r2 + 2 * y * y) + 2 * A.k[3] * x * y;
            x = (x0 - dX) * icdist;
            y = (y0 - dY) * icdist;
        }
        k.x = (float)(A.fx * x + A.cx); k.y = (float)(A.fy * y + A.cy);
    }
    out_[(size_t)f * kp_stride + i] = k;
}

extern "C" int orbhip_undistort_keypoints_device(orbhip_ctx *ctx, const orbhip_keypoint *d_kp, const int32_t *d_n, int frames, int max_n,
                                                 size_t frame_stride_kp, float fx, float fy, float cx, float cy, const float *dist_coef,
                                                 int n_dist, orbhip_keypoint *d_kp_un)
{
    if (!ctx || !d_kp || !d_n || frames <= 0 || max_n <= 0 || !dist_coef || n_dist < 4 || n_dist > 5 || !d_kp_un) return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    UndistortArgs A;
    A.fx = fx; A.fy = fy; A.cx = cx; A.cy = cy; A.ifx = 1.0 / (double)fx; A.ify = 1.0 / (double)fy;
    for (int i = 0; i < 5; i++) A.k[i] = i < n_dist ? (double)dist_coef[i] : 0.0;
    A.copy_only = dist_coef[0] == 0.0f;                                    // Frame.cc:740-744
    hipLaunchKernelGGL(k_undistort, dim3((max_n + 255) / 256, frames), dim3(256), 0, orbhip_ctx_stream_internal(ctx), d_kp, d_n, max_n,
                       frame_stride_kp, A, d_kp_un);
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}

// Frame::AssignFeaturesToGrid (Frame.cc:377-408, Nleft == -1) as a CSR per frame (the layout the windowed matchers build privately
// in LDS, exported for host-side GetFeaturesInArea callers): one wave per frame.
__global__ __launch_bounds__(64) void k_assign_grid(const orbhip_keypoint *kp_, const int32_t *n_, int max_n, size_t kp_stride, float min_x,
                                                    float min_y, float inv_w, float inv_h, int cap_n, int32_t *cell_start_, int32_t *items_,
                                                    int32_t *status, const int32_t *nleft_, int ncells)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t ag_lds[];
    uint32_t *cell_start = reinterpret_cast<uint32_t *>(ag_lds);             // [ncells + 1]
    float *kx = reinterpret_cast<float *>(cell_start + ncells + 1);
    float *ky = kx + cap_n;
    uint16_t *items = reinterpret_cast<uint16_t *>(ky + cap_n);
    uint16_t *cell_of = items + cap_n, *rank_of = cell_of + cap_n;
    uint8_t *oct = reinterpret_cast<uint8_t *>(rank_of + cap_n);
    const int f = blockIdx.x, lane = threadIdx.x;
    const int n = n_[f];
    const int nleft = nleft_ ? nleft_[f] : -1;
    int32_t *cs = cell_start_ + (size_t)f * (ncells + 1), *it = items_ + (size_t)f * max_n;
    if (n > cap_n || n > max_n) { if (lane == 0) atomicExch(status, ORBHIP_E_CAPACITY); for (int c = lane; c <= ncells; c += 64) cs[c] = 0; return; }
    for (int c = lane; c <= ncells; c += 64) cell_start[c] = 0;
    __syncthreads();
    sbp_build_grid(cell_start, kx, ky, oct, items, cell_of, rank_of, kp_ + (size_t)f * kp_stride, n, min_x, min_y, inv_w, inv_h, lane, ncells, nleft);
    for (int c = lane; c <= ncells; c += 64) cs[c] = (int32_t)cell_start[c];
    const int tot = (int)cell_start[ncells];
    const int first_right = nleft >= 0 ? (int)cell_start[SBP_CELLS] : tot;       // mGridRight holds i - Nleft (Frame.cc:403)
    for (int i = lane; i < tot; i += 64) it[i] = i < first_right ? items[i] : (int)items[i] - nleft;
}

static int assign_grid_launch(orbhip_ctx *ctx, const orbhip_keypoint *d_kp, const int32_t *d_n, const int32_t *d_nleft, int frames, int max_n,
                              size_t frame_stride_kp, float min_x, float min_y, float max_x, float max_y, int32_t *d_cell_start, int32_t *d_items)
{
    const int cap_n = ((max_n < 8192 ? max_n : 8192) + 7) & ~7;
    const int ncells = d_nleft ? 2 * SBP_CELLS : SBP_CELLS;
    const size_t lds = sizeof(uint32_t) * ((size_t)ncells + 1) + (size_t)cap_n * (4 + 4 + 2 + 2 + 2 + 1) + 16;
    if (orb_lds_optin(reinterpret_cast<const void *>(k_assign_grid), orbhip_ctx_device_internal(ctx), lds)) return ORBHIP_E_HIP;
    const float inv_w = (float)SI_COLS / (max_x - min_x), inv_h = (float)SI_ROWS / (max_y - min_y);      // Frame.cc:334-335
    hipLaunchKernelGGL(k_assign_grid, dim3(frames), dim3(64), lds, orbhip_ctx_stream_internal(ctx), d_kp, d_n, max_n, frame_stride_kp, min_x,
                       min_y, inv_w, inv_h, cap_n, d_cell_start, d_items, orbhip_ctx_status_internal(ctx), d_nleft, ncells);
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}

extern "C" int orbhip_assign_features_to_grid_device(orbhip_ctx *ctx, const orbhip_keypoint *d_kp, const int32_t *d_n, int frames, int max_n,
                                                     size_t frame_stride_kp, float min_x, float min_y, float max_x, float max_y,
                                                     int32_t *d_cell_start, int32_t *d_items)
{
    if (!ctx || !d_kp || !d_n || frames <= 0 || max_n <= 0 || !(max_x > min_x) || !(max_y > min_y) || !d_cell_start || !d_items) return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    return assign_grid_launch(ctx, d_kp, d_n, nullptr, frames, max_n, frame_stride_kp, min_x, min_y, max_x, max_y, d_cell_start, d_items);
}

extern "C" int orbhip_assign_features_to_grid_rig_device(orbhip_ctx *ctx, const orbhip_keypoint *d_kp, const int32_t *d_n, const int32_t *d_nleft,
                                                         int frames, int max_n, size_t frame_stride_kp, float min_x, float min_y, float max_x,
                                                         float max_y, int32_t *d_cell_start, int32_t *d_items)
{
    if (!ctx || !d_kp || !d_n || !d_nleft || frames <= 0 || max_n <= 0 || !(max_x > min_x) || !(max_y > min_y) || !d_cell_start || !d_items) return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    return assign_grid_launch(ctx, d_kp, d_n, d_nleft, frames, max_n, frame_stride_kp, min_x, min_y, max_x, max_y, d_cell_start, d_items);
}

// ---------------------------------------------------------------------------- BowVector / FeatureVector assembly
// Second half of TemplatedVocabulary::transform(features, v, fv, levelsup) (TemplatedVocabulary.h:1139-1208; TF_IDF weighting,
// L1 norm: the ORBvoc settings): from the per-feature (word, weight, node) of k_bow_transform build, per frame,
//   fv  = map<NodeId, vector<feature index>>  flattened as the CSR the SearchByBoW kernels read (nodes ascending, indices in feature order),
//   v   = map<WordId, sum of weights>          as sorted (word, value) arrays, L1-normalised.
// std::map order and accumulation order are reproduced exactly: keys (id << 16 | feature index) are sorted (bitonic, LDS), a word's
// weights are added in feature order (BowVector::addWeight), the norm is the SEQUENTIAL sum over ascending words (BowVector::normalize).
#define BV_THREADS 256
__device__ void bv_bitonic_sort(unsigned long long *k, int np2, int tid)
{
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (np2 >> 1); t += BV_THREADS) {
                const int lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = k[lo], b = k[hi];
                if ((a > b) == up) { k[lo] = b; k[hi] = a; }
            }
        }
    __syncthreads();
}
__global__ __launch_bounds__(BV_THREADS) void k_bow_vectors(const int32_t *wid_, const double *w_, const int32_t *nid_, const int32_t *n_, int max_n,
                                                            int cap_n, int max_nodes, int32_t *node_ids_, int32_t *node_start_, int32_t *feat_,
                                                            int32_t *nnodes_, int32_t *word_ids_, double *word_val_, int32_t *nwords_, int32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t bv_lds[];
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(bv_lds);      // [cap_n] (power of two)
    int32_t *head = reinterpret_cast<int32_t *>(keys + cap_n);                         // [cap_n] 1 where a new id starts / exclusive scan
    __shared__ int s_cnt, s_scan[BV_THREADS];
    __shared__ double s_norm;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = n_[f];
    const int32_t *wid = wid_ + (size_t)f * max_n, *nid = nid_ + (size_t)f * max_n;
    const double *w = w_ + (size_t)f * max_n;
    int32_t *node_ids = node_ids_ + (size_t)f * max_nodes, *node_start = node_start_ + (size_t)f * (max_nodes + 1), *feat = feat_ + (size_t)f * max_n;
    int32_t *word_ids = word_ids_ + (size_t)f * max_n;
    double *word_val = word_val_ + (size_t)f * max_n;
    if (n > cap_n || n > max_n) {
        if (tid == 0) { atomicExch(status, ORBHIP_E_CAPACITY); nnodes_[f] = 0; nwords_[f] = 0; node_start[0] = 0; }
        return;
    }
    for (int pass = 0; pass < 2; pass++) {                   // pass 0: nodes -> FeatureVector, pass 1: words -> BowVector
        const int32_t *id = pass == 0 ? nid : wid;
        for (int i = tid; i < cap_n; i += BV_THREADS)        // stopped words (w <= 0) sort to the end and are dropped
            keys[i] = (i < n && w[i] > 0.0) ? (((unsigned long long)(uint32_t)id[i] << 16) | (unsigned)i) : ~0ull;
        if (tid == 0) s_cnt = 0;
        bv_bitonic_sort(keys, cap_n, tid);
        // number of kept entries and segment heads
        int mine = 0;
        for (int i = tid; i < cap_n; i += BV_THREADS) {
            const bool kept = keys[i] != ~0ull;
            mine += kept;
            head[i] = kept && (i == 0 || (keys[i] >> 16) != (keys[i - 1] >> 16)) ? 1 : 0;
        }
        atomicAdd(&s_cnt, mine);
        __syncthreads();
        const int m = s_cnt;
        // exclusive scan of head[] in blocks of cap_n / BV_THREADS consecutive entries per thread
        const int per = (cap_n + BV_THREADS - 1) / BV_THREADS, b0 = tid * per;
        int loc = 0;
        for (int i = b0; i < min(b0 + per, cap_n); i++) loc += head[i];
        s_scan[tid] = loc;
        __syncthreads();
        if (tid == 0) { int run = 0; for (int t = 0; t < BV_THREADS; t++) { const int v = s_scan[t]; s_scan[t] = run; run += v; } s_cnt = run; }
        __syncthreads();
        const int nseg = s_cnt;
        if (nseg > (pass == 0 ? max_nodes : max_n)) { if (tid == 0) atomicExch(status, ORBHIP_E_CAPACITY); }
        int run = s_scan[tid];
        for (int i = b0; i < min(b0 + per, cap_n); i++) {
            if (i >= m) break;
            const int seg = run + head[i] - 1;               // index of the segment entry i belongs to
            if (head[i]) {
                run++;
                if (pass == 0) { if (seg < max_nodes) { node_ids[seg] = (int32_t)(keys[i] >> 16); node_start[seg] = i; } }
                else if (seg < max_n) word_ids[seg] = (int32_t)(keys[i] >> 16);
            }
            if (pass == 0) feat[i] = (int32_t)(keys[i] & 0xFFFFu);
            else head[i] = head[i] ? -(seg + 1) : 0;         // mark heads with their segment for the sums below
        }
        __syncthreads();
        if (pass == 0) {
            if (tid == 0) { nnodes_[f] = min(nseg, max_nodes); node_start[min(nseg, max_nodes)] = m; }
        } else {
            // a word's weights in feature order (addWeight), one thread per word
            for (int i = tid; i < m; i += BV_THREADS) {
                if (head[i] >= 0) continue;
                const int seg = -head[i] - 1;
                double acc = 0.0;
                const unsigned long long wkey = keys[i] >> 16;
                for (int j = i; j < m && (keys[j] >> 16) == wkey; j++) acc += w[(int)(keys[j] & 0xFFFFu)];
                if (seg < max_n) word_val[seg] = acc;
            }
            __syncthreads();
            __threadfence_block();
            if (tid == 0) {                                   // BowVector::normalize(L1): sequential sum in ascending word order
                const int nw = min(nseg, max_n);
                double norm = 0.0;
                for (int k2 = 0; k2 < nw; k2++) norm += fabs(word_val[k2]);
                s_norm = norm;
                nwords_[f] = nw;
            }
            __syncthreads();
            if (s_norm > 0.0) {
                const double norm = s_norm;
                for (int k2 = tid; k2 < min(nseg, max_n); k2 += BV_THREADS) word_val[k2] /= norm;
            }
        }
        __syncthreads();
    }
}

extern "C" int orbhip_bow_vectors_device(orbhip_ctx *ctx, const int32_t *d_word_id, const double *d_weight, const int32_t *d_node_id,
                                         const int32_t *d_n, int frames, int max_n, int max_nodes,
                                         int32_t *d_node_ids, int32_t *d_node_start, int32_t *d_feat, int32_t *d_nnodes,
                                         int32_t *d_bow_word, double *d_bow_value, int32_t *d_nwords)
{
    if (!ctx || !d_word_id || !d_weight || !d_node_id || !d_n || frames <= 0 || max_n <= 0 || max_n > 4096 || max_nodes <= 0 || !d_node_ids ||
        !d_node_start || !d_feat || !d_nnodes || !d_bow_word || !d_bow_value || !d_nwords) return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    int cap_n = 64;
    while (cap_n < max_n) cap_n <<= 1;
    const size_t lds = (size_t)cap_n * (8 + 4) + 16;
    if (orb_lds_optin(reinterpret_cast<const void *>(k_bow_vectors), orbhip_ctx_device_internal(ctx), lds)) return ORBHIP_E_HIP;
    hipLaunchKernelGGL(k_bow_vectors, dim3(frames), dim3(BV_THREADS), lds, orbhip_ctx_stream_internal(ctx), d_word_id, d_weight, d_node_id, d_n, max_n,
                       cap_n, max_nodes, d_node_ids, d_node_start, d_feat, d_nnodes, d_bow_word, d_bow_value, d_nwords, orbhip_ctx_status_internal(ctx));
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}
