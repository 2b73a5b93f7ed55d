// orb_kernels.hip -- hand-written gfx950 kernels for the ORB front-end
// (ORBextractor::operator(), reference src/ORBextractor.cc:1068-1150).
//
// One launch processes a whole batch of frames; blockIdx carries (tile|cell|slot, frame).
// Integer/byte work throughout: HBM/LDS/VALU-bound, no MFMA (nothing here is GEMM-shaped).
// Wave = 64 lanes everywhere.  Compile: hipcc --offload-arch=gfx950 -ffp-contract=off.
#include "orb_internal.h"
#include "wave_dpp.h"
#include <mutex>
#include <vector>
#include <cstdio>
#include <cstdlib>

#define WAVE 64

// ----------------------------------------------------------------------------------
// helpers
// ----------------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v) { return wave_scan_add_dpp(v); }   // DPP path: no ds_bpermute round trips (wave_dpp.h)

// Exclusive scan of one int per thread over a 256-thread block. `wsum` = 8 ints of LDS.
// Returns exclusive prefix; *total = block sum.  Contains two __syncthreads().
__device__ __forceinline__ int block_excl_scan256(int v, int *wsum, int *total)
{
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
    int inc = wave_incl_scan(v);
    if (lane == WAVE - 1) wsum[wid] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        int s = wsum[w];
        if (w < wid) base += s;
    }
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + inc - v;
}

// XCD-aware workgroup order (speed only, never correctness): workgroups are dealt round-robin over the
// 8 XCDs, so raw neighbours b, b+1 sit on different L2s.  This bijection gives every XCD one contiguous
// range of logical ids, so that tiles sharing 128-byte lines (horizontal neighbours, vertical aprons,
// consecutive frames) are fetched from HBM by one L2 instead of up to eight.
__device__ __forceinline__ unsigned xcd_logical_id(unsigned bid, unsigned nblocks)
{
    const unsigned q = nblocks >> 3, r = nblocks & 7, k = bid & 7, j = bid >> 3;
    return k * q + min(k, r) + j;
}


// ----------------------------------------------------------------------------------
// A2/A3  pyramid: level l from level l-1, cv::resize INTER_LINEAR 8UC1 fixed point
// (reference call ORBextractor.cc:1165; arithmetic SURVEY Appendix A.3).
// Each thread produces 4 horizontally adjacent output pixels and stores one dword.
// Coefficient tables are computed on the host in float exactly like OpenCV does.
// ----------------------------------------------------------------------------------
#define RS_TW 64              // output tile
#define RS_TH 32
#define RS_MAXC 144           // staged source columns (scale factor <= 2: 64*2 + apron, dword aligned)
#define RS_MAXR 68            // staged source rows
// Separable inside LDS: (A) stage the source footprint of a 64x32 output tile with aligned dword
// loads (vertical clamping of cv::resize baked into the staged rows), (B) horizontal pass for every
// staged row -> (S[sx]*a0 + S[sx+1]*a1) >> 4 as u16 (<= 32640), (C) vertical pass + rounding,
// 4 pixels per thread, one dword store.  Bit-identical to the scalar formula of Appendix A.3.
__global__ __launch_bounds__(256) void k_resize(OrbParams P, int level)
{
    __shared__ __attribute__((aligned(16))) uint32_t in[RS_MAXR * (RS_MAXC / 4)];
    __shared__ uint16_t hz[RS_MAXR * RS_TW];
    const OrbLevel &D = P.lv[level];
    const OrbLevel &S = P.lv[level - 1];
    const int tid = threadIdx.x;
    const int ntx = (D.w + RS_TW - 1) / RS_TW, nty = (D.h + RS_TH - 1) / RS_TH;
    const unsigned lid = xcd_logical_id(blockIdx.x, gridDim.x);
    const int frame = lid / (ntx * nty), trem = lid - frame * (ntx * nty);
    const int dx0 = (trem % ntx) * RS_TW, dy0 = (trem / ntx) * RS_TH;
    const int dx_last = min(dx0 + RS_TW, D.w) - 1, dy_last = min(dy0 + RS_TH, D.h) - 1;
    const int ybase = D.yofs[dy0];                                   // may be -1
    const int nrows = min(D.yofs[dy_last] + 1 - ybase + 1, RS_MAXR);
    const int xbase = D.xofs[dx0] & ~15;
    const int nc16 = min((D.xofs[dx_last] + 1 - xbase) / 16 + 1, RS_MAXC / 16);
    const uint8_t *src = S.img + (size_t)frame * S.img_frame_stride;
    // coefficient tables of passes B and C: fetched now so that their latency hides behind the staging loads
    const int dxl = tid & 63, dxB = min(dx0 + dxl, D.w - 1);
    const int sxB = D.xofs[dxB], a0 = D.xalpha[2 * dxB], a1 = D.xalpha[2 * dxB + 1];
    int r0C[2], b0C[2], b1C[2];
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int dy = min(dy0 + 2 * (tid >> 4) + rr, D.h - 1);
        r0C[rr] = D.yofs[dy]; b0C[rr] = D.ybeta[2 * dy]; b1C[rr] = D.ybeta[2 * dy + 1];
    }
    // ---- A: 16-byte chunks, 8 chunk slots x 32 rows per sweep (index math is shifts only); the loads of a
    // sweep are issued before its LDS stores
    for (int c = tid & 7; c < nc16; c += 8) {
        const int x = xbase + 16 * c;
        for (int r0 = tid >> 3; r0 < nrows; r0 += 64) {
            uint4 reg[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int r = r0 + 32 * k;
                const int y = min(max(ybase + r, 0), S.h - 1);                   // clip(sy, 0, ssize.height)
                const uint8_t *row = src + (size_t)y * S.img_pitch;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (r < nrows) {
                    if (x + 15 < S.w) v = *reinterpret_cast<const uint4 *>(row + x);
                    else {
                        uint32_t d[4] = {0, 0, 0, 0};
#pragma unroll 1
                        for (int j = 0; j < 4; j++) {                             // right image edge only: keep it small
#pragma unroll
                            for (int m = 0; m < 4; m++) d[m] |= (uint32_t)row[min(x + 4 * m + j, S.w - 1)] << (8 * j);
                        }
                        v = make_uint4(d[0], d[1], d[2], d[3]);
                    }
                }
                reg[k] = v;
            }
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int r = r0 + 32 * k;
                if (r < nrows) *reinterpret_cast<uint4 *>(&in[r * (RS_MAXC / 4) + 4 * c]) = reg[k];
            }
        }
    }
    __syncthreads();
    // ---- B
    {
        if (dx0 + dxl < D.w) {
            const int sx = sxB - xbase;
            const uint8_t *inb = reinterpret_cast<const uint8_t *>(in);
            for (int r = tid >> 6; r < nrows; r += 4) {
                const uint8_t *q = inb + r * RS_MAXC + sx;
                hz[r * RS_TW + dxl] = (uint16_t)((q[0] * a0 + q[1] * a1) >> 4);
            }
        }
    }
    __syncthreads();
    // ---- C
    {
        const int c4 = tid & 15, rg = tid >> 4;
        const int dx = dx0 + 4 * c4;
        if (dx < D.w) {
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int dy = dy0 + 2 * rg + rr;
                if (dy < D.h) {
                    const int r0 = r0C[rr] - ybase, b0 = b0C[rr], b1 = b1C[rr];
                    const uint2 t0 = *reinterpret_cast<const uint2 *>(&hz[r0 * RS_TW + 4 * c4]);
                    const uint2 t1 = *reinterpret_cast<const uint2 *>(&hz[(r0 + 1) * RS_TW + 4 * c4]);
                    const int u0[4] = {(int)(t0.x & 0xFFFF), (int)(t0.x >> 16), (int)(t0.y & 0xFFFF), (int)(t0.y >> 16)};
                    const int u1[4] = {(int)(t1.x & 0xFFFF), (int)(t1.x >> 16), (int)(t1.y & 0xFFFF), (int)(t1.y >> 16)};
                    uint32_t packed = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int v = (((b0 * u0[j]) >> 16) + ((b1 * u1[j]) >> 16) + 2) >> 2;
                        packed |= (uint32_t)(v & 255) << (8 * j);
                    }
                    *reinterpret_cast<uint32_t *>(D.img + (size_t)frame * D.img_frame_stride + (size_t)dy * D.img_pitch + dx) = packed;
                }
            }
        }
    }
}

// Row-streaming form (the default whenever 4 output columns read at most 8 consecutive source bytes, i.e. scale factors up to
// ~1.5): no LDS, no barriers.  A lane owns 4 output columns and walks RSR output rows down; per output row it loads its 8 source
// bytes of the two source rows (unaligned global_load_dwordx2, L1/L2 hits after the first touch), v_perm_b32 drops (S[sx], S[sx+1])
// into the halves of a dword and ONE v_dot2_u32_u16 is the horizontal pass S[sx]*a0 + S[sx+1]*a1; the vertical pass is
// v_mul_hi_u32_u24 on pre-shifted operands ((b << 12) * (h & ~15)) >> 32 == (b * (h >> 4)) >> 16, the two halves summed with
// the rounding constant by v_add3 and packed.  ~12 vector instructions per output pixel against 33 for the tiled kernel.
#define RSR 16
typedef unsigned short rs_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t rs_hpass(uint2 q, uint32_t sel, uint32_t al)
{
    const uint32_t pq = __builtin_amdgcn_perm(q.y, q.x, sel);
    return __builtin_amdgcn_udot2(__builtin_bit_cast(rs_u16x2, pq), __builtin_bit_cast(rs_u16x2, al), 0u, false) & ~15u;
}
__device__ __forceinline__ uint32_t rs_mulhi24(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// Source bytes of one lane and one source row, as the 8 bytes starting at its (unaligned) base.  MODE 0: one unaligned
// global_load_dwordx2 (the texture addresser splits it: measured 30-40 % slower, as is a 4-byte-aligned dwordx2); MODE 1: aligned
// dwordx3 from base4 <= base, the bytes shifted back by two v_perm_b32 with a per-lane selector (shift 0..4: where the third dword
// would leave a row of the caller's level-0 buffer, base4 is one dword lower instead).
typedef uint32_t rs_u32x3 __attribute__((ext_vector_type(3)));
typedef rs_u32x3 rs_u32x3_a4 __attribute__((aligned(4)));
template <int MODE>
__device__ __forceinline__ uint2 rs_load8(const uint8_t *src, uint32_t off, uint32_t shsel)
{
    if (MODE == 0) return *reinterpret_cast<const uint2 *>(src + off);
    const rs_u32x3 v = *reinterpret_cast<const rs_u32x3_a4 *>(src + off);
    return make_uint2(__builtin_amdgcn_perm(v.y, v.x, shsel), __builtin_amdgcn_perm(v.z, v.y, shsel));
}
template <int MODE>
__global__ __launch_bounds__(256) void k_resize_rows(OrbParams P, int level, int nbx)
{
    extern __shared__ uint4 yl[];                       // (256 / nch + 2) * RSR + 2 entries: small, so that the kernel fits beside LDS-heavy ones
    const OrbLevel &D = P.lv[level];
    const OrbLevel &S = P.lv[level - 1];
    const unsigned lid = xcd_logical_id(blockIdx.x, gridDim.x);
    const unsigned frame = lid / (unsigned)nbx, bx = lid - frame * (unsigned)nbx;          // uniform
    const int nch = (D.w + 3) >> 2, nb = (D.h + RSR - 1) / RSR, hlast = D.h - 1;
    // the vertical table of this workgroup's bands -> LDS (its per-row lookups are then off the global-load chain)
    const int row_lo = (int)((bx * 256) / (unsigned)nch) * RSR;
    const int row_hi = min((int)((bx * 256 + 255) / (unsigned)nch) * RSR + RSR + 1, hlast);
    for (int r = row_lo + (int)threadIdx.x; r <= row_hi; r += 256) yl[r - row_lo] = reinterpret_cast<const uint4 *>(D.ytab)[r];
    __syncthreads();
    const unsigned g = bx * 256 + threadIdx.x;
    if (g >= (unsigned)(nch * nb)) return;
    const unsigned band = g / (unsigned)nch, c = g - band * (unsigned)nch;
    const uint4 *xc = reinterpret_cast<const uint4 *>(D.xchunk) + 3 * c;
    const uint4 x0 = xc[0], x1 = xc[1];
    const uint4 x2 = xc[2];
    const uint32_t al3 = x2.x, shsel = x2.y;
    const uint32_t base = x0.x, sel0 = x0.y, sel1 = x0.z, sel2 = x0.w, sel3 = x1.x, al0 = x1.y, al1 = x1.z, al2 = x1.w;
    const uint8_t *src = S.img + (size_t)frame * S.img_frame_stride;
    uint8_t *dst = D.img + (size_t)frame * D.img_frame_stride;
    const int dy0 = band * RSR;
    const uint4 *yb = yl + (dy0 - row_lo);
    const int klast = hlast - dy0;                                                         // rows of this band that exist: k <= klast
    const uint32_t spitch = (uint32_t)S.img_pitch;
    // the source rows of the NEXT four output rows are in flight while four are computed (the kernel is bound by bytes in flight)
    uint4 yt[2][4];
    uint2 qa[2][4], qb[2][4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        yt[0][u] = yb[min(u, klast)];
        qa[0][u] = rs_load8<MODE>(src, __umul24(yt[0][u].x, spitch) + base, shsel);
        qb[0][u] = rs_load8<MODE>(src, __umul24(yt[0][u].y, spitch) + base, shsel);
    }
    uint32_t doff = __umul24((uint32_t)dy0, (uint32_t)D.img_pitch) + 4 * c;
#pragma unroll
    for (int g = 0; g < RSR / 4; g++) {
        if (g + 1 < RSR / 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint4 t = yb[min(4 * (g + 1) + u, klast)];
                yt[(g + 1) & 1][u] = t;
                qa[(g + 1) & 1][u] = rs_load8<MODE>(src, __umul24(t.x, spitch) + base, shsel);
                qb[(g + 1) & 1][u] = rs_load8<MODE>(src, __umul24(t.y, spitch) + base, shsel);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = 4 * g + u;
            const uint4 t = yt[g & 1][u];
            const uint2 a = qa[g & 1][u], bq = qb[g & 1][u];
            const uint32_t s0 = rs_mulhi24(t.z, rs_hpass(a, sel0, al0)) + rs_mulhi24(t.w, rs_hpass(bq, sel0, al0)) + 2u;
            const uint32_t s1 = rs_mulhi24(t.z, rs_hpass(a, sel1, al1)) + rs_mulhi24(t.w, rs_hpass(bq, sel1, al1)) + 2u;
            const uint32_t s2 = rs_mulhi24(t.z, rs_hpass(a, sel2, al2)) + rs_mulhi24(t.w, rs_hpass(bq, sel2, al2)) + 2u;
            const uint32_t s3 = rs_mulhi24(t.z, rs_hpass(a, sel3, al3)) + rs_mulhi24(t.w, rs_hpass(bq, sel3, al3)) + 2u;
            const uint32_t p01 = (s0 | (s1 << 16)) >> 2, p23 = (s2 | (s3 << 16)) >> 2;          // bytes 0 and 2 hold the pixels
            // rows below the image repeat the last row's (identical) dword: no branch around the store
            *reinterpret_cast<uint32_t *>(dst + doff) = __builtin_amdgcn_perm(p23, p01, 0x06040200u);
            doff += k < klast ? (uint32_t)D.img_pitch : 0u;
        }
    }
}

void orb_launch_resize(const OrbParams &P, int level, hipStream_t s)
{
    const OrbLevel &D = P.lv[level];
    if (D.xchunk && P.batch >= P.rows_min_batch) {
        const int nch = (D.w + 3) >> 2, nb = (D.h + RSR - 1) / RSR, nbx = (nch * nb + 255) / 256;
        const dim3 grid((unsigned)nbx * (unsigned)P.batch);
        const size_t lds = (size_t)((256 / nch + 2) * RSR + 2) * sizeof(uint4);          // <= 4.4 KB (nch >= 1)
        if (D.resize_mode == 1) hipLaunchKernelGGL(k_resize_rows<1>, grid, dim3(256), lds, s, P, level, nbx);
        else hipLaunchKernelGGL(k_resize_rows<0>, grid, dim3(256), lds, s, P, level, nbx);
        return;
    }
    const unsigned nblocks = (unsigned)(((D.w + RS_TW - 1) / RS_TW) * ((D.h + RS_TH - 1) / RS_TH)) * (unsigned)P.batch;
    hipLaunchKernelGGL(k_resize, dim3(nblocks), dim3(256), 0, s, P, level);
}

// ----------------------------------------------------------------------------------
// A3/A4  per-cell FAST-9/16 + score + 3x3 NMS + two-threshold retry, ONE kernel, one wave per cell
// (ORBextractor.cc:783-854 calling cv::FAST twice; OpenCV FAST_t<16> / cornerScore<16>).
//
// Identity used (tests/test_oracle_orb.py::test_fast_arc_score_identity):
//   S = max over the 16 contiguous 9-arcs of min(v-p) and of min(p-v);
//   corner at threshold t  <=>  S > t ;  cornerScore == S-1.
// On RAW pixel values:  S = max(v - Hi, Lo - v),  Hi = min over arcs of (max of the arc's pixels), Lo = max over arcs of
// (min of the arc's pixels): the sixteen subtractions v - p disappear from the arc search.
//
// Per cell (one wave, no workgroup barriers): (1) the cell's sub-image (band + 3 px) is staged into LDS as packed u16
// PAIRS -- dword p of a row holds pixels (2p-1, 2p) relative to the sub-image, so a band pixel pair and its circle
// neighbours at even dx are plain dword reads and odd dx is one v_alignbit; (2) every band pixel pair runs the cheap
// necessary test ("a 9-arc contains one pixel of every diametral pair", 4 compass pairs, raw values) and is classified
// against BOTH thresholds; pairs that may hold a corner at iniThFAST go to an ordered queue; (3) the queue is scored
// densely (one pair per lane); (4) 3x3 strict NMS from an LDS score tile and ordered emission (cv::FAST order).  Only if
// the cell yields nothing at iniThFAST are the remaining candidates (minThFAST < M <= iniThFAST) scored and the NMS
// repeated at minThFAST (ORBextractor.cc:825-828).  NMS is threshold independent on raw scores: a corner at t survives
// iff score >= t and score > every neighbour's raw score, and a neighbour that is no corner at t has a smaller score
// anyway, so scores that were never computed count as 0.  No score map leaves the kernel.
// ----------------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));

__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s16x2 pkmin(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pkmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
// Three-input packed min / max (round 4).  gfx950 has no v_pk_min3_i16, but v_pk_minimum3_f16 / v_pk_maximum3_f16 -- and a u16 pattern 0..255
// read as an f16 is the subnormal k x 2^-24 (f16 denormals are not flushed), whose order is the integers' order: for pixel values the two
// instructions ARE the integer min / max of three, exactly (all 2^24 triples checked on the device) and at the full rate (4.7 cycles per
// wave-instruction).  Only for operands known to be pixel values (never -1 / NaN patterns).
__device__ __forceinline__ s16x2 pkmin3(s16x2 a, s16x2 b, s16x2 c)
{
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(as_u32(a)), "v"(as_u32(b)), "v"(as_u32(c)));
    return as_s16x2(r);
}
__device__ __forceinline__ s16x2 pkmax3(s16x2 a, s16x2 b, s16x2 c)
{
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(as_u32(a)), "v"(as_u32(b)), "v"(as_u32(c)));
    return as_s16x2(r);
}
// pixel pair starting one pixel to the right of dword `lo`'s first pixel: (lo.hi, hi.lo)
__device__ __forceinline__ s16x2 pair_odd(uint32_t hi, uint32_t lo) { return as_s16x2(__builtin_amdgcn_alignbit(hi, lo, 16)); }

// Necessary-test value M of the band pixel pair whose centre dword is q[0] (q points into the pair tile, PITCH dwords per
// row): M = max(v - max_j min(a_j, b_j), min_j max(a_j, b_j) - v) over the 4 compass diametral pairs; S > t implies M > t.
template <int PITCH>
__device__ __forceinline__ s16x2 fc_compass(const uint32_t *q)
{
    const s16x2 v = as_s16x2(q[0]);
    const s16x2 n = as_s16x2(q[3 * PITCH]), s = as_s16x2(q[-3 * PITCH]);
    const s16x2 e = pair_odd(q[2], q[1]), w = pair_odd(q[-1], q[-2]);
    const s16x2 se = as_s16x2(q[2 * PITCH + 1]), nw = as_s16x2(q[-2 * PITCH - 1]);
    const s16x2 ne = as_s16x2(q[-2 * PITCH + 1]), sw = as_s16x2(q[2 * PITCH - 1]);
    const s16x2 lo = pkmax(pkmax3(pkmin(n, s), pkmin(e, w), pkmin(se, nw)), pkmin(ne, sw));
    const s16x2 hi = pkmin(pkmin3(pkmax(n, s), pkmax(e, w), pkmax(se, nw)), pkmax(ne, sw));
    return pkmax(v - lo, hi - v);
}

// Arc score S (threshold independent, packed) of the band pixel pair at q[0].  Circle in cv::makeOffsets order, k = 0..15:
// (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
template <int PITCH>
__device__ __forceinline__ s16x2 fc_arc_score(const uint32_t *q)
{
    s16x2 p[16];
    {
        const uint32_t *r = q + 3 * PITCH;                      // dy = +3: k = 15, 0, 1
        p[15] = pair_odd(r[0], r[-1]); p[0] = as_s16x2(r[0]); p[1] = pair_odd(r[1], r[0]);
        r = q + 2 * PITCH; p[14] = as_s16x2(r[-1]); p[2] = as_s16x2(r[1]);
        r = q + PITCH; p[13] = pair_odd(r[-1], r[-2]); p[3] = pair_odd(r[2], r[1]);
        p[12] = pair_odd(q[-1], q[-2]); p[4] = pair_odd(q[2], q[1]);
        r = q - PITCH; p[11] = pair_odd(r[-1], r[-2]); p[5] = pair_odd(r[2], r[1]);
        r = q - 2 * PITCH; p[10] = as_s16x2(r[-1]); p[6] = as_s16x2(r[1]);
        r = q - 3 * PITCH; p[9] = pair_odd(r[0], r[-1]); p[8] = as_s16x2(r[0]); p[7] = pair_odd(r[1], r[0]);
    }
    // The two 9-arcs starting at j-1 and at j (j odd) share the 8 pixels j..j+7:
    //   best of the two = combine(run8(j..j+7), better of the end pixels p[j-1], p[j+8])
    // so only the 8 odd-start 8-runs are needed (index m <-> j = 2m+1), built by doubling.
    s16x2 lo2[8], hi2[8], lo4[8], hi4[8];
#pragma unroll
    for (int m = 0; m < 8; m++) { lo2[m] = pkmin(p[2 * m + 1], p[(2 * m + 2) & 15]); hi2[m] = pkmax(p[2 * m + 1], p[(2 * m + 2) & 15]); }
#pragma unroll
    for (int m = 0; m < 8; m++) { lo4[m] = pkmin(lo2[m], lo2[(m + 1) & 7]); hi4[m] = pkmax(hi2[m], hi2[(m + 1) & 7]); }
    // per odd start m: the darkest pixel of the better of its two 9-arcs / the brightest one (three-input forms: one instruction each)
    s16x2 xl[8], xh[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const s16x2 e0 = p[2 * m], e1 = p[(2 * m + 9) & 15];
        xl[m] = pkmin3(lo4[m], lo4[(m + 2) & 7], pkmax(e0, e1));
        xh[m] = pkmax3(hi4[m], hi4[(m + 2) & 7], pkmin(e0, e1));
    }
    // Lo = brightest "darkest pixel of an arc", Hi = darkest "brightest pixel of an arc"
    const s16x2 Lo = pkmax(pkmax3(pkmax3(xl[0], xl[1], xl[2]), xl[3], xl[4]), pkmax3(xl[5], xl[6], xl[7]));
    const s16x2 Hi = pkmin(pkmin3(pkmin3(xh[0], xh[1], xh[2]), xh[3], xh[4]), pkmin3(xh[5], xh[6], xh[7]));
    const s16x2 v = as_s16x2(q[0]);
    return pkmax(v - Hi, Lo - v);
}

#define FC_ID(by, bp) (((by) << 5) | (bp))      // band pixel pair id: row | pair (pairs per row <= 32)

// Scalar description of one cell (ORBextractor.cc:783-806), everything a wave needs in SGPRs
struct FcCell {
    int frame, lvl, ci, cj;             // cj, ci: cell column / row inside the level
    int dw, dh;                         // detection band (dw <= 0: the reference skips the cell)
    int ipitch;                         // image row pitch
    const uint8_t *src;                 // pixel (ini_x - 1, ini_y) of the level image: first byte staged
    int ncols, nrows, wcell, hcell, cell_base, cell_cap, tpr, rpt;
};
#define FC_SGPR(x) __builtin_amdgcn_readfirstlane(x)
__device__ __forceinline__ void fc_cell_geom(const FastParams &P, FcCell &C)
{
    const FcLevel &L = P.lv[C.lvl];
    C.ncols = FC_SGPR(L.ncols); C.nrows = FC_SGPR(L.nrows); C.wcell = FC_SGPR(L.wcell); C.hcell = FC_SGPR(L.hcell);
    C.cell_base = FC_SGPR(L.cell_base); C.cell_cap = FC_SGPR(L.cell_cap); C.tpr = FC_SGPR(L.tpr); C.rpt = FC_SGPR(L.rpt);
    C.ipitch = FC_SGPR(L.img_pitch) & 0xFFFF;
    const int max_bx = FC_SGPR(L.max_bx), max_by = FC_SGPR(L.max_by);
    const int ini_y = ORB_MINB + C.ci * C.hcell, ini_x = ORB_MINB + C.cj * C.wcell;
    int max_y = ini_y + C.hcell + 6, max_x = ini_x + C.wcell + 6;
    C.src = L.img + (size_t)C.frame * L.frame_stride + (size_t)(ini_y * C.ipitch + ini_x - 1);
    C.dw = 0; C.dh = 0;
    if (ini_y >= max_by - 3 || ini_x >= max_bx - 6) return;               // ORBextractor.cc:788,797
    if (max_y > max_by) max_y = max_by;
    if (max_x > max_bx) max_x = max_bx;
    C.dw = max_x - ini_x - 6; C.dh = max_y - ini_y - 6;
    if (C.dw <= 0 || C.dh <= 0) { C.dw = 0; C.dh = 0; }
}
struct __attribute__((packed, aligned(1))) u32x4_unaligned { uint32_t x, y, z, w; };
// Staging loads of a cell: its sub-image rows as NCH 16-byte chunks each (chunk c = pixels ini_x - 1 + 16c .. + 15 = pairs 8c .. 8c+7),
// flattened (row, chunk) tasks, NLD per lane, branch-free (surplus lanes repeat the last task), all in flight together.  A chunk may
// run past the sub-image: those bytes are never used and lie inside the frame (the sub-image ends >= 16 rows above the image's end).
template <int NCH, int NLD, typename CELL>
__device__ __forceinline__ void fc_issue_loads(const CELL &C, int lane, uint4 (&ld)[NLD])
{
    const int last = max((C.dh + 6) * NCH - 1, 0);
#pragma unroll
    for (int t = 0; t < NLD; t++) {
        const int i = min(lane + 64 * t, last);
        const int r = (int)(__umul24((uint32_t)i, 65536u / NCH + 1) >> 16), c = i - r * NCH;      // i / NCH for i < 1024
        const u32x4_unaligned v = *reinterpret_cast<const u32x4_unaligned *>(C.src + (uint32_t)(__umul24((uint32_t)r, (uint32_t)C.ipitch) + 16u * (uint32_t)c));
        ld[t] = make_uint4(v.x, v.y, v.z, v.w);
    }
}

#ifdef FC_PROF
// debug build only (make EXTRA=-DFC_PROF): wave-cycles of k_fast_cells by phase, summed over all waves -- stage (load wait, LDS writes, next cell's
// geometry and loads) / sync / necessary test / arc scores / NMS + emission / tail; [6] cells, [7] whole kernel; [8] LDS writes of the stage,
// [9] next cell's geometry, [10] issue of its loads (the three parts of [0], which is then only the rest)
__device__ unsigned long long g_fc_prof[12];
extern "C" int orbhip_debug_fc_prof(unsigned long long *out8, int reset)
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_fc_prof), 96) != hipSuccess) return -1;
    if (reset) { unsigned long long z[12] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_fc_prof), z, 96) != hipSuccess) return -1; }
    return 0;
}
#define FC_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); fc_acc[i] += t_ - fc_prev; fc_prev = t_; } while (0)
#else
#define FC_T(i) do { } while (0)
#endif
// One wave per workgroup, persistent over a contiguous range of cells (frame-major, then level, then row-major): the next
// cell's staging loads are in flight while the current cell is processed.  LDS (private to the wave, no barriers): pair tile
// PT [rows][PITCH = 8 NCH], score tile SC [srows + 2][SPITCH] (one guard row / pair all around), queue Q.
template <int NCH, int SPITCH, int NLD>
__global__ __launch_bounds__(64) void k_fast_cells(FastParams P)
{
    constexpr int PITCH = 8 * NCH;
    extern __shared__ __attribute__((aligned(16))) uint32_t fc_lds[];
    const int lane = threadIdx.x;
    uint32_t *PT = fc_lds;
    uint16_t *SC = reinterpret_cast<uint16_t *>(PT + P.rows * PITCH);          // scores, one u16 per pixel pair: left | right << 8 (cornerScore <= 254)
    uint16_t *Q = SC + (P.srows + 2) * SPITCH;
    // this wave's cells [g0, g1) of batch * cells_per_frame
    const unsigned lid = xcd_logical_id(blockIdx.x, gridDim.x);
    // the cells of levels [lvl_lo, lvl_hi) are one contiguous range of a frame's cell array
    const int lvl_lo = FC_SGPR(P.lvl_lo), lvl_hi = FC_SGPR(P.lvl_hi);
    const int cell_lo = FC_SGPR(P.lv[lvl_lo].cell_base);
    const int cps = (lvl_hi < P.nlevels ? FC_SGPR(P.lv[lvl_hi].cell_base) : FC_SGPR(P.cells_per_frame)) - cell_lo;
    const long total_cells = (long)P.batch * cps;
    // Work is dealt in CHUNKS of consecutive cells, chunk c to wave c mod #waves: cells differ a lot in cost (texture, clipped border cells,
    // the small top levels), and contiguous equal-count ranges per wave left the kernel waiting for its slowest waves -- its time did not
    // move between 8 and 21 waves per CU.  Neighbouring chunks still run on neighbouring waves of one XCD (shared aprons hit its L2).
    const int chunk = FC_SGPR(P.chunk);
    const long nchunks = (total_cells + chunk - 1) / chunk;
    if ((long)lid >= nchunks) return;
    const int nlevels = lvl_hi, ini_th = FC_SGPR(P.ini_th), min_th = FC_SGPR(P.min_th), cpf = FC_SGPR(P.cells_per_frame);
    auto decode = [&](long gi, FcCell &X) {
        X.frame = (int)(gi / cps);
        const int cell = cell_lo + (int)(gi - (long)X.frame * cps);
        X.lvl = lvl_lo;
        for (int l = lvl_lo + 1; l < nlevels; l++) if (cell >= P.lv[l].cell_base) X.lvl = l;
        const int c = cell - P.lv[X.lvl].cell_base;
        X.ci = c / P.lv[X.lvl].ncols; X.cj = c - X.ci * P.lv[X.lvl].ncols;
        X.frame = FC_SGPR(X.frame); X.lvl = FC_SGPR(X.lvl); X.ci = FC_SGPR(X.ci); X.cj = FC_SGPR(X.cj);
        fc_cell_geom(P, X);
    };
    long ch = lid;                                                  // current chunk
    long g = ch * chunk, g_end = min(g + chunk, total_cells);       // current cell, end of the current chunk
    FcCell C;
    decode(g, C);
    uint4 ld[NLD];
    fc_issue_loads<NCH, NLD>(C, lane, ld);
    const int lo_th = min(ini_th, min_th);
#define FC_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); \
                            __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#ifdef FC_PROF
    unsigned long long fc_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, fc_prev = __builtin_readcyclecounter();
    const unsigned long long fc_t0 = fc_prev;
#endif
    for (bool more = true; more;) {
        const int dw = C.dw, dh = C.dh;
        const int npb = (dw + 1) >> 1;                             // band pixel pairs per row
        const int cell = C.cell_base + C.ci * C.ncols + C.cj;
        uint32_t *cnt_out = P.cell_count + (size_t)C.frame * cpf + cell;
        uint32_t *list = P.cell_list + (size_t)C.frame * P.cell_list_frame_stride + (size_t)cell * C.cell_cap;
        const int key_x0 = 3 + C.cj * C.wcell, key_y0 = 3 + C.ci * C.hcell;
        const int cell_cap = C.cell_cap, tpr = C.tpr, rptT = C.rpt;
#ifdef FC_PROF
        __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): the wait for this cell's staging loads on its own counter
        FC_T(11);
#endif
        // ---- (1) stage: PT[r][p] = pixels (ini_x + 2p - 1, ini_x + 2p) of sub-image row r as u16 | u16 << 16
        if (dw > 0) {
            const int last = (dh + 6) * NCH - 1;
#pragma unroll
            for (int t = 0; t < NLD; t++) {
                const int i = min(lane + 64 * t, last);
                const int r = (int)(__umul24((uint32_t)i, 65536u / NCH + 1) >> 16), c = i - r * NCH;
                uint4 a, b;
                a.x = __builtin_amdgcn_perm(0u, ld[t].x, 0x0c010c00u); a.y = __builtin_amdgcn_perm(0u, ld[t].x, 0x0c030c02u);
                a.z = __builtin_amdgcn_perm(0u, ld[t].y, 0x0c010c00u); a.w = __builtin_amdgcn_perm(0u, ld[t].y, 0x0c030c02u);
                b.x = __builtin_amdgcn_perm(0u, ld[t].z, 0x0c010c00u); b.y = __builtin_amdgcn_perm(0u, ld[t].z, 0x0c030c02u);
                b.z = __builtin_amdgcn_perm(0u, ld[t].w, 0x0c010c00u); b.w = __builtin_amdgcn_perm(0u, ld[t].w, 0x0c030c02u);
                uint4 *dst = reinterpret_cast<uint4 *>(&PT[r * PITCH + 8 * c]);
                dst[0] = a; dst[1] = b;
            }
            // the score tile starts at zero (guards included): a pair that is never scored counts as 0 in the NMS
            uint4 *scz = reinterpret_cast<uint4 *>(SC);
            for (int i = lane; i < ((dh + 2) * SPITCH + 7) / 8; i += 64) scz[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        FC_T(8);
        // ---- next cell (the next one of the chunk, or the first one of this wave's next chunk): its loads stay in flight while this one is processed
        FcCell N = C;
        g++;
        if (g < g_end) {
            N.cj++;
            if (N.cj == C.ncols) { N.cj = 0; N.ci++; }
            if (N.ci == C.nrows) { N.ci = 0; N.lvl++; }
            if (N.lvl == nlevels) { N.lvl = lvl_lo; N.frame++; }
            fc_cell_geom(P, N);
        } else {
            ch += gridDim.x;
            more = ch < nchunks;
            if (more) { g = ch * chunk; g_end = min(g + chunk, total_cells); decode(g, N); }
        }
        FC_T(9);
        if (more) fc_issue_loads<NCH, NLD>(N, lane, ld);
        FC_T(10);
        if (dw <= 0) { if (lane == 0) *cnt_out = 0; C = N; continue; }
        FC_T(0);
        FC_WAVE_SYNC();           // (also: the previous cell's LDS reads are done before this cell's writes -- same wave, program order)
        // Lane layout of the necessary test from the level's nominal cell width (clipped border cells leave lanes idle): tasks per row tpr,
        // rows per trip rpt, two adjacent pairs per lane
        const int tr0 = (int)(__umul24((uint32_t)lane, P.div_magic[tpr]) >> 16), tg = lane - tr0 * tpr;
        const bool lane_rows = tr0 < rptT && 2 * tg < npb, pair1 = 2 * tg + 1 < npb;
        int total = 0;
        FC_T(1);
        for (int pass = 0; pass < 2; pass++) {
            // ---- (2) necessary test on every band pixel pair at this pass's threshold, rows in order (queue order = cv::FAST order).  No
            // per-pair state is kept: the rare second pass (nothing at iniThFAST, ORBextractor.cc:825-828) runs the test again at minThFAST
            // and scores everything it lets through (the pairs of the first pass among them: same scores again)
            const int th = pass == 0 ? ini_th : min_th;
            const s16x2 thP = (s16x2){(short)(th + 1), (short)(th + 1)};
            int nq1 = 0;                                               // entries of Q (wave-uniform)
            for (int y0 = 0; y0 < dh; y0 += rptT) {
                const int by = y0 + tr0;
                bool f0 = false, f1 = false;                           // pair 0 / 1 may hold a corner at th
                if (lane_rows && by < dh) {
                    const uint32_t *q = &PT[(by + 3) * PITCH + 2 * tg + 2];
                    // M - (th + 1) >= 0 in a half <=> its sign bit is clear (|M| <= 255: no overflow).  An odd band's last lane evaluates one pair
                    // beyond the band: its flag is dropped here
                    const uint32_t SIGN = 0x80008000u;
                    const s16x2 M0 = fc_compass<PITCH>(q), M1 = fc_compass<PITCH>(q + 1);      // straight-line: the LDS reads of both pairs go out together
                    f0 = (as_u32(M0 - thP) & SIGN) != SIGN;
                    f1 = ((as_u32(M1 - thP) & SIGN) != SIGN) & pair1;
                }
                const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1);      // (outside the branch: the counts below must stay wave-uniform)
                const int pos = nq1 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
                                (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
                if (f0) Q[pos] = (uint16_t)FC_ID(by, 2 * tg);
                if (f1) Q[pos + (f0 ? 1 : 0)] = (uint16_t)FC_ID(by, 2 * tg + 1);
                nq1 += __popcll(b0) + __popcll(b1);
            }
            FC_WAVE_SYNC();
            FC_T(2);
            // ---- (3) dense arc score, one pair per lane
            for (int i0 = 0; i0 < nq1; i0 += 64) {
                const int i = min(i0 + lane, nq1 - 1);
                const int id = Q[i], by = id >> 5, bp = id & 31;
                const s16x2 S = fc_arc_score<PITCH>(&PT[(by + 3) * PITCH + bp + 2]);
                if (i0 + lane < nq1) {
                    const uint32_t lo = S.x > lo_th ? (uint32_t)(S.x - 1) : 0u;
                    const uint32_t hi = (S.y > lo_th && 2 * bp + 1 < dw) ? (uint32_t)(S.y - 1) : 0u;
                    SC[(by + 1) * SPITCH + bp + 1] = (uint16_t)(lo | (hi << 8));
                }
            }
            FC_WAVE_SYNC();
            FC_T(3);
            // ---- (4) 3x3 strict NMS on raw scores + threshold gate + ordered emission
            for (int i0 = 0; i0 < nq1; i0 += 64) {
                const int i = i0 + lane;
                uint32_t keep = 0, sc = 0;
                int by = 0, bp = 0;
                if (i < nq1) {
                    const int id = Q[i];
                    by = id >> 5; bp = id & 31;
                    const uint16_t *q = &SC[(by + 1) * SPITCH + bp + 1];
#define FC_U(v) as_s16x2(__builtin_amdgcn_perm(0u, (uint32_t)(v), 0x0c010c00u))                          /* left | right << 8  ->  packed halves */
                    sc = q[0];
                    const s16x2 V = pkmax(FC_U(q[-SPITCH]), FC_U(q[SPITCH]));                                  // above / below each pixel
                    const s16x2 Lc = pkmax3(FC_U(q[-SPITCH - 1]), FC_U(q[-1]), FC_U(q[SPITCH - 1]));     // .y: column left of the pair
                    const s16x2 Rc = pkmax3(FC_U(q[-SPITCH + 1]), FC_U(q[1]), FC_U(q[SPITCH + 1]));      // .x: column right of the pair
#undef FC_U
                    const int sl = (int)(sc & 0xFFu), sr = (int)(sc >> 8);
                    const int nbl = max(max((int)V.x, (int)V.y), max((int)Lc.y, sr));                // L: above, below, the R column (3 rows), left column
                    const int nbr = max(max((int)V.x, (int)V.y), max((int)Rc.x, sl));
                    if (sl >= th && sl > nbl) keep |= 1u;
                    if (sr >= th && sr > nbr) keep |= 2u;
                }
                const int mine = __popc(keep);
                const int inc = wave_incl_scan(mine);
                int offs = total + inc - mine;
                if (keep & 1u) { if (offs < cell_cap) list[offs] = ORB_PACK_KEY(2 * bp + key_x0, by + key_y0, sc & 0xFFu); offs++; }
                if (keep & 2u) { if (offs < cell_cap) list[offs] = ORB_PACK_KEY(2 * bp + 1 + key_x0, by + key_y0, (sc >> 8) & 0xFFu); }
                total += __builtin_amdgcn_readlane(inc, 63);
            }
            FC_T(4);
            if (total > 0 || min_th == ini_th) break;              // vKeysCell not empty at iniThFAST: done
            FC_WAVE_SYNC();                                        // the second pass rewrites Q and SC
        }
        if (total > cell_cap) { if (lane == 0) atomicExch(P.status, ORBHIP_E_CAPACITY); total = cell_cap; }
        if (lane == 0) *cnt_out = (uint32_t)total;
        C = N;
        FC_T(5);
#ifdef FC_PROF
        fc_acc[6]++;
#endif
    }
#ifdef FC_PROF
    fc_acc[7] = __builtin_readcyclecounter() - fc_t0;
    if (lane == 0) for (int i = 0; i < 12; i++) atomicAdd(&g_fc_prof[i], fc_acc[i]);
#endif
#undef FC_WAVE_SYNC
}

// ----------------------------------------------------------------------------------
// k_fast_runs (round 4): the same per-cell FAST semantics (ORBextractor.cc:783-854: every cell is FASTed alone at
// iniThFAST, again at minThFAST iff that left it empty; the 3-px ring of its sub-image is excluded, so NMS never sees a neighbour cell)
// with the cell no longer the unit of work.  A wave owns a RUN of up to two horizontally adjacent cells of one cell row -- one
// sub-image of (2 wCell + 6) x (hCell + 6) pixels, staged once (80-byte rows instead of two 41..48-byte ones: less apron, longer lines)
// -- and the cell enters only as
//   (a) a mask in the NMS: a neighbour pixel that lies in the other cell's band counts as 0,
//   (b) the per-cell "empty at iniThFAST -> again at minThFAST" decision (the second pass is restricted to the pixels of empty cells),
//   (c) the emission: per-cell ranks by ballot, per-cell lists and counts.
// The necessary (compass) test, which is most of the kernel, no longer gathers its 11 dwords per pixel pair from LDS: lane = pair COLUMN
// (32 columns x 2 half waves, the halves walking the upper / lower half of the rows), and a lane walks DOWN its column with a
// register ring of the last seven rows' three dwords (columns c-1, c, c+1), so a row step costs 3 LDS instructions (the new row's
// ds_read2 + ds_read, the centre row's two outer dwords) instead of 11 and no address arithmetic; the loop is unrolled by the ring
// period so that every ring access is a fixed register.  Queue (two, one per half wave: concatenated they are in cv::FAST's raster
// order), dense arc scoring and the score tile are k_fast_cells' -- the very same expressions, so every candidate list comes out
// byte-identical (tests/test_gpu_orb.py compares them with the oracle cell by cell).
// ----------------------------------------------------------------------------------
#define FR_PITCH 40            // pair-tile dwords per row: pairs -2 .. 37 of the run's sub-image
#define FR_SPITCH 40           // score-tile u16 per row: 32 pairs + one guard on either side
#define FR_NCH 5               // 16-byte chunks staged per row
struct FrRun {
    int frame, lvl, ci, cj, nc;         // first cell of the run, cells in it (1 or 2)
    int dw0, dw1, dh;                   // detection band widths of the two cells (0: the reference skips the cell), band height
    int ipitch; const uint8_t *src;     // pixel (ini_x - 1, ini_y) of the first cell: first byte staged
    int ncols, nrows, wcell, hcell, cell_base, cell_cap, rpc, rpr;
};
__device__ __forceinline__ void fr_run_geom(const FastParams &P, FrRun &C)
{
    const FcLevel &L = P.lv[C.lvl];
    C.ncols = FC_SGPR(L.ncols); C.nrows = FC_SGPR(L.nrows); C.wcell = FC_SGPR(L.wcell); C.hcell = FC_SGPR(L.hcell);
    C.cell_base = FC_SGPR(L.cell_base); C.cell_cap = FC_SGPR(L.cell_cap); C.rpc = FC_SGPR(L.rpc); C.rpr = FC_SGPR(L.rpr);
    C.ipitch = FC_SGPR(L.img_pitch) & 0xFFFF;
    C.nc = min(C.rpc, C.ncols - C.cj);
    const int max_bx = FC_SGPR(L.max_bx), max_by = FC_SGPR(L.max_by);
    const int ini_y = ORB_MINB + C.ci * C.hcell, ini_x = ORB_MINB + C.cj * C.wcell;
    C.src = L.img + (size_t)C.frame * L.frame_stride + (size_t)(ini_y * C.ipitch + ini_x - 1);
    C.dw0 = 0; C.dw1 = 0; C.dh = 0;
    if (ini_y >= max_by - 3) return;                                          // ORBextractor.cc:788
    C.dh = min(ini_y + C.hcell + 6, max_by) - ini_y - 6;
    if (C.dh <= 0) { C.dh = 0; return; }
    if (ini_x < max_bx - 6) C.dw0 = max(min(ini_x + C.wcell + 6, max_bx) - ini_x - 6, 0);     // :797-802
    const int ini_x1 = ini_x + C.wcell;
    if (C.nc > 1 && ini_x1 < max_bx - 6) C.dw1 = max(min(ini_x1 + C.wcell + 6, max_bx) - ini_x1 - 6, 0);
    if (C.dw0 == 0) C.dw1 = 0;
}

// FR_NLD: staging loads per lane (rows x 5 chunks <= 64 FR_NLD): 4 for cells of up to 45 rows, 6 up to 70
template <int FR_NLD>
__global__ __launch_bounds__(64) void k_fast_runs(FastParams P)
{
    constexpr int PITCH = FR_PITCH, SPITCH = FR_SPITCH;
    extern __shared__ __attribute__((aligned(16))) uint32_t fc_lds[];
    const int lane = threadIdx.x;
    uint32_t *PT = fc_lds;
    uint16_t *SC = reinterpret_cast<uint16_t *>(PT + P.run_rows * PITCH);      // scores, one u16 per pixel pair: left | right << 8
    uint16_t *Q = SC + (P.srows + 2) * SPITCH;                                  // [0, q0): upper half of the rows, [q0, ..): lower half
    const int q0cap = FC_SGPR(P.run_q0);
    const unsigned lid = xcd_logical_id(blockIdx.x, gridDim.x);
    const int lvl_lo = FC_SGPR(P.lvl_lo), lvl_hi = FC_SGPR(P.lvl_hi);
    const int run_lo = FC_SGPR(P.lv[lvl_lo].run_base);
    const int rps = (lvl_hi < P.nlevels ? FC_SGPR(P.lv[lvl_hi].run_base) : FC_SGPR(P.runs_per_frame)) - run_lo;
    const long total_runs = (long)P.batch * rps;
    const int chunk = FC_SGPR(P.chunk);
    const long nchunks = (total_runs + chunk - 1) / chunk;
    if ((long)lid >= nchunks) return;
    const int nlevels = lvl_hi, ini_th = FC_SGPR(P.ini_th), min_th = FC_SGPR(P.min_th), cpf = FC_SGPR(P.cells_per_frame);
    auto decode = [&](long gi, FrRun &X) {
        X.frame = (int)(gi / rps);
        const int run = run_lo + (int)(gi - (long)X.frame * rps);
        X.lvl = lvl_lo;
        for (int l = lvl_lo + 1; l < nlevels; l++) if (run >= P.lv[l].run_base) X.lvl = l;
        const int r = run - P.lv[X.lvl].run_base;
        X.ci = r / P.lv[X.lvl].rpr; X.cj = (r - X.ci * P.lv[X.lvl].rpr) * P.lv[X.lvl].rpc;
        X.frame = FC_SGPR(X.frame); X.lvl = FC_SGPR(X.lvl); X.ci = FC_SGPR(X.ci); X.cj = FC_SGPR(X.cj);
        fr_run_geom(P, X);
    };
    long ch = lid;
    long g = ch * chunk, g_end = min(g + chunk, total_runs);
    FrRun C;
    decode(g, C);
    uint4 ld[FR_NLD];
    fc_issue_loads<FR_NCH, FR_NLD>(C, lane, ld);
    const int lo_th = min(ini_th, min_th);
    const int bp = lane & 31, half = lane >> 5;                                // this lane's pair column and half wave
#define FC_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); \
                            __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
    for (bool more = true; more;) {
        const int dw0 = C.dw0, dw1 = C.dw1, dw = dw0 + dw1, dh = C.dh;
        const int npb = (dw + 1) >> 1;                             // band pixel pairs per row of the run
        const int cell = C.cell_base + C.ci * C.ncols + C.cj;
        uint32_t *cnt_out = P.cell_count + (size_t)C.frame * cpf + cell;
        uint32_t *list = P.cell_list + (size_t)C.frame * P.cell_list_frame_stride + (size_t)cell * C.cell_cap;
        const int key_x0 = 3 + C.cj * C.wcell, key_y0 = 3 + C.ci * C.hcell;
        const int cell_cap = C.cell_cap, nc = C.nc;
        // ---- (1) stage: PT[r][p] = pixels (ini_x + 2p - 1, ini_x + 2p) of sub-image row r as u16 | u16 << 16
        if (dw > 0) {
            const int last = (dh + 6) * FR_NCH - 1;
#pragma unroll
            for (int t = 0; t < FR_NLD; t++) {
                const int i = min(lane + 64 * t, last);
                const int r = (int)(__umul24((uint32_t)i, 65536u / FR_NCH + 1) >> 16), c = i - r * FR_NCH;
                uint4 a, b;
                a.x = __builtin_amdgcn_perm(0u, ld[t].x, 0x0c010c00u); a.y = __builtin_amdgcn_perm(0u, ld[t].x, 0x0c030c02u);
                a.z = __builtin_amdgcn_perm(0u, ld[t].y, 0x0c010c00u); a.w = __builtin_amdgcn_perm(0u, ld[t].y, 0x0c030c02u);
                b.x = __builtin_amdgcn_perm(0u, ld[t].z, 0x0c010c00u); b.y = __builtin_amdgcn_perm(0u, ld[t].z, 0x0c030c02u);
                b.z = __builtin_amdgcn_perm(0u, ld[t].w, 0x0c010c00u); b.w = __builtin_amdgcn_perm(0u, ld[t].w, 0x0c030c02u);
                uint4 *dst = reinterpret_cast<uint4 *>(&PT[r * PITCH + 8 * c]);
                dst[0] = a; dst[1] = b;
            }
            uint4 *scz = reinterpret_cast<uint4 *>(SC);             // the score tile starts at zero (guards included)
            for (int i = lane; i < ((dh + 2) * SPITCH + 7) / 8; i += 64) scz[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        // ---- next run: its loads stay in flight while this one is processed
        FrRun N = C;
        g++;
        if (g < g_end) {
            N.cj += C.rpc;
            if (N.cj >= C.ncols) { N.cj = 0; N.ci++; }
            if (N.ci == C.nrows) { N.ci = 0; N.lvl++; }
            if (N.lvl == nlevels) { N.lvl = lvl_lo; N.frame++; }
            fr_run_geom(P, N);
        } else {
            ch += gridDim.x;
            more = ch < nchunks;
            if (more) { g = ch * chunk; g_end = min(g + chunk, total_runs); decode(g, N); }
        }
        if (more) fc_issue_loads<FR_NCH, FR_NLD>(N, lane, ld);
        if (dw <= 0) { if (lane < nc) cnt_out[lane] = 0; C = N; continue; }
        FC_WAVE_SYNC();
        // the cell of a band pixel: x < dw0 -> the first one.  B = first column of the second cell (or beyond every pixel)
        const int B = dw1 > 0 ? dw0 : (1 << 20);
        const int hh = (dh + 1) >> 1;                               // rows per half wave
        const int rb = half * hh;                                   // first band row of this lane's half
        const int rows_mine = min(hh, dh - rb);                     // (may be < hh for the lower half of an odd band)
        int tot0 = 0, tot1 = 0;                                     // keypoints emitted per cell (uniform)
        uint32_t empty = 0;                                         // second pass: bit c = cell c came out empty at iniThFAST
        for (int pass = 0; pass < 2; pass++) {
            const int th = pass == 0 ? ini_th : min_th;
            const s16x2 thP = (s16x2){(short)(th + 1), (short)(th + 1)};
            // a pair takes part if it lies in the band and, in the second pass, if one of its pixels belongs to an empty cell
            bool col_on = bp < npb;
            if (pass) col_on = col_on && ((((2 * bp >= B) ? 2u : 1u) | ((2 * bp + 1 >= B) ? 2u : 1u)) & empty) != 0;
            // ---- (2) necessary test, every lane walking down its pair column, seven rows per trip: the 13 tile rows a trip touches sit in
            // registers (six carried over from the previous trip), the seven M values are computed as seven independent chains (the
            // packed 16-bit min / max ops need a wait state between dependent ones: alone, a chain was half s_nop), then flagged and queued
            int nq0 = 0, nq1 = 0;
            {
                const uint32_t *col = PT + rb * PITCH + bp + 1;     // dwords [0], [1], [2] of a row = columns c - 1, c, c + 1 (c = bp + 2)
                const int rows_on = col_on ? rows_mine : 0;         // rows of this lane that may flag
                uint32_t rm[13], rc[13], rp[13];
#pragma unroll
                for (int i = 0; i < 6; i++) { rm[i] = col[i * PITCH]; rc[i] = col[i * PITCH + 1]; rp[i] = col[i * PITCH + 2]; }
                const uint32_t SIGN = 0x80008000u;
                for (int y0 = 0; y0 < hh; y0 += 7) {
                    const uint32_t *cy = col + y0 * PITCH;
                    uint32_t cm2[7], cp2[7];
#pragma unroll
                    for (int i = 0; i < 7; i++) {                   // (rows past the band: inside the tile's padding, never flagged)
                        rm[6 + i] = cy[(6 + i) * PITCH]; rc[6 + i] = cy[(6 + i) * PITCH + 1]; rp[6 + i] = cy[(6 + i) * PITCH + 2];
                        cm2[i] = cy[(3 + i) * PITCH - 1]; cp2[i] = cy[(3 + i) * PITCH + 3];
                    }
                    bool f[7];
#pragma unroll
                    for (int i = 0; i < 7; i++) {
                        // fc_compass with q[0] = rc[i + 3]: n = q[3 P], s = q[-3 P], e = (q[1], q[2]), w = (q[-2], q[-1]), se = q[2 P + 1],
                        // nw = q[-2 P - 1], ne = q[-2 P + 1], sw = q[2 P - 1]
                        const s16x2 v = as_s16x2(rc[i + 3]);
                        const s16x2 n = as_s16x2(rc[i + 6]), so = as_s16x2(rc[i]);
                        const s16x2 e = pair_odd(cp2[i], rp[i + 3]), w = pair_odd(rm[i + 3], cm2[i]);
                        const s16x2 se = as_s16x2(rp[i + 5]), nw = as_s16x2(rm[i + 1]);
                        const s16x2 ne = as_s16x2(rp[i + 1]), sw = as_s16x2(rm[i + 5]);
                        const s16x2 lo = pkmax(pkmax3(pkmin(n, so), pkmin(e, w), pkmin(se, nw)), pkmin(ne, sw));
                        const s16x2 hi = pkmin(pkmin3(pkmax(n, so), pkmax(e, w), pkmax(se, nw)), pkmax(ne, sw));
                        const s16x2 M = pkmax(v - lo, hi - v);
                        f[i] = y0 + i < rows_on && (as_u32(M - thP) & SIGN) != SIGN;
                    }
#pragma unroll
                    for (int i = 0; i < 7; i++) {
                        const unsigned long long b = __ballot(f[i]);
                        const uint32_t blo = (uint32_t)b, bhi = (uint32_t)(b >> 32);
                        // entries of the upper half go to Q[nq0 ..), of the lower half to Q[q0cap + nq1 ..): one expression for both
                        const int m = (int)__builtin_amdgcn_mbcnt_hi(bhi, __builtin_amdgcn_mbcnt_lo(blo, 0u));
                        const int pos = m + (half ? q0cap + nq1 - __popc(blo) : nq0);
                        if (f[i]) Q[pos] = (uint16_t)FC_ID(rb + y0 + i, bp);
                        nq0 += __popc(blo); nq1 += __popc(bhi);
                    }
#pragma unroll
                    for (int i = 0; i < 6; i++) { rm[i] = rm[7 + i]; rc[i] = rc[7 + i]; rp[i] = rp[7 + i]; }
                }
            }
            const int nq = nq0 + nq1;
            FC_WAVE_SYNC();
            // ---- (3) dense arc score, one pair per lane (queue entry i: the upper half's entries first)
            for (int i0 = 0; i0 < nq; i0 += 64) {
                const int i = min(i0 + lane, nq - 1);
                const int id = Q[i < nq0 ? i : q0cap + i - nq0], by = id >> 5, qp = id & 31;
                const s16x2 S = fc_arc_score<PITCH>(&PT[(by + 3) * PITCH + qp + 2]);
                if (i0 + lane < nq) {
                    const uint32_t lo = S.x > lo_th ? (uint32_t)(S.x - 1) : 0u;
                    const uint32_t hi = (S.y > lo_th && 2 * qp + 1 < dw) ? (uint32_t)(S.y - 1) : 0u;
                    SC[(by + 1) * SPITCH + qp + 1] = (uint16_t)(lo | (hi << 8));
                }
            }
            FC_WAVE_SYNC();
            // ---- (4) 3x3 strict NMS on raw scores (neighbours in the other cell's band count as 0) + threshold gate + per-cell ordered emission
            for (int i0 = 0; i0 < nq; i0 += 64) {
                const int i = i0 + lane;
                uint32_t keep = 0, sc = 0;
                int by = 0, qp = 0;
                if (i < nq) {
                    const int id = Q[i < nq0 ? i : q0cap + i - nq0];
                    by = id >> 5; qp = id & 31;
                    const uint16_t *q = &SC[(by + 1) * SPITCH + qp + 1];
#define FC_U(v) as_s16x2(__builtin_amdgcn_perm(0u, (uint32_t)(v), 0x0c010c00u))                          /* left | right << 8  ->  packed halves */
                    sc = q[0];
                    const s16x2 V = pkmax(FC_U(q[-SPITCH]), FC_U(q[SPITCH]));                                  // above / below each pixel
                    const s16x2 Lc = pkmax3(FC_U(q[-SPITCH - 1]), FC_U(q[-1]), FC_U(q[SPITCH - 1]));     // .y: column left of the pair
                    const s16x2 Rc = pkmax3(FC_U(q[-SPITCH + 1]), FC_U(q[1]), FC_U(q[SPITCH + 1]));      // .x: column right of the pair
#undef FC_U
                    const int xl = 2 * qp, xr = xl + 1;
                    const int sl = (int)(sc & 0xFFu), sr = (int)(sc >> 8);
                    // L: above / below, the column to its left (not if L is the second cell's first column), the R column (not across the boundary)
                    const int nbl = max(max((int)V.x, xl != B ? (int)Lc.y : 0), xr != B ? max((int)V.y, sr) : 0);
                    const int nbr = max(max((int)V.y, xr + 1 != B ? (int)Rc.x : 0), xr != B ? max((int)V.x, sl) : 0);
                    const bool onL = !pass || ((xl >= B ? 2u : 1u) & empty), onR = !pass || ((xr >= B ? 2u : 1u) & empty);
                    if (sl >= th && sl > nbl && onL) keep |= 1u;
                    if (sr >= th && sr > nbr && onR) keep |= 2u;
                }
                const int xl = 2 * qp;
                const bool kL = keep & 1u, kR = keep & 2u, cL = xl >= B, cR = xl + 1 >= B;      // kept flags, cell (0 / 1) of each pixel
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const bool mL = kL && (int)cL == c, mR = kR && (int)cR == c;
                    const unsigned long long bL = __ballot(mL), bR = __ballot(mR);
                    const int before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bL >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bL, 0u)) +
                                       (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bR >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bR, 0u));
                    const int base = c ? tot1 : tot0;
                    uint32_t *lst = list + (size_t)c * cell_cap;
                    if (mL) { const int o = base + before; if (o < cell_cap) lst[o] = ORB_PACK_KEY(xl + key_x0, by + key_y0, sc & 0xFFu); }
                    if (mR) { const int o = base + before + (mL ? 1 : 0); if (o < cell_cap) lst[o] = ORB_PACK_KEY(xl + 1 + key_x0, by + key_y0, (sc >> 8) & 0xFFu); }
                    const int add = __popcll(bL) + __popcll(bR);
                    if (c) tot1 += add; else tot0 += add;
                }
            }
            if (pass || min_th == ini_th) break;
            empty = (tot0 == 0 && dw0 > 0 ? 1u : 0u) | (tot1 == 0 && dw1 > 0 ? 2u : 0u);        // vKeysCell.empty() at iniThFAST (ORBextractor.cc:825-828)
            if (!empty) break;
            FC_WAVE_SYNC();                                        // the second pass rewrites Q and SC
        }
        if (tot0 > cell_cap || tot1 > cell_cap) { if (lane == 0) atomicExch(P.status, ORBHIP_E_CAPACITY); tot0 = min(tot0, cell_cap); tot1 = min(tot1, cell_cap); }
        if (lane == 0) { cnt_out[0] = (uint32_t)tot0; if (nc > 1) cnt_out[1] = (uint32_t)tot1; }
        C = N;
    }
#undef FC_WAVE_SYNC
}
const void *orb_fast_runs_func(int nld) { return nld == 4 ? reinterpret_cast<const void *>(k_fast_runs<4>) : reinterpret_cast<const void *>(k_fast_runs<6>); }

const void *orb_fast_cells_func(int small);
void orb_launch_fast_cells(const FastParams &F_, hipStream_t s, int max_per_cu, int lvl_lo, int lvl_hi)
{
    FastParams F = F_;
    F.lvl_lo = lvl_lo; F.lvl_hi = lvl_hi < 0 ? F.nlevels : lvl_hi;
    if (F.lvl_lo >= F.lvl_hi) return;
    const bool runs = F.use_runs != 0;
    // persistent single-wave workgroups: as many as the LDS lets a CU hold, a whole number per XCD
    const size_t lds = sizeof(uint32_t) * (size_t)(runs ? F.run_dw : F.wave_dw);
    // as many as the runtime says fit (LDS, wave slots), asked once per (device, kernel variant, LDS size).  The left and right images
    // of a stereo frame are extracted on two threads (Frame.cc:109-110) and a process may drive several GPUs: the table is guarded
    const int nld = (F.run_rows - 7) * FR_NCH <= 256 ? 4 : 6;
    const int v = runs ? (nld == 4 ? 2 : 3) : F.small_cells ? 1 : 0;
    int per_cu;
    {
        struct Occ { int dev, v, n; size_t lds; };
        static std::mutex occ_mu; static std::vector<Occ> occ_tab;
        int dev = 0; (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(occ_mu);
        const Occ *hit = nullptr;
        for (const Occ &o : occ_tab) if (o.dev == dev && o.v == v && o.lds == lds) { hit = &o; break; }
        if (!hit) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, v >= 2 ? orb_fast_runs_func(nld) : orb_fast_cells_func(v), 64, lds) != hipSuccess || n < 1) n = (int)((160 * 1024) / (lds + 512));
            occ_tab.push_back(Occ{dev, v, n < 1 ? 1 : n, lds}); hit = &occ_tab.back();
            if (getenv("ORBHIP_DEBUG_FAST")) fprintf(stderr, "[orbhip] %s: %zu B of LDS per wave, %d waves per CU resident\n", v >= 2 ? "k_fast_runs" : "k_fast_cells", lds, hit->n);
        }
        per_cu = hit->n;
    }
    if (max_per_cu > 0 && per_cu > max_per_cu) per_cu = max_per_cu;      // leave LDS and wave slots to a kernel running beside this one
    long nblocks = (long)(F.n_cus > 0 ? F.n_cus : 256) * per_cu;
    const long total = runs ? (long)F.batch * ((F.lvl_hi < F.nlevels ? F.lv[F.lvl_hi].run_base : F.runs_per_frame) - F.lv[F.lvl_lo].run_base)
                            : (long)F.batch * ((F.lvl_hi < F.nlevels ? F.lv[F.lvl_hi].cell_base : F.cells_per_frame) - F.lv[F.lvl_lo].cell_base);
    if (nblocks > total) nblocks = (total + 7) / 8 * 8;
    // chunks of up to 4 consecutive cells (8 runs): they share aprons and level parameters; at least ~8 chunks per wave so that the shares even out.
    // (Round 4: 16 -> 4 cells.  The waves of an XCD work on neighbouring chunks at the same time, so the seam and apron lines of the k-th cells
    // of neighbouring chunks are touched together, but the last cell of one chunk and the first of the next 15 cell times (~0.1 ms) apart --
    // long after the line left the 4 MB L2.  1024 VGA frames: step 3.59 -> 3.55 ms at 4, 3.68 at 32; 1 and 2 lose to the per-chunk decode.)
    static const int chunk_env = getenv("ORBHIP_TUNE_FAST_CHUNK") ? atoi(getenv("ORBHIP_TUNE_FAST_CHUNK")) : 0;
    const long fair = total / (8 * nblocks);
    const int cmax = runs ? 8 : 4;
    F.chunk = chunk_env > 0 ? chunk_env : (int)(fair < 1 ? 1 : fair > cmax ? cmax : fair);
    if (runs && nld == 4) hipLaunchKernelGGL(k_fast_runs<4>, dim3((unsigned)nblocks), dim3(64), lds, s, F);
    else if (runs) hipLaunchKernelGGL(k_fast_runs<6>, dim3((unsigned)nblocks), dim3(64), lds, s, F);
    else if (F.small_cells) hipLaunchKernelGGL((k_fast_cells<3, 24, 2>), dim3((unsigned)nblocks), dim3(64), lds, s, F);
    else hipLaunchKernelGGL((k_fast_cells<5, 32, 5>), dim3((unsigned)nblocks), dim3(64), lds, s, F);
}
const void *orb_fast_cells_func(int small) { return small ? reinterpret_cast<const void *>(k_fast_cells<3, 24, 2>) : reinterpret_cast<const void *>(k_fast_cells<5, 32, 5>); }

// ----------------------------------------------------------------------------------
// A5  DistributeOctTree (ORBextractor.cc:537-761), one 256-thread workgroup per
// (frame, level), node state resident in LDS.
//
// Reformulation (bit-identical to the oracle's list-based restatement):
//  * every node's vKeys is a subsequence of the level's candidate list in original
//    order (DivideNode pushes in order), so a per-key node id replaces the key vectors;
//  * the std::list order is an explicit position: a step that splits the set `proc`
//    in processing order o=0..m-1 yields  [children(o=m-1) n4..n1, ..., children(o=0)
//    n4..n1, surviving old nodes in old order]  (push_front semantics);
//  * full pass: proc = all nodes with >1 key, processing order = list order;
//  * final phase (ORBextractor.cc:665-735): candidates = children with >1 key created by
//    the previous step, processing order = (size desc, creation desc) == (size desc,
//    list position asc); stop after the first split that reaches N nodes.  The
//    reference breaks ties by node address (non-deterministic); oracle and
//    kernel use creation order -- documented deviation.
// ----------------------------------------------------------------------------------
struct OctLds {
    uint32_t *cc;        // [NC*4] child key counts
    int *ord;            // [NC] processing order of node p, or -1
    int *by_ord;         // [NC] node at order o
    int *excl;           // [NC] exclusive scan over order of nchild
    int *posbase;        // [NC] new position of the first (frontmost) child of processed p
    int *newpos;         // [NC] new position of a surviving node p
    uint32_t *best;      // [NC]
};

int orb_octree_nc(const OrbParams &P)
{
    int nc = 0;
    for (int l = 0; l < P.nlevels; l++) { const int a = P.lv[l].quota + 16, b = 4 * P.lv[l].n_ini + 4; nc = a > nc ? a : nc; nc = b > nc ? b : nc; }
    return nc;
}

size_t orb_octree_lds_bytes(int nc)
{
    return (size_t)nc * 4 * (6 + 4 + 6) + 64;
}

__device__ __forceinline__ int oct_quadrant(uint32_t key, uint32_t b0, uint32_t b1)
{
    // DivideNode (ORBextractor.cc:479-524): halfX = ceil((UR.x-UL.x)/2), children by < on x then y
    const int x0 = b0 & 0xFFFF, y0 = b0 >> 16, x1 = b1 & 0xFFFF, y1 = b1 >> 16;
    const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
    const int kx = ORB_KEY_X(key), ky = ORB_KEY_Y(key);
    return (kx < mx ? 0 : 1) + (ky < my ? 0 : 2);          // n1=0 n2=1 n3=2 n4=3
}

#define OCT_KR 8               // candidates per thread held in registers (x 256 threads)
#ifdef OCT_PROF
__device__ long long g_oct_prof[8];           // debug build only (EXTRA=-DOCT_PROF): cycles of gather / roots / subdivision / best / output+perm of workgroup 0, passes
#define OCT_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = clock64(); g_oct_prof[i] += t_ - t_prev; t_prev = t_; } } while (0)
extern "C" int orbhip_debug_oct_prof(long long *out8, int reset)
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_oct_prof), 64) != hipSuccess) return -1;
    if (reset) { long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_oct_prof), z, 64) != hipSuccess) return -1; }
    return 0;
}
#else
#define OCT_T(i) do { } while (0)
#endif
__global__ __launch_bounds__(256) void k_octree(OrbParams P)
{
    extern __shared__ uint32_t smem[];
    __shared__ int wsum[8];
    __shared__ int s_size, s_front, s_nexpand, s_rstar, s_T, s_nproc;
    const int tid = threadIdx.x;
#ifdef OCT_PROF
    long long t_prev = clock64();
#endif
    const int lvl = blockIdx.x / P.batch;              // level-major: big levels first
    const int frame = blockIdx.x - lvl * P.batch;
    const OrbLevel &L = P.lv[lvl];
    const int N = L.quota;
    const int NC = P.oct_nc;
    // double-buffered node arrays (by list position): box0 = UL.x|UL.y<<16, box1 = BR.x|BR.y<<16, cnt = #keys
    uint32_t *box0 = smem, *box1 = smem + NC, *cnt = smem + 2 * NC;
    uint32_t *nbox0 = smem + 3 * NC, *nbox1 = smem + 4 * NC, *ncnt = smem + 5 * NC;
    OctLds S;
    {
        uint32_t *p = smem + 6 * NC;
        S.cc = p; p += 4 * NC;
        S.ord = (int *)p; p += NC; S.by_ord = (int *)p; p += NC; S.excl = (int *)p; p += NC;
        S.posbase = (int *)p; p += NC; S.newpos = (int *)p; p += NC; S.best = p; p += NC;
    }
    // ---- gather candidates in reference order: cells row-major, list order inside a cell
    const uint32_t *ccount = P.cell_count + (size_t)frame * P.cells_per_frame + L.cell_base;
    const uint32_t *clist = P.cell_list + (size_t)frame * P.cell_list_frame_stride + (size_t)L.cell_base * L.cell_cap;
    uint32_t *keys = P.keys + (size_t)frame * P.keys_per_frame + L.key_base;
    uint16_t *node_of = P.node_of + (size_t)frame * P.keys_per_frame + L.key_base;
    const int ncells = L.ncols * L.nrows;
    uint32_t rk[OCT_KR]; int rn[OCT_KR];
    int running = 0;
    int32_t *count_out = P.lvl_count + frame * P.nlevels + lvl;
    if (ncells + 1 <= 16 * NC) {
        // cell offsets -> LDS (the node arrays are not in use yet), then ONE flat pass over the candidates: candidate k finds its cell by
        // binary search over the offsets and is fetched straight into its register slot (all loads of a thread independent and in
        // flight together; the per-cell copy loops this replaces were chains of dependent global round trips: 37 % of the kernel)
        int *coff = reinterpret_cast<int *>(smem);
        for (int c0 = 0; c0 < ncells; c0 += 256) {
            const int c = c0 + tid;
            const int n = c < ncells ? min((int)ccount[c], L.cell_cap) : 0;
            int total;
            const int off = running + block_excl_scan256(n, wsum, &total);
            if (c < ncells) coff[c] = off;
            running += total;
        }
        if (tid == 0) coff[ncells] = running;
        __syncthreads();
        if (running > L.key_cap) { if (tid == 0) atomicExch(P.status, ORBHIP_E_CAPACITY); running = L.key_cap; }
        const int K0 = running;
        auto fetch = [&](int k) {
            int lo = 0, hi = ncells;                                 // largest c with coff[c] <= k (empty cells share offsets: take the last)
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (coff[mid] <= k) lo = mid; else hi = mid; }
            return clist[(size_t)lo * L.cell_cap + (k - coff[lo])];
        };
#pragma unroll
        for (int i = 0; i < OCT_KR; i++) { const int k = tid + 256 * i; rk[i] = k < K0 ? fetch(k) : 0u; rn[i] = 0; }
#pragma unroll
        for (int i = 0; i < OCT_KR; i++) { const int k = tid + 256 * i; if (k < K0) keys[k] = rk[i]; }
        for (int k = tid + 256 * OCT_KR; k < K0; k += 256) keys[k] = fetch(k);
        __syncthreads();                                         // coff is dead: the node arrays take the LDS over
    } else {
        for (int c0 = 0; c0 < ncells; c0 += 256) {
            const int c = c0 + tid;
            const int n = c < ncells ? (int)ccount[c] : 0;
            int total;
            const int off = running + block_excl_scan256(n, wsum, &total);
            for (int i = 0; i < n; i++)
                if (off + i < L.key_cap) keys[off + i] = clist[(size_t)c * L.cell_cap + i];
            running += total;
        }
        if (running > L.key_cap) { if (tid == 0) atomicExch(P.status, ORBHIP_E_CAPACITY); running = L.key_cap; }
        __syncthreads();                                     // keys[] visible block-wide (same CU, L1 coherent within WG)
        __threadfence_block();
#pragma unroll
        for (int i = 0; i < OCT_KR; i++) { const int k = tid + 256 * i; rk[i] = k < running ? keys[k] : 0u; rn[i] = 0; }
    }
    const int K = running;
    if (tid == 0) P.lvl_ncand[frame * P.nlevels + lvl] = K;
    if (K == 0) { if (tid == 0) *count_out = 0; return; }
    // The first OCT_KR * 256 candidates (all of them unless a level holds more than 2048) live in REGISTERS from here on, key and
    // node id; candidates beyond that keep the global arrays.
    __syncthreads();
    __threadfence_block();
    OCT_T(0);
    // ---- roots (ORBextractor.cc:541-584)
    const int n_ini = L.n_ini;
    const int H = (L.h - ORB_MINB) - ORB_MINB;
    for (int i = tid; i < NC; i += 256) { cnt[i] = 0; ncnt[i] = 0; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OCT_KR; i++)
        if (tid + 256 * i < K) {
            const int r = (int)__fdiv_rn((float)ORB_KEY_X(rk[i]), L.hx);     // ORBextractor.cc:568
            rn[i] = r;
            atomicAdd(&ncnt[r], 1u);
        }
    for (int k = tid + 256 * OCT_KR; k < K; k += 256) {
        const int r = (int)__fdiv_rn((float)ORB_KEY_X(keys[k]), L.hx);
        node_of[k] = (uint16_t)r;
        atomicAdd(&ncnt[r], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        // compact non-empty roots, keep order; remap handled below through newpos
        int m = 0;
        for (int i = 0; i < n_ini; i++) {
            const uint32_t c = ncnt[i];
            S.newpos[i] = m;
            if (c) {
                box0[m] = (uint32_t)(int)__fmul_rn(L.hx, (float)i);              // UL.x | 0<<16
                box1[m] = (uint32_t)(int)__fmul_rn(L.hx, (float)(i + 1)) | ((uint32_t)H << 16);
                cnt[m] = c;
                m++;
            }
        }
        s_size = m; s_front = m;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OCT_KR; i++) if (tid + 256 * i < K) rn[i] = S.newpos[rn[i]];
    for (int k = tid + 256 * OCT_KR; k < K; k += 256) node_of[k] = (uint16_t)S.newpos[node_of[k]];
    __syncthreads();

    OCT_T(1);
    // ---- subdivision loop.  size / front are the same in every thread (derived from block-wide sums): kept in registers, no LDS broadcast
    bool final_phase = false;
    int size = s_size, front = s_front;
    for (int guard = 0; guard < 64; guard++) {
        // 1. candidate set + processing order
        int C;   // number of candidates
        if (!final_phase) {
            int run = 0;
            for (int p0 = 0; p0 < size; p0 += 256) {
                const int p = p0 + tid;
                const int f = (p < size && cnt[p] > 1) ? 1 : 0;
                int tot;
                const int o = run + block_excl_scan256(f, wsum, &tot);
                if (p < size) { S.ord[p] = f ? o : -1; if (f) S.by_ord[o] = p; }
                run += tot;
            }
            C = run;
        } else {
            // candidates: p < front with cnt > 1 ; order by (cnt desc, p asc)
            int run = 0;
            for (int p0 = 0; p0 < size; p0 += 256) {
                const int p = p0 + tid;
                const int f = (p < front && cnt[p] > 1) ? 1 : 0;
                int tot;
                const int o = run + block_excl_scan256(f, wsum, &tot);
                if (p < size) S.ord[p] = -1;
                if (f) S.excl[o] = p;                   // temp: compacted candidate list (ascending p)
                run += tot;
            }
            C = run;
            __syncthreads();
            for (int i = tid; i < C; i += 256) {
                const int p = S.excl[i];
                const uint32_t ci = cnt[p];
                int r = 0;
                for (int j = 0; j < C; j++) {
                    const uint32_t cj = cnt[S.excl[j]];
                    r += (cj > ci || (cj == ci && j < i)) ? 1 : 0;
                }
                S.ord[p] = r;
                S.by_ord[r] = p;
            }
        }
        __syncthreads();
        if (C == 0) break;                               // size == prevSize -> bFinish
        // 2. child key counts of every candidate
        for (int i = tid; i < 4 * size; i += 256) S.cc[i] = 0;
        if (tid == 0) { s_nexpand = 0; s_rstar = C - 1; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < OCT_KR; i++)
            if (tid + 256 * i < K) {
                const int p = rn[i];
                if (S.ord[p] >= 0) atomicAdd(&S.cc[4 * p + oct_quadrant(rk[i], box0[p], box1[p])], 1u);
            }
        for (int k = tid + 256 * OCT_KR; k < K; k += 256) {
            const int p = node_of[k];
            if (S.ord[p] >= 0) atomicAdd(&S.cc[4 * p + oct_quadrant(keys[k], box0[p], box1[p])], 1u);
        }
        __syncthreads();
        // 3. exclusive scan over order of the non-empty-children counts
        {
            int run = 0;
            for (int o0 = 0; o0 < C; o0 += 256) {
                const int o = o0 + tid;
                int nch = 0;
                if (o < C) { const uint32_t *q = &S.cc[4 * S.by_ord[o]]; nch = (q[0] > 0) + (q[1] > 0) + (q[2] > 0) + (q[3] > 0); }
                int tot;
                const int e = run + block_excl_scan256(nch, wsum, &tot);
                if (o < C) {
                    S.excl[o] = e;
                    // final phase: stop after the first split that reaches N (ORBextractor.cc:727-728)
                    if (final_phase && size + e + nch - (o + 1) >= N) atomicMin(&s_rstar, o);
                }
                run += tot;
            }
        }
        __syncthreads();
        const int rstar = s_rstar;
        int T, nproc;
        {   // every thread reads the same LDS words: no single-thread phase, no barrier
            const uint32_t *q = &S.cc[4 * S.by_ord[rstar]];
            T = S.excl[rstar] + (q[0] > 0) + (q[1] > 0) + (q[2] > 0) + (q[3] > 0);
            nproc = rstar + 1;
        }
        // 4. positions of surviving nodes: T + rank among non-processed (old order)
        {
            int run = 0;
            for (int p0 = 0; p0 < size; p0 += 256) {
                const int p = p0 + tid;
                const int surv = (p < size && !(S.ord[p] >= 0 && S.ord[p] <= rstar)) ? 1 : 0;
                int tot;
                const int e = run + block_excl_scan256(surv, wsum, &tot);
                if (p < size && surv) S.newpos[p] = T + e;
                run += tot;
            }
        }
        // 5. build the new node arrays
        for (int p = tid; p < size; p += 256) {
            const int o = S.ord[p];
            if (o >= 0 && o <= rstar) {
                const uint32_t *q = &S.cc[4 * p];
                const int nch = (q[0] > 0) + (q[1] > 0) + (q[2] > 0) + (q[3] > 0);
                const int base = T - (S.excl[o] + nch);          // frontmost child (n4 side)
                S.posbase[p] = base;
                const uint32_t b0 = box0[p], b1 = box1[p];
                const int x0 = b0 & 0xFFFF, y0 = b0 >> 16, x1 = b1 & 0xFFFF, y1 = b1 >> 16;
                const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
                int pos = base, nexp = 0;
                for (int qd = 3; qd >= 0; qd--) {
                    if (q[qd]) {
                        const int cx0 = (qd & 1) ? mx : x0, cx1 = (qd & 1) ? x1 : mx;
                        const int cy0 = (qd & 2) ? my : y0, cy1 = (qd & 2) ? y1 : my;
                        nbox0[pos] = (uint32_t)cx0 | ((uint32_t)cy0 << 16);
                        nbox1[pos] = (uint32_t)cx1 | ((uint32_t)cy1 << 16);
                        ncnt[pos] = q[qd];
                        nexp += q[qd] > 1;
                        pos++;
                    }
                }
                if (nexp) atomicAdd(&s_nexpand, nexp);
            } else {
                const int np = S.newpos[p];
                nbox0[np] = box0[p]; nbox1[np] = box1[p]; ncnt[np] = cnt[p];
            }
        }
        __syncthreads();
        // 6. re-label keys
        auto relabel = [&](uint32_t key, int p) {
            const int o = S.ord[p];
            if (o >= 0 && o <= rstar) {
                const int qd = oct_quadrant(key, box0[p], box1[p]);
                const uint32_t *q = &S.cc[4 * p];
                int r = 0;                                         // non-empty siblings in front (q' > qd)
                for (int j = 3; j > qd; j--) r += q[j] > 0;
                return S.posbase[p] + r;
            }
            return S.newpos[p];
        };
#pragma unroll
        for (int i = 0; i < OCT_KR; i++) if (tid + 256 * i < K) rn[i] = relabel(rk[i], rn[i]);
        for (int k = tid + 256 * OCT_KR; k < K; k += 256) node_of[k] = (uint16_t)relabel(keys[k], node_of[k]);
        __syncthreads();                                  // relabelled keys / new node arrays / s_nexpand complete; the next pass re-initialises s_nexpand
        const int new_size = T + (size - nproc);            // only after its own barrier (step 2), i.e. after every thread has read it here
        const int nexpand = s_nexpand;
        { uint32_t *t; t = box0; box0 = nbox0; nbox0 = t; t = box1; box1 = nbox1; nbox1 = t; t = cnt; cnt = ncnt; ncnt = t; }
        const int old_size = size;
        size = new_size; front = T;
#ifdef OCT_PROF
        if (blockIdx.x == 0 && tid == 0) g_oct_prof[5] += 1;
#endif
        // 7. termination (ORBextractor.cc:661-735)
        if (new_size >= N || new_size == old_size) break;
        if (!final_phase && new_size + 3 * nexpand > N) final_phase = true;
    }
    __syncthreads();
    OCT_T(2);
    // ---- best key per node: max response, first in list order wins (ORBextractor.cc:739-758)
    for (int i = tid; i < size; i += 256) S.best[i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OCT_KR; i++)
        if (tid + 256 * i < K) atomicMax(&S.best[rn[i]], ((uint32_t)ORB_KEY_S(rk[i]) << 20) | (uint32_t)(0xFFFFF - (tid + 256 * i)));
    for (int k = tid + 256 * OCT_KR; k < K; k += 256)
        atomicMax(&S.best[node_of[k]], ((uint32_t)ORB_KEY_S(keys[k]) << 20) | (uint32_t)(0xFFFFF - k));
    __syncthreads();
    OCT_T(3);
    uint32_t *out = P.lvl_kp + (size_t)frame * P.kps_per_frame + L.kp_base;
    int nout = size;
    if (nout > L.kp_cap) { if (tid == 0) atomicExch(P.status, ORBHIP_E_CAPACITY); nout = L.kp_cap; }
    // processing order for k_orient_desc: rows of 32-pixel tiles, x inside a row (rank by counting over LDS; nout <= quota + 8)
    uint16_t *perm = P.lvl_perm + (size_t)frame * P.kps_per_frame + L.kp_base;
    for (int i = tid; i < nout; i += 256) out[i] = keys[0xFFFFF - (S.best[i] & 0xFFFFF)];
    if (P.batch < ORB_PERM_MIN_BATCH) { if (tid == 0) *count_out = nout; return; }     // few frames: cache reuse is not the limit, latency is
    __syncthreads();                                      // everyone has read S.best: it now holds the spatial sort keys
    for (int i = tid; i < nout; i += 256) { const uint32_t ki = out[i]; S.best[i] = ((uint32_t)(ORB_KEY_Y(ki) >> 5) << 16) | (uint32_t)ORB_KEY_X(ki); }
    __syncthreads();
    for (int i = tid; i < nout; i += 256) {
        const uint32_t si = S.best[i];
        int rank = 0;
        for (int j = 0; j < nout; j++) { const uint32_t sj = S.best[j]; rank += (sj < si) || (sj == si && j < i); }
        perm[rank] = (uint16_t)i;
    }
    if (tid == 0) *count_out = nout;
    OCT_T(4);
}

const void *orb_octree_func() { return reinterpret_cast<const void *>(k_octree); }

void orb_launch_octree(const OrbParams &P, hipStream_t s)
{
    const size_t lds = orb_octree_lds_bytes(P.oct_nc);      // > 64 KB from about 4000 features per level on: opted in by the caller
    hipLaunchKernelGGL(k_octree, dim3(P.nlevels * P.batch), dim3(256), lds, s, P);
}

// ----------------------------------------------------------------------------------
// A7  7x7 sigma-2 Gaussian, 8-bit fixed point, BORDER_REFLECT_101 (ORBextractor.cc:1114-1115;
// arithmetic SURVEY Appendix A.7).  Separable inside LDS: the row pass of an 8-bit image
// with the q8 kernel (sum 257) fits u16 exactly (255*257 = 65535), column pass
// (sum + 2^15) >> 16 saturated.  Tile: 64x32 outputs per 256-thread workgroup.
// ----------------------------------------------------------------------------------
#define BL_TW 64
#define BL_TH 32
#define BL_ROWS (BL_TH + 6)
#define BL_IPD 20             // staged-row pitch (dwords): 3 pad | left apron | 16 tile dwords; the right apron is dword 0 of the next row
__device__ __forceinline__ int reflect101(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p;
}

// four pixels x..x+3 of an image row with BORDER_REFLECT_101 outside [0, w)  (x % 4 == 0)
__device__ __forceinline__ uint32_t bl_load4(const uint8_t *row, int x, int w)
{
    if (x >= 0 && x + 3 < w) return *reinterpret_cast<const uint32_t *>(row + x);
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int xx = reflect101(x + j, w);
        xx = min(max(xx, 0), w - 1);
        v |= (uint32_t)row[xx] << (8 * j);
    }
    return v;
}

// horizontal 7-tap of four neighbouring pixels from the 12-byte window d0|d1|d2 (pixel 0 = byte 4), u16 each
__device__ __forceinline__ void bl_row4(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t klo, uint32_t khi, uint32_t (&o)[4])
{
    o[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), klo, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), khi, 0u, false), false);
    o[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), klo, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), khi, 0u, false), false);
    o[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), klo, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), khi, 0u, false), false);
    o[3] = __builtin_amdgcn_udot4(d1, klo, __builtin_amdgcn_udot4(d2, khi, 0u, false), false);
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bl_dot2(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
}

// ----------------------------------------------------------------------------------
// A7 over every pyramid level in ONE launch: the 64x32 tile (+3 rows / +4 columns of apron, BORDER_REFLECT_101) is staged
// with one 16-byte load per lane; row pass on v_dot4_u32_u8 (two rows per task, u16 sums written vertically PAIRED),
// column pass on v_dot2_u32_u16 (a 7-tap column = four dot2), saturate + pack, dword stores.
// Bound by vector-instruction issue (~19 lane-ops per pixel).  Algorithmic bytes: S read + S written.
// ----------------------------------------------------------------------------------
// tile cursor of a persistent k_blur workgroup: (level, frame, tile row / column), advanced without divisions
struct BlTile { int level, frame, tx, ty, ntx, nty, w, h, pitch; const uint8_t *src; uint8_t *dst; int dpitch; };
__device__ __forceinline__ void bl_tile_level(const OrbParams &P, BlTile &T)
{
    const OrbLevel &L = P.lv[T.level];
    T.w = __builtin_amdgcn_readfirstlane(L.w); T.h = __builtin_amdgcn_readfirstlane(L.h);
    T.pitch = __builtin_amdgcn_readfirstlane(L.img_pitch); T.dpitch = __builtin_amdgcn_readfirstlane(L.blur_pitch);
    T.ntx = (T.w + BL_TW - 1) / BL_TW; T.nty = (T.h + BL_TH - 1) / BL_TH;
}
__device__ __forceinline__ void bl_tile_frame(const OrbParams &P, BlTile &T)
{
    const OrbLevel &L = P.lv[T.level];
    T.src = L.img + (size_t)T.frame * L.img_frame_stride; T.dst = L.blur + (size_t)T.frame * L.blur_frame_stride;
}
// this lane's share of a tile's staging: lanes 0..151 one 16-byte chunk (38 rows x 4), lanes 152..227 one apron dword
__device__ __forceinline__ uint4 bl_stage_load(const BlTile &T, int tid)
{
    const bool chunk = tid < 4 * BL_ROWS;
    const int a = tid - 4 * BL_ROWS;
    const int r = chunk ? tid >> 2 : a >> 1;
    const int x0 = T.tx * BL_TW, y0 = T.ty * BL_TH, w = T.w;
    int y = reflect101(y0 + r - 3, T.h);
    y = min(max(y, 0), T.h - 1);
    const uint8_t *row = T.src + (uint32_t)__umul24((uint32_t)y, (uint32_t)T.pitch);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (chunk) {
        const int x = x0 + 16 * (tid & 3);
        if (x + 15 < w) v = *reinterpret_cast<const uint4 *>(row + x);
        else { v.x = bl_load4(row, x, w); v.y = bl_load4(row, x + 4, w); v.z = bl_load4(row, x + 8, w); v.w = bl_load4(row, x + 12, w); }
    } else if (a < 2 * BL_ROWS) {
        v.x = bl_load4(row, (a & 1) ? x0 + BL_TW : x0 - 4, w);
    }
    return v;
}

__global__ __launch_bounds__(256) void k_blur(OrbParams P)
{
    __shared__ __attribute__((aligned(16))) uint32_t in[BL_ROWS * BL_IPD + 4];
    __shared__ __attribute__((aligned(16))) uint32_t hz2[(BL_ROWS / 2) * BL_TW];     // [row pair][x]: row 2m | row 2m+1 << 16
    const int tid = threadIdx.x;
    // persistent workgroup: a contiguous range of tiles in (level, frame, tile row, tile column) order -- each XCD's contiguous range of
    // logical ids stays inside few frames of one level; the next tile's loads are in flight while the current one is filtered
    const unsigned lid = xcd_logical_id(blockIdx.x, gridDim.x);
    const long total = (long)P.bs_tiles[P.nlevels] * P.batch;
    const long t0 = total * lid / gridDim.x, t1 = total * (lid + 1) / gridDim.x;
    if (t0 >= t1) return;
    BlTile T;
    {
        T.level = 0;
        for (int l = 1; l < P.nlevels; l++) if (t0 >= (long)P.bs_tiles[l] * P.batch) T.level = l;
        bl_tile_level(P, T);
        const long rel = t0 - (long)P.bs_tiles[T.level] * P.batch;
        T.frame = (int)(rel / (T.ntx * T.nty));
        const int trem = (int)(rel - (long)T.frame * (T.ntx * T.nty));
        T.ty = trem / T.ntx; T.tx = trem - T.ty * T.ntx;
        bl_tile_frame(P, T);
    }
    const uint32_t klo = (uint32_t)P.gauss_q8[0] | ((uint32_t)P.gauss_q8[1] << 8) | ((uint32_t)P.gauss_q8[2] << 16) | ((uint32_t)P.gauss_q8[3] << 24);
    const uint32_t khi = (uint32_t)P.gauss_q8[4] | ((uint32_t)P.gauss_q8[5] << 8) | ((uint32_t)P.gauss_q8[6] << 16);
    const uint32_t k0 = P.gauss_q8[0], k1 = P.gauss_q8[1], k2 = P.gauss_q8[2], k3 = P.gauss_q8[3];
    const uint32_t wt[2][4] = {{k0 | (k1 << 16), k2 | (k3 << 16), k2 | (k1 << 16), k0},
                               {k0 << 16, k1 | (k2 << 16), k3 | (k2 << 16), k1 | (k0 << 16)}};
    // where this lane's staging share lands in LDS (same for every tile)
    const bool chunk = tid < 4 * BL_ROWS;
    const int sa = tid - 4 * BL_ROWS;
    const int sidx = chunk ? (tid >> 2) * BL_IPD + 4 + 4 * (tid & 3) : (sa >> 1) * BL_IPD + ((sa & 1) ? BL_IPD : 3);
    auto advance = [&](BlTile &X) {
        X.tx++;
        if (X.tx == X.ntx) { X.tx = 0; X.ty++; }
        if (X.ty == X.nty) {
            X.ty = 0; X.frame++;
            if (X.frame == P.batch) { X.frame = 0; X.level++; bl_tile_level(P, X); }
            bl_tile_frame(P, X);
        }
    };
    auto body = [&](uint4 &ld, BlTile &T, long t) {
        const int x0 = T.tx * BL_TW, y0 = T.ty * BL_TH, w = T.w, h = T.h, dpitch = T.dpitch;
        uint8_t *bdst = T.dst;
        if (chunk) *reinterpret_cast<uint4 *>(&in[sidx]) = ld;
        else if (sa < 2 * BL_ROWS) in[sidx] = ld.x;
        __syncthreads();
        if (t + 1 < t1) { advance(T); ld = bl_stage_load(T, tid); }                   // the next tile's loads are in flight during the passes
        // ---- row pass: two rows x four pixels per task, u16 sums (q8 kernel, sum 257 -> 255*257 fits)
        for (int i = tid; i < (BL_ROWS / 2) * (BL_TW / 4); i += 256) {
            const int rp = i >> 4, c4 = i & 15;
            const uint32_t *q = &in[2 * rp * BL_IPD + 3 + c4];
            uint32_t oa[4], ob[4];
            bl_row4(q[0], q[1], q[2], klo, khi, oa);
            bl_row4(q[BL_IPD], q[BL_IPD + 1], q[BL_IPD + 2], klo, khi, ob);
            uint4 pk;
            pk.x = oa[0] | (ob[0] << 16); pk.y = oa[1] | (ob[1] << 16); pk.z = oa[2] | (ob[2] << 16); pk.w = oa[3] | (ob[3] << 16);
            *reinterpret_cast<uint4 *>(&hz2[rp * BL_TW + 4 * c4]) = pk;
        }
        __syncthreads();
        // ---- column pass: rows come vertically paired, so a 7-tap column is four v_dot2_u32_u16
        {
            const int c4 = tid & 15, rg = tid >> 4;
            const int x = x0 + 4 * c4;
            if (x < w) {
                uint4 pr[4];
#pragma unroll
                for (int k = 0; k < 4; k++) pr[k] = *reinterpret_cast<const uint4 *>(&hz2[(rg + k) * BL_TW + 4 * c4]);
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int y = y0 + 2 * rg + rr;
                    if (y < h) {
                        uint32_t acc[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            uint32_t a = 32768u;
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const uint32_t pv = j == 0 ? pr[k].x : j == 1 ? pr[k].y : j == 2 ? pr[k].z : pr[k].w;
                                a = bl_dot2(pv, wt[rr][k], a);
                            }
                            acc[j] = min(a, 0x00FFFFFFu);                  // byte 2 = min((sum + 2^15) >> 16, 255)
                        }
                        const uint32_t p01 = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u);
                        const uint32_t p23 = __builtin_amdgcn_perm(acc[3], acc[2], 0x0c0c0602u);
                        *reinterpret_cast<uint32_t *>(bdst + (uint32_t)__umul24((uint32_t)y, (uint32_t)dpitch) + x) = p01 | (p23 << 16);
                    }
                }
            }
        }
        // (the next tile's staging writes to `in` come after this tile's row pass: second barrier above;
        //  its row-pass writes to `hz2` come after its first barrier, i.e. after this column pass)
    };
    uint4 ld = bl_stage_load(T, tid);
    for (long t = t0; t < t1; t++) body(ld, T, t);
}

// Row-streaming form of the same filter (the default): no LDS, no barriers, so its waves fit into the wave slots k_fast_cells
// leaves free (that kernel is LDS-limited to ~17 waves per CU) and fill its idle issue cycles.  A lane owns 4 output columns and
// walks BLR_R output rows down: per source row one aligned dwordx3 (x-4 .. x+7), the row pass on v_dot4_u32_u8 as above, the
// horizontal sums of consecutive rows paired in a register ring of 6 (statically indexed: the row loop is unrolled by 6), the
// column pass = three v_dot2_u32_u16 on the pairs + one v_mad_u32_u24.  Chunks whose 12-byte window crosses the left / right
// image border (BORDER_REFLECT_101) are a separate lane class at the end of each (frame, level) lane range and assemble their
// windows bytewise.
template <bool EDGE>
__device__ __forceinline__ void blr_load(const uint8_t *src, int r, int spitch, int x, int w, uint32_t &d0, uint32_t &d1, uint32_t &d2)
{
    if (EDGE) {
        const uint8_t *row = src + (uint32_t)__umul24((uint32_t)r, (uint32_t)spitch);
        d0 = bl_load4(row, x - 4, w); d1 = bl_load4(row, x, w); d2 = bl_load4(row, x + 4, w);
    } else {
        const rs_u32x3 v = *reinterpret_cast<const rs_u32x3_a4 *>(src + ((uint32_t)__umul24((uint32_t)r, (uint32_t)spitch) + (uint32_t)(x - 4)));
        d0 = v.x; d1 = v.y; d2 = v.z;
    }
}
template <bool EDGE>
__device__ __forceinline__ void blr_band(const uint8_t *src, uint8_t *dst, int spitch, int dpitch, int w, int h, int x, int dy0,
                                         uint32_t klo, uint32_t khi, uint32_t k01, uint32_t k23, uint32_t k21, uint32_t k0)
{
    auto srow = [&](int i) { int r = dy0 - 3 + i; r = max(r, -r); return min(r, 2 * h - 2 - r); };    // BORDER_REFLECT_101 (h >= BLR_R + 4)
    // the six source rows of the NEXT group are in flight while this group is filtered (24 unique bytes per lane: the kernel is
    // bound by bytes in flight, not by issue); buffers and ring are statically indexed, the whole band is unrolled
    uint32_t ring[6][4], prev[4] = {0, 0, 0, 0};
    uint32_t buf[2][6][3];
#pragma unroll
    for (int u = 0; u < 6; u++) blr_load<EDGE>(src, srow(u), spitch, x, w, buf[0][u][0], buf[0][u][1], buf[0][u][2]);
    uint32_t doff = (uint32_t)__umul24((uint32_t)dy0, (uint32_t)dpitch) + (uint32_t)x;
#pragma unroll
    for (int g = 0; g < (BLR_R + 6) / 6; g++) {
        if (g + 1 < (BLR_R + 6) / 6) {
#pragma unroll
            for (int u = 0; u < 6; u++) blr_load<EDGE>(src, srow(6 * (g + 1) + u), spitch, x, w, buf[(g + 1) & 1][u][0], buf[(g + 1) & 1][u][1], buf[(g + 1) & 1][u][2]);
        }
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const int i = 6 * g + u;
            uint32_t o[4];
            bl_row4(buf[g & 1][u][0], buf[g & 1][u][1], buf[g & 1][u][2], klo, khi, o);
            if (g > 0) {
                uint32_t acc[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t a = bl_dot2(ring[(u + 1) % 6][j], k01, 32768u);
                    a = bl_dot2(ring[(u + 3) % 6][j], k23, a);
                    a = bl_dot2(ring[(u + 5) % 6][j], k21, a);
                    a = __umul24(o[j], k0) + a;
                    acc[j] = min(a, 0x00FFFFFFu);                  // byte 2 = min((sum + 2^15) >> 16, 255)
                }
                const uint32_t p01 = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u);
                const uint32_t p23 = __builtin_amdgcn_perm(acc[3], acc[2], 0x0c0c0602u);
                if (dy0 + i - 6 < h) *reinterpret_cast<uint32_t *>(dst + doff) = p01 | (p23 << 16);
                doff += (uint32_t)dpitch;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) { ring[u][j] = prev[j] | (o[j] << 16); prev[j] = o[j]; }
        }
    }
}

__global__ __launch_bounds__(256) void k_blur_rows(OrbParams P, int frame0)
{
    const int per_frame = P.br_blocks[P.nlevels];
    const unsigned lid = xcd_logical_id(blockIdx.x, gridDim.x);
    const int fr = (int)(lid / (unsigned)per_frame), rem = (int)(lid - (unsigned)fr * (unsigned)per_frame);     // uniform
    const int frame = frame0 + fr;
    int level = 0;
    for (int l = 1; l < P.nlevels; l++) if (rem >= P.br_blocks[l]) level = l;
    const OrbLevel &L = P.lv[level];
    const int w = L.w, h = L.h;
    const int nch = (w + 3) >> 2, nint = (w - 8) >> 2, nedge = nch - nint, nb = (h + BLR_R - 1) / BLR_R;
    const int idx = (rem - P.br_blocks[level]) * 256 + (int)threadIdx.x;
    if (idx >= nch * nb) return;
    const uint8_t *src = L.img + (size_t)frame * L.img_frame_stride;
    uint8_t *dst = L.blur + (size_t)frame * L.blur_frame_stride;
    const uint32_t klo = (uint32_t)P.gauss_q8[0] | ((uint32_t)P.gauss_q8[1] << 8) | ((uint32_t)P.gauss_q8[2] << 16) | ((uint32_t)P.gauss_q8[3] << 24);
    const uint32_t khi = (uint32_t)P.gauss_q8[4] | ((uint32_t)P.gauss_q8[5] << 8) | ((uint32_t)P.gauss_q8[6] << 16);
    const uint32_t k0 = P.gauss_q8[0], k1 = P.gauss_q8[1], k2 = P.gauss_q8[2], k3 = P.gauss_q8[3];
    const uint32_t k01 = k0 | (k1 << 16), k23 = k2 | (k3 << 16), k21 = k2 | (k1 << 16);
    if (idx < nint * nb) {
        const int band = idx / nint, c = 1 + idx - band * nint;
        blr_band<false>(src, dst, L.img_pitch, L.blur_pitch, w, h, 4 * c, band * BLR_R, klo, khi, k01, k23, k21, k0);
    } else {
        const int e = idx - nint * nb, band = e / nedge, ce = e - band * nedge;
        const int c = ce == 0 ? 0 : nint + ce;
        blr_band<true>(src, dst, L.img_pitch, L.blur_pitch, w, h, 4 * c, band * BLR_R, klo, khi, k01, k23, k21, k0);
    }
}

// ----------------------------------------------------------------------------------
// A7 on the matrix cores (round 3).  The blur is exact integer arithmetic (taps q8, no intermediate rounding, one (sum + 2^15) >> 16 at the
// end), so it can be regrouped freely: a 64 x 64 window of pixels times a banded 64 x 16 matrix of taps is the row pass for 16 output
// columns, and the column pass is the same product along the other axis.  Why: k_fast_cells and the blur share the CUs and are bound by the
// SUM of their vector instructions; k_blur_rows costs 14 per pixel, this form ~6, the multiply-adds go to the otherwise idle
// matrix pipe.  No LDS (k_fast_cells owns it), no cross-lane movement:
//   pass 1  C1[row][out col] = sum_k A[row][k] B[k][out col]: A = the window's pixels as int8 (p - 128: one xor per dword), lane (row & 15,
//           g = lane >> 4) holds the 16 bytes of chunk g of its row -- ONE 16-byte load; B = the taps as a band, from a host-built table per
//           (level, tile column, block) that also folds BORDER_REFLECT_101 in (a reflected column's tap is added to the column it mirrors)
//           and mirrors the chunk rule below; the accumulators start at 128 * (sum of taps), which undoes the bias: C1 = the plain 16-bit row sum.
//   pass 2  C2[out col][out row] = sum_k A2[out col][k] B2[k][out row]: A2 = C1 itself -- its layout (column on the lane, rows 4g + r in the
//           registers of the four row tiles) IS an A operand whose k slot (g, 4t + r) means window row 16t + 4g + r; the band table is built
//           in that slot order (any order works as long as both operands agree, tools/mfma_i8_probe.hip).  16-bit sums go in as two int8
//           products (high and low byte, each biased), 256 * hi + lo + 2^15 is formed in the epilogue; the result tile has 4 consecutive
//           COLUMNS of one output row per lane: one dword store.
// A wave owns a 32-column tile column and walks down in steps of 58 rows (64-row windows, 3 rows of apron each side; rows are reflected by
// index at load time).  Chunk rule (the table's too): chunk g of the window starts at 32 tx - 16 + 16 g, moved to 0 when negative and to
// w - 16 when it would cross the row's end -- loads never leave [0, w) of a row (level 0 may be the caller's buffer).
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i bm_as_v4i(uint4 v) { v4i r; r[0] = (int)v.x; r[1] = (int)v.y; r[2] = (int)v.z; r[3] = (int)v.w; return r; }
__global__ __launch_bounds__(256) void k_blur_mfma(OrbParams P, int frame0, int nframes)
{
    const int per_frame = P.bm_cols[P.nlevels];
    // XCD-contiguous order: the four tile columns of a workgroup span 128 + 32 columns, so raw neighbours (dealt to different L2s) fetched
    // the 128-byte lines at their seams twice or three times (2.5 GB read per 1024 VGA frames against 0.98 GB of pixels, profiles/r04_pmc_traffic)
    const unsigned wid = xcd_logical_id(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const int fr = (int)(wid / (unsigned)per_frame), rem = (int)(wid - (unsigned)fr * (unsigned)per_frame);
    if (fr >= nframes) return;
    int level = 0;
    for (int l = 1; l < P.nlevels; l++) if (rem >= P.bm_cols[l]) level = l;
    const OrbLevel &L = P.lv[level];
    const int w = L.w, h = L.h, spitch = L.img_pitch, dpitch = L.blur_pitch;
    const int tx = rem - P.bm_cols[level], X0 = 32 * tx;
    const uint8_t *src = L.img + (size_t)(frame0 + fr) * L.img_frame_stride;
    uint8_t *dst = L.blur + (size_t)(frame0 + fr) * L.blur_frame_stride;
    int cx = X0 - 16 + 16 * g;
    cx = cx < 0 ? 0 : cx;
    if (cx + 16 > w) cx = w - 16;
    const uint4 *th = P.bm_th + ((size_t)(P.bm_cols[level] + tx) * 2) * 64 + lane;
    const v4i B0 = bm_as_v4i(th[0]), B1 = bm_as_v4i(th[64]);
    v4i TV[4];
#pragma unroll
    for (int b = 0; b < 4; b++) TV[b] = bm_as_v4i(P.bm_tv[b * 64 + lane]);
    const int init = P.bm_init;
    const v4i c_init = {init, init, init, init}, c_init_lo = {init + 32768, init + 32768, init + 32768, init + 32768};
    const int nty = (h + 57) / 58;
    // the window of the NEXT step is in flight while this one is multiplied (one 16-byte load per lane and row tile)
    auto load_window = [&](int Y0, u32x4_unaligned (&raw)[4]) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            int ya = Y0 - 3 + 16 * t + m;
            ya = ya < 0 ? -ya : ya;
            ya = ya >= h ? 2 * h - 2 - ya : ya;                       // BORDER_REFLECT_101; rows far below the image (last window) are never used
            ya = min(max(ya, 0), h - 1);
            raw[t] = *reinterpret_cast<const u32x4_unaligned *>(src + (uint32_t)(__umul24((uint32_t)ya, (uint32_t)spitch) + (uint32_t)cx));
        }
    };
    u32x4_unaligned raw[4];
    load_window(0, raw);
    for (int ty = 0; ty < nty; ty++) {
        const int Y0 = 58 * ty;
        v4i A[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            A[t][0] = (int)(raw[t].x ^ 0x80808080u); A[t][1] = (int)(raw[t].y ^ 0x80808080u); A[t][2] = (int)(raw[t].z ^ 0x80808080u); A[t][3] = (int)(raw[t].w ^ 0x80808080u);
        }
        if (ty + 1 < nty) load_window(Y0 + 58, raw);
        v4i C1[2][4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            C1[0][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], B0, c_init, 0, 0, 0);
            C1[1][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], B1, c_init, 0, 0, 0);
        }
#pragma unroll
        for (int cb = 0; cb < 2; cb++) {
            v4i Ahi, Alo;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t pab = __builtin_amdgcn_perm((uint32_t)C1[cb][t][1], (uint32_t)C1[cb][t][0], 0x05040100u);      // (lo, hi) byte pairs of rows r = 0, 1
                const uint32_t pcd = __builtin_amdgcn_perm((uint32_t)C1[cb][t][3], (uint32_t)C1[cb][t][2], 0x05040100u);
                Alo[t] = (int)(__builtin_amdgcn_perm(pcd, pab, 0x06040200u) ^ 0x80808080u);
                Ahi[t] = (int)(__builtin_amdgcn_perm(pcd, pab, 0x07050301u) ^ 0x80808080u);
            }
            const int col = X0 + 16 * cb + 4 * g;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const v4i Chi = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ahi, TV[b], c_init, 0, 0, 0);
                const v4i Clo = __builtin_amdgcn_mfma_i32_16x16x64_i8(Alo, TV[b], c_init_lo, 0, 0, 0);
                uint32_t v[4];
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = min(((uint32_t)Chi[r] << 8) + (uint32_t)Clo[r], 0x00FFFFFFu);      // byte 2 = min((sum + 2^15) >> 16, 255)
                const uint32_t p01 = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u), p23 = __builtin_amdgcn_perm(v[3], v[2], 0x0c0c0602u);
                const int row = Y0 + 16 * b + m;
                if (16 * b + m < 58 && row < h && col < w) *reinterpret_cast<uint32_t *>(dst + (uint32_t)(__umul24((uint32_t)row, (uint32_t)dpitch) + (uint32_t)col)) = p01 | (p23 << 16);
            }
        }
    }
}

void orb_launch_blur(const OrbParams &P, hipStream_t s, int wgs_per_cu, int frame0, int nframes)
{
    if (P.bm_cols[P.nlevels] > 0 && P.batch >= P.bm_min_batch) {
        if (nframes < 0) nframes = P.batch - frame0;
        const long waves = (long)P.bm_cols[P.nlevels] * nframes;
        if (nframes > 0) hipLaunchKernelGGL(k_blur_mfma, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, P, frame0, nframes);
        return;
    }
    if (P.br_blocks[P.nlevels] > 0 && P.batch >= P.rows_min_batch) {
        if (nframes < 0) nframes = P.batch - frame0;
        if (nframes > 0) hipLaunchKernelGGL(k_blur_rows, dim3((unsigned)P.br_blocks[P.nlevels] * (unsigned)nframes), dim3(256), 0, s, P, frame0);
        return;
    }
    if (frame0 > 0) return;                                    // the tile kernel always takes the whole batch (first call)
    const long total = (long)P.bs_tiles[P.nlevels] * P.batch;
    long nblocks = (long)(P.n_cus > 0 ? P.n_cus : 256) * wgs_per_cu;                           // persistent: 8 workgroups (32 waves) per CU when alone on the chip
    if (nblocks > total) nblocks = total;
    hipLaunchKernelGGL(k_blur, dim3((unsigned)nblocks), dim3(256), 0, s, P);
}

// ----------------------------------------------------------------------------------
// A6 + A8  IC_Angle (ORBextractor.cc:75-102) + steered BRIEF (ORBextractor.cc:106-145),
// one wave per keypoint.  Orientation: 2 patch rows per step (lanes 0-30 / 32-62),
// int32 moments, wave reduction.  Descriptor: lane l evaluates tests l, l+64, l+128,
// l+192; a 64-bit ballot is 8 descriptor bytes (LSB-first, as the reference packs them).
// ----------------------------------------------------------------------------------
__constant__ int8_t c_pattern[1024] = {
#include "orb_pattern.inc"
};

// cv::fastAtan2 scalar path, op-by-op, no contraction.
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, (float)2.2204460492503131e-16));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, (float)2.2204460492503131e-16));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// "orb_sincos" (DESIGN.md): fixed IEEE-double fma sequence, identical to the oracle's.
__device__ __forceinline__ void sincos_det(double x, double *s_out, double *c_out)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const int k = (int)__dadd_rn(__dmul_rn(x, TWO_OVER_PI), 0.5);
    const double dk = (double)k;
    double r = fma(-dk, PIO2_HI, x);
    r = fma(-dk, PIO2_LO, r);
    const double z = __dmul_rn(r, r);
    double ps = fma(z, S6, S5); ps = fma(z, ps, S4); ps = fma(z, ps, S3); ps = fma(z, ps, S2); ps = fma(z, ps, S1);
    const double s = fma(__dmul_rn(r, z), ps, r);
    double pc = fma(z, C6, C5); pc = fma(z, pc, C4); pc = fma(z, pc, C3); pc = fma(z, pc, C2); pc = fma(z, pc, C1);
    const double c = fma(__dmul_rn(z, z), pc, fma(z, -0.5, 1.0));
    switch (k & 3) {
    case 0: *s_out = s; *c_out = c; break;
    case 1: *s_out = c; *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
    }
}

// Sixteen lanes per keypoint (four keypoints per wave), persistent waves with a grid stride.
// Per keypoint the 16 lanes stage the 31x31 un-blurred patch and the 39x39 blurred patch into LDS
// with coalesced dword loads, accumulate the 749-pixel moments from LDS (two patch columns per
// lane, reduced over the 16 lanes), evaluate cv::fastAtan2 / orb_sincos once per keypoint (the four
// keypoints of a wave share those instructions), and each lane produces 16 of the 256 rBRIEF bits
// = two consecutive descriptor bytes.
#define OD_UP 48              // un-blurred patch pitch: 31 + up to 3 alignment bytes, staged as three 16-byte chunks
#define OD_BP 48              // blurred patch pitch: 39 + up to 3, three 16-byte chunks
#define OD_KP_LDS ((39 * OD_BP) / 4)                     // dwords per keypoint: the un-blurred patch (moments) and then the blurred one (rBRIEF) share it
#define OD_THREADS 128        // 8 keypoints per workgroup: 19.6 KB of LDS -> 8 workgroups (16 waves) per CU
// 16-byte fetch at a dword-aligned x: one dwordx4 per lane instead of four dword loads (the kernel's time
// follows the number of load wave-instructions).  A chunk may run up to 16 bytes past the end of its row:
// those bytes are never sampled (patches lie inside the image) and the read stays inside the buffer -- the
// un-blurred patch ends at image row <= h-5 (so the overrun lands in a later row, also when level 0 aliases
// the caller's frames), the blurred levels are allocated with 64 bytes of slack (orbhip_extractor_reserve).
__device__ __forceinline__ uint4 od_load16(const uint8_t *img, uint32_t row_off, int x)
{
    return *reinterpret_cast<const uint4 *>(img + (row_off + (uint32_t)x));      // uniform base + 32-bit lane offset
}

__global__ __launch_bounds__(OD_THREADS) void k_orient_desc(OrbParams P)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[OD_THREADS / 16][OD_KP_LDS];
    __shared__ float4 pat_t[16 * 16];            // pat_t[t][l] = test 16*l + t as floats x0,y0,x1,y1
    __shared__ uint32_t od_msk[16 * 8];          // od_msk[|v|][j]: bytes of columns -15+4j..-12+4j inside the circular patch
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, sub = lane >> 4;
    for (int e = tid; e < 256; e += OD_THREADS) {
        const int l = e & 15, t = e >> 4;         // 16 x 16 entries
        const int8_t *pp = &c_pattern[4 * (16 * l + t)];
        pat_t[t * 16 + l] = make_float4((float)pp[0], (float)pp[1], (float)pp[2], (float)pp[3]);
    }
    if (tid < 16 * 8) {
        const int d = P.umax[tid >> 3], j = tid & 7;
        uint32_t m = 0;
        for (int k = 0; k < 4; k++) { const int u = -ORB_HALF_PATCH + 4 * j + k; if (u >= -d && u <= d) m |= 0xFFu << (8 * k); }
        od_msk[tid] = m;
    }
    __syncthreads();
    // XCD-aware work split (speed only): workgroups with equal blockIdx.x % 8 share an XCD and its L2, so
    // frame f is handled by the workgroups of XCD (f % 8): every 128-byte line of a frame's pyramid is then
    // fetched from HBM by one L2 instead of eight.
    const int xcd = blockIdx.x & 7, nblk_x = gridDim.x >> 3;             // gridDim.x is a multiple of 8
    const long wave_x = (long)(blockIdx.x >> 3) * (OD_THREADS / 64) + (tid >> 6), nwaves_x = (long)nblk_x * (OD_THREADS / 64);
    const int gpf = (P.kps_per_frame + 3) >> 2;                          // groups of 4 slots per frame
    const int nframes_x = (P.batch - xcd + 7) >> 3;                      // frames xcd, xcd+8, ...
    uint32_t *up32 = lds_all[(tid >> 4)], *bp32 = up32;
    const uint8_t *bp = reinterpret_cast<const uint8_t *>(bp32);
    const float factor_pi = (float)(3.1415926535897932384626433832795 / 180.f);
    const long ngroups = (long)nframes_x * gpf;
    for (long q = wave_x; q < ngroups; q += nwaves_x) {
        // every per-level staging slice is a multiple of 4 slots (orbhip_extractor_reserve), so the 4 keypoints
        // of a wave share frame and level: level parameters stay in scalar registers
        const int fk = __builtin_amdgcn_readfirstlane((int)(q / gpf));
        const int frame = xcd + 8 * fk;
        const int slot0 = __builtin_amdgcn_readfirstlane(4 * (int)(q - (long)fk * gpf));
        int lvl = 0;
        for (int l = 1; l < P.nlevels; l++) if (slot0 >= P.lv[l].kp_base) lvl = l;
        lvl = __builtin_amdgcn_readfirstlane(lvl);
        const OrbLevel &L = P.lv[lvl];
        const int slot_w = slot0 + sub;                                  // work slot: the slot_w-th keypoint of the level in spatial order
        const int nk = P.lvl_count[frame * P.nlevels + lvl];
        const bool valid = (slot_w - L.kp_base) < nk;
        const int slot = (valid && P.batch >= ORB_PERM_MIN_BATCH) ? L.kp_base + (int)P.lvl_perm[(size_t)frame * P.kps_per_frame + slot_w] : slot_w;
        if (slot0 - L.kp_base >= nk) continue;
        int x = 19, y = 19;
        if (valid) {
            const uint32_t key = P.lvl_kp[(size_t)frame * P.kps_per_frame + slot];
            x = ORB_KEY_X(key) + ORB_MINB; y = ORB_KEY_Y(key) + ORB_MINB;           // ORBextractor.cc:868-869
        }
        // ---- stage both patches (16 lanes per keypoint)
        const int uxs = (x - 15) & ~3, uoff = (x - 15) - uxs;        // un-blurred: cols x-15..x+15
        const int bxs = (x - 19) & ~3, boff = (x - 19) - bxs;        // blurred:    cols x-19..x+19
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // previous keypoints' LDS reads are done
        // the 16 lanes sweep the patch rows in 16-byte chunks (3 per row): 6 + 8 dwordx4 loads per lane, all issued
        // before the first LDS store so that they are in flight together; the blurred chunks wait in registers
        // until the moments are done with the shared LDS region
        uint4 br[8];
#pragma unroll
        for (int k = 0; k < 8; k++) br[k] = make_uint4(0, 0, 0, 0);
        if (valid) {
            const uint8_t *uimg = L.img + (size_t)frame * L.img_frame_stride, *bimg = L.blur + (size_t)frame * L.blur_frame_stride;
            const int upitch = L.img_pitch & 0xFFFF, bpitch = L.blur_pitch & 0xFFFF;     // known-small operands: 24-bit multiplies
            const uint32_t urow = (uint32_t)(((y - 15) & 0xFFFF) * upitch), brow = (uint32_t)(((y - 19) & 0xFFFF) * bpitch);
            uint4 ur[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const int i = l16 + 16 * k, r = (i * 43) >> 7, c3 = i - r * 3;      // i / 3 for i < 128
                ur[k] = i < 31 * 3 ? od_load16(uimg, urow + (uint32_t)(r * upitch), uxs + 16 * c3) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = l16 + 16 * k, r = (i * 43) >> 7, c3 = i - r * 3;      // i / 3 for i < 128
                if (i < 39 * 3) br[k] = od_load16(bimg, brow + (uint32_t)(r * bpitch), bxs + 16 * c3);
            }
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const int i = l16 + 16 * k, r = (i * 43) >> 7, c3 = i - r * 3;      // i / 3 for i < 128
                if (i < 31 * 3) *reinterpret_cast<uint4 *>(&up32[r * (OD_UP / 4) + 4 * c3]) = ur[k];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- IC_Angle (ORBextractor.cc:75-102)
        // lane (j = l16 & 7, parity = l16 >> 3) sweeps patch columns -15+4j .. -12+4j of every second row:
        // m10 = sum (u+16) I - 16 sum I and m01 = sum v * rowsum, both on v_dot4_u32_u8 (exact integers)
        int m10, m01 = 0;
        {
            const int j = l16 & 7, par = l16 >> 3;
            const uint32_t wts = 0x04030201u + 0x04040404u * (uint32_t)j;
            uint32_t accP = 0, accS = 0;
#pragma unroll 4
            for (int i = 0; i < 16; i++) {
                const int v = -ORB_HALF_PATCH + 2 * i + par, vc = min(v, ORB_HALF_PATCH);
                const uint32_t *rq = up32 + (vc + ORB_HALF_PATCH) * (OD_UP / 4) + j;
                uint32_t wv = __builtin_amdgcn_alignbyte(rq[1], rq[0], (uint32_t)uoff) & od_msk[(vc < 0 ? -vc : vc) * 8 + j];
                if (v > ORB_HALF_PATCH) wv = 0;
                const uint32_t sr = __builtin_amdgcn_udot4(wv, 0x01010101u, 0u, false);
                accP = __builtin_amdgcn_udot4(wv, wts, accP, false);
                accS += sr;
                m01 += v * (int)sr;
            }
            m10 = (int)accP - 16 * (int)accS;
        }
        // the moments' LDS reads are done: the blurred patch takes the region over
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = l16 + 16 * k, r = (i * 43) >> 7, c3 = i - r * 3;      // i / 3 for i < 128
                if (i < 39 * 3) *reinterpret_cast<uint4 *>(&bp32[r * (OD_BP / 4) + 4 * c3]) = br[k];
            }
        }
        m10 = row16_allreduce_add_dpp(m10); m01 = row16_allreduce_add_dpp(m01);        // the keypoint's 16 lanes = one DPP row
        const float angle = fast_atan2_deg((float)m01, (float)m10);
        // ---- steered BRIEF on the blurred level (ORBextractor.cc:106-145)
        double sd, cd;
        sincos_det((double)__fmul_rn(angle, factor_pi), &sd, &cd);
        const float a = (float)cd, b = (float)sd;
        // cvRound by the 1.5*2^23 trick: fl(r + M) holds round-half-even(r) in its low mantissa bits, i.e.
        // as_int = K + n with K = 0x4B400000; row*48 + col is then shifts and adds on the raw bits with the
        // 49*K excess folded into the patch-centre offset.  Bits are shifted in MSB-first, so t runs down.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float MAGIC = 12582912.0f;
        const uint32_t offk = (uint32_t)(19 * OD_BP + boff + 19) - 49u * 0x4B400000u;
        uint32_t bits = 0;
#pragma unroll 4
        for (int t = 15; t >= 0; t--) {
            const float4 pk = pat_t[t * 16 + l16];
            const uint32_t r0 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(pk.x, b), __fmul_rn(pk.y, a)), MAGIC));
            const uint32_t c0 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(pk.x, a), __fmul_rn(pk.y, b)), MAGIC));
            const uint32_t r1 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(pk.z, b), __fmul_rn(pk.w, a)), MAGIC));
            const uint32_t c1 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(pk.z, a), __fmul_rn(pk.w, b)), MAGIC));
            const uint32_t i0 = (r0 << 5) + (r0 << 4) + c0 + offk, i1 = (r1 << 5) + (r1 << 4) + c1 + offk;
            const uint32_t t0 = bp[i0], t1 = bp[i1];
            bits = __builtin_amdgcn_alignbit(bits, t0 - t1, 31);        // bits << 1 | (t0 < t1)
        }
        if (valid) {
            uint8_t *desc = P.lvl_desc + ((size_t)frame * P.kps_per_frame + slot) * 32;
            reinterpret_cast<uint16_t *>(desc)[l16] = (uint16_t)bits;              // tests 16*l16 .. 16*l16+15 = bytes 2*l16, 2*l16+1
            if (l16 == 0) P.lvl_angle[(size_t)frame * P.kps_per_frame + slot] = angle;
        }
    }
}

void orb_launch_orient_desc(const OrbParams &P, hipStream_t s)
{
    // one group of 4 keypoint slots per wave and trip; enough workgroups that a wave makes few trips (in-order dispatch keeps the
    // set of patches in flight spatially tight: 4096 x 8 workgroups 0.58 ms vs 0.72 ms with 256 x 8 persistent ones at batch 1024),
    // no more than the busiest XCD needs (small batches), the same number on every XCD
    const long groups_x = (long)((P.batch + 7) / 8) * ((P.kps_per_frame + 3) >> 2);      // frames are pinned to XCDs: work per XCD
    long blocks_x = (groups_x + (OD_THREADS / 64) - 1) / (OD_THREADS / 64);
    if (blocks_x > 4096) blocks_x = 4096;
    const long blocks = 8 * blocks_x;
    hipLaunchKernelGGL(k_orient_desc, dim3((unsigned)blocks), dim3(OD_THREADS), 0, s, P);
}

// ----------------------------------------------------------------------------------
// A9  output assembly: level order, pt *= scale for level>0, lapping-area split
// (ORBextractor.cc:1104-1149).  One workgroup per frame.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assemble(OrbParams P)
{
    __shared__ int wsum[8];
    __shared__ int s_cnt[ORB_MAX_LEVELS + 1];
    const int tid = threadIdx.x, frame = blockIdx.x;
    if (tid == 0) {
        int n = 0;
        for (int l = 0; l < P.nlevels; l++) { s_cnt[l] = n; n += P.lvl_count[frame * P.nlevels + l]; }
        s_cnt[P.nlevels] = n;
    }
    __syncthreads();
    const int n = s_cnt[P.nlevels];
    orbhip_keypoint *okp = P.out_kp + (size_t)frame * P.max_kp;
    uint8_t *odesc = P.out_desc + (size_t)frame * P.max_kp * 32;
    int mono_run = 0;
    for (int g0 = 0; g0 < n; g0 += 256) {
        const int g = g0 + tid;
        int lvl = 0, stereo = 0;
        float fx = 0, fy = 0;
        size_t src = 0;
        if (g < n) {
            for (int l = 1; l < P.nlevels; l++) if (g >= s_cnt[l]) lvl = l;
            const OrbLevel &L = P.lv[lvl];
            src = (size_t)frame * P.kps_per_frame + L.kp_base + (g - s_cnt[lvl]);
            const uint32_t key = P.lvl_kp[src];
            fx = (float)(ORB_KEY_X(key) + ORB_MINB);
            fy = (float)(ORB_KEY_Y(key) + ORB_MINB);
            if (lvl != 0) { fx = __fmul_rn(fx, L.scale); fy = __fmul_rn(fy, L.scale); }    // :1131-1133
            stereo = (fx >= (float)P.lap0 && fx <= (float)P.lap1) ? 1 : 0;                 // :1135
        }
        int tot;
        const int mono_before = mono_run + block_excl_scan256((g < n && !stereo) ? 1 : 0, wsum, &tot);
        if (g < n) {
            const int stereo_before = g - mono_before;
            const int slot = stereo ? (n - 1 - stereo_before) : mono_before;
            const OrbLevel &L = P.lv[lvl];
            orbhip_keypoint kp;
            kp.x = fx; kp.y = fy; kp.size = L.size; kp.angle = P.lvl_angle[src];
            kp.response = (float)ORB_KEY_S(P.lvl_kp[src]);
            kp.octave = lvl; kp.class_id = -1;
            okp[slot] = kp;
            const uint4 *d = reinterpret_cast<const uint4 *>(P.lvl_desc + src * 32);
            uint4 *o = reinterpret_cast<uint4 *>(odesc + (size_t)slot * 32);
            o[0] = d[0]; o[1] = d[1];
        }
        mono_run += tot;
    }
    if (tid == 0) { P.out_count[frame] = n; P.out_mono[frame] = mono_run; }
}

void orb_launch_assemble(const OrbParams &P, hipStream_t s)
{
    hipLaunchKernelGGL(k_assemble, dim3(P.batch), dim3(256), 0, s, P);
}
