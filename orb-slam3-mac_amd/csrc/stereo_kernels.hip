// Frame::ComputeStereoMatches (reference src/Frame.cc:802-980), batched over frames.
//
// Every left keypoint is independent, so the row table of the reference (vRowIndices, :811-830) is not
// materialised: a right keypoint iR is a candidate of left keypoint iL iff floor(yR - r) <= (int)vL <= ceil(yR + r)
// with r = 2*scale[octave_R] -- exactly the rows the reference pushes iR into -- and scanning iR in index order
// reproduces the row list's order (first best wins, :878-882).
//   k_stereo_match   one wave per 64 left keypoints: (1) lane = left keypoint, right keypoints streamed through
//                    LDS, Hamming only for the few that pass the row / octave / disparity gates; (2) the matched
//                    lanes are refined one after the other by the whole wave: 11x11 patch and 11x21 strip staged in
//                    LDS, 11 SADs, parabola, disparity gates (:890-963).
//   k_stereo_median  one workgroup per frame: 1.5*1.4*median SAD filter (:966-980) by rank counting.
// Columns left of the image (the strip reaches scaleduR0-10) read the reflect-101 padding of mvImagePyramid in
// the reference; here the level images are un-padded and the column index is reflected instead (same bytes).
#include <hip/hip_runtime.h>
#include <limits.h>
#include "orb_internal.h"
#include "wave_dpp.h"

#define ST_TH_HIGH 100        // ORBmatcher::TH_HIGH (ORBmatcher.cc:40)
#define ST_TH_ORB 75          // (TH_HIGH + TH_LOW) / 2 (Frame.cc:807)

__device__ __forceinline__ int st_hamming(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__global__ __launch_bounds__(64) void k_stereo_match(StereoArgs A)
{
    __shared__ float rx[64];
    __shared__ int16_t rmin[64], rmax[64];
    __shared__ int8_t roct[64];
    __shared__ uint8_t pl[11 * 11];
    __shared__ uint8_t pr[11 * 21];
    const int f = blockIdx.y, lane = threadIdx.x;
    const int nL = A.nL[f], nR = A.nR[f];
    const int iL = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= nL) return;
    const orbhip_keypoint *kpL = A.kpL + (size_t)f * A.max_kp, *kpR = A.kpR + (size_t)f * A.max_kp;
    const uint4 *dL = reinterpret_cast<const uint4 *>(A.descL + (size_t)f * A.max_kp * 32);
    const uint4 *dR = reinterpret_cast<const uint4 *>(A.descR + (size_t)f * A.max_kp * 32);
    const bool valid = iL < nL;
    float uL = 0, vL = 0; int levelL = 0;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (valid) { const orbhip_keypoint k = kpL[iL]; uL = k.x; vL = k.y; levelL = k.octave; a0 = dL[2 * iL]; a1 = dL[2 * iL + 1]; }
    const float maxD = __fdiv_rn(A.mbf, A.mb);                        // minZ = mb, maxD = mbf/minZ, minD = 0 (Frame.cc:833-835)
    const int row = (int)vL;                                          // vRowIndices[vL] (Frame.cc:848)
    const float minU = __fsub_rn(uL, maxD), maxU = uL;
    const bool searching = valid && row >= 0 && row < A.rows0 && !(maxU < 0);
    int best = ST_TH_HIGH, bestR = 0;
    // ---- (1) row-band candidate search (Frame.cc:865-887)
    for (int t0 = 0; t0 < nR; t0 += 64) {
        __syncthreads();
        if (t0 + lane < nR) {
            const orbhip_keypoint k = kpR[t0 + lane];
            const float r = __fmul_rn(2.0f, A.lv[k.octave].scale);
            rx[lane] = k.x;
            rmax[lane] = (int16_t)min(max((int)ceilf(__fadd_rn(k.y, r)), -32768), 32767);
            rmin[lane] = (int16_t)min(max((int)floorf(__fsub_rn(k.y, r)), -32768), 32767);
            roct[lane] = (int8_t)k.octave;
        }
        __syncthreads();
        const int tn = min(64, nR - t0);
        if (searching) {
            for (int j = 0; j < tn; j++) {
                const int o = roct[j];
                if (row < rmin[j] || row > rmax[j] || o < levelL - 1 || o > levelL + 1) continue;
                const float uR = rx[j];
                if (uR >= minU && uR <= maxU) {
                    const int dist = st_hamming(a0, a1, dR[2 * (t0 + j)], dR[2 * (t0 + j) + 1]);
                    if (dist < best) { best = dist; bestR = t0 + j; }
                }
            }
        }
    }
    // ---- (2) sub-pixel refinement by correlation (Frame.cc:890-963), one matched keypoint at a time, whole wave
    float out_u = -1.0f, out_d = -1.0f; int out_sad = -1;
    unsigned long long todo = __ballot(searching && best < ST_TH_ORB);
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        // lane l's values as scalars (v_readlane), the SAD sums on the DPP path: no ds_bpermute in this per-keypoint chain
        const float uLl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, uL), l)), vLl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vL), l));
        const int lev = __builtin_amdgcn_readlane(levelL, l), bR = __builtin_amdgcn_readlane(bestR, l);
        const float uR0 = kpR[bR].x;
        const StereoLevel &SL = A.lv[lev];
        const float sfac = SL.inv_scale;
        const float scaleduL = roundf(__fmul_rn(uLl, sfac)), scaledvL = roundf(__fmul_rn(vLl, sfac)), scaleduR0 = roundf(__fmul_rn(uR0, sfac));
        const float iniu = scaleduR0, endu = __fadd_rn(scaleduR0, 11.0f);          // scaleduR0+L-w, scaleduR0+L+w+1 (L = w = 5)
        if (iniu < 0 || endu >= (float)SL.wR) continue;                              // Frame.cc:915-916
        const int cu = (int)scaleduL, cv = (int)scaledvL, cr = (int)scaleduR0;
        const uint8_t *IL = SL.imgL + (size_t)f * SL.fsL, *IR = SL.imgR + (size_t)f * SL.fsR;
        __syncthreads();
        for (int e = lane; e < 121; e += 64) { const int dy = e / 11, dx = e - dy * 11; pl[e] = IL[(size_t)(cv - 5 + dy) * SL.pitchL + (cu - 5 + dx)]; }
        for (int e = lane; e < 231; e += 64) {
            const int dy = e / 21, dx = e - dy * 21;
            int c = cr - 10 + dx;
            if (c < 0) c = -c;                                                       // BORDER_REFLECT_101 padding of mvImagePyramid
            pr[e] = IR[(size_t)(cv - 5 + dy) * SL.pitchR + c];
        }
        __syncthreads();
        int s[11];
#pragma unroll
        for (int k = 0; k < 11; k++) s[k] = 0;
        const int cL = pl[5 * 11 + 5];
        for (int e = lane; e < 121; e += 64) {
            const int dy = e / 11, dx = e - dy * 11;
            const int a = (int)pl[e] - cL;
#pragma unroll
            for (int k = 0; k < 11; k++) {                                           // incR = k - 5
                const int b = (int)pr[dy * 21 + dx + k] - (int)pr[5 * 21 + 5 + k];
                s[k] += abs(a - b);
            }
        }
#pragma unroll
        for (int k = 0; k < 11; k++) s[k] = wave_sum_dpp(s[k]);
        int bd = INT_MAX, binc = 0;
#pragma unroll
        for (int k = 0; k < 11; k++) if ((float)s[k] < (float)bd) { bd = s[k]; binc = k - 5; }   // float dist < int bestDist (Frame.cc:929)
        if (binc == -5 || binc == 5) continue;                                       // Frame.cc:938-939
        float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
        for (int k = 1; k < 10; k++) if (k - 5 == binc) { d1 = (float)s[k - 1]; d2 = (float)s[k]; d3 = (float)s[k + 1]; }
        const float deltaR = __fdiv_rn(__fsub_rn(d1, d3), __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
        if (deltaR < -1 || deltaR > 1) continue;
        float bestuR = __fmul_rn(SL.scale, __fadd_rn(__fadd_rn(scaleduR0, (float)binc), deltaR));
        float disparity = __fsub_rn(uLl, bestuR);
        if (disparity >= 0 && disparity < maxD) {                                    // Frame.cc:953-963
            if (disparity <= 0) { disparity = (float)0.01; bestuR = (float)((double)uLl - 0.01); }
            if (lane == l) { out_d = __fdiv_rn(A.mbf, disparity); out_u = bestuR; out_sad = bd; }
        }
    }
    if (valid) {
        const size_t o = (size_t)f * A.max_kp + iL;
        A.u_right[o] = out_u; A.depth[o] = out_d; A.sad[o] = out_sad;
    }
}

// 1.5 * 1.4 * median filter (Frame.cc:966-980): keys (SAD << 16 | iL) are the sorted pairs of the reference.
__global__ __launch_bounds__(256) void k_stereo_median(StereoArgs A)
{
    extern __shared__ uint32_t keys[];
    __shared__ int s_n, s_med, s_removed;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int nL = A.nL[f];
    const int32_t *sad = A.sad + (size_t)f * A.max_kp;
    if (tid == 0) { s_n = 0; s_med = -1; s_removed = 0; }
    __syncthreads();
    for (int i0 = 0; i0 < nL; i0 += 256) {
        const int i = i0 + tid;
        const int sv = i < nL ? sad[i] : -1;
        if (sv >= 0) keys[atomicAdd(&s_n, 1)] = ((uint32_t)sv << 16) | (uint32_t)i;   // order irrelevant: ranks are counted
    }
    __syncthreads();
    const int V = s_n;
    if (V == 0) { if (tid == 0 && A.n_kept) A.n_kept[f] = 0; return; }
    for (int k = tid; k < V; k += 256) {
        const uint32_t me = keys[k];
        int rank = 0;
        for (int j = 0; j < V; j++) rank += keys[j] < me;
        if (rank == V / 2) s_med = (int)(me >> 16);                                    // vDistIdx[size/2].first
    }
    __syncthreads();
    const float thDist = __fmul_rn(__fmul_rn(1.5f, 1.4f), (float)s_med);
    int removed = 0;
    for (int k = tid; k < V; k += 256) {
        const uint32_t me = keys[k];
        if (!((float)(me >> 16) < thDist)) {
            const size_t o = (size_t)f * A.max_kp + (me & 0xFFFFu);
            A.u_right[o] = -1.0f; A.depth[o] = -1.0f; A.sad[o] = -1;
            removed++;
        }
    }
    if (removed) atomicAdd(&s_removed, removed);
    __syncthreads();
    if (tid == 0 && A.n_kept) A.n_kept[f] = V - s_removed;
}

void orb_launch_stereo(const StereoArgs &A, hipStream_t s)
{
    hipLaunchKernelGGL(k_stereo_match, dim3((A.max_kp + 63) / 64, A.batch), dim3(64), 0, s, A);
    hipLaunchKernelGGL(k_stereo_median, dim3(A.batch), dim3(256), sizeof(uint32_t) * (size_t)A.max_kp, s, A);
}
