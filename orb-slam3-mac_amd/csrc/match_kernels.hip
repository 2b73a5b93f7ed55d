// match_kernels.hip -- Hamming matching kernels (reference src/ORBmatcher.cc, src/Frame.cc).
//   M1  ORBmatcher::DescriptorDistance          ORBmatcher.cc:2353-2369
//   M2  all-pairs 2-NN + ratio                  Frame.cc:43,1146-1153 (cv::BFMatcher knnMatch k=2)
//   M3  Frame grid + GetFeaturesInArea          Frame.cc:377-408,645-726
//   M4  SearchForInitialization                 ORBmatcher.cc:710-825, ComputeThreeMaxima :2307-2348
// Integer popcount work: v_bcnt_u32_b32 on 8 dwords per pair; train descriptors are
// staged in LDS and read as wave-uniform (broadcast) 128-bit loads.
#include "orb_internal.h"
#include <climits>
#include <cstring>
#include <string>

struct orbhip_ctx;
hipStream_t orbhip_ctx_stream_internal(orbhip_ctx *c);
int orbhip_ctx_device_internal(orbhip_ctx *c);

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
    int best = INT_MAX, second = INT_MAX, bi = -1, si = -1;
    for (int t0 = 0; t0 < nb; t0 += BF_TILE) {
        const int tn = min(BF_TILE, nb - t0);
        __syncthreads();
        for (int i = tid; i < tn * 2; i += 256) tile[i] = B[2 * t0 + i];
        __syncthreads();
        for (int j = 0; j < tn; j++) {
            const int d = hamming256(a0, a1, tile[2 * j], tile[2 * j + 1]);
            if (d < best) { second = best; si = bi; best = d; bi = t0 + j; }
            else if (d < second) { second = d; si = t0 + j; }
        }
    }
    if (q < na) {
        const size_t o = ((size_t)pair * max_n + q) * 2;
        idx2[o] = bi; idx2[o + 1] = si; dist2[o] = best; dist2[o + 1] = second;
        // Frame.cc:1153: (*it).size() >= 2 && (*it)[0].distance < (*it)[1].distance * 0.7  (float < float*double)
        accept[(size_t)pair * max_n + q] = (si >= 0 && (double)(float)best < (double)(float)second * ratio) ? 1 : 0;
    }
}

extern "C" int orbhip_match_bf2nn_device(orbhip_ctx *ctx, const uint8_t *d_descA, const int32_t *d_nA, size_t strideA,
                                         const uint8_t *d_descB, const int32_t *d_nB, size_t strideB, int pairs,
                                         int max_n, double ratio, int32_t *d_idx2, int32_t *d_dist2, uint8_t *d_accept)
{
    if (!ctx || !d_descA || !d_descB || !d_nA || !d_nB || pairs <= 0 || max_n <= 0 || !d_idx2 || !d_dist2 || !d_accept)
        return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    dim3 grid((max_n + 255) / 256, pairs);
    hipLaunchKernelGGL(k_bf2nn, grid, dim3(256), 0, orbhip_ctx_stream_internal(ctx), d_descA, d_nA, strideA, d_descB, d_nB,
                       strideB, max_n, ratio, d_idx2, d_dist2, d_accept);
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}
