// Wave64 scans and reductions on the DPP data path (gfx9 family: row_shr within rows of 16 lanes, then row_bcast:15 / row_bcast:31
// across rows) instead of ds_bpermute round trips: hipcc lowers __shfl_xor / __shfl_up to ds_bpermute_b32 (LDS crossbar latency per
// step), which is what the latency-bound kernels (sequential query loops, block reductions) were waiting on.
// All functions must be called by all 64 lanes of the wave (EXEC full); inactive data lanes contribute the identity.
#ifndef ORBHIP_WAVE_DPP_H
#define ORBHIP_WAVE_DPP_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ORB_DPP_ROW_SHR(n) (0x110 + (n))
#define ORB_DPP_ROW_BCAST15 0x142
#define ORB_DPP_ROW_BCAST31 0x143

// value of lane (l - shift) inside the row / the broadcast lane, or `ident` where there is no source lane
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int ident, int v)
{
    return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xF, false);
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ int wave_scan_add_dpp(int v)
{
    v += dpp_mov<ORB_DPP_ROW_SHR(1), 0xF>(0, v);
    v += dpp_mov<ORB_DPP_ROW_SHR(2), 0xF>(0, v);
    v += dpp_mov<ORB_DPP_ROW_SHR(4), 0xF>(0, v);
    v += dpp_mov<ORB_DPP_ROW_SHR(8), 0xF>(0, v);
    v += dpp_mov<ORB_DPP_ROW_BCAST15, 0xA>(0, v);
    v += dpp_mov<ORB_DPP_ROW_BCAST31, 0xC>(0, v);
    return v;
}
__device__ __forceinline__ int wave_sum_dpp(int v) { return __builtin_amdgcn_readlane(wave_scan_add_dpp(v), 63); }

__device__ __forceinline__ int wave_max_dpp(int v)       // identity INT_MIN
{
    const int I = (int)0x80000000;
    v = max(v, dpp_mov<ORB_DPP_ROW_SHR(1), 0xF>(I, v));
    v = max(v, dpp_mov<ORB_DPP_ROW_SHR(2), 0xF>(I, v));
    v = max(v, dpp_mov<ORB_DPP_ROW_SHR(4), 0xF>(I, v));
    v = max(v, dpp_mov<ORB_DPP_ROW_SHR(8), 0xF>(I, v));
    v = max(v, dpp_mov<ORB_DPP_ROW_BCAST15, 0xA>(I, v));
    v = max(v, dpp_mov<ORB_DPP_ROW_BCAST31, 0xC>(I, v));
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t v)     // identity 0xFFFFFFFF
{
    const int I = -1;
#define ORB_STEP(CTRL, RM) v = min(v, (uint32_t)dpp_mov<CTRL, RM>(I, (int)v))
    ORB_STEP(ORB_DPP_ROW_SHR(1), 0xF); ORB_STEP(ORB_DPP_ROW_SHR(2), 0xF); ORB_STEP(ORB_DPP_ROW_SHR(4), 0xF); ORB_STEP(ORB_DPP_ROW_SHR(8), 0xF);
    ORB_STEP(ORB_DPP_ROW_BCAST15, 0xA); ORB_STEP(ORB_DPP_ROW_BCAST31, 0xC);
#undef ORB_STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// the two smallest values of the wave's (k1 <= k2) pairs; identity (0xFFFFFFFF, 0xFFFFFFFF)
__device__ __forceinline__ void wave_min2_u32_dpp(uint32_t &k1, uint32_t &k2)
{
    const int I = -1;
#define ORB_STEP(CTRL, RM) do { const uint32_t o1 = (uint32_t)dpp_mov<CTRL, RM>(I, (int)k1), o2 = (uint32_t)dpp_mov<CTRL, RM>(I, (int)k2); \
                                k2 = min(max(k1, o1), min(k2, o2)); k1 = min(k1, o1); } while (0)
    ORB_STEP(ORB_DPP_ROW_SHR(1), 0xF); ORB_STEP(ORB_DPP_ROW_SHR(2), 0xF); ORB_STEP(ORB_DPP_ROW_SHR(4), 0xF); ORB_STEP(ORB_DPP_ROW_SHR(8), 0xF);
    ORB_STEP(ORB_DPP_ROW_BCAST15, 0xA); ORB_STEP(ORB_DPP_ROW_BCAST31, 0xC);
#undef ORB_STEP
    k1 = (uint32_t)__builtin_amdgcn_readlane((int)k1, 63);
    k2 = (uint32_t)__builtin_amdgcn_readlane((int)k2, 63);
}

// sum of a double over the wave, FIXED association (the scan's tree), result uniform
__device__ __forceinline__ double wave_sum_f64_dpp(double v)
{
#define ORB_STEP(CTRL, RM) do { const unsigned long long u = __builtin_bit_cast(unsigned long long, v); \
        const unsigned lo = (unsigned)dpp_mov<CTRL, RM>(0, (int)(u & 0xffffffffu)), hi = (unsigned)dpp_mov<CTRL, RM>(0, (int)(u >> 32)); \
        v += __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); } while (0)
    ORB_STEP(ORB_DPP_ROW_SHR(1), 0xF); ORB_STEP(ORB_DPP_ROW_SHR(2), 0xF); ORB_STEP(ORB_DPP_ROW_SHR(4), 0xF); ORB_STEP(ORB_DPP_ROW_SHR(8), 0xF);
    ORB_STEP(ORB_DPP_ROW_BCAST15, 0xA); ORB_STEP(ORB_DPP_ROW_BCAST31, 0xC);
#undef ORB_STEP
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffu), 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// sum of a double over each row of 16 lanes, every lane of the row gets a full sum (rotations; each lane's association is fixed)
__device__ __forceinline__ double row16_allreduce_f64_dpp(double v)
{
#define ORB_STEP(N) do { const unsigned long long u = __builtin_bit_cast(unsigned long long, v); \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), 0x120 + (N), 0xF, 0xF, false), \
                       hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u >> 32), 0x120 + (N), 0xF, 0xF, false); \
        v += __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); } while (0)
    ORB_STEP(8); ORB_STEP(4); ORB_STEP(2); ORB_STEP(1);
#undef ORB_STEP
    return v;
}

// sum over each row of 16 lanes, result in all 16 lanes (rotations inside the row)
__device__ __forceinline__ int row16_allreduce_add_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x120 + 8, 0xF, 0xF, false);       // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x120 + 4, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x120 + 2, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x120 + 1, 0xF, 0xF, false);
    return v;
}
#endif
