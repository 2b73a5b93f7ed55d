// Issue-rate probe for the vector instructions k_fast_cells leans on (gfx950): every lane runs 8 independent chains of ONE instruction,
// 4 waves per SIMD, all CUs; prints cycles per wave-instruction per SIMD.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define DEF(name, INSN)                                                                                        \
    __global__ __launch_bounds__(256) void name(uint32_t *out, uint32_t seed)                                     \
    {                                                                                                             \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;  \
        const uint32_t b = seed * 2654435761u, c = seed ^ 0x01020304u;                                            \
        for (int i = 0; i < ITER; i++) {                                                                          \
            asm volatile(INSN(%0) "\n" INSN(%1) "\n" INSN(%2) "\n" INSN(%3) "\n" INSN(%4) "\n" INSN(%5) "\n" INSN(%6) "\n" INSN(%7)       \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));             \
        }                                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                               \
    }
#define I_PKMIN(r) "v_pk_min_i16 " #r ", " #r ", %8"
#define I_PKMAX(r) "v_pk_max_i16 " #r ", " #r ", %8"
#define I_PKSUB(r) "v_pk_sub_i16 " #r ", " #r ", %8"
#define I_PKADDU(r) "v_pk_add_u16 " #r ", " #r ", %8"
#define I_ALIGN(r) "v_alignbit_b32 " #r ", " #r ", %8, 16"
#define I_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %9"
#define I_MIN32(r) "v_min_i32 " #r ", " #r ", %8"
#define I_ADD32(r) "v_add_u32 " #r ", " #r ", %8"
#define I_AND(r) "v_and_b32 " #r ", " #r ", %8"
#define I_MIN3(r) "v_min3_i32 " #r ", " #r ", %8, %9"
#define I_MAX3I16(r) "v_max3_i16 " #r ", " #r ", %8, %9"
#define I_MINI16(r) "v_min_i16 " #r ", " #r ", %8"
#define I_SAD(r) "v_sad_u8 " #r ", " #r ", %8, %9"
#define I_LSHL(r) "v_lshlrev_b32 " #r ", 1, " #r
#define I_DOT4(r) "v_dot4_u32_u8 " #r ", " #r ", %8, %9"
#define I_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9"
#define I_PKMAD(r) "v_pk_mad_i16 " #r ", " #r ", %8, %9"
#define I_BFE(r) "v_bfe_u32 " #r ", " #r ", 3, 8"
DEF(k_pkmin, I_PKMIN) DEF(k_pkmax, I_PKMAX) DEF(k_pksub, I_PKSUB) DEF(k_pkaddu, I_PKADDU) DEF(k_align, I_ALIGN) DEF(k_perm, I_PERM)
DEF(k_min32, I_MIN32) DEF(k_add32, I_ADD32) DEF(k_and, I_AND) DEF(k_min3, I_MIN3) DEF(k_max3i16, I_MAX3I16) DEF(k_mini16, I_MINI16)
DEF(k_sad, I_SAD) DEF(k_lshl, I_LSHL) DEF(k_dot4, I_DOT4) DEF(k_mad24, I_MAD24) DEF(k_pkmad, I_PKMAD) DEF(k_bfe, I_BFE)

// co-issue probes: the same 8 vector instructions per iteration with 8 independent scalar ones / 4 LDS reads interleaved
__global__ __launch_bounds__(256) void k_mix_salu(uint32_t *out, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const uint32_t b = seed * 2654435761u;
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_pk_min_i16 %0, %0, %12\n s_add_u32 %8, %8, 3\n v_pk_min_i16 %1, %1, %12\n s_xor_b32 %9, %9, 5\n v_pk_min_i16 %2, %2, %12\n s_add_u32 %10, %10, 7\n"
                     "v_pk_min_i16 %3, %3, %12\n s_xor_b32 %11, %11, 9\n v_pk_min_i16 %4, %4, %12\n s_add_u32 %8, %8, 11\n v_pk_min_i16 %5, %5, %12\n s_xor_b32 %9, %9, 13\n"
                     "v_pk_min_i16 %6, %6, %12\n s_add_u32 %10, %10, 15\n v_pk_min_i16 %7, %7, %12\n s_xor_b32 %11, %11, 17"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b) : "scc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3;
}
__global__ __launch_bounds__(256) void k_salu_only(uint32_t *out, uint32_t seed)
{
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    for (int i = 0; i < ITER; i++) {
        asm volatile("s_add_u32 %0, %0, 3\n s_xor_b32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_xor_b32 %3, %3, 9\n s_add_u32 %0, %0, 11\n s_xor_b32 %1, %1, 13\n s_add_u32 %2, %2, 15\n s_xor_b32 %3, %3, 17"
                     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = s0 ^ s1 ^ s2 ^ s3;
}
__global__ __launch_bounds__(256) void k_mix_lds(uint32_t *out, uint32_t seed)
{
    __shared__ uint32_t sh[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sh[i] = i * seed;
    __syncthreads();
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const uint32_t b = seed * 2654435761u;
    uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    const uint32_t addr = (uint32_t)(uintptr_t)sh + 4 * threadIdx.x;
    for (int i = 0; i < ITER; i++) {
        asm volatile("ds_read_b32 %8, %13\n v_pk_min_i16 %0, %0, %12\n v_pk_min_i16 %1, %1, %12\n ds_read_b32 %9, %13 offset:1024\n v_pk_min_i16 %2, %2, %12\n v_pk_min_i16 %3, %3, %12\n"
                     "ds_read_b32 %10, %13 offset:2048\n v_pk_min_i16 %4, %4, %12\n v_pk_min_i16 %5, %5, %12\n ds_read_b32 %11, %13 offset:3072\n v_pk_min_i16 %6, %6, %12\n v_pk_min_i16 %7, %7, %12\n s_waitcnt lgkmcnt(0)"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3) : "v"(b), "v"(addr) : "memory");
        a0 ^= l0 & 1; a1 ^= l1 & 1; a2 ^= l2 & 1; a3 ^= l3 & 1;
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// cross-lane probes (round 4): what a wave-wide broadcast costs.  v_readlane writes an SGPR: 8 of them, then one consumer each
__global__ __launch_bounds__(256) void k_readlane(uint32_t *out, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_readlane_b32 %8, %0, 3\n v_readlane_b32 %9, %1, 5\n v_readlane_b32 %10, %2, 7\n v_readlane_b32 %11, %3, 9\n"
                     "v_readlane_b32 %12, %4, 11\n v_readlane_b32 %13, %5, 13\n v_readlane_b32 %14, %6, 15\n v_readlane_b32 %15, %7, 17"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                       "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7));
        a0 += s0; a1 += s1; a2 += s2; a3 += s3; a4 += s4; a5 += s5; a6 += s6; a7 += s7;          // 8 v_add_u32 with an SGPR operand
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
#define I_DPPNB(r) "v_mov_b32_dpp " #r ", " #r " row_newbcast:5 row_mask:0xf bank_mask:0xf"
#define I_DPPQP(r) "v_mov_b32_dpp " #r ", " #r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define I_DPPSHR(r) "v_mov_b32_dpp " #r ", " #r " row_shr:1 row_mask:0xf bank_mask:0xf"
DEF(k_dpp_newbcast, I_DPPNB) DEF(k_dpp_quadperm, I_DPPQP) DEF(k_dpp_rowshr, I_DPPSHR)
__global__ __launch_bounds__(256) void k_fma64(uint32_t *out, uint32_t seed)
{
    double a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const double b = 1.0000001, c = 1e-9;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ __launch_bounds__(256) void k_fma64_dpp(uint32_t *out, uint32_t seed)
{
    double a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const double b = 1.0000001, c = 1e-9;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f64_dpp %2, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f64_dpp %4, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f64_dpp %6, %8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:10 row_mask:0xf bank_mask:0xf"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ __launch_bounds__(256) void k_bpermute(uint32_t *out, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const uint32_t idx = 4 * 7;
    for (int i = 0; i < ITER; i++) {
        asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                     "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(idx) : "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * 4;            // 4 workgroups x 4 waves per CU = 4 waves per SIMD
    uint32_t *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    struct { const char *name; void (*fn)(uint32_t *, uint32_t); } ks[] = {
        {"v_pk_min_i16", k_pkmin}, {"v_pk_max_i16", k_pkmax}, {"v_pk_sub_i16", k_pksub}, {"v_pk_add_u16", k_pkaddu}, {"v_alignbit_b32", k_align},
        {"v_perm_b32", k_perm}, {"v_min_i32", k_min32}, {"v_add_u32", k_add32}, {"v_and_b32", k_and}, {"v_min3_i32", k_min3}, {"v_max3_i16", k_max3i16},
        {"v_min_i16", k_mini16}, {"v_sad_u8", k_sad}, {"v_lshlrev_b32", k_lshl}, {"v_dot4_u32_u8", k_dot4}, {"v_mad_u32_u24", k_mad24}, {"v_pk_mad_i16", k_pkmad},
        {"v_bfe_u32", k_bfe}, {"8 v_readlane + 8 v_add(sgpr) (per 16)", k_readlane}, {"v_mov_dpp row_newbcast", k_dpp_newbcast}, {"v_mov_dpp quad_perm", k_dpp_quadperm}, {"v_mov_dpp row_shr", k_dpp_rowshr}, {"v_fma_f64", k_fma64}, {"v_fmac_f64_dpp row_newbcast", k_fma64_dpp}, {"ds_bpermute_b32 (8 + wait)", k_bpermute}, {"8 v_pk_min + 8 SALU (per 8 vector)", k_mix_salu}, {"8 SALU only (per 8 scalar)", k_salu_only}, {"8 v_pk_min + 4 ds_read + 4 v_and/v_xor (per 8)", k_mix_lds}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("%d CUs, clock attribute %.0f MHz; cycles per wave-instruction per SIMD assume that clock\n", cus, clk_khz / 1e3);
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 2u + r);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = 5.0 * ITER * 8 * 4;                       // 4 waves per SIMD
        const double cyc = ms * 1e-3 * clk_khz * 1e3 / insts_per_simd;
        printf("%-16s %.3f ms  -> %.2f cycles per wave-instruction\n", k.name, ms / 5, cyc);
    }
    return 0;
}
