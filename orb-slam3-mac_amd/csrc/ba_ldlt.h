// ba_ldlt.h -- the single-workgroup dense LDL^T solve shared by the local-BA tick (ba_kernels.hip) and the inertial local BA
// (iba_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
// Reduced pose system: dense LDL^T without pivoting + solve, one workgroup per graph (stands in for
// Eigen::SimplicialLDLT, linear_solver_eigen.h:94-125; fails on a zero / non-finite pivot, in which case
// x is left untouched exactly as the reference does).
// Blocked right-looking factorisation: a 32-column panel (all rows below the diagonal block) lives in
// LDS, is factored there (32 steps of LDS-only updates, the right-hand side rides along), is written
// back once, and the trailing matrix gets ONE rank-32 update per panel from LDS (4x4 register tiles).
// The trailing matrix therefore makes 9 instead of 288 round trips through L2 for n = 288.
#define BA_LDLT_MAXN 480
#define LD_NB 32
#define LD_PP 33
#ifndef LDLT_DIAG_DPP
#define LDLT_DIAG_DPP 1           // 0: the v_readlane form of the diagonal block for every panel (A/B)
#endif
static inline size_t ba_ldlt_lds_bytes(int max_n) { return sizeof(double) * ((size_t)max_n * LD_PP + 3 * (size_t)max_n + LD_NB + 32 * 32); }

// One row of a panel through the nb elimination steps of its diagonal block (right-looking LDL^T without pivoting):
//   l = a[jj] / d_jj;  a[kk] -= l * U[jj][kk]  (jj < kk <= min(r, nb-1));  a[jj] = l;  y_r -= l * y_jj.
// DIAG: the caller is the single wave that owns rows 0..nb-1 (lane = row); it produces d_jj, U[jj][.] and the final y_jj as it goes
// (lock-step execution orders the LDS traffic).  Otherwise U, d and y[0..nb) are complete and rows are independent.
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffu), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
typedef __attribute__((address_space(3))) double lds_f64;        // explicit LDS pointers: the accesses below must stay ds_read / ds_write
// 1 / d by v_rcp_f64 + two Newton steps (5 dependent instructions; the compiler's 1.0 / d adds range scaling and a fix-up, 12).
// Pivots of these systems are far from the denormal / overflow range the extra steps exist for.
__device__ __forceinline__ double ldlt_rcp(double d)
{
#pragma clang fp contract(fast)
    double r = __builtin_amdgcn_rcp(d);
    r = r + r * (1.0 - d * r);
    r = r + r * (1.0 - d * r);
    return r;
}
// FULL: the panel has all LD_NB columns (no per-step column test: straight-line code the scheduler can pipeline).
// A zero / non-finite pivot only clears *s_ok; the arithmetic runs on (its results are discarded by the caller).
template <bool DIAG, bool FULL>
__device__ __forceinline__ void ldlt_rows2(lds_f64 *P, lds_f64 *U, lds_f64 *dv, lds_f64 *yv, int r, int nb, int *s_ok)
{
#pragma clang fp contract(fast)      // a -= l * u as one v_fma_f64 (the library builds with -ffp-contract=off for the bit-exact ORB paths)
    // No per-element predicates: entries right of the diagonal (kk > r) and, in a partial last panel, columns >= nb carry
    // garbage that is never read back (keeps the unrolled code at ~3 instructions per update).
    double a[LD_NB];
    asm volatile("" : "+v"(U), "+v"(yv), "+v"(dv));       // vector base registers + immediate offsets (else ~500 hoisted scalar addresses spill)
#pragma unroll
    for (int c = 0; c < LD_NB; c++) a[c] = P[r * LD_PP + c];
    double yr = yv[r];
    bool ok = true;
#pragma unroll
    for (int jj = 0; jj < LD_NB; jj++) {
        if (!FULL && jj >= nb) continue;                    // uniform
        if (DIAG) {
            U[jj * LD_NB + r] = a[jj];                      // unscaled column jj (entry jj = the pivot)
            if (r == jj) yv[jj] = yr;                        // y_jj is final
        }
        // DIAG: pivot, y_jj and the unscaled column jj (U[jj][kk] = A[kk][jj] = lane kk's a[jj]) straight from the lanes' registers:
        // no LDS round trip on the critical path, and the broadcasts issue while the reciprocal is still in flight
        const double d = DIAG ? readlane_f64(a[jj], jj) : dv[jj];
        if (DIAG) {
            ok = ok && d != 0.0 && isfinite(d);
            if (r == jj) dv[jj] = d;
        }
        const double yj = DIAG ? readlane_f64(yr, jj) : yv[jj];
        const double l = a[jj] * ldlt_rcp(d);                // one reciprocal per column, not a division per entry
        yr -= l * yj;
#pragma unroll
        for (int kk = jj + 1; kk < LD_NB; kk++) a[kk] -= l * (DIAG ? readlane_f64(a[jj], kk) : U[jj * LD_NB + kk]);
        if (!DIAG || r > jj) a[jj] = l;
    }
    if (DIAG && !ok && r == 0) *s_ok = 0;
#pragma unroll
    for (int c = 0; c < LD_NB; c++) if (!DIAG || c <= r) P[r * LD_PP + c] = a[c];
    if (!DIAG) yv[r] = yr;
}
// ---- the FULL 32 x 32 diagonal block with row-broadcast multiply-adds (round 4).  ldlt_rows2<true> fetches U[jj][kk] = lane kk's a[jj] with
// two v_readlane per entry, and the v_readlane -> s_nop -> v_fma_f64 chain of an entry costs ~20 cycles on a wave that has its SIMD to
// itself: 496 entries x 32 steps were 36 % of the whole solve.  gfx950 can do the broadcast inside the multiply-add: v_fmac_f64 with the DPP
// control row_newbcast:k reads lane k of the lane's 16-lane ROW as its first operand, at the rate of a plain v_fma_f64
// (tools/valu_rate.hip: 4.8 cycles per wave-instruction).  The 32 rows sit on two DPP rows (lanes 0..15, 16..31), so per step ONE
// v_permlane16_swap per dword makes two registers of the pivot column: y0 = row 0's sixteen values in both rows, y1 = row 1's; the update of
// column kk is then a single instruction on y0 (kk < 16) or y1.  Same operands, same fused multiply-add per entry: bit-identical results.
template <int K>
__device__ __forceinline__ void ldlt_fmac_bcast(double &a, double y, double nl)      // a += (lane K of this lane's row of y) * nl
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(y), "v"(nl), "n"(K));
}
__device__ __forceinline__ void ldlt_row_pair(double c, double &y0, double &y1)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, c);
    unsigned clo = (unsigned)u, chi = (unsigned)(u >> 32), tlo, thi;
    // (the wait states around the swaps and in front of the DPP reads that follow are ours: the hazard recogniser does not look into inline asm)
    asm volatile("v_mov_b32 %0, %2\n v_mov_b32 %1, %3\n s_nop 1\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n s_nop 1"
                 : "=&v"(tlo), "=&v"(thi), "+v"(clo), "+v"(chi));
    y0 = __builtin_bit_cast(double, ((unsigned long long)thi << 32) | tlo);
    y1 = __builtin_bit_cast(double, ((unsigned long long)chi << 32) | clo);
}
template <int KK>
__device__ __forceinline__ void ldlt_diag_row_update(double (&a)[LD_NB], double y0, double y1, double nl)
{
    if constexpr (KK < LD_NB) {
        ldlt_fmac_bcast<(KK & 15)>(a[KK], KK < 16 ? y0 : y1, nl);
        ldlt_diag_row_update<KK + 1>(a, y0, y1, nl);
    }
}
template <int JJ>
__device__ __forceinline__ void ldlt_diag_step(double (&a)[LD_NB], double &yr, bool &ok, lds_f64 *U, double &d_own, double &y_own, int r)
{
#pragma clang fp contract(fast)
    U[JJ * LD_NB + r] = a[JJ];                               // unscaled column JJ for the rows below the block (entry JJ = the pivot)
    const double d = readlane_f64(a[JJ], JJ);
    ok = ok && d != 0.0 && isfinite(d);
    const double yj = readlane_f64(yr, JJ);
    // lane JJ keeps its pivot and its (now final) right-hand side in registers; both go to LDS once, after the last step (a masked
    // store per step meant an EXEC mask restored from a spilled SGPR pair by two v_readlane, twice per step)
    d_own = r == JJ ? d : d_own;
    y_own = r == JJ ? yr : y_own;
    double y0, y1;
    ldlt_row_pair(a[JJ], y0, y1);
    const double l = a[JJ] * ldlt_rcp(d);
    yr -= l * yj;
    ldlt_diag_row_update<JJ + 1>(a, y0, y1, -l);
    if (r > JJ) a[JJ] = l;
    if constexpr (JJ + 1 < LD_NB) ldlt_diag_step<JJ + 1>(a, yr, ok, U, d_own, y_own, r);
}
// all 32 lanes of rows 0..31 active (the caller's `tid < 32`), nb == LD_NB
__device__ __forceinline__ void ldlt_diag_full(lds_f64 *P, lds_f64 *U, lds_f64 *dv, lds_f64 *yv, int r, int *s_ok)
{
    double a[LD_NB];
    asm volatile("" : "+v"(U), "+v"(yv), "+v"(dv));
#pragma unroll
    for (int c = 0; c < LD_NB; c++) a[c] = P[r * LD_PP + c];
    double yr = yv[r];
    bool ok = true;
    double d_own = 0.0, y_own = 0.0;
    ldlt_diag_step<0>(a, yr, ok, U, d_own, y_own, r);
    dv[r] = d_own; yv[r] = y_own;
    if (!ok && r == 0) *s_ok = 0;
#pragma unroll
    for (int c = 0; c < LD_NB; c++) if (c <= r) P[r * LD_PP + c] = a[c];
}

template <bool DIAG>
__device__ __forceinline__ void ldlt_rows(lds_f64 *P, lds_f64 *U, lds_f64 *dv, lds_f64 *yv, int r, int nb, int *s_ok)
{
    ldlt_rows2<DIAG, false>(P, U, dv, yv, r, nb, s_ok);
}

// broadcast of lane OS of every quad to the quad's four lanes
template <int OS>
__device__ __forceinline__ double quad_bcast_f64(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), OS * 0x55, 0xF, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u >> 32), OS * 0x55, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// The rows below the diagonal block, FOUR lanes per row (lane s of a quad owns columns s, s+4, ..., s+28): per elimination step one
// multiply, one quad broadcast and <= 8 fused multiply-adds per lane instead of <= 31 on a single lane -- the dependency chain of a
// panel is ~4x shorter and four times as many waves hide it.  Same arithmetic per entry as ldlt_rows<false>.
template <int JJ, bool FULL>
__device__ __forceinline__ void ldlt_quad_step(double (&a)[8], double &yr, lds_f64 *U, lds_f64 *rdv, lds_f64 *yv, int s, int nb)
{
#pragma clang fp contract(fast)
    if (FULL || JJ < nb) {                                  // uniform
        constexpr int OI = JJ >> 2, OS = JJ & 3, I0 = (JJ + 1) >> 2;
        const double l = quad_bcast_f64<OS>(a[OI]) * rdv[JJ];       // the reciprocal pivots were published by the diagonal wave
        yr -= l * yv[JJ];
        if constexpr (I0 < 8) {                             // the group that holds column JJ + 1: only the lanes right of the pivot column update
            const double t = a[I0] - l * U[JJ * LD_NB + 4 * I0 + s];
            a[I0] = (4 * I0 + s > JJ) ? t : a[I0];
        }
#pragma unroll
        for (int i = I0 + 1; i < 8; i++) a[i] -= l * U[JJ * LD_NB + 4 * i + s];
        a[OI] = (s == OS) ? l : a[OI];
    }
    if constexpr (FULL && (JJ & 1)) __builtin_amdgcn_sched_barrier(0);      // two steps' loads in flight at most (else all 256 are hoisted and spill)
    if constexpr (JJ + 1 < LD_NB) ldlt_quad_step<JJ + 1, FULL>(a, yr, U, rdv, yv, s, nb);
}
template <bool FULL>
__device__ __forceinline__ void ldlt_rows_quad(lds_f64 *P, lds_f64 *U, lds_f64 *rdv, lds_f64 *yv, int r, int s, int nb)
{
    double a[8];
    asm volatile("" : "+v"(U), "+v"(yv), "+v"(rdv));
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = P[r * LD_PP + 4 * i + s];
    double yr = yv[r];
    ldlt_quad_step<0, FULL>(a, yr, U, rdv, yv, s, nb);
#pragma unroll
    for (int i = 0; i < 8; i++) P[r * LD_PP + 4 * i + s] = a[i];
    if (s == 0) yv[r] = yr;
}

// trailing update S[i][k] -= sum_c L[i][c] d_c L[k][c]  (i >= k >= p0 + nb) in TS x TS register tiles, lower-triangular tiles only,
// evenly dealt over the block.  TS = 4 when that fills the block, 2 for the small trailing matrices near the end.
template <int TS>
__device__ __forceinline__ void ldlt_trailing(double *S, int ld, lds_f64 *P, lds_f64 *dv, int p0, int nb, int m)
{
#pragma clang fp contract(fast)
    const int tid = threadIdx.x, nth = blockDim.x;
    const int m2 = m - nb, T = (m2 + TS - 1) / TS, ntri = T * (T + 1) / 2;
    for (int t = tid; t < ntri; t += nth) {
        int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while (ti * (ti + 1) / 2 > t) ti--;
        while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
        const int tk = t - ti * (ti + 1) / 2;
        const int i0 = nb + TS * ti, k0 = nb + TS * tk;
        double acc[TS][TS], sv[TS][TS];
#pragma unroll
        for (int a = 0; a < TS; a++)
#pragma unroll
            for (int b2 = 0; b2 < TS; b2++) {               // the tile's current values: in flight while the products are formed
                acc[a][b2] = 0.0;
                sv[a][b2] = (i0 + a < m && k0 + b2 < m && i0 + a >= k0 + b2) ? S[(size_t)(p0 + i0 + a) * ld + p0 + k0 + b2] : 0.0;
            }
        for (int c = 0; c < nb; c++) {
            const double dc = dv[c];
            double av[TS], bv[TS];
#pragma unroll
            for (int a = 0; a < TS; a++) {
                av[a] = (i0 + a < m) ? P[(i0 + a) * LD_PP + c] * dc : 0.0;
                bv[a] = (k0 + a < m) ? P[(k0 + a) * LD_PP + c] : 0.0;
            }
#pragma unroll
            for (int a = 0; a < TS; a++)
#pragma unroll
                for (int b2 = 0; b2 < TS; b2++) acc[a][b2] += av[a] * bv[b2];
        }
#pragma unroll
        for (int a = 0; a < TS; a++)
#pragma unroll
            for (int b2 = 0; b2 < TS; b2++)
                if (i0 + a < m && k0 + b2 < m && i0 + a >= k0 + b2)
                    S[(size_t)(p0 + i0 + a) * ld + p0 + k0 + b2] = sv[a][b2] - acc[a][b2];
    }
}

// The same trailing update on the FP64 matrix cores: one wave per 16 x 16 tile of the lower triangle, 8 x v_mfma_f64_16x16x4_f64 over the
// panel's 32 columns.  On this chip the FP64 MFMA rate equals the vector FMA rate; what it buys is LDS traffic: a lane fetches ONE double
// of L_i d and ONE of L_k per MFMA (1 byte per multiply-add) where the 4 x 4 register tiles read 8 doubles per 16 multiply-adds (4 bytes)
// -- and 16 waves x 1024 threads of those tiles were bound by the LDS pipe (measured at n = 288: 356 k of the solve's 732 k cycles).
// A[i][k] = lane (i = l & 15, k = l >> 4), B[k][j] = lane (j = l & 15, k = l >> 4); D: col j = l & 15, row i = (l >> 4) + 4 reg.
typedef double ldlt_v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ldlt_trailing_mfma(double *S, int ld, lds_f64 *P, lds_f64 *dv, int p0, int nb, int m)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int m2 = m - nb, T = (m2 + 15) >> 4, ntri = T * (T + 1) / 2;
    const int li = lane & 15, lk = lane >> 4;
    for (int t = wave; t < ntri; t += nwaves) {
        int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while (ti * (ti + 1) / 2 > t) ti--;
        while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
        const int tk = t - ti * (ti + 1) / 2;
        const int i0 = nb + 16 * ti, k0 = nb + 16 * tk;
        const int ri = min(i0 + li, m - 1), rk = min(k0 + li, m - 1);     // rows beyond the matrix: clamped reads, results never stored
        // the tile's current values (row i = lk + 4 r, column li): in flight while the products are formed
        double sv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = i0 + lk + 4 * r, k = k0 + li;
            sv[r] = (i < m && k < m && i >= k) ? S[(size_t)(p0 + i) * ld + p0 + k] : 0.0;
        }
        ldlt_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < LD_NB / 4; q++) {
            const int c = 4 * q + lk;
            const bool live = c < nb;                                     // a partial last panel: columns >= nb hold garbage
            const double a = live ? P[ri * LD_PP + c] * dv[c] : 0.0;
            const double b = live ? P[rk * LD_PP + c] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = i0 + lk + 4 * r, k = k0 + li;
            if (i < m && k < m && i >= k) S[(size_t)(p0 + i) * ld + p0 + k] = sv[r] - acc[r];
        }
    }
}

// S: [n][ld] row-major, lower triangle read and overwritten with L / d; rhs, x: n doubles in global memory (x may alias rhs); uses
// the first ba_ldlt_lds_bytes(max_ld) bytes of the kernel's dynamic LDS, max_ld >= n.  Every thread of the (<= 1024-thread, >= n - 32) block calls
// it; the return value is block-uniform: false = zero / non-finite pivot, x untouched.
#ifdef LDLT_PROF
__device__ long long g_ldlt_prof[8];          // debug build only: shader cycles of load / diag / rows / trailing / back substitution
#define LDLT_T(i) do { if (threadIdx.x == 0) { const long long t_ = clock64(); g_ldlt_prof[i] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define LDLT_T(i) do { } while (0)
#endif
__device__ __forceinline__ bool ldlt_solve_wg(double *S, int ld, int n, const double *rhs, double *x, int max_ld)
{
    // the kernel's dynamic LDS, named HERE so that every panel access stays an LDS instruction (through a pointer parameter the
    // address space is lost and the panel code turns into flat loads with a full wait after each)
#pragma clang fp contract(fast)
    extern __shared__ double lds[];
    __shared__ int s_ok;
#ifdef LDLT_PROF
    long long t_prev = clock64();
#endif
    const int tid = threadIdx.x, nth = blockDim.x;
    lds_f64 *P = (lds_f64 *)lds;                 // panel [rows][LD_PP]
    lds_f64 *y = P + (size_t)max_ld * LD_PP;     // right-hand side / solution
    lds_f64 *dval = y + max_ld;                  // pivots d_c of every column
    lds_f64 *rdval = dval + max_ld;              // their reciprocals
    lds_f64 *red = rdval + max_ld;               // [32][32] partial sums of the back substitution
    lds_f64 *U = red;                            // [32][32] unscaled columns of the current diagonal block (factorisation phase)
    for (int i = tid; i < n; i += nth) y[i] = rhs[i];
    if (tid == 0) s_ok = 1;
    __syncthreads();
    for (int p0 = 0; p0 < n; p0 += LD_NB) {
        const int nb = min(LD_NB, n - p0), m = n - p0;
        for (int idx = tid; idx < m * LD_NB; idx += nth) {
            const int r = idx >> 5, c = idx & 31;
            if (c < nb) P[r * LD_PP + c] = S[(size_t)(p0 + r) * ld + p0 + c];
        }
        __syncthreads();
        LDLT_T(0);
        // (1) the nb x nb diagonal block: ONE wave, one row per lane, the row in registers, no block barriers.  Step jj publishes the
        //     still unscaled column jj (U[jj][r] = A[r][jj]) in LDS; every lane reads it back as wave-wide broadcasts.
        if (tid < nb) {
            // (measured alternatives: the column through LDS with look-ahead publishing of column jj + 1: 29.6 k cycles per 32 x 32 block
            //  against 23.5 k for the register broadcasts below; round 4: steps 0..23 reading the published column back as wave-wide
            //  broadcast reads and only the last eight by v_readlane: 29.5 k -- the write -> read -> multiply-add turnaround of a step
            //  outlasts its 2 x (31 - jj) v_readlane; straight-line FULL code: the scheduler hoists every step's broadcasts
            //  and spills ~700 SGPRs)
            if (nb == LD_NB && LDLT_DIAG_DPP) ldlt_diag_full(P, U, dval + p0, y + p0, tid, &s_ok);     // row-broadcast multiply-adds (full panels)
            else ldlt_rows2<true, false>(P, U, dval + p0, y + p0, tid, nb, &s_ok);
            rdval[p0 + tid] = ldlt_rcp(dval[p0 + tid]);          // same lane wrote dval[p0 + tid]
        }
        __syncthreads();
        LDLT_T(1);
        if (!s_ok) break;
        // (2) the rows below the block: independent forward substitutions against U / d, four lanes per row
        // (the straight-line FULL instantiation makes the register allocator hoist and spill the U loads: not used)
        for (int q = tid >> 2; q < m - nb; q += nth >> 2) ldlt_rows_quad<false>(P, U, rdval + p0, y + p0, nb + q, tid & 3, nb);
        __syncthreads();
        LDLT_T(2);
        if (!s_ok) break;
        // write the factored panel back: L below the diagonal, d on it
        for (int idx = tid; idx < m * LD_NB; idx += nth) {
            const int r = idx >> 5, c = idx & 31;
            if (c < nb && r >= c) S[(size_t)(p0 + r) * ld + p0 + c] = (r == c) ? dval[p0 + c] : P[r * LD_PP + c];
        }
        if (m > nb) {
            const int T4 = (m - nb + 3) >> 2;
#ifdef LDLT_VECTOR_TRAILING
            if (T4 * (T4 + 1) / 2 >= nth) ldlt_trailing<4>(S, ld, P, dval + p0, p0, nb, m);
            else ldlt_trailing<2>(S, ld, P, dval + p0, p0, nb, m);
#else
            (void)T4;
            ldlt_trailing_mfma(S, ld, P, dval + p0, p0, nb, m);
#endif
        }
        __syncthreads();
        LDLT_T(3);
    }
    __syncthreads();
    if (!s_ok) return false;                             // x untouched (as the reference on failure)
    // y <- D^-1 y, then L^T x = y panel by panel from the bottom
    for (int i = tid; i < n; i += nth) y[i] *= rdval[i];
    __syncthreads();
    const int last_p0 = ((n - 1) / LD_NB) * LD_NB;
    for (int p0 = last_p0; p0 >= 0; p0 -= LD_NB) {
        const int nb = min(LD_NB, n - p0), m = n - p0;
        // contributions of the rows below the diagonal block: t_c = sum_{r >= nb} L[p0+r][p0+c] x[p0+r]
        {
            const int c = tid & 31;                         // 32 columns x 32 row groups (1024 threads: one group each; fewer threads take several)
            for (int rg = tid >> 5; rg < 32; rg += nth >> 5) {
                double part = 0.0;
                if (c < nb)
                    for (int r = nb + rg; r < m; r += 32) part += S[(size_t)(p0 + r) * ld + p0 + c] * y[p0 + r];
                red[rg * 32 + c] = part;
            }
        }
        for (int idx = tid; idx < nb * LD_NB; idx += nth) {   // diagonal block of L into LDS
            const int r = idx >> 5, c = idx & 31;
            if (c < nb) P[r * LD_PP + c] = S[(size_t)(p0 + r) * ld + p0 + c];
        }
        __syncthreads();
        if (tid < LD_NB) {                                    // 32 lanes: lane c holds y_c and column c of the block's L; x_jj by readlane
            double yc = 0.0, col[LD_NB];
            if (tid < nb) {
                double t = 0.0;
                for (int rg = 0; rg < 32; rg++) t += red[rg * 32 + tid];
                yc = y[p0 + tid] - t;
            }
#pragma unroll
            for (int jj = 0; jj < LD_NB; jj++) col[jj] = (jj > tid && jj < nb) ? P[jj * LD_PP + tid] : 0.0;
#pragma unroll
            for (int jj = LD_NB - 1; jj >= 1; jj--) yc -= col[jj] * readlane_f64(yc, jj);      // col is zero for jj <= c and jj >= nb
            if (tid < nb) y[p0 + tid] = yc;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += nth) x[i] = y[i];
    LDLT_T(4);
    return true;
}
