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
static inline size_t ba_ldlt_lds_bytes(int max_n) { return sizeof(double) * ((size_t)max_n * LD_PP + 2 * (size_t)max_n + LD_NB + 32 * 32); }

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
template <bool DIAG>
__device__ __forceinline__ void ldlt_rows(double *P, double *U, double *dv, double *yv, int r, int nb, int *s_ok)
{
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
        if (jj >= nb || !ok) continue;                      // uniform
        if (DIAG) {
            U[jj * LD_NB + r] = a[jj];                      // unscaled column jj (entry jj = the pivot)
            if (r == jj) yv[jj] = yr;                        // y_jj is final
        }
        // DIAG: pivot and y_jj straight from lane jj's registers (no LDS round trip on the critical path)
        const double d = DIAG ? readlane_f64(a[jj], jj) : dv[jj];
        if (DIAG) {
            if (d == 0.0 || !isfinite(d)) { ok = false; if (r == 0) *s_ok = 0; continue; }
            if (r == jj) dv[jj] = d;
        }
        const double yj = DIAG ? readlane_f64(yr, jj) : yv[jj];
        const double l = a[jj] / d;
        yr -= l * yj;
#pragma unroll
        for (int kk = jj + 1; kk < LD_NB; kk++) a[kk] -= l * U[jj * LD_NB + kk];
        if (!DIAG || r > jj) a[jj] = l;
    }
#pragma unroll
    for (int c = 0; c < LD_NB; c++) if (!DIAG || c <= r) P[r * LD_PP + c] = a[c];
    if (!DIAG) yv[r] = yr;
}

// S: [n][ld] row-major, lower triangle read and overwritten with L / d; rhs, x: n doubles in global memory (x may alias rhs); lds:
// ba_ldlt_lds_bytes(max_ld) bytes of dynamic LDS with max_ld >= n.  Every thread of the (<= 1024-thread, >= n - 32) block calls
// it; the return value is block-uniform: false = zero / non-finite pivot, x untouched.
__device__ __forceinline__ bool ldlt_solve_wg(double *S, int ld, int n, const double *rhs, double *x, double *lds, int max_ld)
{
    __shared__ int s_ok;
    const int tid = threadIdx.x, nth = blockDim.x;
    double *P = lds;                              // panel [rows][LD_PP]
    double *y = P + (size_t)max_ld * LD_PP;     // right-hand side / solution
    double *dval = y + max_ld;                  // pivots d_c of every column
    double *red = dval + max_ld;                // [32][32] partial sums of the back substitution
    double *U = red;                              // [32][32] unscaled columns of the current diagonal block (factorisation phase)
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
        // (1) the nb x nb diagonal block: ONE wave, one row per lane, the row in registers, no block barriers.  Step jj publishes the
        //     still unscaled column jj (U[jj][r] = A[r][jj]) in LDS; every lane reads it back as wave-wide broadcasts.
        if (tid < nb) ldlt_rows<true>(P, U, dval + p0, y + p0, tid, nb, &s_ok);
        __syncthreads();
        if (!s_ok) break;
        // (2) the rows below the block: independent forward substitutions against U / d, one row per thread
        if (nb + tid < m) ldlt_rows<false>(P, U, dval + p0, y + p0, nb + tid, nb, &s_ok);     // m <= BA_LDLT_MAXN < blockDim
        __syncthreads();
        if (!s_ok) break;
        // write the factored panel back: L below the diagonal, d on it
        for (int idx = tid; idx < m * LD_NB; idx += nth) {
            const int r = idx >> 5, c = idx & 31;
            if (c < nb && r >= c) S[(size_t)(p0 + r) * ld + p0 + c] = (r == c) ? dval[p0 + c] : P[r * LD_PP + c];
        }
        // trailing update S[i][k] -= sum_c L[i][c] d_c L[k][c]  (i >= k >= p0+nb), 4x4 register tiles
        const int m2 = m - nb;
        if (m2 > 0) {
            const int T = (m2 + 3) >> 2, ntri = T * (T + 1) / 2;
            for (int t = tid; t < ntri; t += nth) {               // lower-triangular tiles only, evenly dealt (row ti, column tk <= ti)
                int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
                while (ti * (ti + 1) / 2 > t) ti--;
                while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
                const int tk = t - ti * (ti + 1) / 2;
                const int i0 = nb + 4 * ti, k0 = nb + 4 * tk;
                double acc[4][4], sv[4][4];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b2 = 0; b2 < 4; b2++) {               // the tile's current values: in flight while the products are formed
                        acc[a][b2] = 0.0;
                        sv[a][b2] = (i0 + a < m && k0 + b2 < m && i0 + a >= k0 + b2) ? S[(size_t)(p0 + i0 + a) * ld + p0 + k0 + b2] : 0.0;
                    }
                for (int c = 0; c < nb; c++) {
                    const double dc = dval[p0 + c];
                    double av[4], bv[4];
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        av[a] = (i0 + a < m) ? P[(i0 + a) * LD_PP + c] * dc : 0.0;
                        bv[a] = (k0 + a < m) ? P[(k0 + a) * LD_PP + c] : 0.0;
                    }
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b2 = 0; b2 < 4; b2++) acc[a][b2] += av[a] * bv[b2];
                }
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b2 = 0; b2 < 4; b2++)
                        if (i0 + a < m && k0 + b2 < m && i0 + a >= k0 + b2)
                            S[(size_t)(p0 + i0 + a) * ld + p0 + k0 + b2] = sv[a][b2] - acc[a][b2];
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (!s_ok) return false;                             // x untouched (as the reference on failure)
    // y <- D^-1 y, then L^T x = y panel by panel from the bottom
    for (int i = tid; i < n; i += nth) y[i] /= dval[i];
    __syncthreads();
    const int last_p0 = ((n - 1) / LD_NB) * LD_NB;
    for (int p0 = last_p0; p0 >= 0; p0 -= LD_NB) {
        const int nb = min(LD_NB, n - p0), m = n - p0;
        // contributions of the rows below the diagonal block: t_c = sum_{r >= nb} L[p0+r][p0+c] x[p0+r]
        {
            const int c = tid & 31, rg = tid >> 5;          // 1024 threads = 32 columns x 32 row groups
            double part = 0.0;
            if (c < nb)
                for (int r = nb + rg; r < m; r += 32) part += S[(size_t)(p0 + r) * ld + p0 + c] * y[p0 + r];
            red[rg * 32 + c] = part;
        }
        for (int idx = tid; idx < nb * LD_NB; idx += nth) {   // diagonal block of L into LDS
            const int r = idx >> 5, c = idx & 31;
            if (c < nb) P[r * LD_PP + c] = S[(size_t)(p0 + r) * ld + p0 + c];
        }
        __syncthreads();
        if (tid < nb) {
            double t = 0.0;
            for (int rg = 0; rg < 32; rg++) t += red[rg * 32 + tid];
            y[p0 + tid] -= t;
        }
        __syncthreads();
        if (tid < 64) {                                       // 32x32 triangular solve inside one wave
            for (int jj = nb - 1; jj >= 0; jj--) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const double xj = y[p0 + jj];
                if (tid < jj) y[p0 + tid] -= P[jj * LD_PP + tid] * xj;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += nth) x[i] = y[i];
    return true;
}
