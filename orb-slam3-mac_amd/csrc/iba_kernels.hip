// iba_kernels.hip -- Optimizer::LocalInertialBA's numerical core on gfx950 (reference src/Optimizer.cc:4574-5187; vertices and
// edges include/G2oTypes.h + src/G2oTypes.cc: ImuCamPose :25-220, EdgeMono / EdgeStereo :349-482, EdgeInertial :693-800,
// EdgeGyroRW / EdgeAccRW G2oTypes.h:632-700; bias-corrected preintegration src/ImuTypes.cc:351-378; g2o Levenberg-Marquardt
// with the landmark Schur complement as in ba_kernels.hip).
//
// Shape of the problem: <= 25 keyframes with 15 unknowns each (pose 6, velocity 3, gyro bias 3, accelerometer bias 3), up to 200
// fixed keyframes, a few thousand landmarks, a chain of <= 25 inertial edges.  The reduced system is at most 375 x 375 and every
// LM trial is a short dependent chain (errors -> Schur -> LDL^T -> update -> errors), so ONE workgroup of 1024 threads owns a
// window for the whole optimisation: no host round trips, no inter-workgroup synchronisation, all LM decisions taken redundantly
// by every thread from block-uniform sums.  A batch of windows is a grid of such workgroups (256 CUs = 256 windows in flight).
// Every sum has a fixed order (per-thread sequential, DPP tree inside a wave, waves in index order): results are deterministic.
//
// Scratch (Jacobian blocks W, H, S ...) lives in global memory and stays L2-resident (a window's working set is < 4 MB); the
// LDL^T panel is the only LDS user (ba_ldlt.h).
#include "orb_internal.h"
#include "wave_dpp.h"
#include "ba_ldlt.h"
#include "ba_camera.h"
// The library is built with -ffp-contract=off for the bit-exact integer / float ORB paths.  The double-precision optimisers are
// compared with the oracle to 1e-4, not bit for bit: let a * b + c contract to v_fma_f64 here (half the FP64 instructions).
#pragma clang fp contract(fast)
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#include <thread>
#include <functional>
#include <mutex>
#include <fcntl.h>
#include <errno.h>
#include <unistd.h>
#include <sys/file.h>
#include <sys/stat.h>

hipStream_t orbhip_ctx_stream_internal(orbhip_ctx *c);
int orbhip_ctx_device_internal(orbhip_ctx *c);
void *orbhip_ctx_scratch_internal(orbhip_ctx *c, size_t bytes);
void orbhip_set_last_error_internal(const char *msg);

#define IBA_KF ORBHIP_IBA_KF
#define IBA_PRE ORBHIP_IBA_PREINT
#define IBA_THREADS 512
#define IBA_WAVES (IBA_THREADS / 64)
#define IBA_STAGE 544          // doubles of LDS a wave of workgroup 0 stages an inertial edge in (9 x 24 Jacobian, its Omega-weighted copy, Omega e at 432, the 9 x 9 information at 448)
#define IBA_MAXG 32
#define IBA_KF_CHUNK 256
#define IBA_PAIR_CHUNK 256
enum { K_R = 0, K_T = 9, K_V = 12, K_BG = 15, K_BA = 18 };
enum { P_DT = 0, P_DR = 1, P_DV = 10, P_DP = 13, P_JRG = 16, P_JVG = 25, P_JVA = 34, P_JPG = 43, P_JPA = 52, P_BG = 61, P_BA = 64 };

// Pointers into global memory carry their address space in the TYPE on the device side: the phase functions are not inlined (register
// pressure) and reach IbaArgs through a pointer, where a plain `double *` loaded from memory is a generic pointer and every access
// through it a FLAT instruction (1100 of them in the round-3 kernel: slower than global ones, and counted on BOTH wait counters, so that
// they serialise with the LDS traffic).  gen() hands out the generic pointer the code computes with; the backend's address-space
// inference sees the cast from the global space and emits global instructions.  Same layout on the host, which fills the struct.
#if defined(__HIP_DEVICE_COMPILE__)
#define GPTR(...) __attribute__((address_space(1))) __VA_ARGS__ *
#else
#define GPTR(...) __VA_ARGS__ *
#endif
struct IbaWin {
    int n_kf, L, E, M, n, nfree, ncolors, npairs, nktask, nptask;
    int kf_off, pt_off, e_off, m_off, x_off, free_off, ptstart_off, kfe_off, ktask_off, ktstart_off, ptask_off, ptstart2_off;
    long long pent_off, h_off;
    double Rcb[9], tcb[3], fx, fy, cx, cy, bf;
    int cam_model, has_cam2, cam2_model;            // camera 0; a second camera of the rig (EdgeMono(1) edges, edge type 2)
    double kb[4], Rcb2[9], tcb2[3], fx2, fy2, cx2, cy2, kb2[4];
};

struct IbaArgs {
    GPTR(const IbaWin) win;
    GPTR(const int) kf_xoff;                 // [sumKF] first unknown of the keyframe's block, -1 = fixed
    GPTR(const uint8_t) kf_imu;
    GPTR(const int) free_kf;                 // [sumFree] keyframe of free block f
    GPTR(const int) edge_kf; GPTR(const int) edge_point;    // [sumE] window-local indices
    GPTR(const double) edge_obs; GPTR(const double) edge_is2;
    GPTR(const uint8_t) edge_stereo; GPTR(const uint8_t) edge_close;
    GPTR(const int) pt_start;                // per window L + 1 entries
    GPTR(const int) kf_edges;                // the visual edges of every free keyframe, keyframe by keyframe
    GPTR(const int4) kf_task;                // {free block f, first, end (into kf_edges), 0}: chunks of <= IBA_KF_CHUNK edges
    GPTR(const int) kf_task_start;           // per window nfree + 1: the chunks of block f
    GPTR(const int2) pair_ent;               // {edge of block i, edge of block j} of every landmark both see, pair by pair (i <= j, row-major)
    GPTR(const int4) pair_task;              // {pair, first, end (into pair_ent), i == j}: chunks of <= IBA_PAIR_CHUNK entries
    GPTR(const int) pair_task_start;         // per window npairs + 1
    GPTR(const int) in_kf1; GPTR(const int) in_kf2; GPTR(const int) in_color; GPTR(const uint8_t) in_robust;
    GPTR(const double) in_pre; GPTR(const double) in_info; GPTR(const double) in_info_g; GPTR(const double) in_info_a;
    GPTR(double) kfs; GPTR(double) cam; GPTR(double) pts;            // estimates, two buffers each: [2][sumKF][21], [2][sumKF][2 cameras][12] (Rcw, tcw), [2][sumL][3]
    long long kfs_stride, cam_stride, pts_stride;
    GPTR(double) err; GPTR(double) chi2; GPTR(double) W; GPTR(double) Hll; GPTR(double) bl; GPTR(double) Dinv; GPTR(double) db; GPTR(double) xl;
    GPTR(double) ierr; GPTR(double) ichi2; GPTR(double) Jb; GPTR(double) OJ; GPTR(double) Oe;
    GPTR(double) H; GPTR(double) S; GPTR(double) b; GPTR(double) bs; GPTR(double) x;
    GPTR(double) kpart; GPTR(double) ppart;              // [kf tasks][27], [pair tasks][42] partial sums
    GPTR(unsigned) counters; GPTR(int) fail; GPTR(int) okflag;   // per window: team barrier counter, barrier time-out flag, LDL^T status
    GPTR(double) wpart;                      // per window [2][IBA_MAXG][2]
    int G, n_windows;                   // workgroups per window
    GPTR(uint8_t) outlier;
    GPTR(orbhip_iba_stats) stats;
    int iterations, max_trials, large, max_n;
    double lambda_init;
    GPTR(long long) prof;                    // optional [windows][16] shader-clock cycles per phase (ORBHIP_IBA_PROF=1): errors, build, prep+Schur, LDL^T, update
};

#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
__device__ __forceinline__ T *gen(__attribute__((address_space(1))) T *p) { return (T *)p; }
#else
template <class T>
__host__ __device__ inline T *gen(T *p) { return p; }      // (the host pass only parses the device functions)
#endif
#define GA(f) gen(A.f)
// ------------------------------------------------------------------ small dense helpers (row-major 3x3)
namespace {
__device__ __forceinline__ void mm3(const double *A, const double *B, double *C)
{
    double t[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) C[i] = t[i];
}
__device__ __forceinline__ void mtm3(const double *A, const double *B, double *C)     // A^T B
{
    double t[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) C[i] = t[i];
}
__device__ __forceinline__ void mv3(const double *A, const double *v, double *o)
{
    const double a = A[0] * v[0] + A[1] * v[1] + A[2] * v[2], b = A[3] * v[0] + A[4] * v[1] + A[5] * v[2], c = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = a; o[1] = b; o[2] = c;
}
__device__ __forceinline__ void mtv3(const double *A, const double *v, double *o)     // A^T v
{
    const double a = A[0] * v[0] + A[3] * v[1] + A[6] * v[2], b = A[1] * v[0] + A[4] * v[1] + A[7] * v[2], c = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = a; o[1] = b; o[2] = c;
}
__device__ __forceinline__ void skew3(const double *w, double *W)
{
    W[0] = 0; W[1] = -w[2]; W[2] = w[1]; W[3] = w[2]; W[4] = 0; W[5] = -w[0]; W[6] = -w[1]; W[7] = w[0]; W[8] = 0;
}
__device__ __forceinline__ void inv3(const double *A, double *I)
{
    const double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
    const double id = 1.0 / (A[0] * c0 + A[1] * c1 + A[2] * c2);
    I[0] = c0 * id; I[1] = (A[2] * A[7] - A[1] * A[8]) * id; I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    I[3] = c1 * id; I[4] = (A[0] * A[8] - A[2] * A[6]) * id; I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    I[6] = c2 * id; I[7] = (A[1] * A[6] - A[0] * A[7]) * id; I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}
// nearest rotation (IMU::NormalizeRotation, ImuTypes.cc:30-36, takes U V^T of an SVD): two Newton polar steps
__device__ __forceinline__ void normalize_rotation(double *R)
{
    for (int it = 0; it < 2; it++) {
        double I[9];
        inv3(R, I);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) R[3 * i + j] = 0.5 * (R[3 * i + j] + I[3 * j + i]);
    }
}
__device__ __forceinline__ void rodrigues(const double *w, double a, double b, double *R)     // I + a W + b W W
{
    double W[9], W2[9];
    skew3(w, W); mm3(W, W, W2);
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = a * W[i] + b * W2[i];
    R[0] += 1; R[4] += 1; R[8] += 1;
}
__device__ void exp_so3(const double *w, double *R, double eps, bool normalize)     // G2oTypes.cc:991-1008 (1e-5, normalised) / ImuTypes.cc:48-60 (1e-4)
{
    const double d2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], d = sqrt(d2);
    if (d < eps) rodrigues(w, 1.0, 0.5, R);
    else rodrigues(w, sin(d) / d, (1.0 - cos(d)) / d2, R);
    if (normalize) normalize_rotation(R);
}
__device__ void log_so3(const double *R, double *w)     // G2oTypes.cc:1010-1025
{
    const double tr = R[0] + R[4] + R[8];
    w[0] = (R[7] - R[5]) / 2; w[1] = (R[2] - R[6]) / 2; w[2] = (R[3] - R[1]) / 2;
    const double costheta = (tr - 1.0) * 0.5;
    if (costheta > 1 || costheta < -1) return;
    const double theta = acos(costheta), s = sin(theta);
    if (fabs(s) < 1e-5) return;
    for (int i = 0; i < 3; i++) w[i] = theta * w[i] / s;
}
__device__ void inv_right_jac(const double *v, double *J)     // G2oTypes.cc:1032-1044
{
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrt(d2);
    if (d < 1e-5) { for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    rodrigues(v, 0.5, 1.0 / d2 - (1.0 + cos(d)) / (2.0 * d * sin(d)), J);
}
__device__ void right_jac(const double *v, double *J)         // G2oTypes.cc:1046-1061
{
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrt(d2);
    if (d < 1e-5) { for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    rodrigues(v, -(1.0 - cos(d)) / d2, (d - sin(d)) / (d2 * d), J);
}
__device__ __forceinline__ void huber(double e, double delta, double dsqr, double &rho0, double &rho1)
{
    if (e <= dsqr) { rho0 = e; rho1 = 1.0; }
    else { const double s = sqrt(e); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
}

// sum over each group of 8 consecutive lanes, every lane of the group gets the total (quad butterflies + half-row mirror;
// each lane's association is fixed)
__device__ __forceinline__ double oct_allreduce_f64(double v)
{
#define IBA_STEP(CTRL) do { const unsigned long long u = __builtin_bit_cast(unsigned long long, v); \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), CTRL, 0xF, 0xF, false), \
                       hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, 0xF, 0xF, false); \
        v += __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); } while (0)
    IBA_STEP(0xB1);      // quad_perm [1,0,3,2]
    IBA_STEP(0x4E);      // quad_perm [2,3,0,1]
    IBA_STEP(0x141);     // row_half_mirror
#undef IBA_STEP
    return v;
}

// camera `idx` of the rig: extrinsics w.r.t. the body (G2oTypes.cc:49-52, 57-67) and intrinsics
struct CamView { const double *Rcb, *tcb, *kb; double fx, fy, cx, cy; int model; };
__device__ __forceinline__ CamView cam_view(const IbaWin &W, int idx)
{
    CamView V;
    if (idx == 0) { V.Rcb = W.Rcb; V.tcb = W.tcb; V.kb = W.kb; V.fx = W.fx; V.fy = W.fy; V.cx = W.cx; V.cy = W.cy; V.model = W.cam_model; }
    else { V.Rcb = W.Rcb2; V.tcb = W.tcb2; V.kb = W.kb2; V.fx = W.fx2; V.fy = W.fy2; V.cx = W.cx2; V.cy = W.cy2; V.model = W.cam2_model; }
    return V;
}
// Rcw = Rcb Rbw, tcw = Rcb tbw + tcb for every camera of the keyframe (ImuCamPose::Update, G2oTypes.cc:212-219); c: [2][12]
__device__ void cam_pose(const IbaWin &W, const double *s, double *c)
{
    double Rbw[9], tbw[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Rbw[3 * i + j] = s[K_R + 3 * j + i];
    mv3(Rbw, s + K_T, tbw);
    for (int i = 0; i < 3; i++) tbw[i] = -tbw[i];
    for (int cam = 0; cam <= W.has_cam2; cam++) {
        const CamView V = cam_view(W, cam);
        double R[9], t[3];
        mm3(V.Rcb, Rbw, R);
        mv3(V.Rcb, tbw, t);
        for (int i = 0; i < 9; i++) c[12 * cam + i] = R[i];
        for (int i = 0; i < 3; i++) c[12 * cam + 9 + i] = t[i] + V.tcb[i];
    }
}

// EdgeMono / EdgeStereo::computeError (G2oTypes.h:350-355, G2oTypes.cc:170-185); type 0 EdgeMono(0), 1 EdgeStereo(0), 2 EdgeMono(1);
// c = the pose of the edge's camera
__device__ __forceinline__ void visual_error(const IbaWin &W, const CamView &V, const double *c, const double *X, const double *obs, int type, double *er, double *Xc)
{
    mv3(c, X, Xc);
    Xc[0] += c[9]; Xc[1] += c[10]; Xc[2] += c[11];
    double uv[2];
    cam_project(V.fx, V.fy, V.cx, V.cy, V.model, V.kb, Xc, uv);
    er[0] = obs[0] - uv[0]; er[1] = obs[1] - uv[1];
    er[2] = type == 1 ? obs[2] - (uv[0] - W.bf * (1 / Xc[2])) : 0.0;
}

// linearizeOplus (G2oTypes.cc:349-373, :397-423): Jx [3][3], Jp [3][6]; row 2 zero when mono
__device__ void visual_jac(const IbaWin &W, const CamView &V, const double *c, const double *Xc, int type, double *Jx, double *Jp)
{
    double pj[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    cam_project_jac(V.fx, V.fy, V.model, V.kb, Xc, pj);
    if (type == 1) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + W.bf * (1.0 / (Xc[2] * Xc[2])); }
#pragma unroll
    for (int d = 0; d < 3; d++)
#pragma unroll
        for (int j = 0; j < 3; j++) Jx[3 * d + j] = -(pj[3 * d] * c[j] + pj[3 * d + 1] * c[3 + j] + pj[3 * d + 2] * c[6 + j]);
    double Xb[3];
    const double d0[3] = {Xc[0] - V.tcb[0], Xc[1] - V.tcb[1], Xc[2] - V.tcb[2]};
    mtv3(V.Rcb, d0, Xb);
    double PR[9];
    mm3(pj, V.Rcb, PR);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const double p0 = PR[3 * d], p1 = PR[3 * d + 1], p2 = PR[3 * d + 2];
        Jp[6 * d + 0] = p1 * -Xb[2] + p2 * Xb[1];            // SE3deriv columns: (0,-z,y) (z,0,-x) (-y,x,0) | I
        Jp[6 * d + 1] = p0 * Xb[2] + p2 * -Xb[0];
        Jp[6 * d + 2] = p0 * -Xb[1] + p1 * Xb[0];
        Jp[6 * d + 3] = p0; Jp[6 * d + 4] = p1; Jp[6 * d + 5] = p2;
    }
}

// EdgeInertial::computeError (+ linearizeOplus into J [9][24] when J != nullptr), G2oTypes.cc:720-800
__device__ __noinline__ void inertial_edge(const double *s1, const double *s2, const double *pi, double *err, double *J)
{
    const double dt = pi[P_DT];
    const double g[3] = {0, 0, -(double)9.81f};
    double dbg[3], dba[3], w[3], E[9], dR[9], dV[3], dP[3], t[3];
    for (int i = 0; i < 3; i++) { dbg[i] = s1[K_BG + i] - pi[P_BG + i]; dba[i] = s1[K_BA + i] - pi[P_BA + i]; }
    mv3(pi + P_JRG, dbg, w);
    exp_so3(w, E, 1e-4, false);
    mm3(pi + P_DR, E, dR);
    normalize_rotation(dR);
    mv3(pi + P_JVG, dbg, dV); mv3(pi + P_JVA, dba, t);
    for (int i = 0; i < 3; i++) dV[i] = pi[P_DV + i] + dV[i] + t[i];
    mv3(pi + P_JPG, dbg, dP); mv3(pi + P_JPA, dba, t);
    for (int i = 0; i < 3; i++) dP[i] = pi[P_DP + i] + dP[i] + t[i];
    double R12[9], eR[9], er[3], a[3], b[3], va[3], vb[3];
    mtm3(s1 + K_R, s2 + K_R, R12);
    mtm3(dR, R12, eR);
    log_so3(eR, er);
    for (int i = 0; i < 3; i++) {
        a[i] = s2[K_V + i] - s1[K_V + i] - g[i] * dt;
        b[i] = s2[K_T + i] - s1[K_T + i] - s1[K_V + i] * dt - g[i] * dt * dt / 2;
    }
    mtv3(s1 + K_R, a, va); mtv3(s1 + K_R, b, vb);
    for (int i = 0; i < 3; i++) { err[i] = er[i]; err[3 + i] = va[i] - dV[i]; err[6 + i] = vb[i] - dP[i]; }
    if (!J) return;
    for (int i = 0; i < 216; i++) J[i] = 0.0;
    double invJr[9], T[9], S[9];
    inv_right_jac(er, invJr);
#define SETB(r0, c0, M, sgn) for (int i_ = 0; i_ < 3; i_++) for (int j_ = 0; j_ < 3; j_++) J[24 * ((r0) + i_) + (c0) + j_] = (sgn) * (M)[3 * i_ + j_]
    mtm3(s2 + K_R, s1 + K_R, T); mm3(invJr, T, T); SETB(0, 0, T, -1.0);
    skew3(va, S); SETB(3, 0, S, 1.0);
    skew3(vb, S); SETB(6, 0, S, 1.0);
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    SETB(6, 3, I3, -1.0);
    double Rbw1[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rbw1[3 * i + j] = s1[K_R + 3 * j + i];
    SETB(3, 6, Rbw1, -1.0);
    SETB(6, 6, Rbw1, -dt);
    double RJ[9], eRt[9];
    right_jac(w, RJ);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) eRt[3 * i + j] = eR[3 * j + i];
    mm3(invJr, eRt, T); mm3(T, RJ, T); mm3(T, pi + P_JRG, T); SETB(0, 9, T, -1.0);
    SETB(3, 9, pi + P_JVG, -1.0);
    SETB(6, 9, pi + P_JPG, -1.0);
    SETB(3, 12, pi + P_JVA, -1.0);
    SETB(6, 12, pi + P_JPA, -1.0);
    SETB(0, 15, invJr, 1.0);
    SETB(6, 18, R12, 1.0);
    SETB(3, 21, Rbw1, 1.0);
#undef SETB
}

// ------------------------------------------------------------------ a window's team of workgroups
// G workgroups work on one window (G = 1 for big batches: plain __syncthreads semantics).  team_sync is a monotonic-counter
// barrier (MI355X_MICROARCH.md "barrier-counter": lane-0 agent release fence + drained vmcnt before the arrive, relaxed polls
// with s_sleep, agent acquire fence after); the grid is sized by the host so that all workgroups are resident (cooperative
// launch), and every spin is bounded: a barrier that does not complete sets *fail and lets the kernel run out.
struct Team {
    int w, g, G, gtid, gsize, gwave, gwaves;
    unsigned *counter;          // per window, zeroed by the host
    unsigned epoch;             // barriers passed so far (uniform)
    double *wpart;              // [2][IBA_MAXG][2] cross-workgroup partial sums, double buffered
    int slot;
    int *fail;                  // per window
};

__device__ __forceinline__ void team_sync(Team &T)
{
    __syncthreads();
    if (T.G > 1) {
        T.epoch++;
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(T.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = T.epoch * (unsigned)T.G;
            long long spins = 0;
            while (__hip_atomic_load(T.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1ll << 22) || __hip_atomic_load(T.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {       // seconds: never in a resident grid
                    __hip_atomic_store(T.fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
}

// team-uniform sums of two per-thread values: DPP tree inside each wave, the 16 wave sums in index order, then the G workgroup
// sums in index order (fixed association for a given G)
__device__ __forceinline__ void team_sum2(Team &T, double &a, double &b, double *red)
{
    a = wave_sum_f64_dpp(a); b = wave_sum_f64_dpp(b);
    __syncthreads();                                         // red may still be read by the previous call
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = a; red[2 * (threadIdx.x >> 6) + 1] = b; }
    __syncthreads();
    double sa = 0, sb = 0;
#pragma unroll
    for (int w = 0; w < IBA_WAVES; w++) { sa += red[2 * w]; sb += red[2 * w + 1]; }
    if (T.G > 1) {
        double *wp = T.wpart + (size_t)T.slot * IBA_MAXG * 2;
        if (threadIdx.x == 0) { wp[2 * T.g] = sa; wp[2 * T.g + 1] = sb; }
        team_sync(T);
        sa = 0; sb = 0;
        for (int g = 0; g < T.G; g++) { sa += wp[2 * g]; sb += wp[2 * g + 1]; }
        T.slot ^= 1;
    }
    a = sa; b = sb;
}

struct IbaCtx {            // per-window views (all threads hold the same values)
    GPTR(const IbaWin) W;
    GPTR(const IbaArgs) A;
    double delta_m, dsqr_m, delta_s, dsqr_s, delta_i, dsqr_i;
};

// computeActiveErrors at estimate buffer `buf`: per-edge errors / chi2 to global memory; returns this THREAD's share of
// activeRobustChi2 (the caller sums over the team)
__device__ __noinline__ double iba_errors(const IbaCtx &C, const Team &T, int buf)
{
    const IbaWin &W = *gen(C.W); const IbaArgs &A = *gen(C.A);
    const double *cam = GA(cam) + buf * A.cam_stride + (size_t)W.kf_off * 24;
    const double *pts = GA(pts) + buf * A.pts_stride + (size_t)W.pt_off * 3;
    const double *kfs = GA(kfs) + buf * A.kfs_stride + (size_t)W.kf_off * IBA_KF;
    double part = 0.0;
    for (int e = T.gtid; e < W.E; e += T.gsize) {
        const size_t ge = (size_t)W.e_off + e;
        const int type = GA(edge_stereo)[ge], st = type == 1;
        double er[3], Xc[3];
        visual_error(W, cam_view(W, type == 2), cam + 24 * GA(edge_kf)[ge] + 12 * (type == 2), pts + 3 * GA(edge_point)[ge], GA(edge_obs) + 3 * ge, type, er, Xc);
        const double chi = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * GA(edge_is2)[ge];
        GA(err)[3 * ge] = er[0]; GA(err)[3 * ge + 1] = er[1]; GA(err)[3 * ge + 2] = er[2];
        GA(chi2)[ge] = chi;
        double r0, r1;
        huber(chi, st ? C.delta_s : C.delta_m, st ? C.dsqr_s : C.dsqr_m, r0, r1);
        part += r0;
    }
    // inertial edges: the last threads of the team (the first ones carry the most visual edges)
    const int m = T.gsize - 1 - T.gtid;
    if (m < W.M) {
        const size_t gm = (size_t)W.m_off + m;
        const double *s1 = kfs + IBA_KF * GA(in_kf1)[gm], *s2 = kfs + IBA_KF * GA(in_kf2)[gm];
        double er[15];
        inertial_edge(s1, s2, GA(in_pre) + IBA_PRE * gm, er, nullptr);
        for (int i = 0; i < 3; i++) { er[9 + i] = s2[K_BG + i] - s1[K_BG + i]; er[12 + i] = s2[K_BA + i] - s1[K_BA + i]; }
        const double *I9 = GA(in_info) + 81 * gm, *Ig = GA(in_info_g) + 9 * gm, *Ia = GA(in_info_a) + 9 * gm;
        double c9 = 0, cg = 0, ca = 0;
        for (int i = 0; i < 9; i++) { double r = 0; for (int j = 0; j < 9; j++) r += I9[9 * i + j] * er[j]; c9 += er[i] * r; }
        for (int i = 0; i < 3; i++) {
            double r = 0, q = 0;
            for (int j = 0; j < 3; j++) { r += Ig[3 * i + j] * er[9 + j]; q += Ia[3 * i + j] * er[12 + j]; }
            cg += er[9 + i] * r; ca += er[12 + i] * q;
        }
        for (int i = 0; i < 15; i++) GA(ierr)[15 * gm + i] = er[i];
        GA(ichi2)[3 * gm] = c9; GA(ichi2)[3 * gm + 1] = cg; GA(ichi2)[3 * gm + 2] = ca;
        double r0 = c9, r1;
        if (GA(in_robust)[gm]) huber(c9, C.delta_i, C.dsqr_i, r0, r1);
        part += r0 + cg + ca;
    }
    return part;
}

// One chunk of a free keyframe's visual edges -> partial sums of its 6x6 pose block (lower triangle, 21) and right-hand side (6).
// HALF 0: rows 0..3 of the block (10 entries) + the right-hand side; HALF 1: rows 4, 5 (11 entries): two waves per chunk keep the
// accumulators + the 3x6 Jacobian inside the 128-VGPR budget of a 16-wave workgroup.
template <int HALF>
__device__ __forceinline__ void kf_block_task(const IbaCtx &C, const double *cam, const double *pts, int4 task, double *out)
{
    const IbaWin &W = *gen(C.W); const IbaArgs &A = *gen(C.A);
    constexpr int A0 = HALF ? 4 : 0, A1 = HALF ? 6 : 4, Q0 = HALF ? 10 : 0, NQ = HALF ? 11 : 10, NB = HALF ? 0 : 6;
    const int lane = threadIdx.x & 63;
    const int k = GA(free_kf)[W.free_off + task.x];
    const int *kf_edges = GA(kf_edges) + W.kfe_off;
    double acc[NQ + NB + 1];
#pragma unroll
    for (int i = 0; i < NQ + NB; i++) acc[i] = 0.0;
    for (int j = task.y + lane; j < task.z; j += 64) {
        const size_t ge = (size_t)W.e_off + kf_edges[j];
        const int type = GA(edge_stereo)[ge], st = type == 1;
        const double *c = cam + 24 * k + 12 * (type == 2);
        const double *X = pts + 3 * GA(edge_point)[ge];
        double Xc[3], Jx[9], Jp[18], r0, r1;
        mv3(c, X, Xc); Xc[0] += c[9]; Xc[1] += c[10]; Xc[2] += c[11];
        visual_jac(W, cam_view(W, type == 2), c, Xc, type, Jx, Jp);
        huber(GA(chi2)[ge], st ? C.delta_s : C.delta_m, st ? C.dsqr_s : C.dsqr_m, r0, r1);
        const double w = r1 * GA(edge_is2)[ge];
        int q = 0;
#pragma unroll
        for (int a = A0; a < A1; a++)
#pragma unroll
            for (int cc = 0; cc <= a; cc++, q++)
                acc[q] += Jp[a] * w * Jp[cc] + Jp[6 + a] * w * Jp[6 + cc] + Jp[12 + a] * w * Jp[12 + cc];      // row 2 of Jp is zero when mono
        if (!HALF) {
            const double e0 = -w * GA(err)[3 * ge], e1 = -w * GA(err)[3 * ge + 1], e2 = st ? -w * GA(err)[3 * ge + 2] : 0.0;
#pragma unroll
            for (int a = 0; a < 6; a++) acc[NQ + a] += Jp[a] * e0 + Jp[6 + a] * e1 + Jp[12 + a] * e2;
        }
    }
#pragma unroll
    for (int i = 0; i < NQ + NB; i++) acc[i] = wave_sum_f64_dpp(acc[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NQ; i++) out[Q0 + i] = acc[i];
#pragma unroll
        for (int i = 0; i < NB; i++) out[21 + i] = acc[NQ + i];
    }
}

// buildSystem at buffer `buf` from the stored errors.  Team: Hll, bl and the pose-landmark blocks W per landmark; per-chunk partial
// pose blocks; the Omega-weighted inertial Jacobians.  Workgroup 0 then assembles H (dense, both triangles) and b.
__device__ __noinline__ void iba_build(const IbaCtx &C, Team &T, int buf)
{
    const IbaWin &W = *gen(C.W); const IbaArgs &A = *gen(C.A);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = W.n;
    const double *cam = GA(cam) + buf * A.cam_stride + (size_t)W.kf_off * 24;
    const double *pts = GA(pts) + buf * A.pts_stride + (size_t)W.pt_off * 3;
    const double *kfs = GA(kfs) + buf * A.kfs_stride + (size_t)W.kf_off * IBA_KF;
    double *H = GA(H) + W.h_off, *b = GA(b) + W.x_off;
    const bool bprof = GA(prof) && T.g == 0 && tid == 0;
    long long *bpf = GA(prof) ? GA(prof) + 16 * T.w + 8 : nullptr;
    long long tb = bprof ? clock64() : 0;
#define BPROF(k) do { if (bprof) { const long long t_ = clock64(); bpf[k] += t_ - tb; tb = t_; } } while (0)
    if (T.g == 0) {
        for (int i = tid; i < n * n; i += IBA_THREADS) H[i] = 0.0;
        for (int i = tid; i < n; i += IBA_THREADS) b[i] = 0.0;
    }
    // (1) inertial Jacobians (the last threads of the team, as in iba_errors)
    {
        const int m = T.gsize - 1 - T.gtid;
        if (m < W.M) {
            const size_t gm = (size_t)W.m_off + m;
            double er[9];
            inertial_edge(kfs + IBA_KF * GA(in_kf1)[gm], kfs + IBA_KF * GA(in_kf2)[gm], GA(in_pre) + IBA_PRE * gm, er, GA(Jb) + 216 * gm);
        }
    }
    BPROF(0);
    // (2) landmarks: Hll, bl and the pose-landmark blocks; 8 lanes share a landmark's edges
    const int *pt_start = GA(pt_start) + W.ptstart_off;
    const int sub = tid & 7;
    for (int l0 = (T.gtid >> 3); l0 < ((W.L + 7) & ~7); l0 += (T.gsize >> 3)) {      // whole waves stay in the loop (DPP reductions)
        const int l = min(l0, W.L - 1);
        double h[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
        const double *X = pts + 3 * l;
        const int e_end = l0 < W.L ? pt_start[l + 1] : 0;
        for (int e = pt_start[l] + sub; e < e_end; e += 8) {
            const size_t ge = (size_t)W.e_off + e;
            const int type = GA(edge_stereo)[ge], st = type == 1, k = GA(edge_kf)[ge];
            const double *c = cam + 24 * k + 12 * (type == 2);
            double Xc[3], Jx[9], Jp[18], r0, r1;
            mv3(c, X, Xc); Xc[0] += c[9]; Xc[1] += c[10]; Xc[2] += c[11];
            visual_jac(W, cam_view(W, type == 2), c, Xc, type, Jx, Jp);
            huber(GA(chi2)[ge], st ? C.delta_s : C.delta_m, st ? C.dsqr_s : C.dsqr_m, r0, r1);
            const double w = r1 * GA(edge_is2)[ge];
            const double es[3] = {-w * GA(err)[3 * ge], -w * GA(err)[3 * ge + 1], st ? -w * GA(err)[3 * ge + 2] : 0.0};
            int q = 0;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                bl[a] += Jx[a] * es[0] + Jx[3 + a] * es[1] + Jx[6 + a] * es[2];      // row 2 of Jx is zero when mono
#pragma unroll
                for (int cc = 0; cc <= a; cc++, q++) h[q] += Jx[a] * w * Jx[cc] + Jx[3 + a] * w * Jx[3 + cc] + Jx[6 + a] * w * Jx[6 + cc];
            }
            if (GA(kf_xoff)[W.kf_off + k] >= 0) {
                double *Wd = GA(W) + 18 * ge;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int cc = 0; cc < 3; cc++) Wd[3 * a + cc] = Jp[a] * w * Jx[cc] + Jp[6 + a] * w * Jx[3 + cc] + Jp[12 + a] * w * Jx[6 + cc];
            }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) h[i] = oct_allreduce_f64(h[i]);
#pragma unroll
        for (int i = 0; i < 3; i++) bl[i] = oct_allreduce_f64(bl[i]);
        if (sub == 0 && l0 < W.L) {
            double *Hl = GA(Hll) + 6 * ((size_t)W.pt_off + l), *Bl = GA(bl) + 3 * ((size_t)W.pt_off + l);
            for (int i = 0; i < 6; i++) Hl[i] = h[i];
            for (int i = 0; i < 3; i++) Bl[i] = bl[i];
        }
    }
    BPROF(1);
    // (3) per-chunk partial pose blocks, two waves per chunk
    {
        const int4 *kf_task = GA(kf_task) + W.ktask_off;
        double *kpart = GA(kpart) + 27 * (size_t)W.ktask_off;
        for (int t = T.gwave; t < 2 * W.nktask; t += T.gwaves) {
            const int4 task = kf_task[t >> 1];
            if (t & 1) kf_block_task<1>(C, cam, pts, task, kpart + 27 * (size_t)(t >> 1));
            else kf_block_task<0>(C, cam, pts, task, kpart + 27 * (size_t)(t >> 1));
        }
    }
    BPROF(2);
    team_sync(T);                      // Jb, kpart complete
    BPROF(3);
    if (T.g != 0) return;
    BPROF(4);
    // ---- workgroup 0: pose blocks: the chunks of every keyframe summed in order
    {
        const int *kts = GA(kf_task_start) + W.ktstart_off;
        const double *kpart = GA(kpart) + 27 * (size_t)W.ktask_off;
        for (int idx = tid; idx < W.nfree * 27; idx += IBA_THREADS) {
            const int f = idx / 27, q = idx - 27 * f;
            const int o = GA(kf_xoff)[W.kf_off + GA(free_kf)[W.free_off + f]];
            double s = 0;
            for (int t = kts[f]; t < kts[f + 1]; t++) s += kpart[27 * (size_t)t + q];
            if (q >= 21) b[o + q - 21] = s;
            else {
                int a = 0, rem = q;
                while (rem > a) { rem -= a + 1; a++; }          // q -> (a, cc) of the lower triangle
                H[(size_t)(o + a) * n + o + rem] = s; H[(size_t)(o + rem) * n + o + a] = s;
            }
        }
    }
    // what the colour loop needs to know about an edge, fetched for all edges at once (colour, first unknowns of its two keyframes or -1,
    // robust flag): inside the loop these were four dependent L2 round trips per edge
    extern __shared__ double iba_dyn_lds[];
    typedef __attribute__((address_space(3))) int lds_i32;
    lds_i32 *recs = (lds_i32 *)((lds_f64 *)iba_dyn_lds + (size_t)IBA_WAVES * IBA_STAGE);
    for (int m = tid; m < W.M; m += IBA_THREADS) {
        const size_t gm = (size_t)W.m_off + m;
        const int *kx = GA(kf_xoff) + W.kf_off;
        recs[4 * m] = GA(in_color)[gm]; recs[4 * m + 1] = kx[GA(in_kf1)[gm]]; recs[4 * m + 2] = kx[GA(in_kf2)[gm]]; recs[4 * m + 3] = GA(in_robust)[gm];
    }
    __syncthreads();
    BPROF(5);
    // inertial + random-walk edges into H / b: edges of one colour share no keyframe, colours run one after the other.  A wave takes an
    // edge and keeps it in LDS (the solver's panel area is idle during the build): Jacobian, information and error are fetched in ONE
    // round trip, the Omega-weighted copies (Huber weight x information x J, -weight x information x error) are formed in LDS -- round 3
    // wrote them to global memory in a pass of its own, one L2 round trip per output -- and the 24 x 24 products read LDS only; the
    // entries of H a lane updates are read together before and written together after.  Same expressions, same order of sums.
    lds_f64 *sw = (lds_f64 *)iba_dyn_lds + (size_t)wave * IBA_STAGE;      // J [9][24] | Omega J [9][24] at 216 | Omega e [9] at 432 | information [9][9] at 448 | error [15] at 529
    for (int col = 0; col < W.ncolors; col++) {
        for (int m = 0; m < W.M; m++) {                             // (M is small: every wave scans the records, in LDS)
            const int cr = recs[4 * m];
            if ((cr & 255) != col || ((cr >> 8) % IBA_WAVES) != wave) continue;
            const size_t gm = (size_t)W.m_off + m;
            const int o1 = recs[4 * m + 1], o2 = recs[4 * m + 2];
            const double *Jg = GA(Jb) + 216 * gm, *Ig = GA(in_info) + 81 * gm, *erg = GA(ierr) + 15 * gm;
            double jv[4], iv[2];
#pragma unroll
            for (int t = 0; t < 4; t++) jv[t] = lane + 64 * t < 216 ? Jg[lane + 64 * t] : 0.0;
            iv[0] = Ig[lane]; iv[1] = lane + 64 < 81 ? Ig[lane + 64] : 0.0;
            const double erv = lane < 15 ? erg[lane] : 0.0;
            const double chi_i = GA(ichi2)[3 * gm];
            // EdgeGyroRW / EdgeAccRW (G2oTypes.h:648-651, J = [-I, I]): lanes 32..49 = 2 edges x 3 x 3 of H, lanes 50..55 = 2 x 3 of b
            double rwv = 0.0, rw0 = 0.0, rw1 = 0.0, rw2 = 0.0;
            if (lane >= 32 && lane < 50) { const int t = lane - 32, which = t / 9; rwv = ((which ? GA(in_info_a) : GA(in_info_g)) + 9 * gm)[t % 9]; }
            if (lane >= 50 && lane < 56) {
                const int t = lane - 50, which = t / 3, r = t % 3;
                const double *If = (which ? GA(in_info_a) : GA(in_info_g)) + 9 * gm;
                rw0 = If[3 * r]; rw1 = If[3 * r + 1]; rw2 = If[3 * r + 2];
            }
#pragma unroll
            for (int t = 0; t < 4; t++) if (lane + 64 * t < 216) sw[lane + 64 * t] = jv[t];
            sw[448 + lane] = iv[0];
            if (lane + 64 < 81) sw[448 + 64 + lane] = iv[1];
            if (lane < 15) sw[529 + lane] = erv;
            double r0, r1 = 1.0;
            if (recs[4 * m + 3]) huber(chi_i, C.delta_i, C.dsqr_i, r0, r1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int idx = lane; idx < 216; idx += 64) {
                const int i = idx / 24, c = idx - i * 24;
                double s = 0;
                for (int k = 0; k < 9; k++) s += r1 * sw[448 + 9 * i + k] * sw[24 * k + c];
                sw[216 + idx] = s;
            }
            if (lane < 9) {
                double q = 0;
                for (int k = 0; k < 9; k++) q += sw[448 + 9 * lane + k] * sw[529 + k];
                sw[432 + lane] = -r1 * q;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                double hv[9]; size_t ad[9];
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    const int idx = lane + 64 * j, a = idx / 24, c = idx - a * 24;
                    const int ga = a < 15 ? (o1 < 0 ? -1 : o1 + a) : (o2 < 0 ? -1 : o2 + a - 15);
                    const int gc = c < 15 ? (o1 < 0 ? -1 : o1 + c) : (o2 < 0 ? -1 : o2 + c - 15);
                    ad[j] = (ga < 0 || gc < 0) ? (size_t)-1 : (size_t)ga * n + gc;
                    hv[j] = ad[j] != (size_t)-1 ? H[ad[j]] : 0.0;
                }
                // the right-hand side entry of lanes 0..23 and the random-walk entries travel with the same round trip
                const int ba_ = lane < 24 ? (lane < 15 ? (o1 < 0 ? -1 : o1 + lane) : (o2 < 0 ? -1 : o2 + lane - 15)) : -1;
                double bv = ba_ >= 0 ? b[ba_] : 0.0;
                size_t a11 = 0, a22 = 0, a12 = 0, a21 = 0;
                int rb1 = -1, rb2 = -1;
                if (lane >= 32 && lane < 50) {
                    const int t = lane - 32, which = t / 9, r = (t % 9) / 3, c = t % 3, base = which ? 12 : 9;
                    if (o1 >= 0) a11 = (size_t)(o1 + base + r) * n + o1 + base + c;
                    if (o2 >= 0) a22 = (size_t)(o2 + base + r) * n + o2 + base + c;
                    if (o1 >= 0 && o2 >= 0) { a12 = (size_t)(o1 + base + r) * n + o2 + base + c; a21 = (size_t)(o2 + base + r) * n + o1 + base + c; }
                }
                if (lane >= 50 && lane < 56) {
                    const int t = lane - 50, which = t / 3, r = t % 3, base = which ? 12 : 9;
                    if (o1 >= 0) rb1 = o1 + base + r;
                    if (o2 >= 0) rb2 = o2 + base + r;
                }
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    const int idx = lane + 64 * j, a = idx / 24, c = idx - a * 24;
                    double s = 0;
                    for (int k = 0; k < 9; k++) s += sw[24 * k + a] * sw[216 + 24 * k + c];
                    if (ad[j] != (size_t)-1) H[ad[j]] = hv[j] + s;
                }
                if (ba_ >= 0) {
                    double s = 0;
                    for (int k = 0; k < 9; k++) s += sw[24 * k + lane] * sw[432 + k];
                    b[ba_] = bv + s;
                }
                // the random-walk entries are addresses OTHER lanes have just written (blocks 9..14 of both keyframes): those stores are
                // waited for before they are read back (the inertial sums go in first, as in round 3)
                __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
                __builtin_amdgcn_wave_barrier();
                if (lane >= 32 && lane < 50) {
                    if (o1 >= 0) H[a11] += rwv;
                    if (o2 >= 0) H[a22] += rwv;
                    if (o1 >= 0 && o2 >= 0) { H[a12] -= rwv; H[a21] -= rwv; }
                }
                if (lane >= 50 && lane < 56) {
                    const int t = lane - 50, which = t / 3;
                    const double v = rw0 * sw[529 + (which ? 12 : 9)] + rw1 * sw[529 + (which ? 12 : 9) + 1] + rw2 * sw[529 + (which ? 12 : 9) + 2];
                    if (rb1 >= 0) b[rb1] += v;                   // J1 = -I: b1 += -J1^T (-Omega e) = +Omega e
                    if (rb2 >= 0) b[rb2] -= v;
                }
            }
            __builtin_amdgcn_wave_barrier();                     // (the next edge of this wave restages sw)
        }
        __syncthreads();
    }
    BPROF(6);
#undef BPROF
}

// One chunk of one keyframe pair's shared landmarks -> partial W_j D^-1 W_i^T (6x6) and, on the diagonal pairs, W_i D^-1 b_l.
// HALF 0: rows 0..2 + the right-hand side, HALF 1: rows 3..5.
template <int HALF>
__device__ __forceinline__ void pair_task(const IbaCtx &C, int4 task, bool diag, double *out)
{
    const IbaWin &W = *gen(C.W); const IbaArgs &A = *gen(C.A);
    const int lane = threadIdx.x & 63;
    const int2 *pair_ent = GA(pair_ent) + W.pent_off;
    constexpr int NB = HALF ? 0 : 6;
    double acc[18 + NB + 1];
#pragma unroll
    for (int q = 0; q < 18 + NB; q++) acc[q] = 0.0;
    // (round 4: entries that also carry their landmark -- one dependent L2 round trip fewer per 64 entries -- took 0.4 % off the kernel and
    // DOUBLED the host packing of a one-shot batch, whose scattered 16-byte stores are its hot loop: 5.2 -> 13.9 ms per 32 windows; undone)
    for (int t = task.y + lane; t < task.z; t += 64) {
        const int2 en = pair_ent[t];
        const size_t gi = (size_t)W.e_off + en.x, gj = (size_t)W.e_off + en.y;
        const size_t gl = (size_t)W.pt_off + GA(edge_point)[gi];
        const double *Wi = GA(W) + 18 * gi, *Wj = GA(W) + 18 * gj + 9 * HALF, *Di = GA(Dinv) + 6 * gl;
        const double d00 = Di[0], d10 = Di[1], d11 = Di[2], d20 = Di[3], d21 = Di[4], d22 = Di[5];
        double wi[18];
#pragma unroll
        for (int q = 0; q < 18; q++) wi[q] = Wi[q];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double w0 = Wj[3 * a], w1 = Wj[3 * a + 1], w2 = Wj[3 * a + 2];
            const double y0 = w0 * d00 + w1 * d10 + w2 * d20, y1 = w0 * d10 + w1 * d11 + w2 * d21, y2 = w0 * d20 + w1 * d21 + w2 * d22;
#pragma unroll
            for (int c = 0; c < 6; c++) acc[6 * a + c] += y0 * wi[3 * c] + y1 * wi[3 * c + 1] + y2 * wi[3 * c + 2];
        }
        if (!HALF && diag && en.x == en.y) {                  // W db once per edge (a keyframe's left / right twin edges also pair with each other)
            const double *db = GA(db) + 3 * gl;
#pragma unroll
            for (int a = 0; a < 6; a++) acc[18 + a] += wi[3 * a] * db[0] + wi[3 * a + 1] * db[1] + wi[3 * a + 2] * db[2];
        }
    }
#pragma unroll
    for (int q = 0; q < 18 + NB; q++) acc[q] = wave_sum_f64_dpp(acc[q]);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 18; q++) out[18 * HALF + q] = acc[q];
#pragma unroll
        for (int q = 0; q < NB; q++) out[36 + q] = acc[18 + q];
    }
}

// One LM trial: (Hll + lambda)^-1, S = H + lambda I - sum W D^-1 W^T, LDL^T, landmark back-substitution, oplus into buffer buf^1.
// Returns (team-uniform) ok of the linear solve; the return value of *scale_part is this THREAD's share of computeScale's sum
// (levenberg.cpp:187-194).
__device__ __noinline__ bool iba_trial(const IbaCtx &C, Team &T, int buf, double lambda, double *scale_part)
{
    const IbaWin &W = *gen(C.W); const IbaArgs &A = *gen(C.A);
    const long long t_begin = clock64();
    const int tid = threadIdx.x, n = W.n;
    double *H = GA(H) + W.h_off, *S = GA(S) + W.h_off, *b = GA(b) + W.x_off, *bs = GA(bs) + W.x_off, *x = GA(x) + W.x_off;
    const int *pt_start = GA(pt_start) + W.ptstart_off;
    for (int l = (T.gtid >> 3); l < W.L && (tid & 7) == 0; l += (T.gsize >> 3)) {          // the lane that wrote Hll / bl in iba_build
        const size_t gl = (size_t)W.pt_off + l;
        double *Di = GA(Dinv) + 6 * gl, *db = GA(db) + 3 * gl;
        if (pt_start[l + 1] == pt_start[l]) { for (int i = 0; i < 6; i++) Di[i] = 0.0; for (int i = 0; i < 3; i++) db[i] = 0.0; continue; }
        const double *h = GA(Hll) + 6 * gl, *bl = GA(bl) + 3 * gl;
        const double D[9] = {h[0] + lambda, h[1], h[3], h[1], h[2] + lambda, h[4], h[3], h[4], h[5] + lambda};
        double I[9];
        inv3(D, I);
        Di[0] = I[0]; Di[1] = I[3]; Di[2] = I[4]; Di[3] = I[6]; Di[4] = I[7]; Di[5] = I[8];       // lower triangle of the inverse
        mv3(I, bl, db);
    }
    if (T.g == 0) {
        // (four independent loads in flight per thread: one at a time this copy was 44 dependent L2 round trips per trial at n = 150)
        const int nn = n * n;
        for (int i0 = tid; i0 < nn; i0 += 4 * IBA_THREADS) {
            double h[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + u * IBA_THREADS; h[u] = i < nn ? H[i] : 0.0; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * IBA_THREADS;
                if (i < nn) { const int r = i / n, c = i - r * n; S[i] = h[u] + (r == c ? lambda : 0.0); }
            }
        }
        for (int i = tid; i < n; i += IBA_THREADS) bs[i] = b[i];
    }
    team_sync(T);
    // Schur complement partials: (pair of free keyframes i <= j, chunk of the landmarks both see), two waves per chunk
    {
        const int4 *ptask = GA(pair_task) + W.ptask_off;
        double *ppart = GA(ppart) + 42 * (size_t)W.ptask_off;
        const int nf = W.nfree;
        for (int t = T.gwave; t < 2 * W.nptask; t += T.gwaves) {
            const int4 task = ptask[t >> 1];
            const bool diag = task.w != 0;
            if (t & 1) pair_task<1>(C, task, diag, ppart + 42 * (size_t)(t >> 1));
            else pair_task<0>(C, task, diag, ppart + 42 * (size_t)(t >> 1));
        }
        (void)nf;
    }
    team_sync(T);
    bool ok = true;
    long long t_schur = 0, t_ldlt = 0;
    if (T.g == 0) {
        // combine: block (j, i) of the lower triangle -= sum of the pair's chunks, in order
        const int *pts_ = GA(pair_task_start) + W.ptstart2_off;
        const double *ppart = GA(ppart) + 42 * (size_t)W.ptask_off;
        const int nf = W.nfree;
        for (int idx = tid; idx < W.npairs * 42; idx += IBA_THREADS) {
            const int p = idx / 42, q = idx - 42 * p;
            if (pts_[p + 1] == pts_[p]) continue;
            int i = 0, rem = p;
            while (rem >= nf - i) { rem -= nf - i; i++; }
            const int j = i + rem;
            if (q >= 36 && i != j) continue;
            double s = 0;
            for (int t = pts_[p]; t < pts_[p + 1]; t++) s += ppart[42 * (size_t)t + q];
            const int oi = GA(kf_xoff)[W.kf_off + GA(free_kf)[W.free_off + i]], oj = GA(kf_xoff)[W.kf_off + GA(free_kf)[W.free_off + j]];
            if (q >= 36) bs[oi + q - 36] -= s;
            else { const int a = q / 6, c = q - 6 * a; S[(size_t)(oj + a) * n + oi + c] -= s; }
        }
        __syncthreads();
        t_schur = clock64();
        ok = n > 0 ? ldlt_solve_wg(S, n, n, bs, x, A.max_n) : true;
        if (tid == 0) GA(okflag)[T.w] = ok ? 1 : 0;
        t_ldlt = clock64();
    }
    team_sync(T);
    ok = GA(okflag)[T.w] != 0;
    const double *cur_pts = GA(pts) + buf * A.pts_stride + (size_t)W.pt_off * 3;
    double *new_pts = GA(pts) + (buf ^ 1) * A.pts_stride + (size_t)W.pt_off * 3;
    double part = 0.0;
    for (int l0 = (T.gtid >> 3); l0 < ((W.L + 7) & ~7); l0 += (T.gsize >> 3)) {      // 8 lanes per landmark, whole waves stay in the loop
        const int l = min(l0, W.L - 1), sub = tid & 7;
        const size_t gl = (size_t)W.pt_off + l;
        double *xl = GA(xl) + 3 * gl;
        const bool active = l0 < W.L && pt_start[l + 1] > pt_start[l];
        double cl[3] = {0, 0, 0};
        if (ok && active) {                                        // block_solver.hpp:461-481 (skipped when the pose solve failed: x stays stale)
            for (int e = pt_start[l] + sub; e < pt_start[l + 1]; e += 8) {
                const size_t ge = (size_t)W.e_off + e;
                const int o = GA(kf_xoff)[W.kf_off + GA(edge_kf)[ge]];
                if (o < 0) continue;
                const double *We = GA(W) + 18 * ge, *xp = x + o;
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int a = 0; a < 6; a++) cl[c] -= We[3 * a + c] * xp[a];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) cl[c] = oct_allreduce_f64(cl[c]);
        if (sub != 0 || l0 >= W.L) continue;
        if (ok && active) {
            const double *bl = GA(bl) + 3 * gl, *Di = GA(Dinv) + 6 * gl;
            cl[0] += bl[0]; cl[1] += bl[1]; cl[2] += bl[2];
            xl[0] = Di[0] * cl[0] + Di[1] * cl[1] + Di[3] * cl[2];
            xl[1] = Di[1] * cl[0] + Di[2] * cl[1] + Di[4] * cl[2];
            xl[2] = Di[3] * cl[0] + Di[4] * cl[1] + Di[5] * cl[2];
        }
        for (int a = 0; a < 3; a++) {
            const double dx = active ? xl[a] : 0.0;
            new_pts[3 * l + a] = cur_pts[3 * l + a] + dx;
            if (active) part += dx * (lambda * dx + GA(bl)[3 * gl + a]);
        }
    }
    if (T.g == 0) for (int i = tid; i < n; i += IBA_THREADS) part += x[i] * (lambda * x[i] + b[i]);
    // oplus on the keyframe vertices (ImuCamPose::Update, G2oTypes.cc:192-220; the velocity / bias vertices add)
    const double *cur_kf = GA(kfs) + buf * A.kfs_stride + (size_t)W.kf_off * IBA_KF;
    double *new_kf = GA(kfs) + (buf ^ 1) * A.kfs_stride + (size_t)W.kf_off * IBA_KF;
    double *new_cam = GA(cam) + (buf ^ 1) * A.cam_stride + (size_t)W.kf_off * 24;
    for (int k = T.gsize - 1 - T.gtid; k < W.n_kf; k += T.gsize) {
        double s[IBA_KF];
        for (int i = 0; i < IBA_KF; i++) s[i] = cur_kf[IBA_KF * k + i];
        const int o = GA(kf_xoff)[W.kf_off + k];
        if (o >= 0) {
            const double *dx = x + o;
            double t[3], E[9];
            mv3(s + K_R, dx + 3, t);
            for (int i = 0; i < 3; i++) s[K_T + i] += t[i];
            exp_so3(dx, E, 1e-5, true);
            mm3(s + K_R, E, s + K_R);
            if (GA(kf_imu)[W.kf_off + k]) for (int i = 0; i < 9; i++) s[K_V + i] += dx[6 + i];
        }
        for (int i = 0; i < IBA_KF; i++) new_kf[IBA_KF * k + i] = s[i];
        cam_pose(W, s, new_cam + 24 * k);
    }
    team_sync(T);                              // the new estimates are complete
    *scale_part = part;
    if (GA(prof) && T.g == 0 && threadIdx.x == 0) {
        long long *pf = GA(prof) + 16 * T.w;
        pf[2] += t_schur - t_begin; pf[3] += t_ldlt - t_schur; pf[4] += clock64() - t_ldlt;
    }
    return ok;
}
}  // namespace

__global__ __launch_bounds__(IBA_THREADS) void k_iba_solve(const IbaArgs *Ap)      // the argument block sits in global memory: the phase functions read it through a pointer
{
    const IbaArgs &A = *Ap;
    __shared__ double red[2 * IBA_WAVES];
    // blockIdx -> (window, member): with G > 1 the members of a team sit on ONE XCD (workgroups are dealt round-robin over the 8
    // XCDs), so that the team's scratch stays in that XCD's L2; correctness does not depend on it (agent-scope fences)
    int w, g;
    if (A.G == 1) { w = blockIdx.x; g = 0; }
    else { const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3; w = xcd + 8 * (slot / A.G); g = slot % A.G; }
    if (w >= A.n_windows) return;
    const IbaWin &W = GA(win)[w];
    const int tid = threadIdx.x;
    Team T;
    T.w = w; T.g = g; T.G = A.G; T.gtid = g * IBA_THREADS + tid; T.gsize = A.G * IBA_THREADS; T.gwave = g * IBA_WAVES + (tid >> 6); T.gwaves = A.G * IBA_WAVES;
    T.counter = GA(counters) + w; T.epoch = 0; T.wpart = GA(wpart) + (size_t)w * 2 * IBA_MAXG * 2; T.slot = 0; T.fail = GA(fail) + w;
    IbaCtx C;
    C.W = (GPTR(const IbaWin))&W; C.A = (GPTR(const IbaArgs))Ap;
    C.delta_m = (double)sqrtf(5.991f); C.dsqr_m = C.delta_m * C.delta_m;        // thHuberMono etc. are floats (Optimizer.cc:4893-4896)
    C.delta_s = (double)sqrtf(7.815f); C.dsqr_s = C.delta_s * C.delta_s;
    C.delta_i = sqrt(16.92); C.dsqr_i = C.delta_i * C.delta_i;                  // :4838
    int cur = 0;
    {
        const double *kf0 = GA(kfs) + (size_t)W.kf_off * IBA_KF;
        double *cam0 = GA(cam) + (size_t)W.kf_off * 24;
        for (int k = T.gtid; k < W.n_kf; k += T.gsize) cam_pose(W, kf0 + IBA_KF * k, cam0 + 24 * k);
    }
    team_sync(T);
    long long t_err = 0, t_build = 0;
    const long long t_k0 = clock64();
    double chi = iba_errors(C, T, cur), zero = 0.0;        // computeActiveErrors + activeRobustChi2 (:5046-5047)
    team_sum2(T, chi, zero, red);
    t_err += clock64() - t_k0;
    const double err0 = chi;
    double lambda = A.lambda_init, ni = 2.0, last_chi = chi;
    int nbad = 0, trials = 0, its = 0;
    bool errors_current = true;
    for (int it = 0; it < A.iterations; it++) {            // SparseOptimizer::optimize -> OptimizationAlgorithmLevenberg::solve
        long long t0 = clock64();
        // levenberg.cpp:71 computeActiveErrors: after an ACCEPTED trial the stored errors already are those of the current
        // estimate (same inputs, same code: recomputing gives the same bits), only a rejected trial leaves stale ones behind
        if (it > 0 && !errors_current) { chi = iba_errors(C, T, cur); zero = 0.0; team_sum2(T, chi, zero, red); }
        double current_chi = chi;
        const double ini_chi = chi;
        long long t1 = clock64();
        iba_build(C, T, cur);
        t_err += t1 - t0; t_build += clock64() - t1;
        if (it == 0) { lambda = A.lambda_init; ni = 2.0; nbad = 0; }
        double rho = 0.0;
        int qmax = 0;
        do {
            double scale;
            const bool ok = iba_trial(C, T, cur, lambda, &scale);
            t0 = clock64();
            double tmp = iba_errors(C, T, cur ^ 1);
            team_sum2(T, tmp, scale, red);
            t_err += clock64() - t0;
            last_chi = tmp;
            if (!ok) tmp = DBL_MAX;
            rho = (current_chi - tmp) / (scale + 1e-3);
            if (rho > 0 && isfinite(tmp)) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2.0; current_chi = tmp;
                cur ^= 1;
                errors_current = true; chi = tmp;
            } else {
                lambda *= ni; ni *= 2;
                errors_current = false;
            }
            qmax++; trials++;
        } while (rho < 0 && qmax < A.max_trials && !__hip_atomic_load(T.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        its++;
        if (qmax == A.max_trials || rho == 0 || __hip_atomic_load(T.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if ((ini_chi - current_chi) * 1e3 < ini_chi) nbad++; else nbad = 0;
        if (nbad >= 3) break;
    }
    // outlier gates on the stored chi2, depth of the final estimates (Optimizer.cc:5056-5088)
    const double *cam = GA(cam) + cur * A.cam_stride + (size_t)W.kf_off * 24;
    const double *pts = GA(pts) + cur * A.pts_stride + (size_t)W.pt_off * 3;
    double nout = 0.0;
    for (int e = T.gtid; e < W.E; e += T.gsize) {
        const size_t ge = (size_t)W.e_off + e;
        const double c2 = GA(chi2)[ge];
        bool out;
        if (GA(edge_stereo)[ge] == 1) out = c2 > (double)7.815f;
        else {
            const double *c = cam + 24 * GA(edge_kf)[ge] + 12 * (GA(edge_stereo)[ge] == 2), *X = pts + 3 * GA(edge_point)[ge];
            const bool depth_pos = (c[6] * X[0] + c[7] * X[1] + c[8] * X[2] + c[11]) > 0.0;
            const bool close = GA(edge_close)[ge] != 0;
            out = (c2 > (double)5.991f && !close) || (c2 > (double)(1.5f * 5.991f) && close) || !depth_pos;
        }
        GA(outlier)[ge] = out ? 1 : 0;
        nout += out ? 1.0 : 0.0;
    }
    zero = 0.0;
    team_sum2(T, nout, zero, red);
    if (cur == 1) {                                         // results are read from buffer 0
        double *k0 = GA(kfs) + (size_t)W.kf_off * IBA_KF, *p0 = GA(pts) + (size_t)W.pt_off * 3;
        const double *k1 = k0 + A.kfs_stride, *p1 = p0 + A.pts_stride;
        for (int i = T.gtid; i < W.n_kf * IBA_KF; i += T.gsize) k0[i] = k1[i];
        for (int i = T.gtid; i < W.L * 3; i += T.gsize) p0[i] = p1[i];
    }
    if (GA(prof) && g == 0 && tid == 0) { long long *pf = GA(prof) + 16 * w; pf[0] = t_err; pf[1] = t_build; pf[5] = clock64() - t_k0; }
    if (g == 0 && tid == 0) {
        orbhip_iba_stats &st = GA(stats)[w];
        st.iterations_run = its; st.lm_trials = trials; st.n_outliers = (int)nout;
        st.err = err0; st.err_end = last_chi;
        const float fe = (float)err0, fl = (float)last_chi;
        st.failed = ((2 * fe < fl || isnan(fe) || isnan(fl)) && !A.large) ? 1 : 0;        // :5096
    }
}

// ------------------------------------------------------------------ host side
extern "C" void orbhip_iba_default_params(orbhip_iba_params *p, int large)
{
    if (!p) return;
    p->iterations = large ? 4 : 10;
    p->lambda_init = large ? 1e-2 : 1.0;
    p->large = large ? 1 : 0;
    p->max_trials = 100;
}

namespace {
struct Blob {                        // host image of the constant device data; offsets are stable, pointers taken after the upload
    std::vector<uint8_t> bytes;
    template <typename T> size_t put(const std::vector<T> &v)
    {
        size_t off = (bytes.size() + 255) & ~(size_t)255;
        bytes.resize(off + sizeof(T) * std::max<size_t>(v.size(), 1));
        if (!v.empty()) memcpy(bytes.data() + off, v.data(), sizeof(T) * v.size());
        return off;
    }
    size_t alloc(size_t nbytes)
    {
        size_t off = (bytes.size() + 255) & ~(size_t)255;
        bytes.resize(off + std::max<size_t>(nbytes, 1));
        return off;
    }
};
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
// One TEAM grid per device at a time.  The team barrier needs every workgroup of the grid resident; the grid is sized for an otherwise
// empty device, so two team grids in flight (two contexts / threads / processes) could each hold compute units the other waits for.
// A solve that wants teams takes this lock without blocking -- an in-process mutex per device plus an advisory file lock for other
// processes of this library -- from launch to completion; a solve that does not get it runs one workgroup per window (G = 1: the
// barrier is a __syncthreads, no residency assumption).  Kernels WITHOUT spin barriers running beside a team grid only delay it: they
// finish on their own and free their compute units (the bounded spin, seconds long, is the last line of defence).
static thread_local int g_iba_last_team = 0;
extern "C" int orbhip_inertial_ba_last_team_size(void) { return g_iba_last_team; }

struct IbaTeamLock {
    int device = -1, fd = -1; bool held = false;
    static std::mutex &mu(int device) { static std::mutex m[64]; return m[device & 63]; }
    bool try_acquire(int dev)
    {
        if (!mu(dev).try_lock()) return false;
        char path[64];
        snprintf(path, sizeof(path), "/tmp/.orbhip_team_gpu%d.lock", dev);
        // flock works on a read-only descriptor, so processes of OTHER users can take part whatever the creator's umask left of the
        // mode (fchmod by the creator widens it to 0666 anyway); O_NOFOLLOW: the name is predictable in a world-writable directory.
        fd = open(path, O_RDONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0666);
        if (fd >= 0) (void)fchmod(fd, 0666);
        else if (errno == EEXIST) fd = open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
        if (fd < 0) {
            // only a file system that cannot hold the file at all leaves the in-process mutex as the whole lock; anything else
            // (EACCES, ELOOP: somebody else's file / a symlink planted there) means "lock not acquired": the caller runs G = 1
            if (errno == EROFS || errno == ENOENT) { device = dev; held = true; return true; }
            mu(dev).unlock();
            return false;
        }
        if (flock(fd, LOCK_EX | LOCK_NB) != 0) { close(fd); fd = -1; mu(dev).unlock(); return false; }
        device = dev; held = true;
        return true;
    }
    ~IbaTeamLock()
    {
        if (!held) return;
        if (fd >= 0) { flock(fd, LOCK_UN); close(fd); }
        mu(device).unlock();
    }
};

#define ITRY(e) do { if ((e) != hipSuccess) { orbhip_set_last_error_internal(#e); return ORBHIP_E_HIP; } } while (0)
// Host-side packing of windows [w0, w1) into one IbaPack: SoA arrays, per-keyframe edge lists, per-pair block lists, task lists, edge colours.
// Every offset stored in hw[w] is relative to THIS pack; orbhip_inertial_ba_solve_batch packs chunks of windows on several host threads
// (packing was 2.5x the device time of a batch) and rebases them when it lays the chunks out in the upload blob.
struct IbaPack {
    std::vector<int> kf_xoff, free_kf, edge_kf, edge_point, pt_start, kf_edges, in1, in2, in_color, kf_task_start, pair_task_start;
    std::vector<int2> pair_ent;
    std::vector<int4> kf_task, pair_task;
    std::vector<uint8_t> kf_imu, edge_stereo, edge_close, in_robust;
    std::vector<double> edge_obs, edge_is2, in_pre, in_info, in_info_g, in_info_a, kfs, pts;
    size_t sumKF = 0, sumL = 0, sumE = 0, sumM = 0, sumX = 0, sumH = 0;
    int max_n = 0, rc = ORBHIP_OK;
    const char *err = nullptr;
    std::vector<int> tmp_pstart, tmp_kcount, tmp_kpos, tmp_pcount, tmp_ppos;
    std::vector<std::pair<int, int>> tmp_fe;
};
static int iba_pack_range(const orbhip_iba_window *wins, int w0, int w1, double *const *kf_state_inout, double *const *points_inout, IbaWin *hw, IbaPack &P)
{
    auto &kf_xoff = P.kf_xoff; auto &free_kf = P.free_kf; auto &edge_kf = P.edge_kf; auto &edge_point = P.edge_point; auto &pt_start = P.pt_start;
    auto &kf_edges = P.kf_edges; auto &in1 = P.in1; auto &in2 = P.in2; auto &in_color = P.in_color; auto &kf_task_start = P.kf_task_start;
    auto &pair_task_start = P.pair_task_start; auto &pair_ent = P.pair_ent; auto &kf_task = P.kf_task; auto &pair_task = P.pair_task;
    auto &kf_imu = P.kf_imu; auto &edge_stereo = P.edge_stereo; auto &edge_close = P.edge_close; auto &in_robust = P.in_robust;
    auto &edge_obs = P.edge_obs; auto &edge_is2 = P.edge_is2; auto &in_pre = P.in_pre; auto &in_info = P.in_info; auto &in_info_g = P.in_info_g;
    auto &in_info_a = P.in_info_a; auto &kfs = P.kfs; auto &pts = P.pts;
    auto &tmp_pstart = P.tmp_pstart; auto &tmp_kcount = P.tmp_kcount; auto &tmp_kpos = P.tmp_kpos; auto &tmp_pcount = P.tmp_pcount; auto &tmp_ppos = P.tmp_ppos;
    auto &tmp_fe = P.tmp_fe;
    size_t &sumKF = P.sumKF, &sumL = P.sumL, &sumE = P.sumE, &sumM = P.sumM, &sumX = P.sumX, &sumH = P.sumH;
    int &max_n = P.max_n;
    {
        size_t te = 0, tl = 0, tk = 0;
        for (int w = w0; w < w1; w++) { te += (size_t)std::max(wins[w].n_edges, 0); tl += (size_t)std::max(wins[w].n_points, 0); tk += (size_t)std::max(wins[w].n_kf, 0); }
        edge_kf.reserve(te); edge_point.reserve(te); edge_obs.reserve(3 * te); edge_is2.reserve(te); edge_stereo.reserve(te); edge_close.reserve(te);
        kf_edges.reserve(te); pair_ent.reserve(3 * te); pt_start.reserve(tl + (size_t)(w1 - w0)); pts.reserve(3 * tl); kfs.reserve(IBA_KF * tk);
    }
    for (int w = w0; w < w1; w++) {
        const orbhip_iba_window &g = wins[w];
        if (g.n_kf <= 0 || g.n_points < 0 || g.n_edges < 0 || g.n_inertial < 0 || !g.kf_fixed || !g.kf_imu || !kf_state_inout[w] ||
            (g.n_points && !points_inout[w]) ||
            (g.n_edges && (!g.edge_kf || !g.edge_point || !g.edge_obs || !g.edge_stereo || !g.edge_inv_sigma2)) ||
            (g.n_inertial && (!g.in_kf1 || !g.in_kf2 || !g.in_preint || !g.in_info || !g.in_info_g || !g.in_info_a || !g.in_robust)))
            return ORBHIP_E_BADARG;
        IbaWin &W = hw[w];
        memset(&W, 0, sizeof(W));
        W.n_kf = g.n_kf; W.L = g.n_points; W.E = g.n_edges; W.M = g.n_inertial;
        W.kf_off = (int)sumKF; W.pt_off = (int)sumL; W.e_off = (int)sumE; W.m_off = (int)sumM; W.x_off = (int)sumX; W.h_off = (long long)sumH;
        W.free_off = (int)free_kf.size(); W.ptstart_off = (int)pt_start.size();
        W.kfe_off = (int)kf_edges.size(); W.pent_off = (long long)pair_ent.size();
        W.ktask_off = (int)kf_task.size(); W.ktstart_off = (int)kf_task_start.size();
        W.ptask_off = (int)pair_task.size(); W.ptstart2_off = (int)pair_task_start.size();
        memcpy(W.Rcb, g.Rcb, sizeof(W.Rcb)); memcpy(W.tcb, g.tcb, sizeof(W.tcb));
        W.fx = g.fx; W.fy = g.fy; W.cx = g.cx; W.cy = g.cy; W.bf = g.bf;
        W.cam_model = g.camera_model == 1 ? 1 : 0; memcpy(W.kb, g.kb, sizeof(W.kb));
        W.has_cam2 = g.has_cam2 ? 1 : 0;
        if (W.has_cam2) {                                  // Rcb[1] = Rrl Rcb[0], tcb[1] = Rrl tcb[0] + trl (G2oTypes.cc:60-63)
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) W.Rcb2[3 * r + c] = g.Trl[4 * r] * g.Rcb[c] + g.Trl[4 * r + 1] * g.Rcb[3 + c] + g.Trl[4 * r + 2] * g.Rcb[6 + c];
                W.tcb2[r] = g.Trl[4 * r] * g.tcb[0] + g.Trl[4 * r + 1] * g.tcb[1] + g.Trl[4 * r + 2] * g.tcb[2] + g.Trl[4 * r + 3];
            }
            W.fx2 = g.fx2; W.fy2 = g.fy2; W.cx2 = g.cx2; W.cy2 = g.cy2; W.cam2_model = g.camera2_model == 1 ? 1 : 0; memcpy(W.kb2, g.kb2, sizeof(W.kb2));
        }
        std::vector<int> fidx(g.n_kf, -1);
        int n = 0;
        for (int k = 0; k < g.n_kf; k++) {
            kf_imu.push_back(g.kf_imu[k] ? 1 : 0);
            if (g.kf_fixed[k]) { kf_xoff.push_back(-1); continue; }
            kf_xoff.push_back(n); fidx[k] = W.nfree++; free_kf.push_back(k);
            n += g.kf_imu[k] ? 15 : 6;
        }
        W.n = n;
        if (n > BA_LDLT_MAXN || g.n_inertial > IBA_THREADS) { P.err = "inertial BA: more than 480 keyframe unknowns (32 inertial keyframes)"; return ORBHIP_E_CAPACITY; }
        max_n = std::max(max_n, n);
        // edges: grouped by point (the reference creates them point by point, Optimizer.cc:4914-5034)
        std::vector<int> &pstart = tmp_pstart; pstart.assign(g.n_points + 1, 0);
        std::vector<int> &kcount = tmp_kcount; kcount.assign(W.nfree + 1, 0);
        const size_t e_base = edge_kf.size();
        edge_kf.resize(e_base + g.n_edges); edge_point.resize(e_base + g.n_edges); edge_obs.resize(3 * (e_base + g.n_edges));
        edge_is2.resize(e_base + g.n_edges); edge_stereo.resize(e_base + g.n_edges); edge_close.resize(e_base + g.n_edges);
        for (int e = 0; e < g.n_edges; e++) {
            const int k = g.edge_kf[e], l = g.edge_point[e];
            if (k < 0 || k >= g.n_kf || l < 0 || l >= g.n_points || (e && l < g.edge_point[e - 1])) return ORBHIP_E_BADARG;
            pstart[l + 1]++;
            if (fidx[k] >= 0) kcount[fidx[k] + 1]++;
            edge_kf[e_base + e] = k; edge_point[e_base + e] = l;
            if (g.edge_stereo[e] > 2 || (g.edge_stereo[e] == 2 && !g.has_cam2)) return ORBHIP_E_BADARG;
            edge_stereo[e_base + e] = g.edge_stereo[e];
            edge_close[e_base + e] = g.edge_close ? (g.edge_close[e] ? 1 : 0) : 0;
        }
        if (g.n_edges) {
            memcpy(edge_obs.data() + 3 * e_base, g.edge_obs, 24 * (size_t)g.n_edges);
            memcpy(edge_is2.data() + e_base, g.edge_inv_sigma2, 8 * (size_t)g.n_edges);
        }
        for (int l = 0; l < g.n_points; l++) pstart[l + 1] += pstart[l];
        pt_start.insert(pt_start.end(), pstart.begin(), pstart.end());
        // per-keyframe edge lists (counting sort, edge order kept), cut into chunks of <= IBA_KF_CHUNK edges (one wave pair each)
        for (int f = 0; f < W.nfree; f++) kcount[f + 1] += kcount[f];
        {
            const size_t kbase = kf_edges.size();
            kf_edges.resize(kbase + kcount[W.nfree]);
            std::vector<int> &kpos = tmp_kpos; kpos.assign(kcount.begin(), kcount.end() - 1);
            for (int e = 0; e < g.n_edges; e++) { const int f = fidx[g.edge_kf[e]]; if (f >= 0) kf_edges[kbase + kpos[f]++] = e; }
        }
        for (int f = 0; f < W.nfree; f++) {
            kf_task_start.push_back((int)kf_task.size() - W.ktask_off);
            for (int o = kcount[f]; o < kcount[f + 1]; o += IBA_KF_CHUNK)
                kf_task.push_back(make_int4(f, o, std::min<int>(o + IBA_KF_CHUNK, kcount[f + 1]), 0));
        }
        kf_task_start.push_back((int)kf_task.size() - W.ktask_off);
        W.nktask = (int)kf_task.size() - W.ktask_off;
        // pair lists: for every point, every (i <= j) pair of the free keyframes that see it (two counting passes, point order kept)
        W.npairs = W.nfree * (W.nfree + 1) / 2;
        auto pair_id = [&](int i, int j) { return i * W.nfree - i * (i - 1) / 2 + (j - i); };
        std::vector<int> &pcount = tmp_pcount; pcount.assign(W.npairs + 1, 0);
        std::vector<std::pair<int, int>> &fe = tmp_fe;
        for (int pass = 0; pass < 2; pass++) {
            const size_t pbase = pair_ent.size() - (pass ? (size_t)pcount[W.npairs] : 0);
            std::vector<int> &ppos = tmp_ppos;
            if (pass) ppos.assign(pcount.begin(), pcount.end() - 1);
            for (int l = 0; l < g.n_points; l++) {
                fe.clear();
                for (int e = pstart[l]; e < pstart[l + 1]; e++) if (fidx[g.edge_kf[e]] >= 0) fe.push_back({fidx[g.edge_kf[e]], e});
                for (size_t a2 = 0; a2 < fe.size(); a2++)
                    for (size_t b2 = a2; b2 < fe.size(); b2++) {
                        const bool sw = fe[a2].first > fe[b2].first;
                        const int i = sw ? fe[b2].first : fe[a2].first, j = sw ? fe[a2].first : fe[b2].first;
                        // a keyframe's left and right edges to one point (i == j, two edges) add W_a D^-1 W_b^T AND its transpose to the
                        // diagonal block (g2o keeps ONE Hpl block per (pose, point) pair: both edges accumulate into it)
                        const bool twin = b2 != a2 && i == j;
                        if (!pass) pcount[pair_id(i, j) + 1] += twin ? 2 : 1;
                        else {
                            pair_ent[pbase + ppos[pair_id(i, j)]++] = make_int2(sw ? fe[b2].second : fe[a2].second, sw ? fe[a2].second : fe[b2].second);
                            if (twin) pair_ent[pbase + ppos[pair_id(i, j)]++] = make_int2(fe[b2].second, fe[a2].second);
                        }
                    }
            }
            if (!pass) {
                for (int q = 0; q < W.npairs; q++) pcount[q + 1] += pcount[q];
                pair_ent.resize(pair_ent.size() + pcount[W.npairs]);
            }
        }
        for (int i = 0, q = 0; i < W.nfree; i++)
            for (int j = i; j < W.nfree; j++, q++) {
                pair_task_start.push_back((int)pair_task.size() - W.ptask_off);
                for (int o = pcount[q]; o < pcount[q + 1]; o += IBA_PAIR_CHUNK)
                    pair_task.push_back(make_int4(q, o, std::min<int>(o + IBA_PAIR_CHUNK, pcount[q + 1]), i == j ? 1 : 0));
            }
        pair_task_start.push_back((int)pair_task.size() - W.ptask_off);
        W.nptask = (int)pair_task.size() - W.ptask_off;
        // inertial edges + greedy colouring (edges of one colour share no keyframe)
        std::vector<std::vector<int>> used(g.n_kf);
        std::vector<int> col_count;
        for (int m = 0; m < g.n_inertial; m++) {
            const int k1 = g.in_kf1[m], k2 = g.in_kf2[m];
            if (k1 < 0 || k1 >= g.n_kf || k2 < 0 || k2 >= g.n_kf || k1 == k2 || !g.kf_imu[k1] || !g.kf_imu[k2]) return ORBHIP_E_BADARG;
            int c = 0;
            auto taken = [&](int col) { for (int u : used[k1]) if (u == col) return true; for (int u : used[k2]) if (u == col) return true; return false; };
            while (taken(c)) c++;
            used[k1].push_back(c); used[k2].push_back(c);
            W.ncolors = std::max(W.ncolors, c + 1);
            if ((int)col_count.size() <= c) col_count.resize(c + 1, 0);
            if (c > 255) return ORBHIP_E_CAPACITY;               // (a keyframe with more than 255 inertial edges)
            // colour | rank inside the colour << 8: workgroup 0 deals the edges of a colour to its waves by rank
            in1.push_back(k1); in2.push_back(k2); in_color.push_back(c | (col_count[c]++ << 8)); in_robust.push_back(g.in_robust[m] ? 1 : 0);
        }
        if (g.n_inertial) {
            in_pre.insert(in_pre.end(), g.in_preint, g.in_preint + (size_t)IBA_PRE * g.n_inertial);
            in_info.insert(in_info.end(), g.in_info, g.in_info + 81 * (size_t)g.n_inertial);
            in_info_g.insert(in_info_g.end(), g.in_info_g, g.in_info_g + 9 * (size_t)g.n_inertial);
            in_info_a.insert(in_info_a.end(), g.in_info_a, g.in_info_a + 9 * (size_t)g.n_inertial);
        }
        kfs.insert(kfs.end(), kf_state_inout[w], kf_state_inout[w] + (size_t)IBA_KF * g.n_kf);
        if (g.n_points) pts.insert(pts.end(), points_inout[w], points_inout[w] + 3 * (size_t)g.n_points);
        sumKF += g.n_kf; sumL += g.n_points; sumE += g.n_edges; sumM += g.n_inertial; sumX += n; sumH += (size_t)n * n;
    }
    return ORBHIP_OK;
}
}  // namespace

// A batch of windows resident on the device: the constant part (topology, task lists, preintegrations: everything the host packs) is
// built and uploaded ONCE, a solve only uploads the initial states and launches -- the one-shot entry point below is create + solve +
// download on the context's arena.  (Host packing was 0.5 ms per 13 k-edge window against 60 us of device time.)
struct orbhip_iba_batch {
    orbhip_ctx *ctx;
    int n_windows, max_n;
    bool own;                               // device memory owned by the batch (hipMalloc) or the context's scratch arena (one-shot call)
    uint8_t *d;
    size_t bytes;
    std::vector<IbaWin> hw;
    std::vector<double> kfs, pts;           // initial states, concatenated (host): a solve starts from them
    std::vector<uint8_t> blob;              // host image of the constant part while its upload may still be in flight (one-shot call)
    size_t sumKF, sumL, sumE, sumM, sumX, sumH;
    size_t w_kfs, w_pts, w_xl, w_x, w_W, w_sync, w_stats, w_out, w_prof, w_args;
    IbaArgs A;                              // kernel argument, iteration parameters filled per solve
    double pack_ms;
};

static int iba_create_impl(orbhip_ctx *ctx, const orbhip_iba_window *wins, int n_windows, double *const *kf_state_inout, double *const *points_inout,
                           bool own, orbhip_iba_batch **out)
{
    const auto t_host0 = std::chrono::steady_clock::now();
    std::vector<IbaWin> hw(n_windows);
    // ---- packing: chunks of windows on host threads (windows are independent), then one layout pass
    int nthr = (int)std::thread::hardware_concurrency();
    if (const char *ev = getenv("ORBHIP_IBA_HOST_THREADS")) nthr = atoi(ev);
    nthr = std::max(1, std::min(std::min(nthr, 16), n_windows / 2));
    std::vector<IbaPack> packs(nthr);
    auto chunk_lo = [&](int c) { return (int)((long long)n_windows * c / nthr); };
    {
        std::vector<std::thread> th;
        for (int c = 1; c < nthr; c++)
            th.emplace_back([&, c] { packs[c].rc = iba_pack_range(wins, chunk_lo(c), chunk_lo(c + 1), kf_state_inout, points_inout, hw.data(), packs[c]); });
        packs[0].rc = iba_pack_range(wins, chunk_lo(0), chunk_lo(1), kf_state_inout, points_inout, hw.data(), packs[0]);
        for (auto &t : th) t.join();
    }
    for (int c = 0; c < nthr; c++)
        if (packs[c].rc != ORBHIP_OK) { if (packs[c].err) orbhip_set_last_error_internal(packs[c].err); return packs[c].rc; }
    // rebase every window's offsets by what the chunks in front of it hold
    size_t sumKF = 0, sumL = 0, sumE = 0, sumM = 0, sumX = 0, sumH = 0, n_kftask = 0, n_pairtask = 0;
    int max_n = 0;
    struct Base { size_t kf, l, e, m, x, h, fr, pst, kfe, pent, ktask, ktstart, ptask, ptstart2; };
    std::vector<Base> base(nthr);
    {
        Base b = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < nthr; c++) {
            const IbaPack &P = packs[c];
            base[c] = b;
            for (int w = chunk_lo(c); w < chunk_lo(c + 1); w++) {
                IbaWin &W = hw[w];
                W.kf_off += (int)b.kf; W.pt_off += (int)b.l; W.e_off += (int)b.e; W.m_off += (int)b.m; W.x_off += (int)b.x; W.h_off += (long long)b.h;
                W.free_off += (int)b.fr; W.ptstart_off += (int)b.pst; W.kfe_off += (int)b.kfe; W.pent_off += (long long)b.pent;
                W.ktask_off += (int)b.ktask; W.ktstart_off += (int)b.ktstart; W.ptask_off += (int)b.ptask; W.ptstart2_off += (int)b.ptstart2;
            }
            b.kf += P.sumKF; b.l += P.sumL; b.e += P.sumE; b.m += P.sumM; b.x += P.sumX; b.h += P.sumH; b.fr += P.free_kf.size(); b.pst += P.pt_start.size();
            b.kfe += P.kf_edges.size(); b.pent += P.pair_ent.size(); b.ktask += P.kf_task.size(); b.ktstart += P.kf_task_start.size();
            b.ptask += P.pair_task.size(); b.ptstart2 += P.pair_task_start.size();
            max_n = std::max(max_n, P.max_n);
        }
        sumKF = b.kf; sumL = b.l; sumE = b.e; sumM = b.m; sumX = b.x; sumH = b.h; n_kftask = b.ktask; n_pairtask = b.ptask;
    }
    // the upload blob: every array is the concatenation of the chunks' pieces; the pieces are copied in place by the same threads
    Blob B;
    std::vector<std::function<void(int)>> copies;
    auto lay = [&](auto member, size_t elem_bytes) {
        size_t total = 0;
        std::vector<size_t> offs(nthr);
        for (int c = 0; c < nthr; c++) { offs[c] = total; total += (packs[c].*member).size(); }
        const size_t o = B.alloc(elem_bytes * std::max<size_t>(total, 1));
        copies.push_back([&, member, offs, o, elem_bytes](int c) {
            const auto &v = packs[c].*member;
            if (!v.empty()) memcpy(B.bytes.data() + o + elem_bytes * offs[c], v.data(), elem_bytes * v.size());
        });
        return o;
    };
    const size_t o_win = B.alloc(sizeof(IbaWin) * (size_t)n_windows);
    const size_t o_xoff = lay(&IbaPack::kf_xoff, 4), o_imu = lay(&IbaPack::kf_imu, 1), o_free = lay(&IbaPack::free_kf, 4), o_ekf = lay(&IbaPack::edge_kf, 4),
                 o_ept = lay(&IbaPack::edge_point, 4), o_obs = lay(&IbaPack::edge_obs, 8), o_is2 = lay(&IbaPack::edge_is2, 8), o_est = lay(&IbaPack::edge_stereo, 1),
                 o_ecl = lay(&IbaPack::edge_close, 1), o_pst = lay(&IbaPack::pt_start, 4), o_ked = lay(&IbaPack::kf_edges, 4), o_pre = lay(&IbaPack::pair_ent, 8),
                 o_ktk = lay(&IbaPack::kf_task, 16), o_kts = lay(&IbaPack::kf_task_start, 4), o_ptk = lay(&IbaPack::pair_task, 16),
                 o_pts = lay(&IbaPack::pair_task_start, 4), o_in1 = lay(&IbaPack::in1, 4), o_in2 = lay(&IbaPack::in2, 4), o_col = lay(&IbaPack::in_color, 4),
                 o_rob = lay(&IbaPack::in_robust, 1), o_ipr = lay(&IbaPack::in_pre, 8), o_inf = lay(&IbaPack::in_info, 8), o_ig = lay(&IbaPack::in_info_g, 8),
                 o_ia = lay(&IbaPack::in_info_a, 8);
    std::vector<double> kfs((size_t)IBA_KF * sumKF), pts(3 * sumL);
    {
        auto copy_chunk = [&](int c) {
            for (auto &f : copies) f(c);
            if (!packs[c].kfs.empty()) memcpy(kfs.data() + (size_t)IBA_KF * base[c].kf, packs[c].kfs.data(), 8 * packs[c].kfs.size());
            if (!packs[c].pts.empty()) memcpy(pts.data() + 3 * base[c].l, packs[c].pts.data(), 8 * packs[c].pts.size());
        };
        std::vector<std::thread> th;
        for (int c = 1; c < nthr; c++) th.emplace_back([&, c] { copy_chunk(c); });
        memcpy(B.bytes.data() + o_win, hw.data(), sizeof(IbaWin) * (size_t)n_windows);
        copy_chunk(0);
        for (auto &t : th) t.join();
    }
    const size_t constant_bytes = al256(B.bytes.size());
    // work area
    size_t off = constant_bytes;
    auto take = [&](size_t bytes) { const size_t o = off; off = al256(off + std::max<size_t>(bytes, 8)); return o; };
    const size_t w_kfs = take(16 * IBA_KF * sumKF), w_cam = take(16 * 24 * sumKF), w_pts = take(16 * 3 * sumL), w_err = take(24 * sumE), w_chi = take(8 * sumE),
                 w_W = take(144 * sumE), w_Hll = take(48 * sumL), w_bl = take(24 * sumL), w_Di = take(48 * sumL), w_db = take(24 * sumL), w_xl = take(24 * sumL),
                 w_ierr = take(120 * sumM), w_ichi = take(24 * sumM), w_Jb = take(1728 * sumM), w_OJ = take(1728 * sumM), w_Oe = take(120 * sumM),
                 w_H = take(8 * sumH), w_S = take(8 * sumH), w_b = take(8 * sumX), w_bs = take(8 * sumX), w_x = take(8 * sumX), w_out = take(sumE),
                 w_kpart = take(27 * 8 * n_kftask), w_ppart = take(42 * 8 * n_pairtask),
                 w_stats = take(sizeof(orbhip_iba_stats) * n_windows), w_prof = take(128 * (size_t)n_windows),
                 w_sync = take((4 + 4 + 4) * (size_t)n_windows), w_wpart = take(8 * 2 * IBA_MAXG * 2 * (size_t)n_windows), w_args = take(sizeof(IbaArgs));
    ITRY(hipSetDevice(orbhip_ctx_device_internal(ctx)));
    hipStream_t s = orbhip_ctx_stream_internal(ctx);
    uint8_t *d = nullptr;
    if (own) { if (hipMalloc((void **)&d, off) != hipSuccess) { orbhip_set_last_error_internal("hipMalloc(inertial BA batch)"); return ORBHIP_E_HIP; } }
    else d = (uint8_t *)orbhip_ctx_scratch_internal(ctx, off);
    if (!d) return ORBHIP_E_HIP;
    orbhip_iba_batch *b = new orbhip_iba_batch();
    b->ctx = ctx; b->n_windows = n_windows; b->own = own; b->d = d; b->bytes = off; b->hw = std::move(hw); b->kfs = std::move(kfs); b->pts = std::move(pts);
    b->sumKF = sumKF; b->sumL = sumL; b->sumE = sumE; b->sumM = sumM; b->sumX = sumX; b->sumH = sumH;
    b->w_kfs = w_kfs; b->w_pts = w_pts; b->w_xl = w_xl; b->w_x = w_x; b->w_W = w_W; b->w_sync = w_sync; b->w_stats = w_stats; b->w_out = w_out; b->w_prof = w_prof; b->w_args = w_args;
    b->max_n = std::max(max_n, 32);
    b->blob = std::move(B.bytes);                                    // stays alive until the copy has been waited for
    if (hipMemcpyAsync(d, b->blob.data(), b->blob.size(), hipMemcpyHostToDevice, s) != hipSuccess || (own && hipStreamSynchronize(s) != hipSuccess)) {
        if (own) (void)hipFree(d);
        delete b; orbhip_set_last_error_internal("upload of the inertial BA batch"); return ORBHIP_E_HIP;
    }
    if (own) std::vector<uint8_t>().swap(b->blob);                   // a persistent batch keeps no host copy
    IbaArgs &A = b->A;
    memset(&A, 0, sizeof(A));
#define CP(T, o) (GPTR(const T))(d + (o))
#define WP(T, o) (GPTR(T))(d + (o))
    A.win = CP(IbaWin, o_win); A.kf_xoff = CP(int, o_xoff); A.kf_imu = CP(uint8_t, o_imu); A.free_kf = CP(int, o_free);
    A.edge_kf = CP(int, o_ekf); A.edge_point = CP(int, o_ept); A.edge_obs = CP(double, o_obs); A.edge_is2 = CP(double, o_is2);
    A.edge_stereo = CP(uint8_t, o_est); A.edge_close = CP(uint8_t, o_ecl); A.pt_start = CP(int, o_pst);
    A.kf_edges = CP(int, o_ked); A.pair_ent = CP(int2, o_pre); A.kf_task = CP(int4, o_ktk); A.kf_task_start = CP(int, o_kts);
    A.pair_task = CP(int4, o_ptk); A.pair_task_start = CP(int, o_pts); A.in_kf1 = CP(int, o_in1); A.in_kf2 = CP(int, o_in2);
    A.in_color = CP(int, o_col); A.in_robust = CP(uint8_t, o_rob); A.in_pre = CP(double, o_ipr); A.in_info = CP(double, o_inf);
    A.in_info_g = CP(double, o_ig); A.in_info_a = CP(double, o_ia);
    A.kfs = WP(double, w_kfs); A.cam = WP(double, w_cam); A.pts = WP(double, w_pts);
    A.kfs_stride = (long long)IBA_KF * sumKF; A.cam_stride = 24 * (long long)sumKF; A.pts_stride = 3 * (long long)sumL;
    A.err = WP(double, w_err); A.chi2 = WP(double, w_chi); A.W = WP(double, w_W); A.Hll = WP(double, w_Hll); A.bl = WP(double, w_bl);
    A.Dinv = WP(double, w_Di); A.db = WP(double, w_db); A.xl = WP(double, w_xl); A.ierr = WP(double, w_ierr); A.ichi2 = WP(double, w_ichi);
    A.Jb = WP(double, w_Jb); A.OJ = WP(double, w_OJ); A.Oe = WP(double, w_Oe); A.H = WP(double, w_H); A.S = WP(double, w_S);
    A.b = WP(double, w_b); A.bs = WP(double, w_bs); A.x = WP(double, w_x); A.outlier = WP(uint8_t, w_out); A.stats = WP(orbhip_iba_stats, w_stats);
    A.kpart = WP(double, w_kpart); A.ppart = WP(double, w_ppart);
    A.counters = WP(unsigned, w_sync); A.fail = WP(int, w_sync + 4 * (size_t)n_windows); A.okflag = WP(int, w_sync + 8 * (size_t)n_windows);
    A.wpart = WP(double, w_wpart);
#undef CP
#undef WP
    A.max_n = b->max_n;
    A.n_windows = n_windows;
    b->pack_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    *out = b;
    return ORBHIP_OK;
}

static void iba_destroy_impl(orbhip_iba_batch *b)
{
    if (!b) return;
    if (b->own && b->d) { (void)hipStreamSynchronize(orbhip_ctx_stream_internal(b->ctx)); (void)hipFree(b->d); }
    delete b;
}

static int iba_solve_impl(orbhip_iba_batch *b, const orbhip_iba_params *params)
{
    orbhip_ctx *ctx = b->ctx;
    const int n_windows = b->n_windows;
    uint8_t *d = b->d;
    const size_t sumL = b->sumL, sumE = b->sumE, sumX = b->sumX;
    const bool want_prof = getenv("ORBHIP_IBA_PROF") != nullptr;
    const auto t_host1 = std::chrono::steady_clock::now();
    ITRY(hipSetDevice(orbhip_ctx_device_internal(ctx)));
    hipStream_t s = orbhip_ctx_stream_internal(ctx);
    ITRY(hipMemcpyAsync(d + b->w_kfs, b->kfs.data(), 8 * b->kfs.size(), hipMemcpyHostToDevice, s));
    if (!b->pts.empty()) ITRY(hipMemcpyAsync(d + b->w_pts, b->pts.data(), 8 * b->pts.size(), hipMemcpyHostToDevice, s));
    ITRY(hipMemsetAsync(d + b->w_xl, 0, 24 * sumL + 8, s));
    ITRY(hipMemsetAsync(d + b->w_x, 0, 8 * sumX + 8, s));
    ITRY(hipMemsetAsync(d + b->w_W, 0, 144 * sumE + 8, s));
    ITRY(hipMemsetAsync(d + b->w_sync, 0, 12 * (size_t)n_windows, s));
    IbaArgs A = b->A;
    if (want_prof) { ITRY(hipMemsetAsync(d + b->w_prof, 0, 128 * (size_t)n_windows, s)); A.prof = (GPTR(long long))(d + b->w_prof); }
    A.iterations = params->iterations; A.max_trials = params->max_trials; A.large = params->large; A.lambda_init = params->lambda_init;
    const int device = orbhip_ctx_device_internal(ctx);
    const size_t lds = std::max(ba_ldlt_lds_bytes(A.max_n), sizeof(double) * (size_t)IBA_WAVES * IBA_STAGE + 16 * (size_t)IBA_THREADS);      // the solver's panel area; the build stages inertial edges (and their records) in it
    if (orb_lds_optin((const void *)k_iba_solve, device, lds) != 0) return ORBHIP_E_HIP;
    // team size: every workgroup of a team must be resident (the barrier spins), so teams are only used while the whole grid fits
    // the device at one 1024-thread workgroup per CU; bigger batches run one workgroup per window
    int cus = 0, per_cu = 0;
    ITRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    ITRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_iba_solve, IBA_THREADS, lds));
    const int forced = getenv("ORBHIP_IBA_TEAM") ? atoi(getenv("ORBHIP_IBA_TEAM")) : 0;
    const int slots_per_xcd = std::max(1, (cus / 8) * std::min(per_cu, 1));          // co-resident workgroups per XCD we rely on
    const int win_per_xcd = (n_windows + 7) / 8;
    int G = std::min(IBA_MAXG, slots_per_xcd / win_per_xcd);
    if (forced > 0) G = std::min(forced, G);
    if (G < 2 || per_cu < 1) G = 1;
    IbaTeamLock team_lock;                                           // released when this (synchronous) call returns
    if (G > 1 && !team_lock.try_acquire(device)) G = 1;              // another team grid is in flight on this device
    A.G = G;
    g_iba_last_team = G;
    // the argument block goes to the device (the kernel's phase functions read it through a global pointer; a by-value kernel argument
    // whose address is taken is copied to scratch by every thread)
    const IbaArgs *d_args = reinterpret_cast<const IbaArgs *>(d + b->w_args);
    ITRY(hipMemcpyAsync(d + b->w_args, &A, sizeof(IbaArgs), hipMemcpyHostToDevice, s));
    if (G == 1) {
        hipLaunchKernelGGL(k_iba_solve, dim3(n_windows), dim3(IBA_THREADS), lds, s, d_args);
        ITRY(hipGetLastError());
    } else {
        // the whole grid must be resident (the team barrier spins): a plain launch has the same residency as a cooperative one
        // (MI355X_MICROARCH.md), so the size rule is checked here instead of by hipLaunchCooperativeKernel (whose launches rocprofv3
        // cannot trace without crashing in the runtime's exit handler on this image: profiles/r03_iba_coop_exit_crash.txt)
        if (8 * G * win_per_xcd > cus * per_cu) { orbhip_set_last_error_internal("inertial BA: team grid larger than the device"); return ORBHIP_E_HIP; }
        hipLaunchKernelGGL(k_iba_solve, dim3(8 * G * win_per_xcd), dim3(IBA_THREADS), lds, s, d_args);
        ITRY(hipGetLastError());
    }
    std::vector<int> failv(n_windows);
    ITRY(hipMemcpyAsync(failv.data(), d + b->w_sync + 4 * (size_t)n_windows, 4 * (size_t)n_windows, hipMemcpyDeviceToHost, s));
    ITRY(hipStreamSynchronize(s));
    for (int w = 0; w < n_windows; w++)
        if (failv[w]) { orbhip_set_last_error_internal("inertial BA: a team barrier did not complete (workgroups not co-resident)"); return ORBHIP_E_HIP; }
#ifdef LDLT_PROF
    {
        long long lp[8];
        ITRY(hipMemcpyFromSymbol(lp, HIP_SYMBOL(g_ldlt_prof), sizeof(lp)));
        fprintf(stderr, "[orbhip iba] LDLT cycles (cumulative): load %lld diag %lld rows %lld trailing %lld backsub %lld\n", lp[0], lp[1], lp[2], lp[3], lp[4]);
    }
#endif
    if (want_prof) {
        const auto t_host2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[orbhip iba] %d windows: host packing %.3f ms (at creation), state upload + kernel %.3f ms (%zu B on the device)\n", n_windows, b->pack_ms,
                std::chrono::duration<double, std::milli>(t_host2 - t_host1).count(), b->bytes);
        long long pf[16];
        ITRY(hipMemcpy(pf, d + b->w_prof, 128, hipMemcpyDeviceToHost));
        fprintf(stderr, "[orbhip iba] build phases (workgroup 0, cycles): inertial Jacobians %lld, landmarks %lld, pose chunks %lld, team barrier %lld, Omega J %lld, pose blocks %lld, colours %lld\n",
                pf[8], pf[9], pf[10], pf[11], pf[12], pf[13], pf[14]);
        fprintf(stderr, "[orbhip iba] G=%d window 0 shader cycles: errors %lld, build %lld, prep+schur %lld, ldlt %lld, update %lld, total %lld\n", G, pf[0], pf[1], pf[2], pf[3], pf[4], pf[5]);
    }
    return ORBHIP_OK;
}

static int iba_download_impl(orbhip_iba_batch *b, double *const *kf_state_out, double *const *points_out, uint8_t *const *edge_outlier_out,
                             orbhip_iba_stats *stats_out)
{
    const int n_windows = b->n_windows;
    uint8_t *d = b->d;
    hipStream_t s = orbhip_ctx_stream_internal(b->ctx);
    ITRY(hipSetDevice(orbhip_ctx_device_internal(b->ctx)));
    std::vector<orbhip_iba_stats> st(n_windows);
    std::vector<double> kfo((size_t)IBA_KF * b->sumKF), pto(3 * b->sumL);
    std::vector<uint8_t> outl(b->sumE);
    ITRY(hipMemcpyAsync(st.data(), d + b->w_stats, sizeof(orbhip_iba_stats) * n_windows, hipMemcpyDeviceToHost, s));
    ITRY(hipMemcpyAsync(kfo.data(), d + b->w_kfs, 8 * kfo.size(), hipMemcpyDeviceToHost, s));
    if (b->sumL) ITRY(hipMemcpyAsync(pto.data(), d + b->w_pts, 8 * pto.size(), hipMemcpyDeviceToHost, s));
    if (b->sumE) ITRY(hipMemcpyAsync(outl.data(), d + b->w_out, b->sumE, hipMemcpyDeviceToHost, s));
    ITRY(hipStreamSynchronize(s));
    for (int w = 0; w < n_windows; w++) {
        const IbaWin &W = b->hw[w];
        if (!st[w].failed) {                               // "FAIL LOCAL-INERTIAL BA": the reference returns before any write-back (Optimizer.cc:5096-5100)
            if (kf_state_out && kf_state_out[w]) memcpy(kf_state_out[w], kfo.data() + (size_t)W.kf_off * IBA_KF, 8 * (size_t)IBA_KF * W.n_kf);
            if (W.L && points_out && points_out[w]) memcpy(points_out[w], pto.data() + (size_t)W.pt_off * 3, 24 * (size_t)W.L);
        }
        if (edge_outlier_out && edge_outlier_out[w] && W.E) memcpy(edge_outlier_out[w], outl.data() + W.e_off, W.E);
        if (stats_out) stats_out[w] = st[w];
    }
    return ORBHIP_OK;
}

static bool iba_params_ok(const orbhip_iba_params *p) { return p && p->iterations >= 0 && p->max_trials >= 1 && p->lambda_init > 0; }

extern "C" int orbhip_inertial_ba_solve_batch(orbhip_ctx *ctx, const orbhip_iba_window *wins, int n_windows, const orbhip_iba_params *params,
                                              double *const *kf_state_inout, double *const *points_inout, uint8_t *const *edge_outlier_out,
                                              orbhip_iba_stats *stats_out)
{
    if (!ctx || n_windows < 0 || (n_windows && (!wins || !kf_state_inout || !points_inout)) || !params) return ORBHIP_E_BADARG;
    if (n_windows == 0) return ORBHIP_OK;
    if (!iba_params_ok(params)) return ORBHIP_E_BADARG;
    orbhip_iba_batch *b = nullptr;
    int rc = iba_create_impl(ctx, wins, n_windows, kf_state_inout, points_inout, false, &b);
    if (rc != ORBHIP_OK) return rc;
    rc = iba_solve_impl(b, params);
    if (rc == ORBHIP_OK) rc = iba_download_impl(b, kf_state_inout, points_inout, edge_outlier_out, stats_out);
    iba_destroy_impl(b);
    return rc;
}

extern "C" int orbhip_iba_batch_create(orbhip_ctx *ctx, const orbhip_iba_window *wins, int n_windows, double *const *kf_states, double *const *points,
                                       orbhip_iba_batch **out)
{
    if (!ctx || n_windows <= 0 || !wins || !kf_states || !points || !out) return ORBHIP_E_BADARG;
    return iba_create_impl(ctx, wins, n_windows, kf_states, points, true, out);
}
extern "C" int orbhip_iba_batch_set_states(orbhip_iba_batch *b, double *const *kf_states, double *const *points)
{
    if (!b || !kf_states || !points) return ORBHIP_E_BADARG;
    for (int w = 0; w < b->n_windows; w++) {
        const IbaWin &W = b->hw[w];
        if (!kf_states[w] || (W.L && !points[w])) return ORBHIP_E_BADARG;
        memcpy(b->kfs.data() + (size_t)W.kf_off * IBA_KF, kf_states[w], 8 * (size_t)IBA_KF * W.n_kf);
        if (W.L) memcpy(b->pts.data() + (size_t)W.pt_off * 3, points[w], 24 * (size_t)W.L);
    }
    return ORBHIP_OK;
}
extern "C" int orbhip_iba_batch_solve(orbhip_iba_batch *b, const orbhip_iba_params *params)
{
    if (!b || !iba_params_ok(params)) return ORBHIP_E_BADARG;
    return iba_solve_impl(b, params);
}
extern "C" int orbhip_iba_batch_download(orbhip_iba_batch *b, double *const *kf_states_out, double *const *points_out, uint8_t *const *edge_outlier_out,
                                         orbhip_iba_stats *stats_out)
{
    if (!b) return ORBHIP_E_BADARG;
    return iba_download_impl(b, kf_states_out, points_out, edge_outlier_out, stats_out);
}
extern "C" void orbhip_iba_batch_destroy(orbhip_iba_batch *b) { iba_destroy_impl(b); }
