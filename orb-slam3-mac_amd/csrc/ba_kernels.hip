// ba_kernels.hip -- batched local bundle adjustment on gfx950: the numerical core of
// Optimizer::LocalBundleAdjustment (reference src/Optimizer.cc:1699-2344), i.e. g2o's
// Levenberg-Marquardt + Schur complement (BlockSolver_6_3), for G independent graphs at once.
//
//   B1  SE3Quat exp / oplus                  Thirdparty/g2o/g2o/types/se3quat.h:104-110,223-257
//   B2  residuals + chi2                     include/OptimizableTypes.h:99-110, types_six_dof_expmap.cpp:190-197
//   B3  Jacobians                            src/OptimizableTypes.cpp:139-160, types_six_dof_expmap.cpp:228-274
//   B4  Huber-weighted quadratic form        g2o/core/base_binary_edge.hpp:55-120, robust_kernel_impl.cpp:65-91
//   B5  system layout, lambda on diagonals   g2o/core/block_solver.hpp:502-604
//   B6  Schur complement + reduced solve     g2o/core/block_solver.hpp:354-486, solvers/linear_solver_eigen.h:94-125
//   B7  LM control                           g2o/core/optimization_algorithm_levenberg.cpp:61-194, sparse_optimizer.cpp:354-419
//   B8  two-pass schedule + outlier gates    src/Optimizer.cc:2041-2181
//
// Design: everything FP64.  The LM state machine of every graph lives on the device
// (BaState); the host only "ticks" a fixed kernel sequence (one LM trial per tick per
// graph) and polls one integer.  Edges arrive point-major (the reference's insertion
// order), so per-point assembly needs no atomics; pose blocks use a host-built pose-major
// edge list (deterministic reductions).  The only dense contraction, S -= (W D^-1) W^T, runs
// on the FP64 matrix cores (v_mfma_f64_16x16x4_f64) over a K-padded dense W panel.
#include "orb_internal.h"
#include "wave_dpp.h"
#include "ba_ldlt.h"
#include "ba_camera.h"
// The library is built with -ffp-contract=off for the bit-exact integer / float ORB paths.  The double-precision optimisers are
// compared with the oracle to 1e-4, not bit for bit: let a * b + c contract to v_fma_f64 here (half the FP64 instructions).
#pragma clang fp contract(fast)
#include <cfloat>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <algorithm>
#include <atomic>

struct orbhip_ctx;
hipStream_t orbhip_ctx_stream_internal(orbhip_ctx *c);
int orbhip_ctx_device_internal(orbhip_ctx *c);
int orbhip_ctx_ba_schur_mode_internal(orbhip_ctx *c);

typedef double v4d __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ device structures
struct BaGraphDev {
    int n_poses, n_points, n_edges, nf, n, ld;      // n = 6*nf, ld = n rounded up to 96
    int pose_off, point_off, edge_off, free_off;     // offsets into the concatenated arrays
    int ptstart_off, posestart_off;                  // into pt_start / pose_start (+g extra entries)
    size_t s_off, spart_off, xl_off;                 // element offsets
    int gemm_off, stage_off, n_stages;               // Schur GEMM work lists: first entry of gemm_task / gemm_stage, number of stages
    int gemm_ps;                                     // points per stage (<= GEMM_PS; fewer when ld is large: the LDS panel is [2][gemm_ps][3][ld+16])
    int ks;                                          // split-K factor of the Schur GEMM (ranges of stages)
    int nt16;                                        // 16-column tiles per side (ld / 16)
    int ngrp;                                        // chunk groups (each workgroup keeps <= GEMM_WAVES*GEMM_NCH chunks of GEMM_C tiles in registers)
    double fx, fy, cx, cy, bf;
    int cam_model;                                   // 0 Pinhole, 1 KannalaBrandt8 (monocular edges)
    double kb[4];
    double Trl[7], fx2, fy2, cx2, cy2, kb2[4];       // second camera of a rigid pair (edge type 2, EdgeSE3ProjectXYZToBody)
    int cam2_model;
    // windows with more than 80 free keyframes (n > BA_LDLT_MAXN): global-memory Schur complement and blocked LDL^T
    size_t pair_off, pent_off;                       // into big_pair_start (nf (nf + 1) / 2 + 1 entries) / big_pair_ent
    int cam_off;                                     // >= 0: per-keyframe calibration -- pose i of this graph uses cams[cam_off + pose_cam[pose_off + i]] (round 4)
};
// One calibration (round 4: the reference gives every edge its keyframe's own, Optimizer.cc:1961, :1990-1994, :2021-2023).  The edge
// functions below are templates over the calibration object: a BaGraphDev (one calibration per graph) or a BaCamDev -- same field names.
struct BaCamDev {
    double fx, fy, cx, cy, bf;
    int cam_model;
    double kb[4];
    double Trl[7], fx2, fy2, cx2, cy2, kb2[4];
    int cam2_model;
};

struct BaState {
    double lambda, ni, current_chi, ini_chi, chi_first, chi_last;
    int pass, iter, qmax, nbad;
    int need_build, need_lambda_init, active, cur;   // cur: which pose/point buffer is current
    int ok;                                          // last LDLT status
    int errors_fresh;                                // the stored errors / chi2 are those of the current estimate (the last trial was accepted)
    int iters_run[2], lm_trials, n_outliers;
    int robust;                                      // 0 after setRobustKernel(0) (merge variant, second pass)
    int apply_levels;                                // set when pass 1 starts: k_ba_levels classifies the edges once
    double rho_dbg;
};

struct BaBatch {      // kernel argument (by value)
    int G, max_edges, max_points, max_nf, max_ld;
    const BaGraphDev *gd;
    const BaCamDev *cams; const int *pose_cam;       // per-keyframe calibration tables ([sum of n_cameras], [sumP]); see BaGraphDev::cam_off
    BaState *st;
    // graph topology
    const int *hidx;            // [sumP]
    const int *edge_pose, *edge_point;   // [sumE] local indices
    const double *edge_obs, *edge_is2;   // [sumE*3], [sumE]
    const uint8_t *edge_stereo;
    // a point seen by the same keyframe in both cameras of a rig has TWO edges to one pose; g2o maps both onto the same
    // Hpl block (they accumulate).  edge_dup[e] = 1: an earlier edge of the same point has the same pose (its Hpl share is
    // added by that first edge); edge_next[e] = the next such edge of the chain or -1.
    const uint8_t *edge_dup;
    const int *edge_next;
    const int *pt_start;        // per graph n_points+1
    const int *pose_start;      // per graph nf+1  (free poses only, by hessian index)
    const int *pose_edges;      // [sumE'] edge ids (graph-local) grouped by free pose
    // static edge data once more in that pose-major order, so that k_ba_build_poses streams it (only the point is a gather)
    const int *pm_point, *pm_task; const uint8_t *pm_type; const double *pm_is2, *pm_obs;
    // estimates, double buffered: buffer b at poses + b*sumP*7
    double *poses, *points;
    int sumP, sumL, sumE, sumF;
    // per-edge
    double *err, *chi2, *rho0;
    // system
    double *Hll, *bl, *Dinv, *db;        // [sumL*6] [sumL*3] [sumL*6] [sumL*3]
    double *Hpp, *bp, *bs;               // [sumF*36] [sumF*6] [sumF*6]
    double *Wsp;                         // [tasks][18] Hpl block of edge_task[e], [b*6 + a] = (J_T^T w Omega J_X)[a][b] (first edge of a twin chain: the sum)
    const int *edge_task;                // [sumE] index of the edge's Hpl block in Wsp / gemm_task (batch-global), -1: fixed pose or later twin
    double *Linv;                        // [sumL*6] rows of C^-1 (lower), D = Hll + lambda I = C C^T: l00 l10 l11 l20 l21 l22
    // Schur GEMM work lists (static): the edges with a free pose and no earlier twin, point-major, cut into stages of
    // <= GEMM_PS points and <= GEMM_STAGE_EDGES edges
    const int4 *gemm_task;               // {graph-local edge, 6*h, point, graph}
    const int4 *gemm_stage;              // per stage {first point, number of points, first task, number of tasks}
    const uint32_t *ptmask;              // [sumL] bit t: the point's Hpl column is non-zero inside columns [16t, 16t+16) (static)
    double *S, *Spart;                   // per graph [ld*ld], [ks][ld*ld]
    double *xp, *xl;                     // [sumF*6], [sumL*3]
    double *scale_pt, *scale_pose;       // partial sums of computeScale
    double *chi, *scale, *maxdiag;       // [G]
    int *n_active;                       // [1]
    uint8_t *outlier;                    // [sumE]
    uint8_t *level;                      // [sumE] 1 = excluded from the optimisation (setLevel(1), merge variant)
    // params
    double delta_m, dsqr_m, delta_s, dsqr_s, gate_m, gate_s, user_lambda, tau;
    int iters[2], max_trials;
    int ex2, nr2;                        // merge variant: exclude first-pass outliers / drop the robust kernel in pass 2
    // landmark-sharded solve (optional single-graph mode): this rank owns a slice of every graph's points and their
    // edges, poses are replicated; partial sums meet in xbuf [world][xcount] (slot r = rank r's payload, filled by the caller's
    // all-gather) and are reduced in rank order on every rank, so all ranks take identical LM decisions.
    int rank, world;
    double *xbuf;
    size_t xcount;
    const int *x1_off, *x2_off;          // [G] payload offsets of exchange 1 (Hpp, bp, chi2, max |Hll diag|) and 2 (sum_ks Spart, W db)
    double *bacc;                        // [sumF*6] sum over this rank's edges of W db (k_ba_bschur)
    int *x_abort;                        // [1] any rank saw the abort flag (exchange 3)
    double *edges_total;                 // [G] number of edges over all ranks (>= 50 % outlier rule)
    // big windows (any graph of the batch with n > BA_LDLT_MAXN): per graph, per pair of free poses i <= j (row-major over the upper
    // triangle) the Hpl blocks of the points both see -- the Schur complement is summed per 6x6 block in a fixed order
    int big;
    int pair_schur;                                          // the Schur complement comes from the pair lists (k_ba_schur_big); always with big
    const int *big_pair_start; const int2 *big_pair_ent; const int *big_pair_pt;      // big_pair_pt: the point of every entry     // entries: {Hpl block of pose i, Hpl block of pose j} (batch-global block ids)
    const int2 *big_pair_jr;                                 // of every entry: {Hpl block of pose j, rank of pose i's block in pose i's own (diagonal) list} (k_ba_schur_rows)
    double *big_y, *big_d, *big_U;                           // [sumF*6] forward-substituted right-hand side, pivots; [G][32*32] unscaled diagonal-block columns
    int *big_fail;                                           // [G] a zero / non-finite pivot was met
};

// The calibration the edges of pose `pi` (local index) of graph G project through: the keyframe's own camera when the graph carries a
// camera table, else the graph's single calibration.  A REFERENCE, not a copy: BaGraphDev's calibration fields (fx ... cam2_model) are
// laid out exactly like a BaCamDev, so both cases are one pointer and a field is loaded where it is used (a 30-double copy per edge
// cost the general build kernel its second wave per SIMD: 232 -> 280 registers)
static_assert(offsetof(BaGraphDev, cam2_model) - offsetof(BaGraphDev, fx) == offsetof(BaCamDev, cam2_model) && offsetof(BaGraphDev, kb) - offsetof(BaGraphDev, fx) == offsetof(BaCamDev, kb) &&
              offsetof(BaGraphDev, Trl) - offsetof(BaGraphDev, fx) == offsetof(BaCamDev, Trl) && offsetof(BaGraphDev, kb2) - offsetof(BaGraphDev, fx) == offsetof(BaCamDev, kb2),
              "BaGraphDev's calibration fields must mirror BaCamDev");
template <bool GENERAL>
__device__ __forceinline__ const BaCamDev &ba_cam_ref(const BaBatch &B, const BaGraphDev &G, int pi)
{
    if (GENERAL && G.cam_off >= 0) return B.cams[G.cam_off + B.pose_cam[G.pose_off + pi]];
    return *reinterpret_cast<const BaCamDev *>(&G.fx);
}

// ------------------------------------------------------------------ SE3 helpers (B1)
__device__ __forceinline__ void quat_to_R(const double *q, double *R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void quat_rot(const double *q, const double *v, double *o)
{
    double u0 = q[1] * v[2] - q[2] * v[1], u1 = q[2] * v[0] - q[0] * v[2], u2 = q[0] * v[1] - q[1] * v[0];
    u0 += u0; u1 += u1; u2 += u2;
    o[0] = v[0] + q[3] * u0 + (q[1] * u2 - q[2] * u1);
    o[1] = v[1] + q[3] * u1 + (q[2] * u0 - q[0] * u2);
    o[2] = v[2] + q[3] * u2 + (q[0] * u1 - q[1] * u0);
}
__device__ __forceinline__ void quat_norm_rot(double *q)
{
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__device__ __forceinline__ void R_to_quat(const double *R, double *q)
{
    // Eigen::Quaterniond(Matrix3d): trace branch, else the largest diagonal element picks (i,j,k); written out
    // per case so that every index is a compile-time constant
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > (i == 0 ? R[0] : R[4])) i = 2;
        if (i == 0) {          // j = 1, k = 2
            t = sqrt(R[0] - R[4] - R[8] + 1.0);
            q[0] = 0.5 * t; t = 0.5 / t;
            q[3] = (R[7] - R[5]) * t; q[1] = (R[3] + R[1]) * t; q[2] = (R[6] + R[2]) * t;
        } else if (i == 1) {   // j = 2, k = 0
            t = sqrt(R[4] - R[8] - R[0] + 1.0);
            q[1] = 0.5 * t; t = 0.5 / t;
            q[3] = (R[2] - R[6]) * t; q[2] = (R[7] + R[5]) * t; q[0] = (R[1] + R[3]) * t;
        } else {               // j = 0, k = 1
            t = sqrt(R[8] - R[0] - R[4] + 1.0);
            q[2] = 0.5 * t; t = 0.5 / t;
            q[3] = (R[3] - R[1]) * t; q[0] = (R[2] + R[6]) * t; q[1] = (R[5] + R[7]) * t;
        }
    }
}
// T_new = exp(u) * T  (VertexSE3Expmap::oplusImpl)
__device__ __forceinline__ void se3_oplus(const double *u, const double *pose, double *out)
{
    const double om0 = u[0], om1 = u[1], om2 = u[2];
    const double theta = sqrt(om0 * om0 + om1 * om1 + om2 * om2);
    const double O[9] = {0, -om2, om1, om2, 0, -om0, -om1, om0, 0};
    double O2[9], R[9], V[9];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += O[3 * i + k] * O[3 * k + j];
            O2[3 * i + j] = s;
        }
    }
    if (theta < 0.00001) {
#pragma unroll
        for (int i = 0; i < 9; i++) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        double sn, cs;
        sincos(theta, &sn, &cs);                             // one argument reduction for both (the same values sin() and cos() return)
        const double a = sn / theta, b = (1 - cs) / (theta * theta);
        const double c = (theta - sn) / (theta * theta * theta);
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const double I = (i % 4 == 0 ? 1.0 : 0.0);
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    double qe[4], te[3], rt[3], qn[4];
    R_to_quat(R, qe);
#pragma unroll
    for (int i = 0; i < 3; i++) te[i] = V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
    quat_norm_rot(qe);
    quat_rot(qe, pose + 4, rt);
    const double *b = pose;
    qn[3] = qe[3] * b[3] - qe[0] * b[0] - qe[1] * b[1] - qe[2] * b[2];
    qn[0] = qe[3] * b[0] + qe[0] * b[3] + qe[1] * b[2] - qe[2] * b[1];
    qn[1] = qe[3] * b[1] + qe[1] * b[3] + qe[2] * b[0] - qe[0] * b[2];
    qn[2] = qe[3] * b[2] + qe[2] * b[3] + qe[0] * b[1] - qe[1] * b[0];
    quat_norm_rot(qn);
    out[0] = qn[0]; out[1] = qn[1]; out[2] = qn[2]; out[3] = qn[3];
    out[4] = te[0] + rt[0]; out[5] = te[1] + rt[1]; out[6] = te[2] + rt[2];
}

// ------------------------------------------------------------------ second camera (EdgeSE3ProjectXYZToBody)
// SE3Quat::operator* (se3quat.h:104-110): o = a * b
__device__ __forceinline__ void se3_mul(const double *a, const double *b, double *o)
{
    double rt[3], q[4];
    quat_rot(a, b + 4, rt);
    q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    q[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    q[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    quat_norm_rot(q);
    o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
    o[4] = a[4] + rt[0]; o[5] = a[5] + rt[1]; o[6] = a[6] + rt[2];
}
// EdgeSE3ProjectXYZToBody::computeError (OptimizableTypes.h:121-126): obs - cam2.project((mTrl * T_lw).map(X)); P = that point
template <class CAM>
__device__ __forceinline__ void tobody_error(const CAM &g, const double *pose, const double *X, const double *obs, double *P, double *err)
{
    double Trw[7], uv[2];
    se3_mul(g.Trl, pose, Trw);
    quat_rot(Trw, X, P);
    P[0] += Trw[4]; P[1] += Trw[5]; P[2] += Trw[6];
    cam_project(g.fx2, g.fy2, g.cx2, g.cy2, g.cam2_model, g.kb2, P, uv);
    err[0] = obs[0] - uv[0]; err[1] = obs[1] - uv[1]; err[2] = 0;
}
// EdgeSE3ProjectXYZToBody::linearizeOplus (OptimizableTypes.cpp:192-213); rows 2 of Jx / Jt zeroed
template <class CAM>
__device__ __forceinline__ void tobody_jacobians(const CAM &g, const double *pose, const double *X, double *Jx, double *Jt)
{
    double Trw[7], Xl[3], Xr[3], J[6], Rrw[9], Rrl[9], M[6];
    se3_mul(g.Trl, pose, Trw);
    quat_rot(pose, X, Xl); Xl[0] += pose[4]; Xl[1] += pose[5]; Xl[2] += pose[6];
    quat_rot(g.Trl, Xl, Xr); Xr[0] += g.Trl[4]; Xr[1] += g.Trl[5]; Xr[2] += g.Trl[6];
    cam_project_jac(g.fx2, g.fy2, g.cam2_model, g.kb2, Xr, J);
    quat_to_R(Trw, Rrw); quat_to_R(g.Trl, Rrl);
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            Jx[3 * r + c] = -(J[3 * r] * Rrw[c] + J[3 * r + 1] * Rrw[3 + c] + J[3 * r + 2] * Rrw[6 + c]);
            M[3 * r + c] = J[3 * r] * Rrl[c] + J[3 * r + 1] * Rrl[3 + c] + J[3 * r + 2] * Rrl[6 + c];
        }
    const double x = Xl[0], y = Xl[1], z = Xl[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const double m0 = M[3 * r], m1 = M[3 * r + 1], m2 = M[3 * r + 2];
        Jt[6 * r + 0] = -(-m1 * z + m2 * y); Jt[6 * r + 1] = -(m0 * z - m2 * x); Jt[6 * r + 2] = -(-m0 * y + m1 * x);
        Jt[6 * r + 3] = -m0; Jt[6 * r + 4] = -m1; Jt[6 * r + 5] = -m2;
    }
#pragma unroll
    for (int k = 6; k < 9; k++) Jx[k] = 0;
#pragma unroll
    for (int k = 12; k < 18; k++) Jt[k] = 0;
}
// z of the edge's camera-frame point (isDepthPositive of the three edge types)
template <class CAM>
__device__ __forceinline__ double edge_depth(const CAM &g, const double *pose, const double *X, int type)
{
    double P[3];
    if (type == 2) { double Trw[7]; se3_mul(g.Trl, pose, Trw); quat_rot(Trw, X, P); return P[2] + Trw[6]; }
    quat_rot(pose, X, P);
    return P[2] + pose[6];
}

// ------------------------------------------------------------------ edge math (B2, B3)
template <bool KB = true, class CAM = BaGraphDev>      // KB = false: Pinhole only (the KannalaBrandt8 branch and its registers compile away)
__device__ __forceinline__ void edge_error(const CAM &g, const double *pose, const double *X, const double *obs,
                                           int stereo, double *P, double *err)
{
    quat_rot(pose, X, P);
    P[0] += pose[4]; P[1] += pose[5]; P[2] += pose[6];
    if (KB && !stereo && g.cam_model == 1) {   // KannalaBrandt8::project, KannalaBrandt8.cpp:52-69; atan2f as the float rounding of the double atan2 (see oracle/ba_oracle.c)
        const double x2y2 = P[0] * P[0] + P[1] * P[1];
        const double theta = (double)(float)atan2((double)sqrtf((float)x2y2), (double)(float)P[2]);
        const double psi = (double)(float)atan2((double)(float)P[1], (double)(float)P[0]);
        const double t2 = theta * theta, t3 = theta * t2, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
        const double r = theta + g.kb[0] * t3 + g.kb[1] * t5 + g.kb[2] * t7 + g.kb[3] * t9;
        err[0] = obs[0] - (g.fx * r * cos(psi) + g.cx);
        err[1] = obs[1] - (g.fy * r * sin(psi) + g.cy);
        err[2] = 0;
        return;
    }
    if (!stereo) {
        err[0] = obs[0] - (g.fx * P[0] / P[2] + g.cx);
        err[1] = obs[1] - (g.fy * P[1] / P[2] + g.cy);
        err[2] = 0;
    } else {   // float invz / float bf, types_six_dof_expmap.cpp:190-197
        const float invz = (float)(1.0 / P[2]);
        const float bff = (float)g.bf;
        const double r0 = P[0] * invz * g.fx + g.cx;
        err[0] = obs[0] - r0;
        err[1] = obs[1] - (P[1] * invz * g.fy + g.cy);
        err[2] = obs[2] - (r0 - (double)__fmul_rn(bff, invz));
    }
}

// Jacobians at camera-frame point P with rotation R.  Jx: D x 3, Jt: D x 6 (row-major)
template <bool KB = true, class CAM = BaGraphDev>
__device__ __forceinline__ void edge_jacobians(const CAM &g, const double *P, const double *R, int stereo, double *Jx, double *Jt)
{
    const double x = P[0], y = P[1], z = P[2];
    if (KB && !stereo && g.cam_model == 1) {   // KannalaBrandt8::projectJac, KannalaBrandt8.cpp:166-195
        const double x2 = x * x, y2 = y * y, z2 = z * z, r2 = x2 + y2, r = sqrt(r2), r3 = r2 * r;
        const double theta = atan2(r, z);
        const double t2 = theta * theta, t3 = t2 * theta, t4 = t2 * t2, t5 = t4 * theta, t6 = t2 * t4, t7 = t6 * theta, t8 = t4 * t4, t9 = t8 * theta;
        const double f = theta + t3 * g.kb[0] + t5 * g.kb[1] + t7 * g.kb[2] + t9 * g.kb[3];
        const double fd = 1 + 3 * g.kb[0] * t2 + 5 * g.kb[1] * t4 + 7 * g.kb[2] * t6 + 9 * g.kb[3] * t8;
        const double J00 = g.fx * (fd * z * x2 / (r2 * (r2 + z2)) + f * y2 / r3);
        const double J10 = g.fy * (fd * z * y * x / (r2 * (r2 + z2)) - f * y * x / r3);
        const double J01 = g.fx * (fd * z * y * x / (r2 * (r2 + z2)) - f * y * x / r3);
        const double J11 = g.fy * (fd * z * y2 / (r2 * (r2 + z2)) + f * x2 / r3);
        const double J02 = -g.fx * fd * x / (r2 + z2), J12 = -g.fy * fd * y / (r2 + z2);
        for (int c = 0; c < 3; c++) {                // Jx = -projectJac * R  (OptimizableTypes.cpp:139-160)
            Jx[c] = -(J00 * R[c] + J01 * R[3 + c] + J02 * R[6 + c]);
            Jx[3 + c] = -(J10 * R[c] + J11 * R[3 + c] + J12 * R[6 + c]);
        }
        // Jt = -projectJac * [0 z -y 1 0 0; -z 0 x 0 1 0; y -x 0 0 0 1]
        Jt[0] = -(-J01 * z + J02 * y); Jt[1] = -(J00 * z - J02 * x); Jt[2] = -(-J00 * y + J01 * x); Jt[3] = -J00; Jt[4] = -J01; Jt[5] = -J02;
        Jt[6] = -(-J11 * z + J12 * y); Jt[7] = -(J10 * z - J12 * x); Jt[8] = -(-J10 * y + J11 * x); Jt[9] = -J10; Jt[10] = -J11; Jt[11] = -J12;
        return;
    }
    if (!stereo) {
        const double iz = 1.0 / z;
        const double p00 = -(g.fx / z), p02 = g.fx * x / (z * z), p11 = -(g.fy / z), p12 = g.fy * y / (z * z);
        (void)iz;
        for (int c = 0; c < 3; c++) {
            Jx[c] = p00 * R[c] + p02 * R[6 + c];
            Jx[3 + c] = p11 * R[3 + c] + p12 * R[6 + c];
        }
        // SE3deriv = [0 z -y 1 0 0; -z 0 x 0 1 0; y -x 0 0 0 1]
        Jt[0] = p02 * y;            Jt[1] = p00 * z - p02 * x;  Jt[2] = -p00 * y;  Jt[3] = p00; Jt[4] = 0;   Jt[5] = p02;
        Jt[6] = -p11 * z + p12 * y; Jt[7] = -p12 * x;           Jt[8] = p11 * x;   Jt[9] = 0;   Jt[10] = p11; Jt[11] = p12;
    } else {
        const double z2 = z * z, fx = g.fx, fy = g.fy, bf = g.bf;
        for (int c = 0; c < 3; c++) {
            Jx[c] = -fx * R[c] / z + fx * x * R[6 + c] / z2;
            Jx[3 + c] = -fy * R[3 + c] / z + fy * y * R[6 + c] / z2;
            Jx[6 + c] = Jx[c] - bf * R[6 + c] / z2;
        }
        Jt[0] = x * y / z2 * fx; Jt[1] = -(1 + (x * x / z2)) * fx; Jt[2] = y / z * fx;
        Jt[3] = -1. / z * fx; Jt[4] = 0; Jt[5] = x / z2 * fx;
        Jt[6] = (1 + y * y / z2) * fy; Jt[7] = -x * y / z2 * fy; Jt[8] = -x / z * fy;
        Jt[9] = 0; Jt[10] = -1. / z * fy; Jt[11] = y / z2 * fy;
        Jt[12] = Jt[0] - bf * y / z2; Jt[13] = Jt[1] + bf * x / z2; Jt[14] = Jt[2];
        Jt[15] = Jt[3]; Jt[16] = 0; Jt[17] = Jt[5] - bf / z2;
    }
}

__device__ __forceinline__ void huber(double e, double delta, double dsqr, double *rho0, double *rho1)
{
    if (e <= dsqr) { *rho0 = e; *rho1 = 1.; }
    else { const double s = sqrt(e); *rho0 = 2 * s * delta - dsqr; *rho1 = delta / s; }
}

// ------------------------------------------------------------------ kernels
// which: 0 -> evaluate at the CURRENT estimate for graphs that need a (re)build;
//        1 -> evaluate at the TRIAL estimate for every active graph.
// GENERAL = false: every graph of the batch is Pinhole without second-camera edges and without twin edges (the usual local BA)
template <bool GENERAL>
__global__ __launch_bounds__(256) void k_ba_errors(BaBatch B, int which)
{
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    if (!st.active || (which == 0 && (!st.need_build || (st.errors_fresh && B.world == 1)))) return;      // after an accepted trial the stored errors ARE the current ones
    const BaGraphDev &G = B.gd[g];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= G.n_edges) return;
    const int buf = which == 0 ? st.cur : (st.cur ^ 1);
    const int ge = G.edge_off + e;
    if (B.level[ge]) { B.rho0[ge] = 0; return; }      // not an active edge: its stored error / chi2 stay as last computed
    const double *pose = B.poses + ((size_t)buf * B.sumP + G.pose_off + B.edge_pose[ge]) * 7;
    const double *X = B.points + ((size_t)buf * B.sumL + G.point_off + B.edge_point[ge]) * 3;
    double P[3], er[3];
    const int type = B.edge_stereo[ge], stereo = type == 1;          // 0 mono, 1 stereo, 2 second camera (ToBody)
    const BaCamDev &cam = ba_cam_ref<GENERAL>(B, G, B.edge_pose[ge]);
    if (GENERAL && type == 2) tobody_error(cam, pose, X, B.edge_obs + 3 * (size_t)ge, P, er);
    else edge_error<GENERAL>(cam, pose, X, B.edge_obs + 3 * (size_t)ge, stereo, P, er);
    const double chi2 = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * B.edge_is2[ge];
    double r0, r1;
    if (!st.robust) { r0 = chi2; r1 = 1.; }
    else if (stereo) huber(chi2, B.delta_s, B.dsqr_s, &r0, &r1); else huber(chi2, B.delta_m, B.dsqr_m, &r0, &r1);
    B.err[3 * (size_t)ge] = er[0]; B.err[3 * (size_t)ge + 1] = er[1]; B.err[3 * (size_t)ge + 2] = er[2];
    B.chi2[ge] = chi2;
    B.rho0[ge] = r0;
}

// Deterministic per-graph sums: chi = sum rho0 over edges (activeRobustChi2, sparse_optimizer.cpp:100-114);
// mode 1 additionally sums computeScale partials (levenberg.cpp:187-194).
__global__ __launch_bounds__(1024) void k_ba_reduce(BaBatch B, int which)
{
    // 1024 threads, four loads in flight per thread, DPP tree per wave, the 16 wave sums added in order: a fixed association (round 1: 256
    // threads walking 78 dependent loads each and an 8-step LDS tree -- 21 us of a single-window LM tick, twice per tick)
    __shared__ double red[16];
    const int g = blockIdx.x, tid = threadIdx.x;
    const BaState &st = B.st[g];
    if (!st.active || (which == 0 && (!st.need_build || (st.errors_fresh && B.world == 1)))) return;      // after an accepted trial the stored errors ARE the current ones
    const BaGraphDev &G = B.gd[g];
    auto block_sum = [&](double v) {
        v = wave_sum_f64_dpp(v);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        double t = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) t += red[w];
        return t;
    };
    {
        const double *r = B.rho0 + G.edge_off;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int e = tid;
        for (; e + 3072 < G.n_edges; e += 4096) { s0 += r[e]; s1 += r[e + 1024]; s2 += r[e + 2048]; s3 += r[e + 3072]; }
        for (; e < G.n_edges; e += 1024) s0 += r[e];
        const double chi = block_sum((s0 + s1) + (s2 + s3));
        if (tid == 0) B.chi[g] = chi;
    }
    if (which == 0 && tid == 0) B.st[g].apply_levels = 0;      // k_ba_levels (earlier in this tick) has consumed it
    if (which == 1) {
        double t = 0;
        for (int l = tid; l < G.n_points; l += 1024) t += B.scale_pt[G.point_off + l];
        if (B.rank == 0) for (int h = tid; h < G.nf; h += 1024) t += B.scale_pose[G.free_off + h];   // replicated: counted once
        const double sc = block_sum(t);
        if (tid == 0) B.scale[g] = sc;
    }
}

// buildSystem, landmark side: 16 lanes per point share its (contiguous) edges -- one edge per lane and trip, so the
// Jacobians of a point's ~10 observations are evaluated side by side and their scattered Hpl writes are in flight
// together; Hll / bl are reduced over the 16 lanes in a fixed (butterfly) order.
// Hll (sym 6), bl, and the edge's Hpl block Wsp[e][6*b + a] = (J_T^T w Omega J_X)[a][b].
template <bool GENERAL>
__global__ __launch_bounds__(256) void k_ba_build_points(BaBatch B)
{
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    if (!st.active || !st.need_build) return;
    const BaGraphDev &G = B.gd[g];
    const int sub = threadIdx.x & 15;
    const int l = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int *ps = B.pt_start + G.ptstart_off;
    if (l >= G.n_points) return;                      // whole 16-lane groups leave together
    const int e0 = ps[l], e1 = ps[l + 1];
    const double *X = B.points + ((size_t)st.cur * B.sumL + G.point_off + l) * 3;
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // H (6, upper) then b (3)
    for (int e = e0 + sub; e < e1; e += 16) {
        const int ge = G.edge_off + e;
        const int pi = B.edge_pose[ge];
        const int hi = B.hidx[G.pose_off + pi];
        const double *pose = B.poses + ((size_t)st.cur * B.sumP + G.pose_off + pi) * 7;
        const int type = B.edge_stereo[ge], stereo = type == 1;
        double P[3], R[9], Jx[9], Jt[18];
#pragma unroll
        for (int k = 6; k < 9; k++) Jx[k] = 0;
#pragma unroll
        for (int k = 12; k < 18; k++) Jt[k] = 0;                                     // monocular edge: third row empty
        const BaCamDev &cam = ba_cam_ref<GENERAL>(B, G, pi);
        if (GENERAL && type == 2) tobody_jacobians(cam, pose, X, Jx, Jt);
        else {
            quat_rot(pose, X, P);
            P[0] += pose[4]; P[1] += pose[5]; P[2] += pose[6];
            quat_to_R(pose, R);
            edge_jacobians<GENERAL>(cam, P, R, stereo, Jx, Jt);
        }
        const double chi2 = B.chi2[ge];
        double r0, r1;
        if (!st.robust) r1 = 1.;
        else if (stereo) huber(chi2, B.delta_s, B.dsqr_s, &r0, &r1); else huber(chi2, B.delta_m, B.dsqr_m, &r0, &r1);
        const double w = B.level[ge] ? 0.0 : r1 * B.edge_is2[ge];          // level-1 edge: contributes nothing (its Hpl block is zeroed)
        const double es[3] = {B.err[3 * (size_t)ge], B.err[3 * (size_t)ge + 1], stereo ? B.err[3 * (size_t)ge + 2] : 0.0};
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double j0 = Jx[3 * d], j1 = Jx[3 * d + 1], j2 = Jx[3 * d + 2];
            const double we = -w * es[d];
            acc[6] += j0 * we; acc[7] += j1 * we; acc[8] += j2 * we;
            acc[0] += j0 * w * j0; acc[1] += j0 * w * j1; acc[2] += j0 * w * j2;
            acc[3] += j1 * w * j1; acc[4] += j1 * w * j2; acc[5] += j2 * w * j2;
        }
        if (hi >= 0 && (!GENERAL || !B.edge_dup[ge])) {
            double Wb[18];
#pragma unroll
            for (int bb = 0; bb < 3; bb++)
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    double h = 0;
#pragma unroll
                    for (int d = 0; d < 3; d++) h += Jt[6 * d + a] * w * Jx[3 * d + bb];
                    Wb[6 * bb + a] = h;
                }
            for (int e2 = GENERAL ? B.edge_next[ge] : -1; e2 >= 0; e2 = B.edge_next[G.edge_off + e2]) {      // the same keyframe's other camera (rare)
                const int g2 = G.edge_off + e2;
                const int type2 = B.edge_stereo[g2], st2 = type2 == 1;
                double P2[3], R2[9], Jx2[9], Jt2[18];
                for (int k = 6; k < 9; k++) Jx2[k] = 0;
                for (int k = 12; k < 18; k++) Jt2[k] = 0;
                if (type2 == 2) tobody_jacobians(cam, pose, X, Jx2, Jt2);
                else {
                    quat_rot(pose, X, P2);
                    P2[0] += pose[4]; P2[1] += pose[5]; P2[2] += pose[6];
                    quat_to_R(pose, R2);
                    edge_jacobians(cam, P2, R2, st2, Jx2, Jt2);
                }
                double q0, q1;
                if (!st.robust) q1 = 1.;
                else if (st2) huber(B.chi2[g2], B.delta_s, B.dsqr_s, &q0, &q1); else huber(B.chi2[g2], B.delta_m, B.dsqr_m, &q0, &q1);
                const double w2 = B.level[g2] ? 0.0 : q1 * B.edge_is2[g2];
                for (int bb = 0; bb < 3; bb++)
                    for (int a = 0; a < 6; a++) {
                        double h = 0;
                        for (int d = 0; d < 3; d++) h += Jt2[6 * d + a] * w2 * Jx2[3 * d + bb];
                        Wb[6 * bb + a] += h;
                    }
            }
            // (round 4: parking the blocks of a workgroup's 16 points in LDS and writing the contiguous range out with consecutive lanes on
            // consecutive 16 bytes was SLOWER: 0.55 against 0.42 ms per 256 windows -- two more barriers and the LDS round trip)
            double *wo = B.Wsp + (size_t)B.edge_task[ge] * 18;
#pragma unroll
            for (int k = 0; k < 18; k++) wo[k] = Wb[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc[k] = row16_allreduce_f64_dpp(acc[k]);          // the point's 16 lanes = one DPP row
    }
    if (sub == 0) {
        double *Ho = B.Hll + (size_t)(G.point_off + l) * 6, *bo = B.bl + (size_t)(G.point_off + l) * 3;
#pragma unroll
        for (int i = 0; i < 6; i++) Ho[i] = acc[i];
        bo[0] = acc[6]; bo[1] = acc[7]; bo[2] = acc[8];
    }
}

// buildSystem, pose side: one wave per free pose over its edges (pose-major list).
template <bool GENERAL>
__global__ __launch_bounds__(64) void k_ba_build_poses(BaBatch B)
{
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    if (!st.active || !st.need_build) return;
    const BaGraphDev &G = B.gd[g];
    const int h = blockIdx.x;
    if (h >= G.nf) return;
    const int lane = threadIdx.x;
    const int *qs = B.pose_start + G.posestart_off;
    const int *pe = B.pose_edges + G.edge_off;
    double acc[27];
    for (int i = 0; i < 27; i++) acc[i] = 0;
    // every edge of this list has the same pose; the edge's error / chi2 are recomputed from the pose-major copy of its static data
    // (bit-identical to k_ba_errors: same inputs, same code) instead of being gathered from the point-major arrays
    const int pose_i = qs[h] < qs[h + 1] ? B.edge_pose[G.edge_off + pe[qs[h]]] : 0;
    const double *pose = B.poses + ((size_t)st.cur * B.sumP + G.pose_off + pose_i) * 7;
    const BaCamDev &cam = ba_cam_ref<GENERAL>(B, G, pose_i);
    for (int k = qs[h] + lane; k < qs[h + 1]; k += 64) {
        const size_t gk = (size_t)G.edge_off + k;
        const double *X = B.points + ((size_t)st.cur * B.sumL + G.point_off + B.pm_point[gk]) * 3;
        const int type = B.pm_type[gk], stereo = type == 1;
        double P[3], R[9], Jx[9], Jt[18], es[3];
#pragma unroll
        for (int q = 12; q < 18; q++) Jt[q] = 0;                  // monocular edge: third row empty (the loops below run all 3 rows)
        if (GENERAL && type == 2) { tobody_error(cam, pose, X, B.pm_obs + 3 * gk, P, es); tobody_jacobians(cam, pose, X, Jx, Jt); }
        else {
            edge_error<GENERAL>(cam, pose, X, B.pm_obs + 3 * gk, stereo, P, es);
            quat_to_R(pose, R);
            edge_jacobians<GENERAL>(cam, P, R, stereo, Jx, Jt);
        }
        const double is2 = B.pm_is2[gk];
        const double chi2 = (es[0] * es[0] + es[1] * es[1] + es[2] * es[2]) * is2;
        double r0, r1;
        if (!st.robust) r1 = 1.;
        else if (stereo) huber(chi2, B.delta_s, B.dsqr_s, &r0, &r1); else huber(chi2, B.delta_m, B.dsqr_m, &r0, &r1);
        const double w = (B.ex2 && B.level[G.edge_off + pe[k]]) ? 0.0 : r1 * is2;      // levels exist only in the merge variant
#pragma unroll
        for (int d = 0; d < 3; d++) {                             // compile-time indices: acc stays in registers
            const double we = -w * es[d];
            int idx = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const double ja = Jt[6 * d + a];
                acc[21 + a] += ja * we;
#pragma unroll
                for (int bb = a; bb < 6; bb++) acc[idx++] += ja * w * Jt[6 * d + bb];
            }
        }
    }
    for (int i = 0; i < 27; i++) {
        double v = acc[i];
        v = wave_sum_f64_dpp(v);
        acc[i] = v;
    }
    if (lane == 0) {
        double *Hp = B.Hpp + (size_t)(G.free_off + h) * 36;
        int idx = 0;
        for (int a = 0; a < 6; a++)
            for (int bb = a; bb < 6; bb++) { Hp[6 * a + bb] = acc[idx]; Hp[6 * bb + a] = acc[idx]; idx++; }
        for (int a = 0; a < 6; a++) B.bp[(size_t)(G.free_off + h) * 6 + a] = acc[21 + a];
    }
}

// computeLambdaInit (levenberg.cpp:171-185): max |diag| over pose and landmark blocks
__global__ __launch_bounds__(256) void k_ba_maxdiag(BaBatch B)
{
    __shared__ double red[256];
    const int g = blockIdx.x, tid = threadIdx.x;
    const BaState &st = B.st[g];
    if (!st.active || !st.need_build || !st.need_lambda_init) return;
    const BaGraphDev &G = B.gd[g];
    const int *ps = B.pt_start + G.ptstart_off;
    double m = 0;
    if (B.world == 1)                                   // sharded: Hpp is still a partial sum here (k_ba_shard_sum1 finishes the max)
        for (int h = tid; h < G.nf; h += 256)
            for (int a = 0; a < 6; a++) m = fmax(m, fabs(B.Hpp[(size_t)(G.free_off + h) * 36 + 7 * a]));
    for (int l = tid; l < G.n_points; l += 256)
        if (ps[l + 1] > ps[l]) {
            const double *H = B.Hll + (size_t)(G.point_off + l) * 6;
            m = fmax(m, fmax(fabs(H[0]), fmax(fabs(H[3]), fabs(H[5]))));
        }
    red[tid] = m;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if (tid < d) red[tid] = fmax(red[tid], red[tid + d]); __syncthreads(); }
    if (tid == 0) B.maxdiag[g] = red[0];
}

// start of an outer LM iteration (levenberg.cpp:71-92) for graphs that just (re)built
__global__ void k_ba_pretrial(BaBatch B)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B.G) return;
    BaState &st = B.st[g];
    if (!st.active || !st.need_build) return;
    st.current_chi = B.chi[g];
    st.ini_chi = st.current_chi;
    if (st.chi_first < 0) st.chi_first = st.current_chi;
    if (st.need_lambda_init) {
        st.lambda = B.user_lambda > 0 ? B.user_lambda : B.tau * B.maxdiag[g];
        st.ni = 2; st.nbad = 0;
        st.need_lambda_init = 0;
    }
    st.qmax = 0;
    st.need_build = 0;
}

// per trial: D = Hll + lambda I, D^-1 (symmetric), db = D^-1 bl   (block_solver.hpp:395-404)
__global__ __launch_bounds__(256) void k_ba_point_prep(BaBatch B)
{
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= G.n_points) return;
    const int *ps = B.pt_start + G.ptstart_off;
    double *Di = B.Dinv + (size_t)(G.point_off + l) * 6, *db = B.db + (size_t)(G.point_off + l) * 3;
    if (ps[l + 1] == ps[l]) { for (int i = 0; i < 6; i++) { Di[i] = 0; B.Linv[(size_t)(G.point_off + l) * 6 + i] = 0; } db[0] = db[1] = db[2] = 0; return; }
    const double *H = B.Hll + (size_t)(G.point_off + l) * 6;
    const double a00 = H[0] + st.lambda, a01 = H[1], a02 = H[2], a11 = H[3] + st.lambda, a12 = H[4], a22 = H[5] + st.lambda;
    const double c0 = a11 * a22 - a12 * a12, c1 = a12 * a02 - a01 * a22, c2 = a01 * a12 - a11 * a02;
    const double id = 1.0 / (a00 * c0 + a01 * c1 + a02 * c2);
    const double i00 = c0 * id, i01 = c1 * id, i02 = c2 * id;
    const double i11 = (a00 * a22 - a02 * a02) * id, i12 = (a02 * a01 - a00 * a12) * id, i22 = (a00 * a11 - a01 * a01) * id;
    Di[0] = i00; Di[1] = i01; Di[2] = i02; Di[3] = i11; Di[4] = i12; Di[5] = i22;
    {   // D = C C^T (Cholesky), Linv = C^-1: the Schur GEMM multiplies Z = C^-1 W, so that W^T D^-1 W = Z^T Z
        const double c00 = sqrt(a00), l00 = 1.0 / c00;
        const double c10 = a01 * l00, c20 = a02 * l00;
        const double c11 = sqrt(a11 - c10 * c10), l11 = 1.0 / c11;
        const double c21 = (a12 - c20 * c10) * l11;
        const double c22 = sqrt(a22 - c20 * c20 - c21 * c21), l22 = 1.0 / c22;
        const double l10 = -c10 * l00 * l11, l21 = -c21 * l11 * l22, l20 = -(l21 * c10 + l22 * c20) * l00;
        double *Lo = B.Linv + (size_t)(G.point_off + l) * 6;
        Lo[0] = l00; Lo[1] = l10; Lo[2] = l11; Lo[3] = l20; Lo[4] = l21; Lo[5] = l22;
    }
    const double *b = B.bl + (size_t)(G.point_off + l) * 3;
    db[0] = i00 * b[0] + i01 * b[1] + i02 * b[2];
    db[1] = i01 * b[0] + i11 * b[1] + i12 * b[2];
    db[2] = i02 * b[0] + i12 * b[1] + i22 * b[2];
}

// Schur GEMM on the FP64 matrix cores:  S_sub = W^T D^-1 W = Z^T Z  with  Z = C^-1 W,  D = C C^T  (k_ba_point_prep), over the
// K-padded dense panel Z[4l+b][c] (K = 4 per point: 3 + pad), v_mfma_f64_16x16x4_f64:
//   A[i][k] (lane i=l&15,k=l>>4), B[k][j] (lane j=l&15,k=l>>4), C/D 4 regs: col = l&15, row = (l>>4) + 4*reg.
// One 512-thread workgroup (8 waves, up to 256 VGPRs each) owns a range of STAGES (<= GEMM_PS points, <= GEMM_STAGE_EDGES Hpl
// blocks) of one graph.  The dense panel exists only in LDS: per stage every thread fetches ONE column of ONE sparse Hpl block
// (3 doubles of Wsp, loaded a stage ahead into registers), multiplies it by the point's C^-1 and scatters it into the cleared stage
// buffer -- HBM sees the 144-byte blocks (0.7 GB per 256 graphs) instead of a 79 %-zero dense panel (3.5 GB), and both MFMA
// fragments come straight from the same LDS panel (no D^-1 arithmetic between load and MFMA).  The upper 16x16 output tiles stay
// in accumulator registers for the whole launch.
// Tile ownership: every row strip of the upper triangle is cut into CHUNKS of GEMM_C consecutive tiles; chunk i belongs to wave
// i % 8, slot i / 8 (GEMM_NCH slots per wave).  Block sparsity: a point is seen by ~10 of ~48 free keyframes; which (point, chunk,
// tile) combinations of a stage hold data is decided once per stage by all 64 lanes (lane = 8*point + chunk) and three ballots;
// a hit chunk loads its A fragment and GEMM_C B fragments back to back and issues the MFMAs of its non-empty tiles.
// LDS rows are padded to ld+16 doubles so that the K-rows of a fragment fall into different bank halves.
#define GEMM_PS 8
#define GEMM_C 3               // tiles per chunk
#define GEMM_NCH 8             // chunks per wave
#define GEMM_WAVES 8
#define GEMM_LDS_PAD 16        // = 32 dwords mod 64
#define GEMM_LDS_TAIL 32       // a short chunk reads up to GEMM_C-1 tiles past the end of a row (never multiplied)
#define GEMM_STAGE_EDGES 85    // 6 column tasks per Hpl block, one task per thread
#ifndef GEMM_TARGET_WGS
#define GEMM_TARGET_WGS 256     // one resident workgroup per CU (VGPR-bound); more splits only add Spart traffic (512: -2 %)
#endif
#define GEMM_PANEL_LDS_BYTES (126 * 1024)   // of the CU's 160 KB: the rest holds the raw blocks, task descriptors, C^-1 rows, masks
static_assert(GEMM_PS * GEMM_NCH <= 64 && GEMM_NCH * GEMM_C <= 32, "one lane per (stage point, chunk); one valid bit per tile");
static_assert(6 * GEMM_STAGE_EDGES <= 64 * GEMM_WAVES && 6 * GEMM_PS <= 64 * GEMM_WAVES, "one column task per thread");
static inline __host__ __device__ int gemm_strip_chunks(int nt, int tr) { return (nt - tr + GEMM_C - 1) / GEMM_C; }
__global__ __launch_bounds__(64 * GEMM_WAVES) void k_ba_schur_gemm(BaBatch B)
{
    extern __shared__ double glds[];               // [2][gemm_ps][3][ld + pad] Z rows of the stage, double buffered (+ tail)
    __shared__ __attribute__((aligned(16))) double s_raw[2][GEMM_STAGE_EDGES * 18];   // the stage's Hpl blocks as they lie in Wsp
    __shared__ __attribute__((aligned(16))) int4 s_task[2][GEMM_STAGE_EDGES];
    __shared__ __attribute__((aligned(16))) double s_linv[2][GEMM_PS][6];
    __shared__ __attribute__((aligned(16))) uint32_t s_mask[3][GEMM_PS];
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const int nt = G.nt16;
    if ((int)blockIdx.x >= G.ks * G.ngrp) return;
    const int ksi = blockIdx.x / G.ngrp, grp = blockIdx.x - ksi * G.ngrp;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int ld = G.ld, ldp = ld + GEMM_LDS_PAD;
    // this wave's chunks: offsets of the row tile / first column tile inside an LDS row (wave-uniform -> scalar registers)
    int aoff[GEMM_NCH], boffc[GEMM_NCH];
    uint32_t cvalid = 0;                                 // tiles that exist: bit c*GEMM_C + j
    v4d acc[GEMM_NCH * GEMM_C];
#pragma unroll
    for (int c = 0; c < GEMM_NCH; c++) {
        int id = (grp * GEMM_NCH + c) * GEMM_WAVES + wv, tr = 0;
        while (tr < nt && id >= gemm_strip_chunks(nt, tr)) { id -= gemm_strip_chunks(nt, tr); tr++; }
        if (tr < nt) {
            const int tc = tr + id * GEMM_C, len = min(GEMM_C, nt - tc);
            aoff[c] = 16 * tr; boffc[c] = 16 * tc; cvalid |= ((1u << len) - 1u) << (c * GEMM_C);
        } else { aoff[c] = 0; boffc[c] = 0; }
#pragma unroll
        for (int j = 0; j < GEMM_C; j++) acc[c * GEMM_C + j] = (v4d){0, 0, 0, 0};
    }
    // the same table per LANE for the once-per-stage hit evaluation: lane = 8*point + chunk decides (point, chunk) for all
    // GEMM_PS x GEMM_NCH = 64 combinations at once; the ballots replace ~20 scalar instructions per (point, chunk)
    uint32_t lrow = 0, lcolbit[GEMM_C];
    {
        const int c = lane % GEMM_NCH;
        int id = (grp * GEMM_NCH + c) * GEMM_WAVES + wv, tr = 0;
        while (tr < nt && id >= gemm_strip_chunks(nt, tr)) { id -= gemm_strip_chunks(nt, tr); tr++; }
        const int tc = tr + id * GEMM_C;
#pragma unroll
        for (int j = 0; j < GEMM_C; j++) lcolbit[j] = (tr < nt && tc + j < nt) ? 1u << (tc + j) : 0u;
        if (tr < nt) lrow = 1u << tr;
    }
    const int per = (G.n_stages + G.ks - 1) / G.ks;
    const int sb = ksi * per, se = min(G.n_stages, sb + per);
    const int4 *stg = B.gemm_stage + G.stage_off;          // {first point, points, first task, tasks}
    const int4 *tsk = B.gemm_task + G.gemm_off;            // {edge, 6*h, point, -}
    const double live = lk < 3 ? 1.0 : 0.0;                // K-row 3 is the pad: A is zero there, B re-reads row 2 (finite)
    const int boff = min(lk, 2) * ldp + li;
    const size_t stage_doubles = (size_t)G.gemm_ps * 3 * ldp;
    // stage s -> LDS by asynchronous global -> LDS copies (no staging registers): its Hpl blocks (contiguous in Wsp), task
    // descriptors, C^-1 rows and occupancy masks; issued two stages ahead of the multiply
    auto dma16 = [&](const void *src, void *dst, int bytes) {          // wave-strided 1-KiB pieces, 16 B per lane
        for (int piece = wv; piece * 1024 < bytes; piece += GEMM_WAVES)
            if (piece * 1024 + lane * 16 < bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)src + piece * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)((char *)dst + piece * 1024), 16, 0, 0);
    };
    auto load_stage = [&](int s) {
        const int4 sd = stg[s];
        const int b2 = s & 1;
        dma16(B.Wsp + (size_t)(G.gemm_off + sd.z) * 18, &s_raw[b2][0], sd.w * 144);
        dma16(tsk + sd.z, &s_task[b2][0], sd.w * 16);
        if (wv == 0) {
            if (lane * 16 < sd.y * 48)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)(B.Linv + (size_t)(G.point_off + sd.x) * 6) + lane * 16),
                                                 (__attribute__((address_space(3))) void *)&s_linv[b2][0][0], 16, 0, 0);
        } else if (wv == 1) {
            if (lane < sd.y)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(B.ptmask + G.point_off + sd.x + lane),
                                                 (__attribute__((address_space(3))) void *)&s_mask[s % 3][0], 4, 0, 0);
        }
    };
    auto clear = [&](int buf) {                            // the stage buffer starts from zero
        double2 *z = reinterpret_cast<double2 *>(glds + buf * stage_doubles);
        for (int i = tid; i < (int)(stage_doubles / 2); i += 64 * GEMM_WAVES) z[i] = make_double2(0.0, 0.0);
    };
    auto scatter = [&](int s, int buf) {                   // Z = C^-1 W, one column of one block per thread
        const int b2 = s & 1, ntask = stg[s].w, pt0 = stg[s].x;
        if (tid < 6 * ntask) {
            const int i = tid / 6, a = tid - 6 * i;
            const int4 t = s_task[b2][i];
            const double *w = &s_raw[b2][18 * i + a];
            const double w0 = w[0], w1 = w[6], w2 = w[12];
            const int wp = t.z - pt0;
            const double *L = s_linv[b2][wp];
            double *dst = glds + buf * stage_doubles + (size_t)wp * 3 * ldp + t.y + a;
            dst[0] = L[0] * w0;
            dst[ldp] = L[1] * w0 + L[2] * w1;
            dst[2 * ldp] = L[3] * w0 + L[4] * w1 + L[5] * w2;
        }
    };
    if (sb < se) {
        load_stage(sb);
        if (sb + 1 < se) load_stage(sb + 1);
        clear(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        scatter(sb, 0);
    }
    int cur = 0;
    for (int s = sb; s < se; s++, cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's copies of stage s+1 have landed
        __syncthreads();                                    // stage s is scattered; everyone is done with the other buffer and has its copies
        if (s + 1 < se) clear(cur ^ 1);
        __syncthreads();
        if (s + 1 < se) { scatter(s + 1, cur ^ 1); }
        if (s + 2 < se) load_stage(s + 2);                  // reuses the raw / task / C^-1 slots of stage s (scattered one iteration ago)
        const int np = stg[s].y;
        const double *stage = glds + cur * stage_doubles;
        // which (point, chunk, tile) combinations of this stage hold data: one evaluation per lane, GEMM_C ballots
        unsigned long long hit[GEMM_C];
        {
            const int pl = lane / GEMM_NCH;
            const uint32_t m = pl < np ? s_mask[s % 3][pl] : 0u;
            const bool rh = (m & lrow) != 0;
#pragma unroll
            for (int j = 0; j < GEMM_C; j++) hit[j] = __ballot(rh && (m & lcolbit[j]));
        }
        for (int p = 0; p < np; p++) {
            uint32_t tj[GEMM_C], any = 0;
#pragma unroll
            for (int j = 0; j < GEMM_C; j++) { tj[j] = (uint32_t)(hit[j] >> (GEMM_NCH * p)) & ((1u << GEMM_NCH) - 1u); any |= tj[j]; }
            if (any == 0) continue;                         // the point touches none of this wave's chunks
            const double *rows = stage + (size_t)p * 3 * ldp + boff;
#pragma unroll
            for (int c = 0; c < GEMM_NCH; c++) {
                if (any & (1u << c)) {
                    const double az = rows[aoff[c]];
                    double bf[GEMM_C];
#pragma unroll
                    for (int j = 0; j < GEMM_C; j++) bf[j] = rows[boffc[c] + 16 * j];   // tiles past nt: pad garbage, never multiplied
                    __builtin_amdgcn_sched_barrier(0);          // all 1 + GEMM_C LDS reads in flight before the first wait
                    const double a = live * az;
#pragma unroll
                    for (int j = 0; j < GEMM_C; j++)
                        if (tj[j] & (1u << c))                  // the fragments are already in registers: an empty tile costs one scalar test
                            acc[c * GEMM_C + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bf[j], acc[c * GEMM_C + j], 0, 0, 0);
                }
            }
        }
    }
    double *Sp = B.Spart + G.spart_off + (size_t)ksi * ld * ld;
#pragma unroll
    for (int c = 0; c < GEMM_NCH; c++) {
#pragma unroll
        for (int j = 0; j < GEMM_C; j++) {
            if ((cvalid >> (c * GEMM_C + j)) & 1u) {
#pragma unroll
                for (int r = 0; r < 4; r++) Sp[(size_t)(aoff[c] + lk + 4 * r) * ld + boffc[c] + 16 * j + li] = acc[c * GEMM_C + j][r];
            }
        }
    }
}

// S = blockdiag(Hpp + lambda I) - sum_ks Spart (mirrored from the upper tiles); padding rows -> identity
__global__ __launch_bounds__(256) void k_ba_schur_finish(BaBatch B)
{
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= G.ld * G.ld) return;
    const int r = idx / G.ld, c = idx - r * G.ld;
    double v = 0;
    if (r >= G.n || c >= G.n) v = (r == c) ? 1.0 : 0.0;
    else {
        if (r / 6 == c / 6) v = B.Hpp[(size_t)(G.free_off + r / 6) * 36 + (r % 6) * 6 + (c % 6)] + (r == c ? st.lambda : 0.0);
        const int ur = (r / 16 <= c / 16) ? r : c, uc = (r / 16 <= c / 16) ? c : r;   // upper-tile source
        double s = 0;
        if (B.world == 1) {
            const double *Sp = B.Spart + G.spart_off + (size_t)ur * G.ld + uc;
            for (int k = 0; k < G.ks; k++) s += Sp[(size_t)k * G.ld * G.ld];
        } else {
            const double *X = B.xbuf + B.x2_off[g] + (size_t)ur * G.ld + uc;          // every rank's sum_ks Spart, rank order
            for (int r = 0; r < B.world; r++) s += X[(size_t)r * B.xcount];
        }
        v -= s;
    }
    B.S[G.s_off + idx] = v;
}

// bs = bp - sum_{edges of pose} W_e * db_point   (block_solver.hpp:409-416,435-438), one wave per free pose
__global__ __launch_bounds__(64) void k_ba_bschur(BaBatch B)
{
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const int h = blockIdx.x;
    if (h >= G.nf) return;
    const int lane = threadIdx.x;
    const int *qs = B.pose_start + G.posestart_off;
    const int *pe = B.pose_edges + G.edge_off;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int k = qs[h] + lane; k < qs[h + 1]; k += 64) {
        const int task = B.pm_task[(size_t)G.edge_off + k];                // pose-major copies: only the block and db are gathers
        if (task < 0) continue;                                            // one Hpl block per (point, pose)
        const int l = B.pm_point[(size_t)G.edge_off + k];
        const double *db = B.db + (size_t)(G.point_off + l) * 3;
        const double *w = B.Wsp + (size_t)task * 18;
        for (int a = 0; a < 6; a++) acc[a] += w[a] * db[0] + w[6 + a] * db[1] + w[12 + a] * db[2];
    }
    for (int a = 0; a < 6; a++) {
        double v = acc[a];
        v = wave_sum_f64_dpp(v);
        acc[a] = v;
    }
    if (lane == 0)
        for (int a = 0; a < 6; a++) {
            B.bacc[(size_t)(G.free_off + h) * 6 + a] = acc[a];
            B.bs[(size_t)(G.free_off + h) * 6 + a] = B.bp[(size_t)(G.free_off + h) * 6 + a] - acc[a];      // sharded: redone by k_ba_shard_sum2
        }
}

__global__ __launch_bounds__(1024) void k_ba_ldlt(BaBatch B)
{
    const int g = blockIdx.x;
    BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const bool ok = ldlt_solve_wg(B.S + G.s_off, G.ld, G.n, B.bs + (size_t)G.free_off * 6, B.xp + (size_t)G.free_off * 6, B.max_ld);
    if (threadIdx.x == 0) st.ok = ok ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Windows with more than 80 free keyframes (n = 6 nf > BA_LDLT_MAXN; the reference has no cap: Optimizer.cc:1703-1819 takes
// every covisible keyframe, the merge variant :6255 two whole neighbourhoods).  The reduced system no longer fits the
// single-workgroup kernels, so it is built and factored in global memory by many workgroups:
//   k_ba_schur_big      S_ij = sum over the points seen by free poses i and j of W_i D^-1 W_j^T, one wave per 6x6 block pair,
//                       fixed summation order (block_solver.hpp:381-432 restated per output block)
//   k_ba_big_diag / _rows / _trail   right-looking blocked LDL^T, 32-column panels: diagonal block (one wave), the rows below
//                       (one row per thread, many workgroups), rank-32 trailing update (64x64 tiles, many workgroups)
//   k_ba_big_backsub    D^-1, L^T x = y
// The arithmetic per entry is the LDS-panel kernel's (same ldlt_rows), so small and big windows factor alike.
#define BA_BIG_MAXN 4096
// LPP = lanes per pair: 16 (four pairs per wave; batches) or 64 (one pair per wave: with a handful of windows the chip is empty and the
// per-pair chain -- entries / lanes dependent gather rounds -- is what a tick waits for)
template <int LPP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ba_schur_big(BaBatch B, int nblk)
{
    constexpr int PPW = 64 / LPP;                                  // pairs per wave
    // FOUR block pairs per wave, 16 lanes each: the 36 sums of a pair are reduced inside its 16-lane DPP row (4 rotate-add steps
    // instead of 6 scan steps + a readlane over the wave) and four pairs share the issue slots -- 2.6x fewer instructions per pair
    // than one wave per pair, which made this kernel faster than the MFMA panel GEMM for every batch size.
    // blockIdx -> (graph, block of 4 pairs): all blocks of a graph run on ONE XCD (workgroups are dealt round-robin over the 8 XCDs), one
    // graph after the other per XCD, so that the graph's Hpl blocks (2.9 MB at 50 x 2000 x 10; every block is read ~11 times, once per
    // pair it belongs to) stay in that XCD's 4 MB L2 instead of being fetched again from HBM
    int g, blk;
    if (B.G >= 8) { const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3; g = xcd + 8 * (slot / nblk); blk = slot - (slot / nblk) * nblk; }
    else { g = blockIdx.x / nblk; blk = blockIdx.x - g * nblk; }               // few graphs: every graph over the whole chip
    if (g >= B.G) return;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const int nf = G.nf;
    const int lane = threadIdx.x, sub = lane & (LPP - 1);
    const int pair = blk * PPW + lane / LPP;
    if (pair >= nf * (nf + 1) / 2) return;                          // whole rows leave together (row-local DPP below)
    int i = 0, bp = pair;
    while (bp >= nf - i) { bp -= nf - i; i++; }
    const int j = i + bp;
    const int *ps = B.big_pair_start + G.pair_off + pair;
    const int2 *ent = B.big_pair_ent + G.pent_off;
    const int *entl = B.big_pair_pt + G.pent_off;
    double acc[36];
#pragma unroll
    for (int k = 0; k < 36; k++) acc[k] = 0.0;
    // the list entry (two block indices + point) of the NEXT trip is fetched a trip ahead: index load and block gathers are two dependent
    // L2 round trips, this takes the first one off the chain
    const int e_end = ps[1];
    int en = ps[0] + sub;
    int2 tn = make_int2(0, 0); int ln = 0;
    if (en < e_end) { tn = ent[en]; ln = entl[en]; }
    for (int e = en; e < e_end; e += LPP) {
        const int2 t = tn;
        const int l = ln;
        if (e + LPP < e_end) { tn = ent[e + LPP]; ln = entl[e + LPP]; }
        const double *Di = B.Dinv + (size_t)(G.point_off + l) * 6;       // (Hll + lambda I)^-1, upper triangle
        const double i00 = Di[0], i01 = Di[1], i02 = Di[2], i11 = Di[3], i12 = Di[4], i22 = Di[5];
        const double *wa = B.Wsp + (size_t)t.x * 18, *wb = B.Wsp + (size_t)t.y * 18;
        double y0[6], y1[6], y2[6];                                     // Y = W_a D^-1 (block_solver.hpp:398-407), then acc += Y W_b^T
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const double a0 = wa[a], a1 = wa[6 + a], a2 = wa[12 + a];
            y0[a] = a0 * i00 + a1 * i01 + a2 * i02;
            y1[a] = a0 * i01 + a1 * i11 + a2 * i12;
            y2[a] = a0 * i02 + a1 * i12 + a2 * i22;
        }
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const double b0 = wb[c], b1 = wb[6 + c], b2 = wb[12 + c];
#pragma unroll
            for (int r = 0; r < 6; r++) acc[6 * r + c] += y0[r] * b0 + y1[r] * b1 + y2[r] * b2;
        }
    }
    // the finished block goes straight into S = blockdiag(Hpp + lambda I) - sum, both triangles (no partial-sum buffer, no finish
    // kernel on this path; rows / columns >= n of the padded matrix are never read by the factorisations)
    double *S = B.S + G.s_off;
    const double lambda = st.lambda;
    const double *Hd = B.Hpp + (size_t)(G.free_off + i) * 36;
#pragma unroll
    for (int k = 0; k < 36; k++) {
        const double v = LPP == 16 ? row16_allreduce_f64_dpp(acc[k]) : wave_sum_f64_dpp(acc[k]);      // every lane of the group holds the sum
        if (sub == 0) {
            const int r = k / 6, c = k - 6 * r;
            const double out = (i == j ? Hd[k] + (r == c ? lambda : 0.0) : 0.0) - v;
            S[(size_t)(6 * i + r) * G.ld + 6 * j + c] = out;
            if (i != j) S[(size_t)(6 * j + c) * G.ld + 6 * i + r] = out;
        }
    }
    // bs = bp - W D^-1 b_l (block_solver.hpp:409-416, 435-438): the diagonal pair lists every Hpl block of pose i exactly once; a second,
    // short pass over it (its blocks are in L2 now) -- inside the loop above the extra branch and registers cost more than they saved
    if (i == j) {
        double wdb[6] = {0, 0, 0, 0, 0, 0};
        for (int e = ps[0] + sub; e < ps[1]; e += LPP) {
            const double *w = B.Wsp + (size_t)ent[e].x * 18, *db = B.db + (size_t)(G.point_off + entl[e]) * 3;
            const double d0 = db[0], d1 = db[1], d2 = db[2];
#pragma unroll
            for (int a = 0; a < 6; a++) wdb[a] += w[a] * d0 + w[6 + a] * d1 + w[12 + a] * d2;
        }
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const double v = LPP == 16 ? row16_allreduce_f64_dpp(wdb[a]) : wave_sum_f64_dpp(wdb[a]);
            if (sub == 0) { B.bacc[(size_t)(G.free_off + i) * 6 + a] = v; B.bs[(size_t)(G.free_off + i) * 6 + a] = B.bp[(size_t)(G.free_off + i) * 6 + a] - v; }
        }
    }
}

// The same sums, one 256-thread workgroup per ROW i of the block matrix (round 3).  k_ba_schur_big gathers, per list entry, BOTH Hpl blocks
// and the point's D^-1 from L2: 5-6 cache lines for 162 multiply-adds, 9 GB per launch at 256 windows, and the gather path (one line per
// lane per load instruction) is what the kernel waits for.  All pairs (i, j >= i) share pose i's side: the workgroup computes
// Y_il = W_il D_l^-1 for every block of pose i ONCE into LDS (the diagonal pair's list enumerates them in point order; an entry carries
// the rank of its i-side block in that list), then its sixteen 16-lane groups take the pairs of the row round-robin and gather only W_jl:
// 2 lines per entry, a third of the multiply-adds gone.  Expressions and summation order are k_ba_schur_big<16>'s, so the two kernels
// agree bit for bit on every off-diagonal block (the diagonal blocks and bs are summed in another association, see below).  Rows are dealt
// like the pairs were: a graph's rows on one XCD, longest rows first.
#define SROW_THREADS 384                              // 6 waves: 170 VGPRs each at 3 waves per SIMD, two workgroups per CU while a row is <= 500 blocks
#define SROW_GROUPS (SROW_THREADS / 16)
#define SROW_MAX_BLOCKS 1024                          // 147 KB of LDS; batches with a fuller row keep k_ba_schur_big
__global__ __launch_bounds__(SROW_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_ba_schur_rows(BaBatch B, int max_nf)
{
    extern __shared__ __attribute__((aligned(16))) double srow_y[];          // [blocks of pose i][18]: y0[6] y1[6] y2[6]
    __shared__ double s_part[SROW_GROUPS * 27];                              // the diagonal block's and W db's partial sums, one set per 16-lane group
    __shared__ int s_next;                                                   // next pair of the row nobody has taken
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = xcd + 8 * (slot / max_nf), i = slot - (slot / max_nf) * max_nf;
    if (g >= B.G) return;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const int nf = G.nf;
    if (i >= nf) return;
    const int tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
    const int *ps = B.big_pair_start + G.pair_off + (i * nf - i * (i - 1) / 2);      // pairs (i, i), (i, i + 1), ...
    const int2 *ent = B.big_pair_ent + G.pent_off;
    const int *entl = B.big_pair_pt + G.pent_off;
    const int2 *entjr = B.big_pair_jr + G.pent_off;
    if (tid == 0) s_next = i + 1 + SROW_GROUPS;
    // ---- pose i's own blocks, one per thread: Y into LDS, and the diagonal pair (i, i) on the way -- its list IS this enumeration, and as
    // one 16-lane group's chain (400 entries at 50 x 2000 x 10) it would outlast every other pair of the row by a factor of five while the
    // workgroup holds its LDS.  S_ii and bs = bp - W D^-1 b_l are summed per thread, per 16-lane group (DPP), then over the groups in order.
    {
        const int d0 = ps[0], nblk = ps[1] - d0;
        double dacc[21], wdb[6];                       // S_ii is symmetric: its lower triangle, row-major
#pragma unroll
        for (int k = 0; k < 21; k++) dacc[k] = 0.0;
#pragma unroll
        for (int a = 0; a < 6; a++) wdb[a] = 0.0;
        for (int p = tid; p < nblk; p += SROW_THREADS) {
            const int l = entl[d0 + p];
            const double *Di = B.Dinv + (size_t)(G.point_off + l) * 6, *db = B.db + (size_t)(G.point_off + l) * 3;
            const double i00 = Di[0], i01 = Di[1], i02 = Di[2], i11 = Di[3], i12 = Di[4], i22 = Di[5];
            const double d0b = db[0], d1b = db[1], d2b = db[2];
            const double2 *wa2 = reinterpret_cast<const double2 *>(B.Wsp + (size_t)ent[d0 + p].x * 18);       // 144-byte blocks: 16-byte aligned
            double wa[18], y0[6], y1[6], y2[6];
#pragma unroll
            for (int k = 0; k < 9; k++) { const double2 v = wa2[k]; wa[2 * k] = v.x; wa[2 * k + 1] = v.y; }
            double *y = srow_y + 18 * p;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const double a0 = wa[a], a1 = wa[6 + a], a2 = wa[12 + a];
                y0[a] = a0 * i00 + a1 * i01 + a2 * i02;
                y1[a] = a0 * i01 + a1 * i11 + a2 * i12;
                y2[a] = a0 * i02 + a1 * i12 + a2 * i22;
                y[a] = y0[a]; y[6 + a] = y1[a]; y[12 + a] = y2[a];
                wdb[a] += a0 * d0b + a1 * d1b + a2 * d2b;
            }
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = 0; c <= r; c++) dacc[r * (r + 1) / 2 + c] += y0[r] * wa[c] + y1[r] * wa[6 + c] + y2[r] * wa[12 + c];
        }
#pragma unroll
        for (int k = 0; k < 21; k++) { const double v = row16_allreduce_f64_dpp(dacc[k]); if (sub == 0) s_part[27 * grp + k] = v; }
#pragma unroll
        for (int a = 0; a < 6; a++) { const double v = row16_allreduce_f64_dpp(wdb[a]); if (sub == 0) s_part[27 * grp + 21 + a] = v; }
    }
    __syncthreads();
    double *S = B.S + G.s_off;
    if (tid < 27) {
        double v = 0.0;
        for (int q = 0; q < SROW_GROUPS; q++) v += s_part[27 * q + tid];
        if (tid < 21) {
            int r = 0, c = tid;
            while (c > r) { c -= r + 1; r++; }
            const double out = B.Hpp[(size_t)(G.free_off + i) * 36 + 6 * r + c] + (r == c ? st.lambda : 0.0) - v;
            S[(size_t)(6 * i + r) * G.ld + 6 * i + c] = out;
            S[(size_t)(6 * i + c) * G.ld + 6 * i + r] = out;
        } else {
            const size_t o = (size_t)(G.free_off + i) * 6 + tid - 21;
            B.bacc[o] = v; B.bs[o] = B.bp[o] - v;
        }
    }
    // ---- the pairs (i, j > i), handed out dynamically (list lengths differ by an order of magnitude): the first one per group by index, then
    // from the counter
    for (int j = i + 1 + grp; j < nf; j = row16_allreduce_add_dpp(sub == 0 ? atomicAdd(&s_next, 1) : 0)) {
        const int e0 = ps[j - i], e_end = ps[j - i + 1];
        double acc[36];
#pragma unroll
        for (int k = 0; k < 36; k++) acc[k] = 0.0;
        const int en = e0 + sub;                                       // the NEXT trip's list entry is fetched a trip ahead
        int2 tn = make_int2(0, 0);
        if (en < e_end) tn = entjr[en];
        for (int e = en; e < e_end; e += 16) {
            const int bj = tn.x, pos = tn.y;
            if (e + 16 < e_end) tn = entjr[e + 16];
            const double2 *wb2 = reinterpret_cast<const double2 *>(B.Wsp + (size_t)bj * 18);
            double wb[18];
#pragma unroll
            for (int k = 0; k < 9; k++) { const double2 v = wb2[k]; wb[2 * k] = v.x; wb[2 * k + 1] = v.y; }
            const double *y = srow_y + 18 * pos;
            double y0[6], y1[6], y2[6];
#pragma unroll
            for (int a = 0; a < 6; a++) { y0[a] = y[a]; y1[a] = y[6 + a]; y2[a] = y[12 + a]; }
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const double b0 = wb[c], b1 = wb[6 + c], b2 = wb[12 + c];
#pragma unroll
                for (int r = 0; r < 6; r++) acc[6 * r + c] += y0[r] * b0 + y1[r] * b1 + y2[r] * b2;
            }
        }
#pragma unroll
        for (int k = 0; k < 36; k++) {
            const double v = row16_allreduce_f64_dpp(acc[k]);
            if (sub == 0) {
                const int r = k / 6, c = k - 6 * r;
                S[(size_t)(6 * i + r) * G.ld + 6 * j + c] = -v;
                S[(size_t)(6 * j + c) * G.ld + 6 * i + r] = -v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_ba_big_init(BaBatch B)
{
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < G.n) B.big_y[(size_t)G.free_off * 6 + i] = B.bs[(size_t)G.free_off * 6 + i];
    if (i == 0) B.big_fail[g] = 0;
}

__global__ __launch_bounds__(64) void k_ba_big_diag(BaBatch B, int p0)
{
    __shared__ double P[LD_NB * LD_PP], U[LD_NB * LD_NB], dv[LD_NB], yv[LD_NB];
    __shared__ int s_ok;
    const int g = blockIdx.x, lane = threadIdx.x;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    if (p0 >= G.n || B.big_fail[g]) return;
    const int nb = min(LD_NB, G.n - p0), ld = G.ld;
    double *S = B.S + G.s_off, *y = B.big_y + (size_t)G.free_off * 6, *d = B.big_d + (size_t)G.free_off * 6;
    if (lane == 0) s_ok = 1;
    if (lane < nb) {
        for (int c = 0; c < nb; c++) P[lane * LD_PP + c] = S[(size_t)(p0 + lane) * ld + p0 + c];
        yv[lane] = y[p0 + lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < nb) ldlt_rows<true>((lds_f64 *)P, (lds_f64 *)U, (lds_f64 *)dv, (lds_f64 *)yv, lane, nb, &s_ok);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!s_ok) { if (lane == 0) B.big_fail[g] = 1; return; }
    if (lane < nb) {
        for (int c = 0; c <= lane; c++) S[(size_t)(p0 + lane) * ld + p0 + c] = (c == lane) ? dv[c] : P[lane * LD_PP + c];
        d[p0 + lane] = dv[lane]; y[p0 + lane] = yv[lane];
    }
    double *Ug = B.big_U + (size_t)g * LD_NB * LD_NB;
    for (int k = lane; k < LD_NB * LD_NB; k += 64) Ug[k] = U[k];
}

__global__ __launch_bounds__(256) void k_ba_big_rows(BaBatch B, int p0)
{
    extern __shared__ double brl[];                  // P [32 + 256][LD_PP] (rows 0..31 unused), U [32*32], dv [32], yv [32 + 256]
    __shared__ int s_dummy;
    const int g = blockIdx.y, tid = threadIdx.x;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    if (p0 >= G.n || B.big_fail[g]) return;
    const int nb = min(LD_NB, G.n - p0), m = G.n - p0, ld = G.ld;
    const int r = nb + blockIdx.x * 256 + tid;           // row of the panel (relative to p0)
    if (nb + blockIdx.x * 256 >= m) return;
    double *P = brl, *U = P + (size_t)(LD_NB + 256) * LD_PP, *dv = U + LD_NB * LD_NB, *yv = dv + LD_NB;
    double *S = B.S + G.s_off, *y = B.big_y + (size_t)G.free_off * 6;
    const double *Ug = B.big_U + (size_t)g * LD_NB * LD_NB, *d = B.big_d + (size_t)G.free_off * 6;
    for (int k = tid; k < LD_NB * LD_NB; k += 256) U[k] = Ug[k];
    if (tid < nb) { dv[tid] = d[p0 + tid]; yv[tid] = y[p0 + tid]; }
    const int rl = LD_NB + tid;                          // this thread's row inside the LDS panel
    if (r < m) {
        for (int c = 0; c < nb; c++) P[rl * LD_PP + c] = S[(size_t)(p0 + r) * ld + p0 + c];
        yv[rl] = y[p0 + r];
    }
    __syncthreads();
    if (r < m) {
        ldlt_rows<false>((lds_f64 *)P, (lds_f64 *)U, (lds_f64 *)dv, (lds_f64 *)yv, rl, nb, &s_dummy);
        for (int c = 0; c < nb; c++) S[(size_t)(p0 + r) * ld + p0 + c] = P[rl * LD_PP + c];
        y[p0 + r] = yv[rl];
    }
}

// trailing update S[i][k] -= sum_c L[i][c] d_c L[k][c] for i >= k >= p0 + nb: 64x64 tiles of the lower triangle, 4x4 per thread
__global__ __launch_bounds__(256) void k_ba_big_trail(BaBatch B, int p0)
{
    __shared__ double LI[64 * LD_PP], LK[64 * LD_PP], dv[LD_NB];
    const int g = blockIdx.z, tid = threadIdx.x;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    if (p0 >= G.n || B.big_fail[g]) return;
    const int nb = min(LD_NB, G.n - p0), ld = G.ld;
    const int base = p0 + nb, m2 = G.n - base;
    const int ti = blockIdx.y, tk = blockIdx.x;
    if (tk > ti || 64 * ti >= m2) return;
    double *S = B.S + G.s_off;
    const double *d = B.big_d + (size_t)G.free_off * 6;
    for (int idx = tid; idx < 64 * LD_NB; idx += 256) {
        const int r = idx >> 5, c = idx & 31;
        const int gi = base + 64 * ti + r, gk = base + 64 * tk + r;
        LI[r * LD_PP + c] = (c < nb && gi < G.n) ? S[(size_t)gi * ld + p0 + c] : 0.0;
        LK[r * LD_PP + c] = (c < nb && gk < G.n) ? S[(size_t)gk * ld + p0 + c] : 0.0;
    }
    if (tid < LD_NB) dv[tid] = tid < nb ? d[p0 + tid] : 0.0;
    __syncthreads();
    const int a0 = 4 * (tid >> 4), b0 = 4 * (tid & 15);
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b2 = 0; b2 < 4; b2++) acc[a][b2] = 0.0;
    for (int c = 0; c < nb; c++) {
        const double dc = dv[c];
        double av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; a++) { av[a] = LI[(a0 + a) * LD_PP + c] * dc; bv[a] = LK[(b0 + a) * LD_PP + c]; }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b2 = 0; b2 < 4; b2++) acc[a][b2] += av[a] * bv[b2];
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b2 = 0; b2 < 4; b2++) {
            const int gi = base + 64 * ti + a0 + a, gk = base + 64 * tk + b0 + b2;
            if (gi < G.n && gk <= gi) S[(size_t)gi * ld + gk] -= acc[a][b2];
        }
}

__global__ __launch_bounds__(1024) void k_ba_big_backsub(BaBatch B)
{
    extern __shared__ double bbl[];                  // y [max_ld], red [32*32], P [32][LD_PP]
    const int g = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const int n = G.n, ld = G.ld;
    if (B.big_fail[g]) { if (tid == 0) st.ok = 0; return; }      // x untouched (as the reference on failure)
    double *y = bbl, *red = y + B.max_ld, *P = red + 32 * 32;
    const double *S = B.S + G.s_off, *dval = B.big_d + (size_t)G.free_off * 6;
    // y <- D^-1 y, then L^T x = y panel by panel from the bottom
    for (int i = tid; i < n; i += nth) y[i] = B.big_y[(size_t)G.free_off * 6 + i] / dval[i];
    __syncthreads();
    const int last_p0 = ((n - 1) / LD_NB) * LD_NB;
    for (int p0 = last_p0; p0 >= 0; p0 -= LD_NB) {
        const int nb = min(LD_NB, n - p0), m = n - p0;
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
    for (int i = tid; i < n; i += nth) B.xp[(size_t)G.free_off * 6 + i] = y[i];
    if (tid == 0) st.ok = 1;
}

// landmark back-substitution (block_solver.hpp:461-481) + trial update of every vertex
// (sparse_optimizer.cpp:422-435) + computeScale partials (levenberg.cpp:187-194)
__global__ __launch_bounds__(256) void k_ba_backsub_points(BaBatch B)
{
    // 16 lanes per point: one observing pose per lane and trip
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const int sub = threadIdx.x & 15;
    const int l = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (l >= G.n_points) return;
    const int *ps = B.pt_start + G.ptstart_off;
    const size_t gl = (size_t)G.point_off + l;
    double *xl = B.xl + gl * 3;
    const double *Xc = B.points + ((size_t)st.cur * B.sumL + gl) * 3;
    double *Xn = B.points + ((size_t)(st.cur ^ 1) * B.sumL + gl) * 3;
    if (ps[l + 1] == ps[l]) { if (sub == 0) { Xn[0] = Xc[0]; Xn[1] = Xc[1]; Xn[2] = Xc[2]; B.scale_pt[gl] = 0; } return; }
    const double *bl = B.bl + gl * 3;
    double x0 = xl[0], x1 = xl[1], x2 = xl[2];                     // a failed solve keeps the previous increment
    if (st.ok) {
        double c0 = 0, c1 = 0, c2 = 0;
        for (int e = ps[l] + sub; e < ps[l + 1]; e += 16) {
            const int h = B.hidx[G.pose_off + B.edge_pose[G.edge_off + e]];
            if (h < 0 || B.edge_dup[G.edge_off + e]) continue;              // one Hpl block per (point, pose)
            const double *xp = B.xp + (size_t)(G.free_off + h) * 6;
            const double *w = B.Wsp + (size_t)B.edge_task[G.edge_off + e] * 18;
#pragma unroll
            for (int a = 0; a < 6; a++) { c0 -= w[a] * xp[a]; c1 -= w[6 + a] * xp[a]; c2 -= w[12 + a] * xp[a]; }
        }
        c0 = row16_allreduce_f64_dpp(c0); c1 = row16_allreduce_f64_dpp(c1); c2 = row16_allreduce_f64_dpp(c2);
        c0 += bl[0]; c1 += bl[1]; c2 += bl[2];
        const double *Di = B.Dinv + gl * 6;
        x0 = Di[0] * c0 + Di[1] * c1 + Di[2] * c2;
        x1 = Di[1] * c0 + Di[3] * c1 + Di[4] * c2;
        x2 = Di[2] * c0 + Di[4] * c1 + Di[5] * c2;
    }
    if (sub == 0) {
        xl[0] = x0; xl[1] = x1; xl[2] = x2;
        Xn[0] = Xc[0] + x0; Xn[1] = Xc[1] + x1; Xn[2] = Xc[2] + x2;
        B.scale_pt[gl] = x0 * (st.lambda * x0 + bl[0]) + x1 * (st.lambda * x1 + bl[1]) + x2 * (st.lambda * x2 + bl[2]);
    }
}

__global__ __launch_bounds__(64) void k_ba_update_poses(BaBatch B)
{
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= G.n_poses) return;
    const double *pc = B.poses + ((size_t)st.cur * B.sumP + G.pose_off + p) * 7;
    double *pn = B.poses + ((size_t)(st.cur ^ 1) * B.sumP + G.pose_off + p) * 7;
    const int h = B.hidx[G.pose_off + p];
    if (h < 0) { for (int i = 0; i < 7; i++) pn[i] = pc[i]; return; }
    const double *x = B.xp + (size_t)(G.free_off + h) * 6;
    se3_oplus(x, pc, pn);
    const double *bp = B.bp + (size_t)(G.free_off + h) * 6;
    double s = 0;
    for (int a = 0; a < 6; a++) s += x[a] * (st.lambda * x[a] + bp[a]);
    B.scale_pose[G.free_off + h] = s;
}

// LM accept/reject + iteration bookkeeping: levenberg.cpp:121-169, sparse_optimizer.cpp:372-418,
// Optimizer.cc:2048-2122 (two passes).  One thread per graph.
__global__ void k_ba_control(BaBatch B, int abort_arg)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B.G) return;
    const int abort_flag = B.world > 1 ? *B.x_abort : abort_arg;       // sharded: every rank must see the same flag
    BaState &st = B.st[g];
    if (!st.active) return;
    double temp_chi = B.chi[g];
    if (!st.ok) temp_chi = DBL_MAX;
    double rho = st.current_chi - temp_chi;
    const double scale = B.scale[g] + 1e-3;
    rho /= scale;
    if (rho > 0 && isfinite(temp_chi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        const double sf = fmax(1. / 3., alpha);
        st.lambda *= sf; st.ni = 2; st.current_chi = temp_chi;
        st.cur ^= 1;                                   // discardTop: the trial becomes the estimate
        st.errors_fresh = 1;
    } else {
        st.lambda *= st.ni; st.ni *= 2;                // pop: keep the old estimate
        st.errors_fresh = 0;                           // the stored errors belong to the rejected trial
    }
    st.qmax++; st.lm_trials++;
    st.rho_dbg = rho;
    if (rho < 0 && st.qmax < B.max_trials && !abort_flag) return;      // retry with the larger lambda
    // outer iteration finished
    st.iters_run[st.pass]++;
    st.chi_last = st.current_chi;
    int ok = 1;
    if (st.qmax == B.max_trials || rho == 0) ok = 0;
    else {
        if ((st.ini_chi - st.current_chi) * 1e3 < st.ini_chi) st.nbad++; else st.nbad = 0;
        if (st.nbad >= 3) ok = 0;
    }
    st.iter++;
    st.need_build = 1;
    if (!ok || st.iter >= B.iters[st.pass] || abort_flag) {
        if (st.pass == 0 && !abort_flag && B.iters[1] > 0) {
            st.pass = 1; st.iter = 0; st.need_lambda_init = 1; st.errors_fresh = 0;      // levels / robust kernel change: re-evaluate
            if (B.ex2) st.apply_levels = 1;                // Optimizer.cc:6546-6579 (merge variant)
            if (B.nr2) st.robust = 0;
        }
        else { st.active = 0; atomicSub(B.n_active, 1); }
    }
}

// Merge variant, between the passes (Optimizer.cc:6546-6579): chi2 of the last evaluation > gate or depth <= 0 at the
// current estimate => setLevel(1).  Runs at the start of the tick after the pass switch (apply_levels).
// ---- landmark-sharded solve: pack this rank's partial sums into its slot of xbuf / reduce all slots in rank order
__global__ __launch_bounds__(256) void k_ba_shard_pack1(BaBatch B)
{
    const int g = blockIdx.x, tid = threadIdx.x;
    const BaState &st = B.st[g];
    if (!st.active || !st.need_build) return;
    const BaGraphDev &G = B.gd[g];
    double *X = B.xbuf + (size_t)B.rank * B.xcount + B.x1_off[g];
    for (int i = tid; i < G.nf * 36; i += 256) X[i] = B.Hpp[(size_t)G.free_off * 36 + i];
    for (int i = tid; i < G.nf * 6; i += 256) X[G.nf * 36 + i] = B.bp[(size_t)G.free_off * 6 + i];
    if (tid == 0) { X[G.nf * 42] = B.chi[g]; X[G.nf * 42 + 1] = B.maxdiag[g]; }
}
__global__ __launch_bounds__(256) void k_ba_shard_sum1(BaBatch B)
{
    __shared__ double red[256];
    const int g = blockIdx.x, tid = threadIdx.x;
    const BaState &st = B.st[g];
    if (!st.active || !st.need_build) return;
    const BaGraphDev &G = B.gd[g];
    const double *X = B.xbuf + B.x1_off[g];
    double m = 0;
    for (int i = tid; i < G.nf * 36; i += 256) {
        double v = 0;
        for (int r = 0; r < B.world; r++) v += X[(size_t)r * B.xcount + i];
        B.Hpp[(size_t)G.free_off * 36 + i] = v;
        if (i % 36 % 7 == 0) m = fmax(m, fabs(v));
    }
    for (int i = tid; i < G.nf * 6; i += 256) {
        double v = 0;
        for (int r = 0; r < B.world; r++) v += X[(size_t)r * B.xcount + G.nf * 36 + i];
        B.bp[(size_t)G.free_off * 6 + i] = v;
    }
    red[tid] = m;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if (tid < d) red[tid] = fmax(red[tid], red[tid + d]); __syncthreads(); }
    if (tid == 0) {
        double c = 0, mh = red[0];
        for (int r = 0; r < B.world; r++) { c += X[(size_t)r * B.xcount + G.nf * 42]; mh = fmax(mh, X[(size_t)r * B.xcount + G.nf * 42 + 1]); }
        B.chi[g] = c;
        if (st.need_lambda_init) B.maxdiag[g] = mh;
    }
}
__global__ __launch_bounds__(256) void k_ba_shard_pack2(BaBatch B)
{
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const int idx = blockIdx.x * 256 + threadIdx.x, n2 = G.ld * G.ld;
    double *X = B.xbuf + (size_t)B.rank * B.xcount + B.x2_off[g];
    if (idx < n2) {
        const double *Sp = B.Spart + G.spart_off + idx;
        double s = 0;
        for (int k = 0; k < G.ks; k++) s += Sp[(size_t)k * n2];
        X[idx] = s;
    }
    if (idx < G.nf * 6) X[n2 + idx] = B.bacc[(size_t)G.free_off * 6 + idx];
}
__global__ __launch_bounds__(256) void k_ba_shard_sum2(BaBatch B)
{
    const int g = blockIdx.x;
    const BaState &st = B.st[g];
    if (!st.active) return;
    const BaGraphDev &G = B.gd[g];
    const double *X = B.xbuf + B.x2_off[g] + (size_t)G.ld * G.ld;
    for (int i = threadIdx.x; i < G.nf * 6; i += 256) {
        double a = 0;
        for (int r = 0; r < B.world; r++) a += X[(size_t)r * B.xcount + i];
        B.bs[(size_t)G.free_off * 6 + i] = B.bp[(size_t)G.free_off * 6 + i] - a;
    }
}
// stage 3: trial chi2, computeScale partial, abort flag; stage 4: outlier / edge counts.  payload 3 doubles per graph.
__global__ void k_ba_shard_pack34(BaBatch B, int stage, int abort_arg)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B.G) return;
    double *X = B.xbuf + (size_t)B.rank * B.xcount + 3 * (size_t)g;
    if (stage == 3) { X[0] = B.chi[g]; X[1] = B.scale[g]; X[2] = (double)abort_arg; }
    else { X[0] = (double)B.st[g].n_outliers; X[1] = (double)B.gd[g].n_edges; X[2] = 0; }
}
__global__ void k_ba_shard_sum34(BaBatch B, int stage)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B.G) return;
    double a = 0, b2 = 0, c = 0;
    for (int r = 0; r < B.world; r++) { const double *X = B.xbuf + (size_t)r * B.xcount + 3 * (size_t)g; a += X[0]; b2 += X[1]; c = fmax(c, X[2]); }
    if (stage == 3) { if (B.st[g].active) { B.chi[g] = a; B.scale[g] = b2; } if (g == 0) *B.x_abort = c > 0 ? 1 : 0; }
    else { B.st[g].n_outliers = (int)a; B.edges_total[g] = b2; }
}

__global__ __launch_bounds__(256) void k_ba_levels(BaBatch B)
{
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    if (!st.active || !st.apply_levels) return;
    const BaGraphDev &G = B.gd[g];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= G.n_edges) return;
    const int ge = G.edge_off + e;
    const double *pose = B.poses + ((size_t)st.cur * B.sumP + G.pose_off + B.edge_pose[ge]) * 7;
    const double *X = B.points + ((size_t)st.cur * B.sumL + G.point_off + B.edge_point[ge]) * 3;
    const BaCamDev &cam = ba_cam_ref<true>(B, G, B.edge_pose[ge]);      // (mTrl of the edge's own keyframe)
    const double z = edge_depth(cam, pose, X, B.edge_stereo[ge]);
    const double gate = B.edge_stereo[ge] == 1 ? B.gate_s : B.gate_m;
    if ((B.chi2[ge] > gate) || !(z > 0.0)) B.level[ge] = 1;
}

// outlier gates, Optimizer.cc:2126-2173: stored chi2 of the last evaluation; depth at the final estimate
__global__ __launch_bounds__(256) void k_ba_finalize(BaBatch B)
{
    const int g = blockIdx.y;
    const BaState &st = B.st[g];
    const BaGraphDev &G = B.gd[g];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= G.n_edges) return;
    const int ge = G.edge_off + e;
    const double *pose = B.poses + ((size_t)st.cur * B.sumP + G.pose_off + B.edge_pose[ge]) * 7;
    const double *X = B.points + ((size_t)st.cur * B.sumL + G.point_off + B.edge_point[ge]) * 3;
    const BaCamDev &cam = ba_cam_ref<true>(B, G, B.edge_pose[ge]);      // (mTrl of the edge's own keyframe)
    const double z = edge_depth(cam, pose, X, B.edge_stereo[ge]);
    const double gate = B.edge_stereo[ge] == 1 ? B.gate_s : B.gate_m;
    const int out = (B.chi2[ge] > gate) || !(z > 0.0);
    B.outlier[ge] = (uint8_t)out;
    if (out) atomicAdd(&B.st[g].n_outliers, 1);
}

// ------------------------------------------------------------------ host side
static thread_local std::string g_ba_error;

struct orbhip_ba_batch {
    orbhip_ctx *ctx;
    BaBatch B;
    std::vector<BaGraphDev> gd;
    std::vector<void *> allocs;              // ONE device arena per batch (round 4; ~60 separate hipMalloc before), absent when the
    bool ctx_arena, ctx_word;                // one-shot call borrowed the context's cached arena / pinned word instead
    std::vector<double> poses0, points0;     // normalised initial estimates (host copy)
    int *h_n_active;                         // pinned
    size_t wd_total, s_total, spart_total;
    int ticks_last;
    bool no_discard;                         // merge variant: no >= 50 % outlier bail-out
    bool profile;                            // time the Schur GEMM launches with hipEvents
    hipEvent_t ev0, ev1;
    float gemm_ms_total; int gemm_launches;
    double gemm_flops_per_launch;            // MFMA flops actually issued by one launch (all graphs)
    double gemm_flops_dense;                 // what the same upper tiles would cost without block-sparsity skipping
    double gemm_flops_issued;                // MFMA flops issued by one launch (every tile of a chunk whose row tile and one column tile the point touches)
    // landmark-sharded batches: the slice of every full graph this rank owns
    bool general;                            // some graph needs KannalaBrandt8, second-camera edges or twin-edge chains
    struct Slice { int pt0, npts, e0, ne; };
    std::vector<Slice> slices;
    size_t x_need;                           // doubles per rank slot of the exchange buffer
    // one LM tick (about 20 dependent launches) captured as a hipGraph: small batches are launch bound
    hipGraphExec_t tick_graph; bool tick_graph_valid; BaBatch tick_B;
    int max_row_blocks;                      // most Hpl blocks any free pose of the batch holds (LDS of k_ba_schur_rows)
};

// Device memory of a batch: every array is a slice of ONE arena.  ba_create_impl first declares all slices (uploads first: their
// offsets are contiguous), then takes the arena -- its own hipMalloc, or for a one-shot solve (Optimizer::LocalBundleAdjustment's
// shape: create, solve, download, destroy) the context's cached grow-only arena -- gathers the upload slices in the context's
// page-locked staging area and sends them in ONE host-to-device copy.  Round 3 issued ~60 hipMalloc, ~27 synchronous copies from
// pageable vectors and ~60 hipFree per window: 1.8 + 0.96 ms of a 7 ms call (host_smoke latency).
struct BaPlan {
    struct It { void **dst; const void *src; size_t bytes, off; };
    std::vector<It> items;
    size_t total = 0, up_end = 0;
    void add(void **dst, const void *src, size_t bytes)
    {
        items.push_back({dst, src, bytes, total});
        total += (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
        if (src) up_end = total;
    }
};
void *orbhip_ctx_pinned_internal(orbhip_ctx *c, size_t bytes);
void *orbhip_ctx_ba_arena_acquire_internal(orbhip_ctx *c, size_t bytes);
void orbhip_ctx_ba_arena_release_internal(orbhip_ctx *c);
int *orbhip_ctx_pinned_word_internal(orbhip_ctx *c);

extern "C" void orbhip_ba_default_params(orbhip_ba_params *p)
{
    p->iters1 = 5; p->iters2 = 10; p->huber_mono2 = 5.991; p->huber_stereo2 = 7.815;
    p->user_lambda_init = 0.0; p->tau = 1e-50; p->max_trials = 100;
    p->stage2_exclude_outliers = 0; p->stage2_drop_robust = 0; p->no_discard = 0; p->gate_mono2 = 0; p->gate_stereo2 = 0;
}

// Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, pbStopFlag), Optimizer.cc:6255-6800
extern "C" void orbhip_ba_merge_params(orbhip_ba_params *p)
{
    orbhip_ba_default_params(p);
    p->huber_mono2 = 5.99; p->gate_mono2 = 5.991; p->gate_stereo2 = 7.815;       // :6395, :6554, :6571
    p->stage2_exclude_outliers = 1; p->stage2_drop_robust = 1; p->no_discard = 1;
}

// Optimizer::BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust), Optimizer.cc:62-330: ONE optimize(nIterations),
// thHuber2D = sqrt(5.99) / thHuber3D = sqrt(7.815) (:133-134) when bRobust, no outlier stage, results always written back
extern "C" void orbhip_ba_global_params(orbhip_ba_params *p, int iterations, int robust)
{
    orbhip_ba_default_params(p);
    p->iters1 = iterations; p->iters2 = 0; p->no_discard = 1;
    p->huber_mono2 = robust ? 5.99 : 1e300; p->huber_stereo2 = robust ? 7.815 : 1e300;     // delta -> inf: rho(x) = x, no kernel
    p->gate_mono2 = 5.991; p->gate_stereo2 = 7.815;                                          // informational outlier flags only
}

extern "C" void orbhip_ba_batch_destroy(orbhip_ba_batch *b)
{
    if (!b) return;
    (void)hipStreamSynchronize(orbhip_ctx_stream_internal(b->ctx));
    for (void *p : b->allocs) (void)hipFree(p);
    if (b->ctx_arena) orbhip_ctx_ba_arena_release_internal(b->ctx);
    if (b->h_n_active && !b->ctx_word) (void)hipHostFree(b->h_n_active);
    if (b->ev0) { (void)hipEventDestroy(b->ev0); (void)hipEventDestroy(b->ev1); }
    if (b->tick_graph_valid) (void)hipGraphExecDestroy(b->tick_graph);
    delete b;
}

// pose_in_system[g] (optional): which poses take part in the reduced system even without an edge in THIS graph -- a sharded
// batch sees only a slice of the edges but must number the free poses like every other rank.
// ---------------------------------------------------------------------------------------------------------------- pair lists, built on the device
// The lists k_ba_schur_big / k_ba_schur_rows walk -- per graph and pair of free poses (i <= j, row-major over the upper triangle) the Hpl
// blocks of the points both see, in point order -- come from the pose-major edge lists that are uploaded anyway: a wave owns a pair, runs
// over pose i's edges 64 at a time (they are in point order) and finds each point among pose j's by a lower bound (the first edge of a
// (point, pose) chain is the one that carries the block).  Same entries in the same order as the host enumeration of round 3/4 -- every
// summation order downstream is unchanged -- without the 0.5 ms the host spent per 50 x 2000 x 10 window and the 2 MB it uploaded.
// FILL = 0: entries per pair into big_pair_start; k_ba_pair_scan turns them into starts; FILL = 1: the entries.
template <int FILL>
__global__ __launch_bounds__(256) void k_ba_pair_lists(BaBatch B)
{
    const BaGraphDev &G = B.gd[blockIdx.y];
    const int nf = G.nf, npair = nf * (nf + 1) / 2;
    const int lane = threadIdx.x & 63, pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pi >= npair) return;
    int i = 0, j = pi;
    while (j >= nf - i) { j -= nf - i; i++; }
    j += i;
    const int *qs = B.pose_start + G.posestart_off, *pe = B.pose_edges + G.edge_off, *et = B.edge_task + G.edge_off, *ep = B.edge_point + G.edge_off;
    const int a0 = qs[i], a1 = qs[i + 1], c0 = qs[j], c1 = qs[j + 1];
    int *pstart = const_cast<int *>(B.big_pair_start) + G.pair_off;
    int2 *ent = const_cast<int2 *>(B.big_pair_ent) + G.pent_off, *jr = const_cast<int2 *>(B.big_pair_jr) + G.pent_off;
    int *ptl = const_cast<int *>(B.big_pair_pt) + G.pent_off;
    int out = FILL ? pstart[pi] : 0, rank = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int p0 = a0; p0 < a1; p0 += 64) {
        const int p = p0 + lane;
        int ta = -1, pt = -1;
        if (p < a1) { const int ea = pe[p]; ta = et[ea]; pt = ep[ea]; }
        const unsigned long long bb = __ballot(ta >= 0);
        const int myrank = rank + __popcll(bb & lt);             // place of the block in pose i's own (diagonal) list
        rank += __popcll(bb);
        int tc = -1;
        if (ta >= 0) {
            if (i == j) tc = ta;
            else {
                int lo = c0, hi = c1;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (ep[pe[mid]] < pt) lo = mid + 1; else hi = mid; }
                if (lo < c1) { const int ec = pe[lo]; if (ep[ec] == pt) tc = et[ec]; }
            }
        }
        const unsigned long long hb = __ballot(tc >= 0);
        if (FILL && tc >= 0) {
            const int q = out + __popcll(hb & lt);
            ent[q] = make_int2(ta, tc); ptl[q] = pt; jr[q] = make_int2(tc, myrank);
        }
        out += __popcll(hb);
    }
    if (!FILL && lane == 0) pstart[pi] = out;
}
// counts -> exclusive starts, one workgroup per graph; entry npair = the graph's total
__global__ __launch_bounds__(256) void k_ba_pair_scan(BaBatch B)
{
    const BaGraphDev &G = B.gd[blockIdx.x];
    const int npair = G.nf * (G.nf + 1) / 2;
    int *ps = const_cast<int *>(B.big_pair_start) + G.pair_off;
    __shared__ int part[256];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int k0 = 0; k0 <= npair; k0 += 256) {
        const int k = k0 + threadIdx.x;
        const int v = k < npair ? ps[k] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int t = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        const int base = carry;
        if (k <= npair) ps[k] = base + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = base + part[255];
        __syncthreads();
    }
}

static int ba_create_impl(orbhip_ctx *ctx, const orbhip_ba_graph *graphs, int n_graphs, const double *const *poses, const double *const *points,
                          const std::vector<std::vector<uint8_t>> *pose_in_system, int rank, int world, orbhip_ba_batch **out, bool oneshot = false)
{
    static const bool dry = getenv("ORBHIP_BA_CREATE_DRYRUN") != nullptr;       // development: time the host list building without a device
    const auto t_dry0 = std::chrono::steady_clock::now();
    if (!dry && (!ctx || !graphs || n_graphs <= 0 || !poses || !points || !out)) return ORBHIP_E_BADARG;
    if (!dry && hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    orbhip_ba_batch *b = new orbhip_ba_batch();
    b->ctx = ctx; b->h_n_active = nullptr; b->ctx_arena = false; b->ctx_word = false; b->ticks_last = 0; b->tick_graph = nullptr; b->tick_graph_valid = false;
    b->profile = false; b->ev0 = b->ev1 = nullptr; b->gemm_ms_total = 0; b->gemm_launches = 0; b->gemm_flops_per_launch = 0; b->gemm_flops_dense = 0; b->gemm_flops_issued = 0;
    BaBatch &B = b->B;
    memset(&B, 0, sizeof(B));
    B.G = n_graphs; B.rank = rank; B.world = world;
    b->general = false;
    // the host-side lists live in per-thread scratch vectors that keep their capacity from call to call: LocalMapping solves one window
    // per keyframe from the same thread, and fresh multi-megabyte vectors cost more in page faults than in arithmetic
    struct Scratch {
        std::vector<int> x1off, x2off, hidx, epose, epoint, ptstart, posestart, poseedges, etask, pmpoint, pmtask, enext;
        std::vector<uint32_t> ptmask;
        std::vector<int4> gtask, gstage;
        std::vector<uint8_t> pmtype, est, edup;
        std::vector<double> pmis2, pmobs, eobs, eis2;
        void clear()
        {
            x1off.clear(); x2off.clear(); hidx.clear(); epose.clear(); epoint.clear(); ptstart.clear(); posestart.clear(); poseedges.clear(); etask.clear();
            pmpoint.clear(); pmtask.clear(); enext.clear(); ptmask.clear(); gtask.clear(); gstage.clear(); pmtype.clear();
            est.clear(); edup.clear(); pmis2.clear(); pmobs.clear(); eobs.clear(); eis2.clear();
        }
    };
    static thread_local Scratch SC;
    SC.clear();
    std::vector<BaCamDev> cams;
    std::vector<int> posecam;
    std::vector<int> &x1off = SC.x1off, &x2off = SC.x2off;
    size_t x1 = 0, x2 = 0;
    std::vector<int> &hidx = SC.hidx, &epose = SC.epose, &epoint = SC.epoint, &ptstart = SC.ptstart, &posestart = SC.posestart, &poseedges = SC.poseedges;
    std::vector<uint32_t> &ptmask = SC.ptmask;
    std::vector<int4> &gtask = SC.gtask, &gstage = SC.gstage;
    std::vector<int> &etask = SC.etask, &pmpoint = SC.pmpoint, &pmtask = SC.pmtask;
    std::vector<uint8_t> &pmtype = SC.pmtype;
    std::vector<double> &pmis2 = SC.pmis2, &pmobs = SC.pmobs;
    std::vector<double> &eobs = SC.eobs, &eis2 = SC.eis2;
    std::vector<uint8_t> &est = SC.est, &edup = SC.edup;
    std::vector<int> &enext = SC.enext;
    int sumP = 0, sumL = 0, sumE = 0, sumF = 0;
    size_t s = 0, sp = 0;
    bool any_big = false;
    std::vector<std::vector<int>> g_local_h;                 // hessian index of every pose, per graph (big windows: pair lists)
    int mode = dry ? 0 : orbhip_ctx_ba_schur_mode_internal(ctx);
    if (const char *ev = getenv("ORBHIP_BA_PAIRS")) mode = atoi(ev) ? 1 : 2;                                                    // development override
    const bool want_gemm_stats = !(world == 1 && mode != 2);    // the MFMA flop counts of the GEMM form (bench / profiling): not needed by the pair-list form
    {   // every per-edge / per-point / per-pose array is sized once (round 4: ~25 push_back per edge were 1 ms of a one-shot solve)
        size_t te = 0, tl = 0, tp = 0;
        for (int g = 0; g < n_graphs; g++) { te += (size_t)std::max(graphs[g].n_edges, 0); tl += (size_t)std::max(graphs[g].n_points, 0); tp += (size_t)std::max(graphs[g].n_poses, 0); }
        hidx.reserve(tp); epose.reserve(te); epoint.reserve(te); eobs.reserve(3 * te); eis2.reserve(te); est.reserve(te); edup.reserve(te); enext.reserve(te);
        ptstart.reserve(tl + n_graphs); posestart.reserve(tp + n_graphs); poseedges.reserve(te); ptmask.reserve(tl); gtask.reserve(te); gstage.reserve(tl / 4 + 16);
        etask.reserve(te); pmpoint.reserve(te); pmtask.reserve(te); pmtype.reserve(te); pmis2.reserve(te); pmobs.reserve(3 * te);
        b->poses0.reserve(7 * tp); b->points0.reserve(3 * tl);
    }
    // split-K so that the Schur GEMM launches >= ~4096 waves
    for (int g = 0; g < n_graphs; g++) {
        const orbhip_ba_graph &H = graphs[g];
        if (H.n_poses <= 0 || H.n_points <= 0 || H.n_edges < 0) { delete b; return ORBHIP_E_BADARG; }
        BaGraphDev D;
        memset(&D, 0, sizeof(D));
        D.n_poses = H.n_poses; D.n_points = H.n_points; D.n_edges = H.n_edges;
        D.pose_off = sumP; D.point_off = sumL; D.edge_off = sumE; D.free_off = sumF;
        D.fx = H.fx; D.fy = H.fy; D.cx = H.cx; D.cy = H.cy; D.bf = H.bf;
        D.cam_model = H.camera_model; for (int k = 0; k < 4; k++) D.kb[k] = H.kb[k];
        for (int k = 0; k < 7; k++) D.Trl[k] = H.Trl[k];
        D.fx2 = H.fx2; D.fy2 = H.fy2; D.cx2 = H.cx2; D.cy2 = H.cy2; D.cam2_model = H.camera2_model; for (int k = 0; k < 4; k++) D.kb2[k] = H.kb2[k];
        D.cam_off = -1;
        if (H.n_cameras > 0) {                                   // per-keyframe calibration (Optimizer.cc:1961, :1990-1994, :2021-2023)
            if (!H.cameras || !H.pose_camera) { delete b; g_ba_error = "n_cameras > 0 needs cameras and pose_camera"; return ORBHIP_E_BADARG; }
            D.cam_off = (int)cams.size();
            for (int c = 0; c < H.n_cameras; c++) {
                const orbhip_ba_camera &Cc = H.cameras[c];
                BaCamDev K;
                K.fx = Cc.fx; K.fy = Cc.fy; K.cx = Cc.cx; K.cy = Cc.cy; K.bf = Cc.bf; K.cam_model = Cc.camera_model;
                for (int k = 0; k < 4; k++) { K.kb[k] = Cc.kb[k]; K.kb2[k] = Cc.kb2[k]; }
                for (int k = 0; k < 7; k++) K.Trl[k] = Cc.Trl[k];
                K.fx2 = Cc.fx2; K.fy2 = Cc.fy2; K.cx2 = Cc.cx2; K.cy2 = Cc.cy2; K.cam2_model = Cc.camera2_model;
                cams.push_back(K);
            }
            for (int i = 0; i < H.n_poses; i++) {
                if (H.pose_camera[i] < 0 || H.pose_camera[i] >= H.n_cameras) { delete b; g_ba_error = "pose_camera out of range"; return ORBHIP_E_BADARG; }
                posecam.push_back(H.pose_camera[i]);
            }
            b->general = true;                                   // the camera table is read by the general instantiations only
        } else posecam.insert(posecam.end(), (size_t)H.n_poses, 0);
        std::vector<int> has(H.n_poses, 0), local_h(H.n_poses, -1);
        for (int e = 0; e < H.n_edges; e++) {
            if (H.edge_pose[e] < 0 || H.edge_pose[e] >= H.n_poses || H.edge_point[e] < 0 || H.edge_point[e] >= H.n_points ||
                (e > 0 && H.edge_point[e] < H.edge_point[e - 1])) { delete b; g_ba_error = "edges must be point-major with valid ids"; return ORBHIP_E_BADARG; }
            has[H.edge_pose[e]] = 1;
        }
        if (pose_in_system) for (int i = 0; i < H.n_poses; i++) has[i] = (*pose_in_system)[g][i];
        int nf = 0;
        for (int i = 0; i < H.n_poses; i++) { local_h[i] = (!H.pose_fixed[i] && has[i]) ? nf++ : -1; hidx.push_back(local_h[i]); }
        D.nf = nf; D.n = 6 * nf; D.ld = std::max(96, (D.n + 95) / 96 * 96);   // multiple of 16 (MFMA tiles) and of 32 (1-KiB LDS-DMA pieces)
        if (D.n > BA_BIG_MAXN) { delete b; g_ba_error = "more than 682 free keyframes in one window"; return ORBHIP_E_BADARG; }
        const bool big_graph = D.n > BA_LDLT_MAXN;               // > 80 free keyframes: global-memory Schur complement + blocked LDL^T
        if (big_graph && world > 1) { delete b; g_ba_error = "landmark-sharded solve: at most 80 free keyframes per window"; return ORBHIP_E_BADARG; }
        any_big = any_big || big_graph;
        D.ptstart_off = (int)ptstart.size();
        std::vector<int> cnt(H.n_points + 1, 0);
        for (int e = 0; e < H.n_edges; e++) cnt[H.edge_point[e] + 1]++;
        for (int l = 0; l < H.n_points; l++) cnt[l + 1] += cnt[l];
        ptstart.insert(ptstart.end(), cnt.begin(), cnt.end());
        D.posestart_off = (int)posestart.size();
        std::vector<int> pc(nf + 1, 0);
        for (int e = 0; e < H.n_edges; e++) if (local_h[H.edge_pose[e]] >= 0) pc[local_h[H.edge_pose[e]] + 1]++;
        for (int h = 0; h < nf; h++) pc[h + 1] += pc[h];
        std::vector<int> fill(pc.begin(), pc.end() - 1), pel(H.n_edges, 0);
        for (int e = 0; e < H.n_edges; e++) { const int h = local_h[H.edge_pose[e]]; if (h >= 0) pel[fill[h]++] = e; }
        posestart.insert(posestart.end(), pc.begin(), pc.end());
        poseedges.insert(poseedges.end(), pel.begin(), pel.end());
        {                                                         // pose-major copy of the static edge data (entries past pc[nf] unused)
            const size_t o = pmpoint.size(), ne = (size_t)H.n_edges;
            pmpoint.resize(o + ne); pmtype.resize(o + ne); pmis2.resize(o + ne); pmobs.resize(3 * (o + ne));
            for (size_t k = 0; k < ne; k++) {
                const int e = (int)k < pc[nf] ? pel[k] : 0;
                pmpoint[o + k] = H.edge_point[e]; pmtype[o + k] = H.edge_stereo ? H.edge_stereo[e] : 0; pmis2[o + k] = H.edge_inv_sigma2[e];
                pmobs[3 * (o + k)] = H.edge_obs[3 * e]; pmobs[3 * (o + k) + 1] = H.edge_obs[3 * e + 1]; pmobs[3 * (o + k) + 2] = H.edge_obs[3 * e + 2];
            }
        }
        {   // edge types: 0 mono, 1 stereo, 2 second camera (needs a rigid transform mTrl with a non-zero quaternion)
            bool any2 = false;
            for (int e = 0; e < H.n_edges && H.edge_stereo; e++) {
                if (H.edge_stereo[e] > 2) { delete b; g_ba_error = "edge_stereo must be 0, 1 or 2"; return ORBHIP_E_BADARG; }
                any2 = any2 || H.edge_stereo[e] == 2;
            }
            const double qn = H.Trl[0] * H.Trl[0] + H.Trl[1] * H.Trl[1] + H.Trl[2] * H.Trl[2] + H.Trl[3] * H.Trl[3];
            if (any2 && H.n_cameras <= 0 && !(qn > 0.0)) { delete b; g_ba_error = "edges of type 2 need Trl (mTrl) and the second camera"; return ORBHIP_E_BADARG; }
            for (int e = 0; e < H.n_edges && any2 && H.n_cameras > 0 && H.cameras && H.pose_camera; e++) {
                if (H.edge_stereo[e] != 2) continue;
                const int ci = H.pose_camera[H.edge_pose[e]];
                if (ci < 0 || ci >= H.n_cameras) continue;       // (reported below)
                const double *t = H.cameras[ci].Trl;
                if (!(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3] > 0.0)) { delete b; g_ba_error = "edges of type 2 need their keyframe's Trl (mTrl) and second camera"; return ORBHIP_E_BADARG; }
            }
        }
        {   // chains of edges that share (point, pose): edges are point-major, so only a point's own edges are compared
            std::vector<int> nxt(H.n_edges, -1); std::vector<uint8_t> dup(H.n_edges, 0);
            for (int e0 = 0; e0 < H.n_edges;) {
                int e1 = e0;
                while (e1 < H.n_edges && H.edge_point[e1] == H.edge_point[e0]) e1++;
                for (int a = e0; a < e1; a++)
                    for (int c = a + 1; c < e1; c++)
                        if (H.edge_pose[c] == H.edge_pose[a]) { if (nxt[a] < 0) nxt[a] = c; dup[c] = 1; break; }
                e0 = e1;
            }
            edup.insert(edup.end(), dup.begin(), dup.end()); enext.insert(enext.end(), nxt.begin(), nxt.end());
            for (int e = 0; e < H.n_edges; e++) if (dup[e] || (H.edge_stereo && H.edge_stereo[e] == 2)) b->general = true;
            if (H.camera_model != 0) b->general = true;
        }
        epose.insert(epose.end(), H.edge_pose, H.edge_pose + H.n_edges); epoint.insert(epoint.end(), H.edge_point, H.edge_point + H.n_edges);
        eobs.insert(eobs.end(), H.edge_obs, H.edge_obs + 3 * (size_t)H.n_edges); eis2.insert(eis2.end(), H.edge_inv_sigma2, H.edge_inv_sigma2 + H.n_edges);
        if (H.edge_stereo) est.insert(est.end(), H.edge_stereo, H.edge_stereo + H.n_edges); else est.insert(est.end(), (size_t)H.n_edges, (uint8_t)0);
        const int nt = D.ld / 16, ntiles = nt * (nt + 1) / 2;
        int nchunks = 0;
        for (int tr = 0; tr < nt; tr++) nchunks += gemm_strip_chunks(nt, tr);
        D.nt16 = nt; D.ngrp = (nchunks + GEMM_WAVES * GEMM_NCH - 1) / (GEMM_WAVES * GEMM_NCH);
        {   // Schur GEMM work lists: Hpl blocks (free pose, first edge of a twin chain) point-major, cut into stages
            D.gemm_off = (int)gtask.size(); D.stage_off = (int)gstage.size();
            D.gemm_ps = (int)std::min<size_t>(GEMM_PS, GEMM_PANEL_LDS_BYTES / (sizeof(double) * 2 * 3 * (size_t)(D.ld + GEMM_LDS_PAD)));
            if (D.gemm_ps < 1) { delete b; g_ba_error = "too many free keyframes for the LDS panel of the Schur GEMM"; return ORBHIP_E_BADARG; }
            const uint8_t *dupv = edup.data() + (edup.size() - H.n_edges);
            int e = 0, pt0 = 0, npts = 0, t0 = (int)gtask.size() - D.gemm_off, ntask = 0;
            for (int l = 0; l < H.n_points; l++) {
                int e1 = e, cnt = 0;
                while (e1 < H.n_edges && H.edge_point[e1] == l) { if (local_h[H.edge_pose[e1]] >= 0 && !dupv[e1]) cnt++; e1++; }
                if (cnt > GEMM_STAGE_EDGES && !big_graph) { delete b; g_ba_error = "a point has more Hpl blocks than free keyframes fit (ld <= 512)"; return ORBHIP_E_BADARG; }
                if (npts == D.gemm_ps || ntask + cnt > GEMM_STAGE_EDGES) {       // (big windows: the stage lists are built but not used)
                    gstage.push_back(make_int4(pt0, npts, t0, ntask));
                    pt0 = l; npts = 0; t0 += ntask; ntask = 0;
                }
                for (int k = e; k < e1; k++) {
                    const bool blk = local_h[H.edge_pose[k]] >= 0 && !dupv[k];
                    etask.push_back(blk ? (int)gtask.size() : -1);
                    if (blk) gtask.push_back(make_int4(k, 6 * local_h[H.edge_pose[k]], l, g));
                }
                npts++; ntask += cnt; e = e1;
            }
            if (npts) gstage.push_back(make_int4(pt0, npts, t0, ntask));
            D.n_stages = (int)gstage.size() - D.stage_off;
        }
        {   // pose-major copy of edge_task (k_ba_bschur)
            const int *et = etask.data() + (etask.size() - H.n_edges);
            const int *pl = poseedges.data() + (poseedges.size() - H.n_edges);
            const int npm = posestart[posestart.size() - 1];
            for (int k = 0; k < H.n_edges; k++) pmtask.push_back(k < npm ? et[pl[k]] : -1);
        }
        // split the stages so that every CU has a workgroup
        int ks = (GEMM_TARGET_WGS + n_graphs * D.ngrp - 1) / (n_graphs * D.ngrp);
        ks = std::max(1, std::min(ks, std::max(1, D.n_stages / 4)));
        if (big_graph) ks = 1;                                   // k_ba_schur_big writes the whole block matrix once
        D.ks = ks;
        g_local_h.push_back(local_h);
        {   // static block-sparsity masks: 16-column tiles of the point's Hpl column that hold a non-zero block
            double issued = 0;
            ptmask.insert(ptmask.end(), (size_t)H.n_points, 0u);
            uint32_t *pmv = ptmask.data() + (ptmask.size() - H.n_points);
            for (int e = 0; e < H.n_edges && want_gemm_stats; e++) {
                const int h = local_h[H.edge_pose[e]];
                if (h >= 0 && !big_graph) pmv[H.edge_point[e]] |= (1u << ((6 * h) >> 4)) | (1u << ((6 * h + 5) >> 4));
            }
            double chunks = 0;                                                         // the kernel's issue rule: row tile hit AND a column tile of the chunk hit
            for (int l = 0; l < H.n_points && want_gemm_stats; l++) {
                const double k = __builtin_popcount(pmv[l]); issued += k * (k + 1) / 2;
                for (int tr = 0; tr < nt; tr++) {
                    if (!((pmv[l] >> tr) & 1u)) continue;
                    for (int tc = tr; tc < nt; tc += GEMM_C) {
                        const int len = std::min(GEMM_C, nt - tc);
                        const uint32_t hit = pmv[l] & (((1u << len) - 1u) << tc);
                        if (hit) chunks += __builtin_popcount(hit);
                    }
                }
            }
            b->gemm_flops_per_launch += issued * 2048.0;                               // one 16x16x4 f64 MFMA (2048 flop) per (point, upper tile) that holds data
            b->gemm_flops_issued += chunks * 2048.0;
            b->gemm_flops_dense += (double)ntiles * 2048.0 * (double)H.n_points;        // same tiles without the masks
        }
        x1off.push_back((int)x1); x1 += (size_t)nf * 42 + 2;
        x2off.push_back((int)x2); x2 += (size_t)D.ld * D.ld + (size_t)nf * 6;
        D.s_off = s; s += (size_t)D.ld * D.ld;
        D.spart_off = sp; sp += (size_t)ks * D.ld * D.ld;
        B.max_edges = std::max(B.max_edges, H.n_edges); B.max_points = std::max(B.max_points, H.n_points);
        B.max_nf = std::max(B.max_nf, nf); B.max_ld = std::max(B.max_ld, D.ld);
        // initial estimates: SE3Quat ctor normalises the rotation (se3quat.h:58-64)
        for (int i = 0; i < H.n_poses; i++) {
            double q[7];
            memcpy(q, poses[g] + 7 * i, sizeof(q));
            if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
            const double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            for (int k = 0; k < 4; k++) q[k] /= nn;
            b->poses0.insert(b->poses0.end(), q, q + 7);
        }
        b->points0.insert(b->points0.end(), points[g], points[g] + 3 * (size_t)H.n_points);
        sumP += H.n_poses; sumL += H.n_points; sumE += H.n_edges; sumF += nf;
        b->gd.push_back(D);
    }
    B.sumP = sumP; B.sumL = sumL; B.sumE = sumE; B.sumF = sumF;
    b->s_total = s; b->spart_total = sp;
    // big windows: every graph of the batch takes the global-memory path; per graph and pair of free poses (i <= j) the Hpl blocks
    // of the points both see, in point order (the summation order of k_ba_schur_big)
    b->max_row_blocks = 0;
    B.big = any_big ? 1 : 0;
    // Schur complement: per-block-pair lists (k_ba_schur_big) by default -- measured faster than the MFMA panel GEMM at every batch
    // size -- the GEMM on request (orbhip_ctx_set_ba_schur_mode(ctx, 2), a property of the context the batch is created on), in the landmark-sharded mode and never for big windows
    const bool pair_lists = any_big || (world == 1 && mode != 2);
    size_t pair_total_start = 0, pair_total_ent = 0;
    int max_npair = 0;
    if (dry) fprintf(stderr, "[orbhip ba] create (dry run): per-graph lists %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dry0).count());
    B.pair_schur = pair_lists ? 1 : 0;
    if (pair_lists) {
        for (int g = 0; g < n_graphs; g++) {
            BaGraphDev &D = b->gd[g];
            D.ks = 1;                                            // k_ba_schur_big writes slice 0 only
            const orbhip_ba_graph &H = graphs[g];
            const std::vector<int> &lh = g_local_h[g];
            const int nf = D.nf, npair = nf * (nf + 1) / 2;
            D.pair_off = pair_total_start; D.pent_off = pair_total_ent;
            const int *et = etask.data() + D.edge_off;
            // the lists themselves are built on the device (k_ba_pair_lists); the host only needs their sizes -- a point with nb blocks
            // puts one entry into nb (nb + 1) / 2 lists -- and the longest row of blocks (the LDS the row-owner Schur kernel takes)
            std::vector<int> seen(nf, 0);
            size_t tot = 0;
            for (int e0 = 0; e0 < H.n_edges;) {
                int e1 = e0, nb = 0;
                while (e1 < H.n_edges && H.edge_point[e1] == H.edge_point[e0]) { if (et[e1] >= 0) { nb++; seen[lh[H.edge_pose[e1]]]++; } e1++; }
                tot += (size_t)nb * (nb + 1) / 2;
                e0 = e1;
            }
            for (int k = 0; k < nf; k++) b->max_row_blocks = std::max(b->max_row_blocks, seen[k]);
            pair_total_start += (size_t)npair + 1; pair_total_ent += tot;
            max_npair = std::max(max_npair, npair);
        }
    }
    b->x_need = std::max(std::max(x1, x2), (size_t)3 * n_graphs);
    if (dry) {
        fprintf(stderr, "[orbhip ba] create (dry run): host lists %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dry0).count());
        delete b; return ORBHIP_E_NODEVICE;
    }
    BaPlan plan;
#define UP(dst, vec) plan.add((void **)&(dst), (vec).empty() ? (const void *)&plan : (const void *)(vec).data(), (vec).size() * sizeof((vec)[0]))
#define AL(dst, T, n) plan.add((void **)&(dst), nullptr, (size_t)(n) * sizeof(T))
    UP(B.gd, b->gd); UP(B.cams, cams); UP(B.pose_cam, posecam); UP(B.hidx, hidx); UP(B.edge_pose, epose); UP(B.edge_point, epoint); UP(B.edge_obs, eobs);
    UP(B.edge_is2, eis2); UP(B.edge_stereo, est); UP(B.edge_dup, edup); UP(B.edge_next, enext); UP(B.pt_start, ptstart); UP(B.pose_start, posestart); UP(B.pose_edges, poseedges);
    UP(B.ptmask, ptmask); UP(B.gemm_task, gtask); UP(B.gemm_stage, gstage); UP(B.edge_task, etask); UP(B.x1_off, x1off); UP(B.x2_off, x2off);
    UP(B.pm_point, pmpoint); UP(B.pm_task, pmtask); UP(B.pm_type, pmtype); UP(B.pm_is2, pmis2); UP(B.pm_obs, pmobs);
    if (any_big) {
        AL(B.big_y, double, (size_t)sumF * 6); AL(B.big_d, double, (size_t)sumF * 6); AL(B.big_U, double, (size_t)n_graphs * LD_NB * LD_NB);
        AL(B.big_fail, int, n_graphs);
    }
    if (pair_lists) {
        AL(B.big_pair_start, int, pair_total_start); AL(B.big_pair_ent, int2, std::max<size_t>(pair_total_ent, 1)); AL(B.big_pair_pt, int, std::max<size_t>(pair_total_ent, 1));
        AL(B.big_pair_jr, int2, std::max<size_t>(pair_total_ent, 1));
    }
    AL(B.st, BaState, n_graphs);
    AL(B.poses, double, (size_t)2 * sumP * 7); AL(B.points, double, (size_t)2 * sumL * 3);
    AL(B.err, double, (size_t)sumE * 3); AL(B.chi2, double, sumE); AL(B.rho0, double, sumE);
    AL(B.Hll, double, (size_t)sumL * 6); AL(B.bl, double, (size_t)sumL * 3); AL(B.Dinv, double, (size_t)sumL * 6); AL(B.db, double, (size_t)sumL * 3);
    AL(B.Hpp, double, (size_t)sumF * 36); AL(B.bp, double, (size_t)sumF * 6); AL(B.bs, double, (size_t)sumF * 6);
    AL(B.Wsp, double, gtask.size() * 18); AL(B.Linv, double, (size_t)sumL * 6); AL(B.S, double, s); AL(B.Spart, double, sp);
    AL(B.xp, double, (size_t)sumF * 6); AL(B.xl, double, (size_t)sumL * 3);
    AL(B.scale_pt, double, sumL); AL(B.scale_pose, double, sumF);
    AL(B.chi, double, n_graphs); AL(B.scale, double, n_graphs); AL(B.maxdiag, double, n_graphs);
    AL(B.n_active, int, 1); AL(B.outlier, uint8_t, sumE); AL(B.level, uint8_t, sumE);
    AL(B.bacc, double, (size_t)sumF * 6); AL(B.x_abort, int, 1); AL(B.edges_total, double, n_graphs);
#undef UP
#undef AL
    {
        hipStream_t st = orbhip_ctx_stream_internal(ctx);
        uint8_t *arena = nullptr;
        if (oneshot && (arena = (uint8_t *)orbhip_ctx_ba_arena_acquire_internal(ctx, plan.total))) b->ctx_arena = true;
        else {
            void *p = nullptr;
            if (hipMalloc(&p, plan.total) != hipSuccess) { orbhip_ba_batch_destroy(b); g_ba_error = "device allocation failed"; return ORBHIP_E_HIP; }
            b->allocs.push_back(p); arena = (uint8_t *)p;
        }
        uint8_t *stage = (uint8_t *)orbhip_ctx_pinned_internal(ctx, plan.up_end);
        if (!stage) { orbhip_ba_batch_destroy(b); g_ba_error = "page-locked staging allocation failed"; return ORBHIP_E_HIP; }
        for (const BaPlan::It &it : plan.items) {
            *it.dst = arena + it.off;
            if (it.src && it.bytes) memcpy(stage + it.off, it.src, it.bytes);
        }
        // the staging area belongs to the context and the next host-pointer call may reuse it: the copy is waited for here
        bool up_ok = hipMemcpyAsync(arena, stage, plan.up_end, hipMemcpyHostToDevice, st) == hipSuccess;
        if (up_ok && pair_lists && max_npair > 0) {              // the pair lists, from the lists just uploaded (stream order)
            const dim3 grid((unsigned)((max_npair + 3) / 4), (unsigned)n_graphs);
            hipLaunchKernelGGL(k_ba_pair_lists<0>, grid, dim3(256), 0, st, B);
            hipLaunchKernelGGL(k_ba_pair_scan, dim3((unsigned)n_graphs), dim3(256), 0, st, B);
            hipLaunchKernelGGL(k_ba_pair_lists<1>, grid, dim3(256), 0, st, B);
            up_ok = hipGetLastError() == hipSuccess;
        }
        if (!up_ok || hipStreamSynchronize(st) != hipSuccess) {
            orbhip_ba_batch_destroy(b); g_ba_error = "upload of the graph failed"; return ORBHIP_E_HIP;
        }
        if (oneshot && (b->h_n_active = orbhip_ctx_pinned_word_internal(ctx))) b->ctx_word = true;
        else if (hipHostMalloc((void **)&b->h_n_active, sizeof(int)) != hipSuccess) { b->h_n_active = nullptr; orbhip_ba_batch_destroy(b); g_ba_error = "device allocation failed"; return ORBHIP_E_HIP; }
    }
    *out = b;
    return ORBHIP_OK;
}

extern "C" int orbhip_ba_batch_create(orbhip_ctx *ctx, const orbhip_ba_graph *graphs, int n_graphs,
                                      double *const *poses, double *const *points, orbhip_ba_batch **out)
{
    return ba_create_impl(ctx, graphs, n_graphs, poses, points, nullptr, 0, 1, out);
}

// Landmark-sharded batch (optional single-graph mode): every rank passes the SAME full graphs and estimates; rank r
// keeps points [r*L/world, (r+1)*L/world) of every graph with their edges (contiguous: edges are point-major) and all poses.
extern "C" int orbhip_ba_batch_create_sharded(orbhip_ctx *ctx, const orbhip_ba_graph *graphs, int n_graphs, double *const *poses,
                                              double *const *points, int rank, int world, orbhip_ba_batch **out)
{
    if (!ctx || !graphs || n_graphs <= 0 || !poses || !points || !out || world < 1 || rank < 0 || rank >= world) return ORBHIP_E_BADARG;
    std::vector<orbhip_ba_graph> sub(graphs, graphs + n_graphs);
    std::vector<std::vector<int32_t>> ept(n_graphs);
    std::vector<std::vector<uint8_t>> has(n_graphs);
    std::vector<const double *> pts(n_graphs);
    std::vector<orbhip_ba_batch::Slice> slices(n_graphs);
    for (int g = 0; g < n_graphs; g++) {
        const orbhip_ba_graph &H = graphs[g];
        if (H.n_poses <= 0 || H.n_points < world || H.n_edges < 0) { g_ba_error = "a sharded graph needs at least one point per rank"; return ORBHIP_E_BADARG; }
        has[g].assign(H.n_poses, 0);
        for (int e = 0; e < H.n_edges; e++) {
            if (H.edge_pose[e] < 0 || H.edge_pose[e] >= H.n_poses || H.edge_point[e] < 0 || H.edge_point[e] >= H.n_points ||
                (e > 0 && H.edge_point[e] < H.edge_point[e - 1])) { g_ba_error = "edges must be point-major with valid ids"; return ORBHIP_E_BADARG; }
            has[g][H.edge_pose[e]] = 1;
        }
        const int p0 = (int)((long long)rank * H.n_points / world), p1 = (int)((long long)(rank + 1) * H.n_points / world);
        int e0 = 0;
        while (e0 < H.n_edges && H.edge_point[e0] < p0) e0++;
        int e1 = e0;
        while (e1 < H.n_edges && H.edge_point[e1] < p1) e1++;
        slices[g] = {p0, p1 - p0, e0, e1 - e0};
        ept[g].resize(e1 - e0);
        for (int e = e0; e < e1; e++) ept[g][e - e0] = H.edge_point[e] - p0;
        orbhip_ba_graph &S = sub[g];
        S.n_points = p1 - p0; S.n_edges = e1 - e0;
        S.edge_pose = H.edge_pose + e0; S.edge_point = ept[g].data(); S.edge_obs = H.edge_obs + 3 * (size_t)e0;
        S.edge_inv_sigma2 = H.edge_inv_sigma2 + e0; S.edge_stereo = H.edge_stereo ? H.edge_stereo + e0 : nullptr;
        pts[g] = points[g] + 3 * (size_t)p0;
    }
    const int rc = ba_create_impl(ctx, sub.data(), n_graphs, poses, pts.data(), &has, rank, world, out);
    if (rc == ORBHIP_OK) (*out)->slices = slices;
    return rc;
}

extern "C" size_t orbhip_ba_batch_exchange_doubles(const orbhip_ba_batch *b) { return b ? b->x_need : 0; }

// Runs optimize(iters1) + optimize(iters2) + outlier classification for every graph of the batch,
// starting from the initial estimates given at creation.  Device-resident; returns after completion.
static int ba_solve_impl(orbhip_ba_batch *b, const orbhip_ba_params *params, volatile const uint8_t *abort_flag, orbhip_ba_exchange_fn xch, void *xuser)
{
    if (!b || !params) return ORBHIP_E_BADARG;
    const bool sharded = b->B.world > 1;
    if (sharded && (!xch || !b->B.xbuf)) { g_ba_error = "a sharded batch needs orbhip_ba_batch_set_exchange_buffer and an exchange callback"; return ORBHIP_E_BADARG; }
    if (hipSetDevice(orbhip_ctx_device_internal(b->ctx)) != hipSuccess) return ORBHIP_E_HIP;
    if (!sharded && abort_flag && *abort_flag) return ORBHIP_E_ABORTED;       // Optimizer.cc:2041-2043 (sharded: the flag travels through exchange 3)
    hipStream_t s = orbhip_ctx_stream_internal(b->ctx);
    BaBatch &B = b->B;
    B.delta_m = (double)(float)sqrt(params->huber_mono2); B.dsqr_m = (double)(float)(B.delta_m * B.delta_m);
    B.delta_s = (double)(float)sqrt(params->huber_stereo2); B.dsqr_s = (double)(float)(B.delta_s * B.delta_s);
    B.gate_m = params->gate_mono2 > 0 ? params->gate_mono2 : params->huber_mono2;
    B.gate_s = params->gate_stereo2 > 0 ? params->gate_stereo2 : params->huber_stereo2;
    B.ex2 = params->stage2_exclude_outliers; B.nr2 = params->stage2_drop_robust; b->no_discard = params->no_discard != 0;
    B.user_lambda = params->user_lambda_init; B.tau = params->tau;
    B.iters[0] = params->iters1; B.iters[1] = params->iters2; B.max_trials = params->max_trials;
    // reset state + estimates
    std::vector<BaState> st(B.G);
    for (auto &x : st) {
        memset(&x, 0, sizeof(x));
        x.need_build = 1; x.need_lambda_init = 1; x.active = 1; x.ok = 1; x.chi_first = -1.0; x.robust = 1;
    }
    if (params->iters1 <= 0) for (auto &x : st) { x.pass = 1; }
    int n_active = B.G;
    if (params->iters1 <= 0 && params->iters2 <= 0) { for (auto &x : st) x.active = 0; n_active = 0; }
#define TRY(e) do { if ((e) != hipSuccess) { g_ba_error = #e; return ORBHIP_E_HIP; } } while (0)
    TRY(hipMemcpyAsync(B.st, st.data(), sizeof(BaState) * B.G, hipMemcpyHostToDevice, s));
    TRY(hipMemcpyAsync(B.poses, b->poses0.data(), sizeof(double) * b->poses0.size(), hipMemcpyHostToDevice, s));
    TRY(hipMemcpyAsync(B.points, b->points0.data(), sizeof(double) * b->points0.size(), hipMemcpyHostToDevice, s));
    TRY(hipMemcpyAsync(B.n_active, &n_active, sizeof(int), hipMemcpyHostToDevice, s));
    TRY(hipMemsetAsync(B.xp, 0, sizeof(double) * (size_t)B.sumF * 6, s));
    TRY(hipMemsetAsync(B.xl, 0, sizeof(double) * (size_t)B.sumL * 3, s));
    TRY(hipMemsetAsync(B.level, 0, (size_t)std::max(B.sumE, 1), s));
    TRY(hipMemsetAsync(B.x_abort, 0, sizeof(int), s));
    TRY(hipStreamSynchronize(s));     // st / n_active host buffers must outlive the copies
    // sharded: the caller's all-gather runs between two kernels; the stream is drained around it
#define XCHG(stage, cnt) do { TRY(hipStreamSynchronize(s)); if (xch(xuser, (stage), (size_t)(cnt)) != 0) { g_ba_error = "exchange callback failed"; return ORBHIP_E_HIP; } } while (0)
    const size_t x1n = b->x_need, x3n = (size_t)3 * B.G;
    const int G = B.G;
    const dim3 ge((B.max_edges + 255) / 256, G), gp128((B.max_points + 127) / 128, G), gp256((B.max_points + 255) / 256, G);
    const dim3 gf(std::max(B.max_nf, 1), G);
    int max_items = 0, max_poses = 0;
    for (auto &D : b->gd) { max_items = std::max(max_items, D.ks * D.ngrp); max_poses = std::max(max_poses, D.n_poses); }
    size_t gemm_lds = 0;
    for (auto &D : b->gd) gemm_lds = std::max(gemm_lds, sizeof(double) * (2 * (size_t)D.gemm_ps * 3 * (size_t)(D.ld + GEMM_LDS_PAD) + GEMM_LDS_TAIL));
    if (orb_lds_optin(reinterpret_cast<const void *>(k_ba_schur_gemm), orbhip_ctx_device_internal(b->ctx), gemm_lds)) { g_ba_error = "LDS opt-in (k_ba_schur_gemm)"; return ORBHIP_E_HIP; }
    const size_t ldlt_lds = B.big ? 0 : ba_ldlt_lds_bytes(B.max_ld);
    if (!B.big && orb_lds_optin(reinterpret_cast<const void *>(k_ba_ldlt), orbhip_ctx_device_internal(b->ctx), ldlt_lds)) { g_ba_error = "LDS opt-in (k_ba_ldlt)"; return ORBHIP_E_HIP; }
    const size_t rows_lds = sizeof(double) * ((size_t)(LD_NB + 256) * LD_PP + LD_NB * LD_NB + LD_NB + LD_NB + 256);
    const size_t bsub_lds = sizeof(double) * ((size_t)B.max_ld + 32 * 32 + LD_NB * LD_PP);
    int max_n = 0, max_nfp = 0;
    for (auto &D : b->gd) { max_n = std::max(max_n, D.n); max_nfp = std::max(max_nfp, D.nf * (D.nf + 1) / 2); }
    if (B.big) {
        if (orb_lds_optin(reinterpret_cast<const void *>(k_ba_big_rows), orbhip_ctx_device_internal(b->ctx), rows_lds) ||
            orb_lds_optin(reinterpret_cast<const void *>(k_ba_big_backsub), orbhip_ctx_device_internal(b->ctx), bsub_lds)) { g_ba_error = "LDS opt-in (big LDLT)"; return ORBHIP_E_HIP; }
    }
    // the row-owner form of the pair Schur kernel: whenever every pose's blocks fit the LDS (ORBHIP_BA_SCHUR_ROWS=0: keep k_ba_schur_big, for A/B runs)
    const size_t srow_lds = sizeof(double) * 18 * (size_t)std::max(b->max_row_blocks, 1);
    bool schur_rows = B.pair_schur && b->max_row_blocks <= SROW_MAX_BLOCKS && !(getenv("ORBHIP_BA_SCHUR_ROWS") && atoi(getenv("ORBHIP_BA_SCHUR_ROWS")) == 0);
    if (schur_rows && orb_lds_optin(reinterpret_cast<const void *>(k_ba_schur_rows), orbhip_ctx_device_internal(b->ctx), srow_lds)) { g_ba_error = "LDS opt-in (k_ba_schur_rows)"; return ORBHIP_E_HIP; }
    const int max_ticks = (params->iters1 + params->iters2) * params->max_trials + 4;
    int tick = 0;
    // one LM tick: every graph that is still active evaluates, builds, solves and tries one step (inactive graphs return at once)
    auto launch_tick = [&](int ab) -> int {
        if (B.ex2) hipLaunchKernelGGL(k_ba_levels, ge, dim3(256), 0, s, B);
        if (b->general) hipLaunchKernelGGL(k_ba_errors<true>, ge, dim3(256), 0, s, B, 0); else hipLaunchKernelGGL(k_ba_errors<false>, ge, dim3(256), 0, s, B, 0);
        hipLaunchKernelGGL(k_ba_reduce, dim3(G), dim3(1024), 0, s, B, 0);
        if (b->general) hipLaunchKernelGGL(k_ba_build_points<true>, dim3((B.max_points + 15) / 16, G), dim3(256), 0, s, B);
        else hipLaunchKernelGGL(k_ba_build_points<false>, dim3((B.max_points + 15) / 16, G), dim3(256), 0, s, B);
        if (b->general) hipLaunchKernelGGL(k_ba_build_poses<true>, gf, dim3(64), 0, s, B); else hipLaunchKernelGGL(k_ba_build_poses<false>, gf, dim3(64), 0, s, B);
        hipLaunchKernelGGL(k_ba_maxdiag, dim3(G), dim3(256), 0, s, B);
        if (sharded) {                                           // exchange 1: Hpp, bp, chi2, max |Hll diag|
            hipLaunchKernelGGL(k_ba_shard_pack1, dim3(G), dim3(256), 0, s, B);
            XCHG(1, x1n);
            hipLaunchKernelGGL(k_ba_shard_sum1, dim3(G), dim3(256), 0, s, B);
        }
        hipLaunchKernelGGL(k_ba_pretrial, dim3((G + 63) / 64), dim3(64), 0, s, B);
        hipLaunchKernelGGL(k_ba_point_prep, gp256, dim3(256), 0, s, B);
        if (b->profile) TRY(hipEventRecord(b->ev0, s));
        if (B.pair_schur && max_nfp > 0) {                       // (a batch whose poses are all fixed has no reduced system: points only)
            if (G >= 8 && schur_rows) hipLaunchKernelGGL(k_ba_schur_rows, dim3((unsigned)(B.max_nf * (((G + 7) / 8) * 8))), dim3(SROW_THREADS), srow_lds, s, B, B.max_nf);
            else if (G >= 8) hipLaunchKernelGGL(k_ba_schur_big<16>, dim3((unsigned)(((max_nfp + 3) / 4) * (((G + 7) / 8) * 8))), dim3(64), 0, s, B, (max_nfp + 3) / 4);
            else hipLaunchKernelGGL(k_ba_schur_big<64>, dim3((unsigned)(max_nfp * G)), dim3(64), 0, s, B, max_nfp);
        }
        else if (!B.pair_schur) hipLaunchKernelGGL(k_ba_schur_gemm, dim3(max_items, G), dim3(64 * GEMM_WAVES), gemm_lds, s, B);
        if (b->profile) TRY(hipEventRecord(b->ev1, s));
        if (!B.pair_schur) hipLaunchKernelGGL(k_ba_bschur, gf, dim3(64), 0, s, B);      // (the pair kernel's diagonal rows produced bs)
        if (sharded) {                                           // exchange 2: the shared Schur block (sum_ks Spart) and W db
            hipLaunchKernelGGL(k_ba_shard_pack2, dim3((B.max_ld * B.max_ld + 255) / 256, G), dim3(256), 0, s, B);
            XCHG(2, x1n);
            hipLaunchKernelGGL(k_ba_shard_sum2, dim3(G), dim3(256), 0, s, B);
        }
        if (!B.pair_schur) hipLaunchKernelGGL(k_ba_schur_finish, dim3((B.max_ld * B.max_ld + 255) / 256, G), dim3(256), 0, s, B);      // (the pair kernel wrote the finished S)
        if (!B.big) hipLaunchKernelGGL(k_ba_ldlt, dim3(G), dim3(1024), ldlt_lds, s, B);
        else {
            hipLaunchKernelGGL(k_ba_big_init, dim3((max_n + 255) / 256, G), dim3(256), 0, s, B);
            for (int p0 = 0; p0 < max_n; p0 += LD_NB) {
                const int m2 = max_n - p0 - LD_NB;
                hipLaunchKernelGGL(k_ba_big_diag, dim3(G), dim3(64), 0, s, B, p0);
                if (m2 > 0) {
                    hipLaunchKernelGGL(k_ba_big_rows, dim3((m2 + 255) / 256, G), dim3(256), rows_lds, s, B, p0);
                    const int nt = (m2 + 63) / 64;
                    hipLaunchKernelGGL(k_ba_big_trail, dim3(nt, nt, G), dim3(256), 0, s, B, p0);
                }
            }
            hipLaunchKernelGGL(k_ba_big_backsub, dim3(G), dim3(1024), bsub_lds, s, B);
        }
        hipLaunchKernelGGL(k_ba_backsub_points, dim3((B.max_points + 15) / 16, G), dim3(256), 0, s, B);
        hipLaunchKernelGGL(k_ba_update_poses, dim3((max_poses + 63) / 64, G), dim3(64), 0, s, B);
        if (b->general) hipLaunchKernelGGL(k_ba_errors<true>, ge, dim3(256), 0, s, B, 1); else hipLaunchKernelGGL(k_ba_errors<false>, ge, dim3(256), 0, s, B, 1);
        hipLaunchKernelGGL(k_ba_reduce, dim3(G), dim3(1024), 0, s, B, 1);
        if (sharded) {                                           // exchange 3: trial chi2, computeScale, abort flag
            hipLaunchKernelGGL(k_ba_shard_pack34, dim3((G + 63) / 64), dim3(64), 0, s, B, 3, ab);
            XCHG(3, x3n);
            hipLaunchKernelGGL(k_ba_shard_sum34, dim3((G + 63) / 64), dim3(64), 0, s, B, 3);
        }
        hipLaunchKernelGGL(k_ba_control, dim3((G + 63) / 64), dim3(64), 0, s, B, ab);
        return ORBHIP_OK;
    };
    // Replay form of a tick (no exchange, no event timing, abort flag down): captured once per batch and parameter set.  The
    // kernels of a graph that has finished return at once, so several ticks may be queued per host round trip.
    const bool trace = getenv("ORBHIP_BA_TRACE") != nullptr;      // development: graph 0's LM state after every tick on stderr (one tick per round trip)
    // (a one-shot solve would capture, instantiate and destroy a graph for ~8 replays: ORBHIP_BA_ONESHOT_GRAPH=0 / 1 decides)
    static const int oneshot_graph = getenv("ORBHIP_BA_ONESHOT_GRAPH") ? atoi(getenv("ORBHIP_BA_ONESHOT_GRAPH")) : 0;
    const bool use_graph = !sharded && !b->profile && !trace && (!b->ctx_arena || oneshot_graph);
    if (use_graph && !(b->tick_graph_valid && memcmp(&b->tick_B, &B, sizeof(BaBatch)) == 0)) {
        if (b->tick_graph_valid) { (void)hipGraphExecDestroy(b->tick_graph); b->tick_graph_valid = false; }
        hipGraph_t graph = nullptr;
        TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rc = launch_tick(0);
        const hipError_t ce = hipStreamEndCapture(s, &graph);
        if (rc != ORBHIP_OK || ce != hipSuccess) { if (graph) (void)hipGraphDestroy(graph); g_ba_error = "tick capture"; return ORBHIP_E_HIP; }
        const hipError_t ie = hipGraphInstantiate(&b->tick_graph, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) { g_ba_error = "hipGraphInstantiate(tick)"; return ORBHIP_E_HIP; }
        b->tick_graph_valid = true; b->tick_B = B;
    }
    const int ticks_per_sync = use_graph ? (G <= 16 ? 4 : 2) : 1;
    while (tick < max_ticks && n_active > 0) {
        const int ab = (abort_flag && *abort_flag) ? 1 : 0;
        if (use_graph && !ab) {
            for (int k = 0; k < ticks_per_sync && tick < max_ticks; k++, tick++) TRY(hipGraphLaunch(b->tick_graph, s));
        } else {
            const int rc = launch_tick(ab);
            if (rc != ORBHIP_OK) return rc;
            tick++;
        }
        TRY(hipMemcpyAsync(b->h_n_active, B.n_active, sizeof(int), hipMemcpyDeviceToHost, s));
        TRY(hipStreamSynchronize(s));
        TRY(hipGetLastError());                  // a rejected launch fails here, loudly, instead of spinning to max_ticks
        n_active = *b->h_n_active;
        if (trace) {
            BaState t0;
            TRY(hipMemcpy(&t0, B.st, sizeof(BaState), hipMemcpyDeviceToHost));
            fprintf(stderr, "[orbhip ba] tick %d: pass %d iter %d qmax %d lambda %.6e chi %.9e rho %.6e ok %d nbad %d active %d trials %d\n", tick, t0.pass, t0.iter, t0.qmax,
                    t0.lambda, t0.current_chi, t0.rho_dbg, t0.ok, t0.nbad, t0.active, t0.lm_trials);
        }
        if (b->profile) {
            float ms = 0;
            TRY(hipEventElapsedTime(&ms, b->ev0, b->ev1));
            b->gemm_ms_total += ms; b->gemm_launches++;
        }
    }
    b->ticks_last = tick;
    hipLaunchKernelGGL(k_ba_finalize, ge, dim3(256), 0, s, B);
    if (sharded) {                                               // exchange 4: outlier and edge counts for the >= 50 % rule
        hipLaunchKernelGGL(k_ba_shard_pack34, dim3((G + 63) / 64), dim3(64), 0, s, B, 4, 0);
        XCHG(4, x3n);
        hipLaunchKernelGGL(k_ba_shard_sum34, dim3((G + 63) / 64), dim3(64), 0, s, B, 4);
    }
    TRY(hipStreamSynchronize(s));
    TRY(hipGetLastError());
#undef XCHG
#undef TRY
    return ORBHIP_OK;
}

extern "C" int orbhip_ba_batch_solve(orbhip_ba_batch *b, const orbhip_ba_params *params, volatile const uint8_t *abort_flag)
{
    if (b && b->B.world > 1) { g_ba_error = "sharded batch: use orbhip_ba_batch_solve_sharded"; return ORBHIP_E_BADARG; }
    return ba_solve_impl(b, params, abort_flag, nullptr, nullptr);
}

extern "C" int orbhip_ba_batch_set_exchange_buffer(orbhip_ba_batch *b, double *d_buf, size_t capacity_doubles)
{
    if (!b || !d_buf || capacity_doubles < (size_t)b->B.world * b->x_need) return ORBHIP_E_BADARG;
    b->B.xbuf = d_buf; b->B.xcount = b->x_need;
    return ORBHIP_OK;
}

extern "C" int orbhip_ba_batch_solve_sharded(orbhip_ba_batch *b, const orbhip_ba_params *params, volatile const uint8_t *abort_flag,
                                             orbhip_ba_exchange_fn exchange, void *user)
{
    if (!b || !exchange) return ORBHIP_E_BADARG;
    if (b->B.world == 1) return ba_solve_impl(b, params, abort_flag, nullptr, nullptr);
    return ba_solve_impl(b, params, abort_flag, exchange, user);
}

// D2H of the results of the last solve.  poses_out[g]/points_out[g] are written unless the graph
// was discarded (>= 50 % outliers, Optimizer.cc:2177-2181); outlier/stats may be NULL.
extern "C" int orbhip_ba_batch_download(orbhip_ba_batch *b, double *const *poses_out, double *const *points_out,
                                        uint8_t *const *edge_outlier_out, orbhip_ba_stats *stats_out)
{
    if (!b) return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(b->ctx)) != hipSuccess) return ORBHIP_E_HIP;
    BaBatch &B = b->B;
    std::vector<BaState> st(B.G);
    std::vector<double> poses((size_t)2 * B.sumP * 7), points((size_t)2 * B.sumL * 3);
    std::vector<uint8_t> outl(B.sumE ? B.sumE : 1);
    if (hipMemcpy(st.data(), B.st, sizeof(BaState) * B.G, hipMemcpyDeviceToHost) != hipSuccess) return ORBHIP_E_HIP;
    if (hipMemcpy(poses.data(), B.poses, poses.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ORBHIP_E_HIP;
    if (hipMemcpy(points.data(), B.points, points.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ORBHIP_E_HIP;
    if (B.sumE && hipMemcpy(outl.data(), B.outlier, B.sumE, hipMemcpyDeviceToHost) != hipSuccess) return ORBHIP_E_HIP;
    std::vector<double> etot(B.G, 0.0);
    if (B.world > 1 && hipMemcpy(etot.data(), B.edges_total, sizeof(double) * B.G, hipMemcpyDeviceToHost) != hipSuccess) return ORBHIP_E_HIP;
    for (int g = 0; g < B.G; g++) {
        const BaGraphDev &D = b->gd[g];
        const double n_edges_all = B.world > 1 ? etot[g] : (double)D.n_edges;          // sharded: counts of all ranks (exchange 4)
        const int discarded = (!b->no_discard && n_edges_all > 0 && st[g].n_outliers >= n_edges_all * 0.5) ? 1 : 0;
        const size_t pt0 = b->slices.empty() ? 0 : (size_t)b->slices[g].pt0, e0 = b->slices.empty() ? 0 : (size_t)b->slices[g].e0;
        if (!discarded) {                                      // sharded: this rank's points / edges land at their place in the full arrays
            if (poses_out && poses_out[g]) memcpy(poses_out[g], poses.data() + ((size_t)st[g].cur * B.sumP + D.pose_off) * 7, sizeof(double) * 7 * D.n_poses);
            if (points_out && points_out[g]) memcpy(points_out[g] + 3 * pt0, points.data() + ((size_t)st[g].cur * B.sumL + D.point_off) * 3, sizeof(double) * 3 * D.n_points);
        }
        if (edge_outlier_out && edge_outlier_out[g]) memcpy(edge_outlier_out[g] + e0, outl.data() + D.edge_off, D.n_edges);
        if (stats_out) {
            orbhip_ba_stats &o = stats_out[g];
            o.iterations_run[0] = st[g].iters_run[0]; o.iterations_run[1] = st[g].iters_run[1];
            o.lm_trials = st[g].lm_trials; o.n_outliers = st[g].n_outliers; o.discarded = discarded;
            o.chi2_initial = st[g].chi_first; o.chi2_final = st[g].chi_last;
        }
    }
    return ORBHIP_OK;
}

extern "C" int orbhip_ba_batch_ticks(const orbhip_ba_batch *b) { return b ? b->ticks_last : ORBHIP_E_BADARG; }
#ifdef LDLT_PROF
extern "C" int orbhip_debug_ldlt_prof(long long *out8, int reset)      // debug build only (EXTRA=-DLDLT_PROF): cumulative cycles of ldlt_solve_wg's phases
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ldlt_prof), 64) != hipSuccess) return ORBHIP_E_HIP;
    if (reset) { long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ldlt_prof), z, 64) != hipSuccess) return ORBHIP_E_HIP; }
    return ORBHIP_OK;
}
#endif

extern "C" int orbhip_ba_batch_set_profiling(orbhip_ba_batch *b, int enable)
{
    if (!b) return ORBHIP_E_BADARG;
    if (enable && !b->ev0) {
        if (hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess) return ORBHIP_E_HIP;
    }
    b->profile = enable != 0; b->gemm_ms_total = 0; b->gemm_launches = 0;
    return ORBHIP_OK;
}

// Accumulated device time / launch count of the Schur GEMM kernel since profiling was enabled,
// and the MFMA flops one launch really issues (upper 16x16 tiles, static block-sparsity masks).
extern "C" int orbhip_ba_batch_gemm_profile(const orbhip_ba_batch *b, float *total_ms, int *launches, double *flops_per_launch)
{
    if (!b || !total_ms || !launches || !flops_per_launch) return ORBHIP_E_BADARG;
    *total_ms = b->gemm_ms_total; *launches = b->gemm_launches; *flops_per_launch = b->gemm_flops_per_launch;
    return ORBHIP_OK;
}
extern "C" double orbhip_ba_batch_gemm_dense_flops(const orbhip_ba_batch *b) { return b ? b->gemm_flops_dense : 0.0; }
extern "C" double orbhip_ba_batch_gemm_issued_flops(const orbhip_ba_batch *b) { return b ? b->gemm_flops_issued : 0.0; }

// FP64 matrix-core peak of this device, measured: every wave issues independent
// v_mfma_f64_16x16x4_f64 chains (the local micro-architecture guide lists no FP64 MFMA peak).
__global__ __launch_bounds__(256) void k_mfma_f64_peak(double *sink, int iters)
{
    v4d acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (v4d){0, 0, 0, 0};
    const double a = 1.0 + threadIdx.x * 1e-9, bb = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;
}

extern "C" int orbhip_mfma_f64_peak_tflops(orbhip_ctx *ctx, double *tflops_out)
{
    if (!ctx || !tflops_out) return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    hipStream_t s = orbhip_ctx_stream_internal(ctx);
    double *sink = nullptr;
    if (hipMalloc((void **)&sink, 8) != hipSuccess) return ORBHIP_E_HIP;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 4096;
    hipLaunchKernelGGL(k_mfma_f64_peak, dim3(blocks), dim3(256), 0, s, sink, 64);          // warm-up
    (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(k_mfma_f64_peak, dim3(blocks), dim3(256), 0, s, sink, iters);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink);
    const double flops = (double)blocks * 4.0 * iters * 8.0 * 2048.0;
    *tflops_out = flops / (ms * 1e-3) / 1e12;
    return ORBHIP_OK;
}

extern "C" int orbhip_ba_solve_batch(orbhip_ctx *ctx, const orbhip_ba_graph *graphs, int n_graphs,
                                     const orbhip_ba_params *params, volatile const uint8_t *abort_flag,
                                     double *const *poses_inout, double *const *points_inout,
                                     uint8_t *const *edge_outlier_out, orbhip_ba_stats *stats_out)
{
    if (abort_flag && *abort_flag) return ORBHIP_E_ABORTED;
    orbhip_ba_batch *b = nullptr;
    static const bool prof = getenv("ORBHIP_HOST_PROF") != nullptr;       // where a one-shot call spends its time (host_smoke latency)
    const auto t0 = std::chrono::steady_clock::now();
    int rc = ba_create_impl(ctx, graphs, n_graphs, poses_inout, points_inout, nullptr, 0, 1, &b, true);     // borrows the context's cached arena
    if (rc) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    rc = orbhip_ba_batch_solve(b, params, abort_flag);
    const auto t2 = std::chrono::steady_clock::now();
    if (rc == ORBHIP_OK) rc = orbhip_ba_batch_download(b, poses_inout, points_inout, edge_outlier_out, stats_out);
    const auto t3 = std::chrono::steady_clock::now();
    orbhip_ba_batch_destroy(b);
    if (prof) {
        const auto t4 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
        fprintf(stderr, "[orbhip ba] one-shot solve of %d graph(s): create %.3f  solve %.3f  download %.3f  destroy %.3f ms\n", n_graphs, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4));
    }
    return rc;
}


// ====================================================================== pose-only BA
// Optimizer::PoseOptimization (Optimizer.cc:854-1168).  One 256-thread workgroup per frame; thread t owns
// edges t, t+256, ... (outlier level = one bit per owned edge).  The 6x6 system is tiny, so every thread keeps
// the whole LM state (pose, lambda, ...) in registers and runs the scalar control flow redundantly on the
// block-reduced sums: no broadcasts, two barriers per reduction, fixed summation order (run-to-run identical).
#define PO_THREADS 256
#define PO_NRED 28            // robust chi2 + 21 upper-triangle entries of H + 6 of b
#define PO_KR 4               // edges per thread held in registers (x 256 threads: frames of up to 1024 edges never re-read them)
#define PO_IDX(a, c) ((a) * 6 - (a) * ((a) - 1) / 2 + ((c) - (a)))      // packed upper triangle, a <= c
struct PoArgs {
    const double *Xw, *obs, *inv_s2;
    const int32_t *n;
    int max_edges;
    double fx, fy, cx, cy, bf;
    int cam_model; double kb[4];
    const uint8_t *right;               // [frames][max_edges] 1 = observation in the second camera (may be NULL)
    double Trl[7], fx2, fy2, cx2, cy2, kb2[4]; int cam2_model;
    double *pose;
    uint8_t *outlier;
    int32_t *n_inliers, *stats;
    int stage_cap;                      // edges per frame the four-wave kernel stages in LDS (0: none)
};

#ifdef PO_PROF
__device__ long long g_po_prof[8];            // debug build only (EXTRA=-DPO_PROF): cycles of build walk / 28 block sums / solve + oplus / trial walk / its sum / reclassification, trials
#define PO_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_ = clock64(); g_po_prof[i] += t_ - t_prev; t_prev = t_; } } while (0)
extern "C" int orbhip_debug_po_prof(long long *out8, int reset)
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_po_prof), 64) != hipSuccess) return -1;
    if (reset) { long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_po_prof), z, 64) != hipSuccess) return -1; }
    return 0;
}
#else
#define PO_T(i) do { } while (0)
#endif
template <int N, int NT = 256>
__device__ __forceinline__ void po_block_sum(double (&v)[N], double (*red)[PO_NRED])
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; k++) {
        v[k] = wave_sum_f64_dpp(v[k]);                    // DPP path: the ~100 block sums per frame were ds_bpermute bound
    }
    if (NT == 64) return;                              // one wave per frame: no LDS, no barrier
    __syncthreads();                                   // previous readers of red are done
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; k++) red[wv][k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) {                       // the waves' sums in wave order (a fixed association per NT)
        double s = red[0][k];
#pragma unroll
        for (int w = 1; w < NT / 64; w++) s += red[w][k];
        v[k] = s;
    }
}

// The 28 sums of a build step for the four-wave forms (round 4).  28 wave-level DPP trees of doubles were 3.8 k cycles per step -- 17 % of
// a one-frame call (tools/po_prof_probe.py) -- because every one of them is a chain of 6 dependent (2 x v_mov_dpp + v_add_f64).  Here the
// workgroup transposes through LDS instead: every thread stores its 28 partial sums ([k][8 segments of 32 threads, padded to 33]), thread
// (k, seg) adds the 32 entries of its segment (32 independent LDS reads, a balanced tree), the 8 segment sums of a k meet
// in three DPP steps inside 8 adjacent lanes (quad swaps + half-row mirror: every lane ends with the same bits), one lane per k publishes
// the total and every thread reads the 28 totals back (LDS broadcast).  Two barriers as before; a fixed association (run-to-run identical).
#define PO_RED2_K 265                                  // doubles per k: 8 segments x 33
__device__ __forceinline__ double po_dpp_f64(double v, const int ctrl_sel)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned lo, hi;
    if (ctrl_sel == 0) { lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), 0xB1, 0xF, 0xF, false); hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u >> 32), 0xB1, 0xF, 0xF, false); }        // quad_perm:[1,0,3,2]
    else if (ctrl_sel == 1) { lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), 0x4E, 0xF, 0xF, false); hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u >> 32), 0x4E, 0xF, 0xF, false); }   // quad_perm:[2,3,0,1]
    else { lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), 0x141, 0xF, 0xF, false); hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u >> 32), 0x141, 0xF, 0xF, false); }                  // row_half_mirror
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int N>
__device__ __forceinline__ void po_block_sum_lds(double (&v)[N], double *red2, double *tot)
{
    static_assert(N * 8 <= 256, "one (k, segment) task per thread");
    const int tid = threadIdx.x;
    double *mine = red2 + (tid >> 5) * 33 + (tid & 31);
#pragma unroll
    for (int k = 0; k < N; k++) mine[k * PO_RED2_K] = v[k];
    __syncthreads();
    {
        const int k = min(tid >> 3, N - 1), seg = tid & 7;
        const double *p = red2 + k * PO_RED2_K + seg * 33;
        double t[32];
#pragma unroll
        for (int i = 0; i < 32; i++) t[i] = p[i];
        // a balanced tree over the 32 entries (depth 5 instead of a chain of 31 dependent additions: the (k, segment) thread has nothing
        // else to overlap its chain with); the same association in every call, whatever N
#pragma unroll
        for (int w = 16; w >= 1; w >>= 1) {
#pragma unroll
            for (int i = 0; i < w; i++) t[i] = t[2 * i] + t[2 * i + 1];
        }
        double sacc = t[0];
        sacc += po_dpp_f64(sacc, 0);
        sacc += po_dpp_f64(sacc, 1);
        sacc += po_dpp_f64(sacc, 2);
        if (seg == 0 && (tid >> 3) < N) tot[k] = sacc;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = tot[k];
}

// computeError of the two unary edges; returns chi2 = e^T (inv_sigma2 I) e
template <bool GENERAL>
__device__ __forceinline__ double po_edge_chi2(const PoArgs &A, const BaGraphDev &cam, const double *pose, const double *X, const double *ob,
                                               double is2, int right, double *P, double *er)
{
    if (GENERAL && right) {                                       // EdgeSE3ProjectXYZOnlyPoseToBody::computeError, OptimizableTypes.h:69-73
        tobody_error(cam, pose, X, ob, P, er);
        return (er[0] * er[0] + er[1] * er[1]) * is2;
    }
    quat_rot(pose, X, P);
    P[0] += pose[4]; P[1] += pose[5]; P[2] += pose[6];
    if (GENERAL && ob[2] < 0 && A.cam_model == 1) {    // OptimizableTypes.h:41-45 with pCamera = KannalaBrandt8 (:52-69)
        const double x2y2 = P[0] * P[0] + P[1] * P[1];
        const double theta = (double)(float)atan2((double)sqrtf((float)x2y2), (double)(float)P[2]);
        const double psi = (double)(float)atan2((double)(float)P[1], (double)(float)P[0]);
        const double t2 = theta * theta, t3 = theta * t2, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
        const double r = theta + A.kb[0] * t3 + A.kb[1] * t5 + A.kb[2] * t7 + A.kb[3] * t9;
        er[0] = ob[0] - (A.fx * r * cos(psi) + A.cx);
        er[1] = ob[1] - (A.fy * r * sin(psi) + A.cy);
        er[2] = 0;
    } else if (ob[2] < 0) {                            // OptimizableTypes.h:41-45, Pinhole.cpp:41-47
        er[0] = ob[0] - (A.fx * P[0] / P[2] + A.cx);
        er[1] = ob[1] - (A.fy * P[1] / P[2] + A.cy);
        er[2] = 0;
    } else {                                           // types_six_dof_expmap.cpp:339-346: float invz, double bf
        const float invz = (float)(1.0 / P[2]);
        const double r0 = P[0] * invz * A.fx + A.cx;
        er[0] = ob[0] - r0;
        er[1] = ob[1] - (P[1] * invz * A.fy + A.cy);
        er[2] = ob[2] - (r0 - A.bf * invz);
    }
    return (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * is2;
}

// GENERAL = false: Pinhole camera, no second camera (the common case keeps its registers); true: KannalaBrandt8 and / or
// observations in a second, rigidly attached camera.
// NT threads per frame, KR edges per thread in registers.  <256, 0>: four waves per frame, edges re-read from global memory on every walk
// (frames of more than 2048 edges).  <64, 16>: ONE wave per frame -- up to 1024 edges live in registers (the LM loop walks them ~80
// times per frame: build + trial per iteration, 4 x 10 iterations; every walk was a round of dependent global loads), all sums
// are wave-level DPP trees (no LDS, no barrier), and 1024 frames are one wave per SIMD instead of two rounds of 4-wave workgroups.
template <bool GENERAL, int NT, int KR>
__device__ __forceinline__ void po_body(const PoArgs &A, double (*red)[PO_NRED], double *stage = nullptr, int stage_cap = 0, double *red2 = nullptr, double *tot = nullptr)
{
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = A.n[f];
    const double *Xw = A.Xw + (size_t)f * A.max_edges * 3, *obs = A.obs + (size_t)f * A.max_edges * 3;
    const double *is2 = A.inv_s2 + (size_t)f * A.max_edges;
    uint8_t *outl = A.outlier + (size_t)f * A.max_edges;
    for (int e = tid; e < n; e += NT) outl[e] = 0;                       // Optimizer.cc:896
    if (n < 3 || n > A.max_edges) {                                             // Optimizer.cc:1040-1041
        if (tid == 0) { A.n_inliers[f] = 0; if (A.stats) { for (int k = 0; k < 4; k++) A.stats[4 * f + k] = 0; } }
        return;
    }
    BaGraphDev cam; cam.fx = A.fx; cam.fy = A.fy; cam.cx = A.cx; cam.cy = A.cy; cam.bf = A.bf;
    cam.cam_model = A.cam_model; for (int k = 0; k < 4; k++) cam.kb[k] = A.kb[k];
    for (int k = 0; k < 7; k++) cam.Trl[k] = A.Trl[k];
    cam.fx2 = A.fx2; cam.fy2 = A.fy2; cam.cx2 = A.cx2; cam.cy2 = A.cy2; cam.cam2_model = A.cam2_model; for (int k = 0; k < 4; k++) cam.kb2[k] = A.kb2[k];
    const uint8_t *right = (GENERAL && A.right) ? A.right + (size_t)f * A.max_edges : nullptr;
    if (!GENERAL) cam.cam_model = 0;
    double pose0[7], pose[7], pose_ev[7], x[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 7; k++) pose0[k] = A.pose[7 * f + k];
    quat_norm_rot(pose0);                                                       // SE3Quat ctor
    for (int k = 0; k < 7; k++) { pose[k] = pose0[k]; pose_ev[k] = pose0[k]; }
    const double delta_m = (double)(float)sqrt(5.991), dsqr_m = (double)(float)(delta_m * delta_m);   // Optimizer.cc:887-888
    const double delta_s = (double)(float)sqrt(7.815), dsqr_s = (double)(float)(delta_s * delta_s);
    uint32_t level = 0;                                                          // bit k: edge tid + 256*k is an outlier
    double rX[KR ? KR : 1][3], rO[KR ? KR : 1][3], rW[KR ? KR : 1]; int rR[KR ? KR : 1];
#pragma unroll
    for (int k = 0; k < KR; k++) {
        const int e = min(tid + NT * k, n - 1);
        rX[k][0] = Xw[3 * e]; rX[k][1] = Xw[3 * e + 1]; rX[k][2] = Xw[3 * e + 2];
        rO[k][0] = obs[3 * e]; rO[k][1] = obs[3 * e + 1]; rO[k][2] = obs[3 * e + 2];
        rW[k] = is2[e]; rR[k] = (GENERAL && right) ? right[e] : 0;
    }
    // the multi-wave form keeps the first stage_cap edges in LDS (7 doubles each): the ~80 walks over the edges of a frame are then LDS
    // reads instead of rounds of dependent L2 reads -- what a single frame's latency is made of
    const int stage_lo = NT * KR;                                               // edges below live in registers
    const int n_staged = stage ? min(n, stage_lo + stage_cap) : 0;              // edges [stage_lo, n_staged) live in LDS
    if (stage) {
        for (int e = stage_lo + tid; e < n_staged; e += NT) {
            double *q = stage + 7 * (e - stage_lo);
            q[0] = Xw[3 * e]; q[1] = Xw[3 * e + 1]; q[2] = Xw[3 * e + 2]; q[3] = obs[3 * e]; q[4] = obs[3 * e + 1]; q[5] = obs[3 * e + 2]; q[6] = is2[e];
        }
        __syncthreads();
    }
    // body(k, e, X, ob, w0, rt) for every edge of this thread: the KR register-resident ones, then the tail (LDS stage / global memory)
    auto for_tail_edges = [&](auto body) {
        for (int e = tid + NT * KR, k = KR; e < n; e += NT, k++) {
            if (e < n_staged) { const double *q = stage + 7 * (e - stage_lo); body(k, e, q, q + 3, q[6], (GENERAL && right) ? (int)right[e] : 0); }
            else body(k, e, Xw + 3 * e, obs + 3 * e, is2[e], (GENERAL && right) ? (int)right[e] : 0);
        }
    };
    auto for_edges = [&](auto body) {
#pragma unroll
        for (int k = 0; k < KR; k++) { const int e = tid + NT * k; if (e < n) body(k, e, rX[k], rO[k], rW[k], rR[k]); }
        for_tail_edges(body);
    };
    // Frames whose edges are all monocular Pinhole ones (monocular tracking; Tracking.cc:1934) take a STRAIGHT-LINE form of the two walks
    // over the register-resident edges (round 4): one frame is one wave per SIMD, so a walk is a chain of dependent double-precision
    // latencies (quaternion rotation -> two divisions -> Huber's sqrt / division -> the Jacobian's four divisions), and with a branch per
    // edge (outlier level, mono / stereo, Huber's two cases) the four edges of a thread ran one after the other: 4.0 k cycles per trial walk,
    // 4.1 k per build walk (tools/po_prof_probe.py).  Without branches -- masked by selects -- the scheduler interleaves the four chains.
    // Same expressions in the same order as po_edge_chi2 / huber / edge_jacobians' monocular branches: the same bits.
    bool all_mono = false;
    if (!GENERAL && KR > 0) {
        double ns[1] = {0};
        for (int e = tid; e < n; e += NT) ns[0] += !(obs[3 * e + 2] < 0) ? 1.0 : 0.0;
        po_block_sum<1, NT>(ns, red);
        all_mono = ns[0] == 0.0;
    }
    int robust = 1, nbad = 0, lm_trials = 0, lm_iters = 0, rounds = 0;
#ifdef PO_PROF
    long long t_prev = clock64();
#endif
    for (int it = 0; it < 4; it++) {
        for (int k = 0; k < 7; k++) pose[k] = pose0[k];                          // Optimizer.cc:1053
        double cnt[1] = {0};
        for (int e = tid, k = 0; e < n; e += NT, k++) cnt[0] += !((level >> k) & 1u);
        po_block_sum<1, NT>(cnt, red);
        if (cnt[0] > 0) {
            double lambda = 0, ni = 2;
            int nb = 0, ok = 1;
            for (int iter = 0; iter < 10 && ok; iter++) {                        // SparseOptimizer::optimize(10)
                // ---- computeActiveErrors + activeRobustChi2 + buildSystem at the current estimate (LM:69-87)
                double acc[PO_NRED];
#pragma unroll
                for (int k = 0; k < PO_NRED; k++) acc[k] = 0;
                double R[9];
                quat_to_R(pose, R);
                PO_T(7);
                auto build_edge = [&](int k, int e, const double *Xe, const double *ob, double w0, int rt) {
                    (void)e;
                    if ((level >> k) & 1u) return;
                    const int stereo = !(ob[2] < 0);
                    double P[3], er[3], Jx[9], Jt[18], r0, r1;
#pragma unroll
                    for (int k = 12; k < 18; k++) Jt[k] = 0;                     // monocular edge: third row empty (er[2] == 0)
                    const double chi2 = po_edge_chi2<GENERAL>(A, cam, pose, Xe, ob, w0, rt, P, er);
                    if (robust) huber(chi2, stereo ? delta_s : delta_m, stereo ? dsqr_s : dsqr_m, &r0, &r1);
                    else { r0 = chi2; r1 = 1.; }
                    if (GENERAL && rt) tobody_jacobians(cam, pose, Xe, Jx, Jt);   // OptimizableTypes.cpp:82-106 (the pose block of the binary edge)
                    else edge_jacobians(cam, P, R, stereo, Jx, Jt);
                    const double w = r1 * w0;
                    acc[0] += r0;
                    int h = 1;
#pragma unroll
                    for (int a = 0; a < 6; a++) {
#pragma unroll
                        for (int c = a; c < 6; c++) {
                            double s = 0;
#pragma unroll
                            for (int d = 0; d < 3; d++) s += Jt[6 * d + a] * w * Jt[6 * d + c];
                            acc[h++] += s;
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 6; a++) {
                        double s = 0;
#pragma unroll
                        for (int d = 0; d < 3; d++) s += Jt[6 * d + a] * (-w * er[d]);
                        acc[22 + a] += s;
                    }
                };
                if (!GENERAL && KR > 0 && all_mono) {
#pragma unroll
                    for (int k = 0; k < KR; k++) {
                        const bool act = (tid + NT * k < n) && !((level >> k) & 1u);
                        double P[3], er[2], Jt[12];
                        quat_rot(pose, rX[k], P);
                        P[0] += pose[4]; P[1] += pose[5]; P[2] += pose[6];
                        er[0] = rO[k][0] - (A.fx * P[0] / P[2] + A.cx);                // po_edge_chi2, monocular Pinhole branch
                        er[1] = rO[k][1] - (A.fy * P[1] / P[2] + A.cy);
                        const double chi2 = (er[0] * er[0] + er[1] * er[1] + 0.0 * 0.0) * rW[k];
                        const double sq = sqrt(chi2);
                        const bool quad = !robust || chi2 <= dsqr_m;                   // huber(), selects instead of its two cases
                        const double r0 = quad ? chi2 : 2 * sq * delta_m - dsqr_m, r1 = quad ? 1. : delta_m / sq;
                        {                                                              // edge_jacobians, monocular Pinhole branch
                            const double x = P[0], y = P[1], z = P[2];
                            const double p00 = -(A.fx / z), p02 = A.fx * x / (z * z), p11 = -(A.fy / z), p12 = A.fy * y / (z * z);
                            Jt[0] = p02 * y;            Jt[1] = p00 * z - p02 * x;  Jt[2] = -p00 * y;  Jt[3] = p00; Jt[4] = 0;   Jt[5] = p02;
                            Jt[6] = -p11 * z + p12 * y; Jt[7] = -p12 * x;           Jt[8] = p11 * x;   Jt[9] = 0;   Jt[10] = p11; Jt[11] = p12;
                        }
                        const double w = r1 * rW[k];
                        acc[0] += act ? r0 : 0.0;
                        int h = 1;
#pragma unroll
                        for (int a = 0; a < 6; a++) {
#pragma unroll
                            for (int c = a; c < 6; c++) {
                                double sacc = 0;
                                sacc += Jt[a] * w * Jt[c];
                                sacc += Jt[6 + a] * w * Jt[6 + c];
                                acc[h++] += act ? sacc : 0.0;
                            }
                        }
#pragma unroll
                        for (int a = 0; a < 6; a++) {
                            double sacc = 0;
                            sacc += Jt[a] * (-w * er[0]);
                            sacc += Jt[6 + a] * (-w * er[1]);
                            acc[22 + a] += act ? sacc : 0.0;
                        }
                    }
                    for_tail_edges(build_edge);
                } else for_edges(build_edge);
                PO_T(0);
                if (NT == 256 && red2) po_block_sum_lds<PO_NRED>(acc, red2, tot);
                else po_block_sum<PO_NRED, NT>(acc, red);
                PO_T(1);
                for (int k = 0; k < 7; k++) pose_ev[k] = pose[k];
                double current_chi = acc[0];
                const double ini_chi = current_chi;
                // H (symmetric) stays packed as its 21 upper-triangle entries, row-major: Hp[PO_IDX(a, c)], a <= c
                double Hp[21], b[6];
#pragma unroll
                for (int k = 0; k < 21; k++) Hp[k] = acc[1 + k];
#pragma unroll
                for (int a = 0; a < 6; a++) b[a] = acc[22 + a];
                if (iter == 0) {                                                 // computeLambdaInit, LM:171-185 (_tau = 1e-50)
                    double md = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++) md = fmax(fabs(Hp[PO_IDX(a, a)]), md);
                    lambda = 1e-50 * md; ni = 2; nb = 0;
                }
                double rho = 0;
                int qmax = 0;
                do {
                    double pose_bk[7];
                    for (int k = 0; k < 7; k++) pose_bk[k] = pose[k];            // push
                    // LinearSolverDense: LDL^T of H + lambda I; a non-positive pivot fails the solve (x keeps its old value)
                    // Lp[PO_IDX(j, i)] (j <= i) holds L(i, j) below the diagonal and D(j) on it
                    double Lp[21];
#pragma unroll
                    for (int k = 0; k < 21; k++) Lp[k] = Hp[k];
#pragma unroll
                    for (int a = 0; a < 6; a++) Lp[PO_IDX(a, a)] += lambda;
                    bool ok2 = true;
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        double d = Lp[PO_IDX(j, j)];
#pragma unroll
                        for (int k = 0; k < j; k++) d -= Lp[PO_IDX(k, j)] * Lp[PO_IDX(k, j)] * Lp[PO_IDX(k, k)];
                        ok2 = ok2 && (d > 0.0) && isfinite(d);
                        Lp[PO_IDX(j, j)] = d;
#pragma unroll
                        for (int i = j + 1; i < 6; i++) {
                            double sacc = Lp[PO_IDX(j, i)];
#pragma unroll
                            for (int k = 0; k < j; k++) sacc -= Lp[PO_IDX(k, i)] * Lp[PO_IDX(k, j)] * Lp[PO_IDX(k, k)];
                            Lp[PO_IDX(j, i)] = sacc / d;
                        }
                    }
                    if (ok2) {
                        double y[6];
#pragma unroll
                        for (int i = 0; i < 6; i++) {
                            double sacc = b[i];
#pragma unroll
                            for (int k = 0; k < i; k++) sacc -= Lp[PO_IDX(k, i)] * y[k];
                            y[i] = sacc;
                        }
#pragma unroll
                        for (int i = 0; i < 6; i++) y[i] /= Lp[PO_IDX(i, i)];
#pragma unroll
                        for (int i = 5; i >= 0; i--) {
                            double sacc = y[i];
#pragma unroll
                            for (int k = i + 1; k < 6; k++) sacc -= Lp[PO_IDX(i, k)] * y[k];
                            y[i] = sacc;
                        }
#pragma unroll
                        for (int i = 0; i < 6; i++) x[i] = y[i];
                    }
                    double pn[7];
                    se3_oplus(x, pose, pn);                                      // update, SO:422-435
                    for (int k = 0; k < 7; k++) pose[k] = pn[k];
                    PO_T(2);
                    double tc[1] = {0};                                          // computeActiveErrors + activeRobustChi2 at the trial
                    auto trial_edge = [&](int k, int e, const double *Xe, const double *ob, double w0, int rt) {
                        (void)e;
                        if ((level >> k) & 1u) return;
                        const int stereo = !(ob[2] < 0);
                        double P[3], er[3], r0, r1;
                        const double chi2 = po_edge_chi2<GENERAL>(A, cam, pose, Xe, ob, w0, rt, P, er);
                        if (robust) huber(chi2, stereo ? delta_s : delta_m, stereo ? dsqr_s : dsqr_m, &r0, &r1);
                        else r0 = chi2;
                        tc[0] += r0;
                    };
                    if (!GENERAL && KR > 0 && all_mono) {
                        double c2[KR ? KR : 1];
                        bool over = false;
#pragma unroll
                        for (int k = 0; k < KR; k++) {
                            const bool act = (tid + NT * k < n) && !((level >> k) & 1u);
                            double P[3];
                            quat_rot(pose, rX[k], P);
                            P[0] += pose[4]; P[1] += pose[5]; P[2] += pose[6];
                            const double e0 = rO[k][0] - (A.fx * P[0] / P[2] + A.cx), e1 = rO[k][1] - (A.fy * P[1] / P[2] + A.cy);
                            c2[k] = (e0 * e0 + e1 * e1 + 0.0 * 0.0) * rW[k];
                            over = over || (act && c2[k] > dsqr_m);
                        }
                        // Huber's square root only where some lane of the wave needs it (after the first round the gross outliers are
                        // at level 1 and almost every trial of the remaining rounds is quadratic throughout): a wave-uniform branch
                        if (robust && __builtin_amdgcn_ballot_w64(over) != 0) {
#pragma unroll
                            for (int k = 0; k < KR; k++) {
                                const bool act = (tid + NT * k < n) && !((level >> k) & 1u);
                                const double r0 = c2[k] <= dsqr_m ? c2[k] : 2 * sqrt(c2[k]) * delta_m - dsqr_m;
                                tc[0] += act ? r0 : 0.0;
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < KR; k++) tc[0] += ((tid + NT * k < n) && !((level >> k) & 1u)) ? c2[k] : 0.0;
                        }
                        for_tail_edges(trial_edge);
                    } else for_edges(trial_edge);
                    PO_T(3);
                    // (the SAME reduction as the build step's chi2: a trial that does not move the pose must reproduce current_chi bit for
                    // bit -- rho == 0 is how g2o's loop ends at convergence, LM:151-152; with two different summation orders it never did
                    // and every converged iteration burnt trials until lambda overflowed the step: 58 -> 87 trials on the probe frame)
                    if (NT == 256 && red2) po_block_sum_lds<1>(tc, red2, tot);
                    else po_block_sum<1, NT>(tc, red);
                    PO_T(4);
                    for (int k = 0; k < 7; k++) pose_ev[k] = pose[k];
                    double temp_chi = ok2 ? tc[0] : DBL_MAX;
                    rho = current_chi - temp_chi;
                    double scale = 0;                                            // computeScale, LM:187-194
                    for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                    scale += 1e-3;
                    rho /= scale;
                    if (rho > 0 && isfinite(temp_chi)) {
                        const double t3 = 2 * rho - 1;
                        double alpha = 1. - t3 * t3 * t3;
                        alpha = fmin(alpha, 2. / 3.);
                        lambda *= fmax(1. / 3., alpha); ni = 2; current_chi = temp_chi;
                    } else {
                        lambda *= ni; ni *= 2;
                        for (int k = 0; k < 7; k++) pose[k] = pose_bk[k];        // pop
                    }
                    qmax++; lm_trials++;
#ifdef PO_PROF
                    if (blockIdx.x == 0 && threadIdx.x == 0) g_po_prof[6]++;
#endif
                } while (rho < 0 && qmax < 100);
                lm_iters++;
                if (qmax == 100 || rho == 0) ok = 0;                             // LM:151-152
                else {
                    if ((ini_chi - current_chi) * 1e3 < ini_chi) nb++; else nb = 0;   // LM:157-166
                    if (nb >= 3) ok = 0;
                }
            }
        }
        // ---- re-classification (Optimizer.cc:1058-1145): inliers keep the error of the last evaluation (possibly a
        // rejected trial), outliers are re-evaluated at the current estimate; the comparison is in float
        double bad[1] = {0};
        for_edges([&](int k, int e, const double *Xe, const double *ob, double w0, int rt) {
            (void)e;
            double P[3], er[3];
            const double chi2d = po_edge_chi2<GENERAL>(A, cam, ((level >> k) & 1u) ? pose : pose_ev, Xe, ob, w0, rt, P, er);
            const float chi2 = (float)chi2d;
            const float gate = ob[2] < 0 ? 5.991f : 7.815f;
            if (chi2 > gate) { level |= 1u << k; bad[0] += 1; } else level &= ~(1u << k);
        });
        po_block_sum<1, NT>(bad, red);
        PO_T(5);
        nbad = (int)bad[0];
        if (it == 2) robust = 0;                                                 // setRobustKernel(0)
        rounds++;
        if (n < 10) break;                                                       // Optimizer.cc:1147-1148
    }
    for (int e = tid, k = 0; e < n; e += NT, k++) outl[e] = (uint8_t)((level >> k) & 1u);
    if (tid == 0) {
        for (int k = 0; k < 7; k++) A.pose[7 * f + k] = pose[k];
        A.n_inliers[f] = n - nbad;
        if (A.stats) { A.stats[4 * f] = rounds; A.stats[4 * f + 1] = lm_iters; A.stats[4 * f + 2] = lm_trials; A.stats[4 * f + 3] = nbad; }
    }
}

template <bool GENERAL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pose_opt(PoArgs A)
{
    __shared__ double red[4][PO_NRED];
    __shared__ double red2[PO_NRED * PO_RED2_K], tot[32];
    extern __shared__ double po_stage[];                         // [stage_cap][7] (dynamic: 0 when the launch is a big batch)
    po_body<GENERAL, 256, 0>(A, red, A.stage_cap > 0 ? po_stage : nullptr, A.stage_cap, red2, tot);
}
// Latency form (round 4): what Optimizer::PoseOptimization(Frame*) launches -- ONE frame, i.e. four waves on the whole
// chip.  k_pose_opt above is held at two waves per SIMD (256 VGPRs) for launches that fill the device and spilled there (151 VGPRs /
// 472 B of scratch in the Pinhole instantiation); a launch of up to 256 frames is at most one workgroup per CU = one wave per SIMD, so
// this instantiation takes the whole register file (512 VGPRs), keeps the first 4 edges of every thread (1024 per frame) in registers
// instead of LDS and spills nothing; edges beyond 1024 are staged in LDS as before.
template <bool GENERAL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_pose_opt_lat(PoArgs A)
{
    __shared__ double red[4][PO_NRED];
    __shared__ double red2[PO_NRED * PO_RED2_K], tot[32];
    extern __shared__ double po_stage[];                         // [stage_cap][7]: edges 1024 .. of a frame (index = edge - 1024)
    po_body<GENERAL, 256, PO_KR>(A, red, A.stage_cap > 0 ? po_stage : nullptr, A.stage_cap, red2, tot);
}
template <bool GENERAL>
__global__ __launch_bounds__(64) void k_pose_opt_wave(PoArgs A)
{
    po_body<GENERAL, 64, GENERAL ? 8 : 16>(A, nullptr);          // (the fisheye / second-camera edge code needs the registers: 512 edges resident)
}

extern "C" int orbhip_pose_optimization_device(orbhip_ctx *ctx, const double *d_Xw, const double *d_obs,
                                               const double *d_inv_sigma2, const int32_t *d_n_edges, int frames, int max_edges,
                                               double fx, double fy, double cx, double cy, double bf, const double *kb8_k,
                                               const orbhip_camera2 *cam2, const uint8_t *d_right,
                                               double *d_pose, uint8_t *d_outlier, int32_t *d_n_inliers, int32_t *d_stats)
{
    if (!ctx || !d_Xw || !d_obs || !d_inv_sigma2 || !d_n_edges || frames <= 0 || max_edges <= 0 || max_edges > 8192 ||
        !d_pose || !d_outlier || !d_n_inliers) { g_ba_error = "bad argument"; return ORBHIP_E_BADARG; }
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    PoArgs A;
    A.Xw = d_Xw; A.obs = d_obs; A.inv_s2 = d_inv_sigma2; A.n = d_n_edges; A.max_edges = max_edges;
    A.fx = fx; A.fy = fy; A.cx = cx; A.cy = cy; A.bf = bf; A.pose = d_pose; A.outlier = d_outlier; A.n_inliers = d_n_inliers;
    A.cam_model = kb8_k ? 1 : 0; for (int k = 0; k < 4; k++) A.kb[k] = kb8_k ? kb8_k[k] : 0.0;
    if ((d_right != nullptr) != (cam2 != nullptr)) { g_ba_error = "cam2 and d_right go together"; return ORBHIP_E_BADARG; }
    A.right = d_right;
    for (int k = 0; k < 7; k++) A.Trl[k] = cam2 ? cam2->Trl[k] : (k == 3 ? 1.0 : 0.0);
    A.fx2 = cam2 ? cam2->fx : 0; A.fy2 = cam2 ? cam2->fy : 0; A.cx2 = cam2 ? cam2->cx : 0; A.cy2 = cam2 ? cam2->cy : 0;
    A.cam2_model = cam2 ? cam2->camera_model : 0; for (int k = 0; k < 4; k++) A.kb2[k] = cam2 ? cam2->kb[k] : 0.0;
    A.stats = d_stats; A.stage_cap = 0;
    const bool general = A.cam_model || A.right;
    // one wave per frame is the throughput form (1024 frames: 1.26 ms vs 2.0 ms); four waves per frame have the shorter latency while
    // the frames fit one round of workgroups (1 frame: 0.36 ms vs 0.71 ms, 64 frames: 0.64 vs 1.12 ms)
    const int wave_min_frames = getenv("ORBHIP_POSE_WAVE_MIN_FRAMES") ? atoi(getenv("ORBHIP_POSE_WAVE_MIN_FRAMES")) : 513;
    if (max_edges <= 2048 && frames >= wave_min_frames) {                // outlier bits: 32 per lane
        if (general) hipLaunchKernelGGL(k_pose_opt_wave<true>, dim3(frames), dim3(64), 0, orbhip_ctx_stream_internal(ctx), A);
        else hipLaunchKernelGGL(k_pose_opt_wave<false>, dim3(frames), dim3(64), 0, orbhip_ctx_stream_internal(ctx), A);
    } else {
        // up to one workgroup per CU the edges are staged in LDS (latency form); bigger launches keep the LDS-free one (occupancy)
        static const int stage_env = getenv("ORBHIP_POSE_STAGE_EDGES") ? atoi(getenv("ORBHIP_POSE_STAGE_EDGES")) : 2048;
        static const int lat_env = getenv("ORBHIP_POSE_LAT") ? atoi(getenv("ORBHIP_POSE_LAT")) : 1;       // 0: round 3's kernel for every launch (A/B)
        const bool lat = frames <= 256 && lat_env;               // at most one workgroup per CU: one wave per SIMD, the whole register file
        A.stage_cap = frames <= 256 ? std::max(0, std::min(max_edges, stage_env) - (lat ? PO_THREADS * PO_KR : 0)) : 0;
        const size_t lds = (size_t)A.stage_cap * 7 * sizeof(double);
        const void *fn = lat ? (general ? reinterpret_cast<const void *>(k_pose_opt_lat<true>) : reinterpret_cast<const void *>(k_pose_opt_lat<false>))
                             : (general ? reinterpret_cast<const void *>(k_pose_opt<true>) : reinterpret_cast<const void *>(k_pose_opt<false>));
        if (lds > 0 && orb_lds_optin(fn, orbhip_ctx_device_internal(ctx), lds)) { g_ba_error = "LDS opt-in (k_pose_opt)"; return ORBHIP_E_HIP; }
        if (lat) {
            if (general) hipLaunchKernelGGL(k_pose_opt_lat<true>, dim3(frames), dim3(PO_THREADS), lds, orbhip_ctx_stream_internal(ctx), A);
            else hipLaunchKernelGGL(k_pose_opt_lat<false>, dim3(frames), dim3(PO_THREADS), lds, orbhip_ctx_stream_internal(ctx), A);
        } else if (general) hipLaunchKernelGGL(k_pose_opt<true>, dim3(frames), dim3(PO_THREADS), lds, orbhip_ctx_stream_internal(ctx), A);
        else hipLaunchKernelGGL(k_pose_opt<false>, dim3(frames), dim3(PO_THREADS), lds, orbhip_ctx_stream_internal(ctx), A);
    }
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}
