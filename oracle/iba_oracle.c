/* iba_oracle.c -- see iba_oracle.h (TEST INFRASTRUCTURE ONLY; parity unpinned).
 * Reference line numbers: "LIBA" = src/Optimizer.cc, "G2T" = src/G2oTypes.cc, "G2H" = include/G2oTypes.h,
 * "IMU" = src/ImuTypes.cc, "LM" = Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp,
 * "BS" = Thirdparty/g2o/g2o/core/block_solver.hpp, "BME" = Thirdparty/g2o/g2o/core/base_multi_edge.hpp. */
#include "iba_oracle.h"
#include "ba_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { K_R = 0, K_T = 9, K_V = 12, K_BG = 15, K_BA = 18 };
enum { P_DT = 0, P_DR = 1, P_DV = 10, P_DP = 13, P_JRG = 16, P_JVG = 25, P_JVA = 34, P_JPG = 43, P_JPA = 52, P_BG = 61, P_BA = 64 };

void orc_iba_default_params(orc_iba_params *p, int large)
{
    p->iterations = large ? 4 : 10;              /* LIBA:4579-4585 */
    p->lambda_init = large ? 1e-2 : 1.0;         /* LIBA:4699-4710 */
    p->large = large;
    p->max_trials = 100;                         /* LM:50 */
}

/* ------------------------------------------------------------------ 3x3 helpers (row-major) */
static void mm(const double *A, const double *B, double *C)
{
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, t, sizeof(t));
}
static void mtm(const double *A, const double *B, double *C)      /* A^T B */
{
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
    memcpy(C, t, sizeof(t));
}
static void mv(const double *A, const double *v, double *o)
{
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void mtv(const double *A, const double *v, double *o)      /* A^T v */
{
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void skew(const double *w, double *W)      /* G2T:1063-1068 */
{
    W[0] = 0; W[1] = -w[2]; W[2] = w[1]; W[3] = w[2]; W[4] = 0; W[5] = -w[0]; W[6] = -w[1]; W[7] = w[0]; W[8] = 0;
}
static void inv3g(const double *A, double *I)
{
    const double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
    const double id = 1.0 / (A[0] * c0 + A[1] * c1 + A[2] * c2);
    I[0] = c0 * id; I[1] = (A[2] * A[7] - A[1] * A[8]) * id; I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    I[3] = c1 * id; I[4] = (A[0] * A[8] - A[2] * A[6]) * id; I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    I[6] = c2 * id; I[7] = (A[1] * A[6] - A[0] * A[7]) * id; I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}
/* nearest rotation U V^T of a near-orthonormal matrix (IMU:30-36 takes it from a float SVD): two Newton steps of the polar
 * iteration R <- (R + R^-T) / 2, which converges quadratically to the same factor */
static void normalize_rotation(double *R)
{
    for (int it = 0; it < 2; it++) {
        double I[9];
        inv3g(R, I);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = 0.5 * (R[3 * i + j] + I[3 * j + i]);
    }
}
/* I + a W + b W W */
static void rodrigues(const double *w, double a, double b, double *R)
{
    double W[9], W2[9];
    skew(w, W); mm(W, W, W2);
    for (int i = 0; i < 9; i++) R[i] = a * W[i] + b * W2[i];
    R[0] += 1; R[4] += 1; R[8] += 1;
}
void orc_iba_exp_so3(const double w[3], double R[9])     /* G2T:991-1008 (the Eigen overload the vertices use) */
{
    const double d2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], d = sqrt(d2);
    if (d < 1e-5) rodrigues(w, 1.0, 0.5, R);
    else rodrigues(w, sin(d) / d, (1.0 - cos(d)) / d2, R);
    normalize_rotation(R);
}
static void exp_so3_imu(const double w[3], double R[9])   /* IMU:48-60: eps = 1e-4, no normalisation */
{
    const double d2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], d = sqrt(d2);
    if (d < 1e-4) rodrigues(w, 1.0, 0.5, R);
    else rodrigues(w, sin(d) / d, (1.0 - cos(d)) / d2, R);
}
void orc_iba_log_so3(const double R[9], double w[3])     /* G2T:1010-1025 */
{
    const double tr = R[0] + R[4] + R[8];
    w[0] = (R[7] - R[5]) / 2; w[1] = (R[2] - R[6]) / 2; w[2] = (R[3] - R[1]) / 2;
    const double costheta = (tr - 1.0) * 0.5;
    if (costheta > 1 || costheta < -1) return;
    const double theta = acos(costheta), s = sin(theta);
    if (fabs(s) < 1e-5) return;
    for (int i = 0; i < 3; i++) w[i] = theta * w[i] / s;
}
static void inv_right_jac(const double *v, double *J)    /* G2T:1032-1044 */
{
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrt(d2);
    if (d < 1e-5) { memset(J, 0, 72); J[0] = J[4] = J[8] = 1; return; }
    rodrigues(v, 0.5, 1.0 / d2 - (1.0 + cos(d)) / (2.0 * d * sin(d)), J);
}
static void right_jac(const double *v, double *J)        /* G2T:1046-1061 */
{
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrt(d2);
    if (d < 1e-5) { memset(J, 0, 72); J[0] = J[4] = J[8] = 1; return; }
    rodrigues(v, -(1.0 - cos(d)) / d2, (d - sin(d)) / (d2 * d), J);
}

/* ------------------------------------------------------------------ vertices */
void orc_iba_kf_update(double *s, const double *dx, int imu)
{
    /* ImuCamPose::Update, G2T:192-220: twb += Rwb ut; Rwb = Rwb ExpSO3(ur)  (the camera poses follow in cam_pose()) */
    double t[3], E[9];
    mv(s + K_R, dx + 3, t);
    for (int i = 0; i < 3; i++) s[K_T + i] += t[i];
    orc_iba_exp_so3(dx, E);
    mm(s + K_R, E, s + K_R);
    if (imu) for (int i = 0; i < 9; i++) s[K_V + i] += dx[6 + i];       /* G2H:202-206 and the two bias vertices */
}

/* camera cam_idx of the rig: Rcb, tcb (G2T:49-52, 57-67) and its intrinsics */
struct camview { double Rcb[9], tcb[3], fx, fy, cx, cy; int model; const double *kb; };
static void cam_view(const orc_iba_problem *g, int cam_idx, struct camview *c)
{
    if (!cam_idx) {
        memcpy(c->Rcb, g->Rcb, sizeof(c->Rcb)); memcpy(c->tcb, g->tcb, sizeof(c->tcb));
        c->fx = g->fx; c->fy = g->fy; c->cx = g->cx; c->cy = g->cy; c->model = g->camera_model; c->kb = g->kb;
    } else {
        double Rrl[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rrl[3 * i + j] = g->Trl[4 * i + j];
        mm(Rrl, g->Rcb, c->Rcb);
        mv(Rrl, g->tcb, c->tcb);
        for (int i = 0; i < 3; i++) c->tcb[i] += g->Trl[4 * i + 3];
        c->fx = g->fx2; c->fy = g->fy2; c->cx = g->cx2; c->cy = g->cy2; c->model = g->camera2_model; c->kb = g->kb2;
    }
}
/* Rcw = Rcb Rbw, tcw = Rcb tbw + tcb (G2T:212-219) */
static void cam_pose_of(const struct camview *c, const double *s, double *Rcw, double *tcw)
{
    double Rbw[9], tbw[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rbw[3 * i + j] = s[K_R + 3 * j + i];
    mv(Rbw, s + K_T, tbw);
    for (int i = 0; i < 3; i++) tbw[i] = -tbw[i];
    mm(c->Rcb, Rbw, Rcw);
    mv(c->Rcb, tbw, tcw);
    for (int i = 0; i < 3; i++) tcw[i] += c->tcb[i];
}
static void cam_pose(const orc_iba_problem *g, const double *s, double *Rcw, double *tcw)      /* left camera */
{
    struct camview c;
    cam_view(g, 0, &c);
    cam_pose_of(&c, s, Rcw, tcw);
}

/* ------------------------------------------------------------------ edges (type: 0 EdgeMono(0), 1 EdgeStereo(0), 2 EdgeMono(1)) */
static void visual_error(const orc_iba_problem *g, const double *s, const double X[3], const double obs[3], int type, double err[3], double Xc[3], double Rcw[9])
{
    struct camview c;
    double tcw[3], uv[2];
    cam_view(g, type == 2, &c);
    cam_pose_of(&c, s, Rcw, tcw);
    mv(Rcw, X, Xc);
    for (int i = 0; i < 3; i++) Xc[i] += tcw[i];
    if (c.model == 1) orc_kb8_project(Xc, c.fx, c.fy, c.cx, c.cy, c.kb, uv);
    else { uv[0] = c.fx * Xc[0] / Xc[2] + c.cx; uv[1] = c.fy * Xc[1] / Xc[2] + c.cy; }      /* Pinhole::project */
    err[0] = obs[0] - uv[0]; err[1] = obs[1] - uv[1];                                        /* G2H:350-355 */
    err[2] = type == 1 ? obs[2] - (uv[0] - g->bf * (1 / Xc[2])) : 0.0;                       /* G2T:177-185 */
}

void orc_iba_edge_visual(const orc_iba_problem *g, const double *s, const double X[3], const double obs[3], int type,
                         double err[3], double Jx[9], double Jp[18])
{
    struct camview c;
    double Xc[3], Rcw[9];
    cam_view(g, type == 2, &c);
    visual_error(g, s, X, obs, type, err, Xc, Rcw);
    /* G2T:349-373 / :397-423 */
    double pj[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (c.model == 1) orc_kb8_project_jac(Xc, c.fx, c.fy, c.kb, pj);
    else { pj[0] = c.fx / Xc[2]; pj[2] = -c.fx * Xc[0] / (Xc[2] * Xc[2]); pj[4] = c.fy / Xc[2]; pj[5] = -c.fy * Xc[1] / (Xc[2] * Xc[2]); }
    const int stereo = type == 1, D = stereo ? 3 : 2;
    if (stereo) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + g->bf * (1.0 / (Xc[2] * Xc[2])); }
    memset(Jx, 0, 72); memset(Jp, 0, 144);
    for (int d = 0; d < D; d++) for (int j = 0; j < 3; j++) Jx[3 * d + j] = -(pj[3 * d] * Rcw[j] + pj[3 * d + 1] * Rcw[3 + j] + pj[3 * d + 2] * Rcw[6 + j]);
    double Xb[3], d0[3] = {Xc[0] - c.tcb[0], Xc[1] - c.tcb[1], Xc[2] - c.tcb[2]};
    mtv(c.Rcb, d0, Xb);                      /* Xb = Rbc Xc + tbc with Tbc = Tcb^-1 */
    const double SE3[18] = {0, Xb[2], -Xb[1], 1, 0, 0, -Xb[2], 0, Xb[0], 0, 1, 0, Xb[1], -Xb[0], 0, 0, 0, 1};
    double PR[9];
    mm(pj, c.Rcb, PR);
    for (int d = 0; d < D; d++) for (int j = 0; j < 6; j++) Jp[6 * d + j] = PR[3 * d] * SE3[j] + PR[3 * d + 1] * SE3[6 + j] + PR[3 * d + 2] * SE3[12 + j];
}

/* IMU:357-378 with b_ = (bg, ba) of vertex 1 */
static void preint_deltas(const double *pi, const double *bg, const double *ba, double dR[9], double dV[3], double dP[3], double dbg[3])
{
    double dba[3], w[3], E[9], t[3];
    for (int i = 0; i < 3; i++) { dbg[i] = bg[i] - pi[P_BG + i]; dba[i] = ba[i] - pi[P_BA + i]; }
    mv(pi + P_JRG, dbg, w);
    exp_so3_imu(w, E);
    mm(pi + P_DR, E, dR);
    normalize_rotation(dR);
    mv(pi + P_JVG, dbg, dV); mv(pi + P_JVA, dba, t);
    for (int i = 0; i < 3; i++) dV[i] = pi[P_DV + i] + dV[i] + t[i];
    mv(pi + P_JPG, dbg, dP); mv(pi + P_JPA, dba, t);
    for (int i = 0; i < 3; i++) dP[i] = pi[P_DP + i] + dP[i] + t[i];
}


void orc_iba_edge_inertial(const double *s1, const double *s2, const double *pi, double err[9], double J[216])
{
    const double dt = pi[P_DT];
    const double g[3] = {0, 0, -(double)9.81f};          /* g << 0, 0, -IMU::GRAVITY_VALUE (G2T:700; a float constant, ImuTypes.h:40) */
    double dR[9], dV[3], dP[3], dbg[3];
    preint_deltas(pi, s1 + K_BG, s1 + K_BA, dR, dV, dP, dbg);
    /* computeError, G2T:720-740 */
    double R12[9], eR[9], er[3];
    mtm(s1 + K_R, s2 + K_R, R12);                 /* Rbw1 Rwb2 */
    mtm(dR, R12, eR);
    orc_iba_log_so3(eR, er);
    double a[3], b[3], va[3], vb[3];
    for (int i = 0; i < 3; i++) {
        a[i] = s2[K_V + i] - s1[K_V + i] - g[i] * dt;
        b[i] = s2[K_T + i] - s1[K_T + i] - s1[K_V + i] * dt - g[i] * dt * dt / 2;
    }
    mtv(s1 + K_R, a, va); mtv(s1 + K_R, b, vb);
    for (int i = 0; i < 3; i++) { err[i] = er[i]; err[3 + i] = va[i] - dV[i]; err[6 + i] = vb[i] - dP[i]; }
    if (!J) return;
    /* linearizeOplus, G2T:742-800 */
    memset(J, 0, sizeof(double) * 216);
    double invJr[9], T[9], S[9];
    inv_right_jac(er, invJr);
#define SETB(r0, c0, M, sgn) for (int i_ = 0; i_ < 3; i_++) for (int j_ = 0; j_ < 3; j_++) J[24 * ((r0) + i_) + (c0) + j_] = (sgn) * (M)[3 * i_ + j_]
    mtm(s2 + K_R, s1 + K_R, T); mm(invJr, T, T);  SETB(0, 0, T, -1.0);           /* -invJr Rwb2^T Rwb1 */
    skew(va, S); SETB(3, 0, S, 1.0);
    skew(vb, S); SETB(6, 0, S, 1.0);
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    SETB(6, 3, I3, -1.0);
    double Rbw1[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rbw1[3 * i + j] = s1[K_R + 3 * j + i];
    SETB(3, 6, Rbw1, -1.0);
    SETB(6, 6, Rbw1, -dt);
    double w[3], RJ[9];
    mv(pi + P_JRG, dbg, w);
    right_jac(w, RJ);
    double eRt[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) eRt[3 * i + j] = eR[3 * j + i];
    mm(invJr, eRt, T); mm(T, RJ, T); mm(T, pi + P_JRG, T); SETB(0, 9, T, -1.0);
    SETB(3, 9, pi + P_JVG, -1.0);
    SETB(6, 9, pi + P_JPG, -1.0);
    SETB(3, 12, pi + P_JVA, -1.0);
    SETB(6, 12, pi + P_JPA, -1.0);
    SETB(0, 15, invJr, 1.0);
    SETB(6, 18, R12, 1.0);
    SETB(3, 21, Rbw1, 1.0);
#undef SETB
}

/* ------------------------------------------------------------------ the optimiser */
struct iba {
    const orc_iba_problem *g; const orc_iba_params *p;
    int L, E, M, n;
    int *off, *dim;                /* per keyframe: first unknown of its block (-1 fixed), 15 or 6 */
    int *pt_start;
    double *kf, *kf_bk, *pts, *pts_bk;
    double *err, *chi2;            /* visual */
    double *ierr, *ichi2;          /* [M][15] = inertial 9, gyro RW 3, acc RW 3; [M][3] */
    double *H, *b, *Hll, *bl, *W, *Dinv, *S, *bs, *x;
    double lambda, ni; int nbad, lm_trials;
    double delta_m, dsqr_m, delta_s, dsqr_s, delta_i, dsqr_i;
};

static void huber(double e, double delta, double dsqr, double rho[2])       /* robust_kernel_impl.cpp:78-91 */
{
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; }
    else { double s = sqrt(e); rho[0] = 2 * s * delta - dsqr; rho[1] = delta / s; }
}

static double quad(const double *A, const double *e, int n)
{
    double c = 0;
    for (int i = 0; i < n; i++) { double r = 0; for (int j = 0; j < n; j++) r += A[n * i + j] * e[j]; c += e[i] * r; }
    return c;
}

static void compute_errors(struct iba *B)
{
    const orc_iba_problem *g = B->g;
    for (int e = 0; e < B->E; e++) {
        double Xc[3], Rcw[9], *er = B->err + 3 * e;
        visual_error(g, B->kf + ORC_IBA_KF * g->edge_kf[e], B->pts + 3 * g->edge_point[e], g->edge_obs + 3 * e, g->edge_stereo[e], er, Xc, Rcw);
        B->chi2[e] = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * g->edge_inv_sigma2[e];
    }
    for (int m = 0; m < B->M; m++) {
        const double *s1 = B->kf + ORC_IBA_KF * g->in_kf1[m], *s2 = B->kf + ORC_IBA_KF * g->in_kf2[m];
        double *er = B->ierr + 15 * m;
        orc_iba_edge_inertial(s1, s2, g->in_preint + ORC_IBA_PREINT * m, er, NULL);
        for (int i = 0; i < 3; i++) { er[9 + i] = s2[K_BG + i] - s1[K_BG + i]; er[12 + i] = s2[K_BA + i] - s1[K_BA + i]; }   /* G2H:642-646 */
        B->ichi2[3 * m] = quad(g->in_info + 81 * m, er, 9);
        B->ichi2[3 * m + 1] = quad(g->in_info_g + 9 * m, er + 9, 3);
        B->ichi2[3 * m + 2] = quad(g->in_info_a + 9 * m, er + 12, 3);
    }
}

static double robust_chi2(const struct iba *B)
{
    double chi = 0, rho[2];
    for (int e = 0; e < B->E; e++) {
        if (B->g->edge_stereo[e] == 1) huber(B->chi2[e], B->delta_s, B->dsqr_s, rho); else huber(B->chi2[e], B->delta_m, B->dsqr_m, rho);
        chi += rho[0];
    }
    for (int m = 0; m < B->M; m++) {
        if (B->g->in_robust[m]) { huber(B->ichi2[3 * m], B->delta_i, B->dsqr_i, rho); chi += rho[0]; } else chi += B->ichi2[3 * m];
        chi += B->ichi2[3 * m + 1] + B->ichi2[3 * m + 2];
    }
    return chi;
}

/* H += J^T (w Omega) J, b += J^T (-w Omega e) over the unfixed columns (BME:172-215 / base_binary_edge.hpp) */
static void add_quadratic(struct iba *B, const double *J, int ncol, const int *col, const double *Om, const double *e, int D, double w)
{
    const int n = B->n;
    double OJ[9 * 24], Oe[9];
    for (int i = 0; i < D; i++) {
        double r = 0;
        for (int k = 0; k < D; k++) r += Om[D * i + k] * e[k];
        Oe[i] = -w * r;
        for (int c = 0; c < ncol; c++) {
            double s = 0;
            for (int k = 0; k < D; k++) s += w * Om[D * i + k] * J[ncol * k + c];
            OJ[ncol * i + c] = s;
        }
    }
    for (int a = 0; a < ncol; a++) {
        if (col[a] < 0) continue;
        double s = 0;
        for (int k = 0; k < D; k++) s += J[ncol * k + a] * Oe[k];
        B->b[col[a]] += s;
        for (int c = 0; c < ncol; c++) {
            if (col[c] < 0) continue;
            double h = 0;
            for (int k = 0; k < D; k++) h += J[ncol * k + a] * OJ[ncol * k + c];
            B->H[(size_t)col[a] * n + col[c]] += h;
        }
    }
}

static void build_system(struct iba *B)
{
    const orc_iba_problem *g = B->g;
    const int n = B->n;
    memset(B->H, 0, sizeof(double) * (size_t)n * n);
    memset(B->b, 0, sizeof(double) * n);
    memset(B->Hll, 0, sizeof(double) * 9 * B->L);
    memset(B->bl, 0, sizeof(double) * 3 * B->L);
    memset(B->W, 0, sizeof(double) * 18 * B->E);
    for (int e = 0; e < B->E; e++) {
        const int st = g->edge_stereo[e], D = st == 1 ? 3 : 2, k = g->edge_kf[e], li = g->edge_point[e], o = B->off[k];
        double er[3], Jx[9], Jp[18], rho[2];
        orc_iba_edge_visual(g, B->kf + ORC_IBA_KF * k, B->pts + 3 * li, g->edge_obs + 3 * e, st, er, Jx, Jp);
        const double *es = B->err + 3 * e;            /* the stored _error of the last computeActiveErrors (same state) */
        if (st == 1) huber(B->chi2[e], B->delta_s, B->dsqr_s, rho); else huber(B->chi2[e], B->delta_m, B->dsqr_m, rho);
        const double w = rho[1] * g->edge_inv_sigma2[e];
        for (int a = 0; a < 3; a++) {
            double s = 0;
            for (int d = 0; d < D; d++) s += Jx[3 * d + a] * (-w * es[d]);
            B->bl[3 * li + a] += s;
            for (int c = 0; c < 3; c++) {
                double h = 0;
                for (int d = 0; d < D; d++) h += Jx[3 * d + a] * w * Jx[3 * d + c];
                B->Hll[9 * li + 3 * a + c] += h;
            }
        }
        if (o < 0) continue;
        for (int a = 0; a < 6; a++) {
            double s = 0;
            for (int d = 0; d < D; d++) s += Jp[6 * d + a] * (-w * es[d]);
            B->b[o + a] += s;
            for (int c = 0; c < 6; c++) {
                double h = 0;
                for (int d = 0; d < D; d++) h += Jp[6 * d + a] * w * Jp[6 * d + c];
                B->H[(size_t)(o + a) * n + o + c] += h;
            }
            for (int c = 0; c < 3; c++) {
                double h = 0;
                for (int d = 0; d < D; d++) h += Jp[6 * d + a] * w * Jx[3 * d + c];
                B->W[18 * e + 3 * a + c] = h;
            }
        }
    }
    for (int m = 0; m < B->M; m++) {
        const int k1 = g->in_kf1[m], k2 = g->in_kf2[m], o1 = B->off[k1], o2 = B->off[k2];
        double er[9], J[216], rho[2] = {0, 1};
        orc_iba_edge_inertial(B->kf + ORC_IBA_KF * k1, B->kf + ORC_IBA_KF * k2, g->in_preint + ORC_IBA_PREINT * m, er, J);
        if (g->in_robust[m]) huber(B->ichi2[3 * m], B->delta_i, B->dsqr_i, rho);
        int col[24];
        for (int c = 0; c < 15; c++) col[c] = o1 < 0 ? -1 : o1 + c;
        for (int c = 0; c < 9; c++) col[15 + c] = o2 < 0 ? -1 : o2 + c;
        add_quadratic(B, J, 24, col, g->in_info + 81 * m, B->ierr + 15 * m, 9, rho[1]);
        /* EdgeGyroRW / EdgeAccRW: J = [-I, I] (G2H:648-651, :684-687) */
        double Jrw[18] = {-1, 0, 0, 1, 0, 0, 0, -1, 0, 0, 1, 0, 0, 0, -1, 0, 0, 1};
        int cg[6], ca[6];
        for (int c = 0; c < 3; c++) { cg[c] = o1 < 0 ? -1 : o1 + 9 + c; cg[3 + c] = o2 < 0 ? -1 : o2 + 9 + c; ca[c] = o1 < 0 ? -1 : o1 + 12 + c; ca[3 + c] = o2 < 0 ? -1 : o2 + 12 + c; }
        add_quadratic(B, Jrw, 6, cg, g->in_info_g + 9 * m, B->ierr + 15 * m + 9, 3, 1.0);
        add_quadratic(B, Jrw, 6, ca, g->in_info_a + 9 * m, B->ierr + 15 * m + 12, 3, 1.0);
    }
}

static int ldlt_solve(double *A, int n, const double *b, double *x)
{
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * A[(size_t)k * n + k];
        if (d == 0.0 || !isfinite(d)) return 0;
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * A[(size_t)k * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * x[k]; x[i] = s; }
    for (int i = 0; i < n; i++) x[i] /= A[(size_t)i * n + i];
    for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * x[k]; x[i] = s; }
    return 1;
}

/* BS:354-486 */
static int solve_system(struct iba *B, double lambda)
{
    const orc_iba_problem *g = B->g;
    const int n = B->n;
    memcpy(B->S, B->H, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; i++) B->S[(size_t)i * n + i] += lambda;
    memcpy(B->bs, B->b, sizeof(double) * n);
    for (int l = 0; l < B->L; l++) {
        double D[9], *Di = B->Dinv + 9 * l, db[3];
        if (B->pt_start[l + 1] == B->pt_start[l]) { memset(Di, 0, 72); continue; }
        memcpy(D, B->Hll + 9 * l, 72);
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        inv3g(D, Di);
        mv(Di, B->bl + 3 * l, db);
        for (int e1 = B->pt_start[l]; e1 < B->pt_start[l + 1]; e1++) {
            const int o1 = B->off[g->edge_kf[e1]];
            if (o1 < 0) continue;
            const double *W1 = B->W + 18 * e1;
            double Y[18];
            for (int a = 0; a < 6; a++) for (int c = 0; c < 3; c++) Y[3 * a + c] = W1[3 * a] * Di[c] + W1[3 * a + 1] * Di[3 + c] + W1[3 * a + 2] * Di[6 + c];
            for (int a = 0; a < 6; a++) B->bs[o1 + a] -= W1[3 * a] * db[0] + W1[3 * a + 1] * db[1] + W1[3 * a + 2] * db[2];
            for (int e2 = B->pt_start[l]; e2 < B->pt_start[l + 1]; e2++) {
                const int o2 = B->off[g->edge_kf[e2]];
                if (o2 < 0) continue;
                const double *W2 = B->W + 18 * e2;
                for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++)
                    B->S[(size_t)(o1 + a) * n + o2 + c] -= Y[3 * a] * W2[3 * c] + Y[3 * a + 1] * W2[3 * c + 1] + Y[3 * a + 2] * W2[3 * c + 2];
            }
        }
    }
    if (n > 0 && !ldlt_solve(B->S, n, B->bs, B->x)) return 0;
    for (int l = 0; l < B->L; l++) {
        double cl[3] = {B->bl[3 * l], B->bl[3 * l + 1], B->bl[3 * l + 2]};
        for (int e = B->pt_start[l]; e < B->pt_start[l + 1]; e++) {
            const int o = B->off[g->edge_kf[e]];
            if (o < 0) continue;
            const double *We = B->W + 18 * e, *xp = B->x + o;
            for (int c = 0; c < 3; c++) for (int a = 0; a < 6; a++) cl[c] -= We[3 * a + c] * xp[a];
        }
        mv(B->Dinv + 9 * l, cl, B->x + n + 3 * l);
    }
    return 1;
}

static void apply_update(struct iba *B)
{
    for (int k = 0; k < B->g->n_kf; k++)
        if (B->off[k] >= 0) orc_iba_kf_update(B->kf + ORC_IBA_KF * k, B->x + B->off[k], B->dim[k] == 15);
    for (int l = 0; l < B->L; l++)
        if (B->pt_start[l + 1] > B->pt_start[l]) for (int a = 0; a < 3; a++) B->pts[3 * l + a] += B->x[B->n + 3 * l + a];
}

/* LM:61-169, as ba_oracle.c's lm_iteration */
static int lm_iteration(struct iba *B, int iteration, double *chi_out)
{
    compute_errors(B);
    double current_chi = robust_chi2(B), temp_chi = current_chi;
    const double ini_chi = current_chi;
    build_system(B);
    if (iteration == 0) { B->lambda = B->p->lambda_init; B->ni = 2; B->nbad = 0; }      /* setUserLambdaInit > 0, LM:171-174 */
    double rho = 0;
    int qmax = 0;
    do {
        memcpy(B->kf_bk, B->kf, sizeof(double) * ORC_IBA_KF * B->g->n_kf);
        memcpy(B->pts_bk, B->pts, sizeof(double) * 3 * B->L);
        int ok2 = solve_system(B, B->lambda);
        apply_update(B);
        compute_errors(B);
        temp_chi = robust_chi2(B);
        if (!ok2) temp_chi = DBL_MAX;
        rho = current_chi - temp_chi;
        double scale = 0;
        for (int j = 0; j < B->n; j++) scale += B->x[j] * (B->lambda * B->x[j] + B->b[j]);
        for (int j = 0; j < 3 * B->L; j++) scale += B->x[B->n + j] * (B->lambda * B->x[B->n + j] + B->bl[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && isfinite(temp_chi)) {
            double alpha = 1. - pow((2 * rho - 1), 3);
            alpha = fmin(alpha, 2. / 3.);
            B->lambda *= fmax(1. / 3., alpha); B->ni = 2; current_chi = temp_chi;
        } else {
            B->lambda *= B->ni; B->ni *= 2;
            memcpy(B->kf, B->kf_bk, sizeof(double) * ORC_IBA_KF * B->g->n_kf);
            memcpy(B->pts, B->pts_bk, sizeof(double) * 3 * B->L);
        }
        qmax++; B->lm_trials++;
    } while (rho < 0 && qmax < B->p->max_trials);
    *chi_out = current_chi;
    if (qmax == B->p->max_trials || rho == 0) return 0;
    if ((ini_chi - current_chi) * 1e3 < ini_chi) B->nbad++; else B->nbad = 0;
    if (B->nbad >= 3) return 0;
    return 1;
}

int orc_iba_solve(const orc_iba_problem *g, const orc_iba_params *p, double *kf_state, double *points,
                  uint8_t *edge_outlier, orc_iba_stats *stats)
{
    orc_iba_stats st; memset(&st, 0, sizeof(st));
    for (int e = 0; e < g->n_edges; e++) {
        if (g->edge_kf[e] < 0 || g->edge_kf[e] >= g->n_kf || g->edge_point[e] < 0 || g->edge_point[e] >= g->n_points) return -1;
        if (e && g->edge_point[e] < g->edge_point[e - 1]) return -1;
    }
    for (int m = 0; m < g->n_inertial; m++) {
        const int k1 = g->in_kf1[m], k2 = g->in_kf2[m];
        if (k1 < 0 || k1 >= g->n_kf || k2 < 0 || k2 >= g->n_kf || !g->kf_imu[k1] || !g->kf_imu[k2]) return -1;
    }
    struct iba B; memset(&B, 0, sizeof(B));
    B.g = g; B.p = p; B.L = g->n_points; B.E = g->n_edges; B.M = g->n_inertial;
    B.off = (int *)malloc(sizeof(int) * (g->n_kf + 1)); B.dim = (int *)malloc(sizeof(int) * (g->n_kf + 1));
    for (int k = 0; k < g->n_kf; k++) {
        B.dim[k] = g->kf_imu[k] ? 15 : 6;
        if (g->kf_fixed[k]) B.off[k] = -1; else { B.off[k] = B.n; B.n += B.dim[k]; }
    }
    const int n = B.n, L = B.L, E = B.E, M = B.M;
    B.pt_start = (int *)calloc(L + 2, sizeof(int));
    for (int e = 0; e < E; e++) B.pt_start[g->edge_point[e] + 1]++;
    for (int l = 0; l < L; l++) B.pt_start[l + 1] += B.pt_start[l];
#define AL(T, c) (T *)calloc((size_t)(c) > 0 ? (size_t)(c) : 1, sizeof(T))
    B.kf = AL(double, (size_t)ORC_IBA_KF * g->n_kf); B.kf_bk = AL(double, (size_t)ORC_IBA_KF * g->n_kf);
    B.pts = AL(double, 3 * (size_t)L); B.pts_bk = AL(double, 3 * (size_t)L);
    B.err = AL(double, 3 * (size_t)E); B.chi2 = AL(double, E); B.ierr = AL(double, 15 * (size_t)M); B.ichi2 = AL(double, 3 * (size_t)M);
    B.H = AL(double, (size_t)n * n); B.S = AL(double, (size_t)n * n); B.b = AL(double, n); B.bs = AL(double, n);
    B.Hll = AL(double, 9 * (size_t)L); B.bl = AL(double, 3 * (size_t)L); B.Dinv = AL(double, 9 * (size_t)L); B.W = AL(double, 18 * (size_t)E);
    B.x = AL(double, (size_t)n + 3 * L);
#undef AL
    memcpy(B.kf, kf_state, sizeof(double) * ORC_IBA_KF * g->n_kf);
    memcpy(B.pts, points, sizeof(double) * 3 * L);
    /* thHuberMono = sqrt(5.991) etc. are floats handed to setDelta(double) (LIBA:4893-4896, :4838) */
    B.delta_m = (double)sqrtf(5.991f); B.dsqr_m = B.delta_m * B.delta_m;
    B.delta_s = (double)sqrtf(7.815f); B.dsqr_s = B.delta_s * B.delta_s;
    B.delta_i = sqrt(16.92); B.dsqr_i = B.delta_i * B.delta_i;

    /* LIBA:5045-5049: computeActiveErrors, err = activeRobustChi2, optimize(opt_it), err_end = activeRobustChi2 (stored errors) */
    compute_errors(&B);
    st.err = robust_chi2(&B);
    int ok = 1;
    for (int i = 0; i < p->iterations && ok; i++) {             /* SparseOptimizer::optimize, sparse_optimizer.cpp:354-419 */
        double chi;
        ok = lm_iteration(&B, i, &chi);
        st.iterations_run++;
    }
    st.err_end = robust_chi2(&B);
    st.lm_trials = B.lm_trials;
    /* LIBA:5056-5088 (chi2() of the stored errors, isDepthPositive of the final estimates) */
    for (int e = 0; e < E; e++) {
        int out;
        if (g->edge_stereo[e] == 1) out = B.chi2[e] > (double)7.815f;
        else {
            struct camview c;
            double Rcw[9], tcw[3];
            cam_view(g, g->edge_stereo[e] == 2, &c);
            cam_pose_of(&c, B.kf + ORC_IBA_KF * g->edge_kf[e], Rcw, tcw);
            const double *X = B.pts + 3 * g->edge_point[e];
            const int depth_pos = (Rcw[6] * X[0] + Rcw[7] * X[1] + Rcw[8] * X[2] + tcw[2]) > 0.0;      /* isDepthPositive(Xw, cam_idx) */
            const int close = g->edge_close ? g->edge_close[e] : 0;
            out = (B.chi2[e] > (double)5.991f && !close) || (B.chi2[e] > (double)(1.5f * 5.991f) && close) || !depth_pos;
        }
        if (edge_outlier) edge_outlier[e] = (uint8_t)out;
        st.n_outliers += out;
    }
    const float ferr = (float)st.err, fend = (float)st.err_end;
    st.failed = ((2 * ferr < fend || isnan(ferr) || isnan(fend)) && !p->large) ? 1 : 0;       /* LIBA:5096 */
    if (!st.failed) {
        memcpy(kf_state, B.kf, sizeof(double) * ORC_IBA_KF * g->n_kf);
        memcpy(points, B.pts, sizeof(double) * 3 * L);
    }
    if (stats) *stats = st;
    free(B.off); free(B.dim); free(B.pt_start); free(B.kf); free(B.kf_bk); free(B.pts); free(B.pts_bk); free(B.err); free(B.chi2);
    free(B.ierr); free(B.ichi2); free(B.H); free(B.S); free(B.b); free(B.bs); free(B.Hll); free(B.bl); free(B.Dinv); free(B.W); free(B.x);
    return 0;
}
