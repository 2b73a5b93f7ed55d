/*
 * ba_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See ba_oracle.h.
 * Restates, in plain C / FP64, with the reference's control flow:
 *   LM driver      Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-194   ("LM")
 *   optimize loop  Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:354-435                  ("SO")
 *   block solver   Thirdparty/g2o/g2o/core/block_solver.hpp:354-604                      ("BS")
 *   quadratic form Thirdparty/g2o/g2o/core/base_binary_edge.hpp:55-120                   ("BBE")
 *   Huber          Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:65-91                  ("RK")
 *   SE3            Thirdparty/g2o/g2o/types/se3quat.h:41-296                             ("SE3")
 *   edges          src/OptimizableTypes.cpp:139-160, include/OptimizableTypes.h:99-110,
 *                  Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:190-274
 *   camera         src/CameraModels/Pinhole.cpp:41-47,81-91
 *   LBA schedule   src/Optimizer.cc:2041-2181                                            ("LBA")
 * (paths relative to /root/reference).  Eigen's sparse LDLT is replaced by a dense LDL^T
 * without pivoting (same failure rule: zero pivot); op order differs -> tolerance 1e-4.
 */
#include "ba_oracle.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <float.h>

void orc_ba_default_params(orc_ba_params *p)
{
    p->iters1 = 5; p->iters2 = 10;                /* LBA:2048,2122 */
    p->huber_mono2 = 5.991; p->huber_stereo2 = 7.815;   /* LBA:1910-1911 */
    p->user_lambda_init = 0.0; p->tau = 1e-50;    /* LM:47 */
    p->max_trials = 100;                          /* LM:51 */
    p->stage2_exclude_outliers = 0; p->stage2_drop_robust = 0; p->no_discard = 0;
    p->gate_mono2 = 0; p->gate_stereo2 = 0;
}

/* Optimizer.cc:6255-6800: thHuber2D = sqrt(5.99) (:6395), gates 5.991 / 7.815 (:6554,6571), setLevel(1) +
 * setRobustKernel(0) between the passes (:6554-6577), no bail-out */
void orc_ba_merge_params(orc_ba_params *p)
{
    orc_ba_default_params(p);
    p->huber_mono2 = 5.99; p->gate_mono2 = 5.991; p->gate_stereo2 = 7.815;
    p->stage2_exclude_outliers = 1; p->stage2_drop_robust = 1; p->no_discard = 1;
}

/* Optimizer::BundleAdjustment, Optimizer.cc:62-330: one optimize(nIterations) (:284), sqrt(5.99) / sqrt(7.815) (:133-134) */
void orc_ba_global_params(orc_ba_params *p, int iterations, int robust)
{
    orc_ba_default_params(p);
    p->iters1 = iterations; p->iters2 = 0; p->no_discard = 1;
    p->huber_mono2 = robust ? 5.99 : 1e300; p->huber_stereo2 = robust ? 7.815 : 1e300;
    p->gate_mono2 = 5.991; p->gate_stereo2 = 7.815;
}

/* ---------------------------------------------------------------- quaternion / SE3 (B1) */
static void quat_to_R(const double q[4], double R[9])   /* Eigen::Quaterniond::toRotationMatrix */
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void R_to_quat(const double R[9], double q[4])   /* Eigen::Quaterniond(Matrix3d) */
{
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
}
static void quat_mul(const double a[4], const double b[4], double o[4])
{
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
static void quat_rot(const double q[4], const double v[3], double o[3])   /* Eigen q*v */
{
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
static void quat_normalize_rot(double q[4])   /* SE3:280-285 */
{
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* SE3Quat::exp, SE3:223-257 */
void orc_se3_exp(const double u[6], double q[4], double t[3])
{
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9], R[9], V[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0; for (int k = 0; k < 3; k++) s += O[3 * i + k] * O[3 * k + j];
        O2[3 * i + j] = s;
    }
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
        const double c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 9; i++) {
            const double I = (i % 4 == 0 ? 1.0 : 0.0);
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    R_to_quat(R, q);
    for (int i = 0; i < 3; i++) t[i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
    quat_normalize_rot(q);                                  /* SE3Quat(q,t) ctor, SE3:62-64 */
}

/* VertexSE3Expmap::oplusImpl (types_six_dof_expmap.h:73-76): T <- exp(update) * T ; SE3:104-110 */
void orc_se3_oplus(const double u[6], double pose[7])
{
    double qe[4], te[3], qn[4], rt[3];
    orc_se3_exp(u, qe, te);
    quat_rot(qe, pose + 4, rt);
    quat_mul(qe, pose, qn);
    quat_normalize_rot(qn);
    pose[0] = qn[0]; pose[1] = qn[1]; pose[2] = qn[2]; pose[3] = qn[3];
    pose[4] = te[0] + rt[0]; pose[5] = te[1] + rt[1]; pose[6] = te[2] + rt[2];
}

/* ---------------------------------------------------------------- edges (B2, B3) */
static void map_point(const double pose[7], const double X[3], double Xc[3])   /* SE3:217-220 */
{
    quat_rot(pose, X, Xc);
    Xc[0] += pose[4]; Xc[1] += pose[5]; Xc[2] += pose[6];
}

/* KannalaBrandt8::project(Eigen::Vector3d), KannalaBrandt8.cpp:52-69: float atan2f / sqrtf, double cos / sin */
void orc_kb8_project(const double P[3], double fx, double fy, double cx, double cy, const double k[4], double uv[2])
{
    const double x2_plus_y2 = P[0] * P[0] + P[1] * P[1];
    /* the reference calls libm atan2f / sqrtf on floats; libm's atan2f differs between platforms in the last float ulp
     * (host glibc vs the device's), which is enough to flip an LM accept/reject decision, so oracle and kernel both take
     * the float rounding of the DOUBLE atan2 -- within one float ulp of any libm, identical on both sides */
    const double theta = (float)atan2((double)sqrtf((float)x2_plus_y2), (double)(float)P[2]);
    const double psi = (float)atan2((double)(float)P[1], (double)(float)P[0]);
    const double theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const double r = theta + k[0] * theta3 + k[1] * theta5 + k[2] * theta7 + k[3] * theta9;
    uv[0] = fx * r * cos(psi) + cx;
    uv[1] = fy * r * sin(psi) + cy;
}
/* KannalaBrandt8::projectJac(Eigen::Vector3d), KannalaBrandt8.cpp:166-195; J row-major 2x3 */
void orc_kb8_project_jac(const double v[3], double fx, double fy, const double k[4], double J[6])
{
    const double x2 = v[0] * v[0], y2 = v[1] * v[1], z2 = v[2] * v[2];
    const double r2 = x2 + y2, r = sqrt(r2), r3 = r2 * r;
    const double theta = atan2(r, v[2]);
    const double theta2 = theta * theta, theta3 = theta2 * theta, theta4 = theta2 * theta2, theta5 = theta4 * theta;
    const double theta6 = theta2 * theta4, theta7 = theta6 * theta, theta8 = theta4 * theta4, theta9 = theta8 * theta;
    const double f = theta + theta3 * k[0] + theta5 * k[1] + theta7 * k[2] + theta9 * k[3];
    const double fd = 1 + 3 * k[0] * theta2 + 5 * k[1] * theta4 + 7 * k[2] * theta6 + 9 * k[3] * theta8;
    J[0] = fx * (fd * v[2] * x2 / (r2 * (r2 + z2)) + f * y2 / r3);
    J[3] = fy * (fd * v[2] * v[1] * v[0] / (r2 * (r2 + z2)) - f * v[1] * v[0] / r3);
    J[1] = fx * (fd * v[2] * v[1] * v[0] / (r2 * (r2 + z2)) - f * v[1] * v[0] / r3);
    J[4] = fy * (fd * v[2] * y2 / (r2 * (r2 + z2)) + f * x2 / r3);
    J[2] = -fx * fd * v[0] / (r2 + z2);
    J[5] = -fy * fd * v[1] / (r2 + z2);
}

static void edge_error_cam(const double pose[7], const double X[3], const double obs[3], int stereo,
                           double fx, double fy, double cx, double cy, double bf, const double *kb, double err[3])
{
    double P[3];
    map_point(pose, X, P);
    if (!stereo && kb) {                              /* OptimizableTypes.h:99-104 with pCamera = KannalaBrandt8 */
        double uv[2];
        orc_kb8_project(P, fx, fy, cx, cy, kb, uv);
        err[0] = obs[0] - uv[0]; err[1] = obs[1] - uv[1]; err[2] = 0;
        return;
    }
    if (!stereo) {                                    /* OptimizableTypes.h:99-104, Pinhole.cpp:41-47 */
        err[0] = obs[0] - (fx * P[0] / P[2] + cx);
        err[1] = obs[1] - (fy * P[1] / P[2] + cy);
        err[2] = 0;
    } else {                                          /* types_six_dof_expmap.cpp:190-197: float invz, float bf */
        const float invz = (float)(1.0f / P[2]);
        const float bff = (float)bf;
        const double r0 = P[0] * invz * fx + cx;
        err[0] = obs[0] - r0;
        err[1] = obs[1] - (P[1] * invz * fy + cy);
        err[2] = obs[2] - (r0 - (double)(bff * invz));
    }
}

static void edge_error(const double pose[7], const double X[3], const double obs[3], int stereo,
                       double fx, double fy, double cx, double cy, double bf, double err[3])
{
    edge_error_cam(pose, X, obs, stereo, fx, fy, cx, cy, bf, NULL, err);
}

/* monocular edge through KannalaBrandt8: Jx = -projectJac * R (OptimizableTypes.cpp:139-160), Jt = -projectJac * SE3deriv */
void orc_ba_edge_kb8(const double pose[7], const double X[3], const double obs[3],
                     double fx, double fy, double cx, double cy, const double k[4], double *err, double *Jx, double *Jt)
{
    double P[3], R[9], J[6];
    edge_error_cam(pose, X, obs, 0, fx, fy, cx, cy, 0.0, k, err);
    map_point(pose, X, P);
    quat_to_R(pose, R);
    orc_kb8_project_jac(P, fx, fy, k, J);
    const double x = P[0], y = P[1], z = P[2];
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++)
            Jx[3 * r + c] = -(J[3 * r] * R[c] + J[3 * r + 1] * R[3 + c] + J[3 * r + 2] * R[6 + c]);
    const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 6; c++)
            Jt[6 * r + c] = -(J[3 * r] * D[c] + J[3 * r + 1] * D[6 + c] + J[3 * r + 2] * D[12 + c]);
}

void orc_ba_edge(const double pose[7], const double X[3], const double obs[3], int stereo,
                 double fx, double fy, double cx, double cy, double bf, double *err, double *Jx, double *Jt)
{
    double P[3], R[9];
    edge_error(pose, X, obs, stereo, fx, fy, cx, cy, bf, err);
    map_point(pose, X, P);
    quat_to_R(pose, R);
    const double x = P[0], y = P[1], z = P[2];
    if (!stereo) {                                    /* OptimizableTypes.cpp:139-160 */
        const double pj[6] = {-(fx / z), -0.0, -(-fx * x / (z * z)), -0.0, -(fy / z), -(-fy * y / (z * z))};
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++)
                Jx[3 * r + c] = pj[3 * r] * R[c] + pj[3 * r + 1] * R[3 + c] + pj[3 * r + 2] * R[6 + c];
        const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 6; c++)
                Jt[6 * r + c] = pj[3 * r] * D[c] + pj[3 * r + 1] * D[6 + c] + pj[3 * r + 2] * D[12 + c];
    } else {                                          /* types_six_dof_expmap.cpp:228-274 */
        const double z2 = z * z;
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

/* ---- EdgeSE3ProjectXYZToBody (second camera of a rigid pair) */
/* SE3Quat::operator*, SE3:104-110: result = a * b */
static void se3_mul(const double a[7], const double b[7], double o[7])
{
    double rt[3], q[4];
    quat_rot(a, b + 4, rt);
    quat_mul(a, b, q);
    quat_normalize_rot(q);
    o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
    o[4] = a[4] + rt[0]; o[5] = a[5] + rt[1]; o[6] = a[6] + rt[2];
}
static void cam2_project(const orc_ba_graph *g, const double P[3], double uv[2])
{
    if (g->camera2_model == 1) orc_kb8_project(P, g->fx2, g->fy2, g->cx2, g->cy2, g->kb2, uv);
    else { uv[0] = g->fx2 * P[0] / P[2] + g->cx2; uv[1] = g->fy2 * P[1] / P[2] + g->cy2; }      /* Pinhole.cpp:41-47 */
}
static void cam2_project_jac(const orc_ba_graph *g, const double P[3], double J[6])
{
    if (g->camera2_model == 1) orc_kb8_project_jac(P, g->fx2, g->fy2, g->kb2, J);
    else {                                                                                   /* Pinhole.cpp:81-91 */
        J[0] = g->fx2 / P[2]; J[1] = 0; J[2] = -g->fx2 * P[0] / (P[2] * P[2]);
        J[3] = 0; J[4] = g->fy2 / P[2]; J[5] = -g->fy2 * P[1] / (P[2] * P[2]);
    }
}
/* computeError, OptimizableTypes.h:121-126: obs - pCamera->project((mTrl * T_lw).map(X)) ; returns X_r in Pr */
static void tobody_error(const orc_ba_graph *g, const double pose[7], const double X[3], const double obs[3], double err[3], double Pr[3])
{
    double Trw[7], uv[2];
    se3_mul(g->Trl, pose, Trw);
    map_point(Trw, X, Pr);
    cam2_project(g, Pr, uv);
    err[0] = obs[0] - uv[0]; err[1] = obs[1] - uv[1]; err[2] = 0;
}
/* linearizeOplus, OptimizableTypes.cpp:192-213 */
void orc_ba_edge_tobody(const orc_ba_graph *g, const double pose[7], const double X[3], const double obs[3],
                        double *err, double *Jx, double *Jt)
{
    double Pr[3], Trw[7], Xl[3], Xr[3], J[6], Rrw[9], Rrl[9];
    tobody_error(g, pose, X, obs, err, Pr);
    se3_mul(g->Trl, pose, Trw);
    map_point(pose, X, Xl);
    map_point(g->Trl, Xl, Xr);
    cam2_project_jac(g, Xr, J);
    quat_to_R(Trw, Rrw); quat_to_R(g->Trl, Rrl);
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++)
            Jx[3 * r + c] = -(J[3 * r] * Rrw[c] + J[3 * r + 1] * Rrw[3 + c] + J[3 * r + 2] * Rrw[6 + c]);
    const double x = Xl[0], y = Xl[1], z = Xl[2];
    const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
    double M[6];                                                     /* projectJac * R_rl, 2x3 */
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++) M[3 * r + c] = J[3 * r] * Rrl[c] + J[3 * r + 1] * Rrl[3 + c] + J[3 * r + 2] * Rrl[6 + c];
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 6; c++)
            Jt[6 * r + c] = -(M[3 * r] * D[c] + M[3 * r + 1] * D[6 + c] + M[3 * r + 2] * D[12 + c]);
}
/* The graph as the edges of pose `pi` see it: with per-keyframe calibration (n_cameras > 0) a copy whose calibration fields are that
 * keyframe's camera (e->pCamera = pKFi->mpCamera, e->fx = pKFi->fx ..., Optimizer.cc:1961, :1990-1994, :2021-2023), else g itself */
static const orc_ba_graph *graph_of_pose(const orc_ba_graph *g, int pi, orc_ba_graph *tmp)
{
    if (g->n_cameras <= 0) return g;
    const orc_ba_camera *c = g->cameras + g->pose_camera[pi];
    *tmp = *g;
    tmp->fx = c->fx; tmp->fy = c->fy; tmp->cx = c->cx; tmp->cy = c->cy; tmp->bf = c->bf; tmp->camera_model = c->camera_model;
    memcpy(tmp->kb, c->kb, sizeof(tmp->kb)); memcpy(tmp->Trl, c->Trl, sizeof(tmp->Trl));
    tmp->fx2 = c->fx2; tmp->fy2 = c->fy2; tmp->cx2 = c->cx2; tmp->cy2 = c->cy2; tmp->camera2_model = c->camera2_model;
    memcpy(tmp->kb2, c->kb2, sizeof(tmp->kb2));
    return tmp;
}

/* depth of the edge's camera-frame point (isDepthPositive of the three edge types) */
static double edge_depth(const orc_ba_graph *g0, int pi, const double pose[7], const double X[3], int type)
{
    double P[3];
    orc_ba_graph gtmp;
    const orc_ba_graph *g = graph_of_pose(g0, pi, &gtmp);              /* mTrl of the edge's own keyframe */
    if (type == 2) { double Trw[7]; se3_mul(g->Trl, pose, Trw); map_point(Trw, X, P); }
    else map_point(pose, X, P);
    return P[2];
}

/* ---------------------------------------------------------------- solver state */
struct ba {
    const orc_ba_graph *g; const orc_ba_params *p;
    int nf, L, E, n;            /* free poses, points, edges, 6*nf */
    int *hidx;                  /* pose -> hessian index or -1 */
    int *pt_start;              /* [L+1] edge ranges (edges sorted by point) */
    double *poses, *points;     /* current estimates (caller's arrays) */
    double *poses_bk, *points_bk;   /* push/pop backup (SO:600-613) */
    double *err, *chi2;         /* per edge: last evaluated residual (3) and chi2 */
    double *Hpp, *bp, *Hll, *bl, *W;   /* [nf*36] [n] [L*9] [3L] [E*18] */
    double *S, *bs, *x, *Dinv, *dcoef; /* [n*n] [n] [n+3L] [L*9] */
    double delta_m, dsqr_m, delta_s, dsqr_s;
    double lambda, ni; int nbad;
    const volatile uint8_t *abort_flag;
    int lm_trials;
    uint8_t *level;             /* 1 = excluded from the optimisation (setLevel(1)) */
    int robust;                 /* 0 after setRobustKernel(0) */
};

static int terminate(const struct ba *B) { return B->abort_flag ? *B->abort_flag != 0 : 0; }   /* sparse_optimizer.h:188 */

/* computeActiveErrors (SO:61-94) + chi2 (base_edge.h:58-61) */
static void compute_errors(struct ba *B)
{
    const orc_ba_graph *g0 = B->g;
    orc_ba_graph gtmp;
    for (int e = 0; e < B->E; e++) {
        if (B->level[e]) continue;                      /* not an active edge: _error stays as last computed */
        const orc_ba_graph *g = graph_of_pose(g0, g0->edge_pose[e], &gtmp);
        double *er = B->err + 3 * e;
        if (g->edge_stereo[e] == 2) { double Pr[3]; tobody_error(g, B->poses + 7 * g->edge_pose[e], B->points + 3 * g->edge_point[e], g->edge_obs + 3 * e, er, Pr); }
        else
            edge_error_cam(B->poses + 7 * g->edge_pose[e], B->points + 3 * g->edge_point[e], g->edge_obs + 3 * e,
                           g->edge_stereo[e], g->fx, g->fy, g->cx, g->cy, g->bf, g->camera_model == 1 ? g->kb : NULL, er);
        B->chi2[e] = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * g->edge_inv_sigma2[e];
    }
}

/* RobustKernelHuber::robustify, RK:78-91 (dsqr is a float member, robust_kernel_impl.h:84) */
static void huber(double e, double delta, double dsqr, double rho[2])
{
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; }
    else { double s = sqrt(e); rho[0] = 2 * s * delta - dsqr; rho[1] = delta / s; }
}

/* activeRobustChi2, SO:100-114 */
static double robust_chi2(const struct ba *B)
{
    double chi = 0, rho[2];
    for (int e = 0; e < B->E; e++) {
        if (B->level[e]) continue;
        if (!B->robust) { chi += B->chi2[e]; continue; }
        if (B->g->edge_stereo[e] == 1) huber(B->chi2[e], B->delta_s, B->dsqr_s, rho);
        else huber(B->chi2[e], B->delta_m, B->dsqr_m, rho);
        chi += rho[0];
    }
    return chi;
}

/* BlockSolver::buildSystem (BS:502-560) = linearizeOplus + constructQuadraticForm (BBE:55-120) */
static void build_system(struct ba *B)
{
    const orc_ba_graph *g0 = B->g;
    orc_ba_graph gtmp;
    memset(B->Hpp, 0, sizeof(double) * 36 * (B->nf ? B->nf : 1));
    memset(B->bp, 0, sizeof(double) * (B->n ? B->n : 1));
    memset(B->Hll, 0, sizeof(double) * 9 * B->L);
    memset(B->bl, 0, sizeof(double) * 3 * B->L);
    memset(B->W, 0, sizeof(double) * 18 * B->E);
    for (int e = 0; e < B->E; e++) {
        if (B->level[e]) continue;
        const orc_ba_graph *g = graph_of_pose(g0, g0->edge_pose[e], &gtmp);
        const int D = g->edge_stereo[e] == 1 ? 3 : 2;
        const int pi = g->edge_pose[e], li = g->edge_point[e], hi = B->hidx[pi];
        double er[3], Jx[9], Jt[18], rho[2];
        if (g->edge_stereo[e] == 2)
            orc_ba_edge_tobody(g, B->poses + 7 * pi, B->points + 3 * li, g->edge_obs + 3 * e, er, Jx, Jt);
        else if (g->camera_model == 1 && !g->edge_stereo[e])
            orc_ba_edge_kb8(B->poses + 7 * pi, B->points + 3 * li, g->edge_obs + 3 * e, g->fx, g->fy, g->cx, g->cy, g->kb, er, Jx, Jt);
        else
            orc_ba_edge(B->poses + 7 * pi, B->points + 3 * li, g->edge_obs + 3 * e, g->edge_stereo[e],
                        g->fx, g->fy, g->cx, g->cy, g->bf, er, Jx, Jt);
        /* NB: constructQuadraticForm uses the edge's stored _error / chi2() of the last
         * computeActiveErrors, which LM:71 ran at this same state. */
        const double *es = B->err + 3 * e;
        if (!B->robust) { rho[0] = B->chi2[e]; rho[1] = 1.; }
        else if (g->edge_stereo[e] == 1) huber(B->chi2[e], B->delta_s, B->dsqr_s, rho);
        else huber(B->chi2[e], B->delta_m, B->dsqr_m, rho);
        const double w = rho[1] * g->edge_inv_sigma2[e];          /* robustInformation, base_edge.h:96-102 */
        /* point (vertex 0, "from") */
        for (int a = 0; a < 3; a++) {
            double s = 0;
            for (int d = 0; d < D; d++) s += Jx[3 * d + a] * (-w * es[d]);
            B->bl[3 * li + a] += s;
            for (int b = 0; b < 3; b++) {
                double h = 0;
                for (int d = 0; d < D; d++) h += Jx[3 * d + a] * w * Jx[3 * d + b];
                B->Hll[9 * li + 3 * a + b] += h;
            }
        }
        if (hi >= 0) {   /* pose (vertex 1, "to") not fixed */
            for (int a = 0; a < 6; a++) {
                double s = 0;
                for (int d = 0; d < D; d++) s += Jt[6 * d + a] * (-w * es[d]);
                B->bp[6 * hi + a] += s;
                for (int b = 0; b < 6; b++) {
                    double h = 0;
                    for (int d = 0; d < D; d++) h += Jt[6 * d + a] * w * Jt[6 * d + b];
                    B->Hpp[36 * hi + 6 * a + b] += h;
                }
                for (int b = 0; b < 3; b++) {                      /* Hpl block: pose rows x point cols */
                    double h = 0;
                    for (int d = 0; d < D; d++) h += Jt[6 * d + a] * w * Jx[3 * d + b];
                    B->W[18 * e + 3 * a + b] = h;
                }
            }
        }
    }
}

static int inv3(const double *A, double *I)
{
    const double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c0 + A[1] * c1 + A[2] * c2;
    const double id = 1.0 / det;
    I[0] = c0 * id; I[1] = (A[2] * A[7] - A[1] * A[8]) * id; I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    I[3] = c1 * id; I[4] = (A[0] * A[8] - A[2] * A[6]) * id; I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    I[6] = c2 * id; I[7] = (A[1] * A[6] - A[0] * A[7]) * id; I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return det != 0.0;
}

/* dense LDL^T, no pivoting (stands in for Eigen::SimplicialLDLT, linear_solver_eigen.h:94-125;
 * failure rule: zero / non-finite pivot) */
static int ldlt_solve(double *A, int n, const double *b, double *x)
{
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k] * A[k * n + k];
        if (d == 0.0 || !isfinite(d)) return 0;
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k] * A[k * n + k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[i * n + k] * x[k]; x[i] = s; }
    for (int i = 0; i < n; i++) x[i] /= A[i * n + i];
    for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= A[k * n + i] * x[k]; x[i] = s; }
    return 1;
}

/* BlockSolver::solve with Schur complement (BS:354-486), lambda already added virtually. */
static int solve_system(struct ba *B, double lambda)
{
    const orc_ba_graph *g = B->g;
    const int n = B->n;
    memset(B->S, 0, sizeof(double) * (size_t)n * n);
    for (int h = 0; h < B->nf; h++)
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++)
                B->S[(size_t)(6 * h + a) * n + 6 * h + b] = B->Hpp[36 * h + 6 * a + b] + (a == b ? lambda : 0.0);
    memcpy(B->bs, B->bp, sizeof(double) * n);
    for (int l = 0; l < B->L; l++) {
        double D[9], *Di = B->Dinv + 9 * l, db[3];
        if (B->pt_start[l + 1] == B->pt_start[l]) { memset(Di, 0, sizeof(double) * 9); continue; }   /* not an active vertex */
        memcpy(D, B->Hll + 9 * l, sizeof(D));
        D[0] += lambda; D[4] += lambda; D[8] += lambda;                      /* setLambda, BS:564-590 */
        inv3(D, Di);
        for (int a = 0; a < 3; a++) db[a] = Di[3 * a] * B->bl[3 * l] + Di[3 * a + 1] * B->bl[3 * l + 1] + Di[3 * a + 2] * B->bl[3 * l + 2];
        for (int e1 = B->pt_start[l]; e1 < B->pt_start[l + 1]; e1++) {
            const int h1 = B->hidx[g->edge_pose[e1]];
            if (h1 < 0) continue;
            const double *W1 = B->W + 18 * e1;
            double Y[18];                                                     /* BDinv = Bi * Dinv */
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 3; b++)
                    Y[3 * a + b] = W1[3 * a] * Di[b] + W1[3 * a + 1] * Di[3 + b] + W1[3 * a + 2] * Di[6 + b];
            for (int a = 0; a < 6; a++)
                B->bs[6 * h1 + a] -= W1[3 * a] * db[0] + W1[3 * a + 1] * db[1] + W1[3 * a + 2] * db[2];
            for (int e2 = B->pt_start[l]; e2 < B->pt_start[l + 1]; e2++) {
                const int h2 = B->hidx[g->edge_pose[e2]];
                if (h2 < 0) continue;
                const double *W2 = B->W + 18 * e2;
                for (int a = 0; a < 6; a++)
                    for (int b = 0; b < 6; b++)
                        B->S[(size_t)(6 * h1 + a) * n + 6 * h2 + b] -= Y[3 * a] * W2[3 * b] + Y[3 * a + 1] * W2[3 * b + 1] + Y[3 * a + 2] * W2[3 * b + 2];
            }
        }
    }
    if (n > 0 && !ldlt_solve(B->S, n, B->bs, B->x)) return 0;              /* x untouched on failure */
    /* landmark back-substitution, BS:461-481 */
    for (int l = 0; l < B->L; l++) {
        double cl[3] = {B->bl[3 * l], B->bl[3 * l + 1], B->bl[3 * l + 2]};
        for (int e = B->pt_start[l]; e < B->pt_start[l + 1]; e++) {
            const int h = B->hidx[g->edge_pose[e]];
            if (h < 0) continue;
            const double *We = B->W + 18 * e, *xp = B->x + 6 * h;
            for (int b = 0; b < 3; b++)
                for (int a = 0; a < 6; a++) cl[b] -= We[3 * a + b] * xp[a];
        }
        const double *Di = B->Dinv + 9 * l;
        for (int a = 0; a < 3; a++) B->x[n + 3 * l + a] = Di[3 * a] * cl[0] + Di[3 * a + 1] * cl[1] + Di[3 * a + 2] * cl[2];
    }
    return 1;
}

/* SparseOptimizer::update (SO:422-435): oplus on every active vertex */
static void apply_update(struct ba *B)
{
    for (int p = 0; p < B->g->n_poses; p++)
        if (B->hidx[p] >= 0) orc_se3_oplus(B->x + 6 * B->hidx[p], B->poses + 7 * p);
    for (int l = 0; l < B->L; l++)
        if (B->pt_start[l + 1] > B->pt_start[l])
            for (int a = 0; a < 3; a++) B->points[3 * l + a] += B->x[B->n + 3 * l + a];     /* types_sba.h:52-56 */
}

/* OptimizationAlgorithmLevenberg::solve, LM:61-169.  Returns 1 = OK, 0 = Terminate. */
static int lm_iteration(struct ba *B, int iteration, double *chi_out)
{
    compute_errors(B);
    double current_chi = robust_chi2(B), temp_chi = current_chi;
    const double ini_chi = current_chi;
    build_system(B);
    if (iteration == 0) {                                     /* computeLambdaInit, LM:171-185 */
        if (B->p->user_lambda_init > 0) B->lambda = B->p->user_lambda_init;
        else {
            double md = 0;
            for (int h = 0; h < B->nf; h++) for (int a = 0; a < 6; a++) md = fmax(fabs(B->Hpp[36 * h + 7 * a]), md);
            for (int l = 0; l < B->L; l++)
                if (B->pt_start[l + 1] > B->pt_start[l])
                    for (int a = 0; a < 3; a++) md = fmax(fabs(B->Hll[9 * l + 4 * a]), md);
            B->lambda = B->p->tau * md;
        }
        B->ni = 2; B->nbad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
        memcpy(B->poses_bk, B->poses, sizeof(double) * 7 * B->g->n_poses);       /* push */
        memcpy(B->points_bk, B->points, sizeof(double) * 3 * B->L);
        int ok2 = solve_system(B, B->lambda);
        apply_update(B);
        compute_errors(B);
        temp_chi = robust_chi2(B);
        if (!ok2) temp_chi = DBL_MAX;
        rho = current_chi - temp_chi;
        double scale = 0;                                                         /* computeScale, LM:187-194 */
        for (int j = 0; j < B->n; j++) scale += B->x[j] * (B->lambda * B->x[j] + B->bp[j]);
        for (int j = 0; j < 3 * B->L; j++) scale += B->x[B->n + j] * (B->lambda * B->x[B->n + j] + B->bl[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && isfinite(temp_chi)) {
            double alpha = 1. - pow((2 * rho - 1), 3);
            alpha = fmin(alpha, 2. / 3.);
            double sf = fmax(1. / 3., alpha);
            B->lambda *= sf; B->ni = 2; current_chi = temp_chi;                   /* discardTop */
        } else {
            B->lambda *= B->ni; B->ni *= 2;
            memcpy(B->poses, B->poses_bk, sizeof(double) * 7 * B->g->n_poses);   /* pop */
            memcpy(B->points, B->points_bk, sizeof(double) * 3 * B->L);
        }
        qmax++; B->lm_trials++;
        if (getenv("ORC_BA_TRACE"))
            fprintf(stderr, "[oracle ba] iter %d qmax %d lambda %.6e chi %.9e rho %.6e ok %d nbad %d trials %d\n", iteration, qmax, B->lambda, current_chi, rho, ok2, B->nbad, B->lm_trials);
    } while (rho < 0 && qmax < B->p->max_trials && !terminate(B));
    *chi_out = current_chi;
    if (qmax == B->p->max_trials || rho == 0) return 0;
    if ((ini_chi - current_chi) * 1e3 < ini_chi) B->nbad++; else B->nbad = 0;     /* LM:157-166 */
    if (B->nbad >= 3) return 0;
    return 1;
}

/* SparseOptimizer::optimize, SO:354-419 */
static int optimize(struct ba *B, int iterations, double *chi_first, double *chi_last)
{
    int done = 0, ok = 1;
    for (int i = 0; i < iterations && !terminate(B) && ok; i++) {
        double chi;
        if (i == 0 && chi_first) { compute_errors(B); *chi_first = robust_chi2(B); }
        ok = lm_iteration(B, i, &chi);
        if (chi_last) *chi_last = chi;
        done++;
    }
    return done;
}

int orc_ba_solve(const orc_ba_graph *g, const orc_ba_params *p, const volatile uint8_t *abort_flag,
                 double *poses, double *points, uint8_t *edge_outlier, orc_ba_stats *stats)
{
    return orc_ba_solve_ex(g, p, abort_flag, poses, points, edge_outlier, stats, NULL, NULL);
}

/* as orc_ba_solve; edge_chi2_out / edge_depth_out [n_edges] (may be NULL) = the two numbers the outlier gates of LBA:2126-2173 test,
 * so that a checker can tell a flag that sits on the gate from a real disagreement */
int orc_ba_solve_ex(const orc_ba_graph *g, const orc_ba_params *p, const volatile uint8_t *abort_flag,
                    double *poses, double *points, uint8_t *edge_outlier, orc_ba_stats *stats,
                    double *edge_chi2_out, double *edge_depth_out)
{
    orc_ba_stats st; memset(&st, 0, sizeof(st));
    if (abort_flag && *abort_flag) { if (stats) *stats = st; return -5; }          /* LBA:2041-2043 */
    struct ba B; memset(&B, 0, sizeof(B));
    B.g = g; B.p = p; B.L = g->n_points; B.E = g->n_edges; B.abort_flag = abort_flag;
    B.hidx = (int *)malloc(sizeof(int) * (g->n_poses ? g->n_poses : 1));
    int *has_edge = (int *)calloc(g->n_poses ? g->n_poses : 1, sizeof(int));
    for (int e = 0; e < B.E; e++) has_edge[g->edge_pose[e]] = 1;
    for (int i = 0; i < g->n_poses; i++) B.hidx[i] = (!g->pose_fixed[i] && has_edge[i]) ? B.nf++ : -1;
    free(has_edge);
    B.n = 6 * B.nf;
    B.pt_start = (int *)calloc(B.L + 2, sizeof(int));
    for (int e = 0; e < B.E; e++) B.pt_start[g->edge_point[e] + 1]++;
    for (int l = 0; l < B.L; l++) B.pt_start[l + 1] += B.pt_start[l];
    /* working copies so that a discarded solve leaves the caller's arrays untouched */
    B.poses = (double *)malloc(sizeof(double) * 7 * (g->n_poses ? g->n_poses : 1));
    B.points = (double *)malloc(sizeof(double) * 3 * (B.L ? B.L : 1));
    memcpy(B.poses, poses, sizeof(double) * 7 * g->n_poses);
    memcpy(B.points, points, sizeof(double) * 3 * B.L);
    for (int i = 0; i < g->n_poses; i++) quat_normalize_rot(B.poses + 7 * i);      /* SE3Quat ctor */
    B.poses_bk = (double *)malloc(sizeof(double) * 7 * (g->n_poses ? g->n_poses : 1));
    B.points_bk = (double *)malloc(sizeof(double) * 3 * (B.L ? B.L : 1));
    B.err = (double *)calloc(3 * (B.E ? B.E : 1), sizeof(double));
    B.chi2 = (double *)calloc(B.E ? B.E : 1, sizeof(double));
    B.level = (uint8_t *)calloc(B.E ? B.E : 1, 1); B.robust = 1;
    B.Hpp = (double *)calloc(36 * (B.nf ? B.nf : 1), sizeof(double));
    B.bp = (double *)calloc(B.n ? B.n : 1, sizeof(double));
    B.Hll = (double *)calloc(9 * (B.L ? B.L : 1), sizeof(double));
    B.bl = (double *)calloc(3 * (B.L ? B.L : 1), sizeof(double));
    B.W = (double *)calloc(18 * (B.E ? B.E : 1), sizeof(double));
    B.S = (double *)calloc((size_t)(B.n ? B.n : 1) * (B.n ? B.n : 1), sizeof(double));
    B.bs = (double *)calloc(B.n ? B.n : 1, sizeof(double));
    B.x = (double *)calloc(B.n + 3 * B.L + 1, sizeof(double));
    B.Dinv = (double *)calloc(9 * (B.L ? B.L : 1), sizeof(double));
    /* thHuber = (float)sqrt(th2) (LBA:1910-1911); dsqr stored as float (robust_kernel_impl.h:84) */
    B.delta_m = (double)(float)sqrt(p->huber_mono2); B.dsqr_m = (double)(float)(B.delta_m * B.delta_m);
    B.delta_s = (double)(float)sqrt(p->huber_stereo2); B.dsqr_s = (double)(float)(B.delta_s * B.delta_s);

    st.iterations_run[0] = optimize(&B, p->iters1, &st.chi2_initial, &st.chi2_final);      /* LBA:2048 */
    int do_more = !(abort_flag && *abort_flag);                                              /* LBA:2056-2060 */
    const double gate_m = p->gate_mono2 > 0 ? p->gate_mono2 : p->huber_mono2, gate_s = p->gate_stereo2 > 0 ? p->gate_stereo2 : p->huber_stereo2;
    if (do_more && (p->stage2_exclude_outliers || p->stage2_drop_robust)) {                  /* merge variant, Optimizer.cc:6546-6579 */
        for (int e = 0; e < B.E; e++) {
            const double zc = edge_depth(g, g->edge_pose[e], B.poses + 7 * g->edge_pose[e], B.points + 3 * g->edge_point[e], g->edge_stereo[e]);
            if (p->stage2_exclude_outliers && (B.chi2[e] > (g->edge_stereo[e] == 1 ? gate_s : gate_m) || !(zc > 0.0))) B.level[e] = 1;
        }
        if (p->stage2_drop_robust) B.robust = 0;
    }
    if (do_more) st.iterations_run[1] = optimize(&B, p->iters2, NULL, &st.chi2_final);       /* LBA:2121-2122 */
    st.lm_trials = B.lm_trials;
    /* outliers, LBA:2126-2173: chi2 from the stored (last evaluated) error; depth from current estimates */
    for (int e = 0; e < B.E; e++) {
        const double zc = edge_depth(g, g->edge_pose[e], B.poses + 7 * g->edge_pose[e], B.points + 3 * g->edge_point[e], g->edge_stereo[e]);
        const double gate = g->edge_stereo[e] == 1 ? gate_s : gate_m;
        const int out = (B.chi2[e] > gate) || !(zc > 0.0);
        if (edge_outlier) edge_outlier[e] = (uint8_t)out;
        if (edge_chi2_out) edge_chi2_out[e] = B.chi2[e];
        if (edge_depth_out) edge_depth_out[e] = zc;
        st.n_outliers += out;
    }
    if (!p->no_discard && st.n_outliers >= B.E * 0.5 && B.E > 0) st.discarded = 1;           /* LBA:2177-2181 */
    else {
        memcpy(poses, B.poses, sizeof(double) * 7 * g->n_poses);
        memcpy(points, B.points, sizeof(double) * 3 * B.L);
    }
    if (stats) *stats = st;
    free(B.hidx); free(B.pt_start); free(B.poses); free(B.points); free(B.poses_bk); free(B.points_bk);
    free(B.level); free(B.err); free(B.chi2); free(B.Hpp); free(B.bp); free(B.Hll); free(B.bl); free(B.W); free(B.S);
    free(B.bs); free(B.x); free(B.Dinv);
    return 0;
}


/* ================================================================= PoseOptimization (Optimizer.cc:854-1168) */
struct po {
    const orc_pose_problem *P;
    double pose[7], pose_bk[7];
    double *err, *chi2;        /* stored _error / chi2() of every edge (last computeError) */
    uint8_t *level;            /* 1 = outlier, not optimised (setLevel(1), Optimizer.cc:1081) */
    int robust;
    double delta_m, dsqr_m, delta_s, dsqr_s;
    double H[36], b[6], x[6], lambda, ni; int nbad, lm_trials;
};

/* computeError of the two unary edges */
/* camera-2 parameters of a pose problem as a graph struct (what the ToBody helpers read) */
static void po_rig(const orc_pose_problem *P, orc_ba_graph *g)
{
    memset(g, 0, sizeof(*g));
    memcpy(g->Trl, P->Trl, sizeof(g->Trl));
    g->fx2 = P->fx2; g->fy2 = P->fy2; g->cx2 = P->cx2; g->cy2 = P->cy2; g->camera2_model = P->camera2_model;
    memcpy(g->kb2, P->kb2, sizeof(g->kb2));
}
static void po_edge_error(const orc_pose_problem *P, const double pose[7], int e, double er[3])
{
    if (P->right && P->right[e]) {                         /* OptimizableTypes.h:69-73 */
        orc_ba_graph g; double Pr[3];
        po_rig(P, &g);
        tobody_error(&g, pose, P->Xw + 3 * e, P->obs + 3 * e, er, Pr);
        return;
    }
    double Xc[3];
    const double *obs = P->obs + 3 * e;
    map_point(pose, P->Xw + 3 * e, Xc);
    if (obs[2] < 0 && P->camera_model == 1) {              /* OptimizableTypes.h:41-45 with pCamera = KannalaBrandt8 */
        double uv[2];
        orc_kb8_project(Xc, P->fx, P->fy, P->cx, P->cy, P->kb, uv);
        er[0] = obs[0] - uv[0]; er[1] = obs[1] - uv[1]; er[2] = 0;
    } else if (obs[2] < 0) {                               /* OptimizableTypes.h:41-45, Pinhole.cpp:41-47 */
        er[0] = obs[0] - (P->fx * Xc[0] / Xc[2] + P->cx);
        er[1] = obs[1] - (P->fy * Xc[1] / Xc[2] + P->cy);
        er[2] = 0;
    } else {                                               /* types_six_dof_expmap.cpp:339-346: float invz, DOUBLE bf */
        const float invz = (float)(1.0f / Xc[2]);
        const double r0 = Xc[0] * invz * P->fx + P->cx;
        er[0] = obs[0] - r0;
        er[1] = obs[1] - (Xc[1] * invz * P->fy + P->cy);
        er[2] = obs[2] - (r0 - P->bf * invz);
    }
}
static void po_compute_errors(struct po *S)                /* computeActiveErrors: level-0 edges only */
{
    for (int e = 0; e < S->P->n_edges; e++) {
        if (S->level[e]) continue;
        double *er = S->err + 3 * e;
        po_edge_error(S->P, S->pose, e, er);
        S->chi2[e] = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * S->P->inv_sigma2[e];
    }
}
static void po_rho(const struct po *S, int e, double rho[2])
{
    if (!S->robust) { rho[0] = S->chi2[e]; rho[1] = 1.; return; }
    if (S->P->obs[3 * e + 2] < 0) huber(S->chi2[e], S->delta_m, S->dsqr_m, rho);
    else huber(S->chi2[e], S->delta_s, S->dsqr_s, rho);
}
static double po_robust_chi2(const struct po *S)
{
    double chi = 0, rho[2];
    for (int e = 0; e < S->P->n_edges; e++) if (!S->level[e]) { po_rho(S, e, rho); chi += rho[0]; }
    return chi;
}
/* linearizeOplus + BaseUnaryEdge::constructQuadraticForm (base_unary_edge.hpp:56-86) */
static void po_build(struct po *S)
{
    const orc_pose_problem *P = S->P;
    memset(S->H, 0, sizeof(S->H)); memset(S->b, 0, sizeof(S->b));
    for (int e = 0; e < P->n_edges; e++) {
        if (S->level[e]) continue;
        const int stereo = !(P->obs[3 * e + 2] < 0), D = stereo ? 3 : 2;
        double er[3], Jx[9], Jt[18], rho[2];
        if (P->right && P->right[e]) { orc_ba_graph g; po_rig(P, &g); orc_ba_edge_tobody(&g, S->pose, P->Xw + 3 * e, P->obs + 3 * e, er, Jx, Jt); }
        else if (P->camera_model == 1 && !stereo) orc_ba_edge_kb8(S->pose, P->Xw + 3 * e, P->obs + 3 * e, P->fx, P->fy, P->cx, P->cy, P->kb, er, Jx, Jt);
        else orc_ba_edge(S->pose, P->Xw + 3 * e, P->obs + 3 * e, stereo, P->fx, P->fy, P->cx, P->cy, P->bf, er, Jx, Jt);
        const double *es = S->err + 3 * e;
        po_rho(S, e, rho);
        const double w = rho[1] * P->inv_sigma2[e];
        for (int a = 0; a < 6; a++) {
            double s = 0;
            for (int d = 0; d < D; d++) s += Jt[6 * d + a] * (-w * es[d]);
            S->b[a] += s;
            for (int c = 0; c < 6; c++) {
                double h = 0;
                for (int d = 0; d < D; d++) h += Jt[6 * d + a] * w * Jt[6 * d + c];
                S->H[6 * a + c] += h;
            }
        }
    }
}
/* LinearSolverDense: Eigen::LDLT + isPositive() (stand-in: unpivoted LDL^T, fails on a non-positive pivot) */
static int po_solve(struct po *S)
{
    double A[36];
    memcpy(A, S->H, sizeof(A));
    for (int i = 0; i < 6; i++) A[7 * i] += S->lambda;
    for (int j = 0; j < 6; j++) {
        double d = A[7 * j];
        for (int k = 0; k < j; k++) d -= A[6 * j + k] * A[6 * j + k] * A[7 * k];
        if (!(d > 0.0) || !isfinite(d)) return 0;
        A[7 * j] = d;
        for (int i = j + 1; i < 6; i++) {
            double s = A[6 * i + j];
            for (int k = 0; k < j; k++) s -= A[6 * i + k] * A[6 * j + k] * A[7 * k];
            A[6 * i + j] = s / d;
        }
    }
    double x[6];
    for (int i = 0; i < 6; i++) { double s = S->b[i]; for (int k = 0; k < i; k++) s -= A[6 * i + k] * x[k]; x[i] = s; }
    for (int i = 0; i < 6; i++) x[i] /= A[7 * i];
    for (int i = 5; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < 6; k++) s -= A[6 * k + i] * x[k]; x[i] = s; }
    memcpy(S->x, x, sizeof(x));
    return 1;
}
static int po_lm_iteration(struct po *S, int iteration)        /* LM:61-169, as lm_iteration above */
{
    po_compute_errors(S);
    double current_chi = po_robust_chi2(S), temp_chi = current_chi;
    const double ini_chi = current_chi;
    po_build(S);
    if (iteration == 0) {
        double md = 0;
        for (int a = 0; a < 6; a++) md = fmax(fabs(S->H[7 * a]), md);
        S->lambda = 1e-50 * md; S->ni = 2; S->nbad = 0;           /* LM:47,171-185 */
    }
    double rho = 0; int qmax = 0;
    do {
        memcpy(S->pose_bk, S->pose, sizeof(S->pose));
        const int ok2 = po_solve(S);
        orc_se3_oplus(S->x, S->pose);
        po_compute_errors(S);
        temp_chi = po_robust_chi2(S);
        if (!ok2) temp_chi = DBL_MAX;
        rho = current_chi - temp_chi;
        double scale = 0;
        for (int j = 0; j < 6; j++) scale += S->x[j] * (S->lambda * S->x[j] + S->b[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && isfinite(temp_chi)) {
            double alpha = 1. - pow((2 * rho - 1), 3);
            alpha = fmin(alpha, 2. / 3.);
            S->lambda *= fmax(1. / 3., alpha); S->ni = 2; current_chi = temp_chi;
        } else {
            S->lambda *= S->ni; S->ni *= 2;
            memcpy(S->pose, S->pose_bk, sizeof(S->pose));
        }
        qmax++; S->lm_trials++;
    } while (rho < 0 && qmax < 100);
    if (qmax == 100 || rho == 0) return 0;
    if ((ini_chi - current_chi) * 1e3 < ini_chi) S->nbad++; else S->nbad = 0;
    if (S->nbad >= 3) return 0;
    return 1;
}

int orc_pose_optimization(const orc_pose_problem *P, double pose7[7], uint8_t *outlier, orc_pose_stats *stats)
{
    orc_pose_stats st; memset(&st, 0, sizeof(st));
    const int n = P->n_edges;
    if (outlier) memset(outlier, 0, (size_t)(n > 0 ? n : 0));           /* Optimizer.cc:896 */
    if (n < 3) { if (stats) *stats = st; return 0; }                    /* Optimizer.cc:1040-1041 */
    struct po S; memset(&S, 0, sizeof(S));
    S.P = P;
    S.err = (double *)calloc(3 * (size_t)n, sizeof(double));
    S.chi2 = (double *)calloc((size_t)n, sizeof(double));
    S.level = (uint8_t *)calloc((size_t)n, 1);
    S.delta_m = (double)(float)sqrt(5.991); S.dsqr_m = (double)(float)(S.delta_m * S.delta_m);     /* Optimizer.cc:887-888 */
    S.delta_s = (double)(float)sqrt(7.815); S.dsqr_s = (double)(float)(S.delta_s * S.delta_s);
    S.robust = 1;
    double pose0[7];
    memcpy(pose0, pose7, sizeof(pose0));
    quat_normalize_rot(pose0);
    memcpy(S.pose, pose0, sizeof(pose0));
    int nbad = 0;
    for (int it = 0; it < 4; it++) {
        memcpy(S.pose, pose0, sizeof(pose0));                           /* Optimizer.cc:1053 */
        int nact = 0;
        for (int e = 0; e < n; e++) nact += !S.level[e];
        if (nact > 0) {                                                 /* optimize(10), SO:354-419 */
            int ok = 1;
            for (int i = 0; i < 10 && ok; i++) { ok = po_lm_iteration(&S, i); st.iterations[it]++; }
        }
        nbad = 0;
        for (int e = 0; e < n; e++) {                                   /* Optimizer.cc:1058-1145 */
            if (S.level[e]) {                                           /* e->computeError() at the current estimate */
                double *er = S.err + 3 * e;
                po_edge_error(P, S.pose, e, er);
                S.chi2[e] = (er[0] * er[0] + er[1] * er[1] + er[2] * er[2]) * P->inv_sigma2[e];
            }
            const float chi2 = (float)S.chi2[e];
            const float gate = P->obs[3 * e + 2] < 0 ? 5.991f : 7.815f;
            if (chi2 > gate) { S.level[e] = 1; nbad++; } else S.level[e] = 0;
        }
        if (it == 2) S.robust = 0;                                      /* setRobustKernel(0), Optimizer.cc:1084 */
        st.rounds++;
        if (n < 10) break;                                              /* Optimizer.cc:1147-1148 */
    }
    memcpy(pose7, S.pose, sizeof(S.pose));
    if (outlier) memcpy(outlier, S.level, (size_t)n);
    st.lm_trials = S.lm_trials; st.n_bad = nbad;
    if (stats) *stats = st;
    free(S.err); free(S.chi2); free(S.level);
    return n - nbad;
}
