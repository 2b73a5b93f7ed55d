/*
 * ba_oracle.h -- CPU ORACLE (test infrastructure, NOT product code) for the local-BA rows
 * B1-B8 of SURVEY.md section 8a: the numerical core of
 *   Optimizer::LocalBundleAdjustment            /root/reference/src/Optimizer.cc:1699-2344
 * i.e. g2o Levenberg-Marquardt + Schur complement (BlockSolver_6_3) restated in plain C,
 * FP64.  g2o needs Eigen3 (absent) so the reference itself is UNBUILDABLE here; parity
 * target is 1e-4 RMSE on poses/points (BASELINE.json), "parity unpinned" against a real
 * g2o build; pinned by analytic-vs-numeric Jacobian checks, exact-solution recovery on
 * noise-free graphs and committed goldens (tests/test_oracle_ba.py).
 */
#ifndef BA_ORACLE_H
#define BA_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as orbhip_ba_graph / orbhip_ba_params / orbhip_ba_stats (include/orbhip.h). */
typedef struct {
    int32_t n_poses, n_points, n_edges;
    const uint8_t *pose_fixed;
    const int32_t *edge_pose;
    const int32_t *edge_point;
    const double *edge_obs;
    const double *edge_inv_sigma2;
    const uint8_t *edge_stereo;
    double fx, fy, cx, cy, bf;
    int32_t camera_model;       /* 0 = Pinhole, 1 = KannalaBrandt8 (monocular edges; CameraModels/KannalaBrandt8.cpp:52-69,166-195) */
    double kb[4];               /* k1..k4 = mvParameters[4..7] */
    /* second, rigidly attached camera (pKFi->mpCamera2, mTrl): edges with edge_stereo[e] == 2 are
     * EdgeSE3ProjectXYZToBody (OptimizableTypes.h:112-141, OptimizableTypes.cpp:192-213; Optimizer.cc:2001-2032) */
    double Trl[7];              /* mTrl as (qx,qy,qz,qw,tx,ty,tz) */
    double fx2, fy2, cx2, cy2;
    int32_t camera2_model;
    double kb2[4];
    /* per-keyframe calibration: every edge projects through its own keyframe's camera (Optimizer.cc:1961, :1990-1994, :2021-2023).
     * n_cameras > 0: pose i uses cameras[pose_camera[i]], the fields above are ignored */
    int32_t n_cameras;
    const struct orc_ba_camera *cameras;
    const int32_t *pose_camera;
} orc_ba_graph;
typedef struct orc_ba_camera {
    double fx, fy, cx, cy, bf;
    int32_t camera_model;
    double kb[4];
    double Trl[7];
    double fx2, fy2, cx2, cy2;
    int32_t camera2_model;
    double kb2[4];
} orc_ba_camera;

typedef struct {
    int32_t iters1, iters2;
    double huber_mono2, huber_stereo2;
    double user_lambda_init;
    double tau;
    int32_t max_trials;
    int32_t stage2_exclude_outliers, stage2_drop_robust, no_discard;   /* merge-LBA variant, Optimizer.cc:6255-6800 */
    double gate_mono2, gate_stereo2;                                   /* 0 = same as the Huber deltas */
} orc_ba_params;

typedef struct {
    int32_t iterations_run[2];
    int32_t lm_trials;
    int32_t n_outliers;
    int32_t discarded;
    double chi2_initial, chi2_final;
} orc_ba_stats;

void orc_ba_default_params(orc_ba_params *p);
void orc_ba_merge_params(orc_ba_params *p);
/* Optimizer::BundleAdjustment / GlobalBundleAdjustemnt (src/Optimizer.cc:54-330): one pass, no outlier stage */
void orc_ba_global_params(orc_ba_params *p, int iterations, int robust);   /* Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, pbStopFlag) */
/* Returns 0 ok, -5 aborted before start.  poses [n_poses*7] (qx,qy,qz,qw,tx,ty,tz), points [n_points*3]. */
int orc_ba_solve(const orc_ba_graph *g, const orc_ba_params *p, const volatile uint8_t *abort_flag,
                 double *poses, double *points, uint8_t *edge_outlier, orc_ba_stats *stats);
int orc_ba_solve_ex(const orc_ba_graph *g, const orc_ba_params *p, const volatile uint8_t *abort_flag,
                    double *poses, double *points, uint8_t *edge_outlier, orc_ba_stats *stats,
                    double *edge_chi2_out, double *edge_depth_out);   /* + the stored chi2 and the depth the outlier gates test */

/* unit-test hooks */
void orc_se3_exp(const double upd6[6], double q_out[4], double t_out[3]);          /* se3quat.h:223-257 */
void orc_se3_oplus(const double upd6[6], double pose7[7]);                         /* T <- exp(d)*T   */
/* residual (2 or 3) and Jacobians J_point (D x 3), J_pose (D x 6), row-major. */
void orc_ba_edge(const double pose7[7], const double X[3], const double obs[3], int stereo,
                 double fx, double fy, double cx, double cy, double bf,
                 double *err, double *Jx, double *Jt);
/* same for the monocular edge seen through a KannalaBrandt8 camera (k = k1..k4) */
/* EdgeSE3ProjectXYZToBody through camera 2 of the graph g: residual 2, Jx 2x3, Jt 2x6 */
void orc_ba_edge_tobody(const orc_ba_graph *g, const double pose7[7], const double X[3], const double obs[3],
                        double *err, double *Jx, double *Jt);
void orc_ba_edge_kb8(const double pose7[7], const double X[3], const double obs[3],
                     double fx, double fy, double cx, double cy, const double k[4],
                     double *err, double *Jx, double *Jt);

/* ---------------------------------------------------------------------------------------------
 * Optimizer::PoseOptimization (/root/reference/src/Optimizer.cc:854-1168), SURVEY 8f N1: motion-only BA of
 * one frame.  Unary edges EdgeSE3ProjectXYZOnlyPose (include/OptimizableTypes.h:31-57, src/OptimizableTypes.cpp:
 * 49-63) and g2o::EdgeStereoSE3ProjectXYZOnlyPose (Thirdparty/g2o/g2o/types/types_six_dof_expmap.{h:204-236,
 * cpp:339-404}); BlockSolver_6_3 + LinearSolverDense (Eigen::LDLT, linear_solver_dense.h:65-117) + the same
 * Levenberg-Marquardt as local BA; 4 rounds x 10 iterations from the SAME initial pose with outlier
 * re-classification (chi2 as float against 5.991f / 7.815f), Huber kernel dropped for the last round.
 * The fisheye right-camera edge (EdgeSE3ProjectXYZOnlyPoseToBody, mpCamera2 != 0) is not covered.
 * obs [n][3] = (u, v, uRight); uRight < 0 selects the monocular edge (Optimizer.cc:893).  Xw [n][3] are the
 * float map-point coordinates widened to double (Optimizer.cc:913-916).
 * Returns nInitialCorrespondences - nBad (0 if n < 3, pose untouched).  outlier [n] = pFrame->mvbOutlier. */
typedef struct {
    int32_t n_edges;
    const double *Xw, *obs, *inv_sigma2;
    double fx, fy, cx, cy, bf;
    int32_t camera_model;       /* as in orc_ba_graph (pFrame->mpCamera) */
    double kb[4];
    /* pFrame->mpCamera2 != 0 (Optimizer.cc:960-1037): right[e] = 1 marks an observation in the second camera
     * (EdgeSE3ProjectXYZOnlyPoseToBody, OptimizableTypes.h:59-87, OptimizableTypes.cpp:82-106); may be NULL */
    const uint8_t *right;
    double Trl[7], fx2, fy2, cx2, cy2;
    int32_t camera2_model;
    double kb2[4];
} orc_pose_problem;
typedef struct { int32_t rounds, iterations[4], lm_trials, n_bad; } orc_pose_stats;
int orc_pose_optimization(const orc_pose_problem *P, double pose7[7], uint8_t *outlier, orc_pose_stats *stats);
/* KannalaBrandt8::project / projectJac (KannalaBrandt8.cpp:52-69, 166-195), shared with iba_oracle.c; J row-major 2x3 */
void orc_kb8_project(const double P[3], double fx, double fy, double cx, double cy, const double k[4], double uv[2]);
void orc_kb8_project_jac(const double v[3], double fx, double fy, const double k[4], double J[6]);

#ifdef __cplusplus
}
#endif
#endif
