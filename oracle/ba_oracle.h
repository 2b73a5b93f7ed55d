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
} orc_ba_graph;

typedef struct {
    int32_t iters1, iters2;
    double huber_mono2, huber_stereo2;
    double user_lambda_init;
    double tau;
    int32_t max_trials;
} orc_ba_params;

typedef struct {
    int32_t iterations_run[2];
    int32_t lm_trials;
    int32_t n_outliers;
    int32_t discarded;
    double chi2_initial, chi2_final;
} orc_ba_stats;

void orc_ba_default_params(orc_ba_params *p);
/* Returns 0 ok, -5 aborted before start.  poses [n_poses*7] (qx,qy,qz,qw,tx,ty,tz), points [n_points*3]. */
int orc_ba_solve(const orc_ba_graph *g, const orc_ba_params *p, const volatile uint8_t *abort_flag,
                 double *poses, double *points, uint8_t *edge_outlier, orc_ba_stats *stats);

/* unit-test hooks */
void orc_se3_exp(const double upd6[6], double q_out[4], double t_out[3]);          /* se3quat.h:223-257 */
void orc_se3_oplus(const double upd6[6], double pose7[7]);                         /* T <- exp(d)*T   */
/* residual (2 or 3) and Jacobians J_point (D x 3), J_pose (D x 6), row-major. */
void orc_ba_edge(const double pose7[7], const double X[3], const double obs[3], int stereo,
                 double fx, double fy, double cx, double cy, double bf,
                 double *err, double *Jx, double *Jt);
#ifdef __cplusplus
}
#endif
#endif
