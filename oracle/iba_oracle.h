/* iba_oracle.h -- CPU restatement of Optimizer::LocalInertialBA's numerical core (TEST INFRASTRUCTURE ONLY: tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product path never does).
 *
 * Follows  src/Optimizer.cc:4574-5187 (graph construction, LM call, outlier gates, fail check),
 *          include/G2oTypes.h:59-141 + src/G2oTypes.cc:25-220 (ImuCamPose), :349-482 (EdgeMono / EdgeStereo),
 *          :693-800 (EdgeInertial), include/G2oTypes.h:632-700 (EdgeGyroRW / EdgeAccRW), src/G2oTypes.cc:984-1077 (SO3 helpers),
 *          src/ImuTypes.cc:351-378 (bias-corrected preintegrated deltas) and the g2o Levenberg-Marquardt / Schur solver
 *          already restated in ba_oracle.c (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-194,
 *          block_solver.hpp:354-486, base_multi_edge.hpp:36-48 + base_multi_edge.hpp computeQuadraticForm).
 *
 * PARITY UNPINNED: the reference holds no golden vectors for this path and cannot be built here (Eigen, OpenCV absent).
 * Two places where this restatement is NOT the reference's arithmetic, both below the 1e-4 tolerance of the tests:
 *   - Preintegrated::GetDeltaRotation/Velocity/Position (ImuTypes.cc:357-378) and ExpSO3's re-orthonormalisation
 *     (G2oTypes.cc:991-1008 -> IMU::NormalizeRotation, ImuTypes.cc:30-36) run in float cv::Mat arithmetic in the reference
 *     (IMU::Bias members are float); here they are double, the nearest rotation taken by two Newton polar steps.
 *   - Eigen::SimplicialLDLT with AMD ordering is replaced by a dense LDL^T without pivoting (as in ba_oracle.c).
 * ImuCamPose::Update's "NormalizeRotation(Rwb)" (G2oTypes.cc:205-210) discards its result and is a no-op in the reference;
 * it is one here too. */
#ifndef ORC_IBA_ORACLE_H
#define ORC_IBA_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* keyframe state: Rwb[9] row-major, twb[3], velocity[3], gyro bias[3], acc bias[3] */
#define ORC_IBA_KF 21
/* preintegration record of one EdgeInertial (IMU::Preintegrated members): dT, dR[9], dV[3], dP[3], JRg[9], JVg[9], JVa[9],
 * JPg[9], JPa[9], the bias it was integrated with: bg[3], ba[3] */
#define ORC_IBA_PREINT 67

typedef struct {
    int32_t n_kf;
    const uint8_t *kf_fixed;        /* setFixed(true): lFixedKeyFrames (Optimizer.cc:4757-4781) */
    const uint8_t *kf_imu;          /* pKFi->bImu: the keyframe has VertexVelocity / GyroBias / AccBias (:4726-4740) */
    double Rcb[9], tcb[3];          /* mImuCalib.Tcb (G2oTypes.cc:49-52) */
    double fx, fy, cx, cy, bf;      /* pCamera[0] */
    int32_t camera_model;           /* 0 Pinhole, 1 KannalaBrandt8 (project / projectJac of src/CameraModels/) */
    double kb[4];
    int32_t has_cam2;               /* pKF->mpCamera2: ImuCamPose holds a second camera (G2oTypes.cc:57-67) */
    double Trl[12];                 /* pKF->mTrl, 3x4 row-major: Rcb[1] = Rrl Rcb[0], tcb[1] = Rrl tcb[0] + trl */
    double fx2, fy2, cx2, cy2;
    int32_t camera2_model;
    double kb2[4];
    int32_t n_points;
    int32_t n_edges;                /* EdgeMono / EdgeStereo, grouped by point (ascending edge_point), as :4914-5034 creates them */
    const int32_t *edge_kf, *edge_point;
    const double *edge_obs;         /* [3]: kpUn.pt.x, kpUn.pt.y, mvuRight (unused when mono) */
    const uint8_t *edge_stereo;     /* 0 EdgeMono(0), 1 EdgeStereo(0), 2 EdgeMono(1) (right camera, Optimizer.cc:5000-5031); a keyframe may
                                       hold a left and a right edge to the same point */
    const double *edge_inv_sigma2;  /* mvInvLevelSigma2[octave] / uncertainty2 (:4949-4952) */
    const uint8_t *edge_close;      /* pMP->mTrackDepth < 10 (:5063): monocular gate 1.5 x 5.991 */
    int32_t n_inertial;             /* EdgeInertial + EdgeGyroRW + EdgeAccRW triples (:4784-4868) */
    const int32_t *in_kf1, *in_kf2; /* pKFi->mPrevKF, pKFi */
    const double *in_preint;        /* [ORC_IBA_PREINT] */
    const double *in_info;          /* [81] EdgeInertial information (G2oTypes.cc:702-714, x 1e-2 for the edge into the fixed keyframe :4836) */
    const double *in_info_g, *in_info_a;   /* [9] C.rowRange(9,12)^-1, C.rowRange(12,15)^-1 (:4845-4863) */
    const uint8_t *in_robust;       /* Huber sqrt(16.92) (:4828-4838) */
} orc_iba_problem;

typedef struct {
    int32_t iterations;             /* 10, or 4 when bLarge (:4579-4585) */
    double lambda_init;             /* 1.0, or 1e-2 when bLarge (:4699-4710) */
    int32_t large;                  /* bLarge: no fail check (:5096) */
    int32_t max_trials;             /* 100 */
} orc_iba_params;

typedef struct {
    int32_t iterations_run, lm_trials, n_outliers, failed;   /* failed: 2*err < err_end or NaN (:5096-5100): nothing written back */
    double err, err_end;            /* activeRobustChi2 before / after (:5047-5049; float in the reference) */
} orc_iba_stats;

void orc_iba_default_params(orc_iba_params *p, int large);

/* kf_state [n_kf][ORC_IBA_KF], points [n_points][3]: in/out (untouched when failed).  edge_outlier [n_edges] = the
 * observation goes to vToErase (:5056-5088).  Returns 0, or < 0 on a malformed problem. */
int orc_iba_solve(const orc_iba_problem *g, const orc_iba_params *p, double *kf_state, double *points,
                  uint8_t *edge_outlier, orc_iba_stats *stats);

/* pieces, for the Jacobian tests */
void orc_iba_kf_update(double *s, const double *dx, int imu);      /* ImuCamPose::Update + the three "+=" vertices; dx[15] */
/* EdgeInertial::computeError / linearizeOplus: J [9][24], columns = pose1(6) v1(3) bg1(3) ba1(3) pose2(6) v2(3) */
void orc_iba_edge_inertial(const double *s1, const double *s2, const double *preint, double err[9], double J[216]);
/* EdgeMono / EdgeStereo: err[3], Jx [3][3] (point), Jp [3][6] (pose); rows 0..1 only when mono */
void orc_iba_edge_visual(const orc_iba_problem *g, const double *s, const double X[3], const double obs[3], int type,
                         double err[3], double Jx[9], double Jp[18]);
void orc_iba_exp_so3(const double w[3], double R[9]);
void orc_iba_log_so3(const double R[9], double w[3]);

#ifdef __cplusplus
}
#endif
#endif
