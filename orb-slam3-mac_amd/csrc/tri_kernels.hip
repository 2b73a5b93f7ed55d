// tri_kernels.hip -- ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:969-1210) for EVERY camera combination the
// reference supports: single Pinhole or KannalaBrandt8 cameras and two-camera rigs (mpCamera2 != 0: TUM-VI stereo-fisheye, BASELINE
// config #5).  match_kernels.hip keeps the Pinhole / single-camera fast path (k_search_triangulation); this is its general sibling:
//   * rig keyframes: keypoints mvKeys | mvKeysRight, bRight = index >= NLeft, the relative pose and the camera pair of a candidate
//     picked from {ll, lr, rl, rr} (ORBmatcher.cc:994-1008, 1101-1130), no epipole test and no stereo keypoints (:1044, :1091);
//   * GeometricCamera::epipolarConstrain per camera type: Pinhole.cpp:122-144 (distance to the epipolar line of
//     F12 = K1^-T [t12]x R12 K2^-1, host-built per combination) or KannalaBrandt8.cpp:235-238 -> TriangulateMatches (:334-401):
//     ray parallax, linear triangulation through the SVD of a 4x4 float system (cv::SVD::compute = one-sided Jacobi, restated from
//     OpenCV 3.4.1 lapack.cpp JacobiSVDImpl_<float>: parity unpinned), positive depths, reprojection errors in both cameras.
// One thread per KF1 keypoint as in the fast path (this fork never sets vbMatched2: KF1 keypoints are independent); the Jacobi sweeps
// run in the thread's registers.  Every float expression is evaluated op by op (-ffp-contract=off) in the oracle's order; libm calls
// whose results differ between platforms are replaced on BOTH sides by fixed double sequences rounded to float (tanf, cosf, sinf:
// Cody-Waite + fdlibm kernels; atan2f: double atan2; hypot: sqrt(p^2 + beta^2)) -- the deviation DESIGN 2 states for orb_sincos.
#include "orb_internal.h"
#include <cfloat>

hipStream_t orbhip_ctx_stream_internal(orbhip_ctx *c);
int orbhip_ctx_device_internal(orbhip_ctx *c);
int32_t *orbhip_ctx_status_internal(orbhip_ctx *c);

namespace {

__device__ __forceinline__ int tri_hamming256(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__device__ void tri_sincos_signed(double x, double &s_out, double &c_out)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double dk = rint(x * TWO_OVER_PI);
    const int k = (int)dk;
    double r = fma(-dk, PIO2_HI, x);
    r = fma(-dk, PIO2_LO, r);
    const double z = r * r;
    double ps = fma(z, S6, S5); ps = fma(z, ps, S4); ps = fma(z, ps, S3); ps = fma(z, ps, S2); ps = fma(z, ps, S1);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, C6, C5); pc = fma(z, pc, C4); pc = fma(z, pc, C3); pc = fma(z, pc, C2); pc = fma(z, pc, C1);
    const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
    switch (k & 3) {
    case 0: s_out = s; c_out = c; break;
    case 1: s_out = c; c_out = -s; break;
    case 2: s_out = -s; c_out = -c; break;
    default: s_out = -c; c_out = s; break;
    }
}
__device__ __forceinline__ float tri_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }

// GeometricCamera::project(cv::Point3f): Pinhole.cpp:34-37, KannalaBrandt8.cpp:28-45
__device__ void tri_project(int type, const float *p, const float *P, float *uv)
{
    if (type == 0) { uv[0] = p[0] * P[0] / P[2] + p[2]; uv[1] = p[1] * P[1] / P[2] + p[3]; return; }
    const float x2_plus_y2 = P[0] * P[0] + P[1] * P[1];
    const float theta = tri_atan2f(sqrtf(x2_plus_y2), P[2]);
    const float psi = tri_atan2f(P[1], P[0]);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + p[4] * theta3 + p[5] * theta5 + p[6] * theta7 + p[7] * theta9;
    double s, c;
    tri_sincos_signed((double)psi, s, c);
    uv[0] = p[0] * r * (float)c + p[2]; uv[1] = p[1] * r * (float)s + p[3];
}
// GeometricCamera::unproject: Pinhole.cpp:57-60, KannalaBrandt8.cpp:103-130
__device__ void tri_unproject(int type, const float *p, float u, float v, float *ray)
{
    const float pwx = (u - p[2]) / p[0], pwy = (v - p[3]) / p[1];
    if (type == 0) { ray[0] = pwx; ray[1] = pwy; ray[2] = 1.f; return; }
    float scale = 1.f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    theta_d = fminf(fmaxf((float)(-M_PI / 2.f), theta_d), (float)(M_PI / 2.f));
    if ((double)theta_d > 1e-8) {
        float theta = theta_d;
#pragma unroll 1
        for (int j = 0; j < 10; j++) {
            const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            const float k0_theta2 = p[4] * theta2, k1_theta4 = p[5] * theta4, k2_theta6 = p[6] * theta6, k3_theta8 = p[7] * theta8;
            const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                    (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < 1e-6f) break;
        }
        double s, c;
        tri_sincos_signed((double)theta, s, c);
        scale = (float)(s / c) / theta_d;
    }
    ray[0] = pwx * scale; ray[1] = pwy * scale; ray[2] = 1.f;
}

// last row of Vt of cv::SVD::compute(A 4x4 CV_32F): the right singular vector of the smallest singular value.  At[i] = column i of A.
__device__ void tri_svd4_null(float (&At)[4][4], float (&v)[4])
{
    float Vt[4][4];
    double W[4];
    const float eps = FLT_EPSILON * 2;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const float t = At[i][k]; sd += (double)t * t; }
        W[i] = sd;
#pragma unroll
        for (int k = 0; k < 4; k++) Vt[i][k] = i == k ? 1.f : 0.f;
    }
#pragma unroll 1
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = i + 1; j < 4; j++) {
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < 4; k++) p += (double)At[i][k] * At[j][k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = sqrt(p * p + beta * beta);
                float c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = (float)sqrt(delta / gamma);
                    c = (float)(p / (gamma * s * 2));
                } else {
                    c = (float)sqrt((gamma + beta) / (gamma * 2));
                    s = (float)(p / (gamma * c * 2));
                }
                a = b = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float t0 = c * At[i][k] + s * At[j][k];
                    const float t1 = -s * At[i][k] + c * At[j][k];
                    At[i][k] = t0; At[j][k] = t1;
                    a += (double)t0 * t0; b += (double)t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float t0 = c * Vt[i][k] + s * Vt[j][k];
                    const float t1 = -s * Vt[i][k] + c * Vt[j][k];
                    Vt[i][k] = t0; Vt[j][k] = t1;
                }
            }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const float t = At[i][k]; sd += (double)t * t; }
        W[i] = sqrt(sd);
    }
    // the selection sort of JacobiSVDImpl_ (descending); only the row that ends up last is needed, but ties must break as there
#pragma unroll
    for (int i = 0; i < 3; i++) {
        int j = i;
#pragma unroll
        for (int k = i + 1; k < 4; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            const double tw = W[i]; W[i] = W[j]; W[j] = tw;
#pragma unroll
            for (int k = 0; k < 4; k++) { const float t = Vt[i][k]; Vt[i][k] = Vt[j][k]; Vt[j][k] = t; }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = Vt[3][k];
}

// KannalaBrandt8::TriangulateMatches (KannalaBrandt8.cpp:334-401) > 0.0001f
__device__ bool tri_kb8_constrain(int type1, const float *cam1, int type2, const float *cam2, float u1, float v1, float u2, float v2,
                                  const float *R12, const float *t12, float sigmaLevel, float unc)
{
    float r1[3], r2[3], r21[3];
    tri_unproject(type1, cam1, u1, v1, r1);
    tri_unproject(type2, cam2, u2, v2, r2);
#pragma unroll
    for (int i = 0; i < 3; i++) r21[i] = (float)((double)R12[3 * i] * r2[0] + (double)R12[3 * i + 1] * r2[1] + (double)R12[3 * i + 2] * r2[2]);
    const double dot = (double)r1[0] * r21[0] + (double)r1[1] * r21[1] + (double)r1[2] * r21[2];
    const double n1 = sqrt((double)r1[0] * r1[0] + (double)r1[1] * r1[1] + (double)r1[2] * r1[2]);
    const double n2 = sqrt((double)r21[0] * r21[0] + (double)r21[1] * r21[1] + (double)r21[2] * r21[2]);
    const float cosParallaxRays = (float)(dot / (n1 * n2));
    if ((double)cosParallaxRays > 0.9998) return false;
    float R21[9], t21[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R21[3 * i + j] = R12[3 * j + i];
#pragma unroll
    for (int i = 0; i < 3; i++) t21[i] = (float)(-1.0 * ((double)R21[3 * i] * t12[0] + (double)R21[3 * i + 1] * t12[1] + (double)R21[3 * i + 2] * t12[2]));
    // A (KannalaBrandt8.cpp:426-429) with Tcw1 = [I | 0], Tcw2 = [R21 | t21]; stored transposed: At[j] = column j of A
    float At[4][4], v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float T1r0 = j == 0 ? 1.f : 0.f, T1r1 = j == 1 ? 1.f : 0.f, T1r2 = j == 2 ? 1.f : 0.f;
        const float T2r0 = j < 3 ? R21[j] : t21[0], T2r1 = j < 3 ? R21[3 + j] : t21[1], T2r2 = j < 3 ? R21[6 + j] : t21[2];
        At[j][0] = r1[0] * T1r2 - T1r0;
        At[j][1] = r1[1] * T1r2 - T1r1;
        At[j][2] = r2[0] * T2r2 - T2r0;
        At[j][3] = r2[1] * T2r2 - T2r1;
    }
    tri_svd4_null(At, v);
    const float inv = (float)(1.0 / (double)v[3]);
    const float x3D[3] = {v[0] * inv, v[1] * inv, v[2] * inv};
    const float z1 = x3D[2];
    if (!(z1 > 0.f)) return false;
    const float z2 = (float)((double)R21[6] * x3D[0] + (double)R21[7] * x3D[1] + (double)R21[8] * x3D[2] + (double)t21[2]);
    if (z2 <= 0.f) return false;
    float uv1[2], uv2[2], x3D2[3];
    tri_project(type1, cam1, x3D, uv1);
    const float errX1 = uv1[0] - u1, errY1 = uv1[1] - v1;
    if ((double)(errX1 * errX1 + errY1 * errY1) > 5.991 * (double)sigmaLevel) return false;
#pragma unroll
    for (int i = 0; i < 3; i++)
        x3D2[i] = (float)((double)R21[3 * i] * x3D[0] + (double)R21[3 * i + 1] * x3D[1] + (double)R21[3 * i + 2] * x3D[2] + (double)t21[i]);
    tri_project(type2, cam2, x3D2, uv2);
    const float errX2 = uv2[0] - u2, errY2 = uv2[1] - v2;
    if ((double)(errX2 * errX2 + errY2 * errY2) > 5.991 * (double)unc) return false;
    return z1 > 0.0001f;
}

// KannalaBrandt8::matchAndtriangulate (KannalaBrandt8.cpp:240-332): the candidate test of the SearchForTriangulation overload that
// returns the triangulated points (ORBmatcher.cc:1212-1402).  T1 / T2 = rows 0..2 of Tcw1 / Tcw2 (world -> camera, row-major 3x4) of
// the cameras the two keypoints were seen by; the first camera is a KannalaBrandt8 (the virtual call is made on it), the second any.
// Differences from TriangulateMatches: absolute poses in the linear system, no z1 > 1e-4 test, x3D is a WORLD point.
__device__ bool tri_kb8_match_and_triangulate(const float *cam1, int type2, const float *cam2, float u1, float v1, float u2, float v2,
                                              const float *T1, const float *T2, float sigmaLevel1, float sigmaLevel2, float *x3D_out)
{
    float r1[3], r2[3], ray1[3], ray2[3];
    tri_unproject(1, cam1, u1, v1, r1);
    tri_unproject(type2, cam2, u2, v2, r2);
#pragma unroll
    for (int i = 0; i < 3; i++) {                                              // Rwc = Rcw.t(); ray = Rwc * r
        ray1[i] = (float)((double)T1[i] * r1[0] + (double)T1[4 + i] * r1[1] + (double)T1[8 + i] * r1[2]);
        ray2[i] = (float)((double)T2[i] * r2[0] + (double)T2[4 + i] * r2[1] + (double)T2[8 + i] * r2[2]);
    }
    const double dot = (double)ray1[0] * ray2[0] + (double)ray1[1] * ray2[1] + (double)ray1[2] * ray2[2];
    const double n1 = sqrt((double)ray1[0] * ray1[0] + (double)ray1[1] * ray1[1] + (double)ray1[2] * ray1[2]);
    const double n2 = sqrt((double)ray2[0] * ray2[0] + (double)ray2[1] * ray2[1] + (double)ray2[2] * ray2[2]);
    const float cosParallaxRays = (float)(dot / (n1 * n2));
    if ((double)cosParallaxRays > 0.9998) return false;
    // Triangulate(p11, p22, Tcw1, Tcw2, x3D) (KannalaBrandt8.cpp:422-435); stored transposed: At[j] = column j of A
    float At[4][4], v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        At[j][0] = r1[0] * T1[8 + j] - T1[j];
        At[j][1] = r1[1] * T1[8 + j] - T1[4 + j];
        At[j][2] = r2[0] * T2[8 + j] - T2[j];
        At[j][3] = r2[1] * T2[8 + j] - T2[4 + j];
    }
    tri_svd4_null(At, v);
    const float inv = (float)(1.0 / (double)v[3]);
    const float x3D[3] = {v[0] * inv, v[1] * inv, v[2] * inv};
    const float z1 = (float)((double)T1[8] * x3D[0] + (double)T1[9] * x3D[1] + (double)T1[10] * x3D[2] + (double)T1[11]);
    if (!(z1 > 0.f)) return false;
    const float z2 = (float)((double)T2[8] * x3D[0] + (double)T2[9] * x3D[1] + (double)T2[10] * x3D[2] + (double)T2[11]);
    if (!(z2 > 0.f)) return false;
    float uv1[2], uv2[2], xc[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
        xc[i] = (float)((double)T1[4 * i] * x3D[0] + (double)T1[4 * i + 1] * x3D[1] + (double)T1[4 * i + 2] * x3D[2] + (double)T1[4 * i + 3]);
    tri_project(1, cam1, xc, uv1);
    const float errX1 = uv1[0] - u1, errY1 = uv1[1] - v1;
    if ((double)(errX1 * errX1 + errY1 * errY1) > 5.991 * (double)sigmaLevel1) return false;
#pragma unroll
    for (int i = 0; i < 3; i++)
        xc[i] = (float)((double)T2[4 * i] * x3D[0] + (double)T2[4 * i + 1] * x3D[1] + (double)T2[4 * i + 2] * x3D[2] + (double)T2[4 * i + 3]);
    tri_project(type2, cam2, xc, uv2);
    const float errX2 = uv2[0] - u2, errY2 = uv2[1] - v2;
    if ((double)(errX2 * errX2 + errY2 * errY2) > 5.991 * (double)sigmaLevel2) return false;
    x3D_out[0] = x3D[0]; x3D_out[1] = x3D[1]; x3D_out[2] = x3D[2];
    return true;
}

struct TriSideG { const int32_t *node_ids, *node_start, *feat, *nnodes; };
struct TriLevelsG { float sigma2_1[16], scale2[16], sigma2_2[16]; };
#define TRIG_THREADS 128
#define TRIG_TH_LOW 50
#define TRIG_HISTO 30

// BIG (round 4): keyframes of more than 4096 keypoints (to 16384) read KF2's descriptors from global memory instead of LDS.
// MT (round 4): the overload that also returns the triangulated points (ORBmatcher.cc:1212-1402): the candidate test is
// GeometricCamera::matchAndtriangulate with the absolute poses of the two cameras (poses_), no stereo / epipole gates (bOnlyStereo is
// not read there), and the world point of every kept match goes to points12_ [pairs][max_n][3].
template <bool BIG, bool MT>
__global__ __launch_bounds__(TRIG_THREADS) void k_search_triangulation_general(const int32_t *nid1_, const uint8_t *mp1_, const orbhip_keypoint *kp1_,
        const uint8_t *desc1_, const float *ur1_, const int32_t *n1_, TriSideG S2, const uint8_t *mp2_, const orbhip_keypoint *kp2_,
        const uint8_t *desc2_, const float *ur2_, const int32_t *n2_, const orbhip_tri_pair_general *geom_, int max_nodes, int max_n,
        size_t kp_stride, TriLevelsG lv, int check_ori, int cap_n, int32_t *matches12_, int32_t *nmatches_, int32_t *status,
        const orbhip_tri_pair_poses *poses_, float *points12_)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t trig_lds[];
    uint4 *dlds = reinterpret_cast<uint4 *>(trig_lds);                        // [cap_n][2] KF2 descriptors (BIG: absent)
    uint8_t *flag2 = reinterpret_cast<uint8_t *>(dlds + (BIG ? 0 : 2 * (size_t)cap_n));   // [cap_n] bit0: has a map point, bit1: stereo
    int8_t *bin1 = reinterpret_cast<int8_t *>(flag2 + cap_n);                 // [cap_n] rotation bin of KF1 keypoint i's match
    __shared__ orbhip_tri_pair_general g;
    __shared__ orbhip_tri_pair_poses P;
    __shared__ int hist[TRIG_HISTO];
    __shared__ int s_keep[3];
    __shared__ int s_cnt;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int n1 = n1_[pair], n2 = n2_[pair], nn2 = S2.nnodes[pair];
    int32_t *matches12 = matches12_ + (size_t)pair * max_n;
    if (n1 > cap_n || n2 > cap_n || n1 > max_n || n2 > max_n || nn2 > max_nodes) {
        if (tid == 0) { atomicExch(status, ORBHIP_E_CAPACITY); nmatches_[pair] = 0; }
        return;
    }
    const int32_t *nid1 = nid1_ + (size_t)pair * max_n;
    const uint8_t *mp1 = mp1_ + (size_t)pair * max_n, *mp2 = mp2_ + (size_t)pair * max_n;
    const float *ur1 = ur1_ ? ur1_ + (size_t)pair * max_n : nullptr, *ur2 = ur2_ ? ur2_ + (size_t)pair * max_n : nullptr;
    const int32_t *ids2 = S2.node_ids + (size_t)pair * max_nodes, *st2 = S2.node_start + (size_t)pair * (max_nodes + 1), *fe2 = S2.feat + (size_t)pair * max_n;
    const orbhip_keypoint *kp1 = kp1_ + (size_t)pair * kp_stride, *kp2 = kp2_ + (size_t)pair * kp_stride;
    const uint4 *d1 = reinterpret_cast<const uint4 *>(desc1_ + (size_t)pair * kp_stride * 32);
    const uint4 *d2 = reinterpret_cast<const uint4 *>(desc2_ + (size_t)pair * kp_stride * 32);
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(geom_ + pair);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&g);
        for (int i = tid; i < (int)(sizeof(orbhip_tri_pair_general) / 4); i += TRIG_THREADS) dst[i] = src[i];
        if (MT) {
            const uint32_t *psrc = reinterpret_cast<const uint32_t *>(poses_ + pair);
            uint32_t *pdst = reinterpret_cast<uint32_t *>(&P);
            for (int i = tid; i < (int)(sizeof(orbhip_tri_pair_poses) / 4); i += TRIG_THREADS) pdst[i] = psrc[i];
        }
    }
    for (int i = tid; i < TRIG_HISTO; i += TRIG_THREADS) hist[i] = 0;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const bool cam2nd1 = g.nleft1 != -1, cam2nd2 = g.nleft2 != -1;           // pKF->mpCamera2 != 0
    for (int j = tid; j < n2; j += TRIG_THREADS) {
        if (!BIG) { dlds[2 * j] = d2[2 * j]; dlds[2 * j + 1] = d2[2 * j + 1]; }
        flag2[j] = (uint8_t)((mp2[j] ? 1 : 0) | ((!cam2nd2 && ur2 && ur2[j] >= 0.0f) ? 2 : 0));      // :1073
    }
    __syncthreads();
    const float factor = 1.0f / TRIG_HISTO;
    int mine = 0;
    for (int idx1 = tid; idx1 < n1; idx1 += TRIG_THREADS) {
        int best_idx = -1;
        float best_pt[3] = {0.f, 0.f, 0.f};
        bin1[idx1] = -1;
        const bool st1 = !cam2nd1 && ur1 && ur1[idx1] >= 0.0f;                // :1044
        if (!mp1[idx1] && (MT || !(g.only_stereo && !st1))) {                 // (:1264-1265: only the map-point test)
            const int nid = nid1[idx1];
            int lo = 0, hi = nn2;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (ids2[mid] < nid) lo = mid + 1; else hi = mid; }
            if (lo < nn2 && ids2[lo] == nid) {
                const uint4 a0 = d1[2 * idx1], a1 = d1[2 * idx1 + 1];
                const orbhip_keypoint k1 = kp1[idx1];
                const int bRight1 = !(g.nleft1 == -1 || idx1 < g.nleft1);    // :1055-1056
                const float s1 = lv.sigma2_1[k1.octave & 15];
                int best = TRIG_TH_LOW;
                for (int j = st2[lo]; j < st2[lo + 1]; j++) {
                    const int idx2 = fe2[j];
                    const int fl = flag2[idx2];
                    if ((fl & 1) || (!MT && g.only_stereo && !(fl & 2))) continue;
                    const int dist = BIG ? tri_hamming256(a0, a1, d2[2 * idx2], d2[2 * idx2 + 1]) : tri_hamming256(a0, a1, dlds[2 * idx2], dlds[2 * idx2 + 1]);
                    if (dist > best) continue;                                // :1082 (best <= TH_LOW always)
                    const orbhip_keypoint k2 = kp2[idx2];
                    const int bRight2 = !(g.nleft2 == -1 || idx2 < g.nleft2);
                    if (!MT && !st1 && !(fl & 2) && !cam2nd1) {               // :1091-1099
                        const float ex = g.ep_x - k2.x, ey = g.ep_y - k2.y;
                        if (ex * ex + ey * ey < 100.0f * lv.scale2[k2.octave & 15]) continue;
                    }
                    const bool both = cam2nd1 && cam2nd2;                     // :1101-1130
                    const int c = both ? 2 * bRight1 + bRight2 : 0, ci1 = both ? bRight1 : 0, ci2 = both ? bRight2 : 0;
                    const float s2 = lv.sigma2_2[k2.octave & 15];
                    bool ok = !MT && g.coarse != 0;
                    if (MT) {                                                 // :1307-1324: camera and pose by bRight; Pinhole::matchAndtriangulate is
                        float pt[3];                                          // { return false; } (Pinhole.h:91-94)
                        if (g.cam1_type[bRight1] == 1 &&
                            tri_kb8_match_and_triangulate(g.cam1[bRight1], g.cam2_type[bRight2], g.cam2[bRight2], k1.x, k1.y, k2.x, k2.y, P.Tcw1[bRight1],
                                                          P.Tcw2[bRight2], s1, s2, pt)) {
                            ok = true; best_pt[0] = pt[0]; best_pt[1] = pt[1]; best_pt[2] = pt[2];
                        }
                    } else if (!ok) {
                        if (g.cam1_type[ci1] == 0) {                          // Pinhole.cpp:129-143
                            const float *F = g.F12[c];
                            const float la = k1.x * F[0] + k1.y * F[3] + F[6];
                            const float lb = k1.x * F[1] + k1.y * F[4] + F[7];
                            const float lc = k1.x * F[2] + k1.y * F[5] + F[8];
                            const float num = la * k2.x + lb * k2.y + lc;
                            const float den = la * la + lb * lb;
                            if (den != 0.0f) { const float dsqr = num * num / den; ok = (double)dsqr < 3.84 * (double)s2; }
                        } else
                            ok = tri_kb8_constrain(1, g.cam1[ci1], g.cam2_type[ci2], g.cam2[ci2], k1.x, k1.y, k2.x, k2.y, g.R12[c], g.t12[c], s1, s2);
                    }
                    if (ok) { best_idx = idx2; best = dist; }
                }
            }
        }
        if (best_idx >= 0) {
            mine++;
            if (check_ori) {                                                  // :1154-1164
                float rot = kp1[idx1].angle - kp2[best_idx].angle;
                if (rot < 0.0f) rot = rot + 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == TRIG_HISTO) bin = 0;
                atomicAdd(&hist[bin], 1); bin1[idx1] = (int8_t)bin;
            }
        }
        matches12[idx1] = best_idx;
        if (MT) {
            float *pt = points12_ + ((size_t)pair * max_n + idx1) * 3;
            pt[0] = best_pt[0]; pt[1] = best_pt[1]; pt[2] = best_pt[2];
        }
    }
    __syncthreads();
    if (check_ori) {                                                          // :1171-1189
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < TRIG_HISTO; i++) {
                const int sz = hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
                else if (sz > max3) { max3 = sz; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
            s_keep[0] = ind1; s_keep[1] = ind2; s_keep[2] = ind3;
        }
        __syncthreads();
        for (int i = tid; i < n1; i += TRIG_THREADS) {
            const int b = bin1[i];
            if (b < 0 || b == s_keep[0] || b == s_keep[1] || b == s_keep[2]) continue;
            matches12[i] = -1; mine--;
        }
    }
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (tid == 0) nmatches_[pair] = s_cnt;
}

}  // namespace

namespace {
int tri_general_launch(orbhip_ctx *ctx,
        const int32_t *d_nid1, const uint8_t *d_has_mp1, const orbhip_keypoint *d_kp1, const uint8_t *d_desc1, const float *d_u_right1,
        const int32_t *d_n1,
        const int32_t *d_node_ids2, const int32_t *d_node_start2, const int32_t *d_feat2, const int32_t *d_nnodes2,
        const uint8_t *d_has_mp2, const orbhip_keypoint *d_kp2, const uint8_t *d_desc2, const float *d_u_right2, const int32_t *d_n2,
        const orbhip_tri_pair_general *d_pair, const orbhip_tri_pair_poses *d_poses, int pairs, int max_nodes, int max_n, size_t frame_stride_kp,
        const float *level_sigma2_1, const float *scale_factors2, const float *level_sigma2_2, int nlevels, int check_orientation,
        int32_t *d_matches12, float *d_points12, int32_t *d_nmatches)
{
    if (!ctx || !d_nid1 || !d_has_mp1 || !d_kp1 || !d_desc1 || !d_n1 || !d_node_ids2 || !d_node_start2 || !d_feat2 || !d_nnodes2 ||
        !d_has_mp2 || !d_kp2 || !d_desc2 || !d_n2 || !d_pair || pairs <= 0 || max_nodes <= 0 || max_n <= 0 || !level_sigma2_1 || !scale_factors2 ||
        !level_sigma2_2 || nlevels <= 0 || nlevels > 16 || !d_matches12 || !d_nmatches || (d_poses != nullptr) != (d_points12 != nullptr)) return ORBHIP_E_BADARG;
    if (hipSetDevice(orbhip_ctx_device_internal(ctx)) != hipSuccess) return ORBHIP_E_HIP;
    TriLevelsG lv;
    for (int l = 0; l < 16; l++) {
        lv.sigma2_1[l] = l < nlevels ? level_sigma2_1[l] : 0.0f; lv.scale2[l] = l < nlevels ? scale_factors2[l] : 0.0f;
        lv.sigma2_2[l] = l < nlevels ? level_sigma2_2[l] : 0.0f;
    }
    // up to 4096 keypoints KF2's descriptors live in LDS; beyond (to 16384: the 5 x nFeatures keypoints of a monocular map's first keyframes,
    // Tracking.cc:210) they are read from global memory and only the per-keypoint flags stay in LDS
    const bool big = max_n > 4096, mt = d_poses != nullptr;
    const int lim = big ? 16384 : 4096;
    const int cap_n = ((max_n < lim ? max_n : lim) + 15) & ~15;
    const size_t lds = (size_t)cap_n * (big ? (1 + 1) : (32 + 1 + 1)) + 16;
    typedef void (*kern_t)(const int32_t *, const uint8_t *, const orbhip_keypoint *, const uint8_t *, const float *, const int32_t *, TriSideG, const uint8_t *,
                           const orbhip_keypoint *, const uint8_t *, const float *, const int32_t *, const orbhip_tri_pair_general *, int, int, size_t, TriLevelsG, int,
                           int, int32_t *, int32_t *, int32_t *, const orbhip_tri_pair_poses *, float *);
    const kern_t kern = mt ? (big ? k_search_triangulation_general<true, true> : k_search_triangulation_general<false, true>)
                           : (big ? k_search_triangulation_general<true, false> : k_search_triangulation_general<false, false>);
    if (orb_lds_optin(reinterpret_cast<const void *>(kern), orbhip_ctx_device_internal(ctx), lds)) return ORBHIP_E_HIP;
    TriSideG S2 = {d_node_ids2, d_node_start2, d_feat2, d_nnodes2};
    hipLaunchKernelGGL(kern, dim3(pairs), dim3(TRIG_THREADS), lds, orbhip_ctx_stream_internal(ctx), d_nid1, d_has_mp1, d_kp1, d_desc1, d_u_right1, d_n1, S2,
                       d_has_mp2, d_kp2, d_desc2, d_u_right2, d_n2, d_pair, max_nodes, max_n, frame_stride_kp, lv, check_orientation, cap_n, d_matches12,
                       d_nmatches, orbhip_ctx_status_internal(ctx), d_poses, d_points12);
    return hipGetLastError() == hipSuccess ? ORBHIP_OK : ORBHIP_E_HIP;
}
}  // namespace

extern "C" int orbhip_search_for_triangulation_general_device(orbhip_ctx *ctx,
        const int32_t *d_nid1, const uint8_t *d_has_mp1, const orbhip_keypoint *d_kp1, const uint8_t *d_desc1, const float *d_u_right1,
        const int32_t *d_n1,
        const int32_t *d_node_ids2, const int32_t *d_node_start2, const int32_t *d_feat2, const int32_t *d_nnodes2,
        const uint8_t *d_has_mp2, const orbhip_keypoint *d_kp2, const uint8_t *d_desc2, const float *d_u_right2, const int32_t *d_n2,
        const orbhip_tri_pair_general *d_pair, int pairs, int max_nodes, int max_n, size_t frame_stride_kp,
        const float *level_sigma2_1, const float *scale_factors2, const float *level_sigma2_2, int nlevels, int check_orientation,
        int32_t *d_matches12, int32_t *d_nmatches)
{
    return tri_general_launch(ctx, d_nid1, d_has_mp1, d_kp1, d_desc1, d_u_right1, d_n1, d_node_ids2, d_node_start2, d_feat2, d_nnodes2, d_has_mp2, d_kp2, d_desc2,
                              d_u_right2, d_n2, d_pair, nullptr, pairs, max_nodes, max_n, frame_stride_kp, level_sigma2_1, scale_factors2, level_sigma2_2, nlevels,
                              check_orientation, d_matches12, nullptr, d_nmatches);
}

extern "C" int orbhip_match_and_triangulate_device(orbhip_ctx *ctx,
        const int32_t *d_nid1, const uint8_t *d_has_mp1, const orbhip_keypoint *d_kp1, const uint8_t *d_desc1, const int32_t *d_n1,
        const int32_t *d_node_ids2, const int32_t *d_node_start2, const int32_t *d_feat2, const int32_t *d_nnodes2,
        const uint8_t *d_has_mp2, const orbhip_keypoint *d_kp2, const uint8_t *d_desc2, const int32_t *d_n2,
        const orbhip_tri_pair_general *d_pair, const orbhip_tri_pair_poses *d_poses, int pairs, int max_nodes, int max_n, size_t frame_stride_kp,
        const float *level_sigma2_1, const float *level_sigma2_2, int nlevels, int check_orientation,
        int32_t *d_matches12, float *d_points12, int32_t *d_nmatches)
{
    if (!d_poses || !d_points12) return ORBHIP_E_BADARG;
    return tri_general_launch(ctx, d_nid1, d_has_mp1, d_kp1, d_desc1, nullptr, d_n1, d_node_ids2, d_node_start2, d_feat2, d_nnodes2, d_has_mp2, d_kp2, d_desc2,
                              nullptr, d_n2, d_pair, d_poses, pairs, max_nodes, max_n, frame_stride_kp, level_sigma2_1, level_sigma2_2, level_sigma2_2, nlevels,
                              check_orientation, d_matches12, d_points12, d_nmatches);
}
