// cvmath.h -- the handful of CV_32F cv::Mat expressions the reference's matcher / optimiser geometry is written in (Rcw*p3Dw+tcw,
// -Rcw.t()*tcw, R1w*R2w.t(), sRcw/scw, cv::norm, Mat::dot ...), spelled out element by element so that the host shims compile with
// the real OpenCV as well as with cvlite.h and round where OpenCV 3.4 rounds.  OpenCV is not part of the reference tree: this
// restates its arithmetic from the library's published source (modules/core/src/matmul.cpp, convert.cpp) and is PARITY UNPINNED:
//   * cv::gemm on CV_32F with no transposed operand and an inner dimension of 2..4 takes the unrolled small-matrix path: the products
//     of a row are summed in FLOAT (float t = a0*b0 + a1*b1 + a2*b2), then d = (float)(t*alpha + c*beta) with alpha, beta double;
//   * every other CV_32F product (a transposed operand: A.t()*B, A*B.t()) goes through GEMMSingleMul<float,double>: the sum is
//     accumulated in DOUBLE, d = (float)(s*alpha + c*beta);
//   * Mat * scalar, Mat / scalar: convertTo with the scale narrowed to float, i.e. a float multiply (A/s multiplies by (float)(1/s));
//   * Mat::dot and cv::norm(NORM_L2) accumulate in double and return double.
#pragma once
#include <cmath>
#include "cvlite.h"

namespace ORB_SLAM3 {
namespace cvm {

struct M3 { float m[9]; float operator()(int i, int j) const { return m[3 * i + j]; } float &operator()(int i, int j) { return m[3 * i + j]; } };
struct V3 { float v[3]; float operator()(int i) const { return v[i]; } float &operator()(int i) { return v[i]; } };

inline M3 block3(const cv::Mat &T) { M3 R; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = T.at<float>(i, j); return R; }   // rowRange(0,3).colRange(0,3)
inline V3 col3(const cv::Mat &T, int c = 3) { V3 t; for (int i = 0; i < 3; i++) t(i) = T.at<float>(i, c); return t; }                          // rowRange(0,3).col(c)
inline V3 vec3(const cv::Mat &x) { V3 t; for (int i = 0; i < 3; i++) t(i) = x.at<float>(i); return t; }
inline cv::Mat to_mat(const V3 &x) { cv::Mat m(3, 1, CV_32F); for (int i = 0; i < 3; i++) m.at<float>(i) = x(i); return m; }
inline M3 transpose(const M3 &A) { M3 B; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) B(i, j) = A(j, i); return B; }

// A * B (+ C), no transposes: the small-matrix path (float sums)
inline M3 mul(const M3 &A, const M3 &B, double alpha = 1.0)
{
    M3 D;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        const float t = A(i, 0) * B(0, j) + A(i, 1) * B(1, j) + A(i, 2) * B(2, j);
        D(i, j) = (float)(t * alpha);
    }
    return D;
}
inline V3 mul(const M3 &A, const V3 &x, double alpha = 1.0)
{
    V3 d;
    for (int i = 0; i < 3; i++) { const float t = A(i, 0) * x(0) + A(i, 1) * x(1) + A(i, 2) * x(2); d(i) = (float)(t * alpha); }
    return d;
}
inline V3 mul_add(const M3 &A, const V3 &x, const V3 &c, double alpha = 1.0)          // A*x + c
{
    V3 d;
    for (int i = 0; i < 3; i++) { const float t = A(i, 0) * x(0) + A(i, 1) * x(1) + A(i, 2) * x(2); d(i) = (float)(t * alpha + (double)c(i) * 1.0); }
    return d;
}
// a transposed operand: the general path (double sums).  tA / tB: use A^T / B^T
inline M3 mul_t(const M3 &A, bool tA, const M3 &B, bool tB, double alpha = 1.0)
{
    M3 D;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)(tA ? A(k, i) : A(i, k)) * (double)(tB ? B(j, k) : B(k, j));
        D(i, j) = (float)(s * alpha);
    }
    return D;
}
inline V3 mul_t(const M3 &A, const V3 &x, double alpha = 1.0)                            // A^T * x
{
    V3 d;
    for (int i = 0; i < 3; i++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)A(k, i) * (double)x(k); d(i) = (float)(s * alpha); }
    return d;
}
inline V3 mul_t_add(const M3 &A, const V3 &x, const V3 &c, double alpha = 1.0)           // A^T * x + c
{
    V3 d;
    for (int i = 0; i < 3; i++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)A(k, i) * (double)x(k); d(i) = (float)(s * alpha + (double)c(i) * 1.0); }
    return d;
}
inline M3 scale(const M3 &A, double s) { M3 D; const float f = (float)s; for (int i = 0; i < 9; i++) D.m[i] = A.m[i] * f; return D; }
inline V3 scale(const V3 &x, double s) { V3 d; const float f = (float)s; for (int i = 0; i < 3; i++) d(i) = x(i) * f; return d; }
inline V3 sub(const V3 &a, const V3 &b) { V3 d; for (int i = 0; i < 3; i++) d(i) = a(i) - b(i); return d; }
inline double dot(const V3 &a, const V3 &b) { double s = 0; for (int i = 0; i < 3; i++) s += (double)a(i) * (double)b(i); return s; }
inline double norm(const V3 &a) { double s = 0; for (int i = 0; i < 3; i++) s += (double)a(i) * (double)a(i); return std::sqrt(s); }

// cv::invert of a 3x3 CV_32F matrix (DECOMP_LU): cofactors and determinant in double, results narrowed to float (matrix.cpp / lapack.cpp)
inline M3 inv3(const M3 &S)
{
    const double d = (double)S(0, 0) * ((double)S(1, 1) * S(2, 2) - (double)S(1, 2) * S(2, 1)) - (double)S(0, 1) * ((double)S(1, 0) * S(2, 2) - (double)S(1, 2) * S(2, 0)) +
                     (double)S(0, 2) * ((double)S(1, 0) * S(2, 1) - (double)S(1, 1) * S(2, 0));
    M3 D;
    for (int i = 0; i < 9; i++) D.m[i] = 0.f;
    if (d != 0.) {
        const double id = 1. / d;
        double t[9];
        t[0] = ((double)S(1, 1) * S(2, 2) - (double)S(1, 2) * S(2, 1)) * id;
        t[1] = ((double)S(0, 2) * S(2, 1) - (double)S(0, 1) * S(2, 2)) * id;
        t[2] = ((double)S(0, 1) * S(1, 2) - (double)S(0, 2) * S(1, 1)) * id;
        t[3] = ((double)S(1, 2) * S(2, 0) - (double)S(1, 0) * S(2, 2)) * id;
        t[4] = ((double)S(0, 0) * S(2, 2) - (double)S(0, 2) * S(2, 0)) * id;
        t[5] = ((double)S(0, 2) * S(1, 0) - (double)S(0, 0) * S(1, 2)) * id;
        t[6] = ((double)S(1, 0) * S(2, 1) - (double)S(1, 1) * S(2, 0)) * id;
        t[7] = ((double)S(0, 1) * S(2, 0) - (double)S(0, 0) * S(2, 1)) * id;
        t[8] = ((double)S(0, 0) * S(1, 1) - (double)S(0, 1) * S(1, 0)) * id;
        for (int i = 0; i < 9; i++) D.m[i] = (float)t[i];
    }
    return D;
}
inline M3 skew(const V3 &v)                      // Converter / SkewSymmetricMatrix (src/CameraModels/Pinhole.cpp:146-150)
{
    M3 S;
    S(0, 0) = 0; S(0, 1) = -v(2); S(0, 2) = v(1);
    S(1, 0) = v(2); S(1, 1) = 0; S(1, 2) = -v(0);
    S(2, 0) = -v(1); S(2, 1) = v(0); S(2, 2) = 0;
    return S;
}

}  // namespace cvm
}  // namespace ORB_SLAM3
