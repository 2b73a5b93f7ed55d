/*
 * orb_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See orb_oracle.h for scope and parity status ("parity unpinned" at the OpenCV
 * boundary; pinned by the pattern sha256 + SURVEY Appendix C constants).
 *
 * Every function cites the reference lines it restates.  Reference root:
 * /root/reference ; "ORBx" below = src/ORBextractor.cc.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (no FMA contraction: the float
 * expressions of A.6/A.8 must be evaluated op-by-op, exactly like the HIP side).
 */
#include "orb_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <limits.h>

#define PATCH_SIZE 31          /* ORBx:70 */
#define HALF_PATCH_SIZE 15     /* ORBx:71 */
#define EDGE_THRESHOLD 19      /* ORBx:72 */
#define MAX_LEVELS 16

static const int8_t k_pattern[1024] = {
#include "orb_pattern.inc"
};

/* cvRound: round-half-to-even (x86 cvtss2si / cvtsd2si under the default mode). */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }

struct level_state {
    int w, h;                 /* un-padded dims */
    uint8_t *padded;          /* (w+38)*(h+38) */
    uint8_t *blur;            /* w*h or NULL */
    int ncand, cand_cap;
    int *cx, *cy, *cs;        /* pre-octree candidates */
    int nkp;
    orc_keypoint *kp;         /* post-octree, level coords */
};

struct orc_extractor {
    int nfeatures, nlevels, ini_th, min_th;
    float scale_factor;
    float scale[MAX_LEVELS], inv_scale[MAX_LEVELS], sigma2[MAX_LEVELS], inv_sigma2[MAX_LEVELS];
    int per_level[MAX_LEVELS];
    int umax[HALF_PATCH_SIZE + 1];
    struct level_state lv[MAX_LEVELS];
};

/* ------------------------------------------------------------------ A1 ctor */
/* ORBx:408-468 */
orc_extractor *orc_extractor_create(int nfeatures, float scale_factor, int nlevels,
                                    int ini_th, int min_th)
{
    if (nlevels < 1 || nlevels > MAX_LEVELS) return NULL;
    orc_extractor *e = (orc_extractor *)calloc(1, sizeof(*e));
    e->nfeatures = nfeatures; e->nlevels = nlevels; e->ini_th = ini_th; e->min_th = min_th;
    e->scale_factor = scale_factor;
    e->scale[0] = 1.0f; e->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {                   /* ORBx:417-421 */
        e->scale[i] = e->scale[i - 1] * scale_factor;
        e->sigma2[i] = e->scale[i] * e->scale[i];
    }
    for (int i = 0; i < nlevels; i++) {                   /* ORBx:425-429 */
        e->inv_scale[i] = 1.0f / e->scale[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }
    float factor = 1.0f / scale_factor;                   /* ORBx:434-445 */
    float desired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        e->per_level[l] = cv_round_f(desired);
        sum += e->per_level[l];
        desired *= factor;
    }
    e->per_level[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;

    /* umax, ORBx:453-468 */
    int v, v0;
    int vmax = (int)floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = (int)ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    return e;
}

static void level_free(struct level_state *s)
{
    free(s->padded); free(s->blur); free(s->cx); free(s->cy); free(s->cs); free(s->kp);
    memset(s, 0, sizeof(*s));
}

void orc_extractor_destroy(orc_extractor *e)
{
    if (!e) return;
    for (int l = 0; l < MAX_LEVELS; l++) level_free(&e->lv[l]);
    free(e);
}

const float *orc_scale_factors(const orc_extractor *e) { return e->scale; }
const float *orc_inv_scale_factors(const orc_extractor *e) { return e->inv_scale; }
const float *orc_level_sigma2(const orc_extractor *e) { return e->sigma2; }
const float *orc_inv_level_sigma2(const orc_extractor *e) { return e->inv_sigma2; }
const int *orc_features_per_level(const orc_extractor *e) { return e->per_level; }
const int *orc_umax(const orc_extractor *e) { return e->umax; }

/* ------------------------------------------------------------ A.3 resize */
/* cv::resize(..., INTER_LINEAR) for 8UC1, OpenCV 3.4.1 fixed-point path
 * (INTER_RESIZE_COEF_BITS = 11), called at ORBx:1165.  Un-vendored: restated
 * from SURVEY Appendix A.3. */
static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

void orc_resize_linear(const uint8_t *src, int sw, int sh, int sstride,
                       uint8_t *dst, int dw, int dh, int dstride)
{
    double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *row0 = (int *)malloc(sizeof(int) * dw), *row1 = (int *)malloc(sizeof(int) * dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short(cv_round_f((1.f - fx) * 2048));
        ialpha[2 * dx + 1] = sat_short(cv_round_f(fx * 2048));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        short b0 = sat_short(cv_round_f((1.f - fy) * 2048));
        short b1 = sat_short(cv_round_f(fy * 2048));
        int y0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
        const uint8_t *S0 = src + (size_t)y0 * sstride, *S1 = src + (size_t)y1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sx;   /* a1 == 0 whenever sx is clamped */
            row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx1] * ialpha[2 * dx + 1];
            row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx1] * ialpha[2 * dx + 1];
        }
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(row0); free(row1);
}

/* BORDER_REFLECT_101 index (cv::borderInterpolate). */
static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * n - 2 - p;
    }
    return p;
}

/* cv::copyMakeBorder(..., 19,19,19,19, BORDER_REFLECT_101) around the ROI (ORBx:1167,1172). */
static void pad_reflect101(uint8_t *padded, int w, int h)
{
    int pw = w + 2 * EDGE_THRESHOLD, ph = h + 2 * EDGE_THRESHOLD;
    for (int y = 0; y < ph; y++) {
        int sy = reflect101(y - EDGE_THRESHOLD, h) + EDGE_THRESHOLD;
        uint8_t *row = padded + (size_t)y * pw;
        const uint8_t *srow = padded + (size_t)sy * pw;
        if (sy != y) memcpy(row + EDGE_THRESHOLD, srow + EDGE_THRESHOLD, w);
        for (int x = 0; x < EDGE_THRESHOLD; x++) {
            row[x] = srow[reflect101(x - EDGE_THRESHOLD, w) + EDGE_THRESHOLD];
            row[EDGE_THRESHOLD + w + x] = srow[reflect101(w + x, w) + EDGE_THRESHOLD];
        }
    }
}

/* ------------------------------------------------------------- A.7 blur */
/* cv::GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101) 8-bit path (ORBx:1115); un-vendored,
 * restated from SURVEY Appendix A.7: integer kernel cvRound(k*256), no intermediate
 * rounding, (sum + 2^15) >> 16, saturate. */
static void gauss_kernel7_q8(int kq[7])
{
    /* cv::getGaussianKernel(7, 2.0, CV_32F) */
    float cf[7]; double sum = 0;
    double scale2x = -0.5 / (2.0 * 2.0);
    for (int i = 0; i < 7; i++) {
        double x = i - 3.0;
        cf[i] = (float)exp(scale2x * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) {
        cf[i] = (float)(cf[i] * sum);
        kq[i] = cv_round_f(cf[i] * 256.f);
    }
}

void orc_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride)
{
    int kq[7];
    gauss_kernel7_q8(kq);
    int *tmp = (int *)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -3; k <= 3; k++) s += kq[k + 3] * S[reflect101(x + k, w)];
            tmp[(size_t)y * w + x] = s;
        }
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -3; k <= 3; k++) s += kq[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
            s = (s + 32768) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(s > 255 ? 255 : s);
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------- A.4 FAST */
/* Bresenham circle r=3, cv::makeOffsets order (OpenCV fast_score.cpp). */
static const int k_circ_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int k_circ_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* cornerScore<16> (OpenCV fast_score.cpp), literal restatement incl. early-outs. */
static int corner_score16(const uint8_t *ptr, const int pixel[25], int threshold)
{
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < 25; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; j++) a = a < d[k + j] ? a : d[k + j];
        int m = a < d[k] ? a : d[k];
        a0 = a0 > m ? a0 : m;
        m = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > m ? a0 : m;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 5; j++) b = b > d[k + j] ? b : d[k + j];
        if (b >= b0) continue;
        for (int j = 6; j <= 8; j++) b = b > d[k + j] ? b : d[k + j];
        int m = b > d[k] ? b : d[k];
        b0 = b0 < m ? b0 : m;
        m = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < m ? b0 : m;
    }
    return -b0 - 1;
}

/* FAST_t<16>(img, th, nonmax=true) (OpenCV fast.cpp), called at ORBx:808-809,827-828.
 * Literal row-buffered restatement: 3-row score ring, NMS of row i-1 while scanning row i. */
int orc_fast_nms(const uint8_t *img, int w, int h, int stride, int threshold,
                 int *xs, int *ys, int *scores, int cap)
{
    const int K = 8, N = 25;
    int pixel[25], n = 0;
    for (int k = 0; k < 25; k++) pixel[k] = k_circ_dx[k % 16] + k_circ_dy[k % 16] * stride;
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    uint8_t tab[512];
    for (int i = -255; i <= 255; i++)
        tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    if (w < 7 || h < 7) return 0;
    uint8_t *buf = (uint8_t *)calloc((size_t)w * 3, 1);
    int *cp = (int *)malloc(sizeof(int) * 3 * (w + 1));
    uint8_t *bufs[3] = {buf, buf + w, buf + 2 * w};
    int *cps[3] = {cp + 1, cp + 1 + (w + 1), cp + 1 + 2 * (w + 1)};
    for (int i = 3; i < h - 2; i++) {
        const uint8_t *ptr = img + (size_t)i * stride + 3;
        uint8_t *curr = bufs[(i - 3) % 3];
        int *cornerpos = cps[(i - 3) % 3];
        memset(curr, 0, w);
        int ncorners = 0;
        if (i < h - 3) {
            for (int j = 3; j < w - 3; j++, ptr++) {
                int v = ptr[0];
                const uint8_t *t = &tab[0] - v + 255;
                int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
                d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
                d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
                d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
                d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
                d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
                if (d & 1) {
                    int vt = v - threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x < vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else count = 0;
                    }
                }
                if (d & 2) {
                    int vt = v + threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x > vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else count = 0;
                    }
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t *prev = bufs[(i - 4 + 3) % 3];
        const uint8_t *pprev = bufs[(i - 5 + 3) % 3];
        cornerpos = cps[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            int j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] &&
                score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
                if (n < cap) { xs[n] = j; ys[n] = i - 1; scores[n] = score; }
                n++;
            }
        }
    }
    free(buf); free(cp);
    return n;
}

/* Derived identity used by the HIP kernel (proved equal to the literal code by
 * tests/test_oracle_orb.py::test_fast_arc_score_identity on random + adversarial tiles):
 *   S = max( max_arcs min_{k in arc}(v - p_k), max_arcs min_{k in arc}(p_k - v) ), 16 arcs of 9;
 *   FAST-9 corner at threshold t  <=>  S > t ;  cornerScore<16>(.., t) == S - 1 for corners. */
int orc_fast_arc_score(const uint8_t *p, int stride)
{
    int v = p[0], d[16], best = -256;
    for (int k = 0; k < 16; k++) d[k] = v - p[k_circ_dx[k] + k_circ_dy[k] * stride];
    for (int s = 0; s < 16; s++) {
        int mn = 256, mx = -256;
        for (int j = 0; j < 9; j++) {
            int x = d[(s + j) & 15];
            if (x < mn) mn = x;
            if (x > mx) mx = x;
        }
        if (mn > best) best = mn;
        if (-mx > best) best = -mx;
    }
    return best;
}

/* ------------------------------------------------------ A.6 orientation */
/* cv::fastAtan2 scalar path (OpenCV mathfuncs_core), called at ORBx:101. */
float orc_fast_atan2(float y, float x)
{
    static const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* IC_Angle, ORBx:75-102. (x,y) integer pixel centre on the un-blurred level. */
float orc_ic_angle(const uint8_t *img, int stride, int x, int y, const int *umax)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)y * stride + x;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------- A.8 descriptor */
/* orb_sincos: the reference evaluates cos/sin of a float angle through the platform
 * libm (ORBx:111, std::cos(float) overload) -- not reproducible bit-for-bit across
 * libms.  Oracle and HIP kernel both use THIS fixed sequence of IEEE double
 * operations (fma/mul/add only -> identical on x86 and gfx950), then round to float:
 *   x = (double)angle_rad ; k = (int)(x*(2/pi)+0.5) ; r = fma(-k,PIO2_LO, fma(-k,PIO2_HI,x))
 *   fdlibm __kernel_sin/__kernel_cos minimax polynomials on |r| <= pi/4 ; quadrant fix-up.
 * |error| < 1e-15 => equals the correctly rounded float value except ~2^-29 of inputs. */
static void sincos_det(double x, double *s_out, double *c_out)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    int k = (int)(x * TWO_OVER_PI + 0.5);
    double dk = (double)k;
    double r = fma(-dk, PIO2_HI, x);
    r = fma(-dk, PIO2_LO, r);
    double z = r * r;
    double ps = fma(z, S6, S5); ps = fma(z, ps, S4); ps = fma(z, ps, S3); ps = fma(z, ps, S2); ps = fma(z, ps, S1);
    double s = fma(r * z, ps, r);
    double pc = fma(z, C6, C5); pc = fma(z, pc, C4); pc = fma(z, pc, C3); pc = fma(z, pc, C2); pc = fma(z, pc, C1);
    double c = fma(z * z, pc, fma(z, -0.5, 1.0));
    switch (k & 3) {
    case 0: *s_out = s; *c_out = c; break;
    case 1: *s_out = c; *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
    }
}

static const float k_factor_pi = (float)(3.1415926535897932384626433832795 / 180.f); /* ORBx:105 */

void orc_sincos_deg(float angle_deg, float *cos_out, float *sin_out)
{
    float angle = angle_deg * k_factor_pi;                /* ORBx:110 */
    double s, c;
    sincos_det((double)angle, &s, &c);
    *cos_out = (float)c; *sin_out = (float)s;             /* ORBx:111 */
}

/* computeOrbDescriptor, ORBx:106-145. */
void orc_descriptor(const uint8_t *blur, int stride, int x, int y, float angle_deg, uint8_t *desc)
{
    float a, b;
    orc_sincos_deg(angle_deg, &a, &b);
    const uint8_t *center = blur + (size_t)y * stride + x;
    const int8_t *pat = k_pattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            float x0 = (float)pat[4 * j], y0 = (float)pat[4 * j + 1];
            float x1 = (float)pat[4 * j + 2], y1 = (float)pat[4 * j + 3];
            int t0 = center[cv_round_f(x0 * b + y0 * a) * stride + cv_round_f(x0 * a - y0 * b)];
            int t1 = center[cv_round_f(x1 * b + y1 * a) * stride + cv_round_f(x1 * a - y1 * b)];
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

/* --------------------------------------------------------- A.5 octree */
struct onode {
    int x0, y0, x1, y1;       /* UL=(x0,y0) UR=(x1,y0) BL=(x0,y1) BR=(x1,y1) */
    int *keys; int nkeys;
    int no_more;
    int prev, next;           /* std::list links (arena indices, -1 = none) */
    int seq;                  /* creation counter: stands in for the node address in the
                                 reference's sort over pair<int,ExtractorNode*> (ORBx:679-683,
                                 SURVEY F5) -- documented deterministic tie-break */
};
struct olist { struct onode *a; int n, cap, head, tail, size, seq; };

static int ol_new(struct olist *L)
{
    if (L->n == L->cap) { L->cap = L->cap ? 2 * L->cap : 64; L->a = (struct onode *)realloc(L->a, sizeof(struct onode) * L->cap); }
    struct onode *nd = &L->a[L->n];
    memset(nd, 0, sizeof(*nd));
    nd->prev = nd->next = -1; nd->seq = L->seq++;
    return L->n++;
}
static void ol_push_front(struct olist *L, int i)
{
    L->a[i].prev = -1; L->a[i].next = L->head;
    if (L->head >= 0) L->a[L->head].prev = i; else L->tail = i;
    L->head = i; L->size++;
}
static void ol_push_back(struct olist *L, int i)
{
    L->a[i].next = -1; L->a[i].prev = L->tail;
    if (L->tail >= 0) L->a[L->tail].next = i; else L->head = i;
    L->tail = i; L->size++;
}
static int ol_erase(struct olist *L, int i)   /* returns next */
{
    int p = L->a[i].prev, nx = L->a[i].next;
    if (p >= 0) L->a[p].next = nx; else L->head = nx;
    if (nx >= 0) L->a[nx].prev = p; else L->tail = p;
    L->size--;
    free(L->a[i].keys); L->a[i].keys = NULL;
    return nx;
}

/* ExtractorNode::DivideNode, ORBx:479-535.  Children written to arena slots c[0..3] (n1..n4). */
static void divide_node(struct olist *L, int parent, const int *xs, const int *ys, int c[4])
{
    for (int q = 0; q < 4; q++) c[q] = ol_new(L);        /* may realloc: re-read parent after */
    struct onode *P = &L->a[parent];
    int half_x = (int)ceilf((float)(P->x1 - P->x0) / 2);
    int half_y = (int)ceilf((float)(P->y1 - P->y0) / 2);
    int mx = P->x0 + half_x, my = P->y0 + half_y;
    struct onode *n1 = &L->a[c[0]], *n2 = &L->a[c[1]], *n3 = &L->a[c[2]], *n4 = &L->a[c[3]];
    n1->x0 = P->x0; n1->y0 = P->y0; n1->x1 = mx;    n1->y1 = my;
    n2->x0 = mx;    n2->y0 = P->y0; n2->x1 = P->x1; n2->y1 = my;
    n3->x0 = P->x0; n3->y0 = my;    n3->x1 = mx;    n3->y1 = P->y1;
    n4->x0 = mx;    n4->y0 = my;    n4->x1 = P->x1; n4->y1 = P->y1;
    for (int q = 0; q < 4; q++) L->a[c[q]].keys = (int *)malloc(sizeof(int) * (P->nkeys ? P->nkeys : 1));
    for (int i = 0; i < P->nkeys; i++) {
        int k = P->keys[i];
        float px = (float)xs[k], py = (float)ys[k];
        struct onode *t;
        if (px < (float)mx) t = (py < (float)my) ? n1 : n3;
        else t = (py < (float)my) ? n2 : n4;
        t->keys[t->nkeys++] = k;
    }
    for (int q = 0; q < 4; q++) if (L->a[c[q]].nkeys == 1) L->a[c[q]].no_more = 1;
}

struct szptr { int size; int seq; int node; };
static int szptr_cmp(const void *a, const void *b)
{
    const struct szptr *A = (const struct szptr *)a, *B = (const struct szptr *)b;
    if (A->size != B->size) return A->size < B->size ? -1 : 1;
    return A->seq < B->seq ? -1 : (A->seq > B->seq ? 1 : 0);
}

/* ORBextractor::DistributeOctTree, ORBx:537-761. */
int orc_octree(const int *xs, const int *ys, const int *scores, int n,
               int min_x, int max_x, int min_y, int max_y, int N, int *keep_idx, int cap)
{
    if (n == 0) return 0;
    struct olist L; memset(&L, 0, sizeof(L)); L.head = L.tail = -1;
    const int n_ini = (int)roundf((float)(max_x - min_x) / (max_y - min_y));   /* ORBx:541 */
    const float hX = (float)(max_x - min_x) / n_ini;
    int *ini = (int *)malloc(sizeof(int) * (n_ini > 0 ? n_ini : 1));
    for (int i = 0; i < n_ini; i++) {                                           /* ORBx:550-562 */
        int id = ol_new(&L);
        struct onode *nd = &L.a[id];
        nd->x0 = (int)(hX * (float)i); nd->x1 = (int)(hX * (float)(i + 1));
        nd->y0 = 0; nd->y1 = max_y - min_y;
        nd->keys = (int *)malloc(sizeof(int) * n);
        ol_push_back(&L, id);
        ini[i] = id;
    }
    for (int i = 0; i < n; i++) {                                               /* ORBx:565-569 */
        struct onode *nd = &L.a[ini[(int)((float)xs[i] / hX)]];
        nd->keys[nd->nkeys++] = i;
    }
    for (int it = L.head; it >= 0;) {                                           /* ORBx:573-584 */
        if (L.a[it].nkeys == 1) { L.a[it].no_more = 1; it = L.a[it].next; }
        else if (L.a[it].nkeys == 0) it = ol_erase(&L, it);
        else it = L.a[it].next;
    }
    int finish = 0;
    struct szptr *vs = NULL, *vprev = NULL; int nvs = 0, vs_cap = 0, nvprev = 0, vprev_cap = 0;
#define VS_PUSH(sz_, node_) do { if (nvs == vs_cap) { vs_cap = vs_cap ? 2 * vs_cap : 256; vs = (struct szptr *)realloc(vs, sizeof(*vs) * vs_cap); } \
        vs[nvs].size = (sz_); vs[nvs].node = (node_); vs[nvs].seq = L.a[(node_)].seq; nvs++; } while (0)
    while (!finish) {                                                            /* ORBx:593-735 */
        int prev_size = L.size, n_to_expand = 0;
        nvs = 0;
        for (int it = L.head; it >= 0;) {
            if (L.a[it].no_more) { it = L.a[it].next; continue; }
            int c[4];
            divide_node(&L, it, xs, ys, c);
            for (int q = 0; q < 4; q++) {
                if (L.a[c[q]].nkeys > 0) {
                    ol_push_front(&L, c[q]);
                    if (L.a[c[q]].nkeys > 1) { n_to_expand++; VS_PUSH(L.a[c[q]].nkeys, c[q]); }
                } else { free(L.a[c[q]].keys); L.a[c[q]].keys = NULL; }
            }
            it = ol_erase(&L, it);
        }
        if (L.size >= N || L.size == prev_size) finish = 1;                      /* ORBx:661 */
        else if (L.size + n_to_expand * 3 > N) {                                 /* ORBx:665 */
            while (!finish) {
                prev_size = L.size;
                if (nvs > vprev_cap) { vprev_cap = nvs; vprev = (struct szptr *)realloc(vprev, sizeof(*vprev) * vprev_cap); }
                memcpy(vprev, vs, sizeof(*vs) * nvs); nvprev = nvs; nvs = 0;
                qsort(vprev, nvprev, sizeof(*vprev), szptr_cmp);                 /* ORBx:682 */
                for (int j = nvprev - 1; j >= 0; j--) {
                    int c[4], node = vprev[j].node;
                    divide_node(&L, node, xs, ys, c);
                    for (int q = 0; q < 4; q++) {
                        if (L.a[c[q]].nkeys > 0) {
                            ol_push_front(&L, c[q]);
                            if (L.a[c[q]].nkeys > 1) VS_PUSH(L.a[c[q]].nkeys, c[q]);
                        } else { free(L.a[c[q]].keys); L.a[c[q]].keys = NULL; }
                    }
                    ol_erase(&L, node);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prev_size) finish = 1;
            }
        }
    }
#undef VS_PUSH
    int out = 0;                                                                 /* ORBx:739-758 */
    for (int it = L.head; it >= 0; it = L.a[it].next) {
        struct onode *nd = &L.a[it];
        int best = nd->keys[0];
        float max_resp = (float)scores[best];
        for (int k = 1; k < nd->nkeys; k++)
            if ((float)scores[nd->keys[k]] > max_resp) { best = nd->keys[k]; max_resp = (float)scores[best]; }
        if (out < cap) keep_idx[out] = best;
        out++;
    }
    for (int i = 0; i < L.n; i++) free(L.a[i].keys);
    free(L.a); free(ini); free(vs); free(vprev);
    return out;
}

/* ------------------------------------------------ A.4 cell grid + A.2 pyramid */
void orc_cell_grid(int lw, int lh, int *ncols, int *nrows, int *wcell, int *hcell)
{
    const float W = 30;                                   /* ORBx:767 */
    const int min_b = EDGE_THRESHOLD - 3;
    const float width = (float)((lw - EDGE_THRESHOLD + 3) - min_b);
    const float height = (float)((lh - EDGE_THRESHOLD + 3) - min_b);
    *ncols = (int)(width / W); *nrows = (int)(height / W);
    *wcell = (int)ceilf(width / *ncols); *hcell = (int)ceilf(height / *nrows);
}

/* ORBextractor::ComputePyramid, ORBx:1152-1177. */
static void compute_pyramid(orc_extractor *e, const uint8_t *img, int w, int h, int stride)
{
    for (int l = 0; l < e->nlevels; l++) {
        struct level_state *s = &e->lv[l];
        float scale = e->inv_scale[l];
        int lw = cv_round_f((float)w * scale), lh = cv_round_f((float)h * scale);
        level_free(s);
        s->w = lw; s->h = lh;
        int pw = lw + 2 * EDGE_THRESHOLD, ph = lh + 2 * EDGE_THRESHOLD;
        s->padded = (uint8_t *)malloc((size_t)pw * ph);
        uint8_t *roi = s->padded + (size_t)EDGE_THRESHOLD * pw + EDGE_THRESHOLD;
        if (l == 0) {
            for (int y = 0; y < lh; y++) memcpy(roi + (size_t)y * pw, img + (size_t)y * stride, lw);
        } else {
            struct level_state *p = &e->lv[l - 1];
            int ppw = p->w + 2 * EDGE_THRESHOLD;
            orc_resize_linear(p->padded + (size_t)EDGE_THRESHOLD * ppw + EDGE_THRESHOLD, p->w, p->h, ppw,
                              roi, lw, lh, pw);
        }
        pad_reflect101(s->padded, lw, lh);
    }
}

/* ORBextractor::ComputeKeyPointsOctTree, ORBx:763-878. */
static void compute_keypoints(orc_extractor *e)
{
    for (int l = 0; l < e->nlevels; l++) {
        struct level_state *s = &e->lv[l];
        int pw = s->w + 2 * EDGE_THRESHOLD;
        const uint8_t *roi = s->padded + (size_t)EDGE_THRESHOLD * pw + EDGE_THRESHOLD;
        const int min_bx = EDGE_THRESHOLD - 3, min_by = min_bx;
        const int max_bx = s->w - EDGE_THRESHOLD + 3, max_by = s->h - EDGE_THRESHOLD + 3;
        int ncols, nrows, wcell, hcell;
        orc_cell_grid(s->w, s->h, &ncols, &nrows, &wcell, &hcell);
        s->cand_cap = 4096; s->ncand = 0;
        s->cx = (int *)malloc(sizeof(int) * s->cand_cap);
        s->cy = (int *)malloc(sizeof(int) * s->cand_cap);
        s->cs = (int *)malloc(sizeof(int) * s->cand_cap);
        int tcap = 1024, *tx = (int *)malloc(sizeof(int) * tcap), *ty = (int *)malloc(sizeof(int) * tcap), *ts = (int *)malloc(sizeof(int) * tcap);
        for (int i = 0; i < nrows; i++) {
            const float ini_y = (float)(min_by + i * hcell);
            float max_y = ini_y + hcell + 6;
            if (ini_y >= max_by - 3) continue;
            if (max_y > max_by) max_y = (float)max_by;
            for (int j = 0; j < ncols; j++) {
                const float ini_x = (float)(min_bx + j * wcell);
                float max_x = ini_x + wcell + 6;
                if (ini_x >= max_bx - 6) continue;
                if (max_x > max_bx) max_x = (float)max_bx;
                const uint8_t *sub = roi + (size_t)(int)ini_y * pw + (int)ini_x;
                int cw = (int)max_x - (int)ini_x, ch = (int)max_y - (int)ini_y;
                int nk = orc_fast_nms(sub, cw, ch, pw, e->ini_th, tx, ty, ts, tcap);
                if (nk == 0) nk = orc_fast_nms(sub, cw, ch, pw, e->min_th, tx, ty, ts, tcap);
                if (nk > tcap) abort();
                for (int k = 0; k < nk; k++) {
                    if (s->ncand == s->cand_cap) {
                        s->cand_cap *= 2;
                        s->cx = (int *)realloc(s->cx, sizeof(int) * s->cand_cap);
                        s->cy = (int *)realloc(s->cy, sizeof(int) * s->cand_cap);
                        s->cs = (int *)realloc(s->cs, sizeof(int) * s->cand_cap);
                    }
                    s->cx[s->ncand] = tx[k] + j * wcell;
                    s->cy[s->ncand] = ty[k] + i * hcell;
                    s->cs[s->ncand] = ts[k];
                    s->ncand++;
                }
            }
        }
        free(tx); free(ty); free(ts);
        int *keep = (int *)malloc(sizeof(int) * (s->ncand + 1));
        int nk = orc_octree(s->cx, s->cy, s->cs, s->ncand, min_bx, max_bx, min_by, max_by,
                            e->per_level[l], keep, s->ncand + 1);
        s->nkp = nk;
        s->kp = (orc_keypoint *)malloc(sizeof(orc_keypoint) * (nk ? nk : 1));
        const int scaled_patch = (int)(PATCH_SIZE * e->scale[l]);            /* ORBx:862 */
        for (int k = 0; k < nk; k++) {
            orc_keypoint *kp = &s->kp[k];
            kp->x = (float)s->cx[keep[k]] + min_bx;                            /* ORBx:868-871 */
            kp->y = (float)s->cy[keep[k]] + min_by;
            kp->size = (float)scaled_patch;
            kp->response = (float)s->cs[keep[k]];
            kp->octave = l; kp->class_id = -1;
            kp->angle = orc_ic_angle(roi, pw, cv_round_f(kp->x), cv_round_f(kp->y), e->umax); /* ORBx:876-877 */
        }
        free(keep);
    }
}

/* ORBextractor::operator(), ORBx:1068-1150. */
int orc_extract(orc_extractor *e, const uint8_t *img, int w, int h, int stride,
                int lap0, int lap1, orc_keypoint *kp_out, uint8_t *desc_out, int cap, int *n_out)
{
    if (!img || w <= 0 || h <= 0) return -1;
    compute_pyramid(e, img, w, h, stride);
    compute_keypoints(e);
    int n = 0;
    for (int l = 0; l < e->nlevels; l++) n += e->lv[l].nkp;
    if (n_out) *n_out = n;
    if (n > cap) return -2;
    int mono = 0, stereo = n - 1;
    for (int l = 0; l < e->nlevels; l++) {
        struct level_state *s = &e->lv[l];
        if (s->nkp == 0) continue;
        int pw = s->w + 2 * EDGE_THRESHOLD;
        const uint8_t *roi = s->padded + (size_t)EDGE_THRESHOLD * pw + EDGE_THRESHOLD;
        s->blur = (uint8_t *)malloc((size_t)s->w * s->h);
        orc_gaussian_blur7(roi, s->w, s->h, pw, s->blur, s->w);               /* ORBx:1114-1115 */
        float scale = e->scale[l];
        for (int k = 0; k < s->nkp; k++) {
            orc_keypoint kp = s->kp[k];
            uint8_t d[32];
            orc_descriptor(s->blur, s->w, cv_round_f(kp.x), cv_round_f(kp.y), kp.angle, d);
            if (l != 0) { kp.x *= scale; kp.y *= scale; }                     /* ORBx:1131-1133 */
            int slot;
            if (kp.x >= (float)lap0 && kp.x <= (float)lap1) slot = stereo--;   /* ORBx:1135-1144 */
            else slot = mono++;
            kp_out[slot] = kp;
            memcpy(desc_out + (size_t)slot * 32, d, 32);
        }
    }
    return mono;
}

/* ------------------------------------------------------------------ taps */
const uint8_t *orc_pyramid_level(const orc_extractor *e, int l, int *w, int *h, int *stride)
{
    const struct level_state *s = &e->lv[l];
    int pw = s->w + 2 * EDGE_THRESHOLD;
    *w = s->w; *h = s->h; *stride = pw;
    return s->padded + (size_t)EDGE_THRESHOLD * pw + EDGE_THRESHOLD;
}
const uint8_t *orc_pyramid_level_padded(const orc_extractor *e, int l, int *w, int *h, int *stride)
{
    const struct level_state *s = &e->lv[l];
    *w = s->w + 2 * EDGE_THRESHOLD; *h = s->h + 2 * EDGE_THRESHOLD; *stride = *w;
    return s->padded;
}
const uint8_t *orc_blurred_level(const orc_extractor *e, int l, int *w, int *h)
{
    *w = e->lv[l].w; *h = e->lv[l].h;
    return e->lv[l].blur;
}
int orc_fast_candidates(const orc_extractor *e, int l, int *xs, int *ys, int *scores, int cap)
{
    const struct level_state *s = &e->lv[l];
    for (int i = 0; i < s->ncand && i < cap; i++) { xs[i] = s->cx[i]; ys[i] = s->cy[i]; scores[i] = s->cs[i]; }
    return s->ncand;
}
int orc_level_keypoints(const orc_extractor *e, int l, orc_keypoint *out, int cap)
{
    const struct level_state *s = &e->lv[l];
    for (int i = 0; i < s->nkp && i < cap; i++) out[i] = s->kp[i];
    return s->nkp;
}
