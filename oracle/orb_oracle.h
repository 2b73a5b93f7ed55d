/*
 * orb_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's ORB front-end
 *   /root/reference/src/ORBextractor.cc  (+ the OpenCV 3.4.1 primitives it calls,
 *   restated from SURVEY.md Appendix A because OpenCV is un-vendored and absent).
 *
 * PARITY STATUS: "parity unpinned" at the OpenCV boundary -- the reference holds
 * no tests / golden vectors for this path and cannot be
 * compiled here (needs OpenCV).  What IS pinned: the rBRIEF pattern table
 * (sha256), Appendix-C derived constants (level dims, per-level quotas, umax,
 * cell grids), and the SWAR Hamming identity.  See tests/test_oracle_orb.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library.  The product (orb-slam3-mac_amd/) never links or loads it.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

typedef struct orc_extractor orc_extractor;

/* ORBextractor::ORBextractor  (ORBextractor.cc:408-468) */
orc_extractor *orc_extractor_create(int nfeatures, float scale_factor, int nlevels,
                                    int ini_th_fast, int min_th_fast);
void orc_extractor_destroy(orc_extractor *e);

/* Constructor tables (A1). n = nlevels (or 16 for umax). */
const float *orc_scale_factors(const orc_extractor *e);
const float *orc_inv_scale_factors(const orc_extractor *e);
const float *orc_level_sigma2(const orc_extractor *e);
const float *orc_inv_level_sigma2(const orc_extractor *e);
const int *orc_features_per_level(const orc_extractor *e);
const int *orc_umax(const orc_extractor *e);

/* ORBextractor::operator()  (ORBextractor.cc:1068-1150).
 * Returns monoIndex, or -1 if the image is empty.  *n_out = total keypoints.
 * kp_out / desc_out (32 B rows) must hold `cap` entries; -2 if cap too small. */
int orc_extract(orc_extractor *e, const uint8_t *img, int w, int h, int stride,
                int lap0, int lap1, orc_keypoint *kp_out, uint8_t *desc_out, int cap,
                int *n_out);

/* ---- stage taps valid after orc_extract (for HIP-vs-oracle parity per stage) ---- */
/* Un-padded pyramid level (ROI view inside the reflect-101 padded buffer). */
const uint8_t *orc_pyramid_level(const orc_extractor *e, int level, int *w, int *h, int *stride);
/* Whole padded buffer (w+38 x h+38), what mvImagePyramid's parent Mat holds. */
const uint8_t *orc_pyramid_level_padded(const orc_extractor *e, int level, int *w, int *h, int *stride);
/* 7x7 sigma=2 blurred compact copy (stride == w). NULL if level had no keypoints. */
const uint8_t *orc_blurred_level(const orc_extractor *e, int level, int *w, int *h);
/* Pre-octree FAST candidate list of a level, in reference emission order
 * (cell row-major, then row-major inside the cell).  x,y relative to minBorder (16,16). */
int orc_fast_candidates(const orc_extractor *e, int level, int *xs, int *ys, int *scores, int cap);
/* Post-octree keypoints of a level (level coords, angle set, NOT yet scaled), list order. */
int orc_level_keypoints(const orc_extractor *e, int level, orc_keypoint *out, int cap);
/* FAST cell grid of a level (A.4): nCols,nRows,wCell,hCell. */
void orc_cell_grid(int lw, int lh, int *ncols, int *nrows, int *wcell, int *hcell);

/* ---- stand-alone primitives (unit-testable) ---- */
/* cv::resize INTER_LINEAR 8UC1 (Appendix A.3). */
void orc_resize_linear(const uint8_t *src, int sw, int sh, int sstride,
                       uint8_t *dst, int dw, int dh, int dstride);
/* cv::GaussianBlur 7x7 sigma 2, BORDER_REFLECT_101, 8-bit fixed point (Appendix A.7). */
void orc_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
/* cv::FAST(img, th, nms=true) on one sub-image (Appendix A.4).  Returns count. */
int orc_fast_nms(const uint8_t *img, int w, int h, int stride, int threshold,
                 int *xs, int *ys, int *scores, int cap);
/* Per-pixel FAST-9/16 arc score S = max over the 16 9-arcs of min(v-p) / min(p-v);
 * corner at threshold t  <=>  S > t ; cornerScore == S-1.  (derived identity, see .c) */
int orc_fast_arc_score(const uint8_t *p, int stride);
/* cv::fastAtan2 (Appendix A.6), degrees. */
float orc_fast_atan2(float y, float x);
/* IC_Angle (ORBextractor.cc:75-102). */
float orc_ic_angle(const uint8_t *img, int stride, int x, int y, const int *umax);
/* The deterministic sin/cos both sides use for the rBRIEF steering (DESIGN.md "orb_sincos"). */
void orc_sincos_deg(float angle_deg, float *cos_out, float *sin_out);
/* computeOrbDescriptor (ORBextractor.cc:106-145) on a blurred image. */
void orc_descriptor(const uint8_t *blur, int stride, int x, int y, float angle_deg, uint8_t *desc32);
/* DistributeOctTree (ORBextractor.cc:537-761) with the documented deterministic tie-break.
 * In: n candidates (x,y relative to minBorder, score). Out: indices of kept candidates in
 * list order.  Returns count. */
int orc_octree(const int *xs, const int *ys, const int *scores, int n,
               int min_x, int max_x, int min_y, int max_y, int n_features,
               int *keep_idx, int cap);

#ifdef __cplusplus
}
#endif
#endif
