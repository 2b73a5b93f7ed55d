/*
 * orbhip.h -- C ABI of the MI355X-native ORB front-end + local-BA solver.
 *
 * This is the drop-in boundary: plain pointers and sizes,
 * no C++ / torch / OpenCV types.  Every entry point cites the reference interface
 * it replaces (paths relative to the reference repo root).  Status codes: 0 = ok,
 * negative = error (never throws; see ORBHIP_E_*).  All entry points are thread-safe
 * across distinct contexts; one context = one HIP stream + its own device scratch
 * (mirrors "one ORBextractor instance per thread", src/Frame.cc:109-110).
 *
 * The library REQUIRES a gfx950 device: there is no CPU fallback.  orbhip_ctx_create
 * fails with ORBHIP_E_NODEVICE when no HIP device is present.
 */
#ifndef ORBHIP_H
#define ORBHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORBHIP_OK 0
#define ORBHIP_E_BADARG (-1)
#define ORBHIP_E_NODEVICE (-2)
#define ORBHIP_E_HIP (-3)        /* a HIP runtime call failed; see orbhip_last_error() */
#define ORBHIP_E_CAPACITY (-4)   /* a device-side list overflowed its reserved capacity */
#define ORBHIP_E_ABORTED (-5)    /* BA: stop flag raised (Optimizer.cc:2041-2043) */
#define ORBHIP_E_NOTSPD (-6)     /* BA: reduced system not SPD on every LM trial */
#define ORBHIP_E_EMPTY (-7)      /* extractor: empty image (ORBextractor.cc:1072-1073 returns -1) */

/* cv::KeyPoint layout (28 B): what ORBextractor::operator() fills (ORBextractor.cc:1100). */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbhip_keypoint;

typedef struct orbhip_ctx orbhip_ctx;
typedef struct orbhip_extractor orbhip_extractor;

const char *orbhip_version(void);
const char *orbhip_last_error(void);

/* Device context.  `stream` may be NULL (the library creates its own non-blocking
 * stream) or an existing hipStream_t (e.g. torch's current stream) which is borrowed. */
int orbhip_ctx_create(int device, void *stream, orbhip_ctx **out);
void orbhip_ctx_destroy(orbhip_ctx *ctx);
int orbhip_ctx_synchronize(orbhip_ctx *ctx);
/* Stream order across two contexts of one device, without a host synchronisation: everything submitted to `other` so far completes
 * before anything submitted to `ctx` after this call starts (the reference runs its left / right extractors and its Tracking /
 * LocalMapping / LoopClosing work on separate host threads, src/Frame.cc:109-112; contexts are this library's unit of concurrency). */
int orbhip_ctx_wait_for(orbhip_ctx *ctx, orbhip_ctx *other);
/* Synchronises, then returns and clears the context's sticky device-side error word
 * (ORBHIP_E_CAPACITY when a matcher kernel met more keypoints than it can hold). */
int orbhip_ctx_check_status(orbhip_ctx *ctx);
void *orbhip_ctx_stream(orbhip_ctx *ctx);

/* ------------------------------------------------------------------ ORB extractor */
/* Replaces ORBextractor::ORBextractor(int nfeatures, float scaleFactor, int nlevels,
 *   int iniThFAST, int minThFAST)            include/ORBextractor.h:54, src/ORBextractor.cc:408 */
int orbhip_extractor_create(orbhip_ctx *ctx, int nfeatures, float scale_factor, int nlevels,
                            int ini_th_fast, int min_th_fast, orbhip_extractor **out);
void orbhip_extractor_destroy(orbhip_extractor *ext);

/* GetLevels / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares                 include/ORBextractor.h:61-79
 * which: 0 scale, 1 inv scale, 2 sigma2, 3 inv sigma2.  out[nlevels]. */
int orbhip_extractor_levels(const orbhip_extractor *ext);
int orbhip_extractor_table(const orbhip_extractor *ext, int which, float *out);
/* mnFeaturesPerLevel (ORBextractor.cc:434-445) and umax (ORBextractor.cc:453-468). */
int orbhip_extractor_features_per_level(const orbhip_extractor *ext, int *out);
int orbhip_extractor_umax(const orbhip_extractor *ext, int *out16);

/* Reserve device memory for batches of up to max_batch frames of width x height.
 * Called implicitly by the extract calls; explicit call keeps allocation out of timed code. */
int orbhip_extractor_reserve(orbhip_extractor *ext, int width, int height, int max_batch);
/* Upper bound of keypoints per frame (row capacity of the output arrays). */
int orbhip_extractor_max_keypoints(const orbhip_extractor *ext);

/* Batched ORBextractor::operator()           include/ORBextractor.h:57-59, ORBextractor.cc:1068
 * d_images: DEVICE pointer, u8, frame f at d_images + f*frame_stride, rows at row_stride.
 * Results stay on the device (see orbhip_extractor_results).  lap0/lap1 = vLappingArea.
 * Asynchronous on the context's stream. */
int orbhip_extract_batch_device(orbhip_extractor *ext, const uint8_t *d_images, int width,
                                int height, size_t row_stride, size_t frame_stride, int batch,
                                int lap0, int lap1);
/* Device result arrays of the last batch: kp[batch][max_kp], desc[batch][max_kp][32],
 * count[batch] (nkeypoints), mono_index[batch] (operator()'s return value). */
int orbhip_extractor_results(orbhip_extractor *ext, orbhip_keypoint **d_kp, uint8_t **d_desc,
                             int32_t **d_count, int32_t **d_mono_index);
/* Host convenience == one ORBextractor::operator() call per frame: H2D, extract, D2H, sync.
 * kp_out[batch][cap], desc_out[batch][cap][32], count_out[batch], mono_out[batch].
 * Returns ORBHIP_E_EMPTY for an empty image, ORBHIP_E_CAPACITY if cap < keypoints. */
int orbhip_extract_batch_host(orbhip_extractor *ext, const uint8_t *h_images, int width, int height,
                              size_t row_stride, size_t frame_stride, int batch, int lap0, int lap1,
                              orbhip_keypoint *kp_out, uint8_t *desc_out, int cap,
                              int32_t *count_out, int32_t *mono_out);
/* The same call without the copy into caller arrays: *kp_view / *desc_view point at the extractor's page-locked mirror of the device
 * result arrays (row f at kp_view + f * row_capacity, desc_view + f * row_capacity * 32; count_view[f], mono_view[f]), valid until the
 * extractor's next extract call.  What host/ORBextractor.cc copies into the caller's vector<cv::KeyPoint> / cv::Mat. */
int orbhip_extract_batch_host_view(orbhip_extractor *ext, const uint8_t *h_images, int width, int height, size_t row_stride,
                                   size_t frame_stride, int batch, int lap0, int lap1, const orbhip_keypoint **kp_view,
                                   const uint8_t **desc_view, int *row_capacity, const int32_t **count_view, const int32_t **mono_view);
/* Which kernel blurs a batch of `batch` frames on this extractor's geometry: 0 the LDS tile kernel (small batches), 1 the row-streaming
 * kernel, 2 the matrix-core one (images of up to 320 K pixels (VGA) in batches of 128 frames or more; ORBHIP_BLUR_MFMA=0 / 1 overrides).
 * All three are bit-exact; the choice is a measured one. */
int orbhip_extractor_blur_kernel(const orbhip_extractor *ext, int batch);

/* Frame `frame` of the extractor's latest HOST extract call as it still sits on the device (d_kp, d_desc: device pointers into the result
 * arrays) together with its page-locked host mirror and a counter that changes with every extract call.  The *_host_resident matcher
 * entry points below take d_kp / d_desc as their train side, so that Tracking's SearchByProjection(CurrentFrame, ...) right after
 * Frame::ExtractORB (src/Frame.cc:410-417 -> src/Tracking.cc:1911) uploads queries only.  ORBHIP_E_BADARG when the latest call was a
 * device call or failed (no mirror). */
int orbhip_extractor_last_frame(orbhip_extractor *ext, int frame, const orbhip_keypoint **d_kp, const uint8_t **d_desc,
                                const orbhip_keypoint **h_kp_view, const uint8_t **h_desc_view, int32_t *count, unsigned long long *generation);

/* mvImagePyramid[level] of frame `frame` (include/ORBextractor.h:83; read by
 * src/Frame.cc:809,899,913,918).  padded != 0 -> the (w+38)x(h+38) reflect-101 parent
 * buffer, else the w x h ROI.  Synchronous D2H. */
int orbhip_extractor_level_dims(const orbhip_extractor *ext, int level, int *w, int *h);
int orbhip_extractor_get_pyramid_level(orbhip_extractor *ext, int frame, int level, int padded,
                                       uint8_t *h_out, size_t out_stride);
/* All levels of frame `frame` at once, as the reference holds them after operator(): levels_out[l] receives the
 * (w_l+38) x (h_l+38) reflect-101 padded parent of level l (ORBextractor.cc:1160-1173) with row stride strides[l] >= w_l+38;
 * mvImagePyramid[l] is its ROI at (19,19).  One device-to-host pass and one synchronisation for the whole pyramid. */
int orbhip_extractor_get_pyramid_padded(orbhip_extractor *ext, int frame, uint8_t *const *levels_out, const size_t *strides);
/* Parity taps (test hooks; synchronous D2H of intermediate stages of the last batch). */
int orbhip_extractor_get_blurred_level(orbhip_extractor *ext, int frame, int level,
                                       uint8_t *h_out, size_t out_stride);
/* Pre-octree FAST candidates in reference emission order; x,y relative to (16,16). */
int orbhip_extractor_get_fast_candidates(orbhip_extractor *ext, int frame, int level,
                                         int32_t *xs, int32_t *ys, int32_t *scores, int cap,
                                         int32_t *n_out);
/* Post-octree keypoints of one level (level coords, angle set, not scaled), list order. */
int orbhip_extractor_get_level_keypoints(orbhip_extractor *ext, int frame, int level,
                                         orbhip_keypoint *out, int cap, int32_t *n_out);

/* Small batches (the per-frame real-time case, BASELINE config #1) are launch bound: 19 kernel launches per
 * extract call.  With graph mode on, the launches of one orbhip_extract_batch_device / _host call are captured once
 * per (image size, batch, lapping) and replayed as ONE hipGraph; the input is then always staged into the
 * extractor's own level-0 buffer so the captured kernels see fixed addresses.  Results are identical.
 * Ignored while stage profiling is on. */
int orbhip_extractor_set_graph_mode(orbhip_extractor *ext, int enable);

/* Per-stage device time (ms), averaged over the extract calls made since the previous query
 * (at most the 32 most recent), measured with hipEvents recorded on the context's stream
 * around each stage's launches while profiling is enabled.  Stage ids: ORBHIP_STAGE_*. */
#define ORBHIP_STAGE_PYRAMID 0      /* k_resize, nlevels-1 launches */
#define ORBHIP_STAGE_FAST_CELLS 1   /* k_fast_cells (per-cell FAST score + NMS + two-threshold retry, all levels), 1 launch */
#define ORBHIP_STAGE_BLUR 2         /* k_blur (7x7 Gaussian, all levels), 1 launch */
#define ORBHIP_STAGE_OCTREE 3       /* k_octree, 1 launch */
#define ORBHIP_STAGE_DESC 4         /* k_orient_desc, 1 launch */
#define ORBHIP_STAGE_ASSEMBLE 5     /* k_assemble, 1 launch */
#define ORBHIP_STAGE_COUNT 6
int orbhip_extractor_set_profiling(orbhip_extractor *ext, int enable);
int orbhip_extractor_stage_ms(orbhip_extractor *ext, float *ms_out /*[ORBHIP_STAGE_COUNT]*/);

/* ------------------------------------------------------------------ ORB matcher */
/* ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&)
 *                                            include/ORBmatcher.h:45, src/ORBmatcher.cc:2353
 * Host-callable scalar (two 32-byte descriptors). */
int orbhip_descriptor_distance(const uint8_t *a32, const uint8_t *b32);

/* Batched all-pairs 2-NN == cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) as used by
 * Frame::ComputeStereoFishEyeMatches          src/Frame.cc:43,1146-1153
 * For pair p: queries d_descA + p*strideA (nA[p] rows), train d_descB + p*strideB (nB[p] rows).
 * Outputs per query row: idx[2], dist[2] (ascending; strict <, lowest index wins ties;
 * idx -1 / dist INT_MAX when fewer than k train rows) and ratio-test flag
 * (float)d0 < (float)d1*ratio evaluated in double as in Frame.cc:1153 (ratio there: 0.7).
 * Row q of pair p lands at [p*max_n + q].  All pointers DEVICE.  max_n <= 65535 (else ORBHIP_E_BADARG). */
int orbhip_match_bf2nn_device(orbhip_ctx *ctx, const uint8_t *d_descA, const int32_t *d_nA,
                              size_t strideA, const uint8_t *d_descB, const int32_t *d_nB,
                              size_t strideB, int pairs, int max_n, double ratio,
                              int32_t *d_idx2, int32_t *d_dist2, uint8_t *d_accept);

/* Batched ORBmatcher::SearchForInitialization(F1,F2,vbPrevMatched,vnMatches12,windowSize)
 *                                            include/ORBmatcher.h:66, src/ORBmatcher.cc:710-825
 * including Frame::AssignFeaturesToGrid / GetFeaturesInArea semantics
 *                                            src/Frame.cc:377-408,645-726
 * Pair p matches frame A_p against frame B_p.  kp arrays are the extractor's device
 * outputs (row capacity max_n).  d_prev_matched[pairs][max_n][2] (float x,y) is in/out
 * (vbPrevMatched); d_matches12[pairs][max_n] out; d_nmatches[pairs] out (return value).
 * Grid bounds = image bounds (mnMinX..mnMaxX of a distortion-free camera).  At most 8192
 * keypoints and 4096 octave-0 keypoints per frame -- the monocular-initialisation extractor runs
 * 5 x nFeatures (src/Tracking.cc:210) -- else the context's status word is set
 * (orbhip_ctx_check_status).  Scratch (LDS) is sized from max_n. */
int orbhip_search_for_initialization_device(orbhip_ctx *ctx,
        const orbhip_keypoint *d_kpA, const uint8_t *d_descA, const int32_t *d_nA,
        const orbhip_keypoint *d_kpB, const uint8_t *d_descB, const int32_t *d_nB,
        int pairs, int max_n, size_t frame_stride_kp /*entries*/, float min_x, float min_y,
        float max_x, float max_y, int window_size, float nn_ratio, int check_orientation,
        float *d_prev_matched, int32_t *d_matches12, int32_t *d_nmatches);

/* vbPrevMatched[i] = mvKeysUn[i].pt for `frames` frames (src/Tracking.cc:1497-1499), on the device:
 * d_prev_matched[f][i] = (kp[f][i].x, kp[f][i].y), rows of max_n entries. */
int orbhip_prev_matched_init_device(orbhip_ctx *ctx, const orbhip_keypoint *d_kp, size_t frame_stride_kp,
                                    int frames, int max_n, float *d_prev_matched);

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)
 * (src/ORBmatcher.cc:1965-2181, CurrentFrame.Nleft == -1: monocular / rectified stereo) -- the matcher of
 * Tracking::TrackWithMotionModel (src/Tracking.cc:2683-2694).  The projection of the last frame's map
 * points (ORBmatcher.cc:1992-2008) stays host geometry; each surviving point arrives as one query:
 *   u, v       uv = CurrentFrame.mpCamera->project(x3Dc)                       (ORBmatcher.cc:2003)
 *   radius     th * CurrentFrame.mvScaleFactors[nLastOctave]                   (:2014)
 *   ur         uv.x - CurrentFrame.mbf * invzc                                 (:2043)
 *   angle      angle of the last frame's (undistorted) keypoint                (:2067-2073)
 *   min_level, max_level   level arguments of the GetFeaturesInArea call picked by bForward / bBackward
 *              (:2018-2023): (oct,-1) | (0,oct) | (oct-1,oct+1); -1 = open end (src/Frame.cc:676)
 *   has_obs    pMP->Observations() > 0: a keypoint claimed by such a point is no candidate for later
 *              queries (:2037-2039)
 * Train side = CurrentFrame: mvKeysUn (x, y, octave, angle of orbhip_keypoint), descriptors, optional
 * mvuRight (NULL = monocular, all -1), image bounds mnMinX..mnMaxY for the 64x48 grid (src/Frame.cc:377-408,
 * 645-726).  d_train_match [pairs][max_n] is CurrentFrame.mvpMapPoints, in/out: entry -1 = free, anything else
 * = already holds a map point with observations (returned as -2, never matched); on return a value >= 0 is the
 * index of the query that owns the keypoint.  d_nmatches[pair] = the function's return value (rotation
 * consistency, :2156-2178, applied when check_orientation).  th_high = ORBmatcher::TH_HIGH = 100 (:40).
 * At most 2048 queries and 2048 keypoints per pair (else the context status becomes ORBHIP_E_CAPACITY).
 * Pair p reads queries at d_q + p*max_q, keypoints at d_kp + p*frame_stride_kp.  All pointers DEVICE. */
typedef struct orbhip_proj_query {
    float u, v, radius, ur, angle;
    int32_t min_level, max_level, has_obs;
} orbhip_proj_query;
int orbhip_search_by_projection_device(orbhip_ctx *ctx, const orbhip_proj_query *d_q, const uint8_t *d_desc_q,
                                       const int32_t *d_nq, int max_q, const orbhip_keypoint *d_kp,
                                       const uint8_t *d_desc, const float *d_u_right, const int32_t *d_n, int max_n,
                                       size_t frame_stride_kp, int pairs, float min_x, float min_y, float max_x,
                                       float max_y, int th_high, int check_orientation, int32_t *d_train_match,
                                       int32_t *d_nmatches);

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, bFarPoints, thFarPoints)
 * (src/ORBmatcher.cc:48-218, F.Nleft == -1) -- the matcher of Tracking::SearchLocalPoints (src/Tracking.cc:3096).
 * Same layout and claim rule as orbhip_search_by_projection_device; one query per map point that passes :57-67:
 * u, v = mTrackProjX/Y; radius = RadiusByViewingCos(mTrackViewCos) [* th] * F.mvScaleFactors[nPredictedLevel]
 * (:72-79); (min_level, max_level) = (nPredictedLevel-1, nPredictedLevel); ur = mTrackProjXR; angle unused.
 * A match needs best <= th_high and, when best and second best lie in the same octave, best <= nn_ratio * second
 * (mfNNratio, :131-137).  No rotation histogram. */
int orbhip_search_local_map_device(orbhip_ctx *ctx, const orbhip_proj_query *d_q, const uint8_t *d_desc_q,
                                   const int32_t *d_nq, int max_q, const orbhip_keypoint *d_kp,
                                   const uint8_t *d_desc, const float *d_u_right, const int32_t *d_n, int max_n,
                                   size_t frame_stride_kp, int pairs, float min_x, float min_y, float max_x,
                                   float max_y, int th_high, float nn_ratio, int32_t *d_train_match,
                                   int32_t *d_nmatches);

/* The two searches above on frames of a two-camera rig (CurrentFrame.Nleft != -1: TUM-VI stereo-fisheye, src/Frame.cc:1034-1126) --
 * ORBmatcher.cc:2013-2016 + :2089-2153 (mode 0: SearchByProjection(CurrentFrame, LastFrame, ...)) and :113-122 + :136-214 (mode 1:
 * SearchByProjection(F, vpMapPoints, ...)).  The train side is the whole frame: keypoints [0, d_nleft[pair]) = F.mvKeys (left camera),
 * [d_nleft[pair], d_n[pair]) = F.mvKeysRight, descriptors likewise (F.mDescriptors rows, ORBmatcher.cc:176); each camera has its own
 * grid (mGrid / mGridRight, src/Frame.cc:395-405) and there is no uRight gate.  A query searches the LEFT camera's grid, or the RIGHT
 * one's when bit 1 of has_obs is set (bit 0 keeps its meaning); the caller emits a point's left query (u, v = projection in the left
 * camera) and then its right query (projection in the right camera: :2090-2092, or mTrackProjXR / YR with radius
 * RadiusByViewingCos(mTrackViewCosR) * scale[mnTrackScaleLevelR], :151-156) in the reference's order, so the claim rule sees the same
 * sequence.  d_mirror [pairs][max_n] (mode 1; may be NULL): for every keypoint the frame-wide index of the same point's keypoint in the
 * other camera -- mvLeftToRightMatch[i] + Nleft for i < Nleft, mvRightToLeftMatch[i - Nleft] otherwise -- or -1: a match also assigns
 * that keypoint and counts twice (:142-146, :203-207).  Outputs as in the single-camera calls (d_train_match indexed frame-wide). */
int orbhip_search_by_projection_rig_device(orbhip_ctx *ctx, int mode, const orbhip_proj_query *d_q, const uint8_t *d_desc_q,
                                           const int32_t *d_nq, int max_q, const orbhip_keypoint *d_kp, const uint8_t *d_desc,
                                           const int32_t *d_n, const int32_t *d_nleft, const int32_t *d_mirror, int max_n,
                                           size_t frame_stride_kp, int pairs, float min_x, float min_y, float max_x, float max_y,
                                           int th_high, float nn_ratio, int check_orientation, int32_t *d_train_match,
                                           int32_t *d_nmatches);

/* Host-pointer forms of the two calls above for ONE frame (what an ORBmatcher method with the reference's signature needs:
 * host/ORBmatcher.cc): upload into the context's arena, run the same kernel, download, synchronise.  mode 0 =
 * orbhip_search_by_projection_device (nn_ratio unused), 1 = orbhip_search_local_map_device (check_orientation unused).
 * u_right may be NULL (monocular).  train_match_inout [n] as d_train_match.  All pointers HOST. */
int orbhip_search_by_projection_host(orbhip_ctx *ctx, int mode, const orbhip_proj_query *q, const uint8_t *desc_q, int nq,
                                     const orbhip_keypoint *kp, const uint8_t *desc, const float *u_right, int n,
                                     float min_x, float min_y, float max_x, float max_y, int th_high, float nn_ratio,
                                     int check_orientation, int32_t *train_match_inout, int32_t *nmatches_out);
/* ... and of orbhip_search_by_projection_rig_device: kp / desc are the frame's left | right keypoints, nleft = Nleft, mirror [n] or NULL. */
int orbhip_search_by_projection_rig_host(orbhip_ctx *ctx, int mode, const orbhip_proj_query *q, const uint8_t *desc_q, int nq,
                                         const orbhip_keypoint *kp, const uint8_t *desc, int n, int nleft, const int32_t *mirror,
                                         float min_x, float min_y, float max_x, float max_y, int th_high, float nn_ratio,
                                         int check_orientation, int32_t *train_match_inout, int32_t *nmatches_out);
/* Host-pointer form of orbhip_search_for_initialization_device for one frame pair: prev_matched_inout [nA][2] is vbPrevMatched
 * (in/out), matches12_out [nA] is vnMatches12, *nmatches_out the return value.  All pointers HOST. */
int orbhip_search_for_initialization_host(orbhip_ctx *ctx, const orbhip_keypoint *kpA, const uint8_t *descA, int nA,
                                          const orbhip_keypoint *kpB, const uint8_t *descB, int nB, float min_x, float min_y,
                                          float max_x, float max_y, int window_size, float nn_ratio, int check_orientation,
                                          float *prev_matched_inout, int32_t *matches12_out, int32_t *nmatches_out);
/* The same host-pointer calls with the TRAIN side already on the device: d_kp / d_desc (d_kpB / d_descB) are DEVICE pointers to n
 * keypoints and descriptors -- the result arrays of the extraction that produced the frame (orbhip_extractor_last_frame).  This is
 * the shape of Tracking's calls: SearchByProjection(mCurrentFrame, mLastFrame, ...) (src/Tracking.cc:1911), SearchLocalPoints'
 * SearchByProjection(mCurrentFrame, vpMapPoints, ...) (:3083) and SearchForInitialization(mInitialFrame, mCurrentFrame, ...) (:1506)
 * all search the frame whose features the extractor left on the device a moment ago; only the queries (map point projections and
 * descriptors) travel.  host/frame_cache.h decides when a Frame IS that extraction (byte comparison with the page-locked mirror).
 * d_kp may be NULL while d_desc is given: the frame's descriptors are the extraction's but its keypoints are not (mvKeysUn of a camera
 * with distortion, src/Frame.cc:738-771) -- then kp_host [n] is uploaded.  Single-camera frames only.  Everything else HOST, results
 * identical to the forms above. */
int orbhip_search_by_projection_host_resident(orbhip_ctx *ctx, int mode, const orbhip_proj_query *q, const uint8_t *desc_q, int nq,
                                              const orbhip_keypoint *kp_host, const orbhip_keypoint *d_kp, const uint8_t *d_desc, const float *u_right, int n,
                                              float min_x, float min_y, float max_x, float max_y, int th_high, float nn_ratio,
                                              int check_orientation, int32_t *train_match_inout, int32_t *nmatches_out);
int orbhip_search_for_initialization_host_resident(orbhip_ctx *ctx, const orbhip_keypoint *kpA, const uint8_t *descA, int nA,
                                                   const orbhip_keypoint *kpB_host, const orbhip_keypoint *d_kpB, const uint8_t *d_descB, int nB, float min_x, float min_y,
                                                   float max_x, float max_y, int window_size, float nn_ratio, int check_orientation,
                                                   float *prev_matched_inout, int32_t *matches12_out, int32_t *nmatches_out);

/* The search part of ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th, bRight)
 * (src/ORBmatcher.cc:1403-1613, NLeft == -1; LocalMapping::SearchInNeighbors, src/LocalMapping.cc:781-860), batched over
 * (keyframe, point set) pairs.  The per-point geometry of :1430-1497 stays with the caller; each surviving point is one
 * query: u, v = uv; radius = th * pKF->mvScaleFactors[nPredictedLevel]; ur = uv.x - bf*invz; min_level / max_level =
 * nPredictedLevel-1 / nPredictedLevel (angle, has_obs unused).  Per query: the keypoint inside the window, at an allowed
 * octave, whose reprojection error passes e2 * mvInvLevelSigma2[octave] <= 5.99 (7.8 with a right coordinate,
 * mvuRight >= 0) and whose descriptor distance is smallest (first minimum in GetFeaturesInArea order, :1527-1568).
 * d_best_idx / d_best_dist [pairs][max_q] = bestIdx (-1 = none) / bestDist (256 = none); the caller applies
 * bestDist <= TH_LOW and the Replace / AddObservation bookkeeping (:1572-1595) in order.  inv_level_sigma2: HOST array
 * of nlevels floats.  At most 2900 keypoints per keyframe (they and their descriptors are LDS-resident; more sets the
 * context status to ORBHIP_E_CAPACITY).  All other pointers DEVICE. */
int orbhip_fuse_search_device(orbhip_ctx *ctx, const orbhip_proj_query *d_q, const uint8_t *d_desc_q, const int32_t *d_nq,
                              int max_q, const orbhip_keypoint *d_kp, const uint8_t *d_desc, const float *d_u_right,
                              const int32_t *d_n, int max_n, size_t frame_stride_kp, int pairs,
                              const float *inv_level_sigma2, int nlevels, float min_x, float min_y, float max_x, float max_y,
                              int32_t *d_best_idx, int32_t *d_best_dist);

/* ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) (src/ORBmatcher.cc:273-475,
 * F.Nleft == -1) -- the matcher of Tracking::TrackReferenceKeyFrame (src/Tracking.cc:1757) and Relocalization
 * (:3290-3300), batched over (keyframe, frame) pairs.  The DBoW2 FeatureVectors (map<NodeId, vector<feature index>>,
 * pKF->mFeatVec / F.mFeatVec) arrive flattened per pair: node ids ascending [pairs][max_nodes], node_start
 * [pairs][max_nodes+1] into feat [pairs][max_n], number of nodes [pairs] (orbhip_bow_transform_device yields the node of
 * every feature).  d_kf_valid [pairs][max_n]: the keyframe's map point at that feature exists and is not bad (:297-302).
 * Inside a shared node every valid keyframe feature, in order, takes the best still unmatched frame feature if
 * best <= TH_LOW and best < nn_ratio * second (:304-366); rotation consistency (:445-470) when check_orientation.
 * d_match_f [pairs][max_n]: per frame feature the keyframe feature whose map point it receives, or -1
 * (vpMapPointMatches[j] = vpMapPointsKF[d_match_f[j]]); d_nmatches [pairs] = the return value.  At most 4096 features per
 * frame (their descriptors are LDS-resident).  All pointers DEVICE. */
int orbhip_search_by_bow_device(orbhip_ctx *ctx,
        const int32_t *d_kf_node_ids, const int32_t *d_kf_node_start, const int32_t *d_kf_feat, const int32_t *d_kf_nnodes,
        const uint8_t *d_kf_valid, const orbhip_keypoint *d_kf_kp, const uint8_t *d_kf_desc,
        const int32_t *d_f_node_ids, const int32_t *d_f_node_start, const int32_t *d_f_feat, const int32_t *d_f_nnodes,
        const orbhip_keypoint *d_f_kp, const uint8_t *d_f_desc, const int32_t *d_nF,
        int pairs, int max_nodes, int max_n, size_t frame_stride_kp, float nn_ratio, int check_orientation,
        int32_t *d_match_f, int32_t *d_nmatches);

/* orbhip_search_by_bow_device for frames of a two-camera rig (F.Nleft != -1, ORBmatcher.cc:338-359, :393-425): frame features
 * [0, d_nleft[pair]) are the left camera's, the rest the right camera's (F.mDescriptors order; keypoints concatenated mvKeys | mvKeysRight,
 * on the keyframe side too).  Every keyframe feature keeps a best / second best PER CAMERA; when the left best passes TH_LOW the left
 * match needs the ratio test and the right camera's best is taken whenever it passes TH_LOW (the reference's "|| true").  Everything
 * else as orbhip_search_by_bow_device. */
int orbhip_search_by_bow_rig_device(orbhip_ctx *ctx,
        const int32_t *d_kf_node_ids, const int32_t *d_kf_node_start, const int32_t *d_kf_feat, const int32_t *d_kf_nnodes,
        const uint8_t *d_kf_valid, const orbhip_keypoint *d_kf_kp, const uint8_t *d_kf_desc,
        const int32_t *d_f_node_ids, const int32_t *d_f_node_start, const int32_t *d_f_feat, const int32_t *d_f_nnodes,
        const orbhip_keypoint *d_f_kp, const uint8_t *d_f_desc, const int32_t *d_nF, const int32_t *d_nleft,
        int pairs, int max_nodes, int max_n, size_t frame_stride_kp, float nn_ratio, int check_orientation,
        int32_t *d_match_f, int32_t *d_nmatches);

/* Host-pointer form for ONE (keyframe, frame) pair -- what the ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)
 * method of host/ORBmatcher.cc calls: flattened FeatureVectors (node ids ascending, node_start [nnodes + 1], feature indices),
 * kf_valid [nK], keypoints and descriptors of both sides; nleft < 0 for F.Nleft == -1, else the rig form.  match_f_out [nF]. */
int orbhip_search_by_bow_host(orbhip_ctx *ctx,
        const int32_t *kf_node_ids, const int32_t *kf_node_start, const int32_t *kf_feat, int kf_nnodes, const uint8_t *kf_valid,
        const orbhip_keypoint *kf_kp, const uint8_t *kf_desc, int nK,
        const int32_t *f_node_ids, const int32_t *f_node_start, const int32_t *f_feat, int f_nnodes,
        const orbhip_keypoint *f_kp, const uint8_t *f_desc, int nF, int nleft,
        float nn_ratio, int check_orientation, int32_t *match_f_out, int32_t *nmatches_out);
/* ... with the FRAME side (d_f_kp, d_f_desc: device pointers, nF entries) resident as above; single-camera frames. */
int orbhip_search_by_bow_host_resident(orbhip_ctx *ctx,
        const int32_t *kf_node_ids, const int32_t *kf_node_start, const int32_t *kf_feat, int kf_nnodes, const uint8_t *kf_valid,
        const orbhip_keypoint *kf_kp, const uint8_t *kf_desc, int nK,
        const int32_t *f_node_ids, const int32_t *f_node_start, const int32_t *f_feat, int f_nnodes,
        const orbhip_keypoint *f_kp_host, const orbhip_keypoint *d_f_kp, const uint8_t *d_f_desc, int nF,
        float nn_ratio, int check_orientation, int32_t *match_f_out, int32_t *nmatches_out);

/* ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12) (src/ORBmatcher.cc:827-967,
 * NLeft == -1) -- the matcher of LoopClosing's Sim3 candidates (src/LoopClosing.cc:1005, 2284), batched over keyframe
 * pairs.  Layout as orbhip_search_by_bow_device; both sides carry d_valid [pairs][max_n] (map point exists and is not
 * bad, :867-871, :887-894) and their feature counts d_n1 / d_n2 [pairs].  Inside a shared node every valid KF1 feature,
 * in order, takes the best valid, still unclaimed KF2 feature if best < TH_LOW (strict, :909) and best < nn_ratio *
 * second.  d_matches12 [pairs][max_n]: per KF1 feature the KF2 feature whose map point it receives, or -1
 * (vpMatches12[i] = vpMapPoints2[d_matches12[i]]).  At most 4096 features per keyframe.  All pointers DEVICE. */
int orbhip_search_by_bow_kf_device(orbhip_ctx *ctx,
        const int32_t *d_node_ids1, const int32_t *d_node_start1, const int32_t *d_feat1, const int32_t *d_nnodes1,
        const uint8_t *d_valid1, const orbhip_keypoint *d_kp1, const uint8_t *d_desc1, const int32_t *d_n1,
        const int32_t *d_node_ids2, const int32_t *d_node_start2, const int32_t *d_feat2, const int32_t *d_nnodes2,
        const uint8_t *d_valid2, const orbhip_keypoint *d_kp2, const uint8_t *d_desc2, const int32_t *d_n2,
        int pairs, int max_nodes, int max_n, size_t frame_stride_kp, float nn_ratio, int check_orientation,
        int32_t *d_matches12, int32_t *d_nmatches);

/* Per keyframe pair of orbhip_search_for_triangulation_device: F12 = K1^-T [t12]x R12 K2^-1 (row-major; computed by the
 * caller with the reference's own matrix arithmetic, CameraModels/Pinhole.cpp:124-127), the epipole of KF1's centre in KF2
 * (ORBmatcher.cc:978-992), and the bOnlyStereo / bCoarse arguments. */
typedef struct orbhip_tri_pair { float F12[9]; float ep_x, ep_y; int32_t only_stereo, coarse; } orbhip_tri_pair;

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse) (src/ORBmatcher.cc:969-1210;
 * Pinhole cameras, mpCamera2 == 0) -- the matcher of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:459-460),
 * batched over keyframe pairs.  d_nid1 [pairs][max_n]: vocabulary node of every KF1 feature (orbhip_bow_transform_device);
 * KF2's FeatureVector flattened as in orbhip_search_by_bow_device.  d_has_mp* [pairs][max_n]: the keypoint already has a map
 * point (:1039, :1067); d_u_right* [pairs][max_n] = mvuRight or NULL (monocular).  Per KF1 keypoint without a map point:
 * the KF2 keypoint of the same node, without a map point, with the smallest descriptor distance <= TH_LOW (a later
 * candidate at equal distance replaces an earlier one, :1073) that is not within 10*sqrt(scale) px of the epipole (both
 * monocular, :1083-1091) and passes Pinhole::epipolarConstrain (or any, if coarse); this fork never sets vbMatched2, so
 * KF1 keypoints are independent.  Rotation consistency (:1171-1189) when check_orientation.  d_matches12 [pairs][max_n] =
 * vMatches12 (idx2 or -1; vMatchedPairs is its non-negative entries in index order), d_nmatches [pairs] = the return
 * value.  scale_factors / level_sigma2: HOST arrays of nlevels floats (KF2's mvScaleFactors / mvLevelSigma2).  At most 4096
 * features per keyframe.  KannalaBrandt8 pairs (epipolarConstrain by triangulation) and rigs (mpCamera2) are not covered.
 * All other pointers DEVICE. */
int orbhip_search_for_triangulation_device(orbhip_ctx *ctx,
        const int32_t *d_nid1, const uint8_t *d_has_mp1, const orbhip_keypoint *d_kp1, const uint8_t *d_desc1, const float *d_u_right1,
        const int32_t *d_n1,
        const int32_t *d_node_ids2, const int32_t *d_node_start2, const int32_t *d_feat2, const int32_t *d_nnodes2,
        const uint8_t *d_has_mp2, const orbhip_keypoint *d_kp2, const uint8_t *d_desc2, const float *d_u_right2, const int32_t *d_n2,
        const orbhip_tri_pair *d_pair, int pairs, int max_nodes, int max_n, size_t frame_stride_kp,
        const float *scale_factors, const float *level_sigma2, int nlevels, int check_orientation,
        int32_t *d_matches12, int32_t *d_nmatches);

/* ORBmatcher::SearchForTriangulation for EVERY camera combination of the reference (src/ORBmatcher.cc:969-1210): single Pinhole or
 * KannalaBrandt8 cameras and two-camera rigs (pKF->mpCamera2 != 0, NLeft != -1: TUM-VI stereo-fisheye).  Per keyframe pair: */
typedef struct orbhip_tri_pair_general {
    float R12[4][9], t12[4][3];          /* X1 = R12 X2 + t12 of the camera pair [2*bRight1 + bRight2]: ll, lr, rl, rr (:994-1008; the caller
                                            keeps the reference's cv::Mat arithmetic); a single-camera pair uses [0] (:996-997) */
    float F12[4][9];                     /* K1^-T [t12]x R12 K2^-1 of the same combinations, row-major: what Pinhole::epipolarConstrain
                                            builds (src/CameraModels/Pinhole.cpp:124-127); read only when the first camera is a Pinhole */
    float cam1[2][8], cam2[2][8];        /* mvParameters (fx fy cx cy k1..k4) of pKF1->mpCamera / mpCamera2 and of pKF2's */
    int32_t cam1_type[2], cam2_type[2];  /* 0 Pinhole, 1 KannalaBrandt8 */
    float ep_x, ep_y;                    /* pKF2->mpCamera->project(R2w Cw + t2w) (:978-984) */
    int32_t nleft1, nleft2;              /* NLeft; -1 = single camera.  Rig keyframes: keypoints and flags in mvKeys | mvKeysRight order
                                            (= descriptor rows), bRight = index >= NLeft (:1055, :1087) */
    int32_t only_stereo, coarse;
} orbhip_tri_pair_general;
/* Layout as orbhip_search_for_triangulation_device.  The candidate test is GeometricCamera::epipolarConstrain of the FIRST camera of
 * the picked pair: Pinhole -> distance to the epipolar line of F12[c] (Pinhole.cpp:129-143); KannalaBrandt8 -> TriangulateMatches
 * (src/CameraModels/KannalaBrandt8.cpp:235-238, 334-401: ray parallax < 0.9998, linear triangulation by the SVD of a 4x4 system,
 * both depths positive, reprojection errors <= 5.991 sigma^2 in both cameras, z1 > 1e-4).  Rig pairs have no stereo keypoints
 * (bStereo needs mpCamera2 == 0, :1044) and skip the epipole test (:1091).  level_sigma2_1 = pKF1->mvLevelSigma2, scale_factors2 /
 * level_sigma2_2 = pKF2's (HOST arrays of nlevels floats).  cv::SVD (one-sided Jacobi) and the float libm calls are restated as
 * DESIGN.md 2 describes: parity unpinned against an OpenCV build, bit-exact against the oracle. */
int orbhip_search_for_triangulation_general_device(orbhip_ctx *ctx,
        const int32_t *d_nid1, const uint8_t *d_has_mp1, const orbhip_keypoint *d_kp1, const uint8_t *d_desc1, const float *d_u_right1,
        const int32_t *d_n1,
        const int32_t *d_node_ids2, const int32_t *d_node_start2, const int32_t *d_feat2, const int32_t *d_nnodes2,
        const uint8_t *d_has_mp2, const orbhip_keypoint *d_kp2, const uint8_t *d_desc2, const float *d_u_right2, const int32_t *d_n2,
        const orbhip_tri_pair_general *d_pair, int pairs, int max_nodes, int max_n, size_t frame_stride_kp,
        const float *level_sigma2_1, const float *scale_factors2, const float *level_sigma2_2, int nlevels, int check_orientation,
        int32_t *d_matches12, int32_t *d_nmatches);

/* ORBmatcher::SearchForTriangulation, the overload that also returns the triangulated points (src/ORBmatcher.cc:1212-1402, no caller
 * in the reference).  Per candidate the camera and pose of each side are picked by bRight (:1307-1321) and the test is
 * GeometricCamera::matchAndtriangulate of the FIRST camera: KannalaBrandt8 (src/CameraModels/KannalaBrandt8.cpp:240-332: ray parallax
 * < 0.9998 with the rays turned into the world frame, linear triangulation with the ABSOLUTE poses, both depths positive,
 * reprojection errors <= 5.991 sigma^2 in both cameras) or Pinhole, whose implementation is `{ return false; }`
 * (include/CameraModels/Pinhole.h:91-94): a pair whose first camera is a Pinhole matches nothing.  bOnlyStereo is not read by that
 * overload, there is no epipole gate and vbMatched2 is never set.  d_pair: cam1 / cam2 / cam*_type / nleft1 / nleft2 are read (the
 * rest ignored); d_poses: rows 0..2 of GetPose() ([0]) and GetRightPose() ([1]) of both keyframes, row-major 3x4.
 * d_points12 [pairs][max_n][3]: x3Dtriangulated (a WORLD point) of the kept match of KF1 keypoint i (zeros where matches12 < 0;
 * entries of matches the rotation histogram removed keep their point, as vMatchesPoints12 does). */
typedef struct orbhip_tri_pair_poses { float Tcw1[2][12], Tcw2[2][12]; } orbhip_tri_pair_poses;
int orbhip_match_and_triangulate_device(orbhip_ctx *ctx,
        const int32_t *d_nid1, const uint8_t *d_has_mp1, const orbhip_keypoint *d_kp1, const uint8_t *d_desc1, const int32_t *d_n1,
        const int32_t *d_node_ids2, const int32_t *d_node_start2, const int32_t *d_feat2, const int32_t *d_nnodes2,
        const uint8_t *d_has_mp2, const orbhip_keypoint *d_kp2, const uint8_t *d_desc2, const int32_t *d_n2,
        const orbhip_tri_pair_general *d_pair, const orbhip_tri_pair_poses *d_poses, int pairs, int max_nodes, int max_n, size_t frame_stride_kp,
        const float *level_sigma2_1, const float *level_sigma2_2, int nlevels, int check_orientation,
        int32_t *d_matches12, float *d_points12, int32_t *d_nmatches);
/* one pair, HOST pointers (as orbhip_search_for_triangulation_host below); points12_out [n1][3] */
int orbhip_match_and_triangulate_host(orbhip_ctx *ctx,
        const int32_t *nid1, const uint8_t *has_mp1, const orbhip_keypoint *kp1, const uint8_t *desc1, int n1,
        const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2,
        const uint8_t *has_mp2, const orbhip_keypoint *kp2, const uint8_t *desc2, int n2,
        const orbhip_tri_pair_general *pair, const orbhip_tri_pair_poses *poses, const float *level_sigma2_1, const float *level_sigma2_2,
        int nlevels, int check_orientation, int32_t *matches12_out, float *points12_out, int32_t *nmatches_out);

/* Host-pointer forms for ONE keyframe (pair) -- what the ORBmatcher methods of host/ORBmatcher.cc call (upload into the context's
 * arena, the same kernels as the batched device entry points, download, synchronise).  All pointers HOST.
 *   orbhip_search_for_triangulation_host: one pair through the general kernel (nid1 [n1]: vocabulary node of every KF1 feature, -1 =
 *     in no node; KF2's FeatureVector flattened); matches12_out [n1].
 *   orbhip_fuse_search_host: orbhip_fuse_search_device for one keyframe; best_idx_out / best_dist_out [nq].
 *   orbhip_search_by_bow_kf_host: orbhip_search_by_bow_kf_device for one pair; matches12_out [n1].
 *   orbhip_pose_optimization_host: orbhip_pose_optimization_device for one frame; Xw [n][3], obs [n][3], inv_sigma2 [n] doubles,
 *     right [n] or NULL, pose_inout [7], outlier_out [n], *n_inliers_out, stats_out [4] or NULL. */
int orbhip_search_for_triangulation_host(orbhip_ctx *ctx,
        const int32_t *nid1, const uint8_t *has_mp1, const orbhip_keypoint *kp1, const uint8_t *desc1, const float *u_right1, int n1,
        const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2,
        const uint8_t *has_mp2, const orbhip_keypoint *kp2, const uint8_t *desc2, const float *u_right2, int n2,
        const orbhip_tri_pair_general *pair, const float *level_sigma2_1, const float *scale_factors2, const float *level_sigma2_2,
        int nlevels, int check_orientation, int32_t *matches12_out, int32_t *nmatches_out);
int orbhip_fuse_search_host(orbhip_ctx *ctx, const orbhip_proj_query *q, const uint8_t *desc_q, int nq, const orbhip_keypoint *kp,
                            const uint8_t *desc, const float *u_right, int n, const float *inv_level_sigma2, int nlevels,
                            float min_x, float min_y, float max_x, float max_y, int32_t *best_idx_out, int32_t *best_dist_out);
int orbhip_search_by_bow_kf_host(orbhip_ctx *ctx,
        const int32_t *node_ids1, const int32_t *node_start1, const int32_t *feat1, int nnodes1, const uint8_t *valid1,
        const orbhip_keypoint *kp1, const uint8_t *desc1, int n1,
        const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2, const uint8_t *valid2,
        const orbhip_keypoint *kp2, const uint8_t *desc2, int n2,
        float nn_ratio, int check_orientation, int32_t *matches12_out, int32_t *nmatches_out);

/* Frame::UndistortKeyPoints (src/Frame.cc:738-771), batched: d_kp_un = d_kp with pt replaced by
 * cv::undistortPoints(pt, K, mDistCoef, R = I, P = K) (OpenCV 3.4.1 cvUndistortPoints: 5 fixed-point iterations in double,
 * rounded to float).  dist_coef: HOST array (k1, k2, p1, p2[, k3]), n_dist 4 or 5; k1 == 0 copies (Frame.cc:740-744).
 * d_kp and d_kp_un may be the same buffer.  Other pointers DEVICE. */
int orbhip_undistort_keypoints_device(orbhip_ctx *ctx, const orbhip_keypoint *d_kp, const int32_t *d_n, int frames, int max_n,
                                      size_t frame_stride_kp, float fx, float fy, float cx, float cy, const float *dist_coef,
                                      int n_dist, orbhip_keypoint *d_kp_un);

/* Frame::AssignFeaturesToGrid (src/Frame.cc:377-408, Nleft == -1), batched, as a CSR per frame: d_cell_start
 * [frames][64*48+1], d_items [frames][max_n]; cell (ix, iy) = index ix*48+iy holds the keypoint indices
 * d_items[cell_start[c] .. cell_start[c+1]) in insertion (= index) order -- mGrid[ix][iy] of the reference
 * (PosInGrid rounds, Frame.cc:716-726).  At most 8192 keypoints per frame.  All pointers DEVICE. */
int orbhip_assign_features_to_grid_device(orbhip_ctx *ctx, const orbhip_keypoint *d_kp, const int32_t *d_n, int frames, int max_n,
                                          size_t frame_stride_kp, float min_x, float min_y, float max_x, float max_y,
                                          int32_t *d_cell_start, int32_t *d_items);

/* Frame::AssignFeaturesToGrid for frames of a two-camera rig (Nleft != -1, src/Frame.cc:395-405): keypoints [0, d_nleft[f]) fill mGrid,
 * the others mGridRight.  d_cell_start [frames][2*64*48+1]: cells 0 .. 3071 = mGrid (cell ix*48+iy), 3072 .. 6143 = mGridRight, one
 * running CSR into d_items [frames][max_n]; mGrid items are keypoint indices, mGridRight items are i - Nleft (:403). */
int orbhip_assign_features_to_grid_rig_device(orbhip_ctx *ctx, const orbhip_keypoint *d_kp, const int32_t *d_n, const int32_t *d_nleft,
                                              int frames, int max_n, size_t frame_stride_kp, float min_x, float min_y, float max_x,
                                              float max_y, int32_t *d_cell_start, int32_t *d_items);

/* Second half of Frame::ComputeBoW (src/Frame.cc:729-736): TemplatedVocabulary::transform(features, mBowVec, mFeatVec, 4)
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1139-1208, TF_IDF weighting + L1 norm as in ORBvoc) from the per-feature
 * (word id, weight, node id) of orbhip_bow_transform_device, batched over frames:
 *   mFeatVec as the CSR the SearchByBoW / SearchForTriangulation kernels read: d_node_ids [frames][max_nodes] ascending,
 *     d_node_start [frames][max_nodes+1] into d_feat [frames][max_n] (feature indices in feature order), d_nnodes [frames];
 *   mBowVec as d_bow_word / d_bow_value [frames][max_n] (ascending word ids, L1-normalised sums), d_nwords [frames].
 * Features whose word has weight <= 0 ("stopped") are in neither.  std::map order, BowVector::addWeight's accumulation order and
 * BowVector::normalize's summation order (BowVector.cpp) are reproduced, so the values are bit-identical to DBoW2's.
 * max_n <= 4096.  All pointers DEVICE. */
int orbhip_bow_vectors_device(orbhip_ctx *ctx, const int32_t *d_word_id, const double *d_weight, const int32_t *d_node_id,
                              const int32_t *d_n, int frames, int max_n, int max_nodes,
                              int32_t *d_node_ids, int32_t *d_node_start, int32_t *d_feat, int32_t *d_nnodes,
                              int32_t *d_bow_word, double *d_bow_value, int32_t *d_nwords);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:327-403), batched over map points: point p
 * has d_n[p] observing descriptors at d_desc + p*max_n*32 (the loop of :347-361 packs them, left then right index);
 * d_best_idx[p] = BestIdx: the descriptor with the least median Hamming distance to all of them (median = sorted
 * row [int(0.5*(n-1))], first minimum wins).  d_best_desc (may be NULL) receives mDescriptor, [points][32].
 * max_n <= 256.  All pointers DEVICE. */
int orbhip_distinctive_descriptors_device(orbhip_ctx *ctx, const uint8_t *d_desc, const int32_t *d_n, int points, int max_n,
                                          int32_t *d_best_idx, uint8_t *d_best_desc);

/* The per-feature half of Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:729-736 -> DBoW2
 * TemplatedVocabulary::transform, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1139-1260): for every
 * descriptor the tree descent of :1218-1260 -- at each level the child with the smallest Hamming distance, first minimum
 * wins, until a leaf -- giving its word id, the leaf's weight and the node passed at level L - levelsup (0 = root when
 * that level is <= 0; ORB-SLAM3 uses levelsup = 4).  Building BowVector (addWeight in feature order + L1 normalise,
 * :1165-1210) and FeatureVector from these triples stays on the host (std::map containers).
 * Vocabulary, resident on the device as a flat tree: node i has children d_child_ids[d_child_start[i] ..
 * d_child_start[i+1]) (none = leaf; node 0 = root), 32-byte descriptors d_node_desc[i], d_node_word[i], d_node_weight[i].
 * Frame f reads d_n[f] descriptors at d_desc + f*frame_stride*32; outputs are [frames][max_n].  All pointers DEVICE. */
int orbhip_bow_transform_device(orbhip_ctx *ctx, const uint8_t *d_desc, const int32_t *d_n, int frames, int max_n,
                                size_t frame_stride, const uint8_t *d_node_desc, const int32_t *d_child_start,
                                const int32_t *d_child_ids, const int32_t *d_node_word, const double *d_node_weight,
                                int L, int levelsup, int32_t *d_word_id, double *d_weight, int32_t *d_nid);

/* Frame::ComputeStereoMatches (src/Frame.cc:802-980), batched: rectified-stereo association
 * of the keypoints the LEFT and RIGHT extractor produced in their latest extract call (frame f with frame f;
 * both called with lapping {0,0} as the stereo constructor does, src/Frame.cc:109-110).  Per left keypoint:
 * best right keypoint by descriptor distance among those whose row band (+-2*scale[octave]) covers its row,
 * octave within +-1, disparity in [0, mbf/mb] (:833-887); if better than (TH_HIGH+TH_LOW)/2: 11x11 SAD over
 * +-5 px on the left keypoint's pyramid level (both extractors' device pyramids; columns left of the image
 * are the reflect-101 padding of mvImagePyramid), parabola fit, disparity gates (:890-963); finally matches
 * with SAD >= 1.5*1.4*median are dropped (:966-980).
 * d_u_right / d_depth [batch][max_keypoints] = mvuRight / mvDepth (-1 = no match); d_n_matches [batch] (may be
 * NULL) = matches kept.  The extractors must share image size, levels, scale factor and feature budget and may
 * live on different contexts (the right one's stream is waited for).  Runs on the LEFT context's stream. */
int orbhip_compute_stereo_matches_device(orbhip_extractor *left, orbhip_extractor *right, float mb, float mbf,
                                         float *d_u_right, float *d_depth, int32_t *d_n_matches);

/* One stereo frame, HOST outputs: the call behind Frame::ComputeStereoMatches() of host/Frame.cc (src/Frame.cc:802-980, called by the rectified-stereo
 * constructor at :130).  u_right_out / depth_out [n] = mvuRight / mvDepth of frame 0 of the two extractors' latest extractions (n = the left
 * frame's keypoint count); *n_matches_out (may be NULL) = matches kept.  Synchronous. */
int orbhip_compute_stereo_matches_host(orbhip_extractor *left, orbhip_extractor *right, float mb, float mbf,
                                       float *u_right_out, float *depth_out, int n, int32_t *n_matches_out);

/* ------------------------------------------------------------------ local BA */
/* One keyframe-window graph in SoA form: what Optimizer::LocalBundleAdjustment builds
 * between src/Optimizer.cc:1850 and :2034.  Poses world->camera as (qx,qy,qz,qw,tx,ty,tz)
 * doubles (g2o::SE3Quat, Thirdparty/g2o/g2o/types/se3quat.h:41); points xyz doubles.
 * Edges are sorted by point (insertion order of Optimizer.cc:1940-2034 == point-major). */
typedef struct {
    int32_t n_poses;            /* free + fixed keyframes */
    int32_t n_points;
    int32_t n_edges;
    const uint8_t *pose_fixed;  /* [n_poses] 1 = fixed (lFixedCameras, Optimizer.cc:1877-1903) */
    const int32_t *edge_pose;   /* [n_edges] */
    const int32_t *edge_point;  /* [n_edges] non-decreasing */
    const double *edge_obs;     /* [n_edges][3]  u,v,(ur; unused for mono) */
    const double *edge_inv_sigma2; /* [n_edges]  mvInvLevelSigma2[octave] */
    const uint8_t *edge_stereo; /* [n_edges] 0 = EdgeSE3ProjectXYZ, 1 = EdgeStereoSE3ProjectXYZ,
                                   2 = EdgeSE3ProjectXYZToBody (observation in the second camera, see Trl below) */
    double fx, fy, cx, cy, bf;  /* intrinsics (Pinhole.cpp:41-47) + stereo baseline*fx */
    int32_t camera_model;       /* 0 = Pinhole; 1 = KannalaBrandt8 for the monocular edges (src/CameraModels/
                                   KannalaBrandt8.cpp:52-69 project, :166-195 projectJac) */
    double kb[4];               /* k1..k4 (mvParameters[4..7]) when camera_model == 1 */
    /* second, rigidly attached camera for the edges of type 2 (pKFi->mpCamera2 / mTrl, src/Optimizer.cc:2001-2032;
     * include/OptimizableTypes.h:112-141, src/OptimizableTypes.cpp:192-213): */
    double Trl[7];              /* mTrl as (qx,qy,qz,qw,tx,ty,tz): left-camera frame -> right-camera frame */
    double fx2, fy2, cx2, cy2;
    int32_t camera2_model;      /* as camera_model */
    double kb2[4];
    /* Per-keyframe calibration (round 4).  The reference hands every edge its OWN keyframe's camera -- e->pCamera = pKFi->mpCamera
     * (src/Optimizer.cc:1961), e->fx ... e->bf = pKFi->fx ... pKFi->mbf (:1990-1994), e->mTrl / e->pCamera2 = pKFi->mTrl / mpCamera2
     * (:2021-2023) -- so a window of an Atlas map built from two cameras mixes calibrations.  n_cameras > 0: pose i projects through
     * cameras[pose_camera[i]] (all nine groups of fields of the entry; the single-calibration fields above are then ignored);
     * n_cameras == 0: every pose uses the fields above (cameras / pose_camera may be NULL). */
    int32_t n_cameras;
    const struct orbhip_ba_camera *cameras;   /* [n_cameras] */
    const int32_t *pose_camera;               /* [n_poses] index into cameras */
} orbhip_ba_graph;
/* one calibration of a BA graph: the same fields, with the same meaning, as the single-calibration part of orbhip_ba_graph */
typedef struct orbhip_ba_camera {
    double fx, fy, cx, cy, bf;
    int32_t camera_model;
    double kb[4];
    double Trl[7];
    double fx2, fy2, cx2, cy2;
    int32_t camera2_model;
    double kb2[4];
} orbhip_ba_camera;

typedef struct {
    int32_t iters1, iters2;     /* optimize(5) then optimize(10): Optimizer.cc:2048,2122 */
    double huber_mono2, huber_stereo2;   /* 5.991, 7.815 (Optimizer.cc:1910-1911) */
    double user_lambda_init;    /* 0 -> tau*max diag (levenberg.cpp:171-185); 100 if inertial */
    double tau;                 /* 1e-50 (levenberg.cpp:47) */
    int32_t max_trials;         /* 100 (levenberg.cpp:51) */
    /* Variant switches for the map-merge local BA, Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF,
     * pbStopFlag) (src/Optimizer.cc:6255-6800) -- all 0 in orbhip_ba_default_params: */
    int32_t stage2_exclude_outliers;  /* after optimize(5): edges with chi2 > gate or depth <= 0 get level 1 and do not
                                         take part in optimize(10) (:6554-6556, :6571-6573) */
    int32_t stage2_drop_robust;       /* after optimize(5): setRobustKernel(0) on every edge (:6560, :6577) */
    int32_t no_discard;               /* no ">= 50 % outliers" bail-out (the merge variant has none) */
    double gate_mono2, gate_stereo2;  /* outlier gates when they differ from the Huber deltas (merge: Huber sqrt(5.99),
                                         gate 5.991, :6395,6554); 0 = use huber_mono2 / huber_stereo2 */
} orbhip_ba_params;

typedef struct {
    int32_t iterations_run[2];  /* outer iterations executed in pass 1 / pass 2 */
    int32_t lm_trials;          /* total inner trials */
    int32_t n_outliers;         /* chi2 > gate or depth <= 0 (Optimizer.cc:2126-2173) */
    int32_t discarded;          /* 1 if >= 50 % outliers (Optimizer.cc:2177-2181): no write-back */
    double chi2_initial, chi2_final;
} orbhip_ba_stats;

void orbhip_ba_default_params(orbhip_ba_params *p);
/* The parameters of the map-merge variant (src/Optimizer.cc:6255): Huber 5.99 / 7.815, gates 5.991 / 7.815, outliers of
 * the first pass excluded and the robust kernel dropped for the second pass, no bail-out. */
void orbhip_ba_merge_params(orbhip_ba_params *p);

/* The numerical core of Optimizer::LocalBundleAdjustment(KeyFrame*, bool* pbStopFlag, Map*, int&)
 *                                            include/Optimizer.h:58, src/Optimizer.cc:1699-2344
 * i.e. initializeOptimization + optimize(5) + initializeOptimization(0) + optimize(10) +
 * outlier classification, on `n_graphs` independent graphs at once.  Host pointers in the
 * graph structs; poses_inout[g] = double[n_poses*7], points_inout[g] = double[n_points*3]
 * (updated in place unless discarded), edge_outlier_out[g] = uint8[n_edges] (may be NULL),
 * abort = LocalMapping::mbAbortBA (polled between iterations and LM trials; may be NULL). */
int orbhip_ba_solve_batch(orbhip_ctx *ctx, const orbhip_ba_graph *graphs, int n_graphs,
                          const orbhip_ba_params *params, volatile const uint8_t *abort,
                          double *const *poses_inout, double *const *points_inout,
                          uint8_t *const *edge_outlier_out, orbhip_ba_stats *stats_out);


/* Parameters of Optimizer::BundleAdjustment / GlobalBundleAdjustemnt (src/Optimizer.cc:54-330; map initialisation
 * src/Tracking.cc:1603 with 20 iterations, LoopClosing::RunGlobalBundleAdjustment src/LoopClosing.cc:2437 with 10, bRobust = false):
 * the same graph and solver, ONE optimize(iterations) pass, Huber sqrt(5.99) / sqrt(7.815) when robust (else no kernel), no outlier
 * stage and no bail-out.  pose_fixed = (mnId == InitKFid) (:126).  Up to 80 free keyframes the reduced system is built on the FP64
 * matrix cores and factored inside one workgroup; larger windows (to 682 free keyframes) take a global-memory path. */
void orbhip_ba_global_params(orbhip_ba_params *p, int iterations, int robust);

/* Two-phase form of the same solver for callers that keep graphs resident in HBM (bench,
 * map-merge BA): create uploads topology + initial estimates; solve runs both LM passes and
 * the outlier gates entirely on the device (restarting from the initial estimates each call);
 * download copies estimates / outlier flags / stats back. */
typedef struct orbhip_ba_batch orbhip_ba_batch;
/* How BA batches created on this context AFTERWARDS form the Schur complement (a property of the context: two contexts may differ,
 * and orbhip_ba_solve_batch follows its context too): 0 / 1 = from per-block-pair lists, one 16-lane row per pair of free keyframes
 * (default: measured faster at every batch size), 2 = the FP64-MFMA panel GEMM (up to 80 free keyframes; always used by the
 * landmark-sharded mode).  Results agree to rounding (different summation orders); both are parity-tested against the oracle. */
int orbhip_ctx_set_ba_schur_mode(orbhip_ctx *ctx, int mode);
int orbhip_ba_batch_create(orbhip_ctx *ctx, const orbhip_ba_graph *graphs, int n_graphs,
                           double *const *poses, double *const *points, orbhip_ba_batch **out);
int orbhip_ba_batch_solve(orbhip_ba_batch *b, const orbhip_ba_params *params, volatile const uint8_t *abort);
int orbhip_ba_batch_download(orbhip_ba_batch *b, double *const *poses_out, double *const *points_out,
                             uint8_t *const *edge_outlier_out, orbhip_ba_stats *stats_out);
int orbhip_ba_batch_ticks(const orbhip_ba_batch *b);   /* LM trials of the slowest graph, last solve */
void orbhip_ba_batch_destroy(orbhip_ba_batch *b);
/* Landmark-sharded solve of the same graphs by `world` ranks (one process per GPU) -- the optional single-graph mode of the
 * multi-GPU path: g2o iterates landmarks independently (Thirdparty/g2o/g2o/core/block_solver.hpp:381-432), so rank r owns points
 * [r*L/world, (r+1)*L/world) of every graph with their edges, poses are replicated, and the partial sums meet in ONE all-gather
 * per exchange: (1) Hpp, bp, chi2 and max |Hll diag| after buildSystem, (2) the shared Schur block sum_l W D^-1 W^T and W D^-1 b
 * every LM trial, (3) trial chi2, computeScale and the abort flag, (4) outlier / edge counts at the end.  Every rank reduces the
 * gathered slots in rank order, so all ranks solve the same reduced system and take the same LM decisions (deterministic).
 * create_sharded: every rank passes the SAME full graphs and initial estimates.  set_exchange_buffer: a DEVICE buffer of at least
 * world * orbhip_ba_batch_exchange_doubles(b) doubles, laid out [world][exchange_doubles].  solve_sharded: before calling
 * `exchange(user, stage, count)` the library has written `count` doubles into slot `rank` and drained its stream; the callback
 * must all-gather so that slot r of every rank's buffer holds rank r's `count` doubles (RCCL: ncclAllGather(buf + rank*stride,
 * buf, stride, ncclDouble, comm, stream) + stream sync) and return 0.  download writes this rank's points
 * and edge flags at their positions in the caller's full-size arrays (other ranks' entries untouched), all poses, and stats that
 * are identical on every rank.  The abort flag of any rank stops all of them at the same trial. */
typedef int (*orbhip_ba_exchange_fn)(void *user, int stage, size_t count);
int orbhip_ba_batch_create_sharded(orbhip_ctx *ctx, const orbhip_ba_graph *graphs, int n_graphs, double *const *poses,
                                   double *const *points, int rank, int world, orbhip_ba_batch **out);
size_t orbhip_ba_batch_exchange_doubles(const orbhip_ba_batch *b);
int orbhip_ba_batch_set_exchange_buffer(orbhip_ba_batch *b, double *d_buf, size_t capacity_doubles);
int orbhip_ba_batch_solve_sharded(orbhip_ba_batch *b, const orbhip_ba_params *params, volatile const uint8_t *abort,
                                  orbhip_ba_exchange_fn exchange, void *user);
/* Measurement hooks: hipEvent timing of the Schur GEMM launches (on the context's stream), the
 * MFMA flops of the tiles that hold data (flops_per_launch), of everything issued, and a measured FP64 matrix-core peak. */
int orbhip_ba_batch_set_profiling(orbhip_ba_batch *b, int enable);
int orbhip_ba_batch_gemm_profile(const orbhip_ba_batch *b, float *total_ms, int *launches, double *flops_per_launch);
double orbhip_ba_batch_gemm_dense_flops(const orbhip_ba_batch *b);   /* same tiles without block-sparsity skipping */
double orbhip_ba_batch_gemm_issued_flops(const orbhip_ba_batch *b);  /* MFMA flops one launch issues (whole row strips a point touches) */
int orbhip_mfma_f64_peak_tflops(orbhip_ctx *ctx, double *tflops_out);

/* ------------------------------------------------------------------ inertial local BA (LocalInertialBA)
 * The numerical core of Optimizer::LocalInertialBA(KeyFrame*, bool *pbStopFlag, Map*, bool bLarge, bool bRecInit)
 *                                            include/Optimizer.h:98, src/Optimizer.cc:4574-5187
 * i.e. computeActiveErrors + activeRobustChi2, ONE optimize(opt_it) with lambda_init set (:5045-5049; the stop flag is only
 * handed to the optimizer afterwards and has no effect, :5050-5051), the outlier gates (:5056-5088) and the fail check (:5096),
 * on `n_windows` independent windows at once (one workgroup each).  Graph: per keyframe a VertexPose (ImuCamPose: body pose
 * Rwb, twb; the camera pose follows through Tcb) and, when it has IMU states, VertexVelocity / VertexGyroBias / VertexAccBias
 * (include/G2oTypes.h:131-240); EdgeMono / EdgeStereo to the landmarks (src/G2oTypes.cc:349-482), one EdgeInertial + EdgeGyroRW
 * + EdgeAccRW per pair of consecutive keyframes (src/G2oTypes.cc:693-800, include/G2oTypes.h:632-700).  The caller builds the
 * window exactly as :4588-4682 does (temporal keyframes, the fixed previous one, fixed covisible ones) and passes flat arrays.
 * Pinhole and KannalaBrandt8 cameras; keyframes of a two-camera rig (mpCamera2, mTrl) with EdgeMono(1) edges on the right camera.
 * Up to 32 keyframes with IMU states (480 unknowns; the reference caps the window at 10, 25 when bLarge).
 * Deviations from the reference's arithmetic, all far below the 1e-4 parity tolerance: the bias-corrected preintegrated deltas
 * (src/ImuTypes.cc:357-378) and ExpSO3's re-orthonormalisation (src/G2oTypes.cc:991-1008) are evaluated in double instead of
 * float cv::Mat arithmetic; dense LDL^T instead of Eigen::SimplicialLDLT.  Parity is unpinned (no reference vectors exist). */
#define ORBHIP_IBA_KF 21        /* keyframe state: Rwb[9] row-major, twb[3], velocity[3], gyro bias[3], accelerometer bias[3] */
#define ORBHIP_IBA_PREINT 67    /* IMU::Preintegrated: dT, dR[9], dV[3], dP[3], JRg[9], JVg[9], JVa[9], JPg[9], JPa[9], bias bg[3], ba[3] */
typedef struct {
    int32_t n_kf;
    const uint8_t *kf_fixed;        /* setFixed(true): lFixedKeyFrames (:4757-4781) */
    const uint8_t *kf_imu;          /* pKFi->bImu (:4726-4740); keyframes without it have only the pose vertex */
    double Rcb[9], tcb[3];          /* mImuCalib.Tcb */
    double fx, fy, cx, cy, bf;      /* pKF->mpCamera, mbf */
    int32_t camera_model;           /* 0 Pinhole, 1 KannalaBrandt8 */
    double kb[4];
    int32_t has_cam2;               /* pKF->mpCamera2 != NULL: ImuCamPose carries a second camera (src/G2oTypes.cc:57-67) */
    double Trl[12];                 /* pKF->mTrl, 3x4 row-major */
    double fx2, fy2, cx2, cy2;
    int32_t camera2_model;
    double kb2[4];
    int32_t n_points;
    int32_t n_edges;                /* grouped by point: edge_point ascending, as :4914-5034 creates them */
    const int32_t *edge_kf, *edge_point;
    const double *edge_obs;         /* [3]: kpUn.pt.x, kpUn.pt.y, mvuRight */
    const uint8_t *edge_stereo;     /* 0 EdgeMono(0), 1 EdgeStereo(0), 2 EdgeMono(1): right-camera observation (:5000-5031; has_cam2);
                                       a keyframe may hold a left and a right edge to the same point */
    const double *edge_inv_sigma2;  /* mvInvLevelSigma2[octave] / uncertainty2 (:4949-4952) */
    const uint8_t *edge_close;      /* pMP->mTrackDepth < 10 (:5063); may be NULL */
    int32_t n_inertial;
    const int32_t *in_kf1, *in_kf2; /* pKFi->mPrevKF, pKFi (both with IMU states) */
    const double *in_preint;        /* [ORBHIP_IBA_PREINT] */
    const double *in_info;          /* [81] EdgeInertial information (src/G2oTypes.cc:702-714; x 1e-2 on the edge into the fixed keyframe, :4836) */
    const double *in_info_g, *in_info_a;   /* [9] each (:4845-4863) */
    const uint8_t *in_robust;       /* Huber sqrt(16.92) (:4828-4838) */
} orbhip_iba_window;
typedef struct {
    int32_t iterations;             /* 10, or 4 when bLarge */
    double lambda_init;             /* 1.0, or 1e-2 when bLarge */
    int32_t large;                  /* bLarge: no fail check */
    int32_t max_trials;             /* 100 */
} orbhip_iba_params;
typedef struct {
    int32_t iterations_run, lm_trials, n_outliers;
    int32_t failed;                 /* 2*err < err_end or NaN (:5096-5100): the estimates are NOT written back */
    double err, err_end;            /* activeRobustChi2 before / after */
} orbhip_iba_stats;
void orbhip_iba_default_params(orbhip_iba_params *p, int large);
/* Host pointers.  kf_state_inout[w] = double[n_kf * ORBHIP_IBA_KF], points_inout[w] = double[n_points * 3] (updated unless
 * failed), edge_outlier_out[w] = uint8[n_edges] = the observation goes to vToErase (may be NULL), stats_out[n_windows] (may be
 * NULL).  Synchronous. */
int orbhip_inertial_ba_solve_batch(orbhip_ctx *ctx, const orbhip_iba_window *windows, int n_windows, const orbhip_iba_params *params,
                                   double *const *kf_state_inout, double *const *points_inout, uint8_t *const *edge_outlier_out,
                                   orbhip_iba_stats *stats_out);
/* The same, as a RESIDENT batch (the inertial counterpart of orbhip_ba_batch_create): create packs the windows' constant part
 * (topology, observation and preintegration data, the kernels' task lists) and uploads it once into memory the batch owns; a solve
 * uploads only the states (n_kf x 21 + n_points x 3 doubles per window) and launches.  LocalInertialBA is called on a window whose
 * topology the caller rebuilds per call, so the reference path uses the one-shot form; the batch form is for callers that re-solve
 * the same windows (the reference's own second pass with bLarge toggled, relinearisation sweeps, benchmarks) and is what the
 * device-rate figure in bench.py times.  windows' arrays need not outlive create.  set_states replaces the initial states the next
 * solve starts from (create's kf_states/points until then); solve is synchronous and leaves the result on the device; download
 * follows the one-shot call's rules (failed windows are not written).  Not thread-safe per batch; the batch must be destroyed
 * before its context. */
typedef struct orbhip_iba_batch orbhip_iba_batch;
int orbhip_iba_batch_create(orbhip_ctx *ctx, const orbhip_iba_window *windows, int n_windows, double *const *kf_states,
                            double *const *points, orbhip_iba_batch **out);
int orbhip_iba_batch_set_states(orbhip_iba_batch *b, double *const *kf_states, double *const *points);
int orbhip_iba_batch_solve(orbhip_iba_batch *b, const orbhip_iba_params *params);
int orbhip_iba_batch_download(orbhip_iba_batch *b, double *const *kf_states_out, double *const *points_out,
                              uint8_t *const *edge_outlier_out, orbhip_iba_stats *stats_out);
void orbhip_iba_batch_destroy(orbhip_iba_batch *b);
/* Diagnostics: workgroups per window ("team size" G) of the calling thread's latest inertial solve.  Small batches give a window a
 * team of up to 16 workgroups that meet at a device-wide barrier, which needs the whole grid resident: only ONE team grid runs per
 * device at a time (in-process mutex + advisory file lock /tmp/.orbhip_team_gpu<N>.lock); a solve that finds one in flight, and
 * every batch of more than ~128 windows, runs G = 1.  Results for different G agree to rounding (summation order), identical LM
 * decisions in every tested case.  ORBHIP_IBA_TEAM=<n> caps G (tests). */
int orbhip_inertial_ba_last_team_size(void);

/* ------------------------------------------------------------------ pose-only BA
 * Optimizer::PoseOptimization (src/Optimizer.cc:854-1168), batched over frames: per frame one free
 * VertexSE3Expmap and n unary edges -- EdgeSE3ProjectXYZOnlyPose (include/OptimizableTypes.h:31-57) when
 * uRight < 0, g2o::EdgeStereoSE3ProjectXYZOnlyPose (Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:204-236)
 * otherwise -- solved by the same Levenberg-Marquardt as local BA with a dense 6x6 LDL^T
 * (LinearSolverDense, Thirdparty/g2o/g2o/solvers/linear_solver_dense.h:65-117): 4 rounds x 10 iterations, each
 * from the frame's initial pose, outliers (float chi2 > 5.991f / 7.815f) dropped from the next round and
 * re-tested after it, Huber kernel removed for the last round (:1043-1149).  Called by Tracking after every
 * matcher (src/Tracking.cc:1775,1934,1996-2002).  The fisheye right-camera edge (mpCamera2) is not covered.
 * Frame f reads Xw at d_Xw + f*max_edges*3 (map points, float values widened to double, :913-916),
 * d_obs [f][max_edges][3] = (kpUn.pt.x, kpUn.pt.y, mvuRight), d_inv_sigma2 [f][max_edges]
 * (mvInvLevelSigma2[octave]), d_n_edges [f].  d_pose [f][7] = (qx,qy,qz,qw,tx,ty,tz) of Tcw, in/out
 * (untouched when n < 3, :1040-1041).  d_outlier [f][max_edges] = pFrame->mvbOutlier.  d_n_inliers [f] = the
 * return value nInitialCorrespondences - nBad.  d_stats may be NULL, else [f][4] = rounds, LM iterations,
 * LM trials, nBad.  max_edges <= 8192.  kb8_k: HOST pointer to k1..k4 when pFrame->mpCamera is a KannalaBrandt8
 * (monocular edges then project through src/CameraModels/KannalaBrandt8.cpp:52-69,166-195), NULL = Pinhole.
 * cam2 (HOST pointer, may be NULL) + d_right [f][max_edges] (DEVICE, may be NULL): when pFrame->mpCamera2 exists
 * (:960-1037), d_right[e] = 1 marks an observation made in the second camera -- EdgeSE3ProjectXYZOnlyPoseToBody
 * (include/OptimizableTypes.h:59-87, src/OptimizableTypes.cpp:82-106); such rows carry uRight < 0 (2-D residual,
 * monocular gate).  All other pointers DEVICE; asynchronous on the context's stream. */
/* second camera of a rigid pair for orbhip_pose_optimization_device (pFrame->mpCamera2, pFrame->mTrl) */
typedef struct {
    double Trl[7];              /* mTrl as (qx,qy,qz,qw,tx,ty,tz) */
    double fx, fy, cx, cy;
    int32_t camera_model;       /* 0 Pinhole, 1 KannalaBrandt8 */
    double kb[4];
} orbhip_camera2;
int orbhip_pose_optimization_device(orbhip_ctx *ctx, const double *d_Xw, const double *d_obs,
                                    const double *d_inv_sigma2, const int32_t *d_n_edges, int frames, int max_edges,
                                    double fx, double fy, double cx, double cy, double bf, const double *kb8_k,
                                    const orbhip_camera2 *cam2, const uint8_t *d_right,
                                    double *d_pose, uint8_t *d_outlier, int32_t *d_n_inliers, int32_t *d_stats);
int orbhip_pose_optimization_host(orbhip_ctx *ctx, const double *Xw, const double *obs, const double *inv_sigma2, int n,
                                  double fx, double fy, double cx, double cy, double bf, const double *kb8_k,
                                  const orbhip_camera2 *cam2, const uint8_t *right,
                                  double *pose_inout, uint8_t *outlier_out, int32_t *n_inliers_out, int32_t *stats_out);

#ifdef __cplusplus
}
#endif
#endif
