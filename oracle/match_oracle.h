/*
 * match_oracle.h -- CPU ORACLE (test infrastructure, NOT product code) for the ORB matching
 * rows M1-M4 of SURVEY.md section 8a.  Restates /root/reference/src/ORBmatcher.cc and the
 * grid code of /root/reference/src/Frame.cc.  "parity unpinned" only at cv::BFMatcher's
 * tie order (un-vendored OpenCV; oracle rule: strict <, lowest index wins -- SURVEY A.11).
 */
#ifndef MATCH_ORACLE_H
#define MATCH_ORACLE_H
#include <stdint.h>
#include "orb_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif
/* ORBmatcher::DescriptorDistance, ORBmatcher.cc:2353-2369 (SWAR popcount, literal). */
int orc_descriptor_distance(const uint8_t *a32, const uint8_t *b32);
/* cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) + ratio test of Frame.cc:1146-1153.
 * idx2/dist2: [na][2], accept: [na]. */
void orc_bf2nn(const uint8_t *descA, int na, const uint8_t *descB, int nb, double ratio,
               int32_t *idx2, int32_t *dist2, uint8_t *accept);
/* ORBmatcher::SearchForInitialization (ORBmatcher.cc:710-825) over Frame grids built as
 * Frame::AssignFeaturesToGrid / GetFeaturesInArea (Frame.cc:377-408,645-726) with
 * mnMinX..mnMaxX = [min_x,max_x].  prev_matched [na][2] in/out; matches12 [na] out.
 * Returns nmatches. */
int orc_search_for_initialization(const orc_keypoint *kpA, const uint8_t *descA, int na,
                                  const orc_keypoint *kpB, const uint8_t *descB, int nb,
                                  float min_x, float min_y, float max_x, float max_y,
                                  int window_size, float nn_ratio, int check_orientation,
                                  float *prev_matched, int32_t *matches12);
/* Frame::GetFeaturesInArea on a grid built from kp[n] (test hook). Returns count. */
int orc_features_in_area(const orc_keypoint *kp, int n, float min_x, float min_y, float max_x,
                         float max_y, float x, float y, float r, int min_level, int max_level,
                         int32_t *out, int cap);
#ifdef __cplusplus
}
#endif
#endif
