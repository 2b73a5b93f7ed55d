/*
 * stereo_oracle.h -- CPU ORACLE (test infrastructure, NOT product code) for SURVEY 8f N2:
 * Frame::ComputeStereoMatches, /root/reference/src/Frame.cc:802-980 (rectified stereo):
 * row-band candidate search by Hamming distance, 11x11 SAD refinement over +-5 px on the pyramid level
 * of the left keypoint (reads the reflect-101 padding of mvImagePyramid for columns < 0), parabola
 * sub-pixel fit, disparity gates, and the final 1.5*1.4*median SAD filter.
 * "parity unpinned" against a real OpenCV build only through the pyramid bytes it reads.
 */
#ifndef STEREO_ORACLE_H
#define STEREO_ORACLE_H
#include <stdint.h>
#include "orb_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif
/* eL / eR: extractors that have just run orc_extract on the left / right image (their pyramids are read).
 * kp*: mvKeys / mvKeysRight (level order, lapping {0,0}), desc*: 32-byte rows.  mb, mbf as Frame::mb / mbf.
 * u_right / depth [nL] out (mvuRight / mvDepth, -1 = none).  sad [nL] out (optional): the SAD of the accepted
 * match, -1 if none -- test hook.  Returns the number of stereo matches kept. */
int orc_compute_stereo_matches(const orc_extractor *eL, const orc_extractor *eR,
                               const orc_keypoint *kpL, const uint8_t *descL, int nL,
                               const orc_keypoint *kpR, const uint8_t *descR, int nR,
                               float mb, float mbf, float *u_right, float *depth, int32_t *sad);
#ifdef __cplusplus
}
#endif
#endif
