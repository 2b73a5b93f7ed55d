/*
 * stereo_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See stereo_oracle.h.
 * "Frame" = /root/reference/src/Frame.cc.
 */
#include "stereo_oracle.h"
#include "match_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <stddef.h>

#define TH_HIGH 100   /* ORBmatcher.cc:40 */
#define TH_LOW 50     /* ORBmatcher.cc:41 */

struct di { int dist, idx; };
static int cmp_di(const void *a, const void *b)   /* std::sort of pair<int,int>, Frame:961 */
{
    const struct di *x = (const struct di *)a, *y = (const struct di *)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

int orc_compute_stereo_matches(const orc_extractor *eL, const orc_extractor *eR,
                               const orc_keypoint *kpL, const uint8_t *descL, int nL,
                               const orc_keypoint *kpR, const uint8_t *descR, int nR,
                               float mb, float mbf, float *u_right, float *depth, int32_t *sad)
{
    const float *sf = orc_scale_factors(eL), *isf = orc_inv_scale_factors(eL);
    for (int i = 0; i < nL; i++) { u_right[i] = -1.0f; depth[i] = -1.0f; if (sad) sad[i] = -1; }   /* Frame:804-805 */
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;                                                  /* Frame:807 */
    int w0, nRows, s0;
    orc_pyramid_level(eL, 0, &w0, &nRows, &s0);                                                    /* Frame:809 */
    /* row table, Frame:811-830 (rows outside the image are UB in the reference; skipped here) */
    int *row_cnt = (int *)calloc((size_t)nRows + 1, sizeof(int));
    int *minr = (int *)malloc(sizeof(int) * (nR ? nR : 1)), *maxr = (int *)malloc(sizeof(int) * (nR ? nR : 1));
    for (int iR = 0; iR < nR; iR++) {
        const float kpY = kpR[iR].y;
        const float r = 2.0f * sf[kpR[iR].octave];
        maxr[iR] = (int)ceilf(kpY + r);
        minr[iR] = (int)floorf(kpY - r);
        for (int yi = minr[iR]; yi <= maxr[iR]; yi++) if (yi >= 0 && yi < nRows) row_cnt[yi + 1]++;
    }
    for (int y = 0; y < nRows; y++) row_cnt[y + 1] += row_cnt[y];
    int *row_items = (int *)malloc(sizeof(int) * (row_cnt[nRows] ? row_cnt[nRows] : 1));
    int *fill = (int *)calloc((size_t)nRows, sizeof(int));
    for (int iR = 0; iR < nR; iR++)
        for (int yi = minr[iR]; yi <= maxr[iR]; yi++) if (yi >= 0 && yi < nRows) row_items[row_cnt[yi] + fill[yi]++] = iR;
    free(fill);
    const float minZ = mb, minD = 0, maxD = mbf / minZ;                                            /* Frame:833-835 */
    struct di *vDistIdx = (struct di *)malloc(sizeof(struct di) * (nL ? nL : 1));
    int nv = 0;
    for (int iL = 0; iL < nL; iL++) {
        const int levelL = kpL[iL].octave;
        const float vL = kpL[iL].y, uL = kpL[iL].x;
        const int row = (int)vL;                                                                   /* vRowIndices[vL], Frame:848 */
        if (row < 0 || row >= nRows) continue;
        const int c0 = row_cnt[row], c1 = row_cnt[row + 1];
        if (c0 == c1) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH, bestIdxR = 0;
        for (int c = c0; c < c1; c++) {                                                            /* Frame:865-887 */
            const int iR = row_items[c];
            if (kpR[iR].octave < levelL - 1 || kpR[iR].octave > levelL + 1) continue;
            const float uR = kpR[iR].x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_descriptor_distance(descL + 32 * (size_t)iL, descR + 32 * (size_t)iR);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (!(bestDist < thOrbDist)) continue;                                                     /* Frame:890 */
        const float uR0 = kpR[bestIdxR].x;
        const float scaleFactor = isf[levelL];
        const float scaleduL = roundf(uL * scaleFactor), scaledvL = roundf(vL * scaleFactor), scaleduR0 = roundf(uR0 * scaleFactor);
        const int w = 5, L = 5;
        int lw, lh, ls, rw, rh, rs;
        const uint8_t *IL = orc_pyramid_level(eL, levelL, &lw, &lh, &ls);   /* ROI views: negative columns read the padding */
        const uint8_t *IR = orc_pyramid_level(eR, levelL, &rw, &rh, &rs);
        const int cu = (int)scaleduL, cv = (int)scaledvL, cr = (int)scaleduR0;
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;                        /* Frame:913-916 */
        if (iniu < 0 || endu >= rw) continue;
        int best = INT_MAX, bestincR = 0;
        float vDists[11];
        const int cL = IL[(size_t)cv * ls + cu];
        for (int incR = -L; incR <= L; incR++) {                                                    /* Frame:918-936 */
            const int cR = IR[(ptrdiff_t)cv * rs + (cr + incR)];
            int s = 0;
            for (int dy = -w; dy <= w; dy++)
                for (int dx = -w; dx <= w; dx++) {
                    const int a = (int)IL[(ptrdiff_t)(cv + dy) * ls + (cu + dx)] - cL;
                    const int b = (int)IR[(ptrdiff_t)(cv + dy) * rs + (cr + incR + dx)] - cR;
                    s += abs(a - b);
                }
            const float dist = (float)s;
            if (dist < best) { best = (int)dist; bestincR = incR; }
            vDists[L + incR] = dist;
        }
        if (bestincR == -L || bestincR == L) continue;                                             /* Frame:938-939 */
        const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
        const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
        if (deltaR < -1 || deltaR > 1) continue;
        float bestuR = sf[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
        float disparity = (uL - bestuR);
        if (disparity >= minD && disparity < maxD) {                                                /* Frame:953-963 */
            if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
            depth[iL] = mbf / disparity;
            u_right[iL] = bestuR;
            vDistIdx[nv].dist = best; vDistIdx[nv].idx = iL; nv++;
            if (sad) sad[iL] = best;
        }
    }
    int kept = nv;
    if (nv > 0) {                                                                                   /* Frame:966-980 */
        qsort(vDistIdx, (size_t)nv, sizeof(struct di), cmp_di);
        const float median = vDistIdx[nv / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nv - 1; i >= 0; i--) {
            if (vDistIdx[i].dist < thDist) break;
            u_right[vDistIdx[i].idx] = -1; depth[vDistIdx[i].idx] = -1; kept--;
            if (sad) sad[vDistIdx[i].idx] = -1;
        }
    }
    free(vDistIdx); free(row_items); free(minr); free(maxr); free(row_cnt);
    return kept;
}
