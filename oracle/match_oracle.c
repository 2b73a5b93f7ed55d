/*
 * match_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See match_oracle.h.
 * "ORBm" = /root/reference/src/ORBmatcher.cc, "Frame" = /root/reference/src/Frame.cc.
 */
#include "match_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include <float.h>

#define GRID_COLS 64   /* FRAME_GRID_COLS, include/Frame.h:38 */
#define GRID_ROWS 48   /* FRAME_GRID_ROWS, include/Frame.h:39 */
#define TH_LOW 50      /* ORBm:41 */
#define HISTO_LENGTH 30 /* ORBm:42 */

/* ORBm:2353-2369 */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4); memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* Frame.cc:1146-1153 */
void orc_bf2nn(const uint8_t *A, int na, const uint8_t *B, int nb, double ratio,
               int32_t *idx2, int32_t *dist2, uint8_t *accept)
{
    for (int q = 0; q < na; q++) {
        int best = INT_MAX, second = INT_MAX, bi = -1, si = -1;
        for (int j = 0; j < nb; j++) {
            int d = orc_descriptor_distance(A + 32 * (size_t)q, B + 32 * (size_t)j);
            if (d < best) { second = best; si = bi; best = d; bi = j; }
            else if (d < second) { second = d; si = j; }
        }
        idx2[2 * q] = bi; idx2[2 * q + 1] = si; dist2[2 * q] = best; dist2[2 * q + 1] = second;
        accept[q] = (si >= 0 && (float)best < (float)second * ratio) ? 1 : 0;
    }
}

/* Frame grid: vectors of indices per cell, insertion order (Frame.cc:377-408, PosInGrid :716-726). */
struct grid {
    int *cell_start;   /* [GRID_COLS*GRID_ROWS+1], cell id = ix*GRID_ROWS+iy */
    int *items;
    float min_x, min_y, inv_w, inv_h;
};

static void grid_build(struct grid *g, const orc_keypoint *kp, int n, float min_x, float min_y, float max_x, float max_y)
{
    const int nc = GRID_COLS * GRID_ROWS;
    g->min_x = min_x; g->min_y = min_y;
    g->inv_w = (float)GRID_COLS / (max_x - min_x);       /* Frame.cc:334-335 */
    g->inv_h = (float)GRID_ROWS / (max_y - min_y);
    g->cell_start = (int *)calloc(nc + 1, sizeof(int));
    g->items = (int *)malloc(sizeof(int) * (n ? n : 1));
    int *cell = (int *)malloc(sizeof(int) * (n ? n : 1));
    for (int i = 0; i < n; i++) {
        int px = (int)roundf((kp[i].x - min_x) * g->inv_w);   /* Frame.cc:718-719: round, not floor */
        int py = (int)roundf((kp[i].y - min_y) * g->inv_h);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) cell[i] = -1;
        else { cell[i] = px * GRID_ROWS + py; g->cell_start[cell[i] + 1]++; }
    }
    for (int c = 0; c < nc; c++) g->cell_start[c + 1] += g->cell_start[c];
    int *fill = (int *)calloc(nc, sizeof(int));
    for (int i = 0; i < n; i++) if (cell[i] >= 0) g->items[g->cell_start[cell[i]] + fill[cell[i]]++] = i;
    free(fill); free(cell);
}
static void grid_free(struct grid *g) { free(g->cell_start); free(g->items); }

/* Frame::GetFeaturesInArea, Frame.cc:645-714 */
static int grid_query(const struct grid *g, const orc_keypoint *kp, float x, float y, float r,
                      int min_level, int max_level, int32_t *out, int cap)
{
    int n = 0;
    const float fx = r, fy = r;
    int c0 = (int)floorf((x - g->min_x - fx) * g->inv_w); if (c0 < 0) c0 = 0;
    if (c0 >= GRID_COLS) return 0;
    int c1 = (int)ceilf((x - g->min_x + fx) * g->inv_w); if (c1 > GRID_COLS - 1) c1 = GRID_COLS - 1;
    if (c1 < 0) return 0;
    int r0 = (int)floorf((y - g->min_y - fy) * g->inv_h); if (r0 < 0) r0 = 0;
    if (r0 >= GRID_ROWS) return 0;
    int r1 = (int)ceilf((y - g->min_y + fy) * g->inv_h); if (r1 > GRID_ROWS - 1) r1 = GRID_ROWS - 1;
    if (r1 < 0) return 0;
    const int check = (min_level > 0) || (max_level >= 0);
    for (int ix = c0; ix <= c1; ix++)
        for (int iy = r0; iy <= r1; iy++) {
            const int c = ix * GRID_ROWS + iy;
            for (int j = g->cell_start[c]; j < g->cell_start[c + 1]; j++) {
                const orc_keypoint *k = &kp[g->items[j]];
                if (check) {
                    if (k->octave < min_level) continue;
                    if (max_level >= 0 && k->octave > max_level) continue;
                }
                const float dx = k->x - x, dy = k->y - y;
                if (fabsf(dx) < fx && fabsf(dy) < fy) { if (n < cap) out[n] = g->items[j]; n++; }
            }
        }
    return n;
}

int orc_features_in_area(const orc_keypoint *kp, int n, float min_x, float min_y, float max_x, float max_y,
                         float x, float y, float r, int min_level, int max_level, int32_t *out, int cap)
{
    struct grid g;
    grid_build(&g, kp, n, min_x, min_y, max_x, max_y);
    int m = grid_query(&g, kp, x, y, r, min_level, max_level, out, cap);
    grid_free(&g);
    return m;
}

/* ORBm:2307-2348 */
static void three_maxima(const int *hist, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = hist[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* ORBm:710-825 */
int orc_search_for_initialization(const orc_keypoint *kpA, const uint8_t *descA, int na,
                                  const orc_keypoint *kpB, const uint8_t *descB, int nb,
                                  float min_x, float min_y, float max_x, float max_y,
                                  int window_size, float nn_ratio, int check_orientation,
                                  float *prev, int32_t *m12)
{
    int nmatches = 0;
    struct grid g;
    grid_build(&g, kpB, nb, min_x, min_y, max_x, max_y);
    int hist[HISTO_LENGTH]; memset(hist, 0, sizeof(hist));
    int *bin_of = (int *)malloc(sizeof(int) * (na ? na : 1));
    int *matched_dist = (int *)malloc(sizeof(int) * (nb ? nb : 1));
    int *m21 = (int *)malloc(sizeof(int) * (nb ? nb : 1));
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (nb ? nb : 1));
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < na; i++) { m12[i] = -1; bin_of[i] = -1; }
    for (int i = 0; i < nb; i++) { matched_dist[i] = INT_MAX; m21[i] = -1; }
    for (int i1 = 0; i1 < na; i1++) {
        const int level1 = kpA[i1].octave;
        if (level1 > 0) continue;
        int nc = grid_query(&g, kpB, prev[2 * i1], prev[2 * i1 + 1], (float)window_size, level1, level1, cand, nb);
        if (nc == 0) continue;
        int best = INT_MAX, best2 = INT_MAX, best_idx = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            const int dist = orc_descriptor_distance(descA + 32 * (size_t)i1, descB + 32 * (size_t)i2);
            if (matched_dist[i2] <= dist) continue;
            if (dist < best) { best2 = best; best = dist; best_idx = i2; }
            else if (dist < best2) best2 = dist;
        }
        if (best <= TH_LOW) {
            if (best < (float)best2 * nn_ratio) {
                if (m21[best_idx] >= 0) { m12[m21[best_idx]] = -1; nmatches--; }
                m12[i1] = best_idx; m21[best_idx] = i1; matched_dist[best_idx] = best; nmatches++;
                if (check_orientation) {
                    float rot = kpA[i1].angle - kpB[best_idx].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    hist[bin]++; bin_of[i1] = bin;
                }
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i1 = 0; i1 < na; i1++) {
            const int b = bin_of[i1];
            if (b < 0 || b == ind1 || b == ind2 || b == ind3) continue;
            if (m12[i1] >= 0) { m12[i1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < na; i1++)
        if (m12[i1] >= 0) { prev[2 * i1] = kpB[m12[i1]].x; prev[2 * i1 + 1] = kpB[m12[i1]].y; }
    grid_free(&g); free(bin_of); free(matched_dist); free(m21); free(cand);
    return nmatches;
}

/* ORBm:1965-2181, CurrentFrame.Nleft == -1 (monocular / rectified stereo) */
int orc_search_by_projection(const orc_proj_query *q, const uint8_t *desc_q, int nq,
                             const orc_keypoint *kp, const uint8_t *desc, const float *u_right, int n,
                             float min_x, float min_y, float max_x, float max_y,
                             int th_high, int check_orientation, int32_t *train_match)
{
    int nmatches = 0;
    struct grid g;
    grid_build(&g, kp, n, min_x, min_y, max_x, max_y);
    /* rotHist[bin] = list of bestIdx2 (duplicates possible), ORBm:1970-1973, 2083 */
    int *hist_n = (int *)calloc(HISTO_LENGTH, sizeof(int));
    int *hist_items = (int *)malloc(sizeof(int) * (size_t)HISTO_LENGTH * (nq ? nq : 1));
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < n; i++) if (train_match[i] != -1) train_match[i] = -2;
    for (int t = 0; t < nq; t++) {
        const float radius = q[t].radius;
        const int nc = grid_query(&g, kp, q[t].u, q[t].v, radius, q[t].min_level, q[t].max_level, cand, n);
        if (nc == 0) continue;                                             /* ORBm:2025-2026 */
        int best_dist = 256, best_idx = -1;                                /* ORBm:2030-2031 */
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            const int h = train_match[i2];
            if (h <= -2 || (h >= 0 && q[h].has_obs)) continue;             /* ORBm:2037-2039 */
            if (u_right && u_right[i2] > 0) {                              /* ORBm:2041-2047 */
                const float er = fabsf(q[t].ur - u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = orc_descriptor_distance(desc_q + 32 * (size_t)t, desc + 32 * (size_t)i2);
            if (dist < best_dist) { best_dist = dist; best_idx = i2; }
        }
        if (best_dist <= th_high) {                                        /* ORBm:2058-2086 */
            train_match[best_idx] = t;
            nmatches++;
            if (check_orientation) {
                float rot = q[t].angle - kp[best_idx].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                hist_items[(size_t)bin * nq + hist_n[bin]++] = best_idx;
            }
        }
    }
    if (check_orientation) {                                               /* ORBm:2156-2178 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(hist_n, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hist_n[i]; j++) { train_match[hist_items[(size_t)i * nq + j]] = -1; nmatches--; }
    }
    free(cand); free(hist_items); free(hist_n); grid_free(&g);
    return nmatches;
}

/* ORBm:48-218, F.Nleft == -1 */
int orc_search_by_projection_map(const orc_proj_query *q, const uint8_t *desc_q, int nq,
                                 const orc_keypoint *kp, const uint8_t *desc, const float *u_right, int n,
                                 float min_x, float min_y, float max_x, float max_y,
                                 int th_high, float nn_ratio, int32_t *train_match)
{
    int nmatches = 0;
    struct grid g;
    grid_build(&g, kp, n, min_x, min_y, max_x, max_y);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    for (int i = 0; i < n; i++) if (train_match[i] != -1) train_match[i] = -2;
    for (int t = 0; t < nq; t++) {
        const float radius = q[t].radius;
        const int nc = grid_query(&g, kp, q[t].u, q[t].v, radius, q[t].min_level, q[t].max_level, cand, n);   /* ORBm:78-79 */
        if (nc == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;                    /* ORBm:85-89 */
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            const int h = train_match[idx];
            if (h <= -2 || (h >= 0 && q[h].has_obs)) continue;                                                 /* ORBm:96-98 */
            if (u_right && u_right[idx] > 0) {                                                                 /* ORBm:100-105 */
                const float er = fabsf(q[t].ur - u_right[idx]);
                if (er > radius) continue;
            }
            const int dist = orc_descriptor_distance(desc_q + 32 * (size_t)t, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kp[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = kp[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= th_high) {                                                                             /* ORBm:131-150 */
            if (bestLevel == bestLevel2 && bestDist > nn_ratio * bestDist2) continue;
            train_match[bestIdx] = t;
            nmatches++;
        }
    }
    free(cand); grid_free(&g);
    return nmatches;
}

static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
/* MapPoint.cc:366-397 */
int orc_distinctive_descriptor(const uint8_t *desc, int n)
{
    if (n <= 0) return 0;
    int *row = (int *)malloc(sizeof(int) * (size_t)n);
    int best_median = INT_MAX, best_idx = 0;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) row[j] = i == j ? 0 : orc_descriptor_distance(desc + 32 * (size_t)i, desc + 32 * (size_t)j);
        qsort(row, (size_t)n, sizeof(int), cmp_int);
        const int median = row[(int)(0.5 * (n - 1))];
        if (median < best_median) { best_median = median; best_idx = i; }
    }
    free(row);
    return best_idx;
}

/* TemplatedVocabulary.h:1218-1260 */
void orc_bow_transform(const uint8_t *f, const uint8_t *node_desc, const int32_t *child_start, const int32_t *child_ids,
                       const int32_t *node_word, const double *node_weight, int L, int levelsup,
                       int32_t *word_id, double *weight, int32_t *nid)
{
    const int nid_level = L - levelsup;
    if (nid_level <= 0 && nid) *nid = 0;
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const int c0 = child_start[final_id], c1 = child_start[final_id + 1];
        final_id = child_ids[c0];
        double best_d = orc_descriptor_distance(f, node_desc + 32 * (size_t)final_id);
        for (int c = c0 + 1; c < c1; c++) {
            const int id = child_ids[c];
            const double d = orc_descriptor_distance(f, node_desc + 32 * (size_t)id);
            if (d < best_d) { best_d = d; final_id = id; }
        }
        if (nid && current_level == nid_level) *nid = final_id;
    } while (child_start[final_id + 1] > child_start[final_id]);
    *word_id = node_word[final_id];
    *weight = node_weight[final_id];
}

/* ORBm:1499-1570 (NLeft == -1) */
void orc_fuse_search(const orc_proj_query *q, const uint8_t *desc_q, int nq,
                     const orc_keypoint *kp, const uint8_t *desc, const float *u_right, int n,
                     const float *inv_level_sigma2, float min_x, float min_y, float max_x, float max_y,
                     int32_t *best_idx, int32_t *best_dist)
{
    struct grid g;
    grid_build(&g, kp, n, min_x, min_y, max_x, max_y);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    for (int t = 0; t < nq; t++) {
        const int nc = grid_query(&g, kp, q[t].u, q[t].v, q[t].radius, -1, -1, cand, n);     /* ORBm:1503 */
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            const int kpLevel = kp[idx].octave;
            if (kpLevel < q[t].min_level || kpLevel > q[t].max_level) continue;              /* ORBm:1527-1528 */
            if (u_right && u_right[idx] >= 0) {                                              /* ORBm:1530-1545 */
                const float ex = q[t].u - kp[idx].x, ey = q[t].v - kp[idx].y, er = q[t].ur - u_right[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
            } else {                                                                         /* ORBm:1546-1556 */
                const float ex = q[t].u - kp[idx].x, ey = q[t].v - kp[idx].y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
            }
            const int dist = orc_descriptor_distance(desc_q + 32 * (size_t)t, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_idx[t] = bestIdx; best_dist[t] = bestDist;
    }
    free(cand); grid_free(&g);
}

/* ORBm:273-475, F.Nleft == -1 */
int orc_search_by_bow(const int32_t *kf_node_ids, const int32_t *kf_node_start, const int32_t *kf_feat, int kf_nnodes,
                      const uint8_t *kf_valid, const orc_keypoint *kf_kp, const uint8_t *kf_desc,
                      const int32_t *f_node_ids, const int32_t *f_node_start, const int32_t *f_feat, int f_nnodes,
                      const orc_keypoint *f_kp, const uint8_t *f_desc, int nF,
                      float nn_ratio, int check_orientation, int32_t *match_f)
{
    int nmatches = 0;
    int *hist_n = (int *)calloc(HISTO_LENGTH, sizeof(int));
    int *hist_items = (int *)malloc(sizeof(int) * (size_t)HISTO_LENGTH * (nF ? nF : 1));
    const float factor = 1.0f / HISTO_LENGTH;
    for (int j = 0; j < nF; j++) match_f[j] = -1;
    int a = 0, b = 0;
    while (a < kf_nnodes && b < f_nnodes) {                                   /* ORBm:291-443 */
        if (kf_node_ids[a] == f_node_ids[b]) {
            for (int ik = kf_node_start[a]; ik < kf_node_start[a + 1]; ik++) {
                const int realIdxKF = kf_feat[ik];
                if (!kf_valid[realIdxKF]) continue;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int jf = f_node_start[b]; jf < f_node_start[b + 1]; jf++) {
                    const int realIdxF = f_feat[jf];
                    if (match_f[realIdxF] >= 0) continue;                     /* ORBm:321-322 */
                    const int dist = orc_descriptor_distance(kf_desc + 32 * (size_t)realIdxKF, f_desc + 32 * (size_t)realIdxF);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW && (float)bestDist1 < nn_ratio * (float)bestDist2) {
                    match_f[bestIdxF] = realIdxKF;
                    if (check_orientation) {
                        float rot = kf_kp[realIdxKF].angle - f_kp[bestIdxF].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        hist_items[(size_t)bin * nF + hist_n[bin]++] = bestIdxF;
                    }
                    nmatches++;
                }
            }
            a++; b++;
        } else if (kf_node_ids[a] < f_node_ids[b]) { while (a < kf_nnodes && kf_node_ids[a] < f_node_ids[b]) a++; }   /* lower_bound */
        else { while (b < f_nnodes && f_node_ids[b] < kf_node_ids[a]) b++; }
    }
    if (check_orientation) {                                                  /* ORBm:445-470 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(hist_n, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hist_n[i]; j++) { match_f[hist_items[(size_t)i * nF + j]] = -1; nmatches--; }
    }
    free(hist_items); free(hist_n);
    return nmatches;
}

/* Pinhole::epipolarConstrain, Pinhole.cpp:129-143, F12 given */
static int epipolar_ok(const float F[9], const orc_keypoint *k1, const orc_keypoint *k2, float unc)
{
    const float a = k1->x * F[0] + k1->y * F[3] + F[6];
    const float b = k1->x * F[1] + k1->y * F[4] + F[7];
    const float c = k1->x * F[2] + k1->y * F[5] + F[8];
    const float num = a * k2->x + b * k2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * unc;
}
/* ORBm:969-1210 (mpCamera2 == 0, Pinhole) */
int orc_search_for_triangulation(const int32_t *nid1, const uint8_t *has_mp1, const orc_keypoint *kp1, const uint8_t *desc1,
                                 const float *u_right1, int n1,
                                 const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2,
                                 const uint8_t *has_mp2, const orc_keypoint *kp2, const uint8_t *desc2, const float *u_right2,
                                 const float F12[9], float ep_x, float ep_y, const float *scale_factors, const float *level_sigma2,
                                 int only_stereo, int coarse, int check_orientation, int32_t *matches12)
{
    int nmatches = 0;
    int hist[HISTO_LENGTH]; memset(hist, 0, sizeof(hist));
    int *bin_of = (int *)malloc(sizeof(int) * (n1 ? n1 : 1));
    const float factor = 1.0f / HISTO_LENGTH;
    for (int idx1 = 0; idx1 < n1; idx1++) {                         /* node-major in the reference; the keypoints are independent */
        matches12[idx1] = -1; bin_of[idx1] = -1;
        if (has_mp1[idx1]) continue;                                /* ORBm:1029-1034 */
        const int bStereo1 = u_right1 && u_right1[idx1] >= 0;
        if (only_stereo && !bStereo1) continue;
        int lo = 0, hi = nnodes2;                                   /* the node of idx1 in KF2's FeatureVector */
        while (lo < hi) { const int mid = (lo + hi) / 2; if (node_ids2[mid] < nid1[idx1]) lo = mid + 1; else hi = mid; }
        if (lo >= nnodes2 || node_ids2[lo] != nid1[idx1]) continue;
        int bestDist = TH_LOW, bestIdx2 = -1;                       /* ORBm:1048-1049 */
        for (int j = node_start2[lo]; j < node_start2[lo + 1]; j++) {
            const int idx2 = feat2[j];
            if (has_mp2[idx2]) continue;                            /* vbMatched2 is never set in this fork */
            const int bStereo2 = u_right2 && u_right2[idx2] >= 0;
            if (only_stereo && !bStereo2) continue;
            const int dist = orc_descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
            if (dist > TH_LOW || dist > bestDist) continue;         /* ORBm:1073-1074 */
            if (!bStereo1 && !bStereo2) {                           /* ORBm:1083-1091 */
                const float distex = ep_x - kp2[idx2].x, distey = ep_y - kp2[idx2].y;
                if (distex * distex + distey * distey < 100 * scale_factors[kp2[idx2].octave]) continue;
            }
            if (epipolar_ok(F12, &kp1[idx1], &kp2[idx2], level_sigma2[kp2[idx2].octave]) || coarse) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
            matches12[idx1] = bestIdx2; nmatches++;
            if (check_orientation) {
                float rot = kp1[idx1].angle - kp2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                hist[bin]++; bin_of[idx1] = bin;
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < n1; i++)
            if (bin_of[i] >= 0 && bin_of[i] != ind1 && bin_of[i] != ind2 && bin_of[i] != ind3) { matches12[i] = -1; nmatches--; }
    }
    free(bin_of);
    return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12), ORBm:827-967, NLeft == -1 */
int orc_search_by_bow_kf(const int32_t *node_ids1, const int32_t *node_start1, const int32_t *feat1, int nnodes1,
                         const uint8_t *valid1, const orc_keypoint *kp1, const uint8_t *desc1, int n1,
                         const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2,
                         const uint8_t *valid2, const orc_keypoint *kp2, const uint8_t *desc2, int n2,
                         float nn_ratio, int check_orientation, int32_t *matches12)
{
    int nmatches = 0;
    int *hist_n = (int *)calloc(HISTO_LENGTH, sizeof(int));
    int *hist_items = (int *)malloc(sizeof(int) * (size_t)HISTO_LENGTH * (n1 ? n1 : 1));
    uint8_t *matched2 = (uint8_t *)calloc(n2 ? n2 : 1, 1);                    /* vbMatched2, :840 */
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    int a = 0, b = 0;
    while (a < nnodes1 && b < nnodes2) {                                      /* :856-942 */
        if (node_ids1[a] == node_ids2[b]) {
            for (int i1 = node_start1[a]; i1 < node_start1[a + 1]; i1++) {
                const int idx1 = feat1[i1];
                if (!valid1[idx1]) continue;                                  /* :867-871 */
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int i2 = node_start2[b]; i2 < node_start2[b + 1]; i2++) {
                    const int idx2 = feat2[i2];
                    if (matched2[idx2] || !valid2[idx2]) continue;            /* :887-891 */
                    const int dist = orc_descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW && (float)bestDist1 < nn_ratio * (float)bestDist2) {   /* :909-911, strict */
                    matches12[idx1] = bestIdx2; matched2[bestIdx2] = 1;
                    if (check_orientation) {
                        float rot = kp1[idx1].angle - kp2[bestIdx2].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        hist_items[(size_t)bin * n1 + hist_n[bin]++] = idx1;
                    }
                    nmatches++;
                }
            }
            a++; b++;
        } else if (node_ids1[a] < node_ids2[b]) { while (a < nnodes1 && node_ids1[a] < node_ids2[b]) a++; }
        else { while (b < nnodes2 && node_ids2[b] < node_ids1[a]) b++; }
    }
    if (check_orientation) {                                                  /* :944-963 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(hist_n, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hist_n[i]; j++) { matches12[hist_items[(size_t)i * n1 + j]] = -1; nmatches--; }
    }
    free(matched2); free(hist_items); free(hist_n);
    return nmatches;
}

/* The search loop of ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th, ratioHamming), ORBm:553-595
 * (and of its vpPointsKFs twin, :676-705): queries in order, candidates from KeyFrame::GetFeaturesInArea (KeyFrame.cc:770-814),
 * keypoints that already hold a point are skipped, octave in [pred-1, pred], first minimum, claim if <= TH_LOW*ratioHamming. */
int orc_search_by_projection_sim3(const orc_proj_query *q, const uint8_t *desc_q, int nq,
                                  const orc_keypoint *kp, const uint8_t *desc, int n,
                                  float min_x, float min_y, float max_x, float max_y, float ratio_hamming, int32_t *matched)
{
    struct grid g;
    grid_build(&g, kp, n, min_x, min_y, max_x, max_y);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    int nmatches = 0;
    for (int t = 0; t < nq; t++) {
        const int nc = grid_query(&g, kp, q[t].u, q[t].v, q[t].radius, -1, -1, cand, n);     /* ORBm:541 */
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (matched[idx] != -1) continue;                                                /* ORBm:556-557 */
            const int kpLevel = kp[idx].octave;
            if (kpLevel < q[t].min_level || kpLevel > q[t].max_level) continue;              /* ORBm:561-562 */
            const int dist = orc_descriptor_distance(desc_q + 32 * (size_t)t, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW * ratio_hamming) { matched[bestIdx] = t; nmatches++; }        /* ORBm:575-579 */
    }
    free(cand); grid_free(&g);
    return nmatches;
}

/* The per-point search of ORBmatcher::SearchBySim3 (ORBm:1825-1851, 1905-1931) and of Fuse(KeyFrame*, Scw, ...) (ORBm:1697-1720):
 * independent queries, octave in [pred-1, pred], first minimum, no gate.  best_dist = INT_MAX when there is no candidate. */
void orc_window_best(const orc_proj_query *q, const uint8_t *desc_q, int nq, const orc_keypoint *kp, const uint8_t *desc, int n,
                     float min_x, float min_y, float max_x, float max_y, int32_t *best_idx, int32_t *best_dist)
{
    struct grid g;
    grid_build(&g, kp, n, min_x, min_y, max_x, max_y);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    for (int t = 0; t < nq; t++) {
        const int nc = grid_query(&g, kp, q[t].u, q[t].v, q[t].radius, -1, -1, cand, n);
        int bestDist = 2147483647, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (kp[idx].octave < q[t].min_level || kp[idx].octave > q[t].max_level) continue;
            const int dist = orc_descriptor_distance(desc_q + 32 * (size_t)t, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_idx[t] = bestIdx; best_dist[t] = bestDist;
    }
    free(cand); grid_free(&g);
}

/* Frame::AssignFeaturesToGrid, Frame.cc:377-408 (Nleft == -1): the 64x48 cell lists as a CSR, cells numbered ix*48+iy */
void orc_assign_features_to_grid(const orc_keypoint *kp, int n, float min_x, float min_y, float max_x, float max_y,
                                 int32_t *cell_start, int32_t *items)
{
    struct grid g;
    grid_build(&g, kp, n, min_x, min_y, max_x, max_y);
    for (int c = 0; c <= GRID_COLS * GRID_ROWS; c++) cell_start[c] = g.cell_start[c];
    for (int i = 0; i < g.cell_start[GRID_COLS * GRID_ROWS]; i++) items[i] = g.items[i];
    grid_free(&g);
}

/* Frame::UndistortKeyPoints, Frame.cc:738-771: cv::undistortPoints(mat, mat, K, mDistCoef, Mat(), K) (OpenCV 3.4.1
 * cvUndistortPoints, un-vendored: restated -- 5 fixed-point iterations of the inverse Brown model in double, output rounded to
 * float; k = (k1, k2, p1, p2, k3), R = I, P = K).  parity unpinned (no OpenCV here). */
void orc_undistort_keypoints(const orc_keypoint *kp, int n, float fx_, float fy_, float cx_, float cy_, const float *dist, int ndist,
                             orc_keypoint *out)
{
    double k[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < ndist && i < 5; i++) k[i] = dist[i];
    const double fx = fx_, fy = fy_, cx = cx_, cy = cy_, ifx = 1. / fx, ify = 1. / fy;
    for (int i = 0; i < n; i++) {
        out[i] = kp[i];
        if (dist[0] == 0.0f) continue;                             /* Frame.cc:740-744 */
        double x = kp[i].x, y = kp[i].y;
        x = (x - cx) * ifx; y = (y - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        /* RR = P * R = K:  xx = fx x + cx, yy = fy y + cy, ww = 1 */
        const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
        out[i].x = (float)(xx * ww); out[i].y = (float)(yy * ww);
    }
}

/* TemplatedVocabulary::transform(features, v, fv, levelsup), TemplatedVocabulary.h:1139-1208 (TF_IDF, L1) after the per-feature
 * tree descent: std::map semantics restated with sorted arrays. */
void orc_bow_vectors(const int32_t *wid, const double *w, const int32_t *nid, int n,
                     int32_t *node_ids, int32_t *node_start, int32_t *feat, int32_t *nnodes,
                     int32_t *bow_word, double *bow_value, int32_t *nwords)
{
    int nn = 0, nw = 0;
    /* FeatureVector: map<NodeId, vector<i_feature>>; BowVector: map<WordId, double> */
    int32_t *fv_cnt = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
    int32_t **fv_list = (int32_t **)calloc(n ? n : 1, sizeof(int32_t *));
    for (int i = 0; i < n; i++) {
        if (!(w[i] > 0)) continue;                                  /* :1170 not stopped */
        {   /* v.addWeight(id, w), BowVector.cpp */
            int lo = 0;
            while (lo < nw && bow_word[lo] < wid[i]) lo++;
            if (lo < nw && bow_word[lo] == wid[i]) bow_value[lo] += w[i];
            else {
                for (int k = nw; k > lo; k--) { bow_word[k] = bow_word[k - 1]; bow_value[k] = bow_value[k - 1]; }
                bow_word[lo] = wid[i]; bow_value[lo] = w[i]; nw++;
            }
        }
        {   /* fv.addFeature(nid, i_feature), FeatureVector.cpp */
            int lo = 0;
            while (lo < nn && node_ids[lo] < nid[i]) lo++;
            if (!(lo < nn && node_ids[lo] == nid[i])) {
                for (int k = nn; k > lo; k--) { node_ids[k] = node_ids[k - 1]; fv_cnt[k] = fv_cnt[k - 1]; fv_list[k] = fv_list[k - 1]; }
                node_ids[lo] = nid[i]; fv_cnt[lo] = 0; fv_list[lo] = (int32_t *)malloc(sizeof(int32_t) * n); nn++;
            }
            fv_list[lo][fv_cnt[lo]++] = i;
        }
    }
    double norm = 0.0;                                              /* v.normalize(L1), BowVector.cpp */
    for (int k = 0; k < nw; k++) norm += fabs(bow_value[k]);
    if (norm > 0.0) for (int k = 0; k < nw; k++) bow_value[k] /= norm;
    int pos = 0;
    for (int k = 0; k < nn; k++) {
        node_start[k] = pos;
        for (int j = 0; j < fv_cnt[k]; j++) feat[pos++] = fv_list[k][j];
        free(fv_list[k]);
    }
    node_start[nn] = pos;
    *nnodes = nn; *nwords = nw;
    free(fv_list); free(fv_cnt);
}

/* ---- frames of a two-camera rig (Nleft != -1; TUM-VI stereo-fisheye, Frame.cc:1034-1126) ---------------------------------------
 * Keypoints [0, nleft) = F.mvKeys, [nleft, n) = F.mvKeysRight; mGrid / mGridRight (Frame.cc:395-405, :686).  has_obs bit 1 of a query =
 * the search runs in the right camera (GetFeaturesInArea(..., bRight = true)); candidates are frame-wide indices (idx + Nleft). */
/* ORBm:1965-2181 with CurrentFrame.Nleft != -1 (mode 0) and ORBm:48-218 with F.Nleft != -1 (mode 1); the caller lists a point's left
 * query and then its right query (ORBm:2089-2153 / :149-214) */
int orc_search_by_projection_rig(int mode, const orc_proj_query *q, const uint8_t *desc_q, int nq,
                                 const orc_keypoint *kp, const uint8_t *desc, int n, int nleft, const int32_t *mirror,
                                 float min_x, float min_y, float max_x, float max_y,
                                 int th_high, float nn_ratio, int check_orientation, int32_t *train_match)
{
    int nmatches = 0;
    struct grid g[2];
    grid_build(&g[0], kp, nleft, min_x, min_y, max_x, max_y);
    grid_build(&g[1], kp + nleft, n - nleft, min_x, min_y, max_x, max_y);
    int *hist_n = (int *)calloc(HISTO_LENGTH, sizeof(int));
    int *hist_items = (int *)malloc(sizeof(int) * (size_t)HISTO_LENGTH * (nq ? nq : 1));
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < n; i++) if (train_match[i] != -1) train_match[i] = -2;
    for (int t = 0; t < nq; t++) {
        const int cam = (q[t].has_obs >> 1) & 1, off = cam ? nleft : 0;
        const float radius = q[t].radius;
        const int nc = grid_query(&g[cam], kp + off, q[t].u, q[t].v, radius, q[t].min_level, q[t].max_level, cand, n);
        if (nc == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c] + off;
            const int h = train_match[idx];
            if (h <= -2 || (h >= 0 && (q[h].has_obs & 1))) continue;                                           /* ORBm:110-112, 174-176, 2037-2039, 2120-2122 */
            const int dist = orc_descriptor_distance(desc_q + 32 * (size_t)t, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kp[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = kp[idx].octave; bestDist2 = dist; }
        }
        if (bestDist > th_high) continue;
        if (mode == 1) {                                                                                       /* ORBm:131-150, 197-212 */
            if (bestLevel == bestLevel2 && bestDist > nn_ratio * bestDist2) continue;
            train_match[bestIdx] = t; nmatches++;
            if (mirror && mirror[bestIdx] >= 0) { train_match[mirror[bestIdx]] = t; nmatches++; }
        } else {                                                                                               /* ORBm:2058-2086, 2135-2152 */
            train_match[bestIdx] = t; nmatches++;
            if (check_orientation) {
                float rot = q[t].angle - kp[bestIdx].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                hist_items[(size_t)bin * nq + hist_n[bin]++] = bestIdx;
            }
        }
    }
    if (mode == 0 && check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(hist_n, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hist_n[i]; j++) { train_match[hist_items[(size_t)i * nq + j]] = -1; nmatches--; }
    }
    free(cand); free(hist_items); free(hist_n); grid_free(&g[0]); grid_free(&g[1]);
    return nmatches;
}

/* ORBm:273-475 with F.Nleft != -1 */
int orc_search_by_bow_rig(const int32_t *kf_node_ids, const int32_t *kf_node_start, const int32_t *kf_feat, int kf_nnodes,
                          const uint8_t *kf_valid, const orc_keypoint *kf_kp, const uint8_t *kf_desc,
                          const int32_t *f_node_ids, const int32_t *f_node_start, const int32_t *f_feat, int f_nnodes,
                          const orc_keypoint *f_kp, const uint8_t *f_desc, int nF, int nleft,
                          float nn_ratio, int check_orientation, int32_t *match_f)
{
    int nmatches = 0;
    int *hist_n = (int *)calloc(HISTO_LENGTH, sizeof(int));
    int *hist_items = (int *)malloc(sizeof(int) * (size_t)HISTO_LENGTH * (nF ? nF : 1));
    const float factor = 1.0f / HISTO_LENGTH;
    for (int j = 0; j < nF; j++) match_f[j] = -1;
    int a = 0, b = 0;
    while (a < kf_nnodes && b < f_nnodes) {
        if (kf_node_ids[a] == f_node_ids[b]) {
            for (int ik = kf_node_start[a]; ik < kf_node_start[a + 1]; ik++) {
                const int realIdxKF = kf_feat[ik];
                if (!kf_valid[realIdxKF]) continue;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256, bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
                for (int jf = f_node_start[b]; jf < f_node_start[b + 1]; jf++) {                               /* ORBm:338-359 */
                    const int realIdxF = f_feat[jf];
                    if (match_f[realIdxF] >= 0) continue;
                    const int dist = orc_descriptor_distance(kf_desc + 32 * (size_t)realIdxKF, f_desc + 32 * (size_t)realIdxF);
                    if (realIdxF < nleft && dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (realIdxF < nleft && dist < bestDist2) bestDist2 = dist;
                    if (realIdxF >= nleft && dist < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = dist; bestIdxFR = realIdxF; }
                    else if (realIdxF >= nleft && dist < bestDist2R) bestDist2R = dist;
                }
                if (bestDist1 <= TH_LOW) {                                                                     /* ORBm:362-426 */
                    for (int side = 0; side < 2; side++) {
                        int idx;
                        if (side == 0) { if (!((float)bestDist1 < nn_ratio * (float)bestDist2)) continue; idx = bestIdxF; }
                        else { if (bestDist1R > TH_LOW) continue; idx = bestIdxFR; }                           /* "|| true": no ratio test */
                        match_f[idx] = realIdxKF;
                        if (check_orientation) {
                            float rot = kf_kp[realIdxKF].angle - f_kp[idx].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)roundf(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            hist_items[(size_t)bin * nF + hist_n[bin]++] = idx;
                        }
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (kf_node_ids[a] < f_node_ids[b]) { while (a < kf_nnodes && kf_node_ids[a] < f_node_ids[b]) a++; }
        else { while (b < f_nnodes && f_node_ids[b] < kf_node_ids[a]) b++; }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(hist_n, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < hist_n[i]; j++) { match_f[hist_items[(size_t)i * nF + j]] = -1; nmatches--; }
    }
    free(hist_items); free(hist_n);
    return nmatches;
}

/* ================================================================= SearchForTriangulation, general form
 * ORBmatcher::SearchForTriangulation (ORBm:969-1210) for every camera combination the reference supports: Pinhole or
 * KannalaBrandt8 single cameras and two-camera rigs (mpCamera2 != 0, NLeft != -1: keypoints mvKeys | mvKeysRight, the relative
 * pose and the camera pair picked per candidate from {ll, lr, rl, rr}, ORBm:1092-1121).  GeometricCamera::epipolarConstrain:
 * Pinhole.cpp:122-144 (distance to the epipolar line of F12 = K1^-T [t12]x R12 K2^-1) and KannalaBrandt8.cpp:235-238 ->
 * TriangulateMatches (:334-401): ray parallax, linear triangulation by the SVD of a 4x4 system, positive depths, reprojection
 * errors in both cameras.
 * UNPINNED parts restated from OpenCV 3.4.1 (not in the reference tree): cv::SVD::compute on a 4x4 CV_32F matrix = one-sided Jacobi
 * (JacobiSVDImpl_<float>: rotations of the rows of A^T in float, norms / dot products accumulated in double, eps = 2*FLT_EPSILON,
 * at most 30 sweeps, singular vectors sorted by decreasing singular value); cv::Mat float products accumulate in double and round
 * once; Mat / scalar multiplies by the float reciprocal.  Stated deviations, applied identically in the HIP kernel (as for
 * orb_sincos): tanf / cosf / sinf / atan2f of the platform libm are replaced by the float rounding of fixed double
 * sequences (Cody-Waite + fdlibm kernels; double atan2), hypot(p, beta) of the Jacobi rotation by sqrt(p*p + beta*beta). */
static void sincos_signed(double x, double *s_out, double *c_out)
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
    case 0: *s_out = s; *c_out = c; break;
    case 1: *s_out = c; *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
    }
}
static float det_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }

/* GeometricCamera::project(cv::Point3f): Pinhole.cpp:34-37, KannalaBrandt8.cpp:28-45.  p = fx fy cx cy k1 k2 k3 k4 */
void orc_camera_project_f(int type, const float *p, const float P[3], float uv[2])
{
    if (type == 0) { uv[0] = p[0] * P[0] / P[2] + p[2]; uv[1] = p[1] * P[1] / P[2] + p[3]; return; }
    const float x2_plus_y2 = P[0] * P[0] + P[1] * P[1];
    const float theta = det_atan2f(sqrtf(x2_plus_y2), P[2]);
    const float psi = det_atan2f(P[1], P[0]);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + p[4] * theta3 + p[5] * theta5 + p[6] * theta7 + p[7] * theta9;
    double s, c;
    sincos_signed((double)psi, &s, &c);
    uv[0] = p[0] * r * (float)c + p[2]; uv[1] = p[1] * r * (float)s + p[3];
}
/* GeometricCamera::unproject: Pinhole.cpp:57-60, KannalaBrandt8.cpp:103-130 */
void orc_camera_unproject_f(int type, const float *p, float u, float v, float ray[3])
{
    const float pwx = (u - p[2]) / p[0], pwy = (v - p[3]) / p[1];
    if (type == 0) { ray[0] = pwx; ray[1] = pwy; ray[2] = 1.f; return; }
    float scale = 1.f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    theta_d = fminf(fmaxf((float)(-M_PI / 2.f), theta_d), (float)(M_PI / 2.f));     /* (CV_PI / 2.f is a double expression, narrowed by fminf / fmaxf) */
    if (theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            const float k0_theta2 = p[4] * theta2, k1_theta4 = p[5] * theta4, k2_theta6 = p[6] * theta6, k3_theta8 = p[7] * theta8;
            const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                    (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < 1e-6f) break;                                   /* precision(1e-6), KannalaBrandt8.h:53 */
        }
        double s, c;
        sincos_signed((double)theta, &s, &c);
        scale = (float)(s / c) / theta_d;                                          /* std::tan(theta) / theta_d */
    }
    ray[0] = pwx * scale; ray[1] = pwy * scale; ray[2] = 1.f;
}

/* Vt of cv::SVD::compute(A 4x4 CV_32F, FULL_UV): rows = right singular vectors by decreasing singular value (lapack.cpp JacobiSVDImpl_) */
static void jacobi_svd4_vt(const float A[16], float Vt[16])
{
    float At[16];
    double W[4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) At[4 * i + j] = A[4 * j + i];          /* transpose(src, temp_a): m == n */
    const float eps = FLT_EPSILON * 2;
    for (int i = 0; i < 4; i++) {
        double sd = 0;
        for (int k = 0; k < 4; k++) { const float t = At[4 * i + k]; sd += (double)t * t; }
        W[i] = sd;
        for (int k = 0; k < 4; k++) Vt[4 * i + k] = 0;
        Vt[4 * i + i] = 1;
    }
    for (int iter = 0; iter < 30; iter++) {
        int changed = 0;
        for (int i = 0; i < 3; i++)
            for (int j = i + 1; j < 4; j++) {
                float *Ai = At + 4 * i, *Aj = At + 4 * j;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < 4; k++) p += (double)Ai[k] * Aj[k];
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
                for (int k = 0; k < 4; k++) {
                    const float t0 = c * Ai[k] + s * Aj[k];
                    const float t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += (double)t0 * t0; b += (double)t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = 1;
                float *Vi = Vt + 4 * i, *Vj = Vt + 4 * j;
                for (int k = 0; k < 4; k++) {
                    const float t0 = c * Vi[k] + s * Vj[k];
                    const float t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0; Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < 4; i++) {
        double sd = 0;
        for (int k = 0; k < 4; k++) { const float t = At[4 * i + k]; sd += (double)t * t; }
        W[i] = sqrt(sd);
    }
    for (int i = 0; i < 3; i++) {
        int j = i;
        for (int k = i + 1; k < 4; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            const double tw = W[i]; W[i] = W[j]; W[j] = tw;
            for (int k = 0; k < 4; k++) { const float t = Vt[4 * i + k]; Vt[4 * i + k] = Vt[4 * j + k]; Vt[4 * j + k] = t; }
        }
    }
}

/* KannalaBrandt8::TriangulateMatches (KannalaBrandt8.cpp:334-401); returns z1 or -1.  R12 row-major 3x3, t12[3] */
float orc_kb8_triangulate_matches(int type1, const float *cam1, int type2, const float *cam2, float u1, float v1, float u2, float v2,
                                  const float *R12, const float *t12, float sigmaLevel, float unc, float x3D_out[3])
{
    float r1[3], r2[3], r21[3];
    orc_camera_unproject_f(type1, cam1, u1, v1, r1);
    orc_camera_unproject_f(type2, cam2, u2, v2, r2);
    for (int i = 0; i < 3; i++) r21[i] = (float)((double)R12[3 * i] * r2[0] + (double)R12[3 * i + 1] * r2[1] + (double)R12[3 * i + 2] * r2[2]);
    const double dot = (double)r1[0] * r21[0] + (double)r1[1] * r21[1] + (double)r1[2] * r21[2];
    const double n1 = sqrt((double)r1[0] * r1[0] + (double)r1[1] * r1[1] + (double)r1[2] * r1[2]);
    const double n2 = sqrt((double)r21[0] * r21[0] + (double)r21[1] * r21[1] + (double)r21[2] * r21[2]);
    const float cosParallaxRays = (float)(dot / (n1 * n2));
    if (cosParallaxRays > 0.9998) return -1;
    float R21[9], t21[3], T2[12], A[16], Vt[16];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R21[3 * i + j] = R12[3 * j + i];
    for (int i = 0; i < 3; i++) t21[i] = (float)(-1.0 * ((double)R21[3 * i] * t12[0] + (double)R21[3 * i + 1] * t12[1] + (double)R21[3 * i + 2] * t12[2]));
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T2[4 * i + j] = R21[3 * i + j]; T2[4 * i + 3] = t21[i]; }
    static const float T1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int j = 0; j < 4; j++) {                                                  /* KannalaBrandt8.cpp:426-429 */
        A[j] = r1[0] * T1[8 + j] - T1[j];
        A[4 + j] = r1[1] * T1[8 + j] - T1[4 + j];
        A[8 + j] = r2[0] * T2[8 + j] - T2[j];
        A[12 + j] = r2[1] * T2[8 + j] - T2[4 + j];
    }
    jacobi_svd4_vt(A, Vt);
    const float inv = (float)(1.0 / (double)Vt[15]);                               /* x3D.rowRange(0,3) / x3D.at<float>(3) */
    float x3D[3] = {Vt[12] * inv, Vt[13] * inv, Vt[14] * inv};
    const float z1 = x3D[2];
    if (!(z1 > 0)) return -1;                                                      /* (NaN from a zero homogeneous coordinate fails every test below in the reference as well) */
    const float z2 = (float)((double)R21[6] * x3D[0] + (double)R21[7] * x3D[1] + (double)R21[8] * x3D[2] + (double)t21[2]);
    if (z2 <= 0) return -1;
    float uv1[2], uv2[2], x3D2[3];
    orc_camera_project_f(type1, cam1, x3D, uv1);
    const float errX1 = uv1[0] - u1, errY1 = uv1[1] - v1;
    if ((errX1 * errX1 + errY1 * errY1) > 5.991 * sigmaLevel) return -1;
    for (int i = 0; i < 3; i++)
        x3D2[i] = (float)((double)R21[3 * i] * x3D[0] + (double)R21[3 * i + 1] * x3D[1] + (double)R21[3 * i + 2] * x3D[2] + (double)t21[i]);
    orc_camera_project_f(type2, cam2, x3D2, uv2);
    const float errX2 = uv2[0] - u2, errY2 = uv2[1] - v2;
    if ((errX2 * errX2 + errY2 * errY2) > 5.991 * unc) return -1;
    if (x3D_out) { x3D_out[0] = x3D[0]; x3D_out[1] = x3D[1]; x3D_out[2] = x3D[2]; }
    return z1;
}

int orc_search_for_triangulation_general(const int32_t *nid1, const uint8_t *has_mp1, const orc_keypoint *kp1, const uint8_t *desc1,
                                         const float *u_right1, int n1,
                                         const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2,
                                         const uint8_t *has_mp2, const orc_keypoint *kp2, const uint8_t *desc2, const float *u_right2,
                                         const orc_tri_general *g, const float *level_sigma2_1, const float *scale_factors2,
                                         const float *level_sigma2_2, int check_orientation, int32_t *matches12)
{
    int nmatches = 0;
    int hist[HISTO_LENGTH]; memset(hist, 0, sizeof(hist));
    int *bin_of = (int *)malloc(sizeof(int) * (n1 ? n1 : 1));
    const float factor = 1.0f / HISTO_LENGTH;
    const int cam2nd1 = g->nleft1 != -1, cam2nd2 = g->nleft2 != -1;            /* pKF->mpCamera2 != 0 */
    for (int idx1 = 0; idx1 < n1; idx1++) {
        matches12[idx1] = -1; bin_of[idx1] = -1;
        if (has_mp1[idx1]) continue;                                           /* ORBm:1036-1042 */
        const int bStereo1 = !cam2nd1 && u_right1 && u_right1[idx1] >= 0;      /* :1044 */
        if (g->only_stereo && !bStereo1) continue;
        const int bRight1 = !(g->nleft1 == -1 || idx1 < g->nleft1);            /* :1055-1056 */
        int lo = 0, hi = nnodes2;
        while (lo < hi) { const int mid = (lo + hi) / 2; if (node_ids2[mid] < nid1[idx1]) lo = mid + 1; else hi = mid; }
        if (lo >= nnodes2 || node_ids2[lo] != nid1[idx1]) continue;
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int j = node_start2[lo]; j < node_start2[lo + 1]; j++) {
            const int idx2 = feat2[j];
            if (has_mp2[idx2]) continue;
            const int bStereo2 = !cam2nd2 && u_right2 && u_right2[idx2] >= 0;  /* :1073 */
            if (g->only_stereo && !bStereo2) continue;
            const int dist = orc_descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
            if (dist > TH_LOW || dist > bestDist) continue;
            const int bRight2 = !(g->nleft2 == -1 || idx2 < g->nleft2);
            if (!bStereo1 && !bStereo2 && !cam2nd1) {                          /* :1091-1099 */
                const float distex = g->ep_x - kp2[idx2].x, distey = g->ep_y - kp2[idx2].y;
                if (distex * distex + distey * distey < 100 * scale_factors2[kp2[idx2].octave]) continue;
            }
            const int c = (cam2nd1 && cam2nd2) ? 2 * bRight1 + bRight2 : 0;    /* :1101-1130 */
            const int ci1 = (cam2nd1 && cam2nd2) ? bRight1 : 0, ci2 = (cam2nd1 && cam2nd2) ? bRight2 : 0;
            const float s1 = level_sigma2_1[kp1[idx1].octave], s2 = level_sigma2_2[kp2[idx2].octave];
            int ok;
            if (g->cam1_type[ci1] == 0) ok = epipolar_ok(g->F12[c], &kp1[idx1], &kp2[idx2], s2);
            else ok = orc_kb8_triangulate_matches(1, g->cam1[ci1], g->cam2_type[ci2], g->cam2[ci2], kp1[idx1].x, kp1[idx1].y, kp2[idx2].x, kp2[idx2].y,
                                                  g->R12[c], g->t12[c], s1, s2, NULL) > 0.0001f;
            if (ok || g->coarse) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
            matches12[idx1] = bestIdx2; nmatches++;
            if (check_orientation) {
                float rot = kp1[idx1].angle - kp2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                hist[bin]++; bin_of[idx1] = bin;
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < n1; i++)
            if (bin_of[i] >= 0 && bin_of[i] != ind1 && bin_of[i] != ind2 && bin_of[i] != ind3) { matches12[i] = -1; nmatches--; }
    }
    free(bin_of);
    return nmatches;
}

/* ================================================================= SearchForTriangulation, the overload that returns the points
 * ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, vMatchedPoints) (ORBm:1212-1402; no caller in the
 * reference).  Candidate walk as the first overload, with these differences (all read from the text of :1236-1402):
 *   - only the map-point tests gate a keypoint (:1264-1265, :1289-1290): bOnlyStereo is never read, vbMatched2 is never set, and
 *     there is no epipole test;
 *   - camera and pose of each side by bRight (:1307-1321): Tcw = GetPose() / GetRightPose(), camera = mpCamera / mpCamera2;
 *   - the test is pCamera1->matchAndtriangulate (:1324): Pinhole's is `{ return false; }` (include/CameraModels/Pinhole.h:91-94);
 *     KannalaBrandt8's (KannalaBrandt8.cpp:240-332) is restated below;
 *   - the accepted candidate's x3D is kept per KF1 keypoint (:1327, :1338).
 * cv::Mat arithmetic as for TriangulateMatches above (products accumulate in double and round once, Mat::dot / cv::norm in double,
 * s*row - row through addWeighted in float, cv::SVD by the restated Jacobi): parity unpinned against an OpenCV build. */
int orc_kb8_match_and_triangulate(const float *cam1, int type2, const float *cam2, float u1, float v1, float u2, float v2,
                                  const float T1[12], const float T2[12], float sigmaLevel1, float sigmaLevel2, float x3D_out[3])
{
    float r1[3], r2[3], ray1[3], ray2[3];
    orc_camera_unproject_f(1, cam1, u1, v1, r1);                                  /* this->unproject: the callee is a KannalaBrandt8 */
    orc_camera_unproject_f(type2, cam2, u2, v2, r2);
    for (int i = 0; i < 3; i++) {                                                 /* Rwc = Rcw.t(); ray = Rwc * r (:266-267) */
        ray1[i] = (float)((double)T1[i] * r1[0] + (double)T1[4 + i] * r1[1] + (double)T1[8 + i] * r1[2]);
        ray2[i] = (float)((double)T2[i] * r2[0] + (double)T2[4 + i] * r2[1] + (double)T2[8 + i] * r2[2]);
    }
    const double dot = (double)ray1[0] * ray2[0] + (double)ray1[1] * ray2[1] + (double)ray1[2] * ray2[2];
    const double n1 = sqrt((double)ray1[0] * ray1[0] + (double)ray1[1] * ray1[1] + (double)ray1[2] * ray1[2]);
    const double n2 = sqrt((double)ray2[0] * ray2[0] + (double)ray2[1] * ray2[1] + (double)ray2[2] * ray2[2]);
    const float cosParallaxRays = (float)(dot / (n1 * n2));
    if (cosParallaxRays > 0.9998) return 0;                                       /* :272-274 */
    float A[16], Vt[16];
    for (int j = 0; j < 4; j++) {                                                 /* Triangulate, KannalaBrandt8.cpp:426-429 */
        A[j] = r1[0] * T1[8 + j] - T1[j];
        A[4 + j] = r1[1] * T1[8 + j] - T1[4 + j];
        A[8 + j] = r2[0] * T2[8 + j] - T2[j];
        A[12 + j] = r2[1] * T2[8 + j] - T2[4 + j];
    }
    jacobi_svd4_vt(A, Vt);
    const float inv = (float)(1.0 / (double)Vt[15]);
    const float x3D[3] = {Vt[12] * inv, Vt[13] * inv, Vt[14] * inv};
    const float z1 = (float)((double)T1[8] * x3D[0] + (double)T1[9] * x3D[1] + (double)T1[10] * x3D[2] + (double)T1[11]);   /* :292 */
    if (!(z1 > 0)) return 0;
    const float z2 = (float)((double)T2[8] * x3D[0] + (double)T2[9] * x3D[1] + (double)T2[10] * x3D[2] + (double)T2[11]);   /* :298 */
    if (!(z2 > 0)) return 0;
    float uv1[2], uv2[2], xc[3];
    for (int i = 0; i < 3; i++)
        xc[i] = (float)((double)T1[4 * i] * x3D[0] + (double)T1[4 * i + 1] * x3D[1] + (double)T1[4 * i + 2] * x3D[2] + (double)T1[4 * i + 3]);
    orc_camera_project_f(1, cam1, xc, uv1);
    const float errX1 = uv1[0] - u1, errY1 = uv1[1] - v1;
    if ((errX1 * errX1 + errY1 * errY1) > 5.991 * sigmaLevel1) return 0;          /* :311 */
    for (int i = 0; i < 3; i++)
        xc[i] = (float)((double)T2[4 * i] * x3D[0] + (double)T2[4 * i + 1] * x3D[1] + (double)T2[4 * i + 2] * x3D[2] + (double)T2[4 * i + 3]);
    orc_camera_project_f(type2, cam2, xc, uv2);
    const float errX2 = uv2[0] - u2, errY2 = uv2[1] - v2;
    if ((errX2 * errX2 + errY2 * errY2) > 5.991 * sigmaLevel2) return 0;          /* :323 */
    x3D_out[0] = x3D[0]; x3D_out[1] = x3D[1]; x3D_out[2] = x3D[2];
    return 1;
}

int orc_search_for_triangulation_points(const int32_t *nid1, const uint8_t *has_mp1, const orc_keypoint *kp1, const uint8_t *desc1, int n1,
                                        const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2,
                                        const uint8_t *has_mp2, const orc_keypoint *kp2, const uint8_t *desc2,
                                        const orc_tri_general *g, const orc_tri_poses *poses, const float *level_sigma2_1,
                                        const float *level_sigma2_2, int check_orientation, int32_t *matches12, float *points12)
{
    int nmatches = 0;
    int hist[HISTO_LENGTH]; memset(hist, 0, sizeof(hist));
    int *bin_of = (int *)malloc(sizeof(int) * (n1 ? n1 : 1));
    const float factor = 1.0f / HISTO_LENGTH;
    for (int idx1 = 0; idx1 < n1; idx1++) {
        matches12[idx1] = -1; bin_of[idx1] = -1;
        points12[3 * idx1] = points12[3 * idx1 + 1] = points12[3 * idx1 + 2] = 0;
        if (has_mp1[idx1]) continue;                                           /* :1264-1265 */
        const int bRight1 = !(g->nleft1 == -1 || idx1 < g->nleft1);            /* :1271-1272 */
        int lo = 0, hi = nnodes2;
        while (lo < hi) { const int mid = (lo + hi) / 2; if (node_ids2[mid] < nid1[idx1]) lo = mid + 1; else hi = mid; }
        if (lo >= nnodes2 || node_ids2[lo] != nid1[idx1]) continue;
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int j = node_start2[lo]; j < node_start2[lo + 1]; j++) {
            const int idx2 = feat2[j];
            if (has_mp2[idx2]) continue;                                       /* :1289 (vbMatched2 stays false) */
            const int dist = orc_descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
            if (dist > TH_LOW || dist > bestDist) continue;                    /* :1296 */
            const int bRight2 = !(g->nleft2 == -1 || idx2 < g->nleft2);        /* :1304-1305 */
            float x3D[3];
            if (g->cam1_type[bRight1] == 1 &&
                orc_kb8_match_and_triangulate(g->cam1[bRight1], g->cam2_type[bRight2], g->cam2[bRight2], kp1[idx1].x, kp1[idx1].y, kp2[idx2].x,
                                              kp2[idx2].y, poses->Tcw1[bRight1], poses->Tcw2[bRight2], level_sigma2_1[kp1[idx1].octave],
                                              level_sigma2_2[kp2[idx2].octave], x3D)) {
                bestIdx2 = idx2; bestDist = dist;
                points12[3 * idx1] = x3D[0]; points12[3 * idx1 + 1] = x3D[1]; points12[3 * idx1 + 2] = x3D[2];
            }
        }
        if (bestIdx2 >= 0) {
            matches12[idx1] = bestIdx2; nmatches++;
            if (check_orientation) {                                           /* :1342-1352 */
                float rot = kp1[idx1].angle - kp2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                hist[bin]++; bin_of[idx1] = bin;
            }
        }
    }
    if (check_orientation) {                                                   /* :1370-1389 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < n1; i++)
            if (bin_of[i] >= 0 && bin_of[i] != ind1 && bin_of[i] != ind2 && bin_of[i] != ind3) { matches12[i] = -1; nmatches--; }
    }
    free(bin_of);
    return nmatches;
}
