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
/* One projected map point of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono)
 * (ORBmatcher.cc:1965-2181, Nleft == -1 path).  The projection itself (ORBm:1992-2008) is host geometry:
 * the caller supplies uv, radius = th * mvScaleFactors[nLastOctave] (ORBm:2014), the level range of the
 * GetFeaturesInArea call chosen by bForward/bBackward (ORBm:2018-2023; -1 = open), ur = uv.x - mbf*invzc
 * (ORBm:2043), the last-frame keypoint's angle (ORBm:2067-2073) and whether pMP->Observations() > 0. */
typedef struct { float u, v, radius, ur, angle; int32_t min_level, max_level, has_obs; } orc_proj_query;
/* train_match [n] in/out = CurrentFrame.mvpMapPoints: -1 free, <= -2 held by a map point with observations
 * (never a candidate, ORBm:2037-2039); on return >= 0 is the index of the query that claimed the keypoint.
 * u_right may be NULL (monocular: mvuRight == -1).  Returns nmatches (ORBm:2061, 2171). */
int orc_search_by_projection(const orc_proj_query *q, const uint8_t *desc_q, int nq,
                             const orc_keypoint *kp, const uint8_t *desc, const float *u_right, int n,
                             float min_x, float min_y, float max_x, float max_y,
                             int th_high, int check_orientation, int32_t *train_match);
/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, ...) (ORBmatcher.cc:48-218,
 * F.Nleft == -1), the TrackLocalMap matcher.  One query per map point with mbTrackInView that survives :57-65:
 * u,v = mTrackProjX/Y, radius = r * F.mvScaleFactors[nPredictedLevel] with r = RadiusByViewingCos(...) [* th] (:72-79),
 * (min_level, max_level) = (nPredictedLevel-1, nPredictedLevel), ur = mTrackProjXR, has_obs as above; angle unused.
 * Best and second-best distance with their octaves; accepted iff best <= th_high and not (same octave and
 * best > nn_ratio * second) (:131-137).  train_match as in orc_search_by_projection.  Returns nmatches. */
int orc_search_by_projection_map(const orc_proj_query *q, const uint8_t *desc_q, int nq,
                                 const orc_keypoint *kp, const uint8_t *desc, const float *u_right, int n,
                                 float min_x, float min_y, float max_x, float max_y,
                                 int th_high, float nn_ratio, int32_t *train_match);
/* The search part of ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th, bRight)
 * (ORBmatcher.cc:1403-1613, NLeft == -1): per projected map point (the geometry of :1430-1497 stays with the caller) the
 * keypoint of pKF inside the window with the smallest descriptor distance, among those at octave nPredictedLevel-1 or
 * nPredictedLevel (:1527-1528) whose reprojection error passes e2 * invLevelSigma2 <= 5.99 (monocular keypoint) or
 * <= 7.8 with the right coordinate (mvuRight >= 0) (:1530-1560); first minimum in GetFeaturesInArea order wins.
 * q[t]: u, v = uv; radius = th * mvScaleFactors[nPredictedLevel]; ur = uv.x - bf*invz; min_level / max_level =
 * nPredictedLevel-1 / nPredictedLevel.  best_idx[t] = bestIdx or -1, best_dist[t] = bestDist (256 if none); the caller
 * applies bestDist <= TH_LOW and the Replace / AddObservation logic (:1572-1595) in order. */
void orc_fuse_search(const orc_proj_query *q, const uint8_t *desc_q, int nq,
                     const orc_keypoint *kp, const uint8_t *desc, const float *u_right, int n,
                     const float *inv_level_sigma2, float min_x, float min_y, float max_x, float max_y,
                     int32_t *best_idx, int32_t *best_dist);
/* ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) (ORBmatcher.cc:273-475,
 * F.Nleft == -1): the matcher of Tracking::TrackReferenceKeyFrame / Relocalization.  Feature vectors (DBoW2 FeatureVector =
 * map<NodeId, vector<feature index>>) arrive flattened: node ids ascending, node_start[k]..node_start[k+1] into feat[].
 * kf_valid[i] = pKF's map point at feature i exists and is not bad (:297-302).  Within a shared node, every valid KF feature
 * takes the best unmatched F feature of that node if best <= TH_LOW and best < nn_ratio * second (:304-360); rotation
 * consistency (:445-470) when check_orientation.  match_f [nF] out: KF feature index whose map point F's feature got, or -1.
 * Returns nmatches. */
int orc_search_by_bow(const int32_t *kf_node_ids, const int32_t *kf_node_start, const int32_t *kf_feat, int kf_nnodes,
                      const uint8_t *kf_valid, const orc_keypoint *kf_kp, const uint8_t *kf_desc,
                      const int32_t *f_node_ids, const int32_t *f_node_start, const int32_t *f_feat, int f_nnodes,
                      const orc_keypoint *f_kp, const uint8_t *f_desc, int nF,
                      float nn_ratio, int check_orientation, int32_t *match_f);
/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse) (ORBmatcher.cc:969-1210),
 * Pinhole cameras without a second camera (mpCamera2 == 0): for every keypoint of KF1 without a map point, the keypoint of
 * KF2 in the same vocabulary node, without a map point, with the smallest descriptor distance <= TH_LOW (a later candidate
 * at EQUAL distance replaces an earlier one, :1046-1047) that is >= 10*sqrt(scale) px away from the epipole when both are
 * monocular (:1056-1064) and satisfies Pinhole::epipolarConstrain (CameraModels/Pinhole.cpp:122-144) with the
 * fundamental matrix F12 (row-major 3x3, computed by the caller as in :124-127) -- or any candidate when coarse.  This
 * fork never sets vbMatched2, so KF1 keypoints are independent.  Rotation consistency (:1171-1189) when check_orientation.
 * nid1 [n1]: vocabulary node of every KF1 feature; KF2's FeatureVector flattened as in orc_search_by_bow.
 * has_mp* : the keypoint already has a map point; u_right* may be NULL (monocular).  matches12 [n1] out.  Returns nmatches. */
int orc_search_for_triangulation(const int32_t *nid1, const uint8_t *has_mp1, const orc_keypoint *kp1, const uint8_t *desc1,
                                 const float *u_right1, int n1,
                                 const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2,
                                 const uint8_t *has_mp2, const orc_keypoint *kp2, const uint8_t *desc2, const float *u_right2,
                                 const float F12[9], float ep_x, float ep_y, const float *scale_factors, const float *level_sigma2,
                                 int only_stereo, int coarse, int check_orientation, int32_t *matches12);
/* ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12) (ORBmatcher.cc:827-967, NLeft == -1): as
 * orc_search_by_bow but both sides carry a "has a good map point" flag, a KF2 feature is claimed through vbMatched2, the
 * distance test is strict (best < TH_LOW, :909) and the result is indexed by the KF1 feature: matches12 [n1] = idx2 or -1. */
int orc_search_by_bow_kf(const int32_t *node_ids1, const int32_t *node_start1, const int32_t *feat1, int nnodes1,
                         const uint8_t *valid1, const orc_keypoint *kp1, const uint8_t *desc1, int n1,
                         const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2,
                         const uint8_t *valid2, const orc_keypoint *kp2, const uint8_t *desc2, int n2,
                         float nn_ratio, int check_orientation, int32_t *matches12);
/* Search loop of the Sim3 SearchByProjection overloads (ORBmatcher.cc:477-598, 600-708; LoopClosing).  matched [n] in/out:
 * -1 = vpMatched[idx] == NULL, anything else = taken; a claimed keypoint receives the query index.  Returns nmatches. */
int orc_search_by_projection_sim3(const orc_proj_query *q, const uint8_t *desc_q, int nq,
                                  const orc_keypoint *kp, const uint8_t *desc, int n,
                                  float min_x, float min_y, float max_x, float max_y, float ratio_hamming, int32_t *matched);
/* Per-point window search of SearchBySim3 (ORBmatcher.cc:1813-1851, 1893-1931) and Fuse(KeyFrame*, Scw, ...) (:1687-1720). */
void orc_window_best(const orc_proj_query *q, const uint8_t *desc_q, int nq, const orc_keypoint *kp, const uint8_t *desc, int n,
                     float min_x, float min_y, float max_x, float max_y, int32_t *best_idx, int32_t *best_dist);
/* Frame::AssignFeaturesToGrid (src/Frame.cc:377-408, Nleft == -1) as a CSR: cell_start [64*48+1], items [n] (cell ix*48+iy,
 * insertion = index order; keypoints outside the grid are in no cell). */
void orc_assign_features_to_grid(const orc_keypoint *kp, int n, float min_x, float min_y, float max_x, float max_y,
                                 int32_t *cell_start, int32_t *items);
/* Frame::UndistortKeyPoints (src/Frame.cc:738-771): out = kp with pt replaced by cv::undistortPoints(pt, K, dist, R = I, P = K);
 * dist = (k1, k2, p1, p2[, k3]); a zero k1 copies (Frame.cc:740-744). */
void orc_undistort_keypoints(const orc_keypoint *kp, int n, float fx, float fy, float cx, float cy, const float *dist, int ndist,
                             orc_keypoint *out);
/* BowVector / FeatureVector assembly of TemplatedVocabulary::transform (TemplatedVocabulary.h:1139-1208; TF_IDF + L1) from the
 * per-feature (word, weight, node) of orc_bow_transform.  Arrays sized n (node_start n+1). */
void orc_bow_vectors(const int32_t *wid, const double *w, const int32_t *nid, int n,
                     int32_t *node_ids, int32_t *node_start, int32_t *feat, int32_t *nnodes,
                     int32_t *bow_word, double *bow_value, int32_t *nwords);
/* MapPoint::ComputeDistinctiveDescriptors (/root/reference/src/MapPoint.cc:327-403): among the n
 * descriptors that observe a map point, the one with the least median Hamming distance to all of them
 * (median = sorted row [int(0.5*(n-1))], self distance 0 included; first minimum wins).  Returns BestIdx (0 if n<=0). */
int orc_distinctive_descriptor(const uint8_t *desc, int n);
/* DBoW2 TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(feature, word_id, weight, nid, levelsup)
 * (/root/reference/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1260; called per feature by Frame::ComputeBoW,
 * /root/reference/src/Frame.cc:729-736, with levelsup = 4): descend the vocabulary tree from the root, at every level
 * to the child with the smallest Hamming distance (first minimum wins), until a leaf; nid = the node passed at level
 * L - levelsup (0 = root when that level is <= 0).  The tree is given flat: node i has children
 * child_ids[child_start[i] .. child_start[i+1]) (none = leaf), 32-byte descriptors, word_id and weight per node.
 * The reference's ORBvoc file is absent from its tree (.MISSING_LARGE_BLOBS): tests use synthetic vocabularies. */
void orc_bow_transform(const uint8_t *feature32, const uint8_t *node_desc, const int32_t *child_start, const int32_t *child_ids,
                       const int32_t *node_word, const double *node_weight, int L, int levelsup,
                       int32_t *word_id, double *weight, int32_t *nid);
#ifdef __cplusplus
}
#endif
/* two-camera rig frames (Nleft != -1): ORBm:113-122, 136-214, 338-359, 393-425, 2013-2016, 2089-2153; Frame.cc:395-405, 686 */
int orc_search_by_projection_rig(int mode, const orc_proj_query *q, const uint8_t *desc_q, int nq,
                                 const orc_keypoint *kp, const uint8_t *desc, int n, int nleft, const int32_t *mirror,
                                 float min_x, float min_y, float max_x, float max_y,
                                 int th_high, float nn_ratio, int check_orientation, int32_t *train_match);
int orc_search_by_bow_rig(const int32_t *kf_node_ids, const int32_t *kf_node_start, const int32_t *kf_feat, int kf_nnodes,
                          const uint8_t *kf_valid, const orc_keypoint *kf_kp, const uint8_t *kf_desc,
                          const int32_t *f_node_ids, const int32_t *f_node_start, const int32_t *f_feat, int f_nnodes,
                          const orc_keypoint *f_kp, const uint8_t *f_desc, int nF, int nleft,
                          float nn_ratio, int check_orientation, int32_t *match_f);

/* ---- SearchForTriangulation for every camera combination (ORBm:969-1210; Pinhole.cpp:122-144, KannalaBrandt8.cpp:235-238, :334-401) */
typedef struct {
    float R12[4][9], t12[4][3];          /* relative pose X1 = R12 X2 + t12 of the camera pair [2*bRight1 + bRight2]: ll, lr, rl, rr (ORBm:994-1008);
                                            a single-camera pair uses [0] */
    float F12[4][9];                     /* K1^-T [t12]x R12 K2^-1 of the same combinations (what Pinhole::epipolarConstrain builds, Pinhole.cpp:124-127) */
    float cam1[2][8], cam2[2][8];        /* mvParameters (fx fy cx cy k1..k4) of pKF1->mpCamera / mpCamera2 and of pKF2's */
    int32_t cam1_type[2], cam2_type[2];  /* 0 Pinhole, 1 KannalaBrandt8 */
    float ep_x, ep_y;                    /* pKF2->mpCamera->project(R2w Cw + t2w), ORBm:978-984 */
    int32_t nleft1, nleft2;              /* NLeft; -1 = single camera (mpCamera2 == 0).  Rig keyframes: keypoints mvKeys | mvKeysRight */
    int32_t only_stereo, coarse;
} orc_tri_general;
int orc_search_for_triangulation_general(const int32_t *nid1, const uint8_t *has_mp1, const orc_keypoint *kp1, const uint8_t *desc1,
                                         const float *u_right1, int n1,
                                         const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2,
                                         const uint8_t *has_mp2, const orc_keypoint *kp2, const uint8_t *desc2, const float *u_right2,
                                         const orc_tri_general *g, const float *level_sigma2_1, const float *scale_factors2,
                                         const float *level_sigma2_2, int check_orientation, int32_t *matches12);
float orc_kb8_triangulate_matches(int type1, const float *cam1, int type2, const float *cam2, float u1, float v1, float u2, float v2,
                                  const float *R12, const float *t12, float sigmaLevel, float unc, float x3D_out[3]);
/* ---- SearchForTriangulation returning the triangulated points (ORBm:1212-1402; KannalaBrandt8.cpp:240-332; Pinhole.h:91-94) */
typedef struct { float Tcw1[2][12], Tcw2[2][12]; } orc_tri_poses;   /* rows 0..2 of GetPose() [0] / GetRightPose() [1] of pKF1, pKF2 */
int orc_kb8_match_and_triangulate(const float *cam1, int type2, const float *cam2, float u1, float v1, float u2, float v2,
                                  const float T1[12], const float T2[12], float sigmaLevel1, float sigmaLevel2, float x3D_out[3]);
int orc_search_for_triangulation_points(const int32_t *nid1, const uint8_t *has_mp1, const orc_keypoint *kp1, const uint8_t *desc1, int n1,
                                        const int32_t *node_ids2, const int32_t *node_start2, const int32_t *feat2, int nnodes2,
                                        const uint8_t *has_mp2, const orc_keypoint *kp2, const uint8_t *desc2,
                                        const orc_tri_general *g, const orc_tri_poses *poses, const float *level_sigma2_1,
                                        const float *level_sigma2_2, int check_orientation, int32_t *matches12, float *points12);
void orc_camera_project_f(int type, const float *p, const float P[3], float uv[2]);
void orc_camera_unproject_f(int type, const float *p, float u, float v, float ray[3]);

#endif
