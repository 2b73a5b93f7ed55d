// ORBmatcher_keyframe.cc -- the keyframe-side methods of the signature-preserving ORBmatcher (reference include/ORBmatcher.h:54-88):
//   SearchByProjection(Frame&, KeyFrame*, set<MapPoint*>&, th, ORBdist)            src/ORBmatcher.cc:2183-2305   Tracking::Relocalization
//   SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th, ratioHamming)      :477-591                      LoopClosing
//   SearchByProjection(KeyFrame*, Scw, vpPoints, vpPointsKFs, vpMatched, vpMatchedKF, th, ratioHamming)  :593-708 LoopClosing (place recognition)
//   SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)                                 :827-967                      LoopClosing
//   SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, bCoarse)   :969-1210                     LocalMapping::CreateNewMapPoints
//   SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo, vMatchedPoints)   :1212-1402              (no caller)
//   SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)                       :1739-1963                    LoopClosing
//   Fuse(pKF, vpMapPoints, th, bRight)                                             :1403-1613                    LocalMapping::SearchInNeighbors
//   Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)                                   :1615-1737                    LoopClosing
// As in ORBmatcher.cc: the per-point host geometry in front of each search (projection, image / distance / viewing-angle tests,
// predicted level) is kept in the reference's order and arithmetic (cvmath.h spells the cv::Mat expressions out); everything from
// GetFeaturesInArea on -- grid, window, level and reprojection gates, descriptor distances, claim rules, ratio tests, rotation
// histogram, the epipolar tests of both camera models -- runs on the device behind the C ABI; the MapPoint / KeyFrame bookkeeping
// after a search (Replace, AddObservation, AddMapPoint) is applied here in the reference's order.
#include "ORBmatcher.h"
#include <climits>
#include <cstdio>
#include <set>
#include "cvmath.h"
#include "hip_context.h"

namespace ORB_SLAM3 {

namespace {
inline orbhip_ctx *thread_ctx() { return hip::ThreadContext(); }
static_assert(sizeof(cv::KeyPoint) == sizeof(orbhip_keypoint), "KeyPoint layout");

// the 64 x 48 grid of a keyframe covers the undistorted image bounds of its frame (KeyFrame.cc:66-75: copied from Frame)
struct Bounds { float min_x, min_y, max_x, max_y; };
inline Bounds bounds_of(KeyFrame *pKF) { return {(float)pKF->mnMinX, (float)pKF->mnMinY, (float)pKF->mnMaxX, (float)pKF->mnMaxY}; }

inline void put_desc(std::vector<uint8_t> &dq, MapPoint *pMP)
{
    const cv::Mat d = pMP->GetDescriptor();
    dq.insert(dq.end(), d.ptr<uint8_t>(), d.ptr<uint8_t>() + 32);
}

// Decompose Scw (:486-491, :602-607, :1624-1629): sRcw = Scw(0:3,0:3); scw = sqrt(row0 . row0); Rcw = sRcw / scw; tcw = Scw(0:3,3) / scw;
// Ow = -Rcw.t() * tcw
struct Sim3Parts { cvm::M3 Rcw; cvm::V3 tcw, Ow; };
inline Sim3Parts decompose(const cv::Mat &Scw)
{
    Sim3Parts P;
    const cvm::M3 sRcw = cvm::block3(Scw);
    const cvm::V3 r0 = {{sRcw(0, 0), sRcw(0, 1), sRcw(0, 2)}};
    const float scw = std::sqrt(cvm::dot(r0, r0));
    P.Rcw = cvm::scale(sRcw, 1.0 / scw);
    P.tcw = cvm::scale(cvm::col3(Scw), 1.0 / scw);
    P.Ow = cvm::mul_t(P.Rcw, P.tcw, -1.0);
    return P;
}

// the common tail of the projection-type searches: one query per surviving point
struct Queries {
    std::vector<orbhip_proj_query> q; std::vector<uint8_t> dq; std::vector<int> owner;
    void add(float u, float v, float radius, float ur, float angle, int lo, int hi, int has_obs, MapPoint *pMP, int who)
    {
        orbhip_proj_query e;
        e.u = u; e.v = v; e.radius = radius; e.ur = ur; e.angle = angle; e.min_level = lo; e.max_level = hi; e.has_obs = has_obs;
        q.push_back(e); owner.push_back(who); put_desc(dq, pMP);
    }
};

// "depth inside the scale invariance region" + "viewing angle below 60 degrees" + predicted level (:531-549, :645-663, :1667-1685, :1482-1505)
template <class KF>
inline bool scale_and_angle(MapPoint *pMP, const cvm::V3 &p3Dw, const cvm::V3 &Ow, KF *pKF, bool check_normal, int &nPredictedLevel)
{
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    const cvm::V3 PO = cvm::sub(p3Dw, Ow);
    const float dist = cvm::norm(PO);
    if (dist < minDistance || dist > maxDistance) return false;
    if (check_normal) {
        const cvm::V3 Pn = cvm::vec3(pMP->GetNormal());
        if (cvm::dot(PO, Pn) < 0.5 * dist) return false;
    }
    nPredictedLevel = pMP->PredictScale(dist, pKF);
    return true;
}

inline const float *uright_or_null(const std::vector<float> &v, int n) { return (int)v.size() >= n && n > 0 ? v.data() : nullptr; }
}  // namespace

// ---------------------------------------------------------------------------------------------------------------- Relocalization
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist)
{
    const cvm::M3 Rcw = cvm::block3(CurrentFrame.mTcw);
    const cvm::V3 tcw = cvm::col3(CurrentFrame.mTcw);
    const cvm::V3 Ow = cvm::mul_t(Rcw, tcw, -1.0);                                      // -Rcw.t()*tcw (:2189)
    const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
    Queries Q;
    for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {                            // :2199-2233
        MapPoint *pMP = vpMPs[i];
        if (!pMP) continue;
        if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        // Project
        const cvm::V3 x3Dw = cvm::vec3(pMP->GetWorldPos());
        const cvm::V3 x3Dc = cvm::mul_add(Rcw, x3Dw, tcw);
        const cv::Point2f uv = CurrentFrame.mpCamera->project(cvm::to_mat(x3Dc));
        if (uv.x < CurrentFrame.mnMinX || uv.x > CurrentFrame.mnMaxX) continue;
        if (uv.y < CurrentFrame.mnMinY || uv.y > CurrentFrame.mnMaxY) continue;
        // Compute predicted scale level; depth must be inside the scale pyramid of the image (no viewing-angle test here)
        int nPredictedLevel;
        if (!scale_and_angle(pMP, x3Dw, Ow, &CurrentFrame, false, nPredictedLevel)) continue;
        // Search in a window
        const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
        Q.add(uv.x, uv.y, radius, -1.f, pKF->mvKeysUn[i].angle, nPredictedLevel - 1, nPredictedLevel + 1, 1, pMP, (int)i);
    }
    const int n = CurrentFrame.N;
    // any map point already held blocks a keypoint (:2247-2248: if(CurrentFrame.mvpMapPoints[i2]) continue), and every match of this
    // call blocks it for the later points: the claim rule of the last-frame search with has_obs on every query
    std::vector<int32_t> tm(n > 0 ? n : 1, -1);
    for (int i = 0; i < n; i++) tm[i] = CurrentFrame.mvpMapPoints[i] ? -2 : -1;
    int32_t nmatches = 0;
    const int rc = orbhip_search_by_projection_host(thread_ctx(), 0, Q.q.data(), Q.dq.data(), (int)Q.q.size(), (const orbhip_keypoint *)CurrentFrame.mvKeysUn.data(),
                                                    CurrentFrame.mDescriptors.ptr<uint8_t>(), nullptr, n, Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY,
                                                    ORBdist, 0.f, mbCheckOrientation ? 1 : 0, tm.data(), &nmatches);
    if (rc != ORBHIP_OK) { fprintf(stderr, "ORBmatcher (HIP): SearchByProjection(Frame, KeyFrame): %d (%s)\n", rc, orbhip_last_error()); return 0; }
    for (int i = 0; i < n; i++) if (tm[i] >= 0) CurrentFrame.mvpMapPoints[i] = vpMPs[Q.owner[tm[i]]];      // :2258 (rotation rejects come back free: :2291 sets NULL)
    return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------- Sim3 projections
namespace {
// the body both Sim3 SearchByProjection overloads share (:493-588 / :609-705); project_camera: the first overload projects through
// pKF->mpCamera (:525), the second with fx, fy, cx, cy directly (:635-640)
int sim3_projection(ORBmatcher *self, KeyFrame *pKF, const cv::Mat &Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched,
                    std::vector<KeyFrame *> *vpMatchedKF, const std::vector<KeyFrame *> *vpPointsKFs, int th, float ratioHamming, bool project_camera, int TH_LOW_)
{
    (void)self;
    const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
    const Sim3Parts S = decompose(Scw);
    std::set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint *>(NULL));
    Queries Q;
    for (int iMP = 0, iendMP = vpPoints.size(); iMP < iendMP; iMP++) {
        MapPoint *pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        const cvm::V3 p3Dw = cvm::vec3(pMP->GetWorldPos());
        const cvm::V3 p3Dc = cvm::mul_add(S.Rcw, p3Dw, S.tcw);
        if (p3Dc(2) < 0.0) continue;
        float u, v;
        if (project_camera) {
            const cv::Point2f uv = pKF->mpCamera->project(cv::Point3f(p3Dc(0), p3Dc(1), p3Dc(2)));
            u = uv.x; v = uv.y;
        } else {
            const float invz = 1 / p3Dc(2);
            const float x = p3Dc(0) * invz, y = p3Dc(1) * invz;
            u = fx * x + cx; v = fy * y + cy;
        }
        if (!pKF->IsInImage(u, v)) continue;
        int nPredictedLevel;
        if (!scale_and_angle(pMP, p3Dw, S.Ow, pKF, true, nPredictedLevel)) continue;
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        Q.add(u, v, radius, -1.f, 0.f, nPredictedLevel - 1, nPredictedLevel, 1, pMP, iMP);
    }
    const int n = (int)pKF->mvKeysUn.size();
    // vpMatched as the claim array: occupied keypoints are skipped (:563-564), a match occupies its keypoint for the later points
    std::vector<int32_t> tm(n > 0 ? n : 1, -1);
    for (int i = 0; i < n && i < (int)vpMatched.size(); i++) tm[i] = vpMatched[i] ? -2 : -1;
    int32_t nmatches = 0;
    const Bounds b = bounds_of(pKF);
    const int th_high = (int)std::floor(TH_LOW_ * ratioHamming);                        // bestDist <= TH_LOW*ratioHamming on integers (:581)
    const int rc = orbhip_search_by_projection_host(thread_ctx(), 0, Q.q.data(), Q.dq.data(), (int)Q.q.size(), (const orbhip_keypoint *)pKF->mvKeysUn.data(),
                                                    pKF->mDescriptors.ptr<uint8_t>(), nullptr, n, b.min_x, b.min_y, b.max_x, b.max_y, th_high, 0.f, 0, tm.data(),
                                                    &nmatches);
    if (rc != ORBHIP_OK) { fprintf(stderr, "ORBmatcher (HIP): SearchByProjection(KeyFrame, Scw): %d (%s)\n", rc, orbhip_last_error()); return 0; }
    for (int i = 0; i < n; i++)
        if (tm[i] >= 0) {
            vpMatched[i] = vpPoints[Q.owner[tm[i]]];
            if (vpMatchedKF) (*vpMatchedKF)[i] = (*vpPointsKFs)[Q.owner[tm[i]]];
        }
    return nmatches;
}
}  // namespace

int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th, float ratioHamming)
{
    return sim3_projection(this, pKF, Scw, vpPoints, vpMatched, nullptr, nullptr, th, ratioHamming, true, TH_LOW);
}

int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, const std::vector<KeyFrame *> &vpPointsKFs,
                                   std::vector<MapPoint *> &vpMatched, std::vector<KeyFrame *> &vpMatchedKF, int th, float ratioHamming)
{
    return sim3_projection(this, pKF, Scw, vpPoints, vpMatched, &vpMatchedKF, &vpPointsKFs, th, ratioHamming, false, TH_LOW);
}

// ---------------------------------------------------------------------------------------------------------------- SearchByBoW(KF, KF)
namespace {
void flatten(const DBoW2::FeatureVector &fv, std::vector<int32_t> &ids, std::vector<int32_t> &start, std::vector<int32_t> &feat)
{
    start.push_back(0);
    for (const auto &kv : fv) {
        ids.push_back((int32_t)kv.first);
        for (unsigned int i : kv.second) feat.push_back((int32_t)i);
        start.push_back((int32_t)feat.size());
    }
}
}  // namespace

int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12)
{
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint *>(vpMapPoints1.size(), static_cast<MapPoint *>(NULL));
    std::vector<int32_t> i1, s1, f1, i2, s2, f2;
    flatten(pKF1->mFeatVec, i1, s1, f1);
    flatten(pKF2->mFeatVec, i2, s2, f2);
    const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
    // a feature takes part when its map point exists and is not bad (:867-871, :887-894); on rig keyframes only the LEFT camera's
    // features do (:862-864, :882-884: idx >= mvKeysUn.size() is skipped)
    auto validity = [](KeyFrame *pKF, const std::vector<MapPoint *> &mps) {
        std::vector<uint8_t> v(mps.size() ? mps.size() : 1, 0);
        for (size_t k = 0; k < mps.size(); k++) {
            if (pKF->NLeft != -1 && k >= pKF->mvKeysUn.size()) continue;
            v[k] = (mps[k] && !mps[k]->isBad()) ? 1 : 0;
        }
        return v;
    };
    const std::vector<uint8_t> v1 = validity(pKF1, vpMapPoints1), v2 = validity(pKF2, vpMapPoints2);
    // keypoint arrays as long as the feature arrays (angles are read for valid features only: all below mvKeysUn.size())
    std::vector<cv::KeyPoint> k1 = pKF1->mvKeysUn, k2 = pKF2->mvKeysUn;
    k1.resize(n1 > 0 ? n1 : 1); k2.resize(n2 > 0 ? n2 : 1);
    std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
    int32_t nmatches = 0;
    const int rc = orbhip_search_by_bow_kf_host(thread_ctx(), i1.data(), s1.data(), f1.data(), (int)i1.size(), v1.data(), (const orbhip_keypoint *)k1.data(),
                                                pKF1->mDescriptors.ptr<uint8_t>(), n1, i2.data(), s2.data(), f2.data(), (int)i2.size(), v2.data(),
                                                (const orbhip_keypoint *)k2.data(), pKF2->mDescriptors.ptr<uint8_t>(), n2, mfNNratio, mbCheckOrientation ? 1 : 0,
                                                m12.data(), &nmatches);
    if (rc != ORBHIP_OK) { fprintf(stderr, "ORBmatcher (HIP): SearchByBoW(KeyFrame, KeyFrame): %d (%s)\n", rc, orbhip_last_error()); return 0; }
    for (int i = 0; i < n1; i++) if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];        // :911
    return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------- SearchForTriangulation
namespace {
void camera_params(GeometricCamera *cam, float (&p)[8], int32_t &type)
{
    type = cam->GetType() == cam->CAM_FISHEYE ? 1 : 0;
    for (int i = 0; i < 8; i++) p[i] = i < (int)cam->size() ? cam->getParameter(i) : 0.f;
}
// F12 = K1.t().inv() * t12x * R12 * K2.inv() (Pinhole.cpp:124-127), for the camera pairs whose first camera is a Pinhole and for which the
// caller's F12 does not apply (rig combinations)
void fundamental(const float (&c1)[8], const float (&c2)[8], const cvm::M3 &R12, const cvm::V3 &t12, float (&F)[9])
{
    cvm::M3 K1t, K2;
    for (int i = 0; i < 9; i++) K1t.m[i] = K2.m[i] = 0.f;
    K1t(0, 0) = c1[0]; K1t(2, 0) = c1[2]; K1t(1, 1) = c1[1]; K1t(2, 1) = c1[3]; K1t(2, 2) = 1.f;       // K1.t()
    K2(0, 0) = c2[0]; K2(0, 2) = c2[2]; K2(1, 1) = c2[1]; K2(1, 2) = c2[3]; K2(2, 2) = 1.f;
    const cvm::M3 A = cvm::mul(cvm::mul(cvm::mul(cvm::inv3(K1t), cvm::skew(t12)), R12), cvm::inv3(K2));
    for (int i = 0; i < 9; i++) F[i] = A.m[i];
}
}  // namespace

int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>> &vMatchedPairs,
                                       const bool bOnlyStereo, const bool bCoarse)
{
    orbhip_tri_pair_general g;
    memset(&g, 0, sizeof(g));
    // Compute epipole in second image (:978-984)
    const cvm::V3 Cw = cvm::vec3(pKF1->GetCameraCenter());
    const cvm::M3 R2w = cvm::block3(pKF2->GetRotation());
    const cvm::V3 t2w = cvm::vec3(pKF2->GetTranslation());
    const cvm::V3 C2 = cvm::mul_add(R2w, Cw, t2w);
    const cv::Point2f ep = pKF2->mpCamera->project(cvm::to_mat(C2));
    g.ep_x = ep.x; g.ep_y = ep.y;
    const cvm::M3 R1w = cvm::block3(pKF1->GetRotation());
    const cvm::V3 t1w = cvm::vec3(pKF1->GetTranslation());
    auto rel = [](const cvm::M3 &Ra, const cvm::V3 &ta, const cvm::M3 &Rb, const cvm::V3 &tb, float (&R)[9], float (&t)[3]) {
        // R12 = Ra * Rb.t();  t12 = Ra * (-Rb.t() * tb) + ta  -- and for the single-camera pair -Ra*Rb.t()*tb + ta: the same products
        const cvm::M3 Rab = cvm::mul_t(Ra, false, Rb, true);
        const cvm::V3 tmp = cvm::mul_t(Rb, tb, -1.0);
        const cvm::V3 tab = cvm::mul_add(Ra, tmp, ta);
        for (int i = 0; i < 9; i++) R[i] = Rab.m[i];
        for (int i = 0; i < 3; i++) t[i] = tab(i);
    };
    camera_params(pKF1->mpCamera, g.cam1[0], g.cam1_type[0]);
    camera_params(pKF2->mpCamera, g.cam2[0], g.cam2_type[0]);
    const bool rig = pKF1->mpCamera2 && pKF2->mpCamera2;
    if (!pKF1->mpCamera2 && !pKF2->mpCamera2) {                                       // :996-998
        // R12 = R1w*R2w.t(); t12 = -R1w*R2w.t()*t2w + t1w: (-(R1w R2w^T)) is evaluated first, then times t2w plus t1w
        const cvm::M3 R12 = cvm::mul_t(R1w, false, R2w, true);
        const cvm::M3 nR12 = cvm::mul_t(R1w, false, R2w, true, -1.0);
        const cvm::V3 t12 = cvm::mul_add(nR12, t2w, t1w);
        for (int i = 0; i < 9; i++) g.R12[0][i] = R12.m[i];
        for (int i = 0; i < 3; i++) g.t12[0][i] = t12(i);
        for (int i = 0; i < 9; i++) g.F12[0][i] = F12.at<float>(i / 3, i % 3);      // the caller's F12 is the very expression Pinhole::epipolarConstrain evaluates (LocalMapping.cc:1010-1024)
    } else if (rig) {                                                                 // :999-1008
        camera_params(pKF1->mpCamera2, g.cam1[1], g.cam1_type[1]);
        camera_params(pKF2->mpCamera2, g.cam2[1], g.cam2_type[1]);
        const cvm::M3 R1r = cvm::block3(pKF1->GetRightRotation()), R2r = cvm::block3(pKF2->GetRightRotation());
        const cvm::V3 t1r = cvm::vec3(pKF1->GetRightTranslation()), t2r = cvm::vec3(pKF2->GetRightTranslation());
        rel(R1w, t1w, R2w, t2w, g.R12[0], g.t12[0]);                                  // ll
        rel(R1w, t1w, R2r, t2r, g.R12[1], g.t12[1]);                                  // lr
        rel(R1r, t1r, R2w, t2w, g.R12[2], g.t12[2]);                                  // rl
        rel(R1r, t1r, R2r, t2r, g.R12[3], g.t12[3]);                                  // rr
        for (int c = 0; c < 4; c++)
            if (g.cam1_type[c >> 1] == 0) {
                cvm::M3 R; cvm::V3 t;
                for (int i = 0; i < 9; i++) R.m[i] = g.R12[c][i];
                for (int i = 0; i < 3; i++) t(i) = g.t12[c][i];
                fundamental(g.cam1[c >> 1], g.cam2[c & 1], R, t, g.F12[c]);
            }
    } else {
        // exactly one keyframe with a second camera: the reference reads an empty R12 here (:1131); nothing can be matched
        vMatchedPairs.clear();
        return 0;
    }
    g.nleft1 = pKF1->mpCamera2 ? pKF1->NLeft : -1; g.nleft2 = pKF2->mpCamera2 ? pKF2->NLeft : -1;
    g.only_stereo = bOnlyStereo ? 1 : 0; g.coarse = bCoarse ? 1 : 0;

    const int n1 = pKF1->N, n2 = pKF2->N;
    // keypoints in descriptor order: mvKeysUn, or mvKeys | mvKeysRight on rig keyframes (:1050-1052, :1084-1086)
    auto keys = [](KeyFrame *pKF) {
        if (pKF->NLeft == -1) return pKF->mvKeysUn;
        std::vector<cv::KeyPoint> k(pKF->mvKeys.begin(), pKF->mvKeys.begin() + pKF->NLeft);
        k.insert(k.end(), pKF->mvKeysRight.begin(), pKF->mvKeysRight.end());
        return k;
    };
    const std::vector<cv::KeyPoint> k1 = keys(pKF1), k2 = keys(pKF2);
    std::vector<int32_t> nid1(n1 > 0 ? n1 : 1, -1);
    for (const auto &kv : pKF1->mFeatVec) for (unsigned int i : kv.second) if ((int)i < n1) nid1[i] = (int32_t)kv.first;
    std::vector<int32_t> i2, s2, f2;
    flatten(pKF2->mFeatVec, i2, s2, f2);
    std::vector<uint8_t> mp1(n1 > 0 ? n1 : 1, 0), mp2(n2 > 0 ? n2 : 1, 0);
    for (int i = 0; i < n1; i++) mp1[i] = pKF1->GetMapPoint(i) ? 1 : 0;               // :1036-1042
    for (int i = 0; i < n2; i++) mp2[i] = pKF2->GetMapPoint(i) ? 1 : 0;               // :1068-1072
    std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
    int32_t nmatches = 0;
    const int nlevels = (int)pKF2->mvScaleFactors.size();
    const int rc = orbhip_search_for_triangulation_host(thread_ctx(), nid1.data(), mp1.data(), (const orbhip_keypoint *)k1.data(), pKF1->mDescriptors.ptr<uint8_t>(),
                                                        uright_or_null(pKF1->mvuRight, n1), n1, i2.data(), s2.data(), f2.data(), (int)i2.size(), mp2.data(),
                                                        (const orbhip_keypoint *)k2.data(), pKF2->mDescriptors.ptr<uint8_t>(), uright_or_null(pKF2->mvuRight, n2), n2,
                                                        &g, pKF1->mvLevelSigma2.data(), pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(), nlevels,
                                                        mbCheckOrientation ? 1 : 0, m12.data(), &nmatches);
    vMatchedPairs.clear();
    if (rc != ORBHIP_OK) { fprintf(stderr, "ORBmatcher (HIP): SearchForTriangulation: %d (%s)\n", rc, orbhip_last_error()); return 0; }
    vMatchedPairs.reserve(nmatches);
    for (int i = 0; i < n1; i++) {                                                    // :1196-1202
        if (m12[i] < 0) continue;
        vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
    }
    return nmatches;
}

// The overload that returns the triangulated points (:1212-1402).  F12 and bOnlyStereo are not read by the reference's body either
// (the epipole it computes at :1220-1224 is never used); the candidate test is pCamera1->matchAndtriangulate with the absolute
// poses GetPose() / GetRightPose() of the cameras the two keypoints belong to.
int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>> &vMatchedPairs,
                                       const bool bOnlyStereo, std::vector<cv::Mat> &vMatchedPoints)
{
    (void)F12; (void)bOnlyStereo;
    orbhip_tri_pair_general g;
    orbhip_tri_pair_poses P;
    memset(&g, 0, sizeof(g)); memset(&P, 0, sizeof(P));
    camera_params(pKF1->mpCamera, g.cam1[0], g.cam1_type[0]);
    camera_params(pKF2->mpCamera, g.cam2[0], g.cam2_type[0]);
    auto rows = [](const cv::Mat &T, float (&o)[12]) { for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) o[4 * i + j] = T.at<float>(i, j); };
    rows(pKF1->GetPose(), P.Tcw1[0]); rows(pKF2->GetPose(), P.Tcw2[0]);                 // :1311, :1318
    if (pKF1->NLeft != -1) { camera_params(pKF1->mpCamera2, g.cam1[1], g.cam1_type[1]); rows(pKF1->GetRightPose(), P.Tcw1[1]); }   // :1308-1309
    if (pKF2->NLeft != -1) { camera_params(pKF2->mpCamera2, g.cam2[1], g.cam2_type[1]); rows(pKF2->GetRightPose(), P.Tcw2[1]); }   // :1315-1316
    g.nleft1 = pKF1->NLeft; g.nleft2 = pKF2->NLeft;

    const int n1 = pKF1->N, n2 = pKF2->N;
    auto keys = [](KeyFrame *pKF) {                                                   // :1267-1269, :1300-1302
        if (pKF->NLeft == -1) return pKF->mvKeysUn;
        std::vector<cv::KeyPoint> k(pKF->mvKeys.begin(), pKF->mvKeys.begin() + pKF->NLeft);
        k.insert(k.end(), pKF->mvKeysRight.begin(), pKF->mvKeysRight.end());
        return k;
    };
    const std::vector<cv::KeyPoint> k1 = keys(pKF1), k2 = keys(pKF2);
    std::vector<int32_t> nid1(n1 > 0 ? n1 : 1, -1);
    for (const auto &kv : pKF1->mFeatVec) for (unsigned int i : kv.second) if ((int)i < n1) nid1[i] = (int32_t)kv.first;
    std::vector<int32_t> i2, s2, f2;
    flatten(pKF2->mFeatVec, i2, s2, f2);
    std::vector<uint8_t> mp1(n1 > 0 ? n1 : 1, 0), mp2(n2 > 0 ? n2 : 1, 0);
    for (int i = 0; i < n1; i++) mp1[i] = pKF1->GetMapPoint(i) ? 1 : 0;               // :1261-1265
    for (int i = 0; i < n2; i++) mp2[i] = pKF2->GetMapPoint(i) ? 1 : 0;               // :1286-1290
    std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
    std::vector<float> pts(3 * (size_t)(n1 > 0 ? n1 : 1), 0.f);
    int32_t nmatches = 0;
    const int nlevels = (int)pKF2->mvLevelSigma2.size();
    const int rc = orbhip_match_and_triangulate_host(thread_ctx(), nid1.data(), mp1.data(), (const orbhip_keypoint *)k1.data(), pKF1->mDescriptors.ptr<uint8_t>(), n1,
                                                     i2.data(), s2.data(), f2.data(), (int)i2.size(), mp2.data(), (const orbhip_keypoint *)k2.data(),
                                                     pKF2->mDescriptors.ptr<uint8_t>(), n2, &g, &P, pKF1->mvLevelSigma2.data(), pKF2->mvLevelSigma2.data(), nlevels,
                                                     mbCheckOrientation ? 1 : 0, m12.data(), pts.data(), &nmatches);
    vMatchedPairs.clear();
    if (rc != ORBHIP_OK) { fprintf(stderr, "ORBmatcher (HIP): SearchForTriangulation (points): %d (%s)\n", rc, orbhip_last_error()); return 0; }
    vMatchedPairs.reserve(nmatches);
    for (int i = 0; i < n1; i++) {                                                    // :1391-1399 (vMatchedPoints is appended to, not cleared)
        if (m12[i] < 0) continue;
        vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
        cv::Mat x3D(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) x3D.at<float>(k) = pts[3 * (size_t)i + k];
        vMatchedPoints.push_back(x3D);
    }
    return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------- Fuse
int ORBmatcher::Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th, const bool bRight)
{
    cvm::M3 Rcw; cvm::V3 tcw, Ow;
    GeometricCamera *pCamera;
    if (bRight) {                                                                     // :1408-1421
        Rcw = cvm::block3(pKF->GetRightRotation()); tcw = cvm::vec3(pKF->GetRightTranslation()); Ow = cvm::vec3(pKF->GetRightCameraCenter());
        pCamera = pKF->mpCamera2;
    } else {
        Rcw = cvm::block3(pKF->GetRotation()); tcw = cvm::vec3(pKF->GetTranslation()); Ow = cvm::vec3(pKF->GetCameraCenter());
        pCamera = pKF->mpCamera;
    }
    const float &bf = pKF->mbf;
    int nFused = 0;
    const int nMPs = vpMapPoints.size();
    // NB the reference's loop is sequential and its bookkeeping (AddObservation / AddMapPoint / Replace) can change IsInKeyFrame / isBad of a
    // LATER point of the same call only for a point that appears twice in vpMapPoints or is replaced by an earlier one; the searches are
    // therefore issued for every point that passes the tests NOW and the bookkeeping re-checks isBad / IsInKeyFrame in order below.
    Queries Q;
    for (int i = 0; i < nMPs; i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad()) continue;
        else if (pMP->IsInKeyFrame(pKF)) continue;
        const cvm::V3 p3Dw = cvm::vec3(pMP->GetWorldPos());
        const cvm::V3 p3Dc = cvm::mul_add(Rcw, p3Dw, tcw);
        if (p3Dc(2) < 0.0f) continue;                                                  // Depth must be positive
        const float invz = 1 / p3Dc(2);
        const cv::Point2f uv = pCamera->project(cv::Point3f(p3Dc(0), p3Dc(1), p3Dc(2)));
        if (!pKF->IsInImage(uv.x, uv.y)) continue;                                     // Point must be inside the image
        const float ur = uv.x - bf * invz;
        int nPredictedLevel;
        if (!scale_and_angle(pMP, p3Dw, Ow, pKF, true, nPredictedLevel)) continue;
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        Q.add(uv.x, uv.y, radius, ur, 0.f, nPredictedLevel - 1, nPredictedLevel, 0, pMP, i);
    }
    // the keypoints GetFeaturesInArea(.., bRight) walks (KeyFrame.cc:770-814) and what the loop reads beside them (:1527-1568): mvKeysUn,
    // or on rig keyframes mvKeys / mvKeysRight with their own grid; mvuRight is indexed with the camera-local index in both cases
    // (:1537: before idx += NLeft), descriptors with the frame-wide one
    const int nleft = pKF->NLeft;
    const std::vector<cv::KeyPoint> &kps = nleft == -1 ? pKF->mvKeysUn : (!bRight ? pKF->mvKeys : pKF->mvKeysRight);
    const int n = nleft == -1 ? (int)pKF->mvKeysUn.size() : (!bRight ? nleft : (int)pKF->mvKeysRight.size());
    const int row0 = (nleft != -1 && bRight) ? nleft : 0;
    std::vector<int32_t> bestIdx(Q.q.size() ? Q.q.size() : 1, -1), bestDist(Q.q.size() ? Q.q.size() : 1, 256);
    const Bounds b = bounds_of(pKF);
    const int rc = orbhip_fuse_search_host(thread_ctx(), Q.q.data(), Q.dq.data(), (int)Q.q.size(), (const orbhip_keypoint *)kps.data(),
                                           pKF->mDescriptors.ptr<uint8_t>() + (size_t)32 * row0, uright_or_null(pKF->mvuRight, n), n, pKF->mvInvLevelSigma2.data(),
                                           (int)pKF->mvInvLevelSigma2.size(), b.min_x, b.min_y, b.max_x, b.max_y, bestIdx.data(), bestDist.data());
    if (rc != ORBHIP_OK) { fprintf(stderr, "ORBmatcher (HIP): Fuse: %d (%s)\n", rc, orbhip_last_error()); return 0; }
    for (size_t t = 0; t < Q.q.size(); t++) {                                         // :1572-1597, in the order of vpMapPoints
        MapPoint *pMP = vpMapPoints[Q.owner[t]];
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;                          // (changed by an earlier entry of this call)
        if (bestDist[t] > TH_LOW) continue;
        const int idx = bestIdx[t] + row0;                                            // if(bRight) idx += pKF->NLeft (:1562)
        MapPoint *pMPinKF = pKF->GetMapPoint(idx);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, idx);
            pKF->AddMapPoint(pMP, idx);
        }
        nFused++;
    }
    return nFused;
}

int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint)
{
    const Sim3Parts S = decompose(Scw);
    const std::set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();
    int nFused = 0;
    const int nPoints = vpPoints.size();
    Queries Q;
    for (int iMP = 0; iMP < nPoints; iMP++) {                                          // :1640-1690
        MapPoint *pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        const cvm::V3 p3Dw = cvm::vec3(pMP->GetWorldPos());
        const cvm::V3 p3Dc = cvm::mul_add(S.Rcw, p3Dw, S.tcw);
        if (p3Dc(2) < 0.0f) continue;
        const cv::Point2f uv = pKF->mpCamera->project(cv::Point3f(p3Dc(0), p3Dc(1), p3Dc(2)));
        if (!pKF->IsInImage(uv.x, uv.y)) continue;
        int nPredictedLevel;
        if (!scale_and_angle(pMP, p3Dw, S.Ow, pKF, true, nPredictedLevel)) continue;
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        Q.add(uv.x, uv.y, radius, -1.f, 0.f, nPredictedLevel - 1, nPredictedLevel, 0, pMP, iMP);
    }
    // window + level range + nearest descriptor, no reprojection gates (:1700-1717): the Fuse search with open chi-square gates
    // (inverse sigma^2 = 0) and every keypoint treated as monocular
    const int n = (int)pKF->mvKeysUn.size();
    std::vector<float> open_gates(pKF->mvInvLevelSigma2.size() ? pKF->mvInvLevelSigma2.size() : 8, 0.f);
    std::vector<int32_t> bestIdx(Q.q.size() ? Q.q.size() : 1, -1), bestDist(Q.q.size() ? Q.q.size() : 1, 256);
    const Bounds b = bounds_of(pKF);
    const int rc = orbhip_fuse_search_host(thread_ctx(), Q.q.data(), Q.dq.data(), (int)Q.q.size(), (const orbhip_keypoint *)pKF->mvKeysUn.data(),
                                           pKF->mDescriptors.ptr<uint8_t>(), nullptr, n, open_gates.data(), (int)open_gates.size(), b.min_x, b.min_y, b.max_x,
                                           b.max_y, bestIdx.data(), bestDist.data());
    if (rc != ORBHIP_OK) { fprintf(stderr, "ORBmatcher (HIP): Fuse(Scw): %d (%s)\n", rc, orbhip_last_error()); return 0; }
    for (size_t t = 0; t < Q.q.size(); t++) {                                         // :1719-1733
        if (bestDist[t] > TH_LOW) continue;
        const int iMP = Q.owner[t];
        MapPoint *pMP = vpPoints[iMP];
        MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx[t]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bestIdx[t]);
            pKF->AddMapPoint(pMP, bestIdx[t]);
        }
        nFused++;
    }
    return nFused;
}

// ---------------------------------------------------------------------------------------------------------------- SearchBySim3
int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12,
                             const float th)
{
    const float &fx = pKF1->fx, &fy = pKF1->fy, &cx = pKF1->cx, &cy = pKF1->cy;
    // Camera 1 / 2 from world
    const cvm::M3 R1w = cvm::block3(pKF1->GetRotation()), R2w = cvm::block3(pKF2->GetRotation());
    const cvm::V3 t1w = cvm::vec3(pKF1->GetTranslation()), t2w = cvm::vec3(pKF2->GetTranslation());
    // Transformation between cameras (:1756-1758): sR12 = s12*R12; sR21 = (1.0/s12)*R12.t(); t21 = -sR21*t12
    const cvm::M3 R12m = cvm::block3(R12);
    const cvm::V3 t12v = cvm::vec3(t12);
    const cvm::M3 sR12 = cvm::scale(R12m, s12);
    const cvm::M3 sR21 = cvm::scale(cvm::transpose(R12m), 1.0 / s12);
    const cvm::V3 t21 = cvm::mul(sR21, t12v, -1.0);
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = vpMapPoints1.size();
    const std::vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {                                                    // :1769-1779
        MapPoint *pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            int idx2 = std::get<0>(pMP->GetIndexInKeyFrame(pKF2));
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    std::vector<int> vnMatch1(N1, -1), vnMatch2(N2, -1);
    // one direction: the points of keyframe A projected into keyframe B (:1785-1861 with (A,B) = (1,2), :1864-1940 with (2,1))
    auto direction = [&](const std::vector<MapPoint *> &vpA, const std::vector<bool> &done, const cvm::M3 &RAw, const cvm::V3 &tAw, const cvm::M3 &sRBA,
                         const cvm::V3 &tBA, KeyFrame *pKFB, std::vector<int> &vnMatch) -> bool {
        Queries Q;
        for (int i = 0, n = vpA.size(); i < n; i++) {
            MapPoint *pMP = vpA[i];
            if (!pMP || done[i]) continue;
            if (pMP->isBad()) continue;
            const cvm::V3 p3Dw = cvm::vec3(pMP->GetWorldPos());
            const cvm::V3 p3DcA = cvm::mul_add(RAw, p3Dw, tAw);
            const cvm::V3 p3DcB = cvm::mul_add(sRBA, p3DcA, tBA);
            if (p3DcB(2) < 0.0) continue;                                              // Depth must be positive
            const float invz = 1.0 / p3DcB(2);
            const float x = p3DcB(0) * invz, y = p3DcB(1) * invz;
            const float u = fx * x + cx, v = fy * y + cy;
            if (!pKFB->IsInImage(u, v)) continue;                                      // Point must be inside the image
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            const float dist3D = cvm::norm(p3DcB);
            if (dist3D < minDistance || dist3D > maxDistance) continue;                // Depth must be inside the scale invariance region
            const int nPredictedLevel = pMP->PredictScale(dist3D, pKFB);
            const float radius = th * pKFB->mvScaleFactors[nPredictedLevel];
            Q.add(u, v, radius, -1.f, 0.f, nPredictedLevel - 1, nPredictedLevel, 0, pMP, i);
        }
        const int n = (int)pKFB->mvKeysUn.size();
        std::vector<float> open_gates(8, 0.f);
        std::vector<int32_t> bestIdx(Q.q.size() ? Q.q.size() : 1, -1), bestDist(Q.q.size() ? Q.q.size() : 1, 256);
        const Bounds b = bounds_of(pKFB);
        const int rc = orbhip_fuse_search_host(thread_ctx(), Q.q.data(), Q.dq.data(), (int)Q.q.size(), (const orbhip_keypoint *)pKFB->mvKeysUn.data(),
                                               pKFB->mDescriptors.ptr<uint8_t>(), nullptr, n, open_gates.data(), 8, b.min_x, b.min_y, b.max_x, b.max_y,
                                               bestIdx.data(), bestDist.data());
        if (rc != ORBHIP_OK) { fprintf(stderr, "ORBmatcher (HIP): SearchBySim3: %d (%s)\n", rc, orbhip_last_error()); return false; }
        for (size_t t = 0; t < Q.q.size(); t++) if (bestDist[t] <= TH_HIGH) vnMatch[Q.owner[t]] = bestIdx[t];      // :1857-1860
        return true;
    };
    if (!direction(vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2, vnMatch1)) return 0;
    if (!direction(vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12v, pKF1, vnMatch2)) return 0;
    // Check agreement (:1943-1959)
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {
        int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            int idx1 = vnMatch2[idx2];
            if (idx1 == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; nFound++; }
        }
    }
    return nFound;
}

}  // namespace ORB_SLAM3
