// ORBmatcher.cc -- host side of the signature-preserving ORBmatcher (see header).  Each method keeps the reference's per-point
// host loop up to the point where it would call Frame::GetFeaturesInArea, turns every surviving point into one
// orbhip_proj_query, and hands the whole frame to the device: grid, window query, descriptor distances, claim rule
// (ORBmatcher.cc:110-112, 2037-2039), thresholds / ratio tests and the rotation histogram run in k_search_by_projection /
// k_search_init.  Frames of a two-camera rig (Nleft != -1) add the right camera's query per point and go through the rig entry point.
#include "ORBmatcher.h"
#include <cstdio>
#include "hip_context.h"
#include "frame_cache.h"
#include "host_prof.h"
#include <cstdlib>
#include "cvmath.h"

namespace ORB_SLAM3 {

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

namespace {
inline orbhip_ctx *thread_ctx() { return hip::ThreadContext(); }      // one context per calling thread, GPU of hip::GetDevice() (hip_context.h)

static_assert(sizeof(cv::KeyPoint) == sizeof(orbhip_keypoint), "KeyPoint layout");

// x3Dc = Rcw * x3Dw + tcw on CV_32F matrices (cvmath.h: cv::gemm's small-matrix path, float sums)
inline void transform(const cv::Mat &T, const cv::Mat &Xw, float (&Xc)[3])
{
    const cvm::V3 x = cvm::mul_add(cvm::block3(T), cvm::vec3(Xw), cvm::col3(T));
    for (int i = 0; i < 3; i++) Xc[i] = x(i);
}

// mvpMapPoints as the kernels' claim array: -1 = free (no map point, or one without observations: ORBmatcher.cc:110-112)
void claims_from(const std::vector<MapPoint *> &mps, int n, std::vector<int32_t> &tm)
{
    tm.resize(n);
    for (int i = 0; i < n; i++) tm[i] = (mps[i] && mps[i]->Observations() > 0) ? -2 : -1;
}
}  // namespace

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orbhip_descriptor_distance(a.ptr<uint8_t>(), b.ptr<uint8_t>()); }

float ORBmatcher::RadiusByViewingCos(const float &viewCos)
{
    if (viewCos > 0.998) return 2.5;
    else return 4.0;
}

// the keypoints / cross-camera links of a rig frame as the rig entry points take them: mvKeys | mvKeysRight, and for every keypoint the
// frame-wide index of the same point's keypoint in the other camera (mvLeftToRightMatch / mvRightToLeftMatch) or -1
static void rig_arrays(const Frame &F, std::vector<cv::KeyPoint> &kp, std::vector<int32_t> &mirror)
{
    kp.assign(F.mvKeys.begin(), F.mvKeys.begin() + F.Nleft);
    kp.insert(kp.end(), F.mvKeysRight.begin(), F.mvKeysRight.end());
    mirror.assign(kp.size(), -1);
    for (size_t i = 0; i < F.mvLeftToRightMatch.size() && (int)i < F.Nleft; i++) if (F.mvLeftToRightMatch[i] != -1) mirror[i] = F.mvLeftToRightMatch[i] + F.Nleft;
    for (size_t i = 0; i < F.mvRightToLeftMatch.size() && F.Nleft + i < kp.size(); i++) if (F.mvRightToLeftMatch[i] != -1) mirror[F.Nleft + i] = F.mvRightToLeftMatch[i];
}

int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th, const bool bFarPoints, const float thFarPoints)
{
    hip::HostProf prof("SearchByProjection(Frame, vpMapPoints)");
    const bool bFactor = th != 1.0;
    const bool rig = F.Nleft != -1;
    std::vector<orbhip_proj_query> q;
    std::vector<uint8_t> dq;
    std::vector<MapPoint *> owner;
    q.reserve(vpMapPoints.size()); dq.reserve(vpMapPoints.size() * 32); owner.reserve(vpMapPoints.size());
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {                       // ORBmatcher.cc:54-79, 149-156
        MapPoint *pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView && !pMP->mbTrackInViewR) continue;
        if (bFarPoints && pMP->mTrackDepth > thFarPoints) continue;
        if (pMP->isBad()) continue;
        const int obs = pMP->Observations() > 0;
        if (pMP->mbTrackInView) {
            const int &nPredictedLevel = pMP->mnTrackScaleLevel;
            // The size of the window will depend on the viewing direction
            float r = RadiusByViewingCos(pMP->mTrackViewCos);
            if (bFactor) r *= th;
            orbhip_proj_query e;
            e.u = pMP->mTrackProjX; e.v = pMP->mTrackProjY; e.radius = r * F.mvScaleFactors[nPredictedLevel];
            e.ur = pMP->mTrackProjXR; e.angle = 0.f;
            e.min_level = nPredictedLevel - 1; e.max_level = nPredictedLevel;
            e.has_obs = obs;
            q.push_back(e); owner.push_back(pMP);
            const cv::Mat MPdescriptor = pMP->GetDescriptor();
            dq.insert(dq.end(), MPdescriptor.ptr<uint8_t>(), MPdescriptor.ptr<uint8_t>() + 32);
        }
        if (rig && pMP->mbTrackInViewR) {                                         // the same point in the right camera (:149-156)
            const int &nPredictedLevel = pMP->mnTrackScaleLevelR;
            if (nPredictedLevel != -1) {
                float r = RadiusByViewingCos(pMP->mTrackViewCosR);
                orbhip_proj_query e;
                e.u = pMP->mTrackProjXR; e.v = pMP->mTrackProjYR; e.radius = r * F.mvScaleFactors[nPredictedLevel];
                e.ur = -1.f; e.angle = 0.f;
                e.min_level = nPredictedLevel - 1; e.max_level = nPredictedLevel;
                e.has_obs = obs | 2;                                              // search mGridRight
                q.push_back(e); owner.push_back(pMP);
                const cv::Mat MPdescriptor = pMP->GetDescriptor();
                dq.insert(dq.end(), MPdescriptor.ptr<uint8_t>(), MPdescriptor.ptr<uint8_t>() + 32);
            }
        }
    }
    const int n = F.N;
    std::vector<int32_t> tm;
    claims_from(F.mvpMapPoints, n, tm);
    int32_t nmatches = 0;
    int rc;
    prof.mark();
    if (!rig) {
        // F is normally the frame the extractor has just produced: its features are still on the device (frame_cache.h)
        hip::ResidentFrame res = hip::FindResident(hip::GetDevice(), F.mvKeysUn.data(), F.mDescriptors.ptr<uint8_t>(), n);
        if (res)
            rc = orbhip_search_by_projection_host_resident(thread_ctx(), 1, q.data(), dq.data(), (int)q.size(), (const orbhip_keypoint *)F.mvKeysUn.data(), res.d_kp,
                                                           res.d_desc, F.mvuRight.empty() ? nullptr : F.mvuRight.data(), n, Frame::mnMinX, Frame::mnMinY,
                                                           Frame::mnMaxX, Frame::mnMaxY, TH_HIGH, mfNNratio, 0, tm.data(), &nmatches);
        else
            rc = orbhip_search_by_projection_host(thread_ctx(), 1, q.data(), dq.data(), (int)q.size(), (const orbhip_keypoint *)F.mvKeysUn.data(),
                                                  F.mDescriptors.ptr<uint8_t>(), F.mvuRight.empty() ? nullptr : F.mvuRight.data(), n, Frame::mnMinX,
                                                  Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY, TH_HIGH, mfNNratio, 0, tm.data(), &nmatches);
    } else {
        std::vector<cv::KeyPoint> kp; std::vector<int32_t> mirror;
        rig_arrays(F, kp, mirror);
        rc = orbhip_search_by_projection_rig_host(thread_ctx(), 1, q.data(), dq.data(), (int)q.size(), (const orbhip_keypoint *)kp.data(),
                                                  F.mDescriptors.ptr<uint8_t>(), n, F.Nleft, mirror.data(), Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX,
                                                  Frame::mnMaxY, TH_HIGH, mfNNratio, 0, tm.data(), &nmatches);
    }
    prof.mark();
    if (rc != ORBHIP_OK) { fprintf(stderr, "ORBmatcher (HIP): SearchByProjection: %d (%s)\n", rc, orbhip_last_error()); return 0; }
    for (int i = 0; i < n; i++) if (tm[i] >= 0) F.mvpMapPoints[i] = owner[tm[i]];  // F.mvpMapPoints[bestIdx]=pMP (:140, :145, :205, :210)
    return nmatches;
}

int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
{
    hip::HostProf prof("SearchByProjection(CurrentFrame, LastFrame)");
    const bool rig = CurrentFrame.Nleft != -1;
    // twc = -Rcw.t()*tcw ; tlc = Rlw*twc + tlw (ORBmatcher.cc:1976-1984), CV_32F matrix products as cvmath.h spells them
    const cvm::V3 twc_ = cvm::mul_t(cvm::block3(CurrentFrame.mTcw), cvm::col3(CurrentFrame.mTcw), -1.0);
    const cvm::V3 tlc_ = cvm::mul_add(cvm::block3(LastFrame.mTcw), twc_, cvm::col3(LastFrame.mTcw));
    const float tlc[3] = {tlc_(0), tlc_(1), tlc_(2)};
    const bool bForward = tlc[2] > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc[2] > CurrentFrame.mb && !bMono;

    std::vector<orbhip_proj_query> q;
    std::vector<uint8_t> dq;
    std::vector<MapPoint *> owner;
    q.reserve((size_t)LastFrame.N * (rig ? 2 : 1)); dq.reserve((size_t)LastFrame.N * (rig ? 64 : 32)); owner.reserve((size_t)LastFrame.N * (rig ? 2 : 1));
    for (int i = 0; i < LastFrame.N; i++) {                                        // :1989-2023
        MapPoint *pMP = LastFrame.mvpMapPoints[i];
        if (!pMP) continue;
        if (LastFrame.mvbOutlier[i]) continue;
        // Project
        float x3Dc[3];
        transform(CurrentFrame.mTcw, pMP->GetWorldPos(), x3Dc);
        const float invzc = 1.0 / x3Dc[2];
        if (invzc < 0) continue;
        // (the reference hands project() the cv::Mat x3Dc, :2003: Pinhole.cpp:41-47 / KannalaBrandt8.cpp:71-76 read its three floats and call
        // the cv::Point3f overload, which is called directly here -- one heap allocation per map point less)
        cv::Point2f uv = CurrentFrame.mpCamera->project(cv::Point3f(x3Dc[0], x3Dc[1], x3Dc[2]));
        if (uv.x < CurrentFrame.mnMinX || uv.x > CurrentFrame.mnMaxX) continue;
        if (uv.y < CurrentFrame.mnMinY || uv.y > CurrentFrame.mnMaxY) continue;
        int nLastOctave = (LastFrame.Nleft == -1 || i < LastFrame.Nleft) ? LastFrame.mvKeys[i].octave : LastFrame.mvKeysRight[i - LastFrame.Nleft].octave;
        // Search in a window. Size depends on scale
        float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
        orbhip_proj_query e;
        e.u = uv.x; e.v = uv.y; e.radius = radius;
        e.ur = uv.x - CurrentFrame.mbf * invzc;                                   // :2043
        const cv::KeyPoint &kpLF = (LastFrame.Nleft == -1) ? LastFrame.mvKeysUn[i] : (i < LastFrame.Nleft) ? LastFrame.mvKeys[i] : LastFrame.mvKeysRight[i - LastFrame.Nleft];
        e.angle = kpLF.angle;                                                     // :2067-2073
        if (bForward) { e.min_level = nLastOctave; e.max_level = -1; }            // GetFeaturesInArea(.., nLastOctave)
        else if (bBackward) { e.min_level = 0; e.max_level = nLastOctave; }
        else { e.min_level = nLastOctave - 1; e.max_level = nLastOctave + 1; }
        e.has_obs = pMP->Observations() > 0;
        q.push_back(e); owner.push_back(pMP);
        const cv::Mat dMP = pMP->GetDescriptor();
        dq.insert(dq.end(), dMP.ptr<uint8_t>(), dMP.ptr<uint8_t>() + 32);
        if (rig) {                                                                // the same point in the right camera (:2089-2105)
            const cvm::V3 xr_ = cvm::mul_add(cvm::block3(CurrentFrame.mTrl), cvm::V3{{x3Dc[0], x3Dc[1], x3Dc[2]}}, cvm::col3(CurrentFrame.mTrl));   // mTrl.R * x3Dc + mTrl.col(3)
            const float x3Dr[3] = {xr_(0), xr_(1), xr_(2)};
            const cv::Point2f uvr = CurrentFrame.mpCamera->project(cv::Point3f(x3Dr[0], x3Dr[1], x3Dr[2]));      // (the reference projects through mpCamera here, :2092)
            orbhip_proj_query er = e;
            er.u = uvr.x; er.v = uvr.y; er.has_obs = e.has_obs | 2;
            q.push_back(er); owner.push_back(pMP);
            dq.insert(dq.end(), dMP.ptr<uint8_t>(), dMP.ptr<uint8_t>() + 32);
        }
    }
    const int n = CurrentFrame.N;
    std::vector<int32_t> tm;
    claims_from(CurrentFrame.mvpMapPoints, n, tm);
    int32_t nmatches = 0;
    int rc;
    prof.mark();
    if (!rig) {
        // CurrentFrame is the frame the extractor has just produced (Tracking::TrackWithMotionModel, src/Tracking.cc:1911): its keypoints
        // and descriptors are still on the device, only the queries travel (frame_cache.h)
        hip::ResidentFrame res = hip::FindResident(hip::GetDevice(), CurrentFrame.mvKeysUn.data(), CurrentFrame.mDescriptors.ptr<uint8_t>(), n);
        if (res)
            rc = orbhip_search_by_projection_host_resident(thread_ctx(), 0, q.data(), dq.data(), (int)q.size(), (const orbhip_keypoint *)CurrentFrame.mvKeysUn.data(),
                                                           res.d_kp, res.d_desc, CurrentFrame.mvuRight.empty() ? nullptr : CurrentFrame.mvuRight.data(), n, Frame::mnMinX,
                                                           Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY, TH_HIGH, 0.f, mbCheckOrientation ? 1 : 0, tm.data(), &nmatches);
        else
            rc = orbhip_search_by_projection_host(thread_ctx(), 0, q.data(), dq.data(), (int)q.size(), (const orbhip_keypoint *)CurrentFrame.mvKeysUn.data(),
                                                  CurrentFrame.mDescriptors.ptr<uint8_t>(), CurrentFrame.mvuRight.empty() ? nullptr : CurrentFrame.mvuRight.data(),
                                                  n, Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY, TH_HIGH, 0.f, mbCheckOrientation ? 1 : 0,
                                                  tm.data(), &nmatches);
    } else {
        std::vector<cv::KeyPoint> kp; std::vector<int32_t> mirror;
        rig_arrays(CurrentFrame, kp, mirror);
        rc = orbhip_search_by_projection_rig_host(thread_ctx(), 0, q.data(), dq.data(), (int)q.size(), (const orbhip_keypoint *)kp.data(),
                                                  CurrentFrame.mDescriptors.ptr<uint8_t>(), n, CurrentFrame.Nleft, nullptr, Frame::mnMinX, Frame::mnMinY,
                                                  Frame::mnMaxX, Frame::mnMaxY, TH_HIGH, 0.f, mbCheckOrientation ? 1 : 0, tm.data(), &nmatches);
    }
    prof.mark();
    if (rc != ORBHIP_OK) { fprintf(stderr, "ORBmatcher (HIP): SearchByProjection: %d (%s)\n", rc, orbhip_last_error()); return 0; }
    // A keypoint taken in this call holds the taker's map point.  One that was taken and then dropped by the rotation check comes
    // back as free (-1): the reference sets it to NULL (:2170), which it already is at the reference's call sites
    // (Tracking::TrackWithMotionModel fills mvpMapPoints with NULL before both calls, src/Tracking.cc:1901, 1917).
    for (int i = 0; i < n; i++)
        if (tm[i] >= 0) CurrentFrame.mvpMapPoints[i] = owner[tm[i]];
    return nmatches;
}

int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize)
{
    const int n1 = F1.mvKeysUn.size(), n2 = F2.mvKeysUn.size();
    vnMatches12 = std::vector<int>(n1, -1);                                        // :713
    static_assert(sizeof(cv::Point2f) == 8 && sizeof(int) == 4, "layout");
    int32_t nmatches = 0;
    // F2 is the frame just extracted (Tracking::MonocularInitialization, src/Tracking.cc:1506): resident on the device (frame_cache.h)
    hip::ResidentFrame res = hip::FindResident(hip::GetDevice(), F2.mvKeysUn.data(), n2 ? F2.mDescriptors.ptr<uint8_t>() : nullptr, n2);
    const int rc = res ? orbhip_search_for_initialization_host_resident(thread_ctx(), (const orbhip_keypoint *)F1.mvKeysUn.data(), F1.mDescriptors.ptr<uint8_t>(), n1,
                                                                        (const orbhip_keypoint *)F2.mvKeysUn.data(), res.d_kp, res.d_desc, n2, Frame::mnMinX, Frame::mnMinY,
                                                                        Frame::mnMaxX, Frame::mnMaxY, windowSize, mfNNratio, mbCheckOrientation ? 1 : 0,
                                                                        (float *)vbPrevMatched.data(), (int32_t *)vnMatches12.data(), &nmatches)
                       : orbhip_search_for_initialization_host(thread_ctx(), (const orbhip_keypoint *)F1.mvKeysUn.data(), F1.mDescriptors.ptr<uint8_t>(), n1,
                                                               (const orbhip_keypoint *)F2.mvKeysUn.data(), F2.mDescriptors.ptr<uint8_t>(), n2, Frame::mnMinX,
                                                               Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY, windowSize, mfNNratio, mbCheckOrientation ? 1 : 0,
                                                               (float *)vbPrevMatched.data(), (int32_t *)vnMatches12.data(), &nmatches);
    if (rc != ORBHIP_OK) { fprintf(stderr, "ORBmatcher (HIP): SearchForInitialization: %d (%s)\n", rc, orbhip_last_error()); return 0; }
    return nmatches;
}

}  // namespace ORB_SLAM3

namespace ORB_SLAM3 {

// src/ORBmatcher.cc:273-475.  The node-by-node walk over the two FeatureVectors, the per-node best / second-best search with
// the "frame feature already matched" rule, the ratio test and the rotation histogram run in k_search_by_bow; the method
// flattens the two std::map containers in their iteration order and maps the result back to MapPoint pointers.
int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches)
{
    const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));
    auto flatten = [](const DBoW2::FeatureVector &fv, std::vector<int32_t> &ids, std::vector<int32_t> &start, std::vector<int32_t> &feat) {
        start.push_back(0);
        for (const auto &kv : fv) {
            ids.push_back((int32_t)kv.first);
            for (unsigned int i : kv.second) feat.push_back((int32_t)i);
            start.push_back((int32_t)feat.size());
        }
    };
    std::vector<int32_t> kIds, kStart, kFeat, fIds, fStart, fFeat;
    flatten(pKF->mFeatVec, kIds, kStart, kFeat);
    flatten(F.mFeatVec, fIds, fStart, fFeat);
    const bool rig = F.Nleft != -1;
    const int nK = rig ? (int)(pKF->mvKeysUn.size() + pKF->mvKeysRight.size()) : (int)pKF->mvKeysUn.size();
    std::vector<uint8_t> valid(nK > 0 ? nK : 1, 0);
    for (int k = 0; k < nK && k < (int)vpMapPointsKF.size(); k++) {
        MapPoint *pMP = vpMapPointsKF[k];
        valid[k] = (pMP && !pMP->isBad()) ? 1 : 0;                   // :299-305
    }
    std::vector<cv::KeyPoint> kk, fk;                                  // rig: left | right keypoints in descriptor order (:309-311, :393-400)
    const cv::KeyPoint *pk = pKF->mvKeysUn.data(), *pf = F.mvKeys.data();
    if (rig) {
        kk = pKF->mvKeysUn; kk.insert(kk.end(), pKF->mvKeysRight.begin(), pKF->mvKeysRight.end()); pk = kk.data();
        fk = F.mvKeys; fk.insert(fk.end(), F.mvKeysRight.begin(), F.mvKeysRight.end()); pf = fk.data();
    }
    std::vector<int32_t> matchF(F.N > 0 ? F.N : 1, -1);
    int32_t nmatches = 0;
    // F is the frame just extracted (Tracking::TrackReferenceKeyFrame, src/Tracking.cc:1757): resident on the device (frame_cache.h)
    hip::ResidentFrame res;
    if (!rig) res = hip::FindResident(hip::GetDevice(), pf, F.N > 0 ? F.mDescriptors.ptr<uint8_t>() : nullptr, F.N);
    const int rc = res ? orbhip_search_by_bow_host_resident(thread_ctx(), kIds.data(), kStart.data(), kFeat.data(), (int)kIds.size(), valid.data(),
                                                            (const orbhip_keypoint *)pk, pKF->mDescriptors.ptr<uint8_t>(), nK, fIds.data(), fStart.data(), fFeat.data(),
                                                            (int)fIds.size(), (const orbhip_keypoint *)pf, res.d_kp, res.d_desc, F.N, mfNNratio,
                                                            mbCheckOrientation ? 1 : 0, matchF.data(), &nmatches)
                       : orbhip_search_by_bow_host(thread_ctx(), kIds.data(), kStart.data(), kFeat.data(), (int)kIds.size(), valid.data(),
                                                   (const orbhip_keypoint *)pk, pKF->mDescriptors.ptr<uint8_t>(), nK,
                                                   fIds.data(), fStart.data(), fFeat.data(), (int)fIds.size(), (const orbhip_keypoint *)pf,
                                                   F.mDescriptors.ptr<uint8_t>(), F.N, rig ? F.Nleft : -1, mfNNratio, mbCheckOrientation ? 1 : 0,
                                                   matchF.data(), &nmatches);
    if (rc != ORBHIP_OK) { fprintf(stderr, "orbhip SearchByBoW: %s\n", orbhip_last_error()); return 0; }
    for (int j = 0; j < F.N; j++)
        if (matchF[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[matchF[j]];
    return nmatches;
}

}  // namespace ORB_SLAM3
