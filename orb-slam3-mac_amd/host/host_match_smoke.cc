// host_match_smoke.cc -- `host_smoke match <in> <out>`: builds Frames + MapPoints from a flat description, calls the ORBmatcher
// methods with the reference's signatures and dumps what they did to the frames (tests/test_gpu_host_cpp.py compares it with the
// matcher oracle run on the same inputs).
#include <cstdio>
#include <cstdlib>
#include <memory>
#include "ORBmatcher.h"

using namespace ORB_SLAM3;

float Frame::mnMinX = 0.f, Frame::mnMaxX = 0.f, Frame::mnMinY = 0.f, Frame::mnMaxY = 0.f;
float Frame::fx = 0.f, Frame::fy = 0.f, Frame::cx = 0.f, Frame::cy = 0.f;

namespace {
struct Reader {
    FILE *f;
    explicit Reader(const char *p) : f(fopen(p, "rb")) {}
    ~Reader() { if (f) fclose(f); }
    template <typename T> std::vector<T> vec(size_t n) { std::vector<T> v(n); if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } return v; }
};
void put(FILE *f, const std::vector<int32_t> &v) { if (!v.empty()) fwrite(v.data(), 4, v.size(), f); }

cv::Mat desc_mat(const std::vector<uint8_t> &d, int n)
{
    cv::Mat m(n > 0 ? n : 1, 32, CV_8U);
    if (n) memcpy(m.data, d.data(), (size_t)n * 32);
    return m;
}
cv::Mat desc_row(const uint8_t *p) { cv::Mat m(1, 32, CV_8U); memcpy(m.data, p, 32); return m; }
}  // namespace

// in:  int32[8] {n (current frame keypoints), nLast (last frame keypoints), nMap (local map points), bMono, nInit1, nInit2, window, 0}
//      float[12] {minX, minY, maxX, maxY, fx, fy, cx, cy, mbf, mb, th, nnratio}; float[8] mvScaleFactors
//      current frame: KeyPoint[n] mvKeysUn, u8[n*32], float[n] mvuRight, int32[n] initial holder (-1 none, 0 map point without
//        observations, 1 map point with observations), float[16] mTcw
//      last frame: KeyPoint[nLast] mvKeys (= mvKeysUn), float[16] mTcw, int32[nLast] has map point, int32[nLast] outlier,
//        int32[nLast] observations of that point, float[nLast*3] world positions, u8[nLast*32] descriptors
//      local map: float[nMap*8] {mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos, mTrackDepth, level, inView, nObs}, u8[nMap*32]
//      initialisation pair: KeyPoint[nInit1], u8[nInit1*32], KeyPoint[nInit2], u8[nInit2*32], float[nInit1*2] vbPrevMatched
// out: int32 nmatches (last frame), int32[n] index of the LAST-frame keypoint whose point each current keypoint holds (-1 none, -2 kept);
//      int32 nmatches (local map), int32[n] index of the map point held (-1 / -2); int32 nmatches (init), int32[nInit1] vnMatches12,
//      float[nInit1*2] vbPrevMatched
int match_smoke(const char *in, const char *out)
{
    Reader r(in);
    if (!r.f) { fprintf(stderr, "cannot open %s\n", in); return 2; }
    const std::vector<int32_t> hd = r.vec<int32_t>(8);
    const int n = hd[0], nLast = hd[1], nMap = hd[2], nI1 = hd[4], nI2 = hd[5], window = hd[6];
    const bool bMono = hd[3] != 0;
    const std::vector<float> fp = r.vec<float>(12), scales = r.vec<float>(8);
    Frame::mnMinX = fp[0]; Frame::mnMinY = fp[1]; Frame::mnMaxX = fp[2]; Frame::mnMaxY = fp[3];
    GeometricCamera camera({fp[4], fp[5], fp[6], fp[7]}, 0);
    Map map;
    static_assert(sizeof(cv::KeyPoint) == 28, "layout");
    const std::vector<cv::KeyPoint> kp = r.vec<cv::KeyPoint>(n);
    const std::vector<uint8_t> d = r.vec<uint8_t>((size_t)n * 32);
    const std::vector<float> ur = r.vec<float>(n);
    const std::vector<int32_t> holder = r.vec<int32_t>(n);
    const std::vector<float> Tcw = r.vec<float>(16);
    const std::vector<cv::KeyPoint> kpL = r.vec<cv::KeyPoint>(nLast);
    const std::vector<float> Tlw = r.vec<float>(16);
    const std::vector<int32_t> hasMP = r.vec<int32_t>(nLast), outl = r.vec<int32_t>(nLast), nobs = r.vec<int32_t>(nLast);
    const std::vector<float> Xw = r.vec<float>((size_t)nLast * 3);
    const std::vector<uint8_t> dL = r.vec<uint8_t>((size_t)nLast * 32);
    const std::vector<float> mp = r.vec<float>((size_t)nMap * 8);
    const std::vector<uint8_t> dM = r.vec<uint8_t>((size_t)nMap * 32);
    const std::vector<cv::KeyPoint> k1 = r.vec<cv::KeyPoint>(nI1);
    const std::vector<uint8_t> d1 = r.vec<uint8_t>((size_t)nI1 * 32);
    const std::vector<cv::KeyPoint> k2 = r.vec<cv::KeyPoint>(nI2);
    const std::vector<uint8_t> d2 = r.vec<uint8_t>((size_t)nI2 * 32);
    const std::vector<float> prev = r.vec<float>((size_t)nI1 * 2);
    // rig scenario (appended): int32 nleft; int32[n] cross-camera link of every keypoint (frame-wide index or -1); float[12] mTrl (3x4);
    // float[nMap*5] {mTrackProjXR, mTrackProjYR, mTrackViewCosR, mnTrackScaleLevelR, mbTrackInViewR}
    const int nleftRig = r.vec<int32_t>(1)[0];
    const std::vector<int32_t> link = r.vec<int32_t>(n);
    const std::vector<float> trl = r.vec<float>(12);
    const std::vector<float> mpR = r.vec<float>((size_t)nMap * 5);

    auto mat44 = [](const std::vector<float> &p) { cv::Mat m(4, 4, CV_32F); for (int i = 0; i < 16; i++) m.at<float>(i / 4, i % 4) = p[i]; return m; };
    std::vector<std::unique_ptr<MapPoint>> pool;
    auto new_mp = [&](const float *X, const uint8_t *desc, int observations) {
        cv::Mat P(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) P.at<float>(k) = X ? X[k] : 0.f;
        pool.emplace_back(new MapPoint(pool.size(), P, &map));
        pool.back()->mDescriptor = desc_row(desc);
        pool.back()->nObs = observations;
        return pool.back().get();
    };
    static const uint8_t zero32[32] = {0};
    auto make_current = [&](Frame &F) {
        F.N = n; F.mvKeys = kp; F.mvKeysUn = kp; F.mvuRight = ur; F.mDescriptors = desc_mat(d, n); F.mTcw = mat44(Tcw);
        F.mvScaleFactors = scales; F.mpCamera = &camera; F.mbf = fp[8]; F.mb = fp[9]; F.mvbOutlier.assign(n, false);
        F.mvpMapPoints.assign(n, nullptr);
        for (int i = 0; i < n; i++) if (holder[i] >= 0) F.mvpMapPoints[i] = new_mp(nullptr, zero32, holder[i] ? 3 : 0);
    };
    FILE *fo = fopen(out, "wb");
    if (!fo) return 2;
    // ---- SearchByProjection(CurrentFrame, LastFrame, th, bMono): Tracking::TrackWithMotionModel
    {
        Frame cur, last;
        make_current(cur);
        last.N = nLast; last.mvKeys = kpL; last.mvKeysUn = kpL; last.mTcw = mat44(Tlw); last.mvpMapPoints.assign(nLast, nullptr);
        last.mvbOutlier.assign(nLast, false);
        std::vector<MapPoint *> lastMP(nLast, nullptr);
        for (int i = 0; i < nLast; i++) {
            if (hasMP[i]) last.mvpMapPoints[i] = lastMP[i] = new_mp(&Xw[(size_t)3 * i], &dL[(size_t)32 * i], nobs[i]);
            last.mvbOutlier[i] = outl[i] != 0;
        }
        const std::vector<MapPoint *> before = cur.mvpMapPoints;
        ORBmatcher matcher(0.9, true);                                          // Tracking.cc:1881
        const int nm = matcher.SearchByProjection(cur, last, fp[10], bMono);
        std::vector<int32_t> res(n, -1);
        for (int i = 0; i < n; i++) {
            if (cur.mvpMapPoints[i] == nullptr) continue;
            res[i] = -2;
            if (cur.mvpMapPoints[i] != before[i]) for (int j = 0; j < nLast; j++) if (lastMP[j] == cur.mvpMapPoints[i]) res[i] = j;
        }
        fwrite(&nm, 4, 1, fo); put(fo, res);
    }
    // ---- SearchByProjection(F, vpMapPoints, th): Tracking::SearchLocalPoints
    {
        Frame cur;
        make_current(cur);
        std::vector<MapPoint *> local(nMap);
        for (int j = 0; j < nMap; j++) {
            const float *m = &mp[(size_t)8 * j];
            MapPoint *p = new_mp(nullptr, &dM[(size_t)32 * j], (int)m[7]);
            p->mTrackProjX = m[0]; p->mTrackProjY = m[1]; p->mTrackProjXR = m[2]; p->mTrackViewCos = m[3]; p->mTrackDepth = m[4];
            p->mnTrackScaleLevel = (int)m[5]; p->mbTrackInView = m[6] != 0.f;
            local[j] = p;
        }
        const std::vector<MapPoint *> before = cur.mvpMapPoints;
        ORBmatcher matcher(fp[11]);                                             // Tracking.cc:3083: ORBmatcher matcher(0.8)
        const int nm = matcher.SearchByProjection(cur, local, fp[10], true, 40.0f);
        std::vector<int32_t> res(n, -1);
        for (int i = 0; i < n; i++) {
            if (cur.mvpMapPoints[i] == nullptr) continue;
            res[i] = -2;
            if (cur.mvpMapPoints[i] != before[i]) for (int j = 0; j < nMap; j++) if (local[j] == cur.mvpMapPoints[i]) res[i] = j;
        }
        fwrite(&nm, 4, 1, fo); put(fo, res);
    }
    // ---- SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize): Tracking::MonocularInitialization
    {
        Frame F1, F2;
        F1.N = nI1; F1.mvKeysUn = k1; F1.mDescriptors = desc_mat(d1, nI1);
        F2.N = nI2; F2.mvKeysUn = k2; F2.mDescriptors = desc_mat(d2, nI2);
        std::vector<cv::Point2f> vbPrevMatched(nI1);
        for (int i = 0; i < nI1; i++) vbPrevMatched[i] = cv::Point2f(prev[2 * i], prev[2 * i + 1]);
        std::vector<int> vnMatches12;
        ORBmatcher matcher(0.9, true);                                          // Tracking.cc:1506
        const int nm = matcher.SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, window);
        fwrite(&nm, 4, 1, fo);
        std::vector<int32_t> m12(vnMatches12.begin(), vnMatches12.end());
        put(fo, m12);
        if (nI1) fwrite(vbPrevMatched.data(), 8, nI1, fo);
    }
    // ---- the same two searches on a two-camera rig frame (Nleft != -1): left | right keypoints, mvLeftToRightMatch / mvRightToLeftMatch
    auto make_rig = [&](Frame &F) {
        make_current(F);
        F.Nleft = nleftRig; F.Nright = n - nleftRig;
        F.mvKeys.assign(kp.begin(), kp.begin() + nleftRig); F.mvKeysRight.assign(kp.begin() + nleftRig, kp.end()); F.mvKeysUn.clear();
        F.mvuRight.assign(n, -1.f);
        F.mvLeftToRightMatch.assign(nleftRig, -1); F.mvRightToLeftMatch.assign(n - nleftRig, -1);
        for (int i = 0; i < n; i++) {
            if (link[i] < 0) continue;
            if (i < nleftRig) F.mvLeftToRightMatch[i] = link[i] - nleftRig; else F.mvRightToLeftMatch[i - nleftRig] = link[i];
        }
        F.mTrl = cv::Mat(3, 4, CV_32F);
        for (int i = 0; i < 12; i++) F.mTrl.at<float>(i / 4, i % 4) = trl[i];
    };
    {
        Frame cur, last;
        make_rig(cur);
        last.N = nLast; last.mvKeys = kpL; last.mvKeysUn = kpL; last.mTcw = mat44(Tlw); last.mvpMapPoints.assign(nLast, nullptr);
        last.mvbOutlier.assign(nLast, false);
        std::vector<MapPoint *> lastMP(nLast, nullptr);
        for (int i = 0; i < nLast; i++) {
            if (hasMP[i]) last.mvpMapPoints[i] = lastMP[i] = new_mp(&Xw[(size_t)3 * i], &dL[(size_t)32 * i], nobs[i]);
            last.mvbOutlier[i] = outl[i] != 0;
        }
        const std::vector<MapPoint *> before = cur.mvpMapPoints;
        ORBmatcher matcher(0.9, true);
        const int nm = matcher.SearchByProjection(cur, last, fp[10], bMono);
        std::vector<int32_t> res(n, -1);
        for (int i = 0; i < n; i++) {
            if (cur.mvpMapPoints[i] == nullptr) continue;
            res[i] = -2;
            if (cur.mvpMapPoints[i] != before[i]) for (int j = 0; j < nLast; j++) if (lastMP[j] == cur.mvpMapPoints[i]) res[i] = j;
        }
        fwrite(&nm, 4, 1, fo); put(fo, res);
    }
    {
        Frame cur;
        make_rig(cur);
        std::vector<MapPoint *> local(nMap);
        for (int j = 0; j < nMap; j++) {
            const float *m = &mp[(size_t)8 * j], *mr = &mpR[(size_t)5 * j];
            MapPoint *p = new_mp(nullptr, &dM[(size_t)32 * j], (int)m[7]);
            p->mTrackProjX = m[0]; p->mTrackProjY = m[1]; p->mTrackProjXR = mr[0]; p->mTrackProjYR = mr[1]; p->mTrackViewCos = m[3]; p->mTrackDepth = m[4];
            p->mnTrackScaleLevel = (int)m[5]; p->mbTrackInView = m[6] != 0.f;
            p->mTrackViewCosR = mr[2]; p->mnTrackScaleLevelR = (int)mr[3]; p->mbTrackInViewR = mr[4] != 0.f;
            local[j] = p;
        }
        const std::vector<MapPoint *> before = cur.mvpMapPoints;
        ORBmatcher matcher(fp[11]);
        const int nm = matcher.SearchByProjection(cur, local, fp[10], true, 40.0f);
        std::vector<int32_t> res(n, -1);
        for (int i = 0; i < n; i++) {
            if (cur.mvpMapPoints[i] == nullptr) continue;
            res[i] = -2;
            if (cur.mvpMapPoints[i] != before[i]) for (int j = 0; j < nMap; j++) if (local[j] == cur.mvpMapPoints[i]) res[i] = j;
        }
        fwrite(&nm, 4, 1, fo); put(fo, res);
    }
    fclose(fo);
    printf("HOST_MATCH_OK\n");
    return 0;
}
