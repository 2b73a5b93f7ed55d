// host_latency.cc -- `host_smoke latency [reps]`: what a drop-in caller feels.  Wall-clock time per call of the signature-preserving
// classes themselves -- host cv::Mat / std::vector in, results out: argument packing, the PCIe copies, the kernels, the unpacking --
// on the shapes Tracking / LocalMapping use:
//   ORBextractor::operator()                                   include/ORBextractor.h:57       (src/Frame.cc:410-417)
//   ORBmatcher::SearchByProjection(Frame&, const Frame&, ...)  src/ORBmatcher.cc:1965          (src/Tracking.cc:1911)
//   ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&) src/ORBmatcher.cc:48            (src/Tracking.cc:3083)
//   Optimizer::PoseOptimization(Frame*)                        src/Optimizer.cc:854            (src/Tracking.cc:1934)
//   Optimizer::LocalBundleAdjustment(KeyFrame*, ...)           src/Optimizer.cc:1699           (src/LocalMapping.cc:154)
// Prints ONE JSON object on stdout (bench.py folds it into latency.host_classes next to the device-resident twins and the CPU oracle).
// Every call is timed `reps` times after warm-up calls; median and minimum are reported.  Nothing here is checked against the oracle:
// that is tests/test_gpu_host_cpp.py's job on the same classes.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "Optimizer.h"
#include "frame_cache.h"
#include "hip_context.h"
#include "host_prof.h"

extern "C" void synth_frame(uint8_t *out, int w, int h, int stride, unsigned long long seed, int frame_id);

using namespace ORB_SLAM3;

namespace {
typedef std::chrono::steady_clock Clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
struct Stat { double med, mn; };
Stat stat(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return Stat{v[v.size() / 2], v[0]};
}
struct Lcg {
    unsigned long long s;
    explicit Lcg(unsigned long long seed) : s(seed * 6364136223846793005ull + 1442695040888963407ull) {}
    double uni() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)((s >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53); }
    double sym(double a) { return (2.0 * uni() - 1.0) * a; }
};
cv::Mat eye4() { cv::Mat m = cv::Mat::eye(4, 4, CV_32F); return m; }
cv::Mat desc_row(const uint8_t *p) { cv::Mat m(1, 32, CV_8U); memcpy(m.data, p, 32); return m; }
void put(std::string &js, const char *key, Stat s, const char *what, bool last = false)
{
    char b[512];
    snprintf(b, sizeof(b), "  \"%s\": {\"host_class_ms\": %.4f, \"host_class_ms_min\": %.4f, \"what\": \"%s\"}%s\n", key, s.med, s.mn, what, last ? "" : ",");
    js += b;
}

// one 50-keyframe window of config #4's shape: 48 local keyframes (pKF + 47 covisible), 2 fixed ones, 2000 points x 10 observations
struct Window {
    Map map;
    GeometricCamera camera;
    std::vector<std::unique_ptr<KeyFrame>> kfs;
    std::vector<std::unique_ptr<MapPoint>> mps;
    Window() : camera({500.f, 500.f, 320.f, 240.f}, 0) {}
};
std::unique_ptr<Window> make_window(int nKF, int nMP, int nObs)
{
    std::unique_ptr<Window> W(new Window());
    W->map.mnInitKFid = 0;
    const std::vector<float> invS2 = {1.f, 0.694444f, 0.482253f, 0.334898f, 0.232568f, 0.161506f, 0.112157f, 0.077887f};
    Lcg rng(11);
    std::vector<double> cx(nKF);
    for (int i = 0; i < nKF; i++) {
        W->kfs.emplace_back(new KeyFrame(i, &W->map, 500.f, 500.f, 320.f, 240.f, 40.f, &W->camera));
        cx[i] = 0.05 * i;
        cv::Mat T = eye4();
        T.at<float>(0, 3) = (float)(-cx[i] + rng.sym(0.01)); T.at<float>(1, 3) = (float)rng.sym(0.01); T.at<float>(2, 3) = (float)rng.sym(0.01);     // perturbed estimate
        W->kfs[i]->SetPose(T);
        W->kfs[i]->mvInvLevelSigma2 = invS2;
    }
    for (int l = 0; l < nMP; l++) {
        const double X[3] = {-2.0 + 6.5 * rng.uni(), -1.5 + 3.0 * rng.uni(), 4.0 + 4.0 * rng.uni()};
        cv::Mat P(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) P.at<float>(k) = (float)(X[k] + rng.sym(0.02));
        W->mps.emplace_back(new MapPoint(1000 + l, P, &W->map));
        const int first = (l * 7) % (nKF - nObs + 1);
        for (int o = 0; o < nObs; o++) {
            KeyFrame *kf = W->kfs[first + o].get();
            cv::KeyPoint kp;
            kp.pt.x = (float)(500.0 * (X[0] - cx[first + o]) / X[2] + 320.0 + rng.sym(1.0));
            kp.pt.y = (float)(500.0 * X[1] / X[2] + 240.0 + rng.sym(1.0));
            kp.octave = (int)(rng.uni() * 8) & 7;
            const int idx = (int)kf->mvKeysUn.size();
            kf->mvKeysUn.push_back(kp); kf->mvuRight.push_back(-1.f); kf->mvpMapPoints.push_back(W->mps[l].get());
            W->mps[l]->AddObservation(kf, idx);
        }
    }
    for (int c = nKF - 2; c >= 2; c--) W->kfs[nKF - 1]->mvpOrderedConnectedKeyFrames.push_back(W->kfs[c].get());     // keyframes 0, 1 stay outside: fixed
    return W;
}
}  // namespace

int latency_main(int reps)
{
    if (reps < 3) reps = 3;
    const int W = 640, H = 480;
    cv::Mat im0(H, W, CV_8U), im1(H, W, CV_8U), mask;
    synth_frame(im0.data, W, H, W, 7ull, 0);
    synth_frame(im1.data, W, H, W, 7ull, 1);
    std::vector<int> lap = {0, 1000};
    std::string js = "{\n";
    char b[512];
    Frame::mnMinX = 0.f; Frame::mnMinY = 0.f; Frame::mnMaxX = (float)W; Frame::mnMaxY = (float)H;
    Frame::fx = 500.f; Frame::fy = 500.f; Frame::cx = 320.f; Frame::cy = 240.f;
    GeometricCamera camera({500.f, 500.f, 320.f, 240.f}, 0);
    Map map;

    ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kps0, kps1;
    cv::Mat desc0, desc1;
    if (ex(im0, mask, kps0, desc0, lap) < 0) { fprintf(stderr, "latency: no GPU\n"); return 3; }
    // ---- ORBextractor::operator(): lazy pyramid (the default), the first mvImagePyramid read, and the eager copy of round 3
    {
        std::vector<double> t_lazy, t_read, t_eager;
        for (int r = 0; r < reps + 3; r++) {
            const Clock::time_point t0 = Clock::now();
            ex(im1, mask, kps1, desc1, lap);
            const double a = ms_since(t0);
            const Clock::time_point t1 = Clock::now();
            volatile int rows = ex.mvImagePyramid[0].rows;                       // Frame.cc:809: what ComputeStereoMatches does first
            (void)rows;
            const double c = ms_since(t1);
            if (r >= 3) { t_lazy.push_back(a); t_read.push_back(c); }
        }
        ex.SetImagePyramidSync(true);
        for (int r = 0; r < reps + 3; r++) {
            const Clock::time_point t0 = Clock::now();
            ex(im1, mask, kps1, desc1, lap);
            if (r >= 3) t_eager.push_back(ms_since(t0));
        }
        ex.SetImagePyramidSync(false);
        put(js, "extract_one_frame", stat(t_lazy), "ORBextractor::operator()(cv::Mat 640x480) -> vector<cv::KeyPoint>, cv::Mat descriptors; pyramid left on the device");
        put(js, "extract_pyramid_first_read", stat(t_read), "first mvImagePyramid[0] access after operator() (Frame::ComputeStereoMatches): 8 levels device -> host + reflect-101 borders");
        put(js, "extract_one_frame_eager_pyramid", stat(t_eager), "operator() with SetImagePyramidSync(true): round 3's behaviour, the whole padded pyramid copied back inside every call");
    }
    // ---- Frame::ComputeStereoMatches(): the two extractions stay on the device, mvuRight / mvDepth come back
    {
        cv::Mat imR(H, W, CV_8U);
        const int disp = 9;                                                          // right view = left view moved by 9 px (rectified pair)
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) imR.ptr(y)[x] = im1.ptr(y)[std::min(x + disp, W - 1)];
        ORBextractor exR(1000, 1.2f, 8, 20, 7);
        std::vector<int> lap0 = {0, 0};
        Frame F;
        F.mpORBextractorLeft = &ex; F.mpORBextractorRight = &exR; F.mb = 40.f / 458.f; F.mbf = 40.f;
        ex(im1, mask, F.mvKeys, F.mDescriptors, lap0);
        exR(imR, mask, F.mvKeysRight, F.mDescriptorsRight, lap0);
        F.N = (int)F.mvKeys.size();
        std::vector<double> t;
        int nst = 0;
        for (int r = 0; r < reps + 3; r++) {
            const Clock::time_point t0 = Clock::now();
            F.ComputeStereoMatches();
            if (r >= 3) t.push_back(ms_since(t0));
        }
        for (float v : F.mvuRight) nst += v >= 0.f;
        char b[256];
        snprintf(b, sizeof(b), "Frame::ComputeStereoMatches(): %d left / %zu right keypoints on the device, mvuRight / mvDepth back on the host (%d matches)", F.N, F.mvKeysRight.size(), nst);
        put(js, "compute_stereo_matches_one_frame", stat(t), b);
    }
    // ---- Tracking::TrackWithMotionModel's matcher: the current frame is the one just extracted, the last frame holds map points
    ex(im0, mask, kps0, desc0, lap);
    const std::vector<cv::KeyPoint> k0 = kps0; cv::Mat d0 = desc0.clone();
    ex(im1, mask, kps1, desc1, lap);                                             // Current = the extractor's latest frame
    std::vector<float> scales(8); { float s = 1.f; for (int i = 0; i < 8; i++) { scales[i] = s; s *= 1.2f; } }
    std::vector<float> invS2(8); for (int i = 0; i < 8; i++) invS2[i] = 1.f / (scales[i] * scales[i]);
    std::vector<std::unique_ptr<MapPoint>> pool;
    Frame last, cur;
    last.N = (int)k0.size(); last.mvKeys = k0; last.mvKeysUn = k0; last.mTcw = eye4(); last.mvbOutlier.assign(last.N, false); last.mvpMapPoints.assign(last.N, nullptr);
    for (int i = 0; i < last.N; i++) {
        const float z = 2.f + 0.5f * (i % 7);
        cv::Mat P(3, 1, CV_32F);
        P.at<float>(0) = (k0[i].pt.x - 320.f) / 500.f * z; P.at<float>(1) = (k0[i].pt.y - 240.f) / 500.f * z; P.at<float>(2) = z;
        pool.emplace_back(new MapPoint(i, P, &map));
        pool.back()->mDescriptor = desc_row(d0.ptr<uint8_t>() + (size_t)32 * i);
        pool.back()->nObs = 3;
        last.mvpMapPoints[i] = pool.back().get();
    }
    cur.N = (int)kps1.size(); cur.mvKeys = kps1; cur.mvKeysUn = kps1; cur.mDescriptors = desc1; cur.mTcw = eye4(); cur.mvScaleFactors = scales; cur.mvInvLevelSigma2 = invS2;
    cur.mpCamera = &camera; cur.mbf = 40.f; cur.mb = 0.08f; cur.mvbOutlier.assign(cur.N, false); cur.mvpMapPoints.assign(cur.N, nullptr);
    cur.mvuRight.assign(cur.N, -1.f);
    int nm_last = 0, nm_map = 0;
    for (int cache = 0; cache <= 1; cache++) {
        hip::EnableFrameCache(cache != 0);
        hip::host_prof_reset();
        std::vector<double> t;
        for (int r = 0; r < reps + 3; r++) {
            std::fill(cur.mvpMapPoints.begin(), cur.mvpMapPoints.end(), static_cast<MapPoint *>(NULL));       // Tracking.cc:1901
            ORBmatcher matcher(0.9, true);
            const Clock::time_point t0 = Clock::now();
            nm_last = matcher.SearchByProjection(cur, last, 15, true);
            if (r >= 3) t.push_back(ms_since(t0));
        }
        put(js, cache ? "search_by_projection_last_frame" : "search_by_projection_last_frame_uploading_the_frame", stat(t),
            cache ? "ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, 15, bMono): ~1000 map points projected on the host, current frame resident on the device"
                  : "the same call with the frame cache off (keypoints + descriptors of the current frame uploaded)");
    }
    hip::EnableFrameCache(true);
    hip::host_prof_reset();
    {   // Tracking::SearchLocalPoints: the local map projected by the caller (mTrackProjX / Y), ~1000 points in view
        std::vector<MapPoint *> local;
        for (int i = 0; i < last.N; i++) {
            MapPoint *p = pool[i].get();
            p->mTrackProjX = k0[i].pt.x; p->mTrackProjY = k0[i].pt.y; p->mTrackProjXR = -1.f; p->mTrackViewCos = 0.999f; p->mTrackDepth = 3.f;
            p->mnTrackScaleLevel = k0[i].octave; p->mbTrackInView = true;
            local.push_back(p);
        }
        std::vector<double> t;
        for (int r = 0; r < reps + 3; r++) {
            std::fill(cur.mvpMapPoints.begin(), cur.mvpMapPoints.end(), static_cast<MapPoint *>(NULL));
            ORBmatcher matcher(0.8);
            const Clock::time_point t0 = Clock::now();
            nm_map = matcher.SearchByProjection(cur, local, 3, false, 40.f);
            if (r >= 3) t.push_back(ms_since(t0));
        }
        put(js, "search_by_projection_local_map", stat(t), "ORBmatcher::SearchByProjection(CurrentFrame, vpMapPoints, th = 3): ~1000 points in view, current frame resident on the device");
    }
    // ---- Optimizer::PoseOptimization(Frame*): 1000 monocular map point observations, 10 % gross outliers
    {
        Frame F;
        const int N = 1000;
        Lcg rng(5);
        F.N = N; F.mpCamera = &camera; F.mbf = 40.f; F.mb = 0.08f; F.mvInvLevelSigma2 = invS2; F.mvScaleFactors = scales;
        F.mvKeysUn.resize(N); F.mvKeys.resize(N); F.mvuRight.assign(N, -1.f); F.mvbOutlier.assign(N, false); F.mvpMapPoints.assign(N, nullptr);
        std::vector<std::unique_ptr<MapPoint>> pts;
        for (int i = 0; i < N; i++) {
            const double X[3] = {rng.sym(3.0), rng.sym(2.0), 4.0 + 6.0 * rng.uni()};
            cv::Mat P(3, 1, CV_32F);
            for (int k = 0; k < 3; k++) P.at<float>(k) = (float)X[k];
            pts.emplace_back(new MapPoint(i, P, &map));
            F.mvpMapPoints[i] = pts.back().get();
            cv::KeyPoint kp;
            const bool gross = (i % 10) == 9;
            kp.pt.x = (float)(500.0 * X[0] / X[2] + 320.0 + rng.sym(gross ? 40.0 : 1.0));
            kp.pt.y = (float)(500.0 * X[1] / X[2] + 240.0 + rng.sym(gross ? 40.0 : 1.0));
            kp.octave = (int)(rng.uni() * 8) & 7;
            F.mvKeysUn[i] = kp; F.mvKeys[i] = kp;
        }
        cv::Mat T0 = eye4();
        T0.at<float>(0, 3) = 0.03f; T0.at<float>(1, 3) = -0.02f; T0.at<float>(2, 3) = 0.04f;
        std::vector<double> t;
        int inl = 0;
        for (int r = 0; r < reps + 3; r++) {
            F.SetPose(T0);
            const Clock::time_point t0 = Clock::now();
            inl = Optimizer::PoseOptimization(&F);
            if (r >= 3) t.push_back(ms_since(t0));
        }
        snprintf(b, sizeof(b), "Optimizer::PoseOptimization(Frame*): 1000 map point observations gathered from MapPoint objects, 4 x 10 LM iterations (%d inliers)", inl);
        put(js, "pose_optimization_one_frame", stat(t), b);
    }
    // ---- Optimizer::LocalBundleAdjustment(pKF, &stop, pMap, num_fixedKF): 50 keyframes (2 fixed) x 2000 points x 10 observations
    {
        std::vector<double> t;
        int fixed = 0;
        const int lreps = std::max(3, reps / 4);
        for (int r = 0; r < lreps + 1; r++) {
            std::unique_ptr<Window> Wd = make_window(50, 2000, 10);                 // rebuilt per call (the call moves the map): not timed
            bool stop = false;
            const Clock::time_point t0 = Clock::now();
            Optimizer::LocalBundleAdjustment(Wd->kfs[49].get(), &stop, &Wd->map, fixed);
            if (r >= 1) t.push_back(ms_since(t0));
        }
        snprintf(b, sizeof(b), "Optimizer::LocalBundleAdjustment(KeyFrame*, ...): window selection over the KeyFrame / MapPoint graph, packing, 5 + 10 LM iterations, erase + write-back (%d fixed keyframes)", fixed);
        put(js, "local_ba_one_window", stat(t), b, true);
    }
    js += "}\n";
    fputs(js.c_str(), stdout);
    hip::host_prof_dump(stderr);
    fprintf(stderr, "latency: %d / %d matches (last frame / local map), %d reps\n", nm_last, nm_map, reps);
    return 0;
}

// `host_smoke cachecheck`: the resident-frame path (frame_cache.h) and the lazy pyramid must not change a single result.  Extracts two
// frames, runs the Tracking-side matchers on the frame that was extracted last with the cache on (device-resident train side) and off
// (everything uploaded) and compares every output; compares the lazily materialised pyramid with the eager copy; checks that a frame
// with other contents is NOT taken for the extraction.  Prints HOST_CACHE_OK.
int cachecheck_main()
{
    const int W = 640, H = 480;
    cv::Mat im0(H, W, CV_8U), im1(H, W, CV_8U), mask;
    synth_frame(im0.data, W, H, W, 21ull, 0);
    synth_frame(im1.data, W, H, W, 21ull, 1);
    std::vector<int> lap = {0, 1000};
    Frame::mnMinX = 0.f; Frame::mnMinY = 0.f; Frame::mnMaxX = (float)W; Frame::mnMaxY = (float)H;
    GeometricCamera camera({500.f, 500.f, 320.f, 240.f}, 0);
    Map map;
    ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> k0, k1;
    cv::Mat d0, d1;
    if (ex(im0, mask, k0, d0, lap) < 0) { printf("FAIL no GPU\n"); return 3; }
    d0 = d0.clone();
    // lazy pyramid == eager pyramid, byte for byte, parents included
    std::vector<std::vector<uint8_t>> lazy(8);
    std::vector<int> lw(8), lh(8);
    for (int l = 0; l < 8; l++) {
        const cv::Mat &m = ex.mvImagePyramid[l];
        lw[l] = m.cols; lh[l] = m.rows;
        for (int y = -19; y < m.rows + 19; y++) lazy[l].insert(lazy[l].end(), m.ptr(y) - 19, m.ptr(y) + m.cols + 19);
    }
    ex.SetImagePyramidSync(true);
    std::vector<cv::KeyPoint> kk; cv::Mat dd;
    ex(im0, mask, kk, dd, lap);
    for (int l = 0; l < 8; l++) {
        const cv::Mat &m = ex.mvImagePyramid[l];
        if (m.cols != lw[l] || m.rows != lh[l]) { printf("FAIL pyramid dims\n"); return 1; }
        size_t o = 0;
        for (int y = -19; y < m.rows + 19; y++, o += m.cols + 38) if (memcmp(&lazy[l][o], m.ptr(y) - 19, m.cols + 38) != 0) { printf("FAIL lazy pyramid level %d row %d\n", l, y); return 1; }
    }
    ex.SetImagePyramidSync(false);
    if (kk.size() != k0.size() || memcmp(kk.data(), k0.data(), sizeof(cv::KeyPoint) * k0.size()) != 0) { printf("FAIL repeat\n"); return 1; }
    ex(im1, mask, k1, d1, lap);                                                  // the extractor's latest frame
    std::vector<float> scales(8); { float s = 1.f; for (int i = 0; i < 8; i++) { scales[i] = s; s *= 1.2f; } }
    std::vector<std::unique_ptr<MapPoint>> pool;
    Frame last, cur;
    last.N = (int)k0.size(); last.mvKeys = k0; last.mvKeysUn = k0; last.mTcw = eye4(); last.mvbOutlier.assign(last.N, false); last.mvpMapPoints.assign(last.N, nullptr);
    for (int i = 0; i < last.N; i++) {
        const float z = 2.f + 0.5f * (i % 7);
        cv::Mat P(3, 1, CV_32F);
        P.at<float>(0) = (k0[i].pt.x - 320.f) / 500.f * z; P.at<float>(1) = (k0[i].pt.y - 240.f) / 500.f * z; P.at<float>(2) = z;
        pool.emplace_back(new MapPoint(i, P, &map));
        pool.back()->mDescriptor = desc_row(d0.ptr<uint8_t>() + (size_t)32 * i);
        pool.back()->nObs = (i % 5) ? 3 : 0;
        pool.back()->mTrackProjX = k0[i].pt.x; pool.back()->mTrackProjY = k0[i].pt.y; pool.back()->mTrackProjXR = -1.f; pool.back()->mTrackViewCos = 0.999f;
        pool.back()->mTrackDepth = 3.f; pool.back()->mnTrackScaleLevel = k0[i].octave; pool.back()->mbTrackInView = true;
        last.mvpMapPoints[i] = pool.back().get();
    }
    cur.N = (int)k1.size(); cur.mvKeys = k1; cur.mvKeysUn = k1; cur.mDescriptors = d1; cur.mTcw = eye4(); cur.mvScaleFactors = scales;
    cur.mpCamera = &camera; cur.mbf = 40.f; cur.mb = 0.08f; cur.mvbOutlier.assign(cur.N, false); cur.mvuRight.assign(cur.N, -1.f);
    std::vector<MapPoint *> local;
    for (auto &p : pool) local.push_back(p.get());
    Frame ini;                                                                   // SearchForInitialization(F1 = frame 0, F2 = the resident frame)
    ini.N = (int)k0.size(); ini.mvKeysUn = k0; ini.mDescriptors = d0;
    std::vector<MapPoint *> res[2][2]; int nm[2][3]; std::vector<int> m12[2]; std::vector<cv::Point2f> pm[2];
    for (int cache = 0; cache <= 1; cache++) {
        hip::EnableFrameCache(cache != 0);
        cur.mvpMapPoints.assign(cur.N, nullptr);
        { ORBmatcher m(0.9, true); nm[cache][0] = m.SearchByProjection(cur, last, 15, true); }
        res[cache][0] = cur.mvpMapPoints;
        cur.mvpMapPoints.assign(cur.N, nullptr);
        { ORBmatcher m(0.8); nm[cache][1] = m.SearchByProjection(cur, local, 3, false, 40.f); }
        res[cache][1] = cur.mvpMapPoints;
        pm[cache].resize(k0.size());
        for (size_t i = 0; i < k0.size(); i++) pm[cache][i] = k0[i].pt;
        { ORBmatcher m(0.9, true); nm[cache][2] = m.SearchForInitialization(ini, cur, pm[cache], m12[cache], 100); }
    }
    hip::EnableFrameCache(true);
    if (nm[0][0] != nm[1][0] || nm[0][1] != nm[1][1] || nm[0][2] != nm[1][2] || res[0][0] != res[1][0] || res[0][1] != res[1][1] || m12[0] != m12[1] ||
        memcmp(pm[0].data(), pm[1].data(), sizeof(cv::Point2f) * pm[0].size()) != 0) { printf("FAIL cache on/off differ: %d/%d %d/%d %d/%d\n", nm[0][0], nm[1][0], nm[0][1], nm[1][1], nm[0][2], nm[1][2]); return 1; }
    if (nm[1][0] < 50 || nm[1][1] < 50) { printf("FAIL too few matches %d %d\n", nm[1][0], nm[1][1]); return 1; }
    // a frame with one flipped descriptor bit is another frame: never resident
    {
        cv::Mat d2 = d1.clone();
        d2.ptr<uint8_t>()[32 * 7 + 3] ^= 0x10;
        hip::ResidentFrame r = hip::FindResident(hip::GetDevice(), k1.data(), d2.ptr<uint8_t>(), (int)k1.size());
        if (r) { printf("FAIL altered frame taken for the extraction\n"); return 1; }
        hip::ResidentFrame r2 = hip::FindResident(hip::GetDevice(), k1.data(), d1.ptr<uint8_t>(), (int)k1.size());
        if (!r2 || !r2.d_kp) { printf("FAIL the extracted frame is not resident\n"); return 1; }
        std::vector<cv::KeyPoint> k2 = k1;
        k2[3].pt.x += 0.25f;                                                     // undistorted keypoints: descriptors resident, keypoints uploaded
    }
    {
        std::vector<cv::KeyPoint> k2 = k1;
        k2[3].pt.x += 0.25f;
        hip::ResidentFrame r3 = hip::FindResident(hip::GetDevice(), k2.data(), d1.ptr<uint8_t>(), (int)k1.size());
        if (!r3 || r3.d_kp) { printf("FAIL keypoint mismatch not detected\n"); return 1; }
    }
    {   // a frame whose keypoints differ from the extraction (a distorted camera's mvKeysUn): same result as the plain upload
        Frame c2 = cur;
        c2.mvKeysUn[5].pt.x += 0.5f; c2.mvKeysUn[9].pt.y -= 0.5f;
        int a, b2;
        std::vector<MapPoint *> ra, rb;
        hip::EnableFrameCache(true);
        c2.mvpMapPoints.assign(c2.N, nullptr);
        { ORBmatcher m(0.9, true); a = m.SearchByProjection(c2, last, 15, true); } ra = c2.mvpMapPoints;
        hip::EnableFrameCache(false);
        c2.mvpMapPoints.assign(c2.N, nullptr);
        { ORBmatcher m(0.9, true); b2 = m.SearchByProjection(c2, last, 15, true); } rb = c2.mvpMapPoints;
        hip::EnableFrameCache(true);
        if (a != b2 || ra != rb) { printf("FAIL distorted-keypoint frame: %d vs %d\n", a, b2); return 1; }
    }
    printf("HOST_CACHE_OK %d %d %d matches, lazy pyramid == eager\n", nm[1][0], nm[1][1], nm[1][2]);
    return 0;
}
