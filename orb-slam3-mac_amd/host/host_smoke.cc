// host_smoke.cc -- exercises the signature-preserving C++ classes end to end (needs a GPU to run; compiles anywhere).
// Built by the package Makefile, run by tests/test_gpu_host_cpp.py:
//   host_smoke                      ORBextractor / ORBmatcher::DescriptorDistance call shapes of Frame.cc
//   host_smoke lba  <in> <out>      builds a KeyFrame / MapPoint / Map pointer graph from a flat description, runs
//                                   Optimizer::LocalBundleAdjustment on it and dumps the resulting map (the test compares it
//                                   with the CPU oracle run on the same window)
//   host_smoke match <in> <out>     builds Frames + MapPoints, runs the ORBmatcher methods and dumps their results
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "Optimizer.h"

extern "C" void synth_frame(uint8_t *out, int w, int h, int stride, unsigned long long seed, int frame_id);

using namespace ORB_SLAM3;

static int extractor_smoke()
{
    const int W = 640, H = 480;
    cv::Mat im(H, W, CV_8U), mask, desc;
    synth_frame(im.data, W, H, W, 7ull, 0);
    ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> kps;
    std::vector<int> lap = {0, 1000};
    int mono = ex(im, mask, kps, desc, lap);                       // Frame.cc:302 / :412-416 call shape
    if (mono != 0 || kps.size() < 800 || desc.rows != (int)kps.size()) { printf("FAIL extract %d %zu\n", mono, kps.size()); return 1; }
    // mvImagePyramid is filled by operator() itself (Frame.cc:809,899,913,918 read it with no further call)
    if (ex.mvImagePyramid.size() != 8 || ex.mvImagePyramid[0].cols != W || ex.mvImagePyramid[7].cols != 179) { printf("FAIL pyramid\n"); return 1; }
    if (ex.mvImagePyramid[0].ptr(5)[7] != im.ptr(5)[7]) { printf("FAIL pyramid content\n"); return 1; }
    // the levels are ROI views at (19,19) of reflect-101 padded parents (ORBextractor.cc:1160-1173): the SAD windows of
    // ComputeStereoMatches run up to 5+5 columns past the ROI (Frame.cc:899-918)
    {
        const cv::Mat &L3 = ex.mvImagePyramid[3];
        if (L3.ptr(0)[-1] != L3.ptr(0)[1] || L3.ptr(-1)[4] != L3.ptr(1)[4] || L3.ptr(0)[L3.cols] != L3.ptr(0)[L3.cols - 2]) { printf("FAIL pyramid border\n"); return 1; }
    }
    const std::vector<cv::KeyPoint> first = kps;
    cv::Mat empty;
    if (ex(empty, mask, kps, desc, lap) != -1) { printf("FAIL empty\n"); return 1; }
    ex.SetImagePyramidSync(false);                                 // opt-out for throughput: no D2H of the pyramid
    mono = ex(im, mask, kps, desc, lap);
    if (kps.size() != first.size() || memcmp(kps.data(), first.data(), sizeof(cv::KeyPoint) * kps.size()) != 0) { printf("FAIL repeat\n"); return 1; }
    if (ex.GetLevels() != 8 || ex.GetScaleFactors()[1] != 1.2f) { printf("FAIL getters\n"); return 1; }
    int d = ORBmatcher::DescriptorDistance(desc.row(0), desc.row(0));
    int d2 = ORBmatcher::DescriptorDistance(desc.row(0), desc.row(1));
    if (d != 0 || d2 <= 0) { printf("FAIL distance\n"); return 1; }
    printf("HOST_CPP_OK %zu keypoints, d01=%d\n", kps.size(), d2);
    return 0;
}

// ---- flat file helpers (little-endian, written / read by tests/test_gpu_host_cpp.py with numpy)
struct Reader {
    FILE *f;
    explicit Reader(const char *p) : f(fopen(p, "rb")) {}
    ~Reader() { if (f) fclose(f); }
    template <typename T> std::vector<T> vec(size_t n) { std::vector<T> v(n); if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } return v; }
};
struct Writer {
    FILE *f;
    explicit Writer(const char *p) : f(fopen(p, "wb")) {}
    ~Writer() { if (f) fclose(f); }
    template <typename T> void vec(const std::vector<T> &v) { if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f); }
    void i32(int32_t v) { fwrite(&v, 4, 1, f); }
};

static cv::Mat mat44(const float *p)
{
    cv::Mat m(4, 4, CV_32F);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m.at<float>(i, j) = p[4 * i + j];
    return m;
}

// in:  int32[8] {nKF, nMP, nE, index of pKF, initKFid, nCov, inertial, abort}; float[5] fx fy cx cy bf; int32[nKF] mnId;
//      float[nKF*16] Tcw; int32[nCov] covisible keyframes of pKF, best first; float[nMP*3] world positions; int32[nE] edge keyframe,
//      int32[nE] edge map point, float[nE*3] (u, v, uRight or -1), int32[nE] octave; float[8] mvInvLevelSigma2
// out: int32 num_fixedKF; float[nKF*16] Tcw; float[nMP*3]; int32 nErased; int32[nErased*2] (keyframe, map point) of every erased
//      observation; int32 map change index; int32 total UpdateNormalAndDepth calls
static int lba_smoke(const char *in, const char *out)
{
    Reader r(in);
    if (!r.f) { fprintf(stderr, "cannot open %s\n", in); return 2; }
    const std::vector<int32_t> hd = r.vec<int32_t>(8);
    const int nKF = hd[0], nMP = hd[1], nE = hd[2], cur = hd[3], nCov = hd[5];
    const std::vector<float> cam = r.vec<float>(5);
    const std::vector<int32_t> ids = r.vec<int32_t>(nKF);
    const std::vector<float> Tcw = r.vec<float>((size_t)nKF * 16);
    const std::vector<int32_t> cov = r.vec<int32_t>(nCov);
    const std::vector<float> X = r.vec<float>((size_t)nMP * 3);
    const std::vector<int32_t> eKF = r.vec<int32_t>(nE), eMP = r.vec<int32_t>(nE);
    const std::vector<float> eObs = r.vec<float>((size_t)nE * 3);
    const std::vector<int32_t> eOct = r.vec<int32_t>(nE);
    const std::vector<float> invS2 = r.vec<float>(8);

    Map map;
    map.mnInitKFid = hd[4]; map.mbIsInertial = hd[6] != 0;
    GeometricCamera camera({cam[0], cam[1], cam[2], cam[3]}, 0);
    std::vector<std::unique_ptr<KeyFrame>> kfs;
    std::vector<std::unique_ptr<MapPoint>> mps;
    for (int i = 0; i < nKF; i++) {
        kfs.emplace_back(new KeyFrame(ids[i], &map, cam[0], cam[1], cam[2], cam[3], cam[4], &camera));
        kfs[i]->SetPose(mat44(&Tcw[(size_t)16 * i]));
        kfs[i]->mvInvLevelSigma2 = invS2;
    }
    for (int l = 0; l < nMP; l++) {
        cv::Mat P(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) P.at<float>(k) = X[(size_t)3 * l + k];
        mps.emplace_back(new MapPoint(1000 + l, P, &map));
    }
    for (int e = 0; e < nE; e++) {                              // one keypoint per observation (KeyFrame::AddMapPoint + MapPoint::AddObservation)
        KeyFrame *kf = kfs[eKF[e]].get();
        cv::KeyPoint kp; kp.pt.x = eObs[3 * e]; kp.pt.y = eObs[3 * e + 1]; kp.octave = eOct[e];
        const int idx = kf->mvKeysUn.size();
        kf->mvKeysUn.push_back(kp); kf->mvuRight.push_back(eObs[3 * e + 2]); kf->mvpMapPoints.push_back(mps[eMP[e]].get());
        mps[eMP[e]]->AddObservation(kf, idx);
    }
    for (int c : cov) kfs[cur]->mvpOrderedConnectedKeyFrames.push_back(kfs[c].get());
    bool stop = hd[7] != 0;
    int num_fixed = -1;
    Optimizer::LocalBundleAdjustment(kfs[cur].get(), &stop, &map, num_fixed);          // LocalMapping.cc:154 call shape

    Writer w(out);
    w.i32(num_fixed);
    std::vector<float> To((size_t)nKF * 16), Xo((size_t)nMP * 3);
    for (int i = 0; i < nKF; i++) { const cv::Mat T = kfs[i]->GetPose(); for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) To[(size_t)16 * i + 4 * a + b] = T.at<float>(a, b); }
    int updates = 0;
    for (int l = 0; l < nMP; l++) { const cv::Mat P = mps[l]->GetWorldPos(); for (int k = 0; k < 3; k++) Xo[(size_t)3 * l + k] = P.at<float>(k); updates += mps[l]->nNormalUpdates; }
    w.vec(To); w.vec(Xo);
    std::vector<int32_t> erased;
    for (int e = 0; e < nE; e++) {
        KeyFrame *kf = kfs[eKF[e]].get();
        if (mps[eMP[e]]->mObservations.count(kf) == 0) { erased.push_back(eKF[e]); erased.push_back(eMP[e]); }
    }
    w.i32((int32_t)erased.size() / 2); w.vec(erased);
    w.i32(map.mnMapChange); w.i32(updates);
    printf("HOST_LBA_OK fixed=%d erased=%zu\n", num_fixed, erased.size() / 2);
    return 0;
}

int match_smoke(const char *in, const char *out);            // host_match_smoke.cc

int main(int argc, char **argv)
{
    if (argc == 4 && std::string(argv[1]) == "lba") return lba_smoke(argv[2], argv[3]);
    if (argc == 4 && std::string(argv[1]) == "match") return match_smoke(argv[2], argv[3]);
    return extractor_smoke();
}
