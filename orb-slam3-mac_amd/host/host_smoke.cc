// host_smoke.cc -- exercises the signature-preserving C++ classes end to end (needs a GPU to run; compiles anywhere).
// Built by the package Makefile, run by tests/test_gpu_host_cpp.py:
//   host_smoke                      ORBextractor / ORBmatcher::DescriptorDistance call shapes of Frame.cc
//   host_smoke lba  <in> <out>      builds a KeyFrame / MapPoint / Map pointer graph from a flat description, runs
//                                   Optimizer::LocalBundleAdjustment on it and dumps the resulting map (the test compares it
//                                   with the CPU oracle run on the same window)
//   host_smoke match <in> <out>     builds Frames + MapPoints, runs the ORBmatcher methods and dumps their results
//   host_smoke latency [reps]       wall-clock time per call of the classes themselves (host_latency.cc), one JSON object on stdout
//   host_smoke liba <in> <out>      builds a temporal chain of inertial KeyFrames (+ fixed visual ones), runs Optimizer::LocalInertialBA
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
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
    bool more() { const int ch = fgetc(f); if (ch == EOF) return false; ungetc(ch, f); return true; }        // an optional trailer follows
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
    // optional trailer: int32 nCam; float[nCam*5] fx fy cx cy bf; int32[nKF] camera of every keyframe (an Atlas window from several cameras)
    std::vector<float> camTab(cam);
    std::vector<int32_t> kfCam(nKF, 0);
    if (r.more()) { const int nCam = r.vec<int32_t>(1)[0]; camTab = r.vec<float>((size_t)nCam * 5); kfCam = r.vec<int32_t>(nKF); }

    Map map;
    map.mnInitKFid = hd[4]; map.mbIsInertial = hd[6] != 0;
    std::vector<std::unique_ptr<GeometricCamera>> cameras;
    for (size_t c = 0; c < camTab.size() / 5; c++) cameras.emplace_back(new GeometricCamera({camTab[5 * c], camTab[5 * c + 1], camTab[5 * c + 2], camTab[5 * c + 3]}, 0));
    std::vector<std::unique_ptr<KeyFrame>> kfs;
    std::vector<std::unique_ptr<MapPoint>> mps;
    for (int i = 0; i < nKF; i++) {
        const float *kc = &camTab[(size_t)5 * kfCam[i]];
        kfs.emplace_back(new KeyFrame(ids[i], &map, kc[0], kc[1], kc[2], kc[3], kc[4], cameras[kfCam[i]].get()));
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

// in:  int32[8] {nKF, nMP, nE, index of pKF, KeyFramesInMap, bLarge, bRecInit, 0}; float[5] fx fy cx cy bf; float[16] Tcb;
//      int32[nKF] mnId; float[nKF*16] Tcw; int32[nKF] index of mPrevKF or -1; int32[nKF] bImu; float[nKF*3] velocity;
//      float[nKF*6] bias (bax bay baz bwx bwy bwz); int32[nKF] has a preintegration; float[nKF*292] {dT, C 15x15, dR, dV, dP, JRg,
//      JVg, JVa, JPg, JPa, b (bax..bwz)}; float[nMP*3] positions; float[nMP] mTrackDepth; int32[nE] edge keyframe, int32[nE] edge
//      map point, float[nE*3] (u, v, uRight or -1), int32[nE] octave; float[8] mvInvLevelSigma2;
//      when header[7] != 0 (two KannalaBrandt8 cameras): float[12] mTrl, float[4] camera 2 fx fy cx cy, float[4] k (camera 1), float[4] k
//      (camera 2), int32[nE] 1 = the observation was made in the right camera
// out: float[nKF*16] Tcw; float[nKF*3] velocity; float[nKF*6] bias; float[nMP*3]; int32 nErased; int32[nErased*3] (keyframe, map
//      point, camera); int32 map change index
static int liba_smoke(const char *in, const char *out)
{
    Reader r(in);
    if (!r.f) { fprintf(stderr, "cannot open %s\n", in); return 2; }
    const std::vector<int32_t> hd = r.vec<int32_t>(8);
    const int nKF = hd[0], nMP = hd[1], nE = hd[2], cur = hd[3];
    const std::vector<float> cam = r.vec<float>(5), Tcb = r.vec<float>(16);
    const std::vector<int32_t> ids = r.vec<int32_t>(nKF);
    const std::vector<float> Tcw = r.vec<float>((size_t)nKF * 16);
    const std::vector<int32_t> prev = r.vec<int32_t>(nKF), bimu = r.vec<int32_t>(nKF);
    const std::vector<float> vel = r.vec<float>((size_t)nKF * 3), bias = r.vec<float>((size_t)nKF * 6);
    const std::vector<int32_t> hasp = r.vec<int32_t>(nKF);
    const std::vector<float> pre = r.vec<float>((size_t)nKF * 292);
    const std::vector<float> X = r.vec<float>((size_t)nMP * 3), depth = r.vec<float>(nMP);
    const std::vector<int32_t> eKF = r.vec<int32_t>(nE), eMP = r.vec<int32_t>(nE);
    const std::vector<float> eObs = r.vec<float>((size_t)nE * 3);
    const std::vector<int32_t> eOct = r.vec<int32_t>(nE);
    const std::vector<float> invS2 = r.vec<float>(8);
    const bool rig = hd[7] != 0;
    std::vector<float> Trl, cam2v, k1, k2;
    std::vector<int32_t> eRight(nE, 0);
    if (rig) { Trl = r.vec<float>(12); cam2v = r.vec<float>(4); k1 = r.vec<float>(4); k2 = r.vec<float>(4); eRight = r.vec<int32_t>(nE); }

    Map map;
    map.mbIsInertial = true; map.nKeyFrames = hd[4];
    GeometricCamera camera(rig ? std::vector<float>{cam[0], cam[1], cam[2], cam[3], k1[0], k1[1], k1[2], k1[3]} : std::vector<float>{cam[0], cam[1], cam[2], cam[3]}, rig ? 1 : 0);
    GeometricCamera camera2(rig ? std::vector<float>{cam2v[0], cam2v[1], cam2v[2], cam2v[3], k2[0], k2[1], k2[2], k2[3]} : std::vector<float>{0, 0, 0, 0}, 1);
    std::vector<std::unique_ptr<KeyFrame>> kfs;
    std::vector<std::unique_ptr<MapPoint>> mps;
    std::vector<std::unique_ptr<IMU::Preintegrated>> pints;
    auto mat = [](const float *p, int rows, int cols) { cv::Mat m(rows, cols, CV_32F); for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) m.at<float>(i, j) = p[cols * i + j]; return m; };
    for (int i = 0; i < nKF; i++) {
        kfs.emplace_back(new KeyFrame(ids[i], &map, cam[0], cam[1], cam[2], cam[3], cam[4], &camera));
        KeyFrame *k = kfs[i].get();
        k->mImuCalib.Tcb = mat(Tcb.data(), 4, 4);
        k->SetPose(mat(&Tcw[(size_t)16 * i], 4, 4));
        k->mvInvLevelSigma2 = invS2;
        k->bImu = bimu[i] != 0;
        if (rig) { k->mpCamera2 = &camera2; k->mTrl = mat(Trl.data(), 3, 4); }
        k->SetVelocity(mat(&vel[(size_t)3 * i], 3, 1));
        const float *b = &bias[(size_t)6 * i];
        k->SetNewBias(IMU::Bias(b[0], b[1], b[2], b[3], b[4], b[5]));
    }
    for (int i = 0; i < nKF; i++) {
        if (prev[i] >= 0) { kfs[i]->mPrevKF = kfs[prev[i]].get(); kfs[prev[i]]->mNextKF = kfs[i].get(); }
        if (hasp[i]) {
            const float *p = &pre[(size_t)292 * i];
            pints.emplace_back(new IMU::Preintegrated());
            IMU::Preintegrated *q = pints.back().get();
            q->dT = p[0]; q->C = mat(p + 1, 15, 15); q->dR = mat(p + 226, 3, 3); q->dV = mat(p + 235, 3, 1); q->dP = mat(p + 238, 3, 1);
            q->JRg = mat(p + 241, 3, 3); q->JVg = mat(p + 250, 3, 3); q->JVa = mat(p + 259, 3, 3); q->JPg = mat(p + 268, 3, 3); q->JPa = mat(p + 277, 3, 3);
            q->b = IMU::Bias(p[286], p[287], p[288], p[289], p[290], p[291]);
            kfs[i]->mpImuPreintegrated = q;
        }
    }
    for (int l = 0; l < nMP; l++) {
        mps.emplace_back(new MapPoint(1000 + l, mat(&X[(size_t)3 * l], 3, 1), &map));
        mps[l]->mTrackDepth = depth[l];
    }
    {
        // left keypoints first (indices 0 .. NLeft-1), then the right camera's (NLeft ..), as Frame lays a rig frame out
        std::vector<int> leftIdx(nE, -1), rightIdx(nE, -1);
        for (int e = 0; e < nE; e++) if (!eRight[e]) {
            KeyFrame *kf = kfs[eKF[e]].get();
            cv::KeyPoint kp; kp.pt.x = eObs[3 * e]; kp.pt.y = eObs[3 * e + 1]; kp.octave = eOct[e];
            leftIdx[e] = kf->mvKeysUn.size();
            kf->mvKeysUn.push_back(kp); kf->mvuRight.push_back(eObs[3 * e + 2]); kf->mvpMapPoints.push_back(mps[eMP[e]].get());
        }
        if (rig) for (int i = 0; i < nKF; i++) kfs[i]->NLeft = kfs[i]->mvKeysUn.size();
        for (int e = 0; e < nE; e++) if (eRight[e]) {
            KeyFrame *kf = kfs[eKF[e]].get();
            cv::KeyPoint kp; kp.pt.x = eObs[3 * e]; kp.pt.y = eObs[3 * e + 1]; kp.octave = eOct[e];
            rightIdx[e] = kf->NLeft + (int)kf->mvKeysRight.size();
            kf->mvKeysRight.push_back(kp); kf->mvpMapPoints.push_back(mps[eMP[e]].get());
        }
        std::map<std::pair<int, int>, std::pair<int, int>> both;              // (keyframe, map point) -> (left index, right index)
        for (int e = 0; e < nE; e++) {
            auto &b = both.emplace(std::make_pair(eKF[e], eMP[e]), std::make_pair(-1, -1)).first->second;
            if (eRight[e]) b.second = rightIdx[e]; else b.first = leftIdx[e];
        }
        for (auto &kv : both) mps[kv.first.second]->AddObservation(kfs[kv.first.first].get(), kv.second.first, kv.second.second);
    }
    bool stop = false;
    Optimizer::LocalInertialBA(kfs[cur].get(), &stop, &map, hd[5] != 0, hd[6] != 0);      // LocalMapping.cc:131-155 call shape

    Writer w(out);
    std::vector<float> To((size_t)nKF * 16), Vo((size_t)nKF * 3), Bo((size_t)nKF * 6), Xo((size_t)nMP * 3);
    for (int i = 0; i < nKF; i++) {
        const cv::Mat T = kfs[i]->GetPose(), V = kfs[i]->GetVelocity();
        for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) To[(size_t)16 * i + 4 * a + b] = T.at<float>(a, b);
        for (int a = 0; a < 3; a++) Vo[(size_t)3 * i + a] = V.at<float>(a);
        const IMU::Bias bb = kfs[i]->GetImuBias();
        const float bv[6] = {bb.bax, bb.bay, bb.baz, bb.bwx, bb.bwy, bb.bwz};
        for (int a = 0; a < 6; a++) Bo[(size_t)6 * i + a] = bv[a];
    }
    for (int l = 0; l < nMP; l++) { const cv::Mat P = mps[l]->GetWorldPos(); for (int k = 0; k < 3; k++) Xo[(size_t)3 * l + k] = P.at<float>(k); }
    w.vec(To); w.vec(Vo); w.vec(Bo); w.vec(Xo);
    std::vector<int32_t> erased;
    for (int e = 0; e < nE; e++) {
        KeyFrame *kf = kfs[eKF[e]].get();
        if (mps[eMP[e]]->mObservations.count(kf) == 0) { erased.push_back(eKF[e]); erased.push_back(eMP[e]); erased.push_back(eRight[e]); }
    }
    w.i32((int32_t)erased.size() / 3); w.vec(erased);
    w.i32(map.mnMapChange);
    printf("HOST_LIBA_OK erased=%zu\n", erased.size() / 3);
    return 0;
}

// in:  int32[4] {nK, nF, mbCheckOrientation, 0}; float mfNNratio; cv::KeyPoint[nK]; uint8[nK*32]; int32[nK] vocabulary node of every
//      keyframe feature; int32[nK] 1 = has a good map point, 0 = none, 2 = a bad one; cv::KeyPoint[nF]; uint8[nF*32]; int32[nF] node
// out: int32 return value; int32[nF] index of the keyframe feature whose map point the frame feature received, or -1
static int bow_smoke(const char *in, const char *out)
{
    Reader r(in);
    if (!r.f) { fprintf(stderr, "cannot open %s\n", in); return 2; }
    const std::vector<int32_t> hd = r.vec<int32_t>(4);
    const int nK = hd[0], nF = hd[1];
    const float ratio = r.vec<float>(1)[0];
    const std::vector<cv::KeyPoint> kk = r.vec<cv::KeyPoint>(nK);
    const std::vector<uint8_t> dk = r.vec<uint8_t>((size_t)nK * 32);
    const std::vector<int32_t> nodeK = r.vec<int32_t>(nK), mpK = r.vec<int32_t>(nK);
    const std::vector<cv::KeyPoint> kf = r.vec<cv::KeyPoint>(nF);
    const std::vector<uint8_t> df = r.vec<uint8_t>((size_t)nF * 32);
    const std::vector<int32_t> nodeF = r.vec<int32_t>(nF);
    Map map;
    GeometricCamera camera({500.f, 500.f, 320.f, 240.f}, 0);
    KeyFrame K(1, &map, 500.f, 500.f, 320.f, 240.f, 40.f, &camera);
    K.mvKeysUn = kk;
    K.mDescriptors = cv::Mat(nK, 32, CV_8U);
    if (nK) memcpy(K.mDescriptors.data, dk.data(), dk.size());
    std::vector<std::unique_ptr<MapPoint>> pool;
    cv::Mat zero(3, 1, CV_32F);
    for (int i = 0; i < nK; i++) {
        K.mFeatVec.addFeature((DBoW2::NodeId)nodeK[i], (unsigned)i);           // DBoW2 fills it in feature order (TemplatedVocabulary.h:1218-1260)
        MapPoint *p = nullptr;
        if (mpK[i]) { pool.emplace_back(new MapPoint(100 + i, zero, &map)); p = pool.back().get(); p->mbBad = mpK[i] == 2; }
        K.mvpMapPoints.push_back(p);
    }
    Frame F;
    F.N = nF; F.mvKeys = kf; F.mvKeysUn = kf;
    F.mDescriptors = cv::Mat(nF, 32, CV_8U);
    if (nF) memcpy(F.mDescriptors.data, df.data(), df.size());
    for (int i = 0; i < nF; i++) F.mFeatVec.addFeature((DBoW2::NodeId)nodeF[i], (unsigned)i);
    ORBmatcher matcher(ratio, hd[2] != 0);
    std::vector<MapPoint *> vpMapPointMatches;
    const int n = matcher.SearchByBoW(&K, F, vpMapPointMatches);                // Tracking.cc:1757 call shape
    Writer w(out);
    w.i32(n);
    std::vector<int32_t> res(nF, -1);
    for (int j = 0; j < nF && j < (int)vpMapPointMatches.size(); j++)
        if (vpMapPointMatches[j]) res[j] = (int32_t)vpMapPointMatches[j]->mnId - 100;
    w.vec(res);
    printf("HOST_BOW_OK %d matches\n", n);
    return (int)vpMapPointMatches.size() == nF ? 0 : 1;
}

int match_smoke(const char *in, const char *out);            // host_match_smoke.cc

int kf_smoke(const char *in, const char *out);
int poseopt_smoke(const char *in, const char *out);
int mergeba_smoke(const char *in, const char *out);
int stereo_smoke(const char *in, const char *out);
int latency_main(int reps);                                   // host_latency.cc
int cachecheck_main();

int main(int argc, char **argv)
{
    if (argc >= 2 && std::string(argv[1]) == "cachecheck") return cachecheck_main();
    if (argc >= 2 && std::string(argv[1]) == "latency") return latency_main(argc >= 3 ? atoi(argv[2]) : 20);
    if (argc == 4 && std::string(argv[1]) == "lba") return lba_smoke(argv[2], argv[3]);
    if (argc == 4 && std::string(argv[1]) == "match") return match_smoke(argv[2], argv[3]);
    if (argc == 4 && std::string(argv[1]) == "liba") return liba_smoke(argv[2], argv[3]);
    if (argc == 4 && std::string(argv[1]) == "bow") return bow_smoke(argv[2], argv[3]);
    if (argc == 4 && std::string(argv[1]) == "kfmatch") return kf_smoke(argv[2], argv[3]);
    if (argc == 4 && std::string(argv[1]) == "poseopt") return poseopt_smoke(argv[2], argv[3]);
    if (argc == 4 && std::string(argv[1]) == "mergeba") return mergeba_smoke(argv[2], argv[3]);
    if (argc == 4 && std::string(argv[1]) == "stereo") return stereo_smoke(argv[2], argv[3]);
    return extractor_smoke();
}
