// host_kf_smoke.cc -- `host_smoke kfmatch <in> <out>`, `host_smoke poseopt <in> <out>`, `host_smoke mergeba <in> <out>`: builds KeyFrames /
// Frames / MapPoints from named flat arrays (flatfile.h), calls the keyframe-side ORBmatcher methods, Optimizer::PoseOptimization and the
// map-merge Optimizer::LocalBundleAdjustment with the reference's signatures and dumps what they did to the pointer graph;
// tests/test_gpu_host_cpp.py compares it with an independent Python model of the host geometry on top of the matcher / BA oracles.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <set>
#include <thread>
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "Optimizer.h"
#include "flatfile.h"

using namespace ORB_SLAM3;

namespace {

cv::Mat mat_from(const std::vector<float> &v, int rows, int cols)
{
    cv::Mat m(rows, cols, CV_32F);
    for (int i = 0; i < rows * cols; i++) m.at<float>(i / cols, i % cols) = v[i];
    return m;
}
cv::Mat desc_rows(const std::vector<uint8_t> &d, int n)
{
    cv::Mat m(n > 0 ? n : 1, 32, CV_8U);
    if (n) memcpy(m.data, d.data(), (size_t)n * 32);
    return m;
}
std::vector<cv::KeyPoint> keypoints(const std::vector<uint8_t> &raw)
{
    std::vector<cv::KeyPoint> k(raw.size() / sizeof(cv::KeyPoint));
    if (!k.empty()) memcpy((void *)k.data(), raw.data(), k.size() * sizeof(cv::KeyPoint));
    return k;
}

struct World {
    Map map;
    std::unique_ptr<GeometricCamera> camL, camR;
    std::vector<std::unique_ptr<MapPoint>> pts;
    std::unique_ptr<KeyFrame> kf[2];
    std::map<MapPoint *, int> index;
    int idx(MapPoint *p) const { if (!p) return -1; auto it = index.find(p); return it == index.end() ? -2 : it->second; }
};

std::unique_ptr<World> build(const FlatFile &S)
{
    std::unique_ptr<World> W(new World());
    const std::vector<float> &c1 = S.F("cam1"), &c2 = S.F("cam2");
    const int t1 = S.I("cam_type")[0], rig = S.I("rig")[0];
    W->camL.reset(new GeometricCamera(t1 ? std::vector<float>(c1.begin(), c1.begin() + 8) : std::vector<float>(c1.begin(), c1.begin() + 4), t1));
    W->camR.reset(new GeometricCamera(t1 ? std::vector<float>(c2.begin(), c2.begin() + 8) : std::vector<float>(c2.begin(), c2.begin() + 4), t1));
    const std::vector<float> &X = S.F("pt_X"), &Nn = S.F("pt_n"), &D = S.F("pt_dist");
    const std::vector<uint8_t> &pd = S.U("pt_desc");
    const std::vector<int32_t> &pobs = S.I("pt_obs"), &pbad = S.I("pt_bad");
    const int M = pobs.size();
    for (int l = 0; l < M; l++) {
        cv::Mat P(3, 1, CV_32F), n(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) { P.at<float>(k) = X[3 * l + k]; n.at<float>(k) = Nn[3 * l + k]; }
        W->pts.emplace_back(new MapPoint(100 + l, P, &W->map));
        MapPoint *p = W->pts.back().get();
        p->mNormalVector = n; p->mfMinDistance = D[2 * l]; p->mfMaxDistance = D[2 * l + 1];
        p->mDescriptor = cv::Mat(1, 32, CV_8U); memcpy(p->mDescriptor.data, &pd[(size_t)32 * l], 32);
        p->nObs = pobs[l]; p->mbBad = pbad[l] != 0;
        W->index[p] = l;
    }
    const std::vector<float> &b = S.F("bounds");
    for (int k = 0; k < 2; k++) {
        char nm[32];
        auto key = [&](const char *suffix) { snprintf(nm, sizeof(nm), "kf%d_%s", k + 1, suffix); return nm; };
        KeyFrame *kf = new KeyFrame(10 + k, &W->map, c1[0], c1[1], c1[2], c1[3], S.F("mbf")[0], W->camL.get());
        W->kf[k].reset(kf);
        kf->SetPose(mat_from(S.F(key("Tcw")), 4, 4));
        const std::vector<cv::KeyPoint> kp = keypoints(S.U(key("kp")));
        const int n = kp.size();
        kf->N = n;
        const int nleft = S.I(key("nleft"))[0];
        if (rig) {
            kf->NLeft = nleft; kf->mpCamera2 = W->camR.get();
            kf->mvKeys.assign(kp.begin(), kp.begin() + nleft); kf->mvKeysRight.assign(kp.begin() + nleft, kp.end());
            kf->mvKeysUn = kf->mvKeys;                                       // Frame::UndistortKeyPoints with zero distortion (Frame.cc:740-744)
            kf->mTlr = mat_from(S.F("Tlr"), 3, 4); kf->mTrl = mat_from(S.F("Trl"), 3, 4);
        } else { kf->mvKeys = kp; kf->mvKeysUn = kp; }
        kf->mDescriptors = desc_rows(S.U(key("desc")), n);
        kf->mvuRight = S.F(key("ur"));
        kf->mvScaleFactors = S.F("scale"); kf->mvLevelSigma2 = S.F("sigma2"); kf->mvInvLevelSigma2 = S.F("invsigma2");
        kf->mnScaleLevels = 8; kf->mfLogScaleFactor = std::log(1.2f);
        kf->mnMinX = (int)b[0]; kf->mnMinY = (int)b[1]; kf->mnMaxX = (int)b[2]; kf->mnMaxY = (int)b[3];
        kf->mvpMapPoints.assign(n, nullptr);
        const std::vector<int32_t> &mp = S.I(key("mp")), &nid = S.I(key("nid"));
        for (int i = 0; i < n; i++) {
            if (mp[i] >= 0) {
                MapPoint *p = W->pts[mp[i]].get();
                kf->mvpMapPoints[i] = p;
                const int before = p->nObs;
                p->AddObservation(kf, i);
                p->nObs = before;                                            // the scenario fixes Observations(); only the membership is wanted
            }
            if (nid[i] >= 0) kf->mFeatVec.addFeature((unsigned)nid[i], (unsigned)i);
        }
    }
    return W;
}

std::vector<MapPoint *> list_of(const World &W, const std::vector<int32_t> &ids)
{
    std::vector<MapPoint *> v(ids.size(), nullptr);
    for (size_t i = 0; i < ids.size(); i++) if (ids[i] >= 0) v[i] = W.pts[ids[i]].get();
    return v;
}
std::vector<int32_t> kf_points(const World &W, KeyFrame *kf)
{
    std::vector<int32_t> r(kf->mvpMapPoints.size());
    for (size_t i = 0; i < r.size(); i++) r[i] = W.idx(kf->mvpMapPoints[i]);
    return r;
}

}  // namespace

int kf_smoke(const char *in, const char *out)
{
    FlatFile S;
    if (!S.load(in)) { fprintf(stderr, "cannot read %s\n", in); return 2; }
    FlatWriter O(out);
    if (!O.fp) return 2;
    const std::vector<float> &b = S.F("bounds");
    Frame::mnMinX = b[0]; Frame::mnMinY = b[1]; Frame::mnMaxX = b[2]; Frame::mnMaxY = b[3];
    const int rig = S.I("rig")[0];
    {   // ---- Fuse(pKF, vpMapPoints, th, bRight): LocalMapping::SearchInNeighbors
        for (int right = 0; right <= (rig ? 1 : 0); right++) {
            std::unique_ptr<World> W = build(S);
            KeyFrame *kf = W->kf[1].get();
            const std::vector<MapPoint *> cand = list_of(*W, S.I("fuse_list"));
            ORBmatcher matcher;
            const int n = right ? matcher.Fuse(kf, cand, S.F("th_fuse")[0], true) : matcher.Fuse(kf, cand, S.F("th_fuse")[0]);
            std::vector<int32_t> repl(W->pts.size(), -1), nobs(W->pts.size(), 0);
            for (size_t l = 0; l < W->pts.size(); l++) { repl[l] = W->idx(W->pts[l]->GetReplaced()); nobs[l] = W->pts[l]->Observations(); }
            O.one(right ? "fuser_n" : "fuse_n", n); O.ints(right ? "fuser_kfmp" : "fuse_kfmp", kf_points(*W, kf));
            O.ints(right ? "fuser_replaced" : "fuse_replaced", repl); O.ints(right ? "fuser_nobs" : "fuse_nobs", nobs);
        }
    }
    {   // ---- Fuse(pKF, Scw, vpPoints, th, vpReplacePoint): LoopClosing::SearchAndFuse
        std::unique_ptr<World> W = build(S);
        KeyFrame *kf = W->kf[1].get();
        const std::vector<MapPoint *> cand = list_of(*W, S.I("sim3_list"));
        std::vector<MapPoint *> vpReplacePoints(cand.size(), static_cast<MapPoint *>(NULL));
        ORBmatcher matcher(0.8);
        const int n = matcher.Fuse(kf, mat_from(S.F("Scw"), 4, 4), cand, 4, vpReplacePoints);
        std::vector<int32_t> rp(cand.size());
        for (size_t i = 0; i < cand.size(); i++) rp[i] = W->idx(vpReplacePoints[i]);
        O.one("fs_n", n); O.ints("fs_replace", rp); O.ints("fs_kfmp", kf_points(*W, kf));
    }
    {   // ---- SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming) and the overload with keyframes: LoopClosing
        for (int variant = 0; variant < 2; variant++) {
            std::unique_ptr<World> W = build(S);
            KeyFrame *kf = W->kf[1].get();
            const std::vector<MapPoint *> cand = list_of(*W, S.I("sim3_list"));
            std::vector<MapPoint *> vpMatched = kf->GetMapPointMatches();
            for (size_t i = 0; i < vpMatched.size(); i++) if (i % 3) vpMatched[i] = nullptr;        // a partially filled vpMatched, as after SearchByBoW
            ORBmatcher matcher(0.9, true);
            int n;
            std::vector<int32_t> who(vpMatched.size(), -1);
            if (variant == 0) n = matcher.SearchByProjection(kf, mat_from(S.F("Scw"), 4, 4), cand, vpMatched, 3, 1.5);
            else {
                std::vector<KeyFrame *> vpPointsKFs(cand.size());
                for (size_t i = 0; i < cand.size(); i++) vpPointsKFs[i] = W->kf[i & 1].get();
                std::vector<KeyFrame *> vpMatchedKF(vpMatched.size(), static_cast<KeyFrame *>(NULL));
                n = matcher.SearchByProjection(kf, mat_from(S.F("Scw"), 4, 4), cand, vpPointsKFs, vpMatched, vpMatchedKF, 8, 1.5);
                for (size_t i = 0; i < who.size(); i++) who[i] = !vpMatchedKF[i] ? -1 : (vpMatchedKF[i] == W->kf[0].get() ? 0 : 1);
            }
            std::vector<int32_t> m(vpMatched.size());
            for (size_t i = 0; i < m.size(); i++) m[i] = W->idx(vpMatched[i]);
            O.one(variant ? "spk_n" : "sp_n", n); O.ints(variant ? "spk_matched" : "sp_matched", m);
            if (variant) O.ints("spk_kf", who);
        }
    }
    {   // ---- SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)
        std::unique_ptr<World> W = build(S);
        std::vector<MapPoint *> vpMatches12(W->kf[0]->mvpMapPoints.size(), static_cast<MapPoint *>(NULL));
        const std::vector<int32_t> &pre = S.I("s3_pre");
        for (size_t i = 0; i < vpMatches12.size(); i++) if (pre[i] >= 0) vpMatches12[i] = W->pts[pre[i]].get();
        ORBmatcher matcher(0.75, true);
        const float s12 = S.F("s12")[0];
        const int n = matcher.SearchBySim3(W->kf[0].get(), W->kf[1].get(), vpMatches12, s12, mat_from(S.F("R12"), 3, 3), mat_from(S.F("t12"), 3, 1), 7.5);
        std::vector<int32_t> m(vpMatches12.size());
        for (size_t i = 0; i < m.size(); i++) m[i] = W->idx(vpMatches12[i]);
        O.one("s3_n", n); O.ints("s3_matches", m);
    }
    {   // ---- SearchByBoW(pKF1, pKF2, vpMatches12): LoopClosing
        std::unique_ptr<World> W = build(S);
        std::vector<MapPoint *> vpMatches12;
        ORBmatcher matcher(0.9, true);
        const int n = matcher.SearchByBoW(W->kf[0].get(), W->kf[1].get(), vpMatches12);
        std::vector<int32_t> m(vpMatches12.size());
        for (size_t i = 0; i < m.size(); i++) m[i] = W->idx(vpMatches12[i]);
        O.one("bow_n", n); O.ints("bow_matches", m);
    }
    {   // ---- SearchForTriangulation(pKF1, pKF2, F12, vMatchedIndices, false, bCoarse): LocalMapping::CreateNewMapPoints
        for (int coarse = 0; coarse < 2; coarse++) {
            std::unique_ptr<World> W = build(S);
            std::vector<std::pair<size_t, size_t>> vMatchedIndices;
            ORBmatcher matcher(0.6, false);
            const int n = matcher.SearchForTriangulation(W->kf[0].get(), W->kf[1].get(), mat_from(S.F("F12"), 3, 3), vMatchedIndices, false, coarse != 0);
            std::vector<int32_t> pr;
            for (auto &p : vMatchedIndices) { pr.push_back((int32_t)p.first); pr.push_back((int32_t)p.second); }
            O.one(coarse ? "tric_n" : "tri_n", n); O.ints(coarse ? "tric_pairs" : "tri_pairs", pr);
        }
        std::unique_ptr<World> W = build(S);                                 // with the rotation check
        std::vector<std::pair<size_t, size_t>> vMatchedIndices;
        ORBmatcher matcher(0.6, true);
        const int n = matcher.SearchForTriangulation(W->kf[0].get(), W->kf[1].get(), mat_from(S.F("F12"), 3, 3), vMatchedIndices, false);
        std::vector<int32_t> pr;
        for (auto &p : vMatchedIndices) { pr.push_back((int32_t)p.first); pr.push_back((int32_t)p.second); }
        O.one("trio_n", n); O.ints("trio_pairs", pr);
    }
    {   // ---- SearchForTriangulation(pKF1, pKF2, F12, vMatchedIndices, bOnlyStereo, vMatchedPoints): the overload that returns the points
        for (int ori = 0; ori < 2; ori++) {
            std::unique_ptr<World> W = build(S);
            std::vector<std::pair<size_t, size_t>> vMatchedIndices;
            std::vector<cv::Mat> vMatchedPoints;
            ORBmatcher matcher(0.6, ori != 0);
            const int n = matcher.SearchForTriangulation(W->kf[0].get(), W->kf[1].get(), mat_from(S.F("F12"), 3, 3), vMatchedIndices, true, vMatchedPoints);
            std::vector<int32_t> pr;
            std::vector<float> pt;
            for (auto &p : vMatchedIndices) { pr.push_back((int32_t)p.first); pr.push_back((int32_t)p.second); }
            for (auto &x : vMatchedPoints) for (int k = 0; k < 3; k++) pt.push_back(x.at<float>(k));
            O.one(ori ? "trpo_n" : "trp_n", n); O.ints(ori ? "trpo_pairs" : "trp_pairs", pr); O.floats(ori ? "trpo_points" : "trp_points", pt);
        }
    }
    if (!rig) {   // ---- SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist): Tracking::Relocalization
        std::unique_ptr<World> W = build(S);
        Frame F;
        const std::vector<cv::KeyPoint> kp = keypoints(S.U("fr_kp"));
        F.N = kp.size(); F.mvKeys = kp; F.mvKeysUn = kp; F.mDescriptors = desc_rows(S.U("fr_desc"), F.N); F.mTcw = mat_from(S.F("fr_Tcw"), 4, 4);
        F.mvScaleFactors = S.F("scale"); F.mpCamera = W->camL.get(); F.mvuRight.assign(F.N, -1.f); F.mvbOutlier.assign(F.N, false);
        F.mvpMapPoints = list_of(*W, S.I("fr_mp"));
        std::set<MapPoint *> sFound;
        for (int32_t l : S.I("already")) sFound.insert(W->pts[l].get());
        ORBmatcher matcher2(0.9, true);
        const int n = matcher2.SearchByProjection(F, W->kf[0].get(), sFound, 10, 100);
        std::vector<int32_t> m(F.N);
        for (int i = 0; i < F.N; i++) m[i] = W->idx(F.mvpMapPoints[i]);
        O.one("rl_n", n); O.ints("rl_mp", m);
    }
    printf("HOST_KF_OK\n");
    return 0;
}

// in: "cam" f[8] + "cam_type", "rig", "cam2" f[8], "Trl" f[12], "nleft"; "Tcw" f[16]; "kp" KeyPoint bytes; "ur" f[n]; "X" f[n*3] (NaN row = no map point);
//     "invsigma2" f[8], "mbf"      out: "n_inliers", "Tcw" f[16], "outlier" i[n]
int poseopt_smoke(const char *in, const char *out)
{
    FlatFile S;
    if (!S.load(in)) { fprintf(stderr, "cannot read %s\n", in); return 2; }
    FlatWriter O(out);
    const std::vector<float> &c1 = S.F("cam"), &c2 = S.F("cam2");
    const int t1 = S.I("cam_type")[0], rig = S.I("rig")[0];
    GeometricCamera camL(t1 ? std::vector<float>(c1.begin(), c1.begin() + 8) : std::vector<float>(c1.begin(), c1.begin() + 4), t1);
    GeometricCamera camR(std::vector<float>(c2.begin(), c2.begin() + 8), 1);
    Map map;
    Frame F;
    Frame::fx = c1[0]; Frame::fy = c1[1]; Frame::cx = c1[2]; Frame::cy = c1[3];
    const std::vector<cv::KeyPoint> kp = keypoints(S.U("kp"));
    const int n = kp.size();
    F.N = n; F.mbf = S.F("mbf")[0]; F.mpCamera = &camL; F.mTcw = mat_from(S.F("Tcw"), 4, 4); F.mvInvLevelSigma2 = S.F("invsigma2");
    F.mvuRight = S.F("ur"); F.mvbOutlier.assign(n, true);                     // stale flags: the function resets them
    if (rig) {
        const int nleft = S.I("nleft")[0];
        F.Nleft = nleft; F.Nright = n - nleft; F.mpCamera2 = &camR; F.mTrl = mat_from(S.F("Trl"), 3, 4);
        F.mvKeys.assign(kp.begin(), kp.begin() + nleft); F.mvKeysRight.assign(kp.begin() + nleft, kp.end());
    } else { F.mvKeys = kp; F.mvKeysUn = kp; }
    const std::vector<float> &X = S.F("X");
    std::vector<std::unique_ptr<MapPoint>> pool;
    F.mvpMapPoints.assign(n, nullptr);
    for (int i = 0; i < n; i++) {
        if (X[3 * i] != X[3 * i]) continue;
        cv::Mat P(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) P.at<float>(k) = X[3 * i + k];
        pool.emplace_back(new MapPoint(i, P, &map));
        F.mvpMapPoints[i] = pool.back().get();
    }
    const int nin = Optimizer::PoseOptimization(&F);
    std::vector<float> T(16);
    for (int i = 0; i < 16; i++) T[i] = F.mTcw.at<float>(i / 4, i % 4);
    std::vector<int32_t> o(n);
    for (int i = 0; i < n; i++) o[i] = F.mvbOutlier[i] ? 1 : 0;
    O.one("n_inliers", nin); O.floats("Tcw", T); O.ints("outlier", o);
    printf("HOST_POSEOPT_OK\n");
    return 0;
}

// in (as `host_smoke lba`, by name): "nKF", "ids" i, "Tcw" f[nKF*16], "fixed" i[nKF] (1 = in vpFixedKF, 0 = vpAdjustKF), "marked" i[nKF] (the caller
//     set mnBALocalForMerge on this adjustable keyframe), "X" f[nMP*3], "eKF" / "eMP" i[nE], "eObs" f[nE*3], "eOct" i[nE], "invsigma2" f[8], "cam" f[5], "abort"
// out: "Tcw" f[nKF*16], "X" f[nMP*3], "erased" i[2k], "updates" (UpdateNormalAndDepth calls), "bad" i[nMP]
int mergeba_smoke(const char *in, const char *out)
{
    FlatFile S;
    if (!S.load(in)) { fprintf(stderr, "cannot read %s\n", in); return 2; }
    FlatWriter O(out);
    const std::vector<float> &cam = S.F("cam");
    const std::vector<int32_t> &ids = S.I("ids"), &fx = S.I("fixed"), &marked = S.I("marked"), &eKF = S.I("eKF"), &eMP = S.I("eMP"), &eOct = S.I("eOct");
    const std::vector<float> &Tcw = S.F("Tcw"), &X = S.F("X"), &eObs = S.F("eObs");
    const int nKF = ids.size(), nMP = X.size() / 3, nE = eKF.size();
    Map map;
    GeometricCamera camera({cam[0], cam[1], cam[2], cam[3]}, 0);
    std::vector<std::unique_ptr<KeyFrame>> kfs;
    std::vector<std::unique_ptr<MapPoint>> mps;
    for (int i = 0; i < nKF; i++) {
        kfs.emplace_back(new KeyFrame(ids[i], &map, cam[0], cam[1], cam[2], cam[3], cam[4], &camera));
        kfs[i]->SetPose(mat_from(std::vector<float>(Tcw.begin() + 16 * i, Tcw.begin() + 16 * i + 16), 4, 4));
        kfs[i]->mvInvLevelSigma2 = S.F("invsigma2");
    }
    for (int l = 0; l < nMP; l++) {
        cv::Mat P(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) P.at<float>(k) = X[3 * l + k];
        mps.emplace_back(new MapPoint(1000 + l, P, &map));
    }
    for (int e = 0; e < nE; e++) {
        KeyFrame *kf = kfs[eKF[e]].get();
        cv::KeyPoint kp; kp.pt.x = eObs[3 * e]; kp.pt.y = eObs[3 * e + 1]; kp.octave = eOct[e];
        const int idx = kf->mvKeysUn.size();
        kf->mvKeysUn.push_back(kp); kf->mvuRight.push_back(eObs[3 * e + 2]); kf->mvpMapPoints.push_back(mps[eMP[e]].get());
        mps[eMP[e]]->AddObservation(kf, idx);
    }
    KeyFrame *pMainKF = kfs[S.I("main")[0]].get();
    std::vector<KeyFrame *> vpAdjustKF, vpFixedKF;
    for (int i = 0; i < nKF; i++) {
        (fx[i] ? vpFixedKF : vpAdjustKF).push_back(kfs[i].get());
        if (marked[i]) kfs[i]->mnBALocalForMerge = pMainKF->mnId;
    }
    bool bStop = S.I("abort")[0] != 0;
    Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, &bStop);
    std::vector<float> To((size_t)nKF * 16), Xo((size_t)nMP * 3);
    for (int i = 0; i < nKF; i++) { const cv::Mat T = kfs[i]->GetPose(); for (int k = 0; k < 16; k++) To[16 * i + k] = T.at<float>(k / 4, k % 4); }
    int32_t updates = 0;
    std::vector<int32_t> bad(nMP);
    for (int l = 0; l < nMP; l++) { const cv::Mat P = mps[l]->GetWorldPos(); for (int k = 0; k < 3; k++) Xo[3 * l + k] = P.at<float>(k); updates += mps[l]->nNormalUpdates; bad[l] = mps[l]->isBad(); }
    std::vector<int32_t> erased;                              // observations that existed in the input and are gone
    for (int e = 0; e < nE; e++) if (!mps[eMP[e]]->IsInKeyFrame(kfs[eKF[e]].get())) { erased.push_back(eKF[e]); erased.push_back(eMP[e]); }
    O.floats("Tcw", To); O.floats("X", Xo); O.ints("erased", erased); O.one("updates", updates); O.ints("bad", bad); O.one("change", map.mnMapChange);
    printf("HOST_MERGEBA_OK\n");
    return 0;
}

// ---- Frame::ComputeStereoMatches through the class (host/Frame.cc): the rectified-stereo constructor's sequence (src/Frame.cc:96-130) --
// two extractors on two threads with lapping {0, 0}, then ComputeStereoMatches().
// in: "dims" i[3] (w, h, nfeatures), "left" / "right" u8 [h*w], "cal" f[2] (mb, mbf)
// out: "n" / "nr", "kp" raw KeyPoint bytes of the left frame as floats-bit-pattern is not needed: "kpx" "kpy" f, "kpo" i; "uright", "depth" f;
//      "stale": all -1 when the frame is no longer the extractors' latest extraction (must fail loudly, not match something else)
int stereo_smoke(const char *in, const char *out)
{
    FlatFile S;
    if (!S.load(in)) { fprintf(stderr, "cannot read %s\n", in); return 2; }
    FlatWriter O(out);
    const int W = S.I("dims")[0], H = S.I("dims")[1], nfeat = S.I("dims")[2];
    cv::Mat imL(H, W, CV_8U), imR(H, W, CV_8U), mask;
    memcpy(imL.data, S.U("left").data(), (size_t)W * H); memcpy(imR.data, S.U("right").data(), (size_t)W * H);
    ORBextractor exL(nfeat, 1.2f, 8, 20, 7), exR(nfeat, 1.2f, 8, 20, 7);
    Frame F;
    F.mpORBextractorLeft = &exL; F.mpORBextractorRight = &exR;
    F.mb = S.F("cal")[0]; F.mbf = S.F("cal")[1];
    F.mvScaleFactors = exL.GetScaleFactors();
    std::vector<int> lap = {0, 0};
    // Frame::ExtractORB(0, imLeft, 0, 0) / (1, imRight, 0, 0) on two threads (:109-112, :410-417)
    std::thread tl([&] { exL(imL, mask, F.mvKeys, F.mDescriptors, lap); });
    std::thread tr([&] { exR(imR, mask, F.mvKeysRight, F.mDescriptorsRight, lap); });
    tl.join(); tr.join();
    F.N = (int)F.mvKeys.size();
    F.ComputeStereoMatches();
    std::vector<float> kx, ky; std::vector<int32_t> ko;
    for (const cv::KeyPoint &k : F.mvKeys) { kx.push_back(k.pt.x); ky.push_back(k.pt.y); ko.push_back(k.octave); }
    O.one("n", F.N); O.one("nr", (int32_t)F.mvKeysRight.size());
    O.floats("kpx", kx); O.floats("kpy", ky); O.ints("kpo", ko);
    O.floats("uright", F.mvuRight); O.floats("depth", F.mvDepth);
    // a second call gives the same answer (nothing was consumed) ...
    const std::vector<float> ur1 = F.mvuRight, dp1 = F.mvDepth;
    F.ComputeStereoMatches();
    int32_t same = F.mvuRight == ur1 && F.mvDepth == dp1;
    O.one("same", same);
    // ... and once the left extractor has moved on to another image the old frame is refused
    std::vector<cv::KeyPoint> k2; cv::Mat d2;
    exL(imR, mask, k2, d2, lap);
    F.ComputeStereoMatches();
    int32_t stale = 1;
    for (float v : F.mvuRight) if (v != -1.0f) stale = 0;
    for (float v : F.mvDepth) if (v != -1.0f) stale = 0;
    O.one("stale", (F.N > 0 && (k2.size() != F.mvKeys.size() || memcmp(k2.data(), F.mvKeys.data(), sizeof(cv::KeyPoint) * k2.size()) != 0)) ? stale : -1);
    printf("HOST_STEREO_OK\n");
    return 0;
}
