// compile_callers.cc -- compile-only check of the drop-in claim: the CALL LINES of the reference's Tracking, LocalMapping and LoopClosing
// that reach the hot path, written as the reference writes them (file:line beside each), must build against host/ORBextractor.h,
// host/ORBmatcher.h and host/Optimizer.h.  Nothing here runs; the surrounding control flow of the callers is not restated -- only
// the argument types and the call shapes matter.  Built by `make lib/compile_callers.o`, asserted by tests/test_abi_and_host.py.
#include <mutex>
#include <set>
#include <utility>
#include <vector>
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "Optimizer.h"

using namespace std;

namespace ORB_SLAM3 {

struct CallerState {                      // the members of Tracking / LocalMapping / LoopClosing the call lines mention
    Frame mCurrentFrame, mLastFrame, mInitialFrame;
    KeyFrame *mpReferenceKF, *mpCurrentKeyFrame, *mpCurrentKF;
    vector<MapPoint *> mvpLocalMapPoints;
    vector<cv::Point2f> mvbPrevMatched;
    vector<int> mvIniMatches;
    bool mbAbortBA, mbFarPoints;
    float mThFarPoints;
    cv::Mat mScw;
    ORBextractor *mpORBextractorLeft;
};

// Frame::Frame(imLeft, imRight, ...) (rectified stereo), src/Frame.cc:109-130: the two ExtractORB threads, then ComputeStereoMatches()
void frame_stereo_constructor_calls(Frame &F, const cv::Mat &imLeft, const cv::Mat &imRight)
{
    vector<int> vLapping = {0, 0};
    (*F.mpORBextractorLeft)(imLeft, cv::Mat(), F.mvKeys, F.mDescriptors, vLapping);            // :412-413 (flag == 0)
    (*F.mpORBextractorRight)(imRight, cv::Mat(), F.mvKeysRight, F.mDescriptorsRight, vLapping);  // :415-416
    F.N = F.mvKeys.size();                                                                        // :121
    F.ComputeStereoMatches();                                                                     // :130
}

int tracking_calls(CallerState &S, vector<KeyFrame *> &vpCandidateKFs, bool bMono)
{
    Frame &mCurrentFrame = S.mCurrentFrame, &mLastFrame = S.mLastFrame;
    int total = 0;
    {   // Tracking::MonocularInitialization, src/Tracking.cc:1505-1506
        ORBmatcher matcher(0.9, true);
        int nmatches = matcher.SearchForInitialization(S.mInitialFrame, mCurrentFrame, S.mvbPrevMatched, S.mvIniMatches, 100);
        total += nmatches;
    }
    {   // Tracking::TrackReferenceKeyFrame, src/Tracking.cc:1757-1775
        ORBmatcher matcher(0.7, true);
        vector<MapPoint *> vpMapPointMatches;
        int nmatches = matcher.SearchByBoW(S.mpReferenceKF, mCurrentFrame, vpMapPointMatches);
        mCurrentFrame.mvpMapPoints = vpMapPointMatches;
        mCurrentFrame.SetPose(mLastFrame.mTcw);
        Optimizer::PoseOptimization(&mCurrentFrame);
        total += nmatches;
    }
    {   // Tracking::TrackWithMotionModel, src/Tracking.cc:1881, 1911, 1919, 1934
        ORBmatcher matcher(0.9, true);
        int th = 15;
        int nmatches = matcher.SearchByProjection(mCurrentFrame, mLastFrame, th, bMono);
        if (nmatches < 20) {
            fill(mCurrentFrame.mvpMapPoints.begin(), mCurrentFrame.mvpMapPoints.end(), static_cast<MapPoint *>(NULL));
            nmatches = matcher.SearchByProjection(mCurrentFrame, mLastFrame, 2 * th, bMono);
        }
        Optimizer::PoseOptimization(&mCurrentFrame);
        total += nmatches;
    }
    {   // Tracking::TrackLocalMap / SearchLocalPoints, src/Tracking.cc:1996, 2002, 2405, 2428
        Optimizer::PoseOptimization(&mCurrentFrame);
        ORBmatcher matcher(0.8);
        int th = 1;
        int matches = matcher.SearchByProjection(mCurrentFrame, S.mvpLocalMapPoints, th, S.mbFarPoints, S.mThFarPoints);
        total += matches;
    }
    {   // Tracking::Relocalization, src/Tracking.cc:2645, 2683, 2727-2753
        ORBmatcher matcher(0.75, true);
        ORBmatcher matcher2(0.9, true);
        vector<vector<MapPoint *>> vvpMapPointMatches(vpCandidateKFs.size());
        for (size_t i = 0; i < vpCandidateKFs.size(); i++) {
            int nmatches = matcher.SearchByBoW(vpCandidateKFs[i], mCurrentFrame, vvpMapPointMatches[i]);
            set<MapPoint *> sFound;
            int nGood = Optimizer::PoseOptimization(&mCurrentFrame);
            for (int io = 0; io < mCurrentFrame.N; io++)
                if (mCurrentFrame.mvbOutlier[io]) mCurrentFrame.mvpMapPoints[io] = static_cast<MapPoint *>(NULL);
            int nadditional = matcher2.SearchByProjection(mCurrentFrame, vpCandidateKFs[i], sFound, 10, 100);
            nadditional = matcher2.SearchByProjection(mCurrentFrame, vpCandidateKFs[i], sFound, 3, 64);
            total += nmatches + nGood + nadditional;
        }
    }
    {   // Frame::ExtractORB, src/Frame.cc:410-417 (the extractor instances of src/Tracking.cc:206-212)
        vector<int> vLapping = {0, 1000};
        cv::Mat im, mDescriptors;
        vector<cv::KeyPoint> mvKeys;
        int monoLeft = (*S.mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors, vLapping);
        total += monoLeft;
    }
    return total;
}

int local_mapping_calls(CallerState &S, KeyFrame *pKF2, cv::Mat F12, vector<KeyFrame *> &vpTargetKFs, vector<MapPoint *> &vpFuseCandidates, Map *pMap)
{
    KeyFrame *mpCurrentKeyFrame = S.mpCurrentKeyFrame;
    int total = 0;
    {   // LocalMapping::CreateNewMapPoints, src/LocalMapping.cc:407, 459-463
        ORBmatcher matcher(0.6, false);
        vector<pair<size_t, size_t>> vMatchedIndices;
        bool bCoarse = false;
        matcher.SearchForTriangulation(mpCurrentKeyFrame, pKF2, F12, vMatchedIndices, false, bCoarse);
        total += vMatchedIndices.size();
        vector<cv::Mat> vMatchedPoints;                        // the second overload (include/ORBmatcher.h:76-77): declared, never called in the reference
        matcher.SearchForTriangulation(mpCurrentKeyFrame, pKF2, F12, vMatchedIndices, false, vMatchedPoints);
        total += vMatchedPoints.size();
    }
    {   // LocalMapping::SearchInNeighbors, src/LocalMapping.cc:781-789, 816-817
        ORBmatcher matcher;
        vector<MapPoint *> vpMapPointMatches = mpCurrentKeyFrame->GetMapPointMatches();
        for (vector<KeyFrame *>::iterator vit = vpTargetKFs.begin(), vend = vpTargetKFs.end(); vit != vend; vit++) {
            KeyFrame *pKFi = *vit;
            matcher.Fuse(pKFi, vpMapPointMatches);
            if (pKFi->NLeft != -1) matcher.Fuse(pKFi, vpMapPointMatches, true);
        }
        matcher.Fuse(mpCurrentKeyFrame, vpFuseCandidates);
        if (mpCurrentKeyFrame->NLeft != -1) matcher.Fuse(mpCurrentKeyFrame, vpFuseCandidates, true);
    }
    {   // LocalMapping::Run, src/LocalMapping.cc:138-154
        int num_FixedKF_BA = 0;
        Optimizer::LocalInertialBA(mpCurrentKeyFrame, &S.mbAbortBA, mpCurrentKeyFrame->GetMap(), false, true);
        Optimizer::LocalBundleAdjustment(mpCurrentKeyFrame, &S.mbAbortBA, mpCurrentKeyFrame->GetMap(), num_FixedKF_BA);
        (void)pMap;
        total += num_FixedKF_BA;
    }
    return total;
}

int loop_closing_calls(CallerState &S, vector<KeyFrame *> &vpCovKFi, vector<MapPoint *> &vpMapPoints, vector<KeyFrame *> &vpKeyFrames,
                       vector<KeyFrame *> &vpLocalCurrentWindowKFs, vector<KeyFrame *> &vpMergeConnectedKFs, KeyFrame *pKFi, KeyFrame *pKF2)
{
    KeyFrame *mpCurrentKF = S.mpCurrentKF;
    cv::Mat mScw = S.mScw;
    int total = 0;
    {   // LoopClosing::DetectCommonRegionsFromBoW, src/LoopClosing.cc:578-579, 624, 730, 755
        ORBmatcher matcherBoW(0.9, true);
        ORBmatcher matcher(0.75, true);
        vector<vector<MapPoint *>> vvpMatchedMPs(vpCovKFi.size());
        for (size_t j = 0; j < vpCovKFi.size(); ++j) {
            int num = matcherBoW.SearchByBoW(mpCurrentKF, vpCovKFi[j], vvpMatchedMPs[j]);
            total += num;
        }
        vector<MapPoint *> vpMatchedMP(mpCurrentKF->GetMapPointMatches().size(), static_cast<MapPoint *>(NULL));
        vector<KeyFrame *> vpMatchedKF(mpCurrentKF->GetMapPointMatches().size(), static_cast<KeyFrame *>(NULL));
        int numProjMatches = matcher.SearchByProjection(mpCurrentKF, mScw, vpMapPoints, vpKeyFrames, vpMatchedMP, vpMatchedKF, 8, 1.5);
        int numProjOptMatches = matcher.SearchByProjection(mpCurrentKF, mScw, vpMapPoints, vpMatchedMP, 5, 1.0);
        total += numProjMatches + numProjOptMatches;
    }
    {   // LoopClosing::FindMatchesByProjection, src/LoopClosing.cc:1005-1008
        ORBmatcher matcher(0.9, true);
        vector<MapPoint *> vpMatchedMapPoints(mpCurrentKF->GetMapPointMatches().size(), static_cast<MapPoint *>(NULL));
        int num_matches = matcher.SearchByProjection(mpCurrentKF, mScw, vpMapPoints, vpMatchedMapPoints, 3, 1.5);
        total += num_matches;
    }
    {   // LoopClosing::SearchAndFuse, src/LoopClosing.cc:2284-2300, 2326-2340
        ORBmatcher matcher(0.8);
        cv::Mat cvScw = mScw;
        vector<MapPoint *> vpReplacePoints(vpMapPoints.size(), static_cast<MapPoint *>(NULL));
        int numFused = matcher.Fuse(pKFi, cvScw, vpMapPoints, 4, vpReplacePoints);
        total += numFused;
    }
    {   // the Sim3 guided search (include/ORBmatcher.h:82)
        ORBmatcher matcher(0.75, true);
        vector<MapPoint *> vpMatches12(pKFi->GetMapPointMatches().size(), static_cast<MapPoint *>(NULL));
        const float s12 = 1.f;
        cv::Mat R12 = cv::Mat::eye(3, 3, CV_32F), t12 = cv::Mat::zeros(3, 1, CV_32F);
        total += matcher.SearchBySim3(pKFi, pKF2, vpMatches12, s12, R12, t12, 7.5);
    }
    {   // LoopClosing::MergeLocal, src/LoopClosing.cc:1722
        bool bStop = false;
        Optimizer::LocalBundleAdjustment(mpCurrentKF, vpLocalCurrentWindowKFs, vpMergeConnectedKFs, &bStop);
    }
    {   // the lock Optimizer::PoseOptimization holds while it reads map point positions (src/Optimizer.cc:895) and the one
        // MapPoint::SetWorldPos takes (src/MapPoint.cc:118): both must name the same class-wide mutex
        unique_lock<mutex> lock(MapPoint::mGlobalMutex);
        total += lock.owns_lock() ? 1 : 0;
    }
    return total;
}

}  // namespace ORB_SLAM3
