// Optimizer.h -- signature-preserving host mirror of the ORB_SLAM3::Optimizer entry points on the hot path
// (reference include/Optimizer.h:58, :62, :98, :104).  LocalMapping (src/LocalMapping.cc:154) calls it unchanged.
#pragma once
#include <vector>
#include "slam_types.h"

namespace ORB_SLAM3 {

class Optimizer {
public:
    // reference include/Optimizer.h:62, src/Optimizer.cc:854-1168 (Tracking.cc:1775, 1934, 1996, 2002, 2727, 2743)
    int static PoseOptimization(Frame *pFrame);
    // reference include/Optimizer.h:104, src/Optimizer.cc:6255-6911: local BA of the map-merge welding window (LoopClosing::MergeLocal)
    void static LocalBundleAdjustment(KeyFrame *pMainKF, std::vector<KeyFrame *> vpAdjustKF, std::vector<KeyFrame *> vpFixedKF, bool *pbStopFlag);
    // reference include/Optimizer.h:58, src/Optimizer.cc:1699-2344
    void static LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, int &num_fixedKF);
    // reference include/Optimizer.h:98, src/Optimizer.cc:4574-5187 (LocalMapping.cc:131-155 once the IMU is initialised)
    void static LocalInertialBA(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, bool bLarge = false, bool bRecInit = false);
};

}  // namespace ORB_SLAM3
