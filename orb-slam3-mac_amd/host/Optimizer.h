// Optimizer.h -- signature-preserving host mirror of the ORB_SLAM3::Optimizer entry points on the hot path
// (reference include/Optimizer.h:58, :98).  LocalMapping (src/LocalMapping.cc:154) calls it unchanged.
#pragma once
#include "slam_types.h"

namespace ORB_SLAM3 {

class Optimizer {
public:
    // reference include/Optimizer.h:58, src/Optimizer.cc:1699-2344
    void static LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, int &num_fixedKF);
    // reference include/Optimizer.h:98, src/Optimizer.cc:4574-5187 (LocalMapping.cc:131-155 once the IMU is initialised)
    void static LocalInertialBA(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, bool bLarge = false, bool bRecInit = false);
};

}  // namespace ORB_SLAM3
