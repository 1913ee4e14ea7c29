// Optimizer.h -- host facade with the reference's static Optimizer API (include/Optimizer.h:17-98 of
// mc275/MC_SLAM) whose hot path runs on the MI355X backend through the C-ABI (include/vislam_ba.h).
//
// Kept from the reference: graph extraction (src/Optimizer.cpp:49-451) and erase + write-back (:496-623), same
// conventions (void return, results written in place into KeyFrame / MapPoint, early return when *pbStopFlag is
// set, Map::mMutexMapUpdate held during write-back only).  Replaced: everything g2o did in between.
#pragma once
#include <list>
#include <vector>

#include "../../include/vislam_ba.h"
#include "orbslam_min.h"

namespace ORB_SLAM2 {

// wall-clock split of one facade call (LocalBAPRVIDP): graph extraction (src/Optimizer.cpp:49-451), the solve behind the C-ABI,
// erase + write-back under the map lock (:496-623)
struct FacadeTiming { double extract_ms = 0, solve_ms = 0, writeback_ms = 0, total_ms = 0; };

// flat arrays of one window in the layout of vba_problem, plus the bookkeeping the write-back needs
struct PackedWindow {
    vba_problem P;
    std::vector<double> pose, vel, bias, pt, uv, w, meas, info;
    std::vector<int32_t> ref, begin, obsKF, imuI, imuJ;
    std::vector<KeyFrame*> vKF;          // free keyframes first
    std::vector<MapPoint*> vMP;          // one per landmark row
    std::vector<KeyFrame*> vEdgeKF;      // per observation edge
    std::vector<MapPoint*> vEdgeMP;
    std::vector<double> refXY;           // variant 2: normalised reference pixel per landmark
    std::vector<uint8_t> outlier, kfFix;
    std::vector<double> chi2;
    vba_result R;
};

class Optimizer {
public:
    // include/Optimizer.h:22-24 (gw is a cv::Mat 3x1 there)
    static void LocalBAPRVIDP(KeyFrame* pKF, const std::list<KeyFrame*>& lLocalKeyFrames, bool* pbStopFlag, Map* pMap,
                              const Vector3d& gw, LocalMapping* pLM = NULL);
    // include/Optimizer.h:36-38: the same window with world-XYZ landmarks and Levenberg-Marquardt
    static void LocalBundleAdjustmentNavStatePRV(KeyFrame* pKF, const std::list<KeyFrame*>& lLocalKeyFrames, bool* pbStopFlag, Map* pMap,
                                                 const Vector3d& gw, LocalMapping* pLM = NULL);
    static bool PackLocalBundleAdjustmentNavStatePRV(KeyFrame* pKF, const std::list<KeyFrame*>& lLocalKeyFrames, const Vector3d& gw, PackedWindow& W);
    static bool PackLocalVI(KeyFrame* pKF, const std::list<KeyFrame*>& lLocalKeyFrames, const Vector3d& gw, bool idp, PackedWindow& W);
    // include/Optimizer.h:74
    static void LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, LocalMapping* pLM = NULL);

    // include/Optimizer.h:33-35, :62-68: global bundle adjustment.  Monocular observations only, as
    // everywhere in this backend (the reference's VI variant prints "Stereo not supported", src/Optimizer.cpp:819).
    static void GlobalBundleAdjustmentNavStatePRV(Map* pMap, const Vector3d& gw, int nIterations, bool* pbStopFlag,
                                                  const unsigned long nLoopKF, const bool bRobust);
    static void GlobalBundleAdjustment(Map* pMap, int nIterations = 5, bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                       const bool bRobust = true);
    static void BundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, int nIterations = 5,
                                 bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true);
    static bool PackGlobalBundleAdjustmentNavStatePRV(Map* pMap, const Vector3d& gw, int nIterations, bool bRobust, PackedWindow& W);
    static bool PackBundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, int nIterations,
                                     bool bRobust, PackedWindow& W);

    // include/Optimizer.h:57-59: local BA over an explicit keyframe list (the VI sliding window), vision edges only
    static void LocalBundleAdjustment(KeyFrame* pKF, const std::list<KeyFrame*>& lLocalKeyFrames, bool* pbStopFlag, Map* pMap,
                                      LocalMapping* pLM = NULL);
    // include/Optimizer.h:77, :26-31: per-frame pose optimisation, vision only and IMU-aided (last keyframe / last frame)
    static int PoseOptimization(Frame* pFrame);
    static int PoseOptimization(Frame* pFrame, KeyFrame* pLastKF, const IMUPreintegrator& imupreint, const Vector3d& gw,
                                const bool& bComputeMarg = false);
    static int PoseOptimization(Frame* pFrame, Frame* pLastFrame, const IMUPreintegrator& imupreint, const Vector3d& gw,
                                const bool& bComputeMarg = false);
    static bool PackLocalBundleAdjustment(KeyFrame* pKF, const std::list<KeyFrame*>* pList, PackedWindow& W);
    static void LocalBundleAdjustmentImpl(KeyFrame* pKF, const std::list<KeyFrame*>* pList, bool* pbStopFlag, Map* pMap, LocalMapping* pLM);

    // graph extraction only (what the two entry points hand to vba_solve); exposed for tests
    static bool PackLocalBAPRVIDP(KeyFrame* pKF, const std::list<KeyFrame*>& lLocalKeyFrames, const Vector3d& gw, PackedWindow& W);
    static bool PackLocalBundleAdjustment(KeyFrame* pKF, PackedWindow& W);
    static const PackedWindow& LastWindow();
    static const FacadeTiming& LastTiming();   // of this thread's last LocalBAPRVIDP call
    static PackedWindow& LastWindowMutable();   // test harness: extraction-only calls pack into it
    static int Device;  // HIP device of the backend handle (one handle per calling thread)
};

}  // namespace ORB_SLAM2
