// Optimizer.cpp -- see Optimizer.h.  Line references: src/Optimizer.cpp of mc275/MC_SLAM.
#include "Optimizer.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <chrono>

namespace ORB_SLAM2 {

long unsigned int KeyFrame::nNextId = 0;
long unsigned int MapPoint::nNextId = 0;
int Optimizer::Device = 0;

namespace {

thread_local void* t_handle = nullptr;
thread_local PackedWindow t_last;

void* handle() {
    if (!t_handle && vba_create(Optimizer::Device, &t_handle) != 0) t_handle = nullptr;
    return t_handle;
}

Quaterniond MatrixToQuat(const double* m) {  // Eigen::Quaterniond(Matrix3d), then normalised
    Quaterniond q;
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (auto& v : q) v /= n;
    return q;
}

// information of EdgeNavStatePRV: swap V/phi rows+cols of the P,V,phi covariance, invert (:273-280)
bool PRVInformation(const std::array<double, 81>& cov, double* info) {
    static const int perm[9] = {0, 1, 2, 6, 7, 8, 3, 4, 5};
    double M[9][18];
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) { M[i][j] = cov[9 * perm[i] + perm[j]]; M[i][9 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 9; c++) {
        int p = c;
        for (int r = c + 1; r < 9; r++)
            if (std::fabs(M[r][c]) > std::fabs(M[p][c])) p = r;
        if (M[p][c] == 0.0) return false;
        if (p != c)
            for (int j = 0; j < 18; j++) std::swap(M[c][j], M[p][j]);
        const double inv = 1.0 / M[c][c];
        for (int j = 0; j < 18; j++) M[c][j] *= inv;
        for (int r = 0; r < 9; r++) {
            if (r == c || M[r][c] == 0.0) continue;
            const double f = M[r][c];
            for (int j = 0; j < 18; j++) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) info[9 * i + j] = M[i][9 + j];
    return true;
}

// KeyFrame* -> row of the window.  Every observation of the window looks its keyframe up (30 000 times for a 50-keyframe window):
// an open-addressing table over a few hundred slots instead of a std::map walk (the reference pays optimizer.vertex(id), a
// tr1::unordered_map look-up, per edge endpoint: Thirdparty/g2o/g2o/core/hyper_graph.cpp:60-66).
struct KfTable {
    std::vector<std::pair<const KeyFrame*, int>> slot;
    size_t mask = 0;
    void reset(size_t n) {
        size_t cap = 64;
        while (cap < 4 * n) cap *= 2;
        slot.assign(cap, {nullptr, -1});
        mask = cap - 1;
    }
    static size_t hash(const KeyFrame* k) { return (size_t)((reinterpret_cast<uintptr_t>(k) >> 4) * 0x9E3779B97F4A7C15ull >> 20); }
    void put(const KeyFrame* k, int row) {
        if (2 * (size_t)(row + 1) > slot.size()) {   // (the fixed keyframes are discovered on the way: grow before the table fills)
            std::vector<std::pair<const KeyFrame*, int>> old;
            old.swap(slot);
            reset(old.size());
            for (auto& e : old) if (e.first) put(e.first, e.second);
        }
        size_t i = hash(k) & mask;
        while (slot[i].first && slot[i].first != k) i = (i + 1) & mask;
        slot[i] = {k, row};
    }
    int get(const KeyFrame* k) const {
        size_t i = hash(k) & mask;
        while (slot[i].first) {
            if (slot[i].first == k) return slot[i].second;
            i = (i + 1) & mask;
        }
        return -1;
    }
};

void FinishProblem(PackedWindow& W) {
    vba_problem& P = W.P;
    P.n_kf = (int32_t)W.vKF.size();
    P.n_pt = (int32_t)W.vMP.size();
    P.n_obs = (int32_t)W.obsKF.size();
    P.n_imu = (int32_t)W.imuI.size();
    P.kf_pose = W.pose.data(); P.kf_vel = W.vel.data(); P.kf_bias = W.bias.data(); P.pt = W.pt.data();
    P.pt_ref_kf = W.ref.data(); P.pt_obs_begin = W.begin.data(); P.obs_kf = W.obsKF.data();
    P.obs_uv = W.uv.data(); P.obs_w = W.w.data();
    P.imu_kf_i = W.imuI.data(); P.imu_kf_j = W.imuJ.data(); P.imu_meas = W.meas.data(); P.imu_info_prv = W.info.data();
    P.inv_bg_rw2 = 1.0 / IMUData::getGyrBiasRW2();   // :246
    P.inv_ba_rw2 = 1.0 / IMUData::getAccBiasRW2();   // :249
    P.huber_vis = (double)(float)std::sqrt(5.991);             // const float thHuberMono, :327
    P.huber_prv = (double)(float)std::sqrt(100 * 21.666);      // :241
    P.huber_bias = (double)(float)std::sqrt(100 * 16.812);     // :242
    P.its_stage1 = 5; P.its_stage2 = 10;                       // :459, :493
    P.chi2_th = 5.991;
    W.outlier.assign(P.n_obs ? P.n_obs : 1, 0);
    W.chi2.assign(P.n_obs ? P.n_obs : 1, 0.0);
    std::memset(&W.R, 0, sizeof W.R);
    W.R.obs_outlier = W.outlier.data();
    W.R.obs_chi2 = W.chi2.data();
}

// g2o's forceStopFlag is a bool* written by the Tracking thread (&LocalMapping::mbAbortBA, src/LocalMapping.cpp:1769-1772): the
// backend reads the caller's flag at its own width (vba_solve_b), so the pointer goes straight through -- the device sees a raised
// flag at its next poll because the host forwards it whenever it enqueues an iteration and while it waits.  (Until round 4 the ABI
// only took an int*: a thread per call mirrored the bool into one.)
static_assert(sizeof(bool) == 1, "vba_solve_b reads the stop flag as one byte");
inline const volatile unsigned char* StopPtr(bool* pbStopFlag) { return reinterpret_cast<const volatile unsigned char*>(pbStopFlag); }

// wall-clock split of the last facade call of this thread (Optimizer::LastTiming): extraction, solve (vba_solve_b: H2D + structure +
// two-stage solve + D2H), erase + write-back under the map lock
thread_local FacadeTiming t_timing;
inline double NowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

const PackedWindow& Optimizer::LastWindow() { return t_last; }
const FacadeTiming& Optimizer::LastTiming() { return t_timing; }
PackedWindow& Optimizer::LastWindowMutable() { return t_last; }

// ------------------------------------------------------------------------------------------------
bool Optimizer::PackLocalBAPRVIDP(KeyFrame* pCurKF, const std::list<KeyFrame*>& lLocalKeyFrames, const Vector3d& gw,
                                  PackedWindow& W) {
    return PackLocalVI(pCurKF, lLocalKeyFrames, gw, true, W);
}
bool Optimizer::PackLocalBundleAdjustmentNavStatePRV(KeyFrame* pCurKF, const std::list<KeyFrame*>& lLocalKeyFrames, const Vector3d& gw,
                                                     PackedWindow& W) {
    return PackLocalVI(pCurKF, lLocalKeyFrames, gw, false, W);
}

// the two visual-inertial local windows share everything but the landmark model: inverse depth in the reference keyframe
// (LocalBAPRVIDP, :32-625) or world XYZ (LocalBundleAdjustmentNavStatePRV, :937-1388)
bool Optimizer::PackLocalVI(KeyFrame* pCurKF, const std::list<KeyFrame*>& lLocalKeyFrames, const Vector3d& gw, bool idp,
                            PackedWindow& W) {
    static const bool ptm = getenv("VBA_FACADE_TIMING") != nullptr;
    double tp[8]; int ti = 0; tp[ti++] = NowMs();
    W = PackedWindow();
    std::memset(&W.P, 0, sizeof W.P);
    if (pCurKF != lLocalKeyFrames.back()) std::cerr << "pCurKF != lLocalKeyFrames.back. check" << std::endl;  // :37-38
    for (KeyFrame* pKFi : lLocalKeyFrames) pKFi->mnBALocalForKF = pCurKF->mnId;                                 // :49-56
    std::list<MapPoint*> lLocalMapPoints;                                                                          // :59-79
    for (KeyFrame* pKFi : lLocalKeyFrames)
        for (MapPoint* pMP : pKFi->GetMapPointMatches())
            if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pCurKF->mnId) {
                lLocalMapPoints.push_back(pMP);
                pMP->mnBALocalForKF = pCurKF->mnId;
            }
    tp[ti++] = NowMs();
    // One copy of every map point's observation list serves both passes that read it -- the search for the fixed cameras (:82-127) and
    // the edges (:337-451); GetObservations() returns the std::map by value (include/MapPoint.h), the reference copies it twice.
    std::vector<MapPoint*> vLocalMP(lLocalMapPoints.begin(), lLocalMapPoints.end());
    std::vector<mapMapPointObs> vObs(vLocalMP.size());
    size_t nObsTotal = 0;
    for (size_t i = 0; i < vLocalMP.size(); i++) { vObs[i] = vLocalMP[i]->GetObservations(); nObsTotal += vObs[i].size(); }
    tp[ti++] = NowMs();
    std::list<KeyFrame*> lFixedCameras;                                                                           // :82-127
    KeyFrame* pKFPrevLocal = lLocalKeyFrames.front()->GetPrevKeyFrame();
    if (pKFPrevLocal) {
        pKFPrevLocal->mnBAFixedForKF = pCurKF->mnId;
        if (!pKFPrevLocal->isBad()) lFixedCameras.push_back(pKFPrevLocal);
    } else
        std::cerr << "pKFPrevLocal is NULL?" << std::endl;
    for (size_t i = 0; i < vLocalMP.size(); i++)
        for (auto& mit : vObs[i]) {
            KeyFrame* pKFi = mit.first;
            if (pKFi->mnBALocalForKF != pCurKF->mnId && pKFi->mnBAFixedForKF != pCurKF->mnId) {
                pKFi->mnBAFixedForKF = pCurKF->mnId;
                if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
            }
        }
    tp[ti++] = NowMs();
    // vertices -> rows: free keyframes (window order = ascending mnId) first, fixed after            (:159-232)
    KfTable kfIdx;
    kfIdx.reset(lLocalKeyFrames.size() + lFixedCameras.size());
    for (KeyFrame* k : lLocalKeyFrames) { kfIdx.put(k, (int)W.vKF.size()); W.vKF.push_back(k); }
    W.P.n_kf_free = (int32_t)W.vKF.size();
    for (KeyFrame* k : lFixedCameras) { kfIdx.put(k, (int)W.vKF.size()); W.vKF.push_back(k); }
    for (KeyFrame* k : W.vKF) {
        const NavState& ns = k->GetNavState();
        const Vector3d P = ns.Get_P(), V = ns.Get_V(), bg = ns.Get_BiasGyr(), ba = ns.Get_BiasAcc(), dbg = ns.Get_dBias_Gyr(), dba = ns.Get_dBias_Acc();
        const Quaterniond q = ns.Get_R();
        W.pose.insert(W.pose.end(), {P[0], P[1], P[2], q[0], q[1], q[2], q[3]});
        W.vel.insert(W.vel.end(), {V[0], V[1], V[2]});
        W.bias.insert(W.bias.end(), {bg[0], bg[1], bg[2], ba[0], ba[1], ba[2], dbg[0], dbg[1], dbg[2], dba[0], dba[1], dba[2]});
    }
    // IMU factors: one EdgeNavStatePRV + one EdgeNavStateBias per local keyframe                     (:251-312)
    for (KeyFrame* pKF1 : lLocalKeyFrames) {
        KeyFrame* pKF0 = pKF1->GetPrevKeyFrame();
        if (!pKF0 || kfIdx.get(pKF0) < 0) { std::cerr << "pKF0 missing" << std::endl; continue; }
        const IMUPreintegrator& M = pKF1->GetIMUPreInt();
        W.imuI.push_back(kfIdx.get(pKF0));
        W.imuJ.push_back(kfIdx.get(pKF1));
        W.meas.push_back(M.getDeltaTime());
        for (double v : M.getDeltaP()) W.meas.push_back(v);
        for (double v : M.getDeltaV()) W.meas.push_back(v);
        for (const Matrix3d* A : {&M.getDeltaR(), &M.getJPBiasg(), &M.getJPBiasa(), &M.getJVBiasg(), &M.getJVBiasa(), &M.getJRBiasg()})
            for (double v : *A) W.meas.push_back(v);
        double info[81];
        if (!PRVInformation(M.getCovPVPhi(), info)) { std::cerr << "singular preintegration covariance" << std::endl; return false; }
        W.info.insert(W.info.end(), info, info + 81);
    }
    tp[ti++] = NowMs();
    W.obsKF.reserve(nObsTotal); W.uv.reserve(2 * nObsTotal); W.w.reserve(nObsTotal);
    W.vEdgeKF.reserve(nObsTotal); W.vEdgeMP.reserve(nObsTotal);
    W.pt.reserve(3 * vLocalMP.size()); W.ref.reserve(vLocalMP.size()); W.refXY.reserve(2 * vLocalMP.size());
    W.vMP.reserve(vLocalMP.size()); W.begin.reserve(vLocalMP.size() + 1);
    W.begin.push_back(0);
    if (!idp) {   // VertexSBAPointXYZ + one EdgeNavStatePRPointXYZ per observation (:1151-1193)
        for (size_t ip = 0; ip < vLocalMP.size(); ip++) {
            MapPoint* pMP = vLocalMP[ip];
            double Pw[3];
            pMP->GetWorldPos(Pw);
            for (auto& mit : vObs[ip]) {
                KeyFrame* pKFi = mit.first;
                if (pKFi->isBad()) continue;
                if (pKFi->mvuRight[mit.second] >= 0) { std::cerr << "Stereo not supported yet" << std::endl; continue; }
                const KeyPoint& kpUn = pKFi->mvKeysUn[mit.second];
                W.obsKF.push_back(kfIdx.get(pKFi));
                W.uv.push_back(kpUn.pt.x); W.uv.push_back(kpUn.pt.y);
                W.w.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);
                W.vEdgeKF.push_back(pKFi);
                W.vEdgeMP.push_back(pMP);
                W.P.K[0] = pKFi->fx; W.P.K[1] = pKFi->fy; W.P.K[2] = pKFi->cx; W.P.K[3] = pKFi->cy;
            }
            W.pt.insert(W.pt.end(), {Pw[0], Pw[1], Pw[2]});
            W.ref.push_back(0);
            W.vMP.push_back(pMP);
            W.begin.push_back((int32_t)W.obsKF.size());
        }
    }
    // landmarks and EdgePRIDP edges                                                                   (:337-451)
    for (size_t ip = 0; ip < vLocalMP.size(); ip++) {
        if (!idp) break;
        MapPoint* pMP = vLocalMP[ip];
        double Pw[3];
        pMP->GetWorldPos(Pw);
        KeyFrame* pRefKF = pMP->GetReferenceKeyFrame();
        if (!pRefKF || pRefKF->isBad()) continue;                                                                // :349-353
        double Rcw[9], tcw[3];
        pRefKF->GetRotation(Rcw);
        pRefKF->GetTranslation(tcw);
        const double dc = Rcw[6] * Pw[0] + Rcw[7] * Pw[1] + Rcw[8] * Pw[2] + tcw[2];                             // :355-358
        if (dc < 0.01) continue;                                                                                 // :360-365
        const mapMapPointObs& observations = vObs[ip];
        const auto itRef = observations.find(pRefKF);
        if (itRef == observations.end()) { std::cerr << "!observations.count(pRefKF)" << std::endl; continue; }
        const KeyPoint& kpRefUn = pRefKF->mvKeysUn[itRef->second];
        const double normx = (kpRefUn.pt.x - pRefKF->cx) / pRefKF->fx, normy = (kpRefUn.pt.y - pRefKF->cy) / pRefKF->fy;  // :382-385
        const size_t nBefore = W.obsKF.size();
        for (auto& mit : observations) {
            KeyFrame* pKFi = mit.first;
            if (pKFi == pRefKF || pKFi->isBad()) continue;                                                       // :395-400
            if (pKFi->mvuRight[mit.second] >= 0) { std::cerr << "Stereo not supported yet" << std::endl; continue; }
            const KeyPoint& kpUn = pKFi->mvKeysUn[mit.second];
            W.obsKF.push_back(kfIdx.get(pKFi));
            W.uv.push_back(kpUn.pt.x); W.uv.push_back(kpUn.pt.y);
            W.w.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);                                                  // float -> double, :428-429
            W.vEdgeKF.push_back(pKFi);
            W.vEdgeMP.push_back(pMP);
        }
        if (W.obsKF.size() == nBefore) continue;  // no edge -> the vertex is never added (:407-412, 588-589)
        W.pt.insert(W.pt.end(), {1.0 / dc, normx, normy});
        W.ref.push_back(kfIdx.get(pRefKF));
        W.refXY.push_back(normx); W.refXY.push_back(normy);
        W.vMP.push_back(pMP);
        W.begin.push_back((int32_t)W.obsKF.size());
        W.P.K[0] = pRefKF->fx; W.P.K[1] = pRefKF->fy; W.P.K[2] = pRefKF->cx; W.P.K[3] = pRefKF->cy;              // :422
    }
    tp[ti++] = NowMs();
    W.P.variant = idp ? VBA_VARIANT_PRV_IDP : VBA_VARIANT_PRV_XYZ;
    W.P.algo = idp ? VBA_ALGO_GN : VBA_ALGO_LM;                                                                   // :136 / :1028
    Matrix3d Rcb; Vector3d tcb;
    ConfigParam::GetEigT_cb(Rcb, tcb);                                                                            // :41-43
    const Quaterniond qcb = MatrixToQuat(Rcb.data());
    for (int i = 0; i < 3; i++) { W.P.T_cb[i] = tcb[i]; W.P.g_w[i] = gw[i]; }
    for (int i = 0; i < 4; i++) W.P.T_cb[3 + i] = qcb[i];
    W.P.depth_min = idp ? 0.01 : 0.0; W.P.rho_min = 2e-6;                                                         // g2otypes.h:122-127 / :300-303, :484
    if (W.vMP.empty() || W.obsKF.empty()) return false;   // an empty graph: the reference's optimize() changes nothing
    FinishProblem(W);
    tp[ti++] = NowMs();
    if (ptm) { fprintf(stderr, "[facade] pack:"); for (int i = 1; i < ti; i++) fprintf(stderr, " %.3f", tp[i] - tp[i - 1]); fprintf(stderr, " ms (marks+list | obs copies | fixed cams | rows+imu | edges | finish)\n"); }
    return true;
}

void Optimizer::LocalBAPRVIDP(KeyFrame* pCurKF, const std::list<KeyFrame*>& lLocalKeyFrames, bool* pbStopFlag, Map* pMap,
                              const Vector3d& gw, LocalMapping* pLM) {
    PackedWindow& W = t_last;
    t_timing = FacadeTiming();
    const double t0 = NowMs();
    if (!PackLocalBAPRVIDP(pCurKF, lLocalKeyFrames, gw, W)) return;
    t_timing.extract_ms = NowMs() - t0;
    if (pbStopFlag && *pbStopFlag) return;                                                                        // :453-455
    void* h = handle();
    if (!h) { std::cerr << "LocalBAPRVIDP: no HIP device, local BA skipped (the backend has no CPU path)" << std::endl; return; }
    const double t1 = NowMs();
    if (vba_solve_b(h, &W.P, &W.R, StopPtr(pbStopFlag)) != 0) {
        std::cerr << "LocalBAPRVIDP: " << vba_last_error(h) << std::endl;
        return;
    }
    const double t2 = NowMs();
    t_timing.solve_ms = t2 - t1;
    struct WriteBackTimer { double t; ~WriteBackTimer() { t_timing.writeback_ms = NowMs() - t; t_timing.total_ms = t_timing.extract_ms + t_timing.solve_ms + t_timing.writeback_ms; } } wbt{t2};
    if (W.R.status == VBA_ABORTED_BEFORE) return;
    if (W.R.status == VBA_ABORTED_AFTER_STAGE1)
        std::cerr << "Hint: local mapping optimize only 5 iter. Need more computation resource." << std::endl;  // :469-470
    std::vector<std::pair<KeyFrame*, MapPoint*>> vToErase;                                                        // :496-517
    for (size_t i = 0; i < W.vEdgeKF.size(); i++) {
        if (W.vEdgeMP[i]->isBad()) continue;
        if (W.outlier[i]) vToErase.push_back(std::make_pair(W.vEdgeKF[i], W.vEdgeMP[i]));
    }
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);                                                     // :520
    for (auto& e : vToErase) { e.first->EraseMapPointMatch(e.second); e.second->EraseObservation(e.first); }      // :523-532
    for (int i = 0; i < W.P.n_kf_free; i++) {                                                                     // :536-570
        KeyFrame* pKFi = W.vKF[i];
        pKFi->SetNavStatePos({{W.pose[7 * i], W.pose[7 * i + 1], W.pose[7 * i + 2]}});
        pKFi->SetNavStateRot({{W.pose[7 * i + 3], W.pose[7 * i + 4], W.pose[7 * i + 5], W.pose[7 * i + 6]}});
        pKFi->SetNavStateVel({{W.vel[3 * i], W.vel[3 * i + 1], W.vel[3 * i + 2]}});
        pKFi->SetNavStateDeltaBg({{W.bias[12 * i + 6], W.bias[12 * i + 7], W.bias[12 * i + 8]}});
        pKFi->SetNavStateDeltaBa({{W.bias[12 * i + 9], W.bias[12 * i + 10], W.bias[12 * i + 11]}});
        pKFi->UpdatePoseFromNS();
    }
    for (size_t p = 0; p < W.vMP.size(); p++) {                                                                   // :573-606
        MapPoint* pMP = W.vMP[p];
        KeyFrame* pRefKF = pMP->GetReferenceKeyFrame();
        if (pRefKF->isBad()) continue;
        const double rho = W.pt[3 * p];
        const float Pref[3] = {(float)(W.refXY[2 * p] / rho), (float)(W.refXY[2 * p + 1] / rho), (float)(1.0 / rho)};
        double Rcw[9], twr[3];
        pRefKF->GetRotation(Rcw);       // float32 values of the (updated) reference pose
        pRefKF->GetCameraCenter(twr);
        float Pw[3];
        for (int i = 0; i < 3; i++)     // Rwr * Pref + twr in float32 like cv::Mat
            Pw[i] = (float)Rcw[i] * Pref[0] + (float)Rcw[3 + i] * Pref[1] + (float)Rcw[6 + i] * Pref[2] + (float)twr[i];
        pMP->SetWorldPos(Pw);
        pMP->UpdateNormalAndDepth();
    }
    if (pLM) pLM->SetMapUpdateFlagInTracking(true);                                                               // :620-623
}

void Optimizer::LocalBundleAdjustmentNavStatePRV(KeyFrame* pCurKF, const std::list<KeyFrame*>& lLocalKeyFrames, bool* pbStopFlag,
                                                 Map* pMap, const Vector3d& gw, LocalMapping* pLM) {             // :937-1388
    PackedWindow& W = t_last;
    if (!PackLocalBundleAdjustmentNavStatePRV(pCurKF, lLocalKeyFrames, gw, W)) return;
    if (pbStopFlag && *pbStopFlag) return;                                                                        // :1196-1198
    void* h = handle();
    if (!h) { std::cerr << "LocalBundleAdjustmentNavStatePRV: no HIP device, local BA skipped (the backend has no CPU path)" << std::endl; return; }
    {
        if (vba_solve_b(h, &W.P, &W.R, StopPtr(pbStopFlag)) != 0) {
            std::cerr << "LocalBundleAdjustmentNavStatePRV: " << vba_last_error(h) << std::endl;
            return;
        }
    }
    if (W.R.status == VBA_ABORTED_BEFORE) return;
    std::vector<std::pair<KeyFrame*, MapPoint*>> vToErase;                                                        // :1222-1238
    for (size_t i = 0; i < W.vEdgeKF.size(); i++) {
        if (W.vEdgeMP[i]->isBad()) continue;
        if (W.outlier[i]) vToErase.push_back(std::make_pair(W.vEdgeKF[i], W.vEdgeMP[i]));
    }
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
    for (auto& e : vToErase) { e.first->EraseMapPointMatch(e.second); e.second->EraseObservation(e.first); }
    for (int i = 0; i < W.P.n_kf_free; i++) {                                                                     // :1250-1270
        KeyFrame* pKFi = W.vKF[i];
        pKFi->SetNavStatePos({{W.pose[7 * i], W.pose[7 * i + 1], W.pose[7 * i + 2]}});
        pKFi->SetNavStateRot({{W.pose[7 * i + 3], W.pose[7 * i + 4], W.pose[7 * i + 5], W.pose[7 * i + 6]}});
        pKFi->SetNavStateVel({{W.vel[3 * i], W.vel[3 * i + 1], W.vel[3 * i + 2]}});
        pKFi->SetNavStateDeltaBg({{W.bias[12 * i + 6], W.bias[12 * i + 7], W.bias[12 * i + 8]}});
        pKFi->SetNavStateDeltaBa({{W.bias[12 * i + 9], W.bias[12 * i + 10], W.bias[12 * i + 11]}});
        pKFi->UpdatePoseFromNS();
    }
    for (size_t p = 0; p < W.vMP.size(); p++) {                                                                   // :1275-1283
        const float Pw[3] = {(float)W.pt[3 * p], (float)W.pt[3 * p + 1], (float)W.pt[3 * p + 2]};
        W.vMP[p]->SetWorldPos(Pw);
        W.vMP[p]->UpdateNormalAndDepth();
    }
    if (pLM) pLM->SetMapUpdateFlagInTracking(true);
}

// ------------------------------------------------------------------------------------------------
bool Optimizer::PackLocalBundleAdjustment(KeyFrame* pKF, PackedWindow& W) { return PackLocalBundleAdjustment(pKF, nullptr, W); }

bool Optimizer::PackLocalBundleAdjustment(KeyFrame* pKF, const std::list<KeyFrame*>* pList, PackedWindow& W) {
    W = PackedWindow();
    std::memset(&W.P, 0, sizeof W.P);
    std::list<KeyFrame*> lLocalKeyFrames;
    if (pList) {                                                                                                  // :2981-2986
        lLocalKeyFrames = *pList;
        for (KeyFrame* pKFi : lLocalKeyFrames) pKFi->mnBALocalForKF = pKF->mnId;
    } else {                                                                                                      // :3861-3875
        lLocalKeyFrames.push_back(pKF);
        pKF->mnBALocalForKF = pKF->mnId;
        for (KeyFrame* pKFi : pKF->GetVectorCovisibleKeyFrames()) {
            pKFi->mnBALocalForKF = pKF->mnId;
            if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
        }
    }
    std::list<MapPoint*> lLocalMapPoints;                                                                         // :3878-3895
    for (KeyFrame* pKFi : lLocalKeyFrames)
        for (MapPoint* pMP : pKFi->GetMapPointMatches())
            if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) {
                lLocalMapPoints.push_back(pMP);
                pMP->mnBALocalForKF = pKF->mnId;
            }
    std::list<KeyFrame*> lFixedCameras;                                                                           // :3898-3915
    if (pList) {   // the keyframe before the window is fixed first (:3005-3023)
        KeyFrame* pKFPrevLocal = lLocalKeyFrames.front()->GetPrevKeyFrame();
        if (pKFPrevLocal) {
            pKFPrevLocal->mnBAFixedForKF = pKF->mnId;
            if (!pKFPrevLocal->isBad()) lFixedCameras.push_back(pKFPrevLocal);
        }
    }
    for (MapPoint* pMP : lLocalMapPoints)
        for (auto& mit : pMP->GetObservations()) {
            KeyFrame* pKFi = mit.first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
            }
        }
    std::map<KeyFrame*, int> kfIdx;
    std::vector<KeyFrame*> fixedLocal;
    for (KeyFrame* k : lLocalKeyFrames) {
        if (k->mnId == 0) { fixedLocal.push_back(k); continue; }   // vSE3->setFixed(pKFi->mnId == 0), :3944
        kfIdx[k] = (int)W.vKF.size();
        W.vKF.push_back(k);
    }
    W.P.n_kf_free = (int32_t)W.vKF.size();
    for (KeyFrame* k : fixedLocal) { kfIdx[k] = (int)W.vKF.size(); W.vKF.push_back(k); }
    for (KeyFrame* k : lFixedCameras) { kfIdx[k] = (int)W.vKF.size(); W.vKF.push_back(k); }
    if (W.P.n_kf_free == 0) return false;
    for (KeyFrame* k : W.vKF) {  // Converter::toSE3Quat(pKFi->GetPose()): float32 -> double, Quaterniond(R), normalizeRotation
        double R[9], t[3];
        k->GetRotation(R);
        k->GetTranslation(t);
        Quaterniond q = MatrixToQuat(R);
        if (q[3] < 0) for (auto& v : q) v = -v;
        W.pose.insert(W.pose.end(), {t[0], t[1], t[2], q[0], q[1], q[2], q[3]});
        W.vel.insert(W.vel.end(), {0, 0, 0});
        W.bias.insert(W.bias.end(), 12, 0.0);
    }
    W.begin.push_back(0);
    for (MapPoint* pMP : lLocalMapPoints) {                                                                       // :3992-4085
        double Pw[3];
        pMP->GetWorldPos(Pw);
        for (auto& mit : pMP->GetObservations()) {
            KeyFrame* pKFi = mit.first;
            if (pKFi->isBad() || pKFi->mvuRight[mit.second] >= 0) continue;  // stereo branch never taken for mono
            const KeyPoint& kpUn = pKFi->mvKeysUn[mit.second];
            W.obsKF.push_back(kfIdx[pKFi]);
            W.uv.push_back(kpUn.pt.x); W.uv.push_back(kpUn.pt.y);
            W.w.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);
            W.vEdgeKF.push_back(pKFi);
            W.vEdgeMP.push_back(pMP);
            W.P.K[0] = pKFi->fx; W.P.K[1] = pKFi->fy; W.P.K[2] = pKFi->cx; W.P.K[3] = pKFi->cy;
        }
        W.pt.insert(W.pt.end(), {Pw[0], Pw[1], Pw[2]});
        W.ref.push_back(0);
        W.vMP.push_back(pMP);
        W.begin.push_back((int32_t)W.obsKF.size());
    }
    if (W.vMP.empty() || W.obsKF.empty()) return false;   // nothing to optimise: g2o would run on an empty graph and change nothing
    W.P.variant = VBA_VARIANT_SE3_XYZ;
    W.P.algo = VBA_ALGO_LM;                                                                                       // :3928
    W.P.T_cb[6] = 1.0;
    W.P.depth_min = 0.0;
    if (W.vMP.empty() || W.obsKF.empty()) return false;   // an empty graph: the reference's optimize() changes nothing
    FinishProblem(W);
    return true;
}

void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, const std::list<KeyFrame*>& lLocalKeyFrames, bool* pbStopFlag, Map* pMap,
                                      LocalMapping* pLM) {
    if (pKF != lLocalKeyFrames.back()) std::cerr << "pKF != lLocalKeyFrames.back. check" << std::endl;                // :2978-2979
    LocalBundleAdjustmentImpl(pKF, &lLocalKeyFrames, pbStopFlag, pMap, pLM);
}

void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, LocalMapping* pLM) {
    LocalBundleAdjustmentImpl(pKF, nullptr, pbStopFlag, pMap, pLM);
}

void Optimizer::LocalBundleAdjustmentImpl(KeyFrame* pKF, const std::list<KeyFrame*>* pList, bool* pbStopFlag, Map* pMap, LocalMapping* pLM) {
    PackedWindow& W = t_last;
    if (!PackLocalBundleAdjustment(pKF, pList, W)) return;
    if (pbStopFlag && *pbStopFlag) return;                                                                        // :4088-4090
    void* h = handle();
    if (!h) { std::cerr << "LocalBundleAdjustment: no HIP device, local BA skipped (the backend has no CPU path)" << std::endl; return; }
    {
        if (vba_solve_b(h, &W.P, &W.R, StopPtr(pbStopFlag)) != 0) {
            std::cerr << "LocalBundleAdjustment: " << vba_last_error(h) << std::endl;
            return;
        }
    }
    if (W.R.status == VBA_ABORTED_BEFORE) return;
    std::vector<std::pair<KeyFrame*, MapPoint*>> vToErase;                                                        // :4146-4163
    for (size_t i = 0; i < W.vEdgeKF.size(); i++) {
        if (W.vEdgeMP[i]->isBad()) continue;
        if (W.outlier[i]) vToErase.push_back(std::make_pair(W.vEdgeKF[i], W.vEdgeMP[i]));
    }
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);                                                     // :4181
    for (auto& e : vToErase) { e.first->EraseMapPointMatch(e.second); e.second->EraseObservation(e.first); }
    for (int i = 0; i < W.P.n_kf_free; i++) {                                                                     // :4198-4204, Converter::toCvMat(SE3Quat)
        const Matrix3d R = QuatToMatrix({{W.pose[7 * i + 3], W.pose[7 * i + 4], W.pose[7 * i + 5], W.pose[7 * i + 6]}});
        Mat4f T{};
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) T[4 * r + c] = (float)R[3 * r + c];
            T[4 * r + 3] = (float)W.pose[7 * i + r];
        }
        T[15] = 1.0f;
        W.vKF[i]->SetPose(T);
    }
    for (size_t p = 0; p < W.vMP.size(); p++) {                                                                   // :4207-4214
        const float Pw[3] = {(float)W.pt[3 * p], (float)W.pt[3 * p + 1], (float)W.pt[3 * p + 2]};
        W.vMP[p]->SetWorldPos(Pw);
        W.vMP[p]->UpdateNormalAndDepth();
    }
    if (pLM) pLM->SetMapUpdateFlagInTracking(true);
}


// ------------------------------------------------------------------------------------------------
// Global bundle adjustment.  Same factors as the local windows at map scale, one optimize(nIterations) with
// Levenberg-Marquardt, no outlier pass (VBA_PROTO_SINGLE).
// ------------------------------------------------------------------------------------------------
namespace {
// landmarks + monocular reprojection edges of a whole map (src/Optimizer.cpp:776-833 / :3417-3513)
void PackMapPoints(const std::vector<MapPoint*>& vpMP, std::map<KeyFrame*, int>& kfIdx, PackedWindow& W) {
    W.begin.push_back(0);
    for (MapPoint* pMP : vpMP) {
        if (pMP->isBad()) continue;
        const size_t e0 = W.obsKF.size();
        for (auto& mit : pMP->GetObservations()) {
            KeyFrame* pKF = mit.first;
            if (pKF->isBad() || !kfIdx.count(pKF)) continue;       // "|| pKF->mnId > maxKFid"
            if (pKF->mvuRight[mit.second] >= 0) continue;          // stereo: not supported here
            const KeyPoint& kpUn = pKF->mvKeysUn[mit.second];
            W.obsKF.push_back(kfIdx[pKF]);
            W.uv.push_back(kpUn.pt.x); W.uv.push_back(kpUn.pt.y);
            W.w.push_back(pKF->mvInvLevelSigma2[kpUn.octave]);
            W.vEdgeKF.push_back(pKF);
            W.vEdgeMP.push_back(pMP);
            W.P.K[0] = pKF->fx; W.P.K[1] = pKF->fy; W.P.K[2] = pKF->cx; W.P.K[3] = pKF->cy;
        }
        if (W.obsKF.size() == e0) continue;                        // nEdges == 0: vbNotIncludedMP, vertex removed
        double Pw[3];
        pMP->GetWorldPos(Pw);
        W.pt.insert(W.pt.end(), {Pw[0], Pw[1], Pw[2]});
        W.ref.push_back(0);
        W.vMP.push_back(pMP);
        W.begin.push_back((int32_t)W.obsKF.size());
    }
}

bool RunGlobal(PackedWindow& W, bool* pbStopFlag, const char* who) {
    void* h = handle();
    if (!h) { std::cerr << who << ": no HIP device, global BA skipped (the backend has no CPU path)" << std::endl; return false; }
    if (vba_solve_b(h, &W.P, &W.R, StopPtr(pbStopFlag)) != 0) {
        std::cerr << who << ": " << vba_last_error(h) << std::endl;
        return false;
    }
    return true;  // a raised stop flag leaves the estimates untouched; the reference still writes them back
}

void WriteBackMapPoints(PackedWindow& W, const unsigned long nLoopKF) {   // :915-931 / :3583-3604
    for (size_t p = 0; p < W.vMP.size(); p++) {
        const float Pw[3] = {(float)W.pt[3 * p], (float)W.pt[3 * p + 1], (float)W.pt[3 * p + 2]};
        if (nLoopKF == 0) {
            W.vMP[p]->SetWorldPos(Pw);
            W.vMP[p]->UpdateNormalAndDepth();
        } else {
            for (int i = 0; i < 3; i++) W.vMP[p]->mPosGBA[i] = Pw[i];
            W.vMP[p]->mnBAGlobalForKF = nLoopKF;
        }
    }
}
}  // namespace

bool Optimizer::PackGlobalBundleAdjustmentNavStatePRV(Map* pMap, const Vector3d& gw, int nIterations, bool bRobust, PackedWindow& W) {
    W = PackedWindow();
    std::memset(&W.P, 0, sizeof W.P);
    std::map<KeyFrame*, int> kfIdx;
    for (KeyFrame* pKF : pMap->GetAllKeyFrames()) {                                                               // :661-690
        if (pKF->isBad()) continue;
        kfIdx[pKF] = (int)W.vKF.size();
        W.vKF.push_back(pKF);
        const NavState& ns = pKF->GetNavState();
        const Vector3d P = ns.Get_P(), V = ns.Get_V(), bg = ns.Get_BiasGyr(), ba = ns.Get_BiasAcc(), dbg = ns.Get_dBias_Gyr(), dba = ns.Get_dBias_Acc();
        const Quaterniond q = ns.Get_R();
        W.pose.insert(W.pose.end(), {P[0], P[1], P[2], q[0], q[1], q[2], q[3]});
        W.vel.insert(W.vel.end(), {V[0], V[1], V[2]});
        W.bias.insert(W.bias.end(), {bg[0], bg[1], bg[2], ba[0], ba[1], ba[2], dbg[0], dbg[1], dbg[2], dba[0], dba[1], dba[2]});
        W.kfFix.push_back(pKF->mnId == 0 ? 0x5 : 0x0);   // PR and Bias of keyframe 0 fixed, its V free (:667-685)
    }
    if (W.vKF.empty()) return false;
    W.P.n_kf_free = (int32_t)W.vKF.size();
    for (KeyFrame* pKF1 : W.vKF) {                                                                                // :700-770
        KeyFrame* pKF0 = pKF1->GetPrevKeyFrame();
        if (!pKF0 || !kfIdx.count(pKF0)) continue;
        const IMUPreintegrator& pre = pKF1->GetIMUPreInt();
        double info[81];
        if (!PRVInformation(pre.getCovPVPhi(), info)) continue;
        W.imuI.push_back(kfIdx[pKF0]);
        W.imuJ.push_back(kfIdx[pKF1]);
        W.meas.push_back(pre.getDeltaTime());
        for (double v : pre.getDeltaP()) W.meas.push_back(v);
        for (double v : pre.getDeltaV()) W.meas.push_back(v);
        for (double v : pre.getDeltaR()) W.meas.push_back(v);
        for (double v : pre.getJPBiasg()) W.meas.push_back(v);
        for (double v : pre.getJPBiasa()) W.meas.push_back(v);
        for (double v : pre.getJVBiasg()) W.meas.push_back(v);
        for (double v : pre.getJVBiasa()) W.meas.push_back(v);
        for (double v : pre.getJRBiasg()) W.meas.push_back(v);
        W.info.insert(W.info.end(), info, info + 81);
    }
    PackMapPoints(pMap->GetAllMapPoints(), kfIdx, W);
    W.P.variant = VBA_VARIANT_PRV_XYZ;
    W.P.algo = VBA_ALGO_LM;                                                                                       // :652
    Matrix3d Rcb; Vector3d tcb;
    ConfigParam::GetEigT_cb(Rcb, tcb);
    const Quaterniond qcb = MatrixToQuat(Rcb.data());
    for (int i = 0; i < 3; i++) { W.P.T_cb[i] = tcb[i]; W.P.g_w[i] = gw[i]; }
    for (int i = 0; i < 4; i++) W.P.T_cb[3 + i] = qcb[i];
    W.P.depth_min = 0.0;
    if (W.vMP.empty() || W.obsKF.empty()) return false;   // an empty graph: the reference's optimize() changes nothing
    FinishProblem(W);
    W.P.protocol = VBA_PROTO_SINGLE;
    W.P.robust = bRobust ? 1 : 0;
    W.P.its_stage1 = nIterations; W.P.its_stage2 = 0;                                                             // :836
    W.P.huber_vis = (double)(float)std::sqrt(5.99);                                                               // thHuber2D, :773
    W.P.kf_fix = W.kfFix.data();
    return true;
}

void Optimizer::GlobalBundleAdjustmentNavStatePRV(Map* pMap, const Vector3d& gw, int nIterations, bool* pbStopFlag,
                                                  const unsigned long nLoopKF, const bool bRobust) {
    PackedWindow& W = t_last;
    if (!PackGlobalBundleAdjustmentNavStatePRV(pMap, gw, nIterations, bRobust, W)) return;
    if (!RunGlobal(W, pbStopFlag, "GlobalBundleAdjustmentNavStatePRV")) return;
    for (size_t i = 0; i < W.vKF.size(); i++) {                                                                   // :842-893
        KeyFrame* pKF = W.vKF[i];
        NavState ns = pKF->GetNavState();
        ns.Set_Pos({{W.pose[7 * i], W.pose[7 * i + 1], W.pose[7 * i + 2]}});
        ns.Set_Rot({{W.pose[7 * i + 3], W.pose[7 * i + 4], W.pose[7 * i + 5], W.pose[7 * i + 6]}});
        ns.Set_Vel({{W.vel[3 * i], W.vel[3 * i + 1], W.vel[3 * i + 2]}});
        ns.Set_DeltaBiasGyr({{W.bias[12 * i + 6], W.bias[12 * i + 7], W.bias[12 * i + 8]}});
        ns.Set_DeltaBiasAcc({{W.bias[12 * i + 9], W.bias[12 * i + 10], W.bias[12 * i + 11]}});
        if (nLoopKF == 0) {
            pKF->SetNavState(ns);
            pKF->UpdatePoseFromNS();
        } else {   // the map kept growing meanwhile: results parked for LoopClosing to propagate (:873-891)
            KeyFrame tmp;
            tmp.SetNavState(ns);
            tmp.UpdatePoseFromNS();          // same float32 chain: Twb * Tbc, inverted
            pKF->mNavStateGBA = ns;
            pKF->mTcwGBA = tmp.GetPose();
            pKF->mnBAGlobalForKF = nLoopKF;
        }
    }
    WriteBackMapPoints(W, nLoopKF);
}

bool Optimizer::PackBundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, int nIterations,
                                     bool bRobust, PackedWindow& W) {
    W = PackedWindow();
    std::memset(&W.P, 0, sizeof W.P);
    std::map<KeyFrame*, int> kfIdx;
    std::vector<KeyFrame*> fixed;
    for (KeyFrame* pKF : vpKFs) {                                                                                 // :3400-3412
        if (pKF->isBad()) continue;
        if (pKF->mnId == 0) { fixed.push_back(pKF); continue; }   // vSE3->setFixed(pKF->mnId == 0)
        kfIdx[pKF] = (int)W.vKF.size();
        W.vKF.push_back(pKF);
    }
    W.P.n_kf_free = (int32_t)W.vKF.size();
    for (KeyFrame* k : fixed) { kfIdx[k] = (int)W.vKF.size(); W.vKF.push_back(k); }
    if (W.P.n_kf_free == 0) return false;
    for (KeyFrame* k : W.vKF) {  // Converter::toSE3Quat(pKF->GetPose())
        double R[9], t[3];
        k->GetRotation(R);
        k->GetTranslation(t);
        Quaterniond q = MatrixToQuat(R);
        if (q[3] < 0) for (auto& v : q) v = -v;
        W.pose.insert(W.pose.end(), {t[0], t[1], t[2], q[0], q[1], q[2], q[3]});
        W.vel.insert(W.vel.end(), {0, 0, 0});
        W.bias.insert(W.bias.end(), 12, 0.0);
    }
    PackMapPoints(vpMP, kfIdx, W);
    W.P.variant = VBA_VARIANT_SE3_XYZ;
    W.P.algo = VBA_ALGO_LM;                                                                                       // :3393
    W.P.T_cb[6] = 1.0;
    W.P.depth_min = 0.0;
    if (W.vMP.empty() || W.obsKF.empty()) return false;   // an empty graph: the reference's optimize() changes nothing
    FinishProblem(W);
    W.P.protocol = VBA_PROTO_SINGLE;
    W.P.robust = bRobust ? 1 : 0;
    W.P.its_stage1 = nIterations; W.P.its_stage2 = 0;                                                             // :3518
    W.P.huber_vis = (double)(float)std::sqrt(5.99);                                                               // thHuber2D, :3414
    return true;
}

void Optimizer::BundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, int nIterations,
                                 bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust) {
    PackedWindow& W = t_last;
    if (!PackBundleAdjustment(vpKFs, vpMP, nIterations, bRobust, W)) return;
    if (!RunGlobal(W, pbStopFlag, "BundleAdjustment")) return;
    for (int i = 0; i < W.P.n_kf_free; i++) {                                                                     // :3521-3541, Converter::toCvMat(SE3Quat)
        const Matrix3d R = QuatToMatrix({{W.pose[7 * i + 3], W.pose[7 * i + 4], W.pose[7 * i + 5], W.pose[7 * i + 6]}});
        Mat4f T{};
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) T[4 * r + c] = (float)R[3 * r + c];
            T[4 * r + 3] = (float)W.pose[7 * i + r];
        }
        T[15] = 1.0f;
        if (nLoopKF == 0) W.vKF[i]->SetPose(T);
        else { W.vKF[i]->mTcwGBA = T; W.vKF[i]->mnBAGlobalForKF = nLoopKF; }
    }
    // the fixed keyframe is written back too in the reference (its estimate did not move): a float32 round trip
    WriteBackMapPoints(W, nLoopKF);
}

// ------------------------------------------------------------------------------------------------
// Per-frame pose optimisation (src/Optimizer.cpp:3610-3835, 2046-2317, 1671-2044): set-up and write-back here, the four
// optimize(10) rounds + reclassification + marginals behind vba_pose_optimize.
// ------------------------------------------------------------------------------------------------
namespace {
void NavToArray(const NavState& ns, double* v) {
    const Vector3d P = ns.Get_P(), V = ns.Get_V(), bg = ns.Get_BiasGyr(), ba = ns.Get_BiasAcc(), dbg = ns.Get_dBias_Gyr(), dba = ns.Get_dBias_Acc();
    const Quaterniond q = ns.Get_R();
    const double a[22] = {P[0], P[1], P[2], q[0], q[1], q[2], q[3], V[0], V[1], V[2], bg[0], bg[1], bg[2], ba[0], ba[1], ba[2],
                          dbg[0], dbg[1], dbg[2], dba[0], dba[1], dba[2]};
    std::memcpy(v, a, sizeof a);
}
struct FrameObs {
    std::vector<double> pw, uv, w;
    std::vector<size_t> idx;
    std::vector<uint8_t> outl;
};
// monocular correspondences of a frame (:2137-2172); resets mvbOutlier like the reference does
void GatherFrame(Frame* f, FrameObs& o) {
    for (int i = 0; i < f->N; i++) {
        MapPoint* pMP = f->mvpMapPoints[i];
        if (!pMP || f->mvuRight[i] >= 0) continue;   // stereo: "stereo shouldn't in poseoptimization"
        f->mvbOutlier[i] = false;
        double P[3];
        pMP->GetWorldPos(P);
        const KeyPoint& kp = f->mvKeysUn[i];
        o.pw.insert(o.pw.end(), {P[0], P[1], P[2]});
        o.uv.push_back(kp.pt.x); o.uv.push_back(kp.pt.y);
        o.w.push_back(f->mvInvLevelSigma2[kp.octave]);
        o.idx.push_back((size_t)i);
    }
    o.outl.assign(o.idx.size() + 1, 0);
}
void FillCommon(vba_frame_problem& F, Frame* f, const FrameObs& o) {
    F.n_obs = (int32_t)o.idx.size();
    F.obs_pw = o.pw.data(); F.obs_uv = o.uv.data(); F.obs_w = o.w.data();
    F.K[0] = f->fx; F.K[1] = f->fy; F.K[2] = f->cx; F.K[3] = f->cy;
}
void FillImu(vba_frame_problem& F, const IMUPreintegrator& pre, const Vector3d& gw) {
    Matrix3d Rcb; Vector3d tcb;
    ConfigParam::GetEigT_cb(Rcb, tcb);
    const Quaterniond qcb = MatrixToQuat(Rcb.data());
    for (int i = 0; i < 3; i++) { F.T_cb[i] = tcb[i]; F.g_w[i] = gw[i]; }
    for (int i = 0; i < 4; i++) F.T_cb[3 + i] = qcb[i];
    double* m = F.imu_meas;
    m[0] = pre.getDeltaTime();
    std::memcpy(m + 1, pre.getDeltaP().data(), 24); std::memcpy(m + 4, pre.getDeltaV().data(), 24);
    std::memcpy(m + 7, pre.getDeltaR().data(), 72);
    std::memcpy(m + 16, pre.getJPBiasg().data(), 72); std::memcpy(m + 25, pre.getJPBiasa().data(), 72);
    std::memcpy(m + 34, pre.getJVBiasg().data(), 72); std::memcpy(m + 43, pre.getJVBiasa().data(), 72);
    std::memcpy(m + 52, pre.getJRBiasg().data(), 72);
    std::memcpy(F.imu_cov_pvphi, pre.getCovPVPhi().data(), sizeof F.imu_cov_pvphi);
    F.inv_bg_rw2 = 1.0 / IMUData::getGyrBiasRW2();
    F.inv_ba_rw2 = 1.0 / IMUData::getAccBiasRW2();
}
int RunFrame(vba_frame_problem& F, vba_frame_result& R, const char* who) {
    void* h = handle();
    if (!h) { std::cerr << who << ": no HIP device, pose optimisation skipped (the backend has no CPU path)" << std::endl; return -1; }
    vba_frame_problem* pf = &F;
    vba_frame_result* pr = &R;
    if (vba_pose_optimize(h, 1, &pf, &pr) != 0) { std::cerr << who << ": " << vba_last_error(h) << std::endl; return -1; }
    return 0;
}
NavState ArrayToNav(const NavState& base, const double* v) {   // ns_recov: PVR from the PVR vertex, delta biases from the bias vertex
    NavState ns = base;
    ns.Set_Pos({{v[0], v[1], v[2]}});
    ns.Set_Rot({{v[3], v[4], v[5], v[6]}});
    ns.Set_Vel({{v[7], v[8], v[9]}});
    ns.Set_DeltaBiasGyr({{v[16], v[17], v[18]}});
    ns.Set_DeltaBiasAcc({{v[19], v[20], v[21]}});
    return ns;
}
}  // namespace

int Optimizer::PoseOptimization(Frame* pFrame) {                                                                   // :3610-3835
    FrameObs o;
    GatherFrame(pFrame, o);
    if (o.idx.size() < 3) return 0;                                                                               // :3726-3727
    vba_frame_problem F;
    std::memset(&F, 0, sizeof F);
    vba_frame_result R;
    std::memset(&R, 0, sizeof R);
    F.last_is_frame = VBA_FRAME_VISION;
    FillCommon(F, pFrame, o);
    {   // Converter::toSE3Quat(pFrame->mTcw)
        double Rm[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rm[3 * i + j] = pFrame->mTcw[4 * i + j];
        Quaterniond q = MatrixToQuat(Rm);
        if (q[3] < 0) for (auto& v : q) v = -v;
        for (int i = 0; i < 3; i++) F.nav[i] = pFrame->mTcw[4 * i + 3];
        for (int i = 0; i < 4; i++) F.nav[3 + i] = q[i];
    }
    F.T_cb[6] = 1.0;
    R.outlier = o.outl.data();
    if (RunFrame(F, R, "PoseOptimization") != 0) return 0;
    for (size_t k = 0; k < o.idx.size(); k++) pFrame->mvbOutlier[o.idx[k]] = o.outl[k] != 0;
    const Matrix3d Rr = QuatToMatrix({{F.nav[3], F.nav[4], F.nav[5], F.nav[6]}});                                  // Converter::toCvMat(SE3quat_recov)
    Mat4f T{};
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[4 * r + c] = (float)Rr[3 * r + c];
        T[4 * r + 3] = (float)F.nav[r];
    }
    T[15] = 1.0f;
    pFrame->SetPose(T);
    return R.n_inliers;
}

int Optimizer::PoseOptimization(Frame* pFrame, KeyFrame* pLastKF, const IMUPreintegrator& imupreint, const Vector3d& gw,
                                const bool& bComputeMarg) {                                                       // :2046-2317
    FrameObs o;
    GatherFrame(pFrame, o);
    if (o.idx.size() < 3) return 0;
    vba_frame_problem F;
    std::memset(&F, 0, sizeof F);
    vba_frame_result R;
    std::memset(&R, 0, sizeof R);
    F.last_is_frame = VBA_FRAME_KF;
    F.compute_marg = bComputeMarg ? 1 : 0;
    FillCommon(F, pFrame, o);
    FillImu(F, imupreint, gw);
    NavToArray(pFrame->GetNavState(), F.nav);
    NavToArray(pLastKF->GetNavState(), F.nav_last);
    R.outlier = o.outl.data();
    if (RunFrame(F, R, "PoseOptimization") != 0) return 0;
    for (size_t k = 0; k < o.idx.size(); k++) pFrame->mvbOutlier[o.idx[k]] = o.outl[k] != 0;
    const NavState ns = ArrayToNav(pFrame->GetNavState(), F.nav);
    pFrame->SetNavState(ns);
    pFrame->UpdatePoseFromNS();
    if (bComputeMarg) {
        std::memcpy(pFrame->mMargCovInv.data(), R.marg_cov_inv, sizeof R.marg_cov_inv);
        pFrame->mNavStatePrior = ns;
    }
    return R.n_inliers;
}

int Optimizer::PoseOptimization(Frame* pFrame, Frame* pLastFrame, const IMUPreintegrator& imupreint, const Vector3d& gw,
                                const bool& bComputeMarg) {                                                       // :1671-2044
    FrameObs o, ol;
    GatherFrame(pFrame, o);
    GatherFrame(pLastFrame, ol);
    if (o.idx.size() < 3) return 0;
    vba_frame_problem F;
    std::memset(&F, 0, sizeof F);
    vba_frame_result R;
    std::memset(&R, 0, sizeof R);
    F.last_is_frame = VBA_FRAME_FRAME;
    F.compute_marg = bComputeMarg ? 1 : 0;
    FillCommon(F, pFrame, o);
    F.n_obs_last = (int32_t)ol.idx.size();
    F.last_pw = ol.pw.data(); F.last_uv = ol.uv.data(); F.last_w = ol.w.data();
    FillImu(F, imupreint, gw);
    NavToArray(pFrame->GetNavState(), F.nav);
    NavToArray(pLastFrame->GetNavState(), F.nav_last);
    NavToArray(pLastFrame->mNavStatePrior, F.prior_nav);
    std::memcpy(F.prior_info, pLastFrame->mMargCovInv.data(), sizeof F.prior_info);
    R.outlier = o.outl.data();
    R.outlier_last = ol.outl.data();
    if (RunFrame(F, R, "PoseOptimization") != 0) return 0;
    for (size_t k = 0; k < o.idx.size(); k++) pFrame->mvbOutlier[o.idx[k]] = o.outl[k] != 0;
    for (size_t k = 0; k < ol.idx.size(); k++) pLastFrame->mvbOutlier[ol.idx[k]] = ol.outl[k] != 0;
    const NavState ns = ArrayToNav(pFrame->GetNavState(), F.nav);
    pFrame->SetNavState(ns);
    pFrame->UpdatePoseFromNS();
    if (bComputeMarg) {
        std::memcpy(pFrame->mMargCovInv.data(), R.marg_cov_inv, sizeof R.marg_cov_inv);
        pFrame->mNavStatePrior = ns;
    }
    return R.n_inliers;
}

void Optimizer::GlobalBundleAdjustment(Map* pMap, int nIterations, bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust) {
    BundleAdjustment(pMap->GetAllKeyFrames(), pMap->GetAllMapPoints(), nIterations, pbStopFlag, nLoopKF, bRobust);   // :3346-3354
}

}  // namespace ORB_SLAM2
