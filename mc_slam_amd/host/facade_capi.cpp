// facade_capi.cpp -- C entry points that let the test-suite build a KeyFrame / MapPoint map from flat arrays,
// call the C++ facade (Optimizer.h) and read the map back.  Test harness of the facade, not part of the
// product ABI (that is include/vislam_ba.h).
#include <cstring>
#include <map>
#include <memory>

#include "Optimizer.h"

using namespace ORB_SLAM2;

struct FcMap {
    Map map;
    LocalMapping lm;
    std::map<long, std::unique_ptr<KeyFrame>> kfs;
    std::map<long, std::unique_ptr<MapPoint>> mps;
    std::map<long, std::unique_ptr<Frame>> frames;
};

extern "C" {

void* fc_create() { return new FcMap(); }
void fc_destroy(void* m) { delete reinterpret_cast<FcMap*>(m); }
void fc_set_device(int dev) { Optimizer::Device = dev; }
void fc_set_tbc(const double* R9, const double* p3) {
    Matrix3d R; Vector3d p;
    std::memcpy(R.data(), R9, 72); std::memcpy(p.data(), p3, 24);
    ConfigParam::SetTbc(R, p);
}
// nav: P(3) q(4) V(3) bg(3) ba(3) dbg(3) dba(3)
int fc_add_keyframe(void* m, long id, const double* nav, const double* K, long prev_id, int bad) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    std::unique_ptr<KeyFrame> k(new KeyFrame());
    k->mnId = id;
    if ((long unsigned)id + 1 > KeyFrame::nNextId) KeyFrame::nNextId = id + 1;
    k->fx = (float)K[0]; k->fy = (float)K[1]; k->cx = (float)K[2]; k->cy = (float)K[3];
    k->mNavState.Set_Pos({{nav[0], nav[1], nav[2]}});
    k->mNavState.Set_Rot({{nav[3], nav[4], nav[5], nav[6]}});
    k->mNavState.Set_Vel({{nav[7], nav[8], nav[9]}});
    k->mNavState.Set_BiasGyr({{nav[10], nav[11], nav[12]}});
    k->mNavState.Set_BiasAcc({{nav[13], nav[14], nav[15]}});
    k->mNavState.Set_DeltaBiasGyr({{nav[16], nav[17], nav[18]}});
    k->mNavState.Set_DeltaBiasAcc({{nav[19], nav[20], nav[21]}});
    k->mvInvLevelSigma2.resize(8);
    for (int l = 0; l < 8; l++) k->mvInvLevelSigma2[l] = 1.0f / (float)std::pow(1.2, 2 * l);  // ORBextractor.cpp:427-441
    k->mbBad = bad != 0;
    if (prev_id >= 0 && M->kfs.count(prev_id)) k->mpPrevKeyFrame = M->kfs[prev_id].get();
    k->UpdatePoseFromNS();
    M->kfs[id] = std::move(k);
    return 0;
}
int fc_set_pose_tcw(void* m, long id, const float* T16) {  // vision-only path: float32 T_cw given directly
    FcMap* M = reinterpret_cast<FcMap*>(m);
    Mat4f T; std::memcpy(T.data(), T16, 64);
    M->kfs.at(id)->SetPose(T);
    return 0;
}
int fc_set_covisible(void* m, long id, const long* ids, int n) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    for (int i = 0; i < n; i++) M->kfs.at(id)->mvpOrderedConnectedKeyFrames.push_back(M->kfs.at(ids[i]).get());
    return 0;
}
int fc_set_preint(void* m, long id, const double* meas61, const double* cov81) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    IMUPreintegrator& P = M->kfs.at(id)->mIMUPreInt;
    P._delta_time = meas61[0];
    std::memcpy(P._delta_P.data(), meas61 + 1, 24); std::memcpy(P._delta_V.data(), meas61 + 4, 24);
    std::memcpy(P._delta_R.data(), meas61 + 7, 72);
    std::memcpy(P._J_P_Biasg.data(), meas61 + 16, 72); std::memcpy(P._J_P_Biasa.data(), meas61 + 25, 72);
    std::memcpy(P._J_V_Biasg.data(), meas61 + 34, 72); std::memcpy(P._J_V_Biasa.data(), meas61 + 43, 72);
    std::memcpy(P._J_R_Biasg.data(), meas61 + 52, 72);
    std::memcpy(P._cov_P_V_Phi.data(), cov81, 648);
    return 0;
}
int fc_add_mappoint(void* m, long id, const float* Pw, long ref_kf) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    std::unique_ptr<MapPoint> p(new MapPoint());
    p->mnId = id;
    if ((long unsigned)id + 1 > MapPoint::nNextId) MapPoint::nNextId = id + 1;
    p->SetWorldPos(Pw);
    p->mpRefKF = M->kfs.at(ref_kf).get();
    M->mps[id] = std::move(p);
    return 0;
}
int fc_add_observation(void* m, long mp, long kf, float u, float v, int octave) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    KeyFrame* k = M->kfs.at(kf).get();
    MapPoint* p = M->mps.at(mp).get();
    KeyPoint kp; kp.pt.x = u; kp.pt.y = v; kp.octave = octave;
    k->mvKeysUn.push_back(kp);
    k->mvuRight.push_back(-1.0f);
    k->mvpMapPoints.push_back(p);
    p->mObservations[k] = k->mvKeysUn.size() - 1;
    return 0;
}
static std::list<KeyFrame*> window(FcMap* M, const long* ids, int n) {
    std::list<KeyFrame*> l;
    for (int i = 0; i < n; i++) l.push_back(M->kfs.at(ids[i]).get());
    return l;
}
// LocalBAPRVIDP with the CALLER'S flag: `stop` is a bool (one byte) that another thread may raise while the call runs, exactly as
// LocalMapping::InterruptBA does with mbAbortBA (src/LocalMapping.cpp:1769-1772)
int fc_local_ba_prvidp_flag(void* m, const long* ids, int n, const double* gw, bool* stop) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    std::list<KeyFrame*> l = window(M, ids, n);
    const Vector3d g{{gw[0], gw[1], gw[2]}};
    Optimizer::LocalBAPRVIDP(l.back(), l, stop, &M->map, g, &M->lm);
    return 0;
}
// wall-clock split of this thread's last LocalBAPRVIDP: extraction, solve, erase + write-back, total (ms)
void fc_last_timing(double* out4) {
    const FacadeTiming& t = Optimizer::LastTiming();
    out4[0] = t.extract_ms; out4[1] = t.solve_ms; out4[2] = t.writeback_ms; out4[3] = t.total_ms;
}
// mode 0: full LocalBAPRVIDP; 1: extraction only (no GPU needed)
int fc_local_ba_prvidp(void* m, const long* ids, int n, const double* gw, int stop, int mode) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    std::list<KeyFrame*> l = window(M, ids, n);
    bool bstop = stop != 0;
    const Vector3d g{{gw[0], gw[1], gw[2]}};
    if (mode == 1) {
        return Optimizer::PackLocalBAPRVIDP(l.back(), l, g, Optimizer::LastWindowMutable()) ? 0 : -1;
    }
    Optimizer::LocalBAPRVIDP(l.back(), l, &bstop, &M->map, g, &M->lm);
    return 0;
}
// LocalBundleAdjustmentNavStatePRV (VI window, XYZ landmarks, LM); mode 1: extraction only
int fc_local_ba_prv_xyz(void* m, const long* ids, int n, const double* gw, int stop, int mode) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    std::list<KeyFrame*> l = window(M, ids, n);
    bool bstop = stop != 0;
    const Vector3d g{{gw[0], gw[1], gw[2]}};
    if (mode == 1) return Optimizer::PackLocalBundleAdjustmentNavStatePRV(l.back(), l, g, Optimizer::LastWindowMutable()) ? 0 : -1;
    Optimizer::LocalBundleAdjustmentNavStatePRV(l.back(), l, &bstop, &M->map, g, &M->lm);
    return 0;
}
// LocalBundleAdjustment over an explicit keyframe list (include/Optimizer.h:57-59); mode 1: extraction only
int fc_local_ba_vision_list(void* m, const long* ids, int n, int stop, int mode) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    std::list<KeyFrame*> l = window(M, ids, n);
    bool bstop = stop != 0;
    if (mode == 1) return Optimizer::PackLocalBundleAdjustment(l.back(), &l, Optimizer::LastWindowMutable()) ? 0 : -1;
    Optimizer::LocalBundleAdjustment(l.back(), l, &bstop, &M->map, &M->lm);
    return 0;
}
int fc_local_ba_vision(void* m, long cur, int stop) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    bool bstop = stop != 0;
    Optimizer::LocalBundleAdjustment(M->kfs.at(cur).get(), &bstop, &M->map, &M->lm);
    return 0;
}
static void fill_map(FcMap* M) {
    M->map.mspKeyFrames.clear(); M->map.mspMapPoints.clear();
    for (auto& k : M->kfs) M->map.mspKeyFrames.push_back(k.second.get());
    for (auto& p : M->mps) M->map.mspMapPoints.push_back(p.second.get());
}
// mode 0: run; 1: extraction only (no GPU needed)
int fc_global_ba_prv(void* m, const double* gw, int nIterations, long nLoopKF, int bRobust, int stop, int mode) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    fill_map(M);
    bool bstop = stop != 0;
    const Vector3d g{{gw[0], gw[1], gw[2]}};
    if (mode == 1) return Optimizer::PackGlobalBundleAdjustmentNavStatePRV(&M->map, g, nIterations, bRobust != 0, Optimizer::LastWindowMutable()) ? 0 : -1;
    Optimizer::GlobalBundleAdjustmentNavStatePRV(&M->map, g, nIterations, &bstop, (unsigned long)nLoopKF, bRobust != 0);
    return 0;
}
int fc_global_ba_vision(void* m, int nIterations, long nLoopKF, int bRobust, int stop, int mode) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    fill_map(M);
    bool bstop = stop != 0;
    if (mode == 1) return Optimizer::PackBundleAdjustment(M->map.GetAllKeyFrames(), M->map.GetAllMapPoints(), nIterations, bRobust != 0, Optimizer::LastWindowMutable()) ? 0 : -1;
    Optimizer::GlobalBundleAdjustment(&M->map, nIterations, &bstop, (unsigned long)nLoopKF, bRobust != 0);
    return 0;
}
int fc_get_gba(void* m, long id, double* nav22, float* T16, long* nLoop) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    KeyFrame* k = M->kfs.at(id).get();
    const NavState& ns = k->mNavStateGBA;
    const Vector3d P = ns.Get_P(), V = ns.Get_V(), bg = ns.Get_BiasGyr(), ba = ns.Get_BiasAcc(), dbg = ns.Get_dBias_Gyr(), dba = ns.Get_dBias_Acc();
    const Quaterniond q = ns.Get_R();
    const double v[22] = {P[0], P[1], P[2], q[0], q[1], q[2], q[3], V[0], V[1], V[2], bg[0], bg[1], bg[2], ba[0], ba[1], ba[2],
                          dbg[0], dbg[1], dbg[2], dba[0], dba[1], dba[2]};
    std::memcpy(nav22, v, sizeof v);
    std::memcpy(T16, k->mTcwGBA.data(), 64);
    *nLoop = (long)k->mnBAGlobalForKF;
    return 0;
}
int fc_get_mappoint_gba(void* m, long id, float* Pw, long* nLoop) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    MapPoint* p = M->mps.at(id).get();
    std::memcpy(Pw, p->mPosGBA, 12);
    *nLoop = (long)p->mnBAGlobalForKF;
    return 0;
}
// ---- frames (per-frame pose optimisation) ----
static void set_nav(NavState& ns, const double* nav) {
    ns.Set_Pos({{nav[0], nav[1], nav[2]}});
    ns.Set_Rot({{nav[3], nav[4], nav[5], nav[6]}});
    ns.Set_Vel({{nav[7], nav[8], nav[9]}});
    ns.Set_BiasGyr({{nav[10], nav[11], nav[12]}});
    ns.Set_BiasAcc({{nav[13], nav[14], nav[15]}});
    ns.Set_DeltaBiasGyr({{nav[16], nav[17], nav[18]}});
    ns.Set_DeltaBiasAcc({{nav[19], nav[20], nav[21]}});
}
static void get_nav(const NavState& ns, double* nav22) {
    const Vector3d P = ns.Get_P(), V = ns.Get_V(), bg = ns.Get_BiasGyr(), ba = ns.Get_BiasAcc(), dbg = ns.Get_dBias_Gyr(), dba = ns.Get_dBias_Acc();
    const Quaterniond q = ns.Get_R();
    const double v[22] = {P[0], P[1], P[2], q[0], q[1], q[2], q[3], V[0], V[1], V[2], bg[0], bg[1], bg[2], ba[0], ba[1], ba[2],
                          dbg[0], dbg[1], dbg[2], dba[0], dba[1], dba[2]};
    std::memcpy(nav22, v, sizeof v);
}
int fc_add_frame(void* m, long id, const double* nav22, const double* K) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    std::unique_ptr<Frame> f(new Frame());
    f->fx = (float)K[0]; f->fy = (float)K[1]; f->cx = (float)K[2]; f->cy = (float)K[3];
    f->mvInvLevelSigma2.resize(8);
    for (int l = 0; l < 8; l++) f->mvInvLevelSigma2[l] = 1.0f / (float)std::pow(1.2, 2 * l);
    set_nav(f->mNavState, nav22);
    f->UpdatePoseFromNS();
    M->frames[id] = std::move(f);
    return 0;
}
int fc_frame_set_tcw(void* m, long id, const float* T16) {
    Mat4f T; std::memcpy(T.data(), T16, 64);
    reinterpret_cast<FcMap*>(m)->frames.at(id)->SetPose(T);
    return 0;
}
int fc_frame_add_obs(void* m, long frame, long mp, float u, float v, int octave) {   // mp < 0: an unmatched keypoint
    FcMap* M = reinterpret_cast<FcMap*>(m);
    Frame* f = M->frames.at(frame).get();
    KeyPoint kp; kp.pt.x = u; kp.pt.y = v; kp.octave = octave;
    f->mvKeysUn.push_back(kp);
    f->mvuRight.push_back(-1.0f);
    f->mvpMapPoints.push_back(mp >= 0 ? M->mps.at(mp).get() : nullptr);
    f->mvbOutlier.push_back(true);   // whatever the tracker left there: the optimiser resets matched ones
    f->N = (int)f->mvKeysUn.size();
    return 0;
}
int fc_frame_set_prior(void* m, long id, const double* prior_nav22, const double* info225) {
    Frame* f = reinterpret_cast<FcMap*>(m)->frames.at(id).get();
    set_nav(f->mNavStatePrior, prior_nav22);
    std::memcpy(f->mMargCovInv.data(), info225, 225 * 8);
    return 0;
}
static IMUPreintegrator make_preint(const double* meas61, const double* cov81) {
    IMUPreintegrator P;
    P._delta_time = meas61[0];
    std::memcpy(P._delta_P.data(), meas61 + 1, 24); std::memcpy(P._delta_V.data(), meas61 + 4, 24);
    std::memcpy(P._delta_R.data(), meas61 + 7, 72);
    std::memcpy(P._J_P_Biasg.data(), meas61 + 16, 72); std::memcpy(P._J_P_Biasa.data(), meas61 + 25, 72);
    std::memcpy(P._J_V_Biasg.data(), meas61 + 34, 72); std::memcpy(P._J_V_Biasa.data(), meas61 + 43, 72);
    std::memcpy(P._J_R_Biasg.data(), meas61 + 52, 72);
    std::memcpy(P._cov_P_V_Phi.data(), cov81, 648);
    return P;
}
// kind 2: PoseOptimization(Frame*); 0: (Frame*, KeyFrame* last, ...); 1: (Frame*, Frame* last, ...).  Returns the inlier count.
int fc_pose_optimization(void* m, int kind, long frame, long last, const double* meas61, const double* cov81, const double* gw, int marg) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    Frame* f = M->frames.at(frame).get();
    if (kind == 2) return Optimizer::PoseOptimization(f);
    const IMUPreintegrator pre = make_preint(meas61, cov81);
    const Vector3d g{{gw[0], gw[1], gw[2]}};
    if (kind == 0) return Optimizer::PoseOptimization(f, M->kfs.at(last).get(), pre, g, marg != 0);
    return Optimizer::PoseOptimization(f, M->frames.at(last).get(), pre, g, marg != 0);
}
int fc_frame_get(void* m, long id, double* nav22, float* T16, double* marg225, double* prior22, unsigned char* outlier, int cap) {
    Frame* f = reinterpret_cast<FcMap*>(m)->frames.at(id).get();
    get_nav(f->GetNavState(), nav22);
    std::memcpy(T16, f->mTcw.data(), 64);
    std::memcpy(marg225, f->mMargCovInv.data(), 225 * 8);
    get_nav(f->mNavStatePrior, prior22);
    for (int i = 0; i < f->N && i < cap; i++) outlier[i] = f->mvbOutlier[i] ? 1 : 0;
    return f->N;
}
int fc_get_nav(void* m, long id, double* nav22, float* T16) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    KeyFrame* k = M->kfs.at(id).get();
    const NavState& ns = k->GetNavState();
    const Vector3d P = ns.Get_P(), V = ns.Get_V(), bg = ns.Get_BiasGyr(), ba = ns.Get_BiasAcc(), dbg = ns.Get_dBias_Gyr(), dba = ns.Get_dBias_Acc();
    const Quaterniond q = ns.Get_R();
    const double v[22] = {P[0], P[1], P[2], q[0], q[1], q[2], q[3], V[0], V[1], V[2], bg[0], bg[1], bg[2], ba[0], ba[1], ba[2],
                          dbg[0], dbg[1], dbg[2], dba[0], dba[1], dba[2]};
    std::memcpy(nav22, v, sizeof v);
    std::memcpy(T16, k->GetPose().data(), 64);
    return 0;
}
int fc_get_mappoint(void* m, long id, float* Pw, int* n_obs, int* n_updates) {
    FcMap* M = reinterpret_cast<FcMap*>(m);
    MapPoint* p = M->mps.at(id).get();
    std::memcpy(Pw, p->mWorldPos, 12);
    *n_obs = (int)p->mObservations.size();
    *n_updates = p->nNormalUpdates;
    return 0;
}
int fc_map_updated(void* m) { return reinterpret_cast<FcMap*>(m)->lm.mbMapUpdateFlagForTracking ? 1 : 0; }
// last packed window (what the facade handed / would hand to vba_solve)
const vba_problem* fc_last_problem() { return &Optimizer::LastWindow().P; }
int fc_last_mp_ids(long* out, int cap) {  // mnId of the MapPoint behind every landmark row of the last window
    const PackedWindow& W = Optimizer::LastWindow();
    const int n = (int)W.vMP.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = (long)W.vMP[i]->mnId;
    return n;
}
int fc_last_kf_ids(long* out, int cap) {
    const PackedWindow& W = Optimizer::LastWindow();
    const int n = (int)W.vKF.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = (long)W.vKF[i]->mnId;
    return n;
}
const vba_result* fc_last_result() { return &Optimizer::LastWindow().R; }

}  // extern "C"
