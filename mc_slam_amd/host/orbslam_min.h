// orbslam_min.h -- the slice of the reference's host data model that Optimizer::LocalBAPRVIDP /
// LocalBundleAdjustment read and write (SURVEY.md section 8b lists the surface).  Same class, member and
// method names as mc275/MC_SLAM (include/KeyFrame.h, include/MapPoint.h, src/IMU/NavState.h,
// src/IMU/IMUPreintegrator.h, src/IMU/configparam.h, src/IMU/imudata.h), minus OpenCV / Eigen (absent from
// this image): cv::Mat poses become float[16] row-major (they ARE float32 in the reference, CV_32F),
// Eigen vectors become small POD arrays.  It exists so that the facade in Optimizer.cpp compiles against the
// interface it is meant to drop into, and so that tests can drive it; it is not a SLAM system.
#pragma once
#include <array>
#include <cmath>
#include <cstddef>
#include <list>
#include <map>
#include <mutex>
#include <vector>

namespace ORB_SLAM2 {

typedef std::array<double, 3> Vector3d;
typedef std::array<double, 4> Quaterniond;  // x y z w (Eigen coefficient order)
typedef std::array<double, 9> Matrix3d;     // row-major
typedef std::array<float, 16> Mat4f;        // cv::Mat CV_32F 4x4, row-major

inline Matrix3d QuatToMatrix(const Quaterniond& q) {  // Eigen::Quaterniond::toRotationMatrix
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    return {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
}

// src/IMU/NavState.h:124-138
class NavState {
public:
    Vector3d Get_P() const { return _P; }
    Vector3d Get_V() const { return _V; }
    Quaterniond Get_R() const { return _R; }  // Sophus::SO3 (unit quaternion storage)
    Matrix3d Get_RotMatrix() const { return QuatToMatrix(_R); }
    Vector3d Get_BiasGyr() const { return _BiasGyr; }
    Vector3d Get_BiasAcc() const { return _BiasAcc; }
    Vector3d Get_dBias_Gyr() const { return _dBias_g; }
    Vector3d Get_dBias_Acc() const { return _dBias_a; }
    void Set_Pos(const Vector3d& v) { _P = v; }
    void Set_Vel(const Vector3d& v) { _V = v; }
    void Set_Rot(const Quaterniond& q) { _R = q; }
    void Set_BiasGyr(const Vector3d& v) { _BiasGyr = v; }
    void Set_BiasAcc(const Vector3d& v) { _BiasAcc = v; }
    void Set_DeltaBiasGyr(const Vector3d& v) { _dBias_g = v; }
    void Set_DeltaBiasAcc(const Vector3d& v) { _dBias_a = v; }

private:
    Vector3d _P{{0, 0, 0}}, _V{{0, 0, 0}};
    Quaterniond _R{{0, 0, 0, 1}};
    Vector3d _BiasGyr{{0, 0, 0}}, _BiasAcc{{0, 0, 0}}, _dBias_g{{0, 0, 0}}, _dBias_a{{0, 0, 0}};
};

// src/IMU/IMUPreintegrator.h:179-195
class IMUPreintegrator {
public:
    double getDeltaTime() const { return _delta_time; }
    const Vector3d& getDeltaP() const { return _delta_P; }
    const Vector3d& getDeltaV() const { return _delta_V; }
    const Matrix3d& getDeltaR() const { return _delta_R; }
    const Matrix3d& getJPBiasg() const { return _J_P_Biasg; }
    const Matrix3d& getJPBiasa() const { return _J_P_Biasa; }
    const Matrix3d& getJVBiasg() const { return _J_V_Biasg; }
    const Matrix3d& getJVBiasa() const { return _J_V_Biasa; }
    const Matrix3d& getJRBiasg() const { return _J_R_Biasg; }
    const std::array<double, 81>& getCovPVPhi() const { return _cov_P_V_Phi; }
    double _delta_time = 0;
    Vector3d _delta_P{{0, 0, 0}}, _delta_V{{0, 0, 0}};
    Matrix3d _delta_R{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    Matrix3d _J_P_Biasg{}, _J_P_Biasa{}, _J_V_Biasg{}, _J_V_Biasa{}, _J_R_Biasg{};
    std::array<double, 81> _cov_P_V_Phi{};
};

// src/IMU/imudata.cpp:25-26
struct IMUData {
    static double getGyrBiasRW2() { return 2.0e-5 * 2.0e-5; }
    static double getAccBiasRW2() { return 5.0e-3 * 5.0e-3; }
};

// src/IMU/configparam.cpp:55-71 (T_bc from the settings file, rotation re-normalised; T_cb = T_bc^-1)
struct ConfigParam {
    static Matrix3d& Rbc() { static Matrix3d R{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; return R; }
    static Vector3d& Pbc() { static Vector3d p{{0, 0, 0}}; return p; }
    static void SetTbc(const Matrix3d& R, const Vector3d& p) { Rbc() = R; Pbc() = p; }
    static void GetEigT_cb(Matrix3d& Rcb, Vector3d& tcb) {
        const Matrix3d& R = Rbc();
        const Vector3d& p = Pbc();
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rcb[3 * i + j] = R[3 * j + i];
        for (int i = 0; i < 3; i++) tcb[i] = -(Rcb[3 * i] * p[0] + Rcb[3 * i + 1] * p[1] + Rcb[3 * i + 2] * p[2]);
    }
};

struct KeyPoint {  // cv::KeyPoint: pt.x, pt.y are float, octave int
    struct { float x, y; } pt;
    int octave;
};

class KeyFrame;
class MapPoint;

// src/MapPoint.cpp:19-22
struct cmpKeyFrameId {
    bool operator()(const KeyFrame* a, const KeyFrame* b) const;
};
typedef std::map<KeyFrame*, size_t, cmpKeyFrameId> mapMapPointObs;

class Map {
public:
    std::mutex mMutexMapUpdate;  // include/Map.h:72
    std::vector<KeyFrame*> GetAllKeyFrames() { return mspKeyFrames; }    // include/Map.h:44 (a std::set there: id order)
    std::vector<MapPoint*> GetAllMapPoints() { return mspMapPoints; }    // include/Map.h:45
    std::vector<KeyFrame*> mspKeyFrames;
    std::vector<MapPoint*> mspMapPoints;
};

class LocalMapping {
public:
    void SetMapUpdateFlagInTracking(bool b) { mbMapUpdateFlagForTracking = b; }
    bool mbMapUpdateFlagForTracking = false;
};

class KeyFrame {
public:
    long unsigned int mnId = 0;
    static long unsigned int nNextId;
    long unsigned int mnBALocalForKF = (long unsigned int)-1, mnBAFixedForKF = (long unsigned int)-1;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    std::vector<KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;          // negative for monocular points
    std::vector<float> mvInvLevelSigma2;

    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    KeyFrame* GetPrevKeyFrame() { return mpPrevKeyFrame; }
    const NavState& GetNavState() { return mNavState; }
    void SetNavState(const NavState& ns) { mNavState = ns; }
    const IMUPreintegrator& GetIMUPreInt() { return mIMUPreInt; }
    bool isBad() { return mbBad; }
    // float32 camera pose T_cw, kept in sync by UpdatePoseFromNS / SetPose (src/KeyFrame.cpp:96-114)
    const Mat4f& GetPose() const { return Tcw; }
    void SetPose(const Mat4f& T) { Tcw = T; }
    void GetRotation(double R[9]) const { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = Tcw[4 * i + j]; }
    void GetTranslation(double t[3]) const { for (int i = 0; i < 3; i++) t[i] = Tcw[4 * i + 3]; }
    void GetCameraCenter(double c[3]) const {  // -R^T t in float32 arithmetic like cv::Mat (KeyFrame::SetPose)
        for (int i = 0; i < 3; i++) {
            float s = 0;
            for (int k = 0; k < 3; k++) s += Tcw[4 * k + i] * Tcw[4 * k + 3];
            c[i] = -s;
        }
    }
    void SetNavStatePos(const Vector3d& v) { mNavState.Set_Pos(v); }
    void SetNavStateRot(const Quaterniond& q) { mNavState.Set_Rot(q); }
    void SetNavStateVel(const Vector3d& v) { mNavState.Set_Vel(v); }
    void SetNavStateDeltaBg(const Vector3d& v) { mNavState.Set_DeltaBiasGyr(v); }
    void SetNavStateDeltaBa(const Vector3d& v) { mNavState.Set_DeltaBiasAcc(v); }
    void UpdatePoseFromNS() {  // src/KeyFrame.cpp:96-114 with ConfigParam::GetMatTbc(), float32 like cv::Mat
        const Matrix3d Rwb = mNavState.Get_RotMatrix();
        const Vector3d Pwb = mNavState.Get_P();
        float Rwbf[9], Rbcf[9], Pbcf[3], Pwbf[3];
        for (int i = 0; i < 9; i++) { Rwbf[i] = (float)Rwb[i]; Rbcf[i] = (float)ConfigParam::Rbc()[i]; }
        for (int i = 0; i < 3; i++) { Pbcf[i] = (float)ConfigParam::Pbc()[i]; Pwbf[i] = (float)Pwb[i]; }
        float Rwc[9], Pwc[3];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
                float s = 0;
                for (int k = 0; k < 3; k++) s += Rwbf[3 * i + k] * Rbcf[3 * k + j];
                Rwc[3 * i + j] = s;
            }
            float s = 0;
            for (int k = 0; k < 3; k++) s += Rwbf[3 * i + k] * Pbcf[k];
            Pwc[i] = s + Pwbf[i];
        }
        Mat4f T{};
        for (int i = 0; i < 3; i++) {
            float s = 0;
            for (int j = 0; j < 3; j++) { T[4 * i + j] = Rwc[3 * j + i]; s += Rwc[3 * j + i] * Pwc[j]; }
            T[4 * i + 3] = -s;
        }
        T[15] = 1.0f;
        SetPose(T);
    }
    void EraseMapPointMatch(MapPoint* pMP);   // src/KeyFrame.cpp:573-579: through pMP->GetIndexInKeyFrame(this) (below MapPoint)
    std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { return mvpOrderedConnectedKeyFrames; }

    // state (public here: the test harness fills it)
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<KeyFrame*> mvpOrderedConnectedKeyFrames;
    KeyFrame* mpPrevKeyFrame = nullptr;
    NavState mNavState;
    IMUPreintegrator mIMUPreInt;
    bool mbBad = false;
    Mat4f Tcw{};
    // results of a global BA that runs while the map keeps growing (include/KeyFrame.h:180-186)
    NavState mNavStateGBA;
    Mat4f mTcwGBA{};
    long unsigned int mnBAGlobalForKF = 0;
};

class MapPoint {
public:
    long unsigned int mnId = 0;
    static long unsigned int nNextId;
    long unsigned int mnBALocalForKF = (long unsigned int)-1;
    bool isBad() { return mbBad; }
    void GetWorldPos(double P[3]) const { for (int i = 0; i < 3; i++) P[i] = mWorldPos[i]; }  // float32 -> double (Converter::toVector3d)
    void SetWorldPos(const float P[3]) { for (int i = 0; i < 3; i++) mWorldPos[i] = P[i]; }
    mapMapPointObs GetObservations() { return mObservations; }
    KeyFrame* GetReferenceKeyFrame() { return mpRefKF; }
    void EraseObservation(KeyFrame* pKF) { mObservations.erase(pKF); }
    int GetIndexInKeyFrame(KeyFrame* pKF) {   // src/MapPoint.cpp: the keypoint index of this point in pKF, -1 if pKF does not observe it
        const auto it = mObservations.find(pKF);
        return it == mObservations.end() ? -1 : (int)it->second;
    }
    void UpdateNormalAndDepth() { ++nNormalUpdates; }  // bookkeeping of the map, out of scope

    mapMapPointObs mObservations;
    KeyFrame* mpRefKF = nullptr;
    float mWorldPos[3] = {0, 0, 0};
    bool mbBad = false;
    int nNormalUpdates = 0;
    float mPosGBA[3] = {0, 0, 0};          // include/MapPoint.h:90-91
    long unsigned int mnBAGlobalForKF = 0;
};

// include/Frame.h: the slice PoseOptimization reads and writes (:48-67, :104, :214, :230)
class Frame {
public:
    int N = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    std::vector<KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    std::vector<float> mvInvLevelSigma2;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    Mat4f mTcw{};
    std::array<double, 225> mMargCovInv{};   // Matrix<double,15,15>, row-major here
    NavState mNavStatePrior;
    const NavState& GetNavState() const { return mNavState; }
    void SetNavState(const NavState& ns) { mNavState = ns; }
    void SetPose(const Mat4f& T) { mTcw = T; }
    void UpdatePoseFromNS() {  // Frame::UpdatePoseFromNS(ConfigParam::GetMatTbc()): the same float32 chain as KeyFrame's
        KeyFrame k;
        k.SetNavState(mNavState);
        k.UpdatePoseFromNS();
        mTcw = k.GetPose();
    }
    NavState mNavState;
};

inline bool cmpKeyFrameId::operator()(const KeyFrame* a, const KeyFrame* b) const { return a->mnId < b->mnId; }
inline void KeyFrame::EraseMapPointMatch(MapPoint* pMP) {
    const int idx = pMP->GetIndexInKeyFrame(this);
    if (idx >= 0) mvpMapPoints[idx] = nullptr;
}

}  // namespace ORB_SLAM2
