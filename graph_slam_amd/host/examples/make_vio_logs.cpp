// Synthetic input for the reference's offline visual-inertial drivers (gtsam/test_vro_imu_graph.cpp:94-373,
// gtsam/test_ba_imu_graph.cpp:95-459): the files they read -- a VRO record log (`id_to id_from xi[6] Omega_ut[21]`,
// gtsam/gtsam_graph.cpp:1510-1558), a VN100 IMU log (`t ax ay az gx gy gz yaw pitch roll`, gtsam/imu_vn100.cpp:78-105)
// and the image time-stamp file -- for a platform that starts at rest at the origin, like the drivers assume
// (firstNode: identity pose, zero velocity, zero bias).  Frame ids start at 1 (sr_start_frame's default).
//   usage: make_vio_logs <out_dir> [n_keyframes=200] [lookback=3] [seed=44] [vo_fail=f1,f2,...]
// vo_fail: 1-based frame ids whose VRO matching FAILS: every record with that frame as the newer one is written as a void
// edge (information(0,0) = 10000, the reference's sentinel: gtsam/gtsam_graph.cpp:1600,1654), so the drivers add the node
// from the IMU prediction alone and -- plane_aided -- take their plane branch (gtsam/test_vro_imu_graph.cpp:202-314)
// writes <dir>/imu.log, <dir>/img_time.log, <dir>/vro_results.log, <dir>/truth.log
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <random>
#include <set>
#include <string>
#include <vector>
#include <gtsam/navigation/CombinedImuFactor.h>

using namespace gtsam;
using namespace std;

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <out_dir> [n_keyframes] [lookback] [seed]\n", argv[0]); return 1; }
  const string dir = argv[1];
  const int n_kf = argc > 2 ? atoi(argv[2]) : 200, lookback = argc > 3 ? atoi(argv[3]) : 3;
  const unsigned seed = argc > 4 ? (unsigned)atoi(argv[4]) : 44u;
  std::set<int> vo_fail;
  if (argc > 5) { const char *p = argv[5]; while (*p) { char *e = 0; const long f = strtol(p, &e, 10); if (e == p) break; vo_fail.insert((int)f); p = (*e == ',') ? e + 1 : e; } }
  const int SAMPLES_PER_KF = 40;      // 200 Hz IMU, 5 Hz keyframes (test_vro_imu_graph.cpp:111)
  const double DT = 0.005;
  mt19937_64 rng(seed);
  normal_distribution<double> N01(0.0, 1.0);
  uniform_real_distribution<double> U(0.0, 6.0);
  vector<Pose3> pose(1);
  vector<Vector3> vel(1);
  const Vector3 ba(0.03, -0.02, 0.01), bg(0.002, -0.001, 0.0015);
  const imuBias::ConstantBias bias_true(ba, bg);
  // VN100 parameters as gtsam/imu_vn100.cpp:24-67 sets them, gravity 9.71 (gtsam/imu_base.cpp:258-263)
  std::shared_ptr<PreintegratedCombinedMeasurements::Params> P = PreintegratedCombinedMeasurements::Params::MakeSharedD(9.71);
  {
    fgo_imu_params vn; fgo_imu_params_vn100(&vn);
    P->accelerometerCovariance = Matrix3::Identity() * vn.acc_cov; P->gyroscopeCovariance = Matrix3::Identity() * vn.gyro_cov;
    P->integrationCovariance = Matrix3::Identity() * vn.integ_cov; P->biasAccCovariance = Matrix3::Identity() * vn.bias_acc_cov;
    P->biasOmegaCovariance = Matrix3::Identity() * vn.bias_gyro_cov; P->biasAccOmegaInt = Matrix6::Identity() * vn.bias_acc_omega_int;
  }
  ofstream imu((dir + "/imu.log").c_str()), tim((dir + "/img_time.log").c_str());
  imu << setprecision(12);
  tim << setprecision(12);
  double t = 100.0;                 // time of keyframe 1
  tim << 1 << " " << t;
  for (int k = 0; k + 1 < n_kf; ++k) {
    double ph[6];
    for (int i = 0; i < 6; ++i) ph[i] = U(rng);
    PreintegratedCombinedMeasurements pim(P, imuBias::ConstantBias());
    const Rot3 Rt = pose[k].rotation().inverse();
    const Vector3 g_body = Rt * Vector3(0, 0, -9.71), v_body = Rt * vel[k];
    for (int s = 0; s < SAMPLES_PER_KF; ++s) {
      const double ts = s * DT;
      Vector3 acc, gyro;
      for (int i = 0; i < 3; ++i) {
        gyro(i) = 0.25 * sin(2.1 * ts + ph[i]) + bg(i);
        acc(i) = 0.6 * cos(1.3 * ts + ph[3 + i]) + ba(i) + g_body(i) - 0.5 * v_body(i);
      }
      // the log stores what the sensor reports; the reference reads it back through `float` (imu_vn100.cpp:86)
      const float fa[3] = {(float)acc(0), (float)acc(1), (float)acc(2)}, fg[3] = {(float)gyro(0), (float)gyro(1), (float)gyro(2)};
      imu << (t + ts) << " " << fa[0] << " " << fa[1] << " " << fa[2] << " " << fg[0] << " " << fg[1] << " " << fg[2] << " 0 0 0\n";
      pim.integrateMeasurement(Vector3(fa[0], fa[1], fa[2]), Vector3(fg[0], fg[1], fg[2]), DT);
    }
    const NavState nx = pim.predict(NavState(pose[k], vel[k]), bias_true);
    pose.push_back(nx.pose());
    vel.push_back(nx.v());
    t += SAMPLES_PER_KF * DT;
    tim << "\n" << (k + 2) << " " << t;
  }
  // a few trailing samples so that the last keyframe's time stamp is bracketed (CImuBase::findIndexAt); no newline at the
  // end of either file: the readers loop on eof() and would duplicate the last record
  for (int s = 0; s < 4; ++s) imu << (t + s * DT) << " 0 0 -9.71 0 0 0 0 0 0" << (s < 3 ? "\n" : "");
  // VRO records in the CAMERA frame (CGraphGT::addToGTSAM conjugates them back with T_u2c, gtsam_graph.cpp:640): the
  // odometry + look-back records ordered by the newer frame, as VRO emits them; information in [omega; v] order
  const Pose3 Tuc(Rot3::RzRyRx(M_PI / 2., 0., M_PI / 2.), Point3()), Tcu = Tuc.inverse();   // CGraphGT::setCamera2IMU(0)
  const double nz = 0.01;
  ofstream vro((dir + "/vro_results.log").c_str());
  vro << setprecision(12);
  for (int j = 1; j < n_kf; ++j)
    for (int d = 1; d <= lookback + 1 && j - d >= 0; ++d) {
      const int i = j - d;
      Vector6 e;
      for (int q = 0; q < 3; ++q) { e(q) = 0.5 * nz * N01(rng); e(3 + q) = nz * N01(rng); }
      const Pose3 rel_imu = pose[i].between(pose[j]) * Pose3::Expmap(e);
      const Pose3 rel_cam = Tcu * rel_imu * Tuc;
      const Vector6 xi = Pose3::Logmap(rel_cam);
      vro << (j + 1) << " " << (i + 1);
      for (int q = 0; q < 6; ++q) vro << " " << xi(q);
      // the record holds the information of the camera-frame estimate; diagonal in the IMU frame, rotated to the camera
      Matrix6 W = Matrix6::Zero();
      for (int r = 0; r < 6; ++r) W(r, r) = r < 3 ? 1.0 / (0.25 * nz * nz) : 1.0 / (nz * nz);
      const Matrix6 Ad = Tcu.AdjointMap();
      Matrix6 Wc = Ad * W * Ad.transpose();
      if (vo_fail.count(j + 1)) { Wc = Matrix6::Identity(); Wc(0, 0) = 10000; }     // void edge: VRO found no transformation
      for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) vro << " " << Wc(r, c);
      vro << "\n";
    }
  ofstream tr((dir + "/truth.log").c_str());
  tr << setprecision(12);
  for (int k = 0; k < n_kf; ++k) {
    const Quaternion q = pose[k].rotation().toQuaternion();
    tr << (k + 1) << " " << pose[k].x() << " " << pose[k].y() << " " << pose[k].z() << " " << q.x() << " " << q.y() << " " << q.z() << " " << q.w() << "\n";
  }
  return 0;
}
