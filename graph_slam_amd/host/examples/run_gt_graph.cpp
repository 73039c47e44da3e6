// Offline visual-inertial replay through the CGraphGT / CImuVn100 mirrors, following the reference's driver
// gtsam/test_vro_imu_graph.cpp:94-373 step for step:
//   readVRORecord -> setCamera2IMU -> CImuVn100::readImuData -> firstNode -> for every record:
//   addNodeOffline, imu->predictNextFlag, CombinedImuFactor into mp_fac_graph, addToGTSAM(NavState), loop-closure
//   records through addEdgeOffline -> error() -> optimizeGraphBatch() -> error() -> writeTrajectory / writeG2O.
// The SR4000 frames, feature extraction and plane segmentation of the reference driver are replaced by files this
// program synthesises first (a VRO record log, a VN100 IMU log, image time stamps, plane observations): the graph
// side is what is under test.  usage: run_gt_graph <out_dir> [n_keyframes] [use_planes] [optimize_every]
// (the reference calls optimizeGraphIncremental after every record; optimize_every = N does it every N new nodes)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include <gtsam/navigation/CombinedImuFactor.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/Values.h>

#include "camera_node.h"
#include "gt_parameter.h"
#include "gtsam_graph.h"
#include "imu_vn100.h"
#include "matching_result.h"

using namespace gtsam;
using namespace std;
using symbol_shorthand::B;
using symbol_shorthand::V;
using symbol_shorthand::X;

namespace {
const int SAMPLES_PER_KF = 40;      // 200 Hz IMU, 5 Hz keyframes (test_vro_imu_graph.cpp:111)
const double DT = 0.005;

struct World {
  vector<Pose3> pose;               // true IMU-frame poses
  vector<Vector3> vel;
};

// writes <dir>/imu.log, <dir>/img_time.log, <dir>/vro_results.log, <dir>/planes.log, <dir>/truth.log
World synthesise(const string &dir, int n_kf, int lookback, unsigned seed) {
  mt19937_64 rng(seed);
  normal_distribution<double> N01(0.0, 1.0);
  uniform_real_distribution<double> U(0.0, 6.0);
  World w;
  w.pose.push_back(Pose3());
  Vector3 v0; v0(0) = 0.3; v0(1) = 0.1; v0(2) = 0.0;
  w.vel.push_back(v0);
  Vector3 ba, bg; ba(0) = 0.03; ba(1) = -0.02; ba(2) = 0.01; bg(0) = 0.002; bg(1) = -0.001; bg(2) = 0.0015;
  const imuBias::ConstantBias bias_true(ba, bg);
  std::shared_ptr<PreintegratedCombinedMeasurements::Params> P = PreintegratedCombinedMeasurements::Params::MakeSharedD(9.71);
  ofstream imu((dir + "/imu.log").c_str()), tim((dir + "/img_time.log").c_str());
  imu << setprecision(9);
  tim << setprecision(9);
  double t = 100.0;                 // time of keyframe 0
  tim << 0 << " " << t << "\n";
  for (int k = 0; k + 1 < n_kf; ++k) {
    double ph[6];
    for (int i = 0; i < 6; ++i) ph[i] = U(rng);
    PreintegratedCombinedMeasurements pim(P, imuBias::ConstantBias());
    const Rot3 Rt = w.pose[k].rotation().inverse();
    Vector3 gw; gw(0) = 0; gw(1) = 0; gw(2) = -9.71;
    const Vector3 g_body = Rt * gw, v_body = Rt * w.vel[k];
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
      Vector3 a2, g2;
      for (int i = 0; i < 3; ++i) { a2(i) = fa[i]; g2(i) = fg[i]; }
      pim.integrateMeasurement(a2, g2, DT);
    }
    const NavState nx = pim.predict(NavState(w.pose[k], w.vel[k]), bias_true);
    w.pose.push_back(nx.pose());
    w.vel.push_back(nx.v());
    t += SAMPLES_PER_KF * DT;
    tim << (k + 1) << " " << t << "\n";
  }
  // a few trailing samples so that the last keyframe's time stamp is bracketed (findIndexAt)
  for (int s = 0; s < 4; ++s) imu << (t + s * DT) << " 0 0 -9.71 0 0 0 0 0 0\n";

  // VRO records in the CAMERA frame (addToGTSAM conjugates them back with T_u2c): first the odometry + look-back records
  // ordered by the newer frame, as VRO emits them; information diag, [omega; v] order
  const Pose3 Tuc(Rot3::RzRyRx(M_PI / 2., 0., M_PI / 2.), Point3()), Tcu = Tuc.inverse();   // CGraphGT::setCamera2IMU(0)
  const double nz = 0.01;
  ofstream vro((dir + "/vro_results.log").c_str());
  vro << setprecision(12);
  for (int j = 1; j < n_kf; ++j)
    for (int d = 1; d <= lookback + 1 && j - d >= 0; ++d) {
      const int i = j - d;
      Vector6 e;
      for (int q = 0; q < 3; ++q) { e(q) = 0.5 * nz * N01(rng); e(3 + q) = nz * N01(rng); }
      const Pose3 rel_imu = w.pose[i].between(w.pose[j]) * Pose3::Expmap(e);
      const Pose3 rel_cam = Tcu * rel_imu * Tuc;
      const Vector6 xi = Pose3::Logmap(rel_cam);
      vro << j << " " << i;
      for (int q = 0; q < 6; ++q) vro << " " << xi(q);
      for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) vro << " " << (r == c ? (r < 3 ? 1.0 / (0.25 * nz * nz) : 1.0 / (nz * nz)) : 0.0);
      vro << "\n";
    }
  // plane observations: a floor (z = -1.5 in the world) and one wall, measured in the body frame
  ofstream pl((dir + "/planes.log").c_str());
  pl << setprecision(12);
  const double planes_w[2][4] = {{0, 0, 1, 1.5}, {1, 0, 0, -4.0}};     // n . x + d = 0
  for (int k = 0; k < n_kf; ++k)
    for (int l = 0; l < 2; ++l) {
      if (l == 1 && (k % 3) != 0) continue;
      Vector3 n; n(0) = planes_w[l][0]; n(1) = planes_w[l][1]; n(2) = planes_w[l][2];
      const Vector3 nb = w.pose[k].rotation().inverse() * n;           // OrientedPlane3::transform(pose)
      const double db = planes_w[l][3] + n(0) * w.pose[k].x() + n(1) * w.pose[k].y() + n(2) * w.pose[k].z();
      pl << k << " " << l << " " << nb(0) + 1e-3 * N01(rng) << " " << nb(1) + 1e-3 * N01(rng) << " " << nb(2) + 1e-3 * N01(rng) << " " << db + 1e-2 * N01(rng) << "\n";
    }
  ofstream tr((dir + "/truth.log").c_str());
  tr << setprecision(12);
  for (int k = 0; k < n_kf; ++k) tr << k << " " << w.pose[k].x() << " " << w.pose[k].y() << " " << w.pose[k].z() << "\n";
  return w;
}

bool loadImgTime(const string &f, map<int, double> &img_times) {
  ifstream inf(f.c_str());
  if (!inf.is_open()) return false;
  int id; double t;
  while (inf >> id >> t) img_times[id] = t;
  return true;
}
}  // namespace

int main(int argc, char **argv) {
  const string dir = argc > 1 ? argv[1] : ".";
  const int n_kf = argc > 2 ? atoi(argv[2]) : 200;
  const bool use_planes = argc > 3 ? atoi(argv[3]) != 0 : true;
  const int optimize_every = argc > 4 ? atoi(argv[4]) : 25;
  const int g_f_start = 0;
  synthesise(dir, n_kf, 3, 44);

  // ---- the driver proper
  CGTParams::Instance()->m_vro_result = dir + "/vro_recorded.log";
  CGraphGT gt_graph;
  gt_graph.readVRORecord(dir + "/vro_results.log");
  gt_graph.setCamera2IMU(0);

  map<int, double> img_times;
  if (!loadImgTime(dir + "/img_time.log", img_times)) { fprintf(stderr, "failed to read the time file\n"); return 1; }
  gtsam::imuBias::ConstantBias prior_bias;
  CImuVn100 *imu = new CImuVn100(DT, prior_bias);
  if (!imu->readImuData(dir + "/imu.log")) { fprintf(stderr, "failed to load imu data\n"); return 1; }

  CCameraNode *pNewNode = new CCameraNode();
  pNewNode->m_seq_id = g_f_start;
  gt_graph.firstNode(pNewNode, false);
  imu->setStartPoint(img_times[g_f_start]);
  {   // the synthetic platform starts with a known velocity (the reference starts at rest)
    Vector3 v0; v0(0) = 0.3; v0(1) = 0.1; v0(2) = 0.0;
    NavState s0(Pose3(), v0);
    imu->setState(s0);
  }

  multimap<int, vector<double> > plane_obs;
  if (use_planes) {
    ifstream pl((dir + "/planes.log").c_str());
    int k, l; double a, b, c, d;
    while (pl >> k >> l >> a >> b >> c >> d) { vector<double> o; o.push_back(l); o.push_back(a); o.push_back(b); o.push_back(c); o.push_back(d); plane_obs.insert(make_pair(k, o)); }
  }
  auto add_planes = [&](int node_id) {
    pair<multimap<int, vector<double> >::iterator, multimap<int, vector<double> >::iterator> r = plane_obs.equal_range(node_id);
    for (multimap<int, vector<double> >::iterator it = r.first; it != r.second; ++it) {
      Vector4 z; z(0) = it->second[1]; z(1) = it->second[2]; z(2) = it->second[3]; z(3) = it->second[4];
      const double nn = sqrt(z(0) * z(0) + z(1) * z(1) + z(2) * z(2));
      for (int q = 0; q < 3; ++q) z(q) /= nn;
      Matrix3 S = Matrix3::Identity() * 1e-4;                     // gtsam_graph.cpp:1206
      gt_graph.addPlaneFactor(z, S, node_id, (int)it->second[0]);
    }
  };
  add_planes(0);

  int cur_frame_id = g_f_start, cur_node_id = 0;
  for (size_t i = 0; i < gt_graph.mv_vro_res.size(); i++) {
    MatchingResult *pm = gt_graph.mv_vro_res[i];
    if (pm->edge.id2 <= g_f_start) continue;
    if (pm->edge.id2 > cur_frame_id) {            // a new frame: incremental edge + IMU factor (test_vro_imu_graph.cpp:163-199)
      const int cur_imu_id = pm->edge.id2;
      CCameraNode *pNode = new CCameraNode();
      const bool valid_match = gt_graph.addNodeOffline(pNode, pm);
      if (!valid_match) gt_graph.m_graph_map[pNode->m_id] = pNode;
      NavState cur_p;
      const bool imu_available = imu->predictNextFlag(img_times[cur_imu_id], cur_p);
      PreintegratedCombinedMeasurements *preint_imu_combined = dynamic_cast<PreintegratedCombinedMeasurements *>(imu->mp_combined_pre_imu);
      if (imu_available) {
        cur_node_id = pNode->m_id;
        CombinedImuFactor imu_factor(X(cur_node_id - 1), V(cur_node_id - 1), X(cur_node_id), V(cur_node_id), B(cur_node_id - 1), B(cur_node_id),
                                     *preint_imu_combined);
        gt_graph.mp_fac_graph->add(imu_factor);
        gt_graph.mp_new_fac->add(imu_factor);
        gt_graph.addToGTSAM(cur_p, cur_node_id, !valid_match);
        if (optimize_every > 0 && cur_node_id % optimize_every == 0) {
          // test_vro_imu_graph.cpp:343-350: optimise, then restart the preintegration from the optimised bias / state
          gt_graph.optimizeGraphIncremental();
          imu->resetPreintegrationAndBias(gt_graph.mp_node_values->at<imuBias::ConstantBias>(B(cur_node_id)));
          NavState pre_state(gt_graph.mp_node_values->at<Pose3>(X(cur_node_id)), gt_graph.mp_node_values->at<Vector3>(V(cur_node_id)));
          imu->setState(pre_state);
        } else {
          imu->resetPreintegrationAndBias();
          NavState next(gt_graph.mp_node_values->at<Pose3>(X(cur_node_id)), cur_p.v());
          imu->setState(next);
        }
      }
      if (use_planes) add_planes(pNode->m_id);
      cur_frame_id = pm->edge.id2;
    } else {                                      // a look-back / loop-closure record between existing nodes
      gt_graph.addEdgeOffline(pm);
    }
  }

  const double e0 = gt_graph.error();
  gt_graph.optimizeGraphBatch();
  const double e1 = gt_graph.error();
  printf("nodes %zu factors %zu error before %.9e after %.9e\n", gt_graph.camnodeSize(), gt_graph.mp_fac_graph->size(), e0, e1);
  gt_graph.writeTrajectory(dir + "/trajectory.log");
  gt_graph.writeG2O(dir + "/graph.g2o");
  gt_graph.trajectoryPLY(dir + "/trajectory.ply", CG::BLUE);
  const Matrix6 C = gt_graph.marginalCovariance((int)gt_graph.camnodeSize() - 1);
  printf("marginal covariance of the last pose: trace %.6e\n", C.trace());
  delete imu;
  return e1 <= e0 ? 0 : 2;
}
