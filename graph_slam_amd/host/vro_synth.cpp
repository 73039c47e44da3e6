// Synthetic VRO front end: see shim/vro_synth.h
#include "shim/vro_synth.h"
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <algorithm>
#include "../../include/fgo.h"
#include "shim/camera_node.h"

namespace fgo_synth {
World &World::instance() { static World w; return w; }

void World::generate(int64_t n, int lookback, int n_loop, uint64_t seed) {
  n_poses = n;
  const int64_t max_e = n * (1 + lookback + n_loop);
  truth.assign((size_t)n * 7, 0.0); init.assign((size_t)n * 7, 0.0);
  meas.assign((size_t)max_e * 7, 0.0); info.assign((size_t)max_e * 21, 0.0);
  std::vector<int64_t> ei((size_t)max_e), ej((size_t)max_e);
  const int64_t e = fgo_synth_manhattan3d(n, lookback, n_loop, seed, 0.02, 0.005, init.data(), truth.data(), ei.data(),
                                          ej.data(), meas.data(), info.data(), max_e);
  edge_of.clear();
  for (int64_t k = 0; k < e; ++k) edge_of[{(int)ei[k], (int)ej[k]}] = k;
  vo_fail.clear();
  if (const char *vf = std::getenv("FGO_SYNTH_VO_FAIL")) {
    const char *p = vf;
    while (*p) {
      char *end = nullptr;
      const long f = std::strtol(p, &end, 10);
      if (end == p) break;
      vo_fail.insert((int)f);
      p = (*end == ',') ? end + 1 : end;
    }
  }
}

void World::load_truth(const char *path, double margin) {
  std::ifstream in(path);
  body_pose.clear();
  int id;
  double v[7];
  for (int a = 0; a < 3; ++a) { room_lo[a] = 1e300; room_hi[a] = -1e300; }
  while (in >> id >> v[0] >> v[1] >> v[2] >> v[3] >> v[4] >> v[5] >> v[6]) {
    body_pose[id] = std::vector<double>(v, v + 7);
    for (int a = 0; a < 3; ++a) { room_lo[a] = std::min(room_lo[a], v[a]); room_hi[a] = std::max(room_hi[a], v[a]); }
  }
  has_room = !body_pose.empty();
  for (int a = 0; a < 3; ++a) { room_lo[a] -= margin; room_hi[a] += margin; }
}

bool World::camera_pose(int frame_id, double R[9], double t[3]) const {
  std::map<int, std::vector<double> >::const_iterator it = body_pose.find(frame_id);
  if (it == body_pose.end()) return false;
  const std::vector<double> &p = it->second;
  const double x = p[3], y = p[4], z = p[5], w = p[6];
  const double Rb[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                        2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  // T_u2c = RzRyRx(pi/2, 0, pi/2) = Rz(pi/2) Rx(pi/2) = [[0,0,1],[1,0,0],[0,1,0]]: camera z (optical axis) = body x
  static const double Ruc[9] = {0, 0, 1, 1, 0, 0, 0, 1, 0};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R[3 * r + c] = Rb[3 * r] * Ruc[c] + Rb[3 * r + 1] * Ruc[3 + c] + Rb[3 * r + 2] * Ruc[6 + c];
  t[0] = p[0]; t[1] = p[1]; t[2] = p[2];
  return true;
}

bool World::cast(const double R[9], const double t[3], double dx, double dy, double &z, int &wall) const {
  // ray o + s (R d), d = (dx, dy, 1): the depth along the optical axis is s
  const double d[3] = {R[0] * dx + R[1] * dy + R[2], R[3] * dx + R[4] * dy + R[5], R[6] * dx + R[7] * dy + R[8]};
  double best = 1e300;
  int bw = -1;
  for (int a = 0; a < 3; ++a) {
    if (std::fabs(d[a]) < 1e-12) continue;
    const double s = ((d[a] > 0 ? room_hi[a] : room_lo[a]) - t[a]) / d[a];
    if (s > 0 && s < best) { best = s; bw = 2 * a + (d[a] > 0 ? 1 : 0); }
  }
  if (bw < 0) return false;
  z = best; wall = bw;
  return true;
}

void World::ensure() {
  if (!has_room)
    if (const char *tp = std::getenv("FGO_SYNTH_TRUTH")) load_truth(tp, std::getenv("FGO_SYNTH_ROOM_MARGIN") ? std::atof(std::getenv("FGO_SYNTH_ROOM_MARGIN")) : 2.0);
  if (n_poses > 0) return;
  auto env = [](const char *k, long d) { const char *v = std::getenv(k); return v ? std::atol(v) : d; };
  generate(env("FGO_SYNTH_POSES", 1000), (int)env("FGO_SYNTH_LOOKBACK", 4), (int)env("FGO_SYNTH_LOOPS", 0),
           (uint64_t)env("FGO_SYNTH_SEED", 42));
}
}  // namespace fgo_synth

MatchingResult CCameraNode::matchNodePair(CCameraNode *older) {
  MatchingResult mr;
  fgo_synth::World &w = fgo_synth::World::instance();
  w.ensure();
  if (w.vo_fail.count(m_frame)) return mr;              // injected VO failure: this frame matches nothing older (succeed_match stays false)
  auto it = w.edge_of.find({older->m_frame, m_frame});
  if (it == w.edge_of.end()) return mr;                 // no overlap: VRO fails to find a transformation
  const double *z = &w.meas[(size_t)it->second * 7], *om = &w.info[(size_t)it->second * 21];
  mr.edge.id1 = older->m_id; mr.edge.id2 = m_id;
  Eigen::Quaterniond q(z[6], z[3], z[4], z[5]);
  Eigen::Vector3d t; t(0) = z[0]; t(1) = z[1]; t(2) = z[2];
  mr.edge.transform = Eigen::Isometry3d(q.toRotationMatrix(), t);
  int k = 0;
  for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { mr.edge.informationMatrix(r, c) = om[k]; mr.edge.informationMatrix(c, r) = om[k]; ++k; }
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) mr.final_trafo(r, c) = (float)mr.edge.transform.matrix()(r, c);
  // inlier count decides whether a later match may reset the estimate (g2o_graph.cpp:218-220): the odometry
  // match (adjacent frames) has the most inliers, so the chained odometry guess is kept.
  const int gap = m_frame - older->m_frame;
  mr.inlier_matches.resize(gap == 1 ? 200 : (gap < 20 ? 100 - gap : 50));
  mr.succeed_match = true;
  return mr;
}

// ---- GTSAM-side front-end calls (gtsam/gtsam_graph.cpp:256-277, 450-610)
void CCameraNode::computeCov(CCameraNode *older, std::vector<cv::DMatch> &, cov_helper_fn, Eigen::Matrix<double, 6, 6> &cov) {
  fgo_synth::World &w = fgo_synth::World::instance();
  w.ensure();
  cov = Eigen::Matrix<double, 6, 6>::Identity() * 1e-4;
  auto it = older ? w.edge_of.find({older->m_frame, m_frame}) : w.edge_of.end();
  if (it == w.edge_of.end()) return;
  const double *om = &w.info[(size_t)it->second * 21];
  Eigen::Matrix<double, 6, 6> W;
  int k = 0;
  for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { W(r, c) = om[k]; W(c, r) = om[k]; ++k; }
  cov = W.inverse();
}

#include "shim/camera_node_ba.h"
#include "shim/transformation_estimation_euclidean.h"
std::map<int, int> CCameraNodeBA::matchNodePairBA(CCameraNodeBA *older, Eigen::Matrix4f &, CamModel *) {
  std::map<int, int> m;                                   // features that look at the same synthetic world point
  if (!older) return m;
  std::map<int, int> by_point;
  for (size_t i = 0; i < older->mv_world_point.size(); ++i) by_point[older->mv_world_point[i]] = (int)i;
  for (size_t j = 0; j < mv_world_point.size(); ++j) {
    std::map<int, int>::iterator it = by_point.find(mv_world_point[j]);
    if (it != by_point.end()) m[it->second] = (int)j;
  }
  return m;
}
Eigen::Matrix4f getTransformFromMatches(const CCameraNode *newer, const CCameraNode *older, const std::vector<cv::DMatch> &) {
  Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
  fgo_synth::World &w = fgo_synth::World::instance();
  w.ensure();
  if (!newer || !older) return T;
  auto it = w.edge_of.find({older->m_frame, newer->m_frame});
  if (it == w.edge_of.end()) return T;
  const double *z = &w.meas[(size_t)it->second * 7];
  const Eigen::Matrix3d R = Eigen::Quaterniond(z[6], z[3], z[4], z[5]).toRotationMatrix();
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T(r, c) = (float)R(r, c); T(r, 3) = (float)z[r]; }
  return T;
}
