// Multi-frame bundle adjustment through the reference's OWN graph builder (gtsam/gtsam_graph.cpp, compiled in place):
//   CGraphGT::firstNode (:320-367)                    X(0) = identity with a 1e-7 prior (+ V / B priors)
//   CGraphGT::addNodeOffline (:1593-1623)             one VO BetweenFactor per new keyframe, initial X(k) = X(k-1) * increment
//   CGraphGT::addToGTSAM(CCameraNodeBA*, CCameraNodeBA*, matches, CamModel*)  (:370-448)
//       new landmarks Q(id) with PriorFactor<Point3>(sigma 0.014), GenericProjectionFactor<Pose3, Point3, Cal3DS2> with
//       body_P_sensor = camera-to-IMU for every matched feature, landmark ids carried from keyframe to keyframe
//   CGraphGT::optimizeGraphBatch (:1784-1788)         LevenbergMarquardtOptimizer over everything
// The reference's drivers keep the addToGTSAM(CCameraNodeBA...) calls commented out (gtsam/test_ba_imu_graph.cpp:196-219,
// 398-417); this harness issues them in the same pattern.  What it fabricates (own words) is only what a front end would
// deliver: the feature sets of n_kf keyframes moving through a cloud of points -- camera-frame positions with noise, pixel
// positions with noise, and which features of two keyframes show the same point (CCameraNodeBA::matchNodePairBA stand-in).
// With >= 1000 landmarks libfgo eliminates them first (kernels_ba.hip).  FGO_GRAPH_DUMP_PREFIX=<p> writes the graph before
// and after the optimisation to <p>0.txt / <p>1.txt (host/gtsam_bridge.cpp) for an independent evaluation.
//   usage: run_ba_multiframe [n_kf=60] [n_points=6000] [lookback=3] [seed=5]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <gtsam/geometry/Cal3DS2.h>
#include <gtsam/geometry/Pose3.h>
#include <gtsam/inference/Symbol.h>
#include "camera_node_ba.h"
#include "cam_model.h"
#include "matching_result.h"
#include "gtsam_graph.h"

using namespace gtsam;
using symbol_shorthand::X;

static bool project(const Cal3DS2 &K, const Point3 &pc, double &u, double &v) {      // Cal3DS2::uncalibrate, p1 = p2 = 0
  if (pc(2) < 0.8 || pc(2) > 6.0) return false;
  const double x = pc(0) / pc(2), y = pc(1) / pc(2), r2 = x * x + y * y, g = 1 + K.k1() * r2 + K.k2() * r2 * r2;
  u = K.fx() * g * x + K.skew() * g * y + K.px();
  v = K.fy() * g * y + K.py();
  return u > 4 && u < 172 && v > 4 && v < 140 && r2 < 0.16;                           // SR4000 image, distortion model's valid radius
}

int main(int argc, char **argv) {
  const int n_kf = argc > 1 ? atoi(argv[1]) : 60, n_pts = argc > 2 ? atoi(argv[2]) : 6000, lookback = argc > 3 ? atoi(argv[3]) : 3;
  const unsigned seed = argc > 4 ? (unsigned)atoi(argv[4]) : 5u;
  std::mt19937_64 rng(seed);
  std::normal_distribution<double> N01(0.0, 1.0);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  const Cal3DS2 K(250.5773, 250.5773, 0, 90, 70, -0.8466, 0.5370);                   // gtsam_graph.cpp:544
  CamModel cam(250.5773, 250.5773, 90, 70, -0.8466, 0.5370);
  CGraphGT graph;
  graph.setCamera2IMU(0);                                                             // as the drivers do (test_ba_imu_graph.cpp)
  const Pose3 u2c = *graph.mp_u2c;
  // camera motion relative to the first camera frame: forward along the optical axis with a gentle sway
  std::vector<Pose3> M((size_t)n_kf);
  for (int k = 0; k < n_kf; ++k) {
    Vector6 xi; xi << 0.04 * std::sin(0.11 * k), 0.05 * std::sin(0.07 * k), 0.02 * std::sin(0.05 * k), 0.10 * std::sin(0.09 * k), 0.03 * std::cos(0.13 * k) - 0.03, 0.05 * k;
    M[k] = Pose3::Expmap(xi);
  }
  // points in the first camera frame, in a tube around the path
  std::vector<Point3> P((size_t)n_pts);
  const double depth = 0.05 * n_kf + 5.0;
  for (int q = 0; q < n_pts; ++q) P[q] = Point3(3.0 * (U(rng) - 0.5), 2.4 * (U(rng) - 0.5), 0.8 + depth * U(rng));
  std::vector<CCameraNodeBA *> nodes((size_t)n_kf);
  long n_feat = 0;
  for (int k = 0; k < n_kf; ++k) {
    CCameraNodeBA *n = new CCameraNodeBA;
    for (int q = 0; q < n_pts; ++q) {
      const Point3 pc = M[k].transform_to(P[q]);
      double u, v;
      if (!project(K, pc, u, v)) continue;
      Eigen::Vector4f loc; loc(0) = (float)(pc(0) + 0.004 * N01(rng)); loc(1) = (float)(pc(1) + 0.004 * N01(rng)); loc(2) = (float)(pc(2) + 0.008 * N01(rng)); loc(3) = 1.f;
      cv::KeyPoint kp; kp.pt.x = (float)(u + 0.3 * N01(rng)); kp.pt.y = (float)(v + 0.3 * N01(rng));
      n->m_feature_loc_3d.push_back(loc); n->m_feature_loc_2d.push_back(kp); n->mv_feature_qid.push_back(-1); n->mv_world_point.push_back(q);
    }
    n_feat += (long)n->mv_world_point.size();
    nodes[k] = n;
  }
  nodes[0]->m_seq_id = 1;
  graph.firstNode(nodes[0], false);
  for (int k = 1; k < n_kf; ++k) {
    // the VO increment between consecutive camera frames, with noise; information in the camera frame, [omega; v] order
    Vector6 noise; noise << 0.002 * N01(rng), 0.002 * N01(rng), 0.002 * N01(rng), 0.004 * N01(rng), 0.004 * N01(rng), 0.004 * N01(rng);
    const Pose3 inc = M[k - 1].between(M[k]) * Pose3::Expmap(noise);
    MatchingResult mr;
    mr.edge.id1 = k; mr.edge.id2 = k + 1;                                 // sequence ids (1-based), as in a VRO record
    mr.edge.transform = Eigen::Isometry3d(inc.matrix());
    mr.final_trafo = inc.matrix().cast<float>();
    mr.edge.informationMatrix.setZero();
    for (int d = 0; d < 3; ++d) { mr.edge.informationMatrix(d, d) = 1.0 / (0.002 * 0.002); mr.edge.informationMatrix(3 + d, 3 + d) = 1.0 / (0.004 * 0.004); }
    if (!graph.addNodeOffline(nodes[k], &mr, true)) { std::fprintf(stderr, "addNodeOffline(%d) failed\n", k); return 2; }
    for (int back = 1; back <= lookback && k - back >= 0; ++back) {
      CCameraNodeBA *ni = nodes[k - back], *nj = nodes[k];
      Eigen::Matrix4f Tji = M[k].between(M[k - back]).matrix().cast<float>();
      std::map<int, int> matches = nj->matchNodePairBA(ni, Tji, &cam);
      if (!matches.empty()) graph.addToGTSAM(ni, nj, matches, &cam);
    }
  }
  const char *pre = std::getenv("FGO_GRAPH_DUMP_PREFIX");
  if (pre) setenv("FGO_GRAPH_DUMP", (std::string(pre) + "0.txt").c_str(), 1);
  const double e0 = graph.mp_fac_graph->error(*graph.mp_node_values);
  if (pre) unsetenv("FGO_GRAPH_DUMP");
  graph.optimizeGraphBatch();
  if (pre) setenv("FGO_GRAPH_DUMP", (std::string(pre) + "1.txt").c_str(), 1);
  const double e1 = graph.mp_fac_graph->error(*graph.mp_node_values);
  if (pre) unsetenv("FGO_GRAPH_DUMP");
  // body poses against the truth X_k = u2c M_k u2c^-1
  double dt = 0, dr = 0;
  for (int k = 0; k < n_kf; ++k) {
    const Pose3 Xt = u2c * M[k] * u2c.inverse(), Xe = graph.mp_node_values->at<Pose3>(X(k));
    const Eigen::Matrix4d A = Xt.matrix(), B = Xe.matrix();
    for (int r = 0; r < 3; ++r) { dt = std::max(dt, std::fabs(A(r, 3) - B(r, 3))); for (int c = 0; c < 3; ++c) dr = std::max(dr, std::fabs(A(r, c) - B(r, c))); }
  }
  const Eigen::Matrix4d Tu = u2c.matrix();
  const Eigen::Quaterniond uq(Eigen::Matrix3d(Tu.block<3, 3>(0, 0)));
  std::printf("{\"keyframes\": %d, \"features\": %ld, \"landmarks\": %d, \"factors\": %zu, \"error0\": %.12e, \"error1\": %.12e, \"max_abs_dt\": %.6e, "
              "\"max_abs_dR\": %.6e, \"calib\": [%.10g, %.10g, %.10g, %.10g, %.10g, %.10g, %.10g, 0, 0], \"body_P_sensor\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g]}\n",
              n_kf, n_feat, graph.m_sift_landmark_id, graph.mp_fac_graph->size(), e0, e1, dt, dr, K.fx(), K.fy(), K.skew(), K.px(), K.py(), K.k1(), K.k2(),
              Tu(0, 3), Tu(1, 3), Tu(2, 3), uq.x(), uq.y(), uq.z(), uq.w());
  return 0;
}
