// Two-view bundle adjustment through the reference's OWN CGraphGT::bundleAdjust (gtsam/gtsam_graph.cpp:500-610, compiled
// in place): a PriorFactor<Pose3> on the first camera, PriorFactor<Point3>(sigma 0.014) + two
// GenericProjectionFactor<Pose3, Point3, Cal3DS2> per matched feature, LevenbergMarquardtOptimizer, then
// Marginals(...).marginalCovariance of the second pose -> information of the VRO edge (:593-601).
// The harness (own words) only fabricates the two CCameraNodeBA feature sets a front end would deliver: n points in
// front of camera i, observed by i and by j = i * T_true with pixel noise, and prints the recovered relative pose next
// to the truth.    usage: run_bundle_adjust [n_points=60] [seed=3]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <gtsam/geometry/Cal3DS2.h>
#include <gtsam/geometry/Pose3.h>
#include "camera_node_ba.h"
#include "cam_model.h"
#include "gtsam_graph.h"

using namespace gtsam;

static void project(const Cal3DS2 &K, const Point3 &pc, double &u, double &v) {      // Cal3DS2::uncalibrate, p1 = p2 = 0
  const double x = pc(0) / pc(2), y = pc(1) / pc(2), r2 = x * x + y * y, g = 1 + K.k1() * r2 + K.k2() * r2 * r2;
  u = K.fx() * g * x + K.skew() * g * y + K.px();
  v = K.fy() * g * y + K.py();
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 60;
  const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 3u;
  std::mt19937_64 rng(seed);
  std::normal_distribution<double> N01(0.0, 1.0);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  const Cal3DS2 K(250.5773, 250.5773, 0, 90, 70, -0.8466, 0.5370);        // gtsam_graph.cpp:544
  Vector6 xi; xi << 0.02, -0.03, 0.015, 0.10, -0.05, 0.04;
  const Pose3 Tij = Pose3::Expmap(xi);                                    // camera j in the frame of camera i
  CCameraNodeBA *ni = new CCameraNodeBA, *nj = new CCameraNodeBA;
  for (int q = 0; q < n; ++q) {
    const Point3 p(0.35 * U(rng), 0.25 * U(rng), 1.5 + 0.8 * U(rng));     // in camera i (narrow field of view: small distortion radius)
    const Point3 pj = Tij.transform_to(p);
    double ui, vi, uj, vj;
    project(K, p, ui, vi); project(K, pj, uj, vj);
    Eigen::Vector4f loc; loc(0) = (float)(p(0) + 0.005 * N01(rng)); loc(1) = (float)(p(1) + 0.005 * N01(rng)); loc(2) = (float)(p(2) + 0.01 * N01(rng)); loc(3) = 1.f;
    cv::KeyPoint ki, kj;
    ki.pt.x = (float)(ui + 0.3 * N01(rng)); ki.pt.y = (float)(vi + 0.3 * N01(rng));
    kj.pt.x = (float)(uj + 0.3 * N01(rng)); kj.pt.y = (float)(vj + 0.3 * N01(rng));
    ni->m_feature_loc_3d.push_back(loc); ni->m_feature_loc_2d.push_back(ki); ni->mv_feature_qid.push_back(-1); ni->mv_world_point.push_back(q);
    nj->m_feature_loc_3d.push_back(loc); nj->m_feature_loc_2d.push_back(kj); nj->mv_feature_qid.push_back(-1); nj->mv_world_point.push_back(q);
  }
  CGraphGT graph;
  ni->m_id = 0; ni->m_seq_id = 1; nj->m_id = 1; nj->m_seq_id = 2;
  graph.m_graph_map[0] = ni;                                              // bundleAdjust looks the older node up by graph id
  MatchingResult mr;
  mr.edge.id1 = 1; mr.edge.id2 = 2;                                       // sequence ids, as in a VRO record
  mr.final_trafo.setIdentity();
  CamModel cam(250.5773, 250.5773, 90, 70, -0.8466, 0.5370);
  const bool ok = graph.bundleAdjust(&mr, nj, &cam);
  if (!ok) { std::fprintf(stderr, "bundleAdjust returned false\n"); return 2; }
  const Eigen::Matrix4d T = Tij.matrix();
  double dt = 0, dr = 0;
  for (int r = 0; r < 3; ++r) { dt = std::max(dt, std::fabs((double)mr.final_trafo(r, 3) - T(r, 3))); for (int c = 0; c < 3; ++c) dr = std::max(dr, std::fabs((double)mr.final_trafo(r, c) - T(r, c))); }
  const Eigen::Matrix<double, 6, 6> &W = mr.edge.informationMatrix;
  std::printf("{\"points\": %d, \"max_abs_dt\": %.6e, \"max_abs_dR\": %.6e, \"info_diag\": [%.6e, %.6e, %.6e, %.6e, %.6e, %.6e], \"info_sym_err\": %.3e}\n",
              n, dt, dr, W(0, 0), W(1, 1), W(2, 2), W(3, 3), W(4, 4), W(5, 5), std::fabs(W(0, 4) - W(4, 0)) + std::fabs(W(2, 5) - W(5, 2)));
  delete nj;                                                              // (ni is owned by the graph map)
  return 0;
}
