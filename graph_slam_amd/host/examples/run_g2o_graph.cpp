// Config-1 harness in the builder's own words (it drives the reference's OWN CGraphG2O, compiled in place from
// /root/reference/g2o/g2o_graph.cpp against shim/g2o): feeds a synthetic Manhattan-3D sequence through
// CGraphG2O::addNode exactly the way the reference's online driver does (g2o/test_g2o_graph.cpp:60-126:
// addNode per frame, optimise every m_optimize_step keyframes, fake odometry on failure, chi2 before/after
// the final optimizeGraph, trajectory dumps) and prints one JSON line for the tests.
//   usage: run_g2o_graph <n_poses> <lookback> <optimize_step> [out_prefix]
#include <cstdio>
#include <cstdlib>
#include <string>
#include "g2o_graph.h"
#include "g2o_parameter.h"
#include "g2o/core/sparse_optimizer.h"
#include "camera_node.h"
#include "vro_synth.h"

int main(int argc, char **argv) {
  const long n = argc > 1 ? std::atol(argv[1]) : 1000;
  const int lookback = argc > 2 ? std::atoi(argv[2]) : 4;
  const int step = argc > 3 ? std::atoi(argv[3]) : 0;
  const std::string out = argc > 4 ? argv[4] : "";
  fgo_synth::World::instance().generate(n, lookback, 0, 42);
  CG2OParams *p = CG2OParams::Instance();
  p->m_small_translation = 0.04; p->m_small_rotation = 3; p->m_lookback_nodes = lookback; p->m_optimize_step = step;
  CGraphG2O graph;
  if (!graph.mp_optimizer->handle()) { std::fprintf(stderr, "no device: %s\n", graph.mp_optimizer->lastError().c_str()); return 2; }
  int kf = 0, not_kf = 0, fake = 0;
  for (long f = 0; f < n; ++f) {
    CCameraNode *node = new CCameraNode();
    node->m_frame = (int)f;
    const ADD_RET r = graph.addNode(node);
    if (r == SUCC_KF) {
      ++kf;
      if (step > 0 && graph.camnodeSize() % step == 0) graph.optimizeGraph();
    } else if (r == FAIL_NOT_KF) { ++not_kf; delete node; }
    else { ++fake; graph.fakeOdoNode(node); }
  }
  const double before = graph.error();
  graph.optimizeGraph();
  const double after = graph.error();
  if (!out.empty()) {
    graph.trajectoryPLY(out + "_after.ply", RED);
    graph.writeTrajectory(out + "_trajectory.log");
    graph.writeG2O(out + ".g2o");
  }
  std::printf("{\"keyframes\": %d, \"not_kf\": %d, \"fake\": %d, \"nodes\": %zu, \"chi2_before\": %.17g, \"chi2_after\": %.17g}\n",
              kf, not_kf, fake, graph.camnodeSize(), before, after);
  return 0;
}
