// Synthetic stand-in for the un-vendored VRO front end (visual_odometry/src/VRO in the reference's build:
// CMakeLists.txt:22-23).  Instead of SR4000 frames + RANSAC feature matching, a process-wide "world" holds a
// Manhattan-3D pose graph (fgo_synth_manhattan3d) and CCameraNode::matchNodePair looks up the
// relative-pose measurement between two frames.  This is what lets the reference's drivers run unchanged.
#pragma once
#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace fgo_synth {
struct World {
  int64_t n_poses = 0;
  std::vector<double> truth, init;                                   // 7 per pose
  std::map<std::pair<int, int>, int64_t> edge_of;                    // (older frame, newer frame) -> edge index
  std::vector<double> meas, info;                                    // 7 / 21 per edge
  // VO failures (featureless frames): a frame in this set matches NO older frame, so CGraphG2O / CGraphGT::addNode return
  // FAIL_KF and the drivers fall back to fakeOdoNode (g2o/g2o_graph.cpp:136-157, g2o/test_g2o_graph.cpp:90-95).
  // Filled from FGO_SYNTH_VO_FAIL="f1,f2,..." (0-based frame indices) by ensure() / generate().
  std::set<int> vo_fail;
  static World &instance();
  // (re)generate; env overrides: FGO_SYNTH_POSES, FGO_SYNTH_LOOKBACK, FGO_SYNTH_LOOPS, FGO_SYNTH_SEED
  void generate(int64_t n_poses, int lookback, int n_loop, uint64_t seed);
  void ensure();
};
}  // namespace fgo_synth
