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
  // ---- synthetic depth camera for the plane-aided drivers (gtsam/test_vro_imu_graph.cpp:135-145, 202-314).  When
  // FGO_SYNTH_TRUTH names a trajectory file (`frame x y z qx qy qz qw` per line, IMU / body poses: make_vio_logs' truth.log)
  // the frames are rendered as range images of an axis-aligned ROOM around the trajectory (6 walls), seen through the
  // SR4000 pin-hole model and the reference's camera-to-IMU rotation (CGraphGT::setCamera2IMU(0): RzRyRx(pi/2, 0, pi/2)).
  std::map<int, std::vector<double> > body_pose;                      // frame id (1-based, as in the file) -> t(3) q(4)
  double room_lo[3], room_hi[3];
  bool has_room = false;
  void load_truth(const char *path, double margin);
  // camera pose in the world of a frame: R (row-major 3x3), t; false if the frame is unknown
  bool camera_pose(int frame_id, double R[9], double t[3]) const;
  // depth along the optical axis and wall id (0..5: x-, x+, y-, y+, z-, z+) of pixel (u, v); false if nothing is hit
  bool cast(const double R[9], const double t[3], double dx, double dy, double &z, int &wall) const;
  static World &instance();
  // (re)generate; env overrides: FGO_SYNTH_POSES, FGO_SYNTH_LOOKBACK, FGO_SYNTH_LOOPS, FGO_SYNTH_SEED
  void generate(int64_t n_poses, int lookback, int n_loop, uint64_t seed);
  void ensure();
};
}  // namespace fgo_synth
