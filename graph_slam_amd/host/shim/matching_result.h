// Stand-in for VRO's matching_result.h with the members the graph wrappers read
// (g2o/g2o_graph.cpp:98-131,147-151,174-220; gtsam/gtsam_graph.cpp:630-695,1510-1558).
#pragma once
#include <vector>
#include <Eigen/Core>
#include <Eigen/Geometry>
#include "opencv2/opencv.hpp"

struct LoadedEdge3D {
  int id1 = -1, id2 = -1;
  Eigen::Isometry3d transform;
  Eigen::Matrix<double, 6, 6> informationMatrix;
};

class MatchingResult {
 public:
  MatchingResult() : succeed_match(false) { final_trafo.setIdentity(); }
  LoadedEdge3D edge;
  Eigen::Matrix4f final_trafo;
  std::vector<cv::DMatch> inlier_matches;
  bool succeed_match;
};
