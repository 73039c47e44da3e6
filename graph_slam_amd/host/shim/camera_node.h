// Stand-in for VRO's CCameraNode: ids + matchNodePair, served from the synthetic world.
#pragma once
#include "matching_result.h"
#include "cam_model.h"

class CCameraNode {
 public:
  CCameraNode() : m_id(-1), m_seq_id(-1), m_frame(-1) {}
  virtual ~CCameraNode() {}
  int m_id;        // graph id, assigned by the graph wrapper
  int m_seq_id;    // sequence id
  int m_frame;     // synthetic frame index (set by CSparseFeatureVO::featureExtraction)
  // relative pose of `this` expressed in `older` (edge older -> this), as VRO's RANSAC would return it
  MatchingResult matchNodePair(CCameraNode *older);
  static void set_cam_cov(const CamModel &) {}
  // covariance of the VRO estimate from the inlier set (gtsam/gtsam_graph.cpp:256-277): the synthetic front end returns
  // the inverse of the information matrix it generated the measurement with
  typedef Eigen::Matrix<double, 6, 1> (*cov_helper_fn)(Eigen::Matrix4f &);
  void computeCov(CCameraNode *older, std::vector<cv::DMatch> &inliers, cov_helper_fn helper, Eigen::Matrix<double, 6, 6> &cov);
};
