// Stand-in for VRO's CCameraNodeBA (gtsam/gtsam_graph.cpp:370-448,450-610): per-feature 3-D locations, pixel positions
// and landmark ids, served from the synthetic world (vro_synth.cpp).
#pragma once
#include <map>
#include <vector>
#include "camera_node.h"
class CCameraNodeBA : public CCameraNode {
 public:
  std::vector<int> mv_feature_qid;                       // landmark id per feature, -1 = not yet a landmark
  std::vector<Eigen::Vector4f> m_feature_loc_3d;         // feature position in the camera frame (homogeneous)
  std::vector<cv::KeyPoint> m_feature_loc_2d;            // pixel position
  std::vector<int> mv_world_point;                       // synthetic world point behind each feature
  // feature correspondences {index in older -> index in this} given the relative pose (RANSAC inliers in VRO)
  std::map<int, int> matchNodePairBA(CCameraNodeBA *older, Eigen::Matrix4f &Tji, CamModel *pcam);
};
