// Stand-in for VRO's transformation_estimation_euclidean.h (gtsam/gtsam_graph.cpp:41,494): rigid transform from matched
// 3-D feature pairs.  The synthetic front end answers with the relative pose it generated the two frames with.
#pragma once
#include <vector>
#include "camera_node.h"
Eigen::Matrix4f getTransformFromMatches(const CCameraNode *newer, const CCameraNode *older, const std::vector<cv::DMatch> &matches);
