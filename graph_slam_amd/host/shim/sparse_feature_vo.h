#pragma once
// Stand-in for VRO's CSparseFeatureVO (test_g2o_graph.cpp:51,73)
#include "SR_reader_cv.h"
#include "cam_model.h"
#include "camera_node.h"
class CSparseFeatureVO {
 public:
  explicit CSparseFeatureVO(const CamModel &) {}
  void featureExtraction(const cv::Mat &intensity, const cv::Mat &, float, CCameraNode &node) { node.m_frame = intensity.frame; }
};
