// Stand-in for the un-vendored `plane` package's CPlaneNode (gtsam/gtsam_graph.cpp:877-1100,1346-1503; the drivers'
// plane blocks gtsam/test_vro_imu_graph.cpp:202-314): the planes seen from one key frame.
#pragma once
#include <vector>
#include "plane.h"
class CPlaneNode {
 public:
  CPlaneNode() {}
  ~CPlaneNode() { for (size_t i = 0; i < mv_planes.size(); ++i) delete mv_planes[i]; }
  std::vector<CPlane *> mv_planes;
  std::vector<std::vector<int> > mv_indices;     // pixel indices of each plane
  std::vector<int> mv_landmark_id;               // landmark id of each plane, -1 = not associated
  cv::Mat m_dpt;
  bool empty() const { return m_dpt.empty(); }
  void setDpt(const cv::Mat &d) { m_dpt = d.clone(); }
  // plane segmentation (front end, not reproduced): returns the number of planes found
  int extractPlanes(cv::Mat &, cv::Mat &, CamModel *) { return 0; }
  int extractPlanes(CloudPtr &, CamModel *) { return 0; }
  bool mergeOverlappedPlanes(int) { return false; }
};
