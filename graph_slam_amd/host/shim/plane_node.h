// Stand-in for the un-vendored `plane` package's CPlaneNode (gtsam/gtsam_graph.cpp:877-1100,1346-1503; the drivers'
// plane blocks gtsam/test_vro_imu_graph.cpp:202-314): the planes seen from one key frame.
#pragma once
#include <vector>
#include "plane.h"
#include "vro_synth.h"
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
  // Plane segmentation.  The real front end (RANSAC on the range image) is not reproduced; with the synthetic room of
  // shim/vro_synth.h the segmentation is known analytically: every pixel's ray hits one wall, a wall seen in at least
  // MIN_PIXELS pixels within the SR4000's working range becomes a CPlane in the CAMERA frame (n . p + d = 0, oriented so that
  // the camera is on the positive side), with a small deterministic perturbation standing in for the fitting noise.
  // Returns the number of planes found (0 without a synthetic room: the stand-in of rounds 1-2).
  int extractPlanes(cv::Mat &img, cv::Mat &dpt, CamModel *cam) {
    fgo_synth::World &w = fgo_synth::World::instance();
    double R[9], t[3];
    if (!w.has_room || dpt.empty() || !cam || !w.camera_pose(dpt.frame + 1, R, t)) return 0;
    enum { MIN_PIXELS = 1500 };
    std::vector<int> idx[6];
    for (int v = 0; v < dpt.rows; ++v)
      for (int u = 0; u < dpt.cols; ++u) {
        const double z = dpt.at<unsigned short>(v, u) * cam->m_z_scale;
        if (z <= 0.1 || z >= 5.0) continue;
        double zz; int wall;
        if (w.cast(R, t, (u - cam->cx) / cam->fx, (v - cam->cy) / cam->fy, zz, wall)) idx[wall].push_back(v * dpt.cols + u);
      }
    int found = 0;
    for (int wall = 0; wall < 6; ++wall) {
      if ((int)idx[wall].size() < MIN_PIXELS) continue;
      const int a = wall / 2;
      const double bound = (wall & 1) ? w.room_hi[a] : w.room_lo[a];
      // world plane e_a . p - bound = 0  ->  camera frame: n_c = R^T e_a, d_c = t_a - bound
      double n[3] = {R[3 * a], R[3 * a + 1], R[3 * a + 2]}, d = t[a] - bound;
      if (d < 0) { for (int k = 0; k < 3; ++k) n[k] = -n[k]; d = -d; }
      // fitting noise: deterministic in (frame, wall)
      unsigned s = 2654435761u * (unsigned)(dpt.frame * 6 + wall + 1);
      double e[4];
      for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; e[k] = ((s >> 8) / 16777216.0 - 0.5) * 2.0; }
      for (int k = 0; k < 3; ++k) n[k] += 0.002 * e[k];
      const double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      CPlane *p = new CPlane();
      p->nx_ = n[0] / nn; p->ny_ = n[1] / nn; p->nz_ = n[2] / nn; p->d1_ = d + 0.003 * e[3];
      mv_planes.push_back(p);
      mv_indices.push_back(idx[wall]);
      mv_landmark_id.push_back(-1);
      ++found;
    }
    if (found && empty()) setDpt(dpt);
    (void)img;
    return found;
  }
  int extractPlanes(CloudPtr &, CamModel *) { return 0; }      // planes in a left-over point cloud: not reproduced
  bool mergeOverlappedPlanes(int) { return false; }
};
