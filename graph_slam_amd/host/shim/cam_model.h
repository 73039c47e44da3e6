#pragma once
// Stand-in for cam_model's CamModel (g2o/test_g2o_graph.cpp:48-49; gtsam/gtsam_graph.cpp:373,784-800; the drivers'
// set-up block gtsam/test_vro_imu_graph.cpp:84-89): pin-hole intrinsics with two radial terms, depth scale, the
// process-wide instance and the two projections the plane-propagation code calls.
class CamModel {
 public:
  CamModel(double fx_ = 0, double fy_ = 0, double cx_ = 0, double cy_ = 0, double k1_ = 0, double k2_ = 0)
      : fx(fx_), fy(fy_), cx(cx_), cy(cy_), k1(k1_), k2(k2_), z_offset(0), m_z_scale(1) {}
  double fx, fy, cx, cy, k1, k2, z_offset, m_z_scale;
  void setDepthScale(double s) { m_z_scale = s; }
  static CamModel *gCamModel() { static CamModel m; return &m; }
  static void updategCamModel(const CamModel &m) { *gCamModel() = m; }
  void convertUVZ2XYZ(float u, float v, double z, double &ox, double &oy, double &oz) const { ox = (u - cx) * z / fx; oy = (v - cy) * z / fy; oz = z; }
  void convertXYZ2UV(double x, double y, double z, float &u, float &v) const { u = (float)(fx * x / z + cx); v = (float)(fy * y / z + cy); }
};
