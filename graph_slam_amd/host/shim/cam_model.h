#pragma once
// Stand-in for cam_model's CamModel (test_g2o_graph.cpp:48-49)
class CamModel {
 public:
  CamModel(double fx_ = 0, double fy_ = 0, double cx_ = 0, double cy_ = 0, double k1_ = 0, double k2_ = 0)
      : fx(fx_), fy(fy_), cx(cx_), cy(cy_), k1(k1_), k2(k2_), z_offset(0), m_z_scale(1) {}
  double fx, fy, cx, cy, k1, k2, z_offset, m_z_scale;
  void setDepthScale(double s) { m_z_scale = s; }
};
