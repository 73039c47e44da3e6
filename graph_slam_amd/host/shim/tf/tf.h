// tf-free stand-in: tf::Transform / Quaternion / Vector3 as used by CGraphG2O (g2o/g2o_graph.h:44,
// g2o/misc.h:8-28, g2o/g2o_graph.cpp:299-303).
#pragma once
#include <cmath>
// the real tf.h pulls these in (roscpp / boost headers); the reference's g2o_graph.h relies on that (g2o/g2o_graph.h:39-51)
#include <fstream>
#include <iostream>
#include <string>

namespace tf {
class Vector3 {
 public:
  Vector3(double x = 0, double y = 0, double z = 0) : v_{x, y, z} {}
  double x() const { return v_[0]; } double y() const { return v_[1]; } double z() const { return v_[2]; }
  void setX(double a) { v_[0] = a; } void setY(double a) { v_[1] = a; } void setZ(double a) { v_[2] = a; }
 private:
  double v_[3];
};
class Quaternion {
 public:
  Quaternion(double x = 0, double y = 0, double z = 0, double w = 1) : q_{x, y, z, w} {}
  double x() const { return q_[0]; } double y() const { return q_[1]; } double z() const { return q_[2]; } double w() const { return q_[3]; }
  void setX(double a) { q_[0] = a; } void setY(double a) { q_[1] = a; } void setZ(double a) { q_[2] = a; } void setW(double a) { q_[3] = a; }
  Quaternion operator*(const Quaternion &b) const {
    return Quaternion(w() * b.x() + b.w() * x() + y() * b.z() - z() * b.y(), w() * b.y() + b.w() * y() + z() * b.x() - x() * b.z(),
                      w() * b.z() + b.w() * z() + x() * b.y() - y() * b.x(), w() * b.w() - x() * b.x() - y() * b.y() - z() * b.z());
  }
 private:
  double q_[4];
};
class Transform {
 public:
  Transform() {}
  Transform(const Quaternion &q, const Vector3 &t) : q_(q), t_(t) {}
  const Vector3 &getOrigin() const { return t_; }
  Quaternion getRotation() const { return q_; }
  void setOrigin(const Vector3 &t) { t_ = t; }
  void setRotation(const Quaternion &q) { q_ = q; }
  Vector3 operator*(const Vector3 &v) const {
    const double ux = q_.x(), uy = q_.y(), uz = q_.z(), w = q_.w();
    const double cx = uy * v.z() - uz * v.y(), cy = uz * v.x() - ux * v.z(), cz = ux * v.y() - uy * v.x();
    return Vector3(v.x() + 2 * (w * cx + uy * cz - uz * cy) + t_.x(), v.y() + 2 * (w * cy + uz * cx - ux * cz) + t_.y(),
                   v.z() + 2 * (w * cz + ux * cy - uy * cx) + t_.z());
  }
  Transform operator*(const Transform &b) const { return Transform(q_ * b.q_, (*this) * b.t_); }
 private:
  Quaternion q_;
  Vector3 t_;
};
}  // namespace tf
