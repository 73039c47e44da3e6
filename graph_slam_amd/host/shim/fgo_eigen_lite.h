// Minimal stand-in for the slice of Eigen that the reference's graph wrappers and drivers touch
// (Eigen::Isometry3d, Matrix<double,6,6>, Matrix4f/4d, Vector4f, Quaterniond).  Eigen is not installed in
// this image; a deployment that has Eigen simply drops graph_slam_amd/host/shim from the include path and
// the same sources compile against the real library (only the members used below are relied upon).
#pragma once
#include <array>
#include <cmath>
#include <cstddef>

namespace Eigen {

template <typename T, int R, int C>
class Matrix {
 public:
  Matrix() { v_.fill(T(0)); }
  static Matrix Identity() { Matrix m; for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = T(1); return m; }
  static Matrix Zero() { return Matrix(); }
  T &operator()(int r, int c) { return v_[(size_t)r * C + c]; }
  const T &operator()(int r, int c) const { return v_[(size_t)r * C + c]; }
  T &operator()(int i) { return v_[(size_t)i]; }
  const T &operator()(int i) const { return v_[(size_t)i]; }
  T &operator[](int i) { return v_[(size_t)i]; }
  const T &operator[](int i) const { return v_[(size_t)i]; }
  Matrix operator*(T s) const { Matrix m(*this); for (auto &x : m.v_) x *= s; return m; }
  Matrix operator+(const Matrix &o) const { Matrix m(*this); for (size_t i = 0; i < v_.size(); ++i) m.v_[i] += o.v_[i]; return m; }
  template <int K>
  Matrix<T, R, K> operator*(const Matrix<T, C, K> &o) const {
    Matrix<T, R, K> m;
    for (int r = 0; r < R; ++r) for (int k = 0; k < K; ++k) { T s = 0; for (int c = 0; c < C; ++c) s += (*this)(r, c) * o(c, k); m(r, k) = s; }
    return m;
  }
  Matrix<T, C, R> transpose() const { Matrix<T, C, R> m; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) m(c, r) = (*this)(r, c); return m; }
  T trace() const { T s = 0; for (int i = 0; i < (R < C ? R : C); ++i) s += (*this)(i, i); return s; }
  T norm() const { T s = 0; for (auto x : v_) s += x * x; return std::sqrt(s); }
  T x() const { return v_[0]; }
  T y() const { return v_[1]; }
  T z() const { return v_[2]; }
  template <typename U> Matrix<U, R, C> cast() const { Matrix<U, R, C> m; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) m(r, c) = (U)(*this)(r, c); return m; }
  void setIdentity() { *this = Identity(); }
  void setZero() { v_.fill(T(0)); }
  const T *data() const { return v_.data(); }
  T *data() { return v_.data(); }
 private:
  std::array<T, (size_t)R * C> v_;
};
using Matrix3d = Matrix<double, 3, 3>;
using Matrix4d = Matrix<double, 4, 4>;
using Matrix4f = Matrix<float, 4, 4>;
using Vector3d = Matrix<double, 3, 1>;
using Vector4f = Matrix<float, 4, 1>;

class Quaterniond {
 public:
  Quaterniond() : x_(0), y_(0), z_(0), w_(1) {}
  Quaterniond(double w, double x, double y, double z) : x_(x), y_(y), z_(z), w_(w) {}
  explicit Quaterniond(const Matrix3d &R) { *this = R; }
  Quaterniond &operator=(const Matrix3d &R) {
    const double tr = R.trace();
    if (tr > 0) { double s = std::sqrt(tr + 1.0) * 2; w_ = 0.25 * s; x_ = (R(2, 1) - R(1, 2)) / s; y_ = (R(0, 2) - R(2, 0)) / s; z_ = (R(1, 0) - R(0, 1)) / s; }
    else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) { double s = std::sqrt(1.0 + R(0, 0) - R(1, 1) - R(2, 2)) * 2; w_ = (R(2, 1) - R(1, 2)) / s; x_ = 0.25 * s; y_ = (R(0, 1) + R(1, 0)) / s; z_ = (R(0, 2) + R(2, 0)) / s; }
    else if (R(1, 1) > R(2, 2)) { double s = std::sqrt(1.0 + R(1, 1) - R(0, 0) - R(2, 2)) * 2; w_ = (R(0, 2) - R(2, 0)) / s; x_ = (R(0, 1) + R(1, 0)) / s; y_ = 0.25 * s; z_ = (R(1, 2) + R(2, 1)) / s; }
    else { double s = std::sqrt(1.0 + R(2, 2) - R(0, 0) - R(1, 1)) * 2; w_ = (R(1, 0) - R(0, 1)) / s; x_ = (R(0, 2) + R(2, 0)) / s; y_ = (R(1, 2) + R(2, 1)) / s; z_ = 0.25 * s; }
    return *this;
  }
  double x() const { return x_; } double y() const { return y_; } double z() const { return z_; } double w() const { return w_; }
  Matrix3d toRotationMatrix() const {
    Matrix3d R;
    R(0, 0) = 1 - 2 * (y_ * y_ + z_ * z_); R(0, 1) = 2 * (x_ * y_ - z_ * w_); R(0, 2) = 2 * (x_ * z_ + y_ * w_);
    R(1, 0) = 2 * (x_ * y_ + z_ * w_); R(1, 1) = 1 - 2 * (x_ * x_ + z_ * z_); R(1, 2) = 2 * (y_ * z_ - x_ * w_);
    R(2, 0) = 2 * (x_ * z_ - y_ * w_); R(2, 1) = 2 * (y_ * z_ + x_ * w_); R(2, 2) = 1 - 2 * (x_ * x_ + y_ * y_);
    return R;
  }
 private:
  double x_, y_, z_, w_;
};

// rigid transform (R, t): the subset of Eigen::Isometry3d used by the wrappers
class Isometry3d {
 public:
  Isometry3d() { R_.setIdentity(); }
  static Isometry3d Identity() { return Isometry3d(); }
  Isometry3d(const Isometry3d &) = default;
  Isometry3d &operator=(const Isometry3d &) = default;
  Isometry3d(const Matrix3d &R, const Vector3d &t) : R_(R), t_(t) {}
  void setIdentity() { R_.setIdentity(); t_.setZero(); }
  const Vector3d &translation() const { return t_; }
  Vector3d &translation() { return t_; }
  const Matrix3d &rotation() const { return R_; }
  const Matrix3d &linear() const { return R_; }
  Matrix3d &linear() { return R_; }
  Matrix4d matrix() const {
    Matrix4d M = Matrix4d::Identity();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) M(r, c) = R_(r, c); M(r, 3) = t_(r); }
    return M;
  }
  Isometry3d inverse() const { Isometry3d o; o.R_ = R_.transpose(); Vector3d m = o.R_ * t_; for (int i = 0; i < 3; ++i) o.t_(i) = -m(i); return o; }
  Isometry3d operator*(const Isometry3d &b) const { Isometry3d o; o.R_ = R_ * b.R_; Vector3d m = R_ * b.t_; o.t_ = m + t_; return o; }
 private:
  Matrix3d R_;
  Vector3d t_;
};

}  // namespace Eigen
