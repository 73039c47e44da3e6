// The slice of the GTSAM 4.0 API that the reference's CGraphGT wrapper, imu_interface library and VIO / BA drivers bind
// (gtsam/gtsam_graph.cpp, gtsam/imu_base.cpp, gtsam/imu_vn100.cpp, gtsam/test_vro_imu_graph.cpp,
// gtsam/test_ba_imu_graph.cpp), re-implemented on the libfgo C-ABI so that those files compile UNCHANGED, in place,
// and every optimisation they ask for runs on the MI355X:
//   Values / NonlinearFactorGraph      plain host containers (keys -> values, factor descriptors), exactly GTSAM's
//                                      value semantics (copyable, default-constructible, assigned from optimize())
//   LevenbergMarquardtOptimizer        builds an fgo context from (graph, values), fgo_optimize_gtsam, reads back
//   ISAM2                              owns a persistent fgo context; update(newFactors, newTheta) appends to it and runs
//                                      fgo_isam2_update; calculateEstimate() reads the estimate back
//   Marginals                          fgo_marginal_cov on a context built lazily (the reference constructs one per plane
//                                      association and never uses it: gtsam/gtsam_graph.cpp:1357)
//   NonlinearFactorGraph::error        fgo_error (1/2 chi2) at the given values
// GTSAM is not installed in this image.  Nothing here evaluates the optimiser's factors on the CPU (libfgo has no CPU
// fallback); the host side holds only what a graph BUILDER needs: Pose3 / Rot3 / OrientedPlane3 algebra, noise model
// bookkeeping, and the IMU preintegration host loop of the C-ABI (fgo_preint_*).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <Eigen/Core>
#include <Eigen/Geometry>
extern "C" {
#include "fgo.h"
}

// GTSAM 4.0 still uses boost smart pointers (gtsam/imu_base.h:55, gtsam/gtsam_graph.cpp:373)
namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
using std::make_shared;
using std::dynamic_pointer_cast;
struct none_t {};
static const none_t none = none_t();
}  // namespace boost

namespace gtsam {

typedef Eigen::MatrixXd Matrix;
typedef Eigen::VectorXd Vector;
typedef Eigen::Matrix<double, 2, 1> Vector2;
typedef Eigen::Matrix<double, 3, 1> Vector3;
typedef Eigen::Matrix<double, 4, 1> Vector4;
typedef Eigen::Matrix<double, 6, 1> Vector6;
typedef Eigen::Matrix<double, 2, 2> Matrix2;
typedef Eigen::Matrix<double, 3, 3> Matrix3;
typedef Eigen::Matrix<double, 6, 6> Matrix6;
typedef Eigen::Matrix<double, 2, 2> Matrix22;
typedef Eigen::Matrix<double, 2, 3> Matrix23;
typedef Eigen::Matrix<double, 3, 2> Matrix32;
typedef Eigen::Matrix<double, 3, 3> Matrix33;
typedef Eigen::Matrix<double, 3, 6> Matrix36;
typedef Eigen::Matrix<double, 6, 6> Matrix66;
typedef Eigen::Matrix<double, 15, 15> Matrix15;
typedef Eigen::Quaterniond Quaternion;
typedef Vector3 Point3;          // GTSAM 4.0 built with GTSAM_TYPEDEF_POINTS_TO_VECTORS (gtsam_graph.cpp:388 uses operator<< on a Point3)
typedef Vector2 Point2;
typedef Vector3 Velocity3;
typedef uint64_t Key;

class Symbol {
 public:
  Symbol(unsigned char c, uint64_t j) : c_(c), j_(j) {}
  Symbol(Key k) : c_((unsigned char)(k >> 56)), j_(k & ((uint64_t(1) << 56) - 1)) {}
  operator Key() const { return key(); }
  Key key() const { return ((uint64_t)c_ << 56) | j_; }
  unsigned char chr() const { return c_; }
  uint64_t index() const { return j_; }
 private:
  unsigned char c_;
  uint64_t j_;
};
namespace symbol_shorthand {
inline Key mk(char c, uint64_t j) { return ((uint64_t)(unsigned char)c << 56) | j; }
inline Key A(uint64_t j) { return mk('a', j); }
inline Key B(uint64_t j) { return mk('b', j); }
inline Key L(uint64_t j) { return mk('l', j); }
inline Key P(uint64_t j) { return mk('p', j); }
inline Key Q(uint64_t j) { return mk('q', j); }
inline Key U(uint64_t j) { return mk('u', j); }
inline Key V(uint64_t j) { return mk('v', j); }
inline Key X(uint64_t j) { return mk('x', j); }
}  // namespace symbol_shorthand

static const Matrix3 I_3x3 = Matrix3::Identity();
static const Matrix3 Z_3x3 = Matrix3::Zero();
static const Matrix6 I_6x6 = Matrix6::Identity();

inline Matrix3 skewSymmetric(double wx, double wy, double wz) {
  Matrix3 S;
  S(0, 1) = -wz; S(0, 2) = wy; S(1, 0) = wz; S(1, 2) = -wx; S(2, 0) = -wy; S(2, 1) = wx;
  return S;
}
inline Matrix3 skew3(const Vector3 &w) { return skewSymmetric(w(0), w(1), w(2)); }

// ---- Unit3: direction on S^2 with the GTSAM 4.0 tangent basis (b1 = normalise(n x axis of smallest |n_i|), b2 = n x b1)
class Unit3 {
 public:
  Unit3() { p_(2) = 1.0; }
  explicit Unit3(const Vector3 &p) : p_(p.normalized()) {}
  Unit3(double x, double y, double z) { p_(0) = x; p_(1) = y; p_(2) = z; p_.normalize(); }
  const Vector3 &point3() const { return p_; }
  const Vector3 &unitVector() const { return p_; }
  Matrix32 basis() const {
    const double mx = std::fabs(p_(0)), my = std::fabs(p_(1)), mz = std::fabs(p_(2));
    Vector3 axis(0, 0, 1);
    if (mx <= my && mx <= mz) axis = Vector3(1, 0, 0);
    else if (my <= mx && my <= mz) axis = Vector3(0, 1, 0);
    const Vector3 b1 = p_.cross(axis).normalized(), b2 = p_.cross(b1);
    Matrix32 B;
    for (int k = 0; k < 3; ++k) { B(k, 0) = b1(k); B(k, 1) = b2(k); }
    return B;
  }
  Matrix3 skew() const { return skew3(p_); }
  Vector2 errorVector(const Unit3 &q) const { return basis().transpose() * q.p_; }
 private:
  Vector3 p_;
};

class Rot3 {
 public:
  Rot3() { R_.setIdentity(); }
  template <class D> explicit Rot3(const Eigen::MatrixBase<D, double, 3, 3> &R) : R_(R) {}
  explicit Rot3(const Quaternion &q) : R_(q.toRotationMatrix()) {}
  static Rot3 Rx(double t) { Matrix3 R = Matrix3::Identity(); R(1, 1) = std::cos(t); R(1, 2) = -std::sin(t); R(2, 1) = std::sin(t); R(2, 2) = std::cos(t); return Rot3(R); }
  static Rot3 Ry(double t) { Matrix3 R = Matrix3::Identity(); R(0, 0) = std::cos(t); R(0, 2) = std::sin(t); R(2, 0) = -std::sin(t); R(2, 2) = std::cos(t); return Rot3(R); }
  static Rot3 Rz(double t) { Matrix3 R = Matrix3::Identity(); R(0, 0) = std::cos(t); R(0, 1) = -std::sin(t); R(1, 0) = std::sin(t); R(1, 1) = std::cos(t); return Rot3(R); }
  // GTSAM: RzRyRx(x, y, z) = Rz(z) * Ry(y) * Rx(x)
  static Rot3 RzRyRx(double x, double y, double z) { return Rz(z) * Ry(y) * Rx(x); }
  static Rot3 Ypr(double y, double p, double r) { return RzRyRx(r, p, y); }
  static Rot3 Expmap(const Vector3 &w) {
    const double th2 = w.squaredNorm(), th = std::sqrt(th2);
    const Matrix3 W = skew3(w), W2 = W * W;
    double a, b;
    if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; } else { a = std::sin(th) / th; b = (1.0 - std::cos(th)) / th2; }
    return Rot3(Matrix3::Identity() + W * a + W2 * b);
  }
  static Vector3 Logmap(const Rot3 &R) {
    const Quaternion q(R.matrix());
    double w = q.w(), x = q.x(), y = q.y(), z = q.z();
    if (w < 0) { w = -w; x = -x; y = -y; z = -z; }
    const double n = std::sqrt(x * x + y * y + z * z);
    const double k = n < 1e-12 ? 2.0 : 2.0 * std::atan2(n, w) / n;
    return Vector3(k * x, k * y, k * z);
  }
  Rot3 operator*(const Rot3 &o) const { return Rot3(R_ * o.R_); }
  Vector3 operator*(const Vector3 &p) const { return R_ * p; }
  Vector3 rotate(const Vector3 &p) const { return R_ * p; }
  Vector3 unrotate(const Vector3 &p) const { return R_.transpose() * p; }
  // Unit3 form with the Jacobians OrientedPlane3::transform needs: HR (2x3, wrt this rotation), Hp (2x2, wrt p)
  Unit3 unrotate(const Unit3 &p, Matrix23 *HR = 0, Matrix22 *Hp = 0) const {
    const Unit3 q(unrotate(p.point3()));
    if (Hp) *Hp = q.basis().transpose() * R_.transpose() * p.basis();
    if (HR) *HR = q.basis().transpose() * q.skew();
    return q;
  }
  Rot3 inverse() const { return Rot3(R_.transpose()); }
  const Matrix3 &matrix() const { return R_; }
  Quaternion toQuaternion() const { return Quaternion(R_); }
  Vector3 rpy() const { return Vector3(std::atan2(R_(2, 1), R_(2, 2)), std::asin(-R_(2, 0)), std::atan2(R_(1, 0), R_(0, 0))); }
  void print(const std::string &s = "") const { std::cout << s << "\n" << R_ << std::endl; }
 private:
  Matrix3 R_;
};

class Pose3 {
 public:
  Pose3() {}
  Pose3(const Rot3 &R, const Point3 &t) : R_(R), t_(t) {}
  template <class D> explicit Pose3(const Eigen::MatrixBase<D, double, 4, 4> &M) : R_(M.template block<3, 3>(0, 0)), t_(M.template block<3, 1>(0, 3)) {}
  static Pose3 Create(const Rot3 &R, const Point3 &t) { return Pose3(R, t); }
  // full exponential map, tangent [omega; v] (the chart libfgo retracts with: DESIGN.md, GTSAM_POSE3_EXPMAP)
  static Pose3 Expmap(const Vector6 &xi) {
    const Vector3 w(xi(0), xi(1), xi(2)), v(xi(3), xi(4), xi(5));
    const double th2 = w.squaredNorm(), th = std::sqrt(th2);
    const Matrix3 W = skew3(w), W2 = W * W;
    double b, c;
    if (th < 1e-8) { b = 0.5 - th2 / 24.0; c = 1.0 / 6.0 - th2 / 120.0; } else { b = (1.0 - std::cos(th)) / th2; c = (th - std::sin(th)) / (th2 * th); }
    const Matrix3 Vm = Matrix3::Identity() + W * b + W2 * c;
    return Pose3(Rot3::Expmap(w), Vm * v);
  }
  static Vector6 Logmap(const Pose3 &p) {
    const Vector3 w = Rot3::Logmap(p.R_);
    const double th2 = w.squaredNorm(), th = std::sqrt(th2);
    const Matrix3 W = skew3(w), W2 = W * W;
    double c;     // V^-1 = I - W/2 + c W^2
    if (th < 1e-6) c = 1.0 / 12.0 + th2 / 720.0; else c = (1.0 - 0.5 * th * std::sin(th) / (1.0 - std::cos(th))) / th2;
    const Matrix3 Vi = Matrix3::Identity() + W * (-0.5) + W2 * c;
    const Vector3 v = Vi * p.t_;
    Vector6 xi;
    for (int k = 0; k < 3; ++k) { xi(k) = w(k); xi(3 + k) = v(k); }
    return xi;
  }
  struct ChartAtOrigin {
    static Pose3 Retract(const Vector6 &xi) { return Pose3::Expmap(xi); }
    static Vector6 Local(const Pose3 &p) { return Pose3::Logmap(p); }
  };
  Pose3 operator*(const Pose3 &o) const { return Pose3(R_ * o.R_, R_ * o.t_ + t_); }
  Point3 operator*(const Point3 &p) const { return R_ * p + t_; }
  Pose3 compose(const Pose3 &o) const { return (*this) * o; }
  Pose3 inverse() const { const Rot3 Ri = R_.inverse(); return Pose3(Ri, -(Ri * t_)); }
  Pose3 between(const Pose3 &o) const { return inverse() * o; }
  Point3 transform_from(const Point3 &p) const { return R_ * p + t_; }
  Point3 transform_to(const Point3 &p) const { return R_.unrotate(p - t_); }
  Pose3 transform_pose_to(const Pose3 &pose) const { return inverse() * pose; }   // GTSAM 4.0: pose expressed in this frame
  const Rot3 &rotation() const { return R_; }
  const Point3 &translation() const { return t_; }
  double x() const { return t_(0); }
  double y() const { return t_(1); }
  double z() const { return t_(2); }
  Eigen::Matrix4d matrix() const {
    Eigen::Matrix4d M = Eigen::Matrix4d::Identity();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) M(r, c) = R_.matrix()(r, c); M(r, 3) = t_(r); }
    return M;
  }
  // [R 0; [t]x R  R], tangent [omega; v]
  Matrix6 AdjointMap() const {
    Matrix6 A;
    const Matrix3 &R = R_.matrix();
    const Matrix3 TR = skew3(t_) * R;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) { A(r, c) = R(r, c); A(3 + r, c) = TR(r, c); A(3 + r, 3 + c) = R(r, c); }
    return A;
  }
  void print(const std::string &s = "") const {
    std::cout << s << "\nR:\n" << R_.matrix() << "\nt: " << t_(0) << " " << t_(1) << " " << t_(2) << std::endl;
  }
 private:
  Rot3 R_;
  Point3 t_;
};

class NavState {
 public:
  NavState() {}
  NavState(const Pose3 &p, const Vector3 &v) : p_(p), v_(v) {}
  NavState(const Rot3 &R, const Point3 &t, const Vector3 &v) : p_(R, t), v_(v) {}
  const Pose3 &pose() const { return p_; }
  const Rot3 &attitude() const { return p_.rotation(); }
  const Point3 &position() const { return p_.translation(); }
  const Vector3 &v() const { return v_; }
  const Vector3 &velocity() const { return v_; }
  void print(const std::string &s = "") const { p_.print(s); std::cout << "v: " << v_(0) << " " << v_(1) << " " << v_(2) << std::endl; }
 private:
  Pose3 p_;
  Vector3 v_;
};

namespace imuBias {
class ConstantBias {
 public:
  ConstantBias() {}
  ConstantBias(const Vector3 &acc, const Vector3 &gyro) : a_(acc), g_(gyro) {}
  explicit ConstantBias(const Vector6 &v) : a_(v(0), v(1), v(2)), g_(v(3), v(4), v(5)) {}
  const Vector3 &accelerometer() const { return a_; }
  const Vector3 &gyroscope() const { return g_; }
  Vector6 vector() const { Vector6 o; for (int k = 0; k < 3; ++k) { o(k) = a_(k); o(3 + k) = g_(k); } return o; }
  void print(const std::string &s = "") const { std::cout << s << " acc " << a_.transpose() << " gyro " << g_.transpose() << std::endl; }
 private:
  Vector3 a_, g_;
};
}  // namespace imuBias

// unit normal + distance (a, b, c, d); transform / error as in GTSAM 4.0 (pinned by gtsam/test/testOrientedPlane3.cpp)
class OrientedPlane3 {
 public:
  OrientedPlane3() : d_(0) {}
  OrientedPlane3(const Unit3 &n, double d) : n_(n), d_(d) {}
  OrientedPlane3(double a, double b, double c, double d) : n_(a, b, c), d_(d) {}
  explicit OrientedPlane3(const Vector4 &v) : n_(v(0), v(1), v(2)), d_(v(3)) {}
  // plane expressed in the frame of pose xr (world -> xr): n' = R^T n, d' = n . t + d.
  // Hp (3x3): d result / d this plane; Hr (3x6): d result / d pose  (local coordinates of the result)
  OrientedPlane3 transform(const Pose3 &xr, Matrix33 *Hp = 0, Matrix36 *Hr = 0) const {
    Matrix23 D_rotated_plane;
    Matrix22 D_rotated_pose;
    const Unit3 n_rotated = xr.rotation().unrotate(n_, &D_rotated_plane, &D_rotated_pose);
    const Vector3 u = n_rotated.unitVector();
    const double pred_d = n_.unitVector().dot(xr.translation()) + d_;
    if (Hr) {
      Hr->setZero();
      Hr->block<2, 3>(0, 0) = D_rotated_plane;
      for (int c = 0; c < 3; ++c) (*Hr)(2, 3 + c) = u(c);
    }
    if (Hp) {
      const Vector2 hpp = n_.basis().transpose() * xr.translation();
      Hp->setZero();
      Hp->block<2, 2>(0, 0) = D_rotated_pose;
      (*Hp)(2, 0) = hpp(0); (*Hp)(2, 1) = hpp(1); (*Hp)(2, 2) = 1.0;
    }
    return OrientedPlane3(u(0), u(1), u(2), pred_d);
  }
  OrientedPlane3 transform(const Pose3 &xr, Matrix33 &Hp) const { return transform(xr, &Hp, 0); }
  OrientedPlane3 transform(const Pose3 &xr, Matrix33 &Hp, Matrix36 &Hr) const { return transform(xr, &Hp, &Hr); }
  Vector3 errorVector(const OrientedPlane3 &o) const { const Vector2 e = n_.errorVector(o.n_); return Vector3(e(0), e(1), d_ - o.d_); }
  Vector3 error(const OrientedPlane3 &o) const { return errorVector(o); }     // diagnostic use only (gtsam_graph.cpp:1218)
  Vector4 planeCoefficients() const { const Vector3 &u = n_.unitVector(); return Vector4(u(0), u(1), u(2), d_); }
  const Unit3 &normal() const { return n_; }
  double distance() const { return d_; }
  void print(const std::string &s = "") const { std::cout << s << " : " << planeCoefficients().transpose() << std::endl; }
 private:
  Unit3 n_;
  double d_;
};

class Cal3_S2 {
 public:
  Cal3_S2(double fx = 1, double fy = 1, double s = 0, double u0 = 0, double v0 = 0) : fx_(fx), fy_(fy), s_(s), u0_(u0), v0_(v0) {}
  double fx() const { return fx_; } double fy() const { return fy_; } double skew() const { return s_; } double px() const { return u0_; } double py() const { return v0_; }
 protected:
  double fx_, fy_, s_, u0_, v0_;
};
class Cal3DS2 : public Cal3_S2 {
 public:
  Cal3DS2(double fx = 1, double fy = 1, double s = 0, double u0 = 0, double v0 = 0, double k1 = 0, double k2 = 0, double p1 = 0, double p2 = 0)
      : Cal3_S2(fx, fy, s, u0, v0), k1_(k1), k2_(k2), p1_(p1), p2_(p2) {}
  double k1() const { return k1_; } double k2() const { return k2_; } double p1() const { return p1_; } double p2() const { return p2_; }
 private:
  double k1_, k2_, p1_, p2_;
};

namespace noiseModel {
// every model is reduced to what the C-ABI takes: a dense information matrix (dim <= 6), a covariance, or a sigma
struct Base {
  int dim = 0;
  Matrix6 info;        // information (inverse covariance), top-left dim x dim
  Matrix6 cov;         // covariance when the model was given as one (plane factors take covariances)
  bool has_cov = false;
  double sigma = 0;    // isotropic / first sigma
  typedef std::shared_ptr<Base> shared_ptr;
};
struct Gaussian : Base {
  typedef std::shared_ptr<Base> shared_ptr;
  template <class D, int R, int C> static shared_ptr Information(const Eigen::MatrixBase<D, double, R, C> &M) {
    auto m = std::make_shared<Base>(); m->dim = M.rows();
    for (int r = 0; r < M.rows() && r < 6; ++r) for (int c = 0; c < M.cols() && c < 6; ++c) m->info(r, c) = M(r, c);
    return m;
  }
  template <class D, int R, int C> static shared_ptr Covariance(const Eigen::MatrixBase<D, double, R, C> &S) {
    auto m = std::make_shared<Base>(); m->dim = S.rows(); m->has_cov = true;
    for (int r = 0; r < S.rows() && r < 6; ++r) for (int c = 0; c < S.cols() && c < 6; ++c) m->cov(r, c) = S(r, c);
    const Eigen::MatrixXd inv = S.inverse();
    for (int r = 0; r < S.rows() && r < 6; ++r) for (int c = 0; c < S.cols() && c < 6; ++c) m->info(r, c) = inv(r, c);
    return m;
  }
};
struct Diagonal : Base {
  typedef std::shared_ptr<Base> shared_ptr;
  template <class D, int R, int C> static shared_ptr Sigmas(const Eigen::MatrixBase<D, double, R, C> &s) {
    auto m = std::make_shared<Base>(); m->dim = s.size(); m->sigma = s(0);
    for (int k = 0; k < s.size() && k < 6; ++k) m->info(k, k) = 1.0 / (s(k) * s(k));
    return m;
  }
  template <class D, int R, int C> static shared_ptr Variances(const Eigen::MatrixBase<D, double, R, C> &v) {
    auto m = std::make_shared<Base>(); m->dim = v.size(); m->sigma = std::sqrt(v(0));
    for (int k = 0; k < v.size() && k < 6; ++k) m->info(k, k) = 1.0 / v(k);
    return m;
  }
};
struct Isotropic : Base {
  typedef std::shared_ptr<Base> shared_ptr;
  static shared_ptr Sigma(int dim, double s) {
    auto m = std::make_shared<Base>(); m->dim = dim; m->sigma = s;
    for (int k = 0; k < dim && k < 6; ++k) m->info(k, k) = 1.0 / (s * s);
    return m;
  }
};
}  // namespace noiseModel
typedef std::shared_ptr<noiseModel::Base> SharedNoiseModel;

// ---- IMU preintegration (host side of the C-ABI: fgo_preint_*, csrc/imu_preint.cpp)
struct PreintegrationParams {
  Matrix3 accelerometerCovariance, gyroscopeCovariance, integrationCovariance;
  Vector3 n_gravity;
};
struct PreintegrationCombinedParams : PreintegrationParams {
  Matrix3 biasAccCovariance, biasOmegaCovariance;
  Matrix6 biasAccOmegaInt;
  PreintegrationCombinedParams() {
    accelerometerCovariance.setIdentity(); gyroscopeCovariance.setIdentity(); integrationCovariance.setIdentity();
    biasAccCovariance.setIdentity(); biasOmegaCovariance.setIdentity(); biasAccOmegaInt.setIdentity();
  }
  static std::shared_ptr<PreintegrationCombinedParams> MakeSharedD(double g = 9.81) {      // Z-down navigation frame: n_gravity = (0, 0, +g)
    auto o = std::make_shared<PreintegrationCombinedParams>(); o->n_gravity = Vector3(0, 0, g); return o;
  }
  static std::shared_ptr<PreintegrationCombinedParams> MakeSharedU(double g = 9.81) {      // Z-up: n_gravity = (0, 0, -g)
    auto o = std::make_shared<PreintegrationCombinedParams>(); o->n_gravity = Vector3(0, 0, -g); return o;
  }
  // the C-ABI takes isotropic variances (what gtsam/imu_vn100.cpp:46-62 and imu_MEMS.cpp:22-33 set)
  fgo_imu_params c_params() const {
    fgo_imu_params p;
    p.acc_cov = accelerometerCovariance(0, 0); p.gyro_cov = gyroscopeCovariance(0, 0); p.integ_cov = integrationCovariance(0, 0);
    p.bias_acc_cov = biasAccCovariance(0, 0); p.bias_gyro_cov = biasOmegaCovariance(0, 0); p.bias_acc_omega_int = biasAccOmegaInt(0, 0);
    for (int k = 0; k < 3; ++k) p.gravity[k] = n_gravity(k);
    return p;
  }
};
// gtsam::PreintegrationType: the base the imu_interface library holds a pointer to (gtsam/imu_base.h:73)
class PreintegrationType {
 public:
  virtual ~PreintegrationType() {}
  virtual void integrateMeasurement(const Vector3 &acc, const Vector3 &gyro, double dt) = 0;
  virtual void resetIntegrationAndSetBias(const imuBias::ConstantBias &b) = 0;
  virtual NavState predict(const NavState &s, const imuBias::ConstantBias &b) const = 0;
};
class PreintegratedCombinedMeasurements : public PreintegrationType {
 public:
  typedef PreintegrationCombinedParams Params;
  PreintegratedCombinedMeasurements() : params_(Params::MakeSharedD(9.81)) { reset(imuBias::ConstantBias()); }
  PreintegratedCombinedMeasurements(const std::shared_ptr<Params> &p, const imuBias::ConstantBias &b = imuBias::ConstantBias()) : params_(p) { reset(b); }
  void integrateMeasurement(const Vector3 &acc, const Vector3 &gyro, double dt) {
    const double a[3] = {acc(0), acc(1), acc(2)}, g[3] = {gyro(0), gyro(1), gyro(2)};
    const fgo_imu_params p = params_->c_params();
    fgo_preint_integrate(&m_, &p, a, g, dt);
  }
  void resetIntegrationAndSetBias(const imuBias::ConstantBias &b) { reset(b); }
  void resetIntegration() { imuBias::ConstantBias b(Vector3(m_.bhat[0], m_.bhat[1], m_.bhat[2]), Vector3(m_.bhat[3], m_.bhat[4], m_.bhat[5])); reset(b); }
  NavState predict(const NavState &s, const imuBias::ConstantBias &b) const {
    const Quaternion q = s.pose().rotation().toQuaternion();
    const double pi[7] = {s.pose().x(), s.pose().y(), s.pose().z(), q.x(), q.y(), q.z(), q.w()};
    const double vi[3] = {s.v()(0), s.v()(1), s.v()(2)};
    const Vector6 bv = b.vector();
    const double b6[6] = {bv(0), bv(1), bv(2), bv(3), bv(4), bv(5)};
    const double g[3] = {params_->n_gravity(0), params_->n_gravity(1), params_->n_gravity(2)};
    double pj[7], vj[3];
    fgo_preint_predict(&m_, g, pi, vi, b6, pj, vj);
    return NavState(Pose3(Rot3(Quaternion(pj[6], pj[3], pj[4], pj[5])), Point3(pj[0], pj[1], pj[2])), Vector3(vj[0], vj[1], vj[2]));
  }
  Matrix15 preintMeasCov() const { Matrix15 M; for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) M(r, c) = m_.cov[r * 15 + c]; return M; }
  double deltaTij() const { return m_.dt; }
  const fgo_preint &raw() const { return m_; }
  const Params &p() const { return *params_; }
  void print(const std::string &s = "") const { std::printf("%s preintegrated dt %g\n", s.c_str(), m_.dt); }
 private:
  void reset(const imuBias::ConstantBias &b) { const Vector6 v = b.vector(); const double b6[6] = {v(0), v(1), v(2), v(3), v(4), v(5)}; fgo_preint_reset(&m_, b6); }
  std::shared_ptr<Params> params_;
  fgo_preint m_;
};

// ---- values: key -> (kind, 7 doubles), kinds as in the C-ABI (0 pose, 1 plane, 2 point, 3 vec3, 4 bias)
struct ValueRec { int kind; double v[7]; };
template <class T> struct ValueTraits;
template <> struct ValueTraits<Pose3> {
  static ValueRec pack(const Pose3 &p) { ValueRec r; r.kind = 0; const Quaternion q = p.rotation().toQuaternion(); r.v[0] = p.x(); r.v[1] = p.y(); r.v[2] = p.z(); r.v[3] = q.x(); r.v[4] = q.y(); r.v[5] = q.z(); r.v[6] = q.w(); return r; }
  static Pose3 unpack(const ValueRec &r) { return Pose3(Rot3(Quaternion(r.v[6], r.v[3], r.v[4], r.v[5])), Point3(r.v[0], r.v[1], r.v[2])); }
};
template <> struct ValueTraits<OrientedPlane3> {
  static ValueRec pack(const OrientedPlane3 &p) { ValueRec r = {1, {0, 0, 0, 0, 0, 0, 0}}; const Vector4 c = p.planeCoefficients(); for (int k = 0; k < 4; ++k) r.v[k] = c(k); return r; }
  static OrientedPlane3 unpack(const ValueRec &r) { return OrientedPlane3(r.v[0], r.v[1], r.v[2], r.v[3]); }
};
template <> struct ValueTraits<Vector3> {          // Point3 and velocity share the C++ type; the key's character tells them apart
  static ValueRec pack(const Vector3 &p) { ValueRec r = {3, {p(0), p(1), p(2), 0, 0, 0, 0}}; return r; }
  static Vector3 unpack(const ValueRec &r) { return Vector3(r.v[0], r.v[1], r.v[2]); }
};
template <> struct ValueTraits<imuBias::ConstantBias> {
  static ValueRec pack(const imuBias::ConstantBias &b) { ValueRec r = {4, {0, 0, 0, 0, 0, 0, 0}}; const Vector6 v = b.vector(); for (int k = 0; k < 6; ++k) r.v[k] = v(k); return r; }
  static imuBias::ConstantBias unpack(const ValueRec &r) { return imuBias::ConstantBias(Vector3(r.v[0], r.v[1], r.v[2]), Vector3(r.v[3], r.v[4], r.v[5])); }
};

class Values {
 public:
  typedef std::map<Key, ValueRec> Map;
  template <class T> void insert(Key k, const T &v) { ValueRec r = ValueTraits<T>::pack(v); if (r.kind == 3 && landmark_key(k)) r.kind = 2; m_[k] = r; }
  template <class T> void update(Key k, const T &v) { insert<T>(k, v); }
  void insert(const Values &o) { for (Map::const_iterator it = o.m_.begin(); it != o.m_.end(); ++it) m_[it->first] = it->second; }
  template <class T> T at(Key k) const {
    Map::const_iterator it = m_.find(k);
    if (it == m_.end()) { std::fprintf(stderr, "gtsam shim: Values::at: key %c%llu does not exist\n", (char)(k >> 56), (unsigned long long)(k & 0xffffffffffffffULL)); return T(); }
    return ValueTraits<T>::unpack(it->second);
  }
  bool exists(Key k) const { return m_.count(k) != 0; }
  size_t size() const { return m_.size(); }
  bool empty() const { return m_.empty(); }
  void clear() { m_.clear(); }
  void erase(Key k) { m_.erase(k); }
  const Map &map() const { return m_; }
  Map &map() { return m_; }
  void print(const std::string &s = "") const { std::printf("%s Values with %zu entries\n", s.c_str(), m_.size()); }
  // a Vector3 under a landmark-like key character is a Point3 (3-dof landmark block), otherwise a velocity
  static bool landmark_key(Key k) { const char c = (char)(k >> 56); return c == 'q' || c == 'u' || c == 'p' || c == 'l'; }
 private:
  Map m_;
};

// ---- factors: plain descriptors, translated into C-ABI calls when a context is built from a graph
struct FactorDesc {
  enum Kind { PRIOR_POSE, PRIOR_VEC3, PRIOR_BIAS, PRIOR_POINT, BETWEEN, IMU, PLANE, REPROJ } kind;
  Key k[6];
  int nk;
  double t[3], q[4];          // pose payload
  double v6[6];               // vector / bias / plane / pixel payload
  double info21[21];
  double cov6[6];
  double sigma;
  double calib[9], bps[7];    // reprojection: Cal3DS2 (fx fy s u0 v0 k1 k2 p1 p2) and body_P_sensor
  double gravity[3];          // IMU: n_gravity of the preintegration parameters
  fgo_preint pim;
};
inline void pose_payload(const Pose3 &p, double t[3], double q[4]) {
  const Quaternion qq = p.rotation().toQuaternion();
  t[0] = p.x(); t[1] = p.y(); t[2] = p.z();
  q[0] = qq.x(); q[1] = qq.y(); q[2] = qq.z(); q[3] = qq.w();
}
inline void info_ut21(const Matrix6 &M, double out[21]) { int k = 0; for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) out[k++] = M(r, c); }

struct FactorBase { FactorDesc d; };
template <class T> struct PriorFactor;
template <> struct PriorFactor<Pose3> : FactorBase {
  PriorFactor(Key k, const Pose3 &mean, const SharedNoiseModel &n) { d.kind = FactorDesc::PRIOR_POSE; d.k[0] = k; d.nk = 1; pose_payload(mean, d.t, d.q); info_ut21(n->info, d.info21); }
};
template <> struct PriorFactor<Vector3> : FactorBase {          // PriorFactor<Vector3> (velocity) and PriorFactor<Point3> (landmark)
  PriorFactor(Key k, const Vector3 &mean, const SharedNoiseModel &n) {
    d.kind = Values::landmark_key(k) ? FactorDesc::PRIOR_POINT : FactorDesc::PRIOR_VEC3; d.k[0] = k; d.nk = 1;
    for (int i = 0; i < 3; ++i) d.v6[i] = mean(i);
    d.sigma = n->sigma;
  }
};
template <> struct PriorFactor<imuBias::ConstantBias> : FactorBase {
  PriorFactor(Key k, const imuBias::ConstantBias &mean, const SharedNoiseModel &n) { d.kind = FactorDesc::PRIOR_BIAS; d.k[0] = k; d.nk = 1; const Vector6 v = mean.vector(); for (int i = 0; i < 6; ++i) d.v6[i] = v(i); d.sigma = n->sigma; }
};
template <class T> struct BetweenFactor;
template <> struct BetweenFactor<Pose3> : FactorBase {
  BetweenFactor(Key i, Key j, const Pose3 &z, const SharedNoiseModel &n) { d.kind = FactorDesc::BETWEEN; d.k[0] = i; d.k[1] = j; d.nk = 2; pose_payload(z, d.t, d.q); info_ut21(n->info, d.info21); }
};
struct CombinedImuFactor : FactorBase {
  CombinedImuFactor(Key xi, Key vi, Key xj, Key vj, Key bi, Key bj, const PreintegratedCombinedMeasurements &pim) {
    d.kind = FactorDesc::IMU; d.k[0] = xi; d.k[1] = vi; d.k[2] = xj; d.k[3] = vj; d.k[4] = bi; d.k[5] = bj; d.nk = 6; d.pim = pim.raw();
    for (int i = 0; i < 3; ++i) d.gravity[i] = pim.p().n_gravity(i);
  }
};
// GenericProjectionFactor<Pose3, Point3, Cal3DS2>(measured, noise, poseKey, pointKey, K, throwCheirality, verboseCheirality,
// body_P_sensor): gtsam/gtsam_graph.cpp:405-409, 580-581
template <class POSE, class LANDMARK, class CALIBRATION>
struct GenericProjectionFactor : FactorBase {
  GenericProjectionFactor(const Point2 &z, const SharedNoiseModel &n, Key pose, Key point, const std::shared_ptr<CALIBRATION> &K) { init(z, n, pose, point, *K, Pose3()); }
  GenericProjectionFactor(const Point2 &z, const SharedNoiseModel &n, Key pose, Key point, const std::shared_ptr<CALIBRATION> &K, bool, bool, const Pose3 &body_P_sensor) { init(z, n, pose, point, *K, body_P_sensor); }
 private:
  void init(const Point2 &z, const SharedNoiseModel &n, Key pose, Key point, const Cal3DS2 &K, const Pose3 &bps) {
    d.kind = FactorDesc::REPROJ; d.k[0] = pose; d.k[1] = point; d.nk = 2; d.v6[0] = z(0); d.v6[1] = z(1); d.sigma = n->sigma;
    const double c[9] = {K.fx(), K.fy(), K.skew(), K.px(), K.py(), K.k1(), K.k2(), K.p1(), K.p2()};
    for (int i = 0; i < 9; ++i) d.calib[i] = c[i];
    pose_payload(bps, d.bps, d.bps + 3);
  }
};
class Values;
struct OrientedPlane3Factor : FactorBase {
  OrientedPlane3Factor(const Vector4 &z, const SharedNoiseModel &n, Key pose, Key landmark) : z_(z), noise_(n) {
    d.kind = FactorDesc::PLANE; d.k[0] = pose; d.k[1] = landmark; d.nk = 2;
    for (int i = 0; i < 4; ++i) d.v6[i] = z(i);
    int k = 0;
    for (int r = 0; r < 3; ++r) for (int c = r; c < 3; ++c) d.cov6[k++] = n->cov(r, c);
  }
  // Single-factor diagnostics for the graph BUILDER (gtsam/gtsam_graph.cpp:1267): host arithmetic on two values, not
  // the optimiser's evaluation of the graph (that is k_linearize_gtsam on the device).
  Vector3 unwhitenedError(const Values &v) const {
    const OrientedPlane3 pred = v.at<OrientedPlane3>(d.k[1]).transform(v.at<Pose3>(d.k[0]));
    return pred.errorVector(OrientedPlane3(z_));
  }
  double error(const Values &v) const {
    const Vector3 e = unwhitenedError(v);
    double s = 0;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) s += e(r) * noise_->info(r, c) * e(c);
    return 0.5 * s;
  }
  void print(const std::string &s = "") const { std::cout << s << " OrientedPlane3Factor z = " << z_.transpose() << std::endl; }
 private:
  Vector4 z_;
  SharedNoiseModel noise_;
};

class NonlinearFactorGraph {
 public:
  template <class F> void add(const F &f) { f_.push_back(f.d); }
  template <class F> void push_back(const F &f) { f_.push_back(f.d); }
  size_t size() const { return f_.size(); }
  bool empty() const { return f_.empty(); }
  void resize(size_t n) { f_.resize(n); }
  const std::vector<FactorDesc> &factors() const { return f_; }
  double error(const Values &v) const;                                  // 1/2 sum |whitened residual|^2: fgo_error
  void saveGraph(std::ostream &os, const Values &v = Values()) const;   // graphviz, like GTSAM's
  void print(const std::string &s = "") const { std::printf("%s NonlinearFactorGraph with %zu factors\n", s.c_str(), f_.size()); }
 private:
  std::vector<FactorDesc> f_;
};

// ---- the bridge: a libfgo context fed from the containers (gtsam_bridge.cpp)
class FgoBridge {
 public:
  FgoBridge();
  ~FgoBridge();
  bool ok() const { return ctx_ != 0; }
  fgo_ctx *ctx() { return ctx_; }
  // variables not yet known to the context are added, known ones get their value set; then factors [first, size) are added
  bool load(const NonlinearFactorGraph &g, size_t first_factor, const Values &v, bool set_existing);
  bool read_back(Values &v) const;                                      // every variable the context knows
  const std::map<Key, int> &kinds() const { return kinds_; }
 private:
  FgoBridge(const FgoBridge &);
  FgoBridge &operator=(const FgoBridge &);
  bool add_factor(const FactorDesc &d);
  fgo_ctx *ctx_;
  std::map<Key, int> kinds_;
  bool calib_set_, gravity_set_;
  double calib_[16], gravity_[3];
};

class LevenbergMarquardtParams {
 public:
  int maxIterations = 100;
  void setVerbosity(const std::string &) {}
};
class LevenbergMarquardtOptimizer {
 public:
  LevenbergMarquardtOptimizer(const NonlinearFactorGraph &g, const Values &v, const LevenbergMarquardtParams &p = LevenbergMarquardtParams())
      : g_(g), v_(v), p_(p), iterations_(0), error_(0) {}
  // GTSAM 4.0 default parameters (maxIterations 100, lambdaInitial 1e-5, ...): fgo_optimize_gtsam
  const Values &optimize();
  int iterations() const { return iterations_; }
  double error() const { return error_; }
  const Values &values() const { return v_; }
 private:
  const NonlinearFactorGraph &g_;
  Values v_;
  LevenbergMarquardtParams p_;
  int iterations_;
  double error_;
};

// ISAM2Params / ISAM2 as CGraphGT uses them (gtsam/gtsam_graph.cpp:93-99, 529-532, 1768-1776)
struct ISAM2Params {
  double relinearizeThreshold = 0.1;     // GTSAM 4.0 defaults
  int relinearizeSkip = 10;
};
struct ISAM2Result { int variablesRelinearized = 0; double errorAfter = 0; };
class ISAM2 {
 public:
  explicit ISAM2(const ISAM2Params &p = ISAM2Params()) : p_(p), count_(0), loaded_(0) {}
  // update(newFactors, newTheta): relinearisation is considered every relinearizeSkip-th call (ISAM2::update)
  ISAM2Result update(const NonlinearFactorGraph &newFactors = NonlinearFactorGraph(), const Values &newTheta = Values());
  Values calculateEstimate() const;
  template <class T> T calculateEstimate(Key k) const { return calculateEstimate().at<T>(k); }
  Matrix marginalCovariance(Key k);
  const NonlinearFactorGraph &getFactorsUnsafe() const { return all_; }
 private:
  ISAM2Params p_;
  std::shared_ptr<FgoBridge> b_;
  NonlinearFactorGraph all_;             // every factor handed to update(), in order
  int count_;
  size_t loaded_;
};

class Marginals {
 public:
  enum Factorization { CHOLESKY, QR };
  Marginals(const NonlinearFactorGraph &g, const Values &v, Factorization = CHOLESKY) : g_(g), v_(v) {}
  Matrix marginalCovariance(Key k) const;     // dim x dim block of (J' Omega J)^-1 at v
 private:
  NonlinearFactorGraph g_;
  Values v_;
  mutable std::shared_ptr<FgoBridge> b_;      // built on first use
};

// dataset.h: g2o text for the Pose3 part of a graph (BetweenFactor<Pose3> -> EDGE_SE3:QUAT, gtsam_graph.cpp:1941-1945)
void writeG2o(const NonlinearFactorGraph &g, const Values &v, const std::string &filename);

}  // namespace gtsam
