// Stand-in for the slice of GTSAM 4.0 that the reference's CGraphGT / imu_interface / VIO drivers touch
// (gtsam/gtsam_graph.cpp, gtsam/imu_base.cpp, gtsam/test_vro_imu_graph.cpp:94-373), implemented ON TOP OF the
// libfgo C-ABI: Values and NonlinearFactorGraph are handles onto one fgo_ctx, so "insert a value" is fgo_add_*,
// "add a factor" is fgo_add_*, LevenbergMarquardtOptimizer::optimize is fgo_optimize_gtsam on the MI355X and
// Values::at<T> reads the device estimate back.  GTSAM is not installed in this image; nothing here evaluates a
// factor on the CPU (the product has no CPU fallback) -- only the pose algebra a graph BUILDER needs lives here.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>
#include <Eigen/Core>
#include <Eigen/Geometry>
extern "C" {
#include "fgo.h"
}

namespace gtsam {

typedef Eigen::Matrix<double, 3, 1> Vector3;
typedef Eigen::Matrix<double, 4, 1> Vector4;
typedef Eigen::Matrix<double, 6, 1> Vector6;
typedef Eigen::Matrix<double, 3, 3> Matrix3;
typedef Eigen::Matrix<double, 6, 6> Matrix6;
typedef Eigen::Matrix<double, 15, 15> Matrix15;
typedef Eigen::Quaterniond Quaternion;
typedef Vector3 Point3;
typedef uint64_t Key;

namespace symbol_shorthand {
inline Key mk(char c, uint64_t j) { return ((uint64_t)(unsigned char)c << 56) | j; }
inline Key X(uint64_t j) { return mk('x', j); }
inline Key V(uint64_t j) { return mk('v', j); }
inline Key B(uint64_t j) { return mk('b', j); }
inline Key L(uint64_t j) { return mk('l', j); }
}  // namespace symbol_shorthand

inline Matrix3 skew3(const Vector3 &w) {
  Matrix3 S;
  S(0, 1) = -w(2); S(0, 2) = w(1); S(1, 0) = w(2); S(1, 2) = -w(0); S(2, 0) = -w(1); S(2, 1) = w(0);
  return S;
}

class Rot3 {
 public:
  Rot3() { R_.setIdentity(); }
  explicit Rot3(const Matrix3 &R) : R_(R) {}
  static Rot3 Rx(double t) { Matrix3 R = Matrix3::Identity(); R(1, 1) = std::cos(t); R(1, 2) = -std::sin(t); R(2, 1) = std::sin(t); R(2, 2) = std::cos(t); return Rot3(R); }
  static Rot3 Ry(double t) { Matrix3 R = Matrix3::Identity(); R(0, 0) = std::cos(t); R(0, 2) = std::sin(t); R(2, 0) = -std::sin(t); R(2, 2) = std::cos(t); return Rot3(R); }
  static Rot3 Rz(double t) { Matrix3 R = Matrix3::Identity(); R(0, 0) = std::cos(t); R(0, 1) = -std::sin(t); R(1, 0) = std::sin(t); R(1, 1) = std::cos(t); return Rot3(R); }
  // GTSAM: RzRyRx(x, y, z) = Rz(z) * Ry(y) * Rx(x)
  static Rot3 RzRyRx(double x, double y, double z) { return Rz(z) * Ry(y) * Rx(x); }
  static Rot3 Expmap(const Vector3 &w) {
    const double th2 = w(0) * w(0) + w(1) * w(1) + w(2) * w(2), th = std::sqrt(th2);
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
    Vector3 o;
    const double k = n < 1e-12 ? 2.0 : 2.0 * std::atan2(n, w) / n;
    o(0) = k * x; o(1) = k * y; o(2) = k * z;
    return o;
  }
  Rot3 operator*(const Rot3 &o) const { return Rot3(R_ * o.R_); }
  Vector3 operator*(const Vector3 &p) const { return R_ * p; }
  Rot3 inverse() const { return Rot3(R_.transpose()); }
  const Matrix3 &matrix() const { return R_; }
  Quaternion toQuaternion() const { return Quaternion(R_); }
 private:
  Matrix3 R_;
};

class Pose3 {
 public:
  Pose3() {}
  Pose3(const Rot3 &R, const Point3 &t) : R_(R), t_(t) {}
  explicit Pose3(const Eigen::Matrix4d &M) {
    Matrix3 R;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R(r, c) = M(r, c); t_(r) = M(r, 3); }
    R_ = Rot3(R);
  }
  static Pose3 Create(const Rot3 &R, const Point3 &t) { return Pose3(R, t); }
  // full exponential map, tangent [omega; v] (the chart libfgo retracts with: DESIGN.md, GTSAM_POSE3_EXPMAP)
  static Pose3 Expmap(const Vector6 &xi) {
    Vector3 w, v;
    for (int k = 0; k < 3; ++k) { w(k) = xi(k); v(k) = xi(3 + k); }
    const double th2 = w(0) * w(0) + w(1) * w(1) + w(2) * w(2), th = std::sqrt(th2);
    const Matrix3 W = skew3(w), W2 = W * W;
    double b, c;
    if (th < 1e-8) { b = 0.5 - th2 / 24.0; c = 1.0 / 6.0 - th2 / 120.0; } else { b = (1.0 - std::cos(th)) / th2; c = (th - std::sin(th)) / (th2 * th); }
    const Matrix3 Vm = Matrix3::Identity() + W * b + W2 * c;
    return Pose3(Rot3::Expmap(w), Vm * v);
  }
  static Vector6 Logmap(const Pose3 &p) {
    const Vector3 w = Rot3::Logmap(p.R_);
    const double th2 = w(0) * w(0) + w(1) * w(1) + w(2) * w(2), th = std::sqrt(th2);
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
  Pose3 inverse() const { const Rot3 Ri = R_.inverse(); return Pose3(Ri, (Ri * t_) * -1.0); }
  Pose3 between(const Pose3 &o) const { return inverse() * o; }
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
    std::printf("%s t = (%g %g %g)\n", s.c_str(), t_(0), t_(1), t_(2));
  }
 private:
  Rot3 R_;
  Point3 t_;
};

class NavState {
 public:
  NavState() {}
  NavState(const Pose3 &p, const Vector3 &v) : p_(p), v_(v) {}
  const Pose3 &pose() const { return p_; }
  const Vector3 &v() const { return v_; }
  const Vector3 &velocity() const { return v_; }
 private:
  Pose3 p_;
  Vector3 v_;
};

namespace imuBias {
class ConstantBias {
 public:
  ConstantBias() {}
  ConstantBias(const Vector3 &acc, const Vector3 &gyro) : a_(acc), g_(gyro) {}
  const Vector3 &accelerometer() const { return a_; }
  const Vector3 &gyroscope() const { return g_; }
  Vector6 vector() const { Vector6 o; for (int k = 0; k < 3; ++k) { o(k) = a_(k); o(3 + k) = g_(k); } return o; }
 private:
  Vector3 a_, g_;
};
}  // namespace imuBias

// unit normal + distance, (a, b, c, d)
class OrientedPlane3 {
 public:
  OrientedPlane3() { v_(2) = 1.0; }
  OrientedPlane3(double a, double b, double c, double d) { v_(0) = a; v_(1) = b; v_(2) = c; v_(3) = d; }
  explicit OrientedPlane3(const Vector4 &v) : v_(v) {}
  const Vector4 &planeCoefficients() const { return v_; }
 private:
  Vector4 v_;
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
  static shared_ptr Information(const Matrix6 &M) { auto m = std::make_shared<Base>(); m->dim = 6; m->info = M; return m; }
  static shared_ptr Covariance(const Matrix3 &S) {
    auto m = std::make_shared<Base>(); m->dim = 3; m->has_cov = true;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) m->cov(r, c) = S(r, c);
    return m;
  }
};
struct Diagonal : Base {
  typedef std::shared_ptr<Base> shared_ptr;
  static shared_ptr Sigmas(const Vector6 &s) {
    auto m = std::make_shared<Base>(); m->dim = 6; m->sigma = s(0);
    for (int k = 0; k < 6; ++k) m->info(k, k) = 1.0 / (s(k) * s(k));
    return m;
  }
};
struct Isotropic : Base {
  static std::shared_ptr<Base> Sigma(int dim, double s) {
    auto m = std::make_shared<Base>(); m->dim = dim; m->sigma = s;
    for (int k = 0; k < dim && k < 6; ++k) m->info(k, k) = 1.0 / (s * s);
    return m;
  }
};
}  // namespace noiseModel
typedef std::shared_ptr<noiseModel::Base> SharedNoiseModel;

// ---- IMU preintegration (host side of the C-ABI: fgo_preint_*, csrc/imu_preint.cpp)
class PreintegratedCombinedMeasurements {
 public:
  struct Params {
    fgo_imu_params p;
    double n_gravity[3];
    static std::shared_ptr<Params> MakeSharedD(double g) {      // Z-down navigation frame: n_gravity = (0, 0, +g)
      auto o = std::make_shared<Params>();
      fgo_imu_params_vn100(&o->p);
      o->n_gravity[0] = 0; o->n_gravity[1] = 0; o->n_gravity[2] = g;
      return o;
    }
  };
  PreintegratedCombinedMeasurements() { std::shared_ptr<Params> p = Params::MakeSharedD(9.71); params_ = p; reset(imuBias::ConstantBias()); }
  PreintegratedCombinedMeasurements(const std::shared_ptr<Params> &p, const imuBias::ConstantBias &b) : params_(p) { reset(b); }
  void integrateMeasurement(const Vector3 &acc, const Vector3 &gyro, double dt) {
    const double a[3] = {acc(0), acc(1), acc(2)}, g[3] = {gyro(0), gyro(1), gyro(2)};
    fgo_preint_integrate(&m_, &params_->p, a, g, dt);
  }
  void resetIntegrationAndSetBias(const imuBias::ConstantBias &b) { reset(b); }
  NavState predict(const NavState &s, const imuBias::ConstantBias &b) const {
    const Quaternion q = s.pose().rotation().toQuaternion();
    const double pi[7] = {s.pose().x(), s.pose().y(), s.pose().z(), q.x(), q.y(), q.z(), q.w()};
    const double vi[3] = {s.v()(0), s.v()(1), s.v()(2)};
    const Vector6 bv = b.vector();
    double pj[7], vj[3];
    fgo_preint_predict(&m_, params_->n_gravity, pi, vi, bv.data(), pj, vj);
    Point3 t; t(0) = pj[0]; t(1) = pj[1]; t(2) = pj[2];
    Vector3 v; v(0) = vj[0]; v(1) = vj[1]; v(2) = vj[2];
    return NavState(Pose3(Rot3(Quaternion(pj[6], pj[3], pj[4], pj[5]).toRotationMatrix()), t), v);
  }
  Matrix15 preintMeasCov() const { Matrix15 M; for (int k = 0; k < 225; ++k) M.data()[k] = m_.cov[k]; return M; }
  double deltaTij() const { return m_.dt; }
  const fgo_preint &raw() const { return m_; }
  const Params &params() const { return *params_; }
 private:
  void reset(const imuBias::ConstantBias &b) { const Vector6 v = b.vector(); fgo_preint_reset(&m_, v.data()); }
  std::shared_ptr<Params> params_;
  fgo_preint m_;
};
typedef PreintegratedCombinedMeasurements PreintegrationType;

// ---- factors: plain descriptors, translated into C-ABI calls by NonlinearFactorGraph
struct FactorDesc {
  enum Kind { PRIOR_POSE, PRIOR_VEC3, PRIOR_BIAS, BETWEEN, IMU, PLANE } kind;
  Key k[6];
  int nk;
  double t[3], q[4];          // pose payload
  double v6[6];               // vector / bias / plane payload
  double info21[21];
  double cov6[6];
  double sigma;
  fgo_preint pim;
};
inline void pose_payload(const Pose3 &p, double t[3], double q[4]) {
  const Quaternion qq = p.rotation().toQuaternion();
  t[0] = p.x(); t[1] = p.y(); t[2] = p.z();
  q[0] = qq.x(); q[1] = qq.y(); q[2] = qq.z(); q[3] = qq.w();
}
inline void info_ut21(const Matrix6 &M, double out[21]) { int k = 0; for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) out[k++] = M(r, c); }

template <class T> struct PriorFactor;
template <> struct PriorFactor<Pose3> {
  FactorDesc d;
  PriorFactor(Key k, const Pose3 &mean, const SharedNoiseModel &n) { d.kind = FactorDesc::PRIOR_POSE; d.k[0] = k; d.nk = 1; pose_payload(mean, d.t, d.q); info_ut21(n->info, d.info21); }
};
template <> struct PriorFactor<Vector3> {
  FactorDesc d;
  PriorFactor(Key k, const Vector3 &mean, const SharedNoiseModel &n) { d.kind = FactorDesc::PRIOR_VEC3; d.k[0] = k; d.nk = 1; for (int i = 0; i < 3; ++i) d.v6[i] = mean(i); d.sigma = n->sigma; }
};
template <> struct PriorFactor<imuBias::ConstantBias> {
  FactorDesc d;
  PriorFactor(Key k, const imuBias::ConstantBias &mean, const SharedNoiseModel &n) { d.kind = FactorDesc::PRIOR_BIAS; d.k[0] = k; d.nk = 1; const Vector6 v = mean.vector(); for (int i = 0; i < 6; ++i) d.v6[i] = v(i); d.sigma = n->sigma; }
};
template <class T> struct BetweenFactor;
template <> struct BetweenFactor<Pose3> {
  FactorDesc d;
  BetweenFactor(Key i, Key j, const Pose3 &z, const SharedNoiseModel &n) { d.kind = FactorDesc::BETWEEN; d.k[0] = i; d.k[1] = j; d.nk = 2; pose_payload(z, d.t, d.q); info_ut21(n->info, d.info21); }
};
struct CombinedImuFactor {
  FactorDesc d;
  CombinedImuFactor(Key xi, Key vi, Key xj, Key vj, Key bi, Key bj, const PreintegratedCombinedMeasurements &pim) {
    d.kind = FactorDesc::IMU; d.k[0] = xi; d.k[1] = vi; d.k[2] = xj; d.k[3] = vj; d.k[4] = bi; d.k[5] = bj; d.nk = 6; d.pim = pim.raw();
  }
};
struct OrientedPlane3Factor {
  FactorDesc d;
  OrientedPlane3Factor(const Vector4 &z, const SharedNoiseModel &n, Key pose, Key landmark) {
    d.kind = FactorDesc::PLANE; d.k[0] = pose; d.k[1] = landmark; d.nk = 2;
    for (int i = 0; i < 4; ++i) d.v6[i] = z(i);
    int k = 0;
    for (int r = 0; r < 3; ++r) for (int c = r; c < 3; ++c) d.cov6[k++] = n->cov(r, c);
  }
};

// ---- one fgo context shared by the graph and the values of a CGraphGT
struct Backend {
  fgo_ctx *ctx;
  Backend() : ctx(fgo_create(nullptr)) {}
  ~Backend() { if (ctx) fgo_destroy(ctx); }
  Backend(const Backend &) = delete;
  Backend &operator=(const Backend &) = delete;
};

class Values {
 public:
  Values() {}                                             // staging object (ISAM2 "new nodes"): no backend
  explicit Values(const std::shared_ptr<Backend> &b) : b_(b) {}
  bool exists(Key k) const { return b_ && b_->ctx && fgo_has_pose(b_->ctx, (int64_t)k) == 1; }
  void insert(Key k, const Pose3 &p) { ++n_; if (!live()) return; double t[3], q[4]; pose_payload(p, t, q); fgo_add_pose(b_->ctx, (int64_t)k, t, q, 0); }
  void insert(Key k, const Vector3 &v) { ++n_; if (!live()) return; fgo_add_vec3(b_->ctx, (int64_t)k, v.data()); }
  void insert(Key k, const imuBias::ConstantBias &b) { ++n_; if (!live()) return; const Vector6 v = b.vector(); fgo_add_bias(b_->ctx, (int64_t)k, v.data()); }
  void insert(Key k, const OrientedPlane3 &p) { ++n_; if (!live()) return; fgo_add_plane(b_->ctx, (int64_t)k, p.planeCoefficients().data()); }
  void update(Key k, const Pose3 &p) { if (!live()) return; double t[3], q[4]; pose_payload(p, t, q); fgo_set_pose(b_->ctx, (int64_t)k, t, q); }
  template <class T> T at(Key k) const;
  size_t size() const { return n_; }
  void clear() { if (!b_) n_ = 0; }
 private:
  bool live() const { return b_ && b_->ctx; }
  bool read(Key k, double o[7]) const { return live() && fgo_get_pose(b_->ctx, (int64_t)k, o) == FGO_OK; }
  std::shared_ptr<Backend> b_;
  size_t n_ = 0;
};
template <> inline Pose3 Values::at<Pose3>(Key k) const {
  double o[7] = {0, 0, 0, 0, 0, 0, 1};
  read(k, o);
  Point3 t; t(0) = o[0]; t(1) = o[1]; t(2) = o[2];
  return Pose3(Rot3(Quaternion(o[6], o[3], o[4], o[5]).toRotationMatrix()), t);
}
template <> inline Vector3 Values::at<Vector3>(Key k) const { double o[7] = {0}; read(k, o); Vector3 v; v(0) = o[0]; v(1) = o[1]; v(2) = o[2]; return v; }
template <> inline imuBias::ConstantBias Values::at<imuBias::ConstantBias>(Key k) const {
  double o[7] = {0}; read(k, o);
  Vector3 a, g; for (int i = 0; i < 3; ++i) { a(i) = o[i]; g(i) = o[3 + i]; }
  return imuBias::ConstantBias(a, g);
}
template <> inline OrientedPlane3 Values::at<OrientedPlane3>(Key k) const { double o[7] = {0, 0, 1, 0}; read(k, o); return OrientedPlane3(o[0], o[1], o[2], o[3]); }

class NonlinearFactorGraph {
 public:
  NonlinearFactorGraph() {}                               // staging object (ISAM2 "new factors"): no backend
  explicit NonlinearFactorGraph(const std::shared_ptr<Backend> &b) : b_(b) {}
  template <class F> void add(const F &f) { ++n_; if (b_) pending_.push_back(f.d); }
  size_t size() const { return n_; }
  void resize(size_t n) { if (!b_) n_ = n; }
  // Factors may be added before their variables exist (the drivers add the IMU factor, then insert V/B): they are
  // handed to the C-ABI as soon as all their variables do.  Returns the number still waiting.
  size_t flush() {
    if (!b_ || !b_->ctx) return pending_.size();
    std::vector<FactorDesc> keep;
    for (const FactorDesc &d : pending_) {
      bool ready = true;
      for (int i = 0; i < d.nk; ++i) ready = ready && fgo_has_pose(b_->ctx, (int64_t)d.k[i]) == 1;
      if (!ready) { keep.push_back(d); continue; }
      int rc = FGO_OK;
      switch (d.kind) {
        case FactorDesc::PRIOR_POSE: rc = fgo_add_prior_pose(b_->ctx, (int64_t)d.k[0], d.t, d.q, d.info21); break;
        case FactorDesc::PRIOR_VEC3: rc = fgo_add_prior_vec3(b_->ctx, (int64_t)d.k[0], d.v6, d.sigma); break;
        case FactorDesc::PRIOR_BIAS: rc = fgo_add_prior_bias(b_->ctx, (int64_t)d.k[0], d.v6, d.sigma); break;
        case FactorDesc::BETWEEN: rc = fgo_add_edge_se3(b_->ctx, (int64_t)d.k[0], (int64_t)d.k[1], d.t, d.q, d.info21, FGO_TANGENT_GTSAM); break;
        case FactorDesc::PLANE: rc = fgo_add_plane_factor(b_->ctx, (int64_t)d.k[0], (int64_t)d.k[1], d.v6, d.cov6); break;
        case FactorDesc::IMU: { int64_t ids[6]; for (int i = 0; i < 6; ++i) ids[i] = (int64_t)d.k[i]; rc = fgo_add_imu_combined(b_->ctx, ids, &d.pim); break; }
      }
      if (rc != FGO_OK) std::fprintf(stderr, "gtsam shim: factor rejected by libfgo: %s\n", fgo_last_error(b_->ctx));
    }
    pending_.swap(keep);
    return pending_.size();
  }
  double error(const Values &) { flush(); return (b_ && b_->ctx) ? fgo_error(b_->ctx) : 0.0; }
  const std::shared_ptr<Backend> &backend() const { return b_; }
 private:
  std::shared_ptr<Backend> b_;
  std::vector<FactorDesc> pending_;
  size_t n_ = 0;
};

class LevenbergMarquardtOptimizer {
 public:
  LevenbergMarquardtOptimizer(NonlinearFactorGraph &g, const Values &v) : g_(g), v_(v), iterations_(0) {}
  // GTSAM 4.0 default parameters (maxIterations 100, lambdaInitial 1e-5, ...): fgo_optimize_gtsam
  Values optimize() {
    const size_t waiting = g_.flush();
    if (waiting) std::fprintf(stderr, "gtsam shim: %zu factors reference variables that were never inserted\n", waiting);
    fgo_stats st;
    const int rc = fgo_optimize_gtsam(g_.backend()->ctx, 100, &st);
    if (rc < 0) std::fprintf(stderr, "gtsam shim: fgo_optimize_gtsam: %s\n", fgo_last_error(g_.backend()->ctx));
    iterations_ = rc > 0 ? rc : 0;
    return v_;
  }
  int iterations() const { return iterations_; }
 private:
  NonlinearFactorGraph &g_;
  Values v_;
  int iterations_;
};

// ISAM2Params / ISAM2 as CGraphGT uses them (gtsam/gtsam_graph.cpp:93-99, 1768-1776).  Every factor / value the
// drivers hand to the staging objects is also added to the full graph / values, whose context keeps ISAM2's
// linearisation point and delta (fgo_isam2_update): update() therefore only needs to know that context (attach()).
struct ISAM2Params {
  double relinearizeThreshold = 0.1;     // GTSAM 4.0 defaults
  int relinearizeSkip = 10;
};
class ISAM2 {
 public:
  explicit ISAM2(const ISAM2Params &p = ISAM2Params()) : p_(p), g_(nullptr), v_(nullptr), count_(0) {}
  void attach(NonlinearFactorGraph &full_graph, Values &full_values) { g_ = &full_graph; v_ = &full_values; }
  // update(newFactors, newTheta): relinearisation is considered every relinearizeSkip-th call (ISAM2::update)
  void update(const NonlinearFactorGraph &, const Values &) { step(); }
  void update() { step(); }
  Values calculateEstimate() const { return v_ ? *v_ : Values(); }
  int lastRelinearized() const { return relinearized_; }
 private:
  void step() {
    if (!g_ || !g_->backend() || !g_->backend()->ctx) return;
    const size_t waiting = g_->flush();
    if (waiting) std::fprintf(stderr, "gtsam shim: %zu factors reference variables that were never inserted\n", waiting);
    const bool relin = p_.relinearizeSkip <= 1 || count_ % p_.relinearizeSkip == 0;
    ++count_;
    fgo_stats st;
    const int rc = fgo_isam2_update(g_->backend()->ctx, relin ? p_.relinearizeThreshold : 1e300, &st);
    if (rc < 0) std::fprintf(stderr, "gtsam shim: fgo_isam2_update: %s\n", fgo_last_error(g_->backend()->ctx));
    else relinearized_ = (int)st.reserved[1];
  }
  ISAM2Params p_;
  NonlinearFactorGraph *g_;
  Values *v_;
  int count_, relinearized_ = 0;
};

class Marginals {
 public:
  Marginals(NonlinearFactorGraph &g, const Values &) : g_(g) { g_.flush(); }
  Matrix6 marginalCovariance(Key k) const {
    Matrix6 M;
    if (fgo_marginal_cov(g_.backend()->ctx, (int64_t)k, M.data()) != FGO_OK)
      std::fprintf(stderr, "gtsam shim: fgo_marginal_cov: %s\n", fgo_last_error(g_.backend()->ctx));
    return M;
  }
 private:
  NonlinearFactorGraph &g_;
};

}  // namespace gtsam
