// Device-side GTSAM-semantics Pose3 arithmetic (gfx950, f64): tangent xi = [omega; v] (rotation first),
// retract(x, xi) = x * Expmap(xi), PriorFactor<Pose3> and BetweenFactor<Pose3> residuals with their Jacobians.
// Semantics as the reference uses them: gtsam/gtsam_graph.cpp:338-341 (prior, Diagonal::Sigmas),
// :640,689-692 (BetweenFactor + Gaussian::Information), :675-676 (AdjointMap); chart = full exponential map
// (SURVEY.md Appendix A.2 — GTSAM's chart is a build flag, the choice is documented in DESIGN.md).
#pragma once
#include "se3_device.hpp"

namespace fgo {
namespace dev {

__device__ __forceinline__ M3 skew(V3 w) { return {{0, -w.z, w.y, w.z, 0, -w.x, -w.y, w.x, 0}}; }
__device__ __forceinline__ M3 mscale(const M3 &A, double s) {
  M3 C;
#pragma unroll
  for (int k = 0; k < 9; ++k) C.m[k] = A.m[k] * s;
  return C;
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__device__ __forceinline__ Pose pose_mul(const Pose &a, const Pose &b) {
  Pose c;
  c.t = a.t + mv(qmat(a.q), b.t);
  c.q = qmul(a.q, b.q);
  return c;
}
__device__ __forceinline__ Pose pose_inv(const Pose &a) {
  Pose c;
  c.q = qconj(a.q);
  const V3 r = mv(qmat(c.q), a.t);
  c.t = {-r.x, -r.y, -r.z};
  return c;
}
// Rot3::Expmap as a unit quaternion
__device__ __forceinline__ Q4 so3_exp(V3 w) {
  const double th2 = dot3(w, w), th = sqrt(th2);
  const double s = th < 1e-10 ? 0.5 - th2 / 48.0 : sin(0.5 * th) / th;
  return {s * w.x, s * w.y, s * w.z, cos(0.5 * th)};
}
// Rot3::Logmap (angle in [0, pi])
__device__ __forceinline__ V3 so3_log(Q4 q) {
  if (q.w < 0) q = {-q.x, -q.y, -q.z, -q.w};
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  const double k = n < 1e-10 ? 2.0 + n * n / 3.0 : 2.0 * atan2(n, q.w) / n;
  return {k * q.x, k * q.y, k * q.z};
}
// inverse right Jacobian of SO(3)
__device__ __forceinline__ M3 so3_dlog(V3 w) {
  const double th2 = dot3(w, w), th = sqrt(th2);
  const M3 W = skew(w), W2 = mm(W, W);
  const double c = th < 1e-5 ? 1.0 / 12.0 + th2 / 720.0 : 1.0 / th2 - (1.0 + cos(th)) / (2.0 * th * sin(th));
  M3 J;
#pragma unroll
  for (int k = 0; k < 9; ++k) J.m[k] = 0.5 * W.m[k] + c * W2.m[k];
  J.m[0] += 1; J.m[4] += 1; J.m[8] += 1;
  return J;
}
// Pose3::Expmap
__device__ __forceinline__ Pose se3_exp(const double xi[6]) {
  const V3 w = {xi[0], xi[1], xi[2]}, v = {xi[3], xi[4], xi[5]};
  Pose T;
  T.q = so3_exp(w);
  const double th2 = dot3(w, w);
  if (th2 < 1e-20) { T.t = v; return T; }
  const V3 c = cross(w, v);
  const V3 Rc = mv(qmat(T.q), c);
  const double wv = dot3(w, v);
  T.t = {(c.x - Rc.x + w.x * wv) / th2, (c.y - Rc.y + w.y * wv) / th2, (c.z - Rc.z + w.z * wv) / th2};
  return T;
}
// Pose3::Logmap -> xi[6] = [omega; u]
__device__ __forceinline__ void se3_log(const Pose &T, double xi[6]) {
  const V3 w = so3_log(T.q);
  const double th = sqrt(dot3(w, w));
  xi[0] = w.x; xi[1] = w.y; xi[2] = w.z;
  if (th < 1e-10) { xi[3] = T.t.x; xi[4] = T.t.y; xi[5] = T.t.z; return; }
  const V3 a = {w.x / th, w.y / th, w.z / th};
  const V3 Wt = cross(a, T.t), WWt = cross(a, Wt);
  const double k = 1.0 - th / (2.0 * tan(0.5 * th));
  xi[3] = T.t.x - 0.5 * th * Wt.x + k * WWt.x;
  xi[4] = T.t.y - 0.5 * th * Wt.y + k * WWt.y;
  xi[5] = T.t.z - 0.5 * th * Wt.z + k * WWt.z;
}
// dense 6x6, row-major
struct M6 { double m[36]; };
__device__ __forceinline__ void set33(M6 &A, int r0, int c0, const M3 &B, double s = 1.0) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) A.m[(r0 + r) * 6 + c0 + c] = s * B.m[r * 3 + c];
}
// Pose3::LogmapDerivative at xi: [[Jw, 0], [-Jw Q Jw, Jw]]
__device__ __forceinline__ M6 se3_dlog(const double xi[6]) {
  const V3 w = {xi[0], xi[1], xi[2]}, v = {xi[3], xi[4], xi[5]};
  const M3 Jw = so3_dlog(w), V = skew(v), W = skew(w);
  const double ph2 = dot3(w, w), ph = sqrt(ph2);
  const M3 WV = mm(W, V), VW = mm(V, W), WVW = mm(WV, W), WW = mm(W, W), WWV = mm(WW, V), VWW = mm(VW, W);
  const M3 WVWW = mm(WVW, W), WWVW = mm(W, WVW);
  double c1, c2, c3;
  if (ph > 1e-5) {
    const double s = sin(ph), c = cos(ph), ph3 = ph2 * ph, ph4 = ph2 * ph2, ph5 = ph4 * ph;
    c1 = (ph - s) / ph3;
    c2 = (1 - ph2 / 2 - c) / ph4;
    c3 = -0.5 * ((1 - ph2 / 2 - c) / ph4 - 3 * (ph - s - ph3 / 6.) / ph5);
  } else {
    c1 = 1. / 6.; c2 = 1. / 24.; c3 = -0.5 * (1. / 24. + 3. / 120.);
  }
  M3 Q;
#pragma unroll
  for (int k = 0; k < 9; ++k)
    Q.m[k] = -0.5 * V.m[k] + c1 * (WV.m[k] + VW.m[k] - WVW.m[k]) + c2 * (WWV.m[k] + VWW.m[k] - 3 * WVW.m[k]) + c3 * (WVWW.m[k] + WWVW.m[k]);
  const M3 Q2 = mm(mm(Jw, Q), Jw);
  M6 J;
#pragma unroll
  for (int k = 0; k < 36; ++k) J.m[k] = 0;
  set33(J, 0, 0, Jw); set33(J, 3, 3, Jw); set33(J, 3, 0, Q2, -1.0);
  return J;
}
// Pose3::AdjointMap: [[R, 0], [[t]x R, R]]
__device__ __forceinline__ M6 se3_adjoint(const Pose &T) {
  const M3 R = qmat(T.q), SR = mm(skew(T.t), R);
  M6 A;
#pragma unroll
  for (int k = 0; k < 36; ++k) A.m[k] = 0;
  set33(A, 0, 0, R); set33(A, 3, 3, R); set33(A, 3, 0, SR);
  return A;
}
__device__ __forceinline__ M6 m6mul(const M6 &A, const M6 &B, double s = 1.0) {
  M6 C;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double a = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) a += A.m[r * 6 + k] * B.m[k * 6 + c];
      C.m[r * 6 + c] = s * a;
    }
  return C;
}
__device__ __forceinline__ M6 m6tmul(const M6 &A, const M6 &B) {   // A^T B
  M6 C;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double a = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) a += A.m[k * 6 + r] * B.m[k * 6 + c];
      C.m[r * 6 + c] = a;
    }
  return C;
}
__device__ __forceinline__ Pose retract_pose3(const Pose &X, const double xi[6]) {
  Pose Y = pose_mul(X, se3_exp(xi));
  const double n = sqrt(Y.q.x * Y.q.x + Y.q.y * Y.q.y + Y.q.z * Y.q.z + Y.q.w * Y.q.w);
  Y.q = {Y.q.x / n, Y.q.y / n, Y.q.z / n, Y.q.w / n};
  return Y;
}
// BetweenFactor<Pose3>: e = Logmap(Z^-1 xi^-1 xj), Jj = dLog(e), Ji = -dLog(e) Ad((xi^-1 xj)^-1).  Zinv = Z^-1.
template <bool WITH_JAC>
__device__ __forceinline__ void between_pose3(const Pose &Xi, const Pose &Xj, const Pose &Zinv, double e[6], M6 &Ji, M6 &Jj) {
  const Pose h = pose_mul(pose_inv(Xi), Xj);
  se3_log(pose_mul(Zinv, h), e);
  if (WITH_JAC) {
    Jj = se3_dlog(e);
    Ji = m6mul(Jj, se3_adjoint(pose_inv(h)), -1.0);
  }
}
// PriorFactor<Pose3>: e = Logmap(prior^-1 x), J = dLog(e).  Pinv = prior^-1.
template <bool WITH_JAC>
__device__ __forceinline__ void prior_pose3(const Pose &X, const Pose &Pinv, double e[6], M6 &J) {
  se3_log(pose_mul(Pinv, X), e);
  if (WITH_JAC) J = se3_dlog(e);
}

}  // namespace dev
}  // namespace fgo
