// Device-side arithmetic of the remaining GTSAM-semantics factors on the path (gfx950, f64):
//   OrientedPlane3Factor(z, Gaussian::Covariance(S), X, L)                       gtsam/gtsam_graph.cpp:1265
//   GenericProjectionFactor<Pose3, Point3, Cal3DS2>(z, sigma 1 px, X, Q, K, body_P_sensor)  gtsam/gtsam_graph.cpp:405-409
// and the retraction of the non-pose variables (OrientedPlane3: exponential map on the sphere + offset; vectors: +).
// Plane semantics are the GTSAM 4.0 ones pinned by the golden vectors the reference carries
// (gtsam/test/testOrientedPlane3*.cpp; tests/test_plane_golden.py): r = [-local_{n'}(n_z); d' - d_z] with the
// Jacobians of transform().  Residuals / Jacobians are padded to 6 rows / 6 columns (M6) so every variable is a 6-block.
#pragma once
#include "device_plan.hpp"
#include "pose3_device.hpp"

namespace fgo {
namespace dev {

enum VarKind { VK_POSE = 0, VK_PLANE = 1, VK_POINT = 2, VK_VEC3 = 3, VK_BIAS = 4, VK_PHANTOM = 5 };
enum FactorKind { FK_G2O = 0, FK_BETWEEN = 1, FK_PLANE = 2, FK_REPROJ = 3 };

// VK_PHANTOM: a reserved slot of the incremental mode: no degrees of freedom (identity block, zero gradient) until a real
// variable claims it
__device__ __forceinline__ int var_dim(int vk) { return (vk == VK_POSE || vk == VK_BIAS) ? 6 : (vk == 5 ? 0 : 3); }

struct Basis { V3 b1, b2; };
// Unit3::basis(): b1 = normalise(n x axis of smallest |n_i|), b2 = n x b1 (ties: x, then y, then z)
__device__ __forceinline__ Basis unit3_basis(V3 n) {
  const double mx = fabs(n.x), my = fabs(n.y), mz = fabs(n.z);
  V3 ax = {0, 0, 1};
  if (mx <= my && mx <= mz) ax = {1, 0, 0};
  else if (my <= mx && my <= mz) ax = {0, 1, 0};
  V3 b1 = cross(n, ax);
  const double nb = sqrt(dot3(b1, b1));
  b1 = {b1.x / nb, b1.y / nb, b1.z / nb};
  return {b1, cross(n, b1)};
}
__device__ __forceinline__ V3 unit3_retract(V3 n, double v0, double v1) {
  const Basis B = unit3_basis(n);
  const V3 xi = {B.b1.x * v0 + B.b2.x * v1, B.b1.y * v0 + B.b2.y * v1, B.b1.z * v0 + B.b2.z * v1};
  const double th = sqrt(dot3(xi, xi));
  const double s = th < 1e-300 ? 1.0 : sin(th) / th, c = cos(th);
  V3 o = {c * n.x + s * xi.x, c * n.y + s * xi.y, c * n.z + s * xi.z};
  const double nn = sqrt(dot3(o, o));
  return {o.x / nn, o.y / nn, o.z / nn};
}
__device__ __forceinline__ void unit3_local(V3 n, V3 y, double v[2]) {
  const double x = dot3(n, y);
  if (x > 1.0 - 1e-16) { v[0] = v[1] = 0; return; }
  if (x < -1.0 + 1e-16) { v[0] = 3.14159265358979323846; v[1] = 0; return; }
  const double th = acos(x), k = th / sin(th);
  const Basis B = unit3_basis(n);
  const V3 h = {k * (y.x - x * n.x), k * (y.y - x * n.y), k * (y.z - x * n.z)};
  v[0] = dot3(B.b1, h); v[1] = dot3(B.b2, h);
}

// OrientedPlane3Factor.  X pose, plane (n, d) in world, z measured plane (n_z, d_z) in the pose frame.
template <bool WITH_JAC>
__device__ __forceinline__ void plane_factor(const Pose &X, V3 n, double d, V3 nz, double dz, double e[6], M6 &Jx, M6 &Jp) {
  const M3 R = qmat(X.q);
  const V3 np = mtv(R, n);                       // n' = R^T n
  const double dp = dot3(n, X.t) + d;
  double l[2];
  unit3_local(np, nz, l);
  e[0] = -l[0]; e[1] = -l[1]; e[2] = dp - dz; e[3] = e[4] = e[5] = 0;
  if (WITH_JAC) {
#pragma unroll
    for (int k = 0; k < 36; ++k) { Jx.m[k] = 0; Jp.m[k] = 0; }
    const Basis Bp = unit3_basis(np), B = unit3_basis(n);
    // d n'_local / d omega = B'^T [n']x : row a = (b_a x n')^T ... using b^T [n]x = (n x b)^T * (-1)
    const V3 r1 = cross(Bp.b1, np), r2 = cross(Bp.b2, np);       // b^T [n']x = (b x n')^T
    Jx.m[0] = r1.x; Jx.m[1] = r1.y; Jx.m[2] = r1.z;
    Jx.m[6] = r2.x; Jx.m[7] = r2.y; Jx.m[8] = r2.z;
    Jx.m[12 + 3] = np.x; Jx.m[12 + 4] = np.y; Jx.m[12 + 5] = np.z;
    // plane side: [[B'^T R^T B, 0], [t^T B, 1]]
    const V3 Rb1 = mtv(R, B.b1), Rb2 = mtv(R, B.b2);
    Jp.m[0] = dot3(Bp.b1, Rb1); Jp.m[1] = dot3(Bp.b1, Rb2);
    Jp.m[6] = dot3(Bp.b2, Rb1); Jp.m[7] = dot3(Bp.b2, Rb2);
    Jp.m[12] = dot3(B.b1, X.t); Jp.m[13] = dot3(B.b2, X.t); Jp.m[14] = 1.0;
  }
}

// Cal3DS2 + body_P_sensor (fgo::CamCalib, device_plan.hpp) is shared by all reprojection factors of a context
template <bool WITH_JAC>
__device__ __forceinline__ void reproj_factor(const Pose &X, V3 pw, double u, double v, const CamCalib &K, double e[6], M6 &Jx, M6 &Jp) {
  Pose B;
  B.t = {K.bps[0], K.bps[1], K.bps[2]};
  B.q = {K.bps[3], K.bps[4], K.bps[5], K.bps[6]};
  const Pose cam = pose_mul(X, B);
  const M3 Rc = qmat(cam.q);
  const V3 q = mtv(Rc, pw - cam.t);
  e[2] = e[3] = e[4] = e[5] = 0;
  if (WITH_JAC) {
#pragma unroll
    for (int k = 0; k < 36; ++k) { Jx.m[k] = 0; Jp.m[k] = 0; }
  }
  if (q.z <= 0) { e[0] = e[1] = 2.0 * K.fx; return; }          // throwCheirality = false
  const double dz = 1.0 / q.z, xn = q.x * dz, yn = q.y * dz;
  const double xx = xn * xn, yy = yn * yn, xy = xn * yn, rr = xx + yy;
  const double g = 1. + K.k1 * rr + K.k2 * rr * rr;
  const double dx = 2. * K.p1 * xy + K.p2 * (rr + 2. * xx), dy = 2. * K.p2 * xy + K.p1 * (rr + 2. * yy);
  const double pnx = g * xn + dx, pny = g * yn + dy;
  e[0] = K.fx * pnx + K.s * pny + K.u0 - u;
  e[1] = K.fy * pny + K.v0 - v;
  if (WITH_JAC) {
    const double drdx = 2. * xn, drdy = 2. * yn;
    const double dgdx = K.k1 * drdx + K.k2 * 2. * rr * drdx, dgdy = K.k1 * drdy + K.k2 * 2. * rr * drdy;
    const double dDxdx = 2. * K.p1 * yn + K.p2 * (drdx + 4. * xn), dDxdy = 2. * K.p1 * xn + K.p2 * drdy;
    const double dDydx = 2. * K.p2 * yn + K.p1 * drdx, dDydy = 2. * K.p2 * xn + K.p1 * (drdy + 4. * yn);
    const double D00 = g + xn * dgdx + dDxdx, D01 = xn * dgdy + dDxdy, D10 = yn * dgdx + dDydx, D11 = g + yn * dgdy + dDydy;
    const double A00 = K.fx * D00 + K.s * D10, A01 = K.fx * D01 + K.s * D11, A10 = K.fy * D10, A11 = K.fy * D11;
    const double Dp0[6] = {xy, -(1 + xx), yn, -dz, 0, dz * xn}, Dp1[6] = {1 + yy, -xy, -xn, 0, -dz, dz * yn};
    double Hc0[6], Hc1[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { Hc0[c] = A00 * Dp0[c] + A01 * Dp1[c]; Hc1[c] = A10 * Dp0[c] + A11 * Dp1[c]; }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double t0 = 0, t1 = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { t0 += Hc0[k] * K.ad[k * 6 + c]; t1 += Hc1[k] * K.ad[k * 6 + c]; }
      Jx.m[c] = t0; Jx.m[6 + c] = t1;
    }
    // d(xn, yn)/d p_c = [[dz, 0, -dz xn], [0, dz, -dz yn]],  d p_c / d p_w = Rc^T
    const V3 E0 = {dz, 0, -dz * xn}, E1 = {0, dz, -dz * yn};
    const V3 ER0 = mv(Rc, E0), ER1 = mv(Rc, E1);          // (E Rc^T)_row = Rc E_row
    Jp.m[0] = A00 * ER0.x + A01 * ER1.x; Jp.m[1] = A00 * ER0.y + A01 * ER1.y; Jp.m[2] = A00 * ER0.z + A01 * ER1.z;
    Jp.m[6] = A10 * ER0.x + A11 * ER1.x; Jp.m[7] = A10 * ER0.y + A11 * ER1.y; Jp.m[8] = A10 * ER0.z + A11 * ER1.z;
  }
}

}  // namespace dev
}  // namespace fgo
