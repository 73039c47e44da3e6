// Device-side SE3 arithmetic for the pose-graph factors (gfx950, f64).
// Semantics restated from g2o's VertexSE3 / EdgeSE3 as used by the reference
// (g2o/g2o_graph.cpp:88,115-119,125-132): increment d = [dt; dq], X <- X * (R(dq), dt);
// error e = [t(D); vec q(D)], D = Z^-1 Xi^-1 Xj, quaternion normalised to w >= 0.
#pragma once
#include <hip/hip_runtime.h>

namespace fgo {
namespace dev {

struct V3 { double x, y, z; };
struct Q4 { double x, y, z, w; };
struct M3 { double m[9]; };   // row-major

__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.x + b.w * a.x + (a.y * b.z - a.z * b.y),
          a.w * b.y + b.w * a.y + (a.z * b.x - a.x * b.z),
          a.w * b.z + b.w * a.z + (a.x * b.y - a.y * b.x),
          a.w * b.w - (a.x * b.x + a.y * b.y + a.z * b.z)};
}
__device__ __forceinline__ Q4 qconj(Q4 a) { return {-a.x, -a.y, -a.z, a.w}; }
__device__ __forceinline__ M3 qmat(Q4 q) {
  M3 R;
  R.m[0] = 1 - 2 * (q.y * q.y + q.z * q.z); R.m[1] = 2 * (q.x * q.y - q.z * q.w);     R.m[2] = 2 * (q.x * q.z + q.y * q.w);
  R.m[3] = 2 * (q.x * q.y + q.z * q.w);     R.m[4] = 1 - 2 * (q.x * q.x + q.z * q.z); R.m[5] = 2 * (q.y * q.z - q.x * q.w);
  R.m[6] = 2 * (q.x * q.z - q.y * q.w);     R.m[7] = 2 * (q.y * q.z + q.x * q.w);     R.m[8] = 1 - 2 * (q.x * q.x + q.y * q.y);
  return R;
}
__device__ __forceinline__ V3 mv(const M3 &R, V3 v) {
  return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
          R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
__device__ __forceinline__ V3 mtv(const M3 &R, V3 v) {   // R^T v
  return {R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z,
          R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z};
}
__device__ __forceinline__ M3 mm(const M3 &A, const M3 &B) {
  M3 C;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C.m[r * 3 + c] = A.m[r * 3] * B.m[c] + A.m[r * 3 + 1] * B.m[3 + c] + A.m[r * 3 + 2] * B.m[6 + c];
  return C;
}
__device__ __forceinline__ M3 mtm(const M3 &A, const M3 &B) {   // A^T B
  M3 C;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C.m[r * 3 + c] = A.m[r] * B.m[c] + A.m[3 + r] * B.m[3 + c] + A.m[6 + r] * B.m[6 + c];
  return C;
}
__device__ __forceinline__ M3 madd(const M3 &A, const M3 &B) {
  M3 C;
#pragma unroll
  for (int k = 0; k < 9; ++k) C.m[k] = A.m[k] + B.m[k];
  return C;
}
__device__ __forceinline__ M3 mtrans(const M3 &A) {
  return {{A.m[0], A.m[3], A.m[6], A.m[1], A.m[4], A.m[7], A.m[2], A.m[5], A.m[8]}};
}
__device__ __forceinline__ M3 mzero() { return {{0, 0, 0, 0, 0, 0, 0, 0, 0}}; }

struct Pose { V3 t; Q4 q; };

// poses are stored padded to 8 doubles (64 B): tx ty tz qx qy qz qw pad
__device__ __forceinline__ Pose load_pose(const double *__restrict__ p) {
  const double4 a = *reinterpret_cast<const double4 *>(p);
  const double4 b = *reinterpret_cast<const double4 *>(p + 4);
  return {{a.x, a.y, a.z}, {a.w, b.x, b.y, b.z}};
}
__device__ __forceinline__ void store_pose(double *__restrict__ p, const Pose &X) {
  *reinterpret_cast<double4 *>(p) = make_double4(X.t.x, X.t.y, X.t.z, X.q.x);
  *reinterpret_cast<double4 *>(p + 4) = make_double4(X.q.y, X.q.z, X.q.w, 0.0);
}

// VertexSE3::oplusImpl with fromVectorMQT: identity rotation when |dq|^2 > 1
__device__ __forceinline__ Pose oplus(const Pose &X, const double d[6]) {
  double w = 1.0 - (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
  Q4 dq;
  if (w < 0) dq = {0, 0, 0, 1}; else dq = {d[3], d[4], d[5], sqrt(w)};
  const M3 R = qmat(X.q);
  Pose Y;
  Y.t = X.t + mv(R, V3{d[0], d[1], d[2]});
  Q4 q = qmul(X.q, dq);
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  Y.q = {q.x / n, q.y / n, q.z / n, q.w / n};
  return Y;
}

// Linearisation of one SE3 edge.  Jacobians are block upper-triangular:
//   Ji = [[Ai, Bi], [0, Ci]],  Jj = [[Aj, 0], [0, Cj]]
struct EdgeLin {
  double e[6];
  M3 Ai, Bi, Ci, Aj, Cj;
};

// A = Z^-1 (precomputed at upload: g2o keeps _inverseMeasurement too)
template <bool WITH_JAC>
__device__ __forceinline__ void edge_se3(const Pose &Xi, const Pose &Xj, const Pose &A, EdgeLin &L) {
  const Q4 qic = qconj(Xi.q);
  const Q4 qb = qmul(qic, Xj.q);
  const M3 Rit = qmat(qic);
  const V3 tb = mv(Rit, Xj.t - Xi.t);
  const Q4 qe = qmul(A.q, qb);
  const M3 Ra = qmat(A.q);
  const V3 te = mv(Ra, tb) + A.t;
  const double s = (qe.w < 0) ? -1.0 : 1.0;
  L.e[0] = te.x; L.e[1] = te.y; L.e[2] = te.z;
  L.e[3] = s * qe.x; L.e[4] = s * qe.y; L.e[5] = s * qe.z;
  if (WITH_JAC) {
    const double w = s * qe.w, vx = s * qe.x, vy = s * qe.y, vz = s * qe.z;
    L.Aj = qmat(qe);
    L.Cj = {{w, -vz, vy, vz, w, -vx, -vy, vx, w}};
#pragma unroll
    for (int k = 0; k < 9; ++k) L.Ai.m[k] = -Ra.m[k];
    const M3 S2 = {{0, -2 * tb.z, 2 * tb.y, 2 * tb.z, 0, -2 * tb.x, -2 * tb.y, 2 * tb.x, 0}};
    L.Bi = mm(Ra, S2);
    const M3 P = {{qb.w, qb.z, -qb.y, -qb.z, qb.w, qb.x, qb.y, -qb.x, qb.w}};        // wb I - [vb]x
    const M3 Q = {{A.q.w, -A.q.z, A.q.y, A.q.z, A.q.w, -A.q.x, -A.q.y, A.q.x, A.q.w}};  // wa I + [va]x
    M3 M = mm(P, Q);
    const double vb[3] = {qb.x, qb.y, qb.z}, va[3] = {A.q.x, A.q.y, A.q.z};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) L.Ci.m[r * 3 + c] = -s * (M.m[r * 3 + c] - vb[r] * va[c]);
  }
}

}  // namespace dev
}  // namespace fgo
