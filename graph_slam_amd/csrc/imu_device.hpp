// Device-side CombinedImuFactor (gfx950, f64): 15-dim residual over (X_i, V_i, X_j, V_j, B_i, B_j), on-manifold
// preintegration payload, first-order bias correction.  Built by the reference at gtsam/test_ba_imu_graph.cpp:239-244 /
// test_vro_imu_graph.cpp:191-196 from the payload of CImuBase::predictNext (gtsam/imu_base.cpp:72-87).
// Residual: [ Log(Rj^T Ri dRc) ; Rj^T (p_pred - pj) ; Rj^T (v_pred - vj) ; b_i - b_j ], order theta p v ba bg.
#pragma once
#include "device_plan.hpp"
#include "pose3_device.hpp"

namespace fgo {
namespace dev {

__device__ __forceinline__ M3 ld3(const double *p) { return {{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]}}; }
// right Jacobian of SO(3)
__device__ __forceinline__ M3 so3_dexp(V3 w) {
  const double th2 = dot3(w, w), th = sqrt(th2);
  const M3 W = skew(w), W2 = mm(W, W);
  double a, b;
  if (th < 1e-5) { a = 0.5 - th2 / 24.0; b = 1.0 / 6.0 - th2 / 120.0; }
  else { a = (1 - cos(th)) / th2; b = (th - sin(th)) / (th2 * th); }
  M3 J;
#pragma unroll
  for (int k = 0; k < 9; ++k) J.m[k] = -a * W.m[k] + b * W2.m[k];
  J.m[0] += 1; J.m[4] += 1; J.m[8] += 1;
  return J;
}
// The factor's Jacobian has a fixed pattern: fifteen 3x3 pieces (below, in this order) plus constant +-1 entries.  imu_factor
// writes the pieces one after the other (9 doubles each, row-major); imu_piece_pos maps entry k of that list to its place in
// the 15 x 36 Jacobian (variable u in columns 6 u .. 6 u + 5, zero padded) stored as rows of `js` doubles.
constexpr int IMU_NPIECE = 15;
__device__ __forceinline__ int imu_piece_pos(int k, int js) {
  //                                  pose_i        vel_i  pose_j    vel_j bias_i
  constexpr unsigned char U[15]  = {0, 0, 0, 0,   1, 1,  2, 2, 2,  3,    4, 4, 4, 4, 4};
  constexpr unsigned char R0[15] = {0, 3, 3, 6,   3, 6,  0, 3, 6,  6,    0, 3, 3, 6, 6};
  constexpr unsigned char C0[15] = {0, 0, 3, 0,   0, 0,  0, 0, 0,  0,    3, 0, 3, 0, 3};
  const int n = k / 9, e = k - 9 * n, a = e / 3, b = e - 3 * a;
  return (R0[n] + a) * js + 6 * U[n] + C0[n] + b;
}
// the constant entries: d(rp)/d(p_j) = -I (pose_j columns 3..5) and the bias rows, +I for bias_i and -I for bias_j
__device__ __forceinline__ void imu_const_entries(double *J, int js, int lane) {
  if (lane < 3) J[(3 + lane) * js + 12 + 3 + lane] = -1.0;
  else if (lane < 9) J[(9 + lane - 3) * js + 24 + lane - 3] = 1.0;
  else if (lane < 15) J[(9 + lane - 9) * js + 30 + lane - 9] = -1.0;
}
__device__ __forceinline__ void put_piece(double *J, int n, const M3 &M, double s) {
#pragma unroll
  for (int e = 0; e < 9; ++e) J[9 * n + e] = s * M.m[e];
}

// vals: the six variables' 8-slot values.  r[15]; with WITH_JAC the 15 Jacobian pieces (9 * IMU_NPIECE doubles) into J.
template <bool WITH_JAC>
__device__ void imu_factor(const ImuPayload &m, const double *const v[6], const double g[3], double r[15], double *J) {
  const Pose Xi = load_pose(v[0]), Xj = load_pose(v[2]);
  const V3 vi = {v[1][0], v[1][1], v[1][2]}, vj = {v[3][0], v[3][1], v[3][2]};
  const V3 dba = {v[4][0] - m.bhat[0], v[4][1] - m.bhat[1], v[4][2] - m.bhat[2]};
  const V3 dbg = {v[4][3] - m.bhat[3], v[4][4] - m.bhat[4], v[4][5] - m.bhat[5]};
  const M3 JRbg = ld3(m.J_R_bg), Jpba = ld3(m.J_p_ba), Jpbg = ld3(m.J_p_bg), Jvba = ld3(m.J_v_ba), Jvbg = ld3(m.J_v_bg);
  const V3 bo = mv(JRbg, dbg);
  const Q4 dR = {m.dR[0], m.dR[1], m.dR[2], m.dR[3]};
  const Q4 qcorr = qmul(dR, so3_exp(bo));
  const V3 dpc = V3{m.dp[0], m.dp[1], m.dp[2]} + mv(Jpba, dba) + mv(Jpbg, dbg);
  const V3 dvc = V3{m.dv[0], m.dv[1], m.dv[2]} + mv(Jvba, dba) + mv(Jvbg, dbg);
  const M3 Ri = qmat(Xi.q), Rj = qmat(Xj.q);
  const M3 RjTRi = mtm(Rj, Ri);
  const double dt = m.dt;
  const V3 G = {g[0], g[1], g[2]};
  const Q4 qe = qmul(qmul(qconj(Xj.q), Xi.q), qcorr);
  const V3 rR = so3_log(qe);
  const V3 dpos = {Xi.t.x + vi.x * dt + 0.5 * G.x * dt * dt - Xj.t.x, Xi.t.y + vi.y * dt + 0.5 * G.y * dt * dt - Xj.t.y,
                   Xi.t.z + vi.z * dt + 0.5 * G.z * dt * dt - Xj.t.z};
  const V3 dvel = {vi.x + G.x * dt - vj.x, vi.y + G.y * dt - vj.y, vi.z + G.z * dt - vj.z};
  const V3 rp = mtv(Rj, dpos + mv(Ri, dpc)), rv = mtv(Rj, dvel + mv(Ri, dvc));
  r[0] = rR.x; r[1] = rR.y; r[2] = rR.z; r[3] = rp.x; r[4] = rp.y; r[5] = rp.z; r[6] = rv.x; r[7] = rv.y; r[8] = rv.z;
#pragma unroll
  for (int k = 0; k < 6; ++k) r[9 + k] = v[4][k] - v[5][k];
  if (WITH_JAC) {
    const M3 Jri = so3_dlog(rR), C = qmat(qcorr), E = qmat(qe);
    const M3 RjT = mtrans(Rj);
    const M3 Jbg = mm(mm(Jri, so3_dexp(bo)), JRbg);
    // pose_i
    put_piece(J, 0, mm(Jri, mtrans(C)), 1.0);
    put_piece(J, 1, mm(RjTRi, skew(dpc)), -1.0);
    put_piece(J, 2, RjTRi, 1.0);
    put_piece(J, 3, mm(RjTRi, skew(dvc)), -1.0);
    // vel_i
    put_piece(J, 4, RjT, dt);
    put_piece(J, 5, RjT, 1.0);
    // pose_j
    put_piece(J, 6, mm(Jri, mtrans(E)), -1.0);
    put_piece(J, 7, skew(rp), 1.0);
    put_piece(J, 8, skew(rv), 1.0);
    // vel_j
    put_piece(J, 9, RjT, -1.0);
    // bias_i
    put_piece(J, 10, Jbg, 1.0);
    put_piece(J, 11, mm(RjTRi, Jpba), 1.0);
    put_piece(J, 12, mm(RjTRi, Jpbg), 1.0);
    put_piece(J, 13, mm(RjTRi, Jvba), 1.0);
    put_piece(J, 14, mm(RjTRi, Jvbg), 1.0);
  }
}

}  // namespace dev
}  // namespace fgo
