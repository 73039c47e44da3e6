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
__device__ __forceinline__ void put33(double *J /*15x6*/, int r0, int c0, const M3 &M, double s) {
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) J[(r0 + a) * 6 + c0 + b] = s * M.m[a * 3 + b];
}

// vals: the six variables' 8-slot values.  r[15]; J[6][90] (15x6 each, zero padded) when WITH_JAC.
// COOP: the caller is a whole wave working on ONE factor with J in LDS -- J was zeroed by the caller and only
// `writer` lanes store the Jacobian blocks (every lane still evaluates the same 3x3 algebra in registers).
template <bool WITH_JAC, bool COOP = false>
__device__ void imu_factor(const ImuPayload &m, const double *const v[6], const double g[3], double r[15], double (*J)[90],
                           bool writer = true) {
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
    if (!COOP)
      for (int u = 0; u < 6; ++u)
        for (int k = 0; k < 90; ++k) J[u][k] = 0;
    const M3 Jri = so3_dlog(rR), C = qmat(qcorr), E = qmat(qe);
    const M3 RjT = mtrans(Rj);
    const M3 Jbg = mm(mm(Jri, so3_dexp(bo)), JRbg);
    if (COOP && !writer) return;
    // pose_i
    put33(J[0], 0, 0, mm(Jri, mtrans(C)), 1.0);
    put33(J[0], 3, 0, mm(RjTRi, skew(dpc)), -1.0);
    put33(J[0], 3, 3, RjTRi, 1.0);
    put33(J[0], 6, 0, mm(RjTRi, skew(dvc)), -1.0);
    // vel_i
    put33(J[1], 3, 0, RjT, dt);
    put33(J[1], 6, 0, RjT, 1.0);
    // pose_j
    put33(J[2], 0, 0, mm(Jri, mtrans(E)), -1.0);
    put33(J[2], 3, 0, skew(rp), 1.0);
    J[2][3 * 6 + 3] = -1.0; J[2][4 * 6 + 4] = -1.0; J[2][5 * 6 + 5] = -1.0;
    put33(J[2], 6, 0, skew(rv), 1.0);
    // vel_j
    put33(J[3], 6, 0, RjT, -1.0);
    // bias_i
    put33(J[4], 0, 3, Jbg, 1.0);
    put33(J[4], 3, 0, mm(RjTRi, Jpba), 1.0);
    put33(J[4], 3, 3, mm(RjTRi, Jpbg), 1.0);
    put33(J[4], 6, 0, mm(RjTRi, Jvba), 1.0);
    put33(J[4], 6, 3, mm(RjTRi, Jvbg), 1.0);
    for (int k = 0; k < 6; ++k) { J[4][(9 + k) * 6 + k] = 1.0; J[5][(9 + k) * 6 + k] = -1.0; }
  }
}

}  // namespace dev
}  // namespace fgo
