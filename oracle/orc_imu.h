/* ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: GTSAM 4.0 is not vendored and its preintegration type is a
 * build flag (GTSAM_TANGENT_PREINTEGRATION; gtsam/imu_base.h:73 uses the PreintegrationType typedef); the
 * reference's only IMU example needs an external CSV (gtsam/test/ImuFactorsExample.cpp:75).  The ON-MANIFOLD
 * preintegration (Forster et al.; GTSAM's ManifoldPreintegration) is restated and documented in DESIGN.md.
 *
 * What the reference does with it:
 *   params     PreintegratedCombinedMeasurements::Params::MakeSharedD(9.71)  -> n_gravity = (0, 0, +g)   imu_base.cpp:258-263
 *              VN100 noise: gyro/acc white noise, bias random walk, integration 1e-4, biasAccOmegaInt 1e-3    imu_vn100.cpp:24-67
 *   integrate  integrateMeasurement(acc = imu.tail<3>(), gyro = imu.head<3>(), dt) per sample               imu_base.cpp:72-87
 *   factor     CombinedImuFactor(X(i-1), V(i-1), X(i), V(i), B(i-1), B(i), *preint)                         test_ba_imu_graph.cpp:239-244
 *
 * Preintegrated quantities: dR (quaternion), dp, dv, dt, bias Jacobians dR/dbg, dp/dba, dp/dbg, dv/dba, dv/dbg, the
 * linearisation bias bhat = [ba; bg], and the 15x15 covariance in the order [theta, p, v, ba, bg].
 * Factor residual (15): [ Log(Rj^T Ri dRc) ; Rj^T (p_pred - pj) ; Rj^T (v_pred - vj) ; b_i - b_j ] with
 *   dRc = dR Exp(J_Rbg dbg), dpc = dp + J_pba dba + J_pbg dbg, dvc likewise, db = b_i - bhat,
 *   p_pred = pi + vi dt + g dt^2/2 + Ri dpc,  v_pred = vi + g dt + Ri dvc.
 */
#ifndef ORC_IMU_H
#define ORC_IMU_H
#include "orc_pose3.h"

typedef struct {
  double dt;
  double dR[4];          /* quaternion x y z w */
  double dp[3], dv[3];
  double J_R_bg[9], J_p_ba[9], J_p_bg[9], J_v_ba[9], J_v_bg[9];   /* row-major 3x3 */
  double bhat[6];        /* acc(3), gyro(3) */
  double cov[225];       /* row-major 15x15, order theta p v ba bg */
} orc_preint;

typedef struct {
  double acc_cov, gyro_cov, integ_cov, bias_acc_cov, bias_gyro_cov, bias_acc_omega_int;   /* isotropic variances */
  double gravity[3];
} orc_imu_params;

/* the reference's VN100 settings (imu_vn100.cpp:31-62) with MakeSharedD(9.71) gravity (imu_base.cpp:261) */
static inline void orc_imu_params_vn100(orc_imu_params *p) {
  const double fps = 200, hour = 3600, g = 9.81, d2r = M_PI / 180.0;
  const double accel_noise_sigma = 0.14 * 1e-3 * g, gyro_noise_sigma = 0.0035 * d2r;
  const double accel_bias_rw_sigma = (0.04 * 1e-3 * g) * sqrt(fps), gyro_bias_rw_sigma = (10 * d2r / hour) * sqrt(fps);
  p->acc_cov = accel_noise_sigma * accel_noise_sigma; p->gyro_cov = gyro_noise_sigma * gyro_noise_sigma;
  p->integ_cov = 1e-4; p->bias_acc_cov = accel_bias_rw_sigma * accel_bias_rw_sigma;
  p->bias_gyro_cov = gyro_bias_rw_sigma * gyro_bias_rw_sigma; p->bias_acc_omega_int = 1e-3;
  p->gravity[0] = 0; p->gravity[1] = 0; p->gravity[2] = 9.71;
}

static inline void orc_preint_reset(orc_preint *m, const double bhat[6]) {
  memset(m, 0, sizeof(*m));
  m->dR[3] = 1;
  memcpy(m->bhat, bhat, 6 * sizeof(double));
}

/* right Jacobian of SO(3): Rot3::ExpmapDerivative */
static inline void orc_so3_dexp(const double w[3], double J[9]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
  double W[9], W2[9];
  orc_skew(w, W);
  orc_m3mul(W, W, W2);
  double a, b;
  if (th < 1e-5) { a = 0.5 - th2 / 24.0; b = 1.0 / 6.0 - th2 / 120.0; }
  else { a = (1 - cos(th)) / th2; b = (th - sin(th)) / (th2 * th); }
  for (int k = 0; k < 9; ++k) J[k] = -a * W[k] + b * W2[k];
  J[0] += 1; J[4] += 1; J[8] += 1;
}

static inline void orc_m3t(const double A[9], double T[9]) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T[c * 3 + r] = A[r * 3 + c]; }

/* PreintegratedCombinedMeasurements::integrateMeasurement(acc, gyro, dt) */
static inline void orc_preint_integrate(orc_preint *m, const orc_imu_params *P, const double acc_meas[3], const double gyro_meas[3], double dt) {
  const double a[3] = {acc_meas[0] - m->bhat[0], acc_meas[1] - m->bhat[1], acc_meas[2] - m->bhat[2]};
  const double w[3] = {gyro_meas[0] - m->bhat[3], gyro_meas[1] - m->bhat[4], gyro_meas[2] - m->bhat[5]};
  double R[9], Sa[9], wdt[3] = {w[0] * dt, w[1] * dt, w[2] * dt}, qinc[4], Rinc[9], RincT[9], Jr[9];
  orc_qmat(m->dR, R);                       /* old dR */
  orc_skew(a, Sa);
  orc_so3_exp(wdt, qinc);
  orc_qmat(qinc, Rinc);
  orc_m3t(Rinc, RincT);
  orc_so3_dexp(wdt, Jr);
  const double dt22 = 0.5 * dt * dt;
  /* ---- covariance propagation first (uses the OLD state): x+ = f(x, noise), state order theta p v ba bg,
   * error-state in the local frames of the preintegrated NavState (theta, p, v right-perturbed: R Exp(dth), p + R dp, v + R dv) */
  double A[81];   /* 9x9 d(new)/d(old) in that chart */
  memset(A, 0, sizeof(A));
  double RS[9], t1[9];
  orc_m3mul(RincT, R, t1);    /* scratch */
  (void)t1;
  /* theta+ = RincT theta ; p+ (local) = RincT ( p + v dt - dt22 [a]x theta ) ; v+ (local) = RincT ( v - dt [a]x theta ) */
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[r * 9 + c] = RincT[r * 3 + c];
  double M1[9], M2[9];
  orc_m3mul(RincT, Sa, RS);
  for (int k = 0; k < 9; ++k) { M1[k] = -dt22 * RS[k]; M2[k] = -dt * RS[k]; }
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
    A[(3 + r) * 9 + c] = M1[r * 3 + c]; A[(3 + r) * 9 + 3 + c] = RincT[r * 3 + c]; A[(3 + r) * 9 + 6 + c] = dt * RincT[r * 3 + c];
    A[(6 + r) * 9 + c] = M2[r * 3 + c]; A[(6 + r) * 9 + 6 + c] = RincT[r * 3 + c];
  }
  /* d(new)/d acc: p: dt22 RincT, v: dt RincT ;  d(new)/d omega: theta: Jr dt */
  double Bm[27], Cm[27];
  memset(Bm, 0, sizeof(Bm)); memset(Cm, 0, sizeof(Cm));
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Bm[(3 + r) * 3 + c] = dt22 * RincT[r * 3 + c]; Bm[(6 + r) * 3 + c] = dt * RincT[r * 3 + c]; Cm[r * 3 + c] = dt * Jr[r * 3 + c]; }
  /* F 15x15 */
  double F[225];
  memset(F, 0, sizeof(F));
  for (int r = 0; r < 9; ++r) for (int c = 0; c < 9; ++c) F[r * 15 + c] = A[r * 9 + c];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
    F[r * 15 + 12 + c] = -Cm[r * 3 + c];                 /* theta_H_biasOmega */
    F[(3 + r) * 15 + 9 + c] = -Bm[(3 + r) * 3 + c];      /* pos_H_biasAcc */
    F[(6 + r) * 15 + 9 + c] = -Bm[(6 + r) * 3 + c];      /* vel_H_biasAcc */
  }
  for (int k = 9; k < 15; ++k) F[k * 15 + k] = 1;
  /* G Q G^T / dt (GTSAM's optimised form), isotropic variances */
  double G[225];
  memset(G, 0, sizeof(G));
  double thB[9], vB[9];      /* theta_H_biasOmega, vel_H_biasAcc */
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { thB[r * 3 + c] = -Cm[r * 3 + c]; vB[r * 3 + c] = -Bm[(6 + r) * 3 + c]; }
  double thBt[9], vBt[9], tt[9], vv[9];
  orc_m3t(thB, thBt); orc_m3t(vB, vBt);
  orc_m3mul(thB, thBt, tt); orc_m3mul(vB, vBt, vv);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
    G[r * 15 + c] = (1 / dt) * (P->gyro_cov + P->bias_acc_omega_int) * tt[r * 3 + c];
    G[(6 + r) * 15 + 6 + c] = (1 / dt) * (P->acc_cov + P->bias_acc_omega_int) * vv[r * 3 + c];
  }
  for (int k = 0; k < 3; ++k) { G[(3 + k) * 15 + 3 + k] = dt * P->integ_cov; G[(9 + k) * 15 + 9 + k] = dt * P->bias_acc_cov; G[(12 + k) * 15 + 12 + k] = dt * P->bias_gyro_cov; }
  /* (biasAccOmegaInt is diagonal in the reference, so GTSAM's off-diagonal D_v_R term vanishes) */
  double FC[225], N[225];
  for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) { double s = 0; for (int k = 0; k < 15; ++k) s += F[r * 15 + k] * m->cov[k * 15 + c]; FC[r * 15 + c] = s; }
  for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) { double s = 0; for (int k = 0; k < 15; ++k) s += FC[r * 15 + k] * F[c * 15 + k]; N[r * 15 + c] = s + G[r * 15 + c]; }
  memcpy(m->cov, N, sizeof(N));
  /* ---- bias Jacobians (use old dR and old Jacobians) */
  double D_acc_R[9], tmp[9], tmp2[9];
  orc_m3mul(R, Sa, tmp);
  for (int k = 0; k < 9; ++k) D_acc_R[k] = -tmp[k];                    /* -dR [a]x */
  orc_m3mul(D_acc_R, m->J_R_bg, tmp);                                  /* D_acc_biasOmega */
  for (int k = 0; k < 9; ++k) {
    m->J_p_ba[k] += m->J_v_ba[k] * dt - dt22 * R[k];
    m->J_p_bg[k] += m->J_v_bg[k] * dt + dt22 * tmp[k];
  }
  for (int k = 0; k < 9; ++k) { m->J_v_ba[k] += -R[k] * dt; m->J_v_bg[k] += tmp[k] * dt; }
  orc_m3mul(RincT, m->J_R_bg, tmp2);
  for (int k = 0; k < 9; ++k) m->J_R_bg[k] = tmp2[k] - Jr[k] * dt;
  /* ---- state */
  double Ra[3];
  orc_m3v(R, a, Ra);
  for (int k = 0; k < 3; ++k) { m->dp[k] += m->dv[k] * dt + dt22 * Ra[k]; m->dv[k] += dt * Ra[k]; }
  double qn[4];
  orc_qmul(m->dR, qinc, qn);
  memcpy(m->dR, qn, sizeof(qn));
  orc_qnormalize(m->dR);
  m->dt += dt;
}

/* CombinedImuFactor residual (15) and Jacobians.  Variables: pose_i[7], vel_i[3], pose_j[7], vel_j[3], bias_i[6], bias_j[6].
 * J*: row-major 15 x {6,3,6,3,6,6}; any may be NULL. */
static inline void orc_imu_factor(const double xi[7], const double vi[3], const double xj[7], const double vj[3], const double bi[6],
                                  const double bj[6], const orc_preint *m, const double g[3], double r[15], double *Jxi, double *Jvi,
                                  double *Jxj, double *Jvj, double *Jbi, double *Jbj) {
  double dba[3], dbg[3];
  for (int k = 0; k < 3; ++k) { dba[k] = bi[k] - m->bhat[k]; dbg[k] = bi[3 + k] - m->bhat[3 + k]; }
  double bo[3], qc[4], qcorr[4], dpc[3], dvc[3], t3[3], t4[3];
  orc_m3v(m->J_R_bg, dbg, bo);
  orc_so3_exp(bo, qc);
  orc_qmul(m->dR, qc, qcorr);                    /* dRc = dR Exp(J dbg) */
  orc_m3v(m->J_p_ba, dba, t3); orc_m3v(m->J_p_bg, dbg, t4);
  for (int k = 0; k < 3; ++k) dpc[k] = m->dp[k] + t3[k] + t4[k];
  orc_m3v(m->J_v_ba, dba, t3); orc_m3v(m->J_v_bg, dbg, t4);
  for (int k = 0; k < 3; ++k) dvc[k] = m->dv[k] + t3[k] + t4[k];
  double Ri[9], Rj[9], RjT[9], RjTRi[9], C[9], CT[9];
  orc_qmat(xi + 3, Ri); orc_qmat(xj + 3, Rj); orc_m3t(Rj, RjT);
  orc_m3mul(RjT, Ri, RjTRi);
  orc_qmat(qcorr, C); orc_m3t(C, CT);
  const double dt = m->dt;
  double qjc[4], qe1[4], qe[4];
  orc_qconj(xj + 3, qjc);
  orc_qmul(qjc, xi + 3, qe1);
  orc_qmul(qe1, qcorr, qe);                      /* E = Rj^T Ri dRc */
  orc_so3_log(qe, r);
  double Rdp[3], Rdv[3], dpos[3], dvel[3];
  orc_m3v(Ri, dpc, Rdp); orc_m3v(Ri, dvc, Rdv);
  for (int k = 0; k < 3; ++k) {
    dpos[k] = xi[k] + vi[k] * dt + 0.5 * g[k] * dt * dt + Rdp[k] - xj[k];
    dvel[k] = vi[k] + g[k] * dt + Rdv[k] - vj[k];
  }
  orc_m3v(RjT, dpos, r + 3);
  orc_m3v(RjT, dvel, r + 6);
  for (int k = 0; k < 6; ++k) r[9 + k] = bi[k] - bj[k];
  if (!Jxi && !Jvi && !Jxj && !Jvj && !Jbi && !Jbj) return;
  double Jri[9], E[9], ET[9], Sdp[9], Sdv[9], T[9];
  orc_so3_dlog(r, Jri);                          /* Jr^-1(r_R) */
  orc_qmat(qe, E); orc_m3t(E, ET);
  orc_skew(dpc, Sdp); orc_skew(dvc, Sdv);
#define SET33(J, ncols, r0, c0, M, s) for (int a_ = 0; a_ < 3; ++a_) for (int b_ = 0; b_ < 3; ++b_) (J)[((r0) + a_) * (ncols) + (c0) + b_] = (s) * (M)[a_ * 3 + b_]
  if (Jxi) {
    memset(Jxi, 0, 90 * sizeof(double));
    orc_m3mul(Jri, CT, T); SET33(Jxi, 6, 0, 0, T, 1.0);
    orc_m3mul(RjTRi, Sdp, T); SET33(Jxi, 6, 3, 0, T, -1.0);
    SET33(Jxi, 6, 3, 3, RjTRi, 1.0);
    orc_m3mul(RjTRi, Sdv, T); SET33(Jxi, 6, 6, 0, T, -1.0);
  }
  if (Jvi) { memset(Jvi, 0, 45 * sizeof(double)); SET33(Jvi, 3, 3, 0, RjT, dt); SET33(Jvi, 3, 6, 0, RjT, 1.0); }
  if (Jxj) {
    memset(Jxj, 0, 90 * sizeof(double));
    orc_m3mul(Jri, ET, T); SET33(Jxj, 6, 0, 0, T, -1.0);
    double S[9];
    orc_skew(r + 3, S); SET33(Jxj, 6, 3, 0, S, 1.0);
    for (int k = 0; k < 3; ++k) Jxj[(3 + k) * 6 + 3 + k] = -1.0;
    orc_skew(r + 6, S); SET33(Jxj, 6, 6, 0, S, 1.0);
  }
  if (Jvj) { memset(Jvj, 0, 45 * sizeof(double)); SET33(Jvj, 3, 6, 0, RjT, -1.0); }
  if (Jbi) {
    memset(Jbi, 0, 90 * sizeof(double));
    double Jrb[9], T2[9];
    orc_so3_dexp(bo, Jrb);
    orc_m3mul(Jri, Jrb, T); orc_m3mul(T, m->J_R_bg, T2); SET33(Jbi, 6, 0, 3, T2, 1.0);
    orc_m3mul(RjTRi, m->J_p_ba, T); SET33(Jbi, 6, 3, 0, T, 1.0);
    orc_m3mul(RjTRi, m->J_p_bg, T); SET33(Jbi, 6, 3, 3, T, 1.0);
    orc_m3mul(RjTRi, m->J_v_ba, T); SET33(Jbi, 6, 6, 0, T, 1.0);
    orc_m3mul(RjTRi, m->J_v_bg, T); SET33(Jbi, 6, 6, 3, T, 1.0);
    for (int k = 0; k < 6; ++k) Jbi[(9 + k) * 6 + k] = 1.0;
  }
  if (Jbj) { memset(Jbj, 0, 90 * sizeof(double)); for (int k = 0; k < 6; ++k) Jbj[(9 + k) * 6 + k] = -1.0; }
#undef SET33
}
#endif
