// Host-side IMU preintegration (C-ABI fgo_preint_*): the product's counterpart of the reference's imu_interface
// library, which owns a gtsam::PreintegratedCombinedMeasurements and feeds it sample by sample
// (gtsam/imu_base.cpp:72-87 integrateMeasurement loop, :156-170 predictBetween, :258-263 params;
// gtsam/imu_vn100.cpp:24-67 VN100 noise).  Like in the reference this is a per-factor host pre-process; the
// optimiser kernels consume its output (fgo_add_imu_combined).  On-manifold (Forster et al.) form with GTSAM's
// combined 15x15 covariance propagation; error state order theta, p, v, bias_acc, bias_gyro, local (body-frame) charts.
#include <cmath>
#include <cstring>
#include "../../include/fgo.h"

namespace {

struct Mat3 {
  double a[9];
  double &operator()(int r, int c) { return a[r * 3 + c]; }
  double operator()(int r, int c) const { return a[r * 3 + c]; }
};
Mat3 zero3() { Mat3 m; std::memset(m.a, 0, sizeof(m.a)); return m; }
Mat3 mul(const Mat3 &A, const Mat3 &B) {
  Mat3 C = zero3();
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) C(r, c) += A(r, k) * B(k, c);
  return C;
}
Mat3 transpose(const Mat3 &A) { Mat3 T; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T(c, r) = A(r, c); return T; }
Mat3 hat(const double w[3]) { Mat3 S = zero3(); S(0, 1) = -w[2]; S(0, 2) = w[1]; S(1, 0) = w[2]; S(1, 2) = -w[0]; S(2, 0) = -w[1]; S(2, 1) = w[0]; return S; }
Mat3 from_array(const double *p) { Mat3 m; std::memcpy(m.a, p, sizeof(m.a)); return m; }
void apply(const Mat3 &A, const double x[3], double y[3]) { for (int r = 0; r < 3; ++r) y[r] = A(r, 0) * x[0] + A(r, 1) * x[1] + A(r, 2) * x[2]; }

Mat3 rot_of_quat(const double q[4]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  Mat3 R;
  R(0, 0) = 1 - 2 * (y * y + z * z); R(0, 1) = 2 * (x * y - z * w); R(0, 2) = 2 * (x * z + y * w);
  R(1, 0) = 2 * (x * y + z * w); R(1, 1) = 1 - 2 * (x * x + z * z); R(1, 2) = 2 * (y * z - x * w);
  R(2, 0) = 2 * (x * z - y * w); R(2, 1) = 2 * (y * z + x * w); R(2, 2) = 1 - 2 * (x * x + y * y);
  return R;
}
void quat_mul(const double a[4], const double b[4], double r[4]) {
  r[0] = a[3] * b[0] + b[3] * a[0] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] + b[3] * a[1] + a[2] * b[0] - a[0] * b[2];
  r[2] = a[3] * b[2] + b[3] * a[2] + a[0] * b[1] - a[1] * b[0];
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
void quat_exp(const double w[3], double q[4]) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = std::sqrt(t2);
  const double s = t < 1e-10 ? 0.5 - t2 / 48.0 : std::sin(0.5 * t) / t;
  q[0] = s * w[0]; q[1] = s * w[1]; q[2] = s * w[2]; q[3] = std::cos(0.5 * t);
}
// right Jacobian of SO(3)
Mat3 right_jacobian(const double w[3]) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = std::sqrt(t2);
  const Mat3 W = hat(w), W2 = mul(W, W);
  double a, b;
  if (t < 1e-5) { a = 0.5 - t2 / 24.0; b = 1.0 / 6.0 - t2 / 120.0; } else { a = (1 - std::cos(t)) / t2; b = (t - std::sin(t)) / (t2 * t); }
  Mat3 J;
  for (int k = 0; k < 9; ++k) J.a[k] = -a * W.a[k] + b * W2.a[k];
  J(0, 0) += 1; J(1, 1) += 1; J(2, 2) += 1;
  return J;
}

}  // namespace

extern "C" {

void fgo_imu_params_vn100(fgo_imu_params *p) {
  if (!p) return;
  const double fps = 200, hour = 3600, g = 9.81, d2r = M_PI / 180.0;
  const double acc_sigma = 0.14 * 1e-3 * g, gyro_sigma = 0.0035 * d2r;                       // imu_vn100.cpp:33-41
  const double acc_rw = (0.04 * 1e-3 * g) * std::sqrt(fps), gyro_rw = (10 * d2r / hour) * std::sqrt(fps);   // :42-43
  p->acc_cov = acc_sigma * acc_sigma; p->gyro_cov = gyro_sigma * gyro_sigma;
  p->integ_cov = 1e-4;                                                                       // :50
  p->bias_acc_cov = acc_rw * acc_rw; p->bias_gyro_cov = gyro_rw * gyro_rw;
  p->bias_acc_omega_int = 1e-3;                                                              // :53
  p->gravity[0] = 0; p->gravity[1] = 0; p->gravity[2] = 9.71;                                // MakeSharedD(9.71), imu_base.cpp:261
}

void fgo_preint_reset(fgo_preint *m, const double bias_hat6[6]) {
  if (!m) return;
  std::memset(m, 0, sizeof(*m));
  m->dR[3] = 1.0;
  if (bias_hat6) std::memcpy(m->bhat, bias_hat6, 6 * sizeof(double));
}

void fgo_preint_integrate(fgo_preint *m, const fgo_imu_params *P, const double acc_meas[3], const double gyro_meas[3], double dt) {
  if (!m || !P || !acc_meas || !gyro_meas || !(dt > 0)) return;
  double acc[3], om[3], odt[3];
  for (int k = 0; k < 3; ++k) { acc[k] = acc_meas[k] - m->bhat[k]; om[k] = gyro_meas[k] - m->bhat[3 + k]; odt[k] = om[k] * dt; }
  const Mat3 R = rot_of_quat(m->dR);            // rotation BEFORE this sample
  double qinc[4];
  quat_exp(odt, qinc);
  const Mat3 IncT = transpose(rot_of_quat(qinc)), Jr = right_jacobian(odt), A = hat(acc);
  const double h = 0.5 * dt * dt;

  // --- covariance: Sigma <- F Sigma F^T + G Q G^T, error state in the local charts of (dR, dp, dv)
  double F[15][15];
  std::memset(F, 0, sizeof(F));
  const Mat3 IA = mul(IncT, A);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      F[r][c] = IncT(r, c);
      F[3 + r][c] = -h * IA(r, c); F[3 + r][3 + c] = IncT(r, c); F[3 + r][6 + c] = dt * IncT(r, c);
      F[6 + r][c] = -dt * IA(r, c); F[6 + r][6 + c] = IncT(r, c);
      F[r][12 + c] = -dt * Jr(r, c);            // theta wrt gyro bias
      F[3 + r][9 + c] = -h * IncT(r, c);        // p wrt acc bias
      F[6 + r][9 + c] = -dt * IncT(r, c);       // v wrt acc bias
    }
  for (int k = 9; k < 15; ++k) F[k][k] = 1.0;
  double Q[15][15];
  std::memset(Q, 0, sizeof(Q));
  const Mat3 JJ = mul(Jr, transpose(Jr));       // (dt Jr)(dt Jr)^T / dt = dt Jr Jr^T
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      Q[r][c] = dt * (P->gyro_cov + P->bias_acc_omega_int) * JJ(r, c);
      Q[6 + r][6 + c] = (r == c) ? dt * (P->acc_cov + P->bias_acc_omega_int) : 0.0;   // (dt Inc^T)(dt Inc^T)^T / dt = dt I
    }
  for (int k = 0; k < 3; ++k) { Q[3 + k][3 + k] = dt * P->integ_cov; Q[9 + k][9 + k] = dt * P->bias_acc_cov; Q[12 + k][12 + k] = dt * P->bias_gyro_cov; }
  double FS[15][15], S2[15][15];
  for (int r = 0; r < 15; ++r)
    for (int c = 0; c < 15; ++c) { double s = 0; for (int k = 0; k < 15; ++k) s += F[r][k] * m->cov[k * 15 + c]; FS[r][c] = s; }
  for (int r = 0; r < 15; ++r)
    for (int c = 0; c < 15; ++c) { double s = 0; for (int k = 0; k < 15; ++k) s += FS[r][k] * F[c][k]; S2[r][c] = s + Q[r][c]; }
  for (int r = 0; r < 15; ++r) for (int c = 0; c < 15; ++c) m->cov[r * 15 + c] = S2[r][c];

  // --- bias Jacobians (all right-hand sides use the values before this sample)
  const Mat3 RA = mul(R, A);
  const Mat3 JRbg = from_array(m->J_R_bg);
  const Mat3 dacc_dbg = mul(RA, JRbg);          // d(R acc)/d bg = -R [acc]x dR/dbg
  for (int k = 0; k < 9; ++k) {
    m->J_p_ba[k] += m->J_v_ba[k] * dt - h * R.a[k];
    m->J_p_bg[k] += m->J_v_bg[k] * dt - h * dacc_dbg.a[k];
  }
  for (int k = 0; k < 9; ++k) { m->J_v_ba[k] -= R.a[k] * dt; m->J_v_bg[k] -= dacc_dbg.a[k] * dt; }
  const Mat3 newJ = mul(IncT, JRbg);
  for (int k = 0; k < 9; ++k) m->J_R_bg[k] = newJ.a[k] - Jr.a[k] * dt;

  // --- the preintegrated state
  double Ra[3];
  apply(R, acc, Ra);
  for (int k = 0; k < 3; ++k) { m->dp[k] += m->dv[k] * dt + h * Ra[k]; m->dv[k] += Ra[k] * dt; }
  double q[4];
  quat_mul(m->dR, qinc, q);
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) m->dR[k] = q[k] / n;
  m->dt += dt;
}

void fgo_preint_predict(const fgo_preint *m, const double g[3], const double xi[7], const double vi[3], const double bi[6],
                        double xj[7], double vj[3]) {
  if (!m || !g || !xi || !vi || !bi || !xj || !vj) return;
  double dba[3], dbg[3], bo[3], t1[3], t2[3], dpc[3], dvc[3], qc[4], qcorr[4];
  for (int k = 0; k < 3; ++k) { dba[k] = bi[k] - m->bhat[k]; dbg[k] = bi[3 + k] - m->bhat[3 + k]; }
  apply(from_array(m->J_R_bg), dbg, bo);
  quat_exp(bo, qc);
  quat_mul(m->dR, qc, qcorr);
  apply(from_array(m->J_p_ba), dba, t1); apply(from_array(m->J_p_bg), dbg, t2);
  for (int k = 0; k < 3; ++k) dpc[k] = m->dp[k] + t1[k] + t2[k];
  apply(from_array(m->J_v_ba), dba, t1); apply(from_array(m->J_v_bg), dbg, t2);
  for (int k = 0; k < 3; ++k) dvc[k] = m->dv[k] + t1[k] + t2[k];
  const Mat3 Ri = rot_of_quat(xi + 3);
  apply(Ri, dpc, t1); apply(Ri, dvc, t2);
  for (int k = 0; k < 3; ++k) { xj[k] = xi[k] + vi[k] * m->dt + 0.5 * g[k] * m->dt * m->dt + t1[k]; vj[k] = vi[k] + g[k] * m->dt + t2[k]; }
  double q[4];
  quat_mul(xi + 3, qcorr, q);
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) xj[3 + k] = q[k] / n;
}

}  // extern "C"
