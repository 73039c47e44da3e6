// Batched IMU preintegration on the MI355X: the reference integrates the samples between two
// keyframes one factor at a time on the CPU (CImuBase::predictNext, gtsam/imu_base.cpp:72-87: a loop of
// PreintegratedCombinedMeasurements::integrateMeasurement); the factors are independent, so fgo_preint_batch runs one
// wave per factor.  Same arithmetic as the host-side fgo_preint_integrate (csrc/imu_preint.cpp), in the same order:
// every lane carries the small state (dR, dp, dv, the five 3x3 bias Jacobians) redundantly in registers; the 15x15
// covariance propagation  Sigma <- F Sigma F^T + Q  is spread over the lanes through LDS (225 entries, 4 per lane).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include "../../include/fgo.h"

namespace {

struct M3 { double a[9]; };
__device__ __forceinline__ M3 mul3(const M3 &A, const M3 &B) {
  M3 C;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += A.a[r * 3 + k] * B.a[k * 3 + c];
      C.a[r * 3 + c] = s;
    }
  return C;
}
__device__ __forceinline__ M3 tr3(const M3 &A) {
  M3 T;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) T.a[c * 3 + r] = A.a[r * 3 + c];
  return T;
}
__device__ __forceinline__ M3 hat3(const double w[3]) { return M3{{0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}}; }
__device__ __forceinline__ M3 rot_of_quat(const double q[4]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  return M3{{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
             2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
}
__device__ __forceinline__ void quat_mul(const double a[4], const double b[4], double r[4]) {
  r[0] = a[3] * b[0] + b[3] * a[0] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] + b[3] * a[1] + a[2] * b[0] - a[0] * b[2];
  r[2] = a[3] * b[2] + b[3] * a[2] + a[0] * b[1] - a[1] * b[0];
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
__device__ __forceinline__ void quat_exp(const double w[3], double q[4]) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = sqrt(t2);
  const double s = t < 1e-10 ? 0.5 - t2 / 48.0 : sin(0.5 * t) / t;
  q[0] = s * w[0]; q[1] = s * w[1]; q[2] = s * w[2]; q[3] = cos(0.5 * t);
}
__device__ __forceinline__ M3 right_jacobian(const double w[3]) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = sqrt(t2);
  const M3 W = hat3(w), W2 = mul3(W, W);
  double a, b;
  if (t < 1e-5) { a = 0.5 - t2 / 24.0; b = 1.0 / 6.0 - t2 / 120.0; } else { a = (1 - cos(t)) / t2; b = (t - sin(t)) / (t2 * t); }
  M3 J;
#pragma unroll
  for (int k = 0; k < 9; ++k) J.a[k] = -a * W.a[k] + b * W2.a[k];
  J.a[0] += 1; J.a[4] += 1; J.a[8] += 1;
  return J;
}

// sample layout: acc[3 * s], gyro[3 * s]; factor f owns samples [sample_ptr[f], sample_ptr[f + 1])
__global__ __launch_bounds__(64) void k_preint_batch(int64_t n, const int64_t *__restrict__ sample_ptr, const double *__restrict__ acc_all,
                                                     const double *__restrict__ gyro_all, double dt, const double *__restrict__ bias_hat,
                                                     fgo_imu_params Pm, fgo_preint *__restrict__ out) {
  __shared__ double cov[225], F[225], FS[225];
  __shared__ double sm[36];               // IncT (9), IA (9), Jr (9), JJ (9) for the lanes' dynamic indexing
  const int64_t f = blockIdx.x;
  if (f >= n) return;
  const int lane = threadIdx.x;
  double dR[4] = {0, 0, 0, 1}, dp[3] = {0, 0, 0}, dv[3] = {0, 0, 0}, bh[6], tsum = 0;
  M3 JRbg = {{0}}, Jpba = {{0}}, Jpbg = {{0}}, Jvba = {{0}}, Jvbg = {{0}};
#pragma unroll
  for (int k = 0; k < 6; ++k) bh[k] = bias_hat ? bias_hat[6 * f + k] : 0.0;
  for (int e = lane; e < 225; e += 64) cov[e] = 0.0;
  const double h = 0.5 * dt * dt;
  for (int64_t s = sample_ptr[f]; s < sample_ptr[f + 1]; ++s) {
    double acc[3], om[3], odt[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { acc[k] = acc_all[3 * s + k] - bh[k]; om[k] = gyro_all[3 * s + k] - bh[3 + k]; odt[k] = om[k] * dt; }
    const M3 R = rot_of_quat(dR);
    double qinc[4];
    quat_exp(odt, qinc);
    const M3 IncT = tr3(rot_of_quat(qinc)), Jr = right_jacobian(odt), A = hat3(acc);
    const M3 IA = mul3(IncT, A), JJ = mul3(Jr, tr3(Jr));
    __builtin_amdgcn_wave_barrier();      // previous sample's readers of sm / F are done (single wave, in-order LDS)
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) { sm[k] = IncT.a[k]; sm[9 + k] = IA.a[k]; sm[18 + k] = Jr.a[k]; sm[27 + k] = JJ.a[k]; }
    }
    __builtin_amdgcn_wave_barrier();
    // F (15x15), same entries as the host code
    for (int e = lane; e < 225; e += 64) {
      const int r = e / 15, c = e - 15 * r, rb = r / 3, cb = c / 3, i = (r - 3 * rb) * 3 + (c - 3 * cb);
      double v = 0.0;
      if (rb == 0 && cb == 0) v = sm[i];
      else if (rb == 1 && cb == 0) v = -h * sm[9 + i];
      else if (rb == 1 && cb == 1) v = sm[i];
      else if (rb == 1 && cb == 2) v = dt * sm[i];
      else if (rb == 2 && cb == 0) v = -dt * sm[9 + i];
      else if (rb == 2 && cb == 2) v = sm[i];
      else if (rb == 0 && cb == 4) v = -dt * sm[18 + i];
      else if (rb == 1 && cb == 3) v = -h * sm[i];
      else if (rb == 2 && cb == 3) v = -dt * sm[i];
      else if (r >= 9 && r == c) v = 1.0;
      F[e] = v;
    }
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < 225; e += 64) {
      const int r = e / 15, c = e - 15 * r;
      double sacc = 0;
#pragma unroll
      for (int k = 0; k < 15; ++k) sacc += F[r * 15 + k] * cov[k * 15 + c];
      FS[e] = sacc;
    }
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < 225; e += 64) {
      const int r = e / 15, c = e - 15 * r;
      double sacc = 0;
#pragma unroll
      for (int k = 0; k < 15; ++k) sacc += FS[r * 15 + k] * F[c * 15 + k];
      double qv = 0.0;
      if (r < 3 && c < 3) qv = dt * (Pm.gyro_cov + Pm.bias_acc_omega_int) * sm[27 + r * 3 + c];
      else if (r == c) {
        if (r < 6) qv = dt * Pm.integ_cov;
        else if (r < 9) qv = dt * (Pm.acc_cov + Pm.bias_acc_omega_int);
        else if (r < 12) qv = dt * Pm.bias_acc_cov;
        else qv = dt * Pm.bias_gyro_cov;
      }
      cov[e] = sacc + qv;
    }
    // bias Jacobians and the preintegrated state (values before this sample on the right-hand sides)
    const M3 RA = mul3(R, A), dacc_dbg = mul3(RA, JRbg);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      Jpba.a[k] += Jvba.a[k] * dt - h * R.a[k];
      Jpbg.a[k] += Jvbg.a[k] * dt - h * dacc_dbg.a[k];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) { Jvba.a[k] -= R.a[k] * dt; Jvbg.a[k] -= dacc_dbg.a[k] * dt; }
    const M3 newJ = mul3(IncT, JRbg);
#pragma unroll
    for (int k = 0; k < 9; ++k) JRbg.a[k] = newJ.a[k] - Jr.a[k] * dt;
    double Ra[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) Ra[r] = R.a[r * 3] * acc[0] + R.a[r * 3 + 1] * acc[1] + R.a[r * 3 + 2] * acc[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) { dp[k] += dv[k] * dt + h * Ra[k]; dv[k] += Ra[k] * dt; }
    double q[4];
    quat_mul(dR, qinc, q);
    const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) dR[k] = q[k] / nq;
    tsum += dt;
  }
  __builtin_amdgcn_wave_barrier();
  fgo_preint *o = out + f;
  for (int e = lane; e < 225; e += 64) o->cov[e] = cov[e];
  if (lane == 0) {
    o->dt = tsum;
#pragma unroll
    for (int k = 0; k < 4; ++k) o->dR[k] = dR[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o->dp[k] = dp[k]; o->dv[k] = dv[k]; }
#pragma unroll
    for (int k = 0; k < 9; ++k) { o->J_R_bg[k] = JRbg.a[k]; o->J_p_ba[k] = Jpba.a[k]; o->J_p_bg[k] = Jpbg.a[k]; o->J_v_ba[k] = Jvba.a[k]; o->J_v_bg[k] = Jvbg.a[k]; }
#pragma unroll
    for (int k = 0; k < 6; ++k) o->bhat[k] = bh[k];
  }
}

template <class T>
struct Dev {
  T *p = nullptr;
  ~Dev() { if (p) (void)hipFree(p); }
  hipError_t put(const T *h, size_t n) {
    hipError_t e = hipMalloc((void **)&p, sizeof(T) * (n ? n : 1));
    if (e != hipSuccess || !n || !h) return e;
    return hipMemcpy(p, h, sizeof(T) * n, hipMemcpyHostToDevice);
  }
};

}  // namespace

extern "C" int fgo_preint_batch(int device, int64_t n, const int64_t *sample_ptr, const double *acc, const double *gyro, double dt,
                                const double *bias_hat6, const fgo_imu_params *params, fgo_preint *out) {
  if (n < 0 || !sample_ptr || !params || !out || !(dt > 0)) return FGO_EINVAL;
  if (n == 0) return FGO_OK;
  if (sample_ptr[0] < 0) return FGO_EINVAL;           // a negative first offset would read before the sample buffers
  for (int64_t f = 0; f < n; ++f) if (sample_ptr[f + 1] < sample_ptr[f]) return FGO_EINVAL;
  const int64_t ns = sample_ptr[n];
  if (ns > 0 && (!acc || !gyro)) return FGO_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FGO_ENODEV;   // no CPU fallback
  if (hipSetDevice(device) != hipSuccess) return FGO_ENODEV;
  Dev<int64_t> d_ptr;
  Dev<double> d_acc, d_gyro, d_bias;
  Dev<fgo_preint> d_out;
  if (d_ptr.put(sample_ptr, (size_t)n + 1) != hipSuccess || d_acc.put(acc, (size_t)ns * 3) != hipSuccess ||
      d_gyro.put(gyro, (size_t)ns * 3) != hipSuccess || d_out.put(nullptr, (size_t)n) != hipSuccess)
    return FGO_ENOMEM;
  if (bias_hat6 && d_bias.put(bias_hat6, (size_t)n * 6) != hipSuccess) return FGO_ENOMEM;
  hipLaunchKernelGGL(k_preint_batch, dim3((unsigned)n), dim3(64), 0, 0, n, d_ptr.p, d_acc.p, d_gyro.p, dt, bias_hat6 ? d_bias.p : nullptr, *params, d_out.p);
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return FGO_ENUM;
  if (hipMemcpy(out, d_out.p, sizeof(fgo_preint) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return FGO_ENUM;
  return FGO_OK;
}
