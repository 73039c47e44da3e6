// HIP kernels (gfx950, f64) for the GTSAM-semantics factors of the hot path: PriorFactor<Pose3> and
// BetweenFactor<Pose3> in the [omega; v] tangent with the exponential-map retraction — what
// CGraphGT::firstNode / addToGTSAM build (gtsam/gtsam_graph.cpp:338-341, 689-692) and
// LevenbergMarquardtOptimizer linearises each iteration (gtsam/gtsam_graph.cpp:1784-1788).
// Same gather-form assembly as k_linearize (kernels.hip): every H block written once, no FP atomics.
// The Jacobians are dense 6x6 here (dLog and Ad couple rotation and translation).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_plan.hpp"
#include "pose3_device.hpp"

namespace fgo {
using namespace dev;

namespace {
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double bsum4(double v, double *sh) {
  v = wsum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const double s = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return s;
}
__device__ __forceinline__ Pose load_soa_pose(const double *__restrict__ a, int64_t n, int64_t k) {
  Pose A;
  A.t = {a[0 * n + k], a[1 * n + k], a[2 * n + k]};
  A.q = {a[3 * n + k], a[4 * n + k], a[5 * n + k], a[6 * n + k]};
  return A;
}
__device__ __forceinline__ M6 load_soa_info(const double *__restrict__ info, int64_t n, int64_t k) {
  M6 W;
  int p = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) { const double v = info[(int64_t)p * n + k]; W.m[r * 6 + c] = v; W.m[c * 6 + r] = v; ++p; }
  return W;
}
__device__ __forceinline__ void mv6(const M6 &A, const double x[6], double y[6]) {
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double s = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) s += A.m[r * 6 + c] * x[c];
    y[r] = s;
  }
}
__device__ __forceinline__ void mtv6_sub(const M6 &A, const double x[6], double y[6]) {   // y -= A^T x
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double s = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) s += A.m[c * 6 + r] * x[c];
    y[r] -= s;
  }
}
__device__ __forceinline__ void store_block(double *__restrict__ o, const M6 &O, bool transpose) {
  if (!transpose) {
#pragma unroll
    for (int k = 0; k < 36; ++k) o[k] = O.m[k];
  } else {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) o[c * 6 + r] = O.m[r * 6 + c];
  }
}
}  // namespace

template <int G>
__global__ __launch_bounds__(256) void k_linearize_gtsam(DevPlan P, const double *__restrict__ poses,
                                                         double *__restrict__ Hblk, double *__restrict__ bvec,
                                                         double *__restrict__ chi_partial) {
  __shared__ double sh[4];
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t v = tid / G;
  const int g = (int)(tid % G);
  M6 D;
#pragma unroll
  for (int k = 0; k < 36; ++k) D.m[k] = 0;
  double gv[6] = {0, 0, 0, 0, 0, 0};
  double chi = 0;
  const bool live = v < P.n_poses;
  if (live) {
    const int64_t p0 = P.he_ptr[v], p1 = P.he_ptr[v + 1];
    for (int64_t p = p0 + g; p < p1; p += G) {
      const int he = P.he[p];
      const int64_t e = he >> 1;
      const int side = he & 1;
      const Pose Xi = load_pose(poses + 8 * (int64_t)P.edge_i[e]), Xj = load_pose(poses + 8 * (int64_t)P.edge_j[e]);
      const Pose Zinv = load_soa_pose(P.ainv, P.n_edges, e);
      const M6 W = load_soa_info(P.info, P.n_edges, e);
      double r[6], Wr[6];
      M6 Ji, Jj;
      between_pose3<true>(Xi, Xj, Zinv, r, Ji, Jj);
      mv6(W, r, Wr);
      if (side) {
#pragma unroll
        for (int k = 0; k < 6; ++k) chi += r[k] * Wr[k];
        const M6 K = m6mul(W, Jj);
        const M6 JK = m6tmul(Jj, K);
#pragma unroll
        for (int k = 0; k < 36; ++k) D.m[k] += JK.m[k];
        mtv6_sub(Jj, Wr, gv);
        const int slot = P.edge_slot[e];
        if (slot >= 0) store_block(Hblk + 36 * (int64_t)(slot >> 1), m6tmul(Ji, K), (slot & 1) != 0);
      } else {
        const M6 K = m6mul(W, Ji);
        const M6 JK = m6tmul(Ji, K);
#pragma unroll
        for (int k = 0; k < 36; ++k) D.m[k] += JK.m[k];
        mtv6_sub(Ji, Wr, gv);
      }
    }
    if (g == 0 && P.n_priors > 0) {
      for (int64_t q = P.prior_ptr[v]; q < P.prior_ptr[v + 1]; ++q) {
        const Pose X = load_pose(poses + 8 * v);
        const Pose Pinv = load_soa_pose(P.prior_minv, P.n_priors, q);
        const M6 W = load_soa_info(P.prior_info, P.n_priors, q);
        double r[6], Wr[6];
        M6 J;
        prior_pose3<true>(X, Pinv, r, J);
        mv6(W, r, Wr);
#pragma unroll
        for (int k = 0; k < 6; ++k) chi += r[k] * Wr[k];
        const M6 JK = m6tmul(J, m6mul(W, J));
#pragma unroll
        for (int k = 0; k < 36; ++k) D.m[k] += JK.m[k];
        mtv6_sub(J, Wr, gv);
      }
    }
  }
#pragma unroll
  for (int o = 1; o < G; o <<= 1) {
#pragma unroll
    for (int k = 0; k < 36; ++k) D.m[k] += __shfl_xor(D.m[k], o, 64);
#pragma unroll
    for (int k = 0; k < 6; ++k) gv[k] += __shfl_xor(gv[k], o, 64);
  }
  if (live && g == 0) {
    const int col = P.pose_col[v];
    if (col >= 0) {
      double *d = Hblk + 36 * (int64_t)col;
      // symmetrise exactly: J^T (W J) is symmetric up to rounding; the factor reads the lower part
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) d[r * 6 + c] = (c <= r) ? D.m[r * 6 + c] : D.m[c * 6 + r];
      double *b = bvec + 6 * (int64_t)col;
#pragma unroll
      for (int k = 0; k < 6; ++k) b[k] = gv[k];
    }
  }
  const double s = bsum4(chi, sh);
  if (threadIdx.x == 0) chi_partial[blockIdx.x] = s;
}

// shared H blocks (same vertex pair in several factors): one lane per group, serial sum
__global__ void k_dup_offdiag_gtsam(DevPlan P, const double *__restrict__ poses, double *__restrict__ Hblk) {
  const int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx >= P.n_dup_groups) return;
  M6 acc;
  for (int k = 0; k < 36; ++k) acc.m[k] = 0;
  int slot = -1;
  for (int64_t p = P.dup_ptr[gidx]; p < P.dup_ptr[gidx + 1]; ++p) {
    const int64_t e = P.dup_edges[p];
    const Pose Xi = load_pose(poses + 8 * (int64_t)P.edge_i[e]), Xj = load_pose(poses + 8 * (int64_t)P.edge_j[e]);
    const Pose Zinv = load_soa_pose(P.ainv, P.n_edges, e);
    const M6 W = load_soa_info(P.info, P.n_edges, e);
    double r[6];
    M6 Ji, Jj;
    between_pose3<true>(Xi, Xj, Zinv, r, Ji, Jj);
    const M6 O = m6tmul(Ji, m6mul(W, Jj));
    const int s = P.dup_slot[p];
    slot = s >> 1;
    for (int rr = 0; rr < 6; ++rr)
      for (int c = 0; c < 6; ++c) {
        if ((s & 1) == 0) acc.m[rr * 6 + c] += O.m[rr * 6 + c]; else acc.m[c * 6 + rr] += O.m[rr * 6 + c];
      }
  }
  if (slot >= 0)
    for (int k = 0; k < 36; ++k) Hblk[36 * (int64_t)slot + k] = acc.m[k];
}

// sum r' Omega r over all factors (CGraphGT::error is half of it: gtsam_graph.cpp:173-176)
__global__ __launch_bounds__(256) void k_chi2_gtsam(DevPlan P, const double *__restrict__ poses, double *__restrict__ chi_partial) {
  __shared__ double sh[4];
  double chi = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  M6 dummy;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < P.n_edges; e += stride) {
    const Pose Xi = load_pose(poses + 8 * (int64_t)P.edge_i[e]), Xj = load_pose(poses + 8 * (int64_t)P.edge_j[e]);
    const Pose Zinv = load_soa_pose(P.ainv, P.n_edges, e);
    const M6 W = load_soa_info(P.info, P.n_edges, e);
    double r[6], Wr[6];
    between_pose3<false>(Xi, Xj, Zinv, r, dummy, dummy);
    mv6(W, r, Wr);
    for (int k = 0; k < 6; ++k) chi += r[k] * Wr[k];
  }
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < P.n_priors; q += stride) {
    const Pose X = load_pose(poses + 8 * (int64_t)P.prior_pose[q]);
    const Pose Pinv = load_soa_pose(P.prior_minv, P.n_priors, q);
    const M6 W = load_soa_info(P.prior_info, P.n_priors, q);
    double r[6], Wr[6];
    prior_pose3<false>(X, Pinv, r, dummy);
    mv6(W, r, Wr);
    for (int k = 0; k < 6; ++k) chi += r[k] * Wr[k];
  }
  const double s = bsum4(chi, sh);
  if (threadIdx.x == 0) chi_partial[blockIdx.x] = s;
}

// Values::retract on every free pose into the candidate buffer + sum_k x_k (lambda x_k + b_k)
__global__ __launch_bounds__(256) void k_update_gtsam(DevPlan P, const double *__restrict__ poses, double *__restrict__ cand,
                                                      const double *__restrict__ x, const double *__restrict__ b,
                                                      const double *__restrict__ lambda_p, double *__restrict__ scale_partial) {
  __shared__ double sh[4];
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double sc = 0;
  if (v < P.n_poses) {
    Pose X = load_pose(poses + 8 * v);
    const int col = P.pose_col[v];
    if (col >= 0) {
      const double lambda = *lambda_p;
      double d[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        d[k] = x[6 * (int64_t)col + k];
        sc += d[k] * (lambda * d[k] + b[6 * (int64_t)col + k]);
      }
      X = retract_pose3(X, d);
    }
    store_pose(cand + 8 * v, X);
  }
  const double s = bsum4(sc, sh);
  if (threadIdx.x == 0) scale_partial[blockIdx.x] = s;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

void launch_linearize_gtsam(const DevPlan &P, const double *poses, double *Hblk, double *bvec, double *scalar_out, hipStream_t s) {
  constexpr int G = 4;
  const int blocks = cdiv(P.n_poses * G, 256);
  hipLaunchKernelGGL(k_linearize_gtsam<G>, dim3(blocks), dim3(256), 0, s, P, poses, Hblk, bvec, P.partial);
  if (P.n_dup_groups > 0)
    hipLaunchKernelGGL(k_dup_offdiag_gtsam, dim3(cdiv(P.n_dup_groups, 64)), dim3(64), 0, s, P, poses, Hblk);
  launch_reduce(P.partial, blocks, scalar_out, 0, s);
}
void launch_chi2_gtsam(const DevPlan &P, const double *poses, double *scalar_out, hipStream_t s) {
  int blocks = cdiv(P.n_edges > P.n_priors ? P.n_edges : P.n_priors, 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_chi2_gtsam, dim3(blocks), dim3(256), 0, s, P, poses, P.partial);
  launch_reduce(P.partial, blocks, scalar_out, 0, s);
}
void launch_update_gtsam(const DevPlan &P, const double *poses, double *cand, const double *x, const double *b,
                         const double *lambda_p, double *scalar_out, hipStream_t s) {
  const int blocks = cdiv(P.n_poses, 256);
  hipLaunchKernelGGL(k_update_gtsam, dim3(blocks), dim3(256), 0, s, P, poses, cand, x, b, lambda_p, P.partial);
  launch_reduce(P.partial, blocks, scalar_out, 0, s);
}

}  // namespace fgo
