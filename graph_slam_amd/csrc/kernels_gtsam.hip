// HIP kernels (gfx950, f64) for the GTSAM-semantics part of the hot path: what CGraphGT builds
// (gtsam/gtsam_graph.cpp) and LevenbergMarquardtOptimizer linearises each iteration (:1784-1788):
//   PriorFactor<Pose3 / Point3 / Vector3 / bias>   :338-341, :359-367, :379,394
//   BetweenFactor<Pose3>                           :689-692
//   OrientedPlane3Factor                           :1265
//   GenericProjectionFactor<Pose3,Point3,Cal3DS2>  :405-409
// Variables of every kind are 6-blocks (3-dof ones padded with an identity diagonal), so the block-sparse
// Cholesky and the solves (kernels.hip) are shared with the g2o path unchanged.
// Same gather-form assembly as k_linearize: every H block written once by one lane group, no FP atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_plan.hpp"
#include "factors_device.hpp"
#include "imu_device.hpp"

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
__device__ __forceinline__ M6 m6zero() {
  M6 W;
#pragma unroll
  for (int k = 0; k < 36; ++k) W.m[k] = 0;
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

// one binary factor: residual (padded to 6), Jacobians w.r.t. its first / second variable, information (padded)
template <bool WITH_JAC>
__device__ __forceinline__ void eval_factor(const DevPlan &P, int64_t e, const double *__restrict__ vals, double r[6], M6 &Ji,
                                            M6 &Jj, M6 &W) {
  const int kind = P.edge_kind[e];
  const double *vi = vals + 8 * (int64_t)P.edge_i[e], *vj = vals + 8 * (int64_t)P.edge_j[e];
  const double *__restrict__ rec = P.ainv + EDGE_REC * e;          // [0..6] measurement payload, [8..28] information
  if (kind == FK_PLANE) {
    const Pose X = load_pose(vi);
    const double4 pl = *reinterpret_cast<const double4 *>(vj);
    plane_factor<WITH_JAC>(X, V3{pl.x, pl.y, pl.z}, pl.w, V3{rec[0], rec[1], rec[2]}, rec[3], r, Ji, Jj);
    W = m6zero();
    const double w00 = rec[8], w01 = rec[9], w02 = rec[10], w11 = rec[11], w12 = rec[12], w22 = rec[13];
    W.m[0] = w00; W.m[1] = w01; W.m[2] = w02; W.m[6] = w01; W.m[7] = w11; W.m[8] = w12; W.m[12] = w02; W.m[13] = w12; W.m[14] = w22;
  } else if (kind == FK_REPROJ) {
    const Pose X = load_pose(vi);
    const double4 pt = *reinterpret_cast<const double4 *>(vj);
    reproj_factor<WITH_JAC>(X, V3{pt.x, pt.y, pt.z}, rec[0], rec[1], P.cam, r, Ji, Jj);
    W = m6zero();
    const double w = rec[8];
    W.m[0] = w; W.m[7] = w;
  } else {
    between_pose3<WITH_JAC>(load_pose(vi), load_pose(vj), load_soa_pose(rec, 1, 0), r, Ji, Jj);
    W = load_soa_info(rec + 8, 1, 0);
  }
}
// one unary prior
template <bool WITH_JAC>
__device__ __forceinline__ void eval_prior(const DevPlan &P, int64_t q, int64_t v, const double *__restrict__ vals, double r[6], M6 &J) {
  const int vk = P.var_kind[v];
  if (vk == VK_POSE) {
    prior_pose3<WITH_JAC>(load_pose(vals + 8 * v), load_soa_pose(P.prior_minv, P.n_priors, q), r, J);
  } else {
    const int dim = var_dim(vk);
    if (WITH_JAC) J = m6zero();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      r[k] = k < dim ? vals[8 * v + k] - P.prior_minv[(int64_t)k * P.n_priors + q] : 0.0;   // raw mean for vector kinds
      if (WITH_JAC && k < dim) J.m[k * 6 + k] = 1.0;
    }
  }
}
}  // namespace

// HUB = false: G lanes per variable gather its half-edges.  A variable with more than HUB_DEG half-edges (a plane or
// landmark seen from thousands of keyframes) would serialise thousands of factor evaluations on those G lanes, so it
// is skipped here and gets a whole 256-thread workgroup of the HUB = true instantiation (blockIdx -> P.hub_list).
// MASKED form (ISAM2 updates on graphs of binary factors and priors only): with `mask` only the variables flagged in it are
// gathered again -- the others keep their diagonal block, their part of b and the off-diagonal blocks their half-edges store
// from the previous call on the same buffers -- and chi2 is kept per variable in `chi_var` (summed by the caller).
template <int G, bool HUB>
__global__ __launch_bounds__(256) void k_linearize_gtsam(DevPlan P, const double *__restrict__ vals,
                                                         double *__restrict__ Hblk, double *__restrict__ bvec,
                                                         double *__restrict__ chi_partial, const unsigned char *__restrict__ mask = nullptr,
                                                         double *__restrict__ chi_var = nullptr) {
  __shared__ double sh[4];
  __shared__ double red[4][42];
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t v = HUB ? (int64_t)P.hub_list[blockIdx.x] : tid / G;
  const int hs = HUB ? P.hub_slice[blockIdx.x] : 0x10000;          // slice | slices << 16 (device_plan.hpp "Linearisation hubs")
  const int sl = hs & 0xffff, ns = hs >> 16;
  const int g = HUB ? sl * 256 + (int)threadIdx.x : (int)(tid % G);
  const int STRIDE = HUB ? 256 * ns : G;
  M6 D = m6zero();
  double gv[6] = {0, 0, 0, 0, 0, 0};
  double chi = 0;
  bool live = v < P.n_poses;
  if (!HUB && live && P.n_hubs > 0 && P.he_ptr[v + 1] - P.he_ptr[v] > P.hub_deg) live = false;
  if (live && P.ba.n_lm > 0 && P.pose_col[v] >= P.nb) live = false;      // eliminated landmark: k_ba_linearize (kernels_ba.hip)
  if (!HUB && live && mask && !mask[v]) live = false;
  if (live) {
    const int64_t p0 = P.he_ptr[v], p1 = P.he_ptr[v + 1];
    for (int64_t p = p0 + g; p < p1; p += STRIDE) {
      const int he = P.he[p];
      const int64_t e = he >> 1;
      const int side = he & 1;
      double r[6], Wr[6];
      M6 Ji, Jj, W;
      eval_factor<true>(P, e, vals, r, Ji, Jj, W);
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
    if (g == 0 && P.n_priors > 0 && P.lin_priors) {
      for (int64_t q = P.prior_ptr[v]; q < P.prior_ptr[v + 1]; ++q) {
        const M6 W = load_soa_info(P.prior_info, P.n_priors, q);
        double r[6], Wr[6];
        M6 J;
        eval_prior<true>(P, q, v, vals, r, J);
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
  for (int o = 1; o < (HUB ? 64 : G); o <<= 1) {
#pragma unroll
    for (int k = 0; k < 36; ++k) D.m[k] += __shfl_xor(D.m[k], o, 64);
#pragma unroll
    for (int k = 0; k < 6; ++k) gv[k] += __shfl_xor(gv[k], o, 64);
  }
  if (HUB) {                                   // four wave totals -> thread 0, fixed order
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int k = 0; k < 36; ++k) red[threadIdx.x >> 6][k] = D.m[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) red[threadIdx.x >> 6][36 + k] = gv[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 36; ++k) D.m[k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
#pragma unroll
      for (int k = 0; k < 6; ++k) gv[k] = ((red[0][36 + k] + red[1][36 + k]) + red[2][36 + k]) + red[3][36 + k];
    }
  }
  if (HUB && ns > 1) {                         // one slice of several: the partial sums go to k_hub_combine_gtsam
    if (threadIdx.x == 0) {
      double *o = P.hub_part + (int64_t)blockIdx.x * HUB_PART;
#pragma unroll
      for (int k = 0; k < 36; ++k) o[k] = D.m[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) o[36 + k] = gv[k];
    }
  } else if (live && (HUB ? threadIdx.x == 0 : g == 0)) {
    const int col = P.pose_col[v];
    if (col >= 0) {
      const int dim = var_dim(P.var_kind[v]);
      double *d = Hblk + 36 * (int64_t)col;
      // symmetrise exactly (J^T (W J) is symmetric up to rounding); identity on the padding of 3-dof variables
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double x = (c <= r) ? D.m[r * 6 + c] : D.m[c * 6 + r];
          if (r == c && r >= dim && (!P.var_mine || P.var_mine[v])) x += 1.0;   // (distributed: once, on the variable's rank)
          d[r * 6 + c] = x;
        }
      double *b = bvec + 6 * (int64_t)col;
#pragma unroll
      for (int k = 0; k < 6; ++k) b[k] = gv[k];
    }
  }
  if (!HUB && chi_var) {                        // (per variable: a variable that is skipped keeps its value)
#pragma unroll
    for (int o = 1; o < G; o <<= 1) chi += __shfl_xor(chi, o, 64);
    if (live && g == 0) chi_var[v] = chi;
    return;
  }
  const double s = bsum4(chi, sh);
  if (threadIdx.x == 0) chi_partial[blockIdx.x] = s;
}

// hubs linearised in several slices: the slices' partial sums in entry order, then what the single-slice path does
__global__ __launch_bounds__(64) void k_hub_combine_gtsam(DevPlan P, double *__restrict__ Hblk, double *__restrict__ bvec) {
  __shared__ double sum[HUB_PART];
  const int v = P.hubm[3 * blockIdx.x], e0 = P.hubm[3 * blockIdx.x + 1], ns = P.hubm[3 * blockIdx.x + 2];
  if (threadIdx.x < 42) {
    double a = 0;
    for (int q = 0; q < ns; ++q) a += P.hub_part[(int64_t)(e0 + q) * HUB_PART + threadIdx.x];
    sum[threadIdx.x] = a;
  }
  __syncthreads();
  const int col = P.pose_col[v];
  if (col < 0) return;
  if (threadIdx.x < 36) {
    const int r = threadIdx.x / 6, c = threadIdx.x % 6;
    const int dim = var_dim(P.var_kind[v]);
    double x = (c <= r) ? sum[r * 6 + c] : sum[c * 6 + r];
    if (r == c && r >= dim && (!P.var_mine || P.var_mine[v])) x += 1.0;
    Hblk[36 * (int64_t)col + threadIdx.x] = x;
  } else if (threadIdx.x < 42) {
    bvec[6 * (int64_t)col + (threadIdx.x - 36)] = sum[threadIdx.x];
  }
}

// shared H blocks (same variable pair in several factors): one lane per group, serial sum
__global__ void k_dup_offdiag_gtsam(DevPlan P, const double *__restrict__ vals, double *__restrict__ Hblk) {
  const int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx >= P.n_dup_groups) return;
  M6 acc = m6zero();
  int slot = -1;
  for (int64_t p = P.dup_ptr[gidx]; p < P.dup_ptr[gidx + 1]; ++p) {
    const int64_t e = P.dup_edges[p];
    double r[6];
    M6 Ji, Jj, W;
    eval_factor<true>(P, e, vals, r, Ji, Jj, W);
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
__global__ __launch_bounds__(256) void k_chi2_gtsam(DevPlan P, const double *__restrict__ vals, double *__restrict__ chi_partial) {
  __shared__ double sh[4];
  double chi = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  M6 d0, d1, W;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < P.n_edges; e += stride) {
    double r[6], Wr[6];
    eval_factor<false>(P, e, vals, r, d0, d1, W);
    mv6(W, r, Wr);
    for (int k = 0; k < 6; ++k) chi += r[k] * Wr[k];
  }
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < P.n_priors; q += stride) {
    W = load_soa_info(P.prior_info, P.n_priors, q);
    double r[6], Wr[6];
    eval_prior<false>(P, q, (int64_t)P.prior_pose[q], vals, r, d0);
    mv6(W, r, Wr);
    for (int k = 0; k < 6; ++k) chi += r[k] * Wr[k];
  }
  const double s = bsum4(chi, sh);
  if (threadIdx.x == 0) chi_partial[blockIdx.x] = s;
}

// Values::retract of one variable: out = vals (+) d for the variable kind (`active` false: plain copy)
__device__ __forceinline__ void retract_store(int vk, const double *__restrict__ vals, double *__restrict__ out, const double d[6], bool active) {
  if (vk == VK_POSE) {
    Pose X = load_pose(vals);
    if (active) X = retract_pose3(X, d);
    store_pose(out, X);
  } else {
    double4 a = *reinterpret_cast<const double4 *>(vals), c = *reinterpret_cast<const double4 *>(vals + 4);
    if (active) {
      if (vk == VK_PLANE) {
        const V3 n = unit3_retract(V3{a.x, a.y, a.z}, d[0], d[1]);
        a = make_double4(n.x, n.y, n.z, a.w + d[2]);
      } else if (vk == VK_BIAS) {
        a = make_double4(a.x + d[0], a.y + d[1], a.z + d[2], a.w + d[3]);
        c.x += d[4]; c.y += d[5];
      } else {
        a.x += d[0]; a.y += d[1]; a.z += d[2];
      }
    }
    *reinterpret_cast<double4 *>(out) = a;
    *reinterpret_cast<double4 *>(out + 4) = c;
  }
}

// Values::retract on every free variable into the candidate buffer + sum_k x_k (lambda x_k + b_k)
__global__ __launch_bounds__(256) void k_update_gtsam(DevPlan P, const double *__restrict__ vals, double *__restrict__ cand,
                                                      const double *__restrict__ x, const double *__restrict__ b,
                                                      const double *__restrict__ lambda_p, double *__restrict__ scale_partial) {
  __shared__ double sh[4];
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double sc = 0;
  if (v < P.n_poses) {
    const int col = P.pose_col[v];
    double d[6] = {0, 0, 0, 0, 0, 0};
    if (col >= 0) {
      const double lambda = *lambda_p;
      const bool mine = !P.var_mine || P.var_mine[v];      // distributed: every column counted once
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        d[k] = x[6 * (int64_t)col + k];
        if (mine) sc += d[k] * (lambda * d[k] + b[6 * (int64_t)col + k]);
      }
    }
    retract_store(P.var_kind[v], vals + 8 * v, cand + 8 * v, d, col >= 0);
  }
  const double s = bsum4(sc, sh);
  if (threadIdx.x == 0) scale_partial[blockIdx.x] = s;
}

// ---- ISAM2 semantics on the batch machinery (gtsam/gtsam_graph.cpp:1768-1776, parameters :93-99).
// ISAM2 keeps a linearisation point theta and a linear solution delta; update() moves theta only for the variables
// whose delta exceeds relinearizeThreshold (theta <- theta (+) delta, delta <- 0), relinearises the factors touching
// them and re-solves; calculateEstimate() = theta (+) delta.  Every factor is always linearised at the current theta
// of its variables, so re-linearising ALL factors at theta and solving the whole system gives what ISAM2's partial
// re-elimination gives with wildfireThreshold -> 0.
// Step 1 (before the linearisation): fluid relinearisation, one lane per variable; partial[] = number of variables moved
__global__ __launch_bounds__(256) void k_isam2_relin(DevPlan P, double *__restrict__ theta, double *__restrict__ delta, double thr,
                                                     double *__restrict__ partial, unsigned char *__restrict__ moved_out) {
  __shared__ double sh[4];
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double moved = 0;
  if (v < P.n_poses && P.pose_col[v] >= 0 && P.var_kind[v] != VK_PHANTOM) {
    double d[6], mx = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { d[k] = delta[6 * v + k]; mx = fmax(mx, fabs(d[k])); }
    if (mx >= thr) {                                  // ISAM2::Impl::CheckRelinearizationFull: any |delta_k| >= threshold
      retract_store(P.var_kind[v], theta + 8 * v, theta + 8 * v, d, true);
#pragma unroll
      for (int k = 0; k < 6; ++k) delta[6 * v + k] = 0.0;
      moved = 1;
    }
  }
  if (moved_out && v < P.n_poses) moved_out[v] = moved != 0;        // which variables moved: the partial re-factorisation starts from them
  const double s = bsum4(moved, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// Step 2 (after the solve): delta <- x (variable order), estimate = theta (+) delta
// (moved_next: what k_isam2_relin of the NEXT update will decide for this variable at threshold thr_next -- same test)
__global__ __launch_bounds__(256) void k_isam2_estimate(DevPlan P, const double *__restrict__ theta, const double *__restrict__ x,
                                                        double *__restrict__ delta, double *__restrict__ est,
                                                        unsigned char *__restrict__ moved_next, double thr_next) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= P.n_poses) return;
  const int col = P.pose_col[v];
  double d[6] = {0, 0, 0, 0, 0, 0};
  if (col >= 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { d[k] = x[6 * (int64_t)col + k]; delta[6 * v + k] = d[k]; }
  }
  if (moved_next) {
    double mx = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) mx = fmax(mx, fabs(d[k]));
    moved_next[v] = col >= 0 && P.var_kind[v] != VK_PHANTOM && mx >= thr_next;
  }
  retract_store(P.var_kind[v], theta + 8 * v, est + 8 * v, d, col >= 0);
}

// ---- CombinedImuFactor (6 variables, 15 residuals): H += J^T W J, b -= J^T W r, chi2 += r^T W r in two kernels.
// This rank's factors are listed by colour (device_plan.hpp); the factors of one colour share no variable, so k_imu_blocks --
// one launch per colour -- adds the 21 blocks of every factor straight into H: no atomics, a fixed order, no intermediate.
// (Until round 3 the blocks went through a 6 KB-per-factor scratch area and a gather kernel, one lane group per variable:
// 0.5 + 0.37 ms at cfg 4 for what is 12 KB of H traffic per factor.)  The H off-diagonal area is zeroed before the
// binary-factor kernel whenever IMU factors exist (blocks touched only by IMU factors have no storing writer).
__device__ __forceinline__ int pair_index(int u, int w) { return 5 * u - u * (u - 1) / 2 + (w - u - 1); }   // u < w

// 21 block pairs (u <= w) of the 6-variable factor, row-major upper triangle of the 6x6 block grid
__device__ __forceinline__ int pair21(int u, int w) { return 6 * u - u * (u - 1) / 2 + (w - u); }

// Step 1, one LANE per IMU factor: the residual / Jacobian algebra is a serial program of ~1 500 f64 instructions with
// ~290 live registers (one wave per SIMD) -- 64 factors per wave make it 800 waves at cfg 4, one pass over the chip: 39 us.
// r and the fifteen 3x3 Jacobian pieces (150 doubles) go to P.imu_stash.  (With one factor per wave, all lanes computing
// the same, the issue slots of the algebra alone cost 0.12 ms and dragged the dense part down to one wave per SIMD.)
constexpr int IMU_NS = 9 * IMU_NPIECE + 15;
__global__ __launch_bounds__(64) void k_imu_eval(DevPlan P, const double *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= P.imu_fn) return;
  const int64_t f = P.imu_list[i];
  const int *ids = P.imu_ids + 6 * f;
  const double *pv[6];
#pragma unroll
  for (int u = 0; u < 6; ++u) pv[u] = vals + 8 * (int64_t)ids[u];
  double r[15];
  double *J = P.imu_stash + (size_t)f * IMU_NS;
  imu_factor<true>(P.imu[f], pv, P.gravity, r, J);
#pragma unroll
  for (int a = 0; a < 15; ++a) J[9 * IMU_NPIECE + a] = r[a];
}

// Step 2, one wave per IMU_PER_WAVE factors of ONE colour.  First the targets of all its factors at once (two dependent
// index round trips per WAVE, not per factor): per factor the 21 pair blocks (H block << 1 | transposed, or -1) and the six
// gradient columns.  Then factor by factor: the 150 doubles of step 1 are expanded into the image F = [J | r] (15 x 37, zero
// padded to 16 x 48) in LDS and the dense part runs on v_mfma_f64_16x16x4_f64:  V = W F  (1 x 3 tiles; the A operand W
// straight from the payload, V stays in registers: the result layout is a legal B operand when the K index of step k is
// lq + 4 k), then S = F^T V = [[J^T W J, J^T W r], [., r^T W r]] on the six tiles on and above the tile diagonal: 36 MFMAs and
// 24 LDS operand reads per lane.  S goes back to LDS (over F) and the wave walks the factor's 21 target blocks element by
// element -- 64 consecutive doubles of H per step, whatever the orientation of a block -- adding S's entries: block (u, w),
// u < w, from S's rows 6 u .. and columns 6 w ..; a diagonal block symmetric from S's upper triangle; the gradient from
// column 36; chi2 is S[36][36].  The H values are requested before the dense part and the next factor's operands before
// the current one is worked on, so a factor costs its LDS / MFMA work and not five dependent round trips.
constexpr int IMU_PER_WAVE = 8;
typedef double imu_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_imu_blocks(DevPlan P, double *__restrict__ Hblk, double *__restrict__ bvec,
                                                   double *__restrict__ chi_partial, int l0, int n) {
  constexpr int JS = 48, NIT = (21 * 36 + 63) / 64;
  __shared__ __attribute__((aligned(16))) double S[37 * JS];       // rows 0 .. 15 hold F first
  __shared__ int tgt[IMU_PER_WAVE][32], flist[IMU_PER_WAVE];
  const int lane = threadIdx.x, ln = lane & 15, lq = lane >> 4;
  const int i0 = (int)blockIdx.x * IMU_PER_WAVE;
  const int nf = n - i0 < IMU_PER_WAVE ? n - i0 : IMU_PER_WAVE;
  if (lane < nf) flist[lane] = P.imu_list[l0 + i0 + lane];
#pragma unroll
  for (int x4 = 0; x4 < IMU_PER_WAVE * 32 / 64; ++x4) {
    const int x = lane + 64 * x4, t = x >> 5, sl = x & 31;
    int code = -1;
    if (t < nf && sl < 27) {
      const int64_t f = P.imu_list[l0 + i0 + t];
      const int *ids = P.imu_ids + 6 * f;
      if (sl < 21) {
        const int u = (sl >= 6) + (sl >= 11) + (sl >= 15) + (sl >= 18) + (sl >= 20), w = u + (sl - pair21(u, u));
        if (u == w) { const int col = P.pose_col[ids[u]]; code = col >= 0 ? 2 * col : -1; }
        else code = P.imu_slot[15 * f + pair_index(u, w)];
      } else code = P.pose_col[ids[sl - 21]];
    }
    tgt[t][sl] = code;
  }
  // where entries lane, lane + 64, lane + 128 of a factor's list go in F
  int pos[3];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    const int k = lane + 64 * s3;
    pos[s3] = k < 9 * IMU_NPIECE ? imu_piece_pos(k, JS) : (k < IMU_NS ? (k - 9 * IMU_NPIECE) * JS + 36 : -1);
  }
  // this lane's elements of the 21 blocks: e = lane + 64 it -> pair e / 36, entry e % 36
  __builtin_amdgcn_wave_barrier();
  // A operand of V = W F: W[ln][4 q + lq] (row / column 15: zero); the factor's list
  auto load_ops = [&](int t, double (&wa)[4], double (&sv)[3]) {
    const int64_t f = flist[t];
    const ImuPayload &m = P.imu[f];
    const double *st = P.imu_stash + (size_t)f * IMU_NS;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int b = 4 * q + lq; wa[q] = (ln < 15 && b < 15) ? m.info[ln * 15 + b] : 0.0; }
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) sv[s3] = pos[s3] >= 0 ? st[lane + 64 * s3] : 0.0;
  };
  double wa[4], wn[4], sv[3], sn[3];
  load_ops(0, wa, sv);
  double chi = 0;
  for (int t = 0; t < nf; ++t) {
    // what H and b hold now
    double h[NIT], gb = 0;
    int hc[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = lane + 64 * it;
      hc[it] = e < 21 * 36 ? tgt[t][e / 36] : -1;
      h[it] = hc[it] >= 0 ? Hblk[36 * (int64_t)(hc[it] >> 1) + (e - 36 * (e / 36))] : 0.0;
    }
    const int gcol = lane < 36 ? tgt[t][21 + lane / 6] : -1;
    if (gcol >= 0) gb = bvec[6 * (int64_t)gcol + (lane - 6 * (lane / 6))];
    if (t + 1 < nf) load_ops(t + 1, wn, sn);
#pragma unroll
    for (int k = 0; k < 16 * JS / 64; ++k) S[k * 64 + lane] = 0.0;
    __builtin_amdgcn_wave_barrier();
    imu_const_entries(S, JS, lane);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
      if (pos[s3] >= 0) S[pos[s3]] = sv[s3];
    __builtin_amdgcn_wave_barrier();
    // fb: F[4 q + lq][16 T + ln], the B operand of V = W F;  fa: F[lq + 4 k][16 I + ln], the A operand (= F^T) of F^T V
    double fb[3][4], fa[3][4];
#pragma unroll
    for (int T = 0; T < 3; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) { fb[T][q] = S[(4 * q + lq) * JS + 16 * T + ln]; fa[T][q] = S[(lq + 4 * q) * JS + 16 * T + ln]; }
    __builtin_amdgcn_wave_barrier();
    imu_d4 vb[3];
#pragma unroll
    for (int T = 0; T < 3; ++T) {
      imu_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[q], fb[T][q], acc, 0, 0, 0);
      vb[T] = acc;                                                                   // V[lq + 4 k][16 T + ln], k = 0 .. 3
    }
#pragma unroll
    for (int I = 0; I < 3; ++I)
#pragma unroll
      for (int T = I; T < 3; ++T) {
        imu_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[I][k], vb[T][k], acc, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                               // result layout: row lq + 4 k, column ln
          const int i = 16 * I + lq + 4 * k;
          if (i < 37) S[i * JS + 16 * T + ln] = acc[k];
        }
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (hc[it] < 0) continue;
      const int e = lane + 64 * it, pr = e / 36, k = e - 36 * pr;
      const int u = (pr >= 6) + (pr >= 11) + (pr >= 15) + (pr >= 18) + (pr >= 20), w = u + (pr - pair21(u, u));
      const int k6 = k / 6, km = k - 6 * k6;
      int i = 6 * u + ((hc[it] & 1) ? km : k6), j = 6 * w + ((hc[it] & 1) ? k6 : km);
      if (u == w && i > j) { const int x = i; i = j; j = x; }
      Hblk[36 * (int64_t)(hc[it] >> 1) + k] = h[it] + S[i * JS + j];
    }
    if (gcol >= 0) bvec[6 * (int64_t)gcol + (lane - 6 * (lane / 6))] = gb - S[lane * JS + 36];
    if (lane == 0) chi += S[36 * JS + 36];
#pragma unroll
    for (int q = 0; q < 4; ++q) wa[q] = wn[q];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) sv[s3] = sn[s3];
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) chi_partial[blockIdx.x] = chi;
}

__global__ __launch_bounds__(64) void k_chi2_imu(DevPlan P, const double *__restrict__ vals, double *__restrict__ chi_partial) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double chi = 0;
  if (f < P.n_imu) {
    const int *ids = P.imu_ids + 6 * f;
    const double *pv[6];
    for (int u = 0; u < 6; ++u) pv[u] = vals + 8 * (int64_t)ids[u];
    const ImuPayload &m = P.imu[f];
    double r[15];
    imu_factor<false>(m, pv, P.gravity, r, nullptr);
    for (int a = 0; a < 15; ++a)
      for (int b = 0; b < 15; ++b) chi += r[a] * m.info[a * 15 + b] * r[b];
  }
  chi = wsum(chi);
  if (threadIdx.x == 0) chi_partial[blockIdx.x] = chi;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// sum of n doubles, stage 1: a workgroup per 4 096 entries (fixed order) -> partial[blockIdx]
__global__ __launch_bounds__(256) void k_sum_chunks(const double *__restrict__ in, int64_t n, double *__restrict__ partial) {
  __shared__ double sh[4];
  double acc = 0;
  const int64_t i0 = (int64_t)blockIdx.x * 4096;
#pragma unroll 4
  for (int q = 0; q < 16; ++q) { const int64_t i = i0 + q * 256 + threadIdx.x; if (i < n) acc += in[i]; }
  const double s = bsum4(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
bool linearize_gtsam_maskable(const DevPlan &P) {
  return P.n_imu == 0 && P.n_hubs == 0 && P.n_hub_multi == 0 && P.n_dup_groups == 0 && P.ba.n_lm == 0 && !P.zero_offdiag && !P.var_mine;
}
void launch_linearize_gtsam_masked(const DevPlan &P, const double *poses, double *Hblk, double *bvec, double *scalar_out, hipStream_t s,
                                   const unsigned char *mask, double *chi_var) {
  constexpr int G = 4;
  hipLaunchKernelGGL((k_linearize_gtsam<G, false>), dim3(cdiv(P.n_poses * G, 256)), dim3(256), 0, s, P, poses, Hblk, bvec, P.partial, mask, chi_var);
  const int chunks = cdiv(P.n_poses, 4096);
  hipLaunchKernelGGL(k_sum_chunks, dim3(chunks), dim3(256), 0, s, chi_var, (int64_t)P.n_poses, P.partial);
  launch_reduce(P.partial, chunks, scalar_out, 0, s);
}
void launch_linearize_gtsam(const DevPlan &P, const double *poses, double *Hblk, double *bvec, double *scalar_out, hipStream_t s,
                            double *ba_W, double *ba_Hpp, double *ba_bp) {
  constexpr int G = 4;
  int blocks = cdiv(P.n_poses * G, 256);
  if (P.n_imu > 0 || P.zero_offdiag)   // blocks without a storing writer (IMU-only pairs, other ranks' edges) must start from zero
    launch_zero(Hblk + 36 * (int64_t)P.nb, 36 * (int64_t)(P.n_hblocks - P.nb), s);
  hipLaunchKernelGGL((k_linearize_gtsam<G, false>), dim3(blocks), dim3(256), 0, s, P, poses, Hblk, bvec, P.partial);
  if (P.n_hubs > 0)
    hipLaunchKernelGGL((k_linearize_gtsam<G, true>), dim3(P.n_hubs), dim3(256), 0, s, P, poses, Hblk, bvec, P.partial + blocks);
  if (P.n_hub_multi > 0) hipLaunchKernelGGL(k_hub_combine_gtsam, dim3(P.n_hub_multi), dim3(64), 0, s, P, Hblk, bvec);
  blocks += P.n_hubs;
  if (P.n_dup_groups > 0)
    hipLaunchKernelGGL(k_dup_offdiag_gtsam, dim3(cdiv(P.n_dup_groups, 64)), dim3(64), 0, s, P, poses, Hblk);
  int total = blocks;
  if (P.n_imu > 0) {
    if (P.imu_fn > 0) hipLaunchKernelGGL(k_imu_eval, dim3((unsigned)cdiv(P.imu_fn, 64)), dim3(64), 0, s, P, poses);
    for (int col = 0; col < P.imu_ncolor; ++col) {
      const int l0 = P.imu_color_ptr_h[col], n = P.imu_color_ptr_h[col + 1] - l0;
      if (n <= 0) continue;
      const int iw = cdiv(n, IMU_PER_WAVE);
      hipLaunchKernelGGL(k_imu_blocks, dim3((unsigned)iw), dim3(64), 0, s, P, Hblk, bvec, P.partial + total, l0, n);
      total += iw;
    }
  }
  if (P.ba.n_lm > 0) {                      // the landmark side of the eliminated observations (and their chi2)
    launch_ba_linearize(P, poses, ba_W, ba_Hpp, ba_bp, Hblk, bvec, P.partial + total, s);
    total += ba_linearize_blocks(P);
  }
  launch_reduce(P.partial, total, scalar_out, 0, s);
}
void launch_chi2_gtsam(const DevPlan &P, const double *poses, double *scalar_out, hipStream_t s) {
  int blocks = cdiv(P.n_edges > P.n_priors ? P.n_edges : P.n_priors, 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_chi2_gtsam, dim3(blocks), dim3(256), 0, s, P, poses, P.partial);
  int total = blocks;
  if (P.n_imu > 0) {
    const int ib = cdiv(P.n_imu, 64);
    hipLaunchKernelGGL(k_chi2_imu, dim3(ib), dim3(64), 0, s, P, poses, P.partial + blocks);
    total += ib;
  }
  launch_reduce(P.partial, total, scalar_out, 0, s);
}
void launch_update_gtsam(const DevPlan &P, const double *poses, double *cand, const double *x, const double *b,
                         const double *lambda_p, double *scalar_out, hipStream_t s) {
  const int blocks = cdiv(P.n_poses, 256);
  hipLaunchKernelGGL(k_update_gtsam, dim3(blocks), dim3(256), 0, s, P, poses, cand, x, b, lambda_p, P.partial);
  launch_reduce(P.partial, blocks, scalar_out, 0, s);
}

void launch_isam2_relin(const DevPlan &P, double *theta, double *delta, double thr, double *count_out, hipStream_t s, unsigned char *moved_out) {
  const int blocks = cdiv(P.n_poses, 256);
  hipLaunchKernelGGL(k_isam2_relin, dim3(blocks), dim3(256), 0, s, P, theta, delta, thr, P.partial, moved_out);
  launch_reduce(P.partial, blocks, count_out, 0, s);
}
void launch_isam2_estimate(const DevPlan &P, const double *theta, const double *x, double *delta, double *est, hipStream_t s,
                           unsigned char *moved_next, double thr_next) {
  hipLaunchKernelGGL(k_isam2_estimate, dim3(cdiv(P.n_poses, 256)), dim3(256), 0, s, P, theta, x, delta, est, moved_next, thr_next);
}

}  // namespace fgo
