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
template <int G, bool HUB>
__global__ __launch_bounds__(256) void k_linearize_gtsam(DevPlan P, const double *__restrict__ vals,
                                                         double *__restrict__ Hblk, double *__restrict__ bvec,
                                                         double *__restrict__ chi_partial) {
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
__global__ __launch_bounds__(256) void k_isam2_estimate(DevPlan P, const double *__restrict__ theta, const double *__restrict__ x,
                                                        double *__restrict__ delta, double *__restrict__ est) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= P.n_poses) return;
  const int col = P.pose_col[v];
  double d[6] = {0, 0, 0, 0, 0, 0};
  if (col >= 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { d[k] = x[6 * (int64_t)col + k]; delta[6 * v + k] = d[k]; }
  }
  retract_store(P.var_kind[v], theta + 8 * v, est + 8 * v, d, col >= 0);
}

// ---- CombinedImuFactor (6 variables, 15 residuals).  One lane per VARIABLE walks its IMU incidences; it adds
// J_p^T W J_p / -J_p^T W r to its own diagonal block / rhs (written before by k_linearize_gtsam) and owns every
// off-diagonal block it shares with a variable of SMALLER index, so all contributions to a block come from one lane in
// a fixed order: deterministic without atomics.  The H off-diagonal area is zeroed before the binary-factor kernel
// whenever IMU factors exist (blocks touched only by IMU factors have no other writer).
__device__ __forceinline__ int pair_index(int u, int w) { return 5 * u - u * (u - 1) / 2 + (w - u - 1); }   // u < w

// 21 block pairs (u <= w) of the 6-variable factor, row-major upper triangle of the 6x6 block grid
__device__ __forceinline__ int pair21(int u, int w) { return 6 * u - u * (u - 1) / 2 + (w - u); }

// Step 1, one WAVE per IMU factor: the residual / Jacobian algebra is evaluated by every lane in registers (3x3
// pieces only), the 15x36 Jacobian goes to LDS, and the dense part -- W J, then the 21 blocks J_u^T W J_w and the six
// gradient pieces -J_u^T W r -- is spread over the lanes.  Output: P.imu_blk[f][21][36], P.imu_g[f][36], chi2 of the
// factor.  (The former one-lane-per-variable kernel kept a 540-double Jacobian per lane in scratch memory and
// evaluated every factor six times: 25 ms at cfg 4.)
__global__ __launch_bounds__(64) void k_imu_blocks(DevPlan P, const double *__restrict__ vals, double *__restrict__ chi_partial) {
  __shared__ __attribute__((aligned(16))) double J[6][90];
  __shared__ __attribute__((aligned(16))) double W[225], WJ[6][90], Wr[16], rs[16];
  const int64_t f = P.imu_list ? (int64_t)P.imu_list[blockIdx.x] : P.imu_f0 + blockIdx.x;   // this rank's factors
  const int lane = threadIdx.x;
  const ImuPayload &m = P.imu[f];
  for (int k = lane; k < 540; k += 64) (&J[0][0])[k] = 0.0;
  for (int k = lane; k < 225; k += 64) W[k] = m.info[k];
  __builtin_amdgcn_wave_barrier();
  const int *ids = P.imu_ids + 6 * f;
  const double *pv[6];
#pragma unroll
  for (int u = 0; u < 6; ++u) pv[u] = vals + 8 * (int64_t)ids[u];
  double r[15];
  imu_factor<true, true>(m, pv, P.gravity, r, J, lane == 0);
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 15; ++a) rs[a] = r[a];
  }
  __builtin_amdgcn_wave_barrier();
  // W J (15 x 36) and W r
  for (int o = lane; o < 555; o += 64) {
    if (o < 540) {
      const int u = o / 90, rem = o - 90 * u, a = rem / 6, c = rem - 6 * a;
      double t = 0;
#pragma unroll
      for (int b = 0; b < 15; ++b) t += W[a * 15 + b] * J[u][b * 6 + c];
      WJ[u][a * 6 + c] = t;
    } else {
      const int a = o - 540;
      double t = 0;
#pragma unroll
      for (int b = 0; b < 15; ++b) t += W[a * 15 + b] * rs[b];
      Wr[a] = t;
    }
  }
  __builtin_amdgcn_wave_barrier();
  double *__restrict__ ob = P.imu_blk + (size_t)f * (21 * 36);
  double *__restrict__ og = P.imu_g + (size_t)f * 36;
  for (int o = lane; o < 21 * 36 + 36; o += 64) {
    if (o < 21 * 36) {
      const int pr = o / 36, e = o - 36 * pr, rr = e / 6, c = e - 6 * rr;
      const int u = (pr >= 6) + (pr >= 11) + (pr >= 15) + (pr >= 18) + (pr >= 20);   // invert pair21
      const int w = u + (pr - pair21(u, u));
      double t = 0;
#pragma unroll
      for (int a = 0; a < 15; ++a) t += J[u][a * 6 + rr] * WJ[w][a * 6 + c];
      ob[o] = t;
    } else {
      const int e = o - 21 * 36, u = e / 6, rr = e - 6 * u;
      double t = 0;
#pragma unroll
      for (int a = 0; a < 15; ++a) t += J[u][a * 6 + rr] * Wr[a];
      og[e] = -t;
    }
  }
  if (lane == 0) {
    double chi = 0;
#pragma unroll
    for (int a = 0; a < 15; ++a) chi += rs[a] * Wr[a];
    chi_partial[blockIdx.x] = chi;
  }
}

// Step 2, one lane per variable: gather the blocks of its (at most two) IMU factors in a fixed order.  The diagonal
// block and the gradient always; an off-diagonal pair block is owned by the variable with the larger index.
__global__ __launch_bounds__(64) void k_imu_gather(DevPlan P, double *__restrict__ Hblk, double *__restrict__ bvec) {
  // six lanes per variable, lane r owns row r of every 6x6 block it touches (10 variables per wave): the 288-byte
  // blocks are read and written as contiguous 48-byte rows instead of 36 scalar accesses per lane
  __shared__ double tile[10][36];
  const int lane = threadIdx.x, g = lane / 6, r = lane - 6 * g;
  const int64_t v = (int64_t)blockIdx.x * 10 + g;
  const bool live = lane < 60 && v < P.n_poses;
  const int64_t q0 = live ? P.imu_inc_ptr[v] : 0, q1 = live ? P.imu_inc_ptr[v + 1] : 0;
  double D[6] = {0, 0, 0, 0, 0, 0}, gv = 0;
  for (int64_t q = q0; q < q1; ++q) {
    const int f = P.imu_inc[q] >> 3, pos = P.imu_inc[q] & 7;
    const int *ids = P.imu_ids + 6 * (int64_t)f;
    const double *__restrict__ blk = P.imu_blk + (size_t)f * (21 * 36);
    const double *__restrict__ d = blk + 36 * pair21(pos, pos) + 6 * r;
#pragma unroll
    for (int c = 0; c < 6; ++c) D[c] += d[c];
    gv += P.imu_g[(size_t)f * 36 + 6 * pos + r];
    for (int u = 0; u < 6; ++u) {
      if (u == pos || ids[u] >= ids[pos]) continue;
      const int lo = u < pos ? u : pos, hi = u < pos ? pos : u;
      const int slot = P.imu_slot[15 * (int64_t)f + pair_index(lo, hi)];
      if (slot < 0) continue;
      double *o = Hblk + 36 * (int64_t)(slot >> 1);
      const double *__restrict__ O = blk + 36 * pair21(lo, hi) + 6 * r;   // row r of J_lo^T W J_hi
      if ((slot & 1) == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) o[6 * r + c] += O[c];
      } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) o[c * 6 + r] += O[c];
      }
    }
  }
  // symmetrise exactly like the former one-lane version: the lower triangle is mirrored
  if (lane < 60) {
#pragma unroll
    for (int c = 0; c < 6; ++c) tile[g][6 * r + c] = D[c];
  }
  __builtin_amdgcn_wave_barrier();
  const int col = live ? P.pose_col[v] : -1;
  if (col >= 0 && q1 > q0) {
    double *d = Hblk + 36 * (int64_t)col + 6 * r;
#pragma unroll
    for (int c = 0; c < 6; ++c) d[c] += (c <= r) ? D[c] : tile[g][6 * c + r];
    bvec[6 * (int64_t)col + r] += gv;
  }
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
    if (P.imu_fn > 0) hipLaunchKernelGGL(k_imu_blocks, dim3((unsigned)P.imu_fn), dim3(64), 0, s, P, poses, P.partial + blocks);
    hipLaunchKernelGGL(k_imu_gather, dim3(cdiv(P.n_poses, 10)), dim3(64), 0, s, P, Hblk, bvec);
    total += (int)P.imu_fn;
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
void launch_isam2_estimate(const DevPlan &P, const double *theta, const double *x, double *delta, double *est, hipStream_t s) {
  hipLaunchKernelGGL(k_isam2_estimate, dim3(cdiv(P.n_poses, 256)), dim3(256), 0, s, P, theta, x, delta, est);
}

}  // namespace fgo
