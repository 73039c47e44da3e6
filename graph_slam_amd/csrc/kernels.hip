// Hand-written HIP kernels for gfx950 (MI355X, wave64) — the optimiser hot path:
//   k_linearize      EdgeSE3::computeError + linearizeOplus + constructQuadraticForm   (HOT LOOP 1+2)
//   k_chol_leaf      light sub-trees of the elimination tree factored (and forward-solved) inside LDS   (HOT LOOP 3)
//   k_chol_acc       external updates of a level's columns, gather form (block-sparse left-looking Cholesky,
//                    6x6 f64 micro-blocks); k_chol_fact: the generic one-workgroup-per-task level kernel
//   k_panel_tri/rows dense part of a panel (<= 16 columns of the skinny top of the tree) on f64 MFMA tiles
//   k_solve_fwd/bwd, k_fwd_*/k_bwd_*   level-scheduled block triangular solves (generic levels / panels)
//   k_update         VertexSE3::oplusImpl over all vertices + the LM 'scale' reduction
//   k_chi2           computeActiveErrors + chi2
// replacing what the reference reaches through mp_optimizer->optimize() (g2o/g2o_graph.cpp:246-249)
// and computeActiveErrors()/chi2() (g2o/g2o_graph.cpp:256-257).
//
// Determinism: no floating-point atomics anywhere.  H and b are produced in "gather" form (every
// output block is written once by one lane group that sums its incident half-edges in a fixed order);
// Cholesky targets are owned by one wave; reductions are two-pass with a fixed tree.
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>
#include <stdint.h>
#include "device_plan.hpp"
#include <cstdlib>
#include "fgo_internal.hpp"
#include "se3_device.hpp"

namespace fgo {
using namespace dev;

#define WAVE 64
constexpr int PM = PANEL_MAX;
constexpr int NJMAX = (6 * PM + 15) / 16;              // 16-wide tile rows of a panel's dense scalar triangle (6 / 12)
typedef double d4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
  return v;
}

// block-level deterministic sum: wave shuffles, then LDS across waves (fixed order)
template <int NW>
__device__ __forceinline__ double block_sum(double v, double *sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NW; ++i) s += sh[i];
  }
  __syncthreads();
  return s;   // valid on thread 0
}

// edge record (device_plan.hpp: EDGE_REC doubles, 256-byte aligned): 16-byte loads
__device__ __forceinline__ Pose load_ainv(const double *__restrict__ ainv, int64_t, int64_t e) {
  const double2 *__restrict__ r = reinterpret_cast<const double2 *>(ainv + EDGE_REC * e);
  const double2 a = r[0], b = r[1], c = r[2], d = r[3];
  Pose A;
  A.t = {a.x, a.y, b.x};
  A.q = {b.y, c.x, c.y, d.x};
  return A;
}

struct Info3 { M3 tt, tq, qq; };   // Omega = [[tt, tq], [tq^T, qq]]
__device__ __forceinline__ Info3 load_info(const double *__restrict__ info, int64_t, int64_t e) {
  double u[22];
  const double2 *__restrict__ r = reinterpret_cast<const double2 *>(info + EDGE_REC * e);
#pragma unroll
  for (int c = 0; c < 11; ++c) { const double2 v = r[c]; u[2 * c] = v.x; u[2 * c + 1] = v.y; }
  // upper-triangular row-major: row0: 0..5, row1: 6..10, row2: 11..14, row3: 15..17, row4: 18..19, row5: 20
  Info3 W;
  W.tt = {{u[0], u[1], u[2], u[1], u[6], u[7], u[2], u[7], u[11]}};
  W.tq = {{u[3], u[4], u[5], u[8], u[9], u[10], u[12], u[13], u[14]}};
  W.qq = {{u[15], u[16], u[17], u[16], u[18], u[19], u[17], u[19], u[20]}};
  return W;
}

// ------------------------------------------------------------------------------------------------
// Linearise + assemble, gather form.  G lanes cooperate on one pose: its incident half-edges are dealt
// round-robin to the G lanes, each lane evaluates residual + Jacobians of its half-edges and keeps
// J_s^T Omega J_s and -J_s^T Omega e in registers; a fixed xor-tree over the G lanes finishes the sum.
// The j-side half-edge of an edge also owns its off-diagonal block J_i^T Omega J_j and its chi2 term.
// HBM traffic per launch: one 256-byte edge record (two whole lines) per half-edge, poses gathered (64 B each),
// H diag/off-diag + b written once.
// HUB = true: a pose with more than HUB_DEG half-edges (skipped by the HUB = false launch) gets a whole 256-thread
// workgroup (blockIdx -> P.hub_list) instead of G lanes; the partial sums go through a fixed shuffle tree + LDS.
template <int G, bool HUB>
__global__ __launch_bounds__(256) void k_linearize(DevPlan P, const double *__restrict__ poses,
                                                   double *__restrict__ Hblk, double *__restrict__ bvec,
                                                   double *__restrict__ chi_partial) {
  __shared__ double sh[4];
  __shared__ double red[4][33];
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t v = HUB ? (int64_t)P.hub_list[blockIdx.x] : tid / G;
  const int hs = HUB ? P.hub_slice[blockIdx.x] : 0x10000;          // slice | slices << 16
  const int sl = hs & 0xffff, ns = hs >> 16;
  const int g = HUB ? sl * 256 + (int)threadIdx.x : (int)(tid % G);
  const int STRIDE = HUB ? 256 * ns : G;
  M3 Dtt = mzero(), Dtq = mzero(), Dqq = mzero();
  double gt[3] = {0, 0, 0}, gq[3] = {0, 0, 0};
  double chi = 0;
  bool live = v < P.n_poses;
  if (!HUB && live && P.n_hubs > 0 && P.he_ptr[v + 1] - P.he_ptr[v] > P.hub_deg) live = false;
  if (live) {
    const int64_t p0 = P.he_ptr[v], p1 = P.he_ptr[v + 1];
    for (int64_t p = p0 + g; p < p1; p += STRIDE) {
      const int he = P.he[p];
      const int64_t e = he >> 1;
      const int side = he & 1;                 // 1: this pose is vertex j of the edge
      const int vi = P.edge_i[e], vj = P.edge_j[e];
      const Pose Xi = load_pose(poses + 8 * (int64_t)vi), Xj = load_pose(poses + 8 * (int64_t)vj);
      const Pose A = load_ainv(P.ainv, P.edge_stride, e);
      const Info3 W = load_info(P.info, P.edge_stride, e);
      EdgeLin L;
      edge_se3<true>(Xi, Xj, A, L);
      const V3 et = {L.e[0], L.e[1], L.e[2]}, eq = {L.e[3], L.e[4], L.e[5]};
      const V3 Wt = mv(W.tt, et) + mv(W.tq, eq);          // (Omega e)_t
      const V3 Wq = mtv(W.tq, et) + mv(W.qq, eq);         // (Omega e)_q
      if (side) {
        chi += et.x * Wt.x + et.y * Wt.y + et.z * Wt.z + eq.x * Wq.x + eq.y * Wq.y + eq.z * Wq.z;
        // X = Omega Jj, Jj = [[Aj, 0], [0, Cj]]
        const M3 X11 = mm(W.tt, L.Aj), X21 = mtm(W.tq, L.Aj), X12 = mm(W.tq, L.Cj), X22 = mm(W.qq, L.Cj);
        Dtt = madd(Dtt, mtm(L.Aj, X11));
        Dtq = madd(Dtq, mtm(L.Aj, X12));
        Dqq = madd(Dqq, mtm(L.Cj, X22));
        const V3 a = mtv(L.Aj, Wt), c = mtv(L.Cj, Wq);
        gt[0] -= a.x; gt[1] -= a.y; gt[2] -= a.z;
        gq[0] -= c.x; gq[1] -= c.y; gq[2] -= c.z;
        const int slot = P.edge_slot[e];
        if (slot >= 0) {
          // O = Ji^T X  (rows: tangent of i, cols: tangent of j)
          const M3 Ott = mtm(L.Ai, X11), Otq = mtm(L.Ai, X12);
          const M3 Oqt = madd(mtm(L.Bi, X11), mtm(L.Ci, X21)), Oqq = madd(mtm(L.Bi, X12), mtm(L.Ci, X22));
          // the block in registers first, then 18 sixteen-byte stores in address order (whole lines leave the L2 once)
          double blk[36];
          if ((slot & 1) == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int c2 = 0; c2 < 3; ++c2) {
                blk[r * 6 + c2] = Ott.m[r * 3 + c2]; blk[r * 6 + 3 + c2] = Otq.m[r * 3 + c2];
                blk[(3 + r) * 6 + c2] = Oqt.m[r * 3 + c2]; blk[(3 + r) * 6 + 3 + c2] = Oqq.m[r * 3 + c2];
              }
          } else {   // store O^T (block row = j)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int c2 = 0; c2 < 3; ++c2) {
                blk[c2 * 6 + r] = Ott.m[r * 3 + c2]; blk[(3 + c2) * 6 + r] = Otq.m[r * 3 + c2];
                blk[c2 * 6 + 3 + r] = Oqt.m[r * 3 + c2]; blk[(3 + c2) * 6 + 3 + r] = Oqq.m[r * 3 + c2];
              }
          }
          double2 *__restrict__ o2 = reinterpret_cast<double2 *>(Hblk + 36 * (int64_t)(slot >> 1));
#pragma unroll
          for (int k = 0; k < 18; ++k) o2[k] = make_double2(blk[2 * k], blk[2 * k + 1]);
        }
      } else {
        // X = Omega Ji, Ji = [[Ai, Bi], [0, Ci]]
        const M3 X11 = mm(W.tt, L.Ai);
        const M3 X12 = madd(mm(W.tt, L.Bi), mm(W.tq, L.Ci));
        const M3 X22 = madd(mtm(W.tq, L.Bi), mm(W.qq, L.Ci));
        Dtt = madd(Dtt, mtm(L.Ai, X11));
        Dtq = madd(Dtq, mtm(L.Ai, X12));
        Dqq = madd(Dqq, madd(mtm(L.Bi, X12), mtm(L.Ci, X22)));
        const V3 a = mtv(L.Ai, Wt), c = mtv(L.Bi, Wt) + mtv(L.Ci, Wq);
        gt[0] -= a.x; gt[1] -= a.y; gt[2] -= a.z;
        gq[0] -= c.x; gq[1] -= c.y; gq[2] -= c.z;
      }
    }
  }
  // fixed xor tree over the G lanes of a pose (HUB: over the wave, then the four wave totals in a fixed order)
#pragma unroll
  for (int o = 1; o < (HUB ? 64 : G); o <<= 1) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      Dtt.m[k] += __shfl_xor(Dtt.m[k], o, WAVE);
      Dtq.m[k] += __shfl_xor(Dtq.m[k], o, WAVE);
      Dqq.m[k] += __shfl_xor(Dqq.m[k], o, WAVE);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { gt[k] += __shfl_xor(gt[k], o, WAVE); gq[k] += __shfl_xor(gq[k], o, WAVE); }
  }
  if (HUB) {
    if ((threadIdx.x & 63) == 0) {
      double *rw = red[threadIdx.x >> 6];
#pragma unroll
      for (int k = 0; k < 9; ++k) { rw[k] = Dtt.m[k]; rw[9 + k] = Dtq.m[k]; rw[18 + k] = Dqq.m[k]; }
#pragma unroll
      for (int k = 0; k < 3; ++k) { rw[27 + k] = gt[k]; rw[30 + k] = gq[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        Dtt.m[k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
        Dtq.m[k] = ((red[0][9 + k] + red[1][9 + k]) + red[2][9 + k]) + red[3][9 + k];
        Dqq.m[k] = ((red[0][18 + k] + red[1][18 + k]) + red[2][18 + k]) + red[3][18 + k];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gt[k] = ((red[0][27 + k] + red[1][27 + k]) + red[2][27 + k]) + red[3][27 + k];
        gq[k] = ((red[0][30 + k] + red[1][30 + k]) + red[2][30 + k]) + red[3][30 + k];
      }
    }
  }
  if (HUB && ns > 1) {                         // one slice of several: the partial sums go to k_hub_combine
    if (threadIdx.x == 0) {
      double *o = P.hub_part + (int64_t)blockIdx.x * HUB_PART;
#pragma unroll
      for (int k = 0; k < 9; ++k) { o[k] = Dtt.m[k]; o[9 + k] = Dtq.m[k]; o[18 + k] = Dqq.m[k]; }
#pragma unroll
      for (int k = 0; k < 3; ++k) { o[27 + k] = gt[k]; o[30 + k] = gq[k]; }
    }
  } else if (live && (HUB ? threadIdx.x == 0 : g == 0)) {
    const int col = P.pose_col[v];
    if (col >= 0) {
      double *d = Hblk + 36 * (int64_t)col;
      const double slot_diag = v >= P.n_real ? 1.0 : 0.0;       // an unclaimed growth slot: identity block, zero gradient
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          d[r * 6 + c] = Dtt.m[r * 3 + c] + (r == c ? slot_diag : 0.0); d[r * 6 + 3 + c] = Dtq.m[r * 3 + c];
          d[(3 + r) * 6 + c] = Dtq.m[c * 3 + r]; d[(3 + r) * 6 + 3 + c] = Dqq.m[r * 3 + c] + (r == c ? slot_diag : 0.0);
        }
      double *b = bvec + 6 * (int64_t)col;
      b[0] = gt[0]; b[1] = gt[1]; b[2] = gt[2]; b[3] = gq[0]; b[4] = gq[1]; b[5] = gq[2];
    }
  }
  const double s = block_sum<4>(chi, sh);
  if (threadIdx.x == 0) chi_partial[blockIdx.x] = s;
}

// hubs linearised in several slices: the slices' partial sums in entry order, then the diagonal block and gradient
__global__ __launch_bounds__(64) void k_hub_combine(DevPlan P, double *__restrict__ Hblk, double *__restrict__ bvec) {
  __shared__ double sum[HUB_PART];
  const int v = P.hubm[3 * blockIdx.x], e0 = P.hubm[3 * blockIdx.x + 1], ns = P.hubm[3 * blockIdx.x + 2];
  if (threadIdx.x < 33) {
    double a = 0;
    for (int q = 0; q < ns; ++q) a += P.hub_part[(int64_t)(e0 + q) * HUB_PART + threadIdx.x];
    sum[threadIdx.x] = a;
  }
  __syncthreads();
  const int col = P.pose_col[v];
  if (col < 0) return;
  if (threadIdx.x < 36) {
    const int r = threadIdx.x / 6, c = threadIdx.x % 6;
    double x;
    if (r < 3 && c < 3) x = sum[r * 3 + c];
    else if (r < 3) x = sum[9 + r * 3 + (c - 3)];
    else if (c < 3) x = sum[9 + c * 3 + (r - 3)];
    else x = sum[18 + (r - 3) * 3 + (c - 3)];
    Hblk[36 * (int64_t)col + threadIdx.x] = x;
  } else if (threadIdx.x < 42) {
    bvec[6 * (int64_t)col + (threadIdx.x - 36)] = sum[27 + (threadIdx.x - 36)];
  }
}

// Off-diagonal blocks shared by several edges (same vertex pair added more than once): one lane per
// group sums the members serially and overwrites the slot.  Rare; keeps the common path write-once.
__global__ void k_dup_offdiag(DevPlan P, const double *__restrict__ poses, double *__restrict__ Hblk) {
  const int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx >= P.n_dup_groups) return;
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0;
  int slot = -1;
  for (int64_t p = P.dup_ptr[gidx]; p < P.dup_ptr[gidx + 1]; ++p) {
    const int64_t e = P.dup_edges[p];
    const int vi = P.edge_i[e], vj = P.edge_j[e];
    const Pose Xi = load_pose(poses + 8 * (int64_t)vi), Xj = load_pose(poses + 8 * (int64_t)vj);
    const Pose A = load_ainv(P.ainv, P.edge_stride, e);
    const Info3 W = load_info(P.info, P.edge_stride, e);
    EdgeLin L;
    edge_se3<true>(Xi, Xj, A, L);
    const M3 X11 = mm(W.tt, L.Aj), X21 = mtm(W.tq, L.Aj), X12 = mm(W.tq, L.Cj), X22 = mm(W.qq, L.Cj);
    const M3 Ott = mtm(L.Ai, X11), Otq = mtm(L.Ai, X12);
    const M3 Oqt = madd(mtm(L.Bi, X11), mtm(L.Ci, X21)), Oqq = madd(mtm(L.Bi, X12), mtm(L.Ci, X22));
    const int s = P.dup_slot[p];     // (block index << 1) | transpose, per member edge
    slot = s >> 1;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        if ((s & 1) == 0) {
          acc[r * 6 + c] += Ott.m[r * 3 + c]; acc[r * 6 + 3 + c] += Otq.m[r * 3 + c];
          acc[(3 + r) * 6 + c] += Oqt.m[r * 3 + c]; acc[(3 + r) * 6 + 3 + c] += Oqq.m[r * 3 + c];
        } else {
          acc[c * 6 + r] += Ott.m[r * 3 + c]; acc[(3 + c) * 6 + r] += Otq.m[r * 3 + c];
          acc[c * 6 + 3 + r] += Oqt.m[r * 3 + c]; acc[(3 + c) * 6 + 3 + r] += Oqq.m[r * 3 + c];
        }
      }
  }
  if (slot >= 0)
    for (int k = 0; k < 36; ++k) Hblk[36 * (int64_t)slot + k] = acc[k];
}

// chi2 only (CGraphG2O::error): edge-parallel, coalesced SoA reads
__global__ __launch_bounds__(256) void k_chi2(DevPlan P, const double *__restrict__ poses,
                                              double *__restrict__ chi_partial) {
  __shared__ double sh[4];
  double chi = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < P.n_edges; e += (int64_t)gridDim.x * blockDim.x) {
    const Pose Xi = load_pose(poses + 8 * (int64_t)P.edge_i[e]), Xj = load_pose(poses + 8 * (int64_t)P.edge_j[e]);
    const Pose A = load_ainv(P.ainv, P.edge_stride, e);
    const Info3 W = load_info(P.info, P.edge_stride, e);
    EdgeLin L;
    edge_se3<false>(Xi, Xj, A, L);
    const V3 et = {L.e[0], L.e[1], L.e[2]}, eq = {L.e[3], L.e[4], L.e[5]};
    const V3 Wt = mv(W.tt, et) + mv(W.tq, eq);
    const V3 Wq = mtv(W.tq, et) + mv(W.qq, eq);
    chi += et.x * Wt.x + et.y * Wt.y + et.z * Wt.z + eq.x * Wq.x + eq.y * Wq.y + eq.z * Wq.z;
  }
  const double s = block_sum<4>(chi, sh);
  if (threadIdx.x == 0) chi_partial[blockIdx.x] = s;
}

// final pass of the two-pass reductions: one workgroup, fixed order.  mode 0: sum, 1: max
__global__ __launch_bounds__(256) void k_reduce(const double *__restrict__ partial, int64_t n, double *out, int mode) {
  __shared__ double sh[4];
  double acc = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) acc = mode ? fmax(acc, partial[i]) : acc + partial[i];
  if (mode) {
    acc = wave_max(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
  } else {
    const double s = block_sum<4>(acc, sh);
    if (threadIdx.x == 0) *out = s;
  }
}

// computeLambdaInit: max |H_kk| over the diagonal blocks of the REAL variables -- the unclaimed slots of growth mode carry an identity
// block that is no part of the problem (with information far below 1 it would set lambda_0: ADVICE r5)
__global__ __launch_bounds__(256) void k_maxdiag(const double *__restrict__ Hblk, const int *__restrict__ pose_col, int64_t n_real, double *partial) {
  __shared__ double sh[4];
  double m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_real * 6; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / 6; const int r = (int)(i % 6);
    const int k = pose_col[v];
    if (k >= 0) m = fmax(m, fabs(Hblk[36 * (int64_t)k + 7 * r]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

// SparseOptimizer::update (oplus on every free vertex) into the candidate pose buffer, plus the
// LM 'scale' = sum_k x_k (lambda x_k + b_k)  (OptimizationAlgorithmLevenberg::computeScale)
__global__ __launch_bounds__(256) void k_update(DevPlan P, const double *__restrict__ poses,
                                                double *__restrict__ cand, const double *__restrict__ x,
                                                const double *__restrict__ b, const double *__restrict__ lambda_p,
                                                double *__restrict__ scale_partial) {
  __shared__ double sh[4];
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double sc = 0;
  if (v < P.n_poses) {
    Pose X = load_pose(poses + 8 * v);
    const int col = P.pose_col[v];
    if (col >= 0 && v < P.n_real) {                           // (growth slots hold no pose yet)
      const double lambda = *lambda_p;
      double d[6];
      const bool mine = !P.var_mine || P.var_mine[v];      // distributed: every column counted once (x of foreign domains is 0)
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        d[k] = x[6 * (int64_t)col + k];
        if (mine) sc += d[k] * (lambda * d[k] + b[6 * (int64_t)col + k]);
      }
      X = oplus(X, d);
    }
    store_pose(cand + 8 * v, X);
  }
  const double s = block_sum<4>(sc, sh);
  if (threadIdx.x == 0) scale_partial[blockIdx.x] = s;
}

// ------------------------------------------------------------------------------------------------
// Block Cholesky.  L blocks are 6x6 row-major (row = later-eliminated pose).
// Lane mapping ("row per lane"): lane = 6 g + r owns ROW r of the target block handled by lane group g
// (10 groups per wave, lanes 60..63 idle).  One update L_t -= L_a L_b^T costs a lane 3 + 18 16-byte loads
// (its row of L_a, all of L_b; the 6 lanes of a group read the same L_b lines) for 36 FMAs, needs no
// cross-lane traffic, and the row stays in registers through the diagonal Cholesky and the TRSM.
struct Row6 { double v[6]; };

__device__ __forceinline__ Row6 load_row(const double *__restrict__ p) {
  const double2 a = *reinterpret_cast<const double2 *>(p), b = *reinterpret_cast<const double2 *>(p + 2),
                c = *reinterpret_cast<const double2 *>(p + 4);
  return {{a.x, a.y, b.x, b.y, c.x, c.y}};
}
__device__ __forceinline__ void store_row(double *__restrict__ p, const Row6 &x) {
  *reinterpret_cast<double2 *>(p) = make_double2(x.v[0], x.v[1]);
  *reinterpret_cast<double2 *>(p + 2) = make_double2(x.v[2], x.v[3]);
  *reinterpret_cast<double2 *>(p + 4) = make_double2(x.v[4], x.v[5]);
}
// acc -= Arow * B^T, B a full 6x6 block in memory
__device__ __forceinline__ void row_update(Row6 &acc, const Row6 &a, const double *__restrict__ B) {
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const Row6 b = load_row(B + 6 * c);
    acc.v[c] -= a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2] + a.v[3] * b.v[3] + a.v[4] * b.v[4] + a.v[5] * b.v[5];
  }
}
// row r of the H block feeding target t (+ lambda on the diagonal of a diagonal block), or zeros (fill-in)
__device__ __forceinline__ Row6 load_A_row(const DevPlan &P, const double *__restrict__ Hblk, int64_t t, int r, double lambda) {
  const int a = P.asrc[t];
  Row6 x = {{0, 0, 0, 0, 0, 0}};
  if (a >= 0) {
    x = load_row(Hblk + 36 * (int64_t)a + 6 * r);
    if (a < P.nb) {                                   // setLambda: H_pp diagonal += lambda
#pragma unroll
      for (int c = 0; c < 6; ++c) x.v[c] += (c == r) ? lambda : 0.0;   // static indexing keeps x in VGPRs
    }
  }
  return x;
}
// where an accumulate target's value starts (AccDesc::a): row r of H block a (+ lambda on the diagonal of a diagonal block), zeros, or its value in L
__device__ __forceinline__ Row6 acc_start_row(const DevPlan &P, const double *__restrict__ Hblk, const double *__restrict__ Lv, const AccDesc &d, int r, double lambda) {
  Row6 x = {{0, 0, 0, 0, 0, 0}};
  if (d.a == -2) x = load_row(Lv + 36 * (int64_t)d.t + 6 * r);
  else if (d.a >= 0) {
    x = load_row(Hblk + 36 * (int64_t)d.a + 6 * r);
    if (d.a < P.nb) {
#pragma unroll
      for (int c = 0; c < 6; ++c) x.v[c] += (c == r) ? lambda : 0.0;
    }
  }
  return x;
}
// Apply ops [o0, o1) with stride `step` to this lane's row, in wave-uniform batches of OPB ops.
// Memory-level parallelism is the point: per batch every lane first issues the index loads of the NEXT batch,
// then 6*OPB independent 16-byte loads (its row of each L_a and ONE row of each L_b); the other five rows of
// L_b come from the five sibling lanes through a wave-private LDS tile (in-order DS pipe, no barrier).
// Lanes that ran out of ops (or idle lanes) read the all-zero block P.zero_blk, which changes nothing.
constexpr int OPB = 2;
__device__ __forceinline__ void apply_ops(const DevPlan &P, const double *__restrict__ Lv, Row6 &acc, int g, int r,
                                          int64_t o0, int64_t o1, int step, double *__restrict__ tile) {
  int64_t o = o0;
  int ia[OPB], ib[OPB];
#pragma unroll
  for (int k = 0; k < OPB; ++k) {
    const int64_t q = o + (int64_t)k * step;
    const bool in = q < o1;
    ia[k] = in ? P.op_a[q] : P.zero_blk;
    ib[k] = in ? P.op_b[q] : P.zero_blk;
  }
  double *mine = tile + 36 * g;
  while (__any(o < o1)) {
    o += (int64_t)OPB * step;
    int na[OPB], nb2[OPB];
#pragma unroll
    for (int k = 0; k < OPB; ++k) {            // next batch's indices first: they complete before the rows below
      const int64_t q = o + (int64_t)k * step;
      const bool in = q < o1;
      na[k] = in ? P.op_a[q] : P.zero_blk;
      nb2[k] = in ? P.op_b[q] : P.zero_blk;
    }
    Row6 a[OPB], b[OPB];
#pragma unroll
    for (int k = 0; k < OPB; ++k) {
      a[k] = load_row(Lv + 36 * (int64_t)ia[k] + 6 * r);
      b[k] = load_row(Lv + 36 * (int64_t)ib[k] + 6 * r);
    }
#pragma unroll
    for (int k = 0; k < OPB; ++k) {
      store_row(mine + 6 * r, b[k]);
      __builtin_amdgcn_wave_barrier();
      row_update(acc, a[k], mine);
      __builtin_amdgcn_wave_barrier();
      ia[k] = na[k]; ib[k] = nb2[k];
    }
  }
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Neighbouring work items (targets of one
// column, row chunks of one panel) read the same source blocks / operand tiles, so work item ids are assigned such
// that every XCD gets a CONTIGUOUS range of them: the shared blocks then hit in that XCD's L2 instead of being
// fetched once per XCD (rocprofv3 FETCH_SIZE of the accumulate kernel: profiles/).
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7;
  return x * q + (x < r ? x : r) + (b >> 3);
}

// partial re-factorisation: is this task part of the sweep?  (task_dirty == NULL: everything is)
__device__ __forceinline__ bool task_runs(const DevPlan &P, int task) { return !P.task_dirty || P.task_dirty[task]; }

// The right-hand side as one more row of the matrix: external part of the forward solve of a panel column,
// x_k <- b_k - sum_{j outside the panel} L_kj y_j, by the forward-solve workgroups of the accumulate launches (NW waves,
// fixed summation order).  Work item w of the level (DevPlan::fwg_*) is either a whole panel column (chunk < 0) or ONE CHUNK
// of FWD_CHUNK entries of a long row -- a hub column's row holds tens of thousands of entries (cfg 4: 54 k = 15 MB for one
// workgroup, 630 us); its chunks are summed by separate workgroups into fpart, and k_fwd_combine -- one more launch, only at
// levels that have such rows -- subtracts them from x in chunk order.  (Doing that inside this launch by "the last workgroup to
// arrive" was tried: the agent-scope fences it needs write back the XCD's L2 while the accumulate workgroups are filling it
// with dirty blocks -- 100 us for 340 chunk workgroups.)  One code path for both forms (the chunked one must not cost the
// accumulate kernels registers: k_chol_acc2<1> runs 7 waves per SIMD).
template <int NW>
__device__ __forceinline__ void fwd_role(const DevPlan &P, const double *__restrict__ Lv, double *__restrict__ x, int base, int off, double *__restrict__ sred) {
  // base >= 0: a level without split rows -- item `off` is entry base + off of task_cols (no table look-up on the launch's
  // critical path); base < 0: items -1 - base + off of the work-item table
  const int ci = base >= 0 ? base + off : P.fwg_ci[-1 - base + off], ch = base >= 0 ? -1 : P.fwg_ch[-1 - base + off];
  if (P.task_dirty && !P.task_dirty[P.tcol_task[ci]]) return;
  const int k = P.task_cols[ci];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, r = lane - 6 * g;
  const int gid = wave * 10 + g;
  constexpr int ST = NW * 10;
  const int64_t rm = P.pp.row_mid[k];
  // whole row: from its first external entry (top: the domain part arrived by all-reduce); chunk: its FWD_CHUNK entries
  const int64_t e0 = ch >= 0 ? P.pp.fchunk_e0[ch] : ((P.dist && k >= P.top_col0) ? P.top_row0[k - P.top_col0] : P.rowptr[k]);
  const int64_t e1 = (ch >= 0 && e0 + FWD_CHUNK < rm) ? e0 + FWD_CHUNK : rm;
  double acc = 0;
  if (lane < 60) {
    for (int64_t e = e0 + gid; e < e1; e += 4 * ST) {
      int bi[4], cj[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t ee = e + ST * q;
        const bool in = ee < e1;
        bi[q] = in ? P.row_blk[ee] : P.zero_blk;
        cj[q] = in ? P.row_col[ee] : 0;
      }
      Row6 l[4], y[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { l[q] = load_row(Lv + 36 * (int64_t)bi[q] + 6 * r); y[q] = load_row(x + 6 * (int64_t)cj[q]); }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc += l[q].v[0] * y[q].v[0] + l[q].v[1] * y[q].v[1] + l[q].v[2] * y[q].v[2] + l[q].v[3] * y[q].v[3] + l[q].v[4] * y[q].v[4] + l[q].v[5] * y[q].v[5];
    }
    sred[gid * 6 + r] = acc;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double sv = ch >= 0 ? 0.0 : x[6 * (int64_t)k + threadIdx.x];
    for (int q = 0; q < ST; ++q) sv -= sred[q * 6 + threadIdx.x];
    if (ch >= 0) P.pp.fpart[6 * (int64_t)ch + threadIdx.x] = -sv; else x[6 * (int64_t)k + threadIdx.x] = sv;
  }
}

// x_k -= the chunk sums of a split row, in chunk order (one wave per split column; launched between the accumulate and the
// triangle launch of a level that has such rows).  The partial sums are fetched by the whole wave, then subtracted one
// after the other: the order is fixed.
__global__ __launch_bounds__(64) void k_fwd_combine(DevPlan P, double *__restrict__ x, int s0) {
  __shared__ double buf[360];
  const int ci = P.fsplit_ci[s0 + blockIdx.x];
  if (P.task_dirty && !P.task_dirty[P.tcol_task[ci]]) return;
  const int k = P.task_cols[ci], f0 = P.fwd_f0[ci], fn = P.fwd_fn[ci];
  double sv = threadIdx.x < 6 ? x[6 * (int64_t)k + threadIdx.x] : 0.0;
  for (int q0 = 0; q0 < fn; q0 += 60) {
    const int nq = fn - q0 < 60 ? fn - q0 : 60;
    __syncthreads();
    for (int i = threadIdx.x; i < 6 * nq; i += 64) buf[i] = P.pp.fpart[6 * (int64_t)(f0 + q0) + i];
    __syncthreads();
    if (threadIdx.x < 6) for (int q = 0; q < nq; ++q) sv -= buf[6 * q + threadIdx.x];
  }
  if (threadIdx.x < 6) x[6 * (int64_t)k + threadIdx.x] = sv;
}

// wide accumulate: external sources only.  One workgroup (4 waves) per 10 target blocks; the 4 waves
// split each target's source list (split-K) and wave 0 combines the partial rows from LDS in a fixed order.
// Workgroups beyond n_acc_wg (panel levels with a fused forward solve) take one panel column of the right-hand
// side each: same dependencies as the accumulation, so it rides in the same launch.
template <int SPLIT>
__global__ __launch_bounds__(SPLIT * 64) void k_chol_acc(DevPlan P, const double *__restrict__ Hblk, double *__restrict__ Lv,
                                                         int64_t first, int64_t count, const double *__restrict__ lambda_p,
                                                         double *__restrict__ x, int n_acc_wg, int col0, int n_long, int64_t long0) {
  __shared__ __attribute__((aligned(16))) double tile[SPLIT][360];
  __shared__ __attribute__((aligned(16))) double part[SPLIT][60][6];
  if ((int)blockIdx.x >= n_acc_wg + n_long) {
    fwd_role<SPLIT>(P, Lv, x, col0, (int)blockIdx.x - n_acc_wg - n_long, &part[0][0][0]);      // col0: first forward work item of the level (fwd_role)
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, r = lane - 6 * g;
  if ((int)blockIdx.x >= n_acc_wg) {
    // a target with a long source list (hub column, top separator): all SPLIT*10 lane groups of the workgroup stride
    // through ITS list, then the partial blocks are summed in a fixed order
    // (n_acc_wg is a multiple of 8 when there are long targets, so the XCD of this workgroup is (blockIdx - n_acc_wg) & 7:
    //  every XCD takes a CONTIGUOUS range of the long targets -- the targets of a column / panel share their sources)
    const int64_t ti = long0 + xcd_contiguous((int)blockIdx.x - n_acc_wg, n_long);      // (long0 = first + count in a full sweep)
    if (P.task_dirty && !P.task_dirty[P.acc_task[ti]]) return;
    // (AccDesc, fgo_structure.cpp: riders have applied the head of the list -> continue from the value in L, or, for a hub target, from H: its
    //  riders left their sums in scratch blocks that the rest of the list subtracts; a top block's value (incl. the domains' updates) sits in L)
    const AccDesc dsc = P.acc_desc[ti];
    const int64_t t = dsc.t;
    const int gid = wave * 10 + g;
    Row6 acc = {{0, 0, 0, 0, 0, 0}};
    if (lane < 60) {
      if (gid == 0) acc = acc_start_row(P, Hblk, Lv, dsc, r, *lambda_p);
      apply_ops(P, Lv, acc, g, r, dsc.o0 + gid, dsc.o1, SPLIT * 10, tile[wave]);
#pragma unroll
      for (int c = 0; c < 6; ++c) part[wave][lane][c] = acc.v[c];
    }
    __syncthreads();
    if (threadIdx.x < 36) {
      const double *pf = &part[0][0][0];               // [(wave * 10 + g) * 36 + 6 r + c]
      double s = 0;
      for (int q = 0; q < SPLIT * 10; ++q) s += pf[q * 36 + threadIdx.x];
      Lv[36 * t + threadIdx.x] = s;
    }
    return;
  }
  const int64_t idx = (int64_t)xcd_contiguous(blockIdx.x, n_acc_wg) * 10 + g;
  bool on = lane < 60 && idx < count;
  if (P.task_dirty) {                                       // partial sweep: targets of clean tasks keep their value
    if (on && !P.task_dirty[P.acc_task[first + idx]]) on = false;      // (the same decision on every wave of the workgroup)
    if (!__syncthreads_or(on)) return;
  }
  Row6 acc = {{0, 0, 0, 0, 0, 0}};
  int64_t t = 0;
  if (on) {
    const AccDesc dsc = P.acc_desc[first + idx];
    t = dsc.t;
    if (wave == 0) acc = acc_start_row(P, Hblk, Lv, dsc, r, *lambda_p);
    apply_ops(P, Lv, acc, g, r, dsc.o0 + wave, dsc.o1, SPLIT, tile[wave]);
    if (wave > 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) part[wave][lane][c] = acc.v[c];
    }
  }
  __syncthreads();
  if (on && wave == 0) {
    for (int w = 1; w < SPLIT; ++w)            // fixed order: deterministic
#pragma unroll
      for (int c = 0; c < 6; ++c) acc.v[c] += part[w][lane][c];
    store_row(Lv + 36 * t + 6 * r, acc);
  }
}

// Column-group accumulate: one workgroup (SPLIT waves) per group of <= 10 targets of ONE column k; lane group g owns
// target (i_g, k), lane r its row r.  The group's entries are the external source columns j, ascending; per entry the B
// operand L(k, j) is the same for every lane of the wave -- its address is wave-uniform, so it is fetched with SCALAR
// loads into SGPRs and costs no vector memory traffic and no LDS exchange -- and lane group g reads only its own A block
// L(i_g, j) (the zero block where row i_g is not in pattern(j)).  Per update 288 bytes instead of 576 and 3 vector loads
// instead of 6 + an LDS round trip; the updates of a target arrive in the same ascending source order and with the same
// arithmetic as in the gather form, so the factor is bit-identical.  SPLIT > 1 splits the entry list across the waves
// (short dependent chains at the skinny top), partial rows combined from LDS in a fixed order.
template <int SPLIT>
__global__ __launch_bounds__(SPLIT * 64) void k_chol_acc2(DevPlan P, const double *__restrict__ Hblk, double *__restrict__ Lv,
                                                          const double *__restrict__ Lsrc, int64_t group0, int n_groups,
                                                          const double *__restrict__ lambda_p, double *__restrict__ x, int col0) {
  __shared__ __attribute__((aligned(16))) double part[SPLIT][60][6];
  if ((int)blockIdx.x >= n_groups) {                      // fused forward solve: one panel column's external part
    fwd_role<SPLIT>(P, Lv, x, col0, (int)blockIdx.x - n_groups, &part[0][0][0]);
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, r = lane - 6 * g;
  const int64_t grp = group0 + xcd_contiguous(blockIdx.x, n_groups);
  if (P.task_dirty && !P.task_dirty[P.g2_task[grp]]) return;
  const int64_t t = lane < 60 ? (int64_t)P.g2_tgt[grp * ACC2_G + g] : -1;
  const bool on = t >= 0;
  Row6 acc = {{0, 0, 0, 0, 0, 0}};
  if (on && wave == 0) acc = (P.dist && t >= P.top_blk0) ? load_row(Lv + 36 * t + 6 * r) : load_A_row(P, Hblk, t, r, *lambda_p);
  const int64_t e0 = P.g2_ptr[grp], e1 = P.g2_ptr[grp + 1];
  const int gg = lane < 60 ? g : 0;
  // two entries in flight: the indices of the next pair are requested before the rows of this pair are used
  int64_t e = e0 + wave;
  int ia0 = e < e1 ? P.g2_a[e * ACC2_G + gg] : P.zero_blk, ib0 = e < e1 ? P.g2_b[e] : P.zero_blk;
  int ia1 = e + SPLIT < e1 ? P.g2_a[(e + SPLIT) * ACC2_G + gg] : P.zero_blk, ib1 = e + SPLIT < e1 ? P.g2_b[e + SPLIT] : P.zero_blk;
  while (e < e1) {
    const int64_t en = e + 2 * SPLIT;
    const int na0 = en < e1 ? P.g2_a[en * ACC2_G + gg] : P.zero_blk, nb0 = en < e1 ? P.g2_b[en] : P.zero_blk;
    const int na1 = en + SPLIT < e1 ? P.g2_a[(en + SPLIT) * ACC2_G + gg] : P.zero_blk, nb1 = en + SPLIT < e1 ? P.g2_b[en + SPLIT] : P.zero_blk;
    const Row6 a0 = load_row(Lsrc + 36 * (int64_t)ia0 + 6 * r), a1 = load_row(Lsrc + 36 * (int64_t)ia1 + 6 * r);
    const double *__restrict__ B0 = Lsrc + 36 * (int64_t)__builtin_amdgcn_readfirstlane(ib0);
    const double *__restrict__ B1 = Lsrc + 36 * (int64_t)__builtin_amdgcn_readfirstlane(ib1);
#pragma unroll
    for (int c = 0; c < 6; ++c)
      acc.v[c] -= a0.v[0] * B0[6 * c] + a0.v[1] * B0[6 * c + 1] + a0.v[2] * B0[6 * c + 2] + a0.v[3] * B0[6 * c + 3] + a0.v[4] * B0[6 * c + 4] + a0.v[5] * B0[6 * c + 5];
#pragma unroll
    for (int c = 0; c < 6; ++c)
      acc.v[c] -= a1.v[0] * B1[6 * c] + a1.v[1] * B1[6 * c + 1] + a1.v[2] * B1[6 * c + 2] + a1.v[3] * B1[6 * c + 3] + a1.v[4] * B1[6 * c + 4] + a1.v[5] * B1[6 * c + 5];
    ia0 = na0; ib0 = nb0; ia1 = na1; ib1 = nb1;
    e = en;
  }
  if (SPLIT > 1) {
    if (on && wave > 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) part[wave][lane][c] = acc.v[c];
    }
    __syncthreads();
    if (on && wave == 0) {
      for (int w = 1; w < SPLIT; ++w)
#pragma unroll
        for (int c = 0; c < 6; ++c) acc.v[c] += part[w][lane][c];
    }
  }
  if (on && wave == 0) store_row(Lv + 36 * t + 6 * r, acc);
}


// ------------------------------------------------------------------------------------------------
// 1 / sqrt(d): hardware estimate (v_rsq_f64: about 2^-23 relative) + ONE third-order step
//   y1 = y0 (1 + e/2 + 3 e^2/8),  e = 1 - d y0^2      (error ~ e^3: below the rounding of the result)
// arranged as a chain of four dependent operations (t, e, {p | y0 e}, fma) -- the factor kernels are latency chains of
// these: two Newton steps cost eight, a correctly rounded sqrt followed by a correctly rounded division three times as many.
__device__ __forceinline__ double rsqrt_nr(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double t = d * y;
  const double e = fma(-t, y, 1.0);              // 1 - d y^2
  const double p = fma(0.375, e, 0.5);
  const double ye = y * e;
  return fma(ye, p, y);
}
// scalar Cholesky of a 6x6 read from LDS (every lane computes the same factor: no second barrier needed).
// L packed lower: index(i,j) = i(i+1)/2 + j; invd[j] = 1 / L_jj.
// RIGHT-LOOKING: column j is scaled, then the trailing triangle is updated at once, so the dependent chain from one pivot
// to the next is rsqrt -> one multiply -> one FMA (about 9 operations per column) instead of the j-deep dot products of
// the left-looking form (these factorisations sit on the critical path of every column of every panel and leaf task).
__device__ __forceinline__ bool chol6_lds(const double *__restrict__ sd, double L[21], double invd[6]) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) if (j <= i) L[i * (i + 1) / 2 + j] = sd[i * 6 + j];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double d = L[j * (j + 1) / 2 + j];
    if (!(d > 0.0)) ok = false;
    const double inv = rsqrt_nr(d);
    invd[j] = inv;
    L[j * (j + 1) / 2 + j] = d * inv;
#pragma unroll
    for (int i = 0; i < 6; ++i) if (i > j) L[i * (i + 1) / 2 + j] *= inv;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (i > j && k > j && k <= i) L[i * (i + 1) / 2 + k] = fma(-L[i * (i + 1) / 2 + j], L[k * (k + 1) / 2 + j], L[i * (i + 1) / 2 + k]);
  }
  return ok;
}
// x = u * L^-T for one row, right-looking as well: x_c = u_c / L_cc, then u_c' -= x_c L_c'c for c' > c
__device__ __forceinline__ Row6 trsm_row(const Row6 &u, const double L[21], const double invd[6]) {
  Row6 x, w = u;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    x.v[c] = w.v[c] * invd[c];
#pragma unroll
    for (int c2 = 0; c2 < 6; ++c2) if (c2 > c) w.v[c2] = fma(-x.v[c], L[c2 * (c2 + 1) / 2 + c], w.v[c2]);
  }
  return x;
}

// one workgroup per task: its columns in ascending order; internal updates, 6x6 Cholesky, block TRSM.
// Up to MAXP*NW*10 blocks of a column stay in registers between the update and the TRSM; larger columns
// take extra passes through HBM/L2.
template <int NW, int MAXP>
__global__ __launch_bounds__(NW * 64) void k_chol_fact(DevPlan P, const double *__restrict__ Hblk,
                                                       double *__restrict__ Lv, int task0,
                                                       const double *__restrict__ lambda_p, int *__restrict__ fail_flag) {
  __shared__ __attribute__((aligned(16))) double tile[NW][360];
  __shared__ double sdiag[36];
  const int task = task0 + blockIdx.x;
  if (!task_runs(P, task)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, r = lane - 6 * g;
  const bool lane_on = lane < 60;
  const double lambda = *lambda_p;
  const int c_begin = P.task_ptr[task], c_end = P.task_ptr[task + 1];
  constexpr int PER_PASS = NW * 10;
  for (int ci = c_begin; ci < c_end; ++ci) {
    const int k = P.task_cols[ci];
    const int64_t b0 = P.colptr[k], b1 = P.colptr[k + 1];
    Row6 acc[MAXP];
    // phase 1: finish the accumulation of every block of column k
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int64_t t = b0 + (int64_t)(p * NW + wave) * 10 + g;
      if (lane_on && t < b1) {
        const int64_t om = P.op_mid[t];
        acc[p] = (om == P.op_ptr[t] && !(P.dist && t >= P.top_blk0)) ? load_A_row(P, Hblk, t, r, lambda) : load_row(Lv + 36 * t + 6 * r);
        apply_ops(P, Lv, acc[p], g, r, om, P.op_ptr[t + 1], 1, tile[wave]);
        if (t == b0) {
#pragma unroll
          for (int c = 0; c < 6; ++c) sdiag[6 * r + c] = acc[p].v[c];
        }
      }
    }
    for (int64_t t = b0 + (int64_t)(MAXP * NW + wave) * 10 + g; lane_on && t < b1; t += PER_PASS) {   // overflow passes
      const int64_t om = P.op_mid[t];
      Row6 a = (om == P.op_ptr[t] && !(P.dist && t >= P.top_blk0)) ? load_A_row(P, Hblk, t, r, lambda) : load_row(Lv + 36 * t + 6 * r);
      apply_ops(P, Lv, a, g, r, om, P.op_ptr[t + 1], 1, tile[wave]);
      store_row(Lv + 36 * t + 6 * r, a);
    }
    __syncthreads();
    // phase 2: every lane factors the diagonal block (same arithmetic everywhere), then scales its rows
    double Lk[21], invd[6];
    const bool ok = chol6_lds(sdiag, Lk, invd);
    if (!ok && threadIdx.x == 0) atomicOr(fail_flag, 1);
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int64_t t = b0 + (int64_t)(p * NW + wave) * 10 + g;
      if (lane_on && t < b1) {
        Row6 x;
        if (t == b0) {
#pragma unroll
          for (int rr = 0; rr < 6; ++rr)       // static indices only: Lk must stay in registers
            if (rr == r) {
#pragma unroll
              for (int c = 0; c < 6; ++c) x.v[c] = (c <= rr) ? Lk[rr * (rr + 1) / 2 + c] : 0.0;
            }
        } else {
          x = trsm_row(acc[p], Lk, invd);
        }
        store_row(Lv + 36 * t + 6 * r, x);
      }
    }
    for (int64_t t = b0 + (int64_t)(MAXP * NW + wave) * 10 + g; lane_on && t < b1; t += PER_PASS) {
      const Row6 u = load_row(Lv + 36 * t + 6 * r);
      store_row(Lv + 36 * t + 6 * r, trsm_row(u, Lk, invd));
    }
    __syncthreads();
  }
}

// Leaf levels: a task is a self-contained light sub-tree -- contiguous columns, hence ONE contiguous range of L blocks,
// no updates from outside.  The whole range lives in LDS under local indices (block t -> t - base) while the task is
// factored: every update reads its two source blocks from LDS instead of L2 / HBM (the generic kernel re-reads each
// block ~20 times; at cfg 5 level 0 alone moved 40 GB per sweep through the caches), and L goes to memory once, as
// one coalesced copy.  The task's op lists are staged in LDS too (16-bit local ids), so the column loop touches no
// global memory at all (with the indices streamed from memory the kernel was bound by two dependent loads per column).
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_chol_leaf(DevPlan P, const double *__restrict__ Hblk, double *__restrict__ Lv, int task0,
                                                       const double *__restrict__ lambda_p, int *__restrict__ fail_flag, int lds_blocks,
                                                       int lds_cols, double *__restrict__ x) {
  // [lds_blocks][36] blocks of L | the diagonal block being factored | right-hand side r and forward solution y of the
  // task's columns | op list offsets | local column of every block's row (0xffff: outside the task) | ops as
  // (a << 16 | b), local ids.  With x != nullptr the forward solve L y = b of the task's columns rides along (x holds b
  // on entry): once column k is final every lane knows L_kk, computes y_k and subtracts L_ik y_k from the rows i of its
  // own blocks -- column-oriented substitution with no extra barrier and no extra pass over L.
  extern __shared__ __attribute__((aligned(16))) double Ls[];
  // one descriptor per task (fgo_structure.cpp "LeafDesc"): in a full sweep the level's tasks by descending work, in a partial sweep
  // (a task range) in task order
  const LeafDesc dsc = (P.task_dirty ? P.pp.leaf_desc : P.pp.leaf_lpt)[task0 + blockIdx.x];
  const int task = dsc.task;
  if (!task_runs(P, task)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, r = lane - 6 * g;
  const bool lane_on = lane < 60;
  const double lambda = *lambda_p;
  const int m = dsc.m;
  const int k0 = dsc.k0;
  const int64_t base = dsc.base;
  const int nblk = dsc.nblk;
  double *__restrict__ sdiag = Ls + 36 * lds_blocks;
  double *__restrict__ xr = sdiag + 36;                              // [lds_cols][6] running right-hand side
  double *__restrict__ yr = xr + 6 * lds_cols;                       // [lds_cols][6] forward solution
  int *__restrict__ lcolp = reinterpret_cast<int *>(yr + 6 * lds_cols);   // [lds_cols + 2] first block of every column (local), staged once: the column loop reads it from LDS
  int *__restrict__ lptr = lcolp + lds_cols + 2;
  unsigned short *__restrict__ lrow = reinterpret_cast<unsigned short *>(lptr + lds_blocks + 1);
  unsigned *__restrict__ lop = reinterpret_cast<unsigned *>(lrow + 2 * ((lds_blocks + 1) / 2));
  const int64_t obase = dsc.obase;
  const int nops = dsc.nops;
  for (int q = threadIdx.x; q <= m; q += NW * 64) lcolp[q] = (int)(P.colptr[k0 + q] - base);
  for (int q = threadIdx.x; q <= nblk; q += NW * 64) lptr[q] = (int)(P.op_ptr[base + q] - obase);
  if (x) {
    for (int q = threadIdx.x; q < nblk; q += NW * 64) { const int i = P.rowidx[base + q] - k0; lrow[q] = (unsigned short)(i < m ? i : 0xffff); }
    for (int i = threadIdx.x; i < 6 * m; i += NW * 64) xr[i] = x[6 * (int64_t)k0 + i];
  }
  for (int i = threadIdx.x; i < nops; i += NW * 64)
    lop[i] = ((unsigned)(P.op_a[obase + i] - (int)base) << 16) | (unsigned)(P.op_b[obase + i] - (int)base);
  for (int q = wave * 10 + g; lane_on && q < nblk; q += NW * 10) store_row(Ls + 36 * q + 6 * r, load_A_row(P, Hblk, base + q, r, lambda));
  __syncthreads();
  for (int ci = 0; ci < m; ++ci) {
    const int b0 = lcolp[ci], b1 = lcolp[ci + 1];
    for (int q = b0 + wave * 10 + g; lane_on && q < b1; q += NW * 10) {
      Row6 acc = load_row(Ls + 36 * q + 6 * r);
      int o = lptr[q];
      const int o1 = lptr[q + 1];
      for (; o + 1 < o1; o += 2) {
        const unsigned p0 = lop[o], p1 = lop[o + 1];
        const Row6 a0 = load_row(Ls + 36 * (int)(p0 >> 16) + 6 * r), a1 = load_row(Ls + 36 * (int)(p1 >> 16) + 6 * r);
        row_update(acc, a0, Ls + 36 * (int)(p0 & 0xffffu));
        row_update(acc, a1, Ls + 36 * (int)(p1 & 0xffffu));
      }
      if (o < o1) { const unsigned p0 = lop[o]; row_update(acc, load_row(Ls + 36 * (int)(p0 >> 16) + 6 * r), Ls + 36 * (int)(p0 & 0xffffu)); }
      store_row(Ls + 36 * q + 6 * r, acc);                              // sources are blocks of earlier columns: no hazard
      if (q == b0) store_row(sdiag + 6 * r, acc);
    }
    __syncthreads();
    double Lk[21], invd[6];
    const bool ok = chol6_lds(sdiag, Lk, invd);
    if (!ok && threadIdx.x == 0) atomicOr(fail_flag, 1);
    double yk[6] = {0, 0, 0, 0, 0, 0};
    if (x) {                                                            // y_k = L_kk^-1 r_k, the same arithmetic on every lane
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) {
        double sv = xr[6 * ci + rr];
#pragma unroll
        for (int c = 0; c < 6; ++c) if (c < rr) sv -= Lk[rr * (rr + 1) / 2 + c] * yk[c];
        yk[rr] = sv * invd[rr];
      }
    }
    for (int q = b0 + wave * 10 + g; lane_on && q < b1; q += NW * 10) {
      Row6 lq;
      if (q == b0) {
#pragma unroll
        for (int rr = 0; rr < 6; ++rr)                                  // static indices only: Lk must stay in registers
          if (rr == r) {
#pragma unroll
            for (int c = 0; c < 6; ++c) lq.v[c] = (c <= rr) ? Lk[rr * (rr + 1) / 2 + c] : 0.0;
          }
        if (x) {
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) if (rr == r) yr[6 * ci + rr] = yk[rr];
        }
      } else {
        lq = trsm_row(load_row(Ls + 36 * q + 6 * r), Lk, invd);
        if (x) {
          const int i2 = lrow[q];
          if (i2 != 0xffff)                                             // rows inside the task; the others gather later
            xr[6 * i2 + r] -= lq.v[0] * yk[0] + lq.v[1] * yk[1] + lq.v[2] * yk[2] + lq.v[3] * yk[3] + lq.v[4] * yk[4] + lq.v[5] * yk[5];
        }
      }
      store_row(Ls + 36 * q + 6 * r, lq);
    }
    __syncthreads();
  }
  if (x) for (int i = threadIdx.x; i < 6 * m; i += NW * 64) x[6 * (int64_t)k0 + i] = yr[i];
  double2 *__restrict__ dst = reinterpret_cast<double2 *>(Lv + 36 * base);
  const double2 *__restrict__ src = reinterpret_cast<const double2 *>(Ls);
  for (int i = threadIdx.x; i < nblk * 18; i += NW * 64) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// Panels (the skinny top of the elimination tree).  A panel = m <= PM columns forming a path of the tree: a dense
// m x m lower triangle of blocks plus off-triangle rows that all start at some column and run to the last one.
// External updates were already applied by k_chol_acc.  k_panel_tri factors the triangle on chip (packed triangle in
// LDS, trailing matrix in f64 MFMA accumulator tiles, one barrier per column) and leaves it behind as 16x16 operand
// tiles; k_panel_rows then finishes the off-triangle rows as a blocked TRSM on MFMA (16 scalar rows per wave, the
// right-hand side riding along as one more row); k_bwd_ext / k_bwd_tri do the backward solve from the same tiles.
struct PairTab { unsigned char a[PM * (PM + 1) / 2], b[PM * (PM + 1) / 2]; };    // (a, b), b <= a, a ascending
constexpr PairTab make_pairs() {
  PairTab t{};
  int q = 0;
  for (int a = 0; a < PM; ++a)
    for (int b = 0; b <= a; ++b) { t.a[q] = (unsigned char)a; t.b[q] = (unsigned char)b; ++q; }
  return t;
}
__constant__ PairTab PAIRS = make_pairs();
#define PAIR_A PAIRS.a
#define PAIR_B PAIRS.b
constexpr int NLT = NJMAX * (NJMAX - 1) / 2;           // strictly-lower tiles
// packed lower triangle of 6x6 blocks in LDS: block (rr, kk), kk <= rr
#define TRI(rr, kk) ((((rr) * ((rr) + 1)) / 2 + (kk)) * 36)

template <bool FROM_LDS>
__device__ __forceinline__ Row6 trsm_row_blk(const Row6 &u, const double *__restrict__ L) {   // L: 6x6 row-major, lower
  Row6 x;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double s = u.v[c];
#pragma unroll
    for (int m = 0; m < 6; ++m) if (m < c) s -= x.v[m] * L[6 * c + m];
    x.v[c] = s / L[7 * c];
  }
  return x;
}

// Rider role of a 16-wave k_panel_tri launch (symbolic.cpp "riders"): workgroup wg takes items 4 wg .. 4 wg + 3, four waves
// each -- the 40 lane groups stride the item's op range like the long-list role of k_chol_acc, partial blocks are summed in
// a fixed order, and the target's value so far goes (back) to L.  smem: 16 x 360 doubles (a wave's exchange tile, then its
// partial rows).
constexpr int RIDE_PER_WG = 4;
__device__ __forceinline__ void ride_items16(const DevPlan &P, const double *__restrict__ Hblk, double *__restrict__ Lv,
                                             const double *__restrict__ lambda_p, int item0, int nitems, int wg, double *__restrict__ smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, r = lane - 6 * g;
  const int part = wave >> 2, w4 = wave & 3;
  const int ii = RIDE_PER_WG * wg + part;
  const bool have = ii < nitems;
  const RideItem it = P.ride_items[item0 + (have ? ii : 0)];
  const bool run = have && (!P.task_dirty || P.task_dirty[it.task]);
  double *__restrict__ tile = smem + 360 * wave;
  if (run && lane < 60) {
    const int gid = w4 * 10 + g;
    Row6 acc = {{0, 0, 0, 0, 0, 0}};
    if (gid == 0 && it.first != 2) acc = it.first ? load_A_row(P, Hblk, it.t, r, *lambda_p) : load_row(Lv + 36 * (int64_t)it.t + 6 * r);
    apply_ops(P, Lv, acc, g, r, it.o0 + gid, it.o0 + it.n, 40, tile);
#pragma unroll
    for (int c = 0; c < 6; ++c) tile[6 * lane + c] = acc.v[c];       // (the wave's own tile: its DS operations are in order)
  }
  __syncthreads();
  const int tl = (int)threadIdx.x - 256 * part;
  if (run && tl < 36) {
    const double *pf = smem + 1440 * part;                            // [(w4 * 10 + g) * 36 + 6 r + c]
    double sum = 0;
    for (int q = 0; q < 40; ++q) sum += pf[q * 36 + tl];
    Lv[36 * (int64_t)it.t + tl] = it.first == 2 ? -sum : sum;         // (a hub piece: the scratch block receives + sum L_a L_b^T)
  }
}

// Dense Cholesky of a panel's triangle (<= PM block columns, <= 6 PM scalar columns), entirely on chip.  The packed
// lower triangle of 6x6 blocks sits in LDS (PM = 32: 528 blocks = 152 KB); the trailing matrix lives in registers as
// 16x16 f64 MFMA accumulator tiles spread over waves 1 .. NW-1.  Wave 0 runs the sequential pivot chain, one block
// column per phase and one barrier per phase:
//   wave 0, phase k:   column k (staged in LDS with the updates of columns < k-1) gets the update of column k-1, the
//                      6x6 diagonal block is factored, the rows below are scaled -> V(k) (final L blocks) in LDS
//   the others:        rank-6 update C -= V(k-1) V(k-1)^T (two MFMAs per tile), then they stage column k+1 (now
//                      carrying the updates of columns <= k-1) for the phase after next
// so the MFMA work and the staging hide behind the pivot chain instead of alternating with it.  Epilogue: L blocks to
// global memory, and the same triangle once more as 16x16 tiles in MFMA operand order (strictly-lower tiles negated,
// diagonal tiles inverted) for k_panel_rows and the panel solves.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_panel_tri(DevPlan P, const double *__restrict__ Hblk, double *__restrict__ Lv, int pn0,
                                                    const double *__restrict__ lambda_p, int *__restrict__ fail_flag, int n_pn, int ride0, int n_ride, int n_real) {
  __shared__ __attribute__((aligned(16))) double T[PM * (PM + 1) / 2 * 36];
  if (NW == 16) {
    // workgroups beyond the level's panels: riders (early accumulate work of later levels, while the pivot chains run)
    __shared__ __attribute__((aligned(16))) double ride_smem[NW == 16 ? 16 * 360 : 2];
    if ((int)blockIdx.x >= n_pn) {
      // every XCD takes a contiguous range of the items (n_pn is padded to a multiple of 8 by the launcher when riders exist, so
      // the XCD of a rider workgroup is (blockIdx - n_pn) & 7): items of neighbouring targets share their source blocks
      const int nwg = (int)gridDim.x - n_pn;
      ride_items16(P, Hblk, Lv, lambda_p, ride0, n_ride, P.ride_xcd ? xcd_contiguous((int)blockIdx.x - n_pn, nwg) : (int)blockIdx.x - n_pn, ride_smem);
      return;
    }
  }
  const long long t_begin = __builtin_readcyclecounter();
  if ((int)blockIdx.x >= n_real) return;                       // padding between the panels and the riders
  // (NW == 8: the throughput form of a wide level -- panels in width order in a full sweep, PanelPlan::tri_order; a partial sweep launches a task range)
  const int pn = (NW == 8 && !P.task_dirty) ? P.pp.tri_order[pn0 + blockIdx.x] : pn0 + (int)blockIdx.x;
  const PanelDesc dsc = P.pp.pdesc[pn];
  if (!task_runs(P, dsc.task)) return;
  const int m = dsc.m;
  const int64_t tri0 = (int64_t)pn * (PM * PM);
  const int *__restrict__ tb = P.pp.ptri_blk + tri0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, r = lane - 6 * g;
  const bool lane_on = lane < 60;
  const int gid = wave * 10 + g;
  const double lambda = *lambda_p;
  const int npair = m * (m + 1) / 2;
  for (int gq = gid; lane_on && gq < npair; gq += NW * 10) {
    const int rr = PAIR_A[gq], kk = PAIR_B[gq];
    const int sc = P.pp.ptri_src[tri0 + rr * PM + kk];
    const double *base = sc >= 0 ? Lv + 36 * (int64_t)sc : (sc <= -2 ? Hblk + 36 * (int64_t)(-2 - sc) : Lv + 36 * (int64_t)P.zero_blk);
    Row6 x = load_row(base + 6 * r);
    if (sc <= -2 && rr == kk) {                         // setLambda on a diagonal block that comes straight from H
#pragma unroll
      for (int c = 0; c < 6; ++c) x.v[c] += (c == r) ? lambda : 0.0;
    }
    store_row(&T[36 * gq + 6 * r], x);                  // PAIR order == packed order
  }
  __syncthreads();
  // (Two separate loops with matching barrier counts: the register allocator then sees max(pivot chain, worker),
  //  not their union, which at 16 waves per workgroup -- 128 VGPRs -- is the difference between fitting and spilling.)
  const int n = 6 * m, nJ = (n + 15) >> 4;
  // profiling hook (FGO_TRI_PROF): stamps of the pivot wave of a single-panel launch, 5 per column + 4 for the kernel
  long long *__restrict__ stamp = (P.prof_tri && n_real == 1 && m == PM) ? reinterpret_cast<long long *>(P.partial) : nullptr;
  if (stamp && threadIdx.x == 0) { stamp[0] = t_begin; stamp[1] = __builtin_readcyclecounter(); }
  if (wave == 0) {
    constexpr int HMAX = (PM + 9) / 10;
    for (int k = 0; k < m; ++k) {
      if (stamp && lane == 0) stamp[8 + 5 * k] = __builtin_readcyclecounter();
      // rows rr = k + g (+10 h): update with column k-1, then the diagonal block goes back to LDS for the 6x6 factor.
      // The update runs over the inner index j on the outside, so that the six accumulators of a row advance together
      // (six independent FMA chains instead of six dot products one after the other: the wave is a latency chain here).
      Row6 acc[HMAX];
#pragma unroll
      for (int h = 0; h < HMAX; ++h) {
        const int rr = k + g + 10 * h;
        if (lane_on && rr < m) {
          acc[h] = load_row(&T[TRI(rr, k) + 6 * r]);
          if (k > 0) {
            const Row6 a = load_row(&T[TRI(rr, k - 1) + 6 * r]);
            const double *__restrict__ B = &T[TRI(k, k - 1)];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
              for (int c = 0; c < 6; ++c) acc[h].v[c] = fma(-a.v[j], B[6 * c + j], acc[h].v[c]);
          }
        }
      }
      if (lane_on && g == 0) store_row(&T[TRI(k, k) + 6 * r], acc[0]);
      __builtin_amdgcn_wave_barrier();
      if (stamp && lane == 0) stamp[8 + 5 * k + 1] = __builtin_readcyclecounter();
      double Lk[21], invd[6];
      const bool ok = chol6_lds(&T[TRI(k, k)], Lk, invd);
      if (!ok && lane == 0) atomicOr(fail_flag, 1);
      __builtin_amdgcn_wave_barrier();                    // everybody has read the diagonal block before it is overwritten
      if (stamp && lane == 0) stamp[8 + 5 * k + 2] = __builtin_readcyclecounter() + (long long)(Lk[0] == 12345.678);
#pragma unroll
      for (int h = 0; h < HMAX; ++h) {
        const int rr = k + g + 10 * h;
        if (lane_on && rr < m) {
          // (the rows of the diagonal block take the same path: row r of D L^-T IS row r of L -- equal to the wave's Lk up to
          //  the last bit, with the entries right of the diagonal forced to zero; no 36-way select, no second store pattern)
          Row6 x = trsm_row(acc[h], Lk, invd);
          if (rr == k) {
#pragma unroll
            for (int c = 0; c < 6; ++c) x.v[c] = (c <= r) ? x.v[c] : 0.0;
          }
          store_row(&T[TRI(rr, k) + 6 * r], x);
        }
      }
      if (stamp && lane == 0) stamp[8 + 5 * k + 3] = __builtin_readcyclecounter();
      __syncthreads();
      if (stamp && lane == 0) stamp[8 + 5 * k + 4] = __builtin_readcyclecounter();
    }
  } else {
    const int nn = lane & 15, q4 = lane >> 4;
    const int ntile = nJ * (nJ + 1) / 2;                     // lower tiles incl. the diagonal ones, PAIR order (I, K), K <= I
    // EXCL (the 16-wave instantiation): the waves that share a SIMD with the pivot wave (4, 8, 12: waves are dealt round-robin
    // to the four SIMDs) take no tiles, so the pivot chain -- the critical path of the kernel -- has a SIMD's issue slots to
    // itself.  Measured with FGO_TRI_PROF (tools/tri_prof.py): a column of the chain costs ~3 400 cycles when the pivot wave
    // shares its SIMD with three worker waves.
    constexpr bool EXCL = NW >= 16;
    constexpr int NWORK = EXCL ? NW - NW / 4 : NW - 1;
    constexpr int NT = (NJMAX * (NJMAX + 1) / 2 + NWORK - 1) / NWORK;
    const bool idle = EXCL && (wave & 3) == 0;
    const int aw = EXCL ? wave - 1 - (wave >> 2) : wave - 1;
    // per owned tile: LDS offsets of the V rows feeding the A / B operands and of the four result rows, packed triangle:
    // element (scalar row i, block column k, in-block column c) sits at  base(i) + 36 k + c,  base(i) = TRI(i / 6, 0) + 6 (i % 6)
    d4_t C[NT];
    bool own[NT];
    int offA[NT], rrA[NT], offB[NT], rrB[NT], cj[NT], offE[NT][4], rrE[NT][4], tI[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int p = aw + NWORK * u;
      own[u] = !idle && p < ntile;
      const int I = PAIR_A[own[u] ? p : 0], K = PAIR_B[own[u] ? p : 0];
      tI[u] = I;
      const int iA = 16 * I + nn, iB = 16 * K + nn;
      rrA[u] = iA < n ? iA / 6 : -1; offA[u] = TRI(iA / 6, 0) + (iA % 6) * 6;
      rrB[u] = iB < n ? iB / 6 : -1; offB[u] = TRI(iB / 6, 0) + (iB % 6) * 6;
      cj[u] = iB % 6;
      C[u] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int i = 16 * I + q4 + 4 * r4;
        rrE[u][r4] = i < n ? i / 6 : -1; offE[u][r4] = TRI(i / 6, 0) + (i % 6) * 6;
        if (own[u] && rrE[u][r4] >= 0 && rrB[u] >= 0 && rrE[u][r4] >= rrB[u]) C[u][r4] = T[offE[u][r4] + 36 * rrB[u] + cj[u]];
      }
    }
    // phase 0 has nothing for the workers (columns 0 and 1 are staged by the initial load): LDS is not written before
    // the first barrier, so reading the initial values above needs no extra barrier
    for (int k = 0; k < m; ++k) {
      if (k > 0) {
#pragma unroll
        for (int u = 0; u < NT; ++u)
          if (own[u] && 16 * tI[u] + 15 >= 6 * k) {             // tile reaches into the part right of column k-1
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
              const int c = 4 * kc + q4;
              double a = 0.0, b = 0.0;
              if (c < 6) {
                if (rrA[u] > k - 1) a = -T[offA[u] + (k - 1) * 36 + c];
                if (rrB[u] > k - 1) b = T[offB[u] + (k - 1) * 36 + c];
              }
              C[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C[u], 0, 0, 0);
            }
          }
#pragma unroll
        for (int u = 0; u < NT; ++u)
          if (own[u] && rrB[u] == k + 1) {                      // stage column k+1 for the phase after next
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
              if (rrE[u][r4] >= k + 1) T[offE[u][r4] + (k + 1) * 36 + cj[u]] = C[u][r4];
          }
      }
      __syncthreads();
    }
  }
  if (stamp && threadIdx.x == 0) stamp[2] = __builtin_readcyclecounter();
  for (int gq = gid; lane_on && gq < npair; gq += NW * 10) {
    const int rr = PAIR_A[gq], kk = PAIR_B[gq];
    const int t = tb[rr * PM + kk];
    if (t >= 0) store_row(Lv + 36 * (int64_t)t + 6 * r, load_row(&T[36 * gq + 6 * r]));
  }
  // ---- the factored triangle once more, as a dense scalar matrix cut into 16x16 tiles in MFMA A-operand order
  // (lane l <-> element [l & 15][4 kc + (l >> 4)] of the tile, i.e. column-major): strictly-lower tiles NEGATED,
  // diagonal tiles INVERTED.  Only the tiles of the nJ tile rows the panel really has are written / read.
  auto Ls = [&](int i, int j) -> double {
    if (i >= n || j >= n) return (i == j) ? 1.0 : 0.0;          // identity padding up to the tile boundary
    const int rr = i / 6, kk = j / 6;
    if (rr < kk) return 0.0;
    return T[TRI(rr, kk) + (i - 6 * rr) * 6 + (j - 6 * kk)];
  };
  double *__restrict__ tp = P.pp.ptop + (int64_t)dsc.top * 256;
  for (int e = threadIdx.x; e < (nJ * (nJ - 1) / 2) * 256; e += NW * 64) {   // tiles (J, I), I < J < nJ, are the first nJ (nJ-1) / 2
    const int tile = e >> 8, kc = (e >> 6) & 3, l = e & 63;
    const int J = PAIR_A[tile] + 1, I = PAIR_B[tile];
    tp[e] = -Ls(16 * J + (l & 15), 16 * I + 4 * kc + (l >> 4));
  }
  __shared__ double Dt[NJMAX * 256];                            // the diagonal tiles, staged in LDS (51 KB with the triangle image)
  for (int e = threadIdx.x; e < nJ * 256; e += NW * 64) Dt[e] = Ls(16 * (e >> 8) + ((e >> 4) & 15), 16 * (e >> 8) + (e & 15));
  __syncthreads();
  if ((int)threadIdx.x < 16 * nJ) {                             // column c of the inverse of diagonal tile J, straight to memory
    const int J = threadIdx.x >> 4, c = threadIdx.x & 15;
    double xc[16];
    // column c of Dt^-1 by forward substitution, right-looking: the 16 reciprocals first (independent), then per step one
    // multiply and the updates of the rows below (chain: 2 operations per row instead of a dot product and a division)
    const double *__restrict__ D = &Dt[J * 256];
#pragma unroll
    for (int i = 0; i < 16; ++i) xc[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      xc[i] *= 1.0 / D[i * 17];                                   // (the reciprocal does not depend on the chain)
#pragma unroll
      for (int i2 = 0; i2 < 16; ++i2) if (i2 > i) xc[i2] = fma(-D[i2 * 16 + i], xc[i], xc[i2]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) tp[(NLT + J) * 256 + 16 * c + i] = xc[i];
  }
  if (stamp && threadIdx.x == 0) { stamp[3] = __builtin_readcyclecounter(); stamp[4] = wall_clock64(); }
}

// ---- THROUGHPUT form of the triangle kernel for the WIDE levels (more panels than k_panel_tri<16> has CUs for): ONE WAVE per
// panel, no workgroup barrier anywhere.  k_panel_tri spends a panel's time in a pivot chain on one wave while the other waves
// of its workgroup wait at barriers -- right where one or two panels exist per level, wasteful where there are thousands
// (cfg 2 level 1: 2 047 panels, two 8-wave workgroups per CU, four rounds of ~40 us).  Here a wave keeps the whole trailing
// matrix of its panel as 21 f64 MFMA accumulator tiles (168 registers) and walks the block columns on its own:
//   column k:  the accumulated updates of its 6 scalar columns leave the tiles through a 4.6 KB LDS image (C layout ->
//              one scalar row per lane), are added to the column's own values (a 48-byte row of a block of L or H per lane,
//              requested one column ahead), 6x6 Cholesky + row TRSM as in k_panel_tri, the finished rows go to L, to the
//              operand tiles `ptop` (strictly-lower tiles negated, in A-operand order) and back to the LDS image, from
//              which every lane picks its MFMA operands:  C(I, K) -= V_I V_K^T, two v_mfma_f64_16x16x4 per live tile.
// Diagonal tiles are collected in LDS and inverted at the end (16 lanes per tile, right-looking substitution).  Same contract
// as k_panel_tri (L blocks + ptop), so k_panel_rows, the solves and the backward kernels do not know which one ran.
constexpr int tri1_tile(int I, int K) { return I * (I + 1) / 2 + K; }
constexpr int TRI1_NR = 16 * NJMAX;                               // scalar rows incl. the padding of the last tile row
struct Tri1Ctx {                                                  // wave-uniform / per-lane constants of a panel
  const double *Hblk; double *Lv; double *tp; int64_t zero_blk;
  int m, n, nJ, lane, nn, q;
  double lambda;
  int rr[2], rho[2];
  bool rv[2], rpad[2];
};
__device__ __forceinline__ const double *tri1_src_row(const Tri1Ctx &X, const int *__restrict__ tsrc, int h, int k) {
  const int sc = tsrc[X.rr[h] * PM + k];
  const double *base = sc >= 0 ? X.Lv + 36 * (int64_t)sc : (sc <= -2 ? X.Hblk + 36 * (int64_t)(-2 - sc) : X.Lv + 36 * X.zero_blk);
  return base + 6 * X.rho[h];
}
// block column K of a panel, K a compile-time constant: every tile index below is static, so the 21 accumulator tiles stay
// in registers (a run-time column index made the compiler address them through v_readlane / v_accvgpr chains: 75 us per panel)
template <int K>
__device__ __forceinline__ void tri1_step(const Tri1Ctx &X, d4_t (&C)[NJMAX * (NJMAX + 1) / 2], Row6 (&sN)[2], bool &ok_all,
                                          double *__restrict__ W, double *__restrict__ Dt, double *__restrict__ D6,
                                          const int *__restrict__ tsrc, const int *__restrict__ tblk) {
  constexpr int c0 = 6 * K, K0 = c0 >> 4, K1 = (c0 + 5) >> 4;       // tile columns the block column lies in
  const int lane = X.lane, nn = X.nn, q = X.q;
  Row6 s[2] = {sN[0], sN[1]};
  if (K + 1 < X.m) {                                               // request the next column's rows now: they arrive behind this column's chain
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (X.rv[h] && X.rr[h] >= K + 1) sN[h] = load_row(tri1_src_row(X, tsrc, h, K + 1));
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)                                      // setLambda on a diagonal block that comes straight from H
    if (X.rv[h] && X.rr[h] == K && tsrc[K * PM + K] <= -2) {
#pragma unroll
      for (int c = 0; c < 6; ++c) s[h].v[c] += (c == X.rho[h]) ? X.lambda : 0.0;
    }
  if constexpr (K > 0) {
    // ---- the updates of columns < K leave the accumulator tiles: element (row 16 I + q + 4 r4, column 16 Kt + nn)
#pragma unroll
    for (int I = 0; I < NJMAX; ++I)
#pragma unroll
      for (int Kt = 0; Kt < NJMAX; ++Kt)
        if ((Kt == K0 || Kt == K1) && Kt <= I && 16 * I + 15 >= c0 && I < X.nJ) {
          const int c = 16 * Kt + nn - c0;
          if (c >= 0 && c < 6) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) W[(16 * I + q + 4 * r4) * 6 + c] = C[tri1_tile(I, Kt)][r4];
          }
        }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (X.rv[h] && X.rr[h] >= K) {
        const Row6 e = load_row(&W[(lane + 64 * h) * 6]);
#pragma unroll
        for (int c = 0; c < 6; ++c) s[h].v[c] += e.v[c];
      }
  }
  // ---- 6x6 factor of the diagonal block, TRSM of the rows below
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (X.rv[h] && X.rr[h] == K) store_row(&D6[6 * X.rho[h]], s[h]);
  __builtin_amdgcn_wave_barrier();
  double Lk[21], invd[6];
  ok_all = chol6_lds(D6, Lk, invd) && ok_all;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h == 1 && 16 * X.nJ <= 64) continue;                       // (wave-uniform: no rows 64 .. 95)
    const int i = lane + 64 * h;
    Row6 v = {{0, 0, 0, 0, 0, 0}};
    if (X.rv[h] && X.rr[h] >= K) {
      v = trsm_row(s[h], Lk, invd);
      if (X.rr[h] == K) {
#pragma unroll
        for (int c = 0; c < 6; ++c) v.v[c] = (c <= X.rho[h]) ? v.v[c] : 0.0;
      }
      const int t = tblk[X.rr[h] * PM + K];
      if (t >= 0) store_row(X.Lv + 36 * (int64_t)t + 6 * X.rho[h], v);
    }
    if (X.rpad[h] && i >= c0) {
      store_row(&W[i * 6], v);                                      // (rows of the padding carry zeros)
      const int J = i >> 4, a = i & 15;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        constexpr int dummy = 0; (void)dummy;
        const int col = c0 + c, I = col >> 4, b = col & 15;         // constants
        if (J > I) X.tp[(J * (J - 1) / 2 + I) * 256 + 16 * b + a] = -v.v[c];
        else if (J == I && X.rv[h]) Dt[J * 256 + 16 * a + b] = v.v[c];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  // ---- rank-6 update of the trailing tiles: operands from the column image (rows at or above the diagonal block -> 0)
  if (K + 1 < X.m) {
    constexpr int I0 = (c0 + 6) >> 4;
    double av[NJMAX][2];
#pragma unroll
    for (int I = 0; I < NJMAX; ++I)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        const int row = 16 * I + nn, c = 4 * kc + q;
        av[I][kc] = (I >= I0 && I < X.nJ && c < 6 && row >= c0 + 6 && row < X.n) ? W[row * 6 + c] : 0.0;
      }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int I = 0; I < NJMAX; ++I)
#pragma unroll
        for (int Kt = 0; Kt < NJMAX; ++Kt)
          if (Kt <= I && Kt >= I0 && I < X.nJ)
            C[tri1_tile(I, Kt)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[I][kc], av[Kt][kc], C[tri1_tile(I, Kt)], 0, 0, 0);
    __builtin_amdgcn_wave_barrier();                                // (the operand reads are done before the next column's image is written)
  }
}
template <int K>
__device__ __forceinline__ void tri1_steps(const Tri1Ctx &X, d4_t (&C)[NJMAX * (NJMAX + 1) / 2], Row6 (&sN)[2], bool &ok_all,
                                           double *__restrict__ W, double *__restrict__ Dt, double *__restrict__ D6,
                                           const int *__restrict__ tsrc, const int *__restrict__ tblk) {
  if constexpr (K < PM) {
    if (K < X.m) {
      tri1_step<K>(X, C, sN, ok_all, W, Dt, D6, tsrc, tblk);
      tri1_steps<K + 1>(X, C, sN, ok_all, W, Dt, D6, tsrc, tblk);
    }
  }
}
__device__ __forceinline__ void tri1_body(const DevPlan &P, const double *__restrict__ Hblk, double *__restrict__ Lv, int pn0,
                                          const double *__restrict__ lambda_p, int *__restrict__ fail_flag) {
  __shared__ __attribute__((aligned(16))) double W[TRI1_NR * 6];   // column image: row i at W + 6 i
  __shared__ __attribute__((aligned(16))) double Dt[NJMAX * 256];  // diagonal tiles, row-major
  __shared__ __attribute__((aligned(16))) double D6[36];
  __shared__ int tsrc[PM * PM], tblk[PM * PM];
  const int pn = !P.task_dirty ? P.pp.tri_order[pn0 + blockIdx.x] : pn0 + (int)blockIdx.x;      // (full sweep: panels in width order)
  const PanelDesc dsc = P.pp.pdesc[pn];
  if (!task_runs(P, dsc.task)) return;
  Tri1Ctx X;
  X.Hblk = Hblk; X.Lv = Lv; X.zero_blk = P.zero_blk;
  X.m = dsc.m; X.n = 6 * dsc.m; X.nJ = (X.n + 15) >> 4;
  X.lane = threadIdx.x; X.nn = X.lane & 15; X.q = X.lane >> 4;
  X.lambda = *lambda_p;
  X.tp = P.pp.ptop + (int64_t)dsc.top * 256;
  const int lane = X.lane, n = X.n, nJ = X.nJ;
#pragma unroll
  for (int e = 0; e < PM * PM / 64; ++e) {
    tsrc[lane + 64 * e] = P.pp.ptri_src[(int64_t)pn * PM * PM + lane + 64 * e];
    tblk[lane + 64 * e] = P.pp.ptri_blk[(int64_t)pn * PM * PM + lane + 64 * e];
  }
  for (int e = lane; e < NJMAX * 256; e += 64) {                    // zero above the diagonal, identity on the padding
    const int i = (e >> 4) & 15, j = e & 15, J = e >> 8;
    Dt[e] = (i == j && 16 * J + i >= n) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {                                     // the lane's scalar rows: i = lane and (lanes 0 .. 31) i = 64 + lane
    const int i = lane + 64 * h;
    X.rv[h] = i < n; X.rpad[h] = i < 16 * nJ && i < TRI1_NR;
    X.rr[h] = i / 6; X.rho[h] = i - 6 * X.rr[h];
  }
  __builtin_amdgcn_wave_barrier();
  d4_t C[NJMAX * (NJMAX + 1) / 2];
#pragma unroll
  for (int p = 0; p < NJMAX * (NJMAX + 1) / 2; ++p) C[p] = d4_t{0.0, 0.0, 0.0, 0.0};
  Row6 sN[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    sN[h] = Row6{{0, 0, 0, 0, 0, 0}};
    if (X.rv[h]) sN[h] = load_row(tri1_src_row(X, tsrc, h, 0));
  }
  bool ok_all = true;
  tri1_steps<0>(X, C, sN, ok_all, W, Dt, D6, tsrc, tblk);
  if (!ok_all && lane == 0) atomicOr(fail_flag, 1);
  __builtin_amdgcn_wave_barrier();
  // ---- inverses of the diagonal tiles: lane (J, c) computes column c of tile J's inverse (right-looking substitution)
#pragma unroll
  for (int pass = 0; pass < (NJMAX + 3) / 4; ++pass) {
    const int J = 4 * pass + X.q, c = X.nn;
    if (J < nJ) {
      const double *__restrict__ D = &Dt[J * 256];
      double xc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) xc[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        xc[i] *= 1.0 / D[i * 17];
#pragma unroll
        for (int i2 = 0; i2 < 16; ++i2) if (i2 > i) xc[i2] = fma(-D[i2 * 16 + i], xc[i], xc[i2]);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) X.tp[(NLT + J) * 256 + 16 * c + i] = xc[i];
    }
  }
}
__global__ __launch_bounds__(64) void k_panel_tri1(DevPlan P, const double *__restrict__ Hblk, double *__restrict__ Lv, int pn0,
                                                   const double *__restrict__ lambda_p, int *__restrict__ fail_flag) {
  tri1_body(P, Hblk, Lv, pn0, lambda_p, fail_flag);
}

// Rider role of a k_panel_rows launch of a narrow level (one wave per workgroup): the wave's 10 lane groups stride one item.
__device__ __forceinline__ void ride_item_wave(const DevPlan &P, const double *__restrict__ Hblk, double *__restrict__ Lv,
                                               const double *__restrict__ lambda_p, int item, double *__restrict__ tile) {
  const RideItem it = P.ride_items[item];
  if (P.task_dirty && !P.task_dirty[it.task]) return;
  const int lane = threadIdx.x;
  const int g = lane / 6, r = lane - 6 * g;
  if (lane < 60) {
    Row6 acc = {{0, 0, 0, 0, 0, 0}};
    if (g == 0) acc = it.first ? load_A_row(P, Hblk, it.t, r, *lambda_p) : load_row(Lv + 36 * (int64_t)it.t + 6 * r);
    apply_ops(P, Lv, acc, g, r, it.o0 + g, it.o0 + it.n, 10, tile);
#pragma unroll
    for (int c = 0; c < 6; ++c) tile[6 * lane + c] = acc.v[c];
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < 36) {
    double sum = 0;
#pragma unroll
    for (int q = 0; q < 10; ++q) sum += tile[q * 36 + lane];
    Lv[36 * (int64_t)it.t + lane] = sum;
  }
}

// Off-triangle rows of a panel: X <- U T^-T for all rows at once, done as the transposed problem
// Y = T^-1 U^T on 16-wide tiles with v_mfma_f64_16x16x4_f64: one wave owns ROW_SETS x 16 scalar rows (the N dimension),
// Y_J = Dinv_J (U^T_J - sum_{I<J} T_JI Y_I).  The f64 MFMA result layout (row = (lane >> 4) + 4 reg) makes the
// result tile Y_I directly usable as the B operand of the next products, so the 6 tiles never leave registers; the
// A operands stream from the panel's operand buffer (written by k_panel_tri), one coalesced 512-byte load each.
// With x != nullptr the right-hand side rides along as scalar row R6 of the panel (x holds b - external sums for
// the panel's columns, left there by fwd_ext_column): the in-panel forward substitution costs nothing extra.
template <bool BYC = false>
__device__ __forceinline__ void panel_rows_body(const DevPlan &P, const double *__restrict__ Hblk, double *__restrict__ Lv, int chunk0,
                                                double *__restrict__ x, int n_chunks, const double *__restrict__ lambda_p, int ride0) {
  if ((int)blockIdx.x >= n_chunks) {                               // riders (symbolic.cpp)
    __shared__ __attribute__((aligned(16))) double ride_tile[360];
    ride_item_wave(P, Hblk, Lv, lambda_p, ride0 + (int)blockIdx.x - n_chunks, ride_tile);
    return;
  }
  const int ci = chunk0 + xcd_contiguous(blockIdx.x, n_chunks);
  const RowChunk rc = P.pp.rchunks[ci];
  const int lane = threadIdx.x, nn = lane & 15, q = lane >> 4;
  // BYC (narrow levels, 16-column panels): the source codes of this lane's scalar row from the table laid out by chunk -- addressed without the
  // descriptor, so both requests travel together (fgo_structure.cpp "rchunk_src")
  int scq[4 * NJMAX];
  if constexpr (BYC) {
    const int *__restrict__ rsq = P.pp.rchunk_src + ((int64_t)(ci - P.pp.rchunk_src0) * 16 + nn) * PANEL_MAX;
#pragma unroll
    for (int e = 0; e < 4 * NJMAX; ++e) scq[e] = rsq[(16 * (e >> 2) + q + 4 * (e & 3)) / 6];
  }
  if (!task_runs(P, rc.task)) return;
  const int m = rc.m;
  const int n = 6 * m, nJ = (n + 15) >> 4;
  const int R6 = rc.R6;
  const int *__restrict__ cols = P.task_cols + rc.cols0;
  const double *__restrict__ tp = P.pp.ptop + (int64_t)rc.top * 256;
  constexpr int RS = ROW_SETS;                                   // sets of 16 scalar rows handled by this wave
  constexpr int NE = 4 * NJMAX;                                  // elements of U^T per lane and set: NJMAX tiles x 4 registers
  // gather U^T in MFMA C layout: (lane, J, r) <-> scalar column c = 16 J + (lane >> 4) + 4 r of scalar row s.
  // Two memory round trips in all: the source codes, then the values (branch-free; absent -> the zero block).
  int sc[RS][NE], rho[RS];
  int64_t rowoff[RS];
  bool valid[RS], rhs[RS];
#pragma unroll
  for (int u = 0; u < RS; ++u) {
    const int s = rc.s0 + 16 * u + nn;                           // scalar row within the panel's off-triangle rows
    valid[u] = s < R6;
    rhs[u] = x != nullptr && s == R6;
    const int br = valid[u] ? s / 6 : 0;
    rho[u] = valid[u] ? s - 6 * br : 0;
    rowoff[u] = (int64_t)(rc.prow0 + br) * PM;
    const int *__restrict__ rs = P.pp.prow_src + rowoff[u];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int c = 16 * (e >> 2) + q + 4 * (e & 3);
      if constexpr (BYC) sc[u][e] = ((valid[u] || rhs[u]) && c < n) ? scq[e] : -1;
      else sc[u][e] = (valid[u] && c < n) ? rs[c / 6] : ((rhs[u] && c < n) ? cols[c / 6] : -1);
    }
  }
  d4_t Y[RS][NJMAX];
#pragma unroll
  for (int u = 0; u < RS; ++u)
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int c = 16 * (e >> 2) + q + 4 * (e & 3);
      const int k = c / 6;
      const int code = sc[u][e];
      const double *base = code >= 0 ? Lv + 36 * (int64_t)code : (code <= -2 ? Hblk + 36 * (int64_t)(-2 - code) : Lv + 36 * (int64_t)P.zero_blk);
      if (rhs[u] && c < n) base = x + 6 * (int64_t)code;      // rho == 0 on this lane
      Y[u][e >> 2][e & 3] = base[6 * rho[u] + (c - 6 * k)];
    }
  // operand tiles stream in one tile row ahead of the MFMAs that use them (tile row J: J negated lower tiles + its
  // inverted diagonal tile = 4 (J + 1) doubles per lane); every tile feeds the MFMAs of all RS row sets
  auto load_tile_row = [&](int J, double (&A)[4 * NJMAX]) {
#pragma unroll
    for (int I = 0; I < NJMAX; ++I)
      if (I < J) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) A[4 * I + kc] = tp[((J * (J - 1) / 2 + I) * 4 + kc) * 64 + lane];
      }
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) A[4 * (NJMAX - 1) + kc] = tp[((NLT + J) * 4 + kc) * 64 + lane];   // slot NJMAX-1 is never a lower tile of row J <= NJMAX-1
  };
  double Abuf[2][4 * NJMAX];
  load_tile_row(0, Abuf[0]);
#pragma unroll
  for (int J = 0; J < NJMAX; ++J)
    if (J < nJ) {
      double (&A)[4 * NJMAX] = Abuf[J & 1];
      if (J + 1 < nJ) load_tile_row(J + 1, Abuf[(J + 1) & 1]);
#pragma unroll
      for (int u = 0; u < RS; ++u) {
        d4_t acc = Y[u][J];
#pragma unroll
        for (int I = 0; I < J; ++I)
#pragma unroll
          for (int kc = 0; kc < 4; ++kc)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[4 * I + kc], Y[u][I][kc], acc, 0, 0, 0);
        d4_t z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) z = __builtin_amdgcn_mfma_f64_16x16x4f64(A[4 * (NJMAX - 1) + kc], acc[kc], z, 0, 0, 0);
        Y[u][J] = z;
      }
    }
#pragma unroll
  for (int u = 0; u < RS; ++u) {
    const int *__restrict__ rb = P.pp.prow_blk + rowoff[u];
    int tt[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int c = 16 * (e >> 2) + q + 4 * (e & 3);
      tt[e] = (valid[u] && c < n) ? rb[c / 6] : -1;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int c = 16 * (e >> 2) + q + 4 * (e & 3);
      if (tt[e] >= 0) Lv[36 * (int64_t)tt[e] + 6 * rho[u] + (c - 6 * (c / 6))] = Y[u][e >> 2][e & 3];
      if (rhs[u] && c < n) x[6 * (int64_t)sc[u][e] + (c - 6 * (c / 6))] = Y[u][e >> 2][e & 3];
    }
  }
}

__global__ __launch_bounds__(64) void k_panel_rows(DevPlan P, const double *__restrict__ Hblk, double *__restrict__ Lv, int chunk0,
                                                   double *__restrict__ x, int n_chunks, const double *__restrict__ lambda_p, int ride0) {
  panel_rows_body<false>(P, Hblk, Lv, chunk0, x, n_chunks, lambda_p, ride0);
}
__global__ __launch_bounds__(64) void k_panel_rows_byc(DevPlan P, const double *__restrict__ Hblk, double *__restrict__ Lv, int chunk0,
                                                       double *__restrict__ x, int n_chunks, const double *__restrict__ lambda_p, int ride0) {
  panel_rows_body<true>(P, Hblk, Lv, chunk0, x, n_chunks, lambda_p, ride0);
}
// ------------------------------------------------------------------------------------------------
// Triangular solves on x (in place, permuted block order).  One workgroup per task, same lane mapping:
// lane group g takes one block of the row/column list, lane r one row (column) of it.
// forward: y_k = L_kk^-1 (b_k - sum_{j in row k} L_kj y_j)
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_solve_fwd(DevPlan P, const double *__restrict__ Lv, double *__restrict__ x,
                                                       int task0) {
  __shared__ double sred[NW * 60];
  const int task = task0 + blockIdx.x;
  if (!task_runs(P, task)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, r = lane - 6 * g;
  const int c_begin = P.task_ptr[task], c_end = P.task_ptr[task + 1];
  for (int ci = c_begin; ci < c_end; ++ci) {
    const int k = P.task_cols[ci];
    const int64_t r0 = (P.dist && k >= P.top_col0) ? P.top_row0[k - P.top_col0] : P.rowptr[k], r1 = P.rowptr[k + 1];
    // wave 0, lanes 0..5 prefetch their row of L_kk and b_k while the sums are formed
    Row6 ld = {{0, 0, 0, 0, 0, 0}};
    double rhs = 0;
    if (threadIdx.x < 6) { ld = load_row(Lv + 36 * P.colptr[k] + 6 * r); rhs = x[6 * (int64_t)k + r]; }
    double acc = 0;
    if (lane < 60)
    {
      // 4 independent (index -> block row, y) chains in flight per lane; fixed summation order
      int64_t e = r0 + wave * 10 + g;
      constexpr int64_t ST = NW * 10;
      for (; e + 3 * ST < r1; e += 4 * ST) {
        const int b0i = P.row_blk[e], b1i = P.row_blk[e + ST], b2i = P.row_blk[e + 2 * ST], b3i = P.row_blk[e + 3 * ST];
        const int c0i = P.row_col[e], c1i = P.row_col[e + ST], c2i = P.row_col[e + 2 * ST], c3i = P.row_col[e + 3 * ST];
        const Row6 l0 = load_row(Lv + 36 * (int64_t)b0i + 6 * r), l1 = load_row(Lv + 36 * (int64_t)b1i + 6 * r);
        const Row6 l2 = load_row(Lv + 36 * (int64_t)b2i + 6 * r), l3 = load_row(Lv + 36 * (int64_t)b3i + 6 * r);
        const Row6 y0 = load_row(x + 6 * (int64_t)c0i), y1 = load_row(x + 6 * (int64_t)c1i);
        const Row6 y2 = load_row(x + 6 * (int64_t)c2i), y3 = load_row(x + 6 * (int64_t)c3i);
        acc += l0.v[0] * y0.v[0] + l0.v[1] * y0.v[1] + l0.v[2] * y0.v[2] + l0.v[3] * y0.v[3] + l0.v[4] * y0.v[4] + l0.v[5] * y0.v[5];
        acc += l1.v[0] * y1.v[0] + l1.v[1] * y1.v[1] + l1.v[2] * y1.v[2] + l1.v[3] * y1.v[3] + l1.v[4] * y1.v[4] + l1.v[5] * y1.v[5];
        acc += l2.v[0] * y2.v[0] + l2.v[1] * y2.v[1] + l2.v[2] * y2.v[2] + l2.v[3] * y2.v[3] + l2.v[4] * y2.v[4] + l2.v[5] * y2.v[5];
        acc += l3.v[0] * y3.v[0] + l3.v[1] * y3.v[1] + l3.v[2] * y3.v[2] + l3.v[3] * y3.v[3] + l3.v[4] * y3.v[4] + l3.v[5] * y3.v[5];
      }
      for (; e < r1; e += ST) {
        const Row6 l = load_row(Lv + 36 * (int64_t)P.row_blk[e] + 6 * r);
        const Row6 y = load_row(x + 6 * (int64_t)P.row_col[e]);
        acc += l.v[0] * y.v[0] + l.v[1] * y.v[1] + l.v[2] * y.v[2] + l.v[3] * y.v[3] + l.v[4] * y.v[4] + l.v[5] * y.v[5];
      }
    }
    if (lane < 60) sred[wave * 60 + lane] = acc;
    __syncthreads();
    if (wave == 0) {                       // whole wave executes the shuffles; lanes 0..5 hold the result
      double s = rhs;
      if (lane < 6)
        for (int q = 0; q < NW * 10; ++q) s -= sred[q * 6 + lane];
      double y = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double yi = __shfl(s, i, WAVE) / __shfl(ld.v[i], i, WAVE);   // y_i = s_i / L_ii
        if (lane == i) y = yi;
        if (lane > i && lane < 6) s -= ld.v[i] * yi;                        // s_r -= L_ri y_i
      }
      if (lane < 6) x[6 * (int64_t)k + lane] = y;
    }
    __syncthreads();
  }
}
// backward: x_k = L_kk^-T (y_k - sum_{i in col k} L_ik^T x_i), tasks' columns in descending order
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_solve_bwd(DevPlan P, const double *__restrict__ Lv, double *__restrict__ x,
                                                       int task0) {
  __shared__ double sred[NW * 60];
  const int task = task0 + blockIdx.x;
  if (!task_runs(P, task)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, cc = lane - 6 * g;
  const int c_begin = P.task_ptr[task], c_end = P.task_ptr[task + 1];
  for (int ci = c_end - 1; ci >= c_begin; --ci) {
    const int k = P.task_cols[ci];
    const int64_t b0 = P.colptr[k] + 1, b1 = P.colptr[k + 1];
    // lanes 0..5 of wave 0 prefetch COLUMN cc of L_kk (row cc of L_kk^T) and y_k
    double lcol[6] = {0, 0, 0, 0, 0, 0};
    double rhs = 0;
    if (threadIdx.x < 6) {
      const double *Ld = Lv + 36 * P.colptr[k];
#pragma unroll
      for (int m = 0; m < 6; ++m) lcol[m] = Ld[m * 6 + cc];
      rhs = x[6 * (int64_t)k + cc];
    }
    double acc = 0;
    if (lane < 60)
      for (int64_t t = b0 + wave * 10 + g; t < b1; t += NW * 10) {
        const double *Lb = Lv + 36 * t + cc;
        const Row6 xi = load_row(x + 6 * (int64_t)P.rowidx[t]);
        acc += Lb[0] * xi.v[0] + Lb[6] * xi.v[1] + Lb[12] * xi.v[2] + Lb[18] * xi.v[3] + Lb[24] * xi.v[4] + Lb[30] * xi.v[5];
      }
    if (lane < 60) sred[wave * 60 + lane] = acc;
    __syncthreads();
    if (wave == 0) {
      double s = rhs;
      if (lane < 6)
        for (int q = 0; q < NW * 10; ++q) s -= sred[q * 6 + lane];
      double y = 0;
#pragma unroll
      for (int i = 5; i >= 0; --i) {
        const double yi = __shfl(s, i, WAVE) / __shfl(lcol[i], i, WAVE);   // x_i = s_i / L_ii
        if (lane == i) y = yi;
        // s_c -= L_ic x_i for c < i ; lane c holds column c of L: L_ic = lcol[i] on lane c
        if (lane < i) s -= lcol[i] * yi;
      }
      if (lane < 6) x[6 * (int64_t)k + lane] = y;
    }
    __syncthreads();
  }
}

// ---- panel solves.  Forward: the external part of every panel column's row list is summed by a wide kernel
// (FWD_CHUNK entries per workgroup, partial sums in a fixed order), then one wave per panel runs the in-panel
// substitution out of LDS.  Backward: same split; the external part runs over the panel's off-triangle rows.
__global__ __launch_bounds__(256) void k_fwd_ext(DevPlan P, const double *__restrict__ Lv, const double *__restrict__ x, int chunk0) {
  __shared__ double sred[240];
  const int ch = chunk0 + blockIdx.x;
  const int k = P.pp.fchunk_col[ch];
  const int64_t e0 = P.pp.fchunk_e0[ch];
  const int64_t rm = P.pp.row_mid[k];
  const int64_t e1 = e0 + FWD_CHUNK < rm ? e0 + FWD_CHUNK : rm;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, r = lane - 6 * g;
  const int gid = wave * 10 + g;
  double acc = 0;
  if (lane < 60) {
    for (int64_t e = e0 + gid; e < e1; e += 160) {
      int bi[4], ci[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t ee = e + 40 * q;
        const bool in = ee < e1;
        bi[q] = in ? P.row_blk[ee] : P.zero_blk;
        ci[q] = in ? P.row_col[ee] : 0;
      }
      Row6 l[4], y[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { l[q] = load_row(Lv + 36 * (int64_t)bi[q] + 6 * r); y[q] = load_row(x + 6 * (int64_t)ci[q]); }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc += l[q].v[0] * y[q].v[0] + l[q].v[1] * y[q].v[1] + l[q].v[2] * y[q].v[2] + l[q].v[3] * y[q].v[3] + l[q].v[4] * y[q].v[4] + l[q].v[5] * y[q].v[5];
    }
    sred[gid * 6 + r] = acc;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double s = 0;
    for (int q = 0; q < 40; ++q) s += sred[q * 6 + threadIdx.x];
    P.pp.fpart[6 * (int64_t)ch + threadIdx.x] = s;
  }
}

__global__ __launch_bounds__(64) void k_fwd_tri(DevPlan P, double *__restrict__ x, int pn0) {
  // stand-alone forward solve (factor already resident): y_T = T^-1 s from the operand tiles, blocked on the 16x16 tiles:
  // w_J = s_J + sum_{I<J} (-T_JI) y_I, y_J = Dinv_J w_J.  A tile is column-major, so a row of it is a stride-16 walk:
  // lane (p, i) takes columns 4p .. 4p+3 of row i, two xor-shuffles finish the sum.
  __shared__ __attribute__((aligned(16))) double sb[16 * NJMAX], wb[16 * NJMAX], yb[16 * NJMAX];
  const int pn = pn0 + blockIdx.x;
  const PanelDesc d = P.pp.pdesc[pn];
  const int n = 6 * d.m, nJ = (n + 15) >> 4;
  const int *__restrict__ cols = P.task_cols + d.cols0;
  const int lane = threadIdx.x, i = lane & 15, p = lane >> 4;
  const double *__restrict__ tp = P.pp.ptop + (int64_t)d.top * 256;
  for (int c = lane; c < 16 * NJMAX; c += 64) {
    double sv = 0.0;
    if (c < n) {
      const int k = c / 6, cc = c - 6 * k;
      sv = x[6 * (int64_t)cols[k] + cc];
      const int64_t ce = (int64_t)pn * PM + k;
      const int f0 = P.pp.pcol_fchunk0[ce], fn = P.pp.pcol_fchunkn[ce];
      for (int q = 0; q < fn; ++q) sv -= P.pp.fpart[6 * (int64_t)(f0 + q) + cc];
    }
    sb[c] = sv;
  }
  __builtin_amdgcn_wave_barrier();
  for (int J = 0; J < nJ; ++J) {
    double acc = 0.0;
    for (int I = 0; I < J; ++I) {
      const double *__restrict__ t = tp + (J * (J - 1) / 2 + I) * 256 + i;
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += t[16 * (4 * p + u)] * yb[16 * I + 4 * p + u];
    }
    acc += __shfl_xor(acc, 16, WAVE);
    acc += __shfl_xor(acc, 32, WAVE);
    if (p == 0) wb[16 * J + i] = sb[16 * J + i] + acc;
    __builtin_amdgcn_wave_barrier();
    const double *__restrict__ dt = tp + (NLT + J) * 256 + i;
    double a2 = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) a2 += dt[16 * (4 * p + u)] * wb[16 * J + 4 * p + u];
    a2 += __shfl_xor(a2, 16, WAVE);
    a2 += __shfl_xor(a2, 32, WAVE);
    if (p == 0) yb[16 * J + i] = a2;
    __builtin_amdgcn_wave_barrier();
  }
  for (int c = lane; c < n; c += 64) x[6 * (int64_t)cols[c / 6] + (c - 6 * (c / 6))] = yb[c];
}

__global__ __launch_bounds__(64) void k_bwd_ext(DevPlan P, const double *__restrict__ Lv, const double *__restrict__ x, int chunk0) {
  __shared__ double red[PANEL_ROWS * PM * 6];
  const int ch = chunk0 + blockIdx.x;
  const BwdChunk bc = P.pp.bchunks[ch];
  if (P.task_dirty && !P.task_dirty[P.pp.pdesc[bc.pn].task]) return;      // (wildfire back-substitution: panel not re-solved)
  const int m = bc.m;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, cc = lane - 6 * g;
  const int gid = wave * 10 + g;
  const bool on = lane < 60 && gid < bc.nrows;
  const int ri = bc.row0 + (on ? gid : 0);
  const int *__restrict__ rb = P.pp.prow_blk + (int64_t)ri * PM;
  Row6 xi = {{0, 0, 0, 0, 0, 0}};
  if (on) xi = load_row(x + 6 * (int64_t)P.pp.prow_idx[ri]);
#pragma unroll
  for (int k = 0; k < PM; ++k) {
    double c = 0;
    if (on && k < m) {
      const int t = rb[k];
      if (t >= 0) {
        const double *Lb = Lv + 36 * (int64_t)t + cc;
        c = Lb[0] * xi.v[0] + Lb[6] * xi.v[1] + Lb[12] * xi.v[2] + Lb[18] * xi.v[3] + Lb[24] * xi.v[4] + Lb[30] * xi.v[5];
      }
    }
    if (lane < 60) red[(gid * PM + k) * 6 + cc] = c;
  }
  __syncthreads();
  for (int q2 = threadIdx.x; q2 < m * 6; q2 += 64) {
    const int k = q2 / 6, c = q2 - 6 * k;
    double s = 0;
    for (int q = 0; q < PANEL_ROWS; ++q) s += red[(q * PM + k) * 6 + c];
    P.pp.bpart[((int64_t)ch * PM + k) * 6 + c] = s;
  }
}

// In-panel backward substitution x_T = T^-T s from the panel's operand tiles (k_panel_tri): one wave, blocked on the
// 16x16 tiles.  A tile in operand order is simply column-major (element (i, j) at 16 j + i), so (T_IJ)^T x_I and
// Dinv_J^T w are dot products of contiguous columns: lane (p, j) takes rows 4p .. 4p+3 of column j, two xor-shuffles
// finish the sum.  The tiles of tile column J (those below the diagonal + the inverted diagonal tile) are loaded one
// step ahead of their use.
__global__ __launch_bounds__(64) void k_bwd_tri(DevPlan P, double *__restrict__ x, int pn0) {
  __shared__ __attribute__((aligned(16))) double sb[16 * NJMAX], wb[16 * NJMAX], xb[16 * NJMAX];
  const int pn = pn0 + blockIdx.x;
  const PanelDesc d = P.pp.pdesc[pn];
  if (!task_runs(P, d.task)) return;
  const int n = 6 * d.m, nJ = (n + 15) >> 4;
  const int *__restrict__ cols = P.task_cols + d.cols0;
  const int lane = threadIdx.x, j = lane & 15, p = lane >> 4;
  const double *__restrict__ tp = P.pp.ptop + (int64_t)d.top * 256;
  // tile column J: lower tiles (I, J), I = J+1 .. nJ-1, into slots I, the diagonal tile into slot J
  auto load_tile_col = [&](int J, double (&A)[NJMAX][4]) {
#pragma unroll
    for (int I = 0; I < NJMAX; ++I) {
      const bool need = I >= J && I < nJ;                      // wave-uniform
      if (need) {
        const int t = I == J ? NLT + J : I * (I - 1) / 2 + J;
        const double2 lo = *reinterpret_cast<const double2 *>(tp + t * 256 + 16 * j + 4 * p);
        const double2 hi = *reinterpret_cast<const double2 *>(tp + t * 256 + 16 * j + 4 * p + 2);
        A[I][0] = lo.x; A[I][1] = lo.y; A[I][2] = hi.x; A[I][3] = hi.y;
      }
    }
  };
  double Abuf[2][NJMAX][4];
  for (int c = lane; c < 16 * NJMAX; c += 64) {
    double s = 0.0;
    if (c < n) {
      s = x[6 * (int64_t)cols[c / 6] + (c - 6 * (c / 6))];
      for (int q = 0; q < d.nchunks; ++q) s -= P.pp.bpart[(int64_t)(d.chunk0 + q) * (6 * PM) + c];
    }
    sb[c] = s;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int J = NJMAX - 1; J >= 0; --J)
    if (J < nJ) {
      double (&A)[NJMAX][4] = Abuf[J & 1];
      if (J == nJ - 1) load_tile_col(J, A);                     // first step: nothing was prefetched (static register indices only)
      if (J > 0) load_tile_col(J - 1, Abuf[(J - 1) & 1]);
      double acc = 0.0;
#pragma unroll
      for (int I = J + 1; I < NJMAX; ++I)
        if (I < nJ) {
#pragma unroll
          for (int u = 0; u < 4; ++u) acc += A[I][u] * xb[16 * I + 4 * p + u];      // strictly-lower tile (I, J), stored negated
        }
      acc += __shfl_xor(acc, 16, WAVE);
      acc += __shfl_xor(acc, 32, WAVE);
      if (p == 0) wb[16 * J + j] = sb[16 * J + j] + acc;
      __builtin_amdgcn_wave_barrier();
      double a2 = 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) a2 += A[J][u] * wb[16 * J + 4 * p + u];
      a2 += __shfl_xor(a2, 16, WAVE);
      a2 += __shfl_xor(a2, 32, WAVE);
      if (p == 0) xb[16 * J + j] = a2;
      __builtin_amdgcn_wave_barrier();
    }
  for (int c = lane; c < n; c += 64) x[6 * (int64_t)cols[c / 6] + (c - 6 * (c / 6))] = xb[c];
}


// Backward solve of a panel in ONE kernel, for levels with few panels (the top of the tree): the 16 waves of the workgroup
// take the panel's off-triangle rows in chunks of 10 (one row per lane group, as k_bwd_ext), keep their partial sums
// s_k -= L_ik^T x_i in registers across chunks, combine them through LDS in a fixed order (chunks of a wave, then waves
// 0 .. 15), and wave 0 finishes with the in-panel substitution x_T = T^-T s from the operand tiles exactly as k_bwd_tri --
// whose tile loads are issued before the row phase, so they are in flight while the rows are summed.  One launch and no
// round trip of the partial sums through memory instead of two launches per level.
constexpr int BWD_NW = 16;
__global__ __launch_bounds__(BWD_NW * 64) void k_bwd_fused(DevPlan P, const double *__restrict__ Lv, double *__restrict__ x, int pn0) {
  constexpr int NW = BWD_NW;
  __shared__ __attribute__((aligned(16))) double slab[NW][60];
  __shared__ __attribute__((aligned(16))) double wtot[NW][PM * 6];
  __shared__ __attribute__((aligned(16))) double sb[16 * NJMAX], wb[16 * NJMAX], xb[16 * NJMAX];
  const int pn = pn0 + blockIdx.x;
  const PanelDesc d = P.pp.pdesc[pn];
  if (!task_runs(P, d.task)) return;                       // (backward sweeps always run whole; kept for symmetry)
  const int m = d.m, n = 6 * m, nJ = (n + 15) >> 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, cc = lane - 6 * g;
  const int *__restrict__ cols = P.task_cols + d.cols0;
  const double *__restrict__ tp = P.pp.ptop + (int64_t)d.top * 256;
  // ---- rows: wave w takes chunks w, w + 16, ... ; lanes 0..5 of every wave accumulate the wave's total per column block
  double tot[PM];
#pragma unroll
  for (int k = 0; k < PM; ++k) tot[k] = 0.0;
  for (int r0 = 10 * wave; r0 < d.nrows; r0 += 10 * NW) {
    const bool on = lane < 60 && r0 + g < d.nrows;
    const int ri = d.prow0 + r0 + (on ? g : 0);
    const int *__restrict__ rb = P.pp.prow_blk + (int64_t)ri * PM;
    Row6 xi = {{0, 0, 0, 0, 0, 0}};
    if (on) xi = load_row(x + 6 * (int64_t)P.pp.prow_idx[ri]);
    // (16 columns at a time: the block ids of one half are one round trip and 16 registers)
#pragma unroll
    for (int kb = 0; kb < PM; kb += 16) {
      int tb[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) tb[k] = (on && kb + k < m) ? rb[kb + k] : -1;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (kb + k < m) {                                    // wave-uniform
          double c = 0.0;
          if (tb[k] >= 0) {
            const double *Lb = Lv + 36 * (int64_t)tb[k] + cc;
            c = Lb[0] * xi.v[0] + Lb[6] * xi.v[1] + Lb[12] * xi.v[2] + Lb[18] * xi.v[3] + Lb[24] * xi.v[4] + Lb[30] * xi.v[5];
          }
          if (lane < 60) slab[wave][lane] = c;
          __builtin_amdgcn_wave_barrier();
          if (lane < 6) {
            double sacc = 0.0;
#pragma unroll
            for (int q = 0; q < 10; ++q) sacc += slab[wave][6 * q + lane];
            tot[kb + k] += sacc;
          }
          __builtin_amdgcn_wave_barrier();
        }
    }
  }
  if (lane < 6) {
#pragma unroll
    for (int k = 0; k < PM; ++k) if (k < m) wtot[wave][6 * k + lane] = tot[k];
  }
  __syncthreads();
  if (wave > 0) return;
  // ---- in-panel substitution (as k_bwd_tri)
  const int j = lane & 15, p = lane >> 4;
  auto load_tile_col = [&](int J, double (&A)[NJMAX][4]) {
#pragma unroll
    for (int I = 0; I < NJMAX; ++I) {
      const bool need = I >= J && I < nJ;
      if (need) {
        const int t = I == J ? NLT + J : I * (I - 1) / 2 + J;
        const double2 lo = *reinterpret_cast<const double2 *>(tp + t * 256 + 16 * j + 4 * p);
        const double2 hi = *reinterpret_cast<const double2 *>(tp + t * 256 + 16 * j + 4 * p + 2);
        A[I][0] = lo.x; A[I][1] = lo.y; A[I][2] = hi.x; A[I][3] = hi.y;
      }
    }
  };
  double Abuf[2][NJMAX][4];
  for (int c = lane; c < 16 * NJMAX; c += 64) {
    double s = 0.0;
    if (c < n) {
      s = x[6 * (int64_t)cols[c / 6] + (c - 6 * (c / 6))];
      for (int w = 0; w < NW; ++w) s -= wtot[w][c];
    }
    sb[c] = s;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int J = NJMAX - 1; J >= 0; --J)
    if (J < nJ) {
      double (&A)[NJMAX][4] = Abuf[J & 1];
      if (J == nJ - 1) load_tile_col(J, A);
      if (J > 0) load_tile_col(J - 1, Abuf[(J - 1) & 1]);
      double acc = 0.0;
#pragma unroll
      for (int I = J + 1; I < NJMAX; ++I)
        if (I < nJ) {
#pragma unroll
          for (int u = 0; u < 4; ++u) acc += A[I][u] * xb[16 * I + 4 * p + u];
        }
      acc += __shfl_xor(acc, 16, WAVE);
      acc += __shfl_xor(acc, 32, WAVE);
      if (p == 0) wb[16 * J + j] = sb[16 * J + j] + acc;
      __builtin_amdgcn_wave_barrier();
      double a2 = 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) a2 += A[J][u] * wb[16 * J + 4 * p + u];
      a2 += __shfl_xor(a2, 16, WAVE);
      a2 += __shfl_xor(a2, 32, WAVE);
      if (p == 0) xb[16 * J + j] = a2;
      __builtin_amdgcn_wave_barrier();
    }
  for (int c = lane; c < n; c += 64) x[6 * (int64_t)cols[c / 6] + (c - 6 * (c / 6))] = xb[c];
}

// ---- Backward solve of the TOP LEVELS in one launch.  The ~24 narrow levels of cfg 2 cost ~19 us each as k_bwd_fused launches:
// ~4 us of launch floor, the descriptor / index / tile round trips, then the dependent part.  Here every panel of those levels
// is a workgroup of ONE launch, ordered root level first; a workgroup does everything that does not depend on x of the levels
// above (descriptors, its first tile column) and then waits until the `need` panels above it have published their part of x
// (one counter, release / acquire at agent scope -- the decoupled look-back idiom).  Workgroups are dispatched in index
// order and wait only for smaller indices, so the launch cannot deadlock whatever the residency.  Same arithmetic, same
// order of operations as k_bwd_fused: bit-identical x.
__global__ __launch_bounds__(BWD_NW * 64) void k_bwd_chain(DevPlan P, const double *__restrict__ Lv, double *__restrict__ x, int n_items, int mode) {
  constexpr int NW = BWD_NW;
  const ChainItem it = P.pp.bchain[blockIdx.x];
  __shared__ __attribute__((aligned(16))) double slab[NW][60];
  __shared__ __attribute__((aligned(16))) double wtot[NW][PM * 6];
  __shared__ __attribute__((aligned(16))) double sb[16 * NJMAX], wb[16 * NJMAX], xb[16 * NJMAX];
  const PanelDesc d = P.pp.pdesc[it.pn];
  const int m = d.m, n = 6 * m, nJ = (n + 15) >> 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, cc = lane - 6 * g;
  const int *__restrict__ cols = P.task_cols + d.cols0;
  const double *__restrict__ tp = P.pp.ptop + (int64_t)d.top * 256;
  unsigned *__restrict__ done = P.pp.bchain_done;
  // first chunk of rows: indices and L values do not depend on the levels above -> requested before the wait
  const int r00 = 10 * wave;
  const bool on0 = lane < 60 && r00 + g < d.nrows;
  const int ri0 = d.prow0 + r00 + (on0 ? g : 0);
  int idx0 = 0;
  int tb0[PM];
#pragma unroll
  for (int k = 0; k < PM; ++k) tb0[k] = -1;
  if (r00 < d.nrows) {
    const int *__restrict__ rb = P.pp.prow_blk + (int64_t)ri0 * PM;
    if (on0) idx0 = P.pp.prow_idx[ri0];
#pragma unroll
    for (int k = 0; k < PM; ++k) tb0[k] = (on0 && k < m) ? rb[k] : -1;
  }
  double warm = 0.0;
  if (mode & 4) {                                            // pull this panel's operands towards this XCD's L2 while the levels above run
#pragma unroll
    for (int k = 0; k < PM; ++k) if (tb0[k] >= 0) warm += Lv[36 * (int64_t)tb0[k] + cc];
    if (wave == 0) for (int e = lane; e < nJ * 32; e += 64) warm += tp[(NLT + (e >> 5)) * 256 + 8 * (e & 31)];
    if (warm == 123.456e300) slab[0][0] = warm;
  }
  // ---- wait for the levels above
  if (it.need > 0) {
    if (threadIdx.x == 0) {
      while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it.need) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    if (!(mode & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  // ---- rows: wave w takes chunks w, w + 16, ... ; lanes 0..5 of every wave accumulate the wave's total per column block
  double tot[PM];
#pragma unroll
  for (int k = 0; k < PM; ++k) tot[k] = 0.0;
  for (int r0 = 10 * wave; r0 < d.nrows; r0 += 10 * NW) {
    const bool first = r0 == r00;
    const bool on = lane < 60 && r0 + g < d.nrows;
    const int ri = d.prow0 + r0 + (on ? g : 0);
    const int *__restrict__ rb = P.pp.prow_blk + (int64_t)ri * PM;
    Row6 xi = {{0, 0, 0, 0, 0, 0}};
    if (on) {
      const double *xp = x + 6 * (int64_t)(first ? idx0 : P.pp.prow_idx[ri]);
      if (mode & 1) {                                      // agent-scope loads instead of an acquire fence (no L2 invalidation)
#pragma unroll
        for (int c = 0; c < 6; ++c) xi.v[c] = __hip_atomic_load(xp + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else xi = load_row(xp);
    }
#pragma unroll
    for (int kb = 0; kb < PM; kb += 16) {
      int tb[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) tb[k] = first ? tb0[kb + k] : ((on && kb + k < m) ? rb[kb + k] : -1);
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (kb + k < m) {                                    // wave-uniform
          double c = 0.0;
          if (tb[k] >= 0) {
            const double *Lb = Lv + 36 * (int64_t)tb[k] + cc;
            c = Lb[0] * xi.v[0] + Lb[6] * xi.v[1] + Lb[12] * xi.v[2] + Lb[18] * xi.v[3] + Lb[24] * xi.v[4] + Lb[30] * xi.v[5];
          }
          if (lane < 60) slab[wave][lane] = c;
          __builtin_amdgcn_wave_barrier();
          if (lane < 6) {
            double sacc = 0.0;
#pragma unroll
            for (int q = 0; q < 10; ++q) sacc += slab[wave][6 * q + lane];
            tot[kb + k] += sacc;
          }
          __builtin_amdgcn_wave_barrier();
        }
    }
  }
  if (lane < 6) {
#pragma unroll
    for (int k = 0; k < PM; ++k) if (k < m) wtot[wave][6 * k + lane] = tot[k];
  }
  __syncthreads();
  if (wave > 0) return;
  // ---- in-panel substitution (as k_bwd_tri)
  const int j = lane & 15, p = lane >> 4;
  auto load_tile_col = [&](int J, double (&A)[NJMAX][4]) {
#pragma unroll
    for (int I = 0; I < NJMAX; ++I) {
      const bool need = I >= J && I < nJ;
      if (need) {
        const int t = I == J ? NLT + J : I * (I - 1) / 2 + J;
        const double2 lo = *reinterpret_cast<const double2 *>(tp + t * 256 + 16 * j + 4 * p);
        const double2 hi = *reinterpret_cast<const double2 *>(tp + t * 256 + 16 * j + 4 * p + 2);
        A[I][0] = lo.x; A[I][1] = lo.y; A[I][2] = hi.x; A[I][3] = hi.y;
      }
    }
  };
  double Abuf[2][NJMAX][4];
  for (int c = lane; c < 16 * NJMAX; c += 64) {
    double s = 0.0;
    if (c < n) {
      s = x[6 * (int64_t)cols[c / 6] + (c - 6 * (c / 6))];
      for (int w = 0; w < NW; ++w) s -= wtot[w][c];
    }
    sb[c] = s;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int J = NJMAX - 1; J >= 0; --J)
    if (J < nJ) {
      double (&A)[NJMAX][4] = Abuf[J & 1];
      if (J == nJ - 1) load_tile_col(J, A);
      if (J > 0) load_tile_col(J - 1, Abuf[(J - 1) & 1]);
      double acc = 0.0;
#pragma unroll
      for (int I = J + 1; I < NJMAX; ++I)
        if (I < nJ) {
#pragma unroll
          for (int u = 0; u < 4; ++u) acc += A[I][u] * xb[16 * I + 4 * p + u];
        }
      acc += __shfl_xor(acc, 16, WAVE);
      acc += __shfl_xor(acc, 32, WAVE);
      if (p == 0) wb[16 * J + j] = sb[16 * J + j] + acc;
      __builtin_amdgcn_wave_barrier();
      double a2 = 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) a2 += A[J][u] * wb[16 * J + 4 * p + u];
      a2 += __shfl_xor(a2, 16, WAVE);
      a2 += __shfl_xor(a2, 32, WAVE);
      if (p == 0) xb[16 * J + j] = a2;
      __builtin_amdgcn_wave_barrier();
    }
  if (mode & 2) {
    for (int c = lane; c < n; c += 64) __hip_atomic_store(x + 6 * (int64_t)cols[c / 6] + (c - 6 * (c / 6)), xb[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            // (the stores are acknowledged before the counter moves)
  } else {
    for (int c = lane; c < n; c += 64) x[6 * (int64_t)cols[c / 6] + (c - 6 * (c / 6))] = xb[c];
    // ---- publish: this panel's part of x is visible at agent scope before the counter moves; the last panel re-arms the counter
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  }
  if (lane == 0) {
    const unsigned prev = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1 == (unsigned)n_items) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// ---- multi-GPU: a rank's contribution to the top of the factor.  One lane group per top block t (row per lane, 10
// blocks per wave):  L[t] = H_partial[t] (+ lambda on the designated rank's diagonal blocks) - sum over the updates whose
// source column lies in THIS rank's domain.  After the all-reduce over the ranks L[t] holds A[t] minus every
// domain-sourced update; the top-sourced updates follow in the ordinary kernels (k_chol_acc from top_ext0, the panels).
__global__ __launch_bounds__(64) void k_dist_acc(DevPlan P, const double *__restrict__ Hblk, double *__restrict__ Lv,
                                                 const double *__restrict__ lambda_p) {
  __shared__ __attribute__((aligned(16))) double tile[360];
  const int lane = threadIdx.x, g = lane / 6, r = lane - 6 * g;
  const int64_t q = (int64_t)blockIdx.x * 10 + g;
  const int64_t nq = (int64_t)P.zero_blk - P.top_blk0;                 // zero_blk == nnzL
  const bool on = lane < 60 && q < nq;
  Row6 acc = {{0, 0, 0, 0, 0, 0}};
  const int64_t t = P.top_blk0 + (on ? q : 0);
  if (on) {
    const int a = P.asrc[t];
    if (a >= 0) {
      acc = load_row(Hblk + 36 * (int64_t)a + 6 * r);
      if (a < P.nb && P.lambda_rank) {
        const double lambda = *lambda_p;
#pragma unroll
        for (int c = 0; c < 6; ++c) acc.v[c] += (c == r) ? lambda : 0.0;
      }
    }
  }
  apply_ops(P, Lv, acc, g, r, on ? P.own_op0[q] : 0, on ? P.own_op1[q] : 0, 1, tile);
  if (on) store_row(Lv + 36 * t + 6 * r, acc);
}
// ... and to the right-hand side of the top columns: x_k = b_partial_k - sum_{j in this rank's domain} L_kj y_j
// (one wave per top column)
// (bfull: landmark elimination in distributed mode -- bvec is this rank's REDUCED gradient b - sum over ITS landmarks, b the completed
//  gradient of the top; b itself is counted on one rank, the landmark terms (bvec - bfull) on every rank)
__global__ __launch_bounds__(64) void k_dist_rhs(DevPlan P, const double *__restrict__ Lv, const double *__restrict__ bvec, double *__restrict__ x,
                                                 const double *__restrict__ bfull) {
  __shared__ double sred[60];
  const int kq = blockIdx.x, k = P.top_col0 + kq;
  const int lane = threadIdx.x, g = lane / 6, r = lane - 6 * g;
  const int64_t e0 = P.own_row0[kq], e1 = P.own_row1[kq];
  double acc = 0;
  if (lane < 60) {
    for (int64_t e = e0 + g; e < e1; e += 40) {
      int bi[4], ci[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t ee = e + 10 * q;
        const bool in = ee < e1;
        bi[q] = in ? P.row_blk[ee] : P.zero_blk;
        ci[q] = in ? P.row_col[ee] : 0;
      }
      Row6 l[4], y[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { l[q] = load_row(Lv + 36 * (int64_t)bi[q] + 6 * r); y[q] = load_row(x + 6 * (int64_t)ci[q]); }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc += l[q].v[0] * y[q].v[0] + l[q].v[1] * y[q].v[1] + l[q].v[2] * y[q].v[2] + l[q].v[3] * y[q].v[3] + l[q].v[4] * y[q].v[4] + l[q].v[5] * y[q].v[5];
    }
    sred[lane] = acc;
  }
  __syncthreads();
  if (lane < 6) {
    double sv = P.lambda_rank ? bvec[6 * (int64_t)k + lane] : (bfull ? bvec[6 * (int64_t)k + lane] - bfull[6 * (int64_t)k + lane] : 0.0);    // b of the top was completed after the linearisation: counted once
    for (int q = 0; q < 10; ++q) sv -= sred[q * 6 + lane];
    x[6 * (int64_t)k + lane] = sv;
  }
}
// poses a rank is responsible for (its domain; rank 0: the top and the fixed ones), zeros elsewhere: the sum over the
// ranks is the complete estimate (end of fgo_optimize in distributed mode)
__global__ void k_mask_poses(int64_t n, const double *__restrict__ poses, double *__restrict__ out, const int *__restrict__ pose_group, int rank, int world) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 8) return;
  const int gq = pose_group[i >> 3];
  const bool mine = gq == rank || (rank == 0 && (gq < 0 || gq >= world));
  out[i] = mine ? poses[i] : 0.0;
}

// zero fill as a KERNEL: everything inside a captured LM trial is a kernel node.  (hipMemsetAsync nodes were seen to
// carry a garbage fill value -- 0x40404040 in the failure flag -- when a second host thread issued HIP calls while
// this thread's stream was being captured: tools/debug_dist.py)
__global__ void k_zero(double *__restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.0;
}
__global__ void k_zero_int(int *__restrict__ p) { *p = 0; }

// partial sweep: the right-hand side of the columns that are re-solved, the saved forward solution y of all others
__global__ void k_mix_rhs(DevPlan P, const double *__restrict__ b, const double *__restrict__ ysaved, double *__restrict__ x, const unsigned char *__restrict__ col_dirty) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)P.nb * 6) return;
  x[i] = col_dirty[i / 6] ? b[i] : ysaved[i];
}
void launch_mix_rhs(const DevPlan &P, const double *b, const double *ysaved, double *x, const unsigned char *col_dirty, hipStream_t s) {
  hipLaunchKernelGGL(k_mix_rhs, dim3((unsigned)(((int64_t)P.nb * 6 + 255) / 256)), dim3(256), 0, s, P, b, ysaved, x, col_dirty);
}
void launch_copy_vec(const double *src, double *dst, int64_t n, hipStream_t s);

// x (permuted) <- b (permuted): plain copy kept as a kernel so the whole trial is capturable in a hipGraph
__global__ void k_copy(const double *__restrict__ src, double *__restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers (no synchronisation, capturable)
#ifndef TRI_NW
#define TRI_NW 16
#endif
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

void launch_zero(double *p, int64_t n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_zero, dim3(cdiv(n, 256) > 2048 ? 2048 : cdiv(n, 256)), dim3(256), 0, s, p, n);
}
__global__ void k_scatter_edges(const double *__restrict__ stage, int64_t n, int64_t e0, double *__restrict__ rec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 28) return;
  const int64_t e = i / 28; const int k = (int)(i - 28 * e);
  rec[EDGE_REC * (e0 + e) + (k < 7 ? k : k + 1)] = stage[i];            // [0..6] measurement, [8..28] information
}
void launch_scatter_edges(const double *stage, int64_t n, int64_t e0, double *rec, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_scatter_edges, dim3(cdiv(n * 28, 256)), dim3(256), 0, s, stage, n, e0, rec);
}
void launch_zero_flag(int *p, hipStream_t s) { hipLaunchKernelGGL(k_zero_int, dim3(1), dim3(1), 0, s, p); }
// distributed trial: [5] <- the failure flag, [6] <- the LM scale partial, next to [4] (chi2 partial): one collective for the three
// (host: pinned, device-visible host memory or NULL -- a single-GPU trial's three results go straight there: the 24-byte read-back as a blit
//  copy of its own cost ~18 us of stream time per trial, as did the 8-byte upload of lambda that k_fetch_scalar replaces; round 6)
__global__ void k_pack_scalars(double *__restrict__ scal, const int *__restrict__ fail, double *__restrict__ host) {
  const double f = (double)*fail, sc = scal[1];
  scal[5] = f; scal[6] = sc;
  if (host) { host[4] = scal[4]; host[5] = f; host[6] = sc; }
}
void launch_pack_scalars(double *scal, const int *fail, hipStream_t s, double *host) { hipLaunchKernelGGL(k_pack_scalars, dim3(1), dim3(1), 0, s, scal, fail, host); }
__global__ void k_fetch_scalar(double *__restrict__ dst, const double *__restrict__ src_host) { *dst = *reinterpret_cast<const volatile double *>(src_host); }
void launch_fetch_scalar(double *dst, const double *src_host, hipStream_t s) { hipLaunchKernelGGL(k_fetch_scalar, dim3(1), dim3(1), 0, s, dst, src_host); }

void launch_linearize(const DevPlan &P, const double *poses, double *Hblk, double *bvec, double *scalar_out,
                      hipStream_t s) {
  constexpr int G = 4;
  int blocks = cdiv(P.n_poses * G, 256);
  if (P.zero_offdiag)   // shard mode: off-diagonal blocks of edges owned by other ranks have no local writer
    launch_zero(Hblk + 36 * (int64_t)P.nb, 36 * (int64_t)(P.n_hblocks - P.nb), s);
  hipLaunchKernelGGL((k_linearize<G, false>), dim3(blocks), dim3(256), 0, s, P, poses, Hblk, bvec, P.partial);
  if (P.n_hubs > 0)
    hipLaunchKernelGGL((k_linearize<G, true>), dim3(P.n_hubs), dim3(256), 0, s, P, poses, Hblk, bvec, P.partial + blocks);
  if (P.n_hub_multi > 0) hipLaunchKernelGGL(k_hub_combine, dim3(P.n_hub_multi), dim3(64), 0, s, P, Hblk, bvec);
  blocks += P.n_hubs;
  if (P.n_dup_groups > 0)
    hipLaunchKernelGGL(k_dup_offdiag, dim3(cdiv(P.n_dup_groups, 64)), dim3(64), 0, s, P, poses, Hblk);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, s, P.partial, (int64_t)blocks, scalar_out, 0);
}
int linearize_blocks(const DevPlan &P) { return cdiv(P.n_poses * 4, 256) + P.n_hubs; }
void launch_reduce(const double *partial, int64_t n, double *out, int mode, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, s, partial, n, out, mode);
}

void launch_chi2(const DevPlan &P, const double *poses, double *scalar_out, hipStream_t s) {
  int blocks = cdiv(P.n_edges, 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_chi2, dim3(blocks), dim3(256), 0, s, P, poses, P.partial);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, s, P.partial, (int64_t)blocks, scalar_out, 0);
}

void launch_maxdiag(const DevPlan &P, const double *Hblk, double *scalar_out, hipStream_t s) {
  int blocks = cdiv((int64_t)P.n_real * 6, 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_maxdiag, dim3(blocks), dim3(256), 0, s, Hblk, P.pose_col, (int64_t)P.n_real, P.partial);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, s, P.partial, (int64_t)blocks, scalar_out, 1);
}

void launch_update(const DevPlan &P, const double *poses, double *cand, const double *x, const double *b,
                   const double *lambda_p, double *scalar_out, hipStream_t s) {
  const int blocks = cdiv(P.n_poses, 256);
  hipLaunchKernelGGL(k_update, dim3(blocks), dim3(256), 0, s, P, poses, cand, x, b, lambda_p, P.partial);
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, s, P.partial, (int64_t)blocks, scalar_out, 0);
}

static void launch_fwd_level(const DevPlan &P, const HostSchedule &H, const double *Lv, double *x, int l, hipStream_t s) {
  const int t0 = H.level_ptr[l], nt = H.level_ptr[l + 1] - t0;
  if (H.level_maxtaskcols[l] == 1 && H.level_maxrow[l] <= 40) hipLaunchKernelGGL(k_solve_fwd<1>, dim3(nt), dim3(64), 0, s, P, Lv, x, t0);
  else if (H.level_maxrow[l] <= 160) hipLaunchKernelGGL(k_solve_fwd<4>, dim3(nt), dim3(256), 0, s, P, Lv, x, t0);
  else hipLaunchKernelGGL(k_solve_fwd<16>, dim3(nt), dim3(1024), 0, s, P, Lv, x, t0);
}
void launch_copy_vec(const double *src, double *dst, int64_t n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_copy, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, s, src, dst, n);
}
static void launch_copy(const double *src, double *dst, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_copy, dim3(cdiv(n, 256) > 1024 ? 1024 : cdiv(n, 256)), dim3(256), 0, s, src, dst, n);
}

// per-device kernel attributes (called from the structure build, with the context's device current): the leaf kernel
// keeps a sub-tree's blocks and op lists in up to ~78 KB of dynamic LDS, above the 64 KB that need no opt-in.  Once per
// device and under a lock: re-setting the attribute while another host thread launches the kernel made launches fail
// (two contexts driven from two threads: tests/test_gpu_shard.py).
void prepare_device_kernels() {
  static std::mutex mu;
  static std::vector<char> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return;
  std::lock_guard<std::mutex> lock(mu);
  if ((int)done.size() <= dev) done.resize(dev + 1, 0);
  if (done[dev]) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chol_leaf<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done[dev] = 1;
}

// With b / x given the forward solve L y = b is fused into the sweep (x <- y): a level's right-hand side is one
// more row of its columns, so its external sums ride in the accumulate launch and the in-panel substitution in the
// row kernel; non-panel levels run the generic forward kernel right after their factor kernel.
static inline bool seg_runs(const HostSchedule &H, int l, int phase) {
  if (H.level_ptr[l + 1] == H.level_ptr[l]) return false;            // empty segment
  if (phase == PHASE_ALL) return true;
  const int gq = H.seg_group[l];
  return phase == PHASE_DOMAIN ? gq == H.rank : gq == H.world;
}

void launch_mask_poses(const DevPlan &P, const double *poses, double *out, const int *pose_group, int rank, int world, hipStream_t s) {
  hipLaunchKernelGGL(k_mask_poses, dim3(cdiv(P.n_poses * 8, 256)), dim3(256), 0, s, P.n_poses, poses, out, pose_group, rank, world);
}

void launch_factor(const DevPlan &P, const HostSchedule &H, const double *Hblk, double *Lv, const double *lambda_p,
                   int *fail_flag, hipStream_t s, const double *b, double *x, int phase, const PartialSweep *ps, const double *b_full) {
  // (partial sweep, P.task_dirty set: the caller has prepared x = b on the dirty columns and the saved y elsewhere)
  if (x && phase != PHASE_TOP && !P.task_dirty) launch_copy(b, x, (int64_t)P.top_col0 * 6, s);   // (the top of x is written by k_dist_rhs when distributed)
  for (int l = 0; l < H.n_levels; ++l) {
    if (!seg_runs(H, l, phase)) continue;
    const int64_t a0 = H.acc_ptr[l], am = H.acc_mid[l], a1 = H.acc_ptr[l + 1];
    // partial sweep: the tasks [ta, tb] of this level cover its dirty ones (nothing_dirty: only the riders of its launches run)
    const int ta = ps ? ps->t_lo[l] : 0, tb = ps ? ps->t_hi[l] : -1;
    const bool nothing_dirty = ps && tb < ta;
    const int64_t sf = !ps ? a0 : (nothing_dirty ? a0 : ps->s0[ta]), sn = !ps ? am - a0 : (nothing_dirty ? 0 : ps->s1[tb] - ps->s0[ta]);   // short targets
    const int64_t lf = !ps ? am : (nothing_dirty ? am : ps->l0[ta]);
    const int n_long = !ps ? (int)(a1 - am) : (nothing_dirty ? 0 : (int)(ps->l1[tb] - ps->l0[ta]));
    const int n_acc_wg = n_long > 0 ? (cdiv(sn, 10) + 7) & ~7 : cdiv(sn, 10);   // (padding workgroups find idx >= count and idle)
    // forward-solve work items of the level: its panel columns, or (a level with long rows) the entries of the work-item table
    const bool fw_table = H.fwg_ptr[l + 1] - H.fwg_ptr[l] != H.level_col_ptr[l + 1] - H.level_col_ptr[l];
    int col0 = fw_table ? -1 - H.fwg_ptr[l] : H.level_col_ptr[l];
    int n_fwd_wg = (x && H.level_panel[l]) ? (fw_table ? H.fwg_ptr[l + 1] - H.fwg_ptr[l] : H.level_col_ptr[l + 1] - H.level_col_ptr[l]) : 0;
    if (ps && n_fwd_wg > 0) {
      if (nothing_dirty) n_fwd_wg = 0;
      else if (!fw_table) { col0 = ps->task_ptr[ta]; n_fwd_wg = ps->task_ptr[tb + 1] - ps->task_ptr[ta]; }
    }
    const int grid = n_acc_wg + n_long + n_fwd_wg;
    const int n_g2_full = H.g2_lvl.empty() ? 0 : (int)(H.g2_lvl[l + 1] - H.g2_lvl[l]);
    const int64_t g2f = (!ps || n_g2_full == 0) ? (H.g2_lvl.empty() ? 0 : H.g2_lvl[l]) : (nothing_dirty ? H.g2_lvl[l] : ps->g0[ta]);
    const int n_g2 = (!ps || n_g2_full == 0) ? n_g2_full : (nothing_dirty ? 0 : ps->g1[tb] - ps->g0[ta]);
    if (n_g2_full > 0) {      // (the symbolic phase builds these lists for the very wide levels only: FGO_ACC2_MIN)
      // column-group form (scalar B operand).  Split the entry lists where there are few groups (short chains at the top)
      static const int g2_narrow = (int)tune("acc2_narrow", 400);
      static const int g2_mid = (int)tune("acc2_mid", 6000);
      const int grid2 = n_g2 + n_fwd_wg;                      // (which instantiation: by the level's FULL list, partial sweep or not)
      if (grid2 <= 0) {}
      else if (n_g2_full <= g2_narrow) hipLaunchKernelGGL(k_chol_acc2<8>, dim3(grid2), dim3(512), 0, s, P, Hblk, Lv, Lv, g2f, n_g2, lambda_p, x, col0);
      else if (n_g2_full <= g2_mid) hipLaunchKernelGGL(k_chol_acc2<4>, dim3(grid2), dim3(256), 0, s, P, Hblk, Lv, Lv, g2f, n_g2, lambda_p, x, col0);
      else hipLaunchKernelGGL(k_chol_acc2<1>, dim3(grid2), dim3(64), 0, s, P, Hblk, Lv, Lv, g2f, n_g2, lambda_p, x, col0);
    } else if (grid > 0) {
      // few targets (the skinny top of the tree): split every source list 8 ways to shorten the dependent chain
      static const int64_t narrow_max = (int64_t)tune("acc_narrow", 4000);
      // very many targets (the lowest panel levels: short lists, 10^5 .. 10^6 targets): one wave per 10 targets, no
      // split-K and no LDS combine -- the launch is bound by how many independent waves are in flight, not by the
      // length of a list (cfg 2: factor sweep 5.98 -> 5.73 ms, cfg 5: 35.9 -> 33.0 ms)
      static const int64_t acc_wide2 = (int64_t)tune("acc_wide2", 60000);
      static const int64_t acc_mid2 = (int64_t)tune("acc_mid2", 15000);   // in between: split two ways (3.78 -> 3.74 ms)
      static const int acc_wide_split = (int)tune("acc_wide_split", 1);
      if (a1 - a0 <= narrow_max)
        hipLaunchKernelGGL(k_chol_acc<8>, dim3(grid), dim3(512), 0, s, P, Hblk, Lv, sf, sn, lambda_p, x, n_acc_wg, col0, n_long, lf);
      else if (a1 - a0 > acc_wide2) {
        if (acc_wide_split == 1) hipLaunchKernelGGL(k_chol_acc<1>, dim3(grid), dim3(64), 0, s, P, Hblk, Lv, sf, sn, lambda_p, x, n_acc_wg, col0, n_long, lf);
        else hipLaunchKernelGGL(k_chol_acc<2>, dim3(grid), dim3(128), 0, s, P, Hblk, Lv, sf, sn, lambda_p, x, n_acc_wg, col0, n_long, lf);
      } else if (a1 - a0 > acc_mid2)
        hipLaunchKernelGGL(k_chol_acc<2>, dim3(grid), dim3(128), 0, s, P, Hblk, Lv, sf, sn, lambda_p, x, n_acc_wg, col0, n_long, lf);
      else
        hipLaunchKernelGGL(k_chol_acc<4>, dim3(grid), dim3(256), 0, s, P, Hblk, Lv, sf, sn, lambda_p, x, n_acc_wg, col0, n_long, lf);
    }
    if (x && H.level_panel[l] && !H.fsplit_ptr.empty() && H.fsplit_ptr[l + 1] > H.fsplit_ptr[l] && !nothing_dirty)
      hipLaunchKernelGGL(k_fwd_combine, dim3(H.fsplit_ptr[l + 1] - H.fsplit_ptr[l]), dim3(64), 0, s, P, x, H.fsplit_ptr[l]);
    const int t0f = H.level_ptr[l], ntf = H.level_ptr[l + 1] - t0f;          // the whole level (kernel choices go by it)
    const int t0 = ps ? (nothing_dirty ? t0f : ta) : t0f, nt = ps ? (nothing_dirty ? 0 : tb - ta + 1) : ntf;
    const int pn0 = H.level_panel[l] ? H.level_pn0[l] + (t0 - t0f) : 0;     // (the panels of a level are numbered in task order)
    if (H.level_panel[l]) {
      // 16 waves hold a panel's trailing matrix with the fewest tiles per wave, but their registers allow one workgroup
      // per CU; levels with more panels than CUs run the 8-wave instantiation, two workgroups per CU
      const int tri_wide = tri_wide_panels(H.cus);
      // (a 4-wave instantiation with four workgroups per CU for the very wide levels -- twice the pivot chains in flight --
      //  was measured slower: cfg 2 factor sweep 3.27 -> 3.34 ms, cfg 5 21.5 -> 22.3 ms)
      // wide levels: the throughput form, one wave per panel (FGO_TRI1=0: the 8-wave latency form, two workgroups per CU)
      static const int tri1_on = (int)tune("tri1", 1);
      // (one wave per panel holds 4 panels per CU: it beats two 8-wave workgroups per CU once there are >= 3 rounds of those)
      const int tri1_min = (int)tune("tri1_min", 3 * H.cus);
      const bool tri1 = tri1_on && ntf > tri_wide && ntf >= tri1_min;
      if (tri1) {
        if (nt > 0) hipLaunchKernelGGL(k_panel_tri1, dim3(nt), dim3(64), 0, s, P, Hblk, Lv, pn0, lambda_p, fail_flag);
      } else if (ntf > tri_wide) {
        if (nt > 0) hipLaunchKernelGGL((k_panel_tri<8>), dim3(nt), dim3(8 * 64), 0, s, P, Hblk, Lv, pn0, lambda_p, fail_flag, nt, 0, 0, nt);
      } else {
        const int r0 = H.ride_ptr.empty() ? 0 : H.ride_ptr[2 * l], nr = H.ride_ptr.empty() ? 0 : H.ride_ptr[2 * l + 1] - r0;
        const int ntp = (nr > 0 && P.ride_xcd) ? (nt + 7) & ~7 : nt;      // riders start at a multiple of 8: XCD = (blockIdx - ntp) & 7
        if (ntp + nr > 0)
          hipLaunchKernelGGL((k_panel_tri<TRI_NW>), dim3(ntp + (nr + RIDE_PER_WG - 1) / RIDE_PER_WG), dim3(TRI_NW * 64), 0, s, P, Hblk, Lv, pn0, lambda_p, fail_flag, ntp, r0, nr, nt);
      }
      const int c0 = !ps ? H.rchunk_ptr[l] : (nothing_dirty ? H.rchunk_ptr[l] : ps->c0[ta]);
      const int nc = !ps ? H.rchunk_ptr[l + 1] - H.rchunk_ptr[l] : (nothing_dirty ? 0 : ps->c1[tb] - ps->c0[ta]);
      const int q0 = H.ride_ptr.empty() ? 0 : H.ride_ptr[2 * l + 1], nq = H.ride_ptr.empty() ? 0 : H.ride_ptr[2 * l + 2] - q0;   // riders of the row launch
      if (nc + nq > 0) {
        if (l >= H.rows_byc_level) hipLaunchKernelGGL(k_panel_rows_byc, dim3(nc + nq), dim3(64), 0, s, P, Hblk, Lv, c0, x, nc, lambda_p, q0);
        else hipLaunchKernelGGL(k_panel_rows, dim3(nc + nq), dim3(64), 0, s, P, Hblk, Lv, c0, x, nc, lambda_p, q0);
      }
      continue;
    }
    if (nt <= 0) continue;                              // (partial sweep: nothing dirty in this level)
    if (!H.level_leaf.empty() && H.level_leaf[l]) {
      const int lb = H.level_leaf_maxblk[l], lc = H.level_maxtaskcols[l];
      const size_t lds = ((size_t)lb + 1) * 36 * sizeof(double) + (size_t)12 * lc * sizeof(double) + ((size_t)lb + 2 + lc + 2) * sizeof(int) +
                         ((size_t)lb + 2) * sizeof(unsigned short) + (size_t)H.level_leaf_maxops[l] * sizeof(unsigned);
      hipLaunchKernelGGL((k_chol_leaf<4>), dim3(nt), dim3(256), lds, s, P, Hblk, Lv, t0, lambda_p, fail_flag, lb, lc, x);
      continue;                                         // (the forward solve of a leaf level is part of the kernel)
    } else if (H.level_maxtaskcols[l] == 1 && H.level_maxcol[l] <= 20)
      // single-column tasks (the landmarks of a bundle adjustment): one wave per task instead of four; columns of <= 20
      // blocks keep two register passes instead of three (fewer VGPRs, more waves per SIMD)
      hipLaunchKernelGGL((k_chol_fact<1, 2>), dim3(nt), dim3(64), 0, s, P, Hblk, Lv, t0, lambda_p, fail_flag);
    else if (H.level_maxtaskcols[l] == 1 && H.level_maxcol[l] <= 30)
      hipLaunchKernelGGL((k_chol_fact<1, 3>), dim3(nt), dim3(64), 0, s, P, Hblk, Lv, t0, lambda_p, fail_flag);
    else if (H.level_maxcol[l] <= 120)
      hipLaunchKernelGGL((k_chol_fact<4, 3>), dim3(nt), dim3(256), 0, s, P, Hblk, Lv, t0, lambda_p, fail_flag);
    else if (H.level_maxcol[l] <= 240)
      hipLaunchKernelGGL((k_chol_fact<8, 3>), dim3(nt), dim3(512), 0, s, P, Hblk, Lv, t0, lambda_p, fail_flag);
    else
      hipLaunchKernelGGL((k_chol_fact<16, 2>), dim3(nt), dim3(1024), 0, s, P, Hblk, Lv, t0, lambda_p, fail_flag);
    if (x) launch_fwd_level(P, H, Lv, x, l, s);
  }
  if (phase == PHASE_DOMAIN) {
    // this rank's contributions to the top: every block of the top columns, and (forward solve fused) their right-hand side
    if (H.n_top_blocks > 0) hipLaunchKernelGGL(k_dist_acc, dim3(cdiv(H.n_top_blocks, 10)), dim3(64), 0, s, P, Hblk, Lv, lambda_p);
    if (x && H.n_top_cols > 0) hipLaunchKernelGGL(k_dist_rhs, dim3(H.n_top_cols), dim3(64), 0, s, P, Lv, b, x, b_full);
  }
}

// ---- wildfire back-substitution (ISAM2 option, fgo_isam2_set_wildfire): below the backward chain a task is solved again only
// if it was re-factored or an x it reads changed by >= thr against the previous update's solution; the others keep that solution.
// decide: one wave per task of a level -> run[task];  mark: after the level, changed[column] for its columns (and the old
// solution back into x where the task was not run: x held the forward solution there).
__global__ __launch_bounds__(64) void k_wild_decide(DevPlan P, const unsigned char *__restrict__ dirty, const unsigned char *__restrict__ chg,
                                                    unsigned char *__restrict__ run, int task0, int is_panel) {
  const int task = task0 + blockIdx.x;
  const int lane = threadIdx.x;
  bool any = dirty[task] != 0;
  if (!any) {
    if (is_panel) {
      const PanelDesc d = P.pp.pdesc[P.pp.task_panel[task]];
      for (int q = lane; q < d.nrows && !any; q += 64) any = chg[P.pp.prow_idx[d.prow0 + q]] != 0;
    } else {
      const int c0 = P.task_ptr[task], c1 = P.task_ptr[task + 1];
      const int k_last = P.task_cols[c1 - 1];
      for (int ci = c0; ci < c1; ++ci) {
        const int k = P.task_cols[ci];
        for (int64_t b = P.colptr[k] + 1 + lane; b < P.colptr[k + 1] && !any; b += 64) { const int i = P.rowidx[b]; any = i > k_last && chg[i] != 0; }
      }
    }
  }
  any = __any(any);
  if (lane == 0) run[task] = any;
}
__global__ __launch_bounds__(64) void k_wild_mark(DevPlan P, double *__restrict__ x, const double *__restrict__ xprev, const unsigned char *__restrict__ run,
                                                  unsigned char *__restrict__ chg, double thr, int task0, const ChainItem *__restrict__ items) {
  // items != nullptr: the panels of the backward chain (always solved), blockIdx -> item; else task0 + blockIdx
  const int task = items ? P.pp.pdesc[items[blockIdx.x].pn].task : task0 + blockIdx.x;
  const bool ran = !run || run[task];
  const int c0 = P.task_ptr[task], c1 = P.task_ptr[task + 1];
  for (int ci = c0 + (int)threadIdx.x; ci < c1; ci += 64) {       // one column per lane
    const int k = P.task_cols[ci];
    bool big = false;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const double xo = xprev[6 * (int64_t)k + e];
      if (ran) big |= fabs(x[6 * (int64_t)k + e] - xo) >= thr;
      else x[6 * (int64_t)k + e] = xo;
    }
    chg[k] = big;
  }
}

// fwd_done: x already holds y (forward solve fused into launch_factor); only the backward sweep runs
void launch_solve(const DevPlan &P, const HostSchedule &H, const double *Lv, const double *b, double *x, hipStream_t s, bool fwd_done, int phase,
                  const Wildfire *wf) {
  if (!fwd_done) {                                            // stand-alone forward solve: single-GPU entry points only
    launch_copy(b, x, (int64_t)P.nb * 6, s);
    for (int l = 0; l < H.n_levels; ++l) {
      if (!seg_runs(H, l, PHASE_ALL)) continue;
      const int t0 = H.level_ptr[l], nt = H.level_ptr[l + 1] - t0;
      if (H.level_panel[l]) {
        const int c0 = H.fchunk_ptr[l], nc = H.fchunk_ptr[l + 1] - c0;
        if (nc > 0) hipLaunchKernelGGL(k_fwd_ext, dim3(nc), dim3(256), 0, s, P, Lv, x, c0);
        hipLaunchKernelGGL(k_fwd_tri, dim3(nt), dim3(64), 0, s, P, x, H.level_pn0[l]);
        continue;
      }
      launch_fwd_level(P, H, Lv, x, l, s);
    }
  }
  // backward sweep.  Distributed: the top first (replicated on every rank), then this rank's own domain -- a domain
  // column needs x of its ancestors only (top + own domain), so no communication
  // (distributed: the chain is the replicated top's, part of the PHASE_TOP pass)
  const bool chain = (phase == PHASE_ALL || phase == PHASE_TOP) && H.bchain_low >= 0 && H.bchain_n > 0;
  static const int chain_mode = (int)tune("bwd_chain_mode", 5);   // 1: agent-scope loads of x instead of an acquire fence (no L2 invalidation), 2: agent-scope stores + store-acknowledge wait instead of the release fence, 4: operands touched before the wait.  cfg 2 backward sweep: no chain 0.799, modes 0 / 1 / 3 / 7: 0.815 / 0.747 / 0.737 / 0.735 ms
  if (chain) {
    // the progress counter starts every launch at zero whatever happened to the launch before (an aborted launch would leave
    // it armed and let every later wait pass early): a 4-byte kernel node in front, part of the captured trial
    hipLaunchKernelGGL(k_zero_int, dim3(1), dim3(1), 0, s, reinterpret_cast<int *>(P.pp.bchain_done));
    hipLaunchKernelGGL(k_bwd_chain, dim3(H.bchain_n), dim3(BWD_NW * 64), 0, s, P, Lv, x, H.bchain_n, chain_mode);      // (items: root level first)
  }
  const bool wild = wf && chain && phase == PHASE_ALL;       // (the chain's levels are always solved: they are the dirty root paths)
  DevPlan Pw = P;
  if (wild) {
    Pw.task_dirty = wf->run;
    hipLaunchKernelGGL(k_wild_mark, dim3(H.bchain_n), dim3(64), 0, s, P, x, wf->xprev, (const unsigned char *)nullptr, wf->chg, wf->thr, 0, P.pp.bchain);
  }
  const DevPlan &PL = wild ? Pw : P;
  for (int pass = 0; pass < (phase == PHASE_ALL ? 1 : 2); ++pass)
  for (int l = H.n_levels - 1; l >= 0; --l) {
    if (!seg_runs(H, l, phase == PHASE_ALL ? PHASE_ALL : (pass == 0 ? PHASE_TOP : PHASE_DOMAIN))) continue;
    if (chain && l >= H.bchain_low && (!P.dist || H.seg_group[l] == H.world)) continue;      // solved by the chain launch
    const int t0 = H.level_ptr[l], nt = H.level_ptr[l + 1] - t0;
    // wildfire: which tasks of the level are solved again -- before its kernels; afterwards which of its columns moved
    auto level_done = [&]() { if (wild) hipLaunchKernelGGL(k_wild_mark, dim3(nt), dim3(64), 0, s, P, x, wf->xprev, wf->run, wf->chg, wf->thr, t0, (const ChainItem *)nullptr); };
    if (wild) hipLaunchKernelGGL(k_wild_decide, dim3(nt), dim3(64), 0, s, P, wf->dirty, wf->chg, wf->run, t0, H.level_panel[l] ? 1 : 0);
    if (H.level_panel[l]) {
      // few panels (the top of the tree): one fused launch per level, a 16-wave workgroup per panel
      static const int bwd_fused_max = (int)tune("bwd_fused", 256);   // swept 0 / 32 / 128 / 256 / 512 / 4096 on cfg 2: 137.3 / 138.4 / 138.9 / 139.0 / 139.0 / 132.6 it/s
      if (nt <= bwd_fused_max) {
        hipLaunchKernelGGL(k_bwd_fused, dim3(nt), dim3(BWD_NW * 64), 0, s, PL, Lv, x, H.level_pn0[l]);
        level_done(); continue;
      }
      const int c0 = H.pchunk_ptr[l], nc = H.pchunk_ptr[l + 1] - c0;
      if (nc > 0) hipLaunchKernelGGL(k_bwd_ext, dim3(nc), dim3(64), 0, s, PL, Lv, x, c0);
      hipLaunchKernelGGL(k_bwd_tri, dim3(nt), dim3(64), 0, s, PL, x, H.level_pn0[l]);
      level_done();
      continue;
    }
    if (H.level_maxtaskcols[l] == 1 && H.level_maxcol[l] <= 20) hipLaunchKernelGGL(k_solve_bwd<1>, dim3(nt), dim3(64), 0, s, PL, Lv, x, t0);
    else if (H.level_maxcol[l] <= 80) hipLaunchKernelGGL(k_solve_bwd<4>, dim3(nt), dim3(256), 0, s, PL, Lv, x, t0);
    else hipLaunchKernelGGL(k_solve_bwd<8>, dim3(nt), dim3(512), 0, s, PL, Lv, x, t0);
    level_done();
  }
}

}  // namespace fgo
