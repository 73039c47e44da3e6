// HIP kernels (gfx950, f64) for bundle adjustment with the landmarks eliminated first (device_plan.hpp "BaPlan").
// What the reference builds: Pose3 keyframes + Point3 landmarks + GenericProjectionFactor<Pose3, Point3, Cal3DS2> with
// body_P_sensor + priors (gtsam/gtsam_graph.cpp:370-448, 500-610), optimised by LevenbergMarquardtOptimizer (:1784-1788).
// A landmark couples to ~10 cameras and to nothing else, so it is eliminated analytically -- 3x3 blocks, 6x3 couplings, no
// padding to 6x6 -- and only the cameras form the block system that the sparse Cholesky (kernels.hip) factors:
//   k_ba_linearize   per landmark: residuals / Jacobians of its observations -> H_pp, b_p, chi2
//   k_ba_cameras     per camera: its observations again -> H_cc, b_c contributions and W = J_c^T w J_p per observation
//   k_ba_points      per landmark and LM trial: L_pp = chol(H_pp + lambda I), y_p = L_pp^-1 b_p
//   k_ba_couplings   per camera and trial: Y = W L_pp^-T for its observations, bred = b - sum Y y_p
//   k_ba_schur       per block of the reduced system: Hred = H - sum over the shared landmarks of Y_row Y_col^T   (gather form:
//                    one workgroup per block, fixed summation order, no FP atomics)
//   k_ba_back        per landmark: x_p = L_pp^-T (y_p - sum Y^T x_c)
// The damping is GTSAM's (lambda I on every diagonal, landmarks included); all sums run in a fixed order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <climits>
#include <algorithm>
#include "device_plan.hpp"
#include "fgo_internal.hpp"
#include "factors_device.hpp"

namespace fgo {
using namespace dev;

namespace {
__device__ __forceinline__ double wsum_ba(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void k_ba_linearize(DevPlan P, const double *__restrict__ vals,
                                                      double *__restrict__ Hpp, double *__restrict__ bp, double *__restrict__ bvec,
                                                      double *__restrict__ chi_partial) {
  __shared__ double sh[4];
  const BaPlan &B = P.ba;
  const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  double chi = 0;
  if (p < B.n_lm && B.lm_mine && !B.lm_mine[p]) {                     // another rank's landmark: no gradient here (the LM scale counts it there)
    double *__restrict__ bv = bvec + 6 * ((int64_t)P.nb + p);
#pragma unroll
    for (int k = 0; k < 6; ++k) bv[k] = 0;
  } else if (p < B.n_lm) {
    const int v = B.lm_var[p];
    const double4 pt4 = *reinterpret_cast<const double4 *>(vals + 8 * (int64_t)v);
    const V3 pt = {pt4.x, pt4.y, pt4.z};
    { double *__restrict__ pv = B.pt_val + 3 * (int64_t)p; pv[0] = pt.x; pv[1] = pt.y; pv[2] = pt.z; }   // for k_ba_cameras: a stream there, a two-level gather from `vals`
    double h00 = 0, h01 = 0, h02 = 0, h11 = 0, h12 = 0, h22 = 0, g0 = 0, g1 = 0, g2 = 0;
    for (int64_t q = B.pt_ptr[p]; q < B.pt_ptr[p + 1]; ++q) {
      const double *__restrict__ uvw = B.pt_uvw + 3 * q;
      const Pose X = load_pose(vals + 8 * (int64_t)B.pt_cam[q]);
      double r[6];
      M6 Jx, Jp;
      reproj_factor<true>(X, pt, uvw[0], uvw[1], P.cam, r, Jx, Jp);
      const double w = uvw[2];
      chi += w * (r[0] * r[0] + r[1] * r[1]);
      const double a0 = Jp.m[0], a1 = Jp.m[1], a2 = Jp.m[2], b0 = Jp.m[6], b1 = Jp.m[7], b2 = Jp.m[8];
      h00 += w * (a0 * a0 + b0 * b0); h01 += w * (a0 * a1 + b0 * b1); h02 += w * (a0 * a2 + b0 * b2);
      h11 += w * (a1 * a1 + b1 * b1); h12 += w * (a1 * a2 + b1 * b2); h22 += w * (a2 * a2 + b2 * b2);
      g0 -= w * (a0 * r[0] + b0 * r[1]); g1 -= w * (a1 * r[0] + b1 * r[1]); g2 -= w * (a2 * r[0] + b2 * r[1]);
    }
    // unary priors on the landmark (PriorFactor<Point3>: raw mean, information in the upper-left 3x3 of the padded block),
    // from the table packed in the landmarks' order (BaPlan::lp_val)
    if (P.n_priors > 0 && P.lin_priors) {
      for (int64_t q = B.lp_ptr[p]; q < B.lp_ptr[p + 1]; ++q) {
        const double *__restrict__ a = B.lp_val + 9 * q;
        const double r0 = pt.x - a[0], r1 = pt.y - a[1], r2 = pt.z - a[2];
        const double i00 = a[3], i01 = a[4], i02 = a[5], i11 = a[6], i12 = a[7], i22 = a[8];
        const double t0 = i00 * r0 + i01 * r1 + i02 * r2, t1 = i01 * r0 + i11 * r1 + i12 * r2, t2 = i02 * r0 + i12 * r1 + i22 * r2;
        chi += r0 * t0 + r1 * t1 + r2 * t2;
        h00 += i00; h01 += i01; h02 += i02; h11 += i11; h12 += i12; h22 += i22;
        g0 -= t0; g1 -= t1; g2 -= t2;
      }
    }
    double *__restrict__ ho = Hpp + 6 * (int64_t)p;
    ho[0] = h00; ho[1] = h01; ho[2] = h02; ho[3] = h11; ho[4] = h12; ho[5] = h22;
    double *__restrict__ bo = bp + 3 * (int64_t)p;
    bo[0] = g0; bo[1] = g1; bo[2] = g2;
    double *__restrict__ bv = bvec + 6 * ((int64_t)P.nb + p);         // the gradient of the virtual column (gain-ratio model)
    bv[0] = g0; bv[1] = g1; bv[2] = g2; bv[3] = 0; bv[4] = 0; bv[5] = 0;
  }
  chi = wsum_ba(chi);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = chi;
  __syncthreads();
  if (threadIdx.x == 0) chi_partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// per landmark and trial: the 3x3 Cholesky factor of H_pp + lambda I and y_p = L_pp^-1 b_p
__global__ __launch_bounds__(256) void k_ba_points(DevPlan P, const double *__restrict__ Hpp, const double *__restrict__ bp,
                                                   const double *__restrict__ lambda_p, int *__restrict__ fail_flag) {
  const BaPlan &B = P.ba;
  const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (p >= B.n_lm || (B.lm_mine && !B.lm_mine[p])) return;
  const double lambda = *lambda_p;
  const double *__restrict__ h = Hpp + 6 * (int64_t)p;
  const double h00 = h[0] + lambda, h01 = h[1], h02 = h[2], h11 = h[3] + lambda, h12 = h[4], h22 = h[5] + lambda;
  bool ok = h00 > 0;
  const double l00 = sqrt(h00), i00 = 1.0 / l00;
  const double l10 = h01 * i00, l20 = h02 * i00;
  const double d11 = h11 - l10 * l10;
  ok = ok && d11 > 0;
  const double l11 = sqrt(d11), i11 = 1.0 / l11;
  const double l21 = (h12 - l20 * l10) * i11;
  const double d22 = h22 - l20 * l20 - l21 * l21;
  ok = ok && d22 > 0;
  const double l22 = sqrt(d22), i22 = 1.0 / l22;
  if (!ok) atomicOr(fail_flag, 1);
  // (H_pp + lambda I)^-1 = M^T M with M = L^-1 (lower), and its product with b_p: what the reduction and the back-substitution
  // multiply by.  (Until round 3 the factor was stored and a pass over all observations wrote Y = W L^-T, 144 bytes per
  // observation and trial, for k_ba_schur / k_ba_back to read again: 0.41 ms of the 3.3 ms trial at cfg 3.)
  const double m00 = i00, m10 = -l10 * i00 * i11, m11 = i11;
  const double m20 = -(l20 * m00 + l21 * m10) * i22, m21 = -l21 * m11 * i22, m22 = i22;
  const double v00 = m00 * m00 + m10 * m10 + m20 * m20, v01 = m10 * m11 + m20 * m21, v02 = m20 * m22;
  const double v11 = m11 * m11 + m21 * m21, v12 = m21 * m22, v22 = m22 * m22;
  double *__restrict__ ho = B.Hinv + 6 * (int64_t)p;
  ho[0] = v00; ho[1] = v01; ho[2] = v02; ho[3] = v11; ho[4] = v12; ho[5] = v22;
  const double *__restrict__ g = bp + 3 * (int64_t)p;
  double *__restrict__ zo = B.zp + 3 * (int64_t)p;
  zo[0] = v00 * g[0] + v01 * g[1] + v02 * g[2];
  zo[1] = v01 * g[0] + v11 * g[1] + v12 * g[2];
  zo[2] = v02 * g[0] + v12 * g[1] + v22 * g[2];
}
// One workgroup of NW waves per block of the reduced system that receives landmark terms:
//   S(row camera, column camera) = H - sum over shared landmarks p of  W_a (H_pp + lambda I)^-1 W_b^T,
// and, in the workgroup of a camera's DIAGONAL block (whose list holds every observation of the camera paired with itself),
// the camera's reduced right-hand side  b - sum W_o (H_pp + lambda I)^-1 b_p.  Lane mapping of the block Cholesky kernels:
// lane 6 g + r owns row r of the block; the NW * 10 lane groups stride the block's (row observation, column observation,
// landmark) list -- per entry a lane reads its row of W_a (3 values), the landmark's inverse (6) and ONE row of W_b, the other
// five rows arrive from the sibling lanes through a wave-private LDS tile -- and the partial blocks are summed in a fixed order.
// (The launch is a latency chain per workgroup -- block descriptor, pair indices, gathers, exchange, combine -- not a
//  bandwidth problem: with the column operand forced to hit in L1 it ran only 15 % faster.  So what counts is workgroups in
//  flight, and the partial blocks re-use the exchange tiles' LDS.)
template <int NW, int NP, int OCC>
__global__ __launch_bounds__(NW * 64, OCC) void k_ba_schur(DevPlan P, const double *__restrict__ W, const double *__restrict__ H, double *__restrict__ Hred,
                                                         const double *__restrict__ b, double *__restrict__ bred, const int *__restrict__ tlist) {
  static_assert(NP <= 2, "exchange tiles: two pairs in flight per lane group");
  __shared__ __attribute__((aligned(16))) double smem[NW * 10 * 36];
  double (*tile)[10][2][18] = reinterpret_cast<double (*)[10][2][18]>(smem);   // [wave][lane group][pair in flight][row-major 6 x 3]
  double (*part)[36] = reinterpret_cast<double (*)[36]>(smem);
  const BaPlan &B = P.ba;
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: every XCD takes a CONTIGUOUS range of the blocks, i.e.
  // of the cameras -- a landmark is seen by keyframes that are neighbours in time, so the W blocks a range touches are those
  // of its own cameras and a few neighbours and stay in that XCD's L2.  (In list order every XCD saw every camera: the PMC
  // pass showed 7.8 GB of HBM reads per launch for 0.72 GB of W -- one miss per pair.)
  const int bq = (int)gridDim.x >> 3, br = (int)gridDim.x & 7, bx = (int)blockIdx.x & 7;
  const int t = tlist[bx * bq + (bx < br ? bx : br) + ((int)blockIdx.x >> 3)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, r = lane - 6 * g;
  const int gid = wave * 10 + g;
  const int64_t o0 = B.tgt_ptr[t], o1 = B.tgt_ptr[t + 1];
  const int64_t blk = B.tgt_blk[t];
  const bool diag = blk < P.nb;                                                   // a camera's own block: its column is blk
  const double h_in = threadIdx.x < 36 ? H[36 * blk + threadIdx.x] : 0.0;       // requested before the gathers: off the dependent chain
  const double b_in = (diag && threadIdx.x < 6) ? b[6 * blk + threadIdx.x] : 0.0;
  double acc[6] = {0, 0, 0, 0, 0, 0}, gacc = 0;
  if (lane < 60) {
    constexpr int ST = NW * 10;
    int64_t o = o0 + gid;
    int ia[NP], ib[NP], lm[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) { const int64_t q = o + k * ST; ia[k] = q < o1 ? B.op_a[q] : -1; ib[k] = q < o1 ? B.op_b[q] : 0; lm[k] = q < o1 ? B.op_lm[q] : 0; }
    while (__any(o < o1)) {
      o += NP * ST;
      int na[NP], nb[NP], nl[NP];                                                // the next pairs' indices first
#pragma unroll
      for (int k = 0; k < NP; ++k) { const int64_t q = o + k * ST; na[k] = q < o1 ? B.op_a[q] : -1; nb[k] = q < o1 ? B.op_b[q] : 0; nl[k] = q < o1 ? B.op_lm[q] : 0; }
      double pa[NP][3], pb[NP][3], hv[NP][6];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const double *__restrict__ wa = W + 18 * (int64_t)(ia[k] < 0 ? 0 : ia[k]) + 3 * r, *__restrict__ wb = W + 18 * (int64_t)ib[k] + 3 * r;
        const double *__restrict__ hp = B.Hinv + 6 * (int64_t)lm[k];
#pragma unroll
        for (int x = 0; x < 3; ++x) { pa[k][x] = wa[x]; pb[k][x] = wb[x]; }
#pragma unroll
        for (int x = 0; x < 6; ++x) hv[k][x] = hp[x];
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        double *__restrict__ m = &tile[wave][g][k][0];
        m[3 * r] = pb[k][0]; m[3 * r + 1] = pb[k][1]; m[3 * r + 2] = pb[k][2];
        if (diag && ia[k] >= 0 && ia[k] == ib[k]) {                              // (diag: uniform in the workgroup) the gradient term of an observation
          const double *__restrict__ z = B.zp + 3 * (int64_t)lm[k];
          gacc += pa[k][0] * z[0] + pa[k][1] * z[1] + pa[k][2] * z[2];
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        // row r of W_a (H_pp + lambda I)^-1
        const bool on = ia[k] >= 0;
        const double a0 = on ? pa[k][0] * hv[k][0] + pa[k][1] * hv[k][1] + pa[k][2] * hv[k][2] : 0.0;
        const double a1 = on ? pa[k][0] * hv[k][1] + pa[k][1] * hv[k][3] + pa[k][2] * hv[k][4] : 0.0;
        const double a2 = on ? pa[k][0] * hv[k][2] + pa[k][1] * hv[k][4] + pa[k][2] * hv[k][5] : 0.0;
        const double *__restrict__ m = &tile[wave][g][k][0];
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] += a0 * m[3 * c] + a1 * m[3 * c + 1] + a2 * m[3 * c + 2];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < NP; ++k) { ia[k] = na[k]; ib[k] = nb[k]; lm[k] = nl[k]; }
    }
  }
  if (NW > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();          // every wave is done with the tiles
  if (lane < 60) {
#pragma unroll
    for (int c = 0; c < 6; ++c) part[gid][6 * r + c] = acc[c];
  }
  if (NW > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
  if (threadIdx.x < 36) {
    double s = 0;
    for (int q = 0; q < NW * 10; ++q) s += part[q][threadIdx.x];
    Hred[36 * blk + threadIdx.x] = h_in - s;
  }
  if (diag) {
    if (NW > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    if (lane < 60) part[gid][r] = gacc;
    if (NW > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    if (threadIdx.x < 6) {
      double s = 0;
      for (int q = 0; q < NW * 10; ++q) s += part[q][threadIdx.x];
      bred[6 * blk + threadIdx.x] = b_in - s;
    }
  }
}

// ---- The reduced system, one workgroup per COLUMN camera (round 4).  k_ba_schur above gathers, per (row observation, column
// observation, landmark) triple, the row camera's W (144 B), the column camera's W (144 B, then exchanged between the six
// lanes through LDS) and the landmark's inverse (48 B): 27.5 M triples x 336 B through the L2s at cfg 3, 0.92 ms.  All blocks
// (row, k) of one column camera k share the SAME column operands: per observation o of camera k the 3x6 matrix
//   B_o = (H_pp + lambda I)^-1 W_o^T
// is formed ONCE, in LDS (batches of 256 observations: 36 KB), and a triple costs its lane one 24-byte row of the row camera's
// W and nine broadcast LDS reads:  S(row, k)[r][:] += W_o2[r][0..2] B_o.  No exchange tile, no wave barrier inside the loop,
// a third of the global bytes.  Lane groups: NG = 80 per workgroup, G = NG / (blocks of the camera) slices per block (each
// strides its block's pair list, which ascends with o); partial blocks are summed in a fixed order.  The camera's reduced
// right-hand side  b_k - sum_o W_o (H_pp + lambda I)^-1 b_p  falls out of the staging pass.
template <int NW, int NP>
__global__ __launch_bounds__(NW * 64) void k_ba_schur_cam(DevPlan P, const double *__restrict__ W, const double *__restrict__ H, double *__restrict__ Hred,
                                                         const double *__restrict__ b, double *__restrict__ bred) {
  constexpr int NB = 256, NG = NW * 10;
  __shared__ __attribute__((aligned(16))) double Bs[NB * 18];       // B_o as [3][6]; re-used for the partial blocks at the end
  __shared__ double gpart[NG][6];
  static_assert(NG * 36 <= NB * 18, "partial blocks alias the staging area");
  const BaPlan &B = P.ba;
  const int nwg = (int)gridDim.x, bq = nwg >> 3, br = nwg & 7, bx = (int)blockIdx.x & 7;
  const int i = B.cam_list[bx * bq + (bx < br ? bx : br) + ((int)blockIdx.x >> 3)];       // XCD-contiguous ranges of the cameras
  const int64_t o0 = B.cam_ptr[i], o1 = B.cam_ptr[i + 1];
  const int64_t t0 = B.cam_t0[i];
  const int nT = (int)(B.cam_t0[i + 1] - t0);
  const int col = B.cam_col[i];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 6, r = lane - 6 * g, gid = wave * 10 + g;
  const bool lane_on = lane < 60;
  const int G = nT > 0 ? NG / nT : 1;                               // slices per block (host: nT <= NG)
  const int slot = gid / G, slice = gid - G * slot;
  const bool work = lane_on && slot < nT;
  const int64_t t = t0 + (work ? slot : 0);
  const int64_t e1 = work ? B.tgt_ptr[t + 1] : 0;
  int64_t e = work ? B.tgt_ptr[t] + slice : 0;
  const int64_t blk = work ? B.tgt_blk[t] : 0;
  const double h_in = (work && slice == 0) ? 0.0 : 0.0; (void)h_in;
  double acc[6] = {0, 0, 0, 0, 0, 0}, gacc = 0;
  // the lane group's next NP pairs (indices ahead of the values): a pair's row operand is a 24-byte gather from another camera's
  // part of W -- HBM latency --, so NP of them are in flight at once (one at a time: 0.88 ms for the launch at cfg 3)
  int oa[NP], ob[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int64_t q = e + (int64_t)k * G;
    const bool in = work && q < e1;
    oa[k] = in ? B.op_a[q] : 0; ob[k] = in ? B.op_b[q] : INT32_MAX;
  }
  for (int64_t base = o0; base < o1; base += NB) {
    const int64_t bend = base + NB < o1 ? base + NB : o1;
    __syncthreads();                                                // the previous batch is consumed
    if (lane_on)
      for (int64_t o = base + gid; o < bend; o += NG) {
        const double *__restrict__ w = W + 18 * o + 3 * r;          // row r of W_o (6 x 3)
        const int p = B.obs_lm[o];
        const double *__restrict__ hp = B.Hinv + 6 * (int64_t)p;
        const double *__restrict__ z = B.zp + 3 * (int64_t)p;
        const double w0 = w[0], w1 = w[1], w2 = w[2];
        const double h00 = hp[0], h01 = hp[1], h02 = hp[2], h11 = hp[3], h12 = hp[4], h22 = hp[5];
        double *__restrict__ d = &Bs[18 * (o - base)];
        d[r] = h00 * w0 + h01 * w1 + h02 * w2;                      // B[j][r] = sum_k Hinv[j][k] W[r][k]
        d[6 + r] = h01 * w0 + h11 * w1 + h12 * w2;
        d[12 + r] = h02 * w0 + h12 * w1 + h22 * w2;
        gacc += w0 * z[0] + w1 * z[1] + w2 * z[2];
      }
    __syncthreads();
    while (ob[0] < (int)bend) {                                     // (INT32_MAX: list exhausted / idle lane group)
      if (ob[NP - 1] < (int)bend) {                                 // the next NP pairs all belong to this batch (the list ascends)
        int na[NP], nb2[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const int64_t q = e + (int64_t)(NP + k) * G;
          const bool in = q < e1;
          na[k] = in ? B.op_a[q] : 0; nb2[k] = in ? B.op_b[q] : INT32_MAX;
        }
        double a[NP][3];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const double *__restrict__ wa = W + 18 * (int64_t)oa[k] + 3 * r;
          a[k][0] = wa[0]; a[k][1] = wa[1]; a[k][2] = wa[2];
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const double *__restrict__ d = &Bs[18 * (ob[k] - (int)base)];
#pragma unroll
          for (int c = 0; c < 6; ++c) acc[c] += a[k][0] * d[c] + a[k][1] * d[6 + c] + a[k][2] * d[12 + c];
        }
        e += (int64_t)NP * G;
#pragma unroll
        for (int k = 0; k < NP; ++k) { oa[k] = na[k]; ob[k] = nb2[k]; }
      } else {                                                      // the batch ends inside the window: one pair, shift
        const int64_t q = e + (int64_t)NP * G;
        const bool in = q < e1;
        const int na = in ? B.op_a[q] : 0, nb2 = in ? B.op_b[q] : INT32_MAX;
        const double *__restrict__ wa = W + 18 * (int64_t)oa[0] + 3 * r;
        const double a0 = wa[0], a1 = wa[1], a2 = wa[2];
        const double *__restrict__ d = &Bs[18 * (ob[0] - (int)base)];
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] += a0 * d[c] + a1 * d[6 + c] + a2 * d[12 + c];
        e += G;
#pragma unroll
        for (int k = 0; k + 1 < NP; ++k) { oa[k] = oa[k + 1]; ob[k] = ob[k + 1]; }
        oa[NP - 1] = na; ob[NP - 1] = nb2;
      }
    }
  }
  __syncthreads();
  double (*part)[36] = reinterpret_cast<double (*)[36]>(Bs);
  if (lane_on) {
#pragma unroll
    for (int c = 0; c < 6; ++c) part[gid][6 * r + c] = acc[c];
    gpart[gid][r] = gacc;
  }
  __syncthreads();
  for (int x = threadIdx.x; x < 36 * nT; x += NW * 64) {
    const int tt = x / 36, el = x - 36 * tt;
    double sacc = 0;
    for (int q = 0; q < G; ++q) sacc += part[tt * G + q][el];
    const int64_t bk = B.tgt_blk[t0 + tt];
    Hred[36 * bk + el] = H[36 * bk + el] - sacc;
  }
  if (threadIdx.x < 6) {
    double sacc = 0;
    for (int q = 0; q < NG; ++q) sacc += gpart[q][threadIdx.x];
    bred[6 * (int64_t)col + threadIdx.x] = b[6 * (int64_t)col + threadIdx.x] - sacc;
  }
  (void)blk;
}


// The camera side of the eliminated observations: H_cc += sum w J_c^T J_c, b_c -= sum w J_c^T r, and the coupling block
// W = J_c^T w J_p of every observation.  Four waves per camera column, one observation per lane and step (the camera's
// observations are contiguous: the pose is the same for the whole workgroup, pixel and weight stream in, the landmark is a
// 32-byte gather); a wave's 64 coupling blocks go through LDS so that they leave as one contiguous 9216-byte store; the 27
// sums are combined across the lanes and waves in a fixed order.  Runs after the generic linearisation, which has written
// the block with the camera's other factors (odometry, priors).  (Six lanes per observation, each keeping one row, were
// measured slower: the factor evaluation is what this kernel spends its time on.)  Held to three waves per SIMD: the compiler
// takes 219 VGPRs when left alone (two waves) and needs 167 -- without scratch -- when told; linearise phase of cfg 3 0.60 -> 0.54 ms
// (four waves: 196 bytes of scratch).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_ba_cameras(DevPlan P, const double *__restrict__ vals, double *__restrict__ W, double *__restrict__ Hblk,
                                                   double *__restrict__ bvec) {
  __shared__ __attribute__((aligned(16))) double wst[4][64 * 18];
  __shared__ double red[4][27];
  const BaPlan &B = P.ba;
  const int i = blockIdx.x;
  const int col = B.cam_col[i];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t o0 = B.cam_ptr[i], o1 = B.cam_ptr[i + 1];
  const Pose X = load_pose(vals + 8 * (int64_t)B.obs_cam[o0]);
  double h[21], gv[6];
#pragma unroll
  for (int k = 0; k < 21; ++k) h[k] = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) gv[k] = 0;
  for (int64_t base = o0 + 64 * wave; base < o1; base += 256) {       // (wave-uniform loop: the LDS hand-over below is per wave)
    const int64_t o = base + lane;
    if (o < o1) {
      const double *__restrict__ uvw = B.obs_uvw + 3 * o;
      const double *__restrict__ pv = B.pt_val + 3 * (int64_t)B.obs_lm[o];      // (a camera's observations ascend with the landmark number)
      double r[6];
      M6 Jx, Jp;
      reproj_factor<true>(X, V3{pv[0], pv[1], pv[2]}, uvw[0], uvw[1], P.cam, r, Jx, Jp);
      const double w = uvw[2];
      const double a0 = Jp.m[0], a1 = Jp.m[1], a2 = Jp.m[2], b0 = Jp.m[6], b1 = Jp.m[7], b2 = Jp.m[8];
      double *__restrict__ wo = &wst[wave][18 * lane];
      int q = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const double x0 = w * Jx.m[a], x1 = w * Jx.m[6 + a];
        wo[3 * a] = x0 * a0 + x1 * b0; wo[3 * a + 1] = x0 * a1 + x1 * b1; wo[3 * a + 2] = x0 * a2 + x1 * b2;
        gv[a] -= x0 * r[0] + x1 * r[1];
#pragma unroll
        for (int b = 0; b <= a; ++b) h[q++] += x0 * Jx.m[b] + x1 * Jx.m[6 + b];
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int64_t nout = 18 * (o1 - base < 64 ? o1 - base : 64);      // doubles of this wave's step, contiguous in W
    double *__restrict__ dst = W + 18 * base;
    for (int64_t e = 2 * lane; e < nout; e += 128) *reinterpret_cast<double2 *>(dst + e) = *reinterpret_cast<const double2 *>(&wst[wave][e]);
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int k = 0; k < 21; ++k) h[k] = wsum_ba(h[k]);
#pragma unroll
  for (int k = 0; k < 6; ++k) gv[k] = wsum_ba(gv[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 21; ++k) red[wave][k] = h[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) red[wave][21 + k] = gv[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 21; ++k) h[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
#pragma unroll
    for (int k = 0; k < 6; ++k) gv[k] = (red[0][21 + k] + red[1][21 + k]) + (red[2][21 + k] + red[3][21 + k]);
    double *__restrict__ d = Hblk + 36 * (int64_t)col;
    int q = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) { d[6 * a + b] += h[q]; if (b != a) d[6 * b + a] += h[q]; ++q; }
#pragma unroll
    for (int k = 0; k < 6; ++k) bvec[6 * (int64_t)col + k] += gv[k];
  }
}

__global__ __launch_bounds__(256) void k_ba_back(DevPlan P, const double *__restrict__ W, const double *__restrict__ bp, double *__restrict__ x) {
  // x_p = (H_pp + lambda I)^-1 (b_p - sum over its observations W_o^T x_camera)
  const BaPlan &B = P.ba;
  const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (p >= B.n_lm) return;
  if (B.lm_mine && !B.lm_mine[p]) {                                  // another rank's landmark: no step here (its value is gathered from the owner)
    double *__restrict__ xz = x + 6 * ((int64_t)P.nb + p);
#pragma unroll
    for (int k = 0; k < 6; ++k) xz[k] = 0;
    return;
  }
  const double *__restrict__ g = bp + 3 * (int64_t)p;
  double t0 = g[0], t1 = g[1], t2 = g[2];
  for (int64_t q = B.pt_ptr[p]; q < B.pt_ptr[p + 1]; ++q) {
    const int o = B.pt_obs[q];
    const int col = B.obs_col[o];
    if (col < 0) continue;
    const double *__restrict__ w = W + 18 * (int64_t)o;
    const double *__restrict__ xc = x + 6 * (int64_t)col;
#pragma unroll
    for (int i = 0; i < 6; ++i) { const double xi = xc[i]; t0 -= w[3 * i] * xi; t1 -= w[3 * i + 1] * xi; t2 -= w[3 * i + 2] * xi; }
  }
  const double *__restrict__ h = B.Hinv + 6 * (int64_t)p;          // h00 h01 h02 h11 h12 h22
  double *__restrict__ xo = x + 6 * ((int64_t)P.nb + p);
  xo[0] = h[0] * t0 + h[1] * t1 + h[2] * t2; xo[1] = h[1] * t0 + h[3] * t1 + h[4] * t2; xo[2] = h[2] * t0 + h[4] * t1 + h[5] * t2;
  xo[3] = 0; xo[4] = 0; xo[5] = 0;
}

__global__ void k_copy_ba(const double *__restrict__ src, double *__restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
}  // namespace

#ifndef BA_NP
#define BA_NP 2
#define BA_OCC 5
#endif
static inline int cdiv_ba(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

int ba_linearize_blocks(const DevPlan &P) { return P.ba.n_lm > 0 ? cdiv_ba(P.ba.n_lm, 256) : 0; }

void launch_ba_linearize(const DevPlan &P, const double *vals, double *W, double *Hpp, double *bp, double *Hblk, double *bvec, double *chi_partial, hipStream_t s) {
  hipLaunchKernelGGL(k_ba_linearize, dim3(cdiv_ba(P.ba.n_lm, 256)), dim3(256), 0, s, P, vals, Hpp, bp, bvec, chi_partial);
  if (P.ba.n_cam > 0) hipLaunchKernelGGL(k_ba_cameras, dim3(P.ba.n_cam), dim3(256), 0, s, P, vals, W, Hblk, bvec);
}

void launch_ba_reduce(const DevPlan &P, const double *W, const double *Hpp, const double *bp, const double *H, const double *b,
                      double *Hred, double *bred, const double *lambda_p, int *fail_flag, hipStream_t s) {
  const BaPlan &B = P.ba;
  hipLaunchKernelGGL(k_ba_points, dim3(cdiv_ba(B.n_lm, 256)), dim3(256), 0, s, P, Hpp, bp, lambda_p, fail_flag);
  // blocks without landmark terms (odometry-only pairs, non-camera variables) are taken over as they are
  const int64_t nh = 36 * P.n_hblocks, nbv = 6 * (int64_t)P.nb;
  hipLaunchKernelGGL(k_copy_ba, dim3((unsigned)std::min<int64_t>(2048, (nh + 255) / 256)), dim3(256), 0, s, H, Hred, nh);
  hipLaunchKernelGGL(k_copy_ba, dim3((unsigned)std::min<int64_t>(2048, (nbv + 255) / 256)), dim3(256), 0, s, b, bred, nbv);
  // one workgroup per column camera (all of its blocks at once); cameras with more blocks than that kernel has slots: block by
  // block -- short lists one wave per block, long lists four waves splitting the list
  // (pairs in flight per lane group, cfg 3 factor phase: 1 -> 1.62 ms, 2 -> 1.47, 4 -> 1.46, 6 / 8 -> 1.43; block-by-block kernel: 1.66)
  if (B.n_cam_list > 0) hipLaunchKernelGGL((k_ba_schur_cam<8, 8>), dim3(B.n_cam_list), dim3(512), 0, s, P, W, H, Hred, b, bred);
  if (B.n_tgt_small > 0) hipLaunchKernelGGL((k_ba_schur<1, BA_NP, BA_OCC>), dim3(B.n_tgt_small), dim3(64), 0, s, P, W, H, Hred, b, bred, B.tgt_list);
  if (B.n_tgt_list > B.n_tgt_small) hipLaunchKernelGGL((k_ba_schur<4, BA_NP, BA_OCC>), dim3(B.n_tgt_list - B.n_tgt_small), dim3(256), 0, s, P, W, H, Hred, b, bred, B.tgt_list + B.n_tgt_small);
}

void launch_ba_back(const DevPlan &P, const double *W, const double *bp, double *x, hipStream_t s) {
  hipLaunchKernelGGL(k_ba_back, dim3(cdiv_ba(P.ba.n_lm, 256)), dim3(256), 0, s, P, W, bp, x);
}

}  // namespace fgo
