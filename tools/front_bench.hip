// Go / no-go microbenchmark for a dense-front (multifrontal-style) update on the WIDE levels (VERDICT r5 "next round" #1).
//
// Today the external updates of a panel level are gathered per TARGET block: L[t] -= sum over source columns L[a] L[b]^T, one 6x6x6
// product per (target, source column) -- 13.7 M of them sourced from panel columns at cfg 2, applied by k_chol_acc / k_chol_acc2 at
// ~15 per ns.  The dense alternative: per front (= panel: m <= 16 columns, r block rows below its triangle) ONE update matrix
// U = X X^T (X = the r x m off-triangle rows, zero where a column lacks a row) on f64 MFMA, r (r + 1) / 2 blocks written once, and per
// target an EXTEND-ADD  L[t] = H[t] - sum U blocks  (1.75 M block adds instead of 13.7 M block products).
//
// This tool measures exactly those two kernels on the REAL front shapes and index lists of a graph (tools/symstats FGO_FRONT_DUMP):
//   k_front_syrk   one workgroup (4 waves) per 128 x 128 super-tile of a front's update matrix; X staged through LDS in chunks of 48
//                  scalar columns (gathered from the block-CSC L through the front's block table, zero-filled), v_mfma_f64_16x16x4,
//                  wave w owns tile rows (w, 7 - w) of the super-tile -- 9 tiles on the diagonal super-tile, 16 off it --, results
//                  stored as packed 6x6 blocks (the layout the extend-add reads)
//   k_extend_add   one lane group (6 lanes, a row each) per target block, 10 targets per wave, one 24-byte descriptor per target:
//                  H row or zero, minus the rows of its U blocks in source-panel order (deterministic), stored to L
// and reports per level: SYRK us, dense-equivalent TF/s (m r (r + 1) / 2 block products x 432 flop), extend-add us and TB/s of
// algorithmic bytes (288 B x (U blocks read + targets written + H blocks read)); the sums are what replaces the panel-sourced part of
// the accumulate launches.  Gate: go if SYRK >= 25 TF/s and extend-add >= 3 TB/s.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/front_bench tools/front_bench.hip
//   FGO_FRONT_DUMP=/tmp/fronts.bin tools/symstats 100000 5 4 64 5000 1152921504606846976 && tools/front_bench /tmp/fronts.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int PM = 16;            // columns per panel
// KC: scalar columns of X per LDS chunk (a multiple of 12); LDS row stride KC + 1 doubles: odd -> the 16 rows of an operand fragment
// hit 16 different 8-byte banks.  KC = 24: 2 x 25 KB per workgroup (three workgroups per CU); KC = 48: 2 x 49 KB (one).

struct SyrkItem { int front, si, sj, pad; };                       // super-tile (si, sj), sj <= si, of a front
struct Front { int m, r; long long ubase; long long tab; };        // tab: first entry of the front's [r][PM] block table
struct ExtDesc { long long t, o0; int hidx, n; };                  // target block of L, its U-block list [o0, o0 + n), H block or -1

// rows [row0, row0 + 128) x columns [k0, k0 + KC) of X into LDS (zero-filled beyond the front / where a column lacks the row)
template <int KC>
__device__ __forceinline__ void stage_rows(double *__restrict__ lds, const double *__restrict__ Lv, const int *__restrict__ tab, int r, int m, int row0, int k0) {
  // one thread per (block row, block column, row inside the block): 6 doubles = three 16-byte loads, six LDS writes
  constexpr int LS = KC + 1;
  const int rb0 = row0 / 6, rb1 = (row0 + 128 + 5) / 6, kb0 = k0 / 6, nkb = KC / 6;
  const int nitem = (rb1 - rb0) * nkb * 6;
  for (int it = threadIdx.x; it < nitem; it += blockDim.x) {
    const int i = it % 6, kb = (it / 6) % nkb, rb = rb0 + it / (6 * nkb);
    const int srow = 6 * rb + i - row0;                            // scalar row inside the 128-row window
    if (srow < 0 || srow >= 128) continue;
    double2 a = make_double2(0, 0), b = a, c = a;
    if (kb0 + kb < m && rb < r) {
      const int blk = tab[rb * PM + kb0 + kb];
      if (blk >= 0) {
        const double2 *__restrict__ p = reinterpret_cast<const double2 *>(Lv + 36 * (long long)blk + 6 * i);
        a = p[0]; b = p[1]; c = p[2];
      }
    }
    double *__restrict__ d = lds + srow * LS + 6 * kb;
    d[0] = a.x; d[1] = a.y; d[2] = b.x; d[3] = b.y; d[4] = c.x; d[5] = c.y;
  }
}

// SKIP: use the staircase (a row's first column) to skip leading all-zero K steps per tile row -- off here, measured separately
template <int KC>
__global__ __launch_bounds__(256) void k_front_syrk(const SyrkItem *__restrict__ items, const Front *__restrict__ fronts, const int *__restrict__ tabs,
                                                    const double *__restrict__ Lv, double *__restrict__ U) {
  constexpr int LS = KC + 1;
  extern __shared__ double smem[];
  double *__restrict__ XA = smem, *__restrict__ XB = smem + 128 * LS;   // the A rows (super-row si) and the B rows (super-row sj)
  const SyrkItem it = items[blockIdx.x];
  const Front F = fronts[it.front];
  const int *__restrict__ tab = tabs + F.tab;
  const int R = 6 * F.r, K = 6 * F.m;
  const bool diag = it.si == it.sj;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nn = lane & 15, q = lane >> 4;
  const int ra = wave, rb = 7 - wave;                              // the two tile rows of this wave inside super-row si
  const int rowA0 = 128 * it.si, rowB0 = 128 * it.sj;
  const bool liveA = rowA0 + 16 * ra < R, liveB = rowA0 + 16 * rb < R;   // (wave-uniform)
  const int ncol = min(8, (R - rowB0 + 15) / 16);                  // tile columns that exist in super-column sj
  d4_t Ca[8], Cb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { Ca[j] = d4_t{0, 0, 0, 0}; Cb[j] = d4_t{0, 0, 0, 0}; }
  const double *__restrict__ xb = diag ? XA : XB;
  for (int k0 = 0; k0 < K; k0 += KC) {
    __syncthreads();
    stage_rows<KC>(XA, Lv, tab, F.r, F.m, rowA0, k0);
    if (!diag) stage_rows<KC>(XB, Lv, tab, F.r, F.m, rowB0, k0);
    __syncthreads();
    const int ksteps = min(KC, K - k0 + 3) / 4;                    // (K is a multiple of 6: the last step of an odd chunk reads staged zeros)
    for (int kk = 0; kk < ksteps; ++kk) {
      const int ko = 4 * kk + q;
      const double fa = liveA ? XA[(16 * ra + nn) * LS + ko] : 0.0;
      const double fb = liveB ? XA[(16 * rb + nn) * LS + ko] : 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < ncol) {                                            // (uniform)
          const double b = xb[(16 * j + nn) * LS + ko];
          if (liveA && (!diag || j <= ra)) Ca[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, b, Ca[j], 0, 0, 0);
          if (liveB && (!diag || j <= rb)) Cb[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb, b, Cb[j], 0, 0, 0);
        }
      }
    }
  }
  // store: C layout of v_mfma_f64_16x16x4: register i of lane (nn, q) holds row q + 4 i of column nn
  double *__restrict__ Uf = U + 36 * F.ubase;
  auto store_tile = [&](const d4_t &C, int trow, int j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gi = rowA0 + 16 * trow + q + 4 * i, gj = rowB0 + 16 * j + nn;
      if (gi < R && gj < R) {
        const int bi = gi / 6, bj = gj / 6;
        if (bj <= bi) Uf[36 * ((long long)bi * (bi + 1) / 2 + bj) + 6 * (gi - 6 * bi) + (gj - 6 * bj)] = C[i];
      }
    }
  };
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < ncol) {
      if (liveA && (!diag || j <= ra)) store_tile(Ca[j], ra, j);
      if (liveB && (!diag || j <= rb)) store_tile(Cb[j], rb, j);
    }
  }
}

// ---- second form: the block ids of a thread's scalar row are read ONCE (two 16-byte loads), the next chunk's blocks are requested before
// the current chunk's MFMAs and written to LDS after them (the loads overlap the arithmetic instead of alternating with it), the six K steps
// of a chunk are unrolled (operand fragments of a step are in flight while the previous step's MFMAs issue), and (SKIP) a tile whose rows
// have no block in a chunk's columns yet -- the staircase: a row starts at some column of the panel -- skips that chunk's MFMAs.
// Thread t owns scalar row (t & 127) of the A window and of the B window, block columns 2 (t >> 7), 2 (t >> 7) + 1 of every chunk of 4.
template <bool DIAG, bool SKIP>
__global__ __launch_bounds__(256) void k_front_syrk2(const SyrkItem *__restrict__ items, const Front *__restrict__ fronts, const int *__restrict__ tabs,
                                                     const double *__restrict__ Lv, double *__restrict__ U) {
  constexpr int KC = 24, LS = KC + 1;
  __shared__ double XA[128 * LS], XBs[DIAG ? 1 : 128 * LS];
  __shared__ int cs[16];                                           // first chunk with data per tile row: [0..7] A window, [8..15] B window
  const SyrkItem it = items[blockIdx.x];
  const Front F = fronts[it.front];
  const int *__restrict__ tab = tabs + F.tab;
  const int R = 6 * F.r, K = 6 * F.m;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nn = lane & 15, q = lane >> 4;
  const int ra = wave, rb = 7 - wave;
  const int rowA0 = 128 * it.si, rowB0 = 128 * it.sj;
  const bool liveA = rowA0 + 16 * ra < R, liveB = rowA0 + 16 * rb < R;
  const int ncol = min(8, (R - rowB0 + 15) / 16);
  const int nchunk = (K + KC - 1) / KC;
  // ---- prologue: block ids of this thread's rows, all chunks
  const int srow = threadIdx.x & 127, half = threadIdx.x >> 7;
  constexpr int NB = DIAG ? 1 : 8;
  int idA[8], idB[NB];                                              // [chunk][2]
  const int gA = rowA0 + srow, gB = rowB0 + srow;
  {
    const bool onA = gA < R, onB = !DIAG && gB < R;
    const int4 *__restrict__ pa = reinterpret_cast<const int4 *>(tab + (onA ? gA / 6 : 0) * PM);
    int4 a[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = pa[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      idA[2 * c] = (onA && 4 * c + 2 * half < F.m) ? (half ? a[c].z : a[c].x) : -1;
      idA[2 * c + 1] = (onA && 4 * c + 2 * half + 1 < F.m) ? (half ? a[c].w : a[c].y) : -1;
    }
    if (!DIAG) {
      const int4 *__restrict__ pb = reinterpret_cast<const int4 *>(tab + (onB ? gB / 6 : 0) * PM);
      int4 b[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = pb[c];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        idB[(2 * c) % NB] = (onB && 4 * c + 2 * half < F.m) ? (half ? b[c].z : b[c].x) : -1;
        idB[(2 * c + 1) % NB] = (onB && 4 * c + 2 * half + 1 < F.m) ? (half ? b[c].w : b[c].y) : -1;
      }
    }
  }
  if (SKIP) {
    if (threadIdx.x < 16) cs[threadIdx.x] = 99;
    __syncthreads();
    int fa = 99, fb = 99;
#pragma unroll
    for (int c = 3; c >= 0; --c) {
      if (idA[2 * c] >= 0 || idA[2 * c + 1] >= 0) fa = c;
      if (!DIAG) { if (idB[(2 * c) % NB] >= 0 || idB[(2 * c + 1) % NB] >= 0) fb = c; }
    }
    if (fa < 99) atomicMin(&cs[srow >> 4], fa);
    if (!DIAG && fb < 99) atomicMin(&cs[8 + (srow >> 4)], fb);
  }
  const int i6A = gA % 6, i6B = gB % 6;
  double2 va[6], vb[DIAG ? 1 : 6];
  auto fetch = [&](int ba0, int ba1, int bb0, int bb1) {
    const int ba[2] = {ba0, ba1}, bb[2] = {bb0, bb1};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      va[3 * h] = va[3 * h + 1] = va[3 * h + 2] = make_double2(0, 0);
      if (ba[h] >= 0) { const double2 *__restrict__ p = reinterpret_cast<const double2 *>(Lv + 36 * (long long)ba[h] + 6 * i6A); va[3 * h] = p[0]; va[3 * h + 1] = p[1]; va[3 * h + 2] = p[2]; }
      if (!DIAG) {
        vb[(3 * h) % (DIAG ? 1 : 6)] = vb[(3 * h + 1) % (DIAG ? 1 : 6)] = vb[(3 * h + 2) % (DIAG ? 1 : 6)] = make_double2(0, 0);
        if (bb[h] >= 0) { const double2 *__restrict__ p = reinterpret_cast<const double2 *>(Lv + 36 * (long long)bb[h] + 6 * i6B); vb[(3 * h) % (DIAG ? 1 : 6)] = p[0]; vb[(3 * h + 1) % (DIAG ? 1 : 6)] = p[1]; vb[(3 * h + 2) % (DIAG ? 1 : 6)] = p[2]; }
      }
    }
  };
  // (idA / idB are indexed with the runtime chunk: written out over the four chunks so that every register index is static)
  auto fetch_c = [&](int c) {
    if (c == 0) fetch(idA[0], idA[1], idB[0 % NB], idB[1 % NB]); else if (c == 1) fetch(idA[2], idA[3], idB[2 % NB], idB[3 % NB]);
    else if (c == 2) fetch(idA[4], idA[5], idB[4 % NB], idB[5 % NB]); else fetch(idA[6], idA[7], idB[6 % NB], idB[7 % NB]);
  };
  auto put = [&]() {
    double *__restrict__ da = XA + srow * LS + 12 * half;
#pragma unroll
    for (int e = 0; e < 6; ++e) { da[2 * e] = va[e].x; da[2 * e + 1] = va[e].y; }
    if (!DIAG) {
      double *__restrict__ db = XBs + srow * LS + 12 * half;
#pragma unroll
      for (int e = 0; e < 6; ++e) { db[2 * e] = vb[e % (DIAG ? 1 : 6)].x; db[2 * e + 1] = vb[e % (DIAG ? 1 : 6)].y; }
    }
  };
  // accumulators.  Diagonal super-tile: tile rows (ra, rb = 7 - ra) hold ra + 1 + rb + 1 = 9 lower tiles -> NINE slots, slot u <= ra is tile
  // (ra, u), slot u > ra is tile (rb, u - ra - 1): the slot's tile column is a run-time LDS address, never a register index.
  constexpr int NS = DIAG ? 9 : 16;
  d4_t C[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) C[u] = d4_t{0, 0, 0, 0};
  const double *__restrict__ xb = DIAG ? XA : XBs;
  fetch_c(0);
  put();
  __syncthreads();
  int csU[NS];                                                       // first chunk in which slot u has anything to multiply
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const bool second = DIAG ? u > ra : u >= 8;
    const int j = DIAG ? (second ? u - ra - 1 : u) : (u & 7);
    csU[u] = SKIP ? max(cs[second ? rb : ra], cs[(DIAG ? 0 : 8) + (j & 7)]) : 0;
  }
  for (int c = 0; c < nchunk; ++c) {
    if (c + 1 < nchunk) fetch_c(c + 1);                              // in flight during this chunk's MFMAs
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      const int ko = 4 * kk + q;
      const double fa = liveA ? XA[(16 * ra + nn) * LS + ko] : 0.0;
      const double fb = liveB ? XA[(16 * rb + nn) * LS + ko] : 0.0;
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const bool second = DIAG ? u > ra : u >= 8;
        const int j = DIAG ? (second ? u - ra - 1 : u) : (u & 7);
        const bool live = (second ? liveB : liveA) && j < ncol && (!DIAG || !second || j <= rb);
        if (live && (!SKIP || c >= csU[u])) {
          const double b = xb[(16 * j + nn) * LS + ko];
          C[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(second ? fb : fa, b, C[u], 0, 0, 0);
        }
      }
    }
    if (c + 1 < nchunk) {
      __syncthreads();
      put();
      __syncthreads();
    }
  }
  double *__restrict__ Uf = U + 36 * F.ubase;
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const bool second = DIAG ? u > ra : u >= 8;
    const int j = DIAG ? (second ? u - ra - 1 : u) : (u & 7);
    const int trow = second ? rb : ra;
    if ((second ? liveB : liveA) && j < ncol && (!DIAG || !second || j <= rb)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gi = rowA0 + 16 * trow + q + 4 * i, gj = rowB0 + 16 * j + nn;
        if (gi < R && gj < R) {
          const int bi = gi / 6, bj = gj / 6;
          if (bj <= bi) Uf[36 * ((long long)bi * (bi + 1) / 2 + bj) + 6 * (gi - 6 * bi) + (gj - 6 * bj)] = C[u][i];
        }
      }
    }
  }
}

__global__ __launch_bounds__(64) void k_extend_add(const ExtDesc *__restrict__ desc, long long n_tgt, const long long *__restrict__ uops,
                                                   const double *__restrict__ Hb, const double *__restrict__ U, double *__restrict__ Lv) {
  const int lane = threadIdx.x, g = lane / 6, r = lane - 6 * g;
  const long long idx = (long long)blockIdx.x * 10 + g;
  if (lane >= 60 || idx >= n_tgt) return;
  const ExtDesc d = desc[idx];
  double2 a = make_double2(0, 0), b = a, c = a;
  if (d.hidx >= 0) {
    const double2 *__restrict__ p = reinterpret_cast<const double2 *>(Hb + 36 * (long long)d.hidx + 6 * r);
    a = p[0]; b = p[1]; c = p[2];
  }
  long long o = d.o0;
  const long long o1 = d.o0 + d.n;
  for (; o + 1 < o1; o += 2) {
    const long long u0 = uops[o], u1 = uops[o + 1];
    const double2 *__restrict__ p0 = reinterpret_cast<const double2 *>(U + 36 * u0 + 6 * r);
    const double2 *__restrict__ p1 = reinterpret_cast<const double2 *>(U + 36 * u1 + 6 * r);
    const double2 x0 = p0[0], y0 = p0[1], z0 = p0[2], x1 = p1[0], y1 = p1[1], z1 = p1[2];
    a.x -= x0.x; a.y -= x0.y; b.x -= y0.x; b.y -= y0.y; c.x -= z0.x; c.y -= z0.y;
    a.x -= x1.x; a.y -= x1.y; b.x -= y1.x; b.y -= y1.y; c.x -= z1.x; c.y -= z1.y;
  }
  if (o < o1) {
    const double2 *__restrict__ p0 = reinterpret_cast<const double2 *>(U + 36 * uops[o] + 6 * r);
    const double2 x0 = p0[0], y0 = p0[1], z0 = p0[2];
    a.x -= x0.x; a.y -= x0.y; b.x -= y0.x; b.y -= y0.y; c.x -= z0.x; c.y -= z0.y;
  }
  double2 *__restrict__ dst = reinterpret_cast<double2 *>(Lv + 36 * d.t + 6 * r);
  dst[0] = a; dst[1] = b; dst[2] = c;
}

template <class T> static T *upload(const std::vector<T> &v) {
  T *p = nullptr;
  CK(hipMalloc(&p, std::max<size_t>(1, v.size()) * sizeof(T)));
  if (!v.empty()) CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

template <int KC>
static void launch_syrk(unsigned grid, const SyrkItem *it, const Front *F, const int *T, const double *L, double *U) {
  static bool once = false;
  const size_t lds = (size_t)2 * 128 * (KC + 1) * sizeof(double);
  if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_front_syrk<KC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
  hipLaunchKernelGGL(k_front_syrk<KC>, dim3(grid), dim3(256), lds, 0, it, F, T, L, U);
}
static int g_kc = 24;
static void syrk(unsigned grid, const SyrkItem *it, const Front *F, const int *T, const double *L, double *U, unsigned n_diag) {
  // (second form: the level's items are sorted [diagonal super-tiles | the others]; one launch each)
  if (g_kc == 2 || g_kc == 3) {
    if (n_diag > 0) { if (g_kc == 2) hipLaunchKernelGGL((k_front_syrk2<true, false>), dim3(n_diag), dim3(256), 0, 0, it, F, T, L, U); else hipLaunchKernelGGL((k_front_syrk2<true, true>), dim3(n_diag), dim3(256), 0, 0, it, F, T, L, U); }
    if (grid > n_diag) { if (g_kc == 2) hipLaunchKernelGGL((k_front_syrk2<false, false>), dim3(grid - n_diag), dim3(256), 0, 0, it + n_diag, F, T, L, U); else hipLaunchKernelGGL((k_front_syrk2<false, true>), dim3(grid - n_diag), dim3(256), 0, 0, it + n_diag, F, T, L, U); }
    return;
  }
  if (g_kc == 48) launch_syrk<48>(grid, it, F, T, L, U); else if (g_kc == 96) launch_syrk<96>(grid, it, F, T, L, U); else launch_syrk<24>(grid, it, F, T, L, U);
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: front_bench <front dump> [sweeps] [max level] [KC = 24 | 48: first form with that chunk; 2: second form (prefetch, unrolled); 3: second form + staircase skip]\n"); return 2; }
  const int sweeps = argc > 2 ? atoi(argv[2]) : 10;
  const int max_level = argc > 3 ? atoi(argv[3]) : 1 << 30;
  if (argc > 4) g_kc = atoi(argv[4]);
  FILE *f = fopen(argv[1], "rb");
  if (!f) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  int64_t hdr[4];
  if (fread(hdr, 8, 4, f) != 4) return 2;
  const int nl = (int)hdr[0];
  const int64_t nnzL = hdr[1], nu = hdr[2];
  std::vector<Front> fronts;
  std::vector<int> tabs;
  std::vector<SyrkItem> items;
  std::vector<ExtDesc> desc;
  std::vector<long long> uops;
  struct Lvl { int f0, f1; long long i0, i1, d0, d1, nops, ndiag; double dense_prod, sparse_prod; };
  std::vector<Lvl> lv((size_t)nl);
  int n_h = 0;
  for (int l = 0; l < nl; ++l) {
    int32_t nf;
    if (fread(&nf, 4, 1, f) != 1) return 2;
    Lvl &L = lv[(size_t)l];
    L.f0 = (int)fronts.size(); L.i0 = (long long)items.size(); L.dense_prod = L.sparse_prod = 0;
    for (int q = 0; q < nf; ++q) {
      int32_t m, r; int64_t ub;
      if (fread(&m, 4, 1, f) != 1 || fread(&r, 4, 1, f) != 1 || fread(&ub, 8, 1, f) != 1) return 2;
      Front F{m, r, ub, (long long)tabs.size()};
      tabs.resize(tabs.size() + (size_t)r * PM);
      if (r > 0 && fread(tabs.data() + F.tab, 4, (size_t)r * PM, f) != (size_t)r * PM) return 2;
      const int ns = (6 * r + 127) / 128;
      for (int si = 0; si < ns; ++si) for (int sj = 0; sj <= si; ++sj) items.push_back(SyrkItem{(int)fronts.size(), si, sj, 0});
      L.dense_prod += (double)m * r * (r + 1) / 2;
      for (int k = 0; k < m; ++k) { double nk = 0; for (int a = 0; a < r; ++a) nk += tabs[F.tab + (size_t)a * PM + k] >= 0; L.sparse_prod += nk * (nk + 1) / 2; }
      fronts.push_back(F);
    }
    L.f1 = (int)fronts.size(); L.i1 = (long long)items.size();
    std::stable_partition(items.begin() + L.i0, items.end(), [](const SyrkItem &x) { return x.si == x.sj; });
    L.ndiag = 0; for (long long q = L.i0; q < L.i1; ++q) L.ndiag += items[(size_t)q].si == items[(size_t)q].sj;
    int64_t nt, no;
    if (fread(&nt, 8, 1, f) != 1 || fread(&no, 8, 1, f) != 1) return 2;
    std::vector<int64_t> tg((size_t)nt), pt((size_t)nt + 1), ub((size_t)no);
    std::vector<int32_t> hh((size_t)nt);
    if (fread(tg.data(), 8, nt, f) != (size_t)nt || fread(hh.data(), 4, nt, f) != (size_t)nt || fread(pt.data(), 8, nt + 1, f) != (size_t)nt + 1 || fread(ub.data(), 8, no, f) != (size_t)no) return 2;
    L.d0 = (long long)desc.size(); L.nops = no;
    const long long ob = (long long)uops.size();
    for (int64_t q = 0; q < nt; ++q) desc.push_back(ExtDesc{tg[q], ob + pt[q], hh[q] ? n_h++ : -1, (int)(pt[q + 1] - pt[q])});
    uops.insert(uops.end(), ub.begin(), ub.end());
    L.d1 = (long long)desc.size();
  }
  fclose(f);
  printf("front dump: %d levels, %zu fronts, %zu SYRK super-tiles, nnz(L) %lld blocks, %lld U blocks (%.0f MB), %zu extend-add targets (%d from H)\n", nl, fronts.size(), items.size(),
         (long long)nnzL, (long long)nu, nu * 288e-6, desc.size(), n_h);
  // L with random entries (|x| < 1), H blocks, U
  double *dL, *dU, *dH;
  CK(hipMalloc(&dL, (size_t)nnzL * 288)); CK(hipMalloc(&dU, (size_t)std::max<int64_t>(1, nu) * 288)); CK(hipMalloc(&dH, (size_t)std::max(1, n_h) * 288));
  {
    std::vector<double> h((size_t)nnzL * 36);
    uint64_t s = 88172645463325252ull;
    for (auto &x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (double)(int64_t)(s >> 11) / (double)(1ull << 52) - 1.0; }
    CK(hipMemcpy(dL, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dH, h.data(), (size_t)std::min<int64_t>(n_h, nnzL) * 288, hipMemcpyHostToDevice));
    CK(hipMemset(dU, 0, (size_t)std::max<int64_t>(1, nu) * 288));
    // correctness of the SYRK on the first, a middle and the last front, against a plain triple loop
    Front *dF = upload(fronts); int *dT = upload(tabs); SyrkItem *dI = upload(items);
    for (const Lvl &L : lv) if (L.i1 > L.i0) syrk((unsigned)(L.i1 - L.i0), dI + L.i0, dF, dT, dL, dU, (unsigned)L.ndiag);
    CK(hipDeviceSynchronize());
    double worst = 0;
    for (size_t fi : {(size_t)0, fronts.size() / 2, fronts.size() / 3, fronts.size() - 1}) {
      const Front &F = fronts[fi];
      if (F.r == 0) continue;
      const int R = 6 * F.r, K = 6 * F.m;
      std::vector<double> X((size_t)R * K, 0.0), Ug((size_t)F.r * (F.r + 1) / 2 * 36);
      for (int a = 0; a < F.r; ++a) for (int k = 0; k < F.m; ++k) { const int b = tabs[F.tab + (size_t)a * PM + k]; if (b >= 0) for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) X[(size_t)(6 * a + i) * K + 6 * k + j] = h[(size_t)b * 36 + 6 * i + j]; }
      CK(hipMemcpy(Ug.data(), dU + 36 * F.ubase, Ug.size() * 8, hipMemcpyDeviceToHost));
      for (int gi = 0; gi < R; ++gi) for (int gj = 0; gj <= gi; ++gj) {
        double sref = 0; for (int k = 0; k < K; ++k) sref += X[(size_t)gi * K + k] * X[(size_t)gj * K + k];
        const int bi = gi / 6, bj = gj / 6;
        worst = std::max(worst, std::fabs(sref - Ug[36 * ((size_t)bi * (bi + 1) / 2 + bj) + 6 * (gi - 6 * bi) + (gj - 6 * bj)]));
      }
    }
    printf("SYRK check (4 fronts, lower triangle, against a host triple loop): max abs error %.3e\n", worst);
    if (!(worst < 1e-10)) { fprintf(stderr, "SYRK WRONG\n"); return 1; }
    CK(hipFree(dF)); CK(hipFree(dT)); CK(hipFree(dI));
  }
  Front *dF = upload(fronts); int *dT = upload(tabs); SyrkItem *dI = upload(items);
  ExtDesc *dD = upload(desc); long long *dO = upload(uops);
  // sweeps in level order: extend-add of level l (reads the U of the levels below), then the SYRK of level l (writes its U)
  std::vector<hipEvent_t> ev((size_t)4 * nl);
  for (auto &e : ev) CK(hipEventCreate(&e));
  std::vector<double> t_syrk((size_t)nl, 0.0), t_ext((size_t)nl, 0.0);
  for (int sw = 0; sw < sweeps + 1; ++sw) {
    for (int l = 0; l < nl && l <= max_level; ++l) {
      const Lvl &L = lv[(size_t)l];
      CK(hipEventRecord(ev[4 * l + 0], 0));
      if (L.d1 > L.d0) hipLaunchKernelGGL(k_extend_add, dim3((unsigned)((L.d1 - L.d0 + 9) / 10)), dim3(64), 0, 0, dD + L.d0, L.d1 - L.d0, dO, dH, dU, dL);
      CK(hipEventRecord(ev[4 * l + 1], 0));
      CK(hipEventRecord(ev[4 * l + 2], 0));
      if (L.i1 > L.i0) syrk((unsigned)(L.i1 - L.i0), dI + L.i0, dF, dT, dL, dU, (unsigned)L.ndiag);
      CK(hipEventRecord(ev[4 * l + 3], 0));
    }
    CK(hipDeviceSynchronize());
    if (sw == 0) continue;                                          // warm-up
    for (int l = 0; l < nl && l <= max_level; ++l) {
      float a = 0, b = 0;
      CK(hipEventElapsedTime(&a, ev[4 * l + 0], ev[4 * l + 1])); CK(hipEventElapsedTime(&b, ev[4 * l + 2], ev[4 * l + 3]));
      t_ext[(size_t)l] += 1e3 * a / sweeps; t_syrk[(size_t)l] += 1e3 * b / sweeps;
    }
  }
  printf("KC = %d scalar columns per LDS chunk\n", g_kc);
  printf("level  fronts  tiles   mean m / r    SYRK us   dense TF/s (sparse-equivalent TF/s)   ext targets  U blocks   extend us   TB/s\n");
  double ts = 0, te = 0, fl = 0, fs = 0, by = 0;
  for (int l = 0; l < nl && l <= max_level; ++l) {
    const Lvl &L = lv[(size_t)l];
    if (L.f1 == L.f0 && L.d1 == L.d0) continue;
    double sm = 0, sr = 0;
    for (int q = L.f0; q < L.f1; ++q) { sm += fronts[q].m; sr += fronts[q].r; }
    const int nf = L.f1 - L.f0;
    long long nh = 0; for (long long q = L.d0; q < L.d1; ++q) nh += desc[(size_t)q].hidx >= 0;
    const double bytes = 288.0 * ((double)L.nops + (double)(L.d1 - L.d0) + (double)nh);
    const double sy = L.i1 > L.i0 ? t_syrk[(size_t)l] : 0, ex = L.d1 > L.d0 ? t_ext[(size_t)l] : 0;
    printf("%5d  %6d  %5lld   %5.1f / %5.1f  %8.1f   %6.2f (%6.2f)   %26lld  %8lld   %8.1f   %5.2f\n", l, nf, L.i1 - L.i0, nf ? sm / nf : 0.0, nf ? sr / nf : 0.0, sy,
           sy > 0 ? 432e-6 * L.dense_prod / sy : 0.0, sy > 0 ? 432e-6 * L.sparse_prod / sy : 0.0, L.d1 - L.d0, L.nops, ex, ex > 0 ? 1e-6 * bytes / ex : 0.0);
    ts += sy; te += ex; fl += L.dense_prod; fs += L.sparse_prod; by += bytes;
  }
  printf("all levels: SYRK %.1f us = %.2f TF/s dense-equivalent (%.2f TF/s of the block products the column patterns need), extend-add %.1f us = %.2f TB/s; together %.1f us\n", ts,
         432e-6 * fl / ts, 432e-6 * fs / ts, te, 1e-6 * by / te, ts + te);
  return 0;
}
