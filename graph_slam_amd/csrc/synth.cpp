// Synthetic "Manhattan-3D" pose graphs: the inputs the benchmark and the parity
// tests feed through the C-ABI.  Host-only.  Edge topology follows what CGraphG2O::addNode builds
// (reference g2o/g2o_graph.cpp:159-239): one odometry edge to the predecessor (:174,186), then
// look-back candidates id-2 ... id-1-lookback (:196-205), edge direction (older -> newer); the
// initial estimate is odometry chaining v2 = v1 * T (:118).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <unordered_map>
#include <vector>
#include "../../include/fgo.h"

namespace {

struct Rng {
  std::mt19937_64 g;
  bool has = false;
  double spare = 0;
  explicit Rng(uint64_t s) : g(s) {}
  double uni() { return (double)(g() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
  double normal() {                                                           // Box-Muller
    if (has) { has = false; return spare; }
    double u1 = 0;
    do { u1 = uni(); } while (u1 <= 0);
    const double u2 = uni();
    const double r = std::sqrt(-2.0 * std::log(u1)), a = 6.283185307179586476925 * u2;
    spare = r * std::sin(a); has = true;
    return r * std::cos(a);
  }
};

struct Mat3i { int m[9]; };
inline Mat3i mul(const Mat3i &a, const Mat3i &b) {
  Mat3i c;
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) {
      int s = 0;
      for (int k = 0; k < 3; ++k) s += a.m[r * 3 + k] * b.m[k * 3 + q];
      c.m[r * 3 + q] = s;
    }
  return c;
}
// exact quaternion (x,y,z,w) of a signed permutation rotation matrix
inline void mat_to_quat(const Mat3i &R, double q[4]) {
  const double m00 = R.m[0], m01 = R.m[1], m02 = R.m[2], m10 = R.m[3], m11 = R.m[4], m12 = R.m[5],
               m20 = R.m[6], m21 = R.m[7], m22 = R.m[8];
  const double tr = m00 + m11 + m22;
  double x, y, z, w;
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0) * 2; w = 0.25 * s; x = (m21 - m12) / s; y = (m02 - m20) / s; z = (m10 - m01) / s;
  } else if (m00 > m11 && m00 > m22) {
    double s = std::sqrt(1.0 + m00 - m11 - m22) * 2; w = (m21 - m12) / s; x = 0.25 * s; y = (m01 + m10) / s; z = (m02 + m20) / s;
  } else if (m11 > m22) {
    double s = std::sqrt(1.0 + m11 - m00 - m22) * 2; w = (m02 - m20) / s; x = (m01 + m10) / s; y = 0.25 * s; z = (m12 + m21) / s;
  } else {
    double s = std::sqrt(1.0 + m22 - m00 - m11) * 2; w = (m10 - m01) / s; x = (m02 + m20) / s; y = (m12 + m21) / s; z = 0.25 * s;
  }
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  q[0] = x / n; q[1] = y / n; q[2] = z / n; q[3] = w / n;
}
inline void qmul(const double a[4], const double b[4], double r[4]) {
  r[0] = a[3] * b[0] + b[3] * a[0] + (a[1] * b[2] - a[2] * b[1]);
  r[1] = a[3] * b[1] + b[3] * a[1] + (a[2] * b[0] - a[0] * b[2]);
  r[2] = a[3] * b[2] + b[3] * a[2] + (a[0] * b[1] - a[1] * b[0]);
  r[3] = a[3] * b[3] - (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
}
inline void qrot(const double q[4], const double v[3], double r[3]) {
  // v + 2 w (u x v) + 2 u x (u x v)
  const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
  const double cx = uy * v[2] - uz * v[1], cy = uz * v[0] - ux * v[2], cz = ux * v[1] - uy * v[0];
  r[0] = v[0] + 2 * (w * cx + (uy * cz - uz * cy));
  r[1] = v[1] + 2 * (w * cy + (uz * cx - ux * cz));
  r[2] = v[2] + 2 * (w * cz + (ux * cy - uy * cx));
}
inline void pose_mul(const double a[7], const double b[7], double c[7]) {
  double rt[3], q[4];
  qrot(a + 3, b, rt);
  qmul(a + 3, b + 3, q);
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  c[0] = a[0] + rt[0]; c[1] = a[1] + rt[1]; c[2] = a[2] + rt[2];
  c[3] = q[0] / n; c[4] = q[1] / n; c[5] = q[2] / n; c[6] = q[3] / n;
}
inline void pose_inv(const double a[7], double c[7]) {
  const double qc[4] = {-a[3], -a[4], -a[5], a[6]};
  double r[3];
  qrot(qc, a, r);
  c[0] = -r[0]; c[1] = -r[1]; c[2] = -r[2]; c[3] = qc[0]; c[4] = qc[1]; c[5] = qc[2]; c[6] = qc[3];
}
inline int64_t cell_key(int x, int y, int z) {
  return ((int64_t)(x + (1 << 20)) << 42) | ((int64_t)(y + (1 << 20)) << 21) | (int64_t)(z + (1 << 20));
}

}  // namespace

extern "C" int64_t fgo_synth_manhattan3d(int64_t n_poses, int lookback, int n_loop, uint64_t seed,
                                          double sigma_t, double sigma_q, double *poses_init7,
                                          double *poses_true7, int64_t *id_i, int64_t *id_j, double *meas7,
                                          double *info_ut21, int64_t max_edges) {
  if (n_poses < 1 || lookback < 0 || n_loop < 0 || !poses_init7 || !id_i || !id_j || !meas7 || !info_ut21)
    return FGO_EINVAL;
  const int64_t N = n_poses;
  Rng rng(seed);
  std::vector<double> truth((size_t)N * 7);
  std::vector<int> px(N), py(N), pz(N);
  // 90-degree body-axis rotations
  const Mat3i RX{{1, 0, 0, 0, 0, -1, 0, 1, 0}}, RY{{0, 0, 1, 0, 1, 0, -1, 0, 0}}, RZ{{0, -1, 0, 1, 0, 0, 0, 0, 1}};
  auto transpose = [](const Mat3i &a) { Mat3i t; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) t.m[r * 3 + c] = a.m[c * 3 + r]; return t; };
  Mat3i R{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  int x = 0, y = 0, z = 0;
  std::unordered_map<int64_t, std::vector<int>> cells;
  cells.reserve((size_t)N * 2);
  for (int64_t k = 0; k < N; ++k) {
    if (k > 0) {
      if (rng.uni() < 0.2) {
        const int axis = (int)(rng.uni() * 3.0) % 3;
        const bool neg = rng.uni() < 0.5;
        Mat3i T = axis == 0 ? RX : (axis == 1 ? RY : RZ);
        if (neg) T = transpose(T);
        R = mul(R, T);
      }
      x += R.m[0]; y += R.m[3]; z += R.m[6];   // one unit along body-x
    }
    px[k] = x; py[k] = y; pz[k] = z;
    double *t = &truth[(size_t)k * 7];
    t[0] = x; t[1] = y; t[2] = z;
    mat_to_quat(R, t + 3);
    cells[cell_key(x, y, z)].push_back((int)k);
  }
  if (poses_true7) std::memcpy(poses_true7, truth.data(), sizeof(double) * 7 * (size_t)N);

  int64_t E = 0;
  double info[21];
  {
    int p = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c) info[p++] = (r == c) ? (r < 3 ? 1.0 / (sigma_t * sigma_t) : 1.0 / (sigma_q * sigma_q)) : 0.0;
  }
  auto add_edge = [&](int64_t a, int64_t b) -> bool {
    if (E >= max_edges) return false;
    double ai[7], zt[7], inc[7], d[6];
    pose_inv(&truth[(size_t)a * 7], ai);
    pose_mul(ai, &truth[(size_t)b * 7], zt);
    for (int c = 0; c < 3; ++c) d[c] = sigma_t * rng.normal();
    for (int c = 3; c < 6; ++c) d[c] = sigma_q * rng.normal();
    double w = 1.0 - (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    inc[0] = d[0]; inc[1] = d[1]; inc[2] = d[2];
    if (w < 0) { inc[3] = inc[4] = inc[5] = 0; inc[6] = 1; } else { inc[3] = d[3]; inc[4] = d[4]; inc[5] = d[5]; inc[6] = std::sqrt(w); }
    pose_mul(zt, inc, meas7 + 7 * E);
    std::memcpy(info_ut21 + 21 * E, info, sizeof(info));
    id_i[E] = a; id_j[E] = b; ++E;
    return true;
  };
  std::memcpy(poses_init7, truth.data(), sizeof(double) * 7);
  std::vector<std::pair<int, int>> cand;   // (squared distance, id)
  std::vector<int> chosen;
  for (int64_t k = 1; k < N; ++k) {
    const int64_t e_odo = E;
    if (!add_edge(k - 1, k)) return FGO_ENOMEM;
    pose_mul(poses_init7 + 7 * (k - 1), meas7 + 7 * e_odo, poses_init7 + 7 * k);
    // look-back candidates k-2 .. k-1-lookback (reference: only once the map holds > 3 nodes)
    int64_t lb_lo = k - 2;
    if (k >= 3)
      for (int j = 0; j < lookback && lb_lo >= 0; ++j, --lb_lo)
        if (!add_edge(lb_lo, k)) return FGO_ENOMEM;
    if (k < 3) continue;
    // loop closures: earlier poses within 2 m, nearest first, ties by id
    cand.clear(); chosen.clear();
    if (n_loop > 0)
      for (int dx = -2; dx <= 2; ++dx)
        for (int dy = -2; dy <= 2; ++dy)
          for (int dz = -2; dz <= 2; ++dz) {
            const int d2 = dx * dx + dy * dy + dz * dz;
            if (d2 > 4) continue;
            auto it = cells.find(cell_key(px[k] + dx, py[k] + dy, pz[k] + dz));
            if (it == cells.end()) continue;
            for (int id : it->second)
              if ((int64_t)id <= lb_lo) cand.emplace_back(d2, id);
          }
    std::sort(cand.begin(), cand.end());
    int got = 0;
    for (auto &c : cand) {
      if (got >= n_loop) break;
      if (!add_edge(c.second, k)) return FGO_ENOMEM;
      chosen.push_back(c.second); ++got;
    }
    // not enough revisits: extend the look-back window instead
    for (int64_t v = lb_lo; v >= 0 && got < n_loop; --v) {
      if (std::find(chosen.begin(), chosen.end(), (int)v) != chosen.end()) continue;
      if (!add_edge(v, k)) return FGO_ENOMEM;
      ++got;
    }
  }
  return E;
}

extern "C" int fgo_shard_range(int64_t n, int rank, int world, int64_t *lo, int64_t *hi) {
  if (world < 1 || rank < 0 || rank >= world || n < 0 || !lo || !hi) return FGO_EINVAL;
  const int64_t q = n / world, r = n % world;
  *lo = rank * q + (rank < r ? rank : r);
  *hi = *lo + q + (rank < r ? 1 : 0);
  return FGO_OK;
}
