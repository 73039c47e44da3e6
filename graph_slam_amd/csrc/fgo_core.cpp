// libfgo C-ABI (include/fgo.h): context life cycle and the host graph store (vertices, factors, values).
#include <cstring>
#include "fgo_ctx.hpp"

using namespace fgo;

namespace fgo {

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
std::string g_create_error;

void destroy_graphs(fgo_ctx *c) {
  for (auto &g : c->trial_graph)
    if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
  for (auto &pg : c->dist_graph)
    for (auto &g : pg)
      if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
}

void pose_inv7(const double *a, double *o) {
  const double qx = -a[3], qy = -a[4], qz = -a[5], qw = a[6];
  // R(qc) * t
  const double tx = a[0], ty = a[1], tz = a[2];
  const double cx = qy * tz - qz * ty, cy = qz * tx - qx * tz, cz = qx * ty - qy * tx;
  const double rx = tx + 2 * (qw * cx + (qy * cz - qz * cy));
  const double ry = ty + 2 * (qw * cy + (qz * cx - qx * cz));
  const double rz = tz + 2 * (qw * cz + (qx * cy - qy * cx));
  o[0] = -rx; o[1] = -ry; o[2] = -rz; o[3] = qx; o[4] = qy; o[5] = qz; o[6] = qw;
}

int upload_poses(fgo_ctx *c) {
  const int64_t N = (int64_t)c->ids.size();
  std::vector<double> p8((size_t)N * 8, 0.0);
  for (int64_t v = 0; v < N; ++v) std::memcpy(&p8[(size_t)v * 8], &c->poses[(size_t)v * 7], 7 * sizeof(double));
  HIPCHK(c, hipMemcpyAsync(c->d_poses[c->cur].p, p8.data(), sizeof(double) * p8.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->host_poses_newer = false;
  c->dev_poses_newer = false;
  c->lin_valid = false;
  c->cov_factor_valid = false;
  return FGO_OK;
}

int download_poses(fgo_ctx *c) {
  if (!c->dev_poses_newer) return FGO_OK;
  const int64_t N = (int64_t)c->ids.size();
  std::vector<double> p8((size_t)N * 8);
  HIPCHK(c, hipMemcpyAsync(p8.data(), c->d_poses[c->cur].p, sizeof(double) * p8.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int64_t v = 0; v < N; ++v) std::memcpy(&c->poses[(size_t)v * 7], &p8[(size_t)v * 8], 7 * sizeof(double));
  c->dev_poses_newer = false;
  return FGO_OK;
}

int ensure_ready(fgo_ctx *c) {
  if (c->structure_dirty && c->inc.valid) {
    const int rc = refresh_factors(c);
    if (rc < 0) return rc;
  }
  if (c->structure_dirty) { int rc = build(c); if (rc) return rc; }
  if (c->host_poses_newer) { int rc = upload_poses(c); if (rc) return rc; }
  return FGO_OK;
}

}  // namespace fgo

extern "C" {

const char *fgo_version(void) { return "fgo-mi355x 0.1 (gfx950, f64)"; }

int fgo_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return FGO_ENODEV;
  return n;
}

fgo_ctx *fgo_create(const fgo_config *cfg) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = "no HIP device available: libfgo has no CPU fallback";
    return nullptr;
  }
  fgo_ctx *c = new (std::nothrow) fgo_ctx();
  if (!c) { g_create_error = "out of host memory"; return nullptr; }
  if (cfg) c->cfg = *cfg;
  if (c->cfg.device < 0 || c->cfg.device >= ndev) { g_create_error = "bad device ordinal"; delete c; return nullptr; }
  if (hipSetDevice(c->cfg.device) != hipSuccess) { g_create_error = "hipSetDevice failed"; delete c; return nullptr; }
  {
    // (developer switch, tools/spec_interference.py: FGO_DEBUG_CU_MASK = "<keep>/<of>[,p]" -- this context's stream may use `keep` of every `of`
    //  consecutive compute units only (p: the units kept are the LAST ones of each group); FGO_DEBUG_STREAM_PRIO = low | high)
    hipError_t se = hipErrorUnknown;
    const char *cm = std::getenv("FGO_DEBUG_CU_MASK"), *pr = std::getenv("FGO_DEBUG_STREAM_PRIO");
    int keep = 0, of = 0;
    if (cm && std::sscanf(cm, "%d/%d", &keep, &of) == 2 && keep > 0 && of >= keep) {
      hipDeviceProp_t prop;
      int ncu = 256;
      if (hipGetDeviceProperties(&prop, c->cfg.device) == hipSuccess) ncu = prop.multiProcessorCount;
      std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
      const bool last = std::strchr(cm, 'p') != nullptr;
      for (int i = 0; i < ncu; ++i) { const int k = i % of; if (last ? k >= of - keep : k < keep) mask[(size_t)i / 32] |= 1u << (i % 32); }
      se = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data());
    } else if (pr) {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      se = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, pr[0] == 'h' ? hi : lo);
    } else {
      se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    }
    if (se != hipSuccess) { g_create_error = "hipStreamCreate failed"; delete c; return nullptr; }
  }
  for (auto &ev : c->ev) (void)hipEventCreate(&ev);
  if (hipHostMalloc((void **)&c->h_scal, sizeof(double) * 8, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void **)&c->h_fail, sizeof(int), hipHostMallocDefault) != hipSuccess) {
    g_create_error = "hipHostMalloc failed"; fgo_destroy(c); return nullptr;
  }
  const char *g = std::getenv("FGO_GRAPH");
  c->use_graph = !(g && g[0] == '0');
  return c;
}

void fgo_destroy(fgo_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  destroy_graphs(c);
  for (auto &ev : c->ev) if (ev) (void)hipEventDestroy(ev);
  if (c->h_scal) (void)hipHostFree(c->h_scal);
  if (c->h_fail) (void)hipHostFree(c->h_fail);
  if (c->h_flags) (void)hipHostFree(c->h_flags);
  if (c->rccl && rccl_api()) (void)rccl_api()->CommDestroy(c->rccl);
  hipStream_t s = c->stream;
  delete c;   // DevBuf destructors free HBM
  if (s) (void)hipStreamDestroy(s);
}

const char *fgo_last_error(const fgo_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int fgo_add_pose(fgo_ctx *c, int64_t id, const double t[3], const double q[4], int fixed) try {
  if (!c || !t || !q) return FGO_EINVAL;
  if (c->id2idx.count(id)) return fail(c, FGO_EINVAL, "pose id already exists");
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (!(n > 0)) return fail(c, FGO_EINVAL, "zero quaternion");
  if (c->dev_poses_newer) { int rc = download_poses(c); if (rc) return rc; }
  c->id2idx[id] = (int)c->ids.size();
  c->ids.push_back(id);
  c->poses.insert(c->poses.end(), {t[0], t[1], t[2], q[0] / n, q[1] / n, q[2] / n, q[3] / n});
  c->fixed.push_back(fixed ? 1 : 0);
  c->var_kind.push_back(0);
  c->structure_dirty = true;
  c->host_poses_newer = true;
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_add_poses(fgo_ctx *c, int64_t n, const int64_t *ids, const double *poses7, const unsigned char *fixed) try {
  if (!c || n < 0 || !poses7) return FGO_EINVAL;
  const int64_t base = (int64_t)c->ids.size();
  for (int64_t i = 0; i < n; ++i) {
    int rc = fgo_add_pose(c, ids ? ids[i] : base + i, poses7 + 7 * i, poses7 + 7 * i + 3, fixed ? fixed[i] : 0);
    if (rc) return rc;
  }
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_set_pose(fgo_ctx *c, int64_t id, const double t[3], const double q[4]) try {
  if (!c || !t || !q) return FGO_EINVAL;
  auto it = c->id2idx.find(id);
  if (it == c->id2idx.end()) return fail(c, FGO_EINVAL, "unknown pose id");
  if (c->dev_poses_newer) { int rc = download_poses(c); if (rc) return rc; }
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (!(n > 0)) return fail(c, FGO_EINVAL, "zero quaternion");
  double *p = &c->poses[(size_t)it->second * 7];
  p[0] = t[0]; p[1] = t[1]; p[2] = t[2]; p[3] = q[0] / n; p[4] = q[1] / n; p[5] = q[2] / n; p[6] = q[3] / n;
  c->host_poses_newer = true;
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_set_fixed(fgo_ctx *c, int64_t id, int fixed) try {
  if (!c) return FGO_EINVAL;
  auto it = c->id2idx.find(id);
  if (it == c->id2idx.end()) return fail(c, FGO_EINVAL, "unknown pose id");
  if (c->dev_poses_newer) { int rc = download_poses(c); if (rc) return rc; }
  const unsigned char f = fixed ? 1 : 0;
  if (c->fixed[(size_t)it->second] != f) {
    c->fixed[(size_t)it->second] = f;
    c->structure_dirty = true;            // the set of free block columns changed
    c->inc.valid = false;                 // (not something the in-place extension of the incremental mode can express)
    c->host_poses_newer = true;
    c->lin_valid = false;
  }
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_get_pose(fgo_ctx *c, int64_t id, double out7[7]) try {
  if (!c || !out7) return FGO_EINVAL;
  auto it = c->id2idx.find(id);
  if (it == c->id2idx.end()) return fail(c, FGO_EINVAL, "unknown pose id");
  if (c->dev_poses_newer) { int rc = download_poses(c); if (rc) return rc; }
  std::memcpy(out7, &c->poses[(size_t)it->second * 7], 7 * sizeof(double));
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_get_poses(fgo_ctx *c, int64_t n, const int64_t *ids, double *poses7) try {
  if (!c || n < 0 || !poses7) return FGO_EINVAL;
  if (c->dev_poses_newer) { int rc = download_poses(c); if (rc) return rc; }
  for (int64_t i = 0; i < n; ++i) {
    int idx;
    if (ids) {
      auto it = c->id2idx.find(ids[i]);
      if (it == c->id2idx.end()) return fail(c, FGO_EINVAL, "unknown pose id");
      idx = it->second;
    } else {
      if (i >= (int64_t)c->ids.size()) return fail(c, FGO_EINVAL, "pose index out of range");
      idx = (int)i;
    }
    std::memcpy(poses7 + 7 * i, &c->poses[(size_t)idx * 7], 7 * sizeof(double));
  }
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_has_pose(const fgo_ctx *c, int64_t id) { return c && c->id2idx.count(id) ? 1 : 0; }
int64_t fgo_num_poses(const fgo_ctx *c) { return c ? (int64_t)c->ids.size() : 0; }
int64_t fgo_num_edges(const fgo_ctx *c) { return c ? (int64_t)c->ei.size() : 0; }

int fgo_add_edge_se3(fgo_ctx *c, int64_t id_i, int64_t id_j, const double t[3], const double q[4],
                     const double info_ut21[21], int tangent_order) try {
  if (!c || !t || !q || !info_ut21) return FGO_EINVAL;
  if (tangent_order != FGO_TANGENT_G2O && tangent_order != FGO_TANGENT_GTSAM) return fail(c, FGO_EINVAL, "bad tangent order");
  auto a = c->id2idx.find(id_i), b = c->id2idx.find(id_j);
  if (a == c->id2idx.end() || b == c->id2idx.end()) return fail(c, FGO_EINVAL, "edge references an unknown pose id");
  if (a->second == b->second) return fail(c, FGO_EINVAL, "edge endpoints must differ");
  if (c->var_kind[a->second] != 0 || c->var_kind[b->second] != 0) return fail(c, FGO_EINVAL, "SE3 edges connect poses");
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (!(n > 0)) return fail(c, FGO_EINVAL, "zero quaternion");
  c->ei.push_back(a->second); c->ej.push_back(b->second);
  c->meas.insert(c->meas.end(), {t[0], t[1], t[2], q[0] / n, q[1] / n, q[2] / n, q[3] / n});
  c->info.insert(c->info.end(), info_ut21, info_ut21 + 21);
  c->torder.push_back(tangent_order);
  c->structure_dirty = true;
  return FGO_OK;
} FGO_CATCH_INT(c)

// bulk form: validate everything first (an error leaves the graph untouched), then grow each array once and fill it in
// parallel -- the scalar entry point costs two hash look-ups and four vector insertions per edge (10 M edges at cfg 5)
int fgo_add_edges_se3(fgo_ctx *c, int64_t n, const int64_t *id_i, const int64_t *id_j, const double *meas7,
                      const double *info_ut21, int tangent_order) try {
  if (!c || n < 0 || !id_i || !id_j || !meas7 || !info_ut21) return FGO_EINVAL;
  if (tangent_order != FGO_TANGENT_G2O && tangent_order != FGO_TANGENT_GTSAM) return fail(c, FGO_EINVAL, "bad tangent order");
  if (n == 0) return FGO_OK;
  if (n > (int64_t)INT32_MAX || (int64_t)c->ei.size() + n > (int64_t)INT32_MAX) return fail(c, FGO_EINVAL, "more than 2^31-1 edges: edge indices are 32-bit");
  // ids are usually the dense range 0 .. N-1 in insertion order: then the index is the id and no hashing is needed
  const int64_t N = (int64_t)c->ids.size();
  bool dense_ids = true;
  for (int64_t v = 0; v < N && dense_ids; ++v) dense_ids = c->ids[(size_t)v] == v;
  std::vector<int> ia((size_t)n), ib((size_t)n);
  std::atomic<int> err{0};
  parallel_ranges((int)std::min<int64_t>(n, INT32_MAX), 1 << 16, [&](int eb, int ee) {
    for (int64_t e = eb; e < ee; ++e) {
      int a, b;
      if (dense_ids) {
        if (id_i[e] < 0 || id_i[e] >= N || id_j[e] < 0 || id_j[e] >= N) { err.store(1); return; }
        a = (int)id_i[e]; b = (int)id_j[e];
      } else {
        auto pa = c->id2idx.find(id_i[e]), pb = c->id2idx.find(id_j[e]);
        if (pa == c->id2idx.end() || pb == c->id2idx.end()) { err.store(1); return; }
        a = pa->second; b = pb->second;
      }
      if (a == b) { err.store(2); return; }
      if (c->var_kind[a] != 0 || c->var_kind[b] != 0) { err.store(3); return; }
      const double *q = meas7 + 7 * e + 3;
      if (!(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] > 0)) { err.store(4); return; }
      ia[(size_t)e] = a; ib[(size_t)e] = b;
    }
  });
  switch (err.load()) {
    case 1: return fail(c, FGO_EINVAL, "edge references an unknown pose id");
    case 2: return fail(c, FGO_EINVAL, "edge endpoints must differ");
    case 3: return fail(c, FGO_EINVAL, "SE3 edges connect poses");
    case 4: return fail(c, FGO_EINVAL, "zero quaternion");
    default: break;
  }
  const size_t E0 = c->ei.size();
  c->ei.resize(E0 + (size_t)n); c->ej.resize(E0 + (size_t)n);
  c->meas.resize((E0 + (size_t)n) * 7); c->info.resize((E0 + (size_t)n) * 21);
  c->torder.resize(E0 + (size_t)n, tangent_order);
  parallel_ranges((int)std::min<int64_t>(n, INT32_MAX), 1 << 16, [&](int eb, int ee) {
    for (int64_t e = eb; e < ee; ++e) {
      c->ei[E0 + (size_t)e] = ia[(size_t)e]; c->ej[E0 + (size_t)e] = ib[(size_t)e];
      const double *m = meas7 + 7 * e;
      const double nq = std::sqrt(m[3] * m[3] + m[4] * m[4] + m[5] * m[5] + m[6] * m[6]);
      double *o = &c->meas[(E0 + (size_t)e) * 7];
      o[0] = m[0]; o[1] = m[1]; o[2] = m[2]; o[3] = m[3] / nq; o[4] = m[4] / nq; o[5] = m[5] / nq; o[6] = m[6] / nq;
      std::memcpy(&c->info[(E0 + (size_t)e) * 21], info_ut21 + 21 * e, 21 * sizeof(double));
    }
  });
  c->structure_dirty = true;
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_add_prior_pose(fgo_ctx *c, int64_t id, const double t[3], const double q[4], const double info_ut21[21]) try {
  if (!c || !t || !q || !info_ut21) return FGO_EINVAL;
  auto it = c->id2idx.find(id);
  if (it == c->id2idx.end()) return fail(c, FGO_EINVAL, "prior references an unknown pose id");
  if (c->var_kind[it->second] != 0) return fail(c, FGO_EINVAL, "fgo_add_prior_pose needs a Pose3 variable");
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (!(n > 0)) return fail(c, FGO_EINVAL, "zero quaternion");
  c->prior_v.push_back(it->second);
  c->prior_mean.insert(c->prior_mean.end(), {t[0], t[1], t[2], q[0] / n, q[1] / n, q[2] / n, q[3] / n});
  c->prior_info.insert(c->prior_info.end(), info_ut21, info_ut21 + 21);
  c->structure_dirty = true;
  return FGO_OK;
} FGO_CATCH_INT(c)

// non-pose variables share the 7-slot value store: plane = (nx, ny, nz, d), point / vector = (x, y, z), bias = 6 values
static int add_var(fgo_ctx *c, int64_t id, int kind, const double vals7[7]) {
  if (c->id2idx.count(id)) return fail(c, FGO_EINVAL, "variable id already exists");
  if (c->dev_poses_newer) { int rc = download_poses(c); if (rc) return rc; }
  c->id2idx[id] = (int)c->ids.size();
  c->ids.push_back(id);
  c->poses.insert(c->poses.end(), vals7, vals7 + 7);
  c->fixed.push_back(0);
  c->var_kind.push_back(kind);
  c->structure_dirty = true;
  c->host_poses_newer = true;
  return FGO_OK;
}

int fgo_add_plane(fgo_ctx *c, int64_t id, const double abcd[4]) try {
  if (!c || !abcd) return FGO_EINVAL;
  const double n = std::sqrt(abcd[0] * abcd[0] + abcd[1] * abcd[1] + abcd[2] * abcd[2]);
  if (!(n > 0)) return fail(c, FGO_EINVAL, "zero plane normal");
  const double v[7] = {abcd[0] / n, abcd[1] / n, abcd[2] / n, abcd[3], 0, 0, 0};   // OrientedPlane3(a,b,c,d): Unit3 + d
  return add_var(c, id, 1, v);
} FGO_CATCH_INT(c)

int fgo_add_point3(fgo_ctx *c, int64_t id, const double xyz[3]) try {
  if (!c || !xyz) return FGO_EINVAL;
  const double v[7] = {xyz[0], xyz[1], xyz[2], 0, 0, 0, 0};
  return add_var(c, id, 2, v);
} FGO_CATCH_INT(c)

int fgo_add_prior_point3(fgo_ctx *c, int64_t id, const double xyz[3], double sigma) try {
  if (!c || !xyz || !(sigma > 0)) return FGO_EINVAL;
  auto it = c->id2idx.find(id);
  if (it == c->id2idx.end() || c->var_kind[it->second] != 2) return fail(c, FGO_EINVAL, "prior references an unknown point id");
  double info[21] = {0};
  const double w = 1.0 / (sigma * sigma);
  info[0] = w; info[6] = w; info[11] = w;                 // upper-triangular positions of (0,0), (1,1), (2,2)
  c->prior_v.push_back(it->second);
  c->prior_mean.insert(c->prior_mean.end(), {xyz[0], xyz[1], xyz[2], 0, 0, 0, 0});
  c->prior_info.insert(c->prior_info.end(), info, info + 21);
  c->structure_dirty = true;
  return FGO_OK;
} FGO_CATCH_INT(c)

static int add_binary(fgo_ctx *c, int64_t id_i, int kind_i, int64_t id_j, int kind_j, int fkind, const double meas7[7],
                      const double info21[21]) {
  auto a = c->id2idx.find(id_i), b = c->id2idx.find(id_j);
  if (a == c->id2idx.end() || b == c->id2idx.end()) return fail(c, FGO_EINVAL, "factor references an unknown variable id");
  if (c->var_kind[a->second] != kind_i || c->var_kind[b->second] != kind_j) return fail(c, FGO_EINVAL, "factor attached to a variable of the wrong type");
  c->ei.push_back(a->second); c->ej.push_back(b->second);
  c->meas.insert(c->meas.end(), meas7, meas7 + 7);
  c->info.insert(c->info.end(), info21, info21 + 21);
  c->torder.push_back(fkind);
  c->structure_dirty = true;
  return FGO_OK;
}

int fgo_add_plane_factor(fgo_ctx *c, int64_t pose_id, int64_t plane_id, const double z_abcd[4], const double cov_ut6[6]) try {
  if (!c || !z_abcd || !cov_ut6) return FGO_EINVAL;
  const double n = std::sqrt(z_abcd[0] * z_abcd[0] + z_abcd[1] * z_abcd[1] + z_abcd[2] * z_abcd[2]);
  if (!(n > 0)) return fail(c, FGO_EINVAL, "zero plane normal");
  // Gaussian::Covariance(S): information = S^-1 (symmetric 3x3, closed form)
  const double s00 = cov_ut6[0], s01 = cov_ut6[1], s02 = cov_ut6[2], s11 = cov_ut6[3], s12 = cov_ut6[4], s22 = cov_ut6[5];
  const double c00 = s11 * s22 - s12 * s12, c01 = s02 * s12 - s01 * s22, c02 = s01 * s12 - s02 * s11;
  const double det = s00 * c00 + s01 * c01 + s02 * c02;
  if (!(std::fabs(det) > 0)) return fail(c, FGO_EINVAL, "singular plane covariance");
  const double c11 = s00 * s22 - s02 * s02, c12 = s01 * s02 - s00 * s12, c22 = s00 * s11 - s01 * s01;
  double info[21] = {0};
  info[0] = c00 / det; info[1] = c01 / det; info[2] = c02 / det; info[3] = c11 / det; info[4] = c12 / det; info[5] = c22 / det;
  const double m[7] = {z_abcd[0] / n, z_abcd[1] / n, z_abcd[2] / n, z_abcd[3], 0, 0, 0};
  return add_binary(c, pose_id, 0, plane_id, 1, 2, m, info);
} FGO_CATCH_INT(c)

int fgo_set_calib_ds2(fgo_ctx *c, double fx, double fy, double s, double u0, double v0, double k1, double k2, double p1,
                      double p2, const double body_P_sensor7[7]) try {
  if (!c) return FGO_EINVAL;
  CamCalib &K = c->cam;
  K.fx = fx; K.fy = fy; K.s = s; K.u0 = u0; K.v0 = v0; K.k1 = k1; K.k2 = k2; K.p1 = p1; K.p2 = p2;
  const double ident[7] = {0, 0, 0, 0, 0, 0, 1};
  const double *b = body_P_sensor7 ? body_P_sensor7 : ident;
  const double n = std::sqrt(b[3] * b[3] + b[4] * b[4] + b[5] * b[5] + b[6] * b[6]);
  if (!(n > 0)) return fail(c, FGO_EINVAL, "zero quaternion");
  for (int k = 0; k < 3; ++k) K.bps[k] = b[k];
  for (int k = 3; k < 7; ++k) K.bps[k] = b[k] / n;
  // AdjointMap(B^-1) = [[R, 0], [[t]x R, R]] of B^-1
  double bi[7];
  pose_inv7(K.bps, bi);
  const double x = bi[3], y = bi[4], z = bi[5], w = bi[6];
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                       2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  const double S[9] = {0, -bi[2], bi[1], bi[2], 0, -bi[0], -bi[1], bi[0], 0};
  for (int k = 0; k < 36; ++k) K.ad[k] = 0;
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) {
      K.ad[r * 6 + q] = R[r * 3 + q]; K.ad[(3 + r) * 6 + 3 + q] = R[r * 3 + q];
      K.ad[(3 + r) * 6 + q] = S[r * 3] * R[q] + S[r * 3 + 1] * R[3 + q] + S[r * 3 + 2] * R[6 + q];
    }
  c->cam_set = true;
  c->structure_dirty = true;       // the calibration travels inside the device plan
  return FGO_OK;
} FGO_CATCH_INT(c)

// bulk forms for bundle adjustment (config 3 adds 500k points and 5M observations)
int fgo_add_points3(fgo_ctx *c, int64_t n, const int64_t *ids, const double *xyz, double prior_sigma) try {
  if (!c || n < 0 || !ids || !xyz) return FGO_EINVAL;
  for (int64_t k = 0; k < n; ++k) {
    int rc = fgo_add_point3(c, ids[k], xyz + 3 * k);
    if (rc) return rc;
    if (prior_sigma > 0) { rc = fgo_add_prior_point3(c, ids[k], xyz + 3 * k, prior_sigma); if (rc) return rc; }
  }
  return FGO_OK;
} FGO_CATCH_INT(c)
int fgo_add_reprojs(fgo_ctx *c, int64_t n, const int64_t *pose_ids, const int64_t *point_ids, const double *uv, double sigma) try {
  if (!c || n < 0 || !pose_ids || !point_ids || !uv) return FGO_EINVAL;
  for (int64_t k = 0; k < n; ++k) {
    const int rc = fgo_add_reproj(c, pose_ids[k], point_ids[k], uv + 2 * k, sigma);
    if (rc) return rc;
  }
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_add_vec3(fgo_ctx *c, int64_t id, const double xyz[3]) try {
  if (!c || !xyz) return FGO_EINVAL;
  const double v[7] = {xyz[0], xyz[1], xyz[2], 0, 0, 0, 0};
  return add_var(c, id, 3, v);
} FGO_CATCH_INT(c)

int fgo_add_bias(fgo_ctx *c, int64_t id, const double b[6]) try {
  if (!c || !b) return FGO_EINVAL;
  const double v[7] = {b[0], b[1], b[2], b[3], b[4], b[5], 0};
  return add_var(c, id, 4, v);
} FGO_CATCH_INT(c)

static int add_vector_prior(fgo_ctx *c, int64_t id, int kind, int dim, const double *mean, double sigma) {
  if (!c || !mean || !(sigma > 0)) return FGO_EINVAL;
  auto it = c->id2idx.find(id);
  if (it == c->id2idx.end() || c->var_kind[it->second] != kind) return fail(c, FGO_EINVAL, "prior references an unknown variable of that type");
  double info[21] = {0}, m[7] = {0};
  const double w = 1.0 / (sigma * sigma);
  int p = 0;
  for (int r = 0; r < 6; ++r)
    for (int q = r; q < 6; ++q, ++p) if (r == q && r < dim) info[p] = w;
  for (int k = 0; k < dim; ++k) m[k] = mean[k];
  c->prior_v.push_back(it->second);
  c->prior_mean.insert(c->prior_mean.end(), m, m + 7);
  c->prior_info.insert(c->prior_info.end(), info, info + 21);
  c->structure_dirty = true;
  return FGO_OK;
}
int fgo_add_prior_vec3(fgo_ctx *c, int64_t id, const double xyz[3], double sigma) try { return add_vector_prior(c, id, 3, 3, xyz, sigma); } FGO_CATCH_INT(c)
int fgo_add_prior_bias(fgo_ctx *c, int64_t id, const double b[6], double sigma) try { return add_vector_prior(c, id, 4, 6, b, sigma); } FGO_CATCH_INT(c)

int fgo_set_gravity(fgo_ctx *c, const double g[3]) try {
  if (!c || !g) return FGO_EINVAL;
  for (int k = 0; k < 3; ++k) c->gravity[k] = g[k];
  c->structure_dirty = true;
  return FGO_OK;
} FGO_CATCH_INT(c)

// information = preintMeasCov^-1 through a Cholesky factorisation (the covariance must be SPD); symmetrised.  Host-only.
// Exposed so that a caller (and the parity tests: product and oracle are handed the SAME matrix) can see exactly the
// weight a CombinedImuFactor gets -- noiseModel::Gaussian::Covariance(preintMeasCov) in GTSAM terms.
int fgo_preint_information(const fgo_preint *pre, double info225[225]) {
  if (!pre || !info225) return FGO_EINVAL;
  double L[225], inv[225];
  std::memset(L, 0, sizeof(L));
  for (int j = 0; j < 15; ++j) {
    double d = pre->cov[j * 15 + j];
    for (int k = 0; k < j; ++k) d -= L[j * 15 + k] * L[j * 15 + k];
    if (!(d > 0)) return FGO_ENUM;
    L[j * 15 + j] = std::sqrt(d);
    for (int i = j + 1; i < 15; ++i) {
      double s = 0.5 * (pre->cov[i * 15 + j] + pre->cov[j * 15 + i]);
      for (int k = 0; k < j; ++k) s -= L[i * 15 + k] * L[j * 15 + k];
      L[i * 15 + j] = s / L[j * 15 + j];
    }
  }
  for (int col = 0; col < 15; ++col) {                       // solve L L^T x = e_col
    double y[15];
    for (int i = 0; i < 15; ++i) { double s = (i == col) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= L[i * 15 + k] * y[k]; y[i] = s / L[i * 15 + i]; }
    for (int i = 14; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 15; ++k) s -= L[k * 15 + i] * inv[k * 15 + col]; inv[i * 15 + col] = s / L[i * 15 + i]; }
  }
  for (int r = 0; r < 15; ++r) for (int q = 0; q < 15; ++q) info225[r * 15 + q] = 0.5 * (inv[r * 15 + q] + inv[q * 15 + r]);
  return FGO_OK;
}

int fgo_add_imu_combined(fgo_ctx *c, const int64_t ids6[6], const fgo_preint *pre) try {
  if (!c || !ids6 || !pre) return FGO_EINVAL;
  static const int want[6] = {0, 3, 0, 3, 4, 4};            // X V X V B B
  int idx[6];
  for (int u = 0; u < 6; ++u) {
    auto it = c->id2idx.find(ids6[u]);
    if (it == c->id2idx.end()) return fail(c, FGO_EINVAL, "IMU factor references an unknown variable id");
    if (c->var_kind[it->second] != want[u]) return fail(c, FGO_EINVAL, "IMU factor keys must be (pose, velocity, pose, velocity, bias, bias)");
    idx[u] = it->second;
  }
  // k_imu_blocks adds a factor's 21 blocks with one lane each: a variable listed twice (X_i == X_j, B_i == B_j) would make
  // two lanes read-modify-write the same block; GTSAM's factor needs six distinct keys as well
  for (int u = 0; u < 6; ++u)
    for (int w = u + 1; w < 6; ++w)
      if (idx[u] == idx[w]) return fail(c, FGO_EINVAL, "IMU factor keys must be six distinct variables");
  if (!(pre->dt > 0)) return fail(c, FGO_EINVAL, "empty preintegration");
  double inv[225];
  if (fgo_preint_information(pre, inv) != FGO_OK) return fail(c, FGO_ENUM, "preintegrated covariance is not positive definite");
  ImuPayload P;
  std::memset(&P, 0, sizeof(P));
  P.dt = pre->dt;
  std::memcpy(P.dR, pre->dR, sizeof(P.dR)); std::memcpy(P.dp, pre->dp, sizeof(P.dp)); std::memcpy(P.dv, pre->dv, sizeof(P.dv));
  std::memcpy(P.J_R_bg, pre->J_R_bg, sizeof(P.J_R_bg)); std::memcpy(P.J_p_ba, pre->J_p_ba, sizeof(P.J_p_ba));
  std::memcpy(P.J_p_bg, pre->J_p_bg, sizeof(P.J_p_bg)); std::memcpy(P.J_v_ba, pre->J_v_ba, sizeof(P.J_v_ba));
  std::memcpy(P.J_v_bg, pre->J_v_bg, sizeof(P.J_v_bg)); std::memcpy(P.bhat, pre->bhat, sizeof(P.bhat));
  std::memcpy(P.info, inv, sizeof(inv));
  c->imu_payload.push_back(P);
  c->imu_ids.insert(c->imu_ids.end(), idx, idx + 6);
  c->structure_dirty = true;
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_add_reproj(fgo_ctx *c, int64_t pose_id, int64_t point_id, const double uv[2], double sigma) try {
  if (!c || !uv || !(sigma > 0)) return FGO_EINVAL;
  double info[21] = {0};
  info[0] = 1.0 / (sigma * sigma);
  const double m[7] = {uv[0], uv[1], 0, 0, 0, 0, 0};
  return add_binary(c, pose_id, 0, point_id, 2, 3, m, info);
} FGO_CATCH_INT(c)

}  // extern "C"
