// C-ABI of libfgo (include/fgo.h): host graph store, structure build, device upload and the
// Levenberg-Marquardt controller that drives the HIP kernels.
//
// LM semantics restated from g2o's OptimizationAlgorithmLevenberg as the reference configures it
// (g2o/g2o_graph.cpp:65-77: LM over BlockSolver<6,3> over a sparse Cholesky; :241-252: optimize(2) x 10):
//   iteration 0 of a call: lambda = 1e-5 * max|diag H|, nu = 2;   each iteration: up to 10 trials of
//   { (H + lambda I) d = b ; x (+) d ; rho = (chi2 - chi2') / (d.(lambda d + b) + 1e-3) } with
//   accept: lambda *= max(1/3, min(1 - (2 rho - 1)^3, 2/3)), nu = 2;  reject: lambda *= nu, nu *= 2.
// All state stays in HBM; per trial only chi2', scale and the failure flag cross PCIe (24 bytes).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/fgo.h"
#include "device_plan.hpp"
#include "fgo_internal.hpp"

using namespace fgo;

namespace {

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
std::string g_create_error;

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0, cap = 0;
  ~DevBuf() { release(); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; cap = 0; } }
  // contents are NOT preserved.  A buffer that has to grow gets 25 % headroom: graphs that grow by a few variables per
  // update (fgo_isam2_update after every record) then rebuild their structure without a round of hipFree / hipMalloc
  hipError_t alloc(size_t count) {
    if (p && count <= cap) { n = count; return hipSuccess; }
    const bool regrow = p != nullptr;
    release();
    cap = (count ? count : 1) + (regrow ? count / 4 : 0);
    const hipError_t e = hipMalloc((void **)&p, sizeof(T) * cap);
    if (e != hipSuccess) { p = nullptr; cap = 0; return e; }
    n = count;
    return hipSuccess;
  }
  template <class A>
  hipError_t upload(const std::vector<T, A> &h, hipStream_t s) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess) return e;
    if (h.empty()) return hipSuccess;
    return hipMemcpyAsync(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, s);
  }
  void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
};

}  // namespace

struct fgo_ctx {
  fgo_config cfg{};
  std::string err;
  // ---- host graph store
  std::unordered_map<int64_t, int> id2idx;
  std::vector<int64_t> ids;
  std::vector<double> poses;        // 7 per pose
  std::vector<unsigned char> fixed;
  std::vector<int> ei, ej;
  std::vector<double> meas, info;   // 7 / 21 per edge
  std::vector<int> torder;
  std::vector<int> prior_v;         // unary Pose3 priors (GTSAM path)
  std::vector<double> prior_mean, prior_info;
  bool gtsam_mode = false;          // decided at build(): GTSAM-semantics factors + exponential-map retraction
  bool structure_dirty = true;      // vertices / edges added since the last build
  bool host_poses_newer = true;     // host copy must be uploaded before the next device use
  bool dev_poses_newer = false;     // device copy must be downloaded before the next host read
  bool lin_valid = false;           // H/b/chi2 on the device match the current device poses
  // ---- device
  hipStream_t stream = nullptr;
  bool use_graph = true;
  Symbolic S;
  HostSchedule sched;
  DevPlan plan{};
  int64_t n_offdiag = 0;
  DevBuf<int> d_pose_col, d_edge_i, d_edge_j, d_edge_slot, d_he, d_dup_slot, d_rowidx, d_asrc, d_op_a, d_op_b,
      d_acc_targets, d_row_blk, d_row_col, d_task_ptr, d_task_cols, d_fail;
  DevBuf<int64_t> d_he_ptr, d_dup_ptr, d_dup_edges, d_colptr, d_op_ptr, d_op_mid, d_rowptr, d_g2_ptr;
  DevBuf<int> d_g2_tgt, d_g2_b, d_g2_a;
  DevBuf<double> d_ainv, d_partial, d_poses[2], d_H[2], d_b[2], d_x, d_L, d_scal;
  DevBuf<int> d_task_panel, d_panel_task, d_ptri_blk, d_prow_ptr, d_prow_idx, d_prow_blk, d_pchunk_panel, d_pchunk_row0,
      d_pchunk_nrows, d_panel_chunk0, d_fchunk_col, d_pcol_fchunk0, d_pcol_fchunkn;
  DevBuf<int64_t> d_row_mid, d_fchunk_e0;
  DevBuf<double> d_fpart, d_bpart, d_ptop, d_imu_blk, d_imu_g;
  DevBuf<int> d_rchunk_panel, d_rchunk_s0, d_ptri_src, d_prow_src;
  DevBuf<int> d_hub_list, d_hub_slice, d_hubm;
  DevBuf<double> d_hub_part;
  DevBuf<PanelDesc> d_pdesc;
  DevBuf<RowChunk> d_rchunks;
  DevBuf<BwdChunk> d_bchunks;
  DevBuf<int64_t> d_prior_ptr;
  DevBuf<int> d_prior_pose, d_var_kind, d_edge_kind;
  std::vector<int> var_kind;        // per variable: 0 pose, 1 plane, 2 point, 3 vec3, 4 bias (factors_device.hpp)
  CamCalib cam{};                   // Cal3DS2 + body_P_sensor for the reprojection factors
  bool cam_set = false;
  std::vector<int> imu_ids;         // 6 internal variable indices per CombinedImuFactor
  std::vector<ImuPayload> imu_payload;
  double gravity[3] = {0.0, 0.0, 9.71};   // MakeSharedD(9.71): gtsam/imu_base.cpp:258-263
  int shard_rank = 0, shard_world = 1;    // multi-GPU: this context owns domain `rank` of `world`
  fgo_allreduce_fn ar_fn = nullptr;       // host-callback transport of the collectives (tests, torch.distributed)
  void *ar_user = nullptr;
  ncclComm_t rccl = nullptr;              // RCCL transport: collectives enqueued on the context's stream (fgo_dist_init_rccl)
  std::vector<int> pose_group;            // per variable: owning rank, world = top, -1 = fixed
  DevBuf<int> d_pose_group, d_imu_list;
  DevBuf<int64_t> d_top_ext0, d_own_op0, d_own_op1, d_top_row0, d_own_row0, d_own_row1;
  DevBuf<unsigned char> d_var_mine;
  DevBuf<double> d_gather;                // [8 N] masked poses (end-of-optimize gather) / [world] scalar exchange
  hipGraphExec_t dist_graph[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [buffer parity][0: domain phase, 1: top phase]
  double xgmi_bytes = 0;                  // bytes this rank handed to the collectives since the last fgo_optimize* call began
  DevBuf<ImuPayload> d_imu;
  DevBuf<int> d_imu_ids, d_imu_inc, d_imu_slot;
  DevBuf<int64_t> d_imu_inc_ptr;
  DevBuf<double> d_prior_minv, d_prior_info;
  int cur = 0;                      // which of the double buffers holds the current estimate
  bool cov_factor_valid = false;    // d_L holds the undamped factor of the current linearisation (marginal covariances)
  std::vector<int> h_pose_col;      // host copy of pose_col (marginal covariances)
  // ---- ISAM2 state (fgo_isam2_update): linearisation point and linear solution per variable, variable order
  DevBuf<double> d_theta, d_delta;  // 8 / 6 doubles per variable
  int64_t isam_n = 0;               // variables the state covers (variables added later start at their initial value, delta 0)
  // ---- incremental mode (fgo_isam2_update on a growing graph): the structure is built for the graph PLUS a reserve of
  // phantom variables, each coupled to the `window` variables before it (DESIGN.md "Incremental updates").  New variables
  // claim phantom slots and new factors whose variable pairs already exist in the structure are appended in place
  // (refresh_factors) -- no ordering, no symbolic factorisation, no re-upload of the index lists.
  bool isam_incremental = false;
  int isam_reserve = -1, isam_window = -1;      // -1: defaults (FGO_ISAM_RESERVE / FGO_ISAM_WINDOW or 384 / 64); reserve 0 disables
  struct Incr {
    int64_t NX = 0, N_done = 0, E_done = 0, NI_done = 0, NP_done = 0, E_cap = 0, NI_cap = 0;
    int nb = 0;
    std::vector<int> hidx, pose_col;              // [NX]
    std::vector<uint64_t> ukey;                   // sorted (a << 32 | b), a < b hessian indices: the structure's off-diagonal pairs
    std::vector<int> edge_h, edge_slot;           // per edge: pair index (-1: none), H slot (-1: none / duplicate group)
    std::vector<int> pair_nbin, pair_first;       // per pair: binary factors on it, the first of them
    std::map<int, std::vector<int64_t>> dups;     // pairs carrying more than one binary factor
    std::vector<int64_t> he_ptr, imu_inc_ptr;     // [NX+1] incidence CSRs, kept on the host so that new factors are INSERTED
    std::vector<int> he, imu_inc;                 // (they attach to the newest variables: short suffix to move and to upload)
    int hub_deg = HUB_DEG;                        // the graph's hub limit (plan_hubs), kept while the structure is extended in place
    size_t hub_cap = 0;                           // hub entries the scratch buffers (d_partial, d_hub_part) were sized for
    bool valid = false;
  } inc;
  DevBuf<double> d_stage;
  int64_t n_priors_dev = 0;
  hipGraphExec_t trial_graph[2] = {nullptr, nullptr};
  hipEvent_t ev[6] = {};
  double *h_scal = nullptr;         // pinned: [0] chi2 cur, [1] scale, [2] maxdiag, [3] lambda, [4] chi2 cand
  int *h_fail = nullptr;
  double chi_cur = 0;
  // ---- results
  fgo_stats last{};
  std::vector<double> tr_chi2, tr_lambda;
};

namespace {

int fail(fgo_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  return code;
}
#define HIPCHK(c, call)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(c, e_ == hipErrorOutOfMemory ? FGO_ENOMEM : FGO_ENODEV,                       \
                  std::string(#call) + ": " + hipGetErrorString(e_));                           \
  } while (0)

void destroy_graphs(fgo_ctx *c) {
  for (auto &g : c->trial_graph)
    if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
  for (auto &pg : c->dist_graph)
    for (auto &g : pg)
      if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
}

// ---- RCCL, resolved at run time: single-GPU users need no RCCL, and a process that already carries one (torch) keeps it
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi *rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {std::getenv("FGO_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      if (!n) continue;
      api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (!api.lib) return;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce) { dlclose(api.lib); api.lib = nullptr; }
  });
  return api.lib ? &api : nullptr;
}

// sum `n` doubles at `buf` (device) over the ranks, in place.  RCCL: enqueued on the context's stream, no host
// synchronisation.  Hook transport: the stream is drained first, the hook returns when the sum is in place.
int dist_allreduce(fgo_ctx *c, double *buf, int64_t n) {
  if (c->shard_world <= 1 || n <= 0) return FGO_OK;
  c->xgmi_bytes += 8.0 * (double)n;
  if (c->rccl) {
    const ncclResult_t r = rccl_api()->AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, c->rccl, c->stream);
    if (r != ncclSuccess) return fail(c, FGO_ENODEV, std::string("ncclAllReduce: ") + (rccl_api()->GetErrorString ? rccl_api()->GetErrorString(r) : "failed"));
    return FGO_OK;
  }
  if (!c->ar_fn) return fail(c, FGO_ESTATE, "distributed mode needs a transport: fgo_dist_init_rccl or fgo_set_allreduce");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->ar_fn(c->ar_user, buf, n) != 0) return fail(c, FGO_ENODEV, "all-reduce hook failed");
  return FGO_OK;
}

// Linearisation hubs (device_plan.hpp): the degree limit of this graph, one entry per slice of every hub variable, and the
// list of the hubs that have several slices.  deg_limit == 0: choose it.
struct HubPlan {
  int deg_limit = HUB_DEG;
  std::vector<int> var, slice;          // per entry
  std::vector<int> multi;               // 3 per multi-slice hub: variable, first entry, slices
};
void plan_hubs(const std::vector<int64_t> &he_ptr, int64_t NX, int deg_limit, HubPlan &hp) {
  static const int env_limit = std::getenv("FGO_HUB_DEG") ? std::atoi(std::getenv("FGO_HUB_DEG")) : 0;
  if (deg_limit <= 0 && env_limit > 0) deg_limit = env_limit;
  if (deg_limit <= 0) {
    deg_limit = HUB_DEG;
    for (int T = 64; T < HUB_DEG; T *= 2) {
      int64_t n = 0;
      for (int64_t v = 0; v < NX && n <= HUB_MAX_VARS; ++v) n += he_ptr[v + 1] - he_ptr[v] > T;
      if (n <= HUB_MAX_VARS) { deg_limit = T; break; }
    }
  }
  hp.deg_limit = deg_limit;
  hp.var.clear(); hp.slice.clear(); hp.multi.clear();
  for (int64_t v = 0; v < NX; ++v) {
    const int64_t d = he_ptr[v + 1] - he_ptr[v];
    if (d <= deg_limit) continue;
    const int ns = (int)std::min<int64_t>(HUB_MAX_SLICES, (d + HUB_SLICE - 1) / HUB_SLICE);
    if (ns > 1) { hp.multi.push_back((int)v); hp.multi.push_back((int)hp.var.size()); hp.multi.push_back(ns); }
    for (int q = 0; q < ns; ++q) { hp.var.push_back((int)v); hp.slice.push_back(q | (ns << 16)); }
  }
}
int upload_hubs(fgo_ctx *c, const HubPlan &hp, size_t entry_cap);

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


// unary priors: CSR per variable (stable in insertion order) + SoA payload with the inverse mean; `mine` selects the
// priors this rank evaluates (distributed mode)
int upload_priors(fgo_ctx *c, int64_t NX, const std::vector<unsigned char> &mine) {
  hipStream_t s = c->stream;
  const int64_t NPall = (int64_t)c->prior_v.size();
  int64_t NP = 0;
  for (int64_t q = 0; q < NPall; ++q) NP += mine[c->prior_v[q]];
  std::vector<int64_t> prior_ptr((size_t)NX + 1, 0);
  std::vector<int> prior_pose((size_t)NP);
  std::vector<double> prior_minv((size_t)7 * NP), prior_info((size_t)21 * NP);
  for (int64_t q = 0; q < NPall; ++q) if (mine[c->prior_v[q]]) prior_ptr[c->prior_v[q] + 1]++;
  for (int64_t v = 0; v < NX; ++v) prior_ptr[v + 1] += prior_ptr[v];
  std::vector<int64_t> fill(prior_ptr.begin(), prior_ptr.end() - 1);
  for (int64_t q = 0; q < NPall; ++q) {
    if (!mine[c->prior_v[q]]) continue;
    const int64_t o = fill[c->prior_v[q]]++;
    prior_pose[o] = c->prior_v[q];
    double a[7];
    if (c->var_kind[c->prior_v[q]] == 0) pose_inv7(&c->prior_mean[(size_t)q * 7], a);
    else std::memcpy(a, &c->prior_mean[(size_t)q * 7], sizeof(a));     // vector-valued variables: raw mean
    for (int k = 0; k < 7; ++k) prior_minv[(size_t)k * NP + o] = a[k];
    for (int k = 0; k < 21; ++k) prior_info[(size_t)k * NP + o] = c->prior_info[(size_t)q * 21 + k];
  }
  HIPCHK(c, c->d_prior_ptr.upload(prior_ptr, s));
  HIPCHK(c, c->d_prior_pose.upload(prior_pose, s));
  HIPCHK(c, c->d_prior_minv.upload(prior_minv, s));
  HIPCHK(c, c->d_prior_info.upload(prior_info, s));
  HIPCHK(c, hipStreamSynchronize(s));                 // the staging vectors die here
  c->n_priors_dev = NP;
  return FGO_OK;
}

// Structure build: ordering, symbolic factorisation, device upload.  Replaces BlockSolver::buildStructure +
// the CSparse symbolic decomposition g2o redoes on iteration 0 of every optimize() call; here it is cached
// until vertices or edges are added.
int build(fgo_ctx *c) {
  const double t0 = now_s();
  const int64_t N = (int64_t)c->ids.size(), E = (int64_t)c->ei.size();
  // one semantics per context: g2o ([t;q] tangent, VertexSE3 oplus) or GTSAM ([w;v] tangent, Expmap retraction)
  int64_t n_gtsam = 0;
  for (int64_t e = 0; e < E; ++e) n_gtsam += c->torder[e] != FGO_TANGENT_G2O;   // torder doubles as the factor kind
  bool non_pose = false;
  for (int64_t v = 0; v < N; ++v) non_pose |= c->var_kind[v] != 0;
  if (non_pose && n_gtsam != E) return fail(c, FGO_EINVAL, "plane / point / vector variables need a GTSAM-semantics graph");
  for (int64_t e = 0; e < E; ++e)
    if (c->torder[e] == 3 && !c->cam_set) return fail(c, FGO_EINVAL, "reprojection factors need fgo_set_calib_ds2 first");
  if ((n_gtsam != 0 && n_gtsam != E) || (n_gtsam == 0 && E > 0 && !c->prior_v.empty()))
    return fail(c, FGO_EINVAL, "a context holds either g2o-semantics edges or GTSAM-semantics factors, not both");
  c->gtsam_mode = n_gtsam > 0 || !c->prior_v.empty() || non_pose || !c->imu_payload.empty();
  if (!c->imu_payload.empty() && n_gtsam != E) return fail(c, FGO_EINVAL, "IMU factors need a GTSAM-semantics graph");
  if (c->dev_poses_newer) { int rc = download_poses(c); if (rc) return rc; }
  destroy_graphs(c);
  prepare_device_kernels();
  // incremental mode: R phantom variables behind the real ones (free, no factors, identity diagonal)
  static const int env_reserve = std::getenv("FGO_ISAM_RESERVE") ? std::atoi(std::getenv("FGO_ISAM_RESERVE")) : 384;
  static const int env_window = std::getenv("FGO_ISAM_WINDOW") ? std::atoi(std::getenv("FGO_ISAM_WINDOW")) : 64;
  const int isam_reserve = c->isam_reserve >= 0 ? c->isam_reserve : env_reserve;
  const int isam_window = c->isam_window > 0 ? c->isam_window : env_window;
  const int64_t R = (c->isam_incremental && c->gtsam_mode && c->shard_world == 1) ? isam_reserve : 0;
  const int64_t NX = N + R;
  c->inc.valid = false;
  // free-variable (hessian) index per pose
  std::vector<int> hidx((size_t)NX, -1);
  int nfree = 0;
  for (int64_t v = 0; v < NX; ++v) if (v >= N || !c->fixed[v]) hidx[v] = nfree++;
  if (nfree == 0 || (E == 0 && c->prior_v.empty() && c->imu_payload.empty()))
    return fail(c, FGO_ESTATE, "nothing to optimise (no free vertex or no factor)");
  const bool prof = std::getenv("FGO_SYM_PROFILE") != nullptr;
  double tprev = now_s();
  auto lap = [&](const char *what) { if (prof) { const double t = now_s(); std::fprintf(stderr, "[fgo build]    %-28s %.1f ms\n", what, 1e3 * (t - tprev)); tprev = t; } };
  // unique vertex pairs
  struct PairRec { int a, b; int64_t e; };
  std::vector<PairRec> pr;
  pr.reserve((size_t)E);
  for (int64_t e = 0; e < E; ++e) {
    const int a = hidx[c->ei[e]], b = hidx[c->ej[e]];
    if (a < 0 || b < 0 || a == b) continue;
    pr.push_back({std::min(a, b), std::max(a, b), e});
  }
  // the 6-variable IMU factors contribute all 15 variable pairs; encoded as e = -1 - (15 f + pair)
  const int64_t NI = (int64_t)c->imu_payload.size();
  for (int64_t f = 0; f < NI; ++f) {
    int q = 0;
    for (int u = 0; u < 6; ++u)
      for (int w = u + 1; w < 6; ++w, ++q) {
        const int a = hidx[c->imu_ids[6 * f + u]], b = hidx[c->imu_ids[6 * f + w]];
        if (a < 0 || b < 0 || a == b) continue;
        pr.push_back({std::min(a, b), std::max(a, b), -1 - (15 * f + q)});
      }
  }
  constexpr int64_t STRUCT_ONLY = std::numeric_limits<int64_t>::min();     // a pair without a factor (yet)
  for (int64_t k = 0; k < R; ++k)                                            // phantom k couples to the `window` variables before it
    for (int64_t u = std::max<int64_t>(0, N + k - isam_window); u < N + k; ++u)
      if (hidx[u] >= 0) pr.push_back({std::min(hidx[u], hidx[N + k]), std::max(hidx[u], hidx[N + k]), STRUCT_ONLY});
  {   // sort by (a, b, e): counting sort on a, then the (short) runs of equal a in parallel
    std::vector<int64_t> start((size_t)nfree + 1, 0);
    for (const PairRec &x : pr) start[x.a + 1]++;
    for (int i = 0; i < nfree; ++i) start[i + 1] += start[i];
    std::vector<PairRec> sorted(pr.size());
    {
      std::vector<int64_t> fill(start.begin(), start.end() - 1);
      for (const PairRec &x : pr) sorted[fill[x.a]++] = x;
    }
    parallel_ranges(nfree, 4096, [&](int ab, int ae) {
      for (int a = ab; a < ae; ++a)
        std::sort(sorted.begin() + start[a], sorted.begin() + start[a + 1],
                  [](const PairRec &x, const PairRec &y) { return x.b != y.b ? x.b < y.b : x.e < y.e; });
    });
    pr.swap(sorted);
  }
  std::vector<int> ua, ub;            // unique pairs
  std::vector<int64_t> ufirst;        // index in pr of the first member
  for (size_t i = 0; i < pr.size(); ++i)
    if (i == 0 || pr[i].a != pr[i - 1].a || pr[i].b != pr[i - 1].b) { ua.push_back(pr[i].a); ub.push_back(pr[i].b); ufirst.push_back((int64_t)i); }
  ufirst.push_back((int64_t)pr.size());
  const int64_t noff = (int64_t)ua.size();
  c->n_offdiag = noff;
  BlockGraph g;
  g.n = nfree;
  g.xadj.assign((size_t)nfree + 1, 0);
  for (int64_t h = 0; h < noff; ++h) { g.xadj[ua[h] + 1]++; g.xadj[ub[h] + 1]++; }
  for (int i = 0; i < nfree; ++i) g.xadj[i + 1] += g.xadj[i];
  g.adj.resize((size_t)g.xadj[nfree]);
  {
    std::vector<int> fill(g.xadj.begin(), g.xadj.end() - 1);
    for (int64_t h = 0; h < noff; ++h) { g.adj[fill[ua[h]]++] = ub[h]; g.adj[fill[ub[h]]++] = ua[h]; }
  }
  lap("pairs + block graph");
  std::vector<int> perm;
  OrderingOptions oo;
  oo.leaf = c->cfg.nd_leaf > 0 ? c->cfg.nd_leaf : (std::getenv("FGO_ND_LEAF") ? std::atoi(std::getenv("FGO_ND_LEAF")) : 64);
  if (const char *df = std::getenv("FGO_DENSE_FACTOR")) oo.dense_factor = std::atof(df);
  const double t_ord0 = now_s();
  nested_dissection(g, oo, perm);
  const double t_ord1 = now_s();
  lap("ordering");
  if ((int)perm.size() != nfree) return fail(c, FGO_EINVAL, "internal: ordering lost vertices");
  const char *wl = std::getenv("FGO_TASK_WORK");
  // light subtrees (one workgroup each, level 0): flat optimum 1250 .. 10000 on cfg 2 since the panel kernels exist
  const int64_t work_limit = wl ? std::atoll(wl) : 5000;
  Symbolic &S = c->S;
  const char *cl = std::getenv("FGO_CHAIN_WORK");
  // chains become panels (<= PANEL_MAX columns); with the LDS panel kernels the work bound no longer pays
  // (cfg 2: 60000 -> 31.6 it/s, unbounded -> 38.7 it/s)
  const int64_t chain_limit = cl ? std::atoll(cl) : (int64_t)1 << 60;
  const int world = c->shard_world, rank = c->shard_rank;
  build_symbolic(g, perm, work_limit, chain_limit, S, world);
  const int nb = nfree;
  const bool dist = world > 1;
  const int top_col0 = dist ? S.dom_col0[world] : nb;
  const int64_t top_blk0 = dist ? S.colptr[top_col0] : S.nnzL;
  lap("build_symbolic");

  // pose -> elimination position
  std::vector<int> pose_col((size_t)NX, -1);
  for (int64_t v = 0; v < NX; ++v) if (hidx[v] >= 0) pose_col[v] = S.iperm[hidx[v]];
  // L block -> H block: column k's original entries are the graph neighbours of perm[k]; stamp them in a scratch row
  // (per host thread) and read the column's pattern against it
  std::vector<int> asrc((size_t)S.nnzL, -1);
  {
    // pair index of every adjacency entry, in the order the block graph lists them
    std::vector<int> adj_pair(g.adj.size());
    {
      std::vector<int> fill(g.xadj.begin(), g.xadj.end() - 1);
      for (int64_t h = 0; h < noff; ++h) { adj_pair[fill[ua[h]]++] = (int)h; adj_pair[fill[ub[h]]++] = (int)h; }
    }
    // one chunk per host thread: the scratch rows are allocated once per chunk
    parallel_ranges(nb, std::max(2048, (nb + host_threads() - 1) / host_threads()), [&](int kb, int ke) {
      std::vector<int> stamp((size_t)nb, -1), pair_of((size_t)nb, -1);
      for (int k = kb; k < ke; ++k) {
        const int ha = S.perm[k];
        for (int p = g.xadj[ha]; p < g.xadj[ha + 1]; ++p) { const int col = S.iperm[g.adj[p]]; stamp[col] = k; pair_of[col] = adj_pair[p]; }
        asrc[S.colptr[k]] = k;
        for (int64_t t = S.colptr[k] + 1; t < S.colptr[k + 1]; ++t) {
          const int i = S.rowidx[t];
          asrc[t] = stamp[i] == k ? nb + pair_of[i] : -1;
        }
      }
    });
  }
  lap("asrc");
  // panel blocks: where a block's value sits when the panel kernels pick it up -- in L (>= 0: block id; the wide
  // accumulate kernel already applied its external updates), still in H (-2 - H block), or nowhere (-1: fill-in
  // without updates).  Structural, so resolved here instead of by three dependent loads per block on the device.
  auto block_src = [&](int t) -> int {
    if (t < 0) return -1;
    if (t >= top_blk0) return t;                  // distributed: a top block's value arrives in L through the collective
    if (S.op_mid[t] > S.op_ptr[t]) return t;
    return asrc[t] >= 0 ? -2 - asrc[t] : -1;
  };
  std::vector<int> ptri_src(S.ptri_blk.size()), prow_src(S.prow_blk.size());
  for (size_t q = 0; q < S.ptri_blk.size(); ++q) ptri_src[q] = block_src(S.ptri_blk[q]);
  for (size_t q = 0; q < S.prow_blk.size(); ++q) prow_src[q] = block_src(S.prow_blk[q]);
  lap("panel sources");
  // multi-GPU: which factors this context linearises (everything when world == 1).  A variable belongs to the rank whose
  // domain holds its column (group `world` = top, -1 = fixed); a factor to the rank of any of its domain variables (they
  // all lie in one domain: a factor is a clique of the block graph and domains are separated by the top), factors among
  // top / fixed variables only are dealt round-robin.  So a domain variable sees ALL its factors locally (complete
  // diagonal block), a top variable a partial sum -- completed by the collective on the tail of L.
  std::vector<int> &pgroup = c->pose_group;
  pgroup.assign((size_t)NX, -1);
  if (dist)
    for (int64_t v = 0; v < N; ++v)
      if (pose_col[v] >= 0) pgroup[v] = (int)(std::upper_bound(S.dom_col0.begin(), S.dom_col0.begin() + world + 1, pose_col[v]) - S.dom_col0.begin()) - 1;
  auto factor_owner = [&](const int *vars, int nv, int64_t salt) -> int {
    if (!dist) return 0;
    int own = -1;
    for (int q = 0; q < nv; ++q) { const int gq = pgroup[vars[q]]; if (gq >= 0 && gq < world) { if (own >= 0 && own != gq) return -2; own = gq; } }
    return own >= 0 ? own : (int)(salt % world);
  };
  std::vector<unsigned char> edge_mine((size_t)E, 1), imu_mine((size_t)NI, 1);
  if (dist) {
    for (int64_t e = 0; e < E; ++e) {
      const int vars[2] = {c->ei[e], c->ej[e]};
      const int o = factor_owner(vars, 2, e);
      if (o == -2) return fail(c, FGO_EINVAL, "internal: a factor spans two domains");
      edge_mine[e] = o == rank;
    }
    for (int64_t f = 0; f < NI; ++f) {
      const int o = factor_owner(&c->imu_ids[6 * f], 6, f);
      if (o == -2) return fail(c, FGO_EINVAL, "internal: an IMU factor spans two domains");
      imu_mine[f] = o == rank;
    }
  }
  std::vector<int> imu_list;
  for (int64_t f = 0; f < NI; ++f) if (imu_mine[f]) imu_list.push_back((int)f);
  // edge -> slot; duplicate groups
  std::vector<int> edge_slot((size_t)E, -1);
  std::vector<int64_t> dup_ptr{0}, dup_edges;
  std::vector<int> dup_slot;
  std::vector<int> imu_slot((size_t)15 * NI, -1);
  for (int64_t h = 0; h < noff; ++h) {
    const int64_t m0 = ufirst[h], m1 = ufirst[h + 1];
    int64_t nbin = 0;
    for (int64_t m = m0; m < m1; ++m) nbin += pr[m].e >= 0;
    for (int64_t m = m0; m < m1; ++m) {
      const int64_t e = pr[m].e;
      if (e == STRUCT_ONLY) continue;
      if (e < 0) {                                  // IMU pair (u < w): stored transposed when w is eliminated later
        const int64_t idx = -1 - e, f = idx / 15;
        int u = 0, w = 1;
        for (int q = (int)(idx % 15); q > 0; --q) { if (++w == 6) { ++u; w = u + 1; } }
        const int cu = pose_col[c->imu_ids[6 * f + u]], cw = pose_col[c->imu_ids[6 * f + w]];
        imu_slot[idx] = (int)(((nb + h) << 1) | (cw > cu ? 1 : 0));
        continue;
      }
      const int ci = pose_col[c->ei[e]], cj = pose_col[c->ej[e]];
      const int slot = (int)(((nb + h) << 1) | (cj > ci ? 1 : 0));
      if (nbin == 1) edge_slot[e] = slot;
      else if (edge_mine[e]) { dup_edges.push_back(e); dup_slot.push_back(slot); }   // owned members only
    }
    if (nbin > 1 && (int64_t)dup_edges.size() > dup_ptr.back()) dup_ptr.push_back((int64_t)dup_edges.size());
  }
  lap("edge slots");
  if (R > 0) {      // what refresh_factors needs to append factors / claim phantom slots without touching the structure
    fgo_ctx::Incr &I = c->inc;
    I.NX = NX; I.N_done = N; I.E_done = E; I.NI_done = NI; I.NP_done = (int64_t)c->prior_v.size(); I.nb = nb;
    I.hidx = hidx; I.pose_col = pose_col;
    I.ukey.resize((size_t)noff);
    for (int64_t h = 0; h < noff; ++h) I.ukey[h] = ((uint64_t)(uint32_t)ua[h] << 32) | (uint32_t)ub[h];
    I.edge_h.assign((size_t)E, -1);
    I.pair_nbin.assign((size_t)noff, 0); I.pair_first.assign((size_t)noff, -1);
    I.dups.clear();
    for (int64_t h = 0; h < noff; ++h)
      for (int64_t m = ufirst[h]; m < ufirst[h + 1]; ++m) {
        const int64_t e = pr[m].e;
        if (e < 0) continue;                         // IMU pair or structure-only
        I.edge_h[e] = (int)h;
        if (I.pair_nbin[h]++ == 0) I.pair_first[h] = (int)e;
      }
    for (int64_t h = 0; h < noff; ++h)
      if (I.pair_nbin[h] > 1)
        for (int64_t m = ufirst[h]; m < ufirst[h + 1]; ++m) if (pr[m].e >= 0) I.dups[(int)h].push_back(pr[m].e);
    I.edge_slot = edge_slot;
  }
  const bool keep_lists = R > 0;
  // per-variable incidence of the IMU factors
  std::vector<int64_t> imu_inc_ptr((size_t)NX + 1, 0);
  std::vector<int> imu_inc((size_t)6 * imu_list.size());
  {
    for (int f : imu_list) for (int u = 0; u < 6; ++u) imu_inc_ptr[c->imu_ids[6 * (int64_t)f + u] + 1]++;
    for (int64_t v = 0; v < NX; ++v) imu_inc_ptr[v + 1] += imu_inc_ptr[v];
    std::vector<int64_t> fill(imu_inc_ptr.begin(), imu_inc_ptr.end() - 1);
    for (int f : imu_list)
      for (int u = 0; u < 6; ++u) imu_inc[fill[c->imu_ids[6 * (int64_t)f + u]]++] = (int)(((int64_t)f << 3) | u);
  }
  // half-edge lists (owned edges only)
  std::vector<int64_t> he_ptr((size_t)NX + 1, 0);
  int64_t n_mine = 0;
  for (int64_t e = 0; e < E; ++e) if (edge_mine[e]) { he_ptr[c->ei[e] + 1]++; he_ptr[c->ej[e] + 1]++; ++n_mine; }
  for (int64_t v = 0; v < NX; ++v) he_ptr[v + 1] += he_ptr[v];
  std::vector<int> he((size_t)2 * n_mine);
  {
    std::vector<int64_t> fill(he_ptr.begin(), he_ptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) {
      if (!edge_mine[e]) continue;
      he[fill[c->ei[e]]++] = (int)(e << 1);
      he[fill[c->ej[e]]++] = (int)((e << 1) | 1);
    }
  }
  // unary terms (priors, the padding identity of 3-dof variables): the variable's rank; top / fixed variables: rank 0
  std::vector<unsigned char> var_mine((size_t)NX, 1);
  if (dist) for (int64_t v = 0; v < N; ++v) var_mine[v] = (pgroup[v] >= 0 && pgroup[v] < world) ? pgroup[v] == rank : rank == 0;
  // distributed: per top block / top column, where the updates sourced from this rank's domain and from the top start
  std::vector<int64_t> top_ext0, own_op0, own_op1, top_row0, own_row0, own_row1;
  if (dist) {
    const int64_t ntb = S.nnzL - top_blk0;
    const int ntc = nb - top_col0;
    const int lo = S.dom_col0[rank], hi = S.dom_col0[rank + 1];
    top_ext0.resize((size_t)ntb); own_op0.resize((size_t)ntb); own_op1.resize((size_t)ntb);
    top_row0.resize((size_t)ntc); own_row0.resize((size_t)ntc); own_row1.resize((size_t)ntc);
    parallel_ranges((int)std::min<int64_t>(ntb, INT32_MAX), 4096, [&](int qb, int qe) {
      for (int64_t q = qb; q < qe; ++q) {
        const int64_t t = top_blk0 + q;
        const int *a0 = S.op_a.data() + S.op_ptr[t], *a1 = S.op_a.data() + S.op_mid[t];     // external ops, ascending source column
        auto first_col_ge = [&](int col) { return (int64_t)(std::partition_point(a0, a1, [&](int blk) { return S.blkcol[blk] < col; }) - S.op_a.data()); };
        own_op0[q] = first_col_ge(lo); own_op1[q] = first_col_ge(hi); top_ext0[q] = first_col_ge(top_col0);
      }
    });
    for (int q = 0; q < ntc; ++q) {
      const int k = top_col0 + q;
      const int *r0 = S.row_col.data() + S.rowptr[k], *r1 = S.row_col.data() + S.row_mid[k];   // entries outside the column's own panel, ascending
      auto first_ge = [&](int col) { return (int64_t)(std::lower_bound(r0, r1, col) - S.row_col.data()); };
      own_row0[q] = first_ge(lo); own_row1[q] = first_ge(hi); top_row0[q] = first_ge(top_col0);
    }
  }
  lap("half-edge lists");
  if (keep_lists) { c->inc.he_ptr = he_ptr; c->inc.he = he; c->inc.imu_inc_ptr = imu_inc_ptr; c->inc.imu_inc = imu_inc; }
  // edge payload: one 256-byte record per edge (device_plan.hpp EDGE_REC)
  // (incremental mode: room for factors that arrive later)
  const int64_t E_cap = R > 0 ? E + std::max<int64_t>(4096, E / 8) : E;
  const int64_t NI_cap = R > 0 ? NI + std::max<int64_t>(256, NI / 8) : NI;
  std::vector<double, NoInitAlloc<double>> erec((size_t)EDGE_REC * E_cap);   // first touched by the threads that fill it
  parallel_ranges((int)std::min<int64_t>(E, INT32_MAX), 8192, [&](int eb, int ee) {
    for (int64_t e = eb; e < ee; ++e) {
      double *o = &erec[(size_t)EDGE_REC * e];
      if (c->torder[e] <= 1) pose_inv7(&c->meas[(size_t)e * 7], o);          // SE3 factors: inverse measurement
      else std::memcpy(o, &c->meas[(size_t)e * 7], 7 * sizeof(double));     // plane / reprojection: raw payload
      o[7] = 0.0;
      std::memcpy(o + 8, &c->info[(size_t)e * 21], 21 * sizeof(double));
      o[29] = o[30] = o[31] = 0.0;
    }
  });
  lap("edge records");
  const double t1 = now_s();

  // ---- upload
  hipStream_t s = c->stream;
  HIPCHK(c, c->d_pose_col.upload(pose_col, s));
  if (R > 0) {     // capacity first (upload() keeps an allocation that is large enough), so that later factors are appended in place
    HIPCHK(c, c->d_edge_i.alloc((size_t)E_cap)); HIPCHK(c, c->d_edge_j.alloc((size_t)E_cap)); HIPCHK(c, c->d_edge_slot.alloc((size_t)E_cap));
    HIPCHK(c, c->d_edge_kind.alloc((size_t)E_cap)); HIPCHK(c, c->d_he.alloc((size_t)2 * E_cap));
    HIPCHK(c, c->d_imu.alloc((size_t)NI_cap)); HIPCHK(c, c->d_imu_ids.alloc((size_t)6 * NI_cap)); HIPCHK(c, c->d_imu_slot.alloc((size_t)15 * NI_cap));
    HIPCHK(c, c->d_imu_inc.alloc((size_t)6 * NI_cap)); HIPCHK(c, c->d_imu_list.alloc((size_t)NI_cap));
  }
  HIPCHK(c, c->d_edge_i.upload(c->ei, s));
  HIPCHK(c, c->d_edge_j.upload(c->ej, s));
  HIPCHK(c, c->d_edge_slot.upload(edge_slot, s));
  HIPCHK(c, c->d_he_ptr.upload(he_ptr, s));
  HubPlan hubs;
  plan_hubs(he_ptr, NX, 0, hubs);
  const size_t hub_cap = hubs.var.size() + (R > 0 ? 256 : 0);       // entries the scratch buffers have room for
  { const int rc = upload_hubs(c, hubs, hub_cap); if (rc) return rc; }
  if (keep_lists) { c->inc.hub_deg = hubs.deg_limit; c->inc.hub_cap = hub_cap; }
  HIPCHK(c, c->d_he.upload(he, s));
  HIPCHK(c, c->d_dup_ptr.upload(dup_ptr, s));
  HIPCHK(c, c->d_dup_edges.upload(dup_edges, s));
  HIPCHK(c, c->d_dup_slot.upload(dup_slot, s));
  HIPCHK(c, c->d_ainv.upload(erec, s));
  { const int rc = upload_priors(c, NX, var_mine); if (rc) return rc; }
  const int64_t NP = c->n_priors_dev;
  HIPCHK(c, c->d_imu.upload(c->imu_payload, s));
  HIPCHK(c, c->d_imu_ids.upload(c->imu_ids, s));
  HIPCHK(c, c->d_imu_inc_ptr.upload(imu_inc_ptr, s));
  HIPCHK(c, c->d_imu_inc.upload(imu_inc, s));
  HIPCHK(c, c->d_imu_slot.upload(imu_slot, s));
  {
    std::vector<int> vk(c->var_kind);
    vk.resize((size_t)NX, 5);                          // phantoms: kind 5 = no degrees of freedom yet (identity block, x = 0)
    HIPCHK(c, c->d_var_kind.upload(vk, s));
    HIPCHK(c, hipStreamSynchronize(s));
  }
  HIPCHK(c, c->d_edge_kind.upload(c->torder, s));
  HIPCHK(c, c->d_imu_list.upload(imu_list, s));
  HIPCHK(c, c->d_pose_group.upload(pgroup, s));
  HIPCHK(c, c->d_var_mine.upload(var_mine, s));
  HIPCHK(c, c->d_top_ext0.upload(top_ext0, s));
  HIPCHK(c, c->d_own_op0.upload(own_op0, s));
  HIPCHK(c, c->d_own_op1.upload(own_op1, s));
  HIPCHK(c, c->d_top_row0.upload(top_row0, s));
  HIPCHK(c, c->d_own_row0.upload(own_row0, s));
  HIPCHK(c, c->d_own_row1.upload(own_row1, s));
  if (dist) HIPCHK(c, c->d_gather.alloc(std::max<size_t>((size_t)N * 8, 64)));
  HIPCHK(c, c->d_colptr.upload(S.colptr, s));
  HIPCHK(c, c->d_rowidx.upload(S.rowidx, s));
  HIPCHK(c, c->d_asrc.upload(asrc, s));
  HIPCHK(c, c->d_op_ptr.upload(S.op_ptr, s));
  HIPCHK(c, c->d_op_mid.upload(S.op_mid, s));
  HIPCHK(c, c->d_op_a.upload(S.op_a, s));
  HIPCHK(c, c->d_op_b.upload(S.op_b, s));
  HIPCHK(c, c->d_acc_targets.upload(S.acc_targets, s));
  for (auto &a : S.g2_a) if (a < 0) a = (int)S.nnzL;        // absent (row, source) pairs read the zero block
  HIPCHK(c, c->d_g2_tgt.upload(S.g2_tgt, s));
  HIPCHK(c, c->d_g2_ptr.upload(S.g2_ptr, s));
  HIPCHK(c, c->d_g2_b.upload(S.g2_b, s));
  HIPCHK(c, c->d_g2_a.upload(S.g2_a, s));
  HIPCHK(c, c->d_rowptr.upload(S.rowptr, s));
  HIPCHK(c, c->d_row_blk.upload(S.row_blk, s));
  HIPCHK(c, c->d_row_col.upload(S.row_col, s));
  HIPCHK(c, c->d_task_ptr.upload(S.task_ptr, s));
  HIPCHK(c, c->d_task_cols.upload(S.task_cols, s));
  HIPCHK(c, c->d_task_panel.upload(S.task_panel, s));
  HIPCHK(c, c->d_panel_task.upload(S.panel_task, s));
  HIPCHK(c, c->d_ptri_blk.upload(S.ptri_blk, s));
  HIPCHK(c, c->d_prow_ptr.upload(S.prow_ptr, s));
  HIPCHK(c, c->d_prow_idx.upload(S.prow_idx, s));
  HIPCHK(c, c->d_prow_blk.upload(S.prow_blk, s));
  HIPCHK(c, c->d_pchunk_panel.upload(S.pchunk_panel, s));
  HIPCHK(c, c->d_pchunk_row0.upload(S.pchunk_row0, s));
  HIPCHK(c, c->d_pchunk_nrows.upload(S.pchunk_nrows, s));
  HIPCHK(c, c->d_panel_chunk0.upload(S.panel_chunk0, s));
  HIPCHK(c, c->d_row_mid.upload(S.row_mid, s));
  HIPCHK(c, c->d_fchunk_col.upload(S.fchunk_col, s));
  HIPCHK(c, c->d_fchunk_e0.upload(S.fchunk_e0, s));
  HIPCHK(c, c->d_pcol_fchunk0.upload(S.pcol_fchunk0, s));
  HIPCHK(c, c->d_pcol_fchunkn.upload(S.pcol_fchunkn, s));
  {
    std::vector<PanelDesc> pd((size_t)S.n_panels);
    std::vector<int> task_level((size_t)S.task_ptr.size() - 1, 0);
    for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l)
      for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) task_level[t] = (int)l;
    int n_top = 0;                                  // operand-tile slots only for panels that run the panel kernels
    for (int pn = 0; pn < S.n_panels; ++pn) {
      const int t = S.panel_task[pn];
      const int rows = S.prow_ptr[pn + 1] - S.prow_ptr[pn];
      pd[pn] = PanelDesc{t, S.task_ptr[t + 1] - S.task_ptr[t], S.task_ptr[t], S.prow_ptr[pn], rows, S.panel_chunk0[pn],
                         (rows + PANEL_ROWS - 1) / PANEL_ROWS, S.level_panel[task_level[t]] ? n_top++ : -1};
    }
    constexpr int NJ = (6 * PANEL_MAX + 15) / 16;            // tile rows of a full panel: NJ (NJ + 1) / 2 operand tiles of 256 doubles
    HIPCHK(c, c->d_ptop.alloc((size_t)n_top * (NJ * (NJ + 1) / 2) * 256));
    std::vector<RowChunk> rc(S.rchunk_panel.size());
    for (size_t q = 0; q < rc.size(); ++q) {
      const PanelDesc &d = pd[S.rchunk_panel[q]];
      rc[q] = RowChunk{S.rchunk_panel[q], d.m, S.rchunk_s0[q], 6 * d.nrows, d.prow0, d.cols0, d.top, 0};
    }
    std::vector<BwdChunk> bc(S.pchunk_panel.size());
    for (size_t q = 0; q < bc.size(); ++q) bc[q] = BwdChunk{S.pchunk_panel[q], pd[S.pchunk_panel[q]].m, S.pchunk_row0[q], S.pchunk_nrows[q]};
    HIPCHK(c, c->d_pdesc.upload(pd, s));
    HIPCHK(c, c->d_rchunks.upload(rc, s));
    HIPCHK(c, c->d_bchunks.upload(bc, s));
    HIPCHK(c, hipStreamSynchronize(s));       // the staging vectors die at the end of this scope
  }
  HIPCHK(c, c->d_ptri_src.upload(ptri_src, s));
  HIPCHK(c, c->d_prow_src.upload(prow_src, s));
  HIPCHK(c, c->d_rchunk_panel.upload(S.rchunk_panel, s));
  HIPCHK(c, c->d_rchunk_s0.upload(S.rchunk_s0, s));
  HIPCHK(c, c->d_fpart.alloc(S.fchunk_col.size() * 6));
  HIPCHK(c, c->d_bpart.alloc(S.pchunk_panel.size() * PANEL_MAX * 6));
  const size_t hblocks = (size_t)nb + (size_t)noff;
  for (int i = 0; i < 2; ++i) {
    HIPCHK(c, c->d_poses[i].alloc((size_t)NX * 8));
    if (R > 0) HIPCHK(c, hipMemsetAsync(c->d_poses[i].p + (size_t)N * 8, 0, sizeof(double) * (size_t)R * 8, s));
    HIPCHK(c, c->d_H[i].alloc(hblocks * 36));
    HIPCHK(c, c->d_b[i].alloc((size_t)nb * 6));
    HIPCHK(c, hipMemsetAsync(c->d_H[i].p, 0, sizeof(double) * hblocks * 36, s));
  }
  HIPCHK(c, c->d_x.alloc((size_t)nb * 6));
  HIPCHK(c, c->d_L.alloc(((size_t)S.nnzL + 1) * 36));
  HIPCHK(c, hipMemsetAsync(c->d_L.p + (size_t)S.nnzL * 36, 0, sizeof(double) * 36, s));   // the zero block
  HIPCHK(c, c->d_scal.alloc(8));
  HIPCHK(c, c->d_fail.alloc(1));
  HIPCHK(c, hipMemsetAsync(c->d_fail.p, 0, sizeof(int), s));
  // two-pass reduction scratch, sized from the real launch shapes: linearise = ceil(4N/256) lane-group workgroups + one
  // per hub variable (bounded by 2E / HUB_DEG, NOT by N / HUB_DEG) + one per IMU factor; chi2 <= 2048 + ceil(NI/64);
  // maxdiag <= 1024; update / relinearise ceil(N/256)
  const size_t npart = std::max<size_t>({(size_t)4096, (size_t)((NX * 4 + 255) / 256) + hub_cap + (size_t)NI_cap + 64,
                                         (size_t)2048 + (size_t)((NI_cap + 63) / 64) + 64, (size_t)((NX + 255) / 256) + 64});
  HIPCHK(c, c->d_imu_blk.alloc((size_t)NI_cap * 21 * 36));
  HIPCHK(c, c->d_imu_g.alloc((size_t)NI_cap * 36));
  HIPCHK(c, c->d_partial.alloc(npart));
  HIPCHK(c, hipStreamSynchronize(s));

  DevPlan &P = c->plan;
  P.n_poses = NX; P.n_edges = E; P.edge_stride = E_cap; P.nb = nb;
  P.pose_col = c->d_pose_col.p; P.edge_i = c->d_edge_i.p; P.edge_j = c->d_edge_j.p;
  P.ainv = c->d_ainv.p; P.info = c->d_ainv.p + 8; P.edge_slot = c->d_edge_slot.p;
  P.he_ptr = c->d_he_ptr.p; P.he = c->d_he.p;
  P.hub_list = c->d_hub_list.p; P.hub_slice = c->d_hub_slice.p; P.n_hubs = (int)hubs.var.size(); P.hub_deg = hubs.deg_limit;
  P.hub_part = c->d_hub_part.p; P.hubm = c->d_hubm.p; P.n_hub_multi = (int)(hubs.multi.size() / 3);
  P.n_dup_groups = (int64_t)dup_ptr.size() - 1; P.dup_ptr = c->d_dup_ptr.p; P.dup_edges = c->d_dup_edges.p; P.dup_slot = c->d_dup_slot.p;
  P.n_priors = NP; P.prior_ptr = c->d_prior_ptr.p; P.prior_pose = c->d_prior_pose.p;
  P.prior_minv = c->d_prior_minv.p; P.prior_info = c->d_prior_info.p;
  P.var_kind = c->d_var_kind.p; P.edge_kind = c->d_edge_kind.p; P.cam = c->cam;
  P.n_imu = NI; P.imu = c->d_imu.p; P.imu_ids = c->d_imu_ids.p; P.imu_inc_ptr = c->d_imu_inc_ptr.p;
  P.imu_inc = c->d_imu_inc.p; P.imu_slot = c->d_imu_slot.p;
  P.imu_blk = c->d_imu_blk.p; P.imu_g = c->d_imu_g.p; P.imu_f0 = 0; P.imu_fn = (int64_t)imu_list.size();
  P.imu_list = dist ? c->d_imu_list.p : nullptr;
  for (int k = 0; k < 3; ++k) P.gravity[k] = c->gravity[k];
  P.n_hblocks = (int64_t)hblocks;
  P.lin_priors = 1;                               // (the prior CSR above already holds this rank's priors only)
  P.zero_offdiag = dist ? 1 : 0;
  P.dist = dist ? 1 : 0; P.top_col0 = top_col0; P.top_blk0 = top_blk0;
  P.top_ext0 = c->d_top_ext0.p; P.own_op0 = c->d_own_op0.p; P.own_op1 = c->d_own_op1.p;
  P.top_row0 = c->d_top_row0.p; P.own_row0 = c->d_own_row0.p; P.own_row1 = c->d_own_row1.p;
  P.var_mine = dist ? c->d_var_mine.p : nullptr; P.lambda_rank = rank == 0 ? 1 : 0;
  c->sched.world = world; c->sched.rank = rank; c->sched.seg_group = S.seg_group;
  c->sched.n_top_blocks = S.nnzL - top_blk0; c->sched.n_top_cols = nb - top_col0;
  P.colptr = c->d_colptr.p; P.rowidx = c->d_rowidx.p; P.asrc = c->d_asrc.p;
  P.zero_blk = (int)S.nnzL;
  P.op_ptr = c->d_op_ptr.p; P.op_mid = c->d_op_mid.p; P.op_a = c->d_op_a.p; P.op_b = c->d_op_b.p;
  P.acc_targets = c->d_acc_targets.p;
  P.g2_tgt = c->d_g2_tgt.p; P.g2_ptr = c->d_g2_ptr.p; P.g2_b = c->d_g2_b.p; P.g2_a = c->d_g2_a.p;
  c->sched.g2_lvl = S.g2_lvl;
  if (S.g2_ptr.size() <= 1) c->sched.g2_lvl.clear();
  P.rowptr = c->d_rowptr.p; P.row_blk = c->d_row_blk.p; P.row_col = c->d_row_col.p;
  P.task_ptr = c->d_task_ptr.p; P.task_cols = c->d_task_cols.p;
  P.partial = c->d_partial.p;
  P.pp.task_panel = c->d_task_panel.p; P.pp.panel_task = c->d_panel_task.p; P.pp.ptri_blk = c->d_ptri_blk.p;
  P.pp.prow_ptr = c->d_prow_ptr.p; P.pp.prow_idx = c->d_prow_idx.p; P.pp.prow_blk = c->d_prow_blk.p;
  P.pp.pchunk_panel = c->d_pchunk_panel.p; P.pp.pchunk_row0 = c->d_pchunk_row0.p; P.pp.pchunk_nrows = c->d_pchunk_nrows.p;
  P.pp.panel_chunk0 = c->d_panel_chunk0.p; P.pp.row_mid = c->d_row_mid.p; P.pp.fchunk_col = c->d_fchunk_col.p;
  P.pp.fchunk_e0 = c->d_fchunk_e0.p; P.pp.pcol_fchunk0 = c->d_pcol_fchunk0.p; P.pp.pcol_fchunkn = c->d_pcol_fchunkn.p;
  P.pp.fpart = c->d_fpart.p; P.pp.bpart = c->d_bpart.p; P.pp.ptop = c->d_ptop.p;
  P.pp.rchunk_panel = c->d_rchunk_panel.p; P.pp.rchunk_s0 = c->d_rchunk_s0.p;
  P.pp.ptri_src = c->d_ptri_src.p; P.pp.prow_src = c->d_prow_src.p;
  P.pp.pdesc = c->d_pdesc.p; P.pp.rchunks = c->d_rchunks.p; P.pp.bchunks = c->d_bchunks.p;
  c->sched.level_panel = S.level_panel; c->sched.pchunk_ptr = S.pchunk_ptr; c->sched.fchunk_ptr = S.fchunk_ptr; c->sched.rchunk_ptr = S.rchunk_ptr;
  if (std::getenv("FGO_NO_PANELS")) std::fill(c->sched.level_panel.begin(), c->sched.level_panel.end(), 0);
  c->sched.level_leaf = S.level_leaf; c->sched.level_leaf_maxblk = S.level_leaf_maxblk; c->sched.level_leaf_maxops = S.level_leaf_maxops;
  if (std::getenv("FGO_NO_LEAF")) std::fill(c->sched.level_leaf.begin(), c->sched.level_leaf.end(), 0);
  c->sched.n_levels = (int)S.level_ptr.size() - 1;
  c->sched.level_ptr = S.level_ptr;
  c->sched.acc_ptr = S.acc_ptr; c->sched.acc_mid = S.acc_mid;
  c->sched.level_pn0.assign(c->sched.n_levels, 0);
  for (int l = 0; l < c->sched.n_levels; ++l)
    if (S.level_panel[l]) {
      c->sched.level_pn0[l] = S.task_panel[S.level_ptr[l]];
      for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t)
        if (S.task_panel[t] != c->sched.level_pn0[l] + (t - S.level_ptr[l])) return fail(c, FGO_EINVAL, "internal: panel ids of a level are not consecutive");
    }
  c->sched.level_col_ptr.resize(c->sched.n_levels + 1);
  for (int l = 0; l <= c->sched.n_levels; ++l) c->sched.level_col_ptr[l] = S.task_ptr[S.level_ptr[l]];
  c->sched.level_maxcol.assign(c->sched.n_levels, 0);
  c->sched.level_maxrow.assign(c->sched.n_levels, 0);
  c->sched.level_maxtaskcols.assign(c->sched.n_levels, 0);
  for (int l = 0; l < c->sched.n_levels; ++l)
    for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
      c->sched.level_maxtaskcols[l] = std::max(c->sched.level_maxtaskcols[l], S.task_ptr[t + 1] - S.task_ptr[t]);
      for (int q = S.task_ptr[t]; q < S.task_ptr[t + 1]; ++q) {
        const int k = S.task_cols[q];
        c->sched.level_maxcol[l] = std::max(c->sched.level_maxcol[l], (int)(S.colptr[k + 1] - S.colptr[k]));
        c->sched.level_maxrow[l] = std::max(c->sched.level_maxrow[l], (int)(S.rowptr[k + 1] - S.rowptr[k]));
      }
    }
  if (R > 0) { c->inc.E_cap = E_cap; c->inc.NI_cap = NI_cap; c->inc.valid = true; }
  c->cur = 0;
  c->cov_factor_valid = false;
  c->h_pose_col.clear();
  c->structure_dirty = false;
  c->host_poses_newer = true;
  c->lin_valid = false;

  fgo_stats &st = c->last;
  std::memset(&st, 0, sizeof(st));
  st.structure_rebuilt = 1;
  st.t_symbolic = t1 - t0;
  st.t_upload = now_s() - t1;
  st.n_free = nb; st.n_edges = E;
  st.nnz_H_blocks = (int64_t)hblocks; st.nnz_L_blocks = S.nnzL; st.n_update_ops = S.nops;
  st.n_levels = c->sched.n_levels; st.n_tasks = (int)S.task_ptr.size() - 1;
  // algorithmic HBM bytes: factor = read H once + write L once; solve = read L twice;
  // linearise = edge payload (232 B) + two 64-B pose gathers per edge, once per half-edge, + H/b written once
  // (the forward solve is fused into the factor sweep: it re-reads L once there; the solve phase is the backward sweep)
  st.bytes_factor = 288.0 * (double)hblocks + 2.0 * 288.0 * (double)S.nnzL + 2.0 * 48.0 * nb;
  st.bytes_solve = 288.0 * (double)S.nnzL + 2.0 * 48.0 * nb;
  st.bytes_linearize = (double)E * (8 + 56 + 168) + (double)E * 2 * 56 + 288.0 * (double)hblocks + 48.0 * nb;
  if (c->cfg.verbose)
    std::fprintf(stderr, "[fgo] build: N=%lld E=%lld free=%d nnzL=%lld ops=%lld levels=%d tasks=%d symbolic %.3fs (ordering %.3fs) upload %.3fs\n",
                 (long long)N, (long long)E, nb, (long long)S.nnzL, (long long)S.nops, st.n_levels, st.n_tasks, st.t_symbolic, t_ord1 - t_ord0, st.t_upload);
  // host copies of the big lists are no longer needed
  IntList().swap(S.op_a); IntList().swap(S.op_b); IntList().swap(S.g2_a); IntList().swap(S.g2_b);
  return FGO_OK;
}

int upload_hubs(fgo_ctx *c, const HubPlan &hp, size_t entry_cap) {
  hipStream_t s = c->stream;
  HIPCHK(c, c->d_hub_list.alloc(std::max(entry_cap, hp.var.size())));
  HIPCHK(c, c->d_hub_slice.alloc(std::max(entry_cap, hp.var.size())));
  HIPCHK(c, c->d_hub_part.alloc(std::max(entry_cap, hp.var.size()) * HUB_PART));
  HIPCHK(c, c->d_hubm.alloc(3 * std::max(entry_cap, hp.var.size())));
  if (!hp.var.empty()) {
    HIPCHK(c, hipMemcpyAsync(c->d_hub_list.p, hp.var.data(), sizeof(int) * hp.var.size(), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_hub_slice.p, hp.slice.data(), sizeof(int) * hp.slice.size(), hipMemcpyHostToDevice, s));
  }
  if (!hp.multi.empty()) HIPCHK(c, hipMemcpyAsync(c->d_hubm.p, hp.multi.data(), sizeof(int) * hp.multi.size(), hipMemcpyHostToDevice, s));
  HIPCHK(c, hipStreamSynchronize(s));        // the caller's HubPlan may die right after
  return FGO_OK;
}

// Incremental mode: the graph grew since the structure was built.  If the new variables fit the phantom slots and every
// new factor couples variables whose pair already exists in the structure, the factor-side device arrays are extended
// in place: returns FGO_OK (done), 1 (does not fit: the caller rebuilds), or an error.
int refresh_factors(fgo_ctx *c) {
  fgo_ctx::Incr &I = c->inc;
  if (!I.valid || !c->isam_incremental || c->shard_world > 1 || !c->gtsam_mode) return 1;
  const double t0 = now_s();
  const int64_t N = (int64_t)c->ids.size(), E = (int64_t)c->ei.size(), NI = (int64_t)c->imu_payload.size();
  if (N > I.NX || E > I.E_cap || NI > I.NI_cap || N < I.N_done || E < I.E_done || NI < I.NI_done) return 1;
  for (int64_t v = I.N_done; v < N; ++v) if (c->fixed[v]) return 1;
  for (int64_t e = I.E_done; e < E; ++e) if (c->torder[e] == FGO_TANGENT_G2O || (c->torder[e] == 3 && !c->cam_set)) return 1;
  auto find_pair = [&](int va, int vb) -> int {            // variable indices -> pair index, -1 none needed, -2 missing
    const int a = I.hidx[va], b = I.hidx[vb];
    if (a < 0 || b < 0 || a == b) return -1;
    const uint64_t key = ((uint64_t)(uint32_t)std::min(a, b) << 32) | (uint32_t)std::max(a, b);
    auto it = std::lower_bound(I.ukey.begin(), I.ukey.end(), key);
    return (it != I.ukey.end() && *it == key) ? (int)(it - I.ukey.begin()) : -2;
  };
  // ---- check everything first: nothing is modified unless the whole delta fits
  std::vector<int> new_h((size_t)(E - I.E_done));
  for (int64_t e = I.E_done; e < E; ++e) { const int h = find_pair(c->ei[e], c->ej[e]); if (h == -2) return 1; new_h[(size_t)(e - I.E_done)] = h; }
  std::vector<int> new_imu_slot((size_t)15 * (NI - I.NI_done), -1);
  for (int64_t f = I.NI_done; f < NI; ++f) {
    int q = 0;
    for (int u = 0; u < 6; ++u)
      for (int w = u + 1; w < 6; ++w, ++q) {
        const int vu = c->imu_ids[6 * f + u], vw = c->imu_ids[6 * f + w];
        const int h = find_pair(vu, vw);
        if (h == -2) return 1;
        if (h >= 0) new_imu_slot[(size_t)15 * (f - I.NI_done) + q] = (int)((((int64_t)I.nb + h) << 1) | (I.pose_col[vw] > I.pose_col[vu] ? 1 : 0));
      }
  }
  {   // hub entries (one workgroup per slice of a hub variable) after the extension: the scratch buffers have room for hub_cap
    std::unordered_map<int, int64_t> deg;
    for (int64_t e = I.E_done; e < E; ++e)
      for (const int v : {c->ei[e], c->ej[e]}) {
        auto it = deg.find(v);
        if (it == deg.end()) it = deg.emplace(v, I.he_ptr[v + 1] - I.he_ptr[v]).first;
        ++it->second;
      }
    auto entries = [&](int64_t d) -> int64_t { return d > I.hub_deg ? std::min<int64_t>(HUB_MAX_SLICES, (d + HUB_SLICE - 1) / HUB_SLICE) : 0; };
    int64_t n_entries = (int64_t)c->plan.n_hubs;
    for (auto &kv : deg) n_entries += entries(kv.second) - entries(I.he_ptr[kv.first + 1] - I.he_ptr[kv.first]);
    if (n_entries > (int64_t)I.hub_cap) return 1;
  }
  if (c->dev_poses_newer) { const int rc = download_poses(c); if (rc) return rc; }
  destroy_graphs(c);                                           // captured trials hold the factor counts by value
  hipStream_t s = c->stream;
  I.valid = false;                                             // (an error below leaves the lists half-extended: the next call rebuilds)
  // ---- variables: claim phantom slots (kind; the value goes up with upload_poses)
  if (N > I.N_done) {
    HIPCHK(c, hipMemcpyAsync(c->d_var_kind.p + I.N_done, c->var_kind.data() + I.N_done, sizeof(int) * (size_t)(N - I.N_done), hipMemcpyHostToDevice, s));
    c->host_poses_newer = true;
  }
  // ---- binary factors: slots, duplicate groups, payload, incidence lists
  I.edge_h.resize((size_t)E, -1); I.edge_slot.resize((size_t)E, -1);
  int64_t slot_lo = E;                                          // lowest edge whose slot entry changed
  for (int64_t e = I.E_done; e < E; ++e) {
    const int h = new_h[(size_t)(e - I.E_done)];
    I.edge_h[e] = h;
    if (h < 0) continue;
    const int slot = (int)((((int64_t)I.nb + h) << 1) | (I.pose_col[c->ej[e]] > I.pose_col[c->ei[e]] ? 1 : 0));
    if (I.pair_nbin[h] == 0) { I.pair_first[h] = (int)e; I.edge_slot[e] = slot; }
    else {
      if (I.pair_nbin[h] == 1) { const int f0 = I.pair_first[h]; I.dups[h].push_back(f0); I.edge_slot[f0] = -1; slot_lo = std::min<int64_t>(slot_lo, f0); }
      I.dups[h].push_back(e);
    }
    ++I.pair_nbin[h];
  }
  slot_lo = std::min(slot_lo, I.E_done);
  if (E > slot_lo) HIPCHK(c, hipMemcpyAsync(c->d_edge_slot.p + slot_lo, I.edge_slot.data() + slot_lo, sizeof(int) * (size_t)(E - slot_lo), hipMemcpyHostToDevice, s));
  std::vector<int64_t> dup_ptr{0}, dup_edges;
  std::vector<int> dup_slot;
  for (auto &kv : I.dups) {
    for (int64_t e : kv.second) {
      dup_edges.push_back(e);
      dup_slot.push_back((int)((((int64_t)I.nb + kv.first) << 1) | (I.pose_col[c->ej[e]] > I.pose_col[c->ei[e]] ? 1 : 0)));
    }
    dup_ptr.push_back((int64_t)dup_edges.size());
  }
  HIPCHK(c, c->d_dup_ptr.upload(dup_ptr, s));
  HIPCHK(c, c->d_dup_edges.upload(dup_edges, s));
  HIPCHK(c, c->d_dup_slot.upload(dup_slot, s));
  const int64_t dE = E - I.E_done;
  std::vector<double> stage((size_t)28 * dE);
  if (dE > 0) {
    HIPCHK(c, hipMemcpyAsync(c->d_edge_i.p + I.E_done, c->ei.data() + I.E_done, sizeof(int) * (size_t)dE, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_edge_j.p + I.E_done, c->ej.data() + I.E_done, sizeof(int) * (size_t)dE, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_edge_kind.p + I.E_done, c->torder.data() + I.E_done, sizeof(int) * (size_t)dE, hipMemcpyHostToDevice, s));
    for (int64_t e = I.E_done; e < E; ++e) {                    // SoA payload: staged contiguously, scattered by a kernel
      double *o = &stage[(size_t)28 * (e - I.E_done)];
      if (c->torder[e] <= 1) pose_inv7(&c->meas[(size_t)e * 7], o); else std::memcpy(o, &c->meas[(size_t)e * 7], 7 * sizeof(double));
      std::memcpy(o + 7, &c->info[(size_t)e * 21], 21 * sizeof(double));
    }
    HIPCHK(c, c->d_stage.alloc(stage.size()));
    HIPCHK(c, hipMemcpyAsync(c->d_stage.p, stage.data(), sizeof(double) * stage.size(), hipMemcpyHostToDevice, s));
    launch_scatter_edges(c->d_stage.p, dE, I.E_done, c->d_ainv.p, s);
  }
  // incidence lists: a new factor's half-edges go to the END of its variables' lists (edge order inside a list is what keeps
  // the sums deterministic); everything behind the lowest touched variable moves up and is uploaded again -- new factors
  // attach to the newest variables, so that is a short suffix
  {
    int64_t v_lo = I.NX;
    for (int64_t e = I.E_done; e < E; ++e) {
      const int ends[2] = {c->ei[e], c->ej[e]};
      for (int sd = 0; sd < 2; ++sd) {
        const int v = ends[sd];
        I.he.insert(I.he.begin() + I.he_ptr[v + 1], (int)((e << 1) | sd));
        for (int64_t w = v + 1; w <= I.NX; ++w) I.he_ptr[w]++;
        v_lo = std::min<int64_t>(v_lo, v);
      }
    }
    if (v_lo < I.NX) {
      const int64_t p0 = I.he_ptr[v_lo];
      HIPCHK(c, hipMemcpyAsync(c->d_he.p + p0, I.he.data() + p0, sizeof(int) * (size_t)((int64_t)I.he.size() - p0), hipMemcpyHostToDevice, s));
      HIPCHK(c, hipMemcpyAsync(c->d_he_ptr.p + v_lo, I.he_ptr.data() + v_lo, sizeof(int64_t) * (size_t)(I.NX + 1 - v_lo), hipMemcpyHostToDevice, s));
    }
    HubPlan hubs;
    plan_hubs(I.he_ptr, I.NX, I.hub_deg, hubs);
    { const int rc = upload_hubs(c, hubs, I.hub_cap); if (rc) return rc; }
    c->plan.n_hubs = (int)hubs.var.size(); c->plan.n_hub_multi = (int)(hubs.multi.size() / 3);
  }
  // ---- priors (few): rebuilt
  if ((int64_t)c->prior_v.size() != I.NP_done) {
    std::vector<unsigned char> all((size_t)I.NX, 1);
    const int rc = upload_priors(c, I.NX, all);
    if (rc) return rc;
  }
  // ---- IMU factors: payload / ids / slots appended, incidence rebuilt
  const int64_t dI = NI - I.NI_done;
  if (dI > 0) {
    HIPCHK(c, hipMemcpyAsync(c->d_imu.p + I.NI_done, c->imu_payload.data() + I.NI_done, sizeof(ImuPayload) * (size_t)dI, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_imu_ids.p + 6 * I.NI_done, c->imu_ids.data() + 6 * I.NI_done, sizeof(int) * (size_t)(6 * dI), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_imu_slot.p + 15 * I.NI_done, new_imu_slot.data(), sizeof(int) * (size_t)(15 * dI), hipMemcpyHostToDevice, s));
  }
  if (dI > 0) {
    int64_t v_lo = I.NX;
    for (int64_t f = I.NI_done; f < NI; ++f)
      for (int u = 0; u < 6; ++u) {
        const int v = c->imu_ids[6 * f + u];
        I.imu_inc.insert(I.imu_inc.begin() + I.imu_inc_ptr[v + 1], (int)((f << 3) | u));
        for (int64_t w = v + 1; w <= I.NX; ++w) I.imu_inc_ptr[w]++;
        v_lo = std::min<int64_t>(v_lo, v);
      }
    const int64_t p0 = I.imu_inc_ptr[v_lo];
    HIPCHK(c, hipMemcpyAsync(c->d_imu_inc.p + p0, I.imu_inc.data() + p0, sizeof(int) * (size_t)((int64_t)I.imu_inc.size() - p0), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_imu_inc_ptr.p + v_lo, I.imu_inc_ptr.data() + v_lo, sizeof(int64_t) * (size_t)(I.NX + 1 - v_lo), hipMemcpyHostToDevice, s));
  }
  HIPCHK(c, hipStreamSynchronize(s));                           // the staging vectors die here
  // ---- plan
  DevPlan &P = c->plan;
  P.n_edges = E; P.hub_list = c->d_hub_list.p; P.hub_slice = c->d_hub_slice.p; P.hubm = c->d_hubm.p; P.hub_part = c->d_hub_part.p;
  P.he_ptr = c->d_he_ptr.p; P.he = c->d_he.p;
  P.n_dup_groups = (int64_t)dup_ptr.size() - 1; P.dup_ptr = c->d_dup_ptr.p; P.dup_edges = c->d_dup_edges.p; P.dup_slot = c->d_dup_slot.p;
  P.n_priors = c->n_priors_dev; P.prior_ptr = c->d_prior_ptr.p; P.prior_pose = c->d_prior_pose.p;
  P.prior_minv = c->d_prior_minv.p; P.prior_info = c->d_prior_info.p;
  P.n_imu = NI; P.imu_fn = NI; P.imu_inc_ptr = c->d_imu_inc_ptr.p; P.imu_inc = c->d_imu_inc.p;
  for (int k = 0; k < 3; ++k) P.gravity[k] = c->gravity[k];
  P.cam = c->cam;
  I.N_done = N; I.E_done = E; I.NI_done = NI; I.NP_done = (int64_t)c->prior_v.size();
  I.valid = true;
  c->structure_dirty = false;
  c->lin_valid = false;
  c->cov_factor_valid = false;
  fgo_stats &st = c->last;
  st.structure_rebuilt = 0;
  st.t_symbolic = now_s() - t0;                                 // host time of the in-place extension
  st.t_upload = 0;
  st.n_edges = E;
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

// one LM trial on the stream: factor, solve, update into the candidate buffers, linearise there
void enqueue_trial(fgo_ctx *c, int cur, bool with_events) {
  const int cand = cur ^ 1;
  hipStream_t s = c->stream;
  double *scal = c->d_scal.p;
  launch_zero_flag(c->d_fail.p, s);
  if (with_events) (void)hipEventRecord(c->ev[0], s);
  launch_factor(c->plan, c->sched, c->d_H[cur].p, c->d_L.p, scal + 3, c->d_fail.p, s, c->d_b[cur].p, c->d_x.p);   // + forward solve
  if (with_events) (void)hipEventRecord(c->ev[1], s);
  launch_solve(c->plan, c->sched, c->d_L.p, c->d_b[cur].p, c->d_x.p, s, true);                                      // backward sweep
  if (with_events) (void)hipEventRecord(c->ev[2], s);
  if (c->gtsam_mode) launch_update_gtsam(c->plan, c->d_poses[cur].p, c->d_poses[cand].p, c->d_x.p, c->d_b[cur].p, scal + 3, scal + 1, s);
  else launch_update(c->plan, c->d_poses[cur].p, c->d_poses[cand].p, c->d_x.p, c->d_b[cur].p, scal + 3, scal + 1, s);
  if (with_events) (void)hipEventRecord(c->ev[3], s);
  if (c->gtsam_mode) launch_linearize_gtsam(c->plan, c->d_poses[cand].p, c->d_H[cand].p, c->d_b[cand].p, scal + 4, s);
  else launch_linearize(c->plan, c->d_poses[cand].p, c->d_H[cand].p, c->d_b[cand].p, scal + 4, s);
  if (with_events) (void)hipEventRecord(c->ev[4], s);
}

// ---- distributed mode (fgo_set_shard, world > 1).  Scalars every rank needs (chi2, the LM scale, failure flags) are
// partial sums: slot `slot .. slot+n` of d_scal is summed over the ranks.
int dist_sum_scalars(fgo_ctx *c, int slot, int n) {
  if (c->shard_world <= 1) return FGO_OK;
  return dist_allreduce(c, c->d_scal.p + slot, n);
}
// max over the ranks of one scalar in d_scal (sum-only transport: every rank deposits its value in its own slot of a
// zeroed vector, the sum is the vector of all values)
int dist_max_scalar(fgo_ctx *c, int slot) {
  if (c->shard_world <= 1) return FGO_OK;
  hipStream_t s = c->stream;
  const int w = c->shard_world;
  HIPCHK(c, hipMemsetAsync(c->d_gather.p, 0, sizeof(double) * w, s));
  HIPCHK(c, hipMemcpyAsync(c->d_gather.p + c->shard_rank, c->d_scal.p + slot, sizeof(double), hipMemcpyDeviceToDevice, s));
  const int rc = dist_allreduce(c, c->d_gather.p, w);
  if (rc) return rc;
  std::vector<double> h((size_t)w);
  HIPCHK(c, hipMemcpyAsync(h.data(), c->d_gather.p, sizeof(double) * w, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  double m = h[0];
  for (int q = 1; q < w; ++q) m = std::max(m, h[q]);
  HIPCHK(c, hipMemcpyAsync(c->d_scal.p + slot, &m, sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(c, hipStreamSynchronize(s));
  return FGO_OK;
}
// every rank's poses are right for its own domain and the top only: sum the masked copies (end of an optimize call)
int dist_gather_poses(fgo_ctx *c) {
  if (c->shard_world <= 1) return FGO_OK;
  hipStream_t s = c->stream;
  launch_mask_poses(c->plan, c->d_poses[c->cur].p, c->d_gather.p, c->d_pose_group.p, c->shard_rank, c->shard_world, s);
  const int rc = dist_allreduce(c, c->d_gather.p, (int64_t)c->plan.n_poses * 8);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_poses[c->cur].p, c->d_gather.p, sizeof(double) * (size_t)c->plan.n_poses * 8, hipMemcpyDeviceToDevice, s));
  HIPCHK(c, hipStreamSynchronize(s));
  return FGO_OK;
}

// the two halves of a distributed trial (each capturable; the collectives sit between them on the same stream):
//   domain phase: this rank's sub-trees (factor + fused forward solve), then its contributions to the tail of L and x
//   top phase:    the top of the tree (replicated), backward sweep (top, then own domain), update, linearise the candidate
void enqueue_dist_phase(fgo_ctx *c, int cur, int which) {
  const int cand = cur ^ 1;
  hipStream_t s = c->stream;
  double *scal = c->d_scal.p;
  if (which == 0) {
    launch_zero_flag(c->d_fail.p, s);
    launch_factor(c->plan, c->sched, c->d_H[cur].p, c->d_L.p, scal + 3, c->d_fail.p, s, c->d_b[cur].p, c->d_x.p, PHASE_DOMAIN);
  } else {
    launch_factor(c->plan, c->sched, c->d_H[cur].p, c->d_L.p, scal + 3, c->d_fail.p, s, c->d_b[cur].p, c->d_x.p, PHASE_TOP);
    launch_solve(c->plan, c->sched, c->d_L.p, c->d_b[cur].p, c->d_x.p, s, true, PHASE_TOP);
    if (c->gtsam_mode) launch_update_gtsam(c->plan, c->d_poses[cur].p, c->d_poses[cand].p, c->d_x.p, c->d_b[cur].p, scal + 3, scal + 1, s);
    else launch_update(c->plan, c->d_poses[cur].p, c->d_poses[cand].p, c->d_x.p, c->d_b[cur].p, scal + 3, scal + 1, s);
    if (c->gtsam_mode) launch_linearize_gtsam(c->plan, c->d_poses[cand].p, c->d_H[cand].p, c->d_b[cand].p, scal + 4, s);
    else launch_linearize(c->plan, c->d_poses[cand].p, c->d_H[cand].p, c->d_b[cand].p, scal + 4, s);
  }
}
int launch_dist_phase(fgo_ctx *c, int which) {
  hipStream_t s = c->stream;
  if (!c->use_graph) { enqueue_dist_phase(c, c->cur, which); return FGO_OK; }
  hipGraphExec_t &ge = c->dist_graph[c->cur][which];
  if (!ge) {
    static std::mutex capture_mutex;
    std::lock_guard<std::mutex> capture_lock(capture_mutex);
    hipGraph_t graph = nullptr;
    HIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    enqueue_dist_phase(c, c->cur, which);
    HIPCHK(c, hipStreamEndCapture(s, &graph));
    HIPCHK(c, hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
  }
  HIPCHK(c, hipGraphLaunch(ge, s));
  return FGO_OK;
}
int run_trial_dist(fgo_ctx *c, double lambda, double *chi_cand, double *scale, int *failed, fgo_stats *st) {
  hipStream_t s = c->stream;
  c->h_scal[3] = lambda;
  HIPCHK(c, hipMemcpyAsync(c->d_scal.p + 3, c->h_scal + 3, sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(c, hipEventRecord(c->ev[0], s));
  int rc = launch_dist_phase(c, 0);
  if (rc) return rc;
  static const bool dbg_fail = std::getenv("FGO_DEBUG_TRIALS") != nullptr;
  if (dbg_fail) { int hf0 = -1; (void)hipMemcpyAsync(&hf0, c->d_fail.p, sizeof(int), hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); std::fprintf(stderr, "[fgo trial] rank %d fail flag after the domain phase: %d\n", c->shard_rank, hf0); }
  // collective 1: the domains' updates into the top of the factor and of the right-hand side (both are contiguous tails)
  rc = dist_allreduce(c, c->d_L.p + 36 * (size_t)c->plan.top_blk0, 36 * c->sched.n_top_blocks);
  if (rc) return rc;
  rc = dist_allreduce(c, c->d_x.p + 6 * (size_t)c->plan.top_col0, 6 * (int64_t)c->sched.n_top_cols);
  if (rc) return rc;
  rc = launch_dist_phase(c, 1);
  if (rc) return rc;
  // the gradient of the top is a partial sum like H_top; it is completed once per linearisation (the LM scale and
  // k_dist_rhs on rank 0 read the complete one)
  rc = dist_allreduce(c, c->d_b[c->cur ^ 1].p + 6 * (size_t)c->plan.top_col0, 6 * (int64_t)c->sched.n_top_cols);
  if (rc) return rc;
  HIPCHK(c, hipEventRecord(c->ev[4], s));
  // collective 2: scalars, one all-reduce of three: [4] chi2 of the candidate (a partial sum over this rank's factors),
  // [5] the failure flag, [6] the LM scale (k_update sums the columns this rank is responsible for)
  launch_pack_scalars(c->d_scal.p, c->d_fail.p, s);
  rc = dist_sum_scalars(c, 4, 3);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->h_scal + 4, c->d_scal.p + 4, sizeof(double) * 3, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  c->h_scal[1] = c->h_scal[6];
  *chi_cand = c->h_scal[4]; *scale = c->h_scal[1]; *failed = c->h_scal[5] != 0.0;
  if (st) { float ms = 0; (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[4]); st->reserved[0] += ms; }
  return FGO_OK;
}

int run_trial(fgo_ctx *c, double lambda, double *chi_cand, double *scale, int *failed, fgo_stats *st) {
  c->cov_factor_valid = false;
  if (c->shard_world > 1) return run_trial_dist(c, lambda, chi_cand, scale, failed, st);
  hipStream_t s = c->stream;
  c->h_scal[3] = lambda;
  HIPCHK(c, hipMemcpyAsync(c->d_scal.p + 3, c->h_scal + 3, sizeof(double), hipMemcpyHostToDevice, s));
  if (c->use_graph) {
    hipGraphExec_t &ge = c->trial_graph[c->cur];
    if (!ge) {
      // one capture at a time per process: two contexts capturing from two host threads at once (thread-local mode)
      // occasionally produced a graph that computes garbage (tools/stress_shard_threads.py: 12 of 60 runs diverged,
      // none without graphs or with this lock)
      static std::mutex capture_mutex;
      std::lock_guard<std::mutex> capture_lock(capture_mutex);
      hipGraph_t graph = nullptr;
      HIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      enqueue_trial(c, c->cur, false);
      HIPCHK(c, hipStreamEndCapture(s, &graph));
      HIPCHK(c, hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
    }
    HIPCHK(c, hipEventRecord(c->ev[0], s));
    HIPCHK(c, hipGraphLaunch(ge, s));
    HIPCHK(c, hipEventRecord(c->ev[4], s));
  } else {
    enqueue_trial(c, c->cur, true);
  }
  HIPCHK(c, hipMemcpyAsync(c->h_scal + 1, c->d_scal.p + 1, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(c->h_scal + 4, c->d_scal.p + 4, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(c->h_fail, c->d_fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  *chi_cand = c->h_scal[4]; *scale = c->h_scal[1]; *failed = *c->h_fail;
  if (st) {
    float ms = 0;
    if (c->use_graph) {
      (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[4]);
      st->reserved[0] += ms;          // whole-trial device ms (graph mode)
    } else {
      (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]); st->ms_factor += ms;
      (void)hipEventElapsedTime(&ms, c->ev[1], c->ev[2]); st->ms_solve += ms;
      (void)hipEventElapsedTime(&ms, c->ev[2], c->ev[3]); st->ms_update += ms;
      (void)hipEventElapsedTime(&ms, c->ev[3], c->ev[4]); st->ms_linearize += ms;
      (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[4]); st->reserved[0] += ms;
    }
  }
  return FGO_OK;
}

int linearize_current(fgo_ctx *c, bool want_maxdiag) {
  hipStream_t s = c->stream;
  c->cov_factor_valid = false;
  if (c->gtsam_mode) launch_linearize_gtsam(c->plan, c->d_poses[c->cur].p, c->d_H[c->cur].p, c->d_b[c->cur].p, c->d_scal.p + 0, s);
  else launch_linearize(c->plan, c->d_poses[c->cur].p, c->d_H[c->cur].p, c->d_b[c->cur].p, c->d_scal.p + 0, s);
  { const int rc = dist_sum_scalars(c, 0, 1); if (rc) return rc; }                    // chi2: partial sums over the ranks' factors
  if (c->shard_world > 1) {                                                           // complete the gradient of the top
    const int rc = dist_allreduce(c, c->d_b[c->cur].p + 6 * (size_t)c->plan.top_col0, 6 * (int64_t)c->sched.n_top_cols);
    if (rc) return rc;
  }
  if (want_maxdiag) {
    if (c->shard_world > 1) {
      // the diagonal blocks of the top are partial sums: complete them (in place, diagonal blocks of the top columns
      // are contiguous in H), then the maximum over own-domain + top diagonals, then the maximum over the ranks.
      // Afterwards only rank 0 keeps the summed top diagonal, so that the partial sums still add up to H.
      double *Htop = c->d_H[c->cur].p + 36 * (size_t)c->plan.top_col0;
      const size_t ntop = 36 * (size_t)c->sched.n_top_cols;
      const int rc = dist_allreduce(c, Htop, (int64_t)ntop);
      if (rc) return rc;
      launch_maxdiag(c->plan, c->d_H[c->cur].p, c->d_scal.p + 2, s);
      if (c->shard_rank != 0) HIPCHK(c, hipMemsetAsync(Htop, 0, sizeof(double) * ntop, s));
      const int rc2 = dist_max_scalar(c, 2);
      if (rc2) return rc2;
    } else {
      launch_maxdiag(c->plan, c->d_H[c->cur].p, c->d_scal.p + 2, s);
    }
  }
  HIPCHK(c, hipMemcpyAsync(c->h_scal, c->d_scal.p, sizeof(double) * 3, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  c->chi_cur = c->h_scal[0];
  c->lin_valid = true;
  return FGO_OK;
}

}  // namespace


// Nothing may unwind through the extern "C" boundary into a C / ctypes / ROS caller (include/fgo.h: every entry point
// returns a code): the entry points below are function-try-blocks.
#define FGO_CATCH_INT(c)                                                                              \
  catch (const std::bad_alloc &) { return fail(c, FGO_ENOMEM, "out of host memory"); }               \
  catch (const std::exception &e) { return fail(c, FGO_EINVAL, std::string("exception: ") + e.what()); } \
  catch (...) { return fail(c, FGO_EINVAL, "unknown exception"); }
#define FGO_CATCH_NAN(c)                                                                              \
  catch (const std::exception &e) { if (c) c->err = std::string("exception: ") + e.what(); return std::numeric_limits<double>::quiet_NaN(); } \
  catch (...) { if (c) c->err = "unknown exception"; return std::numeric_limits<double>::quiet_NaN(); }

// =================================================================================================
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
  if (hipSetDevice(c->cfg.device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    g_create_error = "hipSetDevice / hipStreamCreate failed"; delete c; return nullptr;
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

double fgo_chi2(fgo_ctx *c) try {
  if (!c) return std::numeric_limits<double>::quiet_NaN();
  (void)hipSetDevice(c->cfg.device);
  if (c->ei.empty() && c->prior_v.empty() && c->imu_payload.empty()) return 0.0;
  // a graph with edges but no free vertex still has a chi2; build() refuses it, so evaluate on a minimal plan
  if (ensure_ready(c) != FGO_OK) return std::numeric_limits<double>::quiet_NaN();
  if (c->lin_valid) return c->chi_cur;
  if (c->shard_world > 1) {          // distributed: a rank evaluates its own factors; the linearisation pass sums them (collective!)
    if (linearize_current(c, false) != FGO_OK) return std::numeric_limits<double>::quiet_NaN();
    return c->chi_cur;
  }
  if (c->gtsam_mode) launch_chi2_gtsam(c->plan, c->d_poses[c->cur].p, c->d_scal.p + 0, c->stream);
  else launch_chi2(c->plan, c->d_poses[c->cur].p, c->d_scal.p + 0, c->stream);
  if (hipMemcpyAsync(c->h_scal, c->d_scal.p, sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess) {
    c->err = "chi2 kernel failed";
    return std::numeric_limits<double>::quiet_NaN();
  }
  return c->h_scal[0];
} FGO_CATCH_NAN(c)

int fgo_optimize(fgo_ctx *c, int max_iters, fgo_stats *stats) try {
  if (!c || max_iters < 0) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  const double tstart = now_s();
  const bool was_dirty = c->structure_dirty;
  int rc = ensure_ready(c);
  if (rc) return rc;
  if (c->gtsam_mode) return fail(c, FGO_EINVAL, "GTSAM-semantics graph: use fgo_optimize_gtsam");
  fgo_stats st = c->last;
  if (!was_dirty) { st.structure_rebuilt = 0; st.t_symbolic = 0; st.t_upload = 0; }     // (else: as build() / refresh_factors() left it)
  st.iterations = st.trials = st.terminated = 0;
  st.ms_factor = st.ms_solve = st.ms_update = st.ms_linearize = 0; st.reserved[0] = 0;
  c->tr_chi2.clear(); c->tr_lambda.clear();
  c->xgmi_bytes = 0;
  double lambda = 0, ni = 2;
  int it = 0;
  bool ok = true;
  for (; it < max_iters && ok; ++it) {
    if (!c->lin_valid || it == 0) {
      rc = linearize_current(c, it == 0);
      if (rc) return rc;
    }
    double cur = c->chi_cur;
    if (it == 0) { st.chi2_initial = cur; lambda = 1e-5 * c->h_scal[2]; ni = 2; }
    double rho = 0;
    int q = 0;
    do {
      double tmp = 0, scale = 0;
      int failed = 0;
      rc = run_trial(c, lambda, &tmp, &scale, &failed, &st);
      if (rc) return rc;
      ++st.trials;
      static const bool dbg_trials = std::getenv("FGO_DEBUG_TRIALS") != nullptr;
      if (dbg_trials) std::fprintf(stderr, "[fgo trial] rank %d it %d q %d lambda %.6e chi_cur %.9e chi_cand %.9e scale %.6e failed %d\n", c->shard_rank, it, q, lambda, cur, tmp, scale, failed);
      if (failed || !std::isfinite(tmp)) tmp = std::numeric_limits<double>::max();
      rho = (cur - tmp) / (scale + 1e-3);
      // a non-positive pivot leaves NaNs in x and hence in `scale`: g2o's solver keeps x finite on failure, so its rho is a
      // large negative number and the trial loop retries with a larger lambda (up to 10 times, then 'Terminate')
      if (failed || !std::isfinite(rho)) rho = -1.0;
      if (rho > 0 && std::isfinite(tmp)) {
        double alpha = 1. - std::pow(2 * rho - 1, 3);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        cur = tmp;
        c->cur ^= 1;                 // discardTop: the candidate buffers become current
        c->chi_cur = cur;
        c->lin_valid = true;
        c->dev_poses_newer = true;
      } else {
        lambda *= ni; ni *= 2;       // pop: keep the current buffers
        if (!std::isfinite(lambda)) break;
      }
      ++q;
    } while (rho < 0 && q < 10);
    c->tr_chi2.push_back(cur); c->tr_lambda.push_back(lambda);
    st.chi2_final = cur;
    if (q == 10 || rho == 0 || !std::isfinite(lambda)) { ok = false; st.terminated = 1; }
  }
  if (it > 0) { rc = dist_gather_poses(c); if (rc) return rc; }
  st.iterations = it; st.lambda_final = lambda;
  st.reserved[2] = c->xgmi_bytes;        // bytes this rank handed to the collectives during this call
  st.t_total = now_s() - tstart;
  c->last = st;
  if (stats) *stats = st;
  return it;
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

double fgo_error(fgo_ctx *c) { return 0.5 * fgo_chi2(c); }

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
  if (!(pre->dt > 0)) return fail(c, FGO_EINVAL, "empty preintegration");
  // information = preintMeasCov^-1 through a Cholesky factorisation (the covariance must be SPD)
  double L[225], inv[225];
  std::memset(L, 0, sizeof(L));
  for (int j = 0; j < 15; ++j) {
    double d = pre->cov[j * 15 + j];
    for (int k = 0; k < j; ++k) d -= L[j * 15 + k] * L[j * 15 + k];
    if (!(d > 0)) return fail(c, FGO_ENUM, "preintegrated covariance is not positive definite");
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
  ImuPayload P;
  std::memset(&P, 0, sizeof(P));
  P.dt = pre->dt;
  std::memcpy(P.dR, pre->dR, sizeof(P.dR)); std::memcpy(P.dp, pre->dp, sizeof(P.dp)); std::memcpy(P.dv, pre->dv, sizeof(P.dv));
  std::memcpy(P.J_R_bg, pre->J_R_bg, sizeof(P.J_R_bg)); std::memcpy(P.J_p_ba, pre->J_p_ba, sizeof(P.J_p_ba));
  std::memcpy(P.J_p_bg, pre->J_p_bg, sizeof(P.J_p_bg)); std::memcpy(P.J_v_ba, pre->J_v_ba, sizeof(P.J_v_ba));
  std::memcpy(P.J_v_bg, pre->J_v_bg, sizeof(P.J_v_bg)); std::memcpy(P.bhat, pre->bhat, sizeof(P.bhat));
  for (int r = 0; r < 15; ++r) for (int q = 0; q < 15; ++q) P.info[r * 15 + q] = 0.5 * (inv[r * 15 + q] + inv[q * 15 + r]);
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

// GTSAM 4.0 LevenbergMarquardtOptimizer::optimize() with default LevenbergMarquardtParams:
// lambda0 1e-5, fixed factor 10, lambdaUpper 1e5, identity damping, minModelFidelity 1e-3, relative / absolute
// error tolerance 1e-5, at most 100 iterations.  One iteration = linearise once, then search lambda.
// The linearised cost change b'd - d'Hd/2 is obtained from the damped solve itself:
// (H + lambda I) d = b  =>  d'Hd = b'd - lambda |d|^2, so it equals (b'd + lambda |d|^2) / 2 = scale / 2.
int fgo_optimize_gtsam(fgo_ctx *c, int max_iters, fgo_stats *stats) try {
  if (!c) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  const double tstart = now_s();
  const bool was_dirty = c->structure_dirty;
  int rc = ensure_ready(c);
  if (rc) return rc;
  if (!c->gtsam_mode) return fail(c, FGO_EINVAL, "g2o-semantics graph: use fgo_optimize");
  if (max_iters <= 0) max_iters = 100;
  fgo_stats st = c->last;
  if (!was_dirty) { st.structure_rebuilt = 0; st.t_symbolic = 0; st.t_upload = 0; }     // (else: as build() / refresh_factors() left it)
  st.iterations = st.trials = st.terminated = 0;
  st.ms_factor = st.ms_solve = st.ms_update = st.ms_linearize = 0; st.reserved[0] = 0;
  c->tr_chi2.clear(); c->tr_lambda.clear();
  c->xgmi_bytes = 0;
  const double lambdaFactor = 10.0, lambdaUpper = 1e5, lambdaLower = 0.0, minModelFidelity = 1e-3;
  const double relTol = 1e-5, absTol = 1e-5, errTol = 0.0;
  double lambda = 1e-5;
  if (!c->lin_valid) { rc = linearize_current(c, false); if (rc) return rc; }
  double currentError = 0.5 * c->chi_cur;
  st.chi2_initial = c->chi_cur;
  int iterations = 0;
  while (true) {
    const double errorBefore = currentError;
    if (!c->lin_valid) { rc = linearize_current(c, false); if (rc) return rc; }
    while (true) {
      double chi_cand = 0, scale = 0;
      int failed = 0;
      rc = run_trial(c, lambda, &chi_cand, &scale, &failed, &st);
      if (rc) return rc;
      ++st.trials;
      bool step_ok = false, stop_search = false;
      double newError = currentError;
      if (!failed && std::isfinite(chi_cand)) {
        newError = 0.5 * chi_cand;
        const double linearizedCostChange = 0.5 * scale;
        if (linearizedCostChange >= 0) {
          const double costChange = currentError - newError;
          if (linearizedCostChange > 1e-20 && costChange / linearizedCostChange > minModelFidelity) step_ok = true;
          if (std::fabs(costChange) < relTol * currentError) stop_search = true;
        }
      }
      if (step_ok) {
        currentError = newError;
        c->cur ^= 1;
        c->chi_cur = 2 * newError;
        c->lin_valid = true;
        c->dev_poses_newer = true;
        lambda = std::max(lambdaLower, lambda / lambdaFactor);
        break;
      }
      if (stop_search) break;
      lambda *= lambdaFactor;
      if (lambda >= lambdaUpper) break;
    }
    ++iterations;
    c->tr_chi2.push_back(2 * currentError); c->tr_lambda.push_back(lambda);
    if (iterations >= max_iters || !std::isfinite(currentError) || currentError <= errTol) break;
    const double absDec = errorBefore - currentError, relDec = absDec / errorBefore;
    if (relDec <= relTol || absDec <= absTol) break;
  }
  rc = dist_gather_poses(c);
  if (rc) return rc;
  st.iterations = iterations; st.chi2_final = 2 * currentError; st.lambda_final = lambda;
  st.reserved[2] = c->xgmi_bytes;
  st.t_total = now_s() - tstart;
  c->last = st;
  if (stats) *stats = st;
  return iterations;
} FGO_CATCH_INT(c)

// ISAM2::update + calculateEstimate on the batch machinery (kernels_gtsam.hip: k_isam2_relin / k_isam2_estimate)
int fgo_isam2_update(fgo_ctx *c, double relin_threshold, fgo_stats *stats) try {
  if (!c || !(relin_threshold >= 0)) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  const double tstart = now_s();
  if (c->shard_world > 1) return fail(c, FGO_ESTATE, "fgo_isam2_update is not available in distributed mode");
  // a context that is updated incrementally builds its structure with room to grow (phantom variable slots + factor
  // capacity), so that the per-record updates of the reference's drivers do not pay the structure phase every time
  static const bool incr_off = std::getenv("FGO_ISAM_INCREMENTAL") && std::atoi(std::getenv("FGO_ISAM_INCREMENTAL")) == 0;
  if (!incr_off) c->isam_incremental = true;
  const bool was_dirty = c->structure_dirty;
  int rc = ensure_ready(c);
  if (rc) return rc;
  if (!c->gtsam_mode) return fail(c, FGO_EINVAL, "g2o-semantics graph: ISAM2 semantics need a GTSAM-semantics graph");
  hipStream_t s = c->stream;
  const int64_t NX = c->plan.n_poses, N = (int64_t)c->ids.size();   // NX: incl. the phantom slots of the incremental mode
  if (c->d_theta.n != (size_t)NX * 8) {                 // (re)size the state to the structure; covered variables keep theta / delta
    DevBuf<double> th, de;
    HIPCHK(c, th.alloc((size_t)NX * 8));
    HIPCHK(c, de.alloc((size_t)NX * 6));
    HIPCHK(c, hipMemsetAsync(th.p, 0, sizeof(double) * (size_t)NX * 8, s));
    HIPCHK(c, hipMemsetAsync(de.p, 0, sizeof(double) * (size_t)NX * 6, s));
    if (c->isam_n > 0) {
      HIPCHK(c, hipMemcpyAsync(th.p, c->d_theta.p, sizeof(double) * (size_t)c->isam_n * 8, hipMemcpyDeviceToDevice, s));
      HIPCHK(c, hipMemcpyAsync(de.p, c->d_delta.p, sizeof(double) * (size_t)c->isam_n * 6, hipMemcpyDeviceToDevice, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    c->d_theta.swap(th);
    c->d_delta.swap(de);
  }
  if (c->isam_n < N) {                                  // newTheta: new variables enter at their initial value, delta = 0
    HIPCHK(c, hipMemcpyAsync(c->d_theta.p + (size_t)c->isam_n * 8, c->d_poses[c->cur].p + (size_t)c->isam_n * 8,
                             sizeof(double) * (size_t)(N - c->isam_n) * 8, hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemsetAsync(c->d_delta.p + (size_t)c->isam_n * 6, 0, sizeof(double) * (size_t)(N - c->isam_n) * 6, s));
    c->isam_n = N;
  }
  fgo_stats st = c->last;
  if (!was_dirty) { st.structure_rebuilt = 0; st.t_symbolic = 0; st.t_upload = 0; }     // (else: set by build() / refresh_factors())
  st.iterations = st.trials = 1; st.terminated = 0;
  st.ms_factor = st.ms_solve = st.ms_update = st.ms_linearize = 0; st.reserved[0] = 0;
  double *scal = c->d_scal.p;
  const int w = c->cur ^ 1;                             // H / b of the side buffers: the current ones stay valid for the values
  c->h_scal[3] = 0.0;                                   // Gauss-Newton: no damping (ISAM2GaussNewtonParams)
  HIPCHK(c, hipMemcpyAsync(scal + 3, c->h_scal + 3, sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemsetAsync(c->d_fail.p, 0, sizeof(int), s));
  HIPCHK(c, hipEventRecord(c->ev[0], s));
  launch_isam2_relin(c->plan, c->d_theta.p, c->d_delta.p, relin_threshold, scal + 5, s);
  launch_linearize_gtsam(c->plan, c->d_theta.p, c->d_H[w].p, c->d_b[w].p, scal + 4, s);
  HIPCHK(c, hipEventRecord(c->ev[1], s));
  c->cov_factor_valid = false;
  launch_factor(c->plan, c->sched, c->d_H[w].p, c->d_L.p, scal + 3, c->d_fail.p, s, c->d_b[w].p, c->d_x.p);
  HIPCHK(c, hipEventRecord(c->ev[2], s));
  launch_solve(c->plan, c->sched, c->d_L.p, c->d_b[w].p, c->d_x.p, s, true);
  HIPCHK(c, hipEventRecord(c->ev[3], s));
  HIPCHK(c, hipMemcpyAsync(c->h_fail, c->d_fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(c->h_scal + 4, scal + 4, sizeof(double) * 2, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  st.chi2_initial = c->h_scal[4];                       // chi2 at the linearisation point
  st.reserved[1] = c->h_scal[5];                        // variables relinearised by this update
  if (*c->h_fail) {
    c->last = st;
    return fail(c, FGO_ENUM, "ISAM2 update: linear system not positive definite (IndeterminantLinearSystemException)");
  }
  launch_isam2_estimate(c->plan, c->d_theta.p, c->d_x.p, c->d_delta.p, c->d_poses[c->cur].p, s);
  launch_chi2_gtsam(c->plan, c->d_poses[c->cur].p, scal + 0, s);
  HIPCHK(c, hipEventRecord(c->ev[4], s));
  HIPCHK(c, hipMemcpyAsync(c->h_scal, scal, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]); st.ms_linearize = ms;
  (void)hipEventElapsedTime(&ms, c->ev[1], c->ev[2]); st.ms_factor = ms;
  (void)hipEventElapsedTime(&ms, c->ev[2], c->ev[3]); st.ms_solve = ms;
  (void)hipEventElapsedTime(&ms, c->ev[3], c->ev[4]); st.ms_update = ms;
  (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[4]); st.reserved[0] = ms;
  c->chi_cur = c->h_scal[0];
  c->lin_valid = false;                                 // H / b of the current buffers no longer match the values
  c->dev_poses_newer = true;
  st.chi2_final = c->h_scal[0]; st.lambda_final = 0;
  st.t_total = now_s() - tstart;
  c->last = st;
  if (stats) *stats = st;
  return 1;
} FGO_CATCH_INT(c)

int fgo_isam2_reserve(fgo_ctx *c, int reserve_variables, int window) try {
  if (!c || reserve_variables < 0 || window < 0) return FGO_EINVAL;
  c->isam_reserve = reserve_variables;
  if (window > 0) c->isam_window = window;
  if (c->inc.valid) { c->inc.valid = false; c->structure_dirty = true; }      // the next use rebuilds with the new reserve
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_isam2_reset(fgo_ctx *c) try {
  if (!c) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->d_theta.release(); c->d_delta.release();
  c->isam_n = 0;
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_isam2_get_state(fgo_ctx *c, int64_t id, double theta7[7], double delta6[6]) try {
  if (!c || (!theta7 && !delta6)) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  auto it = c->id2idx.find(id);
  if (it == c->id2idx.end()) return fail(c, FGO_EINVAL, "unknown variable id");
  if (it->second >= c->isam_n) return fail(c, FGO_ESTATE, "variable not yet seen by fgo_isam2_update");
  if (theta7) HIPCHK(c, hipMemcpy(theta7, c->d_theta.p + (size_t)it->second * 8, 7 * sizeof(double), hipMemcpyDeviceToHost));
  if (delta6) HIPCHK(c, hipMemcpy(delta6, c->d_delta.p + (size_t)it->second * 6, 6 * sizeof(double), hipMemcpyDeviceToHost));
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_dist_unique_id(void *id128) {
  if (!id128) return FGO_EINVAL;
  RcclApi *api = rccl_api();
  if (!api) return FGO_ENODEV;
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (api->GetUniqueId(&id) != ncclSuccess) return FGO_ENODEV;
  std::memcpy(id128, &id, sizeof(id));
  return FGO_OK;
}

int fgo_dist_init_rccl(fgo_ctx *c, const void *id128) try {
  if (!c || !id128) return FGO_EINVAL;
  RcclApi *api = rccl_api();
  if (!api) return fail(c, FGO_ENODEV, "librccl not found (set FGO_RCCL_LIB)");
  (void)hipSetDevice(c->cfg.device);
  if (c->rccl) { (void)api->CommDestroy(c->rccl); c->rccl = nullptr; }
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  const ncclResult_t r = api->CommInitRank(&c->rccl, c->shard_world, id, c->shard_rank);
  if (r != ncclSuccess) { c->rccl = nullptr; return fail(c, FGO_ENODEV, std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(r) : "failed")); }
  return FGO_OK;
} FGO_CATCH_INT(c)

// host-only: the domain decomposition fgo_set_shard(., world) would use for the block graph with `n` vertices and the
// undirected edges (a[k], b[k]): group_out[v] = owning rank, `world` = top (tests; needs no device)
int fgo_debug_partition(int n, int64_t n_pairs, const int *a, const int *b, int world, int *group_out) {
  if (n <= 0 || n_pairs < 0 || !a || !b || world < 1 || !group_out) return FGO_EINVAL;
  try {
    std::vector<std::pair<int, int>> pr;
    for (int64_t k = 0; k < n_pairs; ++k) {
      if (a[k] < 0 || b[k] < 0 || a[k] >= n || b[k] >= n) return FGO_EINVAL;
      if (a[k] != b[k]) pr.push_back({std::min(a[k], b[k]), std::max(a[k], b[k])});
    }
    std::sort(pr.begin(), pr.end());
    pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
    BlockGraph g;
    g.n = n;
    g.xadj.assign((size_t)n + 1, 0);
    for (auto &e : pr) { g.xadj[e.first + 1]++; g.xadj[e.second + 1]++; }
    for (int i = 0; i < n; ++i) g.xadj[i + 1] += g.xadj[i];
    g.adj.resize((size_t)g.xadj[n]);
    std::vector<int> fill(g.xadj.begin(), g.xadj.end() - 1);
    for (auto &e : pr) { g.adj[fill[e.first]++] = e.second; g.adj[fill[e.second]++] = e.first; }
    std::vector<int> perm;
    OrderingOptions oo;
    nested_dissection(g, oo, perm);
    Symbolic S;
    build_symbolic(g, perm, 5000, (int64_t)1 << 60, S, world);
    for (int v = 0; v < n; ++v) {
      const int col = S.iperm[v];
      group_out[v] = world == 1 ? 0 : (int)(std::upper_bound(S.dom_col0.begin(), S.dom_col0.begin() + world + 1, col) - S.dom_col0.begin()) - 1;
    }
    return FGO_OK;
  } catch (...) { return FGO_ENOMEM; }
}

// tests: sum `n` host doubles over the ranks through the context's transport, even when world == 1 (exercises the RCCL
// binding on a single-GPU box)
int fgo_debug_allreduce(fgo_ctx *c, double *host_buf, int64_t n) try {
  if (!c || !host_buf || n <= 0) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  DevBuf<double> d;
  HIPCHK(c, d.alloc((size_t)n));
  HIPCHK(c, hipMemcpyAsync(d.p, host_buf, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  if (c->rccl) {
    const ncclResult_t r = rccl_api()->AllReduce(d.p, d.p, (size_t)n, ncclDouble, ncclSum, c->rccl, c->stream);
    if (r != ncclSuccess) return fail(c, FGO_ENODEV, "ncclAllReduce failed");
  } else if (c->ar_fn) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->ar_fn(c->ar_user, d.p, n) != 0) return fail(c, FGO_ENODEV, "all-reduce hook failed");
  } else {
    return fail(c, FGO_ESTATE, "no transport");
  }
  HIPCHK(c, hipMemcpyAsync(host_buf, d.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_set_shard(fgo_ctx *c, int rank, int world) try {
  if (!c || world < 1 || rank < 0 || rank >= world) return FGO_EINVAL;
  if (rank != c->shard_rank || world != c->shard_world) { c->structure_dirty = true; c->inc.valid = false; }
  c->shard_rank = rank; c->shard_world = world;
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_set_allreduce(fgo_ctx *c, fgo_allreduce_fn fn, void *user) try {
  if (!c) return FGO_EINVAL;
  c->ar_fn = fn; c->ar_user = user;
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_debug_read_system(fgo_ctx *c, double *H, double *b, double *chi2) try {
  if (!c) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  int rc = ensure_ready(c);
  if (rc) return rc;
  // linearise WITHOUT the all-reduce so a shard's partial sums can be inspected
  hipStream_t s = c->stream;
  if (c->gtsam_mode) launch_linearize_gtsam(c->plan, c->d_poses[c->cur].p, c->d_H[c->cur].p, c->d_b[c->cur].p, c->d_scal.p + 0, s);
  else launch_linearize(c->plan, c->d_poses[c->cur].p, c->d_H[c->cur].p, c->d_b[c->cur].p, c->d_scal.p + 0, s);
  c->lin_valid = false;
  if (H) HIPCHK(c, hipMemcpyAsync(H, c->d_H[c->cur].p, sizeof(double) * 36 * (size_t)c->plan.n_hblocks, hipMemcpyDeviceToHost, s));
  if (b) HIPCHK(c, hipMemcpyAsync(b, c->d_b[c->cur].p, sizeof(double) * 6 * (size_t)c->plan.nb, hipMemcpyDeviceToHost, s));
  if (chi2) HIPCHK(c, hipMemcpyAsync(chi2, c->d_scal.p, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_trace(const fgo_ctx *c, double *chi2s, double *lambdas, int cap) {
  if (!c || cap < 0) return FGO_EINVAL;
  const int m = std::min<int>(cap, (int)c->tr_chi2.size());
  if (chi2s) std::memcpy(chi2s, c->tr_chi2.data(), sizeof(double) * m);
  if (lambdas) std::memcpy(lambdas, c->tr_lambda.data(), sizeof(double) * m);
  return m;
}

int fgo_get_stats(const fgo_ctx *c, fgo_stats *st) {
  if (!c || !st) return FGO_EINVAL;
  *st = c->last;
  return FGO_OK;
}

int fgo_linearize(fgo_ctx *c, double *chi2_out, double *H_dense, double *b_dense, int64_t *n_free_out) try {
  if (!c) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  int rc = ensure_ready(c);
  if (rc) return rc;
  rc = linearize_current(c, false);
  if (rc) return rc;
  if (chi2_out) *chi2_out = c->chi_cur;
  const int nb = c->plan.nb;
  if (n_free_out) *n_free_out = nb;
  if (!H_dense && !b_dense) return FGO_OK;
  if (nb > 4096) return fail(c, FGO_EINVAL, "dense read-back is limited to 4096 free poses");
  const size_t hblocks = (size_t)nb + (size_t)c->n_offdiag;
  std::vector<double> H(hblocks * 36), b((size_t)nb * 6);
  std::vector<int> asrc((size_t)c->S.nnzL);
  HIPCHK(c, hipMemcpy(H.data(), c->d_H[c->cur].p, sizeof(double) * H.size(), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(b.data(), c->d_b[c->cur].p, sizeof(double) * b.size(), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(asrc.data(), c->d_asrc.p, sizeof(int) * asrc.size(), hipMemcpyDeviceToHost));
  const size_t m = (size_t)nb * 6;
  if (H_dense) {
    std::memset(H_dense, 0, sizeof(double) * m * m);
    for (int k = 0; k < nb; ++k)
      for (int64_t t = c->S.colptr[k]; t < c->S.colptr[k + 1]; ++t) {
        if (asrc[t] < 0) continue;
        const int hr = c->S.perm[c->S.rowidx[t]], hc = c->S.perm[k];   // hessian (ascending-id) indices
        const double *B = &H[(size_t)asrc[t] * 36];
        for (int r = 0; r < 6; ++r)
          for (int q = 0; q < 6; ++q) {
            H_dense[((size_t)hr * 6 + r) * m + (size_t)hc * 6 + q] = B[r * 6 + q];
            H_dense[((size_t)hc * 6 + q) * m + (size_t)hr * 6 + r] = B[r * 6 + q];
          }
      }
  }
  if (b_dense)
    for (int k = 0; k < nb; ++k) std::memcpy(b_dense + (size_t)c->S.perm[k] * 6, &b[(size_t)k * 6], 6 * sizeof(double));
  return FGO_OK;
} FGO_CATCH_INT(c)

// Marginals(graph, values, CHOLESKY).marginalCovariance(key): the (id, id) block of (J' Omega J)^-1 at the current
// linearisation (gtsam/gtsam_graph.cpp:598-601).  The reference pays a full batch factorisation per call (and builds
// one it never uses at :1357); here the factor stays resident in HBM: one undamped factorisation per linearisation
// point, then 6 pairs of triangular solves per requested block.
// one undamped factorisation of the current linearisation, kept resident (c->cov_factor_valid) until the estimate or
// the structure changes; then the requested diagonal blocks of H^-1: 6 pairs of triangular solves per block
static int marginal_blocks(fgo_ctx *c, int64_t n, const int64_t *ids, double *cov36) {
  (void)hipSetDevice(c->cfg.device);
  if (c->shard_world > 1) return fail(c, FGO_ESTATE, "marginal covariances: not available in distributed mode");
  int rc = ensure_ready(c);
  if (rc) return rc;
  std::vector<int> idx((size_t)n);
  for (int64_t q = 0; q < n; ++q) {
    auto it = c->id2idx.find(ids[q]);
    if (it == c->id2idx.end()) return fail(c, FGO_EINVAL, "unknown variable id");
    if (c->fixed[it->second]) return fail(c, FGO_EINVAL, "a fixed vertex has no marginal covariance");
    idx[q] = it->second;
  }
  hipStream_t s = c->stream;
  if (!c->lin_valid) { rc = linearize_current(c, false); if (rc) return rc; c->cov_factor_valid = false; }
  if (!c->cov_factor_valid) {
    c->h_scal[3] = 0.0;
    HIPCHK(c, hipMemcpyAsync(c->d_scal.p + 3, c->h_scal + 3, sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(c->d_fail.p, 0, sizeof(int), s));
    launch_factor(c->plan, c->sched, c->d_H[c->cur].p, c->d_L.p, c->d_scal.p + 3, c->d_fail.p, s);
    HIPCHK(c, hipMemcpyAsync(c->h_fail, c->d_fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (*c->h_fail) return fail(c, FGO_ENUM, "information matrix not positive definite (gauge freedom left?)");
    c->cov_factor_valid = true;
  }
  if (c->h_pose_col.size() != c->ids.size()) {           // permuted column of every variable (host copy, once per structure)
    c->h_pose_col.resize(c->ids.size());
    HIPCHK(c, hipMemcpy(c->h_pose_col.data(), c->d_pose_col.p, sizeof(int) * c->h_pose_col.size(), hipMemcpyDeviceToHost));
  }
  const int nb = c->plan.nb;
  DevBuf<double> rhs;
  HIPCHK(c, rhs.alloc((size_t)nb * 6));
  for (int64_t q = 0; q < n; ++q) {
    const int col = c->h_pose_col[idx[q]];
    double blk[6];
    for (int k = 0; k < 6; ++k) {
      HIPCHK(c, hipMemsetAsync(rhs.p, 0, sizeof(double) * (size_t)nb * 6, s));
      const double one = 1.0;
      HIPCHK(c, hipMemcpyAsync(rhs.p + 6 * (size_t)col + k, &one, sizeof(double), hipMemcpyHostToDevice, s));
      launch_solve(c->plan, c->sched, c->d_L.p, rhs.p, c->d_x.p, s);
      HIPCHK(c, hipMemcpyAsync(blk, c->d_x.p + 6 * (size_t)col, sizeof(blk), hipMemcpyDeviceToHost, s));
      HIPCHK(c, hipStreamSynchronize(s));
      for (int r = 0; r < 6; ++r) cov36[36 * q + r * 6 + k] = blk[r];
    }
  }
  HIPCHK(c, hipGetLastError());
  return FGO_OK;
}

// Marginals(graph, values, CHOLESKY).marginalCovariance(key): the (id, id) block of (J' Omega J)^-1 at the current
// linearisation (gtsam/gtsam_graph.cpp:598-601).  The reference pays a full batch factorisation per Marginals object (and
// builds one it never uses at :1357); here the factor stays resident in HBM across calls.
int fgo_marginal_cov(fgo_ctx *c, int64_t id, double *cov36) try {
  if (!c || !cov36) return FGO_EINVAL;
  return marginal_blocks(c, 1, &id, cov36);
} FGO_CATCH_INT(c)

int fgo_marginal_cov_many(fgo_ctx *c, int64_t n, const int64_t *ids, double *cov36) try {
  if (!c || n < 0 || (n > 0 && (!ids || !cov36))) return FGO_EINVAL;
  return n == 0 ? FGO_OK : marginal_blocks(c, n, ids, cov36);
} FGO_CATCH_INT(c)

int fgo_solve_step(fgo_ctx *c, double lambda, double *delta_out) try {
  if (!c || !delta_out) return FGO_EINVAL;
  if (c->shard_world > 1) return fail(c, FGO_ESTATE, "fgo_solve_step: not available in distributed mode");
  (void)hipSetDevice(c->cfg.device);
  int rc = ensure_ready(c);
  if (rc) return rc;
  if (!c->lin_valid) { rc = linearize_current(c, false); if (rc) return rc; }
  hipStream_t s = c->stream;
  c->h_scal[3] = lambda;
  HIPCHK(c, hipMemcpyAsync(c->d_scal.p + 3, c->h_scal + 3, sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemsetAsync(c->d_fail.p, 0, sizeof(int), s));
  c->cov_factor_valid = false;
  launch_factor(c->plan, c->sched, c->d_H[c->cur].p, c->d_L.p, c->d_scal.p + 3, c->d_fail.p, s);
  launch_solve(c->plan, c->sched, c->d_L.p, c->d_b[c->cur].p, c->d_x.p, s);
  const int nb = c->plan.nb;
  std::vector<double> x((size_t)nb * 6);
  HIPCHK(c, hipMemcpyAsync(x.data(), c->d_x.p, sizeof(double) * x.size(), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(c->h_fail, c->d_fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  for (int k = 0; k < nb; ++k) std::memcpy(delta_out + (size_t)c->S.perm[k] * 6, &x[(size_t)k * 6], 6 * sizeof(double));
  if (*c->h_fail) return fail(c, FGO_ENUM, "block Cholesky: matrix not positive definite");
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_bench_phase(fgo_ctx *c, int phase, int reps, double *ms_out) try {
  if (!c || reps < 1 || !ms_out || phase < 0 || phase > 2) return FGO_EINVAL;
  if (c->shard_world > 1) return fail(c, FGO_ESTATE, "fgo_bench_phase: not available in distributed mode");
  (void)hipSetDevice(c->cfg.device);
  int rc = ensure_ready(c);
  if (rc) return rc;
  if (!c->lin_valid) { rc = linearize_current(c, true); if (rc) return rc; }
  hipStream_t s = c->stream;
  c->cov_factor_valid = false;
  if (phase >= 1) {   // make sure lambda and (for the solve) a valid factor are in place
    c->h_scal[3] = 1e-5 * std::max(1.0, c->h_scal[2]);
    HIPCHK(c, hipMemcpyAsync(c->d_scal.p + 3, c->h_scal + 3, sizeof(double), hipMemcpyHostToDevice, s));
    launch_factor(c->plan, c->sched, c->d_H[c->cur].p, c->d_L.p, c->d_scal.p + 3, c->d_fail.p, s, c->d_b[c->cur].p, c->d_x.p);
  }
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipEventRecord(c->ev[0], s));
  for (int r = 0; r < reps; ++r) {
    if (phase == 0 && c->gtsam_mode) launch_linearize_gtsam(c->plan, c->d_poses[c->cur].p, c->d_H[c->cur].p, c->d_b[c->cur].p, c->d_scal.p + 0, s);
    else if (phase == 0) launch_linearize(c->plan, c->d_poses[c->cur].p, c->d_H[c->cur].p, c->d_b[c->cur].p, c->d_scal.p + 0, s);
    else if (phase == 1) launch_factor(c->plan, c->sched, c->d_H[c->cur].p, c->d_L.p, c->d_scal.p + 3, c->d_fail.p, s, c->d_b[c->cur].p, c->d_x.p);
    else launch_solve(c->plan, c->sched, c->d_L.p, c->d_b[c->cur].p, c->d_x.p, s, true);   // what a trial runs: backward sweep only
  }
  HIPCHK(c, hipEventRecord(c->ev[1], s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  *ms_out = (double)ms / reps;
  return FGO_OK;
} FGO_CATCH_INT(c)

}  // extern "C"
