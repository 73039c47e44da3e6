// Distributed mode of libfgo (fgo_set_shard, world > 1): RCCL / hook transport, the collectives of an LM trial.
#include "fgo_ctx.hpp"

using namespace fgo;

#include <dlfcn.h>

namespace fgo {

// ---- RCCL, resolved at run time: single-GPU users need no RCCL, and a process that already carries one (torch) keeps it
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
    api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
    if (!api.GroupStart || !api.GroupEnd) api.GroupStart = api.GroupEnd = nullptr;
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

// Two sums that belong to the same point of a trial (tail of L + tail of x; the top's gradient + the trial's scalars): on RCCL
// they are grouped -- ONE launch, ONE traversal of the ring instead of two, which is what counts for collectives of a few
// hundred KB over per-link-bound xGMI.  Hook transport: two calls.
int dist_allreduce2(fgo_ctx *c, double *buf_a, int64_t na, double *buf_b, int64_t nb) {
  if (c->shard_world <= 1) return FGO_OK;
  RcclApi *api = c->rccl ? rccl_api() : nullptr;
  const bool group = api && api->GroupStart && na > 0 && nb > 0;
  if (group && api->GroupStart() != ncclSuccess) return fail(c, FGO_ENODEV, "ncclGroupStart failed");
  int rc = dist_allreduce(c, buf_a, na);
  if (rc == FGO_OK) rc = dist_allreduce(c, buf_b, nb);
  if (group && api->GroupEnd() != ncclSuccess && rc == FGO_OK) rc = fail(c, FGO_ENODEV, "ncclGroupEnd failed");
  return rc;
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
  if (c->rccl) {                                         // RCCL knows max: in place, on the stream, no host round trip
    c->xgmi_bytes += 8.0;
    const ncclResult_t r = rccl_api()->AllReduce(c->d_scal.p + slot, c->d_scal.p + slot, 1, ncclDouble, ncclMax, c->rccl, s);
    if (r != ncclSuccess) return fail(c, FGO_ENODEV, "ncclAllReduce(max) failed");
    return FGO_OK;
  }
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
// The distributed entry points are COLLECTIVE: a rank that bailed out before the first collective of a call (its structure
// build failed, it has nothing to optimise, an allocation failed) would leave the others blocked in theirs.  So the
// ranks agree on a status word first: every rank contributes (rc != 0) and all of them return an error if any did.
// (Failures INSIDE the trial loops are already agreed -- the failure flag rides in the scalar collective; a HIP or
// transport error in the middle of a trial is fatal for the communicator: fgo.h.)
int dist_agree(fgo_ctx *c, int rc) {
  if (c->shard_world <= 1) return rc;
  hipStream_t s = c->stream;
  if (c->d_status.alloc(1) != hipSuccess) return rc ? rc : fail(c, FGO_ENOMEM, "status word allocation failed");
  const double mine = rc ? 1.0 : 0.0;
  double sum = mine;
  if (hipMemcpyAsync(c->d_status.p, &mine, sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) return rc ? rc : FGO_ENODEV;
  const std::string keep = c->err;                          // the rank's own message survives the exchange
  const int rc2 = dist_allreduce(c, c->d_status.p, 1);
  if (rc2) return rc ? rc : rc2;
  if (hipMemcpyAsync(&sum, c->d_status.p, sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return rc ? rc : FGO_ENODEV;
  if (rc) { c->err = keep; return rc; }
  if (sum != 0.0) return fail(c, FGO_ESTATE, "distributed mode: another rank failed before the first collective of this call");
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
  // landmark elimination (kernels_ba.hip) in distributed mode: a landmark belongs to the rank of its cameras' domain (its cameras
  // are a clique of the reduced graph: one domain + the top), which forms  H - sum over ITS landmarks  for its domain's blocks
  // (complete) and for the top's (a partial sum like H of the top itself: the collective on the tail of L completes it) -- the north
  // star's "all-reduce on the off-diagonal Hessian contributions" for gtsam/gtsam_graph.cpp:370-448 graphs
  const double *H = c->d_H[cur].p, *b = c->d_b[cur].p;
  if (c->ba.on) { H = c->ba.d_Hred.p; b = c->ba.d_bred.p; }
  if (which == 0) {
    launch_zero_flag(c->d_fail.p, s);
    if (c->ba.on)
      launch_ba_reduce(c->plan, c->ba.d_W[cur].p, c->ba.d_Hpp[cur].p, c->ba.d_bp[cur].p, c->d_H[cur].p, c->d_b[cur].p, c->ba.d_Hred.p, c->ba.d_bred.p, scal + 3, c->d_fail.p, s);
    launch_factor(c->plan, c->sched, H, c->d_L.p, scal + 3, c->d_fail.p, s, b, c->d_x.p, PHASE_DOMAIN, nullptr, c->ba.on ? c->d_b[cur].p : nullptr);
  } else {
    launch_factor(c->plan, c->sched, H, c->d_L.p, scal + 3, c->d_fail.p, s, b, c->d_x.p, PHASE_TOP);
    launch_solve(c->plan, c->sched, c->d_L.p, b, c->d_x.p, s, true, PHASE_TOP);
    if (c->ba.on) launch_ba_back(c->plan, c->ba.d_W[cur].p, c->ba.d_bp[cur].p, c->d_x.p, s);
    if (c->gtsam_mode) launch_update_gtsam(c->plan, c->d_poses[cur].p, c->d_poses[cand].p, c->d_x.p, c->d_b[cur].p, scal + 3, scal + 1, s);
    else launch_update(c->plan, c->d_poses[cur].p, c->d_poses[cand].p, c->d_x.p, c->d_b[cur].p, scal + 3, scal + 1, s);
    if (c->gtsam_mode) launch_linearize_gtsam(c->plan, c->d_poses[cand].p, c->d_H[cand].p, c->d_b[cand].p, scal + 4, s,
                                             c->ba.on ? c->ba.d_W[cand].p : nullptr, c->ba.on ? c->ba.d_Hpp[cand].p : nullptr, c->ba.on ? c->ba.d_bp[cand].p : nullptr);
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
  HIPCHK(c, hipEventRecord(c->ev[1], s));
  static const bool dbg_fail = std::getenv("FGO_DEBUG_TRIALS") != nullptr;
  if (dbg_fail) { int hf0 = -1; (void)hipMemcpyAsync(&hf0, c->d_fail.p, sizeof(int), hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); std::fprintf(stderr, "[fgo trial] rank %d fail flag after the domain phase: %d\n", c->shard_rank, hf0); }
  // collective 1: the domains' updates into the top of the factor and of the right-hand side (both are contiguous tails)
  rc = dist_allreduce2(c, c->d_L.p + 36 * (size_t)c->plan.top_blk0, 36 * c->sched.n_top_blocks,
                       c->d_x.p + 6 * (size_t)c->plan.top_col0, 6 * (int64_t)c->sched.n_top_cols);
  if (rc) return rc;
  HIPCHK(c, hipEventRecord(c->ev[2], s));
  rc = launch_dist_phase(c, 1);
  if (rc) return rc;
  HIPCHK(c, hipEventRecord(c->ev[3], s));
  // the gradient of the top is a partial sum like H_top; it is completed once per linearisation (the LM scale and
  // k_dist_rhs on rank 0 read the complete one)
  // collective 2 (grouped with it): scalars, three: [4] chi2 of the candidate (a partial sum over this rank's factors),
  // [5] the failure flag, [6] the LM scale (k_update sums the columns this rank is responsible for)
  launch_pack_scalars(c->d_scal.p, c->d_fail.p, s);
  rc = dist_allreduce2(c, c->d_b[c->cur ^ 1].p + 6 * (size_t)c->plan.top_col0, 6 * (int64_t)c->sched.n_top_cols, c->d_scal.p + 4, 3);
  if (rc) return rc;
  HIPCHK(c, hipEventRecord(c->ev[4], s));
  HIPCHK(c, hipMemcpyAsync(c->h_scal + 4, c->d_scal.p + 4, sizeof(double) * 3, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  c->h_scal[1] = c->h_scal[6];
  *chi_cand = c->h_scal[4]; *scale = c->h_scal[1]; *failed = c->h_scal[5] != 0.0;
  if (st) {
    // whole trial, and its two halves: ms_factor = the rank's own domain (sub-trees + its contributions to the top), ms_solve = the
    // replicated top + backward sweep + update + linearisation of the candidate (the collectives lie between / after them)
    float ms = 0;
    (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[4]); st->reserved[0] += ms;
    (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]); st->ms_factor += ms;
    (void)hipEventElapsedTime(&ms, c->ev[2], c->ev[3]); st->ms_solve += ms;
  }
  return FGO_OK;
}

}  // namespace fgo

extern "C" {

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
    // the three forms a distributed trial uses: a plain sum, two sums in one group (dist_allreduce2), a max (dist_max_scalar) --
    // on a 1-rank communicator all are the identity, so the buffer comes back unchanged if the calls are accepted
    RcclApi *api = rccl_api();
    ncclResult_t r = api->AllReduce(d.p, d.p, (size_t)n, ncclDouble, ncclSum, c->rccl, c->stream);
    if (r != ncclSuccess) return fail(c, FGO_ENODEV, "ncclAllReduce failed");
    if (api->GroupStart && n >= 2) {
      const size_t h = (size_t)n / 2;
      if (api->GroupStart() != ncclSuccess) return fail(c, FGO_ENODEV, "ncclGroupStart failed");
      r = api->AllReduce(d.p, d.p, h, ncclDouble, ncclSum, c->rccl, c->stream);
      const ncclResult_t r2 = api->AllReduce(d.p + h, d.p + h, (size_t)n - h, ncclDouble, ncclSum, c->rccl, c->stream);
      if (api->GroupEnd() != ncclSuccess || r != ncclSuccess || r2 != ncclSuccess) return fail(c, FGO_ENODEV, "grouped ncclAllReduce failed");
    }
    r = api->AllReduce(d.p, d.p, 1, ncclDouble, ncclMax, c->rccl, c->stream);
    if (r != ncclSuccess) return fail(c, FGO_ENODEV, "ncclAllReduce(max) failed");
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

}  // extern "C"
