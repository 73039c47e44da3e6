// ISAM2 semantics on the batch machinery (fgo_isam2_*).
#include "fgo_ctx.hpp"

using namespace fgo;

extern "C" {

// ISAM2::update + calculateEstimate on the batch machinery (kernels_gtsam.hip: k_isam2_relin / k_isam2_estimate)
int fgo_isam2_update(fgo_ctx *c, double relin_threshold, fgo_stats *stats) try {
  if (!c || !(relin_threshold >= 0)) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  const double tstart = now_s();
  if (c->shard_world > 1) return fail(c, FGO_ESTATE, "fgo_isam2_update is not available in distributed mode");
  // a context that is updated incrementally builds its structure with room to grow (phantom variable slots + factor
  // capacity), so that the per-record updates of the reference's drivers do not pay the structure phase every time
  const bool incr_off = std::getenv("FGO_ISAM_INCREMENTAL") && std::atoi(std::getenv("FGO_ISAM_INCREMENTAL")) == 0;
  const bool incr_was = c->isam_incremental;            // (captured BEFORE the switch below: the failure path restores it -- ADVICE r5)
  if (!incr_off) c->isam_incremental = true;
  // ISAM2 keeps theta / delta for EVERY variable and factors H as linearised: a structure built with the landmarks eliminated
  // (fgo_optimize_gtsam ran first, or FGO_ISAM_INCREMENTAL=0) cannot serve it -> generic form from here on, until
  // fgo_isam2_reset.  A call that fails (build error, g2o-semantics graph) leaves the context as it found it.
  const bool ba_was_disabled = c->ba_disable;
  ba_off(c);
  const bool was_dirty = c->structure_dirty;
  int rc = ensure_ready(c);
  if (rc == FGO_OK && !c->gtsam_mode) rc = fail(c, FGO_EINVAL, "g2o-semantics graph: ISAM2 semantics need a GTSAM-semantics graph");
  if (rc) {
    if (!ba_was_disabled) { c->ba_disable = false; c->structure_dirty = true; }   // (the next build decides again)
    if (c->isam_incremental != incr_was) { c->isam_incremental = incr_was; c->structure_dirty = true; }   // (a structure built with the reserve is not the batch one)
    return rc;
  }
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
  const int64_t isam_n_before = c->isam_n;
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
  // ---- partial re-factorisation.  ISAM2 re-eliminates only the cliques on the paths from the affected variables to the
  // root of the Bayes tree; here: only the TASKS of the elimination tree on those paths are run again (k_* kernels skip
  // the others), every other column keeps its blocks of L and its entry of the forward solution y from the previous
  // step.  Affected = variables that were relinearised + everything that shares a factor with them (those factors are
  // re-linearised) + the variables of factors and variables added since the previous step.
  // FGO_ISAM_PARTIAL=0 switches it off (A/B measurements).  Measured, one new pose per update on a settled graph (1xMI355X,
  // tools/isam_build_breakdown.py): 2 000 poses 1.30 vs 1.35 ms, 20 000 poses 2.13 vs 2.37 ms, 100 000 poses 3.30 vs 5.01 ms
  // device per update -- ~25-30 of 200 / 1 750 / 8 600 tasks are re-run; what remains is one latency-bound pivot chain per level
  // of the path, which is why the gain grows with the graph.
  const char *pe = std::getenv("FGO_ISAM_PARTIAL");
  const bool partial_on = !(pe && std::atoi(pe) == 0);
  const fgo_ctx::Incr &I = c->inc;
  const int64_t E = (int64_t)c->ei.size(), NI = (int64_t)c->imu_payload.size(), NPr = (int64_t)c->prior_v.size();
  const bool have_tables = !c->col_task.empty() && I.valid && c->d_y.p != nullptr;
  const bool partial = partial_on && have_tables && c->isam_L_valid;
  // which variables k_isam2_relin moves is known since the END of the previous update (same delta, same test: k_isam2_estimate)
  // unless the threshold or the structure changed in between -- then the flags come back here, at the price of a synchronisation
  static const bool look_on = tune("isam_lookahead", 1) != 0;
  static const bool mask_on = tune("isam_masked", 1) != 0;
  const bool look = look_on && partial && c->isam_moved_valid && c->isam_moved_thr == relin_threshold && c->isam_moved_nx == NX;
  const bool maskable = mask_on && have_tables && c->d_chi_var.p != nullptr && linearize_gtsam_maskable(c->plan);
  const bool masked = maskable && partial && c->isam_H_valid;
  c->isam_moved_valid = false;
  c->isam_H_valid = false;
  launch_isam2_relin(c->plan, c->d_theta.p, c->d_delta.p, relin_threshold, scal + 5, s, (partial && !look) ? c->d_moved.p : nullptr);
  DevPlan plan = c->plan;
  int64_t n_dirty_tasks = -1;
  if (partial) {
    const int nb = c->plan.nb, ntask = (int)c->S.task_ptr.size() - 1;
    // pinned staging; all-zero between calls: what a call sets it remembers (isam_set_*) and the next one clears -- the host
    // side of an update is O(affected), not O(graph), apart from one pass over the `moved` bytes
    unsigned char *moved = c->h_flags, *task_dirty = moved + NX, *col_dirty = task_dirty + ntask, *aff = col_dirty + nb + NX;
    if (look) moved = col_dirty + nb;
    else {
      HIPCHK(c, hipMemcpyAsync(moved, c->d_moved.p, (size_t)NX, hipMemcpyDeviceToHost, s));
      HIPCHK(c, hipEventRecord(c->ev[5], s));
    }
    if (!masked) {                                        // the full linearisation does not depend on the flags: it runs while the host walks the tree
      if (maskable) launch_linearize_gtsam_masked(c->plan, c->d_theta.p, c->d_H[w].p, c->d_b[w].p, scal + 4, s, nullptr, c->d_chi_var.p);
      else launch_linearize_gtsam(c->plan, c->d_theta.p, c->d_H[w].p, c->d_b[w].p, scal + 4, s);
      HIPCHK(c, hipEventRecord(c->ev[1], s));
    }
    for (int t : c->isam_set_tasks) task_dirty[t] = 0;
    for (int k : c->isam_set_cols) col_dirty[k] = 0;
    for (int64_t v : c->isam_set_aff) aff[v] = 0;
    c->isam_set_tasks.clear(); c->isam_set_cols.clear(); c->isam_set_aff.clear();
    std::vector<int64_t> &set_aff = c->isam_set_aff;
    auto mark = [&](int64_t v) { if (!aff[v]) { aff[v] = 1; set_aff.push_back(v); } };
    for (int64_t v = isam_n_before; v < N; ++v) mark(v);
    for (int64_t e = c->isam_E_seen; e < E; ++e) { mark(c->ei[e]); mark(c->ej[e]); }
    for (int64_t f = c->isam_NI_seen; f < NI; ++f) for (int u = 0; u < 6; ++u) mark(c->imu_ids[6 * f + u]);
    for (int64_t q = c->isam_NP_seen; q < NPr; ++q) mark(c->prior_v[q]);
    if (!look) HIPCHK(c, hipEventSynchronize(c->ev[5]));
    auto moved_var = [&](int64_t v) {
      mark(v);
      for (int64_t p = I.he_ptr[v]; p < I.he_ptr[v + 1]; ++p) { const int64_t e = I.he[p] >> 1; mark(c->ei[e]); mark(c->ej[e]); }
      for (int64_t p = I.imu_inc_ptr[v]; p < I.imu_inc_ptr[v + 1]; ++p) { const int64_t f = I.imu_inc[p] >> 3; for (int u = 0; u < 6; ++u) mark(c->imu_ids[6 * f + u]); }
    };
    {
      int64_t v = 0;
      for (; v + 8 <= N; v += 8) {                        // (eight flags per load: a settled graph has none set)
        uint64_t wd; std::memcpy(&wd, moved + v, 8);
        if (!wd) continue;
        for (int u = 0; u < 8; ++u) if (moved[v + u]) moved_var(v + u);
      }
      for (; v < N; ++v) if (moved[v]) moved_var(v);
    }
    std::vector<int> &set_cols = c->isam_set_cols, &set_tasks = c->isam_set_tasks;
    for (size_t q = 0; q < set_aff.size(); ++q)
      for (int k = I.pose_col[set_aff[q]]; k >= 0 && !col_dirty[k]; k = c->S.parent[k]) { col_dirty[k] = 1; set_cols.push_back(k); }   // up the elimination tree
    for (size_t q = 0, n0 = set_cols.size(); q < n0; ++q) {
      const int t = c->col_task[set_cols[q]];
      if (!task_dirty[t]) { task_dirty[t] = 1; set_tasks.push_back(t); }
    }
    n_dirty_tasks = (int64_t)set_tasks.size();
    // a task is re-run as a whole: all of its columns take part in the forward solve again
    for (int t : set_tasks)
      for (int q = c->S.task_ptr[t]; q < c->S.task_ptr[t + 1]; ++q) { const int k = c->S.task_cols[q]; if (!col_dirty[k]) { col_dirty[k] = 1; set_cols.push_back(k); } }
    // when most of the tree is affected (a relinearisation wave after a loop closure) the full sweep is the faster one:
    // the flag look-ups cost every workgroup two extra dependent loads
    if (n_dirty_tasks > (int64_t)(0.3 * ntask)) n_dirty_tasks = -2;
    if (n_dirty_tasks >= 0) {
      HIPCHK(c, hipMemcpyAsync(c->d_task_dirty.p, task_dirty, (size_t)ntask, hipMemcpyHostToDevice, s));
      HIPCHK(c, hipMemcpyAsync(c->d_col_dirty.p, col_dirty, (size_t)nb, hipMemcpyHostToDevice, s));
      // (pinned staging: no synchronisation needed; the buffers are not touched again before the stream is drained below)
      plan.task_dirty = c->d_task_dirty.p;
    }
    if (masked) {
      HIPCHK(c, hipMemcpyAsync(c->d_lin_mask.p, aff, (size_t)NX, hipMemcpyHostToDevice, s));
      launch_linearize_gtsam_masked(c->plan, c->d_theta.p, c->d_H[w].p, c->d_b[w].p, scal + 4, s, c->d_lin_mask.p, c->d_chi_var.p);
      HIPCHK(c, hipEventRecord(c->ev[1], s));
    }
  }
  if (!partial) {
    if (maskable) launch_linearize_gtsam_masked(c->plan, c->d_theta.p, c->d_H[w].p, c->d_b[w].p, scal + 4, s, nullptr, c->d_chi_var.p);
    else launch_linearize_gtsam(c->plan, c->d_theta.p, c->d_H[w].p, c->d_b[w].p, scal + 4, s);
    HIPCHK(c, hipEventRecord(c->ev[1], s));
  }
  c->cov_factor_valid = false;
  c->isam_L_valid = false;                              // (until this step has gone through)
  if (plan.task_dirty) launch_mix_rhs(plan, c->d_b[w].p, c->d_y.p, c->d_x.p, c->d_col_dirty.p, s);
  PartialSweep ps{};
  const bool ranged = plan.task_dirty && c->tk_ok && !c->task_level.empty();
  if (ranged) {                                           // per level: the tasks that cover its dirty ones
    const int nl = c->sched.n_levels;
    std::fill(c->lvl_lo.begin(), c->lvl_lo.end(), INT32_MAX);
    std::fill(c->lvl_hi.begin(), c->lvl_hi.end(), -1);
    for (int t : c->isam_set_tasks) { const int l = c->task_level[(size_t)t]; c->lvl_lo[(size_t)l] = std::min(c->lvl_lo[(size_t)l], t); c->lvl_hi[(size_t)l] = std::max(c->lvl_hi[(size_t)l], t); }
    if (std::getenv("FGO_ISAM_DEBUG")) {
      std::fprintf(stderr, "[isam] ranged sweep: %zu dirty tasks;", c->isam_set_tasks.size());
      for (int l = 0; l < nl; ++l) std::fprintf(stderr, " %d/%d", c->lvl_hi[(size_t)l] < c->lvl_lo[(size_t)l] ? 0 : c->lvl_hi[(size_t)l] - c->lvl_lo[(size_t)l] + 1, c->sched.level_ptr[l + 1] - c->sched.level_ptr[l]);
      std::fprintf(stderr, "\n");
    }
    ps = PartialSweep{c->lvl_lo.data(), c->lvl_hi.data(), c->tk_s0.data(), c->tk_s1.data(), c->tk_l0.data(), c->tk_l1.data(),
                      c->tk_g0.data(), c->tk_g1.data(), c->tk_c0.data(), c->tk_c1.data(), c->S.task_ptr.data()};
  }
  launch_factor(plan, c->sched, c->d_H[w].p, c->d_L.p, scal + 3, c->d_fail.p, s, c->d_b[w].p, c->d_x.p, PHASE_ALL, ranged ? &ps : nullptr);
  if (have_tables) launch_copy_vec(c->d_x.p, c->d_y.p, (int64_t)c->plan.nb * 6, s);       // y for the next step
  st.reserved[3] = (double)n_dirty_tasks;               // tasks re-run by this step (-1: full sweep, -2: full sweep because most of the tree was affected)
  HIPCHK(c, hipEventRecord(c->ev[2], s));
  // wildfire option: below the backward chain only what was re-factored or reads a delta that moved by >= the threshold
  Wildfire wf{c->d_bwd_run.p, c->d_chg.p, c->d_task_dirty.p, c->d_xprev.p, c->wild_thr};
  const bool wild = c->wild_thr > 0 && plan.task_dirty && c->wild_valid && c->d_xprev.p != nullptr && c->sched.bchain_low >= 0;
  c->wild_valid = false;
  launch_solve(c->plan, c->sched, c->d_L.p, c->d_b[w].p, c->d_x.p, s, true, PHASE_ALL, wild ? &wf : nullptr);
  // (the copy of this solution is what the NEXT update's cut compares against: only kept where a cut can apply at all)
  const bool wild_possible = c->wild_thr > 0 && have_tables && c->d_xprev.p && c->sched.bchain_low >= 0;
  if (wild_possible) launch_copy_vec(c->d_x.p, c->d_xprev.p, (int64_t)c->plan.nb * 6, s);
  st.reserved[4] = wild ? 1.0 : 0.0;                    // this update cut its back-substitution (wildfire)
  if (std::getenv("FGO_ISAM_DEBUG"))
    std::fprintf(stderr, "[isam] wildfire: thr %g, partial %d, solution of the previous update kept %d, backward chain from level %d (%d panels) -> cut %d\n",
                 c->wild_thr, plan.task_dirty != nullptr, (int)c->wild_valid, c->sched.bchain_low, c->sched.bchain_n, (int)wild);
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
  c->isam_L_valid = have_tables;
  c->isam_H_valid = maskable;
  c->wild_valid = wild_possible;
  c->isam_E_seen = E; c->isam_NI_seen = NI; c->isam_NP_seen = NPr;
  const bool look_next = have_tables && c->d_moved_next.p != nullptr && c->h_flags != nullptr;
  launch_isam2_estimate(c->plan, c->d_theta.p, c->d_x.p, c->d_delta.p, c->d_poses[c->cur].p, s, look_next ? c->d_moved_next.p : nullptr, relin_threshold);
  launch_chi2_gtsam(c->plan, c->d_poses[c->cur].p, scal + 0, s);
  HIPCHK(c, hipEventRecord(c->ev[4], s));
  HIPCHK(c, hipMemcpyAsync(c->h_scal, scal, sizeof(double), hipMemcpyDeviceToHost, s));
  if (look_next) {
    const int nb = c->plan.nb, ntask = (int)c->S.task_ptr.size() - 1;
    HIPCHK(c, hipMemcpyAsync(c->h_flags + (size_t)NX + (size_t)ntask + (size_t)nb, c->d_moved_next.p, (size_t)NX, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  if (look_next) { c->isam_moved_valid = true; c->isam_moved_thr = relin_threshold; c->isam_moved_nx = NX; }
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

// growth reserve of g2o-semantics contexts (same machinery, same parameters as fgo_isam2_reserve)
int fgo_set_growth(fgo_ctx *c, int reserve_variables, int window) try {
  if (!c || reserve_variables < 0 || window < 0) return FGO_EINVAL;
  const bool on = reserve_variables > 0;
  if (on != c->grow_incremental || (on && (c->isam_reserve != reserve_variables || (window > 0 && c->isam_window != window)))) {
    if (c->inc.valid || on) { c->inc.valid = false; c->structure_dirty = true; }   // the next use rebuilds with / without the reserve
  }
  c->grow_incremental = on;
  c->grow_auto = on ? 1 : 0;
  if (on) { c->isam_reserve = reserve_variables; if (window > 0) c->isam_window = window; }
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_isam2_set_wildfire(fgo_ctx *c, double threshold) try {
  if (!c || !(threshold >= 0)) return FGO_EINVAL;
  c->wild_thr = threshold;
  c->wild_valid = false;                                // (the next update solves everything and leaves its solution behind)
  return FGO_OK;
} FGO_CATCH_INT(c)

int fgo_isam2_reset(fgo_ctx *c) try {
  if (!c) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->d_theta.release(); c->d_delta.release();
  c->isam_n = 0;
  c->isam_moved_valid = false; c->isam_H_valid = false; c->wild_valid = false;
  c->isam_L_valid = false; c->isam_E_seen = c->isam_NI_seen = c->isam_NP_seen = 0;
  // the growth reserve belongs to the incremental driving mode: a context that leaves it (delete isam2) goes back to a
  // structure without phantom slots at its next use; the next fgo_isam2_update lays a fresh reserve down
  c->isam_incremental = false;
  if (c->n_phantom > 0 || c->inc.valid) { c->inc.valid = false; c->structure_dirty = true; }
  // ... and so does the generic (not landmark-eliminated) form fgo_isam2_update had switched the context to
  if (c->ba_disable) { c->ba_disable = false; c->structure_dirty = true; }
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

}  // extern "C"
