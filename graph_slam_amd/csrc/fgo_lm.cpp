// Levenberg-Marquardt controllers of libfgo (g2o and GTSAM semantics) driving the HIP kernels.
//
// LM semantics restated from g2o's OptimizationAlgorithmLevenberg as the reference configures it
// (g2o/g2o_graph.cpp:65-77: LM over BlockSolver<6,3> over a sparse Cholesky; :241-252: optimize(2) x 10):
//   iteration 0 of a call: lambda = 1e-5 * max|diag H|, nu = 2;   each iteration: up to 10 trials of
//   { (H + lambda I) d = b ; x (+) d ; rho = (chi2 - chi2') / (d.(lambda d + b) + 1e-3) } with
//   accept: lambda *= max(1/3, min(1 - (2 rho - 1)^3, 2/3)), nu = 2;  reject: lambda *= nu, nu *= 2.
// All state stays in HBM; per trial only chi2', scale and the failure flag cross PCIe (24 bytes).
#include "fgo_ctx.hpp"

using namespace fgo;

namespace fgo {

// one LM trial on the stream: factor, solve, update into the candidate buffers, linearise there
void ctx_linearize(fgo_ctx *c, int buf, double *scalar_out) {
  hipStream_t s = c->stream;
  if (c->gtsam_mode)
    launch_linearize_gtsam(c->plan, c->d_poses[buf].p, c->d_H[buf].p, c->d_b[buf].p, scalar_out, s,
                           c->ba.on ? c->ba.d_W[buf].p : nullptr, c->ba.on ? c->ba.d_Hpp[buf].p : nullptr, c->ba.on ? c->ba.d_bp[buf].p : nullptr);
  else launch_linearize(c->plan, c->d_poses[buf].p, c->d_H[buf].p, c->d_b[buf].p, scalar_out, s);
}
void ctx_factor(fgo_ctx *c, int buf, bool with_rhs) {
  hipStream_t s = c->stream;
  const double *H = c->d_H[buf].p, *b = c->d_b[buf].p;
  if (c->ba.on) {
    launch_ba_reduce(c->plan, c->ba.d_W[buf].p, c->ba.d_Hpp[buf].p, c->ba.d_bp[buf].p, H, b, c->ba.d_Hred.p, c->ba.d_bred.p, c->d_scal.p + 3, c->d_fail.p, s);
    H = c->ba.d_Hred.p; b = c->ba.d_bred.p;
  }
  launch_factor(c->plan, c->sched, H, c->d_L.p, c->d_scal.p + 3, c->d_fail.p, s, with_rhs ? b : nullptr, with_rhs ? c->d_x.p : nullptr);
}
void ctx_solve(fgo_ctx *c, int buf, bool fwd_done) {
  hipStream_t s = c->stream;
  launch_solve(c->plan, c->sched, c->d_L.p, c->ba.on ? c->ba.d_bred.p : c->d_b[buf].p, c->d_x.p, s, fwd_done);
  if (c->ba.on) launch_ba_back(c->plan, c->ba.d_W[buf].p, c->ba.d_bp[buf].p, c->d_x.p, s);
}
void ba_off(fgo_ctx *c) {
  if (c->ba_disable) return;
  c->ba_disable = true;
  if (c->ba.on) c->structure_dirty = true;
}

void enqueue_trial(fgo_ctx *c, int cur, bool with_events) {
  const int cand = cur ^ 1;
  hipStream_t s = c->stream;
  double *scal = c->d_scal.p;
  static const bool host_scalars = tune("host_scalars", 1) != 0;   // 0: the blit copies of rounds 1-5 (A/B)
  if (host_scalars) launch_fetch_scalar(scal + 3, c->h_scal + 3, s);      // lambda: the host wrote it into its pinned block before the launch
  launch_zero_flag(c->d_fail.p, s);
  if (with_events) (void)hipEventRecord(c->ev[0], s);
  ctx_factor(c, cur, true);                                                                                        // + forward solve
  if (with_events) (void)hipEventRecord(c->ev[1], s);
  ctx_solve(c, cur, true);                                                                                         // backward sweep
  if (with_events) (void)hipEventRecord(c->ev[2], s);
  if (c->gtsam_mode) launch_update_gtsam(c->plan, c->d_poses[cur].p, c->d_poses[cand].p, c->d_x.p, c->d_b[cur].p, scal + 3, scal + 1, s);
  else launch_update(c->plan, c->d_poses[cur].p, c->d_poses[cand].p, c->d_x.p, c->d_b[cur].p, scal + 3, scal + 1, s);
  if (with_events) (void)hipEventRecord(c->ev[3], s);
  ctx_linearize(c, cand, scal + 4);
  launch_pack_scalars(scal, c->d_fail.p, s, host_scalars ? c->h_scal : nullptr);   // [5] <- failure flag, [6] <- LM scale, next to [4] (chi2) -- and all three into the host's pinned block: no copy in either direction
  if (with_events) (void)hipEventRecord(c->ev[4], s);
}

int run_trial(fgo_ctx *c, double lambda, double *chi_cand, double *scale, int *failed, fgo_stats *st) {
  c->cov_factor_valid = false;
  c->isam_L_valid = false;
  if (c->shard_world > 1) return run_trial_dist(c, lambda, chi_cand, scale, failed, st);
  hipStream_t s = c->stream;
  c->h_scal[3] = lambda;                                 // (read by the trial's first kernel node)
  static const bool host_scalars = tune("host_scalars", 1) != 0;
  if (!host_scalars) HIPCHK(c, hipMemcpyAsync(c->d_scal.p + 3, c->h_scal + 3, sizeof(double), hipMemcpyHostToDevice, s));
  if (c->use_graph) {
    hipGraphExec_t &ge = c->trial_graph[c->cur];
    if (!ge) {
      // one capture at a time per process: two contexts capturing from two host threads at once (thread-local mode)
      // occasionally produced a graph that computes garbage (tools/stress_shard_threads.py: 12 of 60 runs diverged,
      // none without graphs or with this lock)
      static std::mutex capture_mutex;
      std::lock_guard<std::mutex> capture_lock(capture_mutex);
      const double t_cap = now_s();
      hipGraph_t graph = nullptr;
      HIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      enqueue_trial(c, c->cur, false);
      HIPCHK(c, hipStreamEndCapture(s, &graph));
      HIPCHK(c, hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
      if (std::getenv("FGO_SYM_PROFILE")) std::fprintf(stderr, "[fgo lm]       trial graph captured + instantiated in %.1f ms\n", 1e3 * (now_s() - t_cap));
    }
    HIPCHK(c, hipEventRecord(c->ev[0], s));
    HIPCHK(c, hipGraphLaunch(ge, s));
    HIPCHK(c, hipEventRecord(c->ev[4], s));
  } else {
    enqueue_trial(c, c->cur, true);
  }
  if (!host_scalars) HIPCHK(c, hipMemcpyAsync(c->h_scal + 4, c->d_scal.p + 4, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));                     // chi2', failure flag, scale are in h_scal[4..6] (k_pack_scalars wrote them)
  HIPCHK(c, hipGetLastError());
  c->h_scal[1] = c->h_scal[6]; *c->h_fail = c->h_scal[5] != 0.0;
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
  ctx_linearize(c, c->cur, c->d_scal.p + 0);
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

}  // namespace fgo

extern "C" {

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
  if (was_dirty && std::getenv("FGO_SYM_PROFILE")) std::fprintf(stderr, "[fgo lm]       structure ready after %.1f ms\n", 1e3 * (now_s() - tstart));
  if (rc == FGO_OK && c->gtsam_mode) rc = fail(c, FGO_EINVAL, "GTSAM-semantics graph: use fgo_optimize_gtsam");
  rc = dist_agree(c, rc);                 // distributed: all ranks continue or none does
  if (rc) return rc;
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

double fgo_error(fgo_ctx *c) { return 0.5 * fgo_chi2(c); }

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
  if (rc == FGO_OK && !c->gtsam_mode) rc = fail(c, FGO_EINVAL, "g2o-semantics graph: use fgo_optimize");
  rc = dist_agree(c, rc);                 // distributed: all ranks continue or none does
  if (rc) return rc;
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

}  // extern "C"
