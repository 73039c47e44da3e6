// Read-back, marginal covariances, stand-alone solve and the bench hooks of libfgo.
#include "fgo_ctx.hpp"

using namespace fgo;

extern "C" {

int fgo_debug_read_system(fgo_ctx *c, double *H, double *b, double *chi2) try {
  if (!c) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  ba_off(c);                                            // the caller wants the whole system: no landmark elimination
  int rc = ensure_ready(c);
  if (rc) return rc;
  if (c->n_phantom > 0) return fail(c, FGO_ESTATE, "fgo_debug_read_system: the structure carries the growth reserve of the incremental mode (fgo_isam2_reset drops it)");
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

// profiling hook: the first n doubles of the reduction scratch (FGO_TRI_PROF stamps of k_panel_tri land there)
int fgo_debug_read_scratch(fgo_ctx *c, double *out, int64_t n) try {
  if (!c || !out || n <= 0 || (size_t)n > c->d_partial.n) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out, c->d_partial.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
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

// dense image of a block system in the structure's H layout (diagonal blocks in elimination order, then the oriented
// off-diagonal blocks): rows / columns in free-variable (hessian) order, phantom slots left out
static int dense_from_blocks(fgo_ctx *c, const double *dH, const double *db, double *H_dense, double *b_dense) {
  const int nb = c->plan.nb;
  const int nreal = nb - c->n_phantom;
  if (nreal > 4096) return fail(c, FGO_EINVAL, "dense read-back is limited to 4096 free poses");
  const size_t hblocks = (size_t)nb + (size_t)c->n_offdiag;
  std::vector<double> H(hblocks * 36), b((size_t)nb * 6);
  std::vector<int> asrc((size_t)c->S.nnzL);
  HIPCHK(c, hipMemcpy(H.data(), dH, sizeof(double) * H.size(), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(b.data(), db, sizeof(double) * b.size(), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(asrc.data(), c->d_asrc.p, sizeof(int) * asrc.size(), hipMemcpyDeviceToHost));
  const size_t m = (size_t)nreal * 6;
  if (H_dense) {
    std::memset(H_dense, 0, sizeof(double) * m * m);
    for (int k = 0; k < nb; ++k)
      for (int64_t t = c->S.colptr[k]; t < c->S.colptr[k + 1]; ++t) {
        if (asrc[t] < 0) continue;
        const int hr = c->S.perm[c->S.rowidx[t]], hc = c->S.perm[k];   // hessian (ascending-id) indices
        if (hr >= nreal || hc >= nreal) continue;                       // phantom slot
        const double *B = &H[(size_t)asrc[t] * 36];
        for (int r = 0; r < 6; ++r)
          for (int q = 0; q < 6; ++q) {
            H_dense[((size_t)hr * 6 + r) * m + (size_t)hc * 6 + q] = B[r * 6 + q];
            H_dense[((size_t)hc * 6 + q) * m + (size_t)hr * 6 + r] = B[r * 6 + q];
          }
      }
  }
  if (b_dense)
    for (int k = 0; k < nb; ++k) if (c->S.perm[k] < nreal) std::memcpy(b_dense + (size_t)c->S.perm[k] * 6, &b[(size_t)k * 6], 6 * sizeof(double));
  return FGO_OK;
}

int fgo_linearize(fgo_ctx *c, double *chi2_out, double *H_dense, double *b_dense, int64_t *n_free_out) try {
  if (!c) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  if (H_dense || b_dense) ba_off(c);                    // the dense system covers every free variable
  int rc = ensure_ready(c);
  if (rc) return rc;
  rc = linearize_current(c, false);
  if (rc) return rc;
  if (chi2_out) *chi2_out = c->chi_cur;
  // the phantom slots of the incremental mode (the last n_phantom hessian indices) are not the caller's variables: the
  // dense system has 6 * (nb - n_phantom) rows, the phantom rows / columns (identity blocks, zero gradient) are left out.
  // Eliminated landmarks ARE the caller's variables: the count is the same whether or not the dense system is asked for,
  // so the usual query-then-allocate pattern sizes its buffers for the system the second call writes
  if (n_free_out) *n_free_out = (int64_t)c->plan.nb - c->n_phantom + (c->ba.on ? c->ba.n_lm : 0);
  if (!H_dense && !b_dense) return FGO_OK;
  return dense_from_blocks(c, c->d_H[c->cur].p, c->d_b[c->cur].p, H_dense, b_dense);
} FGO_CATCH_INT(c)

// tests: the REDUCED camera system of a structure built with the landmarks eliminated (kernels_ba.hip), i.e.
// S = H_cc - W (H_pp + lambda I)^-1 W^T and g = b_c - W (H_pp + lambda I)^-1 b_p at the current estimate, dense, rows / columns =
// the free non-landmark variables in the order they were added.  lambda enters the landmark blocks only (the cameras' own
// damping is added by the factorisation).  FGO_ESTATE when the context's structure has no eliminated landmarks.
int fgo_debug_read_reduced(fgo_ctx *c, double lambda, double *H_dense, double *b_dense, int64_t *n_out) try {
  if (!c || !(lambda >= 0)) return FGO_EINVAL;
  (void)hipSetDevice(c->cfg.device);
  int rc = ensure_ready(c);
  if (rc) return rc;
  if (!c->ba.on) return fail(c, FGO_ESTATE, "fgo_debug_read_reduced: no landmarks are eliminated in this structure");
  rc = linearize_current(c, false);
  if (rc) return rc;
  if (n_out) *n_out = (int64_t)c->plan.nb - c->n_phantom;
  if (!H_dense && !b_dense) return FGO_OK;
  hipStream_t s = c->stream;
  c->h_scal[3] = lambda;
  HIPCHK(c, hipMemcpyAsync(c->d_scal.p + 3, c->h_scal + 3, sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemsetAsync(c->d_fail.p, 0, sizeof(int), s));
  c->cov_factor_valid = false;
  launch_ba_reduce(c->plan, c->ba.d_W[c->cur].p, c->ba.d_Hpp[c->cur].p, c->ba.d_bp[c->cur].p, c->d_H[c->cur].p, c->d_b[c->cur].p,
                   c->ba.d_Hred.p, c->ba.d_bred.p, c->d_scal.p + 3, c->d_fail.p, s);
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  return dense_from_blocks(c, c->ba.d_Hred.p, c->ba.d_bred.p, H_dense, b_dense);
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
  if (c->ba.on) {
    // cameras: the inverse of the reduced system IS their marginal; an eliminated landmark has no column -> generic form
    if (c->h_pose_col.size() != c->ids.size()) {
      c->h_pose_col.resize(c->ids.size());
      HIPCHK(c, hipMemcpy(c->h_pose_col.data(), c->d_pose_col.p, sizeof(int) * c->h_pose_col.size(), hipMemcpyDeviceToHost));
    }
    bool lm = false;
    for (int64_t q = 0; q < n; ++q) lm = lm || c->h_pose_col[idx[q]] >= c->plan.nb;
    if (lm) { ba_off(c); rc = ensure_ready(c); if (rc) return rc; }
  }
  hipStream_t s = c->stream;
  if (!c->lin_valid) { rc = linearize_current(c, false); if (rc) return rc; c->cov_factor_valid = false; }
  if (!c->cov_factor_valid) {
    c->h_scal[3] = 0.0;
    HIPCHK(c, hipMemcpyAsync(c->d_scal.p + 3, c->h_scal + 3, sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(c->d_fail.p, 0, sizeof(int), s));
    c->isam_L_valid = false;
    ctx_factor(c, c->cur, false);
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
  ba_off(c);                                            // delta_out covers every free variable
  int rc = ensure_ready(c);
  if (rc) return rc;
  if (!c->lin_valid) { rc = linearize_current(c, false); if (rc) return rc; }
  hipStream_t s = c->stream;
  c->h_scal[3] = lambda;
  HIPCHK(c, hipMemcpyAsync(c->d_scal.p + 3, c->h_scal + 3, sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemsetAsync(c->d_fail.p, 0, sizeof(int), s));
  c->cov_factor_valid = false;
  c->isam_L_valid = false;
  launch_factor(c->plan, c->sched, c->d_H[c->cur].p, c->d_L.p, c->d_scal.p + 3, c->d_fail.p, s);
  launch_solve(c->plan, c->sched, c->d_L.p, c->d_b[c->cur].p, c->d_x.p, s);
  const int nb = c->plan.nb;
  std::vector<double> x((size_t)nb * 6);
  HIPCHK(c, hipMemcpyAsync(x.data(), c->d_x.p, sizeof(double) * x.size(), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(c->h_fail, c->d_fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipGetLastError());
  for (int k = 0; k < nb; ++k)                                     // phantom slots (the last hessian indices) are not reported
    if (c->S.perm[k] < nb - c->n_phantom) std::memcpy(delta_out + (size_t)c->S.perm[k] * 6, &x[(size_t)k * 6], 6 * sizeof(double));
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
  c->isam_L_valid = false;
  if (phase >= 1) {   // make sure lambda and (for the solve) a valid factor are in place
    c->h_scal[3] = 1e-5 * std::max(1.0, c->h_scal[2]);
    HIPCHK(c, hipMemcpyAsync(c->d_scal.p + 3, c->h_scal + 3, sizeof(double), hipMemcpyHostToDevice, s));
    ctx_factor(c, c->cur, true);
  }
  HIPCHK(c, hipStreamSynchronize(s));
  HIPCHK(c, hipEventRecord(c->ev[0], s));
  for (int r = 0; r < reps; ++r) {
    if (phase == 0) ctx_linearize(c, c->cur, c->d_scal.p + 0);
    else if (phase == 1) ctx_factor(c, c->cur, true);
    else ctx_solve(c, c->cur, true);   // what a trial runs: backward sweep only
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
