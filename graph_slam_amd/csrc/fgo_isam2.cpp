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
  // the growth reserve belongs to the incremental driving mode: a context that leaves it (delete isam2) goes back to a
  // structure without phantom slots at its next use; the next fgo_isam2_update lays a fresh reserve down
  c->isam_incremental = false;
  if (c->n_phantom > 0 || c->inc.valid) { c->inc.valid = false; c->structure_dirty = true; }
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
