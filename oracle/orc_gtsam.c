/* ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (GTSAM 4.0 un-vendored; see orc_pose3.h).
 *
 * GTSAM-semantics batch optimisation the reference reaches through CGraphGT:
 *   CGraphGT::firstNode          gtsam/gtsam_graph.cpp:320-368   PriorFactor<Pose3>, Diagonal::Sigmas(1e-7 x 6)
 *   CGraphGT::addToGTSAM(mr,..)  gtsam/gtsam_graph.cpp:630-695   BetweenFactor<Pose3>(X1,X2,inc, Gaussian::Information)
 *   CGraphGT::optimizeGraphBatch gtsam/gtsam_graph.cpp:1784-1788 LevenbergMarquardtOptimizer(graph, values).optimize()
 *   CGraphGT::error              gtsam/gtsam_graph.cpp:173-176   0.5 * sum ||whitened r||^2
 * LM restated from GTSAM 4.0's LevenbergMarquardtOptimizer with default LevenbergMarquardtParams
 * (SURVEY.md Appendix A.2): lambda0 1e-5, factor 10 (fixed), lambdaUpper 1e5, lambdaLower 0, identity damping,
 * minModelFidelity 1e-3, maxIterations 100, relativeErrorTol 1e-5, absoluteErrorTol 1e-5, errorTol 0.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"
#include "orc_plane.h"
#include "orc_camera.h"
#include "orc_imu.h"

void orc_set_gtsam(orc_problem *p) {
  p->manifold = 1;
  free(p->kind);
  p->kind = (int *)malloc(sizeof(int) * (p->E ? p->E : 1));
  for (int k = 0; k < p->E; ++k) p->kind[k] = 1;
}

void orc_add_priors(orc_problem *p, int n, const int *ids, const double *mean7, const double *info21) {
  const int m = p->nprior + n;
  p->pv = (int *)realloc(p->pv, sizeof(int) * (m ? m : 1));
  p->pmean = (double *)realloc(p->pmean, sizeof(double) * 7 * (m ? m : 1));
  p->pinfo = (double *)realloc(p->pinfo, sizeof(double) * 21 * (m ? m : 1));
  memcpy(p->pv + p->nprior, ids, sizeof(int) * n);
  memcpy(p->pmean + 7 * p->nprior, mean7, sizeof(double) * 7 * n);
  memcpy(p->pinfo + 21 * p->nprior, info21, sizeof(double) * 21 * n);
  p->nprior = m;
}

double orc_error_gtsam(const orc_problem *p) { return 0.5 * orc_chi2(p); }

void orc_between_eval(const double *xi, const double *xj, const double *z, double *e, double *Ji, double *Jj) {
  orc_between_pose3(xi, xj, z, e, Ji, Jj);
}
void orc_prior_eval(const double *x, const double *prior, double *e, double *J) { orc_prior_pose3(x, prior, e, J); }
void orc_pose3_retract_eval(const double *x, const double *xi, double *out) { orc_pose3_retract(x, xi, out); }
void orc_pose3_logmap_eval(const double *T, double *xi) { orc_se3_log(T, xi); }
void orc_pose3_expmap_eval(const double *xi, double *T) { orc_se3_exp(xi, T); }

/* OrientedPlane3 (orc_plane.h) */
void orc_plane_make(const double *abcd, double *p) { orc_plane_normalize(abcd, p); }
void orc_plane_transform_eval(const double *p, const double *x, double *out, double *Hpose, double *Hplane) {
  orc_plane_transform(p, x, out, Hpose, Hplane);
}
void orc_plane_retract_eval(const double *p, const double *v, double *out) { orc_plane_retract(p, v, out); }
void orc_plane_local_eval(const double *p, const double *q, double *v) { orc_plane_local(p, q, v); }
void orc_plane_error_vector_eval(const double *p, const double *o, double *e) { orc_plane_error_vector(p, o, e); }
void orc_plane_factor_eval(const double *x, const double *plane, const double *z, double *r, double *Hpose, double *Hplane) {
  orc_plane_factor(x, plane, z, r, Hpose, Hplane);
}

void orc_reproj_eval(const double *x, const double *pw, const double *uv, const double *calib, const double *bps, double *r,
                     double *Hx, double *Hp) {
  orc_reproj(x, pw, uv, calib, bps, r, Hx, Hp);
}

/* ---- IMU preintegration + CombinedImuFactor (orc_imu.h) ---- */
int orc_preint_size(void) { return (int)sizeof(orc_preint); }
void orc_preint_run(orc_preint *m, const double *bhat6, int n, const double *acc, const double *gyro, double dt) {
  orc_imu_params P;
  orc_imu_params_vn100(&P);
  orc_preint_reset(m, bhat6);
  for (int k = 0; k < n; ++k) orc_preint_integrate(m, &P, acc + 3 * k, gyro + 3 * k, dt);
}
/* the same with explicit isotropic variances (acc, gyro, integration, bias acc, bias gyro, biasAccOmegaInt): other sensors than
 * the VN100 (the reference's CImuMEMS: gtsam/imu_MEMS.cpp:22-37) */
void orc_preint_run_params(orc_preint *m, const double *bhat6, int n, const double *acc, const double *gyro, double dt, const double *var6) {
  orc_imu_params P;
  orc_imu_params_vn100(&P);
  P.acc_cov = var6[0]; P.gyro_cov = var6[1]; P.integ_cov = var6[2]; P.bias_acc_cov = var6[3]; P.bias_gyro_cov = var6[4]; P.bias_acc_omega_int = var6[5];
  orc_preint_reset(m, bhat6);
  for (int k = 0; k < n; ++k) orc_preint_integrate(m, &P, acc + 3 * k, gyro + 3 * k, dt);
}
void orc_imu_factor_eval(const double *xi, const double *vi, const double *xj, const double *vj, const double *bi, const double *bj,
                         const orc_preint *m, const double *g, double *r, double *Jxi, double *Jvi, double *Jxj, double *Jvj,
                         double *Jbi, double *Jbj) {
  orc_imu_factor(xi, vi, xj, vj, bi, bj, m, g, r, Jxi, Jvi, Jxj, Jvj, Jbi, Jbj);
}
/* NavState predict(state_i, bias_i): the state that makes the R, p, v residual vanish */
void orc_imu_predict(const double *xi, const double *vi, const double *bi, const orc_preint *m, const double *g, double *xj, double *vj) {
  double dba[3], dbg[3], bo[3], qc[4], qcorr[4], t3[3], t4[3], dpc[3], dvc[3], Ri[9], Rdp[3], Rdv[3];
  for (int k = 0; k < 3; ++k) { dba[k] = bi[k] - m->bhat[k]; dbg[k] = bi[3 + k] - m->bhat[3 + k]; }
  orc_m3v(m->J_R_bg, dbg, bo); orc_so3_exp(bo, qc); orc_qmul(m->dR, qc, qcorr);
  orc_m3v(m->J_p_ba, dba, t3); orc_m3v(m->J_p_bg, dbg, t4);
  for (int k = 0; k < 3; ++k) dpc[k] = m->dp[k] + t3[k] + t4[k];
  orc_m3v(m->J_v_ba, dba, t3); orc_m3v(m->J_v_bg, dbg, t4);
  for (int k = 0; k < 3; ++k) dvc[k] = m->dv[k] + t3[k] + t4[k];
  orc_qmat(xi + 3, Ri); orc_m3v(Ri, dpc, Rdp); orc_m3v(Ri, dvc, Rdv);
  for (int k = 0; k < 3; ++k) { xj[k] = xi[k] + vi[k] * m->dt + 0.5 * g[k] * m->dt * m->dt + Rdp[k]; vj[k] = vi[k] + g[k] * m->dt + Rdv[k]; }
  orc_qmul(xi + 3, qcorr, xj + 3);
  orc_qnormalize(xj + 3);
}

/* ---- IMU factors inside a problem ---- */
void orc_add_imu_factors(orc_problem *p, int n, const int *ids6, const void *preint, const double *info225, const double *gravity3) {
  const int m = p->nimu + n;
  p->imu_ids = (int *)realloc(p->imu_ids, sizeof(int) * 6 * (m ? m : 1));
  p->imu_pre = realloc(p->imu_pre, sizeof(orc_preint) * (m ? m : 1));
  p->imu_info = (double *)realloc(p->imu_info, sizeof(double) * 225 * (m ? m : 1));
  memcpy(p->imu_ids + 6 * p->nimu, ids6, sizeof(int) * 6 * n);
  memcpy((orc_preint *)p->imu_pre + p->nimu, preint, sizeof(orc_preint) * n);
  memcpy(p->imu_info + 225 * p->nimu, info225, sizeof(double) * 225 * n);
  memcpy(p->gravity, gravity3, sizeof(double) * 3);
  p->nimu = m;
  p->manifold = 1;
}

/* residual and the six Jacobians padded to 15 x 6 */
static void imu_eval(const orc_problem *p, int f, double r[15], double J[6][90]) {
  const int *id = p->imu_ids + 6 * f;
  const double *v[6];
  for (int k = 0; k < 6; ++k) v[k] = p->poses + 7 * id[k];
  double Jvi[45], Jvj[45];
  orc_imu_factor(v[0], v[1], v[2], v[3], v[4], v[5], (const orc_preint *)p->imu_pre + f, p->gravity, r,
                 J ? J[0] : 0, J ? Jvi : 0, J ? J[2] : 0, J ? Jvj : 0, J ? J[4] : 0, J ? J[5] : 0);
  if (J) {
    memset(J[1], 0, sizeof(double) * 90); memset(J[3], 0, sizeof(double) * 90);
    for (int a = 0; a < 15; ++a) for (int c = 0; c < 3; ++c) { J[1][a * 6 + c] = Jvi[a * 3 + c]; J[3][a * 6 + c] = Jvj[a * 3 + c]; }
  }
}

double orc_imu_chi2(const orc_problem *p) {
  double chi = 0;
  for (int f = 0; f < p->nimu; ++f) {
    double r[15];
    imu_eval(p, f, r, 0);
    const double *W = p->imu_info + 225 * f;
    for (int a = 0; a < 15; ++a) for (int b = 0; b < 15; ++b) chi += r[a] * W[a * 15 + b] * r[b];
  }
  return chi;
}

double orc_imu_linearize(orc_problem *p) {
  double chi = 0;
  for (int f = 0; f < p->nimu; ++f) {
    double r[15], J[6][90], WJ[6][90], Wr[15];
    imu_eval(p, f, r, J);
    const double *W = p->imu_info + 225 * f;
    for (int a = 0; a < 15; ++a) { double t = 0; for (int b = 0; b < 15; ++b) t += W[a * 15 + b] * r[b]; Wr[a] = t; chi += r[a] * t; }
    for (int u = 0; u < 6; ++u)
      for (int a = 0; a < 15; ++a) for (int c = 0; c < 6; ++c) { double t = 0; for (int b = 0; b < 15; ++b) t += W[a * 15 + b] * J[u][b * 6 + c]; WJ[u][a * 6 + c] = t; }
    const int *id = p->imu_ids + 6 * f;
    int q = 0;
    for (int u = 0; u < 6; ++u) {
      const int au = p->hidx[id[u]];
      if (au >= 0) {
        for (int rr = 0; rr < 6; ++rr) {
          for (int c = 0; c < 6; ++c) { double t = 0; for (int a = 0; a < 15; ++a) t += J[u][a * 6 + rr] * WJ[u][a * 6 + c]; p->Hd[36 * au + rr * 6 + c] += t; }
          double t = 0; for (int a = 0; a < 15; ++a) t += J[u][a * 6 + rr] * Wr[a];
          p->b[6 * au + rr] -= t;
        }
      }
      for (int w = u + 1; w < 6; ++w, ++q) {
        const int blk = p->imu_blk[15 * f + q];
        if (blk < 0) continue;
        const int aw = p->hidx[id[w]];
        double *B = p->Ho + 36 * blk;                      /* rows = blk_r (smaller hessian index) */
        for (int rr = 0; rr < 6; ++rr)
          for (int c = 0; c < 6; ++c) {
            double t = 0;
            for (int a = 0; a < 15; ++a) t += J[u][a * 6 + rr] * WJ[w][a * 6 + c];     /* (J_u^T W J_w)[rr][c] */
            if (au < aw) B[rr * 6 + c] += t; else B[c * 6 + rr] += t;
          }
      }
    }
  }
  return chi;
}

/* ---- mixed variable / factor kinds (see orc_internal.h) ---- */
void orc_set_var_kinds(orc_problem *p, const int *vkind) {
  free(p->vkind);
  p->vkind = (int *)malloc(sizeof(int) * (p->N ? p->N : 1));
  memcpy(p->vkind, vkind, sizeof(int) * p->N);
  p->manifold = 1;
}
void orc_set_edge_kinds(orc_problem *p, const int *kind) {
  free(p->kind);
  p->kind = (int *)malloc(sizeof(int) * (p->E ? p->E : 1));
  memcpy(p->kind, kind, sizeof(int) * p->E);
  p->manifold = 1;
}
void orc_set_calibration(orc_problem *p, const double *calib9, const double *body_P_sensor7) {
  memcpy(p->calib, calib9, sizeof(double) * 9);
  memcpy(p->body_P_sensor, body_P_sensor7, sizeof(double) * 7);
}

int orc_var_dim(int vkind) { return (vkind == 0 || vkind == 4) ? 6 : 3; }

void orc_var_retract(int vkind, const double *x, const double *d, double *out) {
  memcpy(out, x, 7 * sizeof(double));
  if (vkind == 1) { double pl[4]; orc_plane_retract(x, d, pl); memcpy(out, pl, sizeof(pl)); }
  else if (vkind == 2 || vkind == 3) { out[0] = x[0] + d[0]; out[1] = x[1] + d[1]; out[2] = x[2] + d[2]; }
  else if (vkind == 4) { for (int k = 0; k < 6; ++k) out[k] = x[k] + d[k]; }
  else orc_pose3_retract(x, d, out);
}

void orc_factor_eval(const orc_problem *p, int k, double e[6], double *Ji, double *Jj, double W[36]) {
  const int kind = p->kind ? p->kind[k] : 0;
  const double *xi = p->poses + 7 * p->ei[k], *xj = p->poses + 7 * p->ej[k], *z = p->meas + 7 * k, *om = p->info + 21 * k;
  if (kind == 0) { orc_edge_se3(xi, xj, z, e, Ji, Jj); orc_info_full(om, W); return; }
  if (kind == 1) { orc_between_pose3(xi, xj, z, e, Ji, Jj); orc_info_full(om, W); return; }
  memset(e, 0, 6 * sizeof(double));
  memset(W, 0, 36 * sizeof(double));
  if (Ji) memset(Ji, 0, 36 * sizeof(double));
  if (Jj) memset(Jj, 0, 36 * sizeof(double));
  if (kind == 2) {            /* OrientedPlane3Factor(pose, plane): 3 rows */
    double r[3], Hx[18], Hp[9];
    orc_plane_factor(xi, xj, z, r, Ji ? Hx : 0, Jj ? Hp : 0);
    memcpy(e, r, sizeof(r));
    if (Ji) for (int a = 0; a < 3; ++a) for (int c = 0; c < 6; ++c) Ji[a * 6 + c] = Hx[a * 6 + c];
    if (Jj) for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) Jj[a * 6 + c] = Hp[a * 3 + c];
    int q = 0;
    for (int a = 0; a < 3; ++a) for (int c = a; c < 3; ++c) { W[a * 6 + c] = om[q]; W[c * 6 + a] = om[q]; ++q; }
  } else {                    /* reprojection(pose, point): 2 rows, isotropic */
    double r[2], Hx[12], Hp[6];
    orc_reproj(xi, xj, z, p->calib, p->body_P_sensor, r, Ji ? Hx : 0, Jj ? Hp : 0);
    e[0] = r[0]; e[1] = r[1];
    if (Ji) for (int a = 0; a < 2; ++a) for (int c = 0; c < 6; ++c) Ji[a * 6 + c] = Hx[a * 6 + c];
    if (Jj) for (int a = 0; a < 2; ++a) for (int c = 0; c < 3; ++c) Jj[a * 6 + c] = Hp[a * 3 + c];
    W[0] = om[0]; W[7] = om[0];
  }
}

/* PriorFactor<T>: Pose3 -> Logmap chart; vector-valued variables (Point3 gtsam_graph.cpp:379,394; velocity and bias
 * :359-367) -> x - mean with identity Jacobian on the variable's dimension */
void orc_prior_dispatch(const orc_problem *p, int k, double e[6], double *J) {
  const int v = p->pv[k], vk = p->vkind ? p->vkind[v] : 0;
  if (vk == 0) { orc_prior_pose3(p->poses + 7 * v, p->pmean + 7 * k, e, J); return; }
  const int dim = orc_var_dim(vk);
  memset(e, 0, 6 * sizeof(double));
  if (J) memset(J, 0, 36 * sizeof(double));
  for (int r = 0; r < dim; ++r) { e[r] = p->poses[7 * v + r] - p->pmean[7 * k + r]; if (J) J[r * 6 + r] = 1.0; }
}

/* GTSAM LevenbergMarquardtOptimizer::optimize() with default parameters.  Returns the number of iterations. */
int orc_optimize_gtsam(orc_problem *p, int max_iterations, orc_stats *st) {
  orc_stats s;
  memset(&s, 0, sizeof(s));
  const double tstart = orc_now_s();
  if (p->nfree == 0 || (p->E == 0 && p->nprior == 0)) { if (st) *st = s; return -1; }
  if (!p->built) { orc_build_structure(p); s.t_symbolic = p->t_symbolic; }
  const int n = p->nfree;
  const double lambdaFactor = 10.0, lambdaUpper = 1e5, lambdaLower = 0.0, minModelFidelity = 1e-3;
  const double relTol = 1e-5, absTol = 1e-5, errTol = 0.0;
  double lambda = 1e-5;
  p->ntrace = 0;
  double currentError = 0.5 * orc_chi2(p);
  s.chi2_initial = 2 * currentError;
  int iterations = 0;
  if (max_iterations <= 0) max_iterations = 100;
  while (1) {
    const double errorBefore = currentError;
    /* ---- iterate(): linearise once, search lambda */
    double t0 = orc_now_s();
    orc_linearize(p);
    s.t_linearize += orc_now_s() - t0;
    while (1) {
      memcpy(p->backup, p->poses, sizeof(double) * 7 * p->N);
      const int bad = orc_solve(p, lambda, &s.t_factor, &s.t_solve);
      ++s.trials;
      int step_ok = 0, stop_search = 0;
      double modelFidelity = 0, newError = currentError;
      if (!bad) {
        /* linearised cost change: 0.5 d'(H d) - ... ; with g = -b:  L(0) - L(d) = b'd - 0.5 d'H d  (undamped H) */
        double bd = 0, dHd = 0;
        /* H d through the block storage */
        double *Hx = (double *)calloc(6 * (size_t)n, sizeof(double));
        for (int a = 0; a < n; ++a)
          for (int r = 0; r < 6; ++r) { double t = 0; for (int c = 0; c < 6; ++c) t += p->Hd[36 * a + r * 6 + c] * p->x[6 * a + c]; Hx[6 * a + r] += t; }
        for (int t = 0; t < p->nblk; ++t) {
          const int a = p->blk_r[t], b = p->blk_c[t];
          const double *B = p->Ho + 36 * t;
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) { Hx[6 * a + r] += B[r * 6 + c] * p->x[6 * b + c]; Hx[6 * b + c] += B[r * 6 + c] * p->x[6 * a + r]; }
        }
        for (int k = 0; k < 6 * n; ++k) { bd += p->b[k] * p->x[k]; dHd += p->x[k] * Hx[k]; }
        free(Hx);
        const double linearizedCostChange = bd - 0.5 * dHd;
        t0 = orc_now_s();
        orc_apply_update(p);
        newError = 0.5 * orc_chi2(p);
        s.t_update += orc_now_s() - t0;
        if (linearizedCostChange >= 0) {
          const double costChange = currentError - newError;
          if (linearizedCostChange > 1e-20) {
            modelFidelity = costChange / linearizedCostChange;
            if (modelFidelity > minModelFidelity) step_ok = 1;
          }
          if (fabs(costChange) < relTol * currentError) stop_search = 1;
        }
      }
      if (step_ok) {
        currentError = newError;
        lambda /= lambdaFactor;                 /* decreaseLambda, fixed factor */
        if (lambda < lambdaLower) lambda = lambdaLower;
        break;
      }
      memcpy(p->poses, p->backup, sizeof(double) * 7 * p->N);
      if (stop_search) break;
      lambda *= lambdaFactor;                   /* increaseLambda */
      if (lambda >= lambdaUpper) break;
    }
    ++iterations;
    if (p->ntrace < 256) { p->tr_chi2[p->ntrace] = 2 * currentError; p->tr_lambda[p->ntrace] = lambda; ++p->ntrace; }
    /* ---- checkConvergence */
    if (iterations >= max_iterations) break;
    if (!isfinite(currentError)) break;
    if (currentError <= errTol) break;
    const double absDec = errorBefore - currentError, relDec = absDec / errorBefore;
    if (relDec <= relTol || absDec <= absTol) break;
  }
  s.iterations = iterations; s.chi2_final = 2 * currentError; s.lambda_final = lambda;
  s.nnz_H_blocks = (long long)p->nblk + n;
  s.nnz_L_scalar = orc_chol_nnz(p->chol);
  s.t_total = orc_now_s() - tstart;
  if (st) *st = s;
  return iterations;
}

/* ISAM2::update + calculateEstimate as CGraphGT::optimizeGraphIncremental issues them (gtsam/gtsam_graph.cpp:1768-1776,
 * ISAM2Params :93-99: relinearizeThreshold, relinearizeSkip = 1, Gauss-Newton).  State = ISAM2's linearisation point
 * theta (7 per variable) and linear solution delta (6 per variable, variable order), kept by the caller so that a grown
 * graph (a new orc_problem) can carry it over; new variables enter with theta = initial value, delta = 0.
 *   1. variables with max|delta_k| >= threshold: theta <- theta (+) delta, delta <- 0   (CheckRelinearizationFull + ExpmapMasked)
 *   2. linearise at theta, solve H delta = b undamped                                   (re-elimination + back-substitution,
 *                                                                                        wildfireThreshold -> 0)
 *   3. estimate = theta (+) delta                                                       (calculateEstimate)
 * p->poses is left at the estimate.  Returns 0, or the Cholesky failure code; *n_relin = variables moved in step 1. */
int orc_isam2_step(orc_problem *p, double threshold, double *theta7, double *delta6, double *est7, int *n_relin) {
  if (p->nfree == 0 || (p->E == 0 && p->nprior == 0)) return -1;
  if (!p->built) orc_build_structure(p);
  int moved = 0;
  for (int v = 0; v < p->N; ++v) {
    if (p->hidx[v] < 0) continue;
    double mx = 0;
    for (int k = 0; k < 6; ++k) { const double a = fabs(delta6[6 * v + k]); if (a > mx) mx = a; }
    if (mx >= threshold) {
      double out[7] = {0, 0, 0, 0, 0, 0, 0};
      if (p->vkind && p->vkind[v]) orc_var_retract(p->vkind[v], theta7 + 7 * v, delta6 + 6 * v, out);
      else orc_pose3_retract(theta7 + 7 * v, delta6 + 6 * v, out);
      memcpy(theta7 + 7 * v, out, sizeof(out));
      memset(delta6 + 6 * v, 0, 6 * sizeof(double));
      ++moved;
    }
  }
  if (n_relin) *n_relin = moved;
  memcpy(p->poses, theta7, sizeof(double) * 7 * p->N);
  orc_linearize(p);
  const int bad = orc_solve(p, 0.0, 0, 0);
  if (bad) return bad;
  for (int v = 0; v < p->N; ++v)
    if (p->hidx[v] >= 0) memcpy(delta6 + 6 * v, p->x + 6 * p->hidx[v], 6 * sizeof(double));
  orc_apply_update(p);
  if (est7) memcpy(est7, p->poses, sizeof(double) * 7 * p->N);
  return 0;
}
