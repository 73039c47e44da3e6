/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_se3.h header).  PARITY UNPINNED (g2o un-vendored).
 *
 * CPU restatement of the batch pose-graph optimisation the reference delegates to g2o:
 *   CGraphG2O::createOptimizer   g2o/g2o_graph.cpp:65-77   LM over BlockSolver<6,3> over CSparse
 *   CGraphG2O::firstNode         g2o/g2o_graph.cpp:80-94   vertex 0 fixed (excluded from H)
 *   CGraphG2O::addToGraph        g2o/g2o_graph.cpp:96-134  VertexSE3 / EdgeSE3(meas, information)
 *   CGraphG2O::optimizeGraph     g2o/g2o_graph.cpp:241-252 10 x optimize(2)
 *   CGraphG2O::error             g2o/g2o_graph.cpp:254-258 chi2 = sum e' Omega e
 * One orc_optimize(p, n) call == one SparseOptimizer::optimize(n) call [UPSTREAM semantics,
 * SURVEY.md Appendix A.1]: structure (re)built, lambda re-initialised to 1e-5 * max diag(H) at
 * iteration 0, g2o's rho / nu update, <= 10 trials per iteration.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "orc_api.h"
#include "orc_pose3.h"

double orc_now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

#include "orc_internal.h"

void orc_edge_se3_eval(const double *xi, const double *xj, const double *z, double *e, double *Ji,
                       double *Jj) {
  orc_edge_se3(xi, xj, z, e, Ji, Jj);
}
void orc_pose_oplus_eval(const double *x, const double *d, double *out) { orc_pose_oplus(x, d, out); }

orc_problem *orc_create(int N, const double *poses7, const unsigned char *fixed, int E, const int *ei,
                        const int *ej, const double *meas7, const double *info21) {
  orc_problem *p = (orc_problem *)calloc(1, sizeof(orc_problem));
  p->N = N; p->E = E;
  p->poses = (double *)malloc(sizeof(double) * 7 * N);
  p->backup = (double *)malloc(sizeof(double) * 7 * N);
  memcpy(p->poses, poses7, sizeof(double) * 7 * N);
  p->fixed = (unsigned char *)malloc(N ? N : 1);
  memcpy(p->fixed, fixed, N);
  p->ei = (int *)malloc(sizeof(int) * (E ? E : 1)); memcpy(p->ei, ei, sizeof(int) * E);
  p->ej = (int *)malloc(sizeof(int) * (E ? E : 1)); memcpy(p->ej, ej, sizeof(int) * E);
  p->meas = (double *)malloc(sizeof(double) * 7 * (E ? E : 1)); memcpy(p->meas, meas7, sizeof(double) * 7 * E);
  p->info = (double *)malloc(sizeof(double) * 21 * (E ? E : 1)); memcpy(p->info, info21, sizeof(double) * 21 * E);
  p->hidx = (int *)malloc(sizeof(int) * (N ? N : 1));
  int k = 0;
  for (int v = 0; v < N; ++v) p->hidx[v] = fixed[v] ? -1 : k++;
  p->nfree = k;
  return p;
}

void orc_free(orc_problem *p) {
  if (!p) return;
  free(p->poses); free(p->backup); free(p->fixed); free(p->ei); free(p->ej); free(p->meas);
  free(p->info); free(p->hidx); free(p->kind); free(p->vkind); free(p->imu_ids); free(p->imu_pre); free(p->imu_info); free(p->imu_blk); free(p->pv); free(p->pmean); free(p->pinfo); free(p->blk_r); free(p->blk_c); free(p->edge_blk); free(p->Hd);
  free(p->Ho); free(p->b); free(p->perm); free(p->iperm); free(p->Cp); free(p->Ci); free(p->Cx);
  free(p->colbase); free(p->colm); free(p->blk_rank); free(p->blk_pc); free(p->blk_tr);
  orc_chol_free(p->chol); orc_sn_free(p->sn); free(p->x); free(p->xp); free(p);
}

void orc_get_poses(const orc_problem *p, double *out) { memcpy(out, p->poses, sizeof(double) * 7 * p->N); }
void orc_set_poses(orc_problem *p, const double *in) { memcpy(p->poses, in, sizeof(double) * 7 * p->N); }

double orc_chi2(const orc_problem *p) {
  double chi = 0;
  for (int k = 0; k < p->E; ++k) {
    double e[6], W[36];
    orc_factor_eval(p, k, e, 0, 0, W);
    double c = 0;
    for (int r = 0; r < 6; ++r) {
      double t = 0;
      for (int q = 0; q < 6; ++q) t += W[r * 6 + q] * e[q];
      c += e[r] * t;
    }
    chi += c;
  }
  for (int k = 0; k < p->nprior; ++k) {
    double e[6], W[36];
    orc_prior_dispatch(p, k, e, 0);
    orc_info_full(p->pinfo + 21 * k, W);
    for (int r = 0; r < 6; ++r) for (int q = 0; q < 6; ++q) chi += e[r] * W[r * 6 + q] * e[q];
  }
  chi += orc_imu_chi2(p);
  return chi;
}

typedef struct { int r, c, e; } pairrec;
static int pair_cmp(const void *a, const void *b) {
  const pairrec *x = (const pairrec *)a, *y = (const pairrec *)b;
  if (x->c != y->c) return x->c < y->c ? -1 : 1;
  if (x->r != y->r) return x->r < y->r ? -1 : 1;
  return 0;
}

void orc_build_structure(orc_problem *p) {
  const double t0 = orc_now_s();
  const int n = p->nfree;
  /* unique off-diagonal blocks */
  pairrec *pr = (pairrec *)malloc(sizeof(pairrec) * (p->E + 15 * p->nimu + 1));
  int m = 0;
  p->edge_blk = (int *)malloc(sizeof(int) * (p->E ? p->E : 1));
  for (int k = 0; k < p->E; ++k) {
    int a = p->hidx[p->ei[k]], b = p->hidx[p->ej[k]];
    p->edge_blk[k] = -1;
    if (a < 0 || b < 0 || a == b) continue;
    pr[m].r = a < b ? a : b; pr[m].c = a < b ? b : a; pr[m].e = k; ++m;
  }
  /* the 6-variable IMU factors contribute all 15 variable pairs; encoded as e = -1 - (15 f + pair) */
  p->imu_blk = (int *)malloc(sizeof(int) * (15 * p->nimu + 1));
  for (int f = 0; f < p->nimu; ++f) {
    int q = 0;
    for (int u = 0; u < 6; ++u)
      for (int w = u + 1; w < 6; ++w, ++q) {
        int a = p->hidx[p->imu_ids[6 * f + u]], b = p->hidx[p->imu_ids[6 * f + w]];
        p->imu_blk[15 * f + q] = -1;
        if (a < 0 || b < 0 || a == b) continue;
        pr[m].r = a < b ? a : b; pr[m].c = a < b ? b : a; pr[m].e = -1 - (15 * f + q); ++m;
      }
  }
  qsort(pr, m, sizeof(pairrec), pair_cmp);
  p->blk_r = (int *)malloc(sizeof(int) * (m ? m : 1));
  p->blk_c = (int *)malloc(sizeof(int) * (m ? m : 1));
  int nb = 0;
  for (int t = 0; t < m; ++t) {
    if (t == 0 || pr[t].r != pr[t - 1].r || pr[t].c != pr[t - 1].c) {
      p->blk_r[nb] = pr[t].r; p->blk_c[nb] = pr[t].c; ++nb;
    }
    if (pr[t].e >= 0) p->edge_blk[pr[t].e] = nb - 1;
    else p->imu_blk[-1 - pr[t].e] = nb - 1;
  }
  free(pr);
  p->nblk = nb;
  p->Hd = (double *)malloc(sizeof(double) * 36 * (n ? n : 1));
  p->Ho = (double *)malloc(sizeof(double) * 36 * (nb ? nb : 1));
  p->b = (double *)malloc(sizeof(double) * 6 * (n ? n : 1));
  p->x = (double *)calloc(6 * (n ? n : 1), sizeof(double));
  p->xp = (double *)calloc(6 * (n ? n : 1), sizeof(double));
  /* block adjacency -> AMD */
  int *xadj = (int *)calloc(n + 1, sizeof(int));
  for (int t = 0; t < nb; ++t) { xadj[p->blk_r[t] + 1]++; xadj[p->blk_c[t] + 1]++; }
  for (int i = 0; i < n; ++i) xadj[i + 1] += xadj[i];
  int *adj = (int *)malloc(sizeof(int) * (2 * nb ? 2 * nb : 1));
  int *fill = (int *)malloc(sizeof(int) * (n ? n : 1));
  memcpy(fill, xadj, sizeof(int) * n);
  for (int t = 0; t < nb; ++t) { adj[fill[p->blk_r[t]]++] = p->blk_c[t]; adj[fill[p->blk_c[t]]++] = p->blk_r[t]; }
  free(fill);
  p->perm = (int *)malloc(sizeof(int) * (n ? n : 1));
  p->iperm = (int *)malloc(sizeof(int) * (n ? n : 1));
  orc_amd_order(n, xadj, adj, p->perm);
  for (int k = 0; k < n; ++k) p->iperm[p->perm[k]] = k;
  free(xadj); free(adj);
  /* permuted block columns */
  p->colm = (int *)calloc(n ? n : 1, sizeof(int));
  p->blk_pc = (int *)malloc(sizeof(int) * (nb ? nb : 1));
  p->blk_rank = (int *)malloc(sizeof(int) * (nb ? nb : 1));
  p->blk_tr = (unsigned char *)malloc(nb ? nb : 1);
  int *prow = (int *)malloc(sizeof(int) * (nb ? nb : 1));
  for (int t = 0; t < nb; ++t) {
    int pa = p->iperm[p->blk_r[t]], pb = p->iperm[p->blk_c[t]];
    if (pa < pb) { p->blk_pc[t] = pb; prow[t] = pa; p->blk_tr[t] = 0; }
    else { p->blk_pc[t] = pa; prow[t] = pb; p->blk_tr[t] = 1; }
    p->blk_rank[t] = p->colm[p->blk_pc[t]]++;
  }
  p->colbase = (long long *)malloc(sizeof(long long) * (n + 1));
  p->colbase[0] = 0;
  for (int c = 0; c < n; ++c) p->colbase[c + 1] = p->colbase[c] + 36LL * p->colm[c] + 21;
  const long long nnz = p->colbase[n];
  p->Cp = (int *)malloc(sizeof(int) * (6 * n + 1));
  p->Ci = (int *)malloc(sizeof(int) * (size_t)(nnz ? nnz : 1));
  p->Cx = (double *)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1));
  for (int c = 0; c < n; ++c)
    for (int s = 0; s < 6; ++s) {
      long long off = p->colbase[c] + (long long)s * 6 * p->colm[c] + s * (s + 1) / 2;
      p->Cp[6 * c + s] = (int)off;
      for (int r = 0; r <= s; ++r) p->Ci[off + 6 * p->colm[c] + r] = 6 * c + r;
    }
  p->Cp[6 * n] = (int)nnz;
  for (int t = 0; t < nb; ++t) {
    const int c = p->blk_pc[t];
    for (int s = 0; s < 6; ++s) {
      long long off = p->colbase[c] + (long long)s * 6 * p->colm[c] + s * (s + 1) / 2 + 6 * p->blk_rank[t];
      for (int r = 0; r < 6; ++r) p->Ci[off + r] = 6 * prow[t] + r;
    }
  }
  free(prow);
  p->chol = orc_chol_symbolic(6 * n, p->Cp, p->Ci);
  p->built = 1;
  p->t_symbolic = orc_now_s() - t0;
}

/* J' W K for row-major 6x6 (out += ) */
static void jtwk_add(const double *J, const double *W, const double *K, double *out) {
  double WK[36];
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) {
      double a = 0;
      for (int k = 0; k < 6; ++k) a += W[r * 6 + k] * K[k * 6 + c];
      WK[r * 6 + c] = a;
    }
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) {
      double a = 0;
      for (int k = 0; k < 6; ++k) a += J[k * 6 + r] * WK[k * 6 + c];
      out[r * 6 + c] += a;
    }
}

/* multi-threaded leg: the binary factors in parallel, every contribution computed once into per-edge storage, then each
 * vertex sums its incident edges (gather, no atomics: the sums do not depend on the thread count) */
static double orc_linearize_mt(orc_problem *p, int nth) {
  const int E = p->E;
  memset(p->Ho, 0, sizeof(double) * 36 * p->nblk);
  int *ptr = (int *)calloc((size_t)p->N + 1, sizeof(int));
  for (int k = 0; k < E; ++k) { ptr[p->ei[k] + 1]++; ptr[p->ej[k] + 1]++; }
  for (int v = 0; v < p->N; ++v) ptr[v + 1] += ptr[v];
  int *inc = (int *)malloc(sizeof(int) * 2 * (size_t)(E ? E : 1)), *fillp = (int *)malloc(sizeof(int) * ((size_t)p->N + 1));
  memcpy(fillp, ptr, sizeof(int) * ((size_t)p->N + 1));
  for (int k = 0; k < E; ++k) { inc[fillp[p->ei[k]]++] = 2 * k; inc[fillp[p->ej[k]]++] = 2 * k + 1; }
  double chi = 0;
  /* one vertex at a time: its incident factors are evaluated and its own side summed in list order; the j side of an
   * edge also owns the off-diagonal block (atomic only because duplicate edges share a block) and the chi2 term */
#pragma omp parallel for num_threads(nth) schedule(dynamic, 256) reduction(+ : chi)
  for (int v = 0; v < p->N; ++v) {
    const int a = p->hidx[v];
    double D[36], g[6];
    memset(D, 0, sizeof(D)); memset(g, 0, sizeof(g));
    for (int q = ptr[v]; q < ptr[v + 1]; ++q) {
      const int k = inc[q] >> 1, side = inc[q] & 1;
      double e[6], Ji[36], Jj[36], W[36], We[6];
      orc_factor_eval(p, k, e, Ji, Jj, W);
      double c = 0;
      for (int r = 0; r < 6; ++r) { double t = 0; for (int u = 0; u < 6; ++u) t += W[r * 6 + u] * e[u]; We[r] = t; c += e[r] * t; }
      const double *J = side ? Jj : Ji;
      if (a >= 0) {
        jtwk_add(J, W, J, D);
        for (int r = 0; r < 6; ++r) { double t = 0; for (int u = 0; u < 6; ++u) t += J[u * 6 + r] * We[u]; g[r] -= t; }
      }
      if (side) {
        chi += c;
        const int ai = p->hidx[p->ei[k]], bj = p->hidx[p->ej[k]];
        if (ai >= 0 && bj >= 0 && ai != bj) {
          double tmp[36];
          memset(tmp, 0, sizeof(tmp));
          if (ai < bj) jtwk_add(Ji, W, Jj, tmp); else jtwk_add(Jj, W, Ji, tmp);
          double *blk = p->Ho + 36 * p->edge_blk[k];
          for (int u = 0; u < 36; ++u) {
#pragma omp atomic
            blk[u] += tmp[u];
          }
        }
      }
    }
    if (a >= 0) { memcpy(p->Hd + 36 * a, D, sizeof(D)); memcpy(p->b + 6 * a, g, sizeof(g)); }
  }
  free(ptr); free(inc); free(fillp);
  for (int k = 0; k < p->nprior; ++k) {
    double e[6], J[36], W[36], We[6];
    orc_prior_dispatch(p, k, e, J);
    orc_info_full(p->pinfo + 21 * k, W);
    for (int r = 0; r < 6; ++r) { double t = 0; for (int q = 0; q < 6; ++q) t += W[r * 6 + q] * e[q]; We[r] = t; chi += e[r] * t; }
    const int a = p->hidx[p->pv[k]];
    if (a < 0) continue;
    jtwk_add(J, W, J, p->Hd + 36 * a);
    for (int r = 0; r < 6; ++r) { double t = 0; for (int q = 0; q < 6; ++q) t += J[q * 6 + r] * We[q]; p->b[6 * a + r] -= t; }
  }
  chi += orc_imu_linearize(p);
  if (p->vkind)
    for (int v = 0; v < p->N; ++v) {
      const int a = p->hidx[v];
      if (a < 0) continue;
      for (int r = orc_var_dim(p->vkind[v]); r < 6; ++r) p->Hd[36 * a + 7 * r] += 1.0;
    }
  return chi;
}

/* computeActiveErrors + buildSystem: returns chi2 at the linearisation point */
double orc_linearize(orc_problem *p) {
  const int n = p->nfree;
  memset(p->Hd, 0, sizeof(double) * 36 * n);
  memset(p->Ho, 0, sizeof(double) * 36 * p->nblk);
  memset(p->b, 0, sizeof(double) * 6 * n);
  double chi = 0;
  const int nth = orc_get_threads();
  if (nth > 1) return orc_linearize_mt(p, nth);
  for (int k = 0; k < p->E; ++k) {
    double e[6], Ji[36], Jj[36], W[36], We[6];
    const int vi = p->ei[k], vj = p->ej[k];
    orc_factor_eval(p, k, e, Ji, Jj, W);
    double c = 0;
    for (int r = 0; r < 6; ++r) {
      double t = 0;
      for (int q = 0; q < 6; ++q) t += W[r * 6 + q] * e[q];
      We[r] = t; c += e[r] * t;
    }
    chi += c;
    const int a = p->hidx[vi], b = p->hidx[vj];
    if (a >= 0) {
      jtwk_add(Ji, W, Ji, p->Hd + 36 * a);
      for (int r = 0; r < 6; ++r) { double t = 0; for (int q = 0; q < 6; ++q) t += Ji[q * 6 + r] * We[q]; p->b[6 * a + r] -= t; }
    }
    if (b >= 0) {
      jtwk_add(Jj, W, Jj, p->Hd + 36 * b);
      for (int r = 0; r < 6; ++r) { double t = 0; for (int q = 0; q < 6; ++q) t += Jj[q * 6 + r] * We[q]; p->b[6 * b + r] -= t; }
    }
    if (a >= 0 && b >= 0 && a != b) {
      double *blk = p->Ho + 36 * p->edge_blk[k];
      if (a < b) jtwk_add(Ji, W, Jj, blk); else jtwk_add(Jj, W, Ji, blk);
    }
  }
  for (int k = 0; k < p->nprior; ++k) {
    double e[6], J[36], W[36], We[6];
    orc_prior_dispatch(p, k, e, J);
    orc_info_full(p->pinfo + 21 * k, W);
    for (int r = 0; r < 6; ++r) { double t = 0; for (int q = 0; q < 6; ++q) t += W[r * 6 + q] * e[q]; We[r] = t; chi += e[r] * t; }
    const int a = p->hidx[p->pv[k]];
    if (a < 0) continue;
    jtwk_add(J, W, J, p->Hd + 36 * a);
    for (int r = 0; r < 6; ++r) { double t = 0; for (int q = 0; q < 6; ++q) t += J[q * 6 + r] * We[q]; p->b[6 * a + r] -= t; }
  }
  chi += orc_imu_linearize(p);
  /* variables with fewer than 6 degrees of freedom are padded to a 6-block: identity on the padding */
  if (p->vkind)
    for (int v = 0; v < p->N; ++v) {
      const int a = p->hidx[v];
      if (a < 0) continue;
      for (int r = orc_var_dim(p->vkind[v]); r < 6; ++r) p->Hd[36 * a + 7 * r] += 1.0;
    }
  return chi;
}

static void fill_csc(orc_problem *p, double lambda) {
  const int n = p->nfree;
  for (int pc = 0; pc < n; ++pc) {
    const int a = p->perm[pc];
    const double *D = p->Hd + 36 * a;
    for (int s = 0; s < 6; ++s) {
      long long off = p->colbase[pc] + (long long)s * 6 * p->colm[pc] + s * (s + 1) / 2 + 6 * p->colm[pc];
      for (int r = 0; r <= s; ++r) p->Cx[off + r] = D[r * 6 + s] + (r == s ? lambda : 0.0);
    }
  }
  for (int t = 0; t < p->nblk; ++t) {
    const int c = p->blk_pc[t];
    const double *B = p->Ho + 36 * t;
    for (int s = 0; s < 6; ++s) {
      long long off = p->colbase[c] + (long long)s * 6 * p->colm[c] + s * (s + 1) / 2 + 6 * p->blk_rank[t];
      if (!p->blk_tr[t]) for (int r = 0; r < 6; ++r) p->Cx[off + r] = B[r * 6 + s];
      else for (int r = 0; r < 6; ++r) p->Cx[off + r] = B[s * 6 + r];
    }
  }
}

/* solves (H + lambda I) x = b into p->x ; returns 0 ok */
int orc_solve(orc_problem *p, double lambda, double *t_factor, double *t_solve) {
  const int n = p->nfree;
  double t0 = orc_now_s();
  fill_csc(p, lambda);
  int rc;
  if (orc_get_solver() == 1) {
    if (!p->sn) { const double ts = orc_now_s(); p->sn = orc_sn_symbolic(6 * n, p->Cp, p->Ci); p->t_symbolic += orc_now_s() - ts; t0 = orc_now_s(); }
    rc = orc_sn_numeric(p->sn, p->Cx, orc_get_threads());
  } else
    rc = orc_get_threads() > 1 ? orc_chol_numeric_mt(p->chol, p->Cp, p->Ci, p->Cx, orc_get_threads())
                               : orc_chol_numeric(p->chol, p->Cp, p->Ci, p->Cx);
  double t1 = orc_now_s();
  if (t_factor) *t_factor += t1 - t0;
  if (rc) { memset(p->x, 0, sizeof(double) * 6 * n); return rc; }
  for (int pc = 0; pc < n; ++pc) memcpy(p->xp + 6 * pc, p->b + 6 * p->perm[pc], 6 * sizeof(double));
  if (orc_get_solver() == 1) orc_sn_solve(p->sn, p->xp); else orc_chol_solve(p->chol, p->xp);
  for (int pc = 0; pc < n; ++pc) memcpy(p->x + 6 * p->perm[pc], p->xp + 6 * pc, 6 * sizeof(double));
  if (t_solve) *t_solve += orc_now_s() - t1;
  return 0;
}

void orc_apply_update(orc_problem *p) {
  for (int v = 0; v < p->N; ++v) {
    const int a = p->hidx[v];
    if (a < 0) continue;
    double out[7];
    if (p->vkind && p->vkind[v]) orc_var_retract(p->vkind[v], p->poses + 7 * v, p->x + 6 * a, out);
    else if (p->manifold) orc_pose3_retract(p->poses + 7 * v, p->x + 6 * a, out);
    else orc_pose_oplus(p->poses + 7 * v, p->x + 6 * a, out);
    memcpy(p->poses + 7 * v, out, sizeof(out));
  }
}

int orc_optimize(orc_problem *p, int iterations, orc_stats *st) {
  orc_stats s;
  memset(&s, 0, sizeof(s));
  const double tstart = orc_now_s();
  if (p->nfree == 0 || (p->E == 0 && p->nprior == 0)) { if (st) *st = s; return -1; }
  if (!p->built) { orc_build_structure(p); s.t_symbolic = p->t_symbolic; }
  const int n = p->nfree;
  double lambda = 0, ni = 2;
  p->ntrace = 0;
  int it = 0, ok = 1;
  for (; it < iterations && ok; ++it) {
    double t0 = orc_now_s();
    double cur = orc_linearize(p);
    s.t_linearize += orc_now_s() - t0;
    if (it == 0) {
      s.chi2_initial = cur;
      double mx = 0;
      for (int a = 0; a < n; ++a) for (int r = 0; r < 6; ++r) { double d = fabs(p->Hd[36 * a + 7 * r]); if (d > mx) mx = d; }
      lambda = 1e-5 * mx; ni = 2;
    }
    double rho = 0, tmp = cur;
    int q = 0;
    do {
      memcpy(p->backup, p->poses, sizeof(double) * 7 * p->N);        /* push */
      int bad = orc_solve(p, lambda, &s.t_factor, &s.t_solve);
      ++s.trials;
      t0 = orc_now_s();
      orc_apply_update(p);
      tmp = orc_chi2(p);
      s.t_update += orc_now_s() - t0;
      if (bad) tmp = 1.7976931348623157e308;
      rho = cur - tmp;
      double scale = 0;
      for (int k = 0; k < 6 * n; ++k) scale += p->x[k] * (lambda * p->x[k] + p->b[k]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tmp)) {
        double alpha = 1. - pow(2 * rho - 1, 3);
        if (alpha > 2. / 3.) alpha = 2. / 3.;
        double sf = alpha < 1. / 3. ? 1. / 3. : alpha;
        lambda *= sf; ni = 2; cur = tmp;                              /* discardTop */
      } else {
        lambda *= ni; ni *= 2;
        memcpy(p->poses, p->backup, sizeof(double) * 7 * p->N);      /* pop */
        if (!isfinite(lambda)) break;
      }
      ++q;
    } while (rho < 0 && q < 10);
    if (p->ntrace < 256) { p->tr_chi2[p->ntrace] = cur; p->tr_lambda[p->ntrace] = lambda; ++p->ntrace; }
    s.chi2_final = cur;
    if (q == 10 || rho == 0 || !isfinite(lambda)) { ok = 0; s.terminated = 1; }
  }
  s.iterations = it; s.lambda_final = lambda;
  s.nnz_H_blocks = (long long)p->nblk + n;
  s.nnz_L_scalar = orc_chol_nnz(p->chol);
  s.t_total = orc_now_s() - tstart;
  if (st) *st = s;
  return it;
}

int orc_trace(const orc_problem *p, double *chi2s, double *lambdas, int cap) {
  int m = p->ntrace < cap ? p->ntrace : cap;
  memcpy(chi2s, p->tr_chi2, sizeof(double) * m);
  memcpy(lambdas, p->tr_lambda, sizeof(double) * m);
  return m;
}

int orc_dense_system(const orc_problem *pc, double *H, double *b, int *n_free_out) {
  orc_problem *p = (orc_problem *)pc;
  if (!p->built) orc_build_structure(p);
  const int n = p->nfree, m = 6 * n;
  orc_linearize(p);
  memset(H, 0, sizeof(double) * (size_t)m * m);
  for (int a = 0; a < n; ++a)
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H[(size_t)(6 * a + r) * m + 6 * a + c] = p->Hd[36 * a + r * 6 + c];
  for (int t = 0; t < p->nblk; ++t)
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
      double v = p->Ho[36 * t + r * 6 + c];
      H[(size_t)(6 * p->blk_r[t] + r) * m + 6 * p->blk_c[t] + c] = v;
      H[(size_t)(6 * p->blk_c[t] + c) * m + 6 * p->blk_r[t] + r] = v;
    }
  memcpy(b, p->b, sizeof(double) * m);
  if (n_free_out) *n_free_out = n;
  return 0;
}

int orc_solve_step(orc_problem *p, double lambda, double *delta) {
  if (!p->built) orc_build_structure(p);
  orc_linearize(p);
  int rc = orc_solve(p, lambda, 0, 0);
  memcpy(delta, p->x, sizeof(double) * 6 * p->nfree);
  return rc;
}
