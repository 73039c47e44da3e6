/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_se3.h header).
 *
 * Scalar simplicial sparse Cholesky, up-looking, single-threaded: the class of solver
 * g2o::LinearSolverCSparse uses ([UPSTREAM] cs_schol + cs_chol + two triangular solves; selected
 * by the reference at g2o/g2o_graph.cpp:30-31,72-74).  Written from the textbook algorithms
 * (Liu's elimination tree with path compression; row-subtree reach; up-looking numeric).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "orc_api.h"

struct orc_chol {
  int n;
  int *parent;
  long long *Lp;   /* column pointers (nnz can exceed 2^31) */
  int *Li;
  double *Lx;
  long long *cnext; /* work: next free slot per column */
  int *flag, *stack;
  double *x;
};

static void etree(int n, const int *Cp, const int *Ci, int *parent) {
  int *anc = (int *)malloc(sizeof(int) * n);
  for (int k = 0; k < n; ++k) {
    parent[k] = -1; anc[k] = -1;
    for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      while (i != -1 && i < k) {
        int nxt = anc[i];
        anc[i] = k;
        if (nxt == -1) parent[i] = k;
        i = nxt;
      }
    }
  }
  free(anc);
}

orc_chol *orc_chol_symbolic(int n, const int *Cp, const int *Ci) {
  orc_chol *c = (orc_chol *)calloc(1, sizeof(orc_chol));
  c->n = n;
  c->parent = (int *)malloc(sizeof(int) * (n ? n : 1));
  etree(n, Cp, Ci, c->parent);
  /* column counts by walking every row subtree once: O(nnz(L)) */
  long long *cnt = (long long *)calloc(n + 1, sizeof(long long));
  int *flag = (int *)malloc(sizeof(int) * (n ? n : 1));
  for (int k = 0; k < n; ++k) flag[k] = -1;
  for (int k = 0; k < n; ++k) {
    flag[k] = k; cnt[k]++;                /* diagonal */
    for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      while (i < k && flag[i] != k) { flag[i] = k; cnt[i]++; i = c->parent[i]; }
    }
  }
  c->Lp = (long long *)malloc(sizeof(long long) * (n + 1));
  c->Lp[0] = 0;
  for (int k = 0; k < n; ++k) c->Lp[k + 1] = c->Lp[k] + cnt[k];
  free(cnt);
  c->Li = (int *)malloc(sizeof(int) * (size_t)(c->Lp[n] ? c->Lp[n] : 1));
  c->Lx = (double *)malloc(sizeof(double) * (size_t)(c->Lp[n] ? c->Lp[n] : 1));
  c->cnext = (long long *)malloc(sizeof(long long) * (n ? n : 1));
  c->flag = flag;
  c->stack = (int *)malloc(sizeof(int) * (n ? n : 1));
  c->x = (double *)calloc(n ? n : 1, sizeof(double));
  return c;
}

int orc_chol_numeric(orc_chol *c, const int *Cp, const int *Ci, const double *Cx) {
  const int n = c->n;
  int *flag = c->flag, *s = c->stack;
  double *x = c->x;
  for (int k = 0; k < n; ++k) { c->cnext[k] = c->Lp[k]; flag[k] = -1; x[k] = 0; }
  for (int k = 0; k < n; ++k) {
    /* pattern of row k of L in topological order -> s[top..n) */
    int top = n;
    flag[k] = k;
    double d = 0;
    for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      if (i > k) continue;
      if (i == k) { d += Cx[p]; continue; }
      x[i] += Cx[p];
      int len = 0;
      while (flag[i] != k) { s[len++] = i; flag[i] = k; i = c->parent[i]; }
      while (len > 0) s[--top] = s[--len];
    }
    for (; top < n; ++top) {
      const int j = s[top];
      const long long pj = c->Lp[j];
      const double lkj = x[j] / c->Lx[pj];
      x[j] = 0;
      const long long pe = c->cnext[j];
      for (long long p = pj + 1; p < pe; ++p) x[c->Li[p]] -= c->Lx[p] * lkj;
      d -= lkj * lkj;
      c->Li[pe] = k; c->Lx[pe] = lkj; c->cnext[j] = pe + 1;
    }
    if (!(d > 0) || !isfinite(d)) return -1;   /* not positive definite */
    const long long pk = c->cnext[k]++;
    c->Li[pk] = k; c->Lx[pk] = sqrt(d);
  }
  return 0;
}

void orc_chol_solve(const orc_chol *c, double *x) {
  const int n = c->n;
  for (int j = 0; j < n; ++j) {            /* L y = b */
    const long long p0 = c->Lp[j], p1 = c->Lp[j + 1];
    const double xj = (x[j] /= c->Lx[p0]);
    for (long long p = p0 + 1; p < p1; ++p) x[c->Li[p]] -= c->Lx[p] * xj;
  }
  for (int j = n - 1; j >= 0; --j) {       /* L' x = y */
    const long long p0 = c->Lp[j], p1 = c->Lp[j + 1];
    double acc = x[j];
    for (long long p = p0 + 1; p < p1; ++p) acc -= c->Lx[p] * x[c->Li[p]];
    x[j] = acc / c->Lx[p0];
  }
}

long long orc_chol_nnz(const orc_chol *c) { return c->Lp[c->n]; }

int orc_chol_etree_height(const orc_chol *c) {
  int n = c->n, h = 0;
  int *lev = (int *)calloc(n ? n : 1, sizeof(int));
  for (int k = 0; k < n; ++k) {            /* parent[k] > k, so one forward sweep suffices */
    if (lev[k] + 1 > h) h = lev[k] + 1;
    int p = c->parent[k];
    if (p >= 0 && lev[p] < lev[k] + 1) lev[p] = lev[k] + 1;
  }
  free(lev);
  return h;
}

void orc_chol_free(orc_chol *c) {
  if (!c) return;
  free(c->parent); free(c->Lp); free(c->Li); free(c->Lx); free(c->cnext); free(c->flag);
  free(c->stack); free(c->x); free(c);
}
