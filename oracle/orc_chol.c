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

/* ---- multi-threaded numeric phase (the "OpenMP on one socket" leg of the CPU baseline).
 * Same up-looking arithmetic, row by row; rows in disjoint sub-trees of the elimination tree touch disjoint columns of L
 * (the pattern of row k is a set of descendants of k), so the tree is cut into sub-trees that run concurrently, one row
 * at a time each with private work arrays; their common ancestors follow serially.  Within a column the entries still
 * arrive in ascending row order (an ancestor's index exceeds every index of the sub-tree), so L is bit-identical to the
 * single-threaded factor. */
#ifdef _OPENMP
#include <omp.h>
#endif
static int g_orc_threads = 1;
void orc_set_threads(int n) { g_orc_threads = n > 0 ? n : 1; }
static int g_orc_solver = 0;
void orc_set_solver(int kind) { g_orc_solver = kind == 1 ? 1 : 0; }
int orc_get_solver(void) { return g_orc_solver; }
int orc_get_threads(void) { return g_orc_threads; }

static int row_upsolve(orc_chol *c, const int *Cp, const int *Ci, const double *Cx, int k, int *flag, int *s, double *x) {
  const int n = c->n;
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
  if (!(d > 0) || !isfinite(d)) return -1;
  const long long pk = c->cnext[k]++;
  c->Li[pk] = k; c->Lx[pk] = sqrt(d);
  return 0;
}

int orc_chol_numeric_mt(orc_chol *c, const int *Cp, const int *Ci, const double *Cx, int nthreads) {
  const int n = c->n;
  if (nthreads <= 1 || n < 1000) return orc_chol_numeric(c, Cp, Ci, Cx);
  /* sub-tree weights ~ flops of the rows below: sum of (column count)^2 */
  double *w = (double *)malloc(sizeof(double) * n);
  int *first = (int *)malloc(sizeof(int) * n), *next = (int *)malloc(sizeof(int) * n), *grp = (int *)malloc(sizeof(int) * n);
  double total = 0;
  for (int k = 0; k < n; ++k) { const double cc = (double)(c->Lp[k + 1] - c->Lp[k]); w[k] = cc * cc; first[k] = -1; next[k] = -1; grp[k] = -2; }
  for (int k = 0; k < n; ++k) { const int p = c->parent[k]; if (p >= 0) w[p] += w[k]; else total += w[k]; }
  for (int k = n - 1; k >= 0; --k) { const int p = c->parent[k]; if (p >= 0) { next[k] = first[p]; first[p] = k; } }
  /* open the heaviest sub-tree until none exceeds total / (0.75 threads): every opened root joins the serial top, so the
   * cut stays as low as balance allows (EPYC 9575F, cfg 2, 64 threads: factor 1.64 s with total / (8 threads), 0.70 s
   * with total / (0.5 threads); 2.1 s single-threaded) -- a simple array scan is enough here */
  int *roots = (int *)malloc(sizeof(int) * n);
  int nroots = 0;
  for (int k = 0; k < n; ++k) if (c->parent[k] < 0) roots[nroots++] = k;
  const char *cap_env = getenv("ORC_MT_CAP");
  const double cap = total / ((cap_env ? atof(cap_env) : 0.75) * nthreads);
  for (;;) {
    int best = -1;
    for (int q = 0; q < nroots; ++q) if (first[roots[q]] >= 0 && (best < 0 || w[roots[q]] > w[roots[best]])) best = q;
    if (best < 0 || w[roots[best]] <= cap || nroots > 64 * nthreads) break;
    const int r = roots[best];
    grp[r] = -1;                                        /* top */
    roots[best] = roots[--nroots];
    for (int ch = first[r]; ch >= 0; ch = next[ch]) roots[nroots++] = ch;
  }
  for (int q = 0; q < nroots; ++q) grp[roots[q]] = q;
  for (int k = n - 1; k >= 0; --k) if (grp[k] == -2) grp[k] = grp[c->parent[k]];     /* parents first */
  /* rows of each sub-tree in ascending order */
  int *cnt = (int *)calloc((size_t)nroots + 2, sizeof(int));
  for (int k = 0; k < n; ++k) cnt[grp[k] + 2]++;                                      /* slot 0: top */
  for (int q = 0; q <= nroots; ++q) cnt[q + 1] += cnt[q];
  int *rows = (int *)malloc(sizeof(int) * n), *fill = (int *)malloc(sizeof(int) * ((size_t)nroots + 2));
  memcpy(fill, cnt, sizeof(int) * ((size_t)nroots + 2));
  for (int k = 0; k < n; ++k) rows[fill[grp[k] + 1]++] = k;
  /* heavy sub-trees first */
  int *order = (int *)malloc(sizeof(int) * (nroots ? nroots : 1));
  for (int q = 0; q < nroots; ++q) order[q] = q;
  for (int a = 1; a < nroots; ++a) { int v = order[a], b = a - 1; while (b >= 0 && w[roots[order[b]]] < w[roots[v]]) { order[b + 1] = order[b]; --b; } order[b + 1] = v; }
  for (int k = 0; k < n; ++k) c->cnext[k] = c->Lp[k];
  int bad = 0;
#pragma omp parallel num_threads(nthreads)
  {
    int *flag = (int *)malloc(sizeof(int) * n), *st = (int *)malloc(sizeof(int) * n);
    double *x = (double *)calloc(n, sizeof(double));
    for (int k = 0; k < n; ++k) flag[k] = -1;
#pragma omp for schedule(dynamic, 1)
    for (int oq = 0; oq < nroots; ++oq) {
      const int q = order[oq];
      for (int r = cnt[q + 1]; r < cnt[q + 2]; ++r)
        if (row_upsolve(c, Cp, Ci, Cx, rows[r], flag, st, x)) {
#pragma omp atomic write
          bad = 1;
          break;
        }
    }
    free(flag); free(st); free(x);
  }
  if (!bad) {
    int *flag = c->flag, *st = c->stack;
    double *x = c->x;
    for (int k = 0; k < n; ++k) { flag[k] = -1; x[k] = 0; }
    for (int r = cnt[0]; r < cnt[1] && !bad; ++r) bad = row_upsolve(c, Cp, Ci, Cx, rows[r], flag, st, x) != 0;
  }
  free(w); free(first); free(next); free(grp); free(roots); free(cnt); free(rows); free(fill); free(order);
  return bad ? -1 : 0;
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
