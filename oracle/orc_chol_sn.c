/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_se3.h header).
 *
 * Supernodal sparse Cholesky, left-looking, dense column-major panels (the CHOLMOD class of solver, i.e. what a tuned
 * CPU deployment of the reference's back-end would use instead of g2o's LinearSolverCSparse / cs_chol; written from the
 * textbook algorithm: fundamental supernodes from the elimination tree and column counts, per-supernode row lists,
 * descendant lists threaded through `head/next`, dense update  U = L_d[rows >= s] L_d[rows in s]^T  scattered through a
 * relative-index map, dense right-looking factorisation of the panel).  No BLAS in this image: the dense kernels are
 * plain C written so that gcc -O3 -march=native vectorises their inner loops.  OpenMP: independent sub-trees of the
 * supernodal elimination tree run concurrently; the supernodes above them are processed one after the other with their
 * descendants' updates computed in parallel and applied in a fixed order (deterministic).
 * This is the second CPU leg of bench.py's cpu_baseline ("supernodal"); the simplicial leg (orc_chol.c) stays the
 * restatement of what the reference actually links. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "orc_api.h"

struct orc_sn {
  int n, ns;
  int *sn_start;        /* [ns+1] first column of every supernode */
  int *col_sn;          /* [n] */
  int *sparent;         /* [ns] supernodal elimination tree */
  long long *rptr;      /* [ns+1] -> ridx */
  int *ridx;            /* rows of a supernode: its own columns first, then the rows below, ascending */
  long long *xptr;      /* [ns+1] -> X */
  double *X;            /* panels, column-major, leading dimension = rows of the supernode */
  /* transposed pattern of the input (lower triangle by column), built once */
  int *Tp, *Ti, *Tmap;  /* Tmap: position in Cx of every transposed entry */
};

static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

orc_sn *orc_sn_symbolic(int n, const int *Cp, const int *Ci) {
  orc_sn *c = (orc_sn *)calloc(1, sizeof(orc_sn));
  c->n = n;
  /* elimination tree + column counts (as orc_chol.c) */
  int *parent = (int *)malloc(sizeof(int) * (n ? n : 1)), *anc = (int *)malloc(sizeof(int) * (n ? n : 1));
  for (int k = 0; k < n; ++k) {
    parent[k] = -1; anc[k] = -1;
    for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      while (i != -1 && i < k) { int nxt = anc[i]; anc[i] = k; if (nxt == -1) parent[i] = k; i = nxt; }
    }
  }
  int *cnt = (int *)calloc(n + 1, sizeof(int)), *flag = anc;
  for (int k = 0; k < n; ++k) flag[k] = -1;
  for (int k = 0; k < n; ++k) {
    flag[k] = k; cnt[k]++;
    for (int p = Cp[k]; p < Cp[k + 1]; ++p) { int i = Ci[p]; while (i < k && flag[i] != k) { flag[i] = k; cnt[i]++; i = parent[i]; } }
  }
  /* fundamental supernodes: column k continues the supernode of k-1 iff k-1's only parent is k, k has no other child and
   * the patterns nest (cnt[k-1] == cnt[k] + 1) */
  int *nchild = (int *)calloc(n + 1, sizeof(int));
  for (int k = 0; k < n; ++k) if (parent[k] >= 0) nchild[parent[k]]++;
  c->col_sn = (int *)malloc(sizeof(int) * (n ? n : 1));
  c->sn_start = (int *)malloc(sizeof(int) * (n + 1));
  int ns = 0;
  for (int k = 0; k < n; ++k) {
    const int cont = k > 0 && parent[k - 1] == k && nchild[k] == 1 && cnt[k - 1] == cnt[k] + 1;
    if (!cont) c->sn_start[ns++] = k;
  }
  c->sn_start[ns] = n;
  /* relaxed amalgamation (as supernodal solvers do to get BLAS-3-sized panels): a supernode g is merged into its parent f
   * when g's columns directly precede f's and the explicit zeros this introduces stay below a share of the merged panel.
   * rows(g + f) = cols(g) + rows(f): cnt of the first column of the merged supernode is set accordingly. */
  {
    const char *ze = getenv("ORC_SN_ZEROS");
    const double zmax = ze ? atof(ze) : 0.1;     /* swept on the 20 000-pose graph: 0 / 0.05 / 0.1 / 0.3 -> 0.42 / 0.38 / 0.34 / 0.65 s per factorisation, 1 thread */
    int m = 0;                                   /* merged supernodes written in place: start[m] */
    long long gz = 0;                            /* explicit zeros of the current group */
    int g0 = c->sn_start[0];
    for (int f = 1; f <= ns; ++f) {
      int merged = 0;
      if (f < ns) {
        const int f0 = c->sn_start[f], f1 = c->sn_start[f + 1];
        if (parent[f0 - 1] == f0) {              /* the group is a child of f and ends right before it */
          const long long gc = f0 - g0, gr = cnt[g0], fr = cnt[f0];
          const long long z = gz + gc * (fr - (gr - gc));        /* rows of f missing from the group's pattern, per group column */
          const long long size = (gc + fr) * (gc + (f1 - f0));
          if ((double)z <= zmax * (double)size || gc + (f1 - f0) <= 12) { cnt[g0] = (int)(gc + fr); gz = z; merged = 1; }
        }
      }
      if (!merged) { c->sn_start[m++] = g0; if (f < ns) { g0 = c->sn_start[f]; gz = 0; } }
    }
    ns = m;
    c->sn_start[ns] = n;
  }
  for (int s = 0; s < ns; ++s) for (int k = c->sn_start[s]; k < c->sn_start[s + 1]; ++k) c->col_sn[k] = s;
  c->sn_start[ns] = n;
  c->ns = ns;
  c->sparent = (int *)malloc(sizeof(int) * (ns ? ns : 1));
  c->rptr = (long long *)malloc(sizeof(long long) * (ns + 1));
  c->xptr = (long long *)malloc(sizeof(long long) * (ns + 1));
  c->rptr[0] = 0; c->xptr[0] = 0;
  for (int s = 0; s < ns; ++s) {
    const int c0 = c->sn_start[s], c1 = c->sn_start[s + 1];
    const long long nr = cnt[c0];                     /* rows of the first column = all rows of the supernode */
    c->rptr[s + 1] = c->rptr[s] + nr;
    c->xptr[s + 1] = c->xptr[s] + nr * (c1 - c0);
    const int last = c1 - 1;
    c->sparent[s] = parent[last] >= 0 ? c->col_sn[parent[last]] : -1;
  }
  /* transposed pattern: lower triangle by column (entry (i, k), i <= k of the upper CSC = entry (k, i) of the lower) */
  c->Tp = (int *)calloc(n + 1, sizeof(int));
  for (int k = 0; k < n; ++k) for (int p = Cp[k]; p < Cp[k + 1]; ++p) if (Ci[p] <= k) c->Tp[Ci[p] + 1]++;
  for (int k = 0; k < n; ++k) c->Tp[k + 1] += c->Tp[k];
  c->Ti = (int *)malloc(sizeof(int) * (size_t)(c->Tp[n] ? c->Tp[n] : 1));
  c->Tmap = (int *)malloc(sizeof(int) * (size_t)(c->Tp[n] ? c->Tp[n] : 1));
  {
    int *fill = (int *)malloc(sizeof(int) * (n ? n : 1));
    memcpy(fill, c->Tp, sizeof(int) * n);
    for (int k = 0; k < n; ++k) for (int p = Cp[k]; p < Cp[k + 1]; ++p) if (Ci[p] <= k) { const int q = fill[Ci[p]]++; c->Ti[q] = k; c->Tmap[q] = p; }
    free(fill);
  }
  /* row lists: rows(s) = own columns, then  (A's rows below, columns of s)  U  (rows of the children below s) ; children
   * come before parents, so one ascending sweep with a marker array suffices */
  c->ridx = (int *)malloc(sizeof(int) * (size_t)(c->rptr[ns] ? c->rptr[ns] : 1));
  int *mark = flag;
  for (int k = 0; k < n; ++k) mark[k] = -1;
  int *chead = (int *)malloc(sizeof(int) * (ns ? ns : 1)), *cnext = (int *)malloc(sizeof(int) * (ns ? ns : 1));
  for (int s = 0; s < ns; ++s) { chead[s] = -1; cnext[s] = -1; }
  for (int s = ns - 1; s >= 0; --s) if (c->sparent[s] >= 0) { cnext[s] = chead[c->sparent[s]]; chead[c->sparent[s]] = s; }
  for (int s = 0; s < ns; ++s) {
    const int c0 = c->sn_start[s], c1 = c->sn_start[s + 1];
    int *r = c->ridx + c->rptr[s];
    long long m = 0;
    for (int k = c0; k < c1; ++k) { r[m++] = k; mark[k] = s; }
    const long long own = m;
    for (int k = c0; k < c1; ++k)
      for (int q = c->Tp[k]; q < c->Tp[k + 1]; ++q) { const int i = c->Ti[q]; if (i >= c1 && mark[i] != s) { mark[i] = s; r[m++] = i; } }
    for (int ch = chead[s]; ch >= 0; ch = cnext[ch]) {
      const int *cr = c->ridx + c->rptr[ch];
      const long long cm = c->rptr[ch + 1] - c->rptr[ch];
      for (long long q = c->sn_start[ch + 1] - c->sn_start[ch]; q < cm; ++q) { const int i = cr[q]; if (i >= c1 && mark[i] != s) { mark[i] = s; r[m++] = i; } }
    }
    qsort(r + own, (size_t)(m - own), sizeof(int), cmp_int);
    if (m != c->rptr[s + 1] - c->rptr[s]) { fprintf(stderr, "orc_sn_symbolic: row count mismatch at supernode %d (%lld vs %lld)\n", s, m, c->rptr[s + 1] - c->rptr[s]); abort(); }
  }
  c->X = (double *)malloc(sizeof(double) * (size_t)(c->xptr[ns] ? c->xptr[ns] : 1));
  if (getenv("ORC_SN_STATS")) {
    int big = 0; long long bigx = 0;
    for (int s2 = 0; s2 < ns; ++s2) if (c->sn_start[s2 + 1] - c->sn_start[s2] >= 48) { ++big; bigx += c->xptr[s2 + 1] - c->xptr[s2]; }
    fprintf(stderr, "[orc sn] n %d supernodes %d (%.1f columns each), panel entries %lld, %d supernodes of >= 48 columns hold %.0f%%\n", n, ns, (double)n / (ns ? ns : 1), c->xptr[ns], big, 100.0 * bigx / (c->xptr[ns] ? c->xptr[ns] : 1));
  }
  free(parent); free(anc); free(cnt); free(nchild); free(chead); free(cnext);
  return c;
}

/* U (m x k1, column-major, ld m) = A A1^T, A = rows [0, m), A1 = rows [0, k1) of the same column-major matrix (ld, nd columns) */
static void syrk_like(const double *restrict Xd, long long ld, int nd, int m, int k1, double *restrict U, int nthreads) {
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1 && (long long)m * k1 * nd > 4000000)
  for (int j0 = 0; j0 < k1; j0 += 4) {
    const int jn = k1 - j0 < 4 ? k1 - j0 : 4;
    for (int i0 = 0; i0 < m; i0 += 16) {
      const int in = m - i0 < 16 ? m - i0 : 16;
      double acc[4][16];
      for (int a = 0; a < 4; ++a) for (int t = 0; t < 16; ++t) acc[a][t] = 0.0;
      if (in == 16 && jn == 4) {
        for (int cc = 0; cc < nd; ++cc) {
          const double *restrict a = Xd + (long long)cc * ld + i0, *restrict b = Xd + (long long)cc * ld + j0;
          const double b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
          for (int t = 0; t < 16; ++t) { acc[0][t] += a[t] * b0; acc[1][t] += a[t] * b1; acc[2][t] += a[t] * b2; acc[3][t] += a[t] * b3; }
        }
      } else {
        for (int cc = 0; cc < nd; ++cc) {
          const double *restrict a = Xd + (long long)cc * ld + i0, *restrict b = Xd + (long long)cc * ld + j0;
          for (int q = 0; q < jn; ++q) { const double bq = b[q]; for (int t = 0; t < in; ++t) acc[q][t] += a[t] * bq; }
        }
      }
      for (int q = 0; q < jn; ++q) for (int t = 0; t < in; ++t) U[(long long)(j0 + q) * m + i0 + t] = acc[q][t];
    }
  }
}

typedef struct { int *map; double *U; long long ucap; } sn_work;

/* dense right-looking Cholesky of the nc x nc top of a column-major nr x nc panel; the rows below are carried along.
 * Blocked: PB columns are factored unblocked (their part of the panel stays in cache), then the trailing columns take the
 * rank-PB update through the register-blocked kernel above (threaded for the big panels at the top of the tree). */
#define PB 24
static int panel_factor(double *restrict P, long long nr, int nc, sn_work *w, int nthreads) {
  for (int j0 = 0; j0 < nc; j0 += PB) {
    const int jb = nc - j0 < PB ? nc - j0 : PB;
    for (int j = j0; j < j0 + jb; ++j) {
      double *restrict cj = P + (long long)j * nr;
      const double d = cj[j];
      if (!(d > 0) || !isfinite(d)) return -1;
      const double s = sqrt(d), inv = 1.0 / s;
      cj[j] = s;
      for (long long i = j + 1; i < nr; ++i) cj[i] *= inv;
      for (int k = j + 1; k < j0 + jb; ++k) {
        double *restrict ck = P + (long long)k * nr;
        const double f = cj[k];
        for (long long i = k; i < nr; ++i) ck[i] -= cj[i] * f;
      }
    }
    const int k0 = j0 + jb;
    if (k0 >= nc) break;
    const int m = (int)(nr - k0), k1 = nc - k0;
    const long long need = (long long)m * k1;
    if (need > w->ucap) { free(w->U); w->ucap = need + need / 4 + 1024; w->U = (double *)malloc(sizeof(double) * (size_t)w->ucap); }
    syrk_like(P + (long long)j0 * nr + k0, nr, jb, m, k1, w->U, nthreads);
    for (int j = 0; j < k1; ++j) {
      double *restrict col = P + (long long)(k0 + j) * nr + k0;
      const double *restrict u = w->U + (long long)j * m;
      for (int i = j; i < m; ++i) col[i] -= u[i];
    }
  }
  return 0;
}


/* assemble A into the panel of s, apply the updates of the descendants in `list` (ascending), factor */
static int sn_one(orc_sn *c, const double *Cx, int s, const int *list, int nlist, long long *dpos, sn_work *w, int inner_threads) {
  const int c0 = c->sn_start[s], c1 = c->sn_start[s + 1], nc = c1 - c0;
  const int *r = c->ridx + c->rptr[s];
  const long long nr = c->rptr[s + 1] - c->rptr[s];
  double *P = c->X + c->xptr[s];
  memset(P, 0, sizeof(double) * (size_t)(nr * nc));
  for (long long q = 0; q < nr; ++q) w->map[r[q]] = (int)q;
  for (int k = c0; k < c1; ++k)
    for (int q = c->Tp[k]; q < c->Tp[k + 1]; ++q) P[(long long)(k - c0) * nr + w->map[c->Ti[q]]] += Cx[c->Tmap[q]];
  /* sizes of the descendants' updates */
  long long need = 0;
  long long *uoff = (long long *)malloc(sizeof(long long) * (nlist + 1));
  int *k1s = (int *)malloc(sizeof(int) * (nlist ? nlist : 1));
  uoff[0] = 0;
  for (int q = 0; q < nlist; ++q) {
    const int d = list[q];
    const int *dr = c->ridx + c->rptr[d];
    const long long dnr = c->rptr[d + 1] - c->rptr[d], p0 = dpos[d];
    long long p1 = p0;
    while (p1 < dnr && dr[p1] < c1) ++p1;
    k1s[q] = (int)(p1 - p0);
    uoff[q + 1] = uoff[q] + (dnr - p0) * (p1 - p0);
    need = uoff[q + 1];
  }
  if (need > w->ucap) { free(w->U); w->ucap = need + need / 4 + 1024; w->U = (double *)malloc(sizeof(double) * (size_t)w->ucap); }
#pragma omp parallel for schedule(dynamic, 1) num_threads(inner_threads) if (inner_threads > 1 && nlist > 1 && need > 200000)
  for (int q = 0; q < nlist; ++q) {
    const int d = list[q];
    const long long dnr = c->rptr[d + 1] - c->rptr[d], p0 = dpos[d];
    const int dnc = c->sn_start[d + 1] - c->sn_start[d];
    syrk_like(c->X + c->xptr[d] + p0, dnr, dnc, (int)(dnr - p0), k1s[q], w->U + uoff[q], 1);
  }
  for (int q = 0; q < nlist; ++q) {                       /* scatter in list order: deterministic */
    const int d = list[q];
    const int *dr = c->ridx + c->rptr[d];
    const long long dnr = c->rptr[d + 1] - c->rptr[d], p0 = dpos[d];
    const int m = (int)(dnr - p0), k1 = k1s[q];
    const double *U = w->U + uoff[q];
    for (int j = 0; j < k1; ++j) {
      double *col = P + (long long)(dr[p0 + j] - c0) * nr;
      const double *u = U + (long long)j * m;
      for (int i = j; i < m; ++i) col[w->map[dr[p0 + i]]] -= u[i];
    }
    dpos[d] = p0 + k1;
  }
  free(uoff); free(k1s);
  return panel_factor(P, nr, nc, w, inner_threads);
}

int orc_sn_numeric(orc_sn *c, const double *Cx, int nthreads) {
  const int ns = c->ns, n = c->n;
  if (nthreads < 1) nthreads = 1;
  long long *dpos = (long long *)malloc(sizeof(long long) * (ns ? ns : 1));
  for (int s = 0; s < ns; ++s) dpos[s] = c->sn_start[s + 1] - c->sn_start[s];      /* first row below the supernode */
  /* descendant lists: lists[t] = supernodes whose next pending row lies in t; built as arrays per target, filled in
   * ascending d (children are numbered below parents) */
  int *lcount = (int *)calloc(ns + 1, sizeof(int));
  /* a descendant d visits: the supernodes of its rows below, one visit per distinct supernode */
  for (int d = 0; d < ns; ++d) {
    const int *dr = c->ridx + c->rptr[d];
    const long long dnr = c->rptr[d + 1] - c->rptr[d];
    int last = -1;
    for (long long q = c->sn_start[d + 1] - c->sn_start[d]; q < dnr; ++q) { const int t = c->col_sn[dr[q]]; if (t != last) { lcount[t + 1]++; last = t; } }
  }
  for (int s = 0; s < ns; ++s) lcount[s + 1] += lcount[s];
  int *lists = (int *)malloc(sizeof(int) * (size_t)(lcount[ns] ? lcount[ns] : 1)), *lfill = (int *)malloc(sizeof(int) * (ns ? ns : 1));
  memcpy(lfill, lcount, sizeof(int) * ns);
  for (int d = 0; d < ns; ++d) {
    const int *dr = c->ridx + c->rptr[d];
    const long long dnr = c->rptr[d + 1] - c->rptr[d];
    int last = -1;
    for (long long q = c->sn_start[d + 1] - c->sn_start[d]; q < dnr; ++q) { const int t = c->col_sn[dr[q]]; if (t != last) { lists[lfill[t]++] = d; last = t; } }
  }
  /* sub-tree decomposition of the supernodal elimination tree by flops ~ sum nr^2 nc */
  int *grp = (int *)malloc(sizeof(int) * (ns ? ns : 1));
  int nroots = 0, *roots = NULL, *order = NULL;
  if (nthreads > 1 && ns > 64 && c->xptr[ns] > 2000000) {
    double *w = (double *)malloc(sizeof(double) * ns);
    int *first = (int *)malloc(sizeof(int) * ns), *next = (int *)malloc(sizeof(int) * ns);
    double total = 0;
    for (int s = 0; s < ns; ++s) { const double nr = (double)(c->rptr[s + 1] - c->rptr[s]), nc = c->sn_start[s + 1] - c->sn_start[s]; w[s] = nr * nr * nc; first[s] = -1; next[s] = -1; grp[s] = -2; }
    for (int s = 0; s < ns; ++s) { const int p = c->sparent[s]; if (p >= 0) w[p] += w[s]; else total += w[s]; }
    for (int s = ns - 1; s >= 0; --s) { const int p = c->sparent[s]; if (p >= 0) { next[s] = first[p]; first[p] = s; } }
    roots = (int *)malloc(sizeof(int) * ns);
    for (int s = 0; s < ns; ++s) if (c->sparent[s] < 0) roots[nroots++] = s;
    const double cap = total / (2.0 * nthreads);
    for (;;) {
      int best = -1;
      for (int q = 0; q < nroots; ++q) if (first[roots[q]] >= 0 && (best < 0 || w[roots[q]] > w[roots[best]])) best = q;
      if (best < 0 || w[roots[best]] <= cap || nroots > 64 * nthreads) break;
      const int r = roots[best];
      grp[r] = -1;
      roots[best] = roots[--nroots];
      for (int ch = first[r]; ch >= 0; ch = next[ch]) roots[nroots++] = ch;
    }
    for (int q = 0; q < nroots; ++q) grp[roots[q]] = q;
    for (int s = ns - 1; s >= 0; --s) if (grp[s] == -2) grp[s] = grp[c->sparent[s]];
    order = (int *)malloc(sizeof(int) * (nroots ? nroots : 1));
    for (int q = 0; q < nroots; ++q) order[q] = q;
    for (int a = 1; a < nroots; ++a) { int v = order[a], b = a - 1; while (b >= 0 && w[roots[order[b]]] < w[roots[v]]) { order[b + 1] = order[b]; --b; } order[b + 1] = v; }
    free(w); free(first); free(next);
  } else {
    for (int s = 0; s < ns; ++s) grp[s] = -1;
  }
  int bad = 0;
  if (nroots > 0) {
#pragma omp parallel num_threads(nthreads)
    {
      sn_work w; w.map = (int *)malloc(sizeof(int) * (n ? n : 1)); w.U = NULL; w.ucap = 0;
#pragma omp for schedule(dynamic, 1)
      for (int oq = 0; oq < nroots; ++oq) {
        const int g = order[oq];
        /* supernodes of the sub-tree in ascending order: a sub-tree of a post-ordered-enough tree is not contiguous in
         * general, so walk all supernodes <= its root and pick the members */
        for (int s = 0; s <= roots[g]; ++s)
          if (grp[s] == g && sn_one(c, Cx, s, lists + lcount[s], lcount[s + 1] - lcount[s], dpos, &w, 1)) {
#pragma omp atomic write
            bad = 1;
          }
      }
      free(w.map); free(w.U);
    }
  }
  if (!bad) {
    sn_work w; w.map = (int *)malloc(sizeof(int) * (n ? n : 1)); w.U = NULL; w.ucap = 0;
    for (int s = 0; s < ns && !bad; ++s)
      if (grp[s] == -1) bad = sn_one(c, Cx, s, lists + lcount[s], lcount[s + 1] - lcount[s], dpos, &w, nthreads) != 0;
    free(w.map); free(w.U);
  }
  free(dpos); free(lcount); free(lists); free(lfill); free(grp); free(roots); free(order);
  return bad ? -1 : 0;
}

void orc_sn_solve(const orc_sn *c, double *x) {
  const int ns = c->ns;
  for (int s = 0; s < ns; ++s) {                           /* L y = b */
    const int c0 = c->sn_start[s], nc = c->sn_start[s + 1] - c0;
    const int *r = c->ridx + c->rptr[s];
    const long long nr = c->rptr[s + 1] - c->rptr[s];
    const double *P = c->X + c->xptr[s];
    for (int j = 0; j < nc; ++j) {
      const double *cj = P + (long long)j * nr;
      const double xj = (x[c0 + j] /= cj[j]);
      for (long long i = j + 1; i < nr; ++i) x[r[i]] -= cj[i] * xj;
    }
  }
  for (int s = ns - 1; s >= 0; --s) {                      /* L' x = y */
    const int c0 = c->sn_start[s], nc = c->sn_start[s + 1] - c0;
    const int *r = c->ridx + c->rptr[s];
    const long long nr = c->rptr[s + 1] - c->rptr[s];
    const double *P = c->X + c->xptr[s];
    for (int j = nc - 1; j >= 0; --j) {
      const double *cj = P + (long long)j * nr;
      double acc = x[c0 + j];
      for (long long i = j + 1; i < nr; ++i) acc -= cj[i] * x[r[i]];
      x[c0 + j] = acc / cj[j];
    }
  }
}

long long orc_sn_nnz(const orc_sn *c) { return c->xptr[c->ns]; }
int orc_sn_count(const orc_sn *c) { return c->ns; }

void orc_sn_free(orc_sn *c) {
  if (!c) return;
  free(c->sn_start); free(c->col_sn); free(c->sparent); free(c->rptr); free(c->ridx); free(c->xptr); free(c->X);
  free(c->Tp); free(c->Ti); free(c->Tmap); free(c);
}
