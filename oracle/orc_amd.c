/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_se3.h header).
 *
 * Approximate-minimum-degree ordering on the BLOCK pattern of H, standing in for what
 * g2o::LinearSolverCSparse does with blockOrdering on ([UPSTREAM] cs_amd on the block pattern,
 * reached from the reference at g2o/g2o_graph.cpp:30-31,72-74).  The ordering does not change
 * the solve's result (only fill and time), so it is restated from the published algorithm
 * (Amestoy, Davis, Duff 1996: quotient graph, element absorption, approximate external degree)
 * rather than from CSparse source.  No supervariables; aggressive element absorption on.
 */
#include <stdlib.h>
#include <string.h>
#include "orc_api.h"

typedef struct { int *v; int n, cap; } ivec;
static void iv_push(ivec *a, int x) {
  if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 8; a->v = (int *)realloc(a->v, sizeof(int) * a->cap); }
  a->v[a->n++] = x;
}

/* adjacency given as CSR (xadj[n+1], adj[]) of an undirected graph WITHOUT self loops.
 * perm[k] = k-th node to eliminate.  Returns 0. */
int orc_amd_order(int n, const int *xadj, const int *adj, int *perm) {
  ivec *A = (ivec *)calloc(n, sizeof(ivec));   /* variable-variable adjacency */
  ivec *Ev = (ivec *)calloc(n, sizeof(ivec));  /* elements adjacent to a variable */
  ivec *Le = (ivec *)calloc(n, sizeof(ivec));  /* variables of an element (element id = pivot) */
  char *state = (char *)calloc(n, 1);          /* 0 = variable, 1 = live element, 2 = dead */
  int *deg = (int *)malloc(sizeof(int) * n);
  int *head = (int *)malloc(sizeof(int) * (n + 1));
  int *next = (int *)malloc(sizeof(int) * n), *prev = (int *)malloc(sizeof(int) * n);
  int *mark = (int *)calloc(n, sizeof(int)), *wst = (int *)calloc(n, sizeof(int));
  int *w = (int *)malloc(sizeof(int) * n);
  int stamp = 0, mindeg = 0;
  for (int d = 0; d <= n; ++d) head[d] = -1;
  for (int i = 0; i < n; ++i) {
    for (int p = xadj[i]; p < xadj[i + 1]; ++p) if (adj[p] != i) iv_push(&A[i], adj[p]);
    deg[i] = A[i].n;
  }
#define DL_INSERT(i) do { int d_ = deg[i]; next[i] = head[d_]; prev[i] = -1; \
    if (head[d_] >= 0) prev[head[d_]] = i; head[d_] = i; } while (0)
#define DL_REMOVE(i) do { if (prev[i] >= 0) next[prev[i]] = next[i]; else head[deg[i]] = next[i]; \
    if (next[i] >= 0) prev[next[i]] = prev[i]; } while (0)
  for (int i = n - 1; i >= 0; --i) DL_INSERT(i);

  ivec Lp = {0, 0, 0};
  for (int k = 0; k < n; ++k) {
    while (head[mindeg] < 0) ++mindeg;
    const int p = head[mindeg];
    DL_REMOVE(p);
    perm[k] = p;
    /* Lp = (A_p U union of L_e, e in E_p) \ {p} */
    ++stamp; Lp.n = 0; mark[p] = stamp;
    for (int t = 0; t < A[p].n; ++t) {
      int v = A[p].v[t];
      if (state[v] == 0 && mark[v] != stamp) { mark[v] = stamp; iv_push(&Lp, v); }
    }
    for (int t = 0; t < Ev[p].n; ++t) {
      int e = Ev[p].v[t];
      if (state[e] != 1) continue;
      for (int u = 0; u < Le[e].n; ++u) {
        int v = Le[e].v[u];
        if (state[v] == 0 && mark[v] != stamp) { mark[v] = stamp; iv_push(&Lp, v); }
      }
      state[e] = 2; free(Le[e].v); Le[e].v = 0; Le[e].n = Le[e].cap = 0;   /* absorbed */
    }
    state[p] = 1;
    free(A[p].v); A[p].v = 0; A[p].n = A[p].cap = 0;
    free(Ev[p].v); Ev[p].v = 0; Ev[p].n = Ev[p].cap = 0;
    Le[p].v = (int *)malloc(sizeof(int) * (Lp.n ? Lp.n : 1)); Le[p].cap = Lp.n; Le[p].n = Lp.n;
    memcpy(Le[p].v, Lp.v, sizeof(int) * Lp.n);
    /* prune lists of i in Lp; first pass of |Le \ Lp| */
    for (int t = 0; t < Lp.n; ++t) {
      const int i = Lp.v[t];
      int m = 0;
      for (int u = 0; u < Ev[i].n; ++u) { int e = Ev[i].v[u]; if (state[e] == 1) Ev[i].v[m++] = e; }
      Ev[i].n = m;
      m = 0;
      for (int u = 0; u < A[i].n; ++u) {
        int v = A[i].v[u];
        if (state[v] == 0 && mark[v] != stamp) A[i].v[m++] = v;   /* covered by element p otherwise */
      }
      A[i].n = m;
      for (int u = 0; u < Ev[i].n; ++u) {
        int e = Ev[i].v[u];
        if (wst[e] != stamp) { wst[e] = stamp; w[e] = Le[e].n; }
        --w[e];
      }
    }
    /* degrees */
    const int lp1 = Lp.n - 1, rem = n - k - 1;
    for (int t = 0; t < Lp.n; ++t) {
      const int i = Lp.v[t];
      int d = A[i].n + lp1, m = 0;
      for (int u = 0; u < Ev[i].n; ++u) {
        int e = Ev[i].v[u];
        if (w[e] == 0) { /* aggressive absorption: L_e subset of L_p */
          if (state[e] == 1) { state[e] = 2; free(Le[e].v); Le[e].v = 0; Le[e].n = Le[e].cap = 0; }
          continue;
        }
        if (state[e] != 1) continue;
        d += w[e]; Ev[i].v[m++] = e;
      }
      Ev[i].n = m;
      iv_push(&Ev[i], p);
      int dold = deg[i] + lp1;
      if (d > dold) d = dold;
      if (d > rem - 1) d = rem - 1;
      if (d < 0) d = 0;
      DL_REMOVE(i);
      deg[i] = d;
      DL_INSERT(i);
      if (d < mindeg) mindeg = d;
    }
  }
  for (int i = 0; i < n; ++i) { free(A[i].v); free(Ev[i].v); free(Le[i].v); }
  free(A); free(Ev); free(Le); free(state); free(deg); free(head); free(next); free(prev);
  free(mark); free(wst); free(w); free(Lp.v);
  return 0;
}
