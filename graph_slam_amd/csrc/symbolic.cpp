// Symbolic phase of the block Cholesky (host, once per graph structure): block elimination tree,
// block column patterns of L, left-looking update lists, row lists for the triangular solves and
// the task/level schedule the device kernels follow.  Replaces what g2o does in
// BlockSolver::buildStructure + LinearSolverCSparse's symbolic decomposition ([UPSTREAM], reached
// from the reference at g2o/g2o_graph.cpp:246-249 on iteration 0 of every optimize() call).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>
#include "fgo_internal.hpp"

namespace fgo {

void build_symbolic(const BlockGraph &g, const std::vector<int> &perm, int64_t task_work_limit, int64_t chain_work_limit, Symbolic &S) {
  const int nb = g.n;
  const bool prof = std::getenv("FGO_SYM_PROFILE") != nullptr;
  auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tprev = tnow();
  auto lap = [&](const char *what) { if (prof) { const double t = tnow(); std::fprintf(stderr, "[fgo symbolic] %-28s %.1f ms\n", what, 1e3 * (t - tprev)); tprev = t; } };
  S = Symbolic();
  S.nb = nb;
  // ---- column patterns: pattern(k) = A(k+1:, k)  U  union over children c of pattern(c) \ {k}
  std::vector<std::vector<int>> pat(nb);
  auto pattern_pass = [&](const std::vector<int> &pm) {
    S.perm = pm;
    S.iperm.assign(nb, -1);
    for (int k = 0; k < nb; ++k) S.iperm[pm[k]] = k;
    S.parent.assign(nb, -1);
    S.colptr.assign(nb + 1, 0);
    S.max_col_blocks = 0;
    std::vector<int> first_child(nb, -1), next_sib(nb, -1), mark(nb, -1);
    std::vector<int> tmp;
    for (int k = 0; k < nb; ++k) {
      tmp.clear();
      mark[k] = k;
      const int v = pm[k];
      for (int p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
        const int i = S.iperm[g.adj[p]];
        if (i > k && mark[i] != k) { mark[i] = k; tmp.push_back(i); }
      }
      for (int c = first_child[k]; c >= 0; c = next_sib[c])
        for (int i : pat[c])
          if (i > k && mark[i] != k) { mark[i] = k; tmp.push_back(i); }
      std::sort(tmp.begin(), tmp.end());
      pat[k] = tmp;
      if (!tmp.empty()) {
        const int par = tmp[0];
        S.parent[k] = par;
        next_sib[k] = first_child[par]; first_child[par] = k;
      }
      S.colptr[k + 1] = S.colptr[k] + 1 + (int64_t)tmp.size();
      S.max_col_blocks = std::max(S.max_col_blocks, 1 + (int)tmp.size());
    }
  };
  pattern_pass(perm);
  {
    // Post-order the elimination tree (same fill, same tree): every sub-tree becomes a contiguous column range, so a
    // light sub-tree's blocks are one contiguous range of L and the leaf kernel can keep them in LDS under local
    // indices.  Children keep their relative order; the nested-dissection order is nearly post-ordered already.
    std::vector<int> head(nb, -1), nxt(nb, -1), post;
    for (int k = nb - 1; k >= 0; --k) if (S.parent[k] >= 0) { nxt[k] = head[S.parent[k]]; head[S.parent[k]] = k; }
    post.reserve(nb);
    std::vector<int> stack;
    for (int r = 0; r < nb; ++r) {
      if (S.parent[r] >= 0) continue;
      stack.push_back(r);
      while (!stack.empty()) {
        const int k = stack.back();
        if (head[k] >= 0) { const int c = head[k]; head[k] = nxt[c]; stack.push_back(c); }   // next unvisited child
        else { post.push_back(k); stack.pop_back(); }
      }
    }
    bool identity = true;
    for (int k = 0; k < nb && identity; ++k) identity = post[k] == k;
    if (!identity) {
      std::vector<int> pm(nb);
      for (int k = 0; k < nb; ++k) pm[k] = S.perm[post[k]];
      for (auto &v : pat) v.clear();
      pattern_pass(pm);
    }
  }
  S.nnzL = S.colptr[nb];
  S.rowidx.resize(S.nnzL);
  S.blkcol.resize(S.nnzL);
  for (int k = 0; k < nb; ++k) {
    int64_t p = S.colptr[k];
    S.rowidx[p] = k; S.blkcol[p] = k; ++p;
    for (int i : pat[k]) { S.rowidx[p] = i; S.blkcol[p] = k; ++p; }
    std::vector<int>().swap(pat[k]);
  }

  lap("column patterns");
  // ---- row lists: row k = { L_kj : j < k }, ascending j
  S.rowptr.assign(nb + 1, 0);
  for (int j = 0; j < nb; ++j)
    for (int64_t p = S.colptr[j] + 1; p < S.colptr[j + 1]; ++p) S.rowptr[S.rowidx[p] + 1]++;
  for (int k = 0; k < nb; ++k) S.rowptr[k + 1] += S.rowptr[k];
  S.row_blk.resize(S.rowptr[nb]);
  S.row_col.resize(S.rowptr[nb]);
  {
    std::vector<int64_t> fill(S.rowptr.begin(), S.rowptr.end() - 1);
    for (int j = 0; j < nb; ++j)
      for (int64_t p = S.colptr[j] + 1; p < S.colptr[j + 1]; ++p) {
        const int64_t o = fill[S.rowidx[p]]++;
        S.row_blk[o] = (int)p; S.row_col[o] = j;
      }
  }
  lap("row lists");

  // ---- update lists, generated per TARGET column k (no shared cursors, so the columns are spread over host threads
  // and the result does not depend on the thread count): for every source column j of row k (ascending j) walk
  // pattern(j) from row k downwards -- it is a subset of {k} U pattern(k) -- and merge it against column k.
  //   fn(u, t, s): target block u (in column k)  -=  L[t] * L[s]^T,  t and s in column j, s = block (k, j)
  auto for_each_op_of_column = [&](int k, auto &&fn) {
    const int64_t k0 = S.colptr[k];
    for (int64_t e = S.rowptr[k]; e < S.rowptr[k + 1]; ++e) {
      const int64_t s = S.row_blk[e], c1 = S.colptr[S.row_col[e] + 1];
      int64_t u = k0;
      for (int64_t t = s; t < c1; ++t) {
        const int i = S.rowidx[t];
        while (S.rowidx[u] != i) ++u;
        fn(u, t, s);
      }
    }
  };
  S.op_ptr.assign(S.nnzL + 1, 0);
  parallel_ranges(nb, 512, [&](int kb, int ke) {
    for (int k = kb; k < ke; ++k) for_each_op_of_column(k, [&](int64_t u, int64_t, int64_t) { S.op_ptr[u + 1]++; });
  });
  for (int64_t t = 0; t < S.nnzL; ++t) S.op_ptr[t + 1] += S.op_ptr[t];
  S.nops = S.op_ptr[S.nnzL];
  lap("update lists: count");
  // ---- schedule.  work(k) ~ block operations needed to finish column k.
  std::vector<int64_t> work(nb), sub(nb);
  for (int k = 0; k < nb; ++k) {
    const int64_t c0 = S.colptr[k], c1 = S.colptr[k + 1];
    work[k] = (S.op_ptr[c1] - S.op_ptr[c0]) + 2 * (c1 - c0);
    sub[k] = work[k];
  }
  std::vector<int> height(nb, 0);
  std::vector<int64_t> subblk(nb);                 // blocks of L in the sub-tree of k
  for (int k = 0; k < nb; ++k) subblk[k] = S.colptr[k + 1] - S.colptr[k];
  for (int k = 0; k < nb; ++k) {
    const int p = S.parent[k];
    if (p >= 0) { sub[p] += sub[k]; subblk[p] += subblk[k]; height[p] = std::max(height[p], height[k] + 1); }
    S.etree_height = std::max(S.etree_height, height[k] + 1);
  }
  // task id per column: maximal subtrees with sub <= limit become one task; above that, chains of
  // columns with a single "heavy" child continue the child's task when the work is small.
  std::vector<int> task_of(nb, -1);
  int ntask = 0;
  // subtree tasks: a column k is "light" if sub[k] <= limit.  Root of a light subtree = light column
  // whose parent is heavy (or none).
  std::vector<char> light(nb);
  // ... and fits the leaf kernel's LDS (LEAF_BLOCKS blocks of L per workgroup)
  static const int64_t leaf_blocks = std::getenv("FGO_LEAF_BLOCKS") ? std::atoll(std::getenv("FGO_LEAF_BLOCKS")) : LEAF_BLOCKS;
  for (int k = 0; k < nb; ++k) light[k] = sub[k] <= task_work_limit && subblk[k] <= leaf_blocks && sub[k] - 2 * subblk[k] <= LEAF_OPS;
  // children are numbered below parents, so a reverse sweep propagates the task id downwards
  for (int k = nb - 1; k >= 0; --k) {
    if (!light[k]) continue;
    const int p = S.parent[k];
    if (p >= 0 && light[p]) task_of[k] = task_of[p];
    else task_of[k] = ntask++;
  }
  // heavy columns: extend the task of a heavy only-child chain while the accumulated work is small
  std::vector<int64_t> task_work(ntask, 0);
  std::vector<int> task_heavy_cols(ntask, 0);
  std::vector<int> heavy_children(nb, 0), last_heavy_child(nb, -1);
  for (int k = 0; k < nb; ++k) {
    if (light[k]) continue;
    const int p = S.parent[k];
    if (p >= 0) { heavy_children[p]++; last_heavy_child[p] = k; }
  }
  for (int k = 0; k < nb; ++k) {
    if (light[k]) continue;
    int t = -1;
    if (heavy_children[k] == 1) {
      const int c = last_heavy_child[k];
      const int tc = task_of[c];
      if (task_work[tc] + work[k] <= chain_work_limit && task_heavy_cols[tc] < PANEL_MAX) t = tc;
    }
    if (t < 0) { t = ntask++; task_work.push_back(0); task_heavy_cols.push_back(0); }
    task_of[k] = t;
    task_work[t] += work[k];
    task_heavy_cols[t]++;
  }
  // levels: level(T) = 1 + max level of tasks owning children of T's columns
  std::vector<int> tlevel(ntask, 0);
  for (int k = 0; k < nb; ++k) {          // ascending: children first
    const int p = S.parent[k];
    if (p < 0) continue;
    const int tk = task_of[k], tp = task_of[p];
    if (tk != tp) tlevel[tp] = std::max(tlevel[tp], tlevel[tk] + 1);
    else tlevel[tp] = std::max(tlevel[tp], tlevel[tk]);
  }
  // NOTE: a task's level can be raised after one of its earlier columns was visited; since the level is a
  // property of the task (max over all its columns' children) and parents are visited after children,
  // a second sweep makes it consistent.
  for (int pass = 0; pass < 2; ++pass)
    for (int k = 0; k < nb; ++k) {
      const int p = S.parent[k];
      if (p < 0) continue;
      const int tk = task_of[k], tp = task_of[p];
      if (tk != tp) tlevel[tp] = std::max(tlevel[tp], tlevel[tk] + 1);
    }
  int nlevels = 0;
  for (int t = 0; t < ntask; ++t) nlevels = std::max(nlevels, tlevel[t] + 1);
  // bucket tasks by level, columns by task (ascending column order inside a task)
  std::vector<int> tcount(ntask, 0);
  for (int k = 0; k < nb; ++k) tcount[task_of[k]]++;
  std::vector<int> order(ntask), lcount(nlevels + 1, 0);
  for (int t = 0; t < ntask; ++t) lcount[tlevel[t] + 1]++;
  for (int l = 0; l < nlevels; ++l) lcount[l + 1] += lcount[l];
  S.level_ptr = lcount;
  {
    std::vector<int> fill(lcount.begin(), lcount.end() - 1);
    for (int t = 0; t < ntask; ++t) order[t] = fill[tlevel[t]]++;   // new index of task t
  }
  S.task_ptr.assign(ntask + 1, 0);
  for (int t = 0; t < ntask; ++t) S.task_ptr[order[t] + 1] = tcount[t];
  for (int t = 0; t < ntask; ++t) S.task_ptr[t + 1] += S.task_ptr[t];
  S.task_cols.resize(nb);
  {
    std::vector<int> fill(S.task_ptr.begin(), S.task_ptr.end() - 1);
    for (int k = 0; k < nb; ++k) S.task_cols[fill[order[task_of[k]]]++] = k;
  }

  lap("tasks and levels");
  // ---- fill the update lists, split [external | internal]: external sources live in other (earlier level) tasks and
  // are applied by the wide accumulate kernel, internal ones by the task's own workgroup.  Both parts in ascending
  // source column order: externals are written from the front, internals from the back and then reversed.
  S.op_a.resize(S.nops);
  S.op_b.resize(S.nops);
  S.op_mid.resize(S.nnzL);
  parallel_ranges(nb, 512, [&](int kb, int ke) {
    std::vector<int64_t> front, back;
    for (int k = kb; k < ke; ++k) {
      const int64_t k0 = S.colptr[k], k1 = S.colptr[k + 1];
      const int T = task_of[k];
      front.assign(S.op_ptr.begin() + k0, S.op_ptr.begin() + k1);
      back.assign(S.op_ptr.begin() + k0 + 1, S.op_ptr.begin() + k1 + 1);
      for_each_op_of_column(k, [&](int64_t u, int64_t t, int64_t s) {
        const int64_t o = (task_of[S.blkcol[t]] != T) ? front[u - k0]++ : --back[u - k0];
        S.op_a[o] = (int)t; S.op_b[o] = (int)s;
      });
      for (int64_t u = k0; u < k1; ++u) {
        S.op_mid[u] = front[u - k0];
        std::reverse(S.op_a.begin() + S.op_mid[u], S.op_a.begin() + S.op_ptr[u + 1]);
        std::reverse(S.op_b.begin() + S.op_mid[u], S.op_b.begin() + S.op_ptr[u + 1]);
      }
    }
  });
  // per level: targets that have external ops
  S.acc_ptr.assign(nlevels + 1, 0);
  for (int l = 0; l < nlevels; ++l) {
    for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t)
      for (int c = S.task_ptr[t]; c < S.task_ptr[t + 1]; ++c) {
        const int k = S.task_cols[c];
        for (int64_t b = S.colptr[k]; b < S.colptr[k + 1]; ++b)
          if (S.op_mid[b] > S.op_ptr[b]) S.acc_targets.push_back((int)b);
      }
    S.acc_ptr[l + 1] = (int64_t)S.acc_targets.size();
    // long source lists last: they get a whole workgroup each (hub columns, the top separators)
    static const int64_t long_ops = std::getenv("FGO_ACC_LONG") ? std::atoll(std::getenv("FGO_ACC_LONG")) : ACC_LONG_OPS;
    auto first_long = std::stable_partition(S.acc_targets.begin() + S.acc_ptr[l], S.acc_targets.end(),
                                            [&](int b) { return S.op_mid[b] - S.op_ptr[b] <= long_ops; });
    S.acc_mid.push_back((int64_t)(first_long - S.acc_targets.begin()));
  }

  lap("update lists: fill + split");
  // ---- panels
  constexpr int PM = PANEL_MAX;
  auto find_blk = [&](int row, int col) -> int {     // block id of (row, col), row > col, or -1
    const int *b = S.rowidx.data() + S.colptr[col] + 1, *e = S.rowidx.data() + S.colptr[col + 1];
    const int *p = std::lower_bound(b, e, row);
    return (p != e && *p == row) ? (int)(p - S.rowidx.data()) : -1;
  };
  // candidate levels: every task a chain of <= PM columns.  (A level of thousands of one- or two-column tasks -- the
  // landmarks of a bundle adjustment: 400k single columns -- is cheaper in the generic one-workgroup-per-task kernel
  // than in 1024-thread panel workgroups.)  Only tasks of candidate levels get panel tables.
  std::vector<char> cand(nlevels, 0);
  for (int l = 0; l < nlevels; ++l) {
    bool all = S.level_ptr[l + 1] > S.level_ptr[l];
    int maxm = 0;
    for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1] && all; ++t) {
      const int c0 = S.task_ptr[t], m = S.task_ptr[t + 1] - c0;
      maxm = std::max(maxm, m);
      all = m <= PM;
      for (int q = 0; q + 1 < m && all; ++q) all = S.parent[S.task_cols[c0 + q]] == S.task_cols[c0 + q + 1];
    }
    if (all && maxm <= 2 && S.level_ptr[l + 1] - S.level_ptr[l] > 2048) all = false;
    cand[l] = all;
  }
  S.task_panel.assign(ntask, -1);
  S.prow_ptr.assign(1, 0);
  for (int l = 0; l < nlevels; ++l)
    if (cand[l])
      for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
        const int last = S.task_cols[S.task_ptr[t + 1] - 1];
        S.task_panel[t] = S.n_panels++;
        S.panel_task.push_back(t);
        S.prow_ptr.push_back(S.prow_ptr.back() + (int)(S.colptr[last + 1] - S.colptr[last] - 1));
      }
  S.ptri_blk.assign((size_t)S.n_panels * PM * PM, -1);
  S.prow_idx.resize((size_t)S.prow_ptr.back());
  S.prow_blk.assign((size_t)S.prow_ptr.back() * PM, -1);
  std::vector<char> panel_ok((size_t)S.n_panels, 1);
  parallel_ranges(S.n_panels, 64, [&](int p0, int p1) {
    for (int pn = p0; pn < p1; ++pn) {
      const int t = S.panel_task[pn];
      const int c0 = S.task_ptr[t], m = S.task_ptr[t + 1] - c0;
      const int *cols = S.task_cols.data() + c0;
      const int last = cols[m - 1];
      int *tri = S.ptri_blk.data() + (size_t)pn * PM * PM;
      int64_t covered = 0, total = 0;
      for (int k = 0; k < m; ++k) {
        tri[k * PM + k] = (int)S.colptr[cols[k]];
        ++covered;
        for (int r = k + 1; r < m; ++r) covered += (tri[r * PM + k] = find_blk(cols[r], cols[k])) >= 0;
        total += S.colptr[cols[k] + 1] - S.colptr[cols[k]];
      }
      bool nested = true;
      int q = S.prow_ptr[pn];
      for (int64_t p = S.colptr[last] + 1; p < S.colptr[last + 1]; ++p, ++q) {
        const int i = S.rowidx[p];
        S.prow_idx[q] = i;
        int *rb = S.prow_blk.data() + (size_t)q * PM;
        bool seen = false;
        for (int k = 0; k < m; ++k) {
          const int b = (k == m - 1) ? (int)p : find_blk(i, cols[k]);
          if (b >= 0) { seen = true; ++covered; } else if (seen) nested = false;   // must be a suffix k >= start
          rb[k] = b;
        }
      }
      // every block of the panel's columns must be covered by the triangle or the rows; otherwise it is not a
      // proper supernode-like path and its level is left to the generic kernels
      panel_ok[pn] = nested && covered == total;
    }
  });
  S.level_panel.assign(nlevels, 0);
  S.pchunk_ptr.assign(nlevels + 1, 0);
  S.fchunk_ptr.assign(nlevels + 1, 0);
  S.rchunk_ptr.assign(nlevels + 1, 0);
  S.panel_chunk0.assign(S.n_panels + 1, 0);
  S.pcol_fchunk0.assign((size_t)S.n_panels * PM, 0);
  S.pcol_fchunkn.assign((size_t)S.n_panels * PM, 0);
  S.row_mid.resize(nb);
  for (int k = 0; k < nb; ++k) S.row_mid[k] = S.rowptr[k + 1];
  for (int l = 0; l < nlevels; ++l) {
    bool all = cand[l];
    for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1] && all; ++t) all = panel_ok[S.task_panel[t]];
    S.level_panel[l] = all;
    if (all)
      for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
        const int pn = S.task_panel[t];
        // panels of one level are numbered consecutively only if earlier levels hold no panel of a mixed level;
        // chunk ranges are stored per panel, so no such assumption is needed
        S.panel_chunk0[pn] = (int)S.pchunk_panel.size();
        for (int r0 = S.prow_ptr[pn]; r0 < S.prow_ptr[pn + 1]; r0 += PANEL_ROWS) {
          S.pchunk_panel.push_back(pn); S.pchunk_row0.push_back(r0);
          S.pchunk_nrows.push_back(std::min(PANEL_ROWS, S.prow_ptr[pn + 1] - r0));
        }
        for (int s0 = 0; s0 < 6 * (S.prow_ptr[pn + 1] - S.prow_ptr[pn]) + 1; s0 += 16 * ROW_SETS) {   // + 1: the right-hand side row
          S.rchunk_panel.push_back(pn); S.rchunk_s0.push_back(s0);
        }
        // forward-solve row lists of the panel's columns: [external | in-panel], external part chunked
        const int c0 = S.task_ptr[t], m = S.task_ptr[t + 1] - c0;
        for (int q = 0; q < m; ++q) {
          const int k = S.task_cols[c0 + q];
          const int64_t r0 = S.rowptr[k], r1 = S.rowptr[k + 1];
          std::vector<std::pair<int, int>> ext, in;
          for (int64_t e = r0; e < r1; ++e) {
            const bool inside = S.row_col[e] >= S.task_cols[c0] && std::binary_search(S.task_cols.begin() + c0, S.task_cols.begin() + c0 + m, S.row_col[e]);
            (inside ? in : ext).push_back({S.row_blk[e], S.row_col[e]});
          }
          int64_t w = r0;
          for (auto &x : ext) { S.row_blk[w] = x.first; S.row_col[w] = x.second; ++w; }
          S.row_mid[k] = w;
          for (auto &x : in) { S.row_blk[w] = x.first; S.row_col[w] = x.second; ++w; }
          S.pcol_fchunk0[(size_t)pn * PM + q] = (int)S.fchunk_col.size();
          for (int64_t e = r0; e < S.row_mid[k]; e += FWD_CHUNK) { S.fchunk_col.push_back(k); S.fchunk_e0.push_back(e); }
          S.pcol_fchunkn[(size_t)pn * PM + q] = (int)S.fchunk_col.size() - S.pcol_fchunk0[(size_t)pn * PM + q];
        }
      }
    S.pchunk_ptr[l + 1] = (int)S.pchunk_panel.size();
    S.fchunk_ptr[l + 1] = (int)S.fchunk_col.size();
    S.rchunk_ptr[l + 1] = (int)S.rchunk_panel.size();
  }
  // ---- leaf levels: every task a self-contained light sub-tree (contiguous columns, no external updates, at most
  // leaf_blocks blocks) -> k_chol_leaf factors it inside LDS.  Levels of one- or two-column tasks stay on the one-wave
  // generic kernel.
  S.level_leaf.assign(nlevels, 0);
  S.level_leaf_maxblk.assign(nlevels, 0);
  S.level_leaf_maxops.assign(nlevels, 0);
  for (int l = 0; l < nlevels; ++l) {
    if (S.level_panel[l] || S.level_ptr[l + 1] == S.level_ptr[l]) continue;
    bool ok = true;
    int maxm = 0;
    int64_t maxblk = 0, maxops = 0;
    for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1] && ok; ++t) {
      const int c0 = S.task_ptr[t], m = S.task_ptr[t + 1] - c0;
      maxm = std::max(maxm, m);
      const int k0 = S.task_cols[c0], k1 = S.task_cols[c0 + m - 1];
      ok = k1 - k0 == m - 1;
      const int64_t b0 = S.colptr[k0], b1 = S.colptr[k1 + 1];
      maxblk = std::max(maxblk, b1 - b0);
      maxops = std::max(maxops, S.op_ptr[b1] - S.op_ptr[b0]);
      ok = ok && b1 - b0 <= leaf_blocks && b1 - b0 < 65536 && S.op_ptr[b1] - S.op_ptr[b0] <= LEAF_OPS;
      for (int64_t u = b0; u < b1 && ok; ++u) ok = S.op_mid[u] == S.op_ptr[u];
    }
    if (ok && maxm > 2) { S.level_leaf[l] = 1; S.level_leaf_maxblk[l] = (int)maxblk; S.level_leaf_maxops[l] = (int)maxops; }
  }
  lap("panels and chunks");
}

}  // namespace fgo
