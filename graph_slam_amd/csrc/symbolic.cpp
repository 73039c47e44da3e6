// Symbolic phase of the block Cholesky (host, once per graph structure): block elimination tree,
// block column patterns of L, left-looking update lists, row lists for the triangular solves and
// the task/level schedule the device kernels follow.  Replaces what g2o does in
// BlockSolver::buildStructure + LinearSolverCSparse's symbolic decomposition ([UPSTREAM], reached
// from the reference at g2o/g2o_graph.cpp:246-249 on iteration 0 of every optimize() call).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <climits>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>
#include "fgo_internal.hpp"

namespace fgo {

void build_symbolic(const BlockGraph &g, const std::vector<int> &perm, int64_t task_work_limit, int64_t chain_work_limit, Symbolic &S, int world, bool analyse_only) {
  const int nb = g.n;
  const bool prof = std::getenv("FGO_SYM_PROFILE") != nullptr;
  auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tprev = tnow();
  auto lap = [&](const char *what) { if (prof) { const double t = tnow(); std::fprintf(stderr, "[fgo symbolic] %-28s %.1f ms\n", what, 1e3 * (t - tprev)); tprev = t; } };
  S = Symbolic();
  S.nb = nb;
  if (world < 1) world = 1;
  S.world = world;
  // ---- column patterns: pattern(k) = A(k+1:, k)  U  union over children c of pattern(c) \ {k}
  // `par` = the elimination tree of the order `pm` when the caller knows it (Liu's algorithm below) and the order is a
  // post-order: disjoint sub-trees are then contiguous column ranges whose patterns depend on nothing outside, so they are
  // computed concurrently (one mark array per worker), the columns above them serially afterwards.
  std::vector<std::vector<int>> pat(nb);
  auto pattern_pass = [&](const std::vector<int> &pm, const std::vector<int> *par) {
    S.perm = pm;
    S.iperm.assign(nb, -1);
    for (int k = 0; k < nb; ++k) S.iperm[pm[k]] = k;
    S.parent.assign(nb, -1);
    S.colptr.assign(nb + 1, 0);
    S.max_col_blocks = 0;
    std::vector<int> first_child(nb, -1), next_sib(nb, -1);
    std::vector<char> done(nb, 0);
    auto column = [&](int k, std::vector<int> &mark, std::vector<int> &tmp) {
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
      S.parent[k] = tmp.empty() ? -1 : tmp[0];
    };
    if (par && host_threads() > 1) {
      // children lists from the known tree (ascending), sub-tree sizes, and the maximal sub-trees below a size bound
      for (int k = nb - 1; k >= 0; --k) if ((*par)[k] >= 0) { next_sib[k] = first_child[(*par)[k]]; first_child[(*par)[k]] = k; }
      std::vector<int> size(nb, 1);
      for (int k = 0; k < nb; ++k) if ((*par)[k] >= 0) size[(*par)[k]] += size[k];
      const int bound = std::max(256, nb / (8 * host_threads()));
      std::vector<int> roots;
      for (int k = 0; k < nb; ++k) if (size[k] <= bound && ((*par)[k] < 0 || size[(*par)[k]] > bound)) roots.push_back(k);
      static std::atomic<uint64_t> calls{0};
      const uint64_t token = ++calls;                                  // (a pool thread's mark array is re-initialised once per call)
      parallel_ranges((int)roots.size(), 1, [&](int r0, int r1) {
        static thread_local std::vector<int> mark;
        static thread_local uint64_t mark_token = 0;
        std::vector<int> tmp;
        if (mark_token != token || (int)mark.size() != nb) { mark.assign(nb, -1); mark_token = token; }
        for (int r = r0; r < r1; ++r) {
          const int top = roots[r];
          for (int k = top - size[top] + 1; k <= top; ++k) { column(k, mark, tmp); done[k] = 1; }
          // (marks left on ancestors are column ids of THIS sub-tree: never equal to a later k of this call)
        }
      });
      std::vector<int> mark(nb, -1), tmp;
      int64_t nser = 0, wser = 0;
      for (int k = 0; k < nb; ++k) if (!done[k]) { column(k, mark, tmp); ++nser; wser += (int64_t)tmp.size(); }
      if (prof) std::fprintf(stderr, "[fgo symbolic]   pattern pass: %zu sub-trees in parallel (<= %d columns), %lld columns / %lld pattern entries serially\n", roots.size(), bound, (long long)nser, (long long)wser);
    } else {
      std::vector<int> mark(nb, -1), tmp;
      for (int k = 0; k < nb; ++k) {
        column(k, mark, tmp);
        if (!tmp.empty()) { const int p = tmp[0]; next_sib[k] = first_child[p]; first_child[p] = k; }
      }
    }
    for (int k = 0; k < nb; ++k) {
      S.colptr[k + 1] = S.colptr[k] + 1 + (int64_t)pat[k].size();
      S.max_col_blocks = std::max(S.max_col_blocks, 1 + (int)pat[k].size());
    }
  };
  {
    // Post-order the elimination tree (same fill, same tree): every sub-tree becomes a contiguous column range, so a
    // light sub-tree's blocks are one contiguous range of L and the leaf kernel can keep them in LDS under local
    // indices.  Children keep their relative order; the nested-dissection order is nearly post-ordered already.
    // The tree comes from the graph alone (Liu's algorithm: ancestors with path compression) -- no column patterns
    // needed -- so the (serial) pattern pass runs once, on the final order.
    std::vector<int> ip(nb), par(nb, -1), anc(nb, -1);
    for (int k = 0; k < nb; ++k) ip[perm[k]] = k;
    for (int k = 0; k < nb; ++k) {
      const int v = perm[k];
      for (int p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
        int r = ip[g.adj[p]];
        if (r >= k) continue;
        while (anc[r] != -1 && anc[r] != k) { const int nx = anc[r]; anc[r] = k; r = nx; }
        if (anc[r] == -1) { anc[r] = k; par[r] = k; }
      }
    }
    std::vector<int> head(nb, -1), nxt(nb, -1), post;
    for (int k = nb - 1; k >= 0; --k) if (par[k] >= 0) { nxt[k] = head[par[k]]; head[par[k]] = k; }
    post.reserve(nb);
    std::vector<int> stack;
    for (int r = 0; r < nb; ++r) {
      if (par[r] >= 0) continue;
      stack.push_back(r);
      while (!stack.empty()) {
        const int k = stack.back();
        if (head[k] >= 0) { const int c = head[k]; head[k] = nxt[c]; stack.push_back(c); }   // next unvisited child
        else { post.push_back(k); stack.pop_back(); }
      }
    }
    lap("  etree + post-order");
    std::vector<int> pm(nb), ipost(nb), par2(nb, -1);
    for (int k = 0; k < nb; ++k) { pm[k] = perm[post[k]]; ipost[post[k]] = k; }
    for (int k = 0; k < nb; ++k) if (par[k] >= 0) par2[ipost[k]] = ipost[par[k]];      // the same tree under the post-order
    pattern_pass(pm, &par2);
  }
  // ---- multi-GPU: domain decomposition of the (post-ordered) elimination tree.  The tree is cut into disjoint
  // sub-trees ("domains", dealt to the ranks by weight) and the set of their common ancestors (the "top": the upper
  // nested-dissection separators).  Columns are re-ordered [domain of rank 0 | rank 1 | ... | top] -- still a
  // topological order of the same tree, so the fill is unchanged; inside a group the post-order is kept, so a sub-tree
  // stays a contiguous column range.  A rank factors only its own columns; every update that crosses from a domain into
  // the top is summed over the ranks by one collective on the (contiguous) tail of L (fgo_api.cpp).
  S.dom_col0.assign((size_t)world + 2, nb);
  S.dom_col0[0] = 0;
  if (world > 1) {
    std::vector<double> subw(nb);
    for (int k = 0; k < nb; ++k) { const double c = (double)(S.colptr[k + 1] - S.colptr[k]); subw[k] = 0.5 * c * (c + 1); }   // updates generated by column k
    std::vector<int> head(nb, -1), nxt(nb, -1);
    double total = 0;
    for (int k = 0; k < nb; ++k) { const int p = S.parent[k]; if (p >= 0) subw[p] += subw[k]; }
    for (int k = nb - 1; k >= 0; --k) { const int p = S.parent[k]; if (p >= 0) { nxt[k] = head[p]; head[p] = k; } else total += subw[k]; }
    // open the heaviest sub-tree until there are enough of them to balance and none is too heavy.  Round 3 (tools/dist_sweep.sh,
    // profiles/r03_e_dist_decomposition.txt): a SMALLER top wins -- every column moved from the replicated, latency-bound top into
    // a domain is work divided by the ranks and bytes taken out of the collective: 2 sub-trees per rank and a cap of one fair
    // share (was 4 and a quarter) give 2.33 instead of 2.85 ms per trial at 8 ranks on cfg 2 with 4.2 instead of 20.2 MB in the
    // all-reduce, 6.9 instead of 8.3 ms on cfg 5 (8.4-23 instead of 20-35 MB)
    std::vector<std::pair<double, int>> heap;                           // (weight, root)
    for (int k = 0; k < nb; ++k) if (S.parent[k] < 0) heap.push_back({subw[k], k});
    std::make_heap(heap.begin(), heap.end());
    std::vector<char> is_top(nb, 0);
    static const double max_share = tune("dist_max_share", 1.0);   // of one rank's fair share
    while (!heap.empty()) {
      const std::pair<double, int> h = heap.front();
      static const int min_trees = (int)tune("dist_min_trees", 2);   // sub-trees per rank at least
      const bool enough = (int)heap.size() >= min_trees * world && h.first <= max_share * total / world;
      if (enough || head[h.second] < 0) break;                          // (a childless column cannot be opened)
      std::pop_heap(heap.begin(), heap.end()); heap.pop_back();
      is_top[h.second] = 1;
      for (int c = head[h.second]; c >= 0; c = nxt[c]) { heap.push_back({subw[c], c}); std::push_heap(heap.begin(), heap.end()); }
    }
    // largest first onto the least loaded rank (ties: lowest rank) -- deterministic, identical on every rank
    std::sort(heap.begin(), heap.end(), [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
    std::vector<double> load(world, 0.0);
    std::vector<int> group(nb, -1);
    for (const auto &h : heap) {
      int r = 0;
      for (int q = 1; q < world; ++q) if (load[q] < load[r]) r = q;
      load[r] += h.first;
      group[h.second] = r;
    }
    for (int k = nb - 1; k >= 0; --k) {                                 // parents before children: push the group down
      if (is_top[k]) { group[k] = world; continue; }
      if (group[k] < 0) group[k] = group[S.parent[k]];
    }
    std::vector<int> cnt((size_t)world + 2, 0);
    for (int k = 0; k < nb; ++k) cnt[group[k] + 1]++;
    for (int q = 0; q <= world; ++q) cnt[q + 1] += cnt[q];
    for (int q = 0; q <= world + 1; ++q) S.dom_col0[q] = cnt[q];
    std::vector<int> pm(nb);
    {
      std::vector<int> fill(cnt.begin(), cnt.end() - 1);
      for (int k = 0; k < nb; ++k) pm[fill[group[k]]++] = S.perm[k];
    }
    for (auto &v : pat) v.clear();
    pattern_pass(pm, nullptr);
    if (prof) {
      std::fprintf(stderr, "[fgo symbolic] domains: %zu sub-trees, top %d columns;", heap.size(), nb - S.dom_col0[world]);
      for (int q = 0; q < world; ++q) std::fprintf(stderr, " %.1f%%", 100.0 * load[q] / total);
      std::fprintf(stderr, " of the update work per rank\n");
    }
  }
  auto group_of = [&](int k) -> int {                                   // 0 .. world-1: a rank's domain, world: the top
    if (world == 1) return 0;
    return (int)(std::upper_bound(S.dom_col0.begin(), S.dom_col0.begin() + world + 1, k) - S.dom_col0.begin()) - 1;
  };
  S.nnzL = S.colptr[nb];
  S.rowidx.resize(S.nnzL);
  S.blkcol.resize(S.nnzL);
  lap("  patterns");
  parallel_ranges(nb, 2048, [&](int kb, int ke) {
    for (int k = kb; k < ke; ++k) {
      int64_t p = S.colptr[k];
      S.rowidx[p] = k; S.blkcol[p] = k; ++p;
      for (int i : pat[k]) { S.rowidx[p] = i; S.blkcol[p] = k; ++p; }
      std::vector<int>().swap(pat[k]);
    }
  });

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
  static const int64_t leaf_blocks = (int64_t)tune("leaf_blocks", LEAF_BLOCKS);
  for (int k = 0; k < nb; ++k)
    light[k] = sub[k] <= task_work_limit && subblk[k] <= leaf_blocks && sub[k] - 2 * subblk[k] <= LEAF_OPS && (world == 1 || k < S.dom_col0[world]);
  // children are numbered below parents, so a reverse sweep propagates the task id downwards
  for (int k = nb - 1; k >= 0; --k) {
    if (!light[k]) continue;
    const int p = S.parent[k];
    if (p >= 0 && light[p]) task_of[k] = task_of[p];
    else task_of[k] = ntask++;
  }
  // heavy columns: extend the task of a heavy only-child chain while the accumulated work is small
  std::vector<int64_t> task_work;
  std::vector<int> task_heavy_cols, heavy_children, last_heavy_child, tl, ch_m1, ch_m2, ch_arg;
  task_work.assign(ntask, 0);
  task_heavy_cols.assign(ntask, 0);
  // A column with several heavy children (the first column of a separator) continues the panel of its TALLEST heavy child
  // when that panel has room: a separator's last, partly filled panel and the head of the parent separator then share a
  // level instead of taking one each (the other children's updates arrive through the accumulate like any external
  // source).  FGO_MERGE_MULTI=0 restores the only-child rule.
  static const bool merge_multi = tune("merge_multi", 1) != 0;
  // WHICH child: the one whose task sits on the highest LEVEL so far (ties: the tallest sub-tree) -- continuing the panel on
  // the critical path saves a level, continuing a taller but lower-levelled one does not (by_level = 0: the tallest, as before).
  static const bool by_level = tune("merge_by_level", 1) != 0;
  heavy_children.assign(nb, 0); last_heavy_child.assign(nb, -1);
  tl.assign((size_t)ntask, 0);                                 // task levels as the sweep sees them (light sub-trees: 0)
  ch_m1.assign(nb, -1); ch_m2.assign(nb, -1); ch_arg.assign(nb, -1);   // per column: highest / second highest level among its children's tasks, a child at the highest
  auto note_child = [&](int p, int k, int lv) {                 // child k (task level lv) of p
    if (lv > ch_m1[p]) { ch_m2[p] = ch_m1[p]; ch_m1[p] = lv; ch_arg[p] = k; }
    else if (lv > ch_m2[p]) ch_m2[p] = lv;
  };
  for (int k = 0; k < nb; ++k) {
    const int p = S.parent[k];
    if (light[k]) { if (p >= 0 && !light[p]) note_child(p, k, 0); continue; }
    int t = -1, lv_new = ch_m1[k] + 1;
    if (heavy_children[k] == 1 || (merge_multi && heavy_children[k] > 1)) {
      const int c = last_heavy_child[k];
      const int tc = task_of[c];
      if (task_work[tc] + work[k] <= chain_work_limit && task_heavy_cols[tc] < PANEL_MAX && group_of(c) == group_of(k)) {
        t = tc;
        const int others = ch_arg[k] == c ? ch_m2[k] : ch_m1[k];      // (several children at the top level: m2 == m1 is noted as m2)
        lv_new = std::max(tl[tc], others + 1);
      }
    }
    if (t < 0) { t = ntask++; task_work.push_back(0); task_heavy_cols.push_back(0); tl.push_back(0); }
    task_of[k] = t;
    task_work[t] += work[k];
    task_heavy_cols[t]++;
    tl[t] = std::max(tl[t], lv_new);
    if (p >= 0) {
      heavy_children[p]++;
      const int q = last_heavy_child[p];
      const bool better = q < 0 || (by_level ? (tl[t] > tl[task_of[q]] || (tl[t] == tl[task_of[q]] && height[k] >= height[q])) : height[k] >= height[q]);
      if (better) last_heavy_child[p] = k;
      note_child(p, k, tl[t]);
    }
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
  // multi-GPU: a schedule "level" below is a SEGMENT = (dependency level, group); a rank runs the segments of its own
  // group, then -- after the collective -- those of the top, both in ascending level order
  if (world > 1) {
    std::vector<int> tgroup(ntask, 0);
    for (int k = 0; k < nb; ++k) tgroup[task_of[k]] = group_of(k);
    const int G = world + 1;
    for (int t = 0; t < ntask; ++t) tlevel[t] = tlevel[t] * G + tgroup[t];
    nlevels *= G;
    S.seg_group.resize(nlevels);
    for (int l = 0; l < nlevels; ++l) S.seg_group[l] = l % G;
  } else {
    S.seg_group.assign(nlevels, 0);
  }
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
  if (analyse_only) return;                 // S.nops, S.nnzL and the levels are known: what an ordering candidate is ranked by
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
  lap("update lists: fill");
  // per level: targets that have external ops
  // (multi-GPU: a top block's domain-sourced updates arrive through the collective; only its top-sourced external
  //  updates -- the tail of the ascending list -- are left to the accumulate kernel)
  const int top_col0 = world > 1 ? S.dom_col0[world] : nb;
  auto ext_ops = [&](int64_t b) -> int64_t {
    if (S.blkcol[b] < top_col0) return S.op_mid[b] - S.op_ptr[b];
    int64_t o = S.op_mid[b];
    while (o > S.op_ptr[b] && S.blkcol[S.op_a[o - 1]] >= top_col0) --o;
    return S.op_mid[b] - o;
  };
  S.acc_ptr.assign(nlevels + 1, 0);
  // (two passes over the columns in task order -- count, prefix, fill -- both in parallel; the order is the serial one)
  std::vector<int64_t> col_nt((size_t)nb + 1, 0);                 // targets of the q-th entry of task_cols
  parallel_ranges(nb, 1024, [&](int q0, int q1) {
    for (int q = q0; q < q1; ++q) {
      const int k = S.task_cols[q];
      int64_t n = 0;
      for (int64_t b = S.colptr[k]; b < S.colptr[k + 1]; ++b) n += ext_ops(b) > 0;
      col_nt[(size_t)q + 1] = n;
    }
  });
  for (int q = 0; q < nb; ++q) col_nt[(size_t)q + 1] += col_nt[q];
  S.acc_targets.resize((size_t)col_nt[nb]);
  parallel_ranges(nb, 1024, [&](int q0, int q1) {
    for (int q = q0; q < q1; ++q) {
      const int k = S.task_cols[q];
      int64_t w = col_nt[q];
      for (int64_t b = S.colptr[k]; b < S.colptr[k + 1]; ++b) if (ext_ops(b) > 0) S.acc_targets[(size_t)w++] = (int)b;
    }
  });
  for (int l = 0; l < nlevels; ++l) {
    S.acc_ptr[l + 1] = col_nt[S.task_ptr[S.level_ptr[l + 1]]];
    // long source lists last: they get a whole workgroup each (hub columns, the top separators)
    static const int64_t long_ops = (int64_t)tune("acc_long", ACC_LONG_OPS);
    auto first_long = std::stable_partition(S.acc_targets.begin() + S.acc_ptr[l], S.acc_targets.begin() + S.acc_ptr[l + 1],
                                            [&](int b) { return ext_ops(b) <= long_ops; });
    S.acc_mid.push_back((int64_t)(first_long - S.acc_targets.begin()));
  }

  lap("accumulate targets");
  if (tune("acc_stats", 0) != 0) {
    // per level: targets, external updates, the longest list, and how many updates come from the level directly below
    // (tools/acc_stats.py; profiles/NOTES.md "staged accumulate")
    for (int l = 0; l < nlevels; ++l) {
      int64_t nt = S.acc_ptr[l + 1] - S.acc_ptr[l], ops = 0, mx = 0, late = 0, mxlate = 0;
      for (int64_t q = S.acc_ptr[l]; q < S.acc_ptr[l + 1]; ++q) {
        const int b = S.acc_targets[q];
        int64_t n = 0, nl = 0;
        for (int64_t o = S.op_mid[b] - ext_ops(b); o < S.op_mid[b]; ++o) {
          ++n;
          if (tlevel[task_of[S.blkcol[S.op_a[o]]]] == l - 1) ++nl;
        }
        ops += n; late += nl; mx = std::max(mx, n); mxlate = std::max(mxlate, nl);
      }
      // distinct source blocks of the level: what it has to fetch at least once
      std::vector<int> srcs;
      for (int64_t q = S.acc_ptr[l]; q < S.acc_ptr[l + 1]; ++q) {
        const int b = S.acc_targets[q];
        for (int64_t o = S.op_mid[b] - ext_ops(b); o < S.op_mid[b]; ++o) { srcs.push_back(S.op_a[o]); srcs.push_back(S.op_b[o]); }
      }
      std::sort(srcs.begin(), srcs.end());
      const int64_t distinct = (int64_t)(std::unique(srcs.begin(), srcs.end()) - srcs.begin());
      {   // columns per task: how full the panels of the level are
        int hist[4] = {0, 0, 0, 0};      // 1-4, 5-8, 9-12, 13-16(+)
        int64_t cols = 0;
        for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) { const int m = S.task_ptr[t + 1] - S.task_ptr[t]; cols += m; hist[std::min(3, (m - 1) / 4)]++; }
        std::fprintf(stderr, "[acc] level %d columns %lld, tasks with 1-4 / 5-8 / 9-12 / 13+ columns: %d %d %d %d\n", l, (long long)cols, hist[0], hist[1], hist[2], hist[3]);
      }
      std::fprintf(stderr, "[acc] level %d tasks %d targets %lld updates %lld longest list %lld from the level below %lld (longest %lld) distinct sources %lld (%.1f MB)\n", l,
                   S.level_ptr[l + 1] - S.level_ptr[l], (long long)nt, (long long)ops, (long long)mx, (long long)late, (long long)mxlate,
                   (long long)distinct, 288e-6 * (double)distinct);
    }
  }

  // ---- column-group accumulate lists.  For a column k: T = its blocks with external updates (in block order), cut into
  // groups of ACC2_G; the external sources of k are the entries (b = block (k, j), j) of its row list whose column j lies
  // in another task (and, distributed, in the top when k is a top column: the domains' part arrives by collective).
  // For every (group, j) with at least one target row in pattern(j) one entry is emitted.
  {
    static const bool use_acc2 = tune("acc_v1", 0) == 0;
    S.g2_lvl.assign(nlevels + 1, 0);
    S.g2_ptr.assign(1, 0);
    if (use_acc2) {
      // pass 1 (parallel over chunks of 64 columns of a level): groups and entries into chunk-local buffers; pass 2 after
      // ALL levels: one allocation, offsets by a (cheap, serial) prefix pass, the copies in parallel
      struct ChunkOut { std::vector<int> tgt, b, a; std::vector<int64_t> ptr; };
      // Only the very wide levels take this form: there it saves a third of the instructions and half of the vector
      // loads per update (cfg 5: factor sweep 26.6 -> 23.7 ms).  Narrower levels are latency-bound on the number of
      // dependent steps per wave, and the union of the source columns over ten targets is longer than one target's
      // own list (measured on cfg 2: k_chol_acc2<8> 28 us vs 15 us for the gather form at the top, 45 vs 32-55 us in
      // the middle), so they keep the gather lists.
      static const int64_t g2_min = (int64_t)tune("acc2_min", 8000);
      constexpr int CH = 64;
      std::vector<std::vector<ChunkOut>> all((size_t)nlevels);
      for (int l = 0; l < nlevels; ++l) {
        if (S.acc_ptr[l + 1] - S.acc_ptr[l] < (int64_t)ACC2_G * g2_min) continue;
        const int c0 = S.task_ptr[S.level_ptr[l]], c1 = S.task_ptr[S.level_ptr[l + 1]];
        std::vector<ChunkOut> &outs = all[(size_t)l];
        outs.resize((size_t)((c1 - c0 + CH - 1) / CH));
        parallel_ranges(c1 - c0, CH, [&](int qb, int qe) {
          std::vector<int> T;
          ChunkOut &o = outs[(size_t)(qb / CH)];
          for (int q = qb; q < qe; ++q) {
            const int k = S.task_cols[c0 + q];
            T.clear();
            for (int64_t u = S.colptr[k]; u < S.colptr[k + 1]; ++u) if (ext_ops(u) > 0) T.push_back((int)u);
            if (T.empty()) continue;
            const int ng = ((int)T.size() + ACC2_G - 1) / ACC2_G;
            const size_t tg0 = o.tgt.size();
            o.tgt.resize(tg0 + (size_t)ng * ACC2_G, -1);
            for (size_t x = 0; x < T.size(); ++x) o.tgt[tg0 + x] = T[x];
            // entries per group, ascending source column: a 10-way merge of the targets' (already filled, ascending) external
            // update lists -- op_b = block (k, j) is the same for all targets of a source column and ascends with j
            for (int gq = 0; gq < ng; ++gq) {
              const int t0 = gq * ACC2_G, t1 = std::min<int>((gq + 1) * ACC2_G, (int)T.size());
              int64_t head[ACC2_G], end[ACC2_G];
              for (int x = t0; x < t1; ++x) { end[x - t0] = S.op_mid[T[x]]; head[x - t0] = end[x - t0] - ext_ops(T[x]); }
              while (true) {
                int sb = INT32_MAX;
                for (int x = 0; x < t1 - t0; ++x) if (head[x] < end[x]) sb = std::min(sb, S.op_b[head[x]]);
                if (sb == INT32_MAX) break;
                int a[ACC2_G];
                for (int x = 0; x < ACC2_G; ++x) a[x] = -1;
                for (int x = 0; x < t1 - t0; ++x) if (head[x] < end[x] && S.op_b[head[x]] == sb) { a[x] = S.op_a[head[x]]; ++head[x]; }
                o.b.push_back(sb);
                o.a.insert(o.a.end(), a, a + ACC2_G);
              }
              o.ptr.push_back((int64_t)o.b.size());
            }
          }
        });
      }
      // pass 2
      std::vector<std::pair<int, int>> chunks;                       // (level, chunk)
      std::vector<int64_t> g0(1, 0), e0(1, 0);
      for (int l = 0; l < nlevels; ++l) {
        for (size_t q = 0; q < all[(size_t)l].size(); ++q) {
          chunks.push_back({l, (int)q});
          g0.push_back(g0.back() + (int64_t)all[(size_t)l][q].ptr.size());
          e0.push_back(e0.back() + (int64_t)all[(size_t)l][q].b.size());
        }
        S.g2_lvl[l + 1] = g0.back();
      }
      S.g2_tgt.resize((size_t)g0.back() * ACC2_G);
      S.g2_ptr.resize((size_t)g0.back() + 1);
      S.g2_b.resize((size_t)e0.back());
      S.g2_a.resize((size_t)e0.back() * ACC2_G);
      parallel_ranges((int)chunks.size(), 4, [&](int xb, int xe) {
        for (int x = xb; x < xe; ++x) {
          const ChunkOut &o = all[(size_t)chunks[x].first][(size_t)chunks[x].second];
          if (o.tgt.empty()) continue;
          std::copy(o.tgt.begin(), o.tgt.end(), S.g2_tgt.begin() + g0[x] * ACC2_G);
          std::copy(o.b.begin(), o.b.end(), S.g2_b.begin() + e0[x]);
          std::copy(o.a.begin(), o.a.end(), S.g2_a.begin() + e0[x] * ACC2_G);
          for (size_t y = 0; y < o.ptr.size(); ++y) S.g2_ptr[(size_t)g0[x] + 1 + y] = e0[x] + o.ptr[y];
        }
      });
    }
  }
  lap("column-group lists (acc2)");
  // ---- panels
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
      all = m <= PANEL_MAX;
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
  S.ptri_blk.assign(S.tri_off(S.n_panels), -1);
  S.prow_idx.resize((size_t)S.prow_ptr.back());
  S.prow_blk.assign(S.row_off(S.prow_ptr.back()), -1);
  std::vector<char> panel_ok((size_t)S.n_panels, 1);
  parallel_ranges(S.n_panels, 64, [&](int p0, int p1) {
    for (int pn = p0; pn < p1; ++pn) {
      const int t = S.panel_task[pn];
      const int c0 = S.task_ptr[t], m = S.task_ptr[t + 1] - c0;
      const int *cols = S.task_cols.data() + c0;
      const int last = cols[m - 1];
      constexpr int PM = PANEL_MAX;
      int *tri = S.ptri_blk.data() + S.tri_off(pn);
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
        int *rb = S.prow_blk.data() + S.row_off(q);
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
  std::vector<int> tlevel_new((size_t)ntask);
  for (int l = 0; l < nlevels; ++l) for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) tlevel_new[t] = l;
  S.pchunk_ptr.assign(nlevels + 1, 0);
  S.fchunk_ptr.assign(nlevels + 1, 0);
  S.rchunk_ptr.assign(nlevels + 1, 0);
  S.panel_chunk0.assign(S.n_panels + 1, 0);
  S.pcol_fchunk0.assign(S.col_off(S.n_panels), 0);
  S.pcol_fchunkn.assign(S.col_off(S.n_panels), 0);
  S.row_mid.resize(nb);
  for (int k = 0; k < nb; ++k) S.row_mid[k] = S.rowptr[k + 1];
  for (int l = 0; l < nlevels; ++l) {
    bool all = cand[l];
    for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1] && all; ++t) all = panel_ok[S.task_panel[t]];
    S.level_panel[l] = all;
  }
  // forward-solve row lists of the panels' columns: [external | in-panel] (disjoint ranges of row_blk / row_col: in parallel)
  parallel_ranges(ntask, 16, [&](int tb, int te) {
    std::vector<std::pair<int, int>> ext, in;
    for (int t = tb; t < te; ++t) {
      if (S.task_panel[t] < 0 || !S.level_panel[tlevel_new[t]]) continue;
      const int c0 = S.task_ptr[t], m = S.task_ptr[t + 1] - c0;
      for (int q = 0; q < m; ++q) {
        const int k = S.task_cols[c0 + q];
        const int64_t r0 = S.rowptr[k], r1 = S.rowptr[k + 1];
        ext.clear(); in.clear();
        for (int64_t e = r0; e < r1; ++e) {
          const bool inside = S.row_col[e] >= S.task_cols[c0] && std::binary_search(S.task_cols.begin() + c0, S.task_cols.begin() + c0 + m, S.row_col[e]);
          (inside ? in : ext).push_back({S.row_blk[e], S.row_col[e]});
        }
        int64_t w = r0;
        for (auto &x : ext) { S.row_blk[w] = x.first; S.row_col[w] = x.second; ++w; }
        S.row_mid[k] = w;
        for (auto &x : in) { S.row_blk[w] = x.first; S.row_col[w] = x.second; ++w; }
      }
    }
  });
  for (int l = 0; l < nlevels; ++l) {
    const bool all = S.level_panel[l];
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
          const int64_t r0 = S.rowptr[k];
          S.pcol_fchunk0[S.col_off(pn) + q] = (int)S.fchunk_col.size();
          for (int64_t e = r0; e < S.row_mid[k]; e += FWD_CHUNK) { S.fchunk_col.push_back(k); S.fchunk_e0.push_back(e); }
          S.pcol_fchunkn[S.col_off(pn) + q] = (int)S.fchunk_col.size() - S.pcol_fchunk0[S.col_off(pn) + q];
        }
      }
    S.pchunk_ptr[l + 1] = (int)S.pchunk_panel.size();
    S.fchunk_ptr[l + 1] = (int)S.fchunk_col.size();
    S.rchunk_ptr[l + 1] = (int)S.rchunk_panel.size();
  }

  lap("panels, chunks");
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
  lap("leaf levels");
  // ---- riders.  On the narrow top levels the triangle kernel is a pivot chain (~26 us) on a handful of CUs while the rest of
  // the chip idles, and the accumulate launches of those levels keep the whole chip gathering for 10-60 us each.  Only the
  // updates whose sources lie in the level directly below a target have to wait for it (10-30 % of a top target's list); the
  // others are applied EARLY by spare workgroups ("riders") of the k_panel_tri launch of an earlier level l: a target of
  // level lt > l takes its updates with source level <= l - 1 there.  Per target the external list is re-ordered by source
  // level (stable: ascending column within a level) so that what a slot applies is a contiguous range; partial values live
  // in L (nobody else touches a block of a later level) and the target's own accumulate launch continues from
  // acc_start.  Slots are filled earliest-deadline-first (the next level's targets first) against a capacity model
  // (a 16-wave rider workgroup = four items, cost t0 + tb * batches of 80 updates; budget = free CUs x window).
  // The summation order of a target differs from the unsplit list, so L is not bit-identical to FGO_RIDE=0 -- it is
  // deterministic, and partial re-factorisations (task_dirty) repeat exactly the same segments.
  S.ride_ptr.assign(2 * nlevels + 1, 0);
  {
    static const int ride_on = (int)tune("ride", 1);
    static const double t0 = tune("ride_t0", 3.0);     // us per item (index + row round trips, combine)
    static const double tb = tune("ride_tb", 1.2);     // us per batch of 80 updates
    static const double win = tune("ride_win", 22.0);  // us of a triangle launch that riders may fill
    static const int64_t cap_ops = (int64_t)tune("ride_ops", 230000);   // and at most this many updates (gather throughput; round 4, 27-level schedule of cfg 2: 100 / 150 / 200 / 250 / 330 / 450 k -> sweep 3.009 / 2.966 / 2.946 / 2.941 / 3.003 / 3.015 ms)
    static const int ride_min = (int)tune("ride_min", 40);    // smallest item worth a half workgroup (a target's last chance: the slot below its level)
    static const int ride_max = (int)tune("ride_max", 480);     // largest item (the rest waits for a later slot or the level's own launch)
    static const int ride_hub = (int)tune("ride_hub", 4096);    // early updates from which a target is a hub (pieces into scratch blocks)
    static const int hub_force = (int)tune("ride_hub_force", 1); // hub targets ride in full in the slot below their level
    static const int ride_min2 = (int)tune("ride_min2", 120);   // ... in earlier slots: wait until more has gathered
    const int n_cu = (int)tune("ride_cus", S.cus);
    // Distributed mode: levels are (dependency level, group) segments and only the TOP (group == world, replicated on every
    // rank) takes riders -- its blocks' lists are [domain-sourced (arrive by collective) | top-sourced], ext_ops() is the
    // top-sourced tail, and their value so far always sits in L.  `dl` = dependency level of a segment.
    const int G = world > 1 ? world + 1 : 1;
    auto dl = [&](int l) { return l / G; };
    auto in_scope = [&](int l) { return world == 1 || S.seg_group[l] == world; };
    std::vector<char> slot(nlevels, 0);
    int first_slot = nlevels;
    for (int l = 0; l < nlevels; ++l) {
      const int nt = S.level_ptr[l + 1] - S.level_ptr[l];
      slot[l] = in_scope(l) && S.level_panel[l] && nt > 0 && nt <= tri_wide_panels(S.cus) && nt < n_cu;
      if (slot[l] && first_slot == nlevels) first_slot = l;
    }
    if (ride_on && !std::getenv("FGO_NO_PANELS") && first_slot + 1 < nlevels) {
      auto is_cand = [&](int lt) { return lt > first_slot && in_scope(lt) && S.g2_lvl[lt + 1] == S.g2_lvl[lt] && S.acc_ptr[lt + 1] > S.acc_ptr[lt]; };
      // 1. external lists of the candidate targets by source level (counting sort, stable: ascending column within a level)
      std::vector<int> col_level((size_t)nb);
      for (int k = 0; k < nb; ++k) col_level[k] = tlevel[task_of[k]] / G;
      std::vector<int> early(S.acc_targets.size(), 0);      // updates of a target that can ride at all: source level <= its level - 2
      auto lvl_of = [&](int blk) { return col_level[S.blkcol[blk]]; };
      for (int lt = first_slot + 1; lt < nlevels; ++lt) {
        if (!is_cand(lt)) continue;
        const int64_t q0 = S.acc_ptr[lt];
        parallel_ranges((int)(S.acc_ptr[lt + 1] - q0), 64, [&](int xb, int xe) {
          std::vector<int> lv, ta, tb2, cnt((size_t)nlevels + 1);
          for (int x = xb; x < xe; ++x) {
            const int b = S.acc_targets[q0 + x];
            const int64_t o1 = S.op_mid[b], o0 = o1 - ext_ops(b);
            const int n = (int)(o1 - o0);
            lv.resize((size_t)n);
            bool sorted = true;
            int ne = 0;
            for (int i = 0; i < n; ++i) { lv[i] = lvl_of(S.op_a[o0 + i]); if (i > 0 && lv[i] < lv[i - 1]) sorted = false; ne += lv[i] <= dl(lt) - 2; }
            early[q0 + x] = ne;
            if (sorted) continue;
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int i = 0; i < n; ++i) cnt[(size_t)lv[i] + 1]++;
            for (int l = 0; l < nlevels; ++l) cnt[(size_t)l + 1] += cnt[l];
            ta.resize((size_t)n); tb2.resize((size_t)n);
            for (int i = 0; i < n; ++i) { const int w = cnt[lv[i]]++; ta[w] = S.op_a[o0 + i]; tb2[w] = S.op_b[o0 + i]; }
            std::copy(ta.begin(), ta.end(), S.op_a.begin() + o0);
            std::copy(tb2.begin(), tb2.end(), S.op_b.begin() + o0);
          }
        });
      }
      lap("riders: lists by source level");
      // 2. earliest deadline first over the slots.  A level gives two: its triangle launch (16-wave workgroups, four items each)
      // and its row launch (one-wave workgroups, one item each: small items only, the launch is ~11 us long)
      static const double win2 = tune("ride_win2", 0.0);   // 0: off -- measured neutral (cfg 2: factor sweep 3.303 with, 3.309 ms without)
      static const double tb2 = tune("ride_tb2", 1.0);     // us per batch of 20 updates
      static const int64_t cap_ops2 = (int64_t)tune("ride_ops2", 90000);
      static const int max2 = (int)tune("ride_max2", 100);       // largest item of a row launch
      // (the targets of a level that can ride at all, in target order: most lists are too short, and `cur` only grows)
      std::vector<std::vector<int64_t>> rideable((size_t)nlevels);
      for (int lt = first_slot + 1; lt < nlevels; ++lt)
        if (is_cand(lt))
          for (int64_t q = S.acc_ptr[lt]; q < S.acc_ptr[lt + 1]; ++q)
            if (early[q] >= ride_min) rideable[(size_t)lt].push_back(q);
      std::vector<int> cur(S.acc_targets.size(), 0);       // updates of a target already given to riders
      std::vector<std::pair<int64_t, int>> hub_piece;      // (target position, scratch block) per piece of a hub target, in schedule order
      std::vector<int> hub_skip(S.acc_targets.size(), 0);  // hub targets: entries at the end of the ridden part that the own launch does NOT skip (the piece updates)
      int64_t ridden = 0, total = 0;
      S.ride_ptr.assign(2 * nlevels + 1, 0);
      for (int l = 0; l < nlevels; ++l) {
        for (int sub = 0; sub < 2; ++sub) {
          S.ride_ptr[2 * l + sub + 1] = (int)S.ride_items.size();
          if (l < first_slot || l + 1 >= nlevels || !slot[l]) continue;
          const int nt = S.level_ptr[l + 1] - S.level_ptr[l];
          const int busy = sub == 0 ? nt : (S.rchunk_ptr[l + 1] - S.rchunk_ptr[l] + 15) / 16;
          if (sub == 1 && (win2 <= 0 || S.rchunk_ptr[l + 1] == S.rchunk_ptr[l])) continue;
          double budget = (double)(n_cu - busy) * (sub == 0 ? win : win2);
          int64_t ops_left = (sub == 0 ? cap_ops : cap_ops2) * (n_cu - busy) / n_cu;
          const size_t first_item = S.ride_items.size();
          for (int lt = l + 1; lt < nlevels; ++lt) {
            if (!is_cand(lt)) continue;
            // (the slot directly below a level is the last chance of that level's hub targets: they ride whatever the slot holds already)
            const bool last_chance = hub_force && sub == 0 && world == 1 && dl(lt) == dl(l) + 1;
            if (!(budget > 0 && ops_left > 0) && !last_chance) break;
            for (const int64_t q : rideable[(size_t)lt]) {
              const bool room = budget > 0 && ops_left > 0;
              if (!room && !last_chance) break;
              if (!room && early[q] < ride_hub) continue;
              if (early[q] - cur[q] < ride_min) continue;
              const int b = S.acc_targets[q];
              const int64_t o1 = S.op_mid[b], o0 = o1 - ext_ops(b);
              // updates with source level <= l - 1: a prefix of the (level-sorted) list
              int64_t lo = o0 + cur[q], hi = o0 + early[q];
              while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (lvl_of(S.op_a[mid]) <= dl(l) - 1) lo = mid + 1; else hi = mid; }
              int64_t avail = lo - (o0 + cur[q]);
              const int nmin = dl(lt) == dl(l) + 1 ? ride_min : ride_min2;
              // A HUB target (a plane or place seen from thousands of keyframes: tens of thousands of early updates) would
              // leave most of its list to ONE workgroup of its level's own launch (cfg 4: 54 374 updates, 625 us).  Its early
              // updates are cut into pieces that several rider workgroups of the same slot sum into SCRATCH blocks behind L;
              // the target's own list then holds one update per piece, (scratch block) x (identity block)^T.
              if (sub == 1 && early[q] >= ride_hub) continue;            // (hubs ride in the triangle launches only)
              const bool hub = sub == 0 && world == 1 && early[q] >= ride_hub;
              if (avail < nmin) continue;
              // What a hub target does not get rid of here, ONE workgroup of its level's own launch walks through at ~1.5 us per
              // 80 updates while the chip idles (100 k poses with a dozen places revisited 1 000 times: 510 k updates left to 28
              // workgroups, 359 us); as pieces they cost the triangle launch ~1 us per 15 k.
              const bool force = hub && last_chance;
              if (!room && !force) continue;
              while (avail > 0 && (force || (budget > 0 && ops_left > 0))) {
                int64_t n = force ? avail : std::min(avail, ops_left);
                n = std::min<int64_t>(n, sub == 1 ? max2 : ride_max);   // (an item is one quarter workgroup's serial work)
                if (n < std::min(nmin, max2)) break;
                if (hub) {
                  S.ride_items.push_back(RideItem{(int)(S.nnzL + 2 + S.n_scratch), order[task_of[S.blkcol[b]]], (long long)(o0 + cur[q]), (int)n, 2});
                  hub_piece.push_back({q, S.n_scratch});
                  ++S.n_scratch;
                } else {
                  S.ride_items.push_back(RideItem{b, order[task_of[S.blkcol[b]]], (long long)(o0 + cur[q]), (int)n, (cur[q] == 0 && world == 1) ? 1 : 0});
                }
                cur[q] += (int)n;
                avail -= n;
                ops_left -= n;
                budget -= sub == 0 ? 0.25 * (t0 + tb * (double)((n + 79) / 80)) : (t0 + tb2 * (double)((n + 19) / 20)) / 16.0;
                if (!hub) break;                                         // an ordinary target: one item per slot (it writes its own block)
              }
            }
          }
          // The rider workgroups of a launch take XCD-contiguous ranges of the items (kernels.hip xcd_contiguous: workgroup b runs
          // on XCD b & 7 and takes range element (b & 7) * q + ... ): the items stay in target order ACROSS the eight ranges --
          // neighbouring targets share their source blocks, which then hit in that XCD's L2 (sweep traffic 5.41 -> 4.67 GB) --
          // and are sorted heavy-first INSIDE a range, so that the long items of an XCD are dispatched first and equal
          // neighbours share a workgroup.
          {
            const int n_it = (int)(S.ride_items.size() - first_item);
            const int per = sub == 0 ? 4 : 1;                                  // items per rider workgroup (kernels.hip RIDE_PER_WG)
            const int nwg = (n_it + per - 1) / per, q8 = nwg >> 3, r8 = nwg & 7;
            for (int x = 0; x < 8; ++x) {
              const int w0 = x * q8 + std::min(x, r8), w1 = w0 + q8 + (x < r8 ? 1 : 0);
              const int i0 = std::min(n_it, w0 * per), i1 = std::min(n_it, w1 * per);
              std::stable_sort(S.ride_items.begin() + first_item + i0, S.ride_items.begin() + first_item + i1, [](const RideItem &u, const RideItem &v) { return u.n > v.n; });
            }
          }
          S.ride_ptr[2 * l + sub + 1] = (int)S.ride_items.size();
        }
      }
      lap("riders: schedule");
      // hub targets: the ridden part of the list moves to a copy behind the lists (the rider items read it there), and the
      // last entries of its place become one update per piece: block (nnzL + 2 + scratch) times the identity block nnzL + 1
      if (!hub_piece.empty()) {
        std::vector<int> npiece(S.acc_targets.size(), 0);
        for (const auto &hp : hub_piece) npiece[(size_t)hp.first]++;
        std::vector<int64_t> copy_at(S.acc_targets.size(), -1);
        int64_t extra = 0;
        for (size_t q = 0; q < npiece.size(); ++q) if (npiece[q] > 0) { copy_at[q] = (int64_t)S.op_a.size() + extra; extra += cur[q]; }
        const size_t base = S.op_a.size();
        S.op_a.resize(base + (size_t)extra); S.op_b.resize(base + (size_t)extra);
        for (size_t q = 0; q < npiece.size(); ++q) {
          if (npiece[q] == 0) continue;
          const int b = S.acc_targets[q];
          const int64_t o0 = S.op_mid[b] - ext_ops(b);
          std::copy(S.op_a.begin() + o0, S.op_a.begin() + o0 + cur[q], S.op_a.begin() + copy_at[q]);
          std::copy(S.op_b.begin() + o0, S.op_b.begin() + o0 + cur[q], S.op_b.begin() + copy_at[q]);
        }
        // the items of hub targets: re-point their op ranges into the copies
        {
          std::vector<int64_t> q_of_scratch((size_t)S.n_scratch, -1);
          for (const auto &hp : hub_piece) q_of_scratch[(size_t)hp.second] = hp.first;
          for (RideItem &it : S.ride_items) {
            if (it.first != 2) continue;
            const int64_t q = q_of_scratch[(size_t)(it.t - (int)(S.nnzL + 2))];
            const int b = S.acc_targets[(size_t)q];
            const int64_t o0 = S.op_mid[b] - ext_ops(b);
            it.o0 = copy_at[(size_t)q] + (it.o0 - o0);
          }
        }
        // one (scratch, identity) update per piece at the end of the ridden part; `cur` shrinks to what the own launch skips
        std::vector<int> placed(S.acc_targets.size(), 0);
        for (const auto &hp : hub_piece) {
          const size_t q = (size_t)hp.first;
          const int b = S.acc_targets[q];
          const int64_t o0 = S.op_mid[b] - ext_ops(b);
          const int64_t at = o0 + cur[q] - npiece[q] + placed[q]++;
          S.op_a[at] = (int)(S.nnzL + 2 + hp.second);
          S.op_b[at] = (int)(S.nnzL + 1);
        }
        for (size_t q = 0; q < npiece.size(); ++q) if (npiece[q] > 0) hub_skip[q] = npiece[q];
      }
      // 3. what is left to the levels' own accumulate launches; short / long split by the REMAINING list
      if (!S.ride_items.empty()) {
        S.acc_start.assign(S.acc_targets.size(), -1);
        static const int64_t long_ops = (int64_t)tune("acc_long", ACC_LONG_OPS);
        for (int lt = first_slot + 1; lt < nlevels; ++lt) {
          if (!is_cand(lt)) continue;
          std::vector<std::pair<int, int64_t>> tg;           // (block, start or -1)
          for (int64_t q = S.acc_ptr[lt]; q < S.acc_ptr[lt + 1]; ++q) {
            const int b = S.acc_targets[q];
            total += ext_ops(b); ridden += cur[q];
            // (a hub target starts from H like a target without riders -- its pieces sit in scratch blocks --, unless it is a block
            //  of the distributed top, whose value is in L anyway: encoded as start | 1 << 62)
            const int64_t st = S.op_mid[b] - ext_ops(b) + cur[q] - hub_skip[q];
            tg.push_back({b, cur[q] > 0 ? (hub_skip[q] > 0 && world == 1 ? (st | ((int64_t)1 << 62)) : st) : (int64_t)-1});
          }
          // (task order first -- the level's list arrives as [short | long] of the full lists, two runs --: partial sweeps launch the
          //  index range that covers their dirty tasks, fgo_structure.cpp "tk_*")
          std::stable_sort(tg.begin(), tg.end(), [&](const std::pair<int, int64_t> &u, const std::pair<int, int64_t> &v) {
            return order[task_of[S.blkcol[u.first]]] < order[task_of[S.blkcol[v.first]]]; });
          auto first_long = std::stable_partition(tg.begin(), tg.end(), [&](const std::pair<int, int64_t> &u) {
            return (u.second < 0 ? ext_ops(u.first) : S.op_mid[u.first] - (u.second & ~((int64_t)1 << 62))) <= long_ops; });
          S.acc_mid[lt] = S.acc_ptr[lt] + (int64_t)(first_long - tg.begin());
          for (size_t x = 0; x < tg.size(); ++x) { S.acc_targets[S.acc_ptr[lt] + x] = tg[x].first; S.acc_start[S.acc_ptr[lt] + x] = tg[x].second; }
        }
        if (tune("ride_stats", 0) != 0) {
          std::fprintf(stderr, "[ride] %zu items, %lld of %lld updates of the candidate levels ride\n", S.ride_items.size(), (long long)ridden, (long long)total);
          for (int l = 0; l < nlevels; ++l)
            if (S.ride_ptr[2 * l + 2] > S.ride_ptr[2 * l]) {
              int64_t ops[2] = {0, 0};
              for (int sub = 0; sub < 2; ++sub)
                for (int i = S.ride_ptr[2 * l + sub]; i < S.ride_ptr[2 * l + sub + 1]; ++i) ops[sub] += S.ride_items[i].n;
              std::fprintf(stderr, "[ride] slot %d: triangle launch %d items, %lld updates; row launch %d items, %lld updates\n", l, S.ride_ptr[2 * l + 1] - S.ride_ptr[2 * l],
                           (long long)ops[0], S.ride_ptr[2 * l + 2] - S.ride_ptr[2 * l + 1], (long long)ops[1]);
            }
          for (int lt = first_slot + 1; lt < nlevels; ++lt) {
            int64_t tot = 0, left = 0;
            for (int64_t q = S.acc_ptr[lt]; q < S.acc_ptr[lt + 1]; ++q) {
              const int b = S.acc_targets[q];
              tot += ext_ops(b); left += S.acc_start[q] < 0 ? ext_ops(b) : S.op_mid[b] - (S.acc_start[q] & ~((int64_t)1 << 62));
            }
            std::fprintf(stderr, "[ride] level %d: %lld targets (%lld long), %lld updates, %lld left to its own launch\n", lt, (long long)(S.acc_ptr[lt + 1] - S.acc_ptr[lt]),
                         (long long)(S.acc_ptr[lt + 1] - S.acc_mid[lt]), (long long)tot, (long long)left);
          }
        }
      }
    }
  }
  lap("riders");
}

}  // namespace fgo
