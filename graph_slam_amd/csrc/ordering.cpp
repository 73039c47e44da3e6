// Fill-reducing ordering for the GPU block Cholesky: nested dissection (level-structure bisection)
// on top, exact minimum degree with halo inside the leaves.
//
// Why not AMD like g2o's LinearSolverCSparse (selected by the reference at
// g2o/g2o_graph.cpp:30-31,72-74)?  A pose graph is chain-like; minimum-degree orderings give tall,
// thin elimination trees (a dependency chain tens of thousands of columns long).  Nested
// dissection gives thousands of independent leaf sub-trees plus a short separator tree, which is what a
// 256-CU device needs.  The ordering changes fill and speed only, never the solution.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include "fgo_internal.hpp"

namespace fgo {
namespace {

struct ND {
  const BlockGraph &g;
  int leaf;
  std::vector<int> region;     // current region label of each vertex (-1 = already ordered)
  std::vector<int> lvl;        // BFS level scratch
  std::vector<int> queue;
  std::vector<int> out;        // elimination order
  std::vector<int> local;      // scratch: global -> local index for leaf MD
  int next_region = 1;

  ND(const BlockGraph &gg, int lf) : g(gg), leaf(lf), region(gg.n, 0), lvl(gg.n, -1), local(gg.n, -1) {
    queue.reserve(gg.n); out.reserve(gg.n);
  }

  // BFS inside region r from s; fills lvl for reached vertices, returns them in `queue` order.
  int bfs(int s, int r, std::vector<int> &order) {
    order.clear();
    order.push_back(s); lvl[s] = 0;
    size_t head = 0;
    int maxl = 0;
    while (head < order.size()) {
      int v = order[head++];
      for (int p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
        int u = g.adj[p];
        if (region[u] != r || lvl[u] >= 0) continue;
        lvl[u] = lvl[v] + 1; maxl = std::max(maxl, lvl[u]);
        order.push_back(u);
      }
    }
    return maxl;
  }
  void clear_lvl(const std::vector<int> &vs) { for (int v : vs) lvl[v] = -1; }

  // exact minimum degree on a small vertex set with halo (neighbours outside the set count towards
  // the degree and receive fill, but are never eliminated here)
  void leaf_md(const std::vector<int> &vs) {
    const int m = (int)vs.size();
    // tiny sets need no search; very large separators end up (nearly) dense whatever the order
    if (m <= 2 || m > 384) { for (int v : vs) { out.push_back(v); region[v] = -1; } return; }
    std::vector<int> ext;   // halo vertices: not yet ordered, outside the set
    for (int i = 0; i < m; ++i) local[vs[i]] = i;
    for (int i = 0; i < m; ++i)
      for (int p = g.xadj[vs[i]]; p < g.xadj[vs[i] + 1]; ++p) {
        int u = g.adj[p];
        if (region[u] == -1) continue;            // eliminated earlier: cannot receive fill
        if (local[u] < 0) { local[u] = m + (int)ext.size(); ext.push_back(u); }
      }
    const int tot = m + (int)ext.size(), W = (tot + 63) / 64;
    std::vector<uint64_t> rows((size_t)tot * W, 0);
    auto setb = [&](int a, int b) { rows[(size_t)a * W + (b >> 6)] |= 1ull << (b & 63); };
    for (int i = 0; i < m; ++i)
      for (int p = g.xadj[vs[i]]; p < g.xadj[vs[i] + 1]; ++p) {
        int u = g.adj[p];
        if (region[u] == -1) continue;
        int lu = local[u];
        setb(i, lu); setb(lu, i);
      }
    std::vector<char> done(m, 0);
    for (int step = 0; step < m; ++step) {
      int best = -1, bestd = 1 << 30;
      for (int i = 0; i < m; ++i) {
        if (done[i]) continue;
        int d = 0;
        for (int w = 0; w < W; ++w) d += __builtin_popcountll(rows[(size_t)i * W + w]);
        if (d < bestd) { bestd = d; best = i; }
      }
      const uint64_t *rb = &rows[(size_t)best * W];
      // neighbours of best become a clique
      for (int w = 0; w < W; ++w) {
        uint64_t bits = rb[w];
        while (bits) {
          int u = (w << 6) + __builtin_ctzll(bits);
          bits &= bits - 1;
          uint64_t *ru = &rows[(size_t)u * W];
          for (int x = 0; x < W; ++x) ru[x] |= rb[x];
          ru[u >> 6] &= ~(1ull << (u & 63));
          ru[best >> 6] &= ~(1ull << (best & 63));
        }
      }
      for (int w = 0; w < W; ++w) rows[(size_t)best * W + w] = 0;
      done[best] = 1;
      out.push_back(vs[best]); region[vs[best]] = -1;
    }
    for (int i = 0; i < m; ++i) local[vs[i]] = -1;
    for (int u : ext) local[u] = -1;
  }

  void order_region(std::vector<int> vs) {
    // explicit stack of regions; each entry is a vertex list with a common label
    struct Item { std::vector<int> vs; bool is_sep; };
    std::vector<Item> stack;
    stack.push_back({std::move(vs), false});
    // Separators must be ordered AFTER both halves: emulate post-order with a second marker.
    // We push [sep(is_sep=true), B, A] so that A is popped first, then B, then the separator.
    std::vector<int> bfs_order, bfs2;
    while (!stack.empty()) {
      Item it = std::move(stack.back());
      stack.pop_back();
      std::vector<int> &S = it.vs;
      if (S.empty()) continue;
      if (it.is_sep || (int)S.size() <= leaf) { leaf_md(S); continue; }
      const int r = next_region++;
      for (int v : S) region[v] = r;
      // connected component of S[0]
      bfs(S[0], r, bfs_order);
      if (bfs_order.size() < S.size()) {
        // disconnected: label ALL components in one linear pass (bundle adjustment leaves hundreds of thousands of
        // isolated points once the cameras are taken out).  Small components are binned into leaf-sized groups.
        std::vector<std::vector<int>> big;
        std::vector<int> bin;
        auto flush = [&]() { if (!bin.empty()) { stack.push_back({std::move(bin), true}); bin.clear(); } };
        std::vector<int> comp = bfs_order;
        size_t next_seed = 0;
        while (true) {
          clear_lvl(comp);
          const int rc = next_region++;
          for (int v : comp) region[v] = rc;
          if ((int)comp.size() > leaf) big.push_back(comp);
          else {
            if ((int)(bin.size() + comp.size()) > leaf) flush();
            bin.insert(bin.end(), comp.begin(), comp.end());
          }
          while (next_seed < S.size() && region[S[next_seed]] != r) ++next_seed;
          if (next_seed >= S.size()) break;
          bfs(S[next_seed], r, comp);
        }
        flush();
        for (auto &b : big) stack.push_back({std::move(b), false});
        continue;
      }
      // pseudo-peripheral start: re-run BFS from the last vertex reached (two sweeps)
      int start = bfs_order.back();
      clear_lvl(bfs_order);
      bfs(start, r, bfs_order);
      start = bfs_order.back();
      clear_lvl(bfs_order);
      const int maxl = bfs(start, r, bfs_order);
      if (maxl < 2) { clear_lvl(bfs_order); leaf_md(S); continue; }   // (near-)clique: no useful cut
      std::vector<int> cnt(maxl + 1, 0);
      for (int v : bfs_order) cnt[lvl[v]]++;
      const int n = (int)S.size();
      int best = -1; double bestscore = 1e300;
      int before = 0;
      for (int l = 0; l <= maxl; ++l) {
        const int after = n - before - cnt[l];
        if (l >= 1 && l < maxl && before > 0 && after > 0) {
          const double bal = (double)std::abs(before - after) / n;    // 0 = perfect balance
          const double score = cnt[l] * (1.0 + 4.0 * std::max(0.0, bal - 0.2));
          if (score < bestscore) { bestscore = score; best = l; }
        }
        before += cnt[l];
      }
      if (best < 0) { clear_lvl(bfs_order); leaf_md(S); continue; }
      std::vector<int> A, B, sep;
      for (int v : bfs_order) {
        if (lvl[v] < best) A.push_back(v);
        else if (lvl[v] > best) B.push_back(v);
        else {
          bool touches_b = false;
          for (int p = g.xadj[v]; p < g.xadj[v + 1] && !touches_b; ++p) {
            int u = g.adj[p];
            if (region[u] == r && lvl[u] == best + 1) touches_b = true;
          }
          (touches_b ? sep : A).push_back(v);
        }
      }
      clear_lvl(bfs_order);
      stack.push_back({std::move(sep), true});
      stack.push_back({std::move(B), false});
      stack.push_back({std::move(A), false});
    }
  }
};

}  // namespace

void nested_dissection(const BlockGraph &g, const OrderingOptions &opt, std::vector<int> &perm) {
  ND nd(g, std::max(4, opt.leaf));
  // Hubs (plane landmarks seen from thousands of poses, cameras in bundle adjustment) destroy level structures:
  // take vertices whose degree is far above the mean out of the dissection and eliminate them LAST ("arrow" /
  // Schur ordering), ordered among themselves by dissecting the graph they induce once the rest is gone.
  std::vector<int> sparse, dense;
  if (opt.dense_factor > 0 && g.n > 0) {
    const double mean = (double)g.xadj[g.n] / g.n;
    const double thr = std::max((double)opt.dense_min, opt.dense_factor * mean);
    for (int v = 0; v < g.n; ++v) ((g.xadj[v + 1] - g.xadj[v]) > thr ? dense : sparse).push_back(v);
  } else {
    sparse.resize(g.n);
    std::iota(sparse.begin(), sparse.end(), 0);
  }
  if (dense.empty() || sparse.empty()) {
    std::vector<int> all(g.n);
    std::iota(all.begin(), all.end(), 0);
    nd.order_region(std::move(all));
    perm = std::move(nd.out);
    return;
  }
  for (int v : dense) nd.region[v] = -3;             // invisible to the BFS, still a halo vertex for the leaf ordering
  nd.order_region(std::move(sparse));
  // induced graph on the hubs: direct edges + cliques through sparse vertices with few hub neighbours
  std::vector<int> lid(g.n, -1);
  for (size_t i = 0; i < dense.size(); ++i) lid[dense[i]] = (int)i;
  std::vector<std::pair<int, int>> pr;
  std::vector<int> hubs;
  for (int v = 0; v < g.n; ++v) {
    hubs.clear();
    for (int p = g.xadj[v]; p < g.xadj[v + 1]; ++p) if (lid[g.adj[p]] >= 0) hubs.push_back(lid[g.adj[p]]);
    if (lid[v] >= 0) { for (int h : hubs) if (h != lid[v]) pr.push_back({std::min(h, lid[v]), std::max(h, lid[v])}); }
    else if (hubs.size() <= 16)
      for (size_t a = 0; a < hubs.size(); ++a) for (size_t b = a + 1; b < hubs.size(); ++b) pr.push_back({std::min(hubs[a], hubs[b]), std::max(hubs[a], hubs[b])});
  }
  std::sort(pr.begin(), pr.end());
  pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
  BlockGraph gd;
  gd.n = (int)dense.size();
  gd.xadj.assign(gd.n + 1, 0);
  for (auto &e : pr) { gd.xadj[e.first + 1]++; gd.xadj[e.second + 1]++; }
  for (int i = 0; i < gd.n; ++i) gd.xadj[i + 1] += gd.xadj[i];
  gd.adj.resize(gd.xadj[gd.n]);
  {
    std::vector<int> fill(gd.xadj.begin(), gd.xadj.end() - 1);
    for (auto &e : pr) { gd.adj[fill[e.first]++] = e.second; gd.adj[fill[e.second]++] = e.first; }
  }
  OrderingOptions o2 = opt;
  o2.dense_factor = 0;                               // one level of hub removal
  std::vector<int> pd;
  nested_dissection(gd, o2, pd);
  perm = std::move(nd.out);
  for (int i : pd) perm.push_back(dense[i]);
}

}  // namespace fgo
