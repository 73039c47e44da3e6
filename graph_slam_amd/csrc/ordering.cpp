// Fill-reducing ordering for the GPU block Cholesky: nested dissection (level-structure bisection)
// on top, exact minimum degree with halo inside the leaves.
//
// Why not AMD like g2o's LinearSolverCSparse (selected by the reference at
// g2o/g2o_graph.cpp:30-31,72-74)?  A pose graph is chain-like; minimum-degree orderings give tall,
// thin elimination trees (a dependency chain tens of thousands of columns long).  Nested
// dissection gives thousands of independent leaf sub-trees plus a short separator tree, which is what a
// 256-CU device needs.  The ordering changes fill and speed only, never the solution.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include "fgo_internal.hpp"

namespace fgo {
namespace {

struct ND {
  const BlockGraph &g;
  int leaf;
  double bal_w = 5.0, bal_t = 0.35;
  bool time_mode = false;         // vertex index = time (a pose graph handed over in creation order): dissect by index cuts only
  std::vector<int> tlabel;        // time mode with RECOVERED labels (round 6: a graph whose ids are not creation order): rank of a vertex; empty = its index
  double time_side = 0.30, time_weight = 0.0;
  std::vector<int> base_region;   // template of the per-worker label arrays: 0, or -3 for a hub (invisible to the BFS,
                                  // still a fill-receiving neighbour in the leaf ordering)
  std::atomic<int> next_region{1};

  // per-worker state.  Regions handled by different workers are never adjacent (a separator lies between them) and a
  // worker only needs to know of an outside vertex that it is not eliminated yet (ancestors' separators, hubs), so
  // every worker labels its own copy of the per-vertex arrays: shared arrays would be race-free too, but regions
  // interleave in index space and the cache lines ping-pong (measured: no speed-up at all from 8 threads).
  struct Scratch {
    std::vector<int> region;   // current region label of each vertex (-1 = already ordered)
    std::vector<int> lvl;      // BFS level
    std::vector<int> local;    // global -> local index for the leaf ordering (covers halo vertices, which ARE shared)
    std::vector<int> out;      // elimination order produced by this worker
    std::vector<int> bfs_order, comp, ext, order1, lvl1;
    std::vector<uint64_t> rows;  // adjacency bit rows of the leaf ordering (kept: big ones would be mmap'ed / unmapped per call)
    std::vector<char> done;
  };

  ND(const BlockGraph &gg, int lf) : g(gg), leaf(lf), base_region(gg.n, 0) {}
  void prepare(Scratch &sc) const {
    if (sc.region.empty()) { sc.region = base_region; sc.lvl.assign(g.n, -1); }
  }

  // BFS inside region r from s; fills lvl for reached vertices, returns them in `queue` order.
  int bfs(int s, int r, std::vector<int> &order, Scratch &sc) {
    std::vector<int> &region = sc.region, &lvl = sc.lvl;
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
  void clear_lvl(const std::vector<int> &vs, Scratch &sc) { for (int v : vs) sc.lvl[v] = -1; }

  // exact minimum degree on a small vertex set with halo (neighbours outside the set count towards
  // the degree and receive fill, but are never eliminated here)
  void leaf_md(const std::vector<int> &vs, Scratch &sc) {
    prepare(sc);
    std::vector<int> &out = sc.out, &local = sc.local, &region = sc.region;
    const int m = (int)vs.size();
    // tiny sets need no search; very large separators end up (nearly) dense whatever the order
    if (m <= 2 || m > 384) { for (int v : vs) { out.push_back(v); region[v] = -1; } return; }
    if (local.empty()) local.assign(g.n, -1);
    std::vector<int> &ext = sc.ext;   // halo vertices: not yet ordered, outside the set
    ext.clear();
    for (int i = 0; i < m; ++i) local[vs[i]] = i;
    for (int i = 0; i < m; ++i)
      for (int p = g.xadj[vs[i]]; p < g.xadj[vs[i] + 1]; ++p) {
        int u = g.adj[p];
        if (region[u] == -1) continue;            // eliminated earlier: cannot receive fill
        if (local[u] < 0) { local[u] = m + (int)ext.size(); ext.push_back(u); }
      }
    const int tot = m + (int)ext.size(), W = (tot + 63) / 64;
    std::vector<uint64_t> &rows = sc.rows;
    rows.assign((size_t)tot * W, 0);
    auto setb = [&](int a, int b) { rows[(size_t)a * W + (b >> 6)] |= 1ull << (b & 63); };
    for (int i = 0; i < m; ++i)
      for (int p = g.xadj[vs[i]]; p < g.xadj[vs[i] + 1]; ++p) {
        int u = g.adj[p];
        if (region[u] == -1) continue;
        int lu = local[u];
        setb(i, lu); setb(lu, i);
      }
    std::vector<char> &done = sc.done;
    done.assign(m, 0);
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
      out.push_back(vs[best]);
    }
    // (marked eliminated only now: a vertex of this set is a fill-receiving neighbour of the others until its turn,
    //  and nobody else looks at these labels meanwhile)
    for (int i = 0; i < m; ++i) { region[vs[i]] = -1; local[vs[i]] = -1; }
    for (int u : ext) local[u] = -1;
  }

  // minimum vertex cover of a bipartite graph (X: nx vertices with adjacency xptr / xadj2 into Y: ny vertices): Hopcroft-Karp maximum
  // matching, then Koenig's construction -- Z = reachable from the unmatched X by alternating paths; cover = (X \ Z) + (Y in Z).
  // Returns the membership flags zx / zy of Z.
  static void bipartite_cover(int nx, int ny, const std::vector<int> &xptr, const std::vector<int> &xadj2, std::vector<char> &zx, std::vector<char> &zy) {
    std::vector<int> mx((size_t)nx, -1), my((size_t)ny, -1), dist((size_t)nx), q, it((size_t)nx), stack;
    auto bfs_layers = [&]() {
      q.clear();
      bool found = false;
      for (int i = 0; i < nx; ++i) { if (mx[i] < 0) { dist[i] = 0; q.push_back(i); } else dist[i] = -1; }
      for (size_t h = 0; h < q.size(); ++h) {
        const int i = q[h];
        for (int e = xptr[i]; e < xptr[i + 1]; ++e) {
          const int j = my[xadj2[e]];
          if (j < 0) found = true;
          else if (dist[j] < 0) { dist[j] = dist[i] + 1; q.push_back(j); }
        }
      }
      return found;
    };
    auto augment = [&](int root) {
      stack.clear(); stack.push_back(root);
      while (!stack.empty()) {
        const int i = stack.back();
        if (it[i] == xptr[i + 1]) { dist[i] = -1; stack.pop_back(); continue; }
        const int y = xadj2[it[i]++];
        const int j = my[y];
        if (j < 0) {
          int yy = y;
          for (int k = (int)stack.size() - 1; k >= 0; --k) { const int x = stack[k]; const int prev = mx[x]; mx[x] = yy; my[yy] = x; yy = prev; }
          return true;
        }
        if (dist[j] == dist[i] + 1) stack.push_back(j);
      }
      return false;
    };
    while (bfs_layers()) {
      for (int i = 0; i < nx; ++i) it[i] = xptr[i];
      for (int i = 0; i < nx; ++i) if (mx[i] < 0) augment(i);
    }
    zx.assign((size_t)nx, 0); zy.assign((size_t)ny, 0);
    q.clear();
    for (int i = 0; i < nx; ++i) if (mx[i] < 0) { zx[i] = 1; q.push_back(i); }
    for (size_t h = 0; h < q.size(); ++h) {
      const int i = q[h];
      for (int e = xptr[i]; e < xptr[i + 1]; ++e) {
        const int y = xadj2[e];
        if (zy[y] || mx[i] == y) continue;
        zy[y] = 1;
        const int j = my[y];
        if (j >= 0 && !zx[j]) { zx[j] = 1; q.push_back(j); }
      }
    }
  }

  enum SplitResult { SPLIT, DISCONNECTED, NO_CUT };
  // TIME dissection (round 5).  A driver hands the poses over in the order it created them (g2o/g2o_graph.cpp:159-239: the vertex id is
  // the key-frame counter), so the vertex index IS time, and a trajectory that does not come back to an earlier place for a while
  // leaves cuts "between rank t - 1 and rank t" that are crossed by the odometry / look-back band only.  cfg 2 has such a cut at 33 %
  // of the trajectory whose minimum vertex cover is 10 vertices, where the level structures of a BFS -- distorted by every loop
  // closure -- settle for a root separator of 101.  Regions stay contiguous in time under these cuts, so the cuts below them stay
  // clean as well (mixing them with level cuts does not work: measured, profiles/NOTES.md).  A region is cut at the rank -- smaller
  // side >= 30 % of its vertices -- whose crossing edges have the smallest MINIMUM VERTEX COVER (exact for the eight ranks with the fewest
  // crossing edges; Hopcroft-Karp + Koenig); no BFS at all.  On five 100k-pose graphs: 20-26 levels instead of 22-32, 0-8 % fewer
  // block updates, predicted sweep + backward time -5 .. -15 %.
  SplitResult split_by_index(const std::vector<int> &S, Scratch &sc, int r, std::vector<int> &A, std::vector<int> &B, std::vector<int> &sep) {
    std::vector<int> &region = sc.region, &lvl = sc.lvl;
    const int n = (int)S.size();
    const double side = time_side;
    static const int keep = (int)tune("nd_time_keep", 8);
    std::vector<int> ids(S);
    if (tlabel.empty()) std::sort(ids.begin(), ids.end());
    else std::sort(ids.begin(), ids.end(), [&](int a, int b) { return tlabel[(size_t)a] < tlabel[(size_t)b]; });
    for (int i = 0; i < n; ++i) lvl[ids[i]] = i;                                    // rank inside the region
    std::vector<int> diff((size_t)n + 2, 0);
    for (int i = 0; i < n; ++i) {
      const int v = ids[i];
      for (int p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
        const int u = g.adj[p];
        if (region[u] != r) continue;
        const int j = lvl[u];
        if (j > i) { diff[(size_t)i + 1]++; diff[(size_t)j + 1]--; }                // crosses every cut t with i < t <= j
      }
    }
    int t_lo = std::max(1, (int)std::ceil(side * n)), t_hi = std::min(n - 1, n - t_lo);
    std::vector<std::pair<int, int>> cand;                                         // (crossing edges, t)
    {
      // (nd_time_weight = w > 0: the balance window is taken in the weight 1 + w * crossing edges instead of in vertices -- a stretch of
      //  the trajectory that many loop closures span will need larger separators further down, i.e. a deeper sub-tree per vertex)
      const double wgt = time_weight;
      if (wgt > 0) {
        std::vector<double> acc((size_t)n + 1, 0.0);
        int cross = 0;
        for (int t = 1; t < n; ++t) { cross += diff[(size_t)t]; acc[(size_t)t] = acc[(size_t)t - 1] + 1.0 + wgt * cross; }
        acc[(size_t)n] = acc[(size_t)n - 1] + 1.0;
        const double tot = acc[(size_t)n];
        t_lo = (int)(std::lower_bound(acc.begin(), acc.end(), side * tot) - acc.begin());
        t_hi = (int)(std::upper_bound(acc.begin(), acc.end(), (1.0 - side) * tot) - acc.begin()) - 1;
        t_lo = std::max(1, std::min(t_lo, n - 1)); t_hi = std::max(t_lo, std::min(t_hi, n - 1));
      }
      int cross = 0;
      for (int t = 1; t < n; ++t) { cross += diff[(size_t)t]; if (t >= t_lo && t <= t_hi) cand.push_back({cross, t}); }
    }
    if (cand.empty()) { for (int i = 0; i < n; ++i) lvl[ids[i]] = -1; return NO_CUT; }
    const size_t nk = std::min<size_t>((size_t)std::max(1, keep), cand.size());
    std::partial_sort(cand.begin(), cand.begin() + (std::ptrdiff_t)nk, cand.end(), [&](const std::pair<int, int> &x, const std::pair<int, int> &y) {
      if (x.first != y.first) return x.first < y.first;
      return std::abs(2 * x.second - n) < std::abs(2 * y.second - n);             // ties: the better balanced cut
    });
    int best_t = -1; size_t best_size = (size_t)-1;
    std::vector<int> best_cover, X, ys, xptr, xadj2, cover;
    std::vector<char> zx, zy;
    // the crossing edges of ALL kept cuts in one more sweep over the region's edges (a sweep per cut was most of the serial top of the
    // ordering: 8 x half the region each): edge (i, j), i < j, crosses the cuts t with i < t <= j
    std::vector<std::vector<std::pair<int, int>>> ces(nk);
    {
      bool any_zero = false;
      for (size_t c = 0; c < nk; ++c) any_zero = any_zero || cand[c].first == 0;
      if (!any_zero) {
        std::vector<std::pair<int, int>> ts(nk);                                     // (t, candidate), ascending t
        for (size_t c = 0; c < nk; ++c) ts[c] = {cand[c].second, (int)c};
        std::sort(ts.begin(), ts.end());
        const int tmin = ts.front().first, tmax = ts.back().first;
        for (size_t c = 0; c < nk; ++c) ces[c].reserve((size_t)cand[c].first);
        for (int i = 0; i < tmax; ++i) {
          const int v = ids[i];
          for (int p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
            const int u = g.adj[p];
            if (region[u] != r) continue;
            const int j = lvl[u];
            if (j <= i || j < tmin) continue;
            for (size_t q = 0; q < nk; ++q) { const int t = ts[q].first; if (t > j) break; if (t > i) ces[(size_t)ts[q].second].push_back({i, j}); }
          }
        }
      }
    }
    for (size_t c = 0; c < nk; ++c) {
      const int t = cand[c].second;
      if ((size_t)cand[c].first == 0) { best_t = t; best_cover.clear(); best_size = 0; break; }   // nothing crosses: the region falls apart here
      // crossing edges as a bipartite graph: X = their left endpoints (rank < t), Y = their right endpoints
      std::vector<std::pair<int, int>> &ce = ces[c];
      std::sort(ce.begin(), ce.end());
      ys.clear();
      for (auto &e : ce) ys.push_back(e.second);
      std::sort(ys.begin(), ys.end()); ys.erase(std::unique(ys.begin(), ys.end()), ys.end());
      X.clear(); xptr.assign(1, 0); xadj2.clear();
      for (size_t k = 0; k < ce.size(); ++k) {
        if (k == 0 || ce[k].first != ce[k - 1].first) { if (k) xptr.push_back((int)xadj2.size()); X.push_back(ce[k].first); }
        xadj2.push_back((int)(std::lower_bound(ys.begin(), ys.end(), ce[k].second) - ys.begin()));
      }
      xptr.push_back((int)xadj2.size());
      bipartite_cover((int)X.size(), (int)ys.size(), xptr, xadj2, zx, zy);
      cover.clear();
      for (size_t i = 0; i < X.size(); ++i) if (!zx[i]) cover.push_back(X[i]);
      for (size_t y = 0; y < ys.size(); ++y) if (zy[y]) cover.push_back(ys[y]);
      if (cover.size() < best_size) { best_size = cover.size(); best_t = t; best_cover = cover; }
    }
    std::vector<char> in_sep((size_t)n, 0);
    for (int q : best_cover) in_sep[(size_t)q] = 1;
    A.clear(); B.clear(); sep.clear();
    for (int i = 0; i < n; ++i) (in_sep[(size_t)i] ? sep : (i < best_t ? A : B)).push_back(ids[i]);
    for (int i = 0; i < n; ++i) lvl[ids[i]] = -1;
    if (A.empty() || B.empty()) return NO_CUT;
    return SPLIT;
  }
  // One bisection of the (freshly labelled) region S by a BFS level structure from a pseudo-peripheral vertex:
  // separator = the cut level trimmed to the vertices that touch the far side.  DISCONNECTED leaves the component of
  // S[0] in sc.bfs_order (levels set) and the region label r on all of S.
  SplitResult split(const std::vector<int> &S, Scratch &sc, int &r, std::vector<int> &A, std::vector<int> &B, std::vector<int> &sep) {
    prepare(sc);
    std::vector<int> &bfs_order = sc.bfs_order, &region = sc.region, &lvl = sc.lvl;
    r = next_region++;
    for (int v : S) region[v] = r;
    if (time_mode) return split_by_index(S, sc, r, A, B, sep);
    bfs(S[0], r, bfs_order, sc);
    if (bfs_order.size() < S.size()) return DISCONNECTED;
    // pseudo-peripheral start: re-run BFS from the last vertex reached (two sweeps); then the level structures rooted at
    // BOTH ends of that pseudo-diameter are searched for the best cut
    const double bal_t = this->bal_t, bal_w = this->bal_w;
    static const int n_starts = (int)tune("nd_starts", 2);
    static const double min_side = tune("nd_min_side", 0.03);   // (0.03 leaves the cuts of the Manhattan benchmark graphs alone: cfg 2 keeps 30 levels)
    int start = bfs_order.back();
    clear_lvl(bfs_order, sc);
    bfs(start, r, bfs_order, sc);
    start = bfs_order.back();
    clear_lvl(bfs_order, sc);
    const int n = (int)S.size();
    int best = -1, bestdir = 0, beststart = -1, maxl = 0; double bestscore = 1e300;
    std::vector<int> cnt, tf, tb;
    // candidate separators: the vertices of level l that touch level l+1 ("forward": the rest of level l joins the
    // near side) or those that touch level l-1 ("backward": the rest joins the far side); scored by their trimmed size
    // with a penalty for unbalanced parts
    auto evaluate = [&](int from) -> int {                       // leaves the level structure of `from` in lvl / bfs_order
      // BFS and the per-level counts in one sweep: when v (level l) leaves the queue every vertex of level l - 1 is labelled,
      // and a neighbour of level l + 1 is either labelled already or gets its label from v right now
      cnt.clear(); tf.clear(); tb.clear();
      bfs_order.clear();
      bfs_order.push_back(from); lvl[from] = 0;
      int ml = 0;
      for (size_t head = 0; head < bfs_order.size(); ++head) {
        const int v = bfs_order[head], l = lvl[v];
        if (l >= (int)cnt.size()) { cnt.push_back(0); tf.push_back(0); tb.push_back(0); }
        cnt[l]++;
        bool up = false, down = false;
        for (int p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
          const int u = g.adj[p];
          if (region[u] != r) continue;
          if (lvl[u] < 0) { lvl[u] = l + 1; ml = l + 1; bfs_order.push_back(u); up = true; }
          else { up = up || lvl[u] == l + 1; down = down || lvl[u] == l - 1; }
        }
        tf[l] += up; tb[l] += down;
      }
      if (ml < 2) return ml;
      int before = 0;
      for (int l = 0; l <= ml; ++l) {
        const int after = n - before - cnt[l];
        if (l >= 1 && l < ml && before > 0 && after > 0) {
          for (int dir = 0; dir < 2; ++dir) {
            const int sz = dir == 0 ? tf[l] : tb[l];
            const int na = dir == 0 ? before + cnt[l] - sz : before, nb2 = dir == 0 ? after : after + cnt[l] - sz;
            const double bal = (double)std::abs(na - nb2) / n;      // 0 = perfect balance
            double score = sz * (1.0 + bal_w * std::max(0.0, bal - bal_t));
            // Mesh-like graphs (a torus, a 2-D grid): the level sizes GROW from the root, so the first levels are tiny and
            // win on size against any penalty that is linear in the imbalance -- the dissection then peels one vertex at a
            // time (etree height n / 2, fill n^1.5: measured on a 40 x 40 torus).  A cut whose smaller side holds less than
            // min_side of the region only competes among its own kind (used if nothing better exists).
            if (std::min(na, nb2) < min_side * n) score += 1e12;
            if (score < bestscore) { bestscore = score; best = l; bestdir = dir; beststart = from; maxl = ml; }
          }
        }
        before += cnt[l];
      }
      return ml;
    };
    const int ml1 = evaluate(start);
    if (ml1 < 2) { clear_lvl(bfs_order, sc); return NO_CUT; }          // (near-)clique: no useful cut
    if (n_starts > 1) {
      int last = bfs_order.back();
      // the first level structure is kept (order and levels: two copies instead of one more sweep if it wins)
      std::vector<int> &order1 = sc.order1, &lvl1 = sc.lvl1;
      order1 = bfs_order;
      lvl1.resize(order1.size());
      for (size_t i = 0; i < order1.size(); ++i) lvl1[i] = lvl[order1[i]];
      clear_lvl(bfs_order, sc);
      evaluate(last);
      if (n_starts > 2) {                                         // a second pseudo-diameter, swept from the middle of the region
        clear_lvl(bfs_order, sc);
        bfs(S[S.size() / 2], r, bfs_order, sc);
        int e3 = bfs_order.back();
        clear_lvl(bfs_order, sc);
        evaluate(e3);
        last = e3;
        if (n_starts > 3) { const int e4 = bfs_order.back(); clear_lvl(bfs_order, sc); evaluate(e4); last = e4; }
      }
      if (beststart != last) {                                                                  // back to the winner's levels
        clear_lvl(bfs_order, sc);
        if (beststart == start) { bfs_order = order1; for (size_t i = 0; i < order1.size(); ++i) lvl[order1[i]] = lvl1[i]; }
        else bfs(beststart, r, bfs_order, sc);
      }
    }
    (void)maxl;
    if (best < 0) { clear_lvl(bfs_order, sc); return NO_CUT; }
    // the cut (level `best`, direction `bestdir`) of the level structure in lvl -> A, B, sep; returns its score
    auto materialize = [&](const int best, const int bestdir, std::vector<int> &A, std::vector<int> &B, std::vector<int> &sep) -> double {
    A.clear(); B.clear(); sep.clear();
    for (int v : bfs_order) {
      if (lvl[v] < best) A.push_back(v);
      else if (lvl[v] > best) B.push_back(v);
      else {
        bool touches = false;
        const int other = bestdir == 0 ? best + 1 : best - 1;
        for (int p = g.xadj[v]; p < g.xadj[v + 1] && !touches; ++p) {
          int u = g.adj[p];
          if (region[u] == r && lvl[u] == other) touches = true;
        }
        (touches ? sep : (bestdir == 0 ? A : B)).push_back(v);
      }
    }
    // The trimmed level X is ONE vertex cover of the edges between it and the far side's neighbours Y; a MINIMUM vertex cover of
    // that bipartite graph (maximum matching + Koenig's construction) separates the same two sides with fewer vertices: X minus
    // the cover joins the near side, the cover's part of Y leaves the far side.
    static const int cover_on = (int)tune("nd_cover", 1);
    if (cover_on && sep.size() >= 2) {
      const int other = bestdir == 0 ? best + 1 : best - 1;
      std::vector<int> &nearv = bestdir == 0 ? A : B, &farv = bestdir == 0 ? B : A;
      constexpr int XB = 1 << 29, YB = 1 << 30;
      const int nx = (int)sep.size();
      const std::vector<int> Xv = sep;
      std::vector<int> Y, xptr((size_t)nx + 1, 0), xadj2;
      for (int i = 0; i < nx; ++i) lvl[sep[i]] = XB + i;
      for (int i = 0; i < nx; ++i) {
        const int v = sep[i];
        for (int p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
          const int u = g.adj[p];
          if (region[u] != r) continue;
          if (lvl[u] == other) { lvl[u] = YB + (int)Y.size(); Y.push_back(u); }
          if (lvl[u] >= YB) xadj2.push_back(lvl[u] - YB);
        }
        xptr[(size_t)i + 1] = (int)xadj2.size();
      }
      const int ny = (int)Y.size();
      // Hopcroft-Karp
      std::vector<int> mx((size_t)nx, -1), my((size_t)ny, -1), dist((size_t)nx), q, it((size_t)nx);
      auto bfs_layers = [&]() {
        q.clear();
        bool found = false;
        for (int i = 0; i < nx; ++i) { if (mx[i] < 0) { dist[i] = 0; q.push_back(i); } else dist[i] = -1; }
        for (size_t h = 0; h < q.size(); ++h) {
          const int i = q[h];
          for (int e = xptr[i]; e < xptr[i + 1]; ++e) {
            const int j = my[xadj2[e]];
            if (j < 0) found = true;
            else if (dist[j] < 0) { dist[j] = dist[i] + 1; q.push_back(j); }
          }
        }
        return found;
      };
      std::vector<int> stack;
      auto augment = [&](int root) {                       // iterative DFS along the layers
        stack.clear(); stack.push_back(root);
        while (!stack.empty()) {
          const int i = stack.back();
          if (it[i] == xptr[i + 1]) { dist[i] = -1; stack.pop_back(); continue; }
          const int y = xadj2[it[i]++];
          const int j = my[y];
          if (j < 0) {                                     // free y: flip the path on the stack
            int yy = y;
            for (int k = (int)stack.size() - 1; k >= 0; --k) { const int x = stack[k]; const int prev = mx[x]; mx[x] = yy; my[yy] = x; yy = prev; }
            return true;
          }
          if (dist[j] == dist[i] + 1) stack.push_back(j);
        }
        return false;
      };
      while (bfs_layers()) {
        for (int i = 0; i < nx; ++i) it[i] = xptr[i];
        for (int i = 0; i < nx; ++i) if (mx[i] < 0) augment(i);
      }
      // Koenig: Z = reachable from the unmatched X by alternating paths; cover = (X \ Z) + (Y in Z)
      std::vector<char> zx((size_t)nx, 0), zy((size_t)ny, 0);
      q.clear();
      for (int i = 0; i < nx; ++i) if (mx[i] < 0) { zx[i] = 1; q.push_back(i); }
      for (size_t h = 0; h < q.size(); ++h) {
        const int i = q[h];
        for (int e = xptr[i]; e < xptr[i + 1]; ++e) {
          const int y = xadj2[e];
          if (zy[y] || mx[i] == y) continue;
          zy[y] = 1;
          const int j = my[y];
          if (j >= 0 && !zx[j]) { zx[j] = 1; q.push_back(j); }
        }
      }
      int csize = 0;
      for (int i = 0; i < nx; ++i) csize += !zx[i];
      for (int y = 0; y < ny; ++y) csize += zy[y];
      if (csize < nx) {
        std::vector<int> nsep;
        for (int i = 0; i < nx; ++i) (zx[i] ? nearv : nsep).push_back(sep[i]);
        size_t w = 0;
        for (size_t k = 0; k < farv.size(); ++k) {
          const int u = farv[k];
          if (lvl[u] >= YB && zy[lvl[u] - YB]) nsep.push_back(u); else farv[w++] = u;
        }
        farv.resize(w);
        sep.swap(nsep);
      }
      for (int i = 0; i < nx; ++i) lvl[Xv[i]] = best;             // (the level structure serves the next candidate)
      for (int u : Y) lvl[u] = other;
    }
    const double bal = (double)std::abs((int)A.size() - (int)B.size()) / n;
    double score = (double)sep.size() * (1.0 + bal_w * std::max(0.0, bal - bal_t));
    if ((int)std::min(A.size(), B.size()) < min_side * n) score += 1e12;
    return score;
    };
    // (the three / six best cuts by trimmed size, compared again after the refinement: 0.5 % less fill, but the level count moves
    //  by +-1-2 either way -- predicted sweep times 3 721 / 3 788 / 3 779 us on cfg 2, three other seeds alike; not kept)
    materialize(best, bestdir, A, B, sep);
    clear_lvl(bfs_order, sc);
    return SPLIT;
  }

  struct Item { std::vector<int> vs; bool is_sep; };
  // A region that split() found DISCONNECTED (the component of S[0] in sc.bfs_order, levels set, label r on all of S): label
  // ALL components in one linear pass (bundle adjustment leaves hundreds of thousands of isolated points once the cameras
  // are taken out).  Small components are binned into leaf-sized groups (is_sep: ordered as they are); appended to `items` in
  // the order the serial dissection pushes them on its stack -- they are processed in the REVERSE order.
  void components(const std::vector<int> &S, Scratch &sc, int r, std::vector<Item> &items) {
    std::vector<int> &region = sc.region;
    std::vector<std::vector<int>> big;
    std::vector<int> bin;
    auto flush = [&]() { if (!bin.empty()) { items.push_back({std::move(bin), true}); bin.clear(); } };
    std::vector<int> &comp = sc.comp;
    comp = sc.bfs_order;
    size_t next_seed = 0;
    while (true) {
      clear_lvl(comp, sc);
      const int rc = next_region++;
      for (int v : comp) region[v] = rc;
      if ((int)comp.size() > leaf) big.push_back(comp);
      else {
        if ((int)(bin.size() + comp.size()) > leaf) flush();
        bin.insert(bin.end(), comp.begin(), comp.end());
      }
      while (next_seed < S.size() && region[S[next_seed]] != r) ++next_seed;
      if (next_seed >= S.size()) break;
      bfs(S[next_seed], r, comp, sc);
    }
    flush();
    for (auto &b : big) items.push_back({std::move(b), false});
  }

  // serial dissection of one region (explicit stack); appends to sc.out
  void order_region(std::vector<int> vs, Scratch &sc) {
    std::vector<Item> stack;
    stack.push_back({std::move(vs), false});
    // Separators must be ordered AFTER both halves: emulate post-order with a second marker.
    // We push [sep(is_sep=true), B, A] so that A is popped first, then B, then the separator.
    prepare(sc);
    std::vector<int> A, B, sep;
    while (!stack.empty()) {
      Item it = std::move(stack.back());
      stack.pop_back();
      std::vector<int> &S = it.vs;
      if (S.empty()) continue;
      if (it.is_sep || (int)S.size() <= leaf) { leaf_md(S, sc); continue; }
      int r = 0;
      const SplitResult res = split(S, sc, r, A, B, sep);
      if (res == NO_CUT) { leaf_md(S, sc); continue; }
      if (res == DISCONNECTED) {
        std::vector<Item> items;
        components(S, sc, r, items);
        for (Item &i2 : items) stack.push_back(std::move(i2));
        continue;
      }
      stack.push_back({std::move(sep), true});
      stack.push_back({std::move(B), false});
      stack.push_back({std::move(A), false});
      A = std::vector<int>(); B = std::vector<int>(); sep = std::vector<int>();
    }
  }

  // The top of the dissection tree is expanded breadth-first, the regions of one depth being bisected concurrently;
  // the subregions below are ordered concurrently, one worker each; the separators of the top tree follow in
  // post-order.  Same order as the serial algorithm: what happens inside a region never depends on the other side
  // of a separator.
  void order_all(std::vector<int> vs, std::vector<int> &out) {
    // an expanded node: its children in the order they are eliminated (two for a bisection, the components for a disconnected
    // region -- the small ones together in ONE child that orders its sets as they are), then its separator
    struct Node { std::vector<int> vs, sep, kids; std::vector<std::vector<int>> sets; bool expanded = false, leaf_only = false; Scratch sc; };
    std::vector<Node> nodes(1);
    nodes[0].vs = std::move(vs);
    const bool prof = std::getenv("FGO_SYM_PROFILE") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = tnow();
    const int want = 2 * host_threads();
    if (host_threads() > 1) {
      std::vector<int> frontier{0};                      // unexpanded regions that may still be bisected
      int settled = 0;                                   // unexpanded regions that will not be (small, or no cut found)
      // ... until there are `want` regions AND none of them is much larger than its share: bisections are not balanced, and a
      // region that stays big is ordered by ONE worker (cfg 5: a region of 274 k of the 1 M poses, 0.40 s of the 0.78 s)
      const size_t big = std::max<size_t>((size_t)8 * leaf, nodes[0].vs.size() / (size_t)want);
      while (!frontier.empty()) {
        const bool more = (int)frontier.size() + settled < want;
        std::vector<int> cand;
        for (int id : frontier) {
          const size_t sz = nodes[id].vs.size();
          if ((int)sz > 8 * leaf && (more || sz > big)) cand.push_back(id); else ++settled;
        }
        if (cand.empty()) break;
        std::vector<std::vector<int>> A(cand.size()), B(cand.size());
        std::vector<std::vector<Item>> comps(cand.size());
        std::vector<char> ok(cand.size(), 0);                // 1: bisected, 2: disconnected (components)
        parallel_ranges((int)cand.size(), 1, [&](int q0, int q1) {
          for (int q = q0; q < q1; ++q) {
            Node &nd = nodes[cand[q]];
            int r = 0;
            const SplitResult res = split(nd.vs, nd.sc, r, A[q], B[q], nd.sep);
            if (res == DISCONNECTED) { components(nd.vs, nd.sc, r, comps[q]); ok[q] = 2; }
            else ok[q] = res == SPLIT;
          }
        });
        frontier.clear();
        for (size_t q = 0; q < cand.size(); ++q) {
          const int id = cand[q];
          if (!ok[q]) { ++settled; nodes[id].sep.clear(); continue; }
          nodes[id].expanded = true;
          std::vector<int>().swap(nodes[id].vs);
          nodes[id].sc = Scratch();                          // its label arrays are no longer needed
          if (ok[q] == 1) {
            const int ia = (int)nodes.size(), ib = ia + 1;
            nodes.resize(nodes.size() + 2);                  // (invalidates references, not indices)
            nodes[id].kids = {ia, ib};
            nodes[ia].vs = std::move(A[q]);
            nodes[ib].vs = std::move(B[q]);
            frontier.push_back(ia); frontier.push_back(ib);
          } else {
            nodes[id].sep.clear();
            std::vector<std::vector<int>> sets;
            for (size_t x = comps[q].size(); x-- > 0;) {      // the order in which the serial dissection pops them
              Item &it = comps[q][x];
              if (it.is_sep) { sets.push_back(std::move(it.vs)); continue; }
              const int ic = (int)nodes.size();
              nodes.resize(nodes.size() + 1);
              nodes[id].kids.push_back(ic);
              nodes[ic].vs = std::move(it.vs);
              frontier.push_back(ic);
            }
            if (!sets.empty()) {                               // (pushed first, hence popped last: behind the large components)
              const int ic = (int)nodes.size();
              nodes.resize(nodes.size() + 1);
              nodes[id].kids.push_back(ic);
              nodes[ic].leaf_only = true;
              nodes[ic].sets = std::move(sets);
              ++settled;
            }
          }
        }
      }
    }
    const double t1 = tnow();
    if (prof) { std::fprintf(stderr, "[fgo ordering] region sizes:"); for (auto &nd : nodes) if (!nd.expanded) std::fprintf(stderr, " %zu", nd.vs.size()); std::fprintf(stderr, "\n"); }
    // order the unexpanded regions concurrently
    std::vector<int> work;
    for (size_t id = 0; id < nodes.size(); ++id) if (!nodes[id].expanded) work.push_back((int)id);
    parallel_ranges((int)work.size(), 1, [&](int w0, int w1) {
      for (int w = w0; w < w1; ++w) {
        Node &nd = nodes[work[w]];
        const double ta = tnow(); const size_t sz = nd.vs.size();
        if (nd.leaf_only) { for (const std::vector<int> &set : nd.sets) leaf_md(set, nd.sc); std::vector<std::vector<int>>().swap(nd.sets); }
        else order_region(std::move(nd.vs), nd.sc);
        if (prof) std::fprintf(stderr, "[fgo ordering]   region of %zu: %.1f ms (start %.1f)\n", sz, 1e3 * (tnow() - ta), 1e3 * (ta - t1));
        std::vector<int>().swap(nd.sc.local); std::vector<int>().swap(nd.sc.region); std::vector<int>().swap(nd.sc.lvl);
      }
    });
    const double t2 = tnow();
    // emit in post-order: A, B, separator
    Scratch top;                                         // for the separators of the top tree: everything emitted so far is eliminated
    const bool any_expanded = nodes[0].expanded;
    if (any_expanded) prepare(top);
    struct Frame { int id; int stage; };
    std::vector<Frame> st{{0, 0}};
    while (!st.empty()) {
      Frame &f = st.back();
      Node &nd = nodes[f.id];
      if (!nd.expanded) {
        out.insert(out.end(), nd.sc.out.begin(), nd.sc.out.end());
        if (any_expanded) for (int v : nd.sc.out) top.region[v] = -1;
        st.pop_back();
        continue;
      }
      if (f.stage < (int)nd.kids.size()) { const int kid = nd.kids[(size_t)f.stage++]; st.push_back({kid, 0}); continue; }
      top.out.clear();
      if (!nd.sep.empty()) leaf_md(nd.sep, top);
      out.insert(out.end(), top.out.begin(), top.out.end());
      st.pop_back();
    }
    if (prof) std::fprintf(stderr, "[fgo ordering] top tree (%zu nodes) %.1f ms, regions %.1f ms, top separators %.1f ms\n", nodes.size(), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (tnow() - t2));
  }
};

}  // namespace

void nested_dissection(const BlockGraph &g, const OrderingOptions &opt, std::vector<int> &perm) {
  ND nd(g, std::max(4, opt.leaf));
  nd.bal_w = opt.bal_w; nd.bal_t = opt.bal_t;
  nd.time_side = opt.time_side; nd.time_weight = opt.time_weight;
  // Hubs (plane landmarks seen from thousands of poses, cameras in bundle adjustment) destroy level structures:
  // take vertices whose degree is far above the mean out of the dissection and eliminate them LAST ("arrow" /
  // Schur ordering), ordered among themselves by dissecting the graph they induce once the rest is gone.
  std::vector<int> sparse, dense;
  std::vector<char> is_hub((size_t)g.n, 0);
  if (opt.dense_factor > 0 && g.n > 0) {
    const double mean = (double)g.xadj[g.n] / g.n;
    const double thr = std::max((double)opt.dense_min, opt.dense_factor * mean);
    for (int v = 0; v < g.n; ++v) { const bool h = (g.xadj[v + 1] - g.xadj[v]) > thr; (h ? dense : sparse).push_back(v); is_hub[(size_t)v] = h; }
  } else {
    sparse.resize(g.n);
    std::iota(sparse.begin(), sparse.end(), 0);
  }
  if (sparse.empty()) std::fill(is_hub.begin(), is_hub.end(), 0);
  // vertex index = time?  A pose graph handed over in creation order has (nearly) all of its edges inside a narrow index band
  // (cfg 2: 98.6 % within 10; a bundle adjustment's camera-landmark edges, a grid: not).  The edges of the hubs do not count: they are
  // eliminated last whatever the rest looks like (round 6).
  static const int time_on = (int)tune("nd_time", 1);
  static const int band = (int)tune("nd_time_band", 32);
  static const double frac = tune("nd_time_frac", 0.9);
  auto band_share = [&](const std::vector<int> *label) {
    int64_t total = 0, shortr = 0;
    for (int v = 0; v < g.n; ++v) {
      if (is_hub[(size_t)v]) continue;
      for (int p = g.xadj[v]; p < g.xadj[v + 1]; ++p) {
        const int u = g.adj[p];
        if (u > v && !is_hub[(size_t)u]) { ++total; shortr += std::abs(label ? (*label)[(size_t)u] - (*label)[(size_t)v] : u - v) <= band; }
      }
    }
    return total > 0 ? (double)shortr / (double)total : 0.0;
  };
  nd.time_mode = time_on && band_share(nullptr) >= frac;
  // ... or a graph that HAS such an order under other labels (round 6; VERDICT r5 next #4a / #6): a .g2o file need not number its vertices in
  // creation order, and GTSAM keys group a VIO graph's variables by TYPE -- X(k) = k, V(k) = K + k, B(k) = 2 K + k (gtsam/gtsam_graph.cpp:50-54:
  // the key is (char << 56) | index) -- although pose, velocity and bias of key frame k are coupled to those of k + 1 only
  // (gtsam/test_vro_imu_graph.cpp:191-198).  One Cuthill-McKee pass (breadth-first from a pseudo-peripheral vertex, neighbours by ascending
  // degree, hubs left out) recovers the order of a band graph exactly and is held to the SAME band test; a graph that fails it under the
  // recovered labels too -- loop closures fold the breadth-first fronts over each other -- keeps the level-structure dissection.
  static const int recover_on = (int)tune("nd_time_recover", 1);
  static const double frac2 = tune("nd_time_frac2", frac);
  if (time_on && recover_on && opt.time_recover && !nd.time_mode && g.n > 2) {
    std::vector<int> label((size_t)g.n, -1), queue, nb;
    queue.reserve((size_t)g.n);
    auto deg = [&](int v) { return g.xadj[v + 1] - g.xadj[v]; };
    auto bfs_from = [&](int s, int stamp, std::vector<int> &seen) {       // returns the last vertex reached (a deepest one)
      size_t head = queue.size();
      queue.push_back(s); seen[(size_t)s] = stamp;
      while (head < queue.size()) {
        const int v = queue[head++];
        nb.clear();
        for (int p = g.xadj[v]; p < g.xadj[v + 1]; ++p) { const int u = g.adj[p]; if (!is_hub[(size_t)u] && seen[(size_t)u] != stamp && seen[(size_t)u] >= -2) nb.push_back(u); }
        std::sort(nb.begin(), nb.end(), [&](int x, int y) { const int dx = deg(x), dy = deg(y); return dx != dy ? dx < dy : x < y; });
        for (int u : nb) if (seen[(size_t)u] != stamp) { seen[(size_t)u] = stamp; queue.push_back(u); }
      }
      return queue.back();
    };
    // seen: -1 untouched, -2 touched by a probing sweep, -3 final (labelled); stamps of the probing sweeps: 1, 2
    std::vector<int> seen((size_t)g.n, -1);
    int next = 0;
    for (int s0 = 0; s0 < g.n; ++s0) {
      if (is_hub[(size_t)s0] || seen[(size_t)s0] == -3) continue;
      // two probing sweeps find a pseudo-peripheral start of this component, the third one labels it
      queue.clear(); const int e1 = bfs_from(s0, 1, seen);
      queue.clear(); const int e2 = bfs_from(e1, 2, seen);
      queue.clear(); bfs_from(e2, 3, seen);
      for (int v : queue) { label[(size_t)v] = next++; seen[(size_t)v] = -3; }
    }
    for (int v : dense) label[(size_t)v] = next++;
    if (band_share(&label) >= frac2) { nd.time_mode = true; nd.tlabel.swap(label); }
  }
  if (dense.empty() || sparse.empty()) {
    std::vector<int> all(g.n);
    std::iota(all.begin(), all.end(), 0);
    nd.order_all(std::move(all), perm);
    return;
  }
  for (int v : dense) nd.base_region[v] = -3;             // invisible to the BFS, still a halo vertex for the leaf ordering
  perm.clear();
  nd.order_all(std::move(sparse), perm);
  // induced graph on the hubs: direct edges + cliques through sparse vertices with few hub neighbours
  std::vector<int> lid(g.n, -1);
  for (size_t i = 0; i < dense.size(); ++i) lid[dense[i]] = (int)i;
  // neighbours of hub a = hubs adjacent to it + hubs that share a sparse vertex (with <= 16 hub neighbours) with it;
  // gathered per hub with a stamp row, hubs spread over the host threads (a bundle adjustment has 5 M camera-point
  // incidences: listing and sorting all hub pairs took most of the ordering time)
  BlockGraph gd;
  gd.n = (int)dense.size();
  std::vector<std::vector<int>> nbrs(dense.size());
  std::vector<char> few((size_t)g.n, 0);           // sparse vertex with 2 .. 16 hub neighbours
  parallel_ranges(g.n, 16384, [&](int v0, int v1) {
    for (int v = v0; v < v1; ++v) {
      if (lid[v] >= 0) continue;
      int nh = 0;
      for (int p = g.xadj[v]; p < g.xadj[v + 1]; ++p) nh += lid[g.adj[p]] >= 0;
      few[v] = nh >= 2 && nh <= 16;
    }
  });
  parallel_ranges(gd.n, std::max(1, (gd.n + 4 * host_threads() - 1) / (4 * host_threads())), [&](int a0, int a1) {
    std::vector<int> stamp((size_t)gd.n, -1);
    for (int a = a0; a < a1; ++a) {
      const int va = dense[a];
      stamp[a] = a;
      std::vector<int> &out = nbrs[a];
      for (int p = g.xadj[va]; p < g.xadj[va + 1]; ++p) {
        const int u = g.adj[p];
        if (lid[u] >= 0) { if (stamp[lid[u]] != a) { stamp[lid[u]] = a; out.push_back(lid[u]); } continue; }
        if (!few[u]) continue;
        for (int q = g.xadj[u]; q < g.xadj[u + 1]; ++q) {
          const int h = lid[g.adj[q]];
          if (h >= 0 && stamp[h] != a) { stamp[h] = a; out.push_back(h); }
        }
      }
      std::sort(out.begin(), out.end());
    }
  });
  gd.xadj.assign(gd.n + 1, 0);
  for (int a = 0; a < gd.n; ++a) gd.xadj[a + 1] = gd.xadj[a] + (int)nbrs[a].size();
  gd.adj.resize(gd.xadj[gd.n]);
  for (int a = 0; a < gd.n; ++a) std::copy(nbrs[a].begin(), nbrs[a].end(), gd.adj.begin() + gd.xadj[a]);
  OrderingOptions o2 = opt;
  o2.dense_factor = 0;                               // one level of hub removal
  std::vector<int> pd;
  nested_dissection(gd, o2, pd);
  for (int i : pd) perm.push_back(dense[i]);
}

}  // namespace fgo
