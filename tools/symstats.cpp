// Developer tool: structure statistics of the symbolic phase on a synthetic graph.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include <algorithm>
#include "../include/fgo.h"
#include "../graph_slam_amd/csrc/fgo_internal.hpp"
using namespace fgo;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  int64_t N = argc > 1 ? atoll(argv[1]) : 10000;
  int lookback = argc > 2 ? atoi(argv[2]) : 5, nloop = argc > 3 ? atoi(argv[3]) : 4;
  int leaf = argc > 4 ? atoi(argv[4]) : 64;
  int64_t limit = argc > 5 ? atoll(argv[5]) : 20000;
  int64_t maxE = N * (1 + lookback + nloop);
  std::vector<double> init(N * 7), truth(N * 7), meas(maxE * 7), info(maxE * 21);
  std::vector<int64_t> ei(maxE), ej(maxE);
  double t0 = now();
  int64_t E = fgo_synth_manhattan3d(N, lookback, nloop, std::getenv("FGO_SYNTH_SEED") ? atoll(std::getenv("FGO_SYNTH_SEED")) : 42, 0.02, 0.005, init.data(), truth.data(), ei.data(), ej.data(), meas.data(), info.data(), maxE);
  if (const char *ef = std::getenv("FGO_EDGES")) {     // other topologies: int64 file [n, E, ei[E], ej[E]] (tools/dump_edges.py)
    FILE *f = fopen(ef, "rb"); int64_t hd[2];
    if (!f || fread(hd, 8, 2, f) != 2) { fprintf(stderr, "cannot read %s\n", ef); return 2; }
    N = hd[0]; E = hd[1]; ei.resize(E); ej.resize(E);
    if (fread(ei.data(), 8, E, f) != (size_t)E || fread(ej.data(), 8, E, f) != (size_t)E) return 2;
    fclose(f);
  }
  printf("N=%lld E=%lld synth %.2fs\n", (long long)N, (long long)E, now() - t0);
  int64_t far = 0; for (int64_t e = 0; e < E; ++e) if (ej[e] - ei[e] > lookback + nloop + 1) ++far;
  printf("edges spanning > window: %lld\n", (long long)far);
  // block graph over free poses (pose 0 fixed)
  int n = (int)N - 1;
  std::vector<std::pair<int,int>> pr;
  for (int64_t e = 0; e < E; ++e) { int a = (int)ei[e] - 1, b = (int)ej[e] - 1; if (a < 0 || b < 0 || a == b) continue; pr.push_back({std::min(a,b), std::max(a,b)}); }
  if (const char *lp = std::getenv("FGO_LAPS")) {       // "T,step": a second lap along the same corridor -- every step-th pose i >= T also sees pose i - T
    int T = 0, step = 10; sscanf(lp, "%d,%d", &T, &step);
    size_t added = 0;
    for (int i = T; T > 0 && i < n; ++i) if (i % step == 0) { pr.push_back({i - T, i}); ++added; }
    printf("lap closures: %zu edges (i - %d, i), every %d-th pose\n", added, T, step);
  }
  if (const char *sh = std::getenv("FGO_SHUFFLE")) {     // the same graph under a random relabelling of its vertices (ids that are NOT creation order)
    std::vector<int> p(n);
    for (int i = 0; i < n; ++i) p[i] = i;
    uint64_t st = 0x9E3779B97F4A7C15ull * (uint64_t)(atoll(sh) + 1);
    for (int i = n - 1; i > 0; --i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; std::swap(p[i], p[(int)(st % (uint64_t)(i + 1))]); }
    for (auto &e : pr) { const int a = p[e.first], b = p[e.second]; e = {std::min(a, b), std::max(a, b)}; }
    printf("vertices relabelled at random (seed %s)\n", sh);
  }
  std::sort(pr.begin(), pr.end()); pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
  BlockGraph g; g.n = n; g.xadj.assign(n + 1, 0);
  for (auto &p : pr) { g.xadj[p.first + 1]++; g.xadj[p.second + 1]++; }
  for (int i = 0; i < n; ++i) g.xadj[i + 1] += g.xadj[i];
  g.adj.resize(g.xadj[n]); { std::vector<int> f(g.xadj.begin(), g.xadj.end() - 1); for (auto &p : pr) { g.adj[f[p.first]++] = p.second; g.adj[f[p.second]++] = p.first; } }
  printf("unique offdiag blocks %zu\n", pr.size());
  std::vector<int> perm; OrderingOptions opt; opt.leaf = leaf; opt.time_side = tune("nd_time_side", opt.time_side); opt.time_weight = tune("nd_time_weight", opt.time_weight);
  if (std::getenv("FGO_ND_TWICE")) { std::vector<int> p2; t0 = now(); nested_dissection(g, opt, p2); printf("ND (first of two) %.3fs\n", now() - t0); }
  t0 = now(); nested_dissection(g, opt, perm); printf("ND %.3fs (perm %zu)\n", now() - t0, perm.size());
  if (const char *pf = std::getenv("FGO_PERM")) {       // an ordering from outside (prototypes): int32 perm[n], perm[k] = vertex eliminated k-th
    FILE *f = fopen(pf, "rb"); std::vector<int> p2(n);
    if (!f || fread(p2.data(), 4, n, f) != (size_t)n) { fprintf(stderr, "cannot read %s\n", pf); return 2; }
    fclose(f); perm.swap(p2); printf("ordering read from %s\n", pf);
  }
  const int world = std::getenv("FGO_WORLD") ? atoi(std::getenv("FGO_WORLD")) : 1;
  if (world > 1) { opt.bal_w = 8.0; nested_dissection(g, opt, perm); }       // (the product's distributed-mode balance weight)
  Symbolic S; t0 = now(); build_symbolic(g, perm, limit, argc > 6 ? atoll(argv[6]) : limit, S, world); printf("symbolic %.2fs\n", now() - t0);
  if (world > 1) {
    // the replicated top of the distributed mode, segment by segment: what an owner-computes split could divide
    printf("distributed, world %d: top columns %d of %d\n", world, n - S.dom_col0[world], n);
    int64_t tot_ops = 0;
    for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l) {
      if (S.seg_group[l] != world || S.level_ptr[l + 1] == S.level_ptr[l]) continue;
      int64_t cols = 0, rows = 0, ext = 0, blks = 0;
      for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
        cols += S.task_ptr[t + 1] - S.task_ptr[t];
        const int pn = S.task_panel[t];
        if (pn >= 0) rows += S.prow_ptr[pn + 1] - S.prow_ptr[pn];
        for (int c = S.task_ptr[t]; c < S.task_ptr[t + 1]; ++c) blks += S.colptr[S.task_cols[c] + 1] - S.colptr[S.task_cols[c]];
      }
      for (int64_t a = S.acc_ptr[l]; a < S.acc_ptr[l + 1]; ++a) { const int64_t t = S.acc_targets[a]; ext += S.op_mid[t] - S.op_ptr[t]; }
      tot_ops += ext;
      printf(" top segment %zu (dependency level %zu): panels %d cols %lld rows %lld blocks of L %lld (%.1f MB) acc targets %lld ext ops %lld%s\n", l, l / (size_t)(world + 1),
             S.level_ptr[l + 1] - S.level_ptr[l], (long long)cols, (long long)rows, (long long)blks, blks * 288e-6, (long long)(S.acc_ptr[l + 1] - S.acc_ptr[l]), (long long)ext, S.level_panel[l] ? "" : " [generic]");
    }
    printf(" top: %lld external ops in all\n", (long long)tot_ops);
  }
  if (std::getenv("FGO_G2_STATS") && !S.g2_lvl.empty()) {
    for (size_t l = 0; l + 1 < S.g2_lvl.size(); ++l) {
      const int64_t g0 = S.g2_lvl[l], g1 = S.g2_lvl[l + 1];
      if (g1 <= g0) continue;
      int64_t ent = S.g2_ptr[g1] - S.g2_ptr[g0], filled = 0, tg = 0;
      for (int64_t e = S.g2_ptr[g0]; e < S.g2_ptr[g1]; ++e) for (int q = 0; q < ACC2_G; ++q) filled += S.g2_a[e * ACC2_G + q] >= 0 && S.g2_a[e * ACC2_G + q] != (int)S.nnzL;
      for (int64_t g = g0; g < g1; ++g) for (int q = 0; q < ACC2_G; ++q) tg += S.g2_tgt[g * ACC2_G + q] >= 0;
      printf("[g2] level %zu: %lld groups, %lld targets (%.1f per group), %lld entries (%.1f per group), filled %.1f %% of entries x 10, %.1f %% of entries x targets\n", l, (long long)(g1 - g0), (long long)tg,
             (double)tg / (g1 - g0), (long long)ent, (double)ent / (g1 - g0), 100.0 * filled / (10.0 * ent), 100.0 * filled / ((double)ent * tg / (g1 - g0)));
    }
  }
  printf("nnzL blocks %lld (%.1fx H lower) nops %lld etree_height %d max_col_blocks %d tasks %zu levels %zu\n",
    (long long)S.nnzL, (double)S.nnzL / (pr.size() + n), (long long)S.nops, S.etree_height, S.max_col_blocks, S.task_ptr.size() - 1, S.level_ptr.size() - 1);
  if (std::getenv("FGO_CRIT")) {
    // the critical path in LEVELS: from the root's task down, always to the child task with the highest level
    const int ntask = (int)S.task_ptr.size() - 1;
    std::vector<int> tof(n), tlev(ntask);
    for (int t = 0; t < ntask; ++t) for (int q = S.task_ptr[t]; q < S.task_ptr[t + 1]; ++q) tof[S.task_cols[q]] = t;
    for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l) for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) tlev[t] = (int)l;
    std::vector<std::vector<int>> kids(ntask);
    for (int k = 0; k < n; ++k) { const int p = S.parent[k]; if (p >= 0 && tof[p] != tof[k]) kids[tof[p]].push_back(tof[k]); }
    int t = tof[n - 1], total = 0;
    printf("critical path (task level: columns, child tasks at level-1 / all):");
    while (true) {
      const int m = S.task_ptr[t + 1] - S.task_ptr[t];
      int best = -1, nk = 0, ntop = 0;
      for (int c : kids[t]) { ++nk; if (best < 0 || tlev[c] > tlev[best] || (tlev[c] == tlev[best] && S.task_ptr[c + 1] - S.task_ptr[c] > S.task_ptr[best + 1] - S.task_ptr[best])) best = c; }
      for (int c : kids[t]) ntop += tlev[c] == tlev[t] - 1;
      printf(" %d:%d(%d/%d)", tlev[t], m, ntop, nk);
      if (m < 16 && tlev[t] > 0) {          // why was this panel not continued?  its top column's parent and that column's first-in-task status
        const int top = S.task_cols[S.task_ptr[t + 1] - 1], p = S.parent[top];
        if (p >= 0) {
          const int tp = tof[p];
          const bool first = S.task_cols[S.task_ptr[tp]] == p;
          // children of p
          printf("[parent col %s of a %d-col task;", first ? "FIRST" : "inner", S.task_ptr[tp + 1] - S.task_ptr[tp]);
          for (int k = 0; k < n; ++k) if (S.parent[k] == p) printf(" kid task lvl %d cols %d%s", tlev[tof[k]], S.task_ptr[tof[k] + 1] - S.task_ptr[tof[k]], tof[k] == tp ? "*" : "");
          printf("]");
        }
      }
      total += m;
      if (best < 0) break;
      t = best;
    }
    printf("\ncolumns on it: %d\n", total);
  }
  if (std::getenv("FGO_MF_EST")) {
    // what a multifrontal organisation of the panel levels would move and compute: per front (= panel: m columns, r block rows
    // below the triangle) an update matrix of r (r + 1) / 2 blocks, formed densely with m r (r + 1) / 2 block products -- against
    // the block products the sparse column patterns actually need (a column that holds n of the r rows feeds n (n + 1) / 2)
    printf("multifrontal estimate per panel level: fronts, mean m, mean r, update-matrix blocks, dense block products, sparse block products (today)\n");
    int64_t tu = 0, td = 0, ts = 0;
    for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l) {
      if (!S.level_panel[l]) continue;
      int64_t nf = 0, sm = 0, sr = 0, ub = 0, dn = 0, sp = 0;
      for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
        const int pn = S.task_panel[t];
        if (pn < 0) continue;
        const int m = S.task_ptr[t + 1] - S.task_ptr[t];
        const int64_t r = S.prow_ptr[pn + 1] - S.prow_ptr[pn];
        ++nf; sm += m; sr += r; ub += r * (r + 1) / 2; dn += (int64_t)m * r * (r + 1) / 2;
        for (int k = 0; k < m; ++k) {
          int64_t nk = 0;
          for (int q = S.prow_ptr[pn]; q < S.prow_ptr[pn + 1]; ++q) nk += S.prow_blk[S.row_off(q) + k] >= 0;
          sp += nk * (nk + 1) / 2;
        }

      }
      if (nf == 0) continue;
      tu += ub; td += dn; ts += sp;
      if (l < 8 || l % 4 == 0) printf(" level %zu: %lld fronts, m %.1f, r %.1f, U blocks %lld (%.0f MB), dense %lld, sparse %lld (x%.2f)\n", l, (long long)nf, (double)sm / nf, (double)sr / nf, (long long)ub, ub * 288e-6, (long long)dn, (long long)sp, sp ? (double)dn / sp : 0.0);
    }
    printf(" all panel levels: U blocks %lld (%.0f MB), dense products %lld, sparse products %lld (x%.2f)\n", (long long)tu, tu * 288e-6, (long long)td, (long long)ts, ts ? (double)td / ts : 0.0);
  }
  if (const char *fd = std::getenv("FGO_FRONT_DUMP")) {
    // Real front shapes for tools/front_bench.hip (round 6: the go / no-go microbenchmark of a dense-front update on the wide levels).
    // Per panel level: its fronts (= panels: m columns, r block rows below the triangle, the block of L behind every (row, column) or -1,
    // where the front's update matrix -- r (r + 1) / 2 blocks, packed lower triangle -- starts in the U buffer) and its extend-add targets
    // (every block of L in the level's columns that an update matrix of a LOWER panel level covers: does it start from an H block, and the
    // U blocks it receives, in source-panel order).  Little-endian int32 / int64 as written below.
    FILE *f = fopen(fd, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", fd); return 2; }
    const int nl = (int)S.level_ptr.size() - 1;
    std::vector<int> col_level(n, 0);
    for (int l = 0; l < nl; ++l) for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) for (int c = S.task_ptr[t]; c < S.task_ptr[t + 1]; ++c) col_level[S.task_cols[c]] = l;
    auto blk_of = [&](int row, int col) -> int64_t {
      if (row == col) return S.colptr[col];
      const int *b = S.rowidx.data() + S.colptr[col] + 1, *e = S.rowidx.data() + S.colptr[col + 1];
      const int *q = std::lower_bound(b, e, row);
      return (q != e && *q == row) ? (int64_t)(q - S.rowidx.data()) : -1;
    };
    std::vector<int64_t> ubase((size_t)S.n_panels + 1, 0);
    std::vector<std::vector<std::pair<int64_t, int64_t>>> lvl_ops((size_t)nl);      // per target level: (target block, U block)
    int64_t nu = 0;
    for (int l = 0; l < nl; ++l) {
      if (!S.level_panel[l]) continue;
      for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
        const int pn = S.task_panel[t];
        ubase[pn] = nu;
        const int r0 = S.prow_ptr[pn], r = S.prow_ptr[pn + 1] - r0;
        for (int a = 0; a < r; ++a)
          for (int b = 0; b <= a; ++b) {
            const int i = S.prow_idx[r0 + a], j = S.prow_idx[r0 + b];
            const int64_t tb = blk_of(i, j);
            if (tb < 0) { fprintf(stderr, "front dump: block (%d, %d) missing\n", i, j); return 3; }
            lvl_ops[(size_t)col_level[j]].push_back({tb, nu + (int64_t)a * (a + 1) / 2 + b});
          }
        nu += (int64_t)r * (r + 1) / 2;
      }
    }
    const int64_t hdr[4] = {(int64_t)nl, S.nnzL, nu, (int64_t)n};
    fwrite(hdr, 8, 4, f);
    for (int l = 0; l < nl; ++l) {
      int32_t nf = 0;
      if (S.level_panel[l]) nf = S.level_ptr[l + 1] - S.level_ptr[l];
      fwrite(&nf, 4, 1, f);
      for (int t = S.level_ptr[l]; nf > 0 && t < S.level_ptr[l + 1]; ++t) {
        const int pn = S.task_panel[t];
        const int32_t m = S.task_ptr[t + 1] - S.task_ptr[t], r = S.prow_ptr[pn + 1] - S.prow_ptr[pn];
        fwrite(&m, 4, 1, f); fwrite(&r, 4, 1, f); fwrite(&ubase[pn], 8, 1, f);
        fwrite(S.prow_blk.data() + S.row_off(S.prow_ptr[pn]), 4, (size_t)r * PANEL_MAX, f);
      }
      // targets of the level, grouped by block (stable: U blocks of a target stay in source-panel order)
      auto &ops = lvl_ops[(size_t)l];
      std::stable_sort(ops.begin(), ops.end(), [](const std::pair<int64_t, int64_t> &a, const std::pair<int64_t, int64_t> &b) { return a.first < b.first; });
      std::vector<int64_t> tgt, ptr(1, 0), ub;
      std::vector<int32_t> hasH;
      for (size_t q = 0; q < ops.size(); ++q) {
        if (q == 0 || ops[q].first != ops[q - 1].first) {
          if (q) ptr.push_back((int64_t)ub.size());
          tgt.push_back(ops[q].first);
          const int col = S.blkcol[ops[q].first], row = S.rowidx[ops[q].first];
          bool h = row == col;
          const int vr = S.perm[row], vc = S.perm[col];
          for (int e = g.xadj[vc]; !h && e < g.xadj[vc + 1]; ++e) h = g.adj[e] == vr;
          hasH.push_back(h ? 1 : 0);
        }
        ub.push_back(ops[q].second);
      }
      ptr.push_back((int64_t)ub.size());
      const int64_t nt = (int64_t)tgt.size(), no = (int64_t)ub.size();
      fwrite(&nt, 8, 1, f); fwrite(&no, 8, 1, f);
      fwrite(tgt.data(), 8, tgt.size(), f); fwrite(hasH.data(), 4, hasH.size(), f); fwrite(ptr.data(), 8, nt + 1, f); fwrite(ub.data(), 8, ub.size(), f);
      if (nf > 0 || nt > 0) printf("[front dump] level %d: %d fronts, %lld extend-add targets, %lld U blocks into them\n", l, nf, (long long)nt, (long long)no);
    }
    fclose(f);
    printf("[front dump] %lld U blocks (%.0f MB) -> %s\n", (long long)nu, nu * 288e-6, fd);
  }
  if (std::getenv("FGO_LEAF_HIST") && !S.level_leaf.empty() && S.level_leaf[0]) {
    std::vector<int> hb(12, 0), ho(12, 0);
    int64_t sb = 0, so = 0; int mb = 0, mo = 0;
    for (int t = S.level_ptr[0]; t < S.level_ptr[1]; ++t) {
      const int k0 = S.task_cols[S.task_ptr[t]], k1 = S.task_cols[S.task_ptr[t + 1] - 1];
      const int64_t nb2 = S.colptr[k1 + 1] - S.colptr[k0], no = S.op_ptr[S.colptr[k1 + 1]] - S.op_ptr[S.colptr[k0]];
      hb[std::min<int64_t>(11, nb2 / 24)]++; ho[std::min<int64_t>(11, no / 400)]++; sb += nb2; so += no; mb = std::max<int>(mb, nb2); mo = std::max<int>(mo, no);
    }
    const int nt = S.level_ptr[1] - S.level_ptr[0];
    printf("leaf level: %d tasks, blocks mean %.1f max %d, ops mean %.1f max %d\n blocks / 24:", nt, (double)sb / nt, mb, (double)so / nt, mo);
    for (int v : hb) printf(" %d", v);
    printf("\n ops / 400:");
    for (int v : ho) printf(" %d", v);
    printf("\n");
  }
  // per level: tasks, max task work, total work
  std::vector<int64_t> colwork(n);
  for (int k = 0; k < n; ++k) colwork[k] = (S.op_ptr[S.colptr[k+1]] - S.op_ptr[S.colptr[k]]) + 2 * (S.colptr[k+1] - S.colptr[k]);
  int64_t crit = 0; 
  for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l) {
    int64_t mx = 0, tot = 0; int mxcols = 0;
    for (int t = S.level_ptr[l]; t < S.level_ptr[l+1]; ++t) { int64_t w = 0; for (int c = S.task_ptr[t]; c < S.task_ptr[t+1]; ++c) w += colwork[S.task_cols[c]]; mx = std::max(mx, w); tot += w; mxcols = std::max(mxcols, S.task_ptr[t+1]-S.task_ptr[t]); }
    crit += mx;
    if (std::getenv("FGO_ALL_LEVELS") && S.level_panel[l]) {
      int64_t cols = 0, rows = 0, maxrows = 0, ext = 0, tg = 0;
      for (int t = S.level_ptr[l]; t < S.level_ptr[l+1]; ++t) { const int p = S.task_panel[t]; cols += S.task_ptr[t+1]-S.task_ptr[t]; rows += S.prow_ptr[p+1]-S.prow_ptr[p]; maxrows = std::max<int64_t>(maxrows, S.prow_ptr[p+1]-S.prow_ptr[p]); }
      for (int64_t a = S.acc_ptr[l]; a < S.acc_ptr[l + 1]; ++a) { const int64_t t = S.acc_targets[a]; ++tg; ext += S.op_mid[t] - S.op_ptr[t]; }
      printf(" level %zu: panels %d cols %lld rows %lld (max %lld) acc targets %lld ext ops %lld\n", l, S.level_ptr[l+1]-S.level_ptr[l], (long long)cols, (long long)rows, (long long)maxrows, (long long)tg, (long long)ext);
    } else
    if (l < 6 || l + 6 >= S.level_ptr.size() || l % 50 == 0) printf(" level %zu: tasks %d maxwork %lld totwork %lld maxcols %d\n", l, S.level_ptr[l+1]-S.level_ptr[l], (long long)mx, (long long)tot, mxcols);
  }
  {
    int npl = 0; int64_t prow = S.prow_idx.size(), maxrows = 0, ext_acc = 0, ext_ops = 0, maxops = 0;
    for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l) if (S.level_panel[l]) {
      ++npl;
      for (int64_t a = S.acc_ptr[l]; a < S.acc_ptr[l + 1]; ++a) { const int64_t t = S.acc_targets[a]; ++ext_acc; ext_ops += S.op_mid[t] - S.op_ptr[t]; maxops = std::max(maxops, S.op_mid[t] - S.op_ptr[t]); }
    }
    for (int p = 0; p < S.n_panels; ++p) maxrows = std::max<int64_t>(maxrows, S.prow_ptr[p + 1] - S.prow_ptr[p]);
    printf("panels %d, panel levels %d of %zu, panel rows %lld (max %lld per panel), row chunks %zu, fwd chunks %zu; panel-level acc targets %lld ext ops %lld (max/target %lld)\n",
           S.n_panels, npl, S.level_ptr.size() - 1, (long long)prow, (long long)maxrows, S.pchunk_panel.size(), S.fchunk_col.size(), (long long)ext_acc, (long long)ext_ops, (long long)maxops);
  }
  if (!S.op_a.empty()) {   // who feeds the external updates: level-0 subtree columns or panel columns?
    std::vector<int> col_level(n, 0);
    for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l)
      for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t)
        for (int c = S.task_ptr[t]; c < S.task_ptr[t + 1]; ++c) col_level[S.task_cols[c]] = (int)l;
    int64_t ext0 = 0, extp = 0, int0 = 0, intp = 0;
    for (int64_t t = 0; t < S.nnzL; ++t) {
      const int lt = col_level[S.blkcol[t]];
      for (int64_t o = S.op_ptr[t]; o < S.op_ptr[t + 1]; ++o) {
        const int ls = col_level[S.blkcol[S.op_a[o]]];
        if (o < S.op_mid[t]) { if (ls == 0) ++ext0; else ++extp; } else { if (lt == 0) ++int0; else ++intp; }
      }
    }
    printf("ops: external from level-0 columns %lld, external from panel-level columns %lld, internal level-0 %lld, internal panel %lld\n",
           (long long)ext0, (long long)extp, (long long)int0, (long long)intp);
  }
  {   // level-0 tasks: contiguous column ranges?  blocks / ops per task (LDS-resident subtree kernel eligibility)
    int contig = 0, n0 = S.level_ptr[1] - S.level_ptr[0];
    int64_t maxblk = 0, maxops = 0, sumblk = 0, maxcolblk = 0; int maxcols = 0;
    std::vector<int64_t> blks;
    for (int t = S.level_ptr[0]; t < S.level_ptr[1]; ++t) {
      const int c0 = S.task_ptr[t], c1 = S.task_ptr[t + 1];
      bool ok = true;
      for (int c = c0; c + 1 < c1; ++c) ok = ok && S.task_cols[c + 1] == S.task_cols[c] + 1;
      contig += ok;
      const int64_t b0 = S.colptr[S.task_cols[c0]], b1 = S.colptr[S.task_cols[c1 - 1] + 1];
      int64_t nb2 = 0; for (int c = c0; c < c1; ++c) { nb2 += S.colptr[S.task_cols[c] + 1] - S.colptr[S.task_cols[c]]; maxcolblk = std::max(maxcolblk, S.colptr[S.task_cols[c] + 1] - S.colptr[S.task_cols[c]]); }
      (void)b0; (void)b1;
      maxblk = std::max(maxblk, nb2); sumblk += nb2; blks.push_back(nb2);
      maxops = std::max(maxops, S.op_ptr[S.colptr[S.task_cols[c1 - 1] + 1]] - S.op_ptr[S.colptr[S.task_cols[c0]]]);
      maxcols = std::max(maxcols, c1 - c0);
    }
    std::sort(blks.begin(), blks.end());
    printf("level 0: %d tasks, %d contiguous; blocks/task mean %.0f median %lld p90 %lld max %lld; max ops/task %lld; max cols %d; max blocks in a column %lld\n", n0, contig,
           (double)sumblk / n0, (long long)blks[blks.size() / 2], (long long)blks[blks.size() * 9 / 10], (long long)maxblk, (long long)maxops, maxcols, (long long)maxcolblk);
  }
  {   // supernodal accumulate: (target panel, source panel) pairs
    std::vector<int> col_panel(n, -1), col_lvl(n, 0);
    for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l)
      for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t)
        for (int c = S.task_ptr[t]; c < S.task_ptr[t + 1]; ++c) { col_panel[S.task_cols[c]] = S.level_panel[l] ? S.task_panel[t] : -1; col_lvl[S.task_cols[c]] = (int)l; }
    int64_t pairs = 0, incid = 0, strips = 0, maxsrc = 0;
    std::vector<int> nsrc(S.n_panels, 0);
    for (int d = 0; d < S.n_panels; ++d) {
      if (!S.level_panel[col_lvl[S.task_cols[S.task_ptr[S.panel_task[d]]]]]) continue;
      int last = -1; const int r0 = S.prow_ptr[d], r1 = S.prow_ptr[d + 1];
      for (int r = r0; r < r1; ++r) {
        const int p = col_panel[S.prow_idx[r]];
        if (p < 0 || p == last) continue;
        last = p; ++pairs; nsrc[p]++;
        incid += r1 - r;                       // rows of D at or above P's first hit
        strips += (6 * (r1 - r) + 15) / 16;
      }
    }
    for (int p = 0; p < S.n_panels; ++p) maxsrc = std::max<int64_t>(maxsrc, nsrc[p]);
    printf("supernodal accumulate: %lld (target, source) panel pairs, %lld block-row incidences, ~%lld strip x source units, max sources per target %lld\n",
           (long long)pairs, (long long)incid, (long long)strips, (long long)maxsrc);
  }
  for (size_t l = 1; l < 8 && l + 1 < S.level_ptr.size(); ++l) {   // panel widths of the lowest panel levels
    int hist[17] = {0};
    for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) hist[std::min(16, S.task_ptr[t + 1] - S.task_ptr[t])]++;
    printf(" level %zu panel widths:", l);
    for (int m = 1; m <= 16; ++m) printf(" %d", hist[m]);
    printf("\n");
  }
  {   // structure checksum: must not depend on FGO_HOST_THREADS (tests/test_symbolic_threads.py)
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t bytes) { const unsigned char *c = (const unsigned char *)p; for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; } };
    mix(S.perm.data(), S.perm.size() * sizeof(int)); mix(S.colptr.data(), S.colptr.size() * sizeof(int64_t));
    mix(S.rowidx.data(), S.rowidx.size() * sizeof(int)); mix(S.op_ptr.data(), S.op_ptr.size() * sizeof(int64_t));
    mix(S.op_mid.data(), S.op_mid.size() * sizeof(int64_t)); mix(S.op_a.data(), S.op_a.size() * sizeof(int));
    mix(S.op_b.data(), S.op_b.size() * sizeof(int)); mix(S.task_ptr.data(), S.task_ptr.size() * sizeof(int));
    mix(S.task_cols.data(), S.task_cols.size() * sizeof(int)); mix(S.level_ptr.data(), S.level_ptr.size() * sizeof(int));
    mix(S.acc_targets.data(), S.acc_targets.size() * sizeof(int)); mix(S.row_blk.data(), S.row_blk.size() * sizeof(int));
    mix(S.row_col.data(), S.row_col.size() * sizeof(int)); mix(S.row_mid.data(), S.row_mid.size() * sizeof(int64_t));
    mix(S.ptri_blk.data(), S.ptri_blk.size() * sizeof(int)); mix(S.prow_blk.data(), S.prow_blk.size() * sizeof(int));
    mix(S.prow_idx.data(), S.prow_idx.size() * sizeof(int));
    printf("structure checksum %016llx\n", (unsigned long long)h);
    h = 1469598103934665603ull;
    mix(S.g2_lvl.data(), S.g2_lvl.size() * sizeof(int64_t)); mix(S.g2_tgt.data(), S.g2_tgt.size() * sizeof(int)); mix(S.g2_ptr.data(), S.g2_ptr.size() * sizeof(int64_t));
    mix(S.g2_b.data(), S.g2_b.size() * sizeof(int)); mix(S.g2_a.data(), S.g2_a.size() * sizeof(int));
    printf("column-group checksum %016llx\n", (unsigned long long)h);
    h = 1469598103934665603ull;
    mix(S.ride_items.data(), S.ride_items.size() * sizeof(RideItem)); mix(S.ride_ptr.data(), S.ride_ptr.size() * sizeof(int)); mix(S.acc_start.data(), S.acc_start.size() * sizeof(int64_t));
    printf("rider checksum %016llx\n", (unsigned long long)h);
  }
  {   // the critical path of the level schedule, top down: width of the task at every level
    std::vector<int> task_of_col(n, -1), tlevel(S.task_ptr.size() - 1, 0);
    for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l)
      for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) { tlevel[t] = (int)l; for (int c = S.task_ptr[t]; c < S.task_ptr[t + 1]; ++c) task_of_col[S.task_cols[c]] = t; }
    // children tasks of each task
    std::vector<std::vector<int>> kids(S.task_ptr.size() - 1);
    for (int k = 0; k < n; ++k) { const int p = S.parent[k]; if (p >= 0 && task_of_col[p] != task_of_col[k]) kids[task_of_col[p]].push_back(task_of_col[k]); }
    int t = -1, top = -1;
    for (size_t q = 0; q + 1 < S.task_ptr.size(); ++q) if (tlevel[q] > top) { top = tlevel[q]; t = (int)q; }
    printf("critical path (level:columns):");
    while (t >= 0) {
      printf(" %d:%d", tlevel[t], S.task_ptr[t + 1] - S.task_ptr[t]);
      int best = -1;
      for (int c : kids[t]) if (best < 0 || tlevel[c] > tlevel[best]) best = c;
      t = best;
    }
    printf("\n");
  }
  printf("critical-path work (sum of per-level max) %lld ; total %lld\n", (long long)crit, (long long)(S.nops + 2*S.nnzL));
  return 0;
}
