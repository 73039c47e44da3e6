// Developer tool: how dense are (target panel, source task) interactions of the accumulate?  For every panel level of the
// cfg-2 style graph: external block updates grouped by (target task, source task); per pair the touched target rows rP,
// target columns cP and source columns nj; density = updates / (rP * cP * nj); MFMA padding estimate.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#include <algorithm>
#include "../include/fgo.h"
#include "../graph_slam_amd/csrc/fgo_internal.hpp"
using namespace fgo;
int main(int argc, char **argv) {
  int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
  int lookback = 5, nloop = 4;
  int64_t maxE = N * (1 + lookback + nloop);
  std::vector<double> init(N * 7), truth(N * 7), meas(maxE * 7), info(maxE * 21);
  std::vector<int64_t> ei(maxE), ej(maxE);
  int64_t E = fgo_synth_manhattan3d(N, lookback, nloop, argc > 2 ? atoll(argv[2]) : 42, 0.02, 0.005, init.data(), truth.data(), ei.data(), ej.data(), meas.data(), info.data(), maxE);
  int n = (int)N - 1;
  std::vector<std::pair<int,int>> pr;
  for (int64_t e = 0; e < E; ++e) { int a = (int)ei[e] - 1, b = (int)ej[e] - 1; if (a < 0 || b < 0 || a == b) continue; pr.push_back({std::min(a,b), std::max(a,b)}); }
  std::sort(pr.begin(), pr.end()); pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
  BlockGraph g; g.n = n; g.xadj.assign(n + 1, 0);
  for (auto &p : pr) { g.xadj[p.first + 1]++; g.xadj[p.second + 1]++; }
  for (int i = 0; i < n; ++i) g.xadj[i + 1] += g.xadj[i];
  g.adj.resize(g.xadj[n]); { std::vector<int> f(g.xadj.begin(), g.xadj.end() - 1); for (auto &p : pr) { g.adj[f[p.first]++] = p.second; g.adj[f[p.second]++] = p.first; } }
  std::vector<int> perm; OrderingOptions opt;
  nested_dissection(g, opt, perm);
  Symbolic S; build_symbolic(g, perm, 5000, (int64_t)1 << 60, S);
  printf("N=%lld E=%lld nnzL %lld nops %lld levels %zu tasks %zu\n", (long long)N, (long long)E, (long long)S.nnzL, (long long)S.nops, S.level_ptr.size() - 1, S.task_ptr.size() - 1);
  std::vector<int> task_of(n), tlevel(S.task_ptr.size() - 1);
  for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l)
    for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) { tlevel[t] = (int)l; for (int c = S.task_ptr[t]; c < S.task_ptr[t + 1]; ++c) task_of[S.task_cols[c]] = t; }
  struct PairStat { int64_t nupd = 0; std::set<int> rows, cols, src; };
  for (size_t l = 1; l + 1 < S.level_ptr.size(); ++l) {
    int64_t tot_upd = 0, tot_dense = 0, tot_tiles = 0, npairs = 0, pairs_big = 0, upd_big = 0, dense_big = 0;
    int64_t tot_src_blocks_distinct = 0;
    int64_t from_leaf = 0;
    double sum_r = 0, sum_c = 0, sum_j = 0;
    for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
      std::map<int, PairStat> by_src;
      for (int c = S.task_ptr[t]; c < S.task_ptr[t + 1]; ++c) {
        const int k = S.task_cols[c];
        for (int64_t u = S.colptr[k]; u < S.colptr[k + 1]; ++u)
          for (int64_t o = S.op_ptr[u]; o < S.op_mid[u]; ++o) {
            const int j = S.blkcol[S.op_a[o]];
            PairStat &ps = by_src[task_of[j]];
            ps.nupd++; ps.rows.insert(S.rowidx[u]); ps.cols.insert(k); ps.src.insert(j);
          }
      }
      for (auto &kv : by_src) {
        const PairStat &ps = kv.second;
        const int64_t r = ps.rows.size(), c = ps.cols.size(), j = ps.src.size();
        tot_upd += ps.nupd; tot_dense += r * c * j; ++npairs;
        sum_r += r; sum_c += c; sum_j += j;
        tot_tiles += ((6 * r + 15) / 16) * ((6 * c + 15) / 16) * ((6 * j + 3) / 4);   // MFMA 16x16x4 instructions
        if (tlevel[kv.first] == 0) from_leaf += ps.nupd;
        if (ps.nupd >= 64) { ++pairs_big; upd_big += ps.nupd; dense_big += r * c * j; }
        tot_src_blocks_distinct += (r + c) * j;   // upper bound of the blocks a dense formulation reads (A rows + B rows)
      }
    }
    if (npairs == 0) continue;
    printf("level %2zu tasks %5d pairs %7lld upd %9lld (from leaf %3.0f%%) dense-vol %10lld density %.2f | avg r %.1f c %.1f nj %.1f | mfma instr %9lld (vs %9lld ideal) | pairs>=64upd %6lld hold %3.0f%% upd, density %.2f | blocks read dense %lld vs gather %lld\n",
           l, S.level_ptr[l + 1] - S.level_ptr[l], (long long)npairs, (long long)tot_upd, 100.0 * from_leaf / tot_upd, (long long)tot_dense, (double)tot_upd / tot_dense,
           sum_r / npairs, sum_c / npairs, sum_j / npairs, (long long)tot_tiles, (long long)(tot_upd * 36 * 6 / (16 * 16 * 4)), (long long)pairs_big, 100.0 * upd_big / tot_upd,
           dense_big ? (double)upd_big / dense_big : 0.0, (long long)tot_src_blocks_distinct, (long long)(2 * tot_upd));
  }
  return 0;
}
