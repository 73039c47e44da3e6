// Developer tool: MFMA instruction count of the tile accumulate under different chunking / masking schemes (cfg-2 graph).
#include <cstdio>
#include <cstdlib>
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
  int64_t E = fgo_synth_manhattan3d(N, lookback, nloop, 42, 0.02, 0.005, init.data(), truth.data(), ei.data(), ej.data(), meas.data(), info.data(), maxE);
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
  setenv("FGO_ACC_TILE", "0", 1);
  Symbolic S; build_symbolic(g, perm, 5000, (int64_t)1 << 60, S);
  const int nlevels = (int)S.level_ptr.size() - 1;
  for (int CH : {8, 4}) for (int sorted = 0; sorted < 2; ++sorted) {
    printf("---- chunk %d sources, %s\n", CH, sorted ? "sources sorted by first stacked row" : "sources ascending");
    for (int l = 1; l < nlevels; ++l) {
      if (!S.level_panel[l]) continue;
      int64_t ideal = 0, full = 0, masked = 0, strips = 0, chunks_tot = 0, aloads = 0, bloads_masked = 0;
      for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
        const int pn = S.task_panel[t];
        const int c0 = S.task_ptr[t], m = S.task_ptr[t + 1] - c0;
        const int nrows = S.prow_ptr[pn + 1] - S.prow_ptr[pn], nstack = m + nrows;
        std::vector<int> stack, srcs;
        for (int k = 0; k < m; ++k) stack.push_back(S.task_cols[c0 + k]);
        for (int r = 0; r < nrows; ++r) stack.push_back(S.prow_idx[S.prow_ptr[pn] + r]);
        for (int k = 0; k < m; ++k) { const int col = S.task_cols[c0 + k]; for (int64_t e = S.rowptr[col]; e < S.row_mid[col]; ++e) srcs.push_back(S.row_col[e]); }
        std::sort(srcs.begin(), srcs.end()); srcs.erase(std::unique(srcs.begin(), srcs.end()), srcs.end());
        const int ns = (int)srcs.size(); if (!ns) continue;
        // presence[u][s]
        std::vector<std::vector<char>> pres(ns, std::vector<char>(nstack, 0));
        std::vector<int> first(ns, nstack);
        for (int u = 0; u < ns; ++u) {
          const int j = srcs[u]; int s = 0;
          const int *pb = S.rowidx.data() + S.colptr[j] + 1, *pe = S.rowidx.data() + S.colptr[j + 1];
          for (const int *p = std::lower_bound(pb, pe, stack[0]); p != pe; ++p) { while (s < nstack && stack[s] < *p) ++s; if (s == nstack) break; if (stack[s] == *p) { pres[u][s] = 1; first[u] = std::min(first[u], s); } }
        }
        std::vector<int> order(ns); for (int u = 0; u < ns; ++u) order[u] = u;
        if (sorted) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return first[a] < first[b]; });
        // ideal block updates
        for (int u = 0; u < ns; ++u) { int cnt_c = 0; for (int s = 0; s < nstack; ++s) if (pres[u][s]) { if (s < m) ++cnt_c; ideal += (s < m) ? cnt_c : cnt_c; } }
        const int nch = (ns + CH - 1) / CH, nT = (6 * m + 15) / 16, nstrips = (6 * nstack + 15) / 16;
        const int per_tile = CH * 6 / 4;
        for (int I = 0; I < nstrips; ++I) {
          const int sb0 = 16 * I / 6, sb1 = std::min((16 * I + 15) / 6, nstack - 1);
          const int Kmax = sb1 < m ? std::min(nT - 1, (6 * sb1 + 5) / 16) : nT - 1;
          bool anystrip = false;
          for (int ch = 0; ch < nch; ++ch) {
            bool anyA = false;
            for (int w = 0; w < CH && ch * CH + w < ns; ++w) for (int s = sb0; s <= sb1; ++s) anyA |= pres[order[ch * CH + w]][s];
            if (!anyA) continue;
            anystrip = true; ++chunks_tot; ++aloads;
            for (int K = 0; K <= Kmax; ++K) {
              bool anyB = false;
              const int cb0 = 16 * K / 6, cb1 = std::min((16 * K + 15) / 6, m - 1);
              for (int w = 0; w < CH && ch * CH + w < ns; ++w) for (int s = cb0; s <= cb1; ++s) anyB |= pres[order[ch * CH + w]][s];
              full += per_tile;
              if (anyB) { masked += per_tile; ++bloads_masked; }
            }
          }
          strips += anystrip;
        }
      }
      printf("level %2d strips %6lld chunk-visits %8lld  mfma full %9lld masked %9lld ideal(=upd*0.21) %9lld  ratio masked/ideal %.1f\n", l, (long long)strips, (long long)chunks_tot,
             (long long)full, (long long)masked, (long long)(ideal * 216 / 1024), (double)masked / (ideal * 216.0 / 1024));
    }
  }
  return 0;
}
