// Developer tool: structure phase (ordering + symbolic) of a block graph read from a binary file: int32 n, int64 m, m x (int32 a, int32 b)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../graph_slam_amd/csrc/fgo_internal.hpp"
using namespace fgo;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
  int n; long long m; if (fread(&n, 4, 1, f) != 1 || fread(&m, 8, 1, f) != 1) return 1;
  std::vector<std::pair<int,int>> pr(m);
  for (long long k = 0; k < m; ++k) { int ab[2]; if (fread(ab, 4, 2, f) != 2) return 1; pr[k] = {std::min(ab[0], ab[1]), std::max(ab[0], ab[1])}; }
  fclose(f);
  std::sort(pr.begin(), pr.end()); pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
  BlockGraph g; g.n = n; g.xadj.assign(n + 1, 0);
  for (auto &p : pr) { g.xadj[p.first + 1]++; g.xadj[p.second + 1]++; }
  for (int i = 0; i < n; ++i) g.xadj[i + 1] += g.xadj[i];
  g.adj.resize(g.xadj[n]); { std::vector<int> fl(g.xadj.begin(), g.xadj.end() - 1); for (auto &p : pr) { g.adj[fl[p.first]++] = p.second; g.adj[fl[p.second]++] = p.first; } }
  std::vector<int> perm; OrderingOptions opt;
  double t0 = now(); nested_dissection(g, opt, perm); double t1 = now();
  Symbolic S; build_symbolic(g, perm, 5000, (long long)1 << 60, S);
  printf("n %d pairs %zu: ordering %.2f s symbolic %.2f s nnzL %lld blocks (%.1f GB) ops %lld levels %zu tasks %zu etree height %d\n", n, pr.size(), t1 - t0, now() - t1,
         (long long)S.nnzL, 288e-9 * S.nnzL, (long long)S.nops, S.level_ptr.size() - 1, S.task_ptr.size() - 1, S.etree_height);
  return 0;
}
