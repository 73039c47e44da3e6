// Load an SE3 pose graph in .g2o text (VERTEX_SE3:QUAT / FIX / EDGE_SE3:QUAT -- what CGraphG2O::writeG2O and
// g2o's own `optimizer->save` write, g2o/g2o_graph.cpp:279-283), optimise it on the MI355X with the reference's schedule
// (10 x optimize(2), g2o/g2o_graph.cpp:241-252) or a given number of iterations, print chi2 before / after and optionally
// save the result.  usage: g2o_file_tool <in.g2o> [iterations] [out.g2o]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include "g2o/core/block_solver.h"
#include "g2o/core/optimization_algorithm_levenberg.h"
#include "g2o/core/sparse_optimizer.h"
#include "g2o/solvers/csparse/linear_solver_csparse.h"

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s <in.g2o> [iterations] [out.g2o]\n", argv[0]); return 1; }
  std::ifstream in(argv[1]);
  if (!in.is_open()) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
  g2o::SparseOptimizer opt;
  {   // the reference's configuration (g2o/g2o_graph.cpp:30-31,69-75)
    typedef g2o::BlockSolver<g2o::BlockSolverTraits<6, 3> > SlamBlockSolver;
    typedef g2o::LinearSolverCSparse<SlamBlockSolver::PoseMatrixType> SlamLinearCSparseSolver;
    opt.setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(new SlamBlockSolver(new SlamLinearCSparseSolver())));
  }
  if (!opt.load(in)) { std::fprintf(stderr, "malformed .g2o file: %s\n", opt.lastError().c_str()); return 1; }
  const int iters = argc > 2 ? std::atoi(argv[2]) : 20;
  opt.initializeOptimization();
  opt.computeActiveErrors();
  const double c0 = opt.chi2();
  int done = 0;
  for (int i = 0; i < iters;) {                      // CGraphG2O::optimizeGraph: chunks of 2
    const int cur = opt.optimize(iters - i < 2 ? iters - i : 2);
    if (cur <= 0) break;
    i += cur; done += cur;
  }
  opt.computeActiveErrors();
  std::printf("vertices %zu edges %zu iterations %d chi2 before %.12e after %.12e\n", opt.numVertices(), opt.numEdges(), done, c0, opt.chi2());
  if (argc > 3) { std::ofstream out(argv[3]); if (!opt.save(out)) return 1; }
  return 0;
}
