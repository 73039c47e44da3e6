// The slice of the g2o API that the reference's pose-graph wrapper binds (g2o/g2o_graph.cpp:5-12,30-35,65-134,241-258,
// 279-349), implemented as thin handles onto one fgo context (include/fgo.h, libfgo.so).  With graph_slam_amd/host/shim
// on the include path the reference's OWN g2o_graph.cpp compiles unchanged and every g2o call it makes lands in the HIP
// back-end:
//   new VertexSE3 / setId / setFixed / setEstimate / addVertex   -> fgo_add_pose (fgo_set_pose / fgo_set_fixed once attached)
//   new EdgeSE3 / vertices()[k] / setMeasurement / setInformation / addEdge -> fgo_add_edge_se3 (tangent order [t; q])
//   BlockSolver<6,3> + LinearSolverCSparse + OptimizationAlgorithmLevenberg + setAlgorithm  -> selects fgo_optimize's
//        restatement of exactly that configuration (LM over a 6x6-block sparse Cholesky)
//   initializeOptimization / optimize(n) / computeActiveErrors / chi2 / save / load / clear
// Ownership follows g2o: the optimizer owns the vertices and edges handed to addVertex / addEdge and the algorithm
// handed to setAlgorithm; the algorithm owns its solver, the block solver its linear solver (g2o's pre-2017 raw
// pointer API, which is what g2o_graph.cpp:72-75 uses).  There is no CPU fallback: without a HIP device the optimizer
// reports the error and every operation fails.
#pragma once
#include <iosfwd>
#include <map>
#include <set>
#include <string>
#include <vector>
#include <Eigen/Core>
#include <Eigen/Geometry>

struct fgo_ctx;

namespace g2o {

class SparseOptimizer;

class HyperGraph {
 public:
  class Vertex {
   public:
    explicit Vertex(int id = -1) : _id(id) {}
    virtual ~Vertex() {}
    int id() const { return _id; }
    virtual void setId(int id) { _id = id; }
   protected:
    int _id;
  };
  class Edge {
   public:
    explicit Edge(int n = 2) : _vertices((size_t)n, (Vertex *)0) {}
    virtual ~Edge() {}
    std::vector<Vertex *> &vertices() { return _vertices; }
    const std::vector<Vertex *> &vertices() const { return _vertices; }
    Vertex *vertex(size_t i) { return _vertices[i]; }
    void setVertex(size_t i, Vertex *v) { _vertices[i] = v; }
   protected:
    std::vector<Vertex *> _vertices;
  };
  typedef std::set<Edge *> EdgeSet;
  typedef std::map<int, Vertex *> VertexIDMap;
  virtual ~HyperGraph() {}
};

class OptimizableGraph : public HyperGraph {
 public:
  class Vertex : public HyperGraph::Vertex {
   public:
    Vertex() : _fixed(false), _graph(0) {}
    bool fixed() const { return _fixed; }
    void setFixed(bool f);
   protected:
    friend class g2o::SparseOptimizer;
    bool _fixed;
    SparseOptimizer *_graph;       // set by addVertex: from then on the estimate lives in the fgo context
  };
  class Edge : public HyperGraph::Edge {
   public:
    Edge() : HyperGraph::Edge(2) {}
  };
};

// VertexSE3 (g2o/types/slam3d): Isometry3d estimate, increment [dt; dq], oplus X <- X * (R(dq), dt)
class VertexSE3 : public OptimizableGraph::Vertex {
 public:
  VertexSE3() {}
  Eigen::Isometry3d estimate() const;
  void setEstimate(const Eigen::Isometry3d &e);
 private:
  friend class SparseOptimizer;
  Eigen::Isometry3d _estimate;     // authoritative only while detached
};

// EdgeSE3: measurement Isometry3d, information 6x6 in [t; q] order, error [t(D); vec q(D)], D = Z^-1 Xi^-1 Xj
class EdgeSE3 : public OptimizableGraph::Edge {
 public:
  EdgeSE3() { _information.setIdentity(); }
  void setMeasurement(const Eigen::Isometry3d &m) { _measurement = m; }
  const Eigen::Isometry3d &measurement() const { return _measurement; }
  void setInformation(const Eigen::Matrix<double, 6, 6> &w) { _information = w; }
  const Eigen::Matrix<double, 6, 6> &information() const { return _information; }
 private:
  Eigen::Isometry3d _measurement;
  Eigen::Matrix<double, 6, 6> _information;
};

// ---- solver / algorithm objects: configuration carriers (the arithmetic they select lives in libfgo)
template <typename MatrixType>
class LinearSolver {
 public:
  virtual ~LinearSolver() {}
};
template <typename MatrixType>
class LinearSolverCSparse : public LinearSolver<MatrixType> {
 public:
  LinearSolverCSparse() : _blockOrdering(true) {}
  void setBlockOrdering(bool b) { _blockOrdering = b; }
  bool blockOrdering() const { return _blockOrdering; }
 private:
  bool _blockOrdering;
};
template <int P, int L>
struct BlockSolverTraits {
  static const int PoseDim = P, LandmarkDim = L;
  struct PoseMatrixType { static const int Rows = P, Cols = P; };   // Eigen::Matrix<double,P,P> in g2o; only named here
};
class Solver {
 public:
  virtual ~Solver() {}
  virtual int poseDim() const = 0;
};
template <typename Traits>
class BlockSolver : public Solver {
 public:
  typedef typename Traits::PoseMatrixType PoseMatrixType;
  typedef LinearSolver<PoseMatrixType> LinearSolverType;
  explicit BlockSolver(LinearSolverType *ls) : _linearSolver(ls) {}
  ~BlockSolver() { delete _linearSolver; }
  int poseDim() const { return Traits::PoseDim; }
 private:
  LinearSolverType *_linearSolver;
};
class OptimizationAlgorithm {
 public:
  virtual ~OptimizationAlgorithm() {}
  virtual const char *name() const = 0;
};
class OptimizationAlgorithmLevenberg : public OptimizationAlgorithm {
 public:
  explicit OptimizationAlgorithmLevenberg(Solver *s) : _solver(s) {}
  ~OptimizationAlgorithmLevenberg() { delete _solver; }
  const char *name() const { return "lm"; }
  Solver *solver() { return _solver; }
 private:
  Solver *_solver;
};

class SparseOptimizer : public OptimizableGraph {
 public:
  SparseOptimizer();
  ~SparseOptimizer();
  void setVerbose(bool v) { _verbose = v; }
  void setAlgorithm(OptimizationAlgorithm *a);
  bool addVertex(HyperGraph::Vertex *v);
  bool addEdge(HyperGraph::Edge *e);
  HyperGraph::Vertex *vertex(int id);
  const VertexIDMap &vertices() const { return _vertices; }
  const EdgeSet &edges() const { return _edgeSet; }
  bool initializeOptimization(int level = 0);
  int optimize(int iterations, bool online = false);   // iterations done; 0 on failure, -1 if nothing to optimise
  void computeActiveErrors() {}                         // chi2() evaluates on the device at the current estimate
  double chi2() const;                                  // sum e' Omega e (no 1/2)
  bool save(std::ostream &os) const;                    // VERTEX_SE3:QUAT / FIX / EDGE_SE3:QUAT
  bool load(std::istream &is);
  void clear();                                         // deletes vertices and edges, fresh context
  // not part of g2o: access to the C-ABI handle and the last error text
  fgo_ctx *handle() { return _ctx; }
  const std::string &lastError() const { return _err; }
  size_t numVertices() const { return _vertices.size(); }
  size_t numEdges() const { return _edgeOrder.size(); }

 private:
  friend class OptimizableGraph::Vertex;
  friend class VertexSE3;
  SparseOptimizer(const SparseOptimizer &);
  SparseOptimizer &operator=(const SparseOptimizer &);
  fgo_ctx *_ctx;
  bool _verbose;
  OptimizationAlgorithm *_algorithm;
  VertexIDMap _vertices;
  EdgeSet _edgeSet;
  std::vector<EdgeSE3 *> _edgeOrder;                    // insertion order (save() writes edges as they were added)
  mutable std::string _err;
};

}  // namespace g2o
