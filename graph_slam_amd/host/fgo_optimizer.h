// g2o::SparseOptimizer as seen by CGraphG2O, implemented on the fgo C-ABI (include/fgo.h).
// The reference's header only forward-declares g2o::SparseOptimizer and keeps a public raw pointer
// `mp_optimizer` (g2o/g2o_graph.h:14-16,43); this class supplies the operations g2o_graph.cpp performs on it
// (g2o/g2o_graph.cpp:69-75,88-91,98-132,246-249,256-257,282,297,327) so the wrapper logic reads the same.
#pragma once
#include <iosfwd>
#include <set>
#include <string>
#include <vector>
#include <Eigen/Core>
#include <Eigen/Geometry>
#include "../../include/fgo.h"

namespace g2o {

class SparseOptimizer {
 public:
  SparseOptimizer();
  ~SparseOptimizer();
  SparseOptimizer(const SparseOptimizer &) = delete;
  SparseOptimizer &operator=(const SparseOptimizer &) = delete;

  void setVerbose(bool v) { verbose_ = v; }
  // VertexSE3: addVertex(new VertexSE3{id, estimate, fixed})
  bool addVertexSE3(int id, const Eigen::Isometry3d &estimate, bool fixed);
  bool hasVertex(int id) const;
  Eigen::Isometry3d estimate(int id) const;
  bool setEstimate(int id, const Eigen::Isometry3d &estimate);
  // EdgeSE3: vertices (id1 -> id2), setMeasurement, setInformation, addEdge
  bool addEdgeSE3(int id1, int id2, const Eigen::Isometry3d &measurement, const Eigen::Matrix<double, 6, 6> &information);
  bool initializeOptimization() { return true; }
  int optimize(int iterations);           // iterations done; 0 on failure, -1 if nothing to optimise
  void computeActiveErrors() {}
  double chi2() const;                    // sum e' Omega e (no 1/2)
  bool save(std::ostream &os) const;      // .g2o text: VERTEX_SE3:QUAT / FIX / EDGE_SE3:QUAT
  // g2o's SparseOptimizer::load for the same tags (what `optimizer->save` of g2o/g2o_graph.cpp:282 wrote, or any SE3
  // pose-graph dataset in .g2o text); unknown tags are skipped.  Returns false on a malformed record.
  bool load(std::istream &is);
  size_t numVertices() const { return vertex_ids_.size(); }
  size_t numEdges() const { return edges_.size(); }
  void clear();
  fgo_ctx *handle() { return ctx_; }
  const std::string &lastError() const { return err_; }

 private:
  fgo_ctx *ctx_;
  bool verbose_;
  mutable std::string err_;
  struct EdgeRec { int a, b; double z[7]; double info[21]; };
  std::vector<int> vertex_ids_;
  std::set<int> fixed_;
  std::vector<EdgeRec> edges_;
};

}  // namespace g2o
