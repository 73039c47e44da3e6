// Implementation of the g2o API slice (shim/g2o/fgo_g2o.h) on the fgo C-ABI: every call the reference's wrapper makes
// on g2o::SparseOptimizer / VertexSE3 / EdgeSE3 (g2o/g2o_graph.cpp:65-134,241-258,279-349) is forwarded to libfgo.so.
#include "g2o/fgo_g2o.h"
#include <cstdio>
#include <iomanip>
#include <istream>
#include <ostream>
#include <sstream>
#include "../../include/fgo.h"

namespace {
void iso_to_pose7(const Eigen::Isometry3d &T, double p[7]) {
  Eigen::Quaterniond q(T.rotation());
  p[0] = T.translation()(0); p[1] = T.translation()(1); p[2] = T.translation()(2);
  p[3] = q.x(); p[4] = q.y(); p[5] = q.z(); p[6] = q.w();
}
Eigen::Isometry3d pose7_to_iso(const double p[7]) {
  Eigen::Quaterniond q(p[6], p[3], p[4], p[5]);
  Eigen::Vector3d t; t(0) = p[0]; t(1) = p[1]; t(2) = p[2];
  return Eigen::Isometry3d(q.toRotationMatrix(), t);
}
}  // namespace

namespace g2o {

// ---- vertices: detached = local value; attached = value lives in the context (device-resident between optimize() calls)
void OptimizableGraph::Vertex::setFixed(bool f) {
  _fixed = f;
  if (_graph && _graph->_ctx && fgo_set_fixed(_graph->_ctx, _id, f ? 1 : 0) != FGO_OK) _graph->_err = fgo_last_error(_graph->_ctx);
}
Eigen::Isometry3d VertexSE3::estimate() const {
  if (!_graph || !_graph->_ctx) return _estimate;
  double p[7] = {0, 0, 0, 0, 0, 0, 1};
  if (fgo_get_pose(_graph->_ctx, _id, p) != FGO_OK) { _graph->_err = fgo_last_error(_graph->_ctx); return _estimate; }
  return pose7_to_iso(p);
}
void VertexSE3::setEstimate(const Eigen::Isometry3d &e) {
  _estimate = e;
  if (!_graph || !_graph->_ctx) return;
  double p[7];
  iso_to_pose7(e, p);
  if (fgo_set_pose(_graph->_ctx, _id, p, p + 3) != FGO_OK) _graph->_err = fgo_last_error(_graph->_ctx);
}

// ---- optimizer
SparseOptimizer::SparseOptimizer() : _ctx(0), _verbose(false), _algorithm(0) {
  _ctx = fgo_create(0);
  if (!_ctx) {   // no silent CPU fallback: the optimiser is unusable without the device
    _err = fgo_last_error(0);
    std::fprintf(stderr, "[fgo] FATAL: %s\n", _err.c_str());
  }
}
SparseOptimizer::~SparseOptimizer() {
  for (VertexIDMap::iterator it = _vertices.begin(); it != _vertices.end(); ++it) delete it->second;
  for (EdgeSet::iterator it = _edgeSet.begin(); it != _edgeSet.end(); ++it) delete *it;
  delete _algorithm;
  fgo_destroy(_ctx);
}
void SparseOptimizer::clear() {
  for (VertexIDMap::iterator it = _vertices.begin(); it != _vertices.end(); ++it) delete it->second;
  for (EdgeSet::iterator it = _edgeSet.begin(); it != _edgeSet.end(); ++it) delete *it;
  _vertices.clear(); _edgeSet.clear(); _edgeOrder.clear();
  fgo_destroy(_ctx);
  _ctx = fgo_create(0);
}
void SparseOptimizer::setAlgorithm(OptimizationAlgorithm *a) {
  if (a != _algorithm) delete _algorithm;
  _algorithm = a;
}

bool SparseOptimizer::addVertex(HyperGraph::Vertex *v) {
  VertexSE3 *se3 = dynamic_cast<VertexSE3 *>(v);
  if (!_ctx || !se3) { if (_ctx) _err = "only VertexSE3 is supported"; return false; }
  if (_vertices.count(se3->id())) { _err = "vertex id already in the graph"; return false; }
  double p[7];
  iso_to_pose7(se3->_estimate, p);
  if (fgo_add_pose(_ctx, se3->id(), p, p + 3, se3->fixed() ? 1 : 0) != FGO_OK) { _err = fgo_last_error(_ctx); return false; }
  se3->_graph = this;
  _vertices[se3->id()] = se3;
  return true;
}
HyperGraph::Vertex *SparseOptimizer::vertex(int id) {
  VertexIDMap::iterator it = _vertices.find(id);
  return it == _vertices.end() ? 0 : it->second;
}

bool SparseOptimizer::addEdge(HyperGraph::Edge *e) {
  EdgeSE3 *se3 = dynamic_cast<EdgeSE3 *>(e);
  if (!_ctx || !se3) { if (_ctx) _err = "only EdgeSE3 is supported"; return false; }
  HyperGraph::Vertex *a = se3->vertices()[0], *b = se3->vertices()[1];
  if (!a || !b || vertex(a->id()) != a || vertex(b->id()) != b) { _err = "edge references a vertex that is not in the graph"; return false; }
  double z[7], info[21];
  iso_to_pose7(se3->measurement(), z);
  int k = 0;
  for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) info[k++] = se3->information()(r, c);
  if (fgo_add_edge_se3(_ctx, a->id(), b->id(), z, z + 3, info, FGO_TANGENT_G2O) != FGO_OK) { _err = fgo_last_error(_ctx); return false; }
  _edgeSet.insert(se3);
  _edgeOrder.push_back(se3);
  return true;
}

bool SparseOptimizer::initializeOptimization(int) { return _ctx != 0 && !_vertices.empty(); }

int SparseOptimizer::optimize(int iterations, bool) {
  if (!_ctx) return 0;
  if (!_algorithm) { _err = "no optimization algorithm set (setAlgorithm)"; std::fprintf(stderr, "[fgo] %s\n", _err.c_str()); return -1; }
  // the only configuration the reference builds: LM over BlockSolver<6,3> over a sparse Cholesky (g2o_graph.cpp:69-75)
  OptimizationAlgorithmLevenberg *lm = dynamic_cast<OptimizationAlgorithmLevenberg *>(_algorithm);
  if (!lm || !lm->solver() || lm->solver()->poseDim() != 6) { _err = "unsupported algorithm / block size"; return -1; }
  fgo_stats st;
  const int rc = fgo_optimize(_ctx, iterations, &st);
  if (rc == FGO_ESTATE) return -1;          // g2o: "0 vertices to optimize, maybe forgot to call initializeOptimization()"
  if (rc < 0) { _err = fgo_last_error(_ctx); std::fprintf(stderr, "[fgo] optimize failed: %s\n", _err.c_str()); return 0; }
  if (_verbose) std::fprintf(stderr, "[fgo] %d iterations, chi2 %.6e -> %.6e, lambda %.3e\n", rc, st.chi2_initial, st.chi2_final, st.lambda_final);
  return rc;
}

double SparseOptimizer::chi2() const { return _ctx ? fgo_chi2(_ctx) : 0.0; }

bool SparseOptimizer::save(std::ostream &os) const {
  if (!_ctx) return false;
  os << std::setprecision(17);
  for (VertexIDMap::const_iterator it = _vertices.begin(); it != _vertices.end(); ++it) {
    double p[7];
    if (fgo_get_pose(_ctx, it->first, p) != FGO_OK) return false;
    os << "VERTEX_SE3:QUAT " << it->first;
    for (int k = 0; k < 7; ++k) os << " " << p[k];
    os << "\n";
    if (static_cast<OptimizableGraph::Vertex *>(it->second)->fixed()) os << "FIX " << it->first << "\n";
  }
  for (size_t q = 0; q < _edgeOrder.size(); ++q) {
    const EdgeSE3 *e = _edgeOrder[q];
    double z[7];
    iso_to_pose7(e->measurement(), z);
    os << "EDGE_SE3:QUAT " << e->vertices()[0]->id() << " " << e->vertices()[1]->id();
    for (int k = 0; k < 7; ++k) os << " " << z[k];
    for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) os << " " << e->information()(r, c);
    os << "\n";
  }
  return os.good();
}

bool SparseOptimizer::load(std::istream &is) {
  if (!_ctx) return false;
  std::string line, tag;
  std::vector<int> to_fix;
  while (std::getline(is, line)) {
    std::istringstream ls(line);
    if (!(ls >> tag)) continue;
    if (tag == "VERTEX_SE3:QUAT") {
      int id; double p[7];
      if (!(ls >> id)) return false;
      for (int k = 0; k < 7; ++k) if (!(ls >> p[k])) return false;
      VertexSE3 *v = new VertexSE3;
      v->setId(id);
      v->setEstimate(pose7_to_iso(p));
      if (!addVertex(v)) { delete v; return false; }
    } else if (tag == "FIX") {
      int id;
      while (ls >> id) to_fix.push_back(id);
    } else if (tag == "EDGE_SE3:QUAT") {
      int a, b; double z[7];
      Eigen::Matrix<double, 6, 6> W;
      if (!(ls >> a >> b)) return false;
      for (int k = 0; k < 7; ++k) if (!(ls >> z[k])) return false;
      for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { double w; if (!(ls >> w)) return false; W(r, c) = w; W(c, r) = w; }
      EdgeSE3 *e = new EdgeSE3;
      e->vertices()[0] = vertex(a); e->vertices()[1] = vertex(b);
      e->setMeasurement(pose7_to_iso(z));
      e->setInformation(W);
      if (!addEdge(e)) { delete e; return false; }
    }
  }
  for (size_t q = 0; q < to_fix.size(); ++q) {            // FIX records may precede or follow the vertex they name
    OptimizableGraph::Vertex *v = static_cast<OptimizableGraph::Vertex *>(vertex(to_fix[q]));
    if (v && !v->fixed()) v->setFixed(true);
  }
  return true;
}

}  // namespace g2o
