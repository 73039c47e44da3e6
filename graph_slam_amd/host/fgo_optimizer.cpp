#include "fgo_optimizer.h"
#include <sstream>
#include <cstdio>
#include <iomanip>
#include <ostream>
#include <vector>

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

SparseOptimizer::SparseOptimizer() : ctx_(nullptr), verbose_(false) {
  ctx_ = fgo_create(nullptr);
  if (!ctx_) {   // no silent CPU fallback: the optimiser is unusable without the device
    err_ = fgo_last_error(nullptr);
    std::fprintf(stderr, "[fgo] FATAL: %s\n", err_.c_str());
  }
}
SparseOptimizer::~SparseOptimizer() { fgo_destroy(ctx_); }

void SparseOptimizer::clear() {
  fgo_destroy(ctx_);
  ctx_ = fgo_create(nullptr);
  vertex_ids_.clear(); fixed_.clear(); edges_.clear();
}

bool SparseOptimizer::addVertexSE3(int id, const Eigen::Isometry3d &est, bool fixed) {
  if (!ctx_) return false;
  double p[7];
  iso_to_pose7(est, p);
  if (fgo_add_pose(ctx_, id, p, p + 3, fixed ? 1 : 0) != FGO_OK) { err_ = fgo_last_error(ctx_); return false; }
  vertex_ids_.push_back(id);
  if (fixed) fixed_.insert(id);
  return true;
}
bool SparseOptimizer::hasVertex(int id) const { return ctx_ && fgo_has_pose(ctx_, id) == 1; }

Eigen::Isometry3d SparseOptimizer::estimate(int id) const {
  double p[7] = {0, 0, 0, 0, 0, 0, 1};
  if (ctx_ && fgo_get_pose(ctx_, id, p) != FGO_OK) err_ = fgo_last_error(ctx_);
  return pose7_to_iso(p);
}
bool SparseOptimizer::setEstimate(int id, const Eigen::Isometry3d &est) {
  if (!ctx_) return false;
  double p[7];
  iso_to_pose7(est, p);
  if (fgo_set_pose(ctx_, id, p, p + 3) != FGO_OK) { err_ = fgo_last_error(ctx_); return false; }
  return true;
}

bool SparseOptimizer::addEdgeSE3(int id1, int id2, const Eigen::Isometry3d &meas, const Eigen::Matrix<double, 6, 6> &info) {
  if (!ctx_) return false;
  EdgeRec e;
  e.a = id1; e.b = id2;
  iso_to_pose7(meas, e.z);
  int k = 0;
  for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) e.info[k++] = info(r, c);
  if (fgo_add_edge_se3(ctx_, id1, id2, e.z, e.z + 3, e.info, FGO_TANGENT_G2O) != FGO_OK) { err_ = fgo_last_error(ctx_); return false; }
  edges_.push_back(e);
  return true;
}

int SparseOptimizer::optimize(int iterations) {
  if (!ctx_) return 0;
  fgo_stats st;
  const int rc = fgo_optimize(ctx_, iterations, &st);
  if (rc == FGO_ESTATE) return -1;          // g2o: "0 vertices to optimize, maybe forgot to call initializeOptimization()"
  if (rc < 0) { err_ = fgo_last_error(ctx_); std::fprintf(stderr, "[fgo] optimize failed: %s\n", err_.c_str()); return 0; }
  if (verbose_) std::fprintf(stderr, "[fgo] %d iterations, chi2 %.6e -> %.6e, lambda %.3e\n", rc, st.chi2_initial, st.chi2_final, st.lambda_final);
  return rc;
}

double SparseOptimizer::chi2() const { return ctx_ ? fgo_chi2(ctx_) : 0.0; }

bool SparseOptimizer::save(std::ostream &os) const {
  if (!ctx_) return false;
  os << std::setprecision(17);
  for (int id : vertex_ids_) {
    double p[7];
    if (fgo_get_pose(ctx_, id, p) != FGO_OK) return false;
    os << "VERTEX_SE3:QUAT " << id;
    for (int k = 0; k < 7; ++k) os << " " << p[k];
    os << "\n";
    if (fixed_.count(id)) os << "FIX " << id << "\n";
  }
  for (const EdgeRec &e : edges_) {
    os << "EDGE_SE3:QUAT " << e.a << " " << e.b;
    for (int k = 0; k < 7; ++k) os << " " << e.z[k];
    for (int k = 0; k < 21; ++k) os << " " << e.info[k];
    os << "\n";
  }
  return os.good();
}

bool SparseOptimizer::load(std::istream &is) {
  if (!ctx_) return false;
  std::string line, tag;
  std::vector<int> to_fix;
  while (std::getline(is, line)) {
    std::istringstream ls(line);
    if (!(ls >> tag)) continue;
    if (tag == "VERTEX_SE3:QUAT") {
      int id; double p[7];
      if (!(ls >> id)) return false;
      for (int k = 0; k < 7; ++k) if (!(ls >> p[k])) return false;
      Eigen::Isometry3d T(Eigen::Quaterniond(p[6], p[3], p[4], p[5]).toRotationMatrix(), Eigen::Vector3d());
      T.translation()(0) = p[0]; T.translation()(1) = p[1]; T.translation()(2) = p[2];
      if (!addVertexSE3(id, T, false)) return false;
    } else if (tag == "FIX") {
      int id;
      while (ls >> id) to_fix.push_back(id);
    } else if (tag == "EDGE_SE3:QUAT") {
      EdgeRec e;
      if (!(ls >> e.a >> e.b)) return false;
      for (int k = 0; k < 7; ++k) if (!(ls >> e.z[k])) return false;
      for (int k = 0; k < 21; ++k) if (!(ls >> e.info[k])) return false;
      if (fgo_add_edge_se3(ctx_, e.a, e.b, e.z, e.z + 3, e.info, FGO_TANGENT_G2O) != FGO_OK) { err_ = fgo_last_error(ctx_); return false; }
      edges_.push_back(e);
    }
  }
  // FIX records may come before or after the vertex they name: re-add those vertices as fixed
  for (int id : to_fix) {
    if (!hasVertex(id) || fixed_.count(id)) continue;
    double p[7];
    if (fgo_get_pose(ctx_, id, p) != FGO_OK) return false;
    if (fgo_set_fixed(ctx_, id, 1) != FGO_OK) { err_ = fgo_last_error(ctx_); return false; }
    fixed_.insert(id);
  }
  return true;
}

}  // namespace g2o
