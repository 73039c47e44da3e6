// The non-inline half of shim/gtsam_lite.h: contexts of the libfgo C-ABI fed from GTSAM-style containers.
// LevenbergMarquardtOptimizer::optimize (gtsam/gtsam_graph.cpp:1784-1788, :589-590), ISAM2::update / calculateEstimate
// (:1768-1776), Marginals::marginalCovariance (:598-601), NonlinearFactorGraph::error (:173-176), writeG2o (:1941-1945).
#include "gtsam_lite.h"
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <limits>

namespace gtsam {

namespace {
const char *key_str(Key k, char buf[32]) { std::snprintf(buf, 32, "%c%llu", (char)(k >> 56), (unsigned long long)(k & 0xffffffffffffffULL)); return buf; }
}  // namespace

FgoBridge::FgoBridge() : ctx_(fgo_create(0)), calib_set_(false), gravity_set_(false) {
  if (!ctx_) std::fprintf(stderr, "gtsam shim: FATAL: %s\n", fgo_last_error(0));   // no CPU fallback
}
FgoBridge::~FgoBridge() { if (ctx_) fgo_destroy(ctx_); }

bool FgoBridge::add_factor(const FactorDesc &d) {
  char kb[32];
  for (int i = 0; i < d.nk; ++i)
    if (!kinds_.count(d.k[i])) { std::fprintf(stderr, "gtsam shim: factor references %s, which has no value\n", key_str(d.k[i], kb)); return false; }
  int rc = FGO_OK;
  switch (d.kind) {
    case FactorDesc::PRIOR_POSE: rc = fgo_add_prior_pose(ctx_, (int64_t)d.k[0], d.t, d.q, d.info21); break;
    case FactorDesc::PRIOR_VEC3: rc = fgo_add_prior_vec3(ctx_, (int64_t)d.k[0], d.v6, d.sigma); break;
    case FactorDesc::PRIOR_BIAS: rc = fgo_add_prior_bias(ctx_, (int64_t)d.k[0], d.v6, d.sigma); break;
    case FactorDesc::PRIOR_POINT: rc = fgo_add_prior_point3(ctx_, (int64_t)d.k[0], d.v6, d.sigma); break;
    case FactorDesc::BETWEEN: rc = fgo_add_edge_se3(ctx_, (int64_t)d.k[0], (int64_t)d.k[1], d.t, d.q, d.info21, FGO_TANGENT_GTSAM); break;
    case FactorDesc::PLANE: rc = fgo_add_plane_factor(ctx_, (int64_t)d.k[0], (int64_t)d.k[1], d.v6, d.cov6); break;
    case FactorDesc::IMU: {
      if (!gravity_set_) { std::memcpy(gravity_, d.gravity, sizeof(gravity_)); fgo_set_gravity(ctx_, gravity_); gravity_set_ = true; }
      else if (std::memcmp(gravity_, d.gravity, sizeof(gravity_)) != 0) { std::fprintf(stderr, "gtsam shim: one n_gravity per graph\n"); return false; }
      int64_t ids[6];
      for (int i = 0; i < 6; ++i) ids[i] = (int64_t)d.k[i];
      rc = fgo_add_imu_combined(ctx_, ids, &d.pim);
      break;
    }
    case FactorDesc::REPROJ: {
      double cb[16];
      std::memcpy(cb, d.calib, sizeof(d.calib)); std::memcpy(cb + 9, d.bps, sizeof(d.bps));
      if (!calib_set_) {
        std::memcpy(calib_, cb, sizeof(calib_));
        rc = fgo_set_calib_ds2(ctx_, cb[0], cb[1], cb[2], cb[3], cb[4], cb[5], cb[6], cb[7], cb[8], cb + 9);
        if (rc != FGO_OK) break;
        calib_set_ = true;
      } else if (std::memcmp(calib_, cb, sizeof(calib_)) != 0) { std::fprintf(stderr, "gtsam shim: one calibration / body_P_sensor per graph\n"); return false; }
      rc = fgo_add_reproj(ctx_, (int64_t)d.k[0], (int64_t)d.k[1], d.v6, d.sigma);
      break;
    }
  }
  if (rc != FGO_OK) { std::fprintf(stderr, "gtsam shim: factor rejected by libfgo: %s\n", fgo_last_error(ctx_)); return false; }
  return true;
}

bool FgoBridge::load(const NonlinearFactorGraph &g, size_t first_factor, const Values &v, bool set_existing) {
  if (!ctx_) return false;
  bool ok = true;
  for (Values::Map::const_iterator it = v.map().begin(); it != v.map().end(); ++it) {
    const Key k = it->first;
    const ValueRec &r = it->second;
    std::map<Key, int>::iterator kn = kinds_.find(k);
    int rc = FGO_OK;
    if (kn == kinds_.end()) {
      switch (r.kind) {
        case 0: rc = fgo_add_pose(ctx_, (int64_t)k, r.v, r.v + 3, 0); break;
        case 1: rc = fgo_add_plane(ctx_, (int64_t)k, r.v); break;
        case 2: rc = fgo_add_point3(ctx_, (int64_t)k, r.v); break;
        case 3: rc = fgo_add_vec3(ctx_, (int64_t)k, r.v); break;
        case 4: rc = fgo_add_bias(ctx_, (int64_t)k, r.v); break;
      }
      if (rc == FGO_OK) kinds_[k] = r.kind;
    } else if (set_existing && r.kind == 0) {
      rc = fgo_set_pose(ctx_, (int64_t)k, r.v, r.v + 3);
    }
    if (rc != FGO_OK) { std::fprintf(stderr, "gtsam shim: value rejected by libfgo: %s\n", fgo_last_error(ctx_)); ok = false; }
  }
  const std::vector<FactorDesc> &f = g.factors();
  for (size_t q = first_factor; q < f.size(); ++q) ok = add_factor(f[q]) && ok;
  return ok;
}

bool FgoBridge::read_back(Values &v) const {
  if (!ctx_) return false;
  for (std::map<Key, int>::const_iterator it = kinds_.begin(); it != kinds_.end(); ++it) {
    ValueRec r;
    r.kind = it->second;
    if (fgo_get_pose(ctx_, (int64_t)it->first, r.v) != FGO_OK) return false;
    v.map()[it->first] = r;
  }
  return true;
}

// FGO_GRAPH_DUMP=<file>: every error() call (over)writes the graph it was evaluated on -- all factor descriptors and all
// values as text -- so that a test can re-evaluate what the reference's wrapper built, factor by factor, with the oracle
// (tests/test_ref_drivers.py).  Lines:  V <key char> <index> <kind> <7 values>  |  F <kind> <nk> <keys...> <payload...>
static void dump_graph(const char *path, const std::vector<FactorDesc> &f, const Values &v, double err) {
  std::ofstream os(path);
  os << std::setprecision(17);
  os << "E " << err << "\n";
  for (Values::Map::const_iterator it = v.map().begin(); it != v.map().end(); ++it) {
    os << "V " << (char)(it->first >> 56) << " " << (unsigned long long)(it->first & 0xffffffffffffffULL) << " " << it->second.kind;
    for (int k = 0; k < 7; ++k) os << " " << it->second.v[k];
    os << "\n";
  }
  for (size_t q = 0; q < f.size(); ++q) {
    const FactorDesc &d = f[q];
    os << "F " << (int)d.kind << " " << d.nk;
    for (int i = 0; i < d.nk; ++i) os << " " << (char)(d.k[i] >> 56) << (unsigned long long)(d.k[i] & 0xffffffffffffffULL);
    switch (d.kind) {
      case FactorDesc::PRIOR_POSE: for (int k = 0; k < 3; ++k) os << " " << d.t[k]; for (int k = 0; k < 4; ++k) os << " " << d.q[k]; for (int k = 0; k < 21; ++k) os << " " << d.info21[k]; break;
      case FactorDesc::BETWEEN: for (int k = 0; k < 3; ++k) os << " " << d.t[k]; for (int k = 0; k < 4; ++k) os << " " << d.q[k]; for (int k = 0; k < 21; ++k) os << " " << d.info21[k]; break;
      case FactorDesc::PRIOR_VEC3: case FactorDesc::PRIOR_BIAS: case FactorDesc::PRIOR_POINT: for (int k = 0; k < 6; ++k) os << " " << d.v6[k]; os << " " << d.sigma; break;
      case FactorDesc::PLANE: for (int k = 0; k < 4; ++k) os << " " << d.v6[k]; for (int k = 0; k < 6; ++k) os << " " << d.cov6[k]; break;
      case FactorDesc::IMU: { for (int k = 0; k < 3; ++k) os << " " << d.gravity[k]; const double *p = reinterpret_cast<const double *>(&d.pim); for (size_t k = 0; k < sizeof(fgo_preint) / sizeof(double); ++k) os << " " << p[k]; break; }
      case FactorDesc::REPROJ: for (int k = 0; k < 2; ++k) os << " " << d.v6[k]; os << " " << d.sigma; break;
    }
    os << "\n";
  }
}

double NonlinearFactorGraph::error(const Values &v) const {
  if (f_.empty()) return 0.0;
  FgoBridge b;
  if (!b.load(*this, 0, v, false)) return std::numeric_limits<double>::quiet_NaN();
  const double err = fgo_error(b.ctx());
  if (const char *dp = std::getenv("FGO_GRAPH_DUMP")) dump_graph(dp, f_, v, err);
  return err;
}

void NonlinearFactorGraph::saveGraph(std::ostream &os, const Values &v) const {
  char ka[32], kb[32];
  os << "graph {\n";
  for (Values::Map::const_iterator it = v.map().begin(); it != v.map().end(); ++it) os << "  " << key_str(it->first, ka) << ";\n";
  for (size_t q = 0; q < f_.size(); ++q) {
    os << "  factor" << q << "[shape=point];\n";
    for (int i = 0; i < f_[q].nk; ++i) os << "  " << key_str(f_[q].k[i], kb) << "--factor" << q << ";\n";
  }
  os << "}\n";
}

const Values &LevenbergMarquardtOptimizer::optimize() {
  FgoBridge b;
  if (!b.load(g_, 0, v_, false)) { std::fprintf(stderr, "gtsam shim: LevenbergMarquardtOptimizer: graph not loadable\n"); return v_; }
  fgo_stats st;
  const int rc = fgo_optimize_gtsam(b.ctx(), p_.maxIterations, &st);
  if (rc < 0) { std::fprintf(stderr, "gtsam shim: fgo_optimize_gtsam: %s\n", fgo_last_error(b.ctx())); return v_; }
  iterations_ = rc;
  error_ = 0.5 * st.chi2_final;
  b.read_back(v_);
  return v_;
}

ISAM2Result ISAM2::update(const NonlinearFactorGraph &newFactors, const Values &newTheta) {
  ISAM2Result res;
  if (!b_) b_.reset(new FgoBridge());
  if (!b_->ok()) return res;
  const std::vector<FactorDesc> &nf = newFactors.factors();
  for (size_t q = 0; q < nf.size(); ++q) { FactorBase fb; fb.d = nf[q]; all_.add(fb); }
  b_->load(all_, loaded_, newTheta, false);            // new variables enter at their initial value; existing ones keep theta
  loaded_ = all_.size();
  const bool relin = p_.relinearizeSkip <= 1 || count_ % p_.relinearizeSkip == 0;
  ++count_;
  fgo_stats st;
  const int rc = fgo_isam2_update(b_->ctx(), relin ? p_.relinearizeThreshold : 1e300, &st);
  if (rc < 0) std::fprintf(stderr, "gtsam shim: fgo_isam2_update: %s\n", fgo_last_error(b_->ctx()));
  else { res.variablesRelinearized = (int)st.reserved[1]; res.errorAfter = 0.5 * st.chi2_final; }
  return res;
}
Values ISAM2::calculateEstimate() const {
  Values v;
  if (b_) b_->read_back(v);
  return v;
}
Matrix ISAM2::marginalCovariance(Key k) {
  Matrix M(6, 6);
  double c[36];
  if (!b_ || fgo_marginal_cov(b_->ctx(), (int64_t)k, c) != FGO_OK) { std::fprintf(stderr, "gtsam shim: fgo_marginal_cov: %s\n", b_ ? fgo_last_error(b_->ctx()) : "no graph"); return M; }
  for (int r = 0; r < 6; ++r) for (int q = 0; q < 6; ++q) M(r, q) = c[r * 6 + q];
  return M;
}

Matrix Marginals::marginalCovariance(Key k) const {
  if (!b_) { b_.reset(new FgoBridge()); b_->load(g_, 0, v_, false); }
  std::map<Key, int>::const_iterator it = b_->kinds().find(k);
  const int dim = (it != b_->kinds().end() && (it->second == 0 || it->second == 4)) ? 6 : 3;
  Matrix M(dim, dim);
  double c[36];
  if (fgo_marginal_cov(b_->ctx(), (int64_t)k, c) != FGO_OK) { std::fprintf(stderr, "gtsam shim: fgo_marginal_cov: %s\n", fgo_last_error(b_->ctx())); return M; }
  for (int r = 0; r < dim; ++r) for (int q = 0; q < dim; ++q) M(r, q) = c[r * 6 + q];
  return M;
}

void writeG2o(const NonlinearFactorGraph &g, const Values &v, const std::string &filename) {
  std::ofstream os(filename.c_str());
  if (!os.is_open()) { std::fprintf(stderr, "gtsam shim: writeG2o: cannot open %s\n", filename.c_str()); return; }
  os << std::setprecision(17);
  for (Values::Map::const_iterator it = v.map().begin(); it != v.map().end(); ++it) {
    if (it->second.kind != 0) continue;
    os << "VERTEX_SE3:QUAT " << (it->first & 0xffffffffffffffULL);
    for (int k = 0; k < 7; ++k) os << " " << it->second.v[k];
    os << "\n";
  }
  // GTSAM's writeG2o permutes the information matrix from its [omega; v] order to g2o's [t; q] order
  const std::vector<FactorDesc> &f = g.factors();
  for (size_t q = 0; q < f.size(); ++q) {
    if (f[q].kind != FactorDesc::BETWEEN) continue;
    double W[6][6];
    int k = 0;
    for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { W[r][c] = W[c][r] = f[q].info21[k++]; }
    os << "EDGE_SE3:QUAT " << (f[q].k[0] & 0xffffffffffffffULL) << " " << (f[q].k[1] & 0xffffffffffffffULL);
    for (int i = 0; i < 3; ++i) os << " " << f[q].t[i];
    for (int i = 0; i < 4; ++i) os << " " << f[q].q[i];
    for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) os << " " << W[(r + 3) % 6][(c + 3) % 6];
    os << "\n";
  }
}

}  // namespace gtsam
