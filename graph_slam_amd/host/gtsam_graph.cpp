// CGraphGT on top of the libfgo C-ABI: see gtsam_graph.h.  Each method cites the reference code it mirrors
// (gtsam/gtsam_graph.cpp); the arithmetic that GTSAM would do on the CPU (linearisation, LM, marginals) happens in
// libfgo on the MI355X.
#include "gtsam_graph.h"

#include <cmath>
#include <cstdio>
#include <iomanip>

#include <gtsam/geometry/Pose3.h>
#include <gtsam/navigation/CombinedImuFactor.h>
#include <gtsam/nonlinear/LevenbergMarquardtOptimizer.h>
#include <gtsam/nonlinear/Marginals.h>
#include <gtsam/slam/BetweenFactor.h>
#include <gtsam/slam/OrientedPlane3Factor.h>
#include <gtsam/slam/PriorFactor.h>

#include "camera_node.h"
#include "gt_parameter.h"
#include "matching_result.h"

using namespace gtsam;
using namespace std;
using symbol_shorthand::B;
using symbol_shorthand::L;
using symbol_shorthand::V;
using symbol_shorthand::X;

namespace {
// Pose3::ChartAtOrigin::Local of a 4x4 float transform (gtsam_graph.cpp:56-62)
Vector6 cov_Helper(const Eigen::Matrix4f &m) { return Pose3::ChartAtOrigin::Local(Pose3(m.cast<double>())); }
}  // namespace

CGraphGT::CGraphGT() : m_sequence_id(0), m_vertex_id(0), mp_rec_file(nullptr) {
  // one device context behind the graph and the values (gtsam_graph.cpp:75-99)
  m_backend = std::make_shared<Backend>();
  if (!m_backend->ctx) fprintf(stderr, "%s: %s\n", __FILE__, fgo_last_error(nullptr));
  mp_fac_graph = new NonlinearFactorGraph(m_backend);
  mp_new_fac = new NonlinearFactorGraph;
  mp_node_values = new Values(m_backend);
  mp_new_node = new Values;
  mp_w2o = new Pose3;
  mp_u2c = new Pose3;
  mp_prev_bias = new imuBias::ConstantBias;
  mp_prev_state = new NavState;
  mb_record_vro_results = CGTParams::Instance()->m_record_vro_results;
  m_plane_landmark_id = 0;
  initISAM2Params();
}

// gtsam_graph.cpp:93-99
void CGraphGT::initISAM2Params() {
  mp_isam2_param = new ISAM2Params;
  mp_isam2_param->relinearizeThreshold = 0.1;
  mp_isam2_param->relinearizeSkip = 1;
  mp_isam2 = new ISAM2(*mp_isam2_param);
  mp_isam2->attach(*mp_fac_graph, *mp_node_values);      // the incremental state lives in the full graph's device context
}

CGraphGT::~CGraphGT() {
  delete mp_isam2;
  delete mp_isam2_param;
  delete mp_prev_bias;
  delete mp_prev_state;
  delete mp_fac_graph;
  delete mp_new_fac;
  delete mp_new_node;
  delete mp_node_values;
  delete mp_w2o;
  delete mp_u2c;
  if (mp_rec_file) { mp_rec_file->close(); delete mp_rec_file; }
  for (map<int, CCameraNode *>::iterator it = m_graph_map.begin(); it != m_graph_map.end(); ++it) delete it->second;   // :152-156
}

ofstream *CGraphGT::getRecFile() {
  if (!mp_rec_file) mp_rec_file = new ofstream(CGTParams::Instance()->m_vro_result.c_str());
  return mp_rec_file;
}

double CGraphGT::error() { return mp_fac_graph->error(*mp_node_values); }

// gtsam_graph.cpp:178-209: world (Z up, X forward) from the camera frame, pitched by p
void CGraphGT::setWorld2Original(double p) {
  const Rot3 R_g2b = Rot3::RzRyRx(-M_PI / 2., 0, -M_PI / 2.);
  const Rot3 R_b2o = Rot3::RzRyRx(p, 0, 0);
  (*mp_w2o) = Pose3::Create(R_g2b * R_b2o, Point3());
}
void CGraphGT::setWorld2Original(double r, double p, double y) {
  (*mp_w2o) = Pose3::Create(Rot3::RzRyRx(r, p, y), Point3());
}
// gtsam_graph.cpp:211-254
void CGraphGT::setCamera2IMUTranslation(double px, double py, double pz) {
  Point3 t; t(0) = px; t(1) = py; t(2) = pz;
  (*mp_u2c) = Pose3::Create(Rot3(), t);
}
void CGraphGT::setCamera2IMU(double p) {
  const Rot3 R_g2b = Rot3::RzRyRx(M_PI / 2., 0., M_PI / 2.);
  const Rot3 R_b2o = Rot3::RzRyRx(p, 0, 0);
  (*mp_u2c) = Pose3::Create(R_g2b * R_b2o, Point3());
}

// gtsam_graph.cpp:320-368
void CGraphGT::firstNode(CCameraNode *n, bool online) {
  n->m_id = (int)m_graph_map.size();
  m_sequence_id = 0;
  if (online) n->m_seq_id = ++m_sequence_id;

  const Pose3 origin_priorMean;
  mp_node_values->insert(X(n->m_id), origin_priorMean);
  mp_new_node->insert(X(n->m_id), origin_priorMean);
  Vector6 s;
  for (int k = 0; k < 6; ++k) s(k) = 1e-7;
  SharedNoiseModel priorNoise = noiseModel::Diagonal::Sigmas(s);
  mp_fac_graph->add(PriorFactor<Pose3>(X(n->m_id), origin_priorMean, priorNoise));
  mp_new_fac->add(PriorFactor<Pose3>(X(n->m_id), origin_priorMean, priorNoise));
  m_graph_map[n->m_id] = n;

  const Vector3 priorVelocity;
  mp_node_values->insert(V(n->m_id), priorVelocity);
  mp_new_node->insert(V(n->m_id), priorVelocity);
  const imuBias::ConstantBias priorBias;
  mp_node_values->insert(B(n->m_id), priorBias);
  mp_new_node->insert(B(n->m_id), priorBias);
  SharedNoiseModel velocity_noise_model = noiseModel::Isotropic::Sigma(3, 1e-3);
  SharedNoiseModel bias_noise_model = noiseModel::Isotropic::Sigma(6, 1e-3);
  mp_fac_graph->add(PriorFactor<Vector3>(V(n->m_id), priorVelocity, velocity_noise_model));
  mp_fac_graph->add(PriorFactor<imuBias::ConstantBias>(B(n->m_id), priorBias, bias_noise_model));
  mp_new_fac->add(PriorFactor<Vector3>(V(n->m_id), priorVelocity, velocity_noise_model));
  mp_new_fac->add(PriorFactor<imuBias::ConstantBias>(B(n->m_id), priorBias, bias_noise_model));
}

// gtsam_graph.cpp:613-628
bool CGraphGT::addToGTSAM(gtsam::NavState &new_state, int vid, bool add_pose) {
  if (add_pose) {
    mp_node_values->insert(X(vid), new_state.pose());
    mp_new_node->insert(X(vid), new_state.pose());
  }
  mp_node_values->insert(V(vid), new_state.v());
  mp_node_values->insert(B(vid), *mp_prev_bias);
  mp_new_node->insert(V(vid), new_state.v());
  mp_new_node->insert(B(vid), *mp_prev_bias);
  return true;
}

// gtsam_graph.cpp:630-695: the VRO transform is conjugated into the IMU frame, T_u2c T T_u2c^-1, its information
// with the adjoint of T_u2c
bool CGraphGT::addToGTSAM(MatchingResult &mr, bool set_estimate) {
  const bool pre_exist = mp_node_values->exists(X(mr.edge.id1));
  const bool cur_exist = mp_node_values->exists(X(mr.edge.id2));
  Pose3 inc_pose(mr.edge.transform.matrix());
  inc_pose = (*mp_u2c) * inc_pose * (*mp_u2c).inverse();

  if (!pre_exist && !cur_exist) {
    fprintf(stderr, "%s two nodes %i and %i both not exist\n", __FILE__, mr.edge.id1, mr.edge.id2);
    return false;
  } else if (!pre_exist) {
    const Pose3 cur_pose = mp_node_values->at<Pose3>(X(mr.edge.id2));
    const Pose3 pre_pose = cur_pose * inc_pose.inverse();
    mp_node_values->insert(X(mr.edge.id1), pre_pose);
    mp_new_node->insert(X(mr.edge.id1), pre_pose);
  } else if (!cur_exist) {
    const Pose3 pre_pose = mp_node_values->at<Pose3>(X(mr.edge.id1));
    const Pose3 cur_pose = pre_pose * inc_pose;
    mp_node_values->insert(X(mr.edge.id2), cur_pose);
    mp_new_node->insert(X(mr.edge.id2), cur_pose);
  } else if (set_estimate) {
    const Pose3 pre_pose = mp_node_values->at<Pose3>(X(mr.edge.id1));
    const Pose3 cur_pose = pre_pose * inc_pose;
    mp_node_values->update(X(mr.edge.id2), cur_pose);
    mp_new_node->update(X(mr.edge.id2), cur_pose);
  }

  const Matrix6 Adj_Tuc = (*mp_u2c).AdjointMap();
  const Matrix6 tmp = Adj_Tuc * mr.edge.informationMatrix * Adj_Tuc.transpose();
  SharedNoiseModel visual_odometry_noise = noiseModel::Gaussian::Information(tmp);
  mp_fac_graph->add(BetweenFactor<Pose3>(X(mr.edge.id1), X(mr.edge.id2), inc_pose, visual_odometry_noise));
  mp_new_fac->add(BetweenFactor<Pose3>(X(mr.edge.id1), X(mr.edge.id2), inc_pose, visual_odometry_noise));

  StoredEdge e;
  e.id1 = mr.edge.id1; e.id2 = mr.edge.id2;
  pose_payload(inc_pose, e.t, e.q);
  info_ut21(tmp, e.info);
  m_edges.push_back(e);
  return true;
}

// gtsam_graph.cpp:697-722
void CGraphGT::fakeOdoNode(CCameraNode *new_node) {
  if (new_node->m_id != (int)m_graph_map.size()) {
    cerr << __FILE__ << " " << __LINE__ << " Here this should not happen!" << endl;
    new_node->m_id = (int)m_graph_map.size();
    new_node->m_seq_id = ++m_sequence_id;
  }
  CCameraNode *pre_node = m_graph_map[new_node->m_id - 1];
  MatchingResult mr;
  mr.edge.id1 = pre_node->m_id;
  mr.edge.id2 = new_node->m_id;
  mr.edge.transform.setIdentity();
  mr.edge.informationMatrix = Eigen::Matrix<double, 6, 6>::Identity() * 1e4;
  addToGTSAM(mr, false);
  m_graph_map[new_node->m_id] = new_node;
  if (mb_record_vro_results) recordVROResult(mr);
}

// gtsam_graph.cpp:1510-1558: "id_to id_from xi[6] info_ut[21]" per line, xi = chart-at-origin coordinates [omega; v]
void CGraphGT::readVRORecord(std::string fname) { readVRORecord(fname, mv_vro_res); }
void CGraphGT::readVRORecord(std::string fname, std::vector<MatchingResult *> &mv) {
  ifstream inf(fname.c_str());
  if (!inf.is_open()) {
    cerr << " failed to open file " << fname << endl;
    return;
  }
  while (!inf.eof()) {
    int id_to, id_from;
    Vector6 r;
    MatchingResult *pm = new MatchingResult;
    inf >> id_to >> id_from;
    for (int i = 0; i < 6; i++) inf >> r(i);
    const Pose3 p = Pose3::ChartAtOrigin::Retract(r);
    const Eigen::Matrix4d md = p.matrix();
    pm->final_trafo = md.cast<float>();
    pm->edge.transform = Eigen::Isometry3d(p.rotation().matrix(), p.translation());
    for (int i = 0; i < 6; i++)
      for (int j = i; j < 6; j++) {
        inf >> pm->edge.informationMatrix(i, j);
        pm->edge.informationMatrix(j, i) = pm->edge.informationMatrix(i, j);
      }
    pm->edge.id2 = id_to;
    pm->edge.id1 = id_from;
    if (inf.fail() || inf.eof()) { delete pm; break; }   // trailing whitespace after the last record (:1545)
    mv.push_back(pm);
  }
  cout << __LINE__ << " read vro records " << mv.size() << endl;
}

// gtsam_graph.cpp:1560-1572
void CGraphGT::printVROResult(ostream &ouf, MatchingResult &m) {
  const Vector6 p = cov_Helper(m.final_trafo);
  ouf << m.edge.id2 << " " << m.edge.id1 << " " << p(0) << " " << p(1) << " " << p(2) << " " << p(3) << " " << p(4) << " " << p(5) << " ";
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) ouf << m.edge.informationMatrix(i, j) << " ";
  ouf << endl;
  ouf.flush();
}

// gtsam_graph.cpp:1574-1590: like printVROResult but with the nodes' sequence ids
void CGraphGT::recordVROResult(MatchingResult &m) {
  CCameraNode *pNow = m_graph_map[m.edge.id2];
  CCameraNode *pOld = m_graph_map[m.edge.id1];
  const Vector6 p = cov_Helper(m.final_trafo);
  ofstream *pf = getRecFile();
  (*pf) << pNow->m_seq_id << " " << pOld->m_seq_id << " " << p(0) << " " << p(1) << " " << p(2) << " " << p(3) << " " << p(4) << " " << p(5) << " ";
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) (*pf) << m.edge.informationMatrix(i, j) << " ";
  (*pf) << endl;
  (*pf).flush();
}

// gtsam_graph.cpp:1593-1623; information(0,0) == 10000 is the "void edge" sentinel of the record log
bool CGraphGT::addNodeOffline(CCameraNode *new_node, MatchingResult *mr, bool only_vo) {
  bool ret = true;
  new_node->m_id = (int)m_graph_map.size();
  new_node->m_seq_id = mr->edge.id2;
  if (only_vo || mr->edge.informationMatrix(0, 0) != 10000) {
    m_graph_map[new_node->m_id] = new_node;
    const int pre_id1 = mr->edge.id1, pre_id2 = mr->edge.id2;
    correctMatchingID(mr);
    addToGTSAM(*mr, true);
    mr->edge.id1 = pre_id1;
    mr->edge.id2 = pre_id2;
  } else {
    ret = false;
  }
  return ret;
}

// gtsam_graph.cpp:1625-1650: sequence ids -> graph ids
void CGraphGT::correctMatchingID(MatchingResult *mr) {
  const int from_id = mr->edge.id1, to_id = mr->edge.id2;
  bool from_good = false, to_good = false;
  for (map<int, CCameraNode *>::iterator it = m_graph_map.begin(); it != m_graph_map.end(); ++it) {
    if (it->second->m_seq_id == from_id) { mr->edge.id1 = it->second->m_id; from_good = true; }
    if (it->second->m_seq_id == to_id) { mr->edge.id2 = it->second->m_id; to_good = true; }
    if (from_good && to_good) break;
  }
}

// gtsam_graph.cpp:1652-1668
void CGraphGT::addEdgeOffline(MatchingResult *mr) {
  if (mr->edge.informationMatrix(0, 0) != 10000) {
    const int pre_id1 = mr->edge.id1, pre_id2 = mr->edge.id2;
    correctMatchingID(mr);
    addToGTSAM(*mr, false);
    mr->edge.id1 = pre_id1;
    mr->edge.id2 = pre_id2;
  }
}

// gtsam_graph.cpp:1768-1788
void CGraphGT::optimizeGraphIncremental() {
  mp_isam2->update(*mp_new_fac, *mp_new_node);
  (*mp_node_values) = mp_isam2->calculateEstimate();
  mp_new_fac->resize(0);
  mp_new_node->clear();
}
void CGraphGT::optimizeGraph() { return CGraphGT::optimizeGraphBatch(); }
void CGraphGT::optimizeGraphBatch() {
  LevenbergMarquardtOptimizer optimizer(*mp_fac_graph, *mp_node_values);
  (*mp_node_values) = optimizer.optimize();
}

// gtsam_graph.cpp:1790-1812
bool CGraphGT::isSmallTrafo(MatchingResult &mr) {
  Eigen::Isometry3d &T = mr.edge.transform;
  if (T.translation().norm() > CGTParams::Instance()->m_small_translation) return false;
  const double angle = acos((T.rotation().trace() - 1) * 0.5) * 180. / M_PI;
  return !(angle > CGTParams::Instance()->m_small_rotation);
}
bool CGraphGT::isLargeTrafo(MatchingResult &mr) {
  Eigen::Isometry3d &T = mr.edge.transform;
  if (T.translation().norm() > CGTParams::Instance()->m_large_translation) return true;
  const double angle = acos((T.rotation().trace() - 1) * 0.5) * 180. / M_PI;
  return angle > CGTParams::Instance()->m_large_rotation;
}

size_t CGraphGT::camnodeSize() { return m_graph_map.size(); }

// gtsam_graph.cpp:1819-1840
bool CGraphGT::writeTrajectory(std::string f) {
  ofstream ouf(f.c_str());
  if (!ouf.is_open()) {
    printf("%s failed to open f: %s to write trajectory!\n", __FILE__, f.c_str());
    return false;
  }
  for (map<int, CCameraNode *>::iterator it = m_graph_map.begin(); it != m_graph_map.end(); ++it) {
    Pose3 p = mp_node_values->at<Pose3>(X(it->first));
    p = (*mp_w2o) * p;
    const gtsam::Quaternion q = p.rotation().toQuaternion();
    ouf << it->first << " " << p.x() << " " << p.y() << " " << p.z() << " " << q.x() << " " << q.y() << " " << q.z() << " " << q.w() << " "
        << it->second->m_seq_id << endl;
  }
  return true;
}

// gtsam_graph.cpp:1842-1864, 1927-1939
void CGraphGT::headerPLY(std::ofstream &ouf, int vertex_number) {
  ouf << "ply" << endl << "format ascii 1.0" << endl << "element vertex " << vertex_number << endl << "property float x" << endl
      << "property float y" << endl << "property float z" << endl << "property uchar red" << endl << "property uchar green" << endl
      << "property uchar blue" << endl << "end_header" << endl;
}
bool CGraphGT::trajectoryPLY(std::string f, CG::COLOR c) {
  ofstream ouf(f.c_str());
  if (!ouf.is_open()) {
    printf("%s %d failed to open f: %s to write trajectory!\n", __FILE__, __LINE__, f.c_str());
    return false;
  }
  headerPLY(ouf, (int)m_graph_map.size());
  for (map<int, CCameraNode *>::iterator it = m_graph_map.begin(); it != m_graph_map.end(); ++it) {
    Pose3 p = mp_node_values->at<Pose3>(X(it->first));
    p = (*mp_w2o) * p;
    ouf << p.x() << " " << p.y() << " " << p.z() << " " << (int)CG::g_color[c][0] << " " << (int)CG::g_color[c][1] << " " << (int)CG::g_color[c][2] << endl;
  }
  return true;
}

// gtsam_graph.cpp:1941-1945 (gtsam::writeG2o): poses as VERTEX_SE3:QUAT, between factors as EDGE_SE3:QUAT with the
// information in g2o's [translation; rotation] order
void CGraphGT::writeG2O(std::string f) {
  ofstream ouf(f.c_str());
  if (!ouf.is_open()) {
    cerr << __FILE__ << ": failed to open file " << f << " to write the g2o structure" << endl;
    return;
  }
  ouf << setprecision(12);
  for (map<int, CCameraNode *>::iterator it = m_graph_map.begin(); it != m_graph_map.end(); ++it) {
    const Pose3 p = mp_node_values->at<Pose3>(X(it->first));
    const gtsam::Quaternion q = p.rotation().toQuaternion();
    ouf << "VERTEX_SE3:QUAT " << it->first << " " << p.x() << " " << p.y() << " " << p.z() << " " << q.x() << " " << q.y() << " " << q.z() << " " << q.w() << endl;
  }
  for (size_t k = 0; k < m_edges.size(); ++k) {
    const StoredEdge &e = m_edges[k];
    Matrix6 M;
    int q = 0;
    for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { M(r, c) = e.info[q]; M(c, r) = e.info[q]; ++q; }
    ouf << "EDGE_SE3:QUAT " << e.id1 << " " << e.id2 << " " << e.t[0] << " " << e.t[1] << " " << e.t[2] << " " << e.q[0] << " " << e.q[1] << " " << e.q[2] << " " << e.q[3];
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c) ouf << " " << M((r + 3) % 6, (c + 3) % 6);     // [omega; v] -> [t; r]
    ouf << endl;
  }
}

// gtsam_graph.cpp:1140-1298, the graph side of it: a new landmark is initialised from the measurement moved into the
// world frame (OrientedPlane3::transform with the inverse pose), then OrientedPlane3Factor(z, Covariance(S), X, L)
bool CGraphGT::addPlaneFactor(const gtsam::Vector4 &z, const gtsam::Matrix3 &S, int pose_id, int landmark) {
  if (!mp_node_values->exists(X(pose_id))) return false;
  if (!mp_node_values->exists(L(landmark))) {
    const Pose3 Twu = mp_node_values->at<Pose3>(X(pose_id));
    // plane (n, d) seen in the body frame -> world: n_w = R n, d_w = d - n_w . t
    Vector3 n; n(0) = z(0); n(1) = z(1); n(2) = z(2);
    const Vector3 nw = Twu.rotation() * n;
    const double dw = z(3) - (nw(0) * Twu.x() + nw(1) * Twu.y() + nw(2) * Twu.z());
    const OrientedPlane3 ONW(nw(0), nw(1), nw(2), dw);
    mp_node_values->insert(L(landmark), ONW);
    mp_new_node->insert(L(landmark), ONW);
    mv_plane_num[landmark] = 0;
    if (landmark >= m_plane_landmark_id) m_plane_landmark_id = landmark + 1;
  }
  OrientedPlane3Factor plane_factor(z, noiseModel::Gaussian::Covariance(S), X(pose_id), L(landmark));
  mp_fac_graph->add(plane_factor);
  mp_new_fac->add(plane_factor);
  mv_plane_num[landmark]++;
  mv_plane_last_seen[landmark] = pose_id;
  return true;
}

gtsam::Matrix6 CGraphGT::marginalCovariance(int pose_id) {
  Marginals marginals(*mp_fac_graph, *mp_node_values);
  return marginals.marginalCovariance(X(pose_id));
}
