// CGraphG2O on the MI355X optimiser.  Behaviour follows the reference wrapper step by step
// (reference g2o/g2o_graph.cpp, cited per method); everything numerical is delegated through
// g2o::SparseOptimizer (fgo_optimizer.h) to the fgo C-ABI.  The Qt thread pool the reference uses for the
// look-back matches (g2o_graph.cpp:207-211) belongs to the front end and is replaced by a plain loop.
#include "g2o_graph.h"
#include <ros/ros.h>
#include <cmath>
#include <iostream>
#include <vector>
#include "camera_node.h"
#include "fgo_optimizer.h"
#include "g2o_parameter.h"
#include "matching_result.h"
#include "misc.h"

CGraphG2O::CGraphG2O() : m_sequence_id(0), mp_optimizer(NULL) { createOptimizer(); }

CGraphG2O::~CGraphG2O() {
  // the graph owns the camera nodes it accepted (reference :44-48) and the optimiser (:50-56)
  for (std::map<int, CCameraNode *>::iterator it = m_graph_map.begin(); it != m_graph_map.end(); ++it) delete it->second;
  m_graph_map.clear();
  delete mp_optimizer;
  mp_optimizer = NULL;
}

void CGraphG2O::setWorld2Original(double) { m_w2o = tf::Transform(); }   // reference :59-63 (identity)

// reference :65-77 — LM over BlockSolver<6,3> over sparse Cholesky; here the choice is made inside libfgo
g2o::SparseOptimizer *CGraphG2O::createOptimizer() {
  delete mp_optimizer;
  mp_optimizer = new g2o::SparseOptimizer();
  mp_optimizer->setVerbose(false);
  return mp_optimizer;
}

// reference :80-94 — vertex 0 at identity, fixed (gauge)
void CGraphG2O::firstNode(CCameraNode *n) {
  n->m_id = (int)m_graph_map.size();
  m_sequence_id = 0;
  n->m_seq_id = ++m_sequence_id;
  mp_optimizer->addVertexSE3(n->m_id, Eigen::Isometry3d::Identity(), true);
  m_graph_map[n->m_id] = n;
}

// reference :96-134 — create the missing endpoint by chaining, optionally reset the estimate, add EdgeSE3
bool CGraphG2O::addToGraph(MatchingResult &mr, bool set_estimate) {
  const bool has1 = mp_optimizer->hasVertex(mr.edge.id1), has2 = mp_optimizer->hasVertex(mr.edge.id2);
  if (!has1 && !has2) {
    ROS_ERROR("%s two nodes %i and %i both not exist!", __FILE__, mr.edge.id1, mr.edge.id2);
    return false;
  } else if (!has1) {
    ROS_WARN("this case is weired, has not solved it");
    mp_optimizer->addVertexSE3(mr.edge.id1, mp_optimizer->estimate(mr.edge.id2) * mr.edge.transform.inverse(), false);
  } else if (!has2) {
    mp_optimizer->addVertexSE3(mr.edge.id2, mp_optimizer->estimate(mr.edge.id1) * mr.edge.transform, false);
  } else if (set_estimate) {
    mp_optimizer->setEstimate(mr.edge.id2, mp_optimizer->estimate(mr.edge.id1) * mr.edge.transform);
  }
  return mp_optimizer->addEdgeSE3(mr.edge.id1, mr.edge.id2, mr.edge.transform, mr.edge.informationMatrix);
}

// reference :136-157 — identity edge with information 1e-3 * I to the previous node
void CGraphG2O::fakeOdoNode(CCameraNode *new_node) {
  if (new_node->m_id != (int)m_graph_map.size()) {
    std::cerr << __FILE__ << " " << __LINE__ << " Here this should not happen!" << std::endl;
    new_node->m_id = (int)m_graph_map.size();
    new_node->m_seq_id = ++m_sequence_id;
  }
  CCameraNode *pre_node = m_graph_map[new_node->m_id - 1];
  MatchingResult mr;
  mr.edge.id1 = pre_node->m_id;
  mr.edge.id2 = new_node->m_id;
  mr.edge.transform.setIdentity();
  mr.edge.informationMatrix = Eigen::Matrix<double, 6, 6>::Identity() * 1e-3;
  addToGraph(mr, false);
  m_graph_map[new_node->m_id] = new_node;
}

// reference :159-239
ADD_RET CGraphG2O::addNode(CCameraNode *new_node) {
  if (m_graph_map.size() == 0) {
    firstNode(new_node);
    return SUCC_KF;
  }
  const size_t old_node_size = camnodeSize();
  new_node->m_id = (int)m_graph_map.size();
  new_node->m_seq_id = ++m_sequence_id;
  CCameraNode *pre_node = m_graph_map[new_node->m_id - 1];
  MatchingResult mr = new_node->matchNodePair(pre_node);          // odometry match (:174)
  size_t current_best_match = 0;
  if (mr.succeed_match) {
    if (isSmallTrafo(mr)) return FAIL_NOT_KF;                      // :179-183
    addToGraph(mr, true);                                          // :186
    m_graph_map[new_node->m_id] = new_node;
    current_best_match = mr.inlier_matches.size();
  } else {
    ROS_ERROR("%s Found no transformation to predecessor", __FILE__);
  }
  // local loop closures: id-2 ... id-1-lookback once the map holds more than 3 nodes (:196-205)
  std::vector<CCameraNode *> nodes_to_comp;
  if (m_graph_map.size() > 3) {
    int n_id = new_node->m_id - 2;
    for (int j = 0; j < CG2OParams::Instance()->m_lookback_nodes && n_id >= 0; ++j) nodes_to_comp.push_back(m_graph_map[n_id--]);
  }
  for (size_t i = 0; i < nodes_to_comp.size(); ++i) {
    MatchingResult r = new_node->matchNodePair(nodes_to_comp[i]);
    if (!r.succeed_match || isSmallTrafo(r)) continue;             // :214-217
    const bool reset_estimate = r.inlier_matches.size() > current_best_match;
    if (reset_estimate) current_best_match = r.inlier_matches.size();
    addToGraph(r, reset_estimate);
    m_graph_map[new_node->m_id] = new_node;
  }
  return camnodeSize() > old_node_size ? SUCC_KF : FAIL_KF;        // :230-238
}

// reference :241-252 — 20 iterations issued as optimize(ceil(20/10)) until the budget is used.  The
// reference's loop never ends if optimize() returns <= 0; here a failed call stops it.
void CGraphG2O::optimizeGraph() {
  const int iter = 20;
  int currIt = 0;
  mp_optimizer->initializeOptimization();
  for (int i = 0; i < iter; i += currIt) {
    currIt = mp_optimizer->optimize((int)std::ceil(iter / 10));
    if (currIt <= 0) break;
  }
}

// reference :254-258
double CGraphG2O::error() {
  mp_optimizer->computeActiveErrors();
  return mp_optimizer->chi2();
}

// reference :261-271 — small translation AND small rotation
bool CGraphG2O::isSmallTrafo(MatchingResult &mr) {
  Eigen::Isometry3d &T = mr.edge.transform;
  if (T.translation().norm() > CG2OParams::Instance()->m_small_translation) return false;
  double c = (T.rotation().trace() - 1) * 0.5;
  c = c > 1 ? 1 : (c < -1 ? -1 : c);
  if (std::acos(c) * 180. / M_PI > CG2OParams::Instance()->m_small_rotation) return false;
  return true;
}

size_t CGraphG2O::camnodeSize() { return m_graph_map.size(); }

void CGraphG2O::writeG2O(std::string f) {          // reference :279-283
  std::ofstream ouf(f.c_str());
  mp_optimizer->save(ouf);
}

// reference :285-308 — "graph_id x y z qx qy qz qw seq_id" per line, pose pre-multiplied by m_w2o
bool CGraphG2O::writeTrajectory(std::string f) {
  std::ofstream ouf(f.c_str());
  if (!ouf.is_open()) {
    ROS_ERROR("%s failed to open file : %s", __FILE__, f.c_str());
    return false;
  }
  for (std::map<int, CCameraNode *>::iterator it = m_graph_map.begin(); it != m_graph_map.end(); ++it) {
    const tf::Transform w2p = m_w2o * eigenTransf2TF(mp_optimizer->estimate(it->second->m_id));
    ouf << it->second->m_id << " " << w2p.getOrigin().x() << " " << w2p.getOrigin().y() << " " << w2p.getOrigin().z() << " "
        << w2p.getRotation().x() << " " << w2p.getRotation().y() << " " << w2p.getRotation().z() << " " << w2p.getRotation().w()
        << " " << it->second->m_seq_id << std::endl;
  }
  return true;
}

// reference :310-335
bool CGraphG2O::trajectoryPLY(std::string f, COLOR c) {
  std::ofstream ouf(f.c_str());
  if (!ouf.is_open()) {
    std::printf("%s %d failed to open f: %s to write trajectory!\n", __FILE__, __LINE__, f.c_str());
    return false;
  }
  headerPLY(ouf, (int)m_graph_map.size());
  for (std::map<int, CCameraNode *>::iterator it = m_graph_map.begin(); it != m_graph_map.end(); ++it) {
    const tf::Transform w2p = m_w2o * eigenTransf2TF(mp_optimizer->estimate(it->second->m_id));
    ouf << w2p.getOrigin().x() << " " << w2p.getOrigin().y() << " " << w2p.getOrigin().z() << " " << (int)g_color[c][0] << " "
        << (int)g_color[c][1] << " " << (int)g_color[c][2] << std::endl;
  }
  return true;
}

// reference :337-349
void CGraphG2O::headerPLY(std::ofstream &ouf, int vertex_number) {
  ouf << "ply\nformat ascii 1.0\nelement vertex " << vertex_number
      << "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header"
      << std::endl;
}
