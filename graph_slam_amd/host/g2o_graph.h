// CGraphG2O — pose-graph wrapper with the reference's public surface (reference g2o/g2o_graph.h:21-52):
// identical method names, argument meaning, return conventions (bool / ADD_RET, never throws) and public data
// members, so drivers such as g2o/test_g2o_graph.cpp compile and link unchanged.  The optimiser behind
// `mp_optimizer` is the MI355X back-end (fgo C-ABI) instead of g2o.
#ifndef FGO_HOST_G2O_GRAPH_H
#define FGO_HOST_G2O_GRAPH_H

#include <fstream>
#include <map>
#include <string>
#include <tf/tf.h>
#include "color.h"

namespace g2o { class SparseOptimizer; }
class CCameraNode;
class MatchingResult;

// SUCC_KF: node added as a keyframe; FAIL_NOT_KF: motion too small, caller keeps ownership and discards;
// FAIL_KF: no transformation found (caller may insert a fake-odometry node)
typedef enum { SUCC_KF, FAIL_NOT_KF, FAIL_KF } ADD_RET;

class CGraphG2O {
 public:
  CGraphG2O();
  virtual ~CGraphG2O();

  g2o::SparseOptimizer *createOptimizer();

  void firstNode(CCameraNode *);
  ADD_RET addNode(CCameraNode *);
  void fakeOdoNode(CCameraNode *);
  void optimizeGraph();
  bool addToGraph(MatchingResult &, bool set_estimate);
  bool isSmallTrafo(MatchingResult &);
  double error();
  size_t camnodeSize();
  void writeG2O(std::string ouf);
  bool writeTrajectory(std::string ouf);

  int m_sequence_id;
  std::map<int, CCameraNode *> m_graph_map;
  g2o::SparseOptimizer *mp_optimizer;
  void setWorld2Original(double p);
  tf::Transform m_w2o;

  void headerPLY(std::ofstream &, int vertex_number);
  bool trajectoryPLY(std::string ouf, COLOR);
};

#endif
