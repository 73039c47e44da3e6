// CGraphGT: the reference's GTSAM-side graph wrapper (gtsam/gtsam_graph.h:46-150) re-implemented on top of the libfgo
// C-ABI.  Same class name, method names, argument meaning, return conventions (bool / ADD_RET, no exceptions) and the
// public members the drivers dereference (mp_fac_graph, mp_node_values, mp_new_fac, mp_new_node, mp_w2o, mp_u2c,
// mp_prev_bias, mp_prev_state, mv_vro_res, m_graph_map) -- for the part of the class that is on the optimiser path
// (b): first node + priors, offline node / edge insertion from VRO records, NavState
// insertion, the VRO record log reader / writer, batch optimisation, error, trajectory / g2o / PLY writers, plane
// factors from plane coefficients.  The front-end pieces of the class (feature matching in addNode, two-view bundle
// adjustment, VRO covariance estimation, plane segmentation / association on images) are out of scope
// and are not declared here.  gtsam:: types come from shim/gtsam_lite.h (handles onto one fgo context).
#ifndef FGO_HOST_GTSAM_GRAPH_H
#define FGO_HOST_GTSAM_GRAPH_H

#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>
#include <gtsam/base/Matrix.h>
#include "gt_color.h"

class CCameraNode;
class MatchingResult;

#ifndef FGO_HOST_ADD_RET
#define FGO_HOST_ADD_RET
typedef enum { SUCC_KF, FAIL_NOT_KF, FAIL_KF } ADD_RET;
#endif

class CGraphGT {
 public:
  CGraphGT();
  virtual ~CGraphGT();

  void firstNode(CCameraNode *, bool online = true);   // gauge by priors (gtsam_graph.cpp:320-368)
  void fakeOdoNode(CCameraNode *);                      // identity edge, information 1e4 I (:697-722)
  void optimizeGraph();                                 // = optimizeGraphBatch (:1779-1782)
  void optimizeGraphBatch();                            // LevenbergMarquardtOptimizer, GTSAM defaults (:1784-1788)
  void optimizeGraphIncremental();                      // ISAM2 update + calculateEstimate (:1768-1776)
  bool addToGTSAM(MatchingResult &, bool set_estimate); // BetweenFactor<Pose3> in the IMU frame (:630-695)
  bool addToGTSAM(gtsam::NavState &, int vid, bool add_pose);   // X / V / B values of a new state (:613-628)

  bool isSmallTrafo(MatchingResult &);
  bool isLargeTrafo(MatchingResult &);
  double error();                                       // mp_fac_graph->error(values) = chi2 / 2 (:173-176)
  size_t camnodeSize();
  void writeG2O(std::string ouf);                       // VERTEX_SE3:QUAT / EDGE_SE3:QUAT (:1941-1945, gtsam::writeG2o)
  bool writeTrajectory(std::string ouf);                // id x y z qx qy qz qw seq_id (:1819-1840)

  int m_sequence_id;
  int m_vertex_id;

  std::map<int, CCameraNode *> m_graph_map;
  gtsam::NonlinearFactorGraph *mp_fac_graph;
  gtsam::Values *mp_node_values;
  void setWorld2Original(double r, double p, double y);
  void setWorld2Original(double p);
  void setCamera2IMU(double p);
  void setCamera2IMUTranslation(double px, double py, double pz);
  gtsam::Pose3 *mp_w2o;
  gtsam::Pose3 *mp_u2c;
  gtsam::imuBias::ConstantBias *mp_prev_bias;
  gtsam::NavState *mp_prev_state;

  bool mb_record_vro_results;
  std::ofstream *getRecFile();
  void recordVROResult(MatchingResult &m);
  void printVROResult(std::ostream &ouf, MatchingResult &m);
  void readVRORecord(std::string inf);
  void readVRORecord(std::string inf, std::vector<MatchingResult *> &mv);

  // offline operation
  std::vector<MatchingResult *> mv_vro_res;
  bool addNodeOffline(CCameraNode *, MatchingResult *, bool only_vo = false);
  void addEdgeOffline(MatchingResult *);
  void correctMatchingID(MatchingResult *mr);

  // ISAM2 staging objects: the drivers add every factor / value to these as well (test_vro_imu_graph.cpp:193-195)
  gtsam::NonlinearFactorGraph *mp_new_fac;
  gtsam::Values *mp_new_node;
  gtsam::ISAM2 *mp_isam2;                               // gtsam_graph.h:107-112
  gtsam::ISAM2Params *mp_isam2_param;
  void initISAM2Params();

  // ply
  void headerPLY(std::ofstream &, int vertex_number);
  bool trajectoryPLY(std::string ouf, CG::COLOR);

  // planes: landmark bookkeeping as in the reference (:1140-1298); the measurement arrives as plane coefficients in
  // the body frame + its 3x3 covariance (what the reference derives from a segmented CPlane)
  std::map<int, int> mv_plane_num;
  std::map<int, int> mv_plane_last_seen;
  int m_plane_landmark_id;
  bool addPlaneFactor(const gtsam::Vector4 &plane_in_body, const gtsam::Matrix3 &S, int pose_id, int landmark);

  // 6x6 marginal covariance of a pose (Marginals::marginalCovariance, :598-601)
  gtsam::Matrix6 marginalCovariance(int pose_id);

 private:
  std::shared_ptr<gtsam::Backend> m_backend;
  struct StoredEdge { int id1, id2; double t[3], q[4], info[21]; };
  std::vector<StoredEdge> m_edges;      // for writeG2O
  std::ofstream *mp_rec_file;
};

#endif
