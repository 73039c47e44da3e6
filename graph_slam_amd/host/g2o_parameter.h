// Keyframe / optimisation knobs the drivers set (g2o/test_g2o_graph.cpp:152-162).  Same member names, member
// order and singleton accessor as the reference's CG2OParams (g2o/g2o_parameter.h:18-39): the reference's
// drivers compile against THEIR copy of this header, so the object layout and the out-of-line symbols
// (Instance(), ctor, dtor, mp_instance) are part of the link-level contract.
#ifndef FGO_HOST_G2O_PARAMETER_H
#define FGO_HOST_G2O_PARAMETER_H
#include <cmath>
#include <string>

#ifndef D2R
#define D2R(d) (((d) * M_PI) / 180.)
#define R2D(r) (((r) * 180.) / M_PI)
#endif

class CG2OParams {
 public:
  ~CG2OParams();
  int m_lookback_nodes;        // look-back candidates per new node
  double m_small_translation;  // [m]   below both thresholds a node is not a keyframe
  double m_small_rotation;     // [deg]
  int m_optimize_step;         // optimise every n keyframes
  std::string m_output_dir;
  double m_initial_pitch;      // [deg]
  static CG2OParams *Instance();

 private:
  CG2OParams();
  static CG2OParams *mp_instance;
};
#endif
