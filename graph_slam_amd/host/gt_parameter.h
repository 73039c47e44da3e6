// Keyframe / optimisation knobs of the GTSAM-side drivers.  Same member names, order and singleton accessor as the
// reference's CGTParams (gtsam/gt_parameter.h:16-38, defaults gtsam/gt_parameter.cpp:15-27).
#ifndef FGO_HOST_GT_PARAMETER_H
#define FGO_HOST_GT_PARAMETER_H
#include <cmath>
#include <string>

#ifndef D2R
#define D2R(d) (((d) * M_PI) / 180.)
#define R2D(r) (((r) * 180.) / M_PI)
#endif

class CGTParams {
 public:
  ~CGTParams();
  int m_lookback_nodes;
  double m_small_translation;   // [m]
  double m_small_rotation;      // [deg]
  double m_large_translation;   // [m]
  double m_large_rotation;      // [deg]
  int m_optimize_step;
  std::string m_output_dir;
  bool m_record_vro_results;
  double m_initial_pitch;       // [deg]
  std::string m_vro_result;
  static CGTParams *Instance();

 private:
  CGTParams();
  static CGTParams *mp_instance;
};
#endif
