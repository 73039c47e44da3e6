// CImuVn100 mirror: gtsam/imu_vn100.cpp.
#include "imu_vn100.h"

#include <cstdio>
#include <fstream>

using namespace std;
using namespace gtsam;

CImuVn100::CImuVn100(double dt, gtsam::imuBias::ConstantBias prior_bias) : CImuBase(dt, prior_bias) {
  std::shared_ptr<PreintegratedCombinedMeasurements::Params> p = getIMUParams();
  mp_combined_pre_imu = new PreintegratedCombinedMeasurements(p, m_prior_imu_bias);
}
CImuVn100::~CImuVn100() {}

// imu_vn100.cpp:24-67: the sensor-spec noise model (0.0035 deg/s/sqrt(Hz), 0.14 mg/sqrt(Hz), 10 deg/h, 0.04 mg,
// integration 1e-4, bias-in-preintegration 1e-3) is what fgo_imu_params_vn100 fills in; gravity stays CImuBase's
std::shared_ptr<PreintegratedCombinedMeasurements::Params> CImuVn100::getIMUParams() {
  std::shared_ptr<PreintegratedCombinedMeasurements::Params> p = CImuBase::getParam();
  static bool b_once = true;
  if (b_once) {
    fgo_imu_params_vn100(&p->p);
    b_once = false;
  }
  return p;
}

bool CImuVn100::getRPYAt(double t, Eigen::Vector3d &rpy) {
  const int index = findIndexAt(t);
  if (index < 0) return false;
  for (int k = 0; k < 3; ++k) rpy(k) = mv_rpy[m_syn_start_id + index](k) - mp_ini_rpy(k);
  return true;
}

// imu_vn100.cpp:78-105
bool CImuVn100::readImuData(string fname) {
  ifstream inf(fname.c_str());
  if (!inf.is_open()) {
    printf("%s failed to open imu file %s\n", __FILE__, fname.c_str());
    return false;
  }
  double t;
  float ax, ay, az, gx, gy, gz, yaw, pitch, roll;      // the reference parses the samples as float (imu_vn100.cpp:86)
  while (inf >> t >> ax >> ay >> az >> gx >> gy >> gz >> yaw >> pitch >> roll) {
    Eigen::Vector6d m;
    m(0) = gx; m(1) = gy; m(2) = gz; m(3) = ax; m(4) = ay; m(5) = az;
    Eigen::Vector3d rpy;
    rpy(0) = roll; rpy(1) = pitch; rpy(2) = yaw;
    mv_measurements.push_back(m);
    mv_timestamps.push_back(t);
    mv_rpy.push_back(rpy);
  }
  printf("%s succeed to load %i imu measurements\n", __FILE__, (int)mv_timestamps.size());
  return true;
}

void CImuVn100::setStartPoint(double t) {
  CImuBase::setStartPoint(t);
  if (m_syn_start_id < (int)mv_rpy.size()) mp_ini_rpy = mv_rpy[m_syn_start_id];
}
