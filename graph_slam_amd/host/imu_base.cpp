// CImuBase mirror: gtsam/imu_base.cpp.
#include "imu_base.h"

#include <cstdio>
#include <iostream>

using namespace gtsam;
using namespace std;

CImuBase::CImuBase(double delta_t, gtsam::imuBias::ConstantBias prior_bias)
    : m_curr_i(0), m_syn_start_id(0), m_prior_imu_bias(prior_bias), m_prev_imu_bias(prior_bias), m_dt((float)delta_t),
      mp_combined_pre_imu(nullptr) {}

CImuBase::~CImuBase() {
  delete mp_combined_pre_imu;
  mp_combined_pre_imu = nullptr;
}

double CImuBase::getLastTimeStamp() {
  if (!mv_timestamps.empty()) return mv_timestamps.back();
  cerr << __FILE__ << " at " << __LINE__ << " no imu timestamps available!" << endl;
  return 0;
}

bool CImuBase::predictNextFlag(double t, gtsam::NavState &s) {
  const int index = findIndexAt(t);
  if (index < 0) {
    cerr << __FILE__ << " failed to predictNext given t = " << t << endl;
    return false;
  }
  return predictNextFlag(index, s);
}

gtsam::NavState CImuBase::predictNext(double t) {
  const int index = findIndexAt(t);
  if (index < 0) {
    cerr << __FILE__ << " failed to predictNext given t = " << t << endl;
    return gtsam::NavState();
  }
  return predictNext(index);
}

bool CImuBase::predictNextFlag(int next_t, gtsam::NavState &s) {
  if (next_t < 0) return false;
  s = predictNext(next_t);
  return true;
}

// imu_base.cpp:72-87: integrateMeasurement(acc, gyro, dt) over [m_curr_i, next_i), then predict
gtsam::NavState CImuBase::predictNext(int next_i) {
  for (int i = m_syn_start_id + m_curr_i; i < m_syn_start_id + next_i; i++) {
    if (i >= (int)mv_measurements.size()) {
      printf("%s i >= mv_measurements.size()\n", __FILE__);
      break;
    }
    const Eigen::Vector6d &imu = mv_measurements[i];
    Vector3 acc, gyro;
    for (int k = 0; k < 3; ++k) { gyro(k) = imu(k); acc(k) = imu(3 + k); }
    mp_combined_pre_imu->integrateMeasurement(acc, gyro, m_dt);
  }
  m_curr_i = next_i;
  return mp_combined_pre_imu->predict(m_prev_state, m_prev_imu_bias);
}

void CImuBase::resetPreintegrationAndBias(gtsam::imuBias::ConstantBias bias) {
  m_prev_imu_bias = bias;
  mp_combined_pre_imu->resetIntegrationAndSetBias(bias);
}
void CImuBase::resetPreintegrationAndBias() { mp_combined_pre_imu->resetIntegrationAndSetBias(m_prev_imu_bias); }

bool CImuBase::readImuData(string) {
  printf("%s readImuData not implemented\n", __FILE__);
  return false;
}

void CImuBase::setStartPoint(double t) {
  m_syn_start_id = 0;
  const int index = findIndexAt(t);
  if (index < 0) {
    cerr << __FILE__ << " failed to synchronize with timestamp t = " << t << endl;
    return;
  }
  m_syn_start_id = index;
}

// imu_base.cpp:123-154: index (relative to the synchronisation point) of the sample nearest to t
int CImuBase::findIndexAt(double t) {
  if (mv_timestamps.size() != mv_measurements.size()) {
    cerr << __FILE__ << " something is wrong: mv_timestamps.size() != mv_measurements.size()" << endl;
    return -1;
  }
  const int e = (int)mv_timestamps.size() - 1;
  for (int i = 0; i + m_syn_start_id <= e; i++) {
    if (mv_timestamps[i + m_syn_start_id] > t) {
      if (i >= 1) {
        if (mv_timestamps[i + m_syn_start_id] - t > t - mv_timestamps[i + m_syn_start_id - 1]) return i - 1;
        return i;
      }
      return i;
    }
  }
  return -1;
}

// imu_base.cpp:156-170
gtsam::NavState CImuBase::predictBetween(int i, int j, gtsam::NavState &state_i, gtsam::imuBias::ConstantBias bias_i) {
  resetPreintegrationAndBias(bias_i);
  for (int m = i; m < j; m++) {
    if (m >= (int)mv_measurements.size()) {
      printf("%s m >= mv_measurements.size()\n", __FILE__);
      break;
    }
    const Eigen::Vector6d &imu = mv_measurements[m];
    Vector3 acc, gyro;
    for (int k = 0; k < 3; ++k) { gyro(k) = imu(k); acc(k) = imu(3 + k); }
    mp_combined_pre_imu->integrateMeasurement(acc, gyro, m_dt);
  }
  return mp_combined_pre_imu->predict(state_i, m_prev_imu_bias);
}

void CImuBase::setState(gtsam::NavState &ns) { m_prev_state = ns; }

void CImuBase::getNormalizedAcc(double &ax, double &ay, double &az) {
  if (m_syn_start_id <= 0) return getNormalizedAcc(1, ax, ay, az);
  return getNormalizedAcc(m_syn_start_id, ax, ay, az);
}
// imu_base.cpp:192-215: mean accelerometer direction over the first `index` samples
void CImuBase::getNormalizedAcc(int index, double &ax, double &ay, double &az) {
  if (index > (int)mv_measurements.size()) index = (int)mv_measurements.size();
  if (index <= 0) {
    cerr << __FILE__ << " getNormalizedAcc at index = " << index << endl;
    return;
  }
  double wx = 0, wy = 0, wz = 0;
  for (int i = 0; i < index; i++) { wx += mv_measurements[i](3); wy += mv_measurements[i](4); wz += mv_measurements[i](5); }
  wx /= index; wy /= index; wz /= index;
  const double norm = sqrt(wx * wx + wy * wy + wz * wz);
  ax = wx / norm; ay = wy / norm; az = wz / norm;
}

void CImuBase::resetGravity(double gx, double gy, double gz) {
  getParam()->n_gravity[0] = gx; getParam()->n_gravity[1] = gy; getParam()->n_gravity[2] = gz;
}

// imu_base.cpp:258-263
std::shared_ptr<gtsam::PreintegratedCombinedMeasurements::Params> CImuBase::getParam() {
  static std::shared_ptr<PreintegratedCombinedMeasurements::Params> p = PreintegratedCombinedMeasurements::Params::MakeSharedD(9.71);
  return p;
}
