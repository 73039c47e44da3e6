// imu_interface mirror: CImuBase (gtsam/imu_base.h:27-78).  Owns the IMU log and a PreintegratedCombinedMeasurements,
// integrates the samples between two keyframes and predicts the next NavState -- the producer of the payload of every
// CombinedImuFactor (gtsam/imu_base.cpp:72-87).  Same member / method names; boost::shared_ptr -> std::shared_ptr
// (boost is not in this image).  The integration itself is the host side of the C-ABI (fgo_preint_*).
#ifndef FGO_HOST_IMU_BASE_H
#define FGO_HOST_IMU_BASE_H
#include <cmath>
#include <memory>
#include <string>
#include <vector>
#include <Eigen/Core>
#include <gtsam/navigation/CombinedImuFactor.h>
#include <gtsam/navigation/ImuFactor.h>

#ifndef D2R
#define D2R(d) (((d) * M_PI) / 180)
#define R2D(r) (((r) * 180) / M_PI)
#endif

namespace Eigen {
typedef Matrix<double, 6, 1> Vector6d;
}
typedef std::vector<Eigen::Vector6d> stdv_eigen_vector6d;

class CImuBase {
 public:
  CImuBase(double delta_t, gtsam::imuBias::ConstantBias prior_bias);
  virtual ~CImuBase();

  virtual void setStartPoint(double t);        // the first imu measurement synchronised with camera data
  virtual bool readImuData(std::string f);
  virtual int findIndexAt(double t);
  virtual bool predictNextFlag(double t, gtsam::NavState &);
  virtual bool predictNextFlag(int next_i, gtsam::NavState &);
  virtual gtsam::NavState predictNext(int next_i);   // integrate [curr_i, next_i), predict from m_prev_state
  virtual gtsam::NavState predictNext(double t);
  int m_curr_i;
  double getLastTimeStamp();
  void resetGravity(double gx, double gy, double gz);
  void getNormalizedAcc(int index, double &ax, double &ay, double &az);
  void getNormalizedAcc(double &ax, double &ay, double &az);

  static std::shared_ptr<gtsam::PreintegratedCombinedMeasurements::Params> getParam();   // gravity 9.71, shared singleton
  virtual std::shared_ptr<gtsam::PreintegratedCombinedMeasurements::Params> getIMUParams() = 0;
  virtual gtsam::NavState predictBetween(int i, int j, gtsam::NavState &state_i, gtsam::imuBias::ConstantBias bias_i);

  virtual void resetPreintegrationAndBias(gtsam::imuBias::ConstantBias bias);
  virtual void resetPreintegrationAndBias();
  void setState(gtsam::NavState &);

  int m_syn_start_id;
  gtsam::imuBias::ConstantBias m_prior_imu_bias;
  gtsam::imuBias::ConstantBias m_prev_imu_bias;
  stdv_eigen_vector6d mv_measurements;          // gx gy gz, ax ay az
  std::vector<double> mv_timestamps;
  float m_dt;
  gtsam::NavState m_prev_state;
  gtsam::PreintegrationType *mp_combined_pre_imu;
};
#endif
