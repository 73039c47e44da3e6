// imu_interface mirror: CImuVn100 (gtsam/imu_vn100.h:19-38): VN100 log reader + noise parameters.
#ifndef FGO_HOST_IMU_VN100_H
#define FGO_HOST_IMU_VN100_H
#include "imu_base.h"

typedef std::vector<Eigen::Vector3d> stdv_eigen_vector3d;

class CImuVn100 : public CImuBase {
 public:
  CImuVn100(double dt, gtsam::imuBias::ConstantBias prior_bias);
  virtual ~CImuVn100();
  virtual void setStartPoint(double t);
  virtual bool readImuData(std::string f);                 // t ax ay az gx gy gz yaw pitch roll per line
  bool getRPYAt(double t, Eigen::Vector3d &rpy);
  std::shared_ptr<gtsam::PreintegratedCombinedMeasurements::Params> getIMUParams();   // VN100 noise (imu_vn100.cpp:24-67)
  stdv_eigen_vector3d mv_rpy;
  Eigen::Vector3d mp_ini_rpy;
};
#endif
