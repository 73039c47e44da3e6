#ifndef FGO_HOST_MISC_H
#define FGO_HOST_MISC_H
#include <tf/tf.h>
#include <Eigen/Core>
#include <Eigen/Geometry>
// Isometry -> tf::Transform (role of the reference's g2o/misc.h)
template <typename T>
tf::Transform eigenTransf2TF(const T &iso) {
  Eigen::Quaterniond q(iso.rotation());
  return tf::Transform(tf::Quaternion(q.x(), q.y(), q.z(), q.w()),
                       tf::Vector3(iso.translation()(0), iso.translation()(1), iso.translation()(2)));
}
#endif
