// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <sophus/so3.hpp>: a unit quaternion with Sophus' storage order data() = [qx, qy, qz, qw]
// (Relocator::UpdateNewSubmap hands data() to Ceres as a 4-block with EigenQuaternionParameterization, relocator.cpp:255-257) and the action on
// vectors (pose_graph.cpp:58,66: so3() * Vector3d::UnitX()).
#pragma once
#include <Eigen/Core>
namespace Sophus {
class SO3d {
 public:
  SO3d() { q_[0] = q_[1] = q_[2] = 0.0; q_[3] = 1.0; }
  explicit SO3d(const double* q4) { for (int i = 0; i < 4; ++i) q_[i] = q4[i]; }
  double* data() { return q_; }
  const double* data() const { return q_; }
  Eigen::Quaterniond unit_quaternion() const { return Eigen::Quaterniond(q_[3], q_[0], q_[1], q_[2]); }
  Eigen::Vector3d operator*(const Eigen::Vector3d& v) const {      // Eigen's QuaternionBase::_transformVector
    const Eigen::Vector3d qv(q_[0], q_[1], q_[2]);
    Eigen::Vector3d uv = qv.cross(v);
    uv = uv + uv;
    const Eigen::Vector3d c2 = qv.cross(uv);
    return Eigen::Vector3d(v.x() + q_[3] * uv.x() + c2.x(), v.y() + q_[3] * uv.y() + c2.y(), v.z() + q_[3] * uv.z() + c2.z());
  }
 private:
  double q_[4];
};
}  // namespace Sophus
