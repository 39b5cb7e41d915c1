// ORACLE — TEST INFRASTRUCTURE ONLY.  Minimal stand-in for <sophus/se3.hpp>: SE3d with Sophus' storage order
// data() = [qx, qy, qz, qw, tx, ty, tz] (SURVEY.md §8 conventions), inverse() (conjugate + rotated negated translation),
// group product and action on points (Hamilton product / unit-quaternion rotation).  The cost functors only read data();
// inverse()/operator* are used by PoseGraphError's first constructor (pose_error.hpp:13-17) and sensor.h's inline methods.
#pragma once
#include <Eigen/Core>
#include <sophus/so3.hpp>

namespace Sophus {

// SE3f: the value of SE3d::cast<float>() — every coefficient rounded to the nearest float — as a container the lidar code hands to
// ceres::SE3TransformPoint<float> through data() (association.cpp:238-239, :286-287)
class SE3f {
 public:
  SE3f() { d_[0] = d_[1] = d_[2] = 0.f; d_[3] = 1.f; d_[4] = d_[5] = d_[6] = 0.f; }
  float* data() { return d_; }
  const float* data() const { return d_; }
 private:
  float d_[7];
};

class SE3d {
 public:
  static constexpr int num_parameters = 7;
  SE3d() { d_[0] = d_[1] = d_[2] = 0.0; d_[3] = 1.0; d_[4] = d_[5] = d_[6] = 0.0; }
  explicit SE3d(const double* data7) { for (int i = 0; i < 7; ++i) d_[i] = data7[i]; }     // shim-only convenience
  SE3d(const SO3d& r, const Eigen::Vector3d& t) { for (int i = 0; i < 4; ++i) d_[i] = r.data()[i]; d_[4] = t.x(); d_[5] = t.y(); d_[6] = t.z(); }
  SO3d so3() const { return SO3d(d_); }
  double* data() { return d_; }
  const double* data() const { return d_; }
  template <typename T> SE3f cast() const { SE3f r; for (int i = 0; i < 7; ++i) r.data()[i] = (T)d_[i]; return r; }
  Eigen::Vector3d translation() const { return Eigen::Vector3d(d_[4], d_[5], d_[6]); }
  Eigen::Quaterniond unit_quaternion() const { return Eigen::Quaterniond(d_[3], d_[0], d_[1], d_[2]); }
  Eigen::Matrix3d rotationMatrix() const { return unit_quaternion().toRotationMatrix(); }
  SE3d inverse() const {
    SE3d r;
    r.d_[0] = -d_[0]; r.d_[1] = -d_[1]; r.d_[2] = -d_[2]; r.d_[3] = d_[3];
    const Eigen::Vector3d t = r.rotate(Eigen::Vector3d(-d_[4], -d_[5], -d_[6]));
    r.d_[4] = t.x(); r.d_[5] = t.y(); r.d_[6] = t.z();
    return r;
  }
  SE3d operator*(const SE3d& o) const {
    SE3d r;
    const double ax = d_[0], ay = d_[1], az = d_[2], aw = d_[3], bx = o.d_[0], by = o.d_[1], bz = o.d_[2], bw = o.d_[3];
    r.d_[3] = aw * bw - ax * bx - ay * by - az * bz;
    r.d_[0] = aw * bx + ax * bw + ay * bz - az * by;
    r.d_[1] = aw * by + ay * bw + az * bx - ax * bz;
    r.d_[2] = aw * bz + az * bw + ax * by - ay * bx;
    const Eigen::Vector3d t = rotate(o.translation());
    r.d_[4] = d_[4] + t.x(); r.d_[5] = d_[5] + t.y(); r.d_[6] = d_[6] + t.z();
    return r;
  }
  Eigen::Vector3d operator*(const Eigen::Vector3d& p) const {
    const Eigen::Vector3d t = rotate(p);
    return Eigen::Vector3d(t.x() + d_[4], t.y() + d_[5], t.z() + d_[6]);
  }

 private:
  // Eigen's QuaternionBase::_transformVector: uv = 2 q.vec x v ; v + w uv + q.vec x uv
  Eigen::Vector3d rotate(const Eigen::Vector3d& v) const {
    const Eigen::Vector3d qv(d_[0], d_[1], d_[2]);
    Eigen::Vector3d uv = qv.cross(v);
    uv = uv + uv;
    const Eigen::Vector3d c2 = qv.cross(uv);
    return Eigen::Vector3d(v.x() + d_[3] * uv.x() + c2.x(), v.y() + d_[3] * uv.y() + c2.y(), v.z() + d_[3] * uv.z() + c2.z());
  }
  double d_[7];
};

}  // namespace Sophus
