// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <opencv2/opencv.hpp>: cv::Mat / cv::Mat_<T> with the comma initialiser
// Camera's constructor uses for K and D (visual/camera.h:84-85), and the point / key-point value types utility.h, frame.h and
// visual/feature.h name.  Values are stored, never read by the factor path.
#pragma once
// the C-compatibility headers the real OpenCV / PCL / Sophus headers pull in: with them libstdc++ puts the float overloads of atan2 /
// sqrt / abs / round into the GLOBAL namespace, which is what the reference's unqualified calls on float arguments resolve to
// (projection.cpp:44,73,79,127,130,275; association.cpp:122) — declared, like the Ceres version, because the reference pins nothing
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <bitset>
#include <cstring>
#include <list>
#include <memory>
#include <string>
#include <vector>
typedef unsigned char uchar;
#define CV_8S 1
#define CV_32S 4
#define CV_32F 5
namespace cv {
struct Scalar { double v[4]; static Scalar all(double x) { Scalar s; s.v[0] = s.v[1] = s.v[2] = s.v[3] = x; return s; } };
struct Point2i { int x, y; Point2i() : x(0), y(0) {} Point2i(int x_, int y_) : x(x_), y(y_) {} };
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Point3f { float x, y, z; Point3f() : x(0), y(0), z(0) {} Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {} };
struct KeyPoint { Point2f pt; float size; KeyPoint() : size(0) {} KeyPoint(Point2f p, float s) : pt(p), size(s) {} };
class Mat {
 public:
  Mat() : rows(0), cols(0), type_(0) {}
  // typed dense matrix filled with a scalar: the range / label / ground images of lidar/projection.h:63-65 (element access only)
  Mat(int r, int c, int type, const Scalar& s) : rows(r), cols(c), type_(type) {
    const size_t e = type == CV_8S ? 1 : 4;
    bytes.assign((size_t)r * c * e, 0);
    for (size_t i = 0; i < (size_t)r * c; ++i) {
      if (type == CV_32F) { const float f = (float)s.v[0]; std::memcpy(&bytes[4 * i], &f, 4); }
      else if (type == CV_32S) { const int k = (int)s.v[0]; std::memcpy(&bytes[4 * i], &k, 4); }
      else bytes[i] = (unsigned char)(signed char)s.v[0];
    }
  }
  template <typename T> T& at(int i, int j) { return *reinterpret_cast<T*>(&bytes[((size_t)i * cols + j) * sizeof(T)]); }
  template <typename T> const T& at(int i, int j) const { return *reinterpret_cast<const T*>(&bytes[((size_t)i * cols + j) * sizeof(T)]); }
  int rows, cols;
  std::vector<double> v;
  std::vector<unsigned char> bytes;
  int type_;
};
template <typename T> class Mat_;
template <typename T>
class MatCommaInitializer_ {
 public:
  MatCommaInitializer_(Mat_<T>* m, T first) : m_(m), k_(0) { put(first); }
  MatCommaInitializer_& operator,(T x) { put(x); return *this; }
  operator Mat() const;
 private:
  void put(T x);
  Mat_<T>* m_; size_t k_;
};
template <typename T>
class Mat_ : public Mat {
 public:
  Mat_(int r, int c) { rows = r; cols = c; v.assign((size_t)r * c, 0.0); }
  MatCommaInitializer_<T> operator<<(T first) { return MatCommaInitializer_<T>(this, first); }
};
template <typename T> void MatCommaInitializer_<T>::put(T x) { if (k_ < m_->v.size()) m_->v[k_] = (double)x; ++k_; }
template <typename T> MatCommaInitializer_<T>::operator Mat() const { return *m_; }
// names the front-end headers mention (frontend.h -> visual/local_map.h, visual/extractor.h) so that src/backend.cpp compiles as a translation
// unit of its own (oracle/ref_driver_backend.cpp); nothing of the front end is ever constructed
template <typename T> struct Ptr : std::shared_ptr<T> { Ptr() {} Ptr(const std::shared_ptr<T>& p) : std::shared_ptr<T>(p) {} };
class DescriptorMatcher { public: static Ptr<DescriptorMatcher> create(const std::string&) { return Ptr<DescriptorMatcher>(); } };
}  // namespace cv
