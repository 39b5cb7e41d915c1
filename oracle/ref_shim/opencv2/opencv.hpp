// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <opencv2/opencv.hpp>: cv::Mat / cv::Mat_<T> with the comma initialiser
// Camera's constructor uses for K and D (visual/camera.h:84-85), and the point / key-point value types utility.h, frame.h and
// visual/feature.h name.  Values are stored, never read by the factor path.
#pragma once
#include <bitset>
#include <vector>
typedef unsigned char uchar;
namespace cv {
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Point3f { float x, y, z; Point3f() : x(0), y(0), z(0) {} Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {} };
struct KeyPoint { Point2f pt; float size; KeyPoint() : size(0) {} KeyPoint(Point2f p, float s) : pt(p), size(s) {} };
class Mat {
 public:
  Mat() : rows(0), cols(0) {}
  int rows, cols;
  std::vector<double> v;
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
}  // namespace cv
