// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/jet.h header).
//
// cr_math.h — atan2 of float arguments CORRECTLY ROUNDED to float: the version-independent reading of the reference's libm calls on
// floats in the range-image code (src/lvio_fusion/src/projection.cpp:44-47,73,79,127,275; association.cpp:122), whose results feed
// integer decisions (pixel row / column, ground flag, segment links, half-scan latch).  Evaluated in fp64 from IEEE +, -, *, / only
// (oracle/Makefile compiles with -ffp-contract=off): reduction to [0, 1], nearest eighth c, atan(t) = atan(c) + atan((t - c)/(1 + t c)),
// 8-term odd series (|z| <= 1/16), one rounding to float.  tests/test_oracle_cloud.py checks it against mpmath (50 digits) including the
// float rounding; the device carries the same operation sequence so the GPU extraction can be compared bit for bit.
#pragma once

namespace lvo {

inline double cr_atan_unit(double t) {   // atan(t), 0 <= t <= 1
  static const double kAtanEighth[9] = {0.0, 0.12435499454676144, 0.24497866312686414, 0.35877067027057225, 0.4636476090008061,
                                        0.5585993153435624, 0.6435011087932844, 0.7188299996216245, 0.7853981633974483};
  const int k = (int)(t * 8.0 + 0.5);
  const double c = (double)k * 0.125;
  const double z = (t - c) / (1.0 + t * c);
  const double w = z * z;
  double s = 0.058823529411764705;
  s = 0.06666666666666667 - w * s;
  s = 0.07692307692307693 - w * s;
  s = 0.09090909090909091 - w * s;
  s = 0.1111111111111111 - w * s;
  s = 0.14285714285714285 - w * s;
  s = 0.2 - w * s;
  s = 0.3333333333333333 - w * s;
  s = 1.0 - w * s;
  return kAtanEighth[k] + z * s;
}

inline float cr_atan2f(float yf, float xf) {
  const double x = (double)xf, y = (double)yf;
  if (x != x || y != y) return xf + yf;
  const double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
  double a;
  if (ax == 0.0 && ay == 0.0) a = 0.0;
  else if (ay <= ax) a = cr_atan_unit(ay / ax);
  else a = 1.5707963267948966 - cr_atan_unit(ax / ay);
  if (x < 0.0 || (x == 0.0 && 1.0 / x < 0.0)) a = 3.141592653589793 - a;
  const float r = (float)a;
  return (y < 0.0 || (y == 0.0 && 1.0 / y < 0.0)) ? -r : r;
}

}  // namespace lvo
