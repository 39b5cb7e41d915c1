// cr_math.hpp — atan2 for float arguments, CORRECTLY ROUNDED to float, with one definition for host and device.
//
// The reference's range-image code takes atan2 of floats through libm (src/lvio_fusion/src/projection.cpp:44-47,73,79,127,275;
// association.cpp:122) and feeds the result into integer decisions (pixel row / column, ground flag, segment links, the half-scan
// latch).  Which libm — and therefore which last bit — is unspecified; the only version-independent reading is "the float nearest to
// the exact value".  That value is produced here in fp64 from IEEE +, -, *, / only (no FMA contraction, no libm): argument reduction to
// [0, 1], nearest eighth c, atan(t) = atan(c) + atan((t - c) / (1 + t c)) with |z| <= 1/16 and an 8-term odd series (truncation < 1e-20),
// total error ~2e-16 relative, then ONE rounding to float.  Because the operation sequence is fixed, the device (gfx950) and any host
// produce the same bits — the per-pixel decisions of lvf_lidar_extract are reproducible bit for bit (tests/test_gpu_extract.py).
#pragma once
#include <hip/hip_runtime.h>

namespace lvf {

__host__ __device__ inline double cr_atan_unit(double t) {          // atan(t), 0 <= t <= 1
#pragma clang fp contract(off)
  const double kAtanEighth[9] = {0.0, 0.12435499454676144, 0.24497866312686414, 0.35877067027057225, 0.4636476090008061,
                                 0.5585993153435624, 0.6435011087932844, 0.7188299996216245, 0.7853981633974483};
  const int k = (int)(t * 8.0 + 0.5);
  const double c = (double)k * 0.125;
  const double z = (t - c) / (1.0 + t * c);
  const double w = z * z;
  double s = 0.058823529411764705;                                  // 1/17
  s = 0.06666666666666667 - w * s;                                  // 1/15
  s = 0.07692307692307693 - w * s;                                  // 1/13
  s = 0.09090909090909091 - w * s;                                  // 1/11
  s = 0.1111111111111111 - w * s;                                   // 1/9
  s = 0.14285714285714285 - w * s;                                  // 1/7
  s = 0.2 - w * s;                                                  // 1/5
  s = 0.3333333333333333 - w * s;                                   // 1/3
  s = 1.0 - w * s;
  return kAtanEighth[k] + z * s;
}

__host__ __device__ inline float cr_atan2f(float yf, float xf) {
#pragma clang fp contract(off)
  const double x = (double)xf, y = (double)yf;
  if (x != x || y != y) return xf + yf;                             // NaN in, NaN out
  const double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
  double a;
  if (ax == 0.0 && ay == 0.0) a = 0.0;
  else if (ay <= ax) a = cr_atan_unit(ay / ax);
  else a = 1.5707963267948966 - cr_atan_unit(ax / ay);
  if (x < 0.0 || (x == 0.0 && 1.0 / x < 0.0)) a = 3.141592653589793 - a;
  const float r = (float)a;
  return (y < 0.0 || (y == 0.0 && 1.0 / y < 0.0)) ? -r : r;
}

}  // namespace lvf
