// ORACLE — TEST INFRASTRUCTURE ONLY.  Shim of the slice of upstream Ceres Solver's public jet.h that the reference's cost
// functor headers need, so that /root/reference/src/lvio_fusion/include/lvio_fusion/ceres/{base,visual_error,lidar_error,
// pose_error}.hpp compile UNMODIFIED into oracle/_ref/liblvf_ref.so (recipe: oracle/Makefile, target `ref`).
// Written from the published definition of ceres::Jet<T, N> (value + N partials, first-order chain rule per operator);
// no Ceres source is present in this image.  Operator formulas follow upstream's operand order so that the double results
// agree to the last bit with a real Ceres build compiled without FMA contraction.
#pragma once
#include <cmath>

namespace ceres {

template <typename T, int N>
struct Jet {
  T a;
  T v[N];
  Jet() : a() { for (int i = 0; i < N; ++i) v[i] = T(); }
  Jet(const T& value) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(); }   // NOLINT (implicit, as upstream)
  Jet(const T& value, int k) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(); v[k] = T(1); }
  Jet& operator+=(const Jet& y) { *this = *this + y; return *this; }
  Jet& operator-=(const Jet& y) { *this = *this - y; return *this; }
  Jet& operator*=(const Jet& y) { *this = *this * y; return *this; }
  Jet& operator/=(const Jet& y) { *this = *this / y; return *this; }
};

#define LVF_REF_JET template <typename T, int N> inline Jet<T, N>
LVF_REF_JET operator+(const Jet<T, N>& f) { return f; }
LVF_REF_JET operator-(const Jet<T, N>& f) { Jet<T, N> r; r.a = -f.a; for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r; }
LVF_REF_JET operator+(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> r; r.a = f.a + g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i]; return r; }
LVF_REF_JET operator+(const Jet<T, N>& f, T s) { Jet<T, N> r = f; r.a = f.a + s; return r; }
LVF_REF_JET operator+(T s, const Jet<T, N>& f) { Jet<T, N> r = f; r.a = f.a + s; return r; }
LVF_REF_JET operator-(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> r; r.a = f.a - g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i]; return r; }
LVF_REF_JET operator-(const Jet<T, N>& f, T s) { Jet<T, N> r = f; r.a = f.a - s; return r; }
LVF_REF_JET operator-(T s, const Jet<T, N>& f) { Jet<T, N> r; r.a = s - f.a; for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r; }
// (f g)' = f g' + f' g
LVF_REF_JET operator*(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> r; r.a = f.a * g.a; for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
LVF_REF_JET operator*(const Jet<T, N>& f, T s) { Jet<T, N> r; r.a = f.a * s; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s; return r; }
LVF_REF_JET operator*(T s, const Jet<T, N>& f) { Jet<T, N> r; r.a = f.a * s; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s; return r; }
// (f / g)' = (f' - (f/g) g') / g, evaluated with one reciprocal as upstream does
LVF_REF_JET operator/(const Jet<T, N>& f, const Jet<T, N>& g) {
  Jet<T, N> r; const T g_a_inverse = T(1.0) / g.a; const T f_a_by_g_a = f.a * g_a_inverse; r.a = f_a_by_g_a;
  for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
  return r;
}
LVF_REF_JET operator/(T s, const Jet<T, N>& g) {
  Jet<T, N> r; const T minus_s_g_a_inverse2 = -s / (g.a * g.a); r.a = s / g.a;
  for (int i = 0; i < N; ++i) r.v[i] = g.v[i] * minus_s_g_a_inverse2;
  return r;
}
LVF_REF_JET operator/(const Jet<T, N>& f, T s) { Jet<T, N> r; const T s_inverse = T(1.0) / s; r.a = f.a * s_inverse; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s_inverse; return r; }

template <typename T, int N> inline bool operator<(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a < g.a; }
template <typename T, int N> inline bool operator>(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a > g.a; }

// scalar overloads so that templated functors can call sqrt / sin / ... unqualified from inside namespace ceres
using std::sqrt; using std::sin; using std::cos; using std::asin; using std::acos; using std::atan2; using std::abs;

LVF_REF_JET sqrt(const Jet<T, N>& f) { Jet<T, N> r; const T tmp = std::sqrt(f.a); const T two_a_inverse = T(1.0) / (T(2.0) * tmp); r.a = tmp; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * two_a_inverse; return r; }
LVF_REF_JET sin(const Jet<T, N>& f) { Jet<T, N> r; r.a = std::sin(f.a); const T c = std::cos(f.a); for (int i = 0; i < N; ++i) r.v[i] = c * f.v[i]; return r; }
LVF_REF_JET cos(const Jet<T, N>& f) { Jet<T, N> r; r.a = std::cos(f.a); const T ms = -std::sin(f.a); for (int i = 0; i < N; ++i) r.v[i] = ms * f.v[i]; return r; }
LVF_REF_JET asin(const Jet<T, N>& f) { Jet<T, N> r; r.a = std::asin(f.a); const T tmp = T(1.0) / std::sqrt(T(1.0) - f.a * f.a); for (int i = 0; i < N; ++i) r.v[i] = tmp * f.v[i]; return r; }
LVF_REF_JET acos(const Jet<T, N>& f) { Jet<T, N> r; r.a = std::acos(f.a); const T tmp = -T(1.0) / std::sqrt(T(1.0) - f.a * f.a); for (int i = 0; i < N; ++i) r.v[i] = tmp * f.v[i]; return r; }
// atan2(g, f): d = (f dg - g df) / (f^2 + g^2)
LVF_REF_JET atan2(const Jet<T, N>& g, const Jet<T, N>& f) {
  Jet<T, N> r; const T tmp = T(1.0) / (f.a * f.a + g.a * g.a); r.a = std::atan2(g.a, f.a);
  for (int i = 0; i < N; ++i) r.v[i] = tmp * (-g.a * f.v[i] + f.a * g.v[i]);
  return r;
}
#undef LVF_REF_JET

}  // namespace ceres
