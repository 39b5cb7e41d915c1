// ORACLE — TEST INFRASTRUCTURE ONLY.  Self-test of the Eigen stand-in (oracle/ref_shim/Eigen/Core) that lets the reference's IMU text
// compile unmodified: every operation imu_error.hpp / preintegration.cpp / utility.h use is checked here against values computed
// independently (plain loops in this file, identities that must hold), so that "oracle/imu.h == reference text" cannot rest on a
// stand-in that is wrong in the same way on both sides.  Prints one JSON line; tests/test_oracle_ref.py runs it and compares the
// dumped matrices with numpy as well.
#include <cmath>
#include <cstdio>
#include <vector>

#include <Eigen/Core>

using namespace Eigen;

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { ++fails; std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } } while (0)

static double rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffffff) / 16777216.0 - 0.5; }

int main() {
  unsigned seed = 12345;
  // comma initialiser fills row by row; block views alias the parent; transpose; storage orders
  Matrix3d A;
  A << 1, 2, 3, 4, 5, 6, 7, 8, 10;
  CHECK(A(0, 1) == 2 && A(1, 0) == 4 && A(2, 2) == 10);
  CHECK(A.data()[1] == 4);                                         // column-major storage
  Matrix<double, 2, 3, RowMajor> Rm;
  Rm << 1, 2, 3, 4, 5, 6;
  CHECK(Rm.data()[1] == 2 && Rm(1, 0) == 4);                       // row-major storage
  double raw[6] = {1, 2, 3, 4, 5, 6};
  Map<Matrix<double, 2, 3, RowMajor>> M(raw);
  CHECK(M(1, 2) == 6);
  M.block<1, 2>(0, 1) = Matrix<double, 1, 2>(9.0, 8.0);
  CHECK(raw[1] == 9 && raw[2] == 8);
  Matrix3d At = A.transpose();
  CHECK(At(0, 1) == 4 && At(2, 0) == 3);
  MatrixXd F = MatrixXd::Zero(15, 15);
  F.block<3, 3>(3, 6) = A;
  CHECK(F(4, 7) == 5 && F(0, 0) == 0 && F.rows() == 15);
  F.block<3, 3>(0, 0) = F.block<3, 3>(3, 6);                       // block = block
  CHECK(F(2, 2) == 10);
  Matrix3d Bm = F.block<3, 3>(3, 6);
  CHECK(Bm == A);
  Matrix<double, 4, 4> Q4 = Matrix<double, 4, 4>::Identity();
  Q4.bottomRightCorner<3, 3>() = A;
  CHECK(Q4(3, 3) == 10 && Q4(0, 0) == 1 && Q4(0, 1) == 0);
  // element-wise operators in parse order; scalar on either side; unary minus; int scalar
  Matrix3d C = -0.25 * A * 2.0 + A - A * 0.5;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) CHECK(C(i, j) == ((-0.25 * A(i, j)) * 2.0 + A(i, j)) - A(i, j) * 0.5);
  Vector3d v(1, -2, 3);
  Vector3d w = 2 * v;
  CHECK(w(0) == 2 && w(1) == -4 && w(2) == 6);
  CHECK((-v)(1) == 2);
  // matrix product: k ascending, dynamic x fixed, aliasing (A = B * A)
  Matrix<double, 15, 15> J = Matrix<double, 15, 15>::Identity(), P = Matrix<double, 15, 15>::Zero();
  MatrixXd G = MatrixXd::Zero(15, 15);
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { G(i, j) = rnd(seed); J(i, j) += 0.1 * rnd(seed); P(i, j) = rnd(seed); }
  Matrix<double, 15, 15> J0 = J, ref;
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { double s = G(i, 0) * J0(0, j); for (int k = 1; k < 15; ++k) s += G(i, k) * J0(k, j); ref(i, j) = s; }
  J = G * J;
  CHECK(J == ref);
  Matrix<double, 15, 15> GPGt = G * P * G.transpose();
  for (int i = 0; i < 15; i += 7) for (int j = 0; j < 15; j += 5) {
    double s = 0;
    for (int k = 0; k < 15; ++k) { double t = 0; for (int m = 0; m < 15; ++m) t += G(i, m) * P(m, k); s += t * G(j, k); }
    CHECK(std::fabs(GPGt(i, j) - s) <= 1e-13 * (1 + std::fabs(s)));
  }
  // inverse (partial-pivot LU) and LLT on an SPD 15 x 15: A A^-1 = I, L L^T = A, matrixL() lower
  Matrix<double, 15, 15> S = G * G.transpose();
  for (int i = 0; i < 15; ++i) S(i, i) += 1.0;
  Matrix<double, 15, 15> Si = S.inverse(), I1 = S * Si;
  double e_inv = 0, e_llt = 0, e_up = 0;
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) e_inv = std::fmax(e_inv, std::fabs(I1(i, j) - (i == j)));
  LLT<Matrix<double, 15, 15>> llt(S);
  Matrix<double, 15, 15> L = llt.matrixL(), LLt = L * L.transpose();
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { e_llt = std::fmax(e_llt, std::fabs(LLt(i, j) - S(i, j))); if (j > i) e_up = std::fmax(e_up, std::fabs(L(i, j))); }
  CHECK(e_inv < 1e-12 && e_llt < 1e-12 && e_up == 0.0);
  // a matrix that NEEDS pivoting
  Matrix3d Pv;
  Pv << 0, 2, 1, 1, 1, 0, 3, 0, 1;
  Matrix3d PvI = Pv.inverse(), I3 = Pv * PvI;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) CHECK(std::fabs(I3(i, j) - (i == j)) < 1e-14);
  // quaternions: (w, x, y, z) constructor, storage x y z w, Hamilton product, rotation = matrix action, inverse, normalisation
  Quaterniond q(0.9, 0.1, -0.3, 0.2), p(0.7, -0.2, 0.4, 0.5);
  CHECK(q.w() == 0.9 && q.x() == 0.1 && q.vec()(2) == 0.2 && q.coeffs()(3) == 0.9);
  Quaterniond qp = q * p;
  CHECK(std::fabs(qp.w() - (0.9 * 0.7 - 0.1 * -0.2 - -0.3 * 0.4 - 0.2 * 0.5)) < 1e-16);
  CHECK(std::fabs(qp.x() - (0.9 * -0.2 + 0.1 * 0.7 + -0.3 * 0.5 - 0.2 * 0.4)) < 1e-16);
  Quaterniond qn = q.normalized();
  CHECK(std::fabs(qn.norm() - 1.0) < 1e-15);
  Matrix3d R = qn.toRotationMatrix(), RRt = R * R.transpose();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) CHECK(std::fabs(RRt(i, j) - (i == j)) < 1e-15);
  Vector3d rv = qn * v, rm = R * v;
  for (int i = 0; i < 3; ++i) CHECK(std::fabs(rv(i) - rm(i)) < 1e-15);
  Quaterniond qi = q.inverse(), one = q * qi;                      // non-unit: inverse = conjugate / squared norm
  CHECK(std::fabs(one.w() - 1.0) < 1e-15 && std::fabs(one.x()) < 1e-16 && std::fabs(one.y()) < 1e-16 && std::fabs(one.z()) < 1e-16);
  Quaterniond fromR(R);                                            // rotation matrix -> quaternion (ImuInitGError's constructor)
  CHECK(std::fabs(std::fabs(fromR.w() * qn.w() + fromR.x() * qn.x() + fromR.y() * qn.y() + fromR.z() * qn.z()) - 1.0) < 1e-15);
  Vector3d cr = v.cross(Vector3d(0.5, 0.25, -1));
  CHECK(cr(0) == -2 * -1 - 3 * 0.25 && cr(1) == 3 * 0.5 - 1 * -1 && cr(2) == 1 * 0.25 - -2 * 0.5);
  Vector3d nv = v; nv.normalize();
  CHECK(std::fabs(nv.norm() - 1.0) < 1e-15 && nv(0) == 1.0 / std::sqrt((1.0 + 4.0) + 9.0));
  // dump S, S^-1 and L for the numpy cross-check
  std::printf("{\"fails\": %d, \"S\": [", fails);
  for (int i = 0; i < 225; ++i) std::printf("%s%.17g", i ? ", " : "", S(i / 15, i % 15));
  std::printf("], \"Sinv\": [");
  for (int i = 0; i < 225; ++i) std::printf("%s%.17g", i ? ", " : "", Si(i / 15, i % 15));
  std::printf("], \"L\": [");
  for (int i = 0; i < 225; ++i) std::printf("%s%.17g", i ? ", " : "", L(i / 15, i % 15));
  std::printf("]}\n");
  return fails ? 1 : 0;
}
