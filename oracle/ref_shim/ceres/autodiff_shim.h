// ORACLE — TEST INFRASTRUCTURE ONLY.  AutoDiffCostFunction / NumericDiffCostFunction of the <ceres/ceres.h> shim, split out so that the two
// stand-in Ceres surfaces share it: ref_shim/ceres/ceres.h (a RECORDING ceres::Problem, CPU pins) and ref_shim_gpu/ceres/ceres.h (include/
// lvf_ceres_compat.h's Problem, the surface the MI355X adapter walks — the compiled drop-in, oracle/ref_driver_dropin.cpp).
// Expects ceres::CostFunction / ceres::SizedCostFunction to be declared already.  AutoDiffCostFunction evaluates the functor's templated
// operator() on Jet<double, sum(Ns)> seeded with the identity — the published forward-mode algorithm.
#pragma once
#include <memory>

#include <ceres/jet.h>

namespace ceres {

namespace internal {
template <int... Ns> struct Sum;
template <> struct Sum<> { static constexpr int value = 0; };
template <int N0, int... Ns> struct Sum<N0, Ns...> { static constexpr int value = N0 + Sum<Ns...>::value; };

// calls functor(p[0], ..., p[K-1], residuals)
template <typename Functor, typename T, int K> struct Call;
#define LVF_REF_CALL(K, ...) \
  template <typename Functor, typename T> struct Call<Functor, T, K> { static bool Run(const Functor& f, T const* const* p, T* r) { return f(__VA_ARGS__, r); } };
LVF_REF_CALL(1, p[0])
LVF_REF_CALL(2, p[0], p[1])
LVF_REF_CALL(3, p[0], p[1], p[2])
LVF_REF_CALL(4, p[0], p[1], p[2], p[3])
LVF_REF_CALL(5, p[0], p[1], p[2], p[3], p[4])
LVF_REF_CALL(6, p[0], p[1], p[2], p[3], p[4], p[5])
#undef LVF_REF_CALL
}  // namespace internal

template <typename CostFunctor, int kNumResiduals, int... Ns>
class AutoDiffCostFunction : public SizedCostFunction<kNumResiduals, Ns...> {
 public:
  explicit AutoDiffCostFunction(CostFunctor* functor) : functor_(functor) {}
  const CostFunctor* functor() const { return functor_.get(); }      // shim-only (oracle/ref_driver_backend.cpp reads the recorded blocks' constants)
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    constexpr int K = sizeof...(Ns);
    constexpr int kTotal = internal::Sum<Ns...>::value;
    const int sizes[K] = {Ns...};
    if (!jacobians) return internal::Call<CostFunctor, double, K>::Run(*functor_, parameters, residuals);
    typedef Jet<double, kTotal> JetT;
    JetT x[kTotal];
    JetT out[kNumResiduals];
    const JetT* blocks[K];
    int off = 0;
    for (int b = 0; b < K; ++b) {
      blocks[b] = x + off;
      for (int j = 0; j < sizes[b]; ++j) x[off + j] = JetT(parameters[b][j], off + j);
      off += sizes[b];
    }
    if (!internal::Call<CostFunctor, JetT, K>::Run(*functor_, blocks, out)) return false;
    for (int r = 0; r < kNumResiduals; ++r) residuals[r] = out[r].a;
    off = 0;
    for (int b = 0; b < K; ++b) {
      if (jacobians[b])
        for (int r = 0; r < kNumResiduals; ++r)
          for (int j = 0; j < sizes[b]; ++j) jacobians[b][r * sizes[b] + j] = out[r].v[off + j];
      off += sizes[b];
    }
    return true;
  }

 private:
  std::unique_ptr<CostFunctor> functor_;
};

// declared so that imu_error.hpp:231-274 (ImuInitGError::Create, initialisation only — not on the hot path) compiles; never evaluated
enum NumericDiffMethodType { CENTRAL, FORWARD, RIDDERS };
template <typename CostFunctor, NumericDiffMethodType kMethod, int kNumResiduals, int... Ns>
class NumericDiffCostFunction : public SizedCostFunction<kNumResiduals, Ns...> {
 public:
  explicit NumericDiffCostFunction(CostFunctor* functor) : functor_(functor) {}
  bool Evaluate(double const* const*, double*, double**) const override { return false; }
 private:
  std::unique_ptr<CostFunctor> functor_;
};

}  // namespace ceres
