// ORACLE — TEST INFRASTRUCTURE ONLY.  Shim of <ceres/ceres.h>: CostFunction, SizedCostFunction and AutoDiffCostFunction with
// the upstream public shape (Evaluate(parameters, residuals, jacobians): row-major num_residuals x block_size Jacobians w.r.t.
// AMBIENT parameters, jacobians / jacobians[i] may be null).  AutoDiffCostFunction evaluates the functor's templated
// operator() on Jet<double, sum(Ns)> seeded with the identity — the published forward-mode algorithm.  No solver inside.
#pragma once
#include <memory>
#include <vector>

#include "jet.h"

namespace ceres {

class CostFunction {
 public:
  CostFunction() : num_residuals_(0) {}
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  const std::vector<int>& parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }

 protected:
  std::vector<int>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }

 private:
  std::vector<int> parameter_block_sizes_;
  int num_residuals_;
};

template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() {
    set_num_residuals(kNumResiduals);
    *mutable_parameter_block_sizes() = std::vector<int>{Ns...};
  }
};

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
