// ORACLE — TEST INFRASTRUCTURE ONLY.  Shim of <ceres/ceres.h>: CostFunction, SizedCostFunction and AutoDiffCostFunction with
// the upstream public shape (Evaluate(parameters, residuals, jacobians): row-major num_residuals x block_size Jacobians w.r.t.
// AMBIENT parameters, jacobians / jacobians[i] may be null).  AutoDiffCostFunction evaluates the functor's templated
// operator() on Jet<double, sum(Ns)> seeded with the identity — the published forward-mode algorithm.  No solver inside.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <memory>
#include <utility>
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

}  // namespace ceres
#include "autodiff_shim.h"
namespace ceres {

// ---- the problem-building surface adapt/problem.h:34-88 and association.cpp:270-384 name: loss functions, local parameterisations (type
// names only), and a ceres::Problem that RECORDS what is added (no solver): the driver reads the blocks back and evaluates them.
class LossFunction { public: virtual ~LossFunction() {} virtual void Evaluate(double s, double out[3]) const = 0; };
class TrivialLoss : public LossFunction { public: void Evaluate(double s, double out[3]) const override { out[0] = s; out[1] = 1.0; out[2] = 0.0; } };
class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {      // upstream loss_function.cc
    if (s > b_) { const double r = std::sqrt(s); rho[0] = 2.0 * a_ * r - b_; rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r); rho[2] = -rho[1] / (2.0 * s); }
    else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
  }
  double a() const { return a_; }
 private:
  double a_, b_;
};
class LocalParameterization { public: virtual ~LocalParameterization() {} };
// (type names only: backend.cpp:99-101 builds ProductParameterization(EigenQuaternionParameterization, IdentityParameterization(3)); the shim's
// Problem records the pointer it is handed and nothing else)
class EigenQuaternionParameterization : public LocalParameterization {};
class IdentityParameterization : public LocalParameterization { public: explicit IdentityParameterization(int size) : size_(size) {} int size_; };
class ProductParameterization : public LocalParameterization {
 public:
  ProductParameterization(LocalParameterization* a, LocalParameterization* b) : a_(a), b_(b) {}
  ~ProductParameterization() override { delete a_; delete b_; }
  LocalParameterization *a_, *b_;
};
namespace internal { struct ResidualBlock { CostFunction* cost; LossFunction* loss; std::vector<double*> params; }; }
typedef internal::ResidualBlock* ResidualBlockId;
class Problem {
 public:
  virtual ~Problem() { for (auto* b : blocks_) delete b; }       // (cost / loss objects are leaked on purpose: shared between blocks)
  template <typename... Ts>
  ResidualBlockId AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0, Ts*... xs) {
    auto* b = new internal::ResidualBlock{cost, loss, {x0, xs...}};
    blocks_.push_back(b);
    return b;
  }
  void AddParameterBlock(double* values, int size) { params_.emplace_back(values, size); }
  void AddParameterBlock(double* values, int size, LocalParameterization*) { params_.emplace_back(values, size); }
  void GetResidualBlocksForParameterBlock(const double* values, std::vector<ResidualBlockId>* out) const {
    out->clear();
    for (auto* b : blocks_) for (double* p : b->params) if (p == values) { out->push_back(b); break; }
  }
  int NumResidualBlocks() const { return (int)blocks_.size(); }
  const std::vector<ResidualBlockId>& recorded_blocks() const { return blocks_; }                    // shim-only
  const std::vector<std::pair<double*, int>>& recorded_parameter_blocks() const { return params_; }   // shim-only
 private:
  std::vector<ResidualBlockId> blocks_;
  std::vector<std::pair<double*, int>> params_;
};
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
struct Solver {
  struct Options {      // the fields backend.cpp:206-209, :262-265 and mapping.cpp:159-161 set (values recorded, no solver behind them)
    LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY; double max_solver_time_in_seconds = 1e9; int max_num_iterations = 50; int num_threads = 1;
  };
  struct Summary { double final_cost = 0; int num_residual_blocks_reduced = 0; };
};
inline void Solve(const Solver::Options&, Problem*, Solver::Summary*) {}      // no solver in the shim (declared semantics: oracle/lm.h, oracle/icp.h)

}  // namespace ceres
