// ORACLE — TEST INFRASTRUCTURE ONLY.  Shim of <ceres/ceres.h>: CostFunction, SizedCostFunction and AutoDiffCostFunction with
// the upstream public shape (Evaluate(parameters, residuals, jacobians): row-major num_residuals x block_size Jacobians w.r.t.
// AMBIENT parameters, jacobians / jacobians[i] may be null).  AutoDiffCostFunction evaluates the functor's templated
// operator() on Jet<double, sum(Ns)> seeded with the identity — the published forward-mode algorithm.  No solver inside.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <memory>
#include <unordered_map>
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
// Local parameterisations: Plus / ComputeJacobian as the upstream classes publish them (DECLARED semantics, oracle/robust.h carries the same
// formulas): EigenQuaternionParameterization on x = [x,y,z,w]: x+ = q_delta (x) x, q_delta = [sin|d|/|d| d, cos|d|]; Identity; Product = blocks side by side.
class LocalParameterization {
 public:
  virtual ~LocalParameterization() {}
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
  virtual bool ComputeJacobian(const double* x, double* jacobian) const = 0;      // GlobalSize x LocalSize, row-major
};
class EigenQuaternionParameterization : public LocalParameterization {
 public:
  int GlobalSize() const override { return 4; }
  int LocalSize() const override { return 3; }
  bool Plus(const double* x, const double* d, double* out) const override {
    const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n > 0.0) {
      const double s = std::sin(n) / n;
      const double qd[4] = {std::cos(n), s * d[0], s * d[1], s * d[2]}, xw[4] = {x[3], x[0], x[1], x[2]};      // Hamilton product in [w,x,y,z]
      const double o0 = qd[0] * xw[0] - qd[1] * xw[1] - qd[2] * xw[2] - qd[3] * xw[3], o1 = qd[0] * xw[1] + qd[1] * xw[0] + qd[2] * xw[3] - qd[3] * xw[2],
                   o2 = qd[0] * xw[2] - qd[1] * xw[3] + qd[2] * xw[0] + qd[3] * xw[1], o3 = qd[0] * xw[3] + qd[1] * xw[2] - qd[2] * xw[1] + qd[3] * xw[0];
      out[0] = o1; out[1] = o2; out[2] = o3; out[3] = o0;
    } else { for (int k = 0; k < 4; ++k) out[k] = x[k]; }
    return true;
  }
  bool ComputeJacobian(const double* x, double* j) const override {
    j[0] = x[3]; j[1] = x[2]; j[2] = -x[1]; j[3] = -x[2]; j[4] = x[3]; j[5] = x[0]; j[6] = x[1]; j[7] = -x[0]; j[8] = x[3]; j[9] = -x[0]; j[10] = -x[1]; j[11] = -x[2];
    return true;
  }
};
class IdentityParameterization : public LocalParameterization {
 public:
  explicit IdentityParameterization(int size) : size_(size) {}
  int GlobalSize() const override { return size_; }
  int LocalSize() const override { return size_; }
  bool Plus(const double* x, const double* d, double* out) const override { for (int k = 0; k < size_; ++k) out[k] = x[k] + d[k]; return true; }
  bool ComputeJacobian(const double*, double* j) const override { for (int r = 0; r < size_; ++r) for (int c = 0; c < size_; ++c) j[r * size_ + c] = r == c ? 1.0 : 0.0; return true; }
  int size_;
};
class ProductParameterization : public LocalParameterization {
 public:
  ProductParameterization(LocalParameterization* a, LocalParameterization* b) : a_(a), b_(b) {}
  ~ProductParameterization() override { delete a_; delete b_; }
  int GlobalSize() const override { return a_->GlobalSize() + b_->GlobalSize(); }
  int LocalSize() const override { return a_->LocalSize() + b_->LocalSize(); }
  bool Plus(const double* x, const double* d, double* out) const override {
    return a_->Plus(x, d, out) && b_->Plus(x + a_->GlobalSize(), d + a_->LocalSize(), out + a_->GlobalSize());
  }
  bool ComputeJacobian(const double* x, double* j) const override {
    const int ga = a_->GlobalSize(), la = a_->LocalSize(), gb = b_->GlobalSize(), lb = b_->LocalSize(), L = la + lb;
    std::vector<double> ja((size_t)ga * la), jb((size_t)gb * lb);
    a_->ComputeJacobian(x, ja.data()); b_->ComputeJacobian(x + ga, jb.data());
    for (int i = 0; i < (ga + gb) * L; ++i) j[i] = 0.0;
    for (int r = 0; r < ga; ++r) for (int c = 0; c < la; ++c) j[r * L + c] = ja[r * la + c];
    for (int r = 0; r < gb; ++r) for (int c = 0; c < lb; ++c) j[(ga + r) * L + la + c] = jb[r * lb + c];
    return true;
  }
  LocalParameterization *a_, *b_;
};
namespace internal { struct ResidualBlock { CostFunction* cost; LossFunction* loss; std::vector<double*> params; }; }
typedef internal::ResidualBlock* ResidualBlockId;
// ceres::Problem as a RECORDER: what is added is kept in insertion order (the drivers read the blocks back); ceres::Solve below runs the
// DECLARED Levenberg-Marquardt loop (solve_shim.h) over what was recorded.
class Problem {
 public:
  struct ParameterBlock { double* values; int size; LocalParameterization* parameterization; bool constant; };
  virtual ~Problem() { for (auto* b : blocks_) delete b; }       // (cost / loss objects are leaked on purpose: shared between blocks)
  template <typename... Ts>
  ResidualBlockId AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0, Ts*... xs) {
    auto* b = new internal::ResidualBlock{cost, loss, {x0, xs...}};
    blocks_.push_back(b);
    const std::vector<int>& sizes = cost->parameter_block_sizes();
    for (size_t k = 0; k < b->params.size(); ++k) touch(b->params[k], k < sizes.size() ? sizes[k] : 0, nullptr);
    return b;
  }
  void AddParameterBlock(double* values, int size) { params_.emplace_back(values, size); touch(values, size, nullptr); }
  void AddParameterBlock(double* values, int size, LocalParameterization* lp) { params_.emplace_back(values, size); touch(values, size, lp); }
  void SetParameterBlockConstant(double* values) { pblocks_[index_.at(values)].constant = true; }
  void SetParameterBlockVariable(double* values) { pblocks_[index_.at(values)].constant = false; }
  void GetResidualBlocksForParameterBlock(const double* values, std::vector<ResidualBlockId>* out) const {
    out->clear();
    for (auto* b : blocks_) for (double* p : b->params) if (p == values) { out->push_back(b); break; }
  }
  int NumResidualBlocks() const { return (int)blocks_.size(); }
  const std::vector<ResidualBlockId>& recorded_blocks() const { return blocks_; }                    // shim-only
  const std::vector<std::pair<double*, int>>& recorded_parameter_blocks() const { return params_; }   // shim-only: one entry per AddParameterBlock CALL
  const std::vector<ParameterBlock>& parameter_blocks() const { return pblocks_; }                    // shim-only: distinct blocks, first-seen order
  int parameter_block_index(const double* values) const { auto it = index_.find(values); return it == index_.end() ? -1 : it->second; }
 private:
  void touch(double* values, int size, LocalParameterization* lp) {
    auto it = index_.find(values);
    if (it == index_.end()) { index_[values] = (int)pblocks_.size(); pblocks_.push_back(ParameterBlock{values, size, lp, false}); }
    else if (lp) pblocks_[it->second].parameterization = lp;
  }
  std::vector<ResidualBlockId> blocks_;
  std::vector<std::pair<double*, int>> params_;
  std::vector<ParameterBlock> pblocks_;
  std::unordered_map<const double*, int> index_;
};
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };
struct Solver {
  struct Options {      // the fields backend.cpp:206-209, :262-265 and mapping.cpp:159-161 set + the upstream defaults the declared loop reads
    LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY; double max_solver_time_in_seconds = 1e9; int max_num_iterations = 50; int num_threads = 1;
    double initial_trust_region_radius = 1e4, function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8, min_relative_decrease = 1e-3;
  };
  struct Summary {
    double initial_cost = 0, final_cost = 0; int num_residual_blocks = 0, num_residual_blocks_reduced = 0, num_successful_steps = 0, num_unsuccessful_steps = 0;
    int num_iterations = 0; TerminationType termination_type = NO_CONVERGENCE;
  };
};
void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary);      // solve_shim.h (defined once, in ref_driver_mapping.cpp)

}  // namespace ceres
