// lvf_ceres_compat.h — the slice of the Ceres Solver PUBLIC API that lvio_fusion's hot path touches, declared
// locally so the adapter (lvf_ceres_adapter.hpp) and its self-test compile in an image that has no Ceres.
//
// In a real catkin workspace this header is NOT used: lvf_ceres_adapter.hpp includes <ceres/ceres.h> when it is
// available (__has_include) and the types below are Ceres' own.  What is declared here follows the upstream
// public interface shape (Ceres 1.14 – 2.1, the range the reference's use of LocalParameterization /
// ProductParameterization implies: src/lvio_fusion/src/backend.cpp:98-101, include/lvio_fusion/adapt/problem.h):
// same class names, method names, argument order and ownership rules, so code written against it is source-
// compatible with the real library.  There is deliberately NO ceres::Solve here: this repository has no CPU solver
// (no fallback) — the only Solve is lvio_fusion::gpu::Solve in the adapter, which runs on the MI355X.
#ifndef LVF_CERES_COMPAT_H_
#define LVF_CERES_COMPAT_H_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

namespace ceres {

typedef int int32;

class CostFunction {
 public:
  CostFunction() : num_residuals_(0) {}
  virtual ~CostFunction() {}
  // parameters[i] -> parameter block i; residuals non-null; jacobians may be null, jacobians[i] may be null;
  // jacobians[i] is row-major num_residuals x parameter_block_sizes()[i].  (Ceres contract.)
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  const std::vector<int32>& parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }

 protected:
  std::vector<int32>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }

 private:
  std::vector<int32> parameter_block_sizes_;
  int num_residuals_;
};

template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() {
    set_num_residuals(kNumResiduals);
    *mutable_parameter_block_sizes() = std::vector<int32>{Ns...};
  }
  virtual ~SizedCostFunction() {}
};

class LossFunction {
 public:
  virtual ~LossFunction() {}
  // out[0] = rho(s), out[1] = rho'(s), out[2] = rho''(s)
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class TrivialLoss : public LossFunction {
 public:
  void Evaluate(double s, double rho[3]) const override { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
};
class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_;
      rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
  }

 private:
  const double a_, b_;
};

class LocalParameterization {
 public:
  virtual ~LocalParameterization() {}
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
  virtual bool ComputeJacobian(const double* x, double* jacobian) const = 0;   // GlobalSize x LocalSize, row-major
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
};
class IdentityParameterization : public LocalParameterization {
 public:
  explicit IdentityParameterization(int size) : size_(size) {}
  bool Plus(const double* x, const double* d, double* o) const override { for (int i = 0; i < size_; ++i) o[i] = x[i] + d[i]; return true; }
  bool ComputeJacobian(const double*, double* j) const override {
    for (int i = 0; i < size_; ++i) for (int k = 0; k < size_; ++k) j[i * size_ + k] = (i == k) ? 1.0 : 0.0;
    return true;
  }
  int GlobalSize() const override { return size_; }
  int LocalSize() const override { return size_; }

 private:
  int size_;
};
// x = [x,y,z,w] (Eigen storage order); Plus(x, d) = q_delta(d) (x) x
class EigenQuaternionParameterization : public LocalParameterization {
 public:
  bool Plus(const double* x, const double* d, double* o) const override {
    const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n > 0.0) {
      const double s = std::sin(n) / n, c = std::cos(n);
      const double dx = s * d[0], dy = s * d[1], dz = s * d[2];
      o[3] = c * x[3] - dx * x[0] - dy * x[1] - dz * x[2];
      o[0] = c * x[0] + dx * x[3] + dy * x[2] - dz * x[1];
      o[1] = c * x[1] - dx * x[2] + dy * x[3] + dz * x[0];
      o[2] = c * x[2] + dx * x[1] - dy * x[0] + dz * x[3];
    } else { for (int i = 0; i < 4; ++i) o[i] = x[i]; }
    return true;
  }
  bool ComputeJacobian(const double* x, double* j) const override {
    j[0] = x[3];  j[1] = x[2];   j[2] = -x[1];
    j[3] = -x[2]; j[4] = x[3];   j[5] = x[0];
    j[6] = x[1];  j[7] = -x[0];  j[8] = x[3];
    j[9] = -x[0]; j[10] = -x[1]; j[11] = -x[2];
    return true;
  }
  int GlobalSize() const override { return 4; }
  int LocalSize() const override { return 3; }
};
class ProductParameterization : public LocalParameterization {
 public:
  ProductParameterization(LocalParameterization* a, LocalParameterization* b) : a_(a), b_(b) {}
  bool Plus(const double* x, const double* d, double* o) const override {
    return a_->Plus(x, d, o) && b_->Plus(x + a_->GlobalSize(), d + a_->LocalSize(), o + a_->GlobalSize());
  }
  bool ComputeJacobian(const double* x, double* j) const override {
    const int G = GlobalSize(), L = LocalSize();
    std::fill(j, j + G * L, 0.0);
    std::vector<double> ja(a_->GlobalSize() * a_->LocalSize()), jb(b_->GlobalSize() * b_->LocalSize());
    if (!a_->ComputeJacobian(x, ja.data()) || !b_->ComputeJacobian(x + a_->GlobalSize(), jb.data())) return false;
    for (int r = 0; r < a_->GlobalSize(); ++r) for (int c = 0; c < a_->LocalSize(); ++c) j[r * L + c] = ja[r * a_->LocalSize() + c];
    for (int r = 0; r < b_->GlobalSize(); ++r)
      for (int c = 0; c < b_->LocalSize(); ++c) j[(a_->GlobalSize() + r) * L + a_->LocalSize() + c] = jb[r * b_->LocalSize() + c];
    return true;
  }
  int GlobalSize() const override { return a_->GlobalSize() + b_->GlobalSize(); }
  int LocalSize() const override { return a_->LocalSize() + b_->LocalSize(); }

 private:
  std::unique_ptr<LocalParameterization> a_, b_;
};

namespace internal {
struct ResidualBlock {
  CostFunction* cost_function;
  LossFunction* loss_function;
  std::vector<double*> parameter_blocks;
  int index;
};
}  // namespace internal
typedef internal::ResidualBlock* ResidualBlockId;

// upstream ceres/crs_matrix.h: compressed-row storage; row i holds values[rows[i] .. rows[i+1]) at columns cols[..]
struct CRSMatrix {
  CRSMatrix() : num_rows(0), num_cols(0) {}
  int num_rows, num_cols;
  std::vector<int> cols, rows;
  std::vector<double> values;
};

enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };

// Bookkeeping only — exactly the observable behaviour the reference relies on: pointer identity defines a
// parameter block, the problem owns cost/loss/parameterisation objects and frees shared ones once
// (default Problem::Options), residual blocks keep insertion order.
class Problem {
 public:
  Problem() {}
  Problem(const Problem&) = delete;
  Problem& operator=(const Problem&) = delete;
  virtual ~Problem() {
    // the Problem owns its cost / loss / parameterisation objects and deletes each ONCE (shared ones: the reference shares one HuberLoss and one
    // ProductParameterization across a problem's blocks, backend.cpp:98-101).  De-duplicated by sort + unique: a std::set of 91 k cost
    // functions was 26 ms of every tick of the stand-in (bench ceres_surface_tick_ms)
    std::vector<CostFunction*> cf; std::vector<LossFunction*> lf; std::vector<LocalParameterization*> lp;
    cf.reserve(residual_blocks_.size());
    for (auto& rb : residual_blocks_) { cf.push_back(rb->cost_function); if (rb->loss_function && (lf.empty() || lf.back() != rb->loss_function)) lf.push_back(rb->loss_function); }
    for (auto& kv : blocks_) if (kv.second.parameterization && (lp.empty() || lp.back() != kv.second.parameterization)) lp.push_back(kv.second.parameterization);
    std::sort(cf.begin(), cf.end()); cf.erase(std::unique(cf.begin(), cf.end()), cf.end());
    std::sort(lf.begin(), lf.end()); lf.erase(std::unique(lf.begin(), lf.end()), lf.end());
    std::sort(lp.begin(), lp.end()); lp.erase(std::unique(lp.begin(), lp.end()), lp.end());
    for (auto* p : cf) delete p;
    for (auto* p : lf) delete p;
    for (auto* p : lp) delete p;
  }

  // upstream Problem::EvaluateOptions (the argument of Problem::Evaluate)
  struct EvaluateOptions {
    EvaluateOptions() : apply_loss_function(true), num_threads(1) {}
    std::vector<double*> parameter_blocks;          // empty = all, in the order they were added; otherwise the column order, others held constant
    std::vector<ResidualBlockId> residual_blocks;   // empty = all, in the order they were added
    bool apply_loss_function;
    int num_threads;
  };

  void AddParameterBlock(double* values, int size) { AddParameterBlock(values, size, nullptr); }
  void AddParameterBlock(double* values, int size, LocalParameterization* local_parameterization) {
    auto it = blocks_.find(values);
    if (it == blocks_.end()) {
      Block b; b.size = size; b.parameterization = local_parameterization; b.constant = false; b.order = (int)order_.size();
      blocks_.emplace(values, b);
      order_.push_back(values);
    } else if (local_parameterization) {
      it->second.parameterization = local_parameterization;
    }
  }
  template <typename... Ts>
  ResidualBlockId AddResidualBlock(CostFunction* cost_function, LossFunction* loss_function, double* x0, Ts*... xs) {
    std::vector<double*> pb{x0, xs...};
    return AddResidualBlock(cost_function, loss_function, pb);
  }
  ResidualBlockId AddResidualBlock(CostFunction* cost_function, LossFunction* loss_function, const std::vector<double*>& pb) {
    const std::vector<int32>& sizes = cost_function->parameter_block_sizes();
    for (size_t i = 0; i < pb.size(); ++i) AddParameterBlock(pb[i], i < sizes.size() ? sizes[i] : 0);
    std::unique_ptr<internal::ResidualBlock> rb(new internal::ResidualBlock{cost_function, loss_function, pb, (int)residual_blocks_.size()});
    ResidualBlockId id = rb.get();
    for (double* p : pb) blocks_[p].residual_blocks.push_back(id);
    residual_blocks_.push_back(std::move(rb));
    return id;
  }
  void SetParameterBlockConstant(double* values) { blocks_.at(values).constant = true; }
  void SetParameterBlockVariable(double* values) { blocks_.at(values).constant = false; }
  bool IsParameterBlockConstant(double* values) const { return blocks_.at(values).constant; }
  bool HasParameterBlock(const double* values) const { return blocks_.count(const_cast<double*>(values)) != 0; }
  int ParameterBlockSize(const double* values) const { return blocks_.at(const_cast<double*>(values)).size; }
  const LocalParameterization* GetParameterization(double* values) const { return blocks_.at(values).parameterization; }
  int NumParameterBlocks() const { return (int)order_.size(); }
  int NumResidualBlocks() const { return (int)residual_blocks_.size(); }
  void GetParameterBlocks(std::vector<double*>* out) const { *out = order_; }
  void GetResidualBlocks(std::vector<ResidualBlockId>* out) const {
    out->clear();
    for (auto& rb : residual_blocks_) out->push_back(rb.get());
  }
  void GetParameterBlocksForResidualBlock(const ResidualBlockId id, std::vector<double*>* out) const { *out = id->parameter_blocks; }
  const CostFunction* GetCostFunctionForResidualBlock(const ResidualBlockId id) const { return id->cost_function; }
  const LossFunction* GetLossFunctionForResidualBlock(const ResidualBlockId id) const { return id->loss_function; }
  void GetResidualBlocksForParameterBlock(const double* values, std::vector<ResidualBlockId>* out) const {
    *out = blocks_.at(const_cast<double*>(values)).residual_blocks;
  }

 private:
  struct Block { int size; LocalParameterization* parameterization; bool constant; int order; std::vector<ResidualBlockId> residual_blocks; };
  std::unordered_map<double*, Block> blocks_;
  std::vector<double*> order_;
  std::vector<std::unique_ptr<internal::ResidualBlock>> residual_blocks_;
};

struct Solver {
  struct Options {
    LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
    int max_num_iterations = 50;
    double max_solver_time_in_seconds = 1e9;
    int num_threads = 1;
    double initial_trust_region_radius = 1e4;
    double function_tolerance = 1e-6;
    double gradient_tolerance = 1e-10;
    double parameter_tolerance = 1e-8;
    double min_relative_decrease = 1e-3;
    bool minimizer_progress_to_stdout = false;
  };
  struct Summary {
    TerminationType termination_type = FAILURE;
    std::string message;
    double initial_cost = -1.0, final_cost = -1.0;
    int num_successful_steps = -1, num_unsuccessful_steps = -1;
    int num_residual_blocks = -1, num_residual_blocks_reduced = -1;
    int num_parameter_blocks = -1, num_parameter_blocks_reduced = -1;
    double total_time_in_seconds = -1.0;
    double preprocessor_time_in_seconds = 0.0, minimizer_time_in_seconds = 0.0, postprocessor_time_in_seconds = 0.0;
    std::string BriefReport() const {
      return "lvf(gfx950): cost " + std::to_string(initial_cost) + " -> " + std::to_string(final_cost) + ", " +
             std::to_string(num_successful_steps + num_unsuccessful_steps) + " iterations, " + message;
    }
    bool IsSolutionUsable() const { return termination_type == CONVERGENCE || termination_type == NO_CONVERGENCE || termination_type == USER_SUCCESS; }
  };
};

}  // namespace ceres
#endif  // LVF_CERES_COMPAT_H_
