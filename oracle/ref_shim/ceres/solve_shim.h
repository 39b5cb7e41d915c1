// ORACLE — TEST INFRASTRUCTURE ONLY.  ceres::Solve of the CPU stand-in (oracle/ref_shim): the DECLARED Levenberg-Marquardt loop of
// oracle/lm.h (header, lm_solve) — Ceres itself is absent from /root/reference, so its trust-region semantics are stated, not pinned —
// written generically over what the recording ceres::Problem holds, so that the reference's own CONTROL code (Mapping::Optimize / Relocate,
// mapping.cpp:114-191,251-300; PoseGraph::BuildProblem / Optimize, pose_graph.cpp:163-224; Relocator::UpdateNewSubmap, relocator.cpp:247-282),
// compiled unmodified, runs END TO END on the CPU and oracle/icp.h, oracle/loop.h and lvio_fusion_amd's scan-match / pose-graph paths
// can be compared with what it leaves behind.  Semantics (same as oracle/lm.h, oracle/icp.h, oracle/loop.h):
//   * cost = 1/2 sum rho(|r|^2); Corrector with rho'' <= 0: residual and Jacobian scaled by sqrt(rho');
//   * tangent Jacobian J_local = J_ambient * ComputeJacobian(x) per parameterised block; constant blocks drop out; residual blocks whose
//     blocks are all constant leave the reduced program (Summary::num_residual_blocks_reduced);
//   * H = J^T J, g = J^T r dense; Jacobi scaling from iteration 0 frozen (robust.h lm_damping); (H + D^2 / radius) dx = -g by Cholesky;
//   * model = -dx^T (g + H dx / 2); TrustRegionMinimizer's order of tests: max iterations, gradient, radius, invalid step (x 5 -> FAILURE),
//     parameter tolerance, function tolerance (before the step-quality test, candidate not taken), rho > min_relative_decrease.
// Include in exactly ONE translation unit of a library built against ref_shim/ceres/ceres.h.
#pragma once
#include <cmath>
#include <vector>

#include <ceres/ceres.h>

namespace ceres {
namespace shim {

inline double lm_damping(double h, double h0) {      // oracle/robust.h lm_damping
  const double s = 1.0 / (1.0 + std::sqrt(h0)), s2 = s * s;
  return std::fmin(std::fmax(h * s2, 1e-6), 1e32) / s2;
}

struct Program {
  struct Var { double* values; int size, local, off; LocalParameterization* lp; };
  std::vector<Var> vars;                 // active (non-constant, used) parameter blocks
  std::vector<int> var_of;               // problem parameter block index -> vars index or -1
  std::vector<ResidualBlockId> blocks;   // residual blocks of the reduced program
  int n = 0;                             // tangent dimension
};

inline void build_program(const Problem& p, Program& g) {
  const auto& pbs = p.parameter_blocks();
  std::vector<char> used(pbs.size(), 0);
  for (ResidualBlockId b : p.recorded_blocks()) {
    bool any = false;
    for (double* q : b->params) if (!pbs[(size_t)p.parameter_block_index(q)].constant) any = true;
    if (!any) continue;
    g.blocks.push_back(b);
    for (double* q : b->params) used[(size_t)p.parameter_block_index(q)] = 1;
  }
  g.var_of.assign(pbs.size(), -1);
  for (size_t i = 0; i < pbs.size(); ++i) {
    if (pbs[i].constant || !used[i]) continue;
    Program::Var v{pbs[i].values, pbs[i].size, pbs[i].parameterization ? pbs[i].parameterization->LocalSize() : pbs[i].size, g.n, pbs[i].parameterization};
    g.var_of[i] = (int)g.vars.size();
    g.vars.push_back(v);
    g.n += v.local;
  }
}

// cost at the state given by `state` (one pointer per problem parameter block); with H / grad non-null also the normal equations
inline double evaluate(const Problem& p, const Program& g, const std::vector<const double*>& state, std::vector<double>* H, std::vector<double>* grad) {
  const int n = g.n;
  if (H) { H->assign((size_t)n * n, 0.0); grad->assign((size_t)n, 0.0); }
  double cost = 0.0;
  std::vector<double> r, jl;
  std::vector<std::vector<double>> jac;
  std::vector<double*> jp;
  std::vector<const double*> prm;
  for (ResidualBlockId b : g.blocks) {
    const int R = b->cost->num_residuals();
    const std::vector<int>& sizes = b->cost->parameter_block_sizes();
    const size_t K = b->params.size();
    prm.resize(K); r.assign((size_t)R, 0.0);
    for (size_t k = 0; k < K; ++k) prm[k] = state[(size_t)p.parameter_block_index(b->params[k])];
    if (H) {
      jac.resize(K); jp.resize(K);
      for (size_t k = 0; k < K; ++k) { jac[k].assign((size_t)R * sizes[k], 0.0); jp[k] = jac[k].data(); }
      b->cost->Evaluate(prm.data(), r.data(), jp.data());
    } else {
      b->cost->Evaluate(prm.data(), r.data(), nullptr);
    }
    double s = 0.0;
    for (int i = 0; i < R; ++i) s += r[i] * r[i];
    double rho[3] = {s, 1.0, 0.0};
    if (b->loss) b->loss->Evaluate(s, rho);
    cost += 0.5 * rho[0];
    if (!H) continue;
    const double sc = std::sqrt(rho[1]);
    // local, robustified Jacobian pieces of the active blocks
    struct Piece { int off, width; std::vector<double> J; };
    std::vector<Piece> pcs;
    for (size_t k = 0; k < K; ++k) {
      const int vi = g.var_of[(size_t)p.parameter_block_index(b->params[k])];
      if (vi < 0) continue;
      const Program::Var& v = g.vars[(size_t)vi];
      Piece pc{v.off, v.local, std::vector<double>((size_t)R * v.local, 0.0)};
      if (v.lp) {
        std::vector<double> P((size_t)v.size * v.local);
        v.lp->ComputeJacobian(prm[k], P.data());
        for (int i = 0; i < R; ++i)
          for (int c = 0; c < v.local; ++c) {
            double a = 0.0;
            for (int q = 0; q < v.size; ++q) a += jac[k][(size_t)i * v.size + q] * P[(size_t)q * v.local + c];
            pc.J[(size_t)i * v.local + c] = sc * a;
          }
      } else {
        for (int i = 0; i < R * v.size; ++i) pc.J[(size_t)i] = sc * jac[k][(size_t)i];
      }
      pcs.push_back(std::move(pc));
    }
    for (const Piece& a : pcs)
      for (int i = 0; i < a.width; ++i) {
        double gi = 0.0;
        for (int q = 0; q < R; ++q) gi += a.J[(size_t)q * a.width + i] * (sc * r[q]);
        (*grad)[(size_t)a.off + i] += gi;
        for (const Piece& c : pcs)
          for (int j = 0; j < c.width; ++j) {
            double h = 0.0;
            for (int q = 0; q < R; ++q) h += a.J[(size_t)q * a.width + i] * c.J[(size_t)q * c.width + j];
            (*H)[(size_t)(a.off + i) * n + c.off + j] += h;
          }
      }
  }
  return cost;
}

inline bool cholesky_solve(int n, std::vector<double>& A, std::vector<double>& x) {      // A x = b in place (b in x); lower Cholesky
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double v = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = v / d;
    }
  }
  for (int i = 0; i < n; ++i) { double v = x[(size_t)i]; for (int k = 0; k < i; ++k) v -= A[(size_t)i * n + k] * x[(size_t)k]; x[(size_t)i] = v / A[(size_t)i * n + i]; }
  for (int i = n - 1; i >= 0; --i) { double v = x[(size_t)i]; for (int k = i + 1; k < n; ++k) v -= A[(size_t)k * n + i] * x[(size_t)k]; x[(size_t)i] = v / A[(size_t)i * n + i]; }
  for (int i = 0; i < n; ++i) if (!std::isfinite(x[(size_t)i])) return false;
  return true;
}

}  // namespace shim

void Solve(const Solver::Options& o, Problem* problem, Solver::Summary* s) {
  using namespace shim;
  Program g;
  build_program(*problem, g);
  const auto& pbs = problem->parameter_blocks();
  *s = Solver::Summary();
  s->num_residual_blocks = problem->NumResidualBlocks();
  s->num_residual_blocks_reduced = (int)g.blocks.size();
  const int n = g.n;
  std::vector<const double*> cur(pbs.size()), cand_state(pbs.size());
  for (size_t i = 0; i < pbs.size(); ++i) cur[i] = pbs[i].values;
  std::vector<std::vector<double>> cand(g.vars.size());
  for (size_t v = 0; v < g.vars.size(); ++v) cand[v].resize((size_t)g.vars[v].size);
  double radius = o.initial_trust_region_radius, decrease = 2.0, cost = 0.0;
  bool first = true;
  int invalid_run = 0;
  std::vector<double> H, grad, h0, A, dx;
  s->termination_type = NO_CONVERGENCE;
  for (;;) {
    cost = evaluate(*problem, g, cur, &H, &grad);
    if (first) { s->initial_cost = cost; first = false; h0.resize((size_t)n); for (int u = 0; u < n; ++u) h0[(size_t)u] = H[(size_t)u * n + u]; }
    if (s->num_iterations >= o.max_num_iterations) break;
    double gmax = 0.0;
    for (int u = 0; u < n; ++u) gmax = std::fmax(gmax, std::fabs(grad[(size_t)u]));
    if (gmax <= o.gradient_tolerance) { s->termination_type = CONVERGENCE; break; }
    if (radius < 1e-32) { s->termination_type = CONVERGENCE; break; }
    A = H; dx.assign((size_t)n, 0.0);
    for (int u = 0; u < n; ++u) { A[(size_t)u * n + u] += lm_damping(H[(size_t)u * n + u], h0[(size_t)u]) / radius; dx[(size_t)u] = -grad[(size_t)u]; }
    const bool ok = n > 0 && cholesky_solve(n, A, dx);
    double model = 0.0;
    if (ok)
      for (int u = 0; u < n; ++u) {
        double hd = 0.0;
        for (int v = 0; v < n; ++v) hd += H[(size_t)u * n + v] * dx[(size_t)v];
        model -= dx[(size_t)u] * (grad[(size_t)u] + 0.5 * hd);
      }
    if (!(ok && model > 0.0)) {
      s->num_iterations += 1; s->num_unsuccessful_steps += 1;
      if (++invalid_run >= 5) { s->termination_type = FAILURE; break; }
      radius *= 0.5;
      continue;
    }
    // candidate x + dx (Plus per block), its ambient step norm and the norm of x over the reduced program
    double d2 = 0.0, x2 = 0.0;
    cand_state = cur;
    for (size_t v = 0; v < g.vars.size(); ++v) {
      const Program::Var& var = g.vars[v];
      if (var.lp) var.lp->Plus(var.values, dx.data() + var.off, cand[v].data());
      else for (int k = 0; k < var.size; ++k) cand[v][(size_t)k] = var.values[k] + dx[(size_t)var.off + k];
      for (int k = 0; k < var.size; ++k) { const double e = cand[v][(size_t)k] - var.values[k]; d2 += e * e; x2 += var.values[k] * var.values[k]; }
    }
    for (size_t i = 0; i < pbs.size(); ++i) if (g.var_of[i] >= 0) cand_state[i] = cand[(size_t)g.var_of[i]].data();
    const double cc = evaluate(*problem, g, cand_state, nullptr, nullptr);
    if (!std::isfinite(cc)) {
      s->num_iterations += 1; s->num_unsuccessful_steps += 1;
      if (++invalid_run >= 5) { s->termination_type = FAILURE; break; }
      radius *= 0.5;
      continue;
    }
    invalid_run = 0;
    if (std::sqrt(d2) <= o.parameter_tolerance * (std::sqrt(x2) + o.parameter_tolerance)) { s->termination_type = CONVERGENCE; break; }
    if (std::fabs(cost - cc) <= o.function_tolerance * cost) { s->termination_type = CONVERGENCE; break; }
    s->num_iterations += 1;
    const double rho = (cost - cc) / model;
    if (rho > o.min_relative_decrease) {
      for (size_t v = 0; v < g.vars.size(); ++v) for (int k = 0; k < g.vars[v].size; ++k) g.vars[v].values[k] = cand[v][(size_t)k];
      cost = cc; s->num_successful_steps += 1;
      const double t = 2.0 * rho - 1.0;
      radius = std::fmin(radius / std::fmax(1.0 / 3.0, 1.0 - t * t * t), 1e16); decrease = 2.0;
    } else {
      radius /= decrease; decrease *= 2.0; s->num_unsuccessful_steps += 1;
    }
  }
  s->final_cost = cost;
}

}  // namespace ceres
