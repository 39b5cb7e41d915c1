// evaluate_local.hip — the batched Problem::Evaluate surface (SURVEY.md §8b "batched-evaluate surface"): residuals and Jacobians of a
// whole factor batch as upstream ceres::Problem::Evaluate(EvaluateOptions, cost, residuals, gradient, CRSMatrix*) reports them —
// with the loss function's Corrector applied (residual and Jacobian rows of a block scaled by sqrt(rho'), rho'' <= 0 for Huber /
// Trivial) and the Jacobian in LOCAL (tangent) coordinates: every 7-sized pose block is multiplied by the plus-Jacobian of
// ProductParameterization(EigenQuaternionParameterization, IdentityParameterization(3)) (src/lvio_fusion/src/backend.cpp:99-101), 7 -> 6
// columns.  Input = the materialised Ceres-layout outputs of lvf_batch_evaluate (ambient coordinates); one thread per residual row.
// include/lvf_ceres_adapter.hpp's gpu::Evaluate scatters these values into the caller's CRSMatrix.
#include "lvf_internal.hpp"

namespace lvf {

struct LocalDesc {
  int n, R, nb, L;                 // blocks, residual rows per block, parameter blocks, local columns per row
  int size[8];                     // ambient size of parameter block b
  const int* kf[8];                // pose blocks: keyframe index per residual block (null for non-pose blocks; negative index = absent block)
  const double* jac[8];            // ambient Jacobians [n][R][size]
  const double* res;               // [n][R]
  const double* poses;             // [n_kf][7]
  double huber;                    // <= 0: no robustification
  int robust_rows;                 // rows whose squared norm feeds the loss (= R for the visual functors)
};

__global__ __launch_bounds__(256) void k_to_local(LocalDesc d, double* __restrict__ out_res, double* __restrict__ out_jac) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= d.n * d.R) return;
  const int i = e / d.R;
  double sc = 1.0;
  if (d.huber > 0.0) {
    double s = 0.0;
    for (int k = 0; k < d.robust_rows; ++k) { const double v = d.res[(size_t)i * d.R + k]; s += v * v; }
    double rho;
    sc = robust_scale(d.huber, s, rho);
  }
  out_res[e] = sc * d.res[e];
  double* o = out_jac + (size_t)e * d.L;
  int col = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    if (b >= d.nb) break;
    const int sz = d.size[b];
    const double* row = d.jac[b] + (size_t)e * sz;
    if (d.kf[b]) {                 // pose block: [q(4) | t(3)] -> [tangent(3) | t(3)]
      const int k = d.kf[b][i];
      if (k >= 0) {
        double l3[3];
        quat_row_to_local(row, d.poses + (size_t)7 * k, l3);
        o[col] = sc * l3[0]; o[col + 1] = sc * l3[1]; o[col + 2] = sc * l3[2];
        o[col + 3] = sc * row[4]; o[col + 4] = sc * row[5]; o[col + 5] = sc * row[6];
      } else {
        for (int c = 0; c < 6; ++c) o[col + c] = 0.0;
      }
      col += 6;
    } else {
      for (int c = 0; c < sz; ++c) o[col + c] = sc * row[c];
      col += sz;
    }
  }
}

}  // namespace lvf

using namespace lvf;

extern "C" {

int lvf_batch_local_columns(const lvf_batch* b) {
  if (!b) return -1;
  int L = 0;
  for (int k = 0; k < b->n_blocks; ++k) L += b->block_size[k] == 7 ? 6 : b->block_size[k];
  return L;
}

int lvf_batch_evaluate_local(lvf_batch* b, const lvf_state* st, double huber_a, double* residuals, double* jacobians_local) {
  LVF_REQUIRE(b && st && residuals, "lvf_batch_evaluate_local: null argument");
  LVF_REQUIRE(b->kind != LVF_K_LIDAR, "lvf_batch_evaluate_local: lidar batches have scalar parameter blocks (their ambient Jacobians are already local)");
  LVF_TRY(lvf_batch_evaluate(b, st, nullptr, jacobians_local ? 1 : 0));
  if (b->n == 0) return LVF_OK;
  hipStream_t s = b->ctx->stream;
  if (!jacobians_local && !(huber_a > 0.0)) return lvf_batch_download_residuals(b, residuals);
  LocalDesc d{};
  d.n = b->n; d.R = b->n_res; d.nb = jacobians_local ? b->n_blocks : 0; d.L = lvf_batch_local_columns(b);
  d.res = b->res.p; d.poses = st->poses.p;
  const bool visual = b->kind == LVF_K_POSE_ONLY || b->kind == LVF_K_TWO_FRAME || b->kind == LVF_K_TWO_CAMERA;
  d.huber = visual ? huber_a : 0.0;      // the reference passes its HuberLoss to the visual blocks only (backend.cpp:124,130,139 vs :159,171,176)
  d.robust_rows = b->n_res;
  for (int k = 0; k < b->n_blocks; ++k) { d.size[k] = b->block_size[k]; d.jac[k] = b->jac[k].p; d.kf[k] = nullptr; }
  switch (b->kind) {
    case LVF_K_POSE_ONLY: d.kf[0] = b->idx_a.p; break;
    case LVF_K_TWO_FRAME: d.kf[1] = b->idx_b.p; d.kf[2] = b->idx_c.p; break;
    case LVF_K_IMU: d.kf[0] = b->idx_a.p; d.kf[4] = b->idx_b.p; break;
    case LVF_K_POSE_PRIOR: d.kf[0] = b->idx_a.p; d.kf[1] = b->idx_b.p; break;
    default: break;
  }
  DevBuf<double> orr, oj;
  LVF_TRY(orr.alloc((size_t)b->n * b->n_res));
  LVF_TRY(oj.alloc(std::max<size_t>(1, (size_t)b->n * b->n_res * d.L)));
  hipLaunchKernelGGL(k_to_local, dim3((b->n * b->n_res + 255) / 256), dim3(256), 0, s, d, orr.p, oj.p);
  LVF_HIP(hipGetLastError());
  LVF_HIP(hipMemcpyAsync(residuals, orr.p, (size_t)b->n * b->n_res * 8, hipMemcpyDeviceToHost, s));
  if (jacobians_local) LVF_HIP(hipMemcpyAsync(jacobians_local, oj.p, (size_t)b->n * b->n_res * d.L * 8, hipMemcpyDeviceToHost, s));
  LVF_HIP(hipStreamSynchronize(s));
  return LVF_OK;
}

}  // extern "C"
