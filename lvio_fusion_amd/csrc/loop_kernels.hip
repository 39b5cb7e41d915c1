// loop_kernels.hip — the loop-correction tail after the multi-GPU candidate gather, on device:
//
//   RelocateRError <7,4>                           src/lvio_fusion/include/lvio_fusion/ceres/pose_error.hpp:192-222
//   Relocator::UpdateNewSubmap's rotation solve    src/lvio_fusion/src/relocator.cpp:247-268
//       one quaternion parameter r under EigenQuaternionParameterization, one RelocateRError(relocated_i, unrelocated_i) per keyframe
//       of the new submap, no loss function, ceres::Solve(DENSE_QR) with default options;
//   PoseGraph::ForwardUpdate                       src/lvio_fusion/src/pose_graph.cpp:245-252   (also Backend::UpdateFrontend, backend.cpp:256)
//       pose <- transform * pose, Vw <- transform.unit_quaternion() * Vw for every keyframe after the corrected section.
//
// RelocateRError goes through SE3Product of the UN-normalised parameter quaternion (the rotation of the translation normalises it,
// the quaternion product does not), so its ambient 7x4 Jacobian is taken exactly as the reference's autodiff does: dual numbers
// (djet.hpp).  The problem has 3 tangent unknowns and a few dozen blocks: the whole LM loop (linearise, damped 3x3 solve, candidate
// cost, accept / reject, trust-region update, termination tests) is ONE launch of one workgroup; the host reads one record.
#include "lvf_internal.hpp"
#include "se3_jet.hpp"

namespace lvf {

constexpr int kLT = 256;

// residual r[7] (+ ambient J[7][4], row-major) of one block at quaternion q (x,y,z,w)
template <bool WITH_J>
__device__ __forceinline__ void relocate_r_eval(const double* __restrict__ relocated, const double* __restrict__ unrelocated, const double q[4], double r[7],
                                                double J[28]) {
  typedef DJet<4> T;
  T R[7], U[7], RU[7];
#pragma unroll
  for (int k = 0; k < 4; ++k) R[k] = WITH_J ? T(q[k], k) : T(q[k]);
  R[4] = T(0.0); R[5] = T(0.0); R[6] = T(0.0);
#pragma unroll
  for (int k = 0; k < 7; ++k) U[k] = T(unrelocated[k]);
  se3_product(R, U, RU);
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    r[k] = relocated[k] - RU[k].a;
    if (WITH_J) {
#pragma unroll
      for (int c = 0; c < 4; ++c) J[4 * k + c] = -RU[k].v[c];
    }
  }
}

template <bool WITH_J>
__global__ __launch_bounds__(64) void k_relocate_r(int n, const double* __restrict__ relocated, const double* __restrict__ unrelocated,
                                                   const double* __restrict__ q4, double* __restrict__ res, double* __restrict__ jac) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const double q[4] = {q4[0], q4[1], q4[2], q4[3]};
  double r[7], J[28];
  relocate_r_eval<WITH_J>(relocated + 7 * i, unrelocated + 7 * i, q, r, J);
#pragma unroll
  for (int k = 0; k < 7; ++k) res[7 * i + k] = r[k];
  if (WITH_J) {
#pragma unroll
    for (int k = 0; k < 28; ++k) jac[28 * i + k] = J[k];
  }
}

struct RelocOpts { int max_iters; double function_tol, gradient_tol, parameter_tol, min_rel_decrease, radius0; };
struct RelocRecord { double q[4]; double initial_cost, final_cost; int iters, successes, termination, pad; };

// sums `v` over the workgroup; every thread returns the total (two barriers)
__device__ __forceinline__ double wg_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < kLT / 64; ++k) s += red[k];
  return s;
}

__global__ __launch_bounds__(kLT) void k_relocate_solve(int n, const double* __restrict__ relocated, const double* __restrict__ unrelocated,
                                                        const double* __restrict__ q_in, RelocOpts o, RelocRecord* __restrict__ out) {
  __shared__ double red[kLT / 64];
  const int tid = threadIdx.x;
  double q[4] = {q_in[0], q_in[1], q_in[2], q_in[3]};
  double radius = o.radius0, decrease = 2.0, cost = 0.0, initial_cost = 0.0;
  int iters = 0, successes = 0, termination = 1;
  // ceres::Solve's TrustRegionMinimizer order (declared semantics: oracle/lm.h lm_solve; this problem's restatement: oracle/loop.h)
  bool first = true;
  double h0[3] = {0.0, 0.0, 0.0};
  int invalid_run = 0;
  for (;;) {
    // EigenQuaternionParameterization::ComputeJacobian at q
    const double P[12] = {q[3], q[2], -q[1], -q[2], q[3], q[0], q[1], -q[0], q[3], -q[0], -q[1], -q[2]};
    double h[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, c = 0.0;
    for (int i = tid; i < n; i += kLT) {
      double r[7], J[28];
      relocate_r_eval<true>(relocated + 7 * i, unrelocated + 7 * i, q, r, J);
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        c += 0.5 * r[k] * r[k];
        double l[3];
#pragma unroll
        for (int x = 0; x < 3; ++x) l[x] = J[4 * k] * P[x] + J[4 * k + 1] * P[3 + x] + J[4 * k + 2] * P[6 + x] + J[4 * k + 3] * P[9 + x];
        g[0] += l[0] * r[k]; g[1] += l[1] * r[k]; g[2] += l[2] * r[k];
        h[0] += l[0] * l[0]; h[1] += l[1] * l[0]; h[2] += l[1] * l[1]; h[3] += l[2] * l[0]; h[4] += l[2] * l[1]; h[5] += l[2] * l[2];
      }
    }
    cost = wg_sum(c, red);
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] = wg_sum(g[k], red);
#pragma unroll
    for (int k = 0; k < 6; ++k) h[k] = wg_sum(h[k], red);
    // every thread holds the same sums and takes the same decisions (no divergence across the barriers below)
    if (first) { initial_cost = cost; first = false; h0[0] = h[0]; h0[1] = h[2]; h0[2] = h[5]; }      // Jacobi scaling: iteration 0, frozen (Ceres default jacobi_scaling; relocator.cpp:259 solves with default options)
    if (iters >= o.max_iters) break;
    if (fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2]))) <= o.gradient_tol) { termination = 0; break; }
    if (radius < 1e-32) { termination = 0; break; }
    const double H[3][3] = {{h[0], h[1], h[3]}, {h[1], h[2], h[4]}, {h[3], h[4], h[5]}};
    double A[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
#pragma unroll
      for (int v = 0; v < 3; ++v) A[u][v] = H[u][v];
      { const double sj = 1.0 / (1.0 + sqrt(h0[u])), s2 = sj * sj; A[u][u] += fmin(fmax(H[u][u] * s2, 1e-6), 1e32) / s2 / radius; }
    }
    const double l00 = sqrt(A[0][0]), l10 = A[1][0] / l00, l20 = A[2][0] / l00;
    const double t11 = A[1][1] - l10 * l10, l11 = sqrt(t11), l21 = (A[2][1] - l20 * l10) / l11;
    const double t22 = A[2][2] - l20 * l20 - l21 * l21, l22 = sqrt(t22);
    const bool ok = A[0][0] > 0.0 && t11 > 0.0 && t22 > 0.0;
    double dx[3] = {0.0, 0.0, 0.0};
    if (ok) {
      const double y0 = -g[0] / l00, y1 = (-g[1] - l10 * y0) / l11, y2 = (-g[2] - l20 * y0 - l21 * y1) / l22;
      dx[2] = y2 / l22; dx[1] = (y1 - l21 * dx[2]) / l11; dx[0] = (y0 - l10 * dx[1] - l20 * dx[2]) / l00;
    }
    double model = 0.0;
#pragma unroll
    for (int u = 0; u < 3; ++u) model -= dx[u] * (g[u] + 0.5 * (H[u][0] * dx[0] + H[u][1] * dx[1] + H[u][2] * dx[2]));
    // EigenQuaternionParameterization::Plus
    double qc[4] = {q[0], q[1], q[2], q[3]};
    const double dn = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
    if (dn > 0.0) {
      const double sn = sin(dn) / dn, cw = cos(dn);
      const double ax = sn * dx[0], ay = sn * dx[1], az = sn * dx[2];
      qc[3] = cw * q[3] - ax * q[0] - ay * q[1] - az * q[2];
      qc[0] = cw * q[0] + ax * q[3] + ay * q[2] - az * q[1];
      qc[1] = cw * q[1] - ax * q[2] + ay * q[3] + az * q[0];
      qc[2] = cw * q[2] + ax * q[1] - ay * q[0] + az * q[3];
    }
    if (!(ok && model > 0.0)) {                      // invalid step (uniform across the workgroup)
      ++iters;
      if (++invalid_run >= 5) { termination = 2; break; }
      radius *= 0.5;
      continue;
    }
    invalid_run = 0;
    const double xn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double sn2 = (qc[0] - q[0]) * (qc[0] - q[0]) + (qc[1] - q[1]) * (qc[1] - q[1]) + (qc[2] - q[2]) * (qc[2] - q[2]) + (qc[3] - q[3]) * (qc[3] - q[3]);
    if (sqrt(sn2) <= o.parameter_tol * (xn + o.parameter_tol)) { termination = 0; break; }
    double cc = 0.0;
    for (int i = tid; i < n; i += kLT) {
      double r[7], J[28];
      relocate_r_eval<false>(relocated + 7 * i, unrelocated + 7 * i, qc, r, J);
#pragma unroll
      for (int k = 0; k < 7; ++k) cc += 0.5 * r[k] * r[k];
    }
    const double cand = wg_sum(cc, red);
    if (fabs(cost - cand) <= o.function_tol * cost) { termination = 0; break; }
    ++iters;
    const double rho = (cost - cand) / model;
    if (rho > o.min_rel_decrease) {
      q[0] = qc[0]; q[1] = qc[1]; q[2] = qc[2]; q[3] = qc[3];
      cost = cand; ++successes;
      const double t = 2.0 * rho - 1.0;
      radius = fmin(radius / fmax(1.0 / 3.0, 1.0 - t * t * t), 1e16); decrease = 2.0;
    } else { radius /= decrease; decrease *= 2.0; }
  }
  if (tid == 0) {
    out->q[0] = q[0]; out->q[1] = q[1]; out->q[2] = q[2]; out->q[3] = q[3];
    out->initial_cost = initial_cost; out->final_cost = cost; out->iters = iters; out->successes = successes; out->termination = termination; out->pad = 0;
  }
}

// PoseGraph::ForwardUpdate: Sophus SE3 product (Hamilton product re-normalised; translation through Eigen's _transformVector)
__global__ __launch_bounds__(kLT) void k_forward_update(int n, const double* __restrict__ T, double* __restrict__ poses, double* __restrict__ vw) {
  const int i = blockIdx.x * kLT + threadIdx.x;
  if (i >= n) return;
  const double qn = sqrt(T[0] * T[0] + T[1] * T[1] + T[2] * T[2] + T[3] * T[3]);
  const double ux = T[0] / qn, uy = T[1] / qn, uz = T[2] / qn, uw = T[3] / qn;
  double* p = poses + (size_t)7 * i;
  const double bx = p[0], by = p[1], bz = p[2], bw = p[3];
  const double w = uw * bw - ux * bx - uy * by - uz * bz, x = uw * bx + ux * bw + uy * bz - uz * by, y = uw * by + uy * bw + uz * bx - ux * bz,
               z = uw * bz + uz * bw + ux * by - uy * bx;
  const double nn = sqrt(w * w + x * x + y * y + z * z);
  {
    const double v0 = p[4], v1 = p[5], v2 = p[6];
    const double cx = 2.0 * (uy * v2 - uz * v1), cy = 2.0 * (uz * v0 - ux * v2), cz = 2.0 * (ux * v1 - uy * v0);
    p[4] = T[4] + (v0 + uw * cx + (uy * cz - uz * cy)); p[5] = T[5] + (v1 + uw * cy + (uz * cx - ux * cz)); p[6] = T[6] + (v2 + uw * cz + (ux * cy - uy * cx));
  }
  p[0] = x / nn; p[1] = y / nn; p[2] = z / nn; p[3] = w / nn;
  if (vw) {
    double* v = vw + (size_t)3 * i;
    const double v0 = v[0], v1 = v[1], v2 = v[2];
    const double cx = 2.0 * (uy * v2 - uz * v1), cy = 2.0 * (uz * v0 - ux * v2), cz = 2.0 * (ux * v1 - uy * v0);
    v[0] = v0 + uw * cx + (uy * cz - uz * cy); v[1] = v1 + uw * cy + (uz * cx - ux * cz); v[2] = v2 + uw * cz + (ux * cy - uy * cx);
  }
}

}  // namespace lvf

using namespace lvf;

extern "C" {

int lvf_relocate_r_evaluate(lvf_ctx* ctx, int n, const double* relocated, const double* unrelocated, const double* q4, double* residuals, double* jacobians) {
  LVF_REQUIRE(ctx && q4 && residuals, "lvf_relocate_r_evaluate: null argument");
  LVF_REQUIRE(n >= 0 && (n == 0 || (relocated && unrelocated)), "lvf_relocate_r_evaluate: bad block arrays");
  if (n == 0) return LVF_OK;
  LVF_TRY(lvf::enter(ctx));
  hipStream_t s = ctx->stream;
  DevBuf<double> a, b, q, r, J;
  LVF_TRY(a.upload(relocated, (size_t)7 * n, s)); LVF_TRY(b.upload(unrelocated, (size_t)7 * n, s)); LVF_TRY(q.upload(q4, 4, s));
  LVF_TRY(r.alloc((size_t)7 * n));
  if (jacobians) {
    LVF_TRY(J.alloc((size_t)28 * n));
    hipLaunchKernelGGL(k_relocate_r<true>, dim3((n + 63) / 64), dim3(64), 0, s, n, a.p, b.p, q.p, r.p, J.p);
  } else {
    hipLaunchKernelGGL(k_relocate_r<false>, dim3((n + 63) / 64), dim3(64), 0, s, n, a.p, b.p, q.p, r.p, (double*)nullptr);
  }
  LVF_HIP(hipGetLastError());
  LVF_HIP(hipMemcpyAsync(residuals, r.p, (size_t)7 * n * 8, hipMemcpyDeviceToHost, s));
  if (jacobians) LVF_HIP(hipMemcpyAsync(jacobians, J.p, (size_t)28 * n * 8, hipMemcpyDeviceToHost, s));
  LVF_HIP(hipStreamSynchronize(s));
  return LVF_OK;
}

int lvf_relocate_rotation_solve(lvf_ctx* ctx, int n, const double* relocated, const double* unrelocated, double* q4, const lvf_solver_options* o,
                                lvf_solver_summary* summary) {
  LVF_REQUIRE(ctx && q4 && o && summary, "lvf_relocate_rotation_solve: null argument");
  LVF_REQUIRE(n >= 0 && (n == 0 || (relocated && unrelocated)), "lvf_relocate_rotation_solve: bad block arrays");
  LVF_REQUIRE(q4[0] * q4[0] + q4[1] * q4[1] + q4[2] * q4[2] + q4[3] * q4[3] > 0.0, "lvf_relocate_rotation_solve: zero quaternion");
  std::memset(summary, 0, sizeof(*summary));
  summary->num_residual_blocks = n;
  if (n == 0) return LVF_OK;
  LVF_TRY(lvf::enter(ctx));
  hipStream_t s = ctx->stream;
  DevBuf<double> a, b, q;
  DevBuf<RelocRecord> rec;
  LVF_TRY(a.upload(relocated, (size_t)7 * n, s)); LVF_TRY(b.upload(unrelocated, (size_t)7 * n, s)); LVF_TRY(q.upload(q4, 4, s)); LVF_TRY(rec.alloc(1));
  const RelocOpts ro{o->max_num_iterations, o->function_tolerance, o->gradient_tolerance, o->parameter_tolerance, o->min_relative_decrease,
                     o->initial_trust_region_radius};
  hipLaunchKernelGGL(k_relocate_solve, dim3(1), dim3(kLT), 0, s, n, a.p, b.p, q.p, ro, rec.p);
  LVF_HIP(hipGetLastError());
  RelocRecord h;
  LVF_HIP(hipMemcpyAsync(&h, rec.p, sizeof(h), hipMemcpyDeviceToHost, s));
  LVF_HIP(hipStreamSynchronize(s));
  if (h.termination == 2 || !std::isfinite(h.final_cost)) { summary->termination = 2; summary->initial_cost = h.initial_cost; summary->final_cost = h.initial_cost; return LVF_OK; }   // fail soft: q4 untouched
  std::memcpy(q4, h.q, 32);
  summary->initial_cost = h.initial_cost; summary->final_cost = h.final_cost; summary->num_iterations = h.iters; summary->num_successful_steps = h.successes;
  summary->termination = h.termination; summary->num_unsuccessful_steps = h.iters - h.successes;
  return LVF_OK;
}

int lvf_forward_update(lvf_ctx* ctx, const double* transform7, int n, double* poses, double* vw) {
  LVF_REQUIRE(ctx && transform7, "lvf_forward_update: null argument");
  LVF_REQUIRE(n >= 0 && (n == 0 || poses), "lvf_forward_update: bad pose array");
  LVF_REQUIRE(transform7[0] * transform7[0] + transform7[1] * transform7[1] + transform7[2] * transform7[2] + transform7[3] * transform7[3] > 0.0,
              "lvf_forward_update: zero quaternion");
  if (n == 0) return LVF_OK;
  LVF_TRY(lvf::enter(ctx));
  hipStream_t s = ctx->stream;
  DevBuf<double> T, P, V;
  LVF_TRY(T.upload(transform7, 7, s)); LVF_TRY(P.upload(poses, (size_t)7 * n, s));
  if (vw) LVF_TRY(V.upload(vw, (size_t)3 * n, s));
  hipLaunchKernelGGL(k_forward_update, dim3((n + kLT - 1) / kLT), dim3(kLT), 0, s, n, T.p, P.p, vw ? V.p : (double*)nullptr);
  LVF_HIP(hipGetLastError());
  LVF_HIP(hipMemcpyAsync(poses, P.p, (size_t)7 * n * 8, hipMemcpyDeviceToHost, s));
  if (vw) LVF_HIP(hipMemcpyAsync(vw, V.p, (size_t)3 * n * 8, hipMemcpyDeviceToHost, s));
  LVF_HIP(hipStreamSynchronize(s));
  return LVF_OK;
}

int lvf_state_forward_update(lvf_state* st, const double* transform7, int first_kf) {
  LVF_REQUIRE(st && transform7, "lvf_state_forward_update: null argument");
  LVF_REQUIRE(first_kf >= 0 && first_kf <= st->n_kf, "lvf_state_forward_update: first_kf %d out of range [0,%d]", first_kf, st->n_kf);
  LVF_REQUIRE(transform7[0] * transform7[0] + transform7[1] * transform7[1] + transform7[2] * transform7[2] + transform7[3] * transform7[3] > 0.0,
              "lvf_state_forward_update: zero quaternion");
  const int n = st->n_kf - first_kf;
  if (n == 0) return LVF_OK;
  LVF_TRY(lvf::enter(st->ctx));
  hipStream_t s = st->ctx->stream;
  DevBuf<double> T;
  LVF_TRY(T.upload(transform7, 7, s));
  hipLaunchKernelGGL(k_forward_update, dim3((n + kLT - 1) / kLT), dim3(kLT), 0, s, n, T.p, st->poses.p + (size_t)7 * first_kf, st->vel.p + (size_t)3 * first_kf);
  LVF_HIP(hipGetLastError());
  LVF_HIP(hipStreamSynchronize(s));     // T is released on return
  return LVF_OK;
}

}  // extern "C"
