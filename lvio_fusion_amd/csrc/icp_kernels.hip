// icp_kernels.hip — one scan-to-map sub-problem solved entirely on device.
//
// Replaces FeatureAssociation::ScanToMapWithGround / ScanToMapWithSegmented (src/lvio_fusion/src/association.cpp:270-384)
// plus the 3-parameter ceres::Solve that follows them (src/lvio_fusion/src/mapping.cpp:154-178, :264-296):
//   association (k_knn3) -> correspondence build (plane normal from the three map neighbours, float -> double as
//   association.cpp:303-314) -> up to max_num_iterations Levenberg-Marquardt steps on (pitch,roll,z) or (yaw,x,y).
// The LM loop never leaves the GPU: parameters, trust-region radius, accept/reject and termination live in a small
// device struct; every iteration is four dependent launches on the context's stream
//   eval(x, with J) -> step (1 thread: 3x3 damped normal equations) -> eval(x + dx, cost only) -> decide (1 thread)
// and the host only reads the result back at the end.  Normal-equation sums are wave-shuffle reduced (10 doubles per
// wave) before touching memory.  Solver semantics as declared for the BA problem (oracle/lm.h header); DENSE_QR and the
// damped normal equations solve the same 3x3 least-squares step.
#include <algorithm>
#include <cmath>
#include "lidar_eval.hpp"
#include "lvf_internal.hpp"
#include "scan_match_dev.hpp"

namespace lvf {

constexpr int kTI = 256;
// (IcpDev, IcpArgs, kIcpMaxBlocks: scan_match_dev.hpp)

__device__ __forceinline__ void param_slots(int mode, int& i0, int& i1, int& i2) {
  if (mode == 0) { i0 = 1; i1 = 2; i2 = 5; } else { i0 = 0; i1 = 3; i2 = 4; }
}

// (the correspondences — scan point, first neighbour pa, unit plane normal, SoA [3][Q] — are written by the association kernel: knn_kernels.hip)
// the sums other workgroups added with L2 atomics, read past this CU's L1
__device__ __forceinline__ double fresh(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// LM damping under Ceres' Jacobi column scaling, in unscaled terms (oracle/robust.h lm_damping): h = H_jj now, h0 = H_jj at iteration 0
__device__ __forceinline__ double lm_damping(double h, double h0) {
  const double sj = 1.0 / (1.0 + sqrt(h0)), s2 = sj * sj;
  return fmin(fmax(h * s2, 1e-6), 1e32) / s2;
}

// 3x3 damped normal equations; PoseErrorRPZ / PoseErrorYXY prior (pose_error.hpp:135-190): residual_k = w (x_k - x0_k)
__device__ void icp_step(const IcpArgs& args, IcpDev* dev) {
  if (dev->done) return;
  const double w2 = args.prior_w * args.prior_w;
  double H[6], g[3];
  for (int q = 0; q < 6; ++q) H[q] = fresh(&dev->acc[q]);
  for (int q = 0; q < 3; ++q) g[q] = fresh(&dev->acc[6 + q]);
  double cost = fresh(&dev->acc[9]);
  if (args.prior_w > 0.0) {
    H[0] += w2; H[2] += w2; H[5] += w2;
    for (int q = 0; q < 3; ++q) { const double dxp = dev->x[q] - dev->x0[q]; g[q] += w2 * dxp; cost += 0.5 * w2 * dxp * dxp; }
  }
  dev->cost_cur = cost;
  if (dev->first) { dev->initial_cost = cost; dev->first = 0; dev->h0[0] = H[0]; dev->h0[1] = H[2]; dev->h0[2] = H[5]; if (dev->count_valid) dev->nvalid = (int)fresh(&dev->acc[10]); }
  const double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
  if (gmax <= args.gradient_tolerance || dev->radius < 1e-32) { dev->done = 1; return; }      // top of ceres::Solve's loop: gradient tolerance, smallest trust region
  const double inv_r = 1.0 / dev->radius;
  const double D0 = lm_damping(H[0], dev->h0[0]) * inv_r, D1 = lm_damping(H[2], dev->h0[1]) * inv_r, D2 = lm_damping(H[5], dev->h0[2]) * inv_r;
  // Cholesky of [[a00,.,.],[a10,a11,.],[a20,a21,a22]]
  const double a00 = H[0] + D0, a10 = H[1], a11 = H[2] + D1, a20 = H[3], a21 = H[4], a22 = H[5] + D2;
  const double l00 = sqrt(a00), l10 = a10 / l00, l20 = a20 / l00;
  const double t11 = a11 - l10 * l10;
  const double l11 = sqrt(t11), l21 = (a21 - l20 * l10) / l11;
  const double t22 = a22 - l20 * l20 - l21 * l21;
  const double l22 = sqrt(t22);
  bool ok = a00 > 0.0 && t11 > 0.0 && t22 > 0.0;
  double dx[3] = {0, 0, 0};
  if (ok) {
    const double y0 = -g[0] / l00, y1 = (-g[1] - l10 * y0) / l11, y2 = (-g[2] - l20 * y0 - l21 * y1) / l22;
    dx[2] = y2 / l22; dx[1] = (y1 - l21 * dx[2]) / l11; dx[0] = (y0 - l10 * dx[1] - l20 * dx[2]) / l00;
    ok = isfinite(dx[0]) && isfinite(dx[1]) && isfinite(dx[2]);
  }
  dev->model = ok ? 0.5 * (dx[0] * (D0 * dx[0] - g[0]) + dx[1] * (D1 * dx[1] - g[1]) + dx[2] * (D2 * dx[2] - g[2])) : -1.0;
  for (int q = 0; q < 3; ++q) dev->xc[q] = dev->x[q] + dx[q];
  const double dn = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
  const double xn = sqrt(dev->x[0] * dev->x[0] + dev->x[1] * dev->x[1] + dev->x[2] * dev->x[2]);
  if (ok && dev->model > 0.0 && dn <= args.parameter_tolerance * (xn + args.parameter_tolerance)) dev->done = 1;      // (a VALID step this small: the candidate is not taken)
}

// The candidate pass has just summed H, g and the cost AT xc into accC (k_icp_eval<true>): judge the step; an accepted candidate's sums
// become the current linearisation (no pass at the new x), a rejected or invalid step keeps the old one (no pass at the old x either)
__device__ void icp_decide(const IcpArgs& args, IcpDev* dev, const bool have_j = true) {
  if (dev->done) return;
  double cand = fresh(&dev->accC[9]);
  if (args.prior_w > 0.0) {
    const double w2 = args.prior_w * args.prior_w;
    for (int q = 0; q < 3; ++q) { const double dxp = dev->xc[q] - dev->x0[q]; cand += 0.5 * w2 * dxp * dxp; }
  }
  // ceres::Solve's TrustRegionMinimizer order (declared in oracle/lm.h lm_solve, restated for this 3-unknown problem in oracle/icp.h)
  const bool valid = dev->model > 0.0 && isfinite(cand);
  if (!valid) {
    dev->iters += 1;
    if (++dev->invalid_run >= 5) dev->done = 1;
    else dev->radius *= 0.5;
  } else {
    dev->invalid_run = 0;
    if (fabs(dev->cost_cur - cand) <= args.function_tolerance * dev->cost_cur) dev->done = 1;      // BEFORE the step-quality test; the candidate is not taken
    else {
      dev->iters += 1;
      const double rho = (dev->cost_cur - cand) / dev->model;
      if (rho > args.min_relative_decrease) {
        for (int q = 0; q < 3; ++q) dev->x[q] = dev->xc[q];
        if (have_j) for (int q = 0; q < 10; ++q) dev->acc[q] = fresh(&dev->accC[q]);
        dev->cost_cur = cand;
        dev->successes += 1;
        const double t = 2.0 * rho - 1.0;
        dev->radius = fmin(dev->radius / fmax(1.0 / 3.0, 1.0 - t * t * t), 1e16);
        dev->decrease = 2.0;
      } else { dev->radius = dev->radius / dev->decrease; dev->decrease *= 2.0; }
    }
  }
  if (dev->iters >= args.max_iters || dev->radius < 1e-32) dev->done = 1;
  for (int q = 0; q < 11; ++q) dev->accC[q] = 0.0;
}

// stand-alone forms (problems with no correspondences never launch k_icp_eval)
__global__ void k_icp_step(const IcpArgs args, IcpDev* dev) { icp_step(args, dev); }
__global__ void k_icp_decide_step(const IcpArgs args, IcpDev* dev) { icp_decide(args, dev); icp_step(args, dev); }

// CAND = false: the pass at x that opens a solve (sums into acc, then the first damped step); CAND = true: the pass at the candidate xc of
// every LM iteration (sums into accC, then the decision and — unless the solve ended — the next step).  ONE launch per LM iteration:
// ceres::Solve evaluates the cost at the candidate and, once the step is accepted, the Jacobian at the same point; here both come from
// the same pass (round 3: a Jacobian pass at x and a cost-only pass at xc per iteration, eight launches for four iterations instead of five).
// WITH_J = false (with CAND): the pass of the LAST iteration the options allow — no step can follow it, so the cost alone is summed.
template <bool CAND, bool WITH_J = true>
__device__ __forceinline__ void icp_eval_body(const int bx, const int nbx, int Q, const double* __restrict__ P, const double* __restrict__ PA,
                                              const double* __restrict__ N, const uint8_t* __restrict__ valid,
                                              const IcpArgs& args, IcpDev* __restrict__ dev) {
  __shared__ LidarU U;
  if (dev->done) return;                              // uniform: the flag is only written by single-thread kernels
  if (threadIdx.x == 0) {
    LidarArgs a;
    for (int k = 0; k < 7; ++k) a.Twc1[k] = args.Twc1[k];
    for (int k = 0; k < 6; ++k) a.rpyxyz[k] = dev->rpyxyz[k];
    int i0, i1, i2;
    param_slots(args.mode, i0, i1, i2);
    const double* x = CAND ? dev->xc : dev->x;
    a.rpyxyz[i0] = x[0]; a.rpyxyz[i1] = x[1]; a.rpyxyz[i2] = x[2];
    a.weight = args.weight; a.mode = args.mode;
    derive_lidar(a, U);
  }
  __syncthreads();
  // grid-stride over the correspondences (the grid is capped at kIcpMaxBlocks): sums stay in registers, then ONE set of
  // atomics per workgroup (wave shuffles -> LDS -> wave 0) instead of one per wave — the ten accumulators are single
  // addresses, and ~900 serialised L2 atomics per address were the kernel's whole run time
  double v[11];
#pragma unroll
  for (int q = 0; q < 11; ++q) v[q] = 0.0;
  for (int i = bx * kTI + threadIdx.x; i < Q; i += nbx * kTI) {
    if (valid && !valid[i]) continue;
    if (WITH_J) v[10] += 1.0;
    const double pp[3] = {P[i], P[Q + i], P[2 * Q + i]}, qa[3] = {PA[i], PA[Q + i], PA[2 * Q + i]}, nn[3] = {N[i], N[Q + i], N[2 * Q + i]};
    double r, J[3];
    lidar_point(U, args.mode, pp, qa, nn, r, J);
    double rho;
    const double sc = robust_scale(args.huber, r * r, rho);
    v[9] += 0.5 * rho;
    if (WITH_J) {
      const double rs = sc * r, j0 = sc * J[0], j1 = sc * J[1], j2 = sc * J[2];
      v[0] += j0 * j0; v[1] += j1 * j0; v[2] += j1 * j1; v[3] += j2 * j0; v[4] += j2 * j1; v[5] += j2 * j2;
      v[6] += j0 * rs; v[7] += j1 * rs; v[8] += j2 * rs;
    }
  }
  __shared__ double s_part[kTI / 64][11];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int q = WITH_J ? 0 : 9; q < (WITH_J ? 11 : 10); ++q) {
    const double s = wave_sum(v[q]);                     // DPP row butterflies + 4 readlanes (lvf_internal.hpp)
    if (lane == 0) s_part[wid][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < 11 && (WITH_J || threadIdx.x == 9)) {
    double s = 0.0;
#pragma unroll
    for (int w2 = 0; w2 < kTI / 64; ++w2) s += s_part[w2][threadIdx.x];
    if (s != 0.0) atomicAdd(CAND ? &dev->accC[threadIdx.x] : &dev->acc[threadIdx.x], s);
  }
  // the LAST workgroup to get here runs the scalar tail of the pass (accept / reject of the candidate, then the 3x3 damped solve for the
  // next one): one launch per LM iteration
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned t = atomicAdd(&dev->ticket, 1u);
    if (t == (unsigned)nbx - 1u) {
      __threadfence();
      dev->ticket = 0;
      if (CAND) icp_decide(args, dev, WITH_J);
      if (WITH_J) icp_step(args, dev);
    }
  }
}
template <bool CAND, bool WITH_J>
__global__ __launch_bounds__(kTI) void k_icp_eval(int Q, const double* __restrict__ P, const double* __restrict__ PA,
                                                  const double* __restrict__ N, const uint8_t* __restrict__ valid,
                                                  const IcpArgs args, IcpDev* __restrict__ dev) {
  icp_eval_body<CAND, WITH_J>(blockIdx.x, gridDim.x, Q, P, PA, N, valid, args, dev);
}
static __host__ __device__ inline int icp_blocks(int Q) { const int b = ((Q > 1 ? Q : 1) + kTI - 1) / kTI; return b < kIcpMaxBlocks ? b : kIcpMaxBlocks; }
// table form: blockIdx.y = candidate; a candidate's own workgroup count is what its ticket counts (one workgroup even for Q = 0: the
// scalar tail — the damped 3 x 3 solve / the decision — must run for problems with no correspondences too)
template <bool CAND, bool WITH_J>
__global__ __launch_bounds__(kTI) void k_icp_eval_b(const IcpJob* __restrict__ jobs, SmDev* __restrict__ devs, int sub) {
  SmDev& D = devs[blockIdx.y];
  if (!D.has[sub]) return;
  const IcpJob& J = jobs[2 * blockIdx.y + sub];
  const int nbx = icp_blocks(J.Q);
  if ((int)blockIdx.x >= nbx) return;
  icp_eval_body<CAND, WITH_J>(blockIdx.x, nbx, J.Q, J.P, J.PA, J.N, J.valid, J.args, &D.icp);
}

}  // namespace lvf

namespace lvf {
static IcpArgs make_args(const double* Twc1, const lvf_icp_options* opt) {
  IcpArgs a;
  std::memcpy(a.Twc1, Twc1, sizeof(a.Twc1));
  a.weight = opt->weight; a.huber = opt->huber_a; a.prior_w = opt->prior_weight;
  a.function_tolerance = 1e-6; a.gradient_tolerance = 1e-10; a.parameter_tolerance = 1e-8; a.min_relative_decrease = 1e-3;
  a.mode = opt->mode; a.max_iters = opt->max_num_iterations;
  return a;
}
int launch_icp_eval_batch(hipStream_t q, const IcpJob* jobs, SmDev* devs, int n, int sub, int max_Q, bool candidate, bool last) {
  if (n <= 0) return LVF_OK;
  const dim3 g(icp_blocks(max_Q), n);
  if (!candidate) hipLaunchKernelGGL((k_icp_eval_b<false, true>), g, dim3(kTI), 0, q, jobs, devs, sub);
  else if (last) hipLaunchKernelGGL((k_icp_eval_b<true, false>), g, dim3(kTI), 0, q, jobs, devs, sub);
  else hipLaunchKernelGGL((k_icp_eval_b<true, true>), g, dim3(kTI), 0, q, jobs, devs, sub);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}
static void init_dev(IcpDev& h, int mode, const double* rpyxyz) {
  std::memset(&h, 0, sizeof(h));
  const int i0 = mode == 0 ? 1 : 0, i1 = mode == 0 ? 2 : 3, i2 = mode == 0 ? 5 : 4;
  h.x[0] = h.x0[0] = rpyxyz[i0]; h.x[1] = h.x0[1] = rpyxyz[i1]; h.x[2] = h.x0[2] = rpyxyz[i2];
  for (int k = 0; k < 6; ++k) h.rpyxyz[k] = rpyxyz[k];
  h.radius = 1e4; h.decrease = 2.0; h.first = 1;
}
// the device-resident LM loop over correspondences P | PA | N (SoA [3][Q]); valid may be null (all rows count)
static int run_lm(hipStream_t q, int Q, const double* P, const double* PA, const double* N, const uint8_t* valid, const IcpArgs& a,
                  IcpDev* dev, IcpDev* host_out) {
  const int grid = std::min((std::max(Q, 1) + kTI - 1) / kTI, kIcpMaxBlocks);
  // the pass at x (linearisation + first step), then ONE pass per LM iteration at its candidate (decision + next step)
  // (the last iteration's pass sums the cost alone: no step can follow it)
  if (Q > 0) hipLaunchKernelGGL((k_icp_eval<false, true>), dim3(grid), dim3(kTI), 0, q, Q, P, PA, N, valid, a, dev);
  else hipLaunchKernelGGL(k_icp_step, dim3(1), dim3(1), 0, q, a, dev);
  for (int it = 0; it < a.max_iters; ++it) {
    if (Q <= 0) hipLaunchKernelGGL(k_icp_decide_step, dim3(1), dim3(1), 0, q, a, dev);
    else if (it + 1 == a.max_iters) hipLaunchKernelGGL((k_icp_eval<true, false>), dim3(grid), dim3(kTI), 0, q, Q, P, PA, N, valid, a, dev);
    else hipLaunchKernelGGL((k_icp_eval<true, true>), dim3(grid), dim3(kTI), 0, q, Q, P, PA, N, valid, a, dev);
  }
  LVF_HIP(hipGetLastError());
  LVF_HIP(hipMemcpyAsync(host_out, dev, sizeof(IcpDev), hipMemcpyDeviceToHost, q));
  LVF_HIP(hipStreamSynchronize(q));
  return LVF_OK;
}
}  // namespace lvf

using namespace lvf;

extern "C" int lvf_icp_solve(lvf_map* m, lvf_scan* sc, const double* map_pose, const double* frame_pose, double* rpyxyz,
                             const lvf_icp_options* opt, lvf_icp_summary* summary) {
  LVF_REQUIRE(m && sc && map_pose && frame_pose && rpyxyz && opt && summary, "lvf_icp_solve: null argument");
  LVF_REQUIRE(m->ctx == sc->ctx, "lvf_icp_solve: map and scan belong to different contexts");
  LVF_REQUIRE(opt->mode == 0 || opt->mode == 1, "lvf_icp_solve: mode must be 0 (ground/RPZ) or 1 (surf/YXY)");
  LVF_REQUIRE(opt->thr > 0.0f && opt->max_num_iterations >= 0, "lvf_icp_solve: bad options");
  LVF_TRY(lvf::enter(m->ctx));
  hipStream_t q = m->ctx->stream;
  const int Q = sc->Q;
  std::memset(summary, 0, sizeof(*summary));
  // 1. association at the frame's current pose (association.cpp:287-301 / :345-359) + 2. the correspondences (:303-314), one launch
  if (sc->corr.n < (size_t)9 * std::max(Q, 1)) LVF_TRY(sc->corr.alloc((size_t)9 * std::max(Q, 1)));
  LVF_TRY(launch_knn3_build(m, sc, frame_pose, opt->thr, sc->corr.p));
  // device solver state
  if (!sc->icp_dev.p) LVF_TRY(sc->icp_dev.alloc(sizeof(IcpDev)));
  double* P = sc->corr.p; double* PA = P + (size_t)3 * Q; double* N = PA + (size_t)3 * Q;
  // (the solver state travels through a pinned mirror owned by the scan: from a stack variable both copies went through the runtime's
  // pageable path, a pin / unpin of the page per solve)
  LVF_TRY(sc->icp_host.reserve(sizeof(IcpDev)));
  IcpDev& h = *reinterpret_cast<IcpDev*>(sc->icp_host.p);
  init_dev(h, opt->mode, rpyxyz);
  h.count_valid = 1;                         // (the accepted queries are counted by the first linearisation pass)
  const int i0 = opt->mode == 0 ? 1 : 0, i1 = opt->mode == 0 ? 2 : 3, i2 = opt->mode == 0 ? 5 : 4;
  IcpDev* dev = reinterpret_cast<IcpDev*>(sc->icp_dev.p);
  LVF_HIP(hipMemcpyAsync(dev, &h, sizeof(h), hipMemcpyHostToDevice, q));
  IcpArgs a = make_args(map_pose, opt);
  // 3. LM iterations, all on device
  LVF_TRY(run_lm(q, Q, P, PA, N, sc->valid.p, a, dev, &h));
  rpyxyz[i0] = h.x[0]; rpyxyz[i1] = h.x[1]; rpyxyz[i2] = h.x[2];
  summary->initial_cost = h.initial_cost; summary->final_cost = h.cost_cur;
  summary->num_residual_blocks = h.nvalid + (opt->prior_weight > 0.0 ? 1 : 0);
  summary->num_iterations = h.iters; summary->num_successful_steps = h.successes;
  return LVF_OK;
}

// adapt::Solve for a problem made of LidarPlaneErrorRPZ/YXY blocks (+ optionally one PoseErrorRPZ/YXY prior) that were
// built by the CALLER (association.cpp:303-333 / :361-384 left on the host): the same device-resident 3-DoF LM as
// lvf_icp_solve, run over the batch's own correspondences.  mode, weight and Twc1 come from the batch.
extern "C" int lvf_lidar_solve(lvf_batch* b, double* rpyxyz, const lvf_icp_options* opt, lvf_icp_summary* summary) {
  LVF_REQUIRE(b && rpyxyz && opt && summary, "lvf_lidar_solve: null argument");
  LVF_REQUIRE(b->kind == LVF_K_LIDAR, "lvf_lidar_solve: not a lidar-plane batch");
  LVF_REQUIRE(opt->max_num_iterations >= 0, "lvf_lidar_solve: bad options");
  LVF_TRY(lvf::enter(b->ctx));
  hipStream_t q = b->ctx->stream;
  std::memset(summary, 0, sizeof(*summary));
  if (!b->icp_dev.p) LVF_TRY(b->icp_dev.alloc(sizeof(IcpDev)));
  lvf_icp_options o = *opt;
  o.mode = b->lidar_mode; o.weight = b->lidar_weight;
  LVF_TRY(b->icp_host.reserve(sizeof(IcpDev)));
  IcpDev& h = *reinterpret_cast<IcpDev*>(b->icp_host.p);
  init_dev(h, o.mode, rpyxyz);
  h.nvalid = b->n;
  IcpDev* dev = reinterpret_cast<IcpDev*>(b->icp_dev.p);
  LVF_HIP(hipMemcpyAsync(dev, &h, sizeof(h), hipMemcpyHostToDevice, q));
  const IcpArgs a = make_args(b->Twc1, &o);
  LVF_TRY(run_lm(q, b->n, b->lp.p, b->lpa.p, b->lnrm.p, nullptr, a, dev, &h));
  const int i0 = o.mode == 0 ? 1 : 0, i1 = o.mode == 0 ? 2 : 3, i2 = o.mode == 0 ? 5 : 4;
  rpyxyz[i0] = h.x[0]; rpyxyz[i1] = h.x[1]; rpyxyz[i2] = h.x[2];
  summary->initial_cost = h.initial_cost; summary->final_cost = h.cost_cur;
  summary->num_residual_blocks = b->n + (o.prior_weight > 0.0 ? 1 : 0);
  summary->num_iterations = h.iters; summary->num_successful_steps = h.successes;
  return LVF_OK;
}

// PoseErrorRPZ / PoseErrorYXY <3,1,1,1> (pose_error.hpp:135-190) as a stand-alone CostFunction::Evaluate: three linear
// residuals.  Inside lvf_icp_solve / lvf_lidar_solve the prior is folded into k_icp_step; this entry point exists so the
// per-block ceres::CostFunction surface of the adapter has a device evaluation for every functor on the path.
namespace lvf {
__global__ void k_prior3(int mode, const double w, const double t0, const double t1, const double t2, const double x0, const double x1,
                         const double x2, double* __restrict__ out /* r[3] | J0[3] | J1[3] | J2[3] */) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // parameter order (p, r, z) / (Y, x, y); residual order (r, p, z) / (Y, x, y)
  for (int k = 0; k < 12; ++k) out[k] = 0.0;
  if (mode == 0) {
    out[0] = w * (x1 - t1); out[1] = w * (x0 - t0); out[2] = w * (x2 - t2);
    out[3 + 1] = w;   // d r1 / d p
    out[6 + 0] = w;   // d r0 / d r
    out[9 + 2] = w;   // d r2 / d z
  } else {
    out[0] = w * (x0 - t0); out[1] = w * (x1 - t1); out[2] = w * (x2 - t2);
    out[3 + 0] = w; out[6 + 1] = w; out[9 + 2] = w;
  }
}
}  // namespace lvf
extern "C" int lvf_prior3_evaluate(lvf_ctx* ctx, int mode, const double* target3, double weight, const double* x3, double* residuals3,
                                   double* jacobians9) {
  LVF_REQUIRE(ctx && target3 && x3 && residuals3, "lvf_prior3_evaluate: null argument");
  LVF_REQUIRE(mode == 0 || mode == 1, "lvf_prior3_evaluate: mode must be 0 (RPZ) or 1 (YXY)");
  LVF_TRY(lvf::enter(ctx));
  DevBuf<double> out;
  LVF_TRY(out.alloc(12));
  hipLaunchKernelGGL(k_prior3, dim3(1), dim3(1), 0, ctx->stream, mode, weight, target3[0], target3[1], target3[2], x3[0], x3[1], x3[2], out.p);
  LVF_HIP(hipGetLastError());
  double h[12];
  LVF_HIP(hipMemcpyAsync(h, out.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  LVF_HIP(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 3; ++k) residuals3[k] = h[k];
  if (jacobians9) for (int k = 0; k < 9; ++k) jacobians9[k] = h[3 + k];
  return LVF_OK;
}
