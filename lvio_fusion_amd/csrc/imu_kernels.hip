// imu_kernels.hip — ImuError factor (15 residuals, 8 parameter blocks) and device pre-integration.
//
// Replaces ImuError::Evaluate                       include/lvio_fusion/ceres/imu_error.hpp:17-113
//          Preintegration::Evaluate                 src/preintegration.cpp:144-165
//          Preintegration::{MidPointIntegration,Propagate}   src/preintegration.cpp:30-127
// of src/lvio_fusion/.  The Jacobians are the reference's VINS-style tangent blocks placed in ambient
// pose columns 0-2 (rotation, imu::O_PR) and 4-6 (translation, imu::O_PT) with column 3 zero — reproduced, not
// "fixed" (SURVEY.md §7 hard parts).
//
// Only ~49 factors exist per window, so this path is latency- not bandwidth-bound: one wave64 per factor, the
// 15x32 pre-weighting Jacobian M and sqrt_info S live in LDS, all 64 lanes share the 15x15 x 15x32 product.
// sqrt_info = LLT(cov^-1).L^T depends only on the pre-integration, so it is factorised ONCE at batch creation
// (k_imu_sqrt_info) instead of on every Evaluate as the reference does (imu_error.hpp:32).
#include "imu_eval.hpp"

namespace lvf {

// --------------------------------------------------------------------------------- sqrt_info (once per batch)
// Declared algorithm (identical to the oracle's, oracle/imu.h): partial-pivot LU inverse of cov, then lower
// Cholesky of the inverse reading the lower triangle; sqrt_info = L^T.  cov has condition number ~1e8+, so the
// operation ORDER matters for 1e-6 parity: contraction is disabled here.
// One WORKGROUP per factor (the first version ran one THREAD per factor with the three 15x15 work arrays in scratch: 0.47 ms for
// the 49 factors of a window, paid at every batch creation and every persistent-window tick).  Every matrix element goes
// through exactly the same sequence of operations as in the scalar algorithm — the elimination step k updates all (i, j) > k
// independently, the 15 columns of the inverse are independent, Cholesky column j is independent over rows i — so the result
// is bit-identical and only the loop nests that were independent run side by side.
__device__ __forceinline__ void imu_sqrt_info_body(const int f, const double* __restrict__ pre, double* __restrict__ sqrt_info) {
#pragma clang fp contract(off)
  __shared__ double LU[225], X[225], L[225];
  __shared__ int piv[15];
  const int tid = threadIdx.x;
  const double* cov = pre + (size_t)f * kPre + OFF_COV;
  if (tid < 225) { LU[tid] = cov[tid]; L[tid] = 0.0; }
  __syncthreads();
  const int ti = tid >> 4, tj = tid & 15;          // element (ti, tj) of a 16 x 16 thread grid
  for (int k = 0; k < 15; ++k) {
    if (tid < 64) {
      // partial pivot across the lanes of wave 0: the FIRST row i >= k with the largest |LU[i][k]| — what the scalar loop (keep the
      // earlier row unless strictly larger) finds; as that loop on one lane it was 15 dependent LDS reads per step
      const bool in = tid >= k && tid < 15;
      const double a = in ? fabs(LU[15 * tid + k]) : -1.0;
      double m = a;
      for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
      const unsigned long long hit = __ballot(in && a == m);
      if (tid == 0) piv[k] = hit ? (int)__ffsll((long long)hit) - 1 : k;
    }
    __syncthreads();
    const int p = piv[k];
    if (p != k && tid < 15) { const double t = LU[15 * k + tid]; LU[15 * k + tid] = LU[15 * p + tid]; LU[15 * p + tid] = t; }
    __syncthreads();
    double l = 0.0;
    const bool row = ti > k && ti < 15;
    if (row) l = LU[15 * ti + k] / LU[15 * k + k];
    __syncthreads();                               // every thread of a row has read LU[i][k] before tj == k overwrites it
    if (row) {
      if (tj == k) LU[15 * ti + k] = l;
      else if (tj > k && tj < 15) LU[15 * ti + tj] -= l * LU[15 * k + tj];
    }
    __syncthreads();
  }
  if (tid < 15) {
    const int c = tid;
    double b[15];
    // the row swaps applied to e_c only move its single 1: track where it ends up (indexing b[] with piv[k] put the array in scratch)
    int pos = c;
    for (int k = 0; k < 15; ++k) { const int pk = piv[k]; pos = (pos == k) ? pk : (pos == pk ? k : pos); }
#pragma unroll
    for (int i = 0; i < 15; ++i) b[i] = (i == pos) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      double s = b[i];
#pragma unroll
      for (int j = 0; j < i; ++j) s -= LU[15 * i + j] * b[j];
      b[i] = s;
    }
#pragma unroll
    for (int i = 14; i >= 0; --i) {
      double s = b[i];
#pragma unroll
      for (int j = i + 1; j < 15; ++j) s -= LU[15 * i + j] * b[j];
      b[i] = s / LU[15 * i + i];
    }
#pragma unroll
    for (int i = 0; i < 15; ++i) X[15 * i + c] = b[i];
  }
  __syncthreads();
  for (int j = 0; j < 15; ++j) {
    if (tid == 0) {
      double d = X[15 * j + j];
      for (int k = 0; k < j; ++k) d -= L[15 * j + k] * L[15 * j + k];
      L[15 * j + j] = sqrt(d);
    }
    __syncthreads();
    if (tid > j && tid < 15) {
      const int i = tid;
      double s_ = X[15 * i + j];
      for (int k = 0; k < j; ++k) s_ -= L[15 * i + k] * L[15 * j + k];
      L[15 * i + j] = s_ / L[15 * j + j];
    }
    __syncthreads();
  }
  double* S = sqrt_info + (size_t)f * 225;
  if (tid < 225) { const int i = tid / 15, j = tid - 15 * i; S[15 * i + j] = L[15 * j + i]; }
}
__global__ __launch_bounds__(256) void k_imu_sqrt_info(int n, const double* __restrict__ pre, double* __restrict__ sqrt_info) {
  if ((int)blockIdx.x >= n) return;
  imu_sqrt_info_body(blockIdx.x, pre, sqrt_info);
}
// The persistent window's form: a factor whose pre-integration has not changed since the previous tick (src[f] = the slot its matrix had
// then, in `prev`) is COPIED; only the others are factored — in steady state the one pair the newest keyframe brought (imu_error.hpp:32
// re-derives the matrix on every Evaluate; round 3 re-factored all of a window's pairs every tick).
__global__ __launch_bounds__(256) void k_imu_sqrt_info_cached(int n, const double* __restrict__ pre, double* __restrict__ sqrt_info, const int* __restrict__ src,
                                                              const double* __restrict__ prev) {
  const int f = blockIdx.x;
  if (f >= n) return;
  const int s = src[f];
  if (s >= 0) {                                      // (workgroup-uniform)
    if (threadIdx.x < 225) sqrt_info[(size_t)f * 225 + threadIdx.x] = prev[(size_t)s * 225 + threadIdx.x];
    return;
  }
  imu_sqrt_info_body(f, pre, sqrt_info);
}

// --------------------------------------------------------------------------------- ImuError evaluate
// column layout of the concatenated 15x32 Jacobian: [pose_i 0..6 | v_i 7..9 | ba_i 10..12 | bg_i 13..15 |
//                                                     pose_j 16..22 | v_j 23..25 | ba_j 26..28 | bg_j 29..31]
template <bool WITH_J>
__device__ __forceinline__ void imu_body(const int bx, const ImuArgs& A) {
  if (A.done && *A.done) return;
  const int n = A.n;
  double* __restrict__ res = A.res; double* __restrict__ cost_stripes = A.cost_stripes;
  const ImuOut& out = A.out;
  if (bx >= n) {            // extra workgroups: clear the solver's accumulators (independent of the factors)
    if (bx >= n + A.zero_wgs) return;
    const ZeroList& zero = A.zero;
    const unsigned long long t = (unsigned long long)(bx - n) * 64 + threadIdx.x, nt = (unsigned long long)A.zero_wgs * 64;
    // (static indices only: a run-time index into the by-value pointer table would put it in scratch memory)
#pragma unroll
    for (int a = 0; a < kZeroListMax; ++a) {
      if (a >= zero.count) break;
      double* p = zero.p[a];
      const unsigned long long cnt = zero.n[a], n2 = cnt / 2;
      double2* p2 = reinterpret_cast<double2*>(p);
      for (unsigned long long i = t; i < n2; i += nt) p2[i] = make_double2(0.0, 0.0);
      if ((cnt & 1) && t == 0) p[cnt - 1] = 0.0;
    }
    return;
  }
  __shared__ double sS[225];
  __shared__ double sM[15 * 32];
  __shared__ double sr0[15];
  __shared__ double scost[15];
  const int f = bx;
  const int lane = threadIdx.x;
  imu_stage<WITH_J>(f, lane, A.sqrt_info, sS, sM);
  __syncthreads();
  if (lane == 0) imu_raw<WITH_J>(f, A.pre + (size_t)f * kPre, A.kf_i, A.kf_j, A.poses, A.vel, A.ba, A.bg, sr0, sM);
  __syncthreads();
  if (lane < 15) {
    const double s = imu_weighted_residual(lane, sS, sr0);
    res[(size_t)f * 15 + lane] = s;
    scost[lane] = 0.5 * s * s;
  }
  // optional: the factor's cost 1/2 |r|^2 straight into the solver's striped accumulator (one atomic per factor) — a separate
  // reduction kernel over 15 n values is a 6 us launch for no work
  if (cost_stripes) {
    __syncthreads();
    if (lane == 0) {
      double c = 0.0;
      for (int k = 0; k < 15; ++k) c += scost[k];
      atomicAdd(cost_stripes + (f & 31), c);
    }
  }
  if (WITH_J) {
    for (int e = lane; e < 480; e += 64) {
      const int r = e >> 5, c = e & 31;
      const double s = imu_weighted_jacobian(e, sS, sM);
      int blk, cc, width;
      if (c < 7) { blk = 0; cc = c; width = 7; }
      else if (c < 16) { blk = 1 + (c - 7) / 3; cc = (c - 7) % 3; width = 3; }
      else if (c < 23) { blk = 4; cc = c - 16; width = 7; }
      else { blk = 5 + (c - 23) / 3; cc = (c - 23) % 3; width = 3; }
      out.j[blk][(size_t)f * 15 * width + r * width + cc] = s;
    }
  }
}

template <bool WITH_J>
__global__ __launch_bounds__(64) void k_imu(ImuArgs a) { imu_body<WITH_J>(blockIdx.x, a); }
// one launch for a batch of windows: blockIdx.y selects the window's argument block
template <bool WITH_J>
__global__ __launch_bounds__(64) void k_imu_b(const ImuArgs* __restrict__ table) { imu_body<WITH_J>(blockIdx.x, table[blockIdx.y]); }

// --------------------------------------------------------------------------------- pre-integration (K5)
// One workgroup per keyframe pair; the chain over IMU samples is sequential (each sample needs the previous delta_q,
// jacobian and covariance) but the 15x15 products  jac <- F jac,  cov <- F cov F^T + V N V^T  are spread over the
// 256 threads with F, V, jac, cov resident in LDS.  Mid-point rule and F/V blocks: preintegration.cpp:30-127.
__device__ __forceinline__ void mm3(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void skew9(const double v[3], double S[9]) {
  S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
__global__ __launch_bounds__(256) void k_preintegrate(const int* __restrict__ offset, const double* __restrict__ samples,
                                                      const double* __restrict__ acc0_, const double* __restrict__ gyr0_,
                                                      const double* __restrict__ ba_, const double* __restrict__ bg_,
                                                      double acc_n, double gyr_n, double acc_w, double gyr_w,
                                                      double* __restrict__ out) {
  __shared__ double sJ[225], sC[225], sF[225], sV[270], sT[225], sN[18], sM[56];
  __shared__ double sState[24];   // dp 0..2, dq 3..6 (x,y,z,w), dv 7..9, acc0 10..12, gyr0 13..15, sum_dt 16
  const int f = blockIdx.x, tid = threadIdx.x;
  const double* lba = ba_ + 3 * f; const double* lbg = bg_ + 3 * f;
  if (tid < 225) { sJ[tid] = (tid % 16 == 0) ? 1.0 : 0.0; sC[tid] = 0.0; }
  if (tid < 18) { const int b = tid / 3; sN[tid] = (b == 0 || b == 2) ? acc_n * acc_n : (b == 1 || b == 3) ? gyr_n * gyr_n : (b == 4 ? acc_w * acc_w : gyr_w * gyr_w); }
  if (tid == 0) {
    for (int k = 0; k < 3; ++k) { sState[k] = 0.0; sState[7 + k] = 0.0; sState[10 + k] = acc0_[3 * f + k]; sState[13 + k] = gyr0_[3 * f + k]; }
    sState[3] = sState[4] = sState[5] = 0.0; sState[6] = 1.0; sState[16] = 0.0;
  }
  for (int k = tid; k < 225; k += 256) sF[k] = 0.0;      // the structural zeros of F (15 x 15) and V (15 x 18): one lane clearing 495 LDS words per
  for (int k = tid; k < 270; k += 256) sV[k] = 0.0;      // sample was a third of the serial part of a step
  __syncthreads();
  for (int sidx = offset[f]; sidx < offset[f + 1]; ++sidx) {
    if (tid == 0) {
      const double* smp = samples + 7 * sidx;
      const double dt = smp[0];
      const double* acc1 = smp + 1; const double* gyr1 = smp + 4;
      double a0b[3], a1b[3], ug[3];
      for (int k = 0; k < 3; ++k) { a0b[k] = sState[10 + k] - lba[k]; a1b[k] = acc1[k] - lba[k]; ug[k] = 0.5 * (sState[13 + k] + gyr1[k]) - lbg[k]; }
      const Qd dq{sState[3], sState[4], sState[5], sState[6]};
      double ua0[3], ua1[3];
      qrot(dq, a0b, ua0);
      const Qd rq = qmul(dq, Qd{ug[0] * dt / 2, ug[1] * dt / 2, ug[2] * dt / 2, 1.0});
      qrot(rq, a1b, ua1);
      double R0[9], R1[9], Rw[9], Ra0[9], Ra1[9], ImRw[9], R0a0[9], R1a1[9], R1a1I[9];
      qmat(dq, R0); qmat(rq, R1); skew9(ug, Rw); skew9(a0b, Ra0); skew9(a1b, Ra1);
      for (int k = 0; k < 9; ++k) ImRw[k] = ((k % 4 == 0) ? 1.0 : 0.0) - Rw[k] * dt;
      mm3(R0, Ra0, R0a0); mm3(R1, Ra1, R1a1); mm3(R1a1, ImRw, R1a1I);
      // the 3 x 3 pieces of F and V go to LDS; nine lanes place them below (one lane writing all ~220 entries was most of what was
      // left of the serial part of a step)
      for (int k = 0; k < 9; ++k) { sM[k] = R0[k]; sM[9 + k] = R1[k]; sM[18 + k] = R0a0[k]; sM[27 + k] = R1a1[k]; sM[36 + k] = R1a1I[k]; sM[45 + k] = ImRw[k]; }
      sM[54] = dt;
      // state update (Propagate tail, preintegration.cpp:116-126): delta_q is re-normalised
      for (int k = 0; k < 3; ++k) {
        const double ua = 0.5 * (ua0[k] + ua1[k]);
        const double np = sState[k] + sState[7 + k] * dt + 0.5 * ua * dt * dt;
        const double nv = sState[7 + k] + ua * dt;
        sState[k] = np; sState[7 + k] = nv; sState[10 + k] = acc1[k]; sState[13 + k] = gyr1[k];
      }
      const double nq = sqrt(rq.x * rq.x + rq.y * rq.y + rq.z * rq.z + rq.w * rq.w);
      sState[3] = rq.x / nq; sState[4] = rq.y / nq; sState[5] = rq.z / nq; sState[6] = rq.w / nq;
      sState[16] += dt;
    }
    __syncthreads();
    if (tid < 9) {      // (sF / sV were cleared once before the loop: the entries assigned here are the same set for every sample)
      const int e = tid, i = e / 3, j = e % 3;
      const double I = (i == j) ? 1.0 : 0.0;
      const double dt = sM[54];
      const double R0e = sM[e], R1e = sM[9 + e], R0a0e = sM[18 + e], R1a1e = sM[27 + e], R1a1Ie = sM[36 + e], ImRwe = sM[45 + e];
#define FF(r, c) sF[(r) * 15 + (c)]
#define VV(r, c) sV[(r) * 18 + (c)]
      FF(i, j) = I;
      FF(i, 3 + j) = -0.25 * R0a0e * dt * dt + -0.25 * R1a1Ie * dt * dt;
      FF(i, 6 + j) = I * dt;
      FF(i, 9 + j) = -0.25 * (R0e + R1e) * dt * dt;
      FF(i, 12 + j) = -0.25 * R1a1e * dt * dt * -dt;
      FF(3 + i, 3 + j) = ImRwe;
      FF(3 + i, 12 + j) = -1.0 * I * dt;
      FF(6 + i, 3 + j) = -0.5 * R0a0e * dt + -0.5 * R1a1Ie * dt;
      FF(6 + i, 6 + j) = I;
      FF(6 + i, 9 + j) = -0.5 * (R0e + R1e) * dt;
      FF(6 + i, 12 + j) = -0.5 * R1a1e * dt * -dt;
      FF(9 + i, 9 + j) = I;
      FF(12 + i, 12 + j) = I;
      VV(i, j) = 0.25 * R0e * dt * dt;
      VV(i, 3 + j) = 0.25 * -R1a1e * dt * dt * 0.5 * dt;
      VV(i, 6 + j) = 0.25 * R1e * dt * dt;
      VV(i, 9 + j) = VV(i, 3 + j);
      VV(3 + i, 3 + j) = 0.5 * I * dt;
      VV(3 + i, 9 + j) = 0.5 * I * dt;
      VV(6 + i, j) = 0.5 * R0e * dt;
      VV(6 + i, 3 + j) = 0.5 * -R1a1e * dt * 0.5 * dt;
      VV(6 + i, 6 + j) = 0.5 * R1e * dt;
      VV(6 + i, 9 + j) = VV(6 + i, 3 + j);
      VV(9 + i, 12 + j) = I * dt;
      VV(12 + i, 15 + j) = I * dt;
#undef FF
#undef VV
    }
    __syncthreads();
    double nj = 0.0, fc = 0.0;
    const int r = tid / 15, c = tid % 15;
    if (tid < 225) {
      for (int k = 0; k < 15; ++k) { nj += sF[r * 15 + k] * sJ[k * 15 + c]; fc += sF[r * 15 + k] * sC[k * 15 + c]; }
      sT[tid] = fc;
    }
    __syncthreads();
    double nc = 0.0;
    if (tid < 225) {
      double a = 0.0, b = 0.0;
      for (int k = 0; k < 15; ++k) a += sT[r * 15 + k] * sF[c * 15 + k];
      for (int k = 0; k < 18; ++k) b += sV[r * 18 + k] * sN[k] * sV[c * 18 + k];
      nc = a + b;
    }
    __syncthreads();
    if (tid < 225) { sJ[tid] = nj; sC[tid] = nc; }
    __syncthreads();
  }
  double* o = out + (size_t)f * kPre;
  if (tid == 0) {
    o[OFF_SUMDT] = sState[16];
    for (int k = 0; k < 3; ++k) { o[OFF_LBA + k] = lba[k]; o[OFF_LBG + k] = lbg[k]; o[OFF_DP + k] = sState[k]; o[OFF_DV + k] = sState[7 + k]; }
    for (int k = 0; k < 4; ++k) o[OFF_DQ + k] = sState[3 + k];
  }
  if (tid < 225) { o[OFF_JAC + tid] = sJ[tid]; o[OFF_COV + tid] = sC[tid]; }
}

int launch_imu_sqrt_info(lvf_batch* b) {
  if (b->n == 0) return LVF_OK;
  hipLaunchKernelGGL(k_imu_sqrt_info, dim3(b->n), dim3(256), 0, b->ctx->stream, b->n, b->pre.p, b->sqrt_info.p);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}
int launch_imu_sqrt_info_cached(lvf_batch* b, const int* src_dev, const double* prev_dev) {
  if (b->n == 0) return LVF_OK;
  hipLaunchKernelGGL(k_imu_sqrt_info_cached, dim3(b->n), dim3(256), 0, b->ctx->stream, b->n, b->pre.p, b->sqrt_info.p, src_dev, prev_dev);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

void fill_imu_args(const lvf_batch* b, const double* poses, const double* vel, const double* ba, const double* bg, double* cost_stripes,
                   const ZeroList* zero, const int* done, ImuArgs* o) {
  o->n = b->n; o->pre = b->pre.p; o->sqrt_info = b->sqrt_info.p; o->kf_i = b->idx_a.p; o->kf_j = b->idx_b.p;
  o->poses = poses; o->vel = vel; o->ba = ba; o->bg = bg; o->res = b->res.p;
  for (int k = 0; k < 8; ++k) o->out.j[k] = b->jac[k].p;
  o->cost_stripes = cost_stripes;
  o->zero = zero ? *zero : ZeroList{};
  o->zero_wgs = zero ? kImuZeroWgs : 0;
  o->done = done;
}
int launch_imu_args(hipStream_t s, const ImuArgs& a, bool want_j) {
  if (a.n + a.zero_wgs == 0) return LVF_OK;
  if (want_j) hipLaunchKernelGGL(k_imu<true>, dim3(a.n + a.zero_wgs), dim3(64), 0, s, a);
  else hipLaunchKernelGGL(k_imu<false>, dim3(a.n + a.zero_wgs), dim3(64), 0, s, a);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}
int launch_imu_table(hipStream_t s, const ImuArgs* table_dev, int n_windows, int max_blocks, bool want_j) {
  if (max_blocks == 0 || n_windows == 0) return LVF_OK;
  if (want_j) hipLaunchKernelGGL(k_imu_b<true>, dim3(max_blocks, n_windows), dim3(64), 0, s, table_dev);
  else hipLaunchKernelGGL(k_imu_b<false>, dim3(max_blocks, n_windows), dim3(64), 0, s, table_dev);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}
int launch_imu(lvf_batch* b, const lvf_state* st, bool want_j, double* cost_stripes, const ZeroList* zero) {
  if (b->n == 0) return LVF_OK;
  ImuArgs a;
  fill_imu_args(b, st->poses.p, st->vel.p, st->ba.p, st->bg.p, cost_stripes, zero, nullptr, &a);
  return launch_imu_args(b->ctx->stream, a, want_j);
}

}  // namespace lvf

using namespace lvf;
extern "C" int lvf_preintegrate(lvf_ctx* ctx, int n, const int32_t* offset, const double* samples, const double* acc0, const double* gyr0,
                                const double* ba, const double* bg, const double* noise4, lvf_preint* out) {
  LVF_REQUIRE(ctx && noise4 && (n == 0 || (offset && acc0 && gyr0 && ba && bg && out)), "lvf_preintegrate: null argument");
  LVF_REQUIRE(n >= 0, "lvf_preintegrate: negative count");
  if (n == 0) return LVF_OK;
  LVF_REQUIRE(offset[0] == 0, "lvf_preintegrate: offset[0] must be 0");
  for (int k = 0; k < n; ++k) LVF_REQUIRE(offset[k + 1] >= offset[k], "lvf_preintegrate: offsets must be non-decreasing");
  const int ns = offset[n];
  LVF_REQUIRE(ns == 0 || samples, "lvf_preintegrate: samples is null");
  LVF_TRY(lvf::enter(ctx));
  hipStream_t q = ctx->stream;
  DevBuf<int> d_off; DevBuf<double> d_s, d_a0, d_g0, d_ba, d_bg, d_out;
  LVF_TRY(d_off.upload(offset, n + 1, q)); LVF_TRY(d_s.upload(samples, (size_t)7 * ns, q));
  LVF_TRY(d_a0.upload(acc0, (size_t)3 * n, q)); LVF_TRY(d_g0.upload(gyr0, (size_t)3 * n, q));
  LVF_TRY(d_ba.upload(ba, (size_t)3 * n, q)); LVF_TRY(d_bg.upload(bg, (size_t)3 * n, q));
  LVF_TRY(d_out.alloc((size_t)kPre * n));
  hipLaunchKernelGGL(k_preintegrate, dim3(n), dim3(256), 0, q, d_off.p, d_s.p, d_a0.p, d_g0.p, d_ba.p, d_bg.p, noise4[0], noise4[1], noise4[2],
                     noise4[3], d_out.p);
  LVF_HIP(hipGetLastError());
  LVF_HIP(hipMemcpyAsync(out, d_out.p, (size_t)kPre * n * 8, hipMemcpyDeviceToHost, q));
  LVF_HIP(hipStreamSynchronize(q));
  return LVF_OK;
}
