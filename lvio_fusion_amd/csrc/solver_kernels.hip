// solver_kernels.hip — the sliding-window BA problem on device: fused robustified linearisation, Schur
// elimination of the 1x1 inverse-depth blocks (MFMA f64), dense Cholesky of the reduced camera system, step
// evaluation.  Replaces what ceres::Solve does under adapt::Solve for Backend::Optimize
// (src/lvio_fusion/include/lvio_fusion/adapt/problem.h:83-88; src/lvio_fusion/src/backend.cpp:96-183, 206-211).
//
// Unknown ordering (reduced system, d = 15 n_kf):  [ pose tangent 6 x n_kf | (v, ba, bg) 9 x n_kf ].
// H = [B E^T; E C], C diagonal (one inverse depth per landmark).  LM step: (H + D^2/radius) dx = -g with
// D^2 = clamp(diag H, 1e-6, 1e32).  S = B + Dc - E^T diag(1/(C + Dl)) E touches only the pose-pose corner, so E is
// kept dense [n_lm x ldE] (ldE = 6 n_kf rounded up to 16, +1 column carrying g_rho) and the rank-n_lm update is one
// v_mfma_f64_16x16x4_f64 SYRK ("MFMA only for the dense Schur reduce").  The right-hand side rides along as an
// augmented ROW of S (index d), so the forward substitution comes out of the Cholesky for free.
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <unordered_map>

#include "factor_eval.hpp"
#include "prior_eval.hpp"
#include "lvf_internal.hpp"
#include "imu_eval.hpp"

namespace lvf {
struct TfWork;
struct LmCtl;
struct Chain;
struct StageClock;
// one (v, ba, bg) block eliminated ahead of the dense factorisation: its 9 columns start at `col`, its `m` neighbour rows
// (later-ordered (v, ba, bg) blocks, poses, the augmented row; ascending) sit at rows[row_off .. row_off + m)
struct SpNode { int col, row_off, m, id; };
constexpr int kSpMaxLevels = 12, kSpMaxRows = 768;
struct SpLevels { int n; int first[kSpMaxLevels]; int count[kSpMaxLevels]; };
// "Early" sparse levels (see Chain::early): the level reads its columns as  B (natural order, what the ImuError factors accumulated) +
// LM damping + the updates of the levels below (all S holds there), instead of entries k_prepare assembled — so it does not have to wait
// for k_prepare and can ride in an earlier launch.  B == nullptr: the classic form (S holds the assembled entries).
// Ceres' Jacobi column scaling (Solver::Options::jacobi_scaling, a default the reference leaves on: backend.cpp:206-211; declared in
// oracle/lm.h's header): s_j = 1 / (1 + sqrt(H0_jj)) with H0 = diag(J^T J) of the solve's FIRST linearisation, frozen for the solve; the LM
// diagonal is clamped on the SCALED system, which in unscaled terms is D_jj = clamp(s_j^2 H_jj, 1e-6, 1e32) / s_j^2 (lm_damping).  h0 holds
// H0 in the natural order [15 n_kf camera unknowns | n_lm inverse depths]; while *frozen == 0 (the first pass of a solve) the kernels that
// form the damping store H_jj there, afterwards they read it (k_lm_decide raises the flag).
struct JacobiDev { GP<double> h0; GP<const int> frozen; };
struct SpSrc {
  GP<const double> B; int ldB, dp; GP<const double> gc; GP<const double> radius; GP<const int> rows_nat;
  // levels CHAINED inside one launch (the Schur complement's: it lasts long enough for three of them): the level waits until `wait_target`
  // workgroups of the level below have arrived at *wait_counter, and arrives at *done_counter itself.  What it reads of the level
  // below are RETURNING atomic adds into S (agent scope), read back with agent-scope atomic loads; the arrival is a RELEASE add, the
  // waiting side follows its spin with an agent-scope ACQUIRE fence in every wave (`fenced`, default; LVF_CHAIN_FENCE=0: the relaxed
  // round-3 form for A/B timing).  Producers carry lower workgroup numbers than their consumers, so they are normally dispatched
  // first — nothing DEPENDS on that: a consumer that does not see its producers within `timeout_ticks` raises the hand-over flag
  // (SC_FAIL >= kFailHandover), the decision ends the loop WITHOUT taking or counting the step (LVF_WHY_HANDOVER) and the host re-runs
  // the iteration with every level in a launch of its own (lvf_problem::no_chain) — a scheduling delay never becomes a numerical outcome.
  GP<int> wait_counter; int wait_target; GP<int> done_counter;
  int fenced; unsigned timeout_ticks;      // wall_clock64() ticks (100 MHz) before the hand-over is given up
  int strip_end;           // S rows / columns below it belong to sparse blocks (lvf_problem::off)
  int rmw_read;            // diagnostic: chained reads by returning atomics instead of agent-scope loads
  GP<unsigned long long> dbg; // LVF_SP_TIMING=1: eight wall_clock64() stamps per workgroup (tile 0 of every node), else null
  int s_zero;              // the level has nothing below it (level 0, early form): its part of S is still all zeros, not read
  JacobiDev jac{nullptr, nullptr};
};
struct SpArgs {          // one sparse level
  GP<const SpNode> nodes; int first, tiles; GP<const int> rows; GP<double> S; int ld; GP<double> W; int wstride; GP<double> Lout; GP<int> fail; int nblocks; GP<const int> done;
  SpSrc src;
};
__device__ __forceinline__ void sp_ride(const int vb, const SpArgs& a);     // workgroup vb of the level (defined with k_sp_eliminate)
// SC_FAIL codes (raised with atomicMax: the largest wins): 1 + kb = dense block step kb met a non-positive pivot, kFailSparse + id = sparse
// block id did, kFailHandover + id = a chained level gave up waiting for the level below (NOT a property of the problem: see SpSrc)
constexpr int kFailSparse = 100000, kFailHandover = 300000;
}
struct lvf_problem {
  lvf_ctx* ctx = nullptr;
  lvf_state* st = nullptr;
  lvf_batch *tc = nullptr, *tf = nullptr, *po = nullptr, *imu = nullptr, *prior = nullptr;
  int n_kf = 0, n_lm = 0, d = 0, dp = 0, ldE = 0, dpad = 0, nb = 0;
  // layout of the factorised matrix S (see "elimination order" below): [sparse (v,ba,bg) blocks | dense (v,ba,bg) blocks | poses | rhs row | pad]
  int ld = 0, off = 0, off_pose = 0, ndense = 0, aug = 0, sp_wstride = 0;
  lvf::SpLevels sp_levels{};
  std::vector<int> sp_tiles, sp_shmem;          // per level: workgroups per node, dynamic LDS bytes
  std::vector<int> sp_item0, sp_items;          // per level: its slice of sp_rows
  std::vector<int32_t> plan_key;                // (n_kf, IMU index pairs) the current plan was built for
  lvf::DevBuf<lvf::SpNode> sp_nodes;
  lvf::DevBuf<int> sp_rows, sp_rows_nat, sp_owner, perm, iperm;      // sp_rows_nat: the natural-order unknown of every entry of sp_rows (-2 = the right-hand-side row)
  lvf::DevBuf<int> lm_kmin, lm_kmax, lm_order, lm_nactive;   // per-landmark keyframe track [kmin, kmax]; Schur row order; #rows with pose blocks
  bool band_ready = false;
  // compact landmark layout + slabs of the atomic-free TwoFrame linearisation (see TfCompact)
  bool compact = false;
  lvf::DevBuf<int> lm_eoff, n_slots, tf_slot, run_first;
  lvf::DevBuf<double> slotB, slabP, slabQ, Ct, grt;
  lvf::StageClock* clk = nullptr;     // lvf_problem_stage_times
  bool accum_clean = false;           // B / gc / C / g_rho / cost stripes are zero (left so by the last iteration's cost + decision launch)
  const double* chain_tcw = nullptr;      // the TwoCamera per-block weight array the current chain was built with
  int band_rows = 64;           // landmark rows per slice of the band Schur complement (a batch uses more: fewer output atomics)
  lvf::DevBuf<int4> band_work; lvf::DevBuf<int> n_band_work_dev; lvf::HostPin<int> h_n_band_work;
  int n_band_work = 0, band_rows_built = 0;
  bool band_pending = false; hipEvent_t ev_band = nullptr;      // the item count of the list is still on its way (awaited just before the Schur launch)
  int band_epoch = 0;                 // bumped whenever the landmark bands change (a batch keeps its own, wider-slice work lists: lvf_problem_batch)
  std::vector<lvf_problem_batch*> batches;      // the batches that borrow this problem (they are told when it is destroyed)
  lvf::HostPin<int> h_run_first;
  lvf::DevBuf<unsigned long long> dbg, dbg_lin, dbg_sp;
  lvf::DevBuf<double> dbg_hist;                 // LVF_LM_HISTORY=1: the decisions of the last solve (lvf_problem_debug_history)
  lvf::DevBuf<double> sp_sync;                  // arrival counters of sparse levels chained inside one launch (one 8-byte slot per level, an int in each; cleared with the accumulators)
  lvf::DevBuf<double> sp_W, sp_L, Dinv;         // Dinv: L_kk^-T of every 64x64 diagonal block of the dense corner
  lvf::DevBuf<double> Ldiag;                    // the factored diagonal blocks L_kk [nb][64][64] (NOT stored back into S: see chol_step_body)
  std::vector<int> perm_h;
  lvf::DevBuf<double> B, gc, C, gr, E, Cd, S, dxc, dxl, scal;
  lvf::DevBuf<double> jh0;                      // Jacobi scaling of the running solve: diag(J^T J) of its first pass (JacobiDev)
  lvf::DevBuf<double> poses2, vel2, ba2, bg2, invd2;   // candidate state x + dx
  lvf::DevBuf<uint8_t> pose_const;
  lvf::DevBuf<int> fail;
  lvf::DevBuf<lvf::TfWork> tf_work;   // per-workgroup runs of same-k2 blocks (empty => generic atomic path)
  lvf::HostPin<lvf::TfWork> h_tf_work;
  std::vector<uint8_t> pose_const_h;
  bool linearized = false;
  // TwoFrame blocks as the solver reads them: the batch's own arrays, or — when the blocks of a current-keyframe run come with their first
  // keyframes in no order (landmark ids not in creation order) — copies sorted by (current, first) keyframe made at problem_configure, so
  // that a wave's 64 blocks share a few first keyframes and their sums go through the group-wise reductions instead of 63 LDS atomics per
  // block (the slowest workgroup of k_lin_visual: 18 -> 12 us).  The batch itself is never reordered (lvf_batch_evaluate keeps its order).
  lvf::DevBuf<double2> tfs_fo, tfs_ob;
  lvf::DevBuf<int> tfs_lm, tfs_k1, tfs_k2, tfs_perm;
  lvf::HostPin<int> h_tfs_perm;     // pinned staging of the permutation (the upload is asynchronous)
  bool tf_sorted_copy = false;
  const double2* tf_fo() const { return tf_sorted_copy ? tfs_fo.p : (const double2*)tf->ob_a.p; }
  const double2* tf_ob() const { return tf_sorted_copy ? tfs_ob.p : (const double2*)tf->ob_b.p; }
  const int* tf_lm() const { return tf_sorted_copy ? tfs_lm.p : tf->idx_a.p; }
  const int* tf_k1() const { return tf_sorted_copy ? tfs_k1.p : tf->idx_b.p; }
  const int* tf_k2() const { return tf_sorted_copy ? tfs_k2.p : tf->idx_c.p; }
  bool tf_unique_lk2 = false;   // no (landmark, current keyframe) pair occurs twice in the TwoFrame batch
  bool tf_k1_first = false;     // every TwoFrame block's first keyframe precedes its current keyframe
  double last_radius = 0;
  // the device-resident LM loop
  lvf::DevBuf<lvf::LmCtl> ctl;        // control block (radius, costs, accept / reject, termination) in HBM
  lvf::LmCtl* rec = nullptr;          // host-visible mirror written by k_lm_decide (hipHostMalloc)
  lvf::HostPin<lvf::LmCtl> h_ctl;     // pinned staging for uploads / read-backs of the control block
  lvf::Chain* chain = nullptr;        // argument blocks of one iteration
  bool chain_ready = false;
  bool no_chain = false;              // a chained hand-over timed out: this problem's levels are launches of their own until kUnchainedSolves solves have gone by
  int unchained_solves = 0;           // solves taken since no_chain was set (chaining is tried again after kUnchainedSolves of them: one scheduling blip
                                      // under Relocator traffic — relocator.cpp:188 — must not cost a persistent window its chained launches for good)
  int handover_retries = 0;           // iterations re-run because of that (reported in lvf_solver_summary::hand_over_retries)
  int force_handover_timeouts = 0;    // test hook (lvf_problem_debug_force_handover_timeout): the next chain is built with an unreachable wait target
  const void* chain_state[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};    // the state pointers the chain was built for
  double huber = 1.0;
  hipGraphExec_t graph_exec = nullptr;
  ~lvf_problem();
};

namespace lvf {

constexpr int kT = 256;
typedef double double4_t __attribute__((ext_vector_type(4)));
// device scalar slots
// Each sum slot is STRIPED over kStripes addresses (workgroup b adds into stripe b % kStripes, the host adds the stripes up): a
// cost pass issues one atomic per wave, ~1100 of them at configs[3], and on ONE address they serialise in L2 (measured: the
// residual-only TwoFrame pass spent 2/3 of its 16 us there).  SC_GMAX is a max (striped too; the decision takes the max over the stripes).
constexpr int kStripes = 32;
enum { SC_COST = 0, SC_COST_NEW = 1 * kStripes, SC_MODEL = 2 * kStripes, SC_DXNORM = 3 * kStripes, SC_XNORM = 4 * kStripes, SC_GMAX = 5 * kStripes,
       SC_N = 6 * kStripes, SC_FAIL = SC_N /* int flag */, SC_TICKET = SC_N + 1 /* int: workgroups of the candidate-cost pass that are done */, SC_ALLOC = SC_N + 2 };
static inline double stripe_sum(const double* h, int slot) { double s = 0.0; for (int k = 0; k < kStripes; ++k) s += h[slot + k]; return s; }

// The LM loop's `done` flag gates every launch of an iteration.  Tested at the top of a kernel it is a dependent global round trip
// (~0.5-1 us) in front of everything; issued FIRST and tested after the kernel's own first loads have been issued, its latency hides
// under theirs (loads return in order: the test waits for the oldest one only).
__device__ __forceinline__ int done_flag_issue(const int* done) { return done ? *reinterpret_cast<const volatile int*>(done) : 0; }

__device__ __forceinline__ void block_add(double v, double* dst) {
  v = wave_sum(v);
  // (x + y: a table launch may carry the window in either grid dimension — the stripes must be spread by the workgroup's number inside the window)
  if ((threadIdx.x & 63) == 0 && v != 0.0) atomicAdd(dst + ((blockIdx.x + blockIdx.y) & (kStripes - 1)), v);
}

struct StateP { GP<const double> poses, vel, ba, bg, inv_depth, w_kf; };

// Device-resident control block of one window's Levenberg-Marquardt loop.  Everything that changes from one iteration to the next
// lives here (trust-region radius, costs, accept / reject, termination), so the arguments of every kernel of an iteration are
// constant across iterations: the host enqueues iteration after iteration without waiting, k_lm_decide closes each one on device
// (what the reference's ceres::Solve does on the host between evaluations).
struct LmCtl {
  double radius, decrease;                         // trust region (in: start values; updated by every iteration)
  double last_radius;                              // the radius the last iteration's step was computed with
  double cost, initial_cost;                       // cost at the current state / at the first linearisation
  double cost_before, cost_after, model, dxnorm, xnorm, gmax;   // scalars of the last iteration
  double huber, function_tol, gradient_tol, parameter_tol, min_rel_decrease;
  int max_iters;
  int iter, successes, invalid_run;                // iterations taken / accepted steps / consecutive unsolvable steps
  int accepted, solved;                            // of the last iteration
  int done, termination;                           // done != 0: the remaining launches of this window return immediately
  int why, rejected;                               // LVF_WHY_* reason of the termination ; rejected / invalid steps so far
  int jfrozen;                                     // Jacobi scaling taken (JacobiDev): 0 until the solve's first pass has been decided on
};

// lower-triangle accumulation of a 6x6 block pair J_a^T J_b into B at (ra, rb) block offsets (ra >= rb required
// for off-diagonal; for ra == rb only the lower half is written)
__device__ __forceinline__ void add_block66(double* __restrict__ B, int ld, int ra, int rb, const double Ja[12], const double Jb[12]) {
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (ra == rb && j > i) continue;
      atomicAdd(&B[(size_t)(ra + i) * ld + rb + j], Ja[i] * Jb[j] + Ja[6 + i] * Jb[6 + j]);
    }
}

// ------------------------------------------------------------------------------------------------ housekeeping
// one launch zeroes every accumulator of a linearisation (B, gc, E, C, gr, scalars) instead of six fill kernels
// (ZeroList: lvf_internal.hpp)
// rows of a lower-triangular matrix (leading dimension ld, even): one wave per row, columns [0, end of the row's 64-column block)
__device__ __forceinline__ void zero_lower(double* __restrict__ p, const int ld, const unsigned long long wave, const unsigned long long n_waves) {
  const int lane = threadIdx.x & 63;
  for (unsigned long long r = wave; r < (unsigned long long)ld; r += n_waves) {
    double2* row = reinterpret_cast<double2*>(p + r * (unsigned long long)ld);
    const int end2 = min(ld, (((int)r | 63) + 1)) / 2;
    for (int c = lane; c < end2; c += 64) row[c] = make_double2(0.0, 0.0);
  }
}
__global__ __launch_bounds__(kT) void k_zero_multi(ZeroList z) {
  double* p = z.p[blockIdx.y];
  if (z.tri[blockIdx.y] > 0) { zero_lower(p, z.tri[blockIdx.y], (unsigned long long)blockIdx.x * (kT / 64) + (threadIdx.x >> 6), (unsigned long long)gridDim.x * (kT / 64)); return; }
  const unsigned long long n = z.n[blockIdx.y], n2 = n / 2;
  double2* p2 = reinterpret_cast<double2*>(p);
  for (unsigned long long i = (unsigned long long)blockIdx.x * kT + threadIdx.x; i < n2; i += (unsigned long long)gridDim.x * kT) p2[i] = make_double2(0.0, 0.0);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = 0.0;
}
// k_zero_multi + k_lm_range_init in one launch (problem_configure): slice z.count of the grid's y sets the landmark tracks to "none yet"
__global__ __launch_bounds__(kT) void k_zero_multi_ranges(ZeroList z, int n_lm, int* __restrict__ kmin, int* __restrict__ kmax) {
  if ((int)blockIdx.y == z.count) {
    for (int l = blockIdx.x * kT + threadIdx.x; l < n_lm; l += gridDim.x * kT) { kmin[l] = 0x7fffffff; kmax[l] = -1; }
    return;
  }
  double* p = z.p[blockIdx.y];
  const unsigned long long n = z.n[blockIdx.y], n2 = n / 2;
  double2* p2 = reinterpret_cast<double2*>(p);
  for (unsigned long long i = (unsigned long long)blockIdx.x * kT + threadIdx.x; i < n2; i += (unsigned long long)gridDim.x * kT) p2[i] = make_double2(0.0, 0.0);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = 0.0;
}
// workgroup `wg` of `n_wgs` (kT threads each) clears its share of every array of the list
// (static indices only: a run-time index into the by-value pointer table would put it in scratch memory)
__device__ __forceinline__ void zero_list_share(const ZeroList& zero, const int wg, const int n_wgs) {
  const unsigned long long t = (unsigned long long)wg * kT + threadIdx.x, nt = (unsigned long long)n_wgs * kT;
#pragma unroll
  for (int a = 0; a < kZeroListMax; ++a) {
    if (a >= zero.count) break;
    double* p = zero.p[a];
    if (zero.tri[a] > 0) { zero_lower(p, zero.tri[a], (unsigned long long)wg * (kT / 64) + (threadIdx.x >> 6), (unsigned long long)n_wgs * (kT / 64)); continue; }
    const unsigned long long cnt = zero.n[a], n2 = cnt / 2;
    double2* p2 = reinterpret_cast<double2*>(p);
    for (unsigned long long i = t; i < n2; i += nt) p2[i] = make_double2(0.0, 0.0);
    if ((cnt & 1) && t == 0) p[cnt - 1] = 0.0;
  }
}
// one launch for a batch of windows: blockIdx.y = window
__global__ __launch_bounds__(kT) void k_zero_table(const ZeroList* __restrict__ t) { zero_list_share(t[blockIdx.y], blockIdx.x, gridDim.x); }

// ------------------------------------------------------------------------------------------------ TwoCamera
template <bool COST_ONLY>
__device__ __forceinline__ void lin_tc_body(const int vb, int n, const double2* __restrict__ lo, const double2* __restrict__ ro,
                                               const int* __restrict__ lm, const int* __restrict__ kf, const double* __restrict__ wblk, StateP s,
                                               CamD left, CamD right, double huber, double* __restrict__ C,
                                               double* __restrict__ gr, double* __restrict__ cost) {
  const int i = vb * kT + threadIdx.x;
  double c = 0.0;
  if (i < n) {
    const int l = lm[i];
    const double2 a = lo[i], b = ro[i];
    double r[2], J[2];
    eval_two_camera<!COST_ONLY>(left, right, a.x, a.y, b.x, b.y, s.inv_depth[l], wblk ? wblk[i] : 5.0 * s.w_kf[kf[i]], r, J);
    double rho;
    const double sc = robust_scale(huber, r[0] * r[0] + r[1] * r[1], rho);
    c = 0.5 * rho;
    if (!COST_ONLY) {
      const double s2 = sc * sc;
      atomicAdd(&C[l], s2 * (J[0] * J[0] + J[1] * J[1]));
      atomicAdd(&gr[l], s2 * (J[0] * r[0] + J[1] * r[1]));
    }
  }
  block_add(c, cost);
}
template <bool COST_ONLY>
__global__ __launch_bounds__(kT) void k_lin_tc(int n, const double2* __restrict__ lo, const double2* __restrict__ ro,
                                               const int* __restrict__ lm, const int* __restrict__ kf, const double* __restrict__ wblk, StateP s,
                                               CamD left, CamD right, double huber, double* __restrict__ C,
                                               double* __restrict__ gr, double* __restrict__ cost) { lin_tc_body<COST_ONLY>(blockIdx.x, n, lo, ro, lm, kf, wblk, s, left, right, huber, C, gr, cost); }

// ------------------------------------------------------------------------------------------------ TwoFrame
template <bool COST_ONLY>
__global__ __launch_bounds__(kT) void k_lin_tf(int n, int n_kf, const double2* __restrict__ fo, const double2* __restrict__ ob,
                                               const int* __restrict__ lm, const int* __restrict__ kf1,
                                               const int* __restrict__ kf2, StateP s, CamD left, CamD right, double huber,
                                               const uint8_t* __restrict__ pose_const, double* __restrict__ B, int ld,
                                               double* __restrict__ gc, double* __restrict__ E, int ldE,
                                               double* __restrict__ C, double* __restrict__ gr, double* __restrict__ cost) {
  __shared__ PoseD s_pose[kMaxStagedKf];
  stage_poses<kT>(s_pose, s.poses, n_kf);
  const int i = blockIdx.x * kT + threadIdx.x;
  double c = 0.0;
  if (i < n) {
    const int l = lm[i], k1 = kf1[i], k2 = kf2[i];
    const double2 a = fo[i], b = ob[i];
    const PoseD P1 = fetch_pose(s_pose, s.poses, n_kf, k1), P2 = fetch_pose(s_pose, s.poses, n_kf, k2);
    double r[2], Jd[2], L1[12], L2[12];
    if (COST_ONLY) { double J1[14], J2[14]; eval_two_frame<false>(P1, P2, left, right, a.x, a.y, b.x, b.y, s.inv_depth[l], s.w_kf[k2], r, Jd, J1, J2); }
    else eval_two_frame_local(P1, P2, left, right, a.x, a.y, b.x, b.y, s.inv_depth[l], s.w_kf[k2], r, Jd, L1, L2);
    double rho;
    const double sc = robust_scale(huber, r[0] * r[0] + r[1] * r[1], rho);
    c = 0.5 * rho;
    if (!COST_ONLY) {
      {
        const double s1 = (pose_const[k1] & 1) ? 0.0 : sc, s2 = (pose_const[k2] & 1) ? 0.0 : sc;
#pragma unroll
        for (int q = 0; q < 12; ++q) { L1[q] *= s1; L2[q] *= s2; }
      }
      const double r0 = sc * r[0], r1 = sc * r[1], d0 = sc * Jd[0], d1 = sc * Jd[1];
      atomicAdd(&C[l], d0 * d0 + d1 * d1);
      atomicAdd(&gr[l], d0 * r0 + d1 * r1);
      if (k1 == k2) {   // degenerate: both pose blocks are the same parameter
#pragma unroll
        for (int q = 0; q < 12; ++q) L1[q] += L2[q];
        add_block66(B, ld, 6 * k1, 6 * k1, L1, L1);
#pragma unroll
        for (int q = 0; q < 6; ++q) { atomicAdd(&gc[6 * k1 + q], L1[q] * r0 + L1[6 + q] * r1); atomicAdd(&E[(size_t)l * ldE + 6 * k1 + q], L1[q] * d0 + L1[6 + q] * d1); }
      } else {
        add_block66(B, ld, 6 * k1, 6 * k1, L1, L1);
        add_block66(B, ld, 6 * k2, 6 * k2, L2, L2);
        if (k2 > k1) add_block66(B, ld, 6 * k2, 6 * k1, L2, L1); else add_block66(B, ld, 6 * k1, 6 * k2, L1, L2);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          atomicAdd(&gc[6 * k1 + q], L1[q] * r0 + L1[6 + q] * r1);
          atomicAdd(&gc[6 * k2 + q], L2[q] * r0 + L2[6 + q] * r1);
          atomicAdd(&E[(size_t)l * ldE + 6 * k1 + q], L1[q] * d0 + L1[6 + q] * d1);
          atomicAdd(&E[(size_t)l * ldE + 6 * k2 + q], L2[q] * d0 + L2[6 + q] * d1);
        }
      }
    }
  }
  block_add(c, cost);
}

// TwoFrame linearisation, fast path: blocks sorted by CURRENT keyframe k2 and every workgroup handed a run of blocks that
// share one k2 (host-built work list).  Then
//   * B[k2,k2] and g[k2] are wave-shuffle reductions (one LDS add per wave, one global flush per workgroup),
//   * B[k1,k1], g[k1] and the cross block B[k2,k1] only depend on k1 <= n_kf: accumulated with ds_add_f64 in a
//     [n_kf][63] LDS table and flushed once per workgroup (non-zero entries only),
//   * only the landmark-indexed sums (C, g_rho, E rows) remain global atomics: 14 per block instead of 134.
struct TfWork { int first, count, k2; };
// Atomic-free outputs of the sorted TwoFrame linearisation ("compact" mode).  Measured on MI355X: the 8 landmark-indexed global f64
// atomics per block were 18 of the 21 us a workgroup spent between loading its blocks and its reductions, and together with the
// per-workgroup flush of the keyframe-indexed sums (~1 M atomics per linearisation at configs[3]) they are a chip-wide L2 bottleneck
// (~30 atomics / ns) that a batch of windows hits W times over.  Instead:
//   * the k2 columns of a landmark's (dense) E row have exactly one writer: plain stores.  The row is NOT cleared per linearisation: its
//     non-zero pattern (the landmark's track) is fixed for a problem, so E is zeroed once per problem_configure and every entry inside
//     the pattern is overwritten by every linearisation;
//   * every block owns a SLOT s = eoff[l] + (k2 - k1 - 1) of its landmark's track and writes there, with plain 16-byte stores, one
//     64-byte record: its contributions to the k1 columns of E (6), to C and to g_rho.  k_prepare reads a landmark's slots as one
//     contiguous range, sums them and completes the row (k1 columns, g_rho column) and Cd;
//   * every workgroup writes its LDS table of keyframe-indexed sums to its own slab (slabP[wg][k1][64], slabQ[wg][32]); k_tf_reduce adds
//     the slabs of a run into B / gc, each entry of B having exactly one owner there.
struct TfCompact { int on; GP<const int> slot; GP<double> slotB, slabP, slabQ; int staged; };
constexpr int kSlabRow = 64, kSlabQ = 32;
constexpr int kAccSlots = 63;   // 21 (B[k1,k1] lower) + 6 (g[k1]) + 36 (cross block, rows = k2 tangent, cols = k1 tangent)
constexpr int kStageWave = 64 * 9 + 32;   // doubles of LDS staging per wave (segmented first-keyframe sums): 64 x (8 + 1 pad) values + 64 ints
// DYN: the LDS tables are carved from the launch's dynamic LDS, sized by the window's n_kf (k_lin_visual: 32 KB at 50 keyframes instead of
// 79 KB of static arrays sized for 64 — three workgroups per CU instead of two)
template <bool DYN = false>
__device__ __forceinline__ void lin_tf_sorted_body(const int vb, const TfWork* __restrict__ work, int n_kf, const double2* __restrict__ fo,
                                                      const double2* __restrict__ ob, const int* __restrict__ lm,
                                                      const int* __restrict__ kf1, StateP s, CamD left, CamD right, double huber,
                                                      const uint8_t* __restrict__ pose_const, double* __restrict__ B, int ld,
                                                      double* __restrict__ gc, double* __restrict__ E, int ldE,
                                                      double* __restrict__ C, double* __restrict__ gr, double* __restrict__ cost, int unique_lk2,
                                                      unsigned long long* dbg = nullptr, const TfCompact cp = TfCompact{0, nullptr, nullptr, nullptr, nullptr, 1}, const int dbg_nw = 0) {
  __shared__ PoseD s_pose_st[DYN ? 1 : kMaxStagedKf];
  __shared__ double s_acc_st[DYN ? 1 : kMaxStagedKf * kAccSlots];
  __shared__ double s_k2_st[DYN ? 1 : 27];
  extern __shared__ double lin_lds[];
  PoseD* s_pose = DYN ? reinterpret_cast<PoseD*>(lin_lds) : s_pose_st;
  double* s_acc = DYN ? lin_lds + (sizeof(PoseD) / 8) * n_kf : s_acc_st;
  double* s_k2 = DYN ? s_acc + kAccSlots * n_kf : s_k2_st;
  // per-wave staging of the segmented first-keyframe sums (DYN only): [64 lanes][8 slots + 1 pad] doubles + the lanes' first keyframes
  double* s_stage = DYN ? s_k2 + 32 + (threadIdx.x >> 6) * kStageWave : nullptr;
  int* s_stage_k1 = reinterpret_cast<int*>(s_stage + 64 * 9);
  const TfWork w = work[vb];
  auto mark = [&](int k) { if (dbg && threadIdx.x == 0) dbg[(size_t)vb * 8 + k] = wall_clock64(); };   // LVF_LIN_TIMING=1: phase stamps (100 MHz)
  // ... and per wave (lane 0 of each): [0] before the evaluation, [1] after it, [2] after the first-keyframe sums, [3] = number of first-keyframe groups
  unsigned long long* dbgw = dbg ? dbg + (size_t)dbg_nw * 8 + 8 + (size_t)vb * 16 + (threadIdx.x >> 6) * 4 : nullptr;
  auto markw = [&](int k) { if (dbgw && (threadIdx.x & 63) == 0) dbgw[k] = wall_clock64(); };
  mark(0);
  for (int e = threadIdx.x; e < n_kf * kAccSlots; e += kT) s_acc[e] = 0.0;
  if (threadIdx.x < 27) s_k2[threadIdx.x] = 0.0;
  stage_poses<kT>(s_pose, s.poses, n_kf);   // ends with __syncthreads()
  mark(1);
  const int k2 = w.k2;
  double c = 0.0;
  double v[27];
#pragma unroll
  for (int q = 0; q < 27; ++q) v[q] = 0.0;
  const bool active = (int)threadIdx.x < w.count;
  markw(0);
  int k1 = 0;
  double L1[12], L2[12], r0 = 0.0, r1 = 0.0;
#pragma unroll
  for (int q = 0; q < 12; ++q) { L1[q] = 0.0; L2[q] = 0.0; }
  if (active) {
    const int i = w.first + threadIdx.x;
    const int l = lm[i];
    k1 = kf1[i];
    const double2 a = fo[i], b = ob[i];
    double r[2], Jd[2];
    eval_two_frame_local(s_pose[k1], s_pose[k2], left, right, a.x, a.y, b.x, b.y, s.inv_depth[l], s.w_kf[k2], r, Jd, L1, L2);      // pose rows in tangent coordinates, closed form
    double rho;
    const double sc = robust_scale(huber, r[0] * r[0] + r[1] * r[1], rho);
    c = 0.5 * rho;
    {
      const double s1 = (pose_const[k1] & 1) ? 0.0 : sc, s2 = (pose_const[k2] & 1) ? 0.0 : sc;
#pragma unroll
      for (int q = 0; q < 12; ++q) { L1[q] *= s1; L2[q] *= s2; }
    }
    r0 = sc * r[0]; r1 = sc * r[1];
    const double d0 = sc * Jd[0], d1 = sc * Jd[1];
    if (cp.on) {
      // atomic-free mode: the k2 columns of the landmark's E row have ONE writer (plain stores); the contributions to C, g_rho and to the
      // k1 columns go to this block's slot of the landmark's track as one 64-byte record (summed per landmark by k_prepare)
      double* el = E + (size_t)l * ldE + 6 * k2;
      double2* rec = reinterpret_cast<double2*>(cp.slotB + (size_t)cp.slot[i] * 8);
      double e1[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) { e1[q] = L1[q] * d0 + L1[6 + q] * d1; el[q] = L2[q] * d0 + L2[6 + q] * d1; }
      rec[0] = make_double2(e1[0], e1[1]); rec[1] = make_double2(e1[2], e1[3]); rec[2] = make_double2(e1[4], e1[5]);
      rec[3] = make_double2(d0 * d0 + d1 * d1, d0 * r0 + d1 * r1);
    } else {
      atomicAdd(&C[l], d0 * d0 + d1 * d1);
      atomicAdd(&gr[l], d0 * r0 + d1 * r1);
      double* el = E + (size_t)l * ldE;
      // E[l][k2 columns] has exactly ONE writer when no landmark is observed twice by a keyframe (checked on the host when the batch
      // is created; always true for what BuildProblem builds): plain stores.
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        atomicAdd(&el[6 * k1 + q], L1[q] * d0 + L1[6 + q] * d1);
        const double e2 = L2[q] * d0 + L2[6 + q] * d1;
        if (unique_lk2) el[6 * k2 + q] = e2; else atomicAdd(&el[6 * k2 + q], e2);
      }
    }
  }
  mark(2); markw(1);
  // k1-indexed sums (B[k1,k1], g[k1], cross block).  A live front-end hands out landmark ids in creation order, so the blocks of a
  // wave usually share their first keyframe (per-lane ds_add_f64 on ONE address serialises 64-fold: +10 us on this kernel with ids in
  // birth order).
  {
    // the k2-indexed products v[27] (reduced below)
    {
      int q = 0;
#pragma unroll
      for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int y = 0; y <= x; ++y) v[q++] = L2[x] * L2[y] + L2[6 + x] * L2[6 + y];
#pragma unroll
      for (int x = 0; x < 6; ++x) v[21 + x] = L2[x] * r0 + L2[6 + x] * r1;
    }
    // The 63 k1-indexed products, slot order = [B(k1,k1) 21 | g(k1) 6 | cross (k2,k1) 36].  Lanes that share their first keyframe
    // are reduced TOGETHER (two transposed reductions of 32 slots, formed one batch at a time to keep the register count where three
    // workgroups fit a CU) — up to four groups of >= 16 lanes per wave, which covers whole waves and the waves that straddle a boundary
    // between two first keyframes; whatever is left (random landmark ids: nearly every lane its own keyframe) goes through per-lane
    // LDS atomics, which are only slow when many lanes hit one address.
    const int lane = threadIdx.x & 63;
    unsigned long long remaining = __ballot(active);
    bool mine_done = !active;
    // a wave with at most four first keyframes (creation-order ids: one or two long runs and a short one at a boundary) is reduced
    // group by group whatever the group sizes: a 10-lane group left to the atomics serialises 10-fold on each of its 63 addresses
    int n_groups = 0;
    for (unsigned long long rem = remaining; rem && n_groups < 5; ++n_groups)
      rem &= ~__ballot(active && k1 == __builtin_amdgcn_readlane(k1, (int)__ffsll((long long)rem) - 1));
    const int min_group = n_groups <= 4 ? 1 : 16;
    if (dbgw && lane == 0) dbgw[3] = (unsigned long long)n_groups;
#pragma unroll 1
    for (int round = 0; round < 4 && remaining; ++round) {
      const int k1u = __builtin_amdgcn_readlane(k1, (int)__ffsll((long long)remaining) - 1);
      const bool sel = active && !mine_done && k1 == k1u;
      const unsigned long long m = __ballot(sel);
      if (__popcll(m) < min_group) break;
      double* accu = s_acc + k1u * kAccSlots;
      // the group's lanes keep their first-keyframe Jacobian, everyone else contributes zeros; the scale is opaque to the compiler so
      // that the 63 products are formed inside this loop (hoisted out of it as loop invariants they cost 126 registers)
      double z = sel ? 1.0 : 0.0;
      asm volatile("" : "+v"(z));
      double M[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) M[i] = L1[i] * z;
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) {
        double t[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) t[i] = 0.0;
        int q = 0;
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
          for (int y = 0; y <= x; ++y) { if ((q >> 5) == bt) t[q & 31] = M[x] * L1[y] + M[6 + x] * L1[6 + y]; ++q; }
#pragma unroll
        for (int x = 0; x < 6; ++x) { if ((q >> 5) == bt) t[q & 31] = M[x] * r0 + M[6 + x] * r1; ++q; }
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
          for (int y = 0; y < 6; ++y) { if ((q >> 5) == bt) t[q & 31] = L2[x] * M[y] + L2[6 + x] * M[6 + y]; ++q; }
        const double tot = wave_sum32(t);
        const int slot = 32 * bt + (lane >> 1);
        if (!(lane & 1) && slot < 63 && tot != 0.0) atomicAdd(&accu[slot], tot);
      }
      mine_done = mine_done || sel;
      remaining &= ~m;
    }
    if (DYN && n_groups > 4 && cp.staged) {
      // Many first keyframes in one wave (the old tail of a late keyframe's run: dozens of groups of one to three blocks; or landmark ids
      // in no order).  LDS f64 atomics retire at ~2 lane-operations per clock per CU (measured: 63 per lane cost a wave 8-10 us with
      // three workgroups on the CU), so their NUMBER is what counts: the 63 products go through a wave-private LDS tile eight slots
      // at a time, lane (slot s, part) walks 8 consecutive lanes' values and adds one partial sum per run of equal first keyframes —
      // 63 x (segments + 7) atomics per wave instead of 63 x 64 (blocks sorted by first keyframe: a few hundred instead of 4 032).
      // (lanes the group rounds above have served contribute zeros below)
      double z = (active && !mine_done) ? 1.0 : 0.0;
      asm volatile("" : "+v"(z));
      double M[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) M[i] = L1[i] * z;
      s_stage_k1[lane] = (active && !mine_done) ? k1 : -1;
      const int ss = lane & 7, part = lane >> 3;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // the first keyframes of this lane's 8 rows (and of the row behind them), once: everything the segment logic needs sits in registers
      // before the first atomic (the compiler orders LDS reads behind LDS atomics it cannot tell apart from them)
      int kk[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) kk[j] = (8 * part + j < 64) ? s_stage_k1[8 * part + j] : -2;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        double t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = 0.0;
        int q = 0;
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
          for (int y = 0; y <= x; ++y) { if ((q >> 3) == ch) t[q & 7] = M[x] * L1[y] + M[6 + x] * L1[6 + y]; ++q; }
#pragma unroll
        for (int x = 0; x < 6; ++x) { if ((q >> 3) == ch) t[q & 7] = M[x] * r0 + M[6 + x] * r1; ++q; }
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
          for (int y = 0; y < 6; ++y) { if ((q >> 3) == ch) t[q & 7] = L2[x] * M[y] + L2[6 + x] * M[6 + y]; ++q; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) s_stage[lane * 9 + i] = t[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int slot = 8 * ch + ss;
        double val[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) val[j] = s_stage[(8 * part + j) * 9 + ss];
        double run = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          run += val[j];
          if (j == 7 || kk[j + 1] != kk[j]) {
            if (kk[j] >= 0 && slot < kAccSlots && run != 0.0) atomicAdd(&s_acc[kk[j] * kAccSlots + slot], run);
            run = 0.0;
          }
        }
      }
    } else if (!mine_done) {
      double* acc = s_acc + k1 * kAccSlots;
      int q = 0;
#pragma unroll
      for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int y = 0; y <= x; ++y) atomicAdd(&acc[q++], L1[x] * L1[y] + L1[6 + x] * L1[6 + y]);
#pragma unroll
      for (int x = 0; x < 6; ++x) atomicAdd(&acc[21 + x], L1[x] * r0 + L1[6 + x] * r1);
#pragma unroll
      for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int y = 0; y < 6; ++y) atomicAdd(&acc[27 + 6 * x + y], L2[x] * L1[y] + L2[6 + x] * L1[6 + y]);
    }
  }
  mark(3); markw(2);
  {
    // the 27 sums of the wave in one transposed reduction: lane l ends up with the total of value l >> 1, the even lanes add them
    double v32[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) v32[q] = q < 27 ? v[q] : 0.0;
    const double tot = wave_sum32(v32);
    const int lane = threadIdx.x & 63;
    if (!(lane & 1) && (lane >> 1) < 27 && tot != 0.0) atomicAdd(&s_k2[lane >> 1], tot);
  }
  block_add(c, cost);
  __syncthreads();
  mark(4);
  if (cp.on) {
    // the workgroup's sums leave as plain, fully coalesced stores into its own slab (rows k1 < k2 only; k_tf_reduce reads exactly those)
    if (threadIdx.x < kSlabQ) cp.slabQ[(size_t)vb * kSlabQ + threadIdx.x] = threadIdx.x < 27 ? s_k2[threadIdx.x] : 0.0;
    double* P = cp.slabP + (size_t)vb * n_kf * kSlabRow;
    for (int e = threadIdx.x; e < k2 * kSlabRow; e += kT) {
      const int kk = e >> 6, slot = e & 63;
      P[e] = slot < kAccSlots ? s_acc[kk * kAccSlots + slot] : 0.0;
    }
  } else {
  if (threadIdx.x < 27) {
    const double val = s_k2[threadIdx.x];
    if (val != 0.0) {
      if (threadIdx.x < 21) {
        int x = 0, rem = threadIdx.x;
        while (rem > x) { rem -= x + 1; ++x; }
        atomicAdd(&B[(size_t)(6 * k2 + x) * ld + 6 * k2 + rem], val);
      } else atomicAdd(&gc[6 * k2 + threadIdx.x - 21], val);
    }
  }
  for (int e = threadIdx.x; e < n_kf * kAccSlots; e += kT) {
    const double val = s_acc[e];
    if (val == 0.0) continue;
    const int k1 = e / kAccSlots, slot = e % kAccSlots;
    if (slot < 21) {
      int x = 0, rem = slot;
      while (rem > x) { rem -= x + 1; ++x; }
      atomicAdd(&B[(size_t)(6 * k1 + x) * ld + 6 * k1 + rem], val);
    } else if (slot < 27) {
      atomicAdd(&gc[6 * k1 + slot - 21], val);
    } else {
      const int x = (slot - 27) / 6, y = (slot - 27) % 6;   // x: k2 tangent index, y: k1 tangent index
      if (k2 > k1) atomicAdd(&B[(size_t)(6 * k2 + x) * ld + 6 * k1 + y], val);
      else atomicAdd(&B[(size_t)(6 * k1 + y) * ld + 6 * k2 + x], val);
    }
  }
  }
  __syncthreads();
  mark(5);
}
// ------------------------------------------------------------------------------------------------ candidate cost, visual factors
// One launch for the residual-only passes of the three reprojection batches (the workgroups of the launch are split into a
// TwoCamera, a TwoFrame and a PoseOnly segment): as three back-to-back launches of 4-9 us they were mostly launch boundaries.
struct CostVisual {
  int n_tc, n_tf, n_po, g_tc, g_tf;
  GP<const double2> tc_lo, tc_ro; GP<const int> tc_lm, tc_kf; GP<const double> tc_w; CamD tc_left, tc_right;
  GP<const double2> tf_fo, tf_ob; GP<const int> tf_lm, tf_k1, tf_k2; CamD tf_left, tf_right;
  GP<const double2> po_ob; GP<const int> po_kf, po_pwi; GP<const double> po_pw; CamD po_cam;
};
struct ImuEvalArgs { int n; GP<const double> pre, sqrt_info; GP<const int> kf_i, kf_j; };      // ImuError factors evaluated inside a merged launch
struct CostArgs {
  CostVisual a; int n_kf; StateP s; double huber; GP<double> cost; int nblocks; GP<const int> done;
  ImuEvalArgs imu; int g_imu;       // workgroups [0, g_imu) evaluate one ImuError factor each, the visual passes follow
  int tiles;                        // tiles of kT blocks per visual workgroup (0 = 1)
  ZeroList zero; int zero_wgs;      // workgroups [nblocks, nblocks + zero_wgs) of the merged cost + decision launch clear the accumulators for the NEXT linearisation
};
// the calling thread's share of the candidate cost (workgroup b of the pass)
__device__ __forceinline__ double cost_visual_value(const int b, const CostArgs& A) {
  const CostVisual& a = A.a;
  const int n_kf = A.n_kf; const StateP s = A.s; const double huber = A.huber;
  const int tiles = A.tiles > 0 ? A.tiles : 1;          // tiles of kT blocks per workgroup (a batch of windows uses fatter workgroups: with
                                                        // 8 x 600 thin ones the launch took 52 us, with a quarter of them 31)
  __shared__ PoseD s_pose[kMaxStagedKf];
  double c = 0.0;
  if (b < a.g_tc) {
    for (int rp = 0; rp < tiles; ++rp) {
      const int i = (b * tiles + rp) * kT + threadIdx.x;
      if (i < a.n_tc) {
        const int l = a.tc_lm[i];
        const double2 lo = a.tc_lo[i], ro = a.tc_ro[i];
        double r[2], J[2];
        eval_two_camera<false>(a.tc_left, a.tc_right, lo.x, lo.y, ro.x, ro.y, s.inv_depth[l], a.tc_w ? a.tc_w[i] : 5.0 * s.w_kf[a.tc_kf[i]], r, J);
        double rho;
        (void)robust_scale(huber, r[0] * r[0] + r[1] * r[1], rho);
        c += 0.5 * rho;
      }
    }
  } else {
    stage_poses<kT>(s_pose, s.poses, n_kf);     // ends with __syncthreads(); the branch is workgroup-uniform
    if (b < a.g_tc + a.g_tf) {
      for (int rp = 0; rp < tiles; ++rp) {
        const int i = ((b - a.g_tc) * tiles + rp) * kT + threadIdx.x;
        if (i < a.n_tf) {
          const int l = a.tf_lm[i], k1 = a.tf_k1[i], k2 = a.tf_k2[i];
          const double2 fo = a.tf_fo[i], ob = a.tf_ob[i];
          const PoseD P1 = fetch_pose(s_pose, s.poses, n_kf, k1), P2 = fetch_pose(s_pose, s.poses, n_kf, k2);
          double r[2], Jd[2], J1[14], J2[14];
          eval_two_frame<false>(P1, P2, a.tf_left, a.tf_right, fo.x, fo.y, ob.x, ob.y, s.inv_depth[l], s.w_kf[k2], r, Jd, J1, J2);
          double rho;
          (void)robust_scale(huber, r[0] * r[0] + r[1] * r[1], rho);
          c += 0.5 * rho;
        }
      }
    } else {
      for (int rp = 0; rp < tiles; ++rp) {
        const int i = ((b - a.g_tc - a.g_tf) * tiles + rp) * kT + threadIdx.x;
        if (i < a.n_po) {
          const int k = a.po_kf[i], l = a.po_pwi[i];
          const double2 o = a.po_ob[i];
          const PoseD P = fetch_pose(s_pose, s.poses, n_kf, k);
          const double pwl[3] = {a.po_pw[3 * l], a.po_pw[3 * l + 1], a.po_pw[3 * l + 2]};
          double r[2], J[14];
          eval_pose_only<false>(P, a.po_cam, o.x, o.y, pwl, s.w_kf[k], r, J);
          double rho;
          (void)robust_scale(huber, r[0] * r[0] + r[1] * r[1], rho);
          c += 0.5 * rho;
        }
      }
    }
  }
  return c;
}
__device__ __forceinline__ void cost_visual_body(const int b, const CostArgs& A) {
  if (b >= A.nblocks || (A.done && *A.done)) return;
  block_add(cost_visual_value(b, A), A.cost);
}
__global__ __launch_bounds__(kT) void k_cost_visual(CostArgs a) { cost_visual_body(blockIdx.x, a); }
__global__ __launch_bounds__(kT) void k_cost_visual_b(const CostArgs* __restrict__ t) { cost_visual_body(blockIdx.x, t[blockIdx.y]); }

// ------------------------------------------------------------------------------------------------ PoseOnly
template <bool COST_ONLY, bool DYN = false>
__device__ __forceinline__ void lin_po_body(const int vb, int n, int n_kf, const double2* __restrict__ ob, const int* __restrict__ kf,
                                               const int* __restrict__ pwi, const double* __restrict__ pw, StateP s, CamD cam,
                                               double huber, const uint8_t* __restrict__ pose_const, double* __restrict__ B,
                                               int ld, double* __restrict__ gc, double* __restrict__ cost) {
  __shared__ PoseD s_pose_st[DYN ? 1 : kMaxStagedKf];
  extern __shared__ double lin_lds[];
  PoseD* s_pose = DYN ? reinterpret_cast<PoseD*>(lin_lds) : s_pose_st;
  stage_poses<kT>(s_pose, s.poses, n_kf);
  const int i = vb * kT + threadIdx.x;
  double c = 0.0;
  int k = -1;
  double v[27];
#pragma unroll
  for (int q = 0; q < 27; ++q) v[q] = 0.0;
  if (i < n) {
    k = kf[i];
    const int l = pwi[i];
    const double2 o = ob[i];
    const PoseD P = fetch_pose(s_pose, s.poses, n_kf, k);
    const double pwl[3] = {pw[3 * l], pw[3 * l + 1], pw[3 * l + 2]};
    double r[2], Lc[12];
    if (COST_ONLY) { double J[14]; eval_pose_only<false>(P, cam, o.x, o.y, pwl, s.w_kf[k], r, J); }
    else eval_pose_only_local(P, cam, o.x, o.y, pwl, s.w_kf[k], r, Lc);
    double rho;
    const double sc = robust_scale(huber, r[0] * r[0] + r[1] * r[1], rho);
    c = 0.5 * rho;
    if (!COST_ONLY) {
      {
        const double sl = (pose_const[k] & 1) ? 0.0 : sc;
#pragma unroll
        for (int q = 0; q < 12; ++q) Lc[q] *= sl;
      }
      const double r0 = sc * r[0], r1 = sc * r[1];
      int q = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) v[q++] = Lc[a] * Lc[b] + Lc[6 + a] * Lc[6 + b];
#pragma unroll
      for (int a = 0; a < 6; ++a) v[21 + a] = Lc[a] * r0 + Lc[6 + a] * r1;
    }
  }
  if (!COST_ONLY) {
    // blocks are normally sorted by keyframe: when the whole wave shares one pose block, reduce the 21+6 sums with wave shuffles
    // (the per-pose-block JtJ/Jtr reduction of the north star).  The wave sums then meet in a per-workgroup LDS table
    // [n_kf][27] and only its non-zero entries go to global memory: ~15 waves per keyframe used to hit the SAME 27 addresses of B
    // with global atomics, which serialise in L2.
    __shared__ double s_acc_st[DYN ? 1 : kMaxStagedKf * 27];
    double* s_acc = DYN ? lin_lds + (sizeof(PoseD) / 8) * n_kf : s_acc_st;
    const bool lds_path = n_kf <= kMaxStagedKf;
    if (lds_path) {
      for (int e = threadIdx.x; e < n_kf * 27; e += kT) s_acc[e] = 0.0;
      __syncthreads();
    }
    const int k0 = __shfl(k, 0);
    const bool uniform = __all(k == k0) && k0 >= 0;
    if (uniform) {
#pragma unroll
      for (int q = 0; q < 27; ++q) v[q] = wave_sum(v[q]);
      if ((threadIdx.x & 63) == 0) {
        if (lds_path) { for (int q = 0; q < 27; ++q) atomicAdd(&s_acc[k0 * 27 + q], v[q]); }
        else {
          int q = 0;
          for (int a = 0; a < 6; ++a) for (int b = 0; b <= a; ++b) atomicAdd(&B[(size_t)(6 * k0 + a) * ld + 6 * k0 + b], v[q++]);
          for (int a = 0; a < 6; ++a) atomicAdd(&gc[6 * k0 + a], v[21 + a]);
        }
      }
    } else if (k >= 0) {
      if (lds_path) { for (int q = 0; q < 27; ++q) atomicAdd(&s_acc[k * 27 + q], v[q]); }
      else {
        int q = 0;
        for (int a = 0; a < 6; ++a) for (int b = 0; b <= a; ++b) atomicAdd(&B[(size_t)(6 * k + a) * ld + 6 * k + b], v[q++]);
        for (int a = 0; a < 6; ++a) atomicAdd(&gc[6 * k + a], v[21 + a]);
      }
    }
    if (lds_path) {
      __syncthreads();
      for (int e = threadIdx.x; e < n_kf * 27; e += kT) {
        const double val = s_acc[e];
        if (val == 0.0) continue;
        const int kk = e / 27, slot = e - kk * 27;
        if (slot < 21) {
          int x = 0, rem = slot;
          while (rem > x) { rem -= x + 1; ++x; }
          atomicAdd(&B[(size_t)(6 * kk + x) * ld + 6 * kk + rem], val);
        } else atomicAdd(&gc[6 * kk + slot - 21], val);
      }
    }
  }
  block_add(c, cost);
}
template <bool COST_ONLY>
__global__ __launch_bounds__(kT) void k_lin_po(int n, int n_kf, const double2* __restrict__ ob, const int* __restrict__ kf,
                                               const int* __restrict__ pwi, const double* __restrict__ pw, StateP s, CamD cam,
                                               double huber, const uint8_t* __restrict__ pose_const, double* __restrict__ B,
                                               int ld, double* __restrict__ gc, double* __restrict__ cost) { lin_po_body<COST_ONLY>(blockIdx.x, n, n_kf, ob, kf, pwi, pw, s, cam, huber, pose_const, B, ld, gc, cost); }

// ------------------------------------------------------------------------------------------------ IMU
// consumes the materialised ImuError outputs (res[n][15], eight Jacobian blocks) of launch_imu; one wave per factor
struct ImuJ { GP<const double> j[8]; };
__global__ __launch_bounds__(64) void k_lin_imu(int n, int n_kf, const double* __restrict__ res, ImuJ J, const int* __restrict__ kf_i,
                                                const int* __restrict__ kf_j, const double* __restrict__ poses,
                                                const uint8_t* __restrict__ pose_const, double* __restrict__ B, int ld,
                                                double* __restrict__ gc, double* __restrict__ cost) {
  __shared__ double sJ[15 * 30];   // local Jacobian: [pose_i 6 | vbb_i 9 | pose_j 6 | vbb_j 9]
  __shared__ double sr[15];
  __shared__ int sidx[30];
  const int f = blockIdx.x, lane = threadIdx.x;
  const int ki = kf_i[f], kj = kf_j[f];
  if (lane < 15) sr[lane] = res[(size_t)f * 15 + lane];
  if (lane < 30) {
    int g;
    if (lane < 6) g = 6 * ki + lane; else if (lane < 15) g = 6 * n_kf + 9 * ki + (lane - 6);
    else if (lane < 21) g = 6 * kj + (lane - 15); else g = 6 * n_kf + 9 * kj + (lane - 21);
    sidx[lane] = g;
  }
  // pose blocks: 15 rows x (7 -> 6)
  for (int e = lane; e < 30; e += 64) {
    const int row = e % 15, which = e / 15;            // which: 0 = pose_i, 1 = pose_j
    const double* Jr = (which ? J.j[4] : J.j[0]) + (size_t)f * 105 + 7 * row;
    const int kk = which ? kj : ki;
    const double* q = poses + 7 * kk;
    const double sc = (pose_const[kk] & 1) ? 0.0 : 1.0;
    double l3[3];
    quat_row_to_local(Jr, q, l3);
    double* o = sJ + row * 30 + (which ? 15 : 0);
    o[0] = sc * l3[0]; o[1] = sc * l3[1]; o[2] = sc * l3[2]; o[3] = sc * Jr[4]; o[4] = sc * Jr[5]; o[5] = sc * Jr[6];
  }
  for (int e = lane; e < 15 * 18; e += 64) {           // six 15x3 blocks
    const int row = e / 18, c = e % 18, blk = c / 3, cc = c % 3;   // blk 0..2 -> (v,ba,bg)_i ; 3..5 -> _j
    const int src = blk < 3 ? 1 + blk : 5 + (blk - 3);
    const double* jp = src == 1 ? J.j[1] : (src == 2 ? J.j[2] : (src == 3 ? J.j[3] : (src == 5 ? J.j[5] : (src == 6 ? J.j[6] : J.j[7]))));
    const double scv = ((pose_const[blk < 3 ? ki : kj] >> (1 + blk % 3)) & 1) ? 0.0 : 1.0;      // constant (v | ba | bg) block: bits 1..3 of the mask
    sJ[row * 30 + (blk < 3 ? 6 + 3 * blk : 21 + 3 * (blk - 3)) + cc] = scv * jp[(size_t)f * 45 + 3 * row + cc];
  }
  __syncthreads();
  double c = 0.0;
  if (lane < 15) c = 0.5 * sr[lane] * sr[lane];
  c = wave_sum(c);
  if (lane == 0) atomicAdd(cost + (blockIdx.x & (kStripes - 1)), c);
  for (int e = lane; e < 30 * 30; e += 64) {
    const int a = e / 30, b = e % 30;
    const int ga = sidx[a], gb = sidx[b];
    if (gb > ga || (ga == gb && a != b)) continue;    // lower triangle in GLOBAL indices (kf_i != kf_j is validated)
    double h = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) h += sJ[k * 30 + a] * sJ[k * 30 + b];
    atomicAdd(&B[(size_t)ga * ld + gb], h);
  }
  if (lane < 30) {
    double g = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) g += sJ[k * 30 + lane] * sr[k];
    atomicAdd(&gc[sidx[lane]], g);
  }
}

// the same accumulation for use inside a 256-thread workgroup: wave w of virtual block vb takes factor 4 vb + w
template <bool DYN = false>
__device__ __forceinline__ void lin_imu_body4(const int vb, int n, int n_kf, const double* __restrict__ res, ImuJ J, const int* __restrict__ kf_i,
                                              const int* __restrict__ kf_j, const double* __restrict__ poses,
                                              const uint8_t* __restrict__ pose_const, double* __restrict__ B, int ld,
                                              double* __restrict__ gc, double* __restrict__ cost) {
  __shared__ double sJ4[DYN ? 1 : 4 * 15 * 30];
  __shared__ double sr4[DYN ? 1 : 4 * 16];
  __shared__ int sidx4[DYN ? 1 : 4 * 32];
  extern __shared__ double lin_lds[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int f = 4 * vb + w;
  const bool active = f < n;
  double* sJ = (DYN ? lin_lds : sJ4) + w * 450;
  double* sr = (DYN ? lin_lds + 1800 : sr4) + w * 16;
  int* sidx = (DYN ? reinterpret_cast<int*>(lin_lds + 1864) : sidx4) + w * 32;
  if (active) {
    const int ki = kf_i[f], kj = kf_j[f];
    if (lane < 15) sr[lane] = res[(size_t)f * 15 + lane];
    if (lane < 30) {
      int g;
      if (lane < 6) g = 6 * ki + lane; else if (lane < 15) g = 6 * n_kf + 9 * ki + (lane - 6);
      else if (lane < 21) g = 6 * kj + (lane - 15); else g = 6 * n_kf + 9 * kj + (lane - 21);
      sidx[lane] = g;
    }
    for (int e = lane; e < 30; e += 64) {
      const int row = e % 15, which = e / 15;            // which: 0 = pose_i, 1 = pose_j
      const double* Jr = (which ? J.j[4] : J.j[0]) + (size_t)f * 105 + 7 * row;
      const int kk = which ? kj : ki;
      const double* q = poses + 7 * kk;
      const double sc = (pose_const[kk] & 1) ? 0.0 : 1.0;
      double l3[3];
      quat_row_to_local(Jr, q, l3);
      double* o = sJ + row * 30 + (which ? 15 : 0);
      o[0] = sc * l3[0]; o[1] = sc * l3[1]; o[2] = sc * l3[2]; o[3] = sc * Jr[4]; o[4] = sc * Jr[5]; o[5] = sc * Jr[6];
    }
    for (int e = lane; e < 15 * 18; e += 64) {           // six 15x3 blocks
      const int row = e / 18, c = e % 18, blk = c / 3, cc = c % 3;   // blk 0..2 -> (v,ba,bg)_i ; 3..5 -> _j
      const int src = blk < 3 ? 1 + blk : 5 + (blk - 3);
      // (static indices only: indexing the by-value pointer table with a run-time value puts it in scratch memory, and a kernel with a
      // scratch segment pays several microseconds of dispatch set-up)
      const double* jp = src == 1 ? J.j[1] : (src == 2 ? J.j[2] : (src == 3 ? J.j[3] : (src == 5 ? J.j[5] : (src == 6 ? J.j[6] : J.j[7]))));
      // a constant velocity / bias block (bits 1..3 of the keyframe's mask; Environment::Optimize holds all of them, environment.cpp:62-68) keeps
      // its residual but gets no Jacobian columns
      const double scv = ((pose_const[blk < 3 ? ki : kj] >> (1 + blk % 3)) & 1) ? 0.0 : 1.0;
      sJ[row * 30 + (blk < 3 ? 6 + 3 * blk : 21 + 3 * (blk - 3)) + cc] = scv * jp[(size_t)f * 45 + 3 * row + cc];
    }
  }
  __syncthreads();
  double c = 0.0;
  if (active && lane < 15) c = 0.5 * sr[lane] * sr[lane];
  c = wave_sum(c);
  if (!active) return;
  if (lane == 0) atomicAdd(cost + (f & (kStripes - 1)), c);
  for (int e = lane; e < 30 * 30; e += 64) {
    const int a = e / 30, b = e % 30;
    const int ga = sidx[a], gb = sidx[b];
    if (gb > ga || (ga == gb && a != b)) continue;    // lower triangle in GLOBAL indices (kf_i != kf_j is validated)
    double h = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) h += sJ[k * 30 + a] * sJ[k * 30 + b];
    atomicAdd(&B[(size_t)ga * ld + gb], h);
  }
  if (lane < 30) {
    double g = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) g += sJ[k * 30 + lane] * sr[k];
    atomicAdd(&gc[sidx[lane]], g);
  }
}

// ------------------------------------------------------------------------------------------------ linearisation, visual factors
// The TwoFrame (sorted fast path), TwoCamera and PoseOnly linearisations as ONE launch: workgroups [0, n_tfw) take the TwoFrame
// work list, the next g_tc the TwoCamera blocks, the rest the PoseOnly blocks.  The two small passes (4.6 + 8.8 us as launches of
// their own) disappear under the TwoFrame pass; all three only meet in B, gc, C, g_rho through atomics.  A fourth segment
// accumulates the ImuError blocks (four factors per workgroup) from the Jacobians k_imu<true> materialised just before.
// ImuError factors of the merged linearisation launch: EVALUATED and accumulated by the same workgroup (one factor per workgroup), so the linearisation needs no IMU launch ahead of it and the weighted Jacobian never leaves LDS:
//   stage sqrt_info -> one lane forms the raw residual and the 15 x 32 pre-weighting Jacobian -> all lanes weight them ->
//   pose columns to tangent coordinates -> J^T J / J^T r into B / gc, 1/2 |r|^2 into the cost.
// LDS (doubles): sS 225 | sM 480 (later the local 15 x 30 Jacobian) | sJw 480 | sr0 16 | sr 16 | sidx 16  = kImuWaveLds.
constexpr int kEndZeroWgs = 192;       // workgroups of the cost + decision launch that clear the accumulators
constexpr int kImuWaveLds = 225 + 480 + 480 + 16 + 16 + 16 + 248 + 32 + 2;      // + the pre-integration's head (OFF_COV doubles) + the two keyframes' states
__device__ __forceinline__ void lin_imu_eval_body(const int f, const ImuEvalArgs& I, int n_kf, const StateP& s, const uint8_t* __restrict__ pose_const,
                                                  double* __restrict__ B, int ld, double* __restrict__ gc, double* __restrict__ cost, unsigned long long* dbg) {
  // ONE factor per workgroup: the evaluation's serial part (one lane) is what it is, everything around it is spread over all kT threads
  extern __shared__ double lin_lds[];
  const int tid = threadIdx.x;
  if (f >= I.n) return;
  if (f != 0 || tid != 0) dbg = nullptr;               // LVF_LIN_TIMING: factor 0 stamps its phases
  if (dbg) dbg[0] = wall_clock64();
  double* sS = lin_lds;
  double* sM = sS + 225;
  double* sJw = sM + 480;
  double* sr0 = sJw + 480;
  double* sr = sr0 + 16;
  int* sidx = reinterpret_cast<int*>(sr + 16);
  double* sP = reinterpret_cast<double*>(sidx + 32);      // [248] what the one-lane section reads of the pre-integration
  double* sX = sP + 248;                                  // [32] pose, v, ba, bg of keyframe i (0..15) and j (16..31)
  // everything the one-lane section reads is staged by the whole workgroup first (as single-lane global reads they were two cold round trips of its 3.6 us)
  const int ki = I.kf_i[f], kj = I.kf_j[f];
  for (int k = tid; k < 225; k += kT) sS[k] = I.sqrt_info[(size_t)f * 225 + k];
  for (int k = tid; k < OFF_COV; k += kT) sP[k] = I.pre[(size_t)f * kPre + k];
  if (tid < 32) {
    const int kk = tid < 16 ? ki : kj, c = tid & 15;
    sX[tid] = c < 7 ? s.poses[7 * kk + c] : (c < 10 ? s.vel[3 * kk + c - 7] : (c < 13 ? s.ba[3 * kk + c - 10] : s.bg[3 * kk + c - 13]));
  }
  int* smask = reinterpret_cast<int*>(sX + 32);           // the two keyframes' constant-block masks
  if (tid >= 32 && tid < 34) smask[tid - 32] = pose_const[tid == 32 ? ki : kj];
  for (int k = tid; k < 480; k += kT) sM[k] = 0.0;
  __syncthreads();
  if (dbg) dbg[1] = wall_clock64();
  if (tid == 0) imu_raw16<true>(sP, sX, sr0, sM);
  __syncthreads();
  if (dbg) dbg[2] = wall_clock64();
  {
    const double r = imu_weighted_residual(tid, sS, sr0);
    if (tid < 15) sr[tid] = r;
    for (int e = tid; e < 480; e += kT) sJw[e] = imu_weighted_jacobian(e, sS, sM);
    if (tid < 30) {
      int g;
      if (tid < 6) g = 6 * ki + tid; else if (tid < 15) g = 6 * n_kf + 9 * ki + (tid - 6);
      else if (tid < 21) g = 6 * kj + (tid - 15); else g = 6 * n_kf + 9 * kj + (tid - 21);
      sidx[tid] = g;
    }
  }
  __syncthreads();                                   // sM is dead from here: it becomes the local Jacobian sJ[15][30]
  double* sJ = sM;
  if (tid < 30) {
    const int row = tid % 15, which = tid / 15;          // which: 0 = pose_i, 1 = pose_j
    const double* Jr = sJw + 32 * row + (which ? 16 : 0);
    const double sc = (smask[which] & 1) ? 0.0 : 1.0;
    double l3[3];
    quat_row_to_local(Jr, sX + 16 * which, l3);
    double* o = sJ + row * 30 + (which ? 15 : 0);
    o[0] = sc * l3[0]; o[1] = sc * l3[1]; o[2] = sc * l3[2]; o[3] = sc * Jr[4]; o[4] = sc * Jr[5]; o[5] = sc * Jr[6];
  }
  for (int e = tid; e < 15 * 18; e += kT) {              // six 15x3 blocks: (v, ba, bg)_i = columns 7..15, (v, ba, bg)_j = columns 23..31
    const int row = e / 18, c = e % 18;
    const int cg = c < 9 ? c / 3 : (c - 9) / 3;          // 0 = v, 1 = ba, 2 = bg
    const double scv = ((smask[c < 9 ? 0 : 1] >> (1 + cg)) & 1) ? 0.0 : 1.0;      // constant (v | ba | bg) block: bits 1..3 of the mask
    sJ[row * 30 + (c < 9 ? 6 + c : 21 + (c - 9))] = scv * sJw[32 * row + (c < 9 ? 7 + c : 23 + (c - 9))];
  }
  __syncthreads();
  if (dbg) dbg[3] = wall_clock64();
  if (tid < 64) {
    double c = tid < 15 ? 0.5 * sr[tid] * sr[tid] : 0.0;
    c = wave_sum(c);
    if (tid == 0) atomicAdd(cost + (f & (kStripes - 1)), c);
  }
  for (int e = tid; e < 30 * 30; e += kT) {
    const int a = e / 30, b = e % 30;
    const int ga = sidx[a], gb = sidx[b];
    if (gb > ga || (ga == gb && a != b)) continue;    // lower triangle in GLOBAL indices (kf_i != kf_j is validated)
    double h = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) h += sJ[k * 30 + a] * sJ[k * 30 + b];
    atomicAdd(&B[(size_t)ga * ld + gb], h);
  }
  if (tid < 30) {
    double g = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) g += sJ[k * 30 + tid] * sr[k];
    atomicAdd(&gc[sidx[tid]], g);
  }
  if (dbg) dbg[4] = wall_clock64();
}

struct LinVisual {
  int n_tfw, g_tc;
  // TwoFrame
  GP<const TfWork> work; GP<const double2> tf_fo, tf_ob; GP<const int> tf_lm, tf_k1; CamD tf_left, tf_right; int unique_lk2; TfCompact cp;
  // TwoCamera
  int n_tc; GP<const double2> tc_lo, tc_ro; GP<const int> tc_lm, tc_kf; GP<const double> tc_w; CamD tc_left, tc_right;
  // PoseOnly
  int n_po, g_po; GP<const double2> po_ob; GP<const int> po_kf, po_pwi; GP<const double> po_pw; CamD po_cam;
  // ImuError: evaluated inside the launch (imu.pre != nullptr) or ahead of it by k_imu<true> (imu_res / imu_J)
  int n_imu; GP<const double> imu_res; ImuJ imu_J; GP<const int> imu_i, imu_j; ImuEvalArgs imu;
};
struct LinArgs {
  LinVisual v; int n_kf; StateP s; double huber; GP<const uint8_t> pose_const; GP<double> B; int ld; GP<double> gc; GP<double> E; int ldE; GP<double> C, gr, cost;
  int nblocks; GP<const int> done; GP<unsigned long long> dbg; int rows;
  GP<double> scal_reset;      // early sparse levels: the per-step scalars and the fail flag are reset HERE (the levels start before k_prepare, which resets them otherwise)
};
__device__ __forceinline__ void reset_step_scalars(double* scal) {
  for (int k = SC_COST_NEW + threadIdx.x; k < SC_N; k += kT) scal[k] = 0.0;
  if (threadIdx.x == 0) { *reinterpret_cast<int*>(scal + SC_FAIL) = 0; *reinterpret_cast<int*>(scal + SC_TICKET) = 0; }
}
__device__ __forceinline__ void lin_visual_body(const int bx, const LinArgs& A) {
  if (bx >= A.nblocks || (A.done && *A.done)) return;
  if (bx == 0 && A.scal_reset) reset_step_scalars(A.scal_reset);
  const LinVisual& a = A.v;
  // the ImuError workgroups come FIRST: each is a ~12 us chain with a one-lane section, and dispatched last (of the last window of a
  // batch) it would stick out behind everything else
  const int g_imu_first = a.imu.pre ? a.n_imu : 0;
  const int b = bx - g_imu_first;
  const int n_kf = A.n_kf; const StateP s = A.s; const double huber = A.huber; const uint8_t* pose_const = A.pose_const;
  double* B = A.B; const int ld = A.ld; double* gc = A.gc; double* E = A.E; const int ldE = A.ldE; double* C = A.C; double* gr = A.gr; double* cost = A.cost;
  if (bx < g_imu_first)
    lin_imu_eval_body(bx, a.imu, n_kf, s, pose_const, B, ld, gc, cost, A.dbg ? A.dbg + (size_t)a.n_tfw * 8 : nullptr);
  else if (b < a.n_tfw)
    lin_tf_sorted_body<true>(b, a.work, n_kf, a.tf_fo, a.tf_ob, a.tf_lm, a.tf_k1, s, a.tf_left, a.tf_right, huber, pose_const, B, ld, gc, E, ldE, C, gr, cost,
                       a.unique_lk2, A.dbg, a.cp, a.n_tfw);
  else if (b < a.n_tfw + a.g_tc)
    lin_tc_body<false>(b - a.n_tfw, a.n_tc, a.tc_lo, a.tc_ro, a.tc_lm, a.tc_kf, a.tc_w, s, a.tc_left, a.tc_right, huber, C, gr, cost);
  else if (b < a.n_tfw + a.g_tc + a.g_po)
    lin_po_body<false, true>(b - a.n_tfw - a.g_tc, a.n_po, n_kf, a.po_ob, a.po_kf, a.po_pwi, a.po_pw, s, a.po_cam, huber, pose_const, B, ld, gc, cost);
  else
    lin_imu_body4<true>(b - a.n_tfw - a.g_tc - a.g_po, a.n_imu, n_kf, a.imu_res, a.imu_J, a.imu_i, a.imu_j, s.poses, pose_const, B, ld, gc, cost);
}
// (three workgroups per CU: the register allocator is told so — left alone it lands one VGPR above the limit)
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(3))) void k_lin_visual(LinArgs a) { lin_visual_body(blockIdx.x, a); }
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(3))) void k_lin_visual_b(const LinArgs* __restrict__ t) { lin_visual_body(blockIdx.x, t[blockIdx.y]); }
__global__ __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(3))) void k_lin_visual_bt(const LinArgs* __restrict__ t) { lin_visual_body(blockIdx.y, t[blockIdx.x]); }

// Adds the TwoFrame slabs of a linearisation into B / gc (compact mode).  Workgroups are sorted by current keyframe: run(k) =
// workgroups [run_first[k], run_first[k+1]).  Every entry of B has ONE owner thread here (plain read-modify-write; the other factor
// types' atomic contributions were complete when the linearisation launch ended):
//   workgroups [0, n_kf)         keyframe k: B[k,k] (21) and g[k] (6) = sum of slabQ over run(k) (k as current keyframe)
//                                                                      + sum of slabP[.][k][0..27) over every later workgroup (k as first keyframe)
//   the rest                     one thread per (k2, k1 < k2, entry of the 6x6 cross block) = sum of slabP[.][k1][27..63) over run(k2)
struct TfReduceArgs {
  int n_kf, n_wg; GP<const int> run_first; GP<const double> slabP, slabQ; GP<double> B; int ld; GP<double> gc; int nblocks; GP<const int> done;
  int own_blocks; SpArgs ride;       // workgroups [own_blocks, nblocks): a sparse level riding in this launch (early form)
};
__device__ __forceinline__ void tf_reduce_body(const int bx0, const TfReduceArgs& A) {
  if (bx0 >= A.nblocks || (A.done && *A.done)) return;
  // (the riding level takes the FIRST workgroups: it is the longest piece of the launch, and in a batch the first workgroups of every
  // window are dispatched first — see the transposed table launches)
  if (bx0 < A.ride.nblocks) { sp_ride(bx0, A.ride); return; }
  const int bx = bx0 - A.ride.nblocks;
  const int n_kf = A.n_kf;
  const int nchunk = (A.n_wg + 63) / 64;
  if (bx < n_kf * nchunk) {
    // keyframe k, chunk c of 64 workgroups: 8 groups of 32 lanes (lane v < 27 owns one value), 8 slab rows per thread, all requested at
    // once; the chunk's 27 sums go to B / gc with one atomic each (n_kf x nchunk x 27 per linearisation)
    __shared__ double part[8][32];
    const int k = bx / nchunk, c = bx - k * nchunk, v = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int r0 = A.run_first[k], r1 = A.run_first[k + 1];
    double acc = 0.0;
    if (v < 27) {
      // the eight slab values are requested together — the row (P or Q slab) chosen by address, the workgroup index clamped — and added in
      // order afterwards: as `if (wg >= r1) acc += P[..]; else if (wg >= r0) acc += Q[..]` every load sat in its own branch behind the
      // previous addition, eight dependent round trips per thread
      const double* slabP = A.slabP; const double* slabQ = A.slabQ;
      double t8[8]; bool keep[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int wg = 64 * c + g + 8 * u, wgc = min(wg, A.n_wg - 1);
        const bool useP = wgc >= r1;               // k as the blocks' first keyframe (later workgroups); else k as their current keyframe
        keep[u] = (wg < A.n_wg) & (wgc >= r0);
        const double* src = useP ? slabP + ((size_t)wgc * n_kf + k) * kSlabRow + v : slabQ + (size_t)wgc * kSlabQ + v;
        t8[u] = *src;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = keep[u] ? acc + t8[u] : acc;
    }
    part[g][v] = acc;
    __syncthreads();
    if (threadIdx.x < 27) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += part[q][threadIdx.x];
      if (t != 0.0) {
        if (threadIdx.x < 21) {
          int x = 0, rem = threadIdx.x;
          while (rem > x) { rem -= x + 1; ++x; }
          atomicAdd(&A.B[(size_t)(6 * k + x) * A.ld + 6 * k + rem], t);
        } else atomicAdd(&A.gc[6 * k + threadIdx.x - 21], t);
      }
    }
    return;
  }
  const int e = (bx - n_kf * nchunk) * kT + threadIdx.x;
  const int npair = n_kf * (n_kf - 1) / 2;
  if (e >= npair * 36) return;
  const int pr = e / 36, el = e - 36 * pr;
  int k2 = (int)((1.0 + sqrt(1.0 + 8.0 * pr)) * 0.5);                 // pr = k2 (k2 - 1) / 2 + k1, k1 < k2
  while (k2 * (k2 - 1) / 2 > pr) --k2;
  while ((k2 + 1) * k2 / 2 <= pr) ++k2;
  const int k1 = pr - k2 * (k2 - 1) / 2;
  double acc = 0.0;
  {
    const double* slabP = A.slabP;
    const int w0 = A.run_first[k2], w1 = A.run_first[k2 + 1];
    for (int wg = w0; wg < w1; wg += 4) {            // four rows per round trip (index clamped), added in order
      double t4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t4[u] = slabP[((size_t)min(wg + u, w1 - 1) * n_kf + k1) * kSlabRow + 27 + el];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = (wg + u < w1) ? acc + t4[u] : acc;
    }
  }
  if (acc != 0.0) {
    const int x = el / 6, y = el - 6 * x;                              // x: k2 tangent index, y: k1 tangent index
    A.B[(size_t)(6 * k2 + x) * A.ld + 6 * k1 + y] += acc;
  }
}
__global__ __launch_bounds__(kT) void k_tf_reduce(TfReduceArgs a) { tf_reduce_body(blockIdx.x, a); }
__global__ __launch_bounds__(kT) void k_tf_reduce_b(const TfReduceArgs* __restrict__ t) { tf_reduce_body(blockIdx.x, t[blockIdx.y]); }
// (transposed: blockIdx.x = window.  Workgroups are dispatched x-fastest, so the first workgroups of EVERY window — the riding / chained
// sparse levels, the ImuError factors — start together at the head of the launch instead of each window's behind the previous window's bulk)
__global__ __launch_bounds__(kT) void k_tf_reduce_bt(const TfReduceArgs* __restrict__ t) { tf_reduce_body(blockIdx.y, t[blockIdx.x]); }


// ------------------------------------------------------------------------------------------------ pose priors
// consumes the materialised PoseGraphError / PoseError outputs of launch_pose_prior (res[n][6], ja/jb [n][6][7]); no loss
// function (backend.cpp:171,176).  <= n_kf blocks: one thread per block, global atomics.
// (res, ja, jb: THIS block's 6 residuals and 6 x 7 ambient Jacobian rows — global arrays of launch_pose_prior, or the thread's own copies)
__device__ __forceinline__ void lin_prior_block(const int i, const int a, const int b, const double* __restrict__ res, const double* __restrict__ ja,
                                                const double* __restrict__ jb, const double* __restrict__ poses, const uint8_t* __restrict__ pose_const,
                                                double* __restrict__ B, int ld, double* __restrict__ gc, double* __restrict__ cost) {
  double r[6], La[36], Lb[36];   // local Jacobians, row-major 6 x 6
  double c = 0.0;
  for (int k = 0; k < 6; ++k) { r[k] = res[k]; c += 0.5 * r[k] * r[k]; }
  for (int k = 0; k < 6; ++k) {
    const double* row = jb + 7 * k;
    const double sc = (pose_const[b] & 1) ? 0.0 : 1.0;
    double l3[3];
    quat_row_to_local(row, poses + 7 * b, l3);
    Lb[6 * k] = sc * l3[0]; Lb[6 * k + 1] = sc * l3[1]; Lb[6 * k + 2] = sc * l3[2];
    Lb[6 * k + 3] = sc * row[4]; Lb[6 * k + 4] = sc * row[5]; Lb[6 * k + 5] = sc * row[6];
    if (a >= 0) {
      const double* rowa = ja + 7 * k;
      const double sa = (pose_const[a] & 1) ? 0.0 : 1.0;
      quat_row_to_local(rowa, poses + 7 * a, l3);
      La[6 * k] = sa * l3[0]; La[6 * k + 1] = sa * l3[1]; La[6 * k + 2] = sa * l3[2];
      La[6 * k + 3] = sa * rowa[4]; La[6 * k + 4] = sa * rowa[5]; La[6 * k + 5] = sa * rowa[6];
    }
  }
  atomicAdd(cost + (i & (kStripes - 1)), c);
  for (int x = 0; x < 6; ++x) {
    double g = 0.0;
    for (int k = 0; k < 6; ++k) g += Lb[6 * k + x] * r[k];
    atomicAdd(&gc[6 * b + x], g);
    for (int y = 0; y <= x; ++y) {
      double h = 0.0;
      for (int k = 0; k < 6; ++k) h += Lb[6 * k + x] * Lb[6 * k + y];
      atomicAdd(&B[(size_t)(6 * b + x) * ld + 6 * b + y], h);
    }
  }
  if (a >= 0) {
    for (int x = 0; x < 6; ++x) {
      double g = 0.0;
      for (int k = 0; k < 6; ++k) g += La[6 * k + x] * r[k];
      atomicAdd(&gc[6 * a + x], g);
      for (int y = 0; y <= x; ++y) {
        double h = 0.0;
        for (int k = 0; k < 6; ++k) h += La[6 * k + x] * La[6 * k + y];
        atomicAdd(&B[(size_t)(6 * a + x) * ld + 6 * a + y], h);
      }
      for (int y = 0; y < 6; ++y) {   // cross block, stored in the lower triangle of B
        double h = 0.0;
        for (int k = 0; k < 6; ++k) h += La[6 * k + x] * Lb[6 * k + y];
        if (a > b) atomicAdd(&B[(size_t)(6 * a + x) * ld + 6 * b + y], h);
        else atomicAdd(&B[(size_t)(6 * b + y) * ld + 6 * a + x], h);
      }
    }
  }
}
__global__ __launch_bounds__(64) void k_lin_prior(int n, const double* __restrict__ res, const double* __restrict__ ja,
                                                  const double* __restrict__ jb, const int* __restrict__ kf_a, const int* __restrict__ kf_b,
                                                  const double* __restrict__ poses, const uint8_t* __restrict__ pose_const,
                                                  double* __restrict__ B, int ld, double* __restrict__ gc, double* __restrict__ cost) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  lin_prior_block(i, kf_a[i], kf_b[i], res + 6 * i, ja + (size_t)42 * i, jb + (size_t)42 * i, poses, pose_const, B, ld, gc, cost);
}
// The same with the evaluation inside (prior_eval.hpp): the block's residuals and ambient Jacobians never leave the thread — one launch
// instead of k_pose_prior + k_lin_prior on the LM loop's path (a window with weak frames pays it every iteration), nothing materialised.
struct PriorArgs { int n; GP<const int> kf_a, kf_b; GP<const double> target, weight, vv; };
__global__ __launch_bounds__(64) void k_prior_lin(PriorArgs P, const double* __restrict__ poses, const uint8_t* __restrict__ pose_const,
                                                  double* __restrict__ B, int ld, double* __restrict__ gc, double* __restrict__ cost) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= P.n) return;
  double res[6], ja[42], jb[42];
  const int a = P.kf_a[i], b = P.kf_b[i];
  pose_prior_eval<true>(a, b, P.target + 7 * i, P.weight[i], P.vv[i], poses, res, ja, jb);
  lin_prior_block(i, a, b, res, ja, jb, poses, pose_const, B, ld, gc, cost);
}
// 1/2 |r|^2 of every prior block at `poses` (the candidate), added to the striped cost: k_pose_prior<false> + k_cost_sq in one launch
__global__ __launch_bounds__(64) void k_prior_cost(PriorArgs P, const double* __restrict__ poses, double* __restrict__ cost) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= P.n) return;
  double res[6];
  pose_prior_eval<false>(P.kf_a[i], P.kf_b[i], P.target + 7 * i, P.weight[i], P.vv[i], poses, res, nullptr, nullptr);
  double c = 0.0;
  for (int k = 0; k < 6; ++k) c += 0.5 * res[k] * res[k];
  atomicAdd(cost + (i & (kStripes - 1)), c);
}
__global__ __launch_bounds__(kT) void k_cost_sq(int n, const double* __restrict__ res, double* __restrict__ cost) {
  const int i = blockIdx.x * kT + threadIdx.x;
  double c = 0.0;
  if (i < n) { const double r = res[i]; c = 0.5 * r * r; }
  block_add(c, cost);
}

// ------------------------------------------------------------------------------------------------ damping / assembly
// (the fp64 square root and two divisions of the literal form cost k_prepare / k_tf_reduce / k_step_tail 6.6 us per iteration between them
// when every diagonal entry took them — profiles/r05_a; they are only needed where the clamp can act:  s^2 h >= 1e-6  <=  h >= 2e-6 (1 + h0),
// because (1 + sqrt(h0))^2 <= 2 (1 + h0); there clamp(s^2 h) / s^2 = h up to one rounding.  A column that was empty at iteration 0 has s = 1.)
__device__ __forceinline__ double lm_damping(double h, double h0) {
  if (h >= 2e-6 * (1.0 + h0) && h < 1e32) return h;
  if (h0 == 0.0) return fmin(fmax(h, 1e-6), 1e32);
  const double sj = 1.0 / (1.0 + sqrt(h0)), s2 = sj * sj;
  return fmin(fmax(h * s2, 1e-6), 1e32) / s2;
}
// the damping of unknown `slot` whose diagonal entry is h in this pass; the thread that OWNS the entry's assembly records H0 in the first
// pass.  h0_loaded = jac.h0[slot], requested by the caller together with its other loads (so that it is not a dependent round trip here).
__device__ __forceinline__ double lm_damping_own(double h, const JacobiDev& j, int slot, int frozen, double h0_loaded) {
  double h0 = h0_loaded;
  if (!frozen) { h0 = h; j.h0[slot] = h; }
  return lm_damping(h, h0);
}

// One launch prepares the damped system of a step:
//   blocks [0, nS_blocks)      : S (lower, in ELIMINATION order: S row I holds unknown iperm[I]) = B (lower, natural order) + Dc on
//                                the diagonal; augmented row (iperm = -2) = -gc ; padding rows (iperm = -1) = identity
//   blocks [nS_blocks, ...)    : Cd = C + clamp(C)/radius ; E[l][dp] = gr[l] (the extra column that makes the SYRK also
//                                produce E^T Cd^-1 g_rho)
//   block 0 / thread 0         : resets the per-step scalars (candidate cost, model change, norms) and the Cholesky fail flag
struct PrepArgs {
  int ld, dpad, jl0 /* = d: the first landmark slot of jac.h0 */; GP<const int> iperm; GP<const double> B, gc; GP<const double> radius; GP<double> S; unsigned nS_blocks; int n_lm, dp, ldE; GP<const double> C, gr;
  GP<double> Cd, E, scal; int nblocks; GP<const int> done;
  // atomic-free mode (slotB != nullptr): per-landmark totals from the slot records
  GP<const int> eoff, kmin, kmax; GP<const double> slotB; GP<double> Ct, grt;
  // early form (early != 0): S was cleared with the accumulators and sparse levels may already have added into the dense corner, so the
  // corner's entries (rows / columns >= off) are ADDED, and the columns of the sparse blocks are left alone (the levels form them themselves)
  int early, off;
  int own_blocks; SpArgs ride;       // workgroups [own_blocks, nblocks): a sparse level riding in this launch
  JacobiDev jac;
};
__device__ __forceinline__ void prepare_body(const unsigned bx0, const PrepArgs& A) {
  if (bx0 >= (unsigned)A.nblocks || (A.done && *A.done)) return;
  if (bx0 < (unsigned)A.ride.nblocks) { sp_ride((int)bx0, A.ride); return; }      // (riders first: tf_reduce_body)
  const unsigned bx = bx0 - (unsigned)A.ride.nblocks;
  const int ld = A.ld, dpad = A.dpad; const int* __restrict__ iperm = A.iperm; const double* __restrict__ B = A.B; const double* __restrict__ gc = A.gc;
  const double inv_radius = 1.0 / *A.radius;
  const int jf = *A.jac.frozen;
  double* __restrict__ S = A.S; const unsigned nS_blocks = A.nS_blocks; const int n_lm = A.n_lm, dp = A.dp, ldE = A.ldE;
  const double* __restrict__ C = A.C; const double* __restrict__ gr = A.gr; double* __restrict__ Cd = A.Cd; double* __restrict__ E = A.E; double* __restrict__ scal = A.scal;
  if (bx == 0 && scal) reset_step_scalars(scal);
  if (bx >= nS_blocks) {
    if (A.slotB) {
      // atomic-free mode: C, g_rho of the landmark = its TwoCamera part (C, gr: atomics of the linearisation) + its slot records; the k1
      // columns of its E row = the sum of the records' first-keyframe parts.  8 lanes per landmark (lane j takes slots j, j + 8, ...: the
      // loads of a track are issued together instead of one dependent loop), DPP sum over the 8 lanes.  The totals go to separate arrays
      // so the pass can be repeated (lvf_problem_download_reduced).
      const int l = ((bx - nS_blocks) * kT + threadIdx.x) >> 3, j0 = threadIdx.x & 7;
      const bool live = l < n_lm;
      const int lc = live ? l : 0;
      const int k1 = A.kmin[lc], len = live ? max(0, A.kmax[lc] - k1) : 0;
      const double h0l = A.jac.h0[A.jl0 + lc];
      const double2* sb = reinterpret_cast<const double2*>(A.slotB + (size_t)A.eoff[lc] * 8);
      double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int j = j0; j < len; j += 8) {
        const double2 b0 = sb[4 * j], b1 = sb[4 * j + 1], b2 = sb[4 * j + 2], cg = sb[4 * j + 3];
        v[0] += cg.x; v[1] += cg.y; v[2] += b0.x; v[3] += b0.y; v[4] += b1.x; v[5] += b1.y; v[6] += b2.x; v[7] += b2.y;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) { v[q] = quad_sum(v[q]); v[q] += quad_perm<0x141>(v[q]); }      // sum over the aligned group of 8 lanes
      if (live && j0 == 0) {
        const double c = C[l] + v[0], g = gr[l] + v[1];
        A.Ct[l] = c; A.grt[l] = g;
        Cd[l] = c + lm_damping_own(c, A.jac, A.jl0 + l, jf, h0l) * inv_radius;
        double* el = E + (size_t)l * ldE;
        el[dp] = g;
        if (len > 0) {
#pragma unroll
          for (int q = 0; q < 6; ++q) el[6 * k1 + q] = v[2 + q];
        }
      }
      return;
    }
    const int l = (bx - nS_blocks) * kT + threadIdx.x;
    if (l >= n_lm) return;
    const double c = C[l], h0l = A.jac.h0[A.jl0 + l];
    Cd[l] = c + lm_damping_own(c, A.jac, A.jl0 + l, jf, h0l) * inv_radius;
    E[(size_t)l * ldE + dp] = gr[l];
    return;
  }
  // Only the LOWER triangle is assembled (nothing downstream reads an entry right of the diagonal as a value: the factorisations work
  // on lower tiles, and what they carry in the upper halves of diagonal tiles only ever feeds those same entries).  The triangle is
  // folded into a rectangle so that consecutive threads still write consecutive entries of a row: rectangle row q holds matrix row q
  // (columns 0..q) followed by matrix row ld-1-q (columns 0..ld-1-q), ld + 1 entries in all.
  const size_t e = (size_t)bx * kT + threadIdx.x;
  const int n0 = A.early ? A.off : 0, nn = ld - n0;                   // early form: the dense corner only
  const int half = (nn + 1) / 2;
  if (e >= (size_t)half * (nn + 1)) return;
  const int q = (int)(e / (nn + 1)), cc = (int)(e % (nn + 1));
  int I, J;
  if (cc <= q) { I = q; J = cc; }
  else { I = nn - 1 - q; J = cc - q - 1; if (I == q) return; }      // (odd size: the middle row is its own partner)
  I += n0; J += n0;
  const int oi = iperm[I], oj = iperm[J];
  double v = 0.0;
  if (oi >= 0) {
    if (oj >= 0 && J <= I) {
      const double h0d = I == J ? A.jac.h0[oi] : 0.0;
      v = B[(size_t)max(oi, oj) * dpad + min(oi, oj)];
      if (I == J) v += lm_damping_own(v, A.jac, oi, jf, h0d) * inv_radius;
    }
  } else if (oi == -2) {
    v = (oj >= 0) ? -gc[oj] : (J == I ? 1e300 : 0.0);   // huge corner keeps the augmented matrix positive definite
  } else if (I == J) {
    v = 1.0;
  }
  if (!A.early) S[(size_t)I * ld + J] = v;
  else if (v != 0.0) atomicAdd(&S[(size_t)I * ld + J], v);
}
__global__ __launch_bounds__(kT) void k_prepare(PrepArgs a) { prepare_body(blockIdx.x, a); }
__global__ __launch_bounds__(kT) void k_prepare_b(const PrepArgs* __restrict__ t) { prepare_body(blockIdx.x, t[blockIdx.y]); }
__global__ __launch_bounds__(kT) void k_prepare_bt(const PrepArgs* __restrict__ t) { const PrepArgs a = t[blockIdx.x]; prepare_body(blockIdx.y, a); }

// ------------------------------------------------------------------------------------------------ Schur reduce (MFMA f64)
// T = Ea^T diag(1/Cd) Ea with Ea = [E | g_rho] (n_lm x ldE).  One wave per (16x16 output tile, K-chunk); tiles on or
// below the diagonal only.  v_mfma_f64_16x16x4_f64: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// D: col = lane&15, row = (lane>>4) + 4*reg.   S[i][j] -= T[i][j] (i,j < dp);  S[d][i] += T[dp][i].
constexpr int kSchurChunk = 512;
__global__ __launch_bounds__(64) void k_schur_syrk(int n_lm, int dp, int ldE, int ntile, const double* __restrict__ E,
                                                   const double* __restrict__ Cd, int d, int ldS, double* __restrict__ S) {
  // decode lower-triangular tile index
  int t = blockIdx.x, ti = 0;
  while (t >= ti + 1) { t -= ti + 1; ++ti; }
  const int tj = t;
  const int lane = threadIdx.x, lk = lane >> 4, lc = lane & 15;
  const int k_begin = blockIdx.y * kSchurChunk, k_end = min(n_lm, k_begin + kSchurChunk);
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = k_begin; k0 < k_end; k0 += 4) {
    const int l = k0 + lk;
    double a = 0.0, b = 0.0;
    if (l < k_end) {
      const double* row = E + (size_t)l * ldE;
      a = row[ti * 16 + lc] / Cd[l];
      b = row[tj * 16 + lc];
    }
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = ti * 16 + lk + 4 * r, gj = tj * 16 + lc;
    const double v = acc[r];
    if (v == 0.0) continue;
    if (gi < dp && gj < dp) { if (gj <= gi) atomicAdd(&S[(size_t)gi * ldS + gj], -v); }
    else if (gi == dp && gj < dp) atomicAdd(&S[(size_t)d * ldS + gj], v);
  }
}

// LDS-staged variant (used whenever ldE <= 320, i.e. up to 53 keyframes; larger windows fall back to k_schur_syrk): workgroup = (tile group g of kSchurGroups, K slice).  The slice's rows
// of Ea are streamed through LDS in 16-row chunks with fully coalesced loads (the next chunk is prefetched into registers while
// the current one feeds the matrix cores); every wave keeps up to kSchurTilesPerWave 16x16 accumulators in registers across
// the whole slice and the workgroup touches S once at the end.  Versus one wave per (tile, 512-row chunk) with strided 8-byte
// global operand loads: the same MFMA count, 1/8 of the atomics, and E is read once per tile group instead of once per tile.
constexpr int kSchurGroups = 8, kSchurTilesPerWave = 8, kSchurRows = 16;
__global__ __launch_bounds__(256) void k_schur_lds(int n_lm, int dp, int ldE, int ntile, int rows_per_slice, const double* __restrict__ E,
                                                   const double* __restrict__ Cd, int d, int ldS, double* __restrict__ S) {
  extern __shared__ double sh[];          // Es[kSchurRows][ldE] | icd[kSchurRows]
  double* Es = sh;
  double* icd = sh + kSchurRows * ldE;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lk = lane >> 4, lc = lane & 15;
  const int g = blockIdx.x, slice = blockIdx.y;
  const int k_begin = slice * rows_per_slice, k_end = min(n_lm, k_begin + rows_per_slice);
  // this wave's tiles: t = g + kSchurGroups * (w + 4 * s), s = 0..; decode (ti, tj) of the lower-triangular enumeration
  int ti[kSchurTilesPerWave], tj[kSchurTilesPerWave], nt = 0;
#pragma unroll
  for (int s_ = 0; s_ < kSchurTilesPerWave; ++s_) {
    const int t = g + kSchurGroups * (w + 4 * s_);
    ti[s_] = 0; tj[s_] = 0;
    if (t < ntile) {
      int a = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
      while ((a + 1) * (a + 2) / 2 <= t) ++a;
      while (a * (a + 1) / 2 > t) --a;
      ti[s_] = a; tj[s_] = t - a * (a + 1) / 2;
      nt = s_ + 1;
    }
  }
  double4_t acc[kSchurTilesPerWave];
#pragma unroll
  for (int s_ = 0; s_ < kSchurTilesPerWave; ++s_) acc[s_] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int per_row = ldE;                 // doubles per staged row
  const int total = kSchurRows * per_row;  // elements per chunk
  constexpr int kPf = 20;                  // prefetch registers per thread: 256 * 20 >= 16 * 304 (ldE <= 320 on this path; larger ldE loops)
  double pf[kPf];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < kPf; ++u) {
      const int e = tid + 256 * u;
      double v = 0.0;
      if (e < total) { const int rr = e / per_row, cc = e - rr * per_row; if (k0 + rr < k_end) v = E[(size_t)(k0 + rr) * ldE + cc]; }
      pf[u] = v;
    }
  };
  auto fetch_tail = [&](int k0) {         // elements beyond 256 * kPf (only when ldE > 320): straight to LDS
    for (int e = tid + 256 * kPf; e < total; e += 256) {
      const int rr = e / per_row, cc = e - rr * per_row;
      Es[e] = (k0 + rr < k_end) ? E[(size_t)(k0 + rr) * ldE + cc] : 0.0;
    }
  };
  if (k_begin < k_end) fetch(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += kSchurRows) {
    __syncthreads();                       // the previous chunk has been consumed
#pragma unroll
    for (int u = 0; u < kPf; ++u) { const int e = tid + 256 * u; if (e < total) Es[e] = pf[u]; }
    fetch_tail(k0);
    if (tid < kSchurRows) icd[tid] = (k0 + tid < k_end) ? 1.0 / Cd[k0 + tid] : 0.0;
    __syncthreads();
    if (k0 + kSchurRows < k_end) fetch(k0 + kSchurRows);   // in flight while the matrix cores work
#pragma unroll
    for (int kk = 0; kk < kSchurRows; kk += 4) {
      const double* row = Es + (kk + lk) * per_row;
      const double wgt = icd[kk + lk];
#pragma unroll
      for (int s_ = 0; s_ < kSchurTilesPerWave; ++s_)
        if (s_ < nt) acc[s_] = __builtin_amdgcn_mfma_f64_16x16x4f64(row[16 * ti[s_] + lc] * wgt, row[16 * tj[s_] + lc], acc[s_], 0, 0, 0);
    }
  }
#pragma unroll
  for (int s_ = 0; s_ < kSchurTilesPerWave; ++s_) {
    if (s_ >= nt) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = ti[s_] * 16 + lk + 4 * r, gj = tj[s_] * 16 + lc;
      const double v = acc[s_][r];
      if (v == 0.0) continue;
      if (gi < dp && gj < dp) { if (gj <= gi) atomicAdd(&S[(size_t)gi * ldS + gj], -v); }
      else if (gi == dp && gj < dp) atomicAdd(&S[(size_t)d * ldS + gj], v);
    }
  }
}
// ---- band-limited variant.  A landmark's row of E is non-zero only at the keyframes that observe it, a contiguous track
// [kmin_l, kmax_l] of the window.  Landmarks are ordered by (track-length class, kmin) once per problem (k_lm_range + k_lm_sort),
// so a slice of kBandRows consecutive landmarks touches a narrow BAND of 16-column tiles [t0, t1]: the workgroup stages only
// those columns (plus the tile holding the g_rho column) and only forms the band's lower-triangular tiles.  At configs[3]
// that is ~1/5 of the MFMAs and ~1/30 of the E bytes of the dense SYRK.  Correctness never depends on the ordering: the
// band of each slice is computed from the actual kmin/kmax of its rows.
constexpr int kBandTilesPerWave = 8;        // output tiles (16 x 16 accumulators) a wave of the band Schur complement carries.  (Measured round 4, same box: 16 — one
// workgroup covers most slices' whole band, E read once instead of ~2x — needs > 256 VGPRs: 0.204 -> 0.254 ms / iteration spilling under the two-waves-per-SIMD
// attribute below, 0.206 / 8 windows 0.375 -> 0.435 ms with one wave per SIMD; 4 — more, smaller workgroups — 8 windows 0.43 -> 0.59 ms.  The launch is bound by
// how many workgroups overlap their fetch -> LDS -> matrix-core chains, not by E's bytes.)
constexpr int kBandRows = 64, kBandRowsMax = 256, kBandTilesPerGroup = 4 * kBandTilesPerWave;     // rows per slice: 64 for one window, up to 256 in a batch (fewer output atomics)
__global__ __launch_bounds__(kT) void k_lm_range(int n, const int* __restrict__ lm, const int* __restrict__ k1, const int* __restrict__ k2,
                                                 int* __restrict__ kmin, int* __restrict__ kmax) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const int l = lm[i], a = min(k1[i], k2[i]), b = max(k1[i], k2[i]);
  atomicMin(&kmin[l], a); atomicMax(&kmax[l], b);
}
__global__ __launch_bounds__(kT) void k_lm_range_init(int n_lm, int* __restrict__ kmin, int* __restrict__ kmax) {
  const int l = blockIdx.x * kT + threadIdx.x;
  if (l < n_lm) { kmin[l] = 0x7fffffff; kmax[l] = -1; }
}
__device__ __forceinline__ int lm_sort_key(int kmin, int kmax, int n_kf) {
  // (branch-free on purpose: with `if (kmax < 0) return ..` in front, the compiler sinks the caller's kmin load into the branch behind the
  // kmax load's wait — two dependent round trips per landmark instead of one)
  const int len = kmax - kmin + 1;
  const int cls = (len > 8) + (len > 16) + (len > 32);
  const int none = kmax >> 31;                          // all ones: no pose-dependent block — nothing to eliminate, ordered last
  return (none & (4 * n_kf)) | (~none & (cls * n_kf + kmin));
}
// counting sort by key, one workgroup (n_lm is ~1e4; 4 n_kf + 1 buckets in LDS)
// (per-wave copies of the histogram — 16x fewer lanes per address — measured SLOWER, 15.7 vs 13.6 us: the two passes are bound by their
// dependent load -> LDS atomic round trips, not by address conflicts)
__device__ __forceinline__ void lm_sort_body(int n_lm, int n_kf, const int* __restrict__ kmin, const int* __restrict__ kmax,
                                             int* __restrict__ order, int* __restrict__ n_active) {
  extern __shared__ int bucket[];
  const int nb = 4 * n_kf + 1;
  for (int b = threadIdx.x; b < nb; b += 1024) bucket[b] = 0;
  __syncthreads();
  // (16 landmarks per thread and pass, their tracks requested together and the keys kept for the second sweep: as `for (l ..) atomicAdd(&bucket[
  // key(kmin[l], kmax[l])], 1)` every iteration waited for its own two loads before its LDS atomic — twice ten dependent round trips at 10 k landmarks)
  constexpr int kG = 16;
  for (int sb = 0; sb < n_lm; sb += kG * 1024) {
    int key[kG];
#pragma unroll
    for (int g = 0; g < kG; ++g) { const int lc = min(sb + g * 1024 + (int)threadIdx.x, n_lm - 1); key[g] = lm_sort_key(kmin[lc], kmax[lc], n_kf); }
#pragma unroll
    for (int g = 0; g < kG; ++g) if (sb + g * 1024 + (int)threadIdx.x < n_lm) atomicAdd(&bucket[key[g]], 1);
  }
  __syncthreads();
  if (nb <= 1024) {                         // exclusive scan of the bucket counts: 16 wave scans + 16 wave totals
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = tid < nb ? bucket[tid] : 0;
    const int incl = wave_incl_scan(c);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (tid < nb) bucket[tid] = base + incl - c;
    if (tid == nb - 1) *n_active = base + incl - c;
  } else if (threadIdx.x == 0) {
    int run = 0;
    for (int b = 0; b < nb; ++b) { const int c = bucket[b]; bucket[b] = run; run += c; }
    *n_active = bucket[nb - 1];
  }
  __syncthreads();
  for (int sb = 0; sb < n_lm; sb += kG * 1024) {
    int key[kG];
#pragma unroll
    for (int g = 0; g < kG; ++g) { const int lc = min(sb + g * 1024 + (int)threadIdx.x, n_lm - 1); key[g] = lm_sort_key(kmin[lc], kmax[lc], n_kf); }
#pragma unroll
    for (int g = 0; g < kG; ++g) { const int l = sb + g * 1024 + (int)threadIdx.x; if (l < n_lm) order[atomicAdd(&bucket[key[g]], 1)] = l; }
  }
}
__global__ __launch_bounds__(1024) void k_lm_sort(int n_lm, int n_kf, const int* __restrict__ kmin, const int* __restrict__ kmax,
                                                  int* __restrict__ order, int* __restrict__ n_active) {
  lm_sort_body(n_lm, n_kf, kmin, kmax, order, n_active);
}

// ---- compact landmark layout, built once per problem_configure
// eoff[l] = first slot of landmark l, len_l = kmax_l - kmin_l slots (one per keyframe after the first); *n_slots = their total
__device__ __forceinline__ void lm_offsets_body(int n_lm, const int* __restrict__ kmin, const int* __restrict__ kmax, int* __restrict__ eoff,
                                                int* __restrict__ n_slots) {
  // 16 k landmarks per pass: every thread requests its 16 (kmin, kmax) pairs up front (a load inside the per-1024 loop was waited for
  // before the next was issued: 10 dependent round trips at 10 k landmarks), scans run inside waves, two workgroup barriers per pass
  constexpr int kG = 16;
  __shared__ int s_cnt[kG * 16];                    // [chunk][wave] totals, then their exclusive prefix
  __shared__ int s_total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int total = 0;
  for (int sb = 0; sb < n_lm; sb += kG * 1024) {
    int c[kG], ex[kG];
#pragma unroll
    for (int g = 0; g < kG; ++g) {
      const int l = sb + g * 1024 + tid, lc = min(l, n_lm - 1);
      const int lo = kmin[lc], hi = kmax[lc];
      c[g] = max(0, hi - lo) & -(int)((l < n_lm) & (hi >= 0));      // (branch-free: behind `l < n_lm && hi >= 0 ? .. : 0` the kmin load was sunk into a branch behind the kmax load's wait, 32 dependent round trips)
    }
#pragma unroll
    for (int g = 0; g < kG; ++g) {
      const int incl = wave_incl_scan(c[g]);
      ex[g] = incl - c[g];
      if (lane == 63) s_cnt[g * 16 + wave] = incl;
    }
    __syncthreads();
    if (wave == 0) {                                 // exclusive scan of the 256 totals (chunk-major = ascending landmark order)
      int carry = 0;
#pragma unroll
      for (int base = 0; base < kG * 16; base += 64) {
        const int v = s_cnt[base + lane];
        const int inc = wave_incl_scan(v);
        s_cnt[base + lane] = carry + inc - v;
        carry += __shfl(inc, 63);
      }
      if (lane == 0) s_total = carry;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < kG; ++g) {
      const int l = sb + g * 1024 + tid;
      if (l < n_lm) eoff[l] = total + s_cnt[g * 16 + wave] + ex[g];
    }
    total += s_total;
    __syncthreads();
  }
  if (tid == 0) *n_slots = total;
}
__global__ __launch_bounds__(1024) void k_lm_offsets(int n_lm, const int* __restrict__ kmin, const int* __restrict__ kmax, int* __restrict__ eoff,
                                                     int* __restrict__ n_slots) {
  lm_offsets_body(n_lm, kmin, kmax, eoff, n_slots);
}
// the counting sort and the slot offsets are two one-workgroup chains over the same (kmin, kmax) with nothing in common but their inputs:
// ONE launch of two workgroups runs them side by side (a persistent window reconfigures every tick: 14 us + 9 us + a launch gap became 14 us)
__global__ __launch_bounds__(1024) void k_lm_sort_offsets(int n_lm, int n_kf, const int* __restrict__ kmin, const int* __restrict__ kmax, int* __restrict__ order,
                                                          int* __restrict__ n_active, int* __restrict__ eoff, int* __restrict__ n_slots) {
  if (blockIdx.x == 0) lm_sort_body(n_lm, n_kf, kmin, kmax, order, n_active);
  else lm_offsets_body(n_lm, kmin, kmax, eoff, n_slots);
}
__global__ __launch_bounds__(kT) void k_tf_slots(int n, const int* __restrict__ lm, const int* __restrict__ k2, const int* __restrict__ kmin,
                                                 const int* __restrict__ eoff, int* __restrict__ slot) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const int l = lm[i];
  slot[i] = eoff[l] + (k2[i] - kmin[l] - 1);
}
// k_tf_slots and k_zero_slots in one launch: workgroups [0, g_slots) take the blocks, the rest clear the slot records
__global__ __launch_bounds__(kT) void k_tf_slots_zero(int n, int g_slots, const int* __restrict__ lm, const int* __restrict__ k2, const int* __restrict__ kmin,
                                                      const int* __restrict__ eoff, int* __restrict__ slot, const int* __restrict__ n_slots, double* __restrict__ slotB) {
  if ((int)blockIdx.x < g_slots) {
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i >= n) return;
    const int l = lm[i];
    slot[i] = eoff[l] + (k2[i] - kmin[l] - 1);
    return;
  }
  const size_t nz = (size_t)*n_slots * 4;     // double2 elements
  double2* b = reinterpret_cast<double2*>(slotB);
  const size_t gz = gridDim.x - g_slots;
  for (size_t i = (size_t)(blockIdx.x - g_slots) * kT + threadIdx.x; i < nz; i += gz * kT) b[i] = make_double2(0.0, 0.0);
}
// slots of keyframes that do not observe their landmark (gaps in a track) are never written by the linearisation: cleared once here
// sorted copies of the TwoFrame block arrays: block i of the copy = block perm[i] of the batch
__global__ __launch_bounds__(kT) void k_tf_gather(int n, const int* __restrict__ perm, const double2* __restrict__ fo, const double2* __restrict__ ob,
                                                   const int* __restrict__ lm, const int* __restrict__ k1, const int* __restrict__ k2,
                                                   double2* __restrict__ fo_s, double2* __restrict__ ob_s, int* __restrict__ lm_s, int* __restrict__ k1_s, int* __restrict__ k2_s) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const int j = perm[i];
  fo_s[i] = fo[j]; ob_s[i] = ob[j]; lm_s[i] = lm[j]; k1_s[i] = k1[j]; k2_s[i] = k2[j];
}
__global__ __launch_bounds__(kT) void k_zero_slots(const int* __restrict__ n_slots, double* __restrict__ slotB) {
  const size_t n = (size_t)*n_slots * 4;     // double2 elements
  double2* b = reinterpret_cast<double2*>(slotB);
  for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) b[i] = make_double2(0.0, 0.0);
}

__device__ __forceinline__ void schur_band_body(const int bx, const int by, int dp, int ldE, const double* __restrict__ E,
                                                const double* __restrict__ Cd, const int* __restrict__ order, const int* __restrict__ n_active_p,
                                                const int* __restrict__ kmin, const int* __restrict__ kmax, int d, int ldS,
                                                double* __restrict__ S, unsigned long long* dbg = nullptr, const int rows = kBandRows,
                                                const int4* __restrict__ item = nullptr) {
  extern __shared__ double sh[];          // Es[kSchurRows][ldl] | icd[kSchurRows] | rowid (int)[kBandRows]
  auto mark = [&](int k) { if (dbg && threadIdx.x == 0) dbg[k] = wall_clock64(); };   // LVF_SCHUR_TIMING=1: phase stamps of this workgroup
  mark(0);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lk = lane >> 4, lc = lane & 15;
  // the slice's extent and band: from the work item when there is one (k_band_work computed them once per configure: three dependent
  // global round trips — n_active, order[], kmin/kmax[] — off the front of every workgroup), else every wave reduces them for itself
  const int k_begin = bx * rows;
  int k_end, lo = 0x7fffffff, hi = -1;
  if (item) { const int4 it = *item; k_end = it.w; lo = it.z & 0xffff; hi = it.z >> 16; }
  else {
    k_end = min(*n_active_p, k_begin + rows);
    if (k_begin >= k_end) return;
    for (int r = k_begin + lane; r < k_end; r += 64) { const int l = order[r]; lo = min(lo, kmin[l]); hi = max(hi, kmax[l]); }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
  }
  const int t0 = (6 * lo) >> 4, t1 = (6 * hi + 5) >> 4, tl = dp >> 4;
  const int nbt = t1 - t0 + 1, tri = nbt * (nbt + 1) / 2;
  const bool extra = tl > t1;                                  // the g_rho column's tile lies outside the band
  const int ntiles = tri + (extra ? nbt : 0);
  const int tbase = by * kBandTilesPerGroup;
  if (tbase >= ntiles) return;
  const int ldl = 16 * (nbt + (extra ? 1 : 0));                // staged doubles per row
  double* Es = sh;
  double* icd = sh + kSchurRows * (ldE + 16);
  int* rowid = reinterpret_cast<int*>(icd + kSchurRows);
  for (int r = tid; r < rows; r += 256) rowid[r] = (k_begin + r < k_end) ? order[k_begin + r] : -1;
  // this wave's tiles: t = tbase + w + 4 s; local tile columns (a = row tile, b = column tile), extra row tile = index nbt
  int ta[kBandTilesPerWave], tb[kBandTilesPerWave], nt = 0;
#pragma unroll
  for (int s_ = 0; s_ < kBandTilesPerWave; ++s_) {
    const int t = tbase + w + 4 * s_;
    ta[s_] = 0; tb[s_] = 0;
    if (t < ntiles) {
      if (t < tri) {
        int a = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((a + 1) * (a + 2) / 2 <= t) ++a;
        while (a * (a + 1) / 2 > t) --a;
        ta[s_] = a; tb[s_] = t - a * (a + 1) / 2;
      } else { ta[s_] = nbt; tb[s_] = t - tri; }
      nt = s_ + 1;
    }
  }
  double4_t acc[kBandTilesPerWave];
#pragma unroll
  for (int s_ = 0; s_ < kBandTilesPerWave; ++s_) acc[s_] = double4_t{0.0, 0.0, 0.0, 0.0};
  // staging: thread (row fr = tid >> 4, column fc = tid & 15) carries column fc of every staged 16-column tile of its row:
  // no index arithmetic beyond one add per tile, 128-byte runs per 16 lanes
  constexpr int kPf = 20;                  // tiles prefetched in registers (ldl <= 320); wider bands finish through fetch_tail
  double pf[kPf];
  double pf_cd = 1.0;
  const int fr = tid >> 4, fc = tid & 15, nst = ldl >> 4;
  __syncthreads();                         // rowid
  auto fetch = [&](int k0) {
    const int l = rowid[k0 - k_begin + fr];
    const double* src = E + (size_t)max(l, 0) * ldE + fc;
    if (fc == 0) pf_cd = (l >= 0) ? Cd[l] : 1.0;
#pragma unroll
    for (int u = 0; u < kPf; ++u) {
      double v = 0.0;
      if (u < nst && l >= 0) v = src[16 * (u < nbt ? t0 + u : tl)];
      pf[u] = v;
    }
  };
  auto fetch_tail = [&](int k0) {
    const int l = rowid[k0 - k_begin + fr];
    for (int u = kPf; u < nst; ++u) Es[fr * ldl + 16 * u + fc] = (l >= 0) ? E[(size_t)l * ldE + 16 * (u < nbt ? t0 + u : tl) + fc] : 0.0;
  };
  fetch(k_begin);
  mark(1);
  for (int k0 = k_begin; k0 < k_end; k0 += kSchurRows) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPf; ++u) if (u < nst) Es[fr * ldl + 16 * u + fc] = pf[u];
    fetch_tail(k0);
    if (fc == 0) icd[fr] = (rowid[k0 - k_begin + fr] >= 0) ? 1.0 / pf_cd : 0.0;
    __syncthreads();
    if (k0 + kSchurRows < k_end) fetch(k0 + kSchurRows);
#pragma unroll
    for (int kk = 0; kk < kSchurRows; kk += 4) {
      const double* row = Es + (kk + lk) * ldl;
      const double wgt = icd[kk + lk];
#pragma unroll
      for (int s_ = 0; s_ < kBandTilesPerWave; ++s_)
        if (s_ < nt) acc[s_] = __builtin_amdgcn_mfma_f64_16x16x4f64(row[16 * ta[s_] + lc] * wgt, row[16 * tb[s_] + lc], acc[s_], 0, 0, 0);
    }
  }
  mark(2);
#pragma unroll
  for (int s_ = 0; s_ < kBandTilesPerWave; ++s_) {
    if (s_ >= nt) continue;
    const int gti = ta[s_] < nbt ? t0 + ta[s_] : tl, gtj = t0 + tb[s_];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = gti * 16 + lk + 4 * r, gj = gtj * 16 + lc;
      const double v = acc[s_][r];
      if (v == 0.0) continue;
      if (gi < dp && gj < dp) { if (gj <= gi) atomicAdd(&S[(size_t)gi * ldS + gj], -v); }
      else if (gi == dp && gj < dp) atomicAdd(&S[(size_t)d * ldS + gj], v);
    }
  }
  mark(3);
}

__global__ __launch_bounds__(256) void k_schur_band(int dp, int ldE, const double* __restrict__ E, const double* __restrict__ Cd,
                                                    const int* __restrict__ order, const int* __restrict__ n_active_p,
                                                    const int* __restrict__ kmin, const int* __restrict__ kmax, int d, int ldS,
                                                    double* __restrict__ S) {
  schur_band_body(blockIdx.x, blockIdx.y, dp, ldE, E, Cd, order, n_active_p, kmin, kmax, d, ldS, S);
}
struct LmBand { GP<const int> order; GP<const int> n_active; GP<const int> kmin; GP<const int> kmax; };   // null order => dense SYRK
static int launch_schur(hipStream_t q, int n_lm, int dp, int ldE, const double* E, const double* Cd, int d, int ldS, double* S, const LmBand& band) {
  const int nt = ldE / 16, ntile = nt * (nt + 1) / 2;
  if (band.order) {
    const size_t shb = ((size_t)kSchurRows * (ldE + 16) + kSchurRows) * sizeof(double) + kBandRowsMax * sizeof(int);
    if (shb <= 64 * 1024) {
      hipLaunchKernelGGL(k_schur_band, dim3((n_lm + kBandRows - 1) / kBandRows, (ntile + kBandTilesPerGroup - 1) / kBandTilesPerGroup), dim3(256), shb, q,
                         dp, ldE, E, Cd, band.order, band.n_active, band.kmin, band.kmax, d, ldS, S);
      LVF_HIP(hipGetLastError());
      return LVF_OK;
    }
  }
  const bool lds_path = ldE <= 320 && ntile <= kSchurGroups * 4 * kSchurTilesPerWave;
  if (lds_path) {
    const int slices = std::max(1, std::min(32, (n_lm + 4 * kSchurRows - 1) / (4 * kSchurRows)));   // 16..128 measured: 32 is the optimum at 10 k rows
    int rows_per_slice = (n_lm + slices - 1) / slices;
    rows_per_slice = ((rows_per_slice + kSchurRows - 1) / kSchurRows) * kSchurRows;
    const size_t shb = ((size_t)kSchurRows * ldE + kSchurRows) * sizeof(double);
    hipLaunchKernelGGL(k_schur_lds, dim3(kSchurGroups, (n_lm + rows_per_slice - 1) / rows_per_slice), dim3(256), shb, q, n_lm, dp, ldE, ntile, rows_per_slice, E,
                       Cd, d, ldS, S);
  } else {
    hipLaunchKernelGGL(k_schur_syrk, dim3(ntile, (n_lm + kSchurChunk - 1) / kSchurChunk), dim3(64), 0, q, n_lm, dp, ldE, ntile, E, Cd, d, ldS, S);
  }
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

// ------------------------------------------------------------------------------------------------ blocked Cholesky (64)
// Right-looking, block 64, ONE launch per block step (k_chol_step): every workgroup of the step's column re-factors the 64x64 diagonal
// block (redundantly: all of them run it concurrently, which removes a dependent launch) while its own panel block rides along in the
// same sweep, and the trailing update of the previous step is folded into the same launch (chol_step_body).
// The sequential critical path is ~64 x (rsqrt + broadcast) per block; everything else is wide.
constexpr int kNB = 64, kLd = 65;

__device__ __forceinline__ void load_row64(const double* __restrict__ g, double a[kNB]) {
  const double2* g2 = reinterpret_cast<const double2*>(g);
#pragma unroll
  for (int c = 0; c < kNB / 2; ++c) { const double2 v = g2[c]; a[2 * c] = v.x; a[2 * c + 1] = v.y; }
}
__device__ __forceinline__ void store_row64(double* __restrict__ g, const double a[kNB]) {
  double2* g2 = reinterpret_cast<double2*>(g);
#pragma unroll
  for (int c = 0; c < kNB / 2; ++c) g2[c] = make_double2(a[2 * c], a[2 * c + 1]);
}


// Factor + panel solve of one 64-wide block column by ONE workgroup of 512 threads = 8 waves, thread = row r = tid & 63:
//   waves 0..3 ("diagonal" waves) hold the diagonal block, wave Q the columns 16Q..16Q+15 of every row;
//   waves 4..7 ("panel" waves) hold the workgroup's panel block the same way; X L^T = A is solved in the same sweep: once column j of L
//   is known, x_rj = b_rj / L_jj is final and the row's later columns take the rank-1 term x_rj L_tj.
// One wave issues at most one VALU instruction per ~8 clocks (tools/ubench), so what a wave costs is its instruction count; the 64
// pivots are a dependent chain through the diagonal waves, and everything that is not the chain is kept out of their instruction stream.
// The wave-uniform operands L_tj of the rank-1 updates are fetched with ONE 8-byte LDS read per pivot (lane t of each 16-lane row takes
// L_tj) and handed out by the DPP row_newbcast operand of v_fmac_f64: as 16-byte broadcast reads they occupied the LDS pipe for 8 clocks
// each, 8 per pivot and wave, and that pipe — shared by all waves — set the pace (measured 375 clocks per pivot with 4 waves, 512 with
// 8); through v_readlane + SGPR operands it was slower still (900).
//   * pivots are taken kPG at a time: the owning wave finishes a group of kPG columns on its own (the later columns of the group take
//     the earlier ones' rank-1 terms from registers and lane broadcasts) and publishes them (Lcol[j][r], 1/L_jj); the other diagonal
//     waves apply a group ONE STEP after it was published, so the owner never waits for them: one workgroup barrier per step, at which
//     the consumers — who have less to do per step — are already waiting when the owner arrives;
//   * the panel arithmetic (three quarters of the flops) lives in its own four waves, which run one more step behind on the published
//     columns: they share the barriers but never hold the chain up;
//   * 1/sqrt is the hardware estimate plus one second-order correction (the library call adds range checks and two dependent selects
//     per pivot); a non-positive pivot is flagged and its NaN/inf only lives until the step is rejected.  Lane j holds A_jj itself, so
//     scaling its entry gives L_jj with no select.
// acc += (lane N of each 16-lane row of lv) * m: the DPP row_newbcast operand of v_fmac_f64 hands a row-uniform value to all 16 lanes
// inside the multiply-add itself (no LDS broadcast read, no v_readlane + SGPR operand)
template <int N>
__device__ __forceinline__ void fmac_row_bcast(double& acc, double lv, double m) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(lv), "v"(m), "n"(N));
}
// (`from` folds to a constant once the caller's loops are unrolled)
__device__ __forceinline__ void rank1_row_bcast(const int from, double a[16], double lv, double m) {
  asm volatile("s_nop 4");                           // (an EXEC change needs 5 wait states before a DPP read; inline asm is not hazard-checked)
  if (from <= 0) fmac_row_bcast<0>(a[0], lv, m);
  if (from <= 1) fmac_row_bcast<1>(a[1], lv, m);
  if (from <= 2) fmac_row_bcast<2>(a[2], lv, m);
  if (from <= 3) fmac_row_bcast<3>(a[3], lv, m);
  if (from <= 4) fmac_row_bcast<4>(a[4], lv, m);
  if (from <= 5) fmac_row_bcast<5>(a[5], lv, m);
  if (from <= 6) fmac_row_bcast<6>(a[6], lv, m);
  if (from <= 7) fmac_row_bcast<7>(a[7], lv, m);
  if (from <= 8) fmac_row_bcast<8>(a[8], lv, m);
  if (from <= 9) fmac_row_bcast<9>(a[9], lv, m);
  if (from <= 10) fmac_row_bcast<10>(a[10], lv, m);
  if (from <= 11) fmac_row_bcast<11>(a[11], lv, m);
  if (from <= 12) fmac_row_bcast<12>(a[12], lv, m);
  if (from <= 13) fmac_row_bcast<13>(a[13], lv, m);
  if (from <= 14) fmac_row_bcast<14>(a[14], lv, m);
  if (from <= 15) fmac_row_bcast<15>(a[15], lv, m);
}
constexpr int kPG = 2, kCT = 512, kGW = 16 / kPG;                    // pivots per group, threads, groups per wave
struct FactorLds { double* Lcol; double* Xcol; double* Linv; };       // Lcol/Xcol: [64 columns][64 rows]

// diagonal wave Q, steps s = kGW qj + i (i unrolled, qj a real loop: the code of one column group is reused four times).  In step s the
// owner of group s runs its pivots while every other diagonal wave applies group s - 1 (published at the end of step s - 1); the
// workgroup meets at one barrier per step.
// `nsteps` (<= kNB / kPG + 1): the last block of the corner is padded with identity rows / columns — their pivots are 1 and nothing
// real depends on them — so the sweep stops one step after the last real pivot group (every wave of the workgroup gets the same count:
// the barriers stay matched)
__device__ __forceinline__ bool factor_diag_wave(double a[16], const int r, const int Q, const FactorLds& F, const int nsteps = kNB / kPG + 1) {
  bool bad = false;
#pragma unroll 1
  for (int qj = 0; qj <= kNB / 16; ++qj) {
#pragma unroll
    for (int i = 0; i < kGW; ++i) {
      if ((qj == kNB / 16 && i > 0) || kGW * qj + i >= nsteps) break;
      const int j0 = 16 * qj + kPG * i;                   // first pivot of group s
      const int prev_owner = i > 0 ? qj : qj - 1;         // owner of group s - 1
      if (j0 > 0 && prev_owner < Q) {                     // columns right of group s - 1: all 16
#pragma unroll
        for (int g = 0; g < kPG; ++g) {
          const int j = j0 - kPG + g;
          const double cr = F.Lcol[j * kNB + r], lv = F.Lcol[j * kNB + 16 * Q + (r & 15)];
          rank1_row_bcast(0, a, lv, -cr);                 // A_rt -= L_rj L_tj (meaningful for r >= t)
        }
      }
      if (qj == Q) {                                      // own group: pivots j0 .. j0 + kPG - 1
#pragma unroll
        for (int g = 0; g < kPG; ++g) {
          const int jj = kPG * i + g, j = j0 + g;
          const double djj = lane_bcast(a[jj], j);        // pivot A_jj sits in lane j (= row j) of this wave
          bad |= !(djj > 0.0);
          const double y0 = __builtin_amdgcn_rsq(djj);
          const double e = fma(-djj * y0, y0, 1.0);
          const double inv_l = fma(y0 * e, fma(e, 0.375, 0.5), y0);
          a[jj] *= inv_l;
          F.Lcol[j * kNB + r] = a[jj];
          F.Linv[j] = inv_l;                              // (every lane stores the same value: no exec juggling on the chain)
#pragma unroll
          for (int h = g + 1; h < kPG; ++h) a[kPG * i + h] -= a[jj] * lane_bcast(a[jj], j0 + h);
        }
        // the rest of this wave's own columns: L_tj of its own rows t comes back from the column it has just published
#pragma unroll
        for (int g = 0; g < kPG; ++g)
          if (kPG * (i + 1) < 16) rank1_row_bcast(kPG * (i + 1), a, F.Lcol[(j0 + g) * kNB + 16 * Q + (r & 15)], -a[kPG * i + g]);
      }
      __syncthreads();
    }
  }
  return bad;
}

// panel wave Q: owns the x columns 16Q..16Q+15 of its rows and runs one step behind the diagonal waves: in step s it applies the x
// group s - 2 its left neighbours published in step s - 1, then — if it owns group s - 1 — finishes those x columns and publishes them.
__device__ __forceinline__ void factor_panel_wave(double b[16], const int r, const int Q, const FactorLds& F, const int nsteps = kNB / kPG + 1) {
#pragma unroll 1
  for (int qj = 0; qj <= kNB / 16; ++qj) {
#pragma unroll
    for (int i = 0; i < kGW; ++i) {
      if ((qj == kNB / 16 && i > 0) || kGW * qj + i >= nsteps) break;
      const int j0 = 16 * qj + kPG * i;
      const int o2 = i > 1 ? qj : qj - 1;                 // owner of group s - 2
      if (j0 >= 2 * kPG && o2 < Q) {
#pragma unroll
        for (int g = 0; g < kPG; ++g) {
          const int j = j0 - 2 * kPG + g;
          const double cx = F.Xcol[j * kNB + r], lv = F.Lcol[j * kNB + 16 * Q + (r & 15)];
          rank1_row_bcast(0, b, lv, -cx);
        }
      }
      const int o1 = i > 0 ? qj : qj - 1, i1 = (i + kGW - 1) % kGW;     // owner of group s - 1 and its index within the wave
      if (j0 > 0 && o1 == Q) {
#pragma unroll
        for (int g = 0; g < kPG; ++g) {
          const int jj = kPG * i1 + g, j = 16 * o1 + jj;
          const double lv = F.Lcol[j * kNB + 16 * Q + (r & 15)];
          b[jj] *= F.Linv[j];
          F.Xcol[j * kNB + r] = b[jj];
          if (jj + 1 < 16) rank1_row_bcast(jj + 1, b, lv, -b[jj]);
        }
      }
      __syncthreads();
    }
  }
}

// ONE launch per block step kb (grid = chol_step_grid), 512 threads per workgroup:
//   workgroups [0, 2 + below)  — the column of step kb.  For kb > 0 each first applies step kb-1's trailing update to the two tiles it
//       reads, A_kk -= P_k P_k^T (diagonal waves) and A_ik -= P_i P_k^T (panel waves) (P = the panel of column kb-1, staged in LDS;
//       v_mfma_f64_16x16x4_f64), instead of waiting for a separate update launch.  Then: workgroup 0 stores the factored diagonal
//       block; workgroups 1..below solve their panel block X L^T = A; the last workgroup solves against the identity and leaves
//       L_kk^-T for the back substitution.
//   workgroups behind them     — the rest of step kb-1's trailing update, A[bi][bj] -= P_bi P_bj^T for kb < bj <= bi, which nothing in
//       this launch reads (the next step does).
struct CholArgs { GP<double> Sd; int ld, nb; GP<int> fail; GP<double> Dinv; GP<const int> done; GP<unsigned long long> dbg; int last_cols; GP<double> Ldiag; };   // dbg: LVF_CHOL_TIMING stamps; last_cols: real (un-padded) columns of the last block
__host__ __device__ inline int chol_step_grid(int nb, int kb) {
  const int below = nb - kb - 1;
  return kb >= nb ? 0 : 2 + below + (kb > 0 ? below * (below + 1) / 2 : 0);
}

// staging of a 64x64 block at g (leading dimension ld) into LDS with row stride kLd by 256 threads (t = 0..255): all eight 16-byte
// loads of a thread are in flight before the first LDS write (written as one loop the compiler waits for each load in turn)
struct Stage64 { double2 v[8]; };
__device__ __forceinline__ void stage_issue(const double* g, int ld, int t, Stage64& st) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = t + 256 * i, rr = e >> 5, c = (e & 31) * 2;
    st.v[i] = *reinterpret_cast<const double2*>(g + (size_t)rr * ld + c);
  }
}
__device__ __forceinline__ void stage_commit(const Stage64& st, int t, double* __restrict__ L) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = t + 256 * i, rr = e >> 5, c = (e & 31) * 2;
    L[rr * kLd + c] = st.v[i].x; L[rr * kLd + c + 1] = st.v[i].y;
  }
}

// trailing update of one tile: A[bi][bj] -= P_bi P_bj^T with the panels of column kp.  Wave wv produces rows 16 (wv & 3) .. + 15 of the
// column half wv >> 2 (A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], D: col = lane&15, row = (lane>>4) + 4 reg).
__device__ __forceinline__ void chol_update_tile(double* S, int ld, int kp, int bi, int bj, double* Pi, double* Pj) {
  // the tile being updated is requested FIRST (it does not depend on the product) so its round trip hides under the panel
  // staging and the matrix-core work; S is not __restrict__ so the loads stay where they are written
  const int wv = threadIdx.x >> 6, w = wv & 3, ch = wv >> 2, lane = threadIdx.x & 63, lk = lane >> 4, lc = lane & 15;
  double* out = S + (size_t)(bi * kNB + 16 * w) * ld + bj * kNB + 32 * ch;
  double o[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) o[ct][rg] = out[(size_t)(lk + 4 * rg) * ld + 16 * ct + lc];
  {
    Stage64 st;
    const int t = threadIdx.x & 255;
    stage_issue(S + (size_t)((ch ? bj : bi) * kNB) * ld + kp * kNB, ld, t, st);
    stage_commit(st, t, ch ? Pj : Pi);
  }
  __syncthreads();
  double4_t acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int k0 = 0; k0 < kNB; k0 += 4) {
    const double av = Pi[(16 * w + lc) * kLd + k0 + lk];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Pj[(32 * ch + 16 * ct + lc) * kLd + k0 + lk], acc[ct], 0, 0, 0);
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) out[(size_t)(lk + 4 * rg) * ld + 16 * ct + lc] = o[ct][rg] - acc[ct][rg];
}

__device__ __forceinline__ void chol_step_body(const int bx, const CholArgs& A, const int kb) {
  const int below = A.nb - kb - 1;
  if (bx >= chol_step_grid(A.nb, kb)) return;                          // (workgroup-uniform: no barrier is skipped by part of a workgroup)
  const int dv = done_flag_issue(A.done);
  double* S = A.Sd; const int ld = A.ld; int* __restrict__ fail = A.fail; double* __restrict__ Dinv = A.Dinv;
  __shared__ double Pi[kNB * kLd];
  __shared__ double Pj[kNB * kLd];
  __shared__ double Linv[kNB];
  if (bx >= 2 + below) {                              // trailing tiles of step kb-1 right of column kb
    int t = bx - (2 + below), ii = 0;
    while (t >= ii + 1) { t -= ii + 1; ++ii; }
    if (dv) return;
    chol_update_tile(S, ld, kb - 1, kb + 1 + ii, kb + 1 + t, Pi, Pj);
    return;
  }
  const int tid = threadIdx.x, r = tid & 63, wv = tid >> 6;
  const bool panel_wave = wv >= 4;                    // wave-uniform
  // column group of the wave.  Waves land on SIMD wv % 4: the panel wave of group q sits two SIMDs away from the diagonal wave of group q,
  // so the two waves that are busiest at the same time (the pivot owner and the panel owner one step behind it) do not share an issue port
  const int q = panel_wave ? ((wv + 2) & 3) : wv;
  unsigned long long* dbg = (A.dbg && bx == 1 && tid == 0) ? A.dbg + 8 * kb : nullptr;
  if (dbg) dbg[0] = wall_clock64();
  const bool inverse_wg = bx == 1 + below, panel_wg = bx > 0 && !inverse_wg;
  double* brow = inverse_wg ? Dinv + (size_t)kb * kNB * kNB + (size_t)r * kNB + 16 * q
                            : S + (size_t)((kb + bx) * kNB + r) * ld + kb * kNB + 16 * q;
  double* drow = S + (size_t)(kb * kNB + r) * ld + kb * kNB + 16 * q;
  double a[16];                                       // diagonal waves: row r of A_kk; panel waves: row r of the panel block / identity
  if (!panel_wave || panel_wg) {
    const double2* g2 = reinterpret_cast<const double2*>(panel_wave ? brow : drow);
#pragma unroll
    for (int c = 0; c < 8; ++c) { const double2 v = g2[c]; a[2 * c] = v.x; a[2 * c + 1] = v.y; }
  } else {
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = (inverse_wg && 16 * q + c == r) ? 1.0 : 0.0;
  }
  if (dv) return;                                     // (the tile loads above are in flight behind the flag's)
  if (kb > 0) {
    // step kb-1's update of the tiles just requested: diagonal wave q forms rows 16q..16q+15 of P_k P_k^T (lower tiles only), panel wave q
    // those of P_i P_k^T; the products go through LDS into the row-per-lane layout of the factorisation
    const int lane = tid & 63, lk = lane >> 4, lc = lane & 15;
    {
      Stage64 st;
      const int t = tid & 255;
      const bool mine = !panel_wave || panel_wg;
      if (mine) stage_issue(S + (size_t)((panel_wave ? kb + bx : kb) * kNB) * ld + (kb - 1) * kNB, ld, t, st);
      if (mine) stage_commit(st, t, panel_wave ? Pi : Pj);
    }
    __syncthreads();
    if (dbg) dbg[1] = wall_clock64();
    // 10 lower tiles of P_k P_k^T + (panel workgroups) 16 tiles of P_i P_k^T, dealt round-robin to the 8 waves (two waves share a SIMD's
    // matrix pipe: 6.5 tile products per SIMD instead of up to 8 with one strip per wave)
    const int n_tiles = panel_wg ? 26 : 10;
    double4_t ac[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    int t_m[4], t_c[4]; bool t_own[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int t = wv + 8 * n;
      t_own[n] = t >= 10;
      if (t_own[n]) { t_m[n] = (t - 10) >> 2; t_c[n] = (t - 10) & 3; }
      else { int m = 0, u = t; while (u > m) { u -= m + 1; ++m; } t_m[n] = m; t_c[n] = u; }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
      if (wv + 8 * n < n_tiles) {
        const double* Pa = (t_own[n] ? Pi : Pj) + (16 * t_m[n] + lc) * kLd + lk;
        const double* Pb = Pj + (16 * t_c[n] + lc) * kLd + lk;
#pragma unroll
        for (int k0 = 0; k0 < kNB; k0 += 4) ac[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pa[k0], Pb[k0], ac[n], 0, 0, 0);
      }
    __syncthreads();                                  // every wave is done reading the panels: the products take their place
    if (dbg) dbg[2] = wall_clock64();
#pragma unroll
    for (int n = 0; n < 4; ++n)
      if (wv + 8 * n < n_tiles) {
        double* Pw = t_own[n] ? Pi : Pj;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Pw[(16 * t_m[n] + lk + 4 * rg) * kLd + 16 * t_c[n] + lc] = ac[n][rg];
      }
    __syncthreads();
    const double* Pw = panel_wave ? Pi : Pj;
    if (!panel_wave || panel_wg)
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] -= Pw[r * kLd + 16 * q + c];     // (entries right of the diagonal are never read as values of A)
    __syncthreads();                                  // the factorisation reuses both buffers
  }
  if (dbg) { dbg[3] = wall_clock64(); dbg[6] = clock64(); }
  const FactorLds F{Pi, Pj, Linv};
  bool bad = false;
  // pivot groups to run: all of them, except in the last block where only the real columns (the rest is identity padding) need any
  const int ncols = (kb == A.nb - 1 && A.last_cols > 0) ? min(A.last_cols, kNB) : kNB;
  const int nsteps = (ncols + kPG - 1) / kPG + 1;
  if (panel_wave) factor_panel_wave(a, r, q, F, nsteps);
  else bad = factor_diag_wave(a, r, q, F, nsteps);
  if (dbg) { dbg[4] = wall_clock64(); dbg[7] = clock64(); }
  if (bad && r == 0) atomicMax(fail, 1 + kb);
  if (bx == 0 && !panel_wave) {
    // The factored diagonal block goes to a SIDE buffer, never back into S: every workgroup of this column loads A_kk from S when it starts,
    // and a workgroup that is dispatched late — a second context's kernels filling the chip (Backend::Optimize beside Relocator,
    // relocator.cpp:188) — would find L_kk there instead of A_kk and factor garbage (measured: 197 of 400 solves took a step as invalid
    // under chip-filling traffic; the per-launch "everyone has loaded long before workgroup 0 stores" held only on an otherwise idle GPU).
    // What reads L_kk afterwards — the right-hand-side row inside the last block, by the back substitution — reads it from there.
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) if (16 * q + tt > r) a[tt] = 0.0;
    double2* g2 = reinterpret_cast<double2*>(A.Ldiag + (size_t)kb * kNB * kNB + (size_t)r * kNB + 16 * q);
#pragma unroll
    for (int c = 0; c < 8; ++c) g2[c] = make_double2(a[2 * c], a[2 * c + 1]);
  }
  if (bx > 0 && panel_wave) {
    double2* g2 = reinterpret_cast<double2*>(brow);
#pragma unroll
    for (int c = 0; c < 8; ++c) g2[c] = make_double2(a[2 * c], a[2 * c + 1]);
  }
  if (dbg) dbg[5] = wall_clock64();
}
__global__ __launch_bounds__(kCT) void k_chol_step(CholArgs a, int kb) { chol_step_body(blockIdx.x, a, kb); }
__global__ __launch_bounds__(kCT) void k_chol_step_b(const CholArgs* __restrict__ t, int kb) { chol_step_body(blockIdx.x, t[blockIdx.y], kb); }
__global__ __launch_bounds__(kCT) void k_chol_step_bt(const CholArgs* __restrict__ t, int kb) { chol_step_body(blockIdx.y, t[blockIdx.x], kb); }

// ------------------------------------------------------------------------------------------------ elimination order
// The (v, ba, bg) blocks only meet each other and the poses through ImuError factors, i.e. along the IMU chain: block k touches
// blocks k-1, k+1 and poses k-1, k, k+1.  Factorising them FIRST, in nested-dissection order (level 0 = every other block of the
// chain, level 1 = every other one of what is left, ...; any coupling graph works, the levels are greedy independent sets with
// fill-in tracked on the host), costs 9 sequential pivots per LEVEL instead of 9 per block, and leaves only the pose corner
// (6 n_kf) for the dense blocked factorisation: at 50 keyframes 6 x 9 + 300 sequential pivots instead of 750.
// k_sp_eliminate, one launch per level, `tiles` workgroups per block b with neighbour rows N (|N| = m):
//   L_bb = chol(S_bb) (wave 0, lane = row, pivots broadcast by v_readlane);  W = S_Nb L_bb^-T (thread per row);
//   S_NN -= W W^T (pairs split over the tiles; atomics, because blocks of one level share neighbours).
// W and L_bb go to side buffers (the eliminated columns of S are never read again), so the tiles of a block never race.
__device__ __forceinline__ void sp_eliminate_body(const int vb, const SpNode* __restrict__ nodes, int first, int tiles, const int* __restrict__ rows,
                                                  double* __restrict__ S, int ld, double* __restrict__ W, int wstride,
                                                  double* __restrict__ Lout, int* __restrict__ fail, const int* done = nullptr,
                                                  const SpSrc src = SpSrc{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 1, 200000u, 0, 0, nullptr, 0}) {
  extern __shared__ double sp_sm[];        // Ws[m][9] | L[81] | linv[9] | rws[m] (int) | rnat[m] (int, early form)
  const int dv = done_flag_issue(done);
  const int ni = first + vb / tiles, tile = vb % tiles, tid = threadIdx.x;
  unsigned long long* stamp = (src.dbg && tile == 0 && tid == 0) ? src.dbg + (size_t)ni * 8 : nullptr;
  if (stamp) stamp[0] = wall_clock64();
  const SpNode nd = nodes[ni];
  const double radius = src.B ? *src.radius : 1.0;
  if (dv) return;
  const bool chained = src.wait_counter != nullptr;
  // an entry of S the level below may have added into during THIS launch: an agent-scope atomic load behind the acquire fence that
  // follows the wait (the adds it must see were RETURNING atomics performed before the producer's release arrival; the wrong results
  // round 3 chased — 5 of 12 runs of the 8-keyframe / 20 000-landmark case — came from RETURNLESS adds on the producer side, not from
  // this load).  rmw_read (LVF_CHAIN_RMW_READ=1, diagnostic): a returning atomic adding zero instead — same results, 7 us slower.
  auto ld_s = [&](double* ptr) -> double {
    if (!chained) return *ptr;
    return src.rmw_read ? __hip_atomic_fetch_add(ptr, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  const int m = nd.m, col = nd.col;
  double* Ws = sp_sm;
  double* L = sp_sm + 9 * m;
  double* linv = L + 81;
  int* rws = reinterpret_cast<int*>(linv + 9);
  int* rnat = rws + m;
  const int nb0 = src.dp + 9 * nd.id;      // early form: the block's first unknown in the natural order (poses | 9 per keyframe)
  // The level is a chain of dependent round trips (node -> row list -> rows -> factor -> W -> updates), so everything is requested as early
  // as its address is known, and the rows are dealt to waves 1..3 first: their requests are in flight while wave 0 factors the block
  // (wave 0 only takes rows when there are more than 192).  Each thread keeps its FIRST row in registers; further rows (m > 256) follow
  // the classic loop behind the barrier.
  const int r0 = (tid + 192) & 255;
  // ---- phase A: what does not depend on the level below
  const bool has0 = r0 < m;
  int rw0 = 0, rn0 = -1;
  if (has0) { rw0 = rows[nd.row_off + r0]; if (src.B) rn0 = src.rows_nat[nd.row_off + r0]; }
  for (int r = tid; r < m; r += 256) { rws[r] = rows[nd.row_off + r]; if (src.B) rnat[r] = src.rows_nat[nd.row_off + r]; }
  double bd[9], sv0[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) { bd[c] = 0.0; sv0[c] = 0.0; }
  if (src.B) {
    if (tid < 9) {                         // what k_prepare would have stored in the diagonal block: B + clamp(diag B) / radius
      const double inv_radius = 1.0 / radius;
      const int jf = *src.jac.frozen;
      const double h0d = src.jac.h0[nb0 + tid];
      // (the row's nine entries are requested together, column clamped to the diagonal: as `c <= tid ? B[..] : 0` each load sat behind its own
      // branch and was waited for there — nine dependent round trips at the head of every sparse level)
      double braw[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) braw[c] = src.B[(size_t)(nb0 + tid) * src.ldB + nb0 + min(c, tid)];
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        double b = c <= tid ? braw[c] : 0.0;
        if (c == tid) b += lm_damping_own(b, src.jac, nb0 + tid, jf, h0d) * inv_radius;      // (every tile workgroup of the node stores the same H0)
        bd[c] = b;
      }
    }
    if (has0) {                            // the row's entries of B (lower triangle, natural order) / of -gc (the right-hand-side row)
      if (rn0 == -2) {
#pragma unroll
        for (int c = 0; c < 9; ++c) sv0[c] = -src.gc[nb0 + c];
      } else if (rn0 >= 0) {
#pragma unroll
        for (int c = 0; c < 9; ++c) { const int oj = nb0 + c; sv0[c] = src.B[(size_t)max(rn0, oj) * src.ldB + min(rn0, oj)]; }
      }
    }
  }
  if (stamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp[1] = wall_clock64(); }      // (timing only: phase A's requests have landed)
  if (chained) {
    // bounded: if the level below does not arrive in time (workgroups of another stream or process took the CUs its producers needed, or
    // a dispatch order this code does not expect) the hand-over flag is raised instead of hanging; the step is then NOT judged: see SpSrc
    if (tid == 0) {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_fetch_add((int*)src.wait_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < src.wait_target) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > (unsigned long long)src.timeout_ticks) { atomicMax(fail, kFailHandover + nd.id); break; }
      }
    }
    asm volatile("s_barrier" ::: "memory");           // (not __syncthreads(): the requests above stay in flight across it)
    // every wave orders its reads of S behind the producers' release arrivals (agent scope: the L1 copy of a line an earlier level's
    // plain loads brought in is dropped; phase A's requests have long landed while the wave sat at the barrier)
    if (src.fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  if (stamp) stamp[2] = wall_clock64();
  // ---- phase B: what the level below added into S
  double a[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) a[c] = 0.0;
  if (tid < 9) {
    double sraw[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) sraw[c] = 0.0;
    if (!src.s_zero) {                      // (all nine requested together, column clamped to the diagonal: see phase A)
#pragma unroll
      for (int c = 0; c < 9; ++c) sraw[c] = ld_s(&S[(size_t)(col + tid) * ld + col + min(c, tid)]);
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) a[c] = (c <= tid ? sraw[c] : 0.0) + bd[c];
  }
  if (has0 && !src.s_zero) {
    double* srow = S + (size_t)rw0 * ld + col;
#pragma unroll
    for (int c = 0; c < 9; ++c) sv0[c] += ld_s(srow + c);
  }
  if (stamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp[3] = wall_clock64(); }      // (timing only: the S entries are here)
  if (tid < 64) {
    const int lane = tid;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const double djj = lane_bcast(a[j], j);
      bad |= !(djj > 0.0);
      // hardware estimate + one second-order correction (as in the dense corner); lane j holds A_jj itself, so no select
      const double y0 = __builtin_amdgcn_rsq(djj);
      const double e = fma(-djj * y0, y0, 1.0);
      const double inv_l = fma(y0 * e, fma(e, 0.375, 0.5), y0);
      a[j] *= inv_l;
      if (lane == j) linv[j] = inv_l;
      // A_rt -= L_rj L_tj (meaningful for t <= r): L_tj sits in lane t of the same 16-lane row — the DPP row broadcast hands it to the
      // multiply-add directly (one instruction per column instead of two v_readlane + one FMA)
      {
        const double m = -a[j];
        asm volatile("s_nop 4");
        if (j < 1) fmac_row_bcast<1>(a[1], a[j], m);
        if (j < 2) fmac_row_bcast<2>(a[2], a[j], m);
        if (j < 3) fmac_row_bcast<3>(a[3], a[j], m);
        if (j < 4) fmac_row_bcast<4>(a[4], a[j], m);
        if (j < 5) fmac_row_bcast<5>(a[5], a[j], m);
        if (j < 6) fmac_row_bcast<6>(a[6], a[j], m);
        if (j < 7) fmac_row_bcast<7>(a[7], a[j], m);
        if (j < 8) fmac_row_bcast<8>(a[8], a[j], m);
      }
    }
    if (lane < 9) {
#pragma unroll
      for (int c = 0; c < 9; ++c) L[lane * 9 + c] = (c <= lane) ? a[c] : 0.0;
    }
    if (bad && lane == 0) atomicMax(fail, kFailSparse + nd.id);
  }
  if (stamp) stamp[4] = wall_clock64();
  __syncthreads();
  auto finish_row = [&](const int r, const double sv[9]) {     // W_r = S_rb L_bb^-T
    double w[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      double v = sv[c];
#pragma unroll
      for (int k = 0; k < c; ++k) v -= w[k] * L[c * 9 + k];
      w[c] = v * linv[c];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) Ws[r * 9 + c] = w[c];
    if (tile == 0) {
#pragma unroll
      for (int c = 0; c < 9; ++c) W[(size_t)c * wstride + nd.row_off + r] = w[c];     // component-major: coalesced here and in the back substitution
    }
  };
  if (has0) finish_row(r0, sv0);
  for (int r = r0 + 256; r < m; r += 256) {
    double* srow = S + (size_t)rws[r] * ld + col;
    double sv[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) sv[c] = src.s_zero ? 0.0 : ld_s(srow + c);
    if (src.B) {
      const int oi = rnat[r];
      if (oi == -2) {
#pragma unroll
        for (int c = 0; c < 9; ++c) sv[c] -= src.gc[nb0 + c];
      } else if (oi >= 0) {
#pragma unroll
        for (int c = 0; c < 9; ++c) { const int oj = nb0 + c; sv[c] += src.B[(size_t)max(oi, oj) * src.ldB + min(oi, oj)]; }
      }
    }
    finish_row(r, sv);
  }
  if (tile == 0 && tid < 9) {              // column tid of L_bb^-1 (forward substitution against e_tid), for the back substitution
    double xcol[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      double v = (r == tid) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < r; ++k) v -= L[r * 9 + k] * xcol[k];
      xcol[r] = v * linv[r];
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) Lout[(size_t)ni * 81 + r * 9 + tid] = xcol[r];
  }
  __syncthreads();
  // S_NN -= W W^T.  The pairs whose COLUMN belongs to a sparse block (rows [0, ns): later (v, ba, bg) blocks come first in the ascending
  // row list) are what the next level reads; they go first, and a chained level signals its successor as soon as THEY are acknowledged —
  // the bulk (the dense corner's entries, four fifths at the top level) and its acknowledgement stay off the chain.
  int ns = 0;
  if (src.done_counter && src.strip_end > 0) {                  // (binary search: rws is ascending)
    int lo = 0, hi = m;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (rws[mid] < src.strip_end) lo = mid + 1; else hi = mid; }
    ns = lo;
  }
  auto pair_value = [&](const int r, const int c2) {
    const double* wr = Ws + r * 9;
    const double* wc = Ws + c2 * 9;
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) v += wr[c] * wc[c];
    return v;
  };
  auto pair_update = [&](const int r, const int c2) {
    const double v = pair_value(r, c2);
    if (v != 0.0) atomicAdd(&S[(size_t)rws[r] * ld + rws[c2]], -v);
  };
  if (ns > 0) {
    // RETURNING atomics: the value only comes back once the add has been performed where the next level will read it, so the barrier
    // below (which waits for the returns) really orders them ahead of the arrival.  With returnless adds the acknowledgement that
    // releases vmcnt came first often enough: 5 of 12 runs of the 8-keyframe / 20 000-landmark case had the next level read entries short of an update.
    double sink = 0.0;
    for (int q = tile * 256 + tid; q < ns * m; q += tiles * 256) {      // the m x ns rectangle, its r < c2 corner skipped
      const int r = q / ns, c2 = q - r * ns;
      if (r >= c2) {
        const double v = pair_value(r, c2);
        if (v != 0.0) sink += __hip_atomic_fetch_add(&S[(size_t)rws[r] * ld + rws[c2]], -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (sink == -1.2345678901234567e301) atomicMax(fail, 50000);     // (never: keeps the returns alive)
  }
  if (stamp) stamp[5] = wall_clock64();
  if (src.done_counter) {
    __syncthreads();                       // every wave has its returns (the barrier drains vmcnt)
    if (tid == 0) {
      if (src.fenced) __hip_atomic_fetch_add((int*)src.done_counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add((int*)src.done_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (stamp) stamp[6] = wall_clock64();
  const int m2 = m - ns, P = m2 * (m2 + 1) / 2;                          // the triangle over rows / columns [ns, m)
  for (int p = tile * 256 + tid; p < P; p += tiles * 256) {
    int r = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= p) ++r;
    while (r * (r + 1) / 2 > p) --r;
    const int c2 = p - r * (r + 1) / 2;
    pair_update(r + ns, c2 + ns);
  }
  if (stamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp[7] = wall_clock64(); }
}
// Work list of the band Schur complement: one item per (slice, tile group) that has tiles to form.  The 2-D grid slices x groups is sized
// for the widest possible band, but most slices (short tracks) need one group: three quarters of its workgroups had nothing to do and
// their dispatch — each needs its LDS slice — was most of the launch's span.  Built once per problem_configure (the bands are fixed).
__global__ __launch_bounds__(64) void k_band_work(int rows, int dp, const int* __restrict__ n_active_p, const int* __restrict__ order,
                                                  const int* __restrict__ kmin, const int* __restrict__ kmax, int4* __restrict__ work, int* __restrict__ n_work) {
  const int n_active = *n_active_p, lane = threadIdx.x;
  const int k_begin = blockIdx.x * rows, k_end = min(n_active, k_begin + rows);
  if (k_begin >= k_end) return;
  int lo = 0x7fffffff, hi = -1;
  for (int r = k_begin + lane; r < k_end; r += 64) { const int l = order[r]; lo = min(lo, kmin[l]); hi = max(hi, kmax[l]); }
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
  if (lane == 0) {
    const int t0 = (6 * lo) >> 4, t1 = (6 * hi + 5) >> 4, tl = dp >> 4, nbt = t1 - t0 + 1;
    const int ntiles = nbt * (nbt + 1) / 2 + (tl > t1 ? nbt : 0);
    const int groups = (ntiles + kBandTilesPerGroup - 1) / kBandTilesPerGroup;
    const int base = atomicAdd(n_work, groups);
    for (int g = 0; g < groups; ++g) work[base + g] = make_int4((int)blockIdx.x, g, lo | (hi << 16), k_end);
  }
}

__device__ __forceinline__ void sp_ride(const int vb, const SpArgs& a) {
  if (vb >= a.nblocks) return;
  sp_eliminate_body(vb, a.nodes, a.first, a.tiles, a.rows, a.S, a.ld, a.W, a.wstride, a.Lout, a.fail, a.done, a.src);
}
__global__ __launch_bounds__(256) void k_sp_eliminate(SpArgs a) { sp_ride(blockIdx.x, a); }
__global__ __launch_bounds__(256) void k_sp_eliminate_b(const SpArgs* __restrict__ t) { sp_ride(blockIdx.x, t[blockIdx.y]); }
// The band-limited Schur complement and the FIRST sparse level in one launch: both only ADD (atomically) into entries of S the other
// does not read — the Schur complement touches the pose corner and the pose part of the rhs row, level 0 reads its own (v,ba,bg)
// columns — so they are independent; later levels depend on level 0 and stay launches of their own.
struct SchurSp0Args {
  int n_slices, n_groups, dp, ldE; GP<const double> E, Cd; GP<const int> order, n_active, kmin, kmax; int d_local, ldS; GP<double> S_pose;
  SpArgs sp;             // the sparse level riding in the launch (sp.nblocks == 0: Schur complement only): level 0, or — early form — the first one left
  SpArgs sp_b, sp_c;     // early form: the next two levels, chained behind it inside the launch (SpSrc::wait_counter)
  int nblocks; GP<const int> done; GP<unsigned long long> dbg; int rows;
  GP<const int4> work; int n_work;      // (slice, group, band lo | hi << 16, slice end) items; the sparse levels run in the first workgroups, the items behind
};
__device__ __forceinline__ void schur_sp0_body(const int b, const SchurSp0Args& A) {
  if (b >= A.nblocks) return;
  if (A.work) {
    const int n_a = A.sp.nblocks, n_b = A.sp_b.nblocks, n_c = A.sp_c.nblocks, n_sp = n_a + n_b + n_c;
    if (b < n_a) sp_ride(b, A.sp);
    else if (b < n_a + n_b) sp_ride(b - n_a, A.sp_b);
    else if (b < n_sp) sp_ride(b - n_a - n_b, A.sp_c);
    else {
      const int dv = done_flag_issue(A.done);
      const int4* item = A.work + (b - n_sp);
      const int4 it = *item;
      if (dv) return;
      schur_band_body(it.x, it.y, A.dp, A.ldE, A.E, A.Cd, A.order, A.n_active, A.kmin, A.kmax, A.d_local, A.ldS, A.S_pose, A.dbg ? A.dbg + (size_t)(b - n_sp) * 8 : nullptr, A.rows, item);
    }
    return;
  }
  if (A.done && *A.done) return;
  const int ns = A.n_slices * A.n_groups;
  if (b < ns) schur_band_body(b % A.n_slices, b / A.n_slices, A.dp, A.ldE, A.E, A.Cd, A.order, A.n_active, A.kmin, A.kmax, A.d_local, A.ldS, A.S_pose, A.dbg ? A.dbg + (size_t)b * 8 : nullptr, A.rows);
  else sp_ride(b - ns, A.sp);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_schur_sp0(SchurSp0Args a) { schur_sp0_body(blockIdx.x, a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_schur_sp0_b(const SchurSp0Args* __restrict__ t) { schur_sp0_body(blockIdx.x, t[blockIdx.y]); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_schur_sp0_bt(const SchurSp0Args* __restrict__ t) { schur_sp0_body(blockIdx.y, t[blockIdx.x]); }
struct SpBack {                    // what the back substitution needs of the plan
  SpLevels lv;
  int item0[kSpMaxLevels], items[kSpMaxLevels];   // the level's slice of rows/owner/W
  GP<const SpNode> nodes; GP<const int> rows; GP<const int> owner; GP<const double> W; GP<const double> Linv; GP<const int> perm;
  int off, aug, d_total, total_items, n_nodes, max_count, linv_in_lds;
  int prod_items;                  // > 0: LDS room for that many (row x 9) products => conflict-free two-stage sums; 0: LDS atomics
  GP<unsigned long long> dbg;         // LVF_BACK_TIMING=1: wall_clock64() stamps (100 MHz) at the phase boundaries, else null
};

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the global prefetches
// that are meant to stay in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// back substitution x = L^-T y with y = augmented row L[d][0..d); ONE workgroup of 512 threads.  S points at the DENSE corner
// (row/column `off` of the full matrix), d = its unknowns, Dinv holds L_kk^-T of its 64x64 diagonal blocks.
// Nothing this kernel LOADS from global memory depends on x, so the chain is kept free of global latency:
//   * at entry every thread requests its share of the sparse levels' (row, owner, W[9]) items (registers) and the stored
//     L_bb^-1 (LDS): they arrive while the dense corner is being solved;
//   * dense corner, bottom-up:  rhs = y_blk - sum_{rows r below} L[r][blk]^T x_r  (thread (column c, part) takes rows part,
//     part + 8, ...), x_blk = Dinv_blk rhs (64x64 mat-vec split 8 ways); the gather operands and inverse block of the NEXT block
//     are prefetched while the current one is reduced;
//   * sparse levels in reverse, x_b = L_bb^-T (y_b - W_b^T x_N): the pre-loaded items are multiplied with x from LDS, summed per
//     block (wave-wide when a wave holds one block's rows, LDS atomics otherwise), one thread per (block, component) applies
//     L_bb^-1.  The solution leaves in the natural unknown order through `perm`.
// (S and Dinv are deliberately NOT __restrict__/invariant: LLVM would sink the prefetch loads past the barriers to their uses.)
template <typename T>
__device__ __forceinline__ T ld_off32(const T* base, unsigned byte_off) {     // wave-uniform base + 32-bit lane offset (saddr + voffset form)
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
constexpr int kBT = 512, kBParts = kBT / 64, kBackPre = 256 / kBParts, kBackInv = 64 / kBParts, kTailPre = 6;
// pose_ready (merged back-substitution + step tail, k_backsolve_tail): once the dense corner is solved the POSE part of the step (natural
// unknowns [0, n_pose)) is written out and *pose_ready is raised (release, agent scope) — what the landmark back-substitution waits for
struct BackArgs { GP<const double> Sd; int ld, d; GP<const double> Dinv; GP<double> xout; SpBack sp; GP<const int> done; GP<const double> Ldiag; GP<int> pose_ready = nullptr; int n_pose = 0; int pose_fenced = 0; };
__device__ __forceinline__ void chol_backsolve_body(const BackArgs& A) {
  const int dv = done_flag_issue(A.done);
  const double* S = A.Sd; const int ld = A.ld, d = A.d; const double* Dinv = A.Dinv; double* xout = A.xout; const SpBack& sp = A.sp;
  extern __shared__ double sm[];          // xs[off] | x[nblk*64] | partial[kBParts][64] | rhs[64] | accs[9 max_count] | linv[81 n_nodes]
  const int tid = threadIdx.x, c = tid & 63, part = tid >> 6;
  const int nblk = (d + kNB - 1) / kNB, n = nblk * kNB;
  double* x = sm + sp.off;
  double* partial = x + n;
  double* rhs = partial + kBParts * kNB;
  double* accs = rhs + kNB;
  double* linv = accs + 9 * sp.max_count;
  double* part1 = linv + (sp.linv_in_lds ? 81 * sp.n_nodes : 0);     // [kBT] stage-1 partial sums
  double* prod = part1 + kBT;                                        // [prod_items][9]
  int* snode = reinterpret_cast<int*>(prod + 9 * (size_t)sp.prod_items);   // [n_nodes][2] = (row_off, m)
  int stamp = 0;
  auto mark = [&]() { if (sp.dbg && tid == 0) sp.dbg[stamp++] = wall_clock64(); };
  mark();
  // ---- dense corner
  const int rend = min(n, d);
  double gl[kBackPre], xi[kBackInv], yv;
  // The kernel is ISSUE-bound (one workgroup, two waves per SIMD): every request below is a wave-uniform base (SALU) plus a 32-bit
  // per-lane byte offset, i.e. one VMEM instruction and no VALU address arithmetic, and the row tests are scalar branches -- with
  // 64-bit per-lane addresses and per-lane predicates issuing one block's 41 requests took 1.2-1.8 us of its 2.3-3.4.
  const int part_u = __builtin_amdgcn_readfirstlane(part);
  const unsigned c_off = 8u * (unsigned)c;
  // (no row tests either: k_prepare writes every entry of the padded corner, rows d.. meet x = 0, so the 8-row groups of the
  // 64 (nblk-1-kb) rows below a block are requested and multiplied whole)
  auto prefetch = [&](int kb) {
    const int r0 = kb * kNB, below = nblk - 1 - kb;
    const double* row = S + (size_t)(r0 + kNB + part_u) * ld + r0;
#pragma unroll
    for (int g = 0; g < kBackPre / 8; ++g)
      if (g < below) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gl[8 * g + j] = ld_off32(row + (size_t)(kBParts * (8 * g + j)) * ld, c_off);
      }
    const double* dv = Dinv + (size_t)kb * kNB * kNB + kBackInv * part_u;
#pragma unroll
    for (int t = 0; t < kBackInv; ++t) xi[t] = ld_off32(dv + t, (unsigned)kNB * c_off);
    // the forward-substituted right-hand side: row d of the factor — a panel tile of S for every block but the last, whose part sits in
    // the factored DIAGONAL block (kept in the side buffer: chol_step_body)
    const double* yrow = (d / kNB == kb) ? A.Ldiag + (size_t)kb * kNB * kNB + (size_t)(d - r0) * kNB : S + (size_t)d * ld + r0;
    yv = (r0 + c < d) ? ld_off32(yrow, c_off) : 0.0;
  };
  prefetch(nblk - 1);
  // the stored L_bb^-1 and the node table go to LDS: ALL of a thread's requests are issued before the first LDS write (written as a
  // copy loop the compiler waits for every load in turn — eight dependent round trips at 50 keyframes, 4 of this kernel's 30 us)
  constexpr int kLinvPre = 8;
  double lpre[kLinvPre];
  const int n_linv = sp.linv_in_lds ? 81 * sp.n_nodes : 0;
#pragma unroll
  for (int u = 0; u < kLinvPre; ++u) { const int i = tid + kBT * u; lpre[u] = i < n_linv ? ld_off32((const double*)sp.Linv, 8u * (unsigned)i) : 0.0; }
  SpNode ndpre = SpNode{0, 0, 0, 0};
  if (tid < sp.n_nodes) ndpre = sp.nodes[tid];
#pragma unroll
  for (int u = 0; u < kLinvPre; ++u) { const int i = tid + kBT * u; if (i < n_linv) linv[i] = lpre[u]; }
  for (int i = tid + kBT * kLinvPre; i < n_linv; i += kBT) linv[i] = sp.Linv[i];
  if (tid < sp.n_nodes) { snode[2 * tid] = ndpre.row_off; snode[2 * tid + 1] = ndpre.m; }
  for (int i = tid + kBT; i < sp.n_nodes; i += kBT) { const SpNode nd = sp.nodes[i]; snode[2 * i] = nd.row_off; snode[2 * i + 1] = nd.m; }
  asm volatile("" ::: "memory");
  // ---- requests for the sparse tail, AFTER the first dense prefetch: loads return in order, so the dense corner does not wait
  // for them and they land while it is being solved
  int tR[kTailPre], tK[kTailPre];
  double tW[kTailPre][9];
#pragma unroll
  for (int u = 0; u < kTailPre; ++u) {
    const int g = tid + kBT * u;
    const bool ok = g < sp.total_items;
    tR[u] = -1; tK[u] = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) tW[u][q] = 0.0;
    if (ok) {
      tR[u] = ld_off32((const int*)sp.rows, 4u * (unsigned)g);
      tK[u] = ld_off32((const int*)sp.owner, 4u * (unsigned)g);
#pragma unroll
      for (int q = 0; q < 9; ++q) tW[u][q] = ld_off32((const double*)sp.W, 8u * ((unsigned)q * (unsigned)sp.total_items + (unsigned)g));
    }
  }
  if (dv) return;                          // (every request above is in flight behind the flag's)
  for (int i = tid; i < sp.off + n; i += kBT) sm[i] = 0.0;
  lds_barrier();
  mark();
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int r0 = kb * kNB;
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < kBackPre / 8; ++g)
      if (g < nblk - 1 - kb) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s += gl[8 * g + j] * x[r0 + kNB + part_u + kBParts * (8 * g + j)];
      }
    for (int r = r0 + kNB + part + kBParts * kBackPre; r < rend; r += kBParts) s += S[(size_t)r * ld + r0 + c] * x[r];
    partial[part * kNB + c] = s;
    double xc[kBackInv];
#pragma unroll
    for (int t = 0; t < kBackInv; ++t) xc[t] = xi[t];
    const double yc = yv;
    if (kb > 0) prefetch(kb - 1);          // in flight during the reductions below
    lds_barrier();
    if (part == 0) {
      double a = yc;
#pragma unroll
      for (int pp = 0; pp < kBParts; ++pp) a -= partial[pp * kNB + c];
      rhs[c] = a;
    }
    lds_barrier();
    double t2 = 0.0;
#pragma unroll
    for (int t = 0; t < kBackInv; ++t) t2 += xc[t] * rhs[kBackInv * part + t];
    partial[part * kNB + c] = t2;
    lds_barrier();
    if (part == 0) {
      double a = 0.0;
#pragma unroll
      for (int pp = 0; pp < kBParts; ++pp) a += partial[pp * kNB + c];
      x[r0 + c] = (r0 + c < d) ? a : 0.0;
    }
    lds_barrier();
    mark();
  }
  if (A.pose_ready) {
    // the pose increments are final (every pose row lives in the dense corner): out they go, so that the landmark back-substitution — which
    // reads nothing else of the step — runs in the sibling workgroups of this launch WHILE the sparse levels below are solved here
    // The increments and the flag travel as agent-scope ATOMICS (written through to the coherence point, read there by the consumers): an
    // agent-scope release / acquire FENCE pair instead writes this XCD's dirty L2 lines back and makes every consumer invalidate its L2 —
    // measured: 6 us on this workgroup's path and the landmark pass twice as long (320 workgroups flushing the E rows out of each other's
    // L2).  pose_fenced (LVF_CHAIN_FENCE=2) adds the fences back for A/B.
    {
      int last = -1;
      for (int i = tid; i < A.n_pose; i += kBT) { __hip_atomic_store(xout + i, sm[sp.perm[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); last = i; }
      // (a store is acknowledged to the wave before it is necessarily performed; a load of the same address is ordered behind it and RETURNS:
      // once it is back — the s_waitcnt below — the store is where the consumers read.  Round 3 met the same with returnless atomic adds.)
      if (last >= 0) { const double chk = __hip_atomic_load(xout + last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("" ::"v"(chk)); }
    }
    if (A.pose_fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");          // every wave's stores have been performed before the flag goes up
    if (tid == 0) __hip_atomic_store((int*)A.pose_ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- sparse levels, last eliminated first
  for (int lv = sp.lv.n - 1; lv >= 0; --lv) {
    const int first = sp.lv.first[lv], count = sp.lv.count[lv], item0 = sp.item0[lv], item1 = item0 + sp.items[lv];
    const bool staged = sp.items[lv] <= sp.prod_items;
    if (staged) {
      // (1) products -W x_r of every (block, row) item into LDS (stride 9 doubles: conflict-free)
#pragma unroll
      for (int u = 0; u < kTailPre; ++u) {
        const int g = tid + kBT * u;
        if (g >= item0 && g < item1) {
          const double xr = (tR[u] == sp.aug) ? -1.0 : sm[tR[u]];          // the rhs row carries y_b itself
#pragma unroll
          for (int q = 0; q < 9; ++q) prod[9 * (g - item0) + q] = -tW[u][q] * xr;
        }
      }
      for (int g = max(item0, kBT * kTailPre) + tid; g < item1; g += kBT) {   // items beyond the register window (large windows only)
        const int R = sp.rows[g];
        const double xr = (R == sp.aug) ? -1.0 : sm[R];
#pragma unroll
        for (int q = 0; q < 9; ++q) prod[9 * (g - item0) + q] = -sp.W[(size_t)q * sp.total_items + g] * xr;
      }
      lds_barrier();
      // (2) thread (block k, component q, part j of J) sums its share of the block's rows; (3) thread (k, q) adds the J parts
      int J = 1;
      while (2 * J * 9 * count <= kBT && J < 32) J *= 2;
      if (tid < 9 * count * J) {
        const int j = tid % J, kq = tid / J, q = kq % 9, k = kq / 9;
        const int base = snode[2 * (first + k)] - item0, m = snode[2 * (first + k) + 1];
        double a = 0.0;
        for (int r = j; r < m; r += J) a += prod[9 * (base + r) + q];
        part1[tid] = a;
      }
      lds_barrier();
      if (tid < 9 * count) {
        double a = 0.0;
        for (int j = 0; j < J; ++j) a += part1[tid * J + j];
        accs[tid] = a;
      }
    } else {                                   // no LDS room for the products: LDS atomics (slow under contention, but general)
      for (int i = tid; i < 9 * count; i += kBT) accs[i] = 0.0;
      lds_barrier();
      for (int g = item0 + tid; g < item1; g += kBT) {
        const int R = sp.rows[g], k = sp.owner[g] - first;
        const double xr = (R == sp.aug) ? -1.0 : sm[R];
#pragma unroll
        for (int q = 0; q < 9; ++q) atomicAdd(&accs[9 * k + q], -sp.W[(size_t)q * sp.total_items + g] * xr);
      }
    }
    lds_barrier();
    for (int idx = tid; idx < 9 * count; idx += kBT) {
      const int kk = idx / 9, qq = idx - 9 * kk;
      double v = 0.0;                                       // x_q = sum_{t >= q} (L^-1)[t][q] acc_t
      if (sp.linv_in_lds) { const double* Li = linv + (size_t)(first + kk) * 81; for (int t = qq; t < 9; ++t) v += Li[t * 9 + qq] * accs[9 * kk + t]; }
      else { const double* Li = sp.Linv + (size_t)(first + kk) * 81; for (int t = qq; t < 9; ++t) v += Li[t * 9 + qq] * accs[9 * kk + t]; }
      sm[9 * (first + kk) + qq] = v;                        // block `first + kk` owns S columns 9 (first + kk) ..
    }
    lds_barrier();
    mark();
  }
  for (int i = tid; i < sp.d_total; i += kBT) xout[i] = sm[sp.perm[i]];
  mark();
  if (sp.dbg && tid == 0) sp.dbg[63] = (unsigned long long)stamp;
}
__global__ __launch_bounds__(kBT) void k_chol_backsolve(BackArgs a) { chol_backsolve_body(a); }
__global__ __launch_bounds__(kBT) void k_chol_backsolve_b(const BackArgs* __restrict__ t) { chol_backsolve_body(t[blockIdx.y]); }

// ------------------------------------------------------------------------------------------------ step pieces
// landmark back-substitution: dl = (-gr - e_l . dx_pose) / Cd ; model terms and norms
// (on DENSE rows one thread per landmark beat a wave per landmark, 25.6 vs 29.9 us; with the row limited to the landmark's track
// a per-thread walk diverges (41-54 us) and 16 lanes per landmark is the right shape)
// COH_DX: the pose increments were published by a sibling workgroup of THIS launch as agent-scope atomic stores: read past the non-coherent cache levels
template <int NT = kT, bool COH_DX = false>
__device__ __forceinline__ void landmark_back_body(const int vb, const int nwg, int n_lm, int dp, int ldE, const double* __restrict__ E,
                                                   const double* __restrict__ C, const double* __restrict__ Cd, const double* __restrict__ gr,
                                                   const double* __restrict__ dxc, const double* __restrict__ inv_depth,
                                                   double* __restrict__ dxl, double* __restrict__ invd2, double* __restrict__ scal,
                                                   const int* __restrict__ kmin, const int* __restrict__ kmax) {
  extern __shared__ double sdx[];
  for (int i = threadIdx.x; i < dp; i += NT) sdx[i] = COH_DX ? __hip_atomic_load(dxc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : dxc[i];
  __syncthreads();
  // 16 lanes per landmark: the band of a row is a few 128-byte runs, read coalesced and reduced with four shuffles; the grid is
  // capped and strides over the landmarks so that the three scalar sums cost one atomic per WORKGROUP (thousands of per-wave
  // atomics on the 32 striped slots were most of this kernel's time)
  const int q = threadIdx.x & 15;
  double m = 0.0, n2 = 0.0, x2 = 0.0, gmx = 0.0;
  for (int l = vb * (NT / 16) + (threadIdx.x >> 4); l < n_lm; l += nwg * (NT / 16)) {
    const double* e = E + (size_t)l * ldE;
    double ed = 0.0;
    const int i0 = kmin ? 6 * min(kmin[l], dp / 6) : 0, i1 = kmin ? 6 * (kmax[l] + 1) : dp;   // the row is zero outside the landmark's track
    for (int i = (i0 & ~15) + q; i < i1; i += 16) ed += e[i] * sdx[i];
    ed = row16_sum(ed);
    if (q == 0) {
      const double g_l = gr[l];
      const double dl = (-g_l - ed) / Cd[l];
      gmx = fmax(gmx, fabs(g_l));                   // Ceres' gradient_max_norm runs over every unknown, the inverse depths included
      dxl[l] = dl;
      invd2[l] = inv_depth[l] + dl;                 // the candidate inverse depth (was a second pass in k_apply_step)
      m += -0.5 * dl * ((Cd[l] - C[l]) * dl - g_l);
      n2 += dl * dl; x2 += inv_depth[l] * inv_depth[l];
    }
  }
  __shared__ double red[4][NT / 64];
  m = wave_sum(m); n2 = wave_sum(n2); x2 = wave_sum(x2);
  for (int o = 32; o > 0; o >>= 1) gmx = fmax(gmx, __shfl_down(gmx, o));
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = m; red[1][threadIdx.x >> 6] = n2; red[2][threadIdx.x >> 6] = x2; red[3][threadIdx.x >> 6] = gmx; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double v = 0.0;
    for (int k = 0; k < NT / 64; ++k) v += red[threadIdx.x][k];
    double* dst = scal + (threadIdx.x == 0 ? SC_MODEL : (threadIdx.x == 1 ? SC_DXNORM : SC_XNORM));
    if (v != 0.0) atomicAdd(dst + (vb & (kStripes - 1)), v);
  } else if (threadIdx.x == 3) {
    double v = 0.0;
    for (int k = 0; k < NT / 64; ++k) v = fmax(v, red[3][k]);
    // (non-negative doubles order like their bit patterns; striped like the sums: hundreds of workgroups hitting ONE address serialise,
    // measured +4 us on this launch)
    if (v != 0.0) atomicMax(reinterpret_cast<unsigned long long*>(scal + SC_GMAX + (vb & (kStripes - 1))), (unsigned long long)__double_as_longlong(v));
  }
}
// Model cost change without a pass over H:  (H + D) dx = -g  =>  -dx^T (g + H dx / 2) = 1/2 sum_i dx_i (D_i dx_i - g_i).
// Camera part in k_apply_step (D_i = clamp(B_ii)/radius), landmark part in k_landmark_back.  SC_MODEL accumulates the NEGATED value
// (host flips the sign), keeping the convention model = -SC_MODEL.
// x_new = x [+] dx  (EigenQuaternionParameterization::Plus on the quaternion, plain add elsewhere)
// ... fused with the camera part of the model cost change (k_model_cam's body; d <= 15 n_kf threads of the same grid)
// PARTS: bit 0 = the pose unknowns (sums over [0, 6 n_kf), candidate poses), bit 1 = the (v, ba, bg) unknowns (sums over [6 n_kf, d), candidate
// velocities / biases); 3 = everything (k_step_tail).  k_backsolve_tail runs the two parts in different workgroups at different times.
template <int NT = kT, int PARTS = 3>
__device__ __forceinline__ void apply_step_body(const int vb, int n_kf, int n_lm, StateP s, const double* __restrict__ dxc, const double* __restrict__ dxl,
                                                double* __restrict__ poses2, double* __restrict__ vel2, double* __restrict__ ba2,
                                                double* __restrict__ bg2, double* __restrict__ invd2, double* __restrict__ scal, int d, int ld,
                                                const double* __restrict__ B, const double* __restrict__ gc, double inv_radius,
                                                const unsigned char* __restrict__ pose_const, const JacobiDev jac) {
  const int i = vb * NT + threadIdx.x;
  // step_norm / x_norm as Ceres takes them (trust_region_minimizer.cc): |x - x_plus_delta| and |x| over the AMBIENT parameter vector of the
  // reduced program — the quaternion's four coefficients, not its three tangent increments; constant pose blocks are not part of it
  double m = 0.0, n2 = 0.0, g = 0.0, x2 = 0.0;
  if (i < d && ((PARTS & 1) || i >= 6 * n_kf) && ((PARTS & 2) || i < 6 * n_kf)) {
    const double dx = dxc[i];
    const double h = B[(size_t)i * ld + i], h0d = jac.h0[i];
    m = -0.5 * dx * (lm_damping(h, *jac.frozen ? h0d : h) * inv_radius * dx - gc[i]);      // (first pass: H0 = H, being recorded by the assembly)
    const bool rot = i < 6 * n_kf && (i % 6) < 3;           // rotation increments enter through the quaternion difference below
    n2 = rot ? 0.0 : dx * dx;
    g = fabs(gc[i]);
  }
  const int cm_kf = (i < n_kf && pose_const) ? pose_const[i] : 0;      // bit 0: pose, bits 1..3: v, ba, bg held constant
  if (i < n_kf && (PARTS & 1)) {
    const double* p = s.poses + 7 * i; const double* dlt = dxc + 6 * i;
    const double nrm = sqrt(dlt[0] * dlt[0] + dlt[1] * dlt[1] + dlt[2] * dlt[2]);
    double* o = poses2 + 7 * i;
    if (nrm > 0.0) {
      const double sn = sin(nrm) / nrm, cw = cos(nrm);
      const double dx = sn * dlt[0], dy = sn * dlt[1], dz = sn * dlt[2];
      const double xw = p[3], xx = p[0], xy = p[1], xz = p[2];   // q_delta (x) x, Hamilton [w,x,y,z]
      o[3] = cw * xw - dx * xx - dy * xy - dz * xz;
      o[0] = cw * xx + dx * xw + dy * xz - dz * xy;
      o[1] = cw * xy - dx * xz + dy * xw + dz * xx;
      o[2] = cw * xz + dx * xy - dy * xx + dz * xw;
    } else { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3]; }
    for (int c = 0; c < 3; ++c) o[4 + c] = p[4 + c] + dlt[3 + c];
    for (int c = 0; c < 4; ++c) n2 += (o[c] - p[c]) * (o[c] - p[c]);
    if (!(cm_kf & 1)) for (int c = 0; c < 7; ++c) x2 += p[c] * p[c];
  }
  if (i < n_kf && (PARTS & 2)) {
    const int cm = cm_kf;
    const double* dv = dxc + 6 * n_kf + 9 * i;
    for (int c = 0; c < 3; ++c) { vel2[3 * i + c] = s.vel[3 * i + c] + dv[c]; ba2[3 * i + c] = s.ba[3 * i + c] + dv[3 + c]; bg2[3 * i + c] = s.bg[3 * i + c] + dv[6 + c]; }
    for (int c = 0; c < 3; ++c)
      x2 += ((cm & 2) ? 0.0 : s.vel[3 * i + c] * s.vel[3 * i + c]) + ((cm & 4) ? 0.0 : s.ba[3 * i + c] * s.ba[3 * i + c]) + ((cm & 8) ? 0.0 : s.bg[3 * i + c] * s.bg[3 * i + c]);
  }
  if (i < n_lm) invd2[i] = s.inv_depth[i] + dxl[i];
  if (vb * NT < d) {      // block-uniform
    block_add(m, scal + SC_MODEL); block_add(n2, scal + SC_DXNORM);
    for (int o = 32; o > 0; o >>= 1) g = fmax(g, __shfl_down(g, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(scal + SC_GMAX + (vb & (kStripes - 1))), (unsigned long long)__double_as_longlong(g));
  }
  block_add(x2, scal + SC_XNORM);
}

// landmark back-substitution and the camera-side step / model terms as ONE launch: workgroups [0, g_lm) walk the landmarks,
// the rest apply dx to the keyframe states (independent of the landmark results)
struct TailArgs {
  int g_lm, n_lm, dp, ldE; GP<const double> E, C, Cd, gr, dxc; GP<double> dxl, scal; GP<const int> kmin, kmax; int n_kf; StateP s;
  GP<double> poses2, vel2, ba2, bg2, invd2; int d, ld; GP<const double> B, gc; GP<const double> radius; int nblocks; GP<const int> done;
  GP<const unsigned char> pose_const; JacobiDev jac;
};
__device__ __forceinline__ void step_tail_body(const int bx, const TailArgs& A) {
  if (bx >= A.nblocks || (A.done && *A.done)) return;
  if (bx < A.g_lm) landmark_back_body(bx, A.g_lm, A.n_lm, A.dp, A.ldE, A.E, A.C, A.Cd, A.gr, A.dxc, A.s.inv_depth, A.dxl, A.invd2, A.scal, A.kmin, A.kmax);
  else apply_step_body(bx - A.g_lm, A.n_kf, 0, A.s, A.dxc, A.dxl, A.poses2, A.vel2, A.ba2, A.bg2, A.invd2, A.scal, A.d, A.ld, A.B, A.gc, 1.0 / *A.radius, A.pose_const, A.jac);
}
__global__ __launch_bounds__(kT) void k_step_tail(TailArgs a) { step_tail_body(blockIdx.x, a); }
__global__ __launch_bounds__(kT) void k_step_tail_b(const TailArgs* __restrict__ t) { step_tail_body(blockIdx.x, t[blockIdx.y]); }
__global__ __launch_bounds__(kT) void k_step_tail_bt(const TailArgs* __restrict__ t) { step_tail_body(blockIdx.y, t[blockIdx.x]); }

// The back substitution and the step tail as ONE launch (single-window chain).  The landmark back-substitution needs the POSE part of the
// step only, and the poses are solved first (dense corner, 16 of the back substitution's 28 us); the sparse levels behind it (the
// velocities' and biases' increments, 11 us in one workgroup) and the landmark pass (10 us in hundreds of workgroups) have nothing to do
// with each other, yet as two launches they ran one after the other.  Here workgroup 0 is the back substitution — it publishes the pose
// increments as soon as the dense corner is done, goes on with the sparse levels and finally applies the step to the keyframe states
// (apply_step_body) — and workgroups 1.. are the landmark pass: they wait for the pose increments inside the launch (bounded, like the
// chained sparse levels: on a time-out the hand-over flag is raised, the pass is not judged and the host re-runs it with the two launches
// of old; SpSrc has the rules) and then walk the landmarks.  One launch boundary less and the two tails overlap: 39 -> 29 us at configs[3].
struct BackTailArgs { BackArgs back; TailArgs tail; int g_lm; int fenced; unsigned timeout_ticks; GP<int> fail; };
__global__ __launch_bounds__(kBT) void k_backsolve_tail(BackTailArgs a) {
  if (a.back.done && *a.back.done) return;                      // (every workgroup tests the same flag: nobody waits for a producer that has left)
  const TailArgs& T = a.tail;
  if (blockIdx.x == 0) {
    chol_backsolve_body(a.back);
    // the whole step is in xout (this workgroup wrote it): the velocities' and biases' part is applied here, with its share of the model cost
    // change.  (Requesting what that part reads — B's diagonal, the gradient, the states — ahead of the back substitution was measured
    // SLOWER: 36.1 vs 33.1 us for the launch; the registers it holds across the solve cost more than the three round trips it saves.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    const double inv_radius = 1.0 / *T.radius;
    for (int vb = 0; vb * kBT < max(T.d, T.n_kf); ++vb)
      apply_step_body<kBT, 2>(vb, T.n_kf, 0, T.s, T.dxc, T.dxl, T.poses2, T.vel2, T.ba2, T.bg2, T.invd2, T.scal, T.d, T.ld, T.B, T.gc, inv_radius, T.pose_const, T.jac);
    if (a.back.sp.dbg && threadIdx.x == 0) a.back.sp.dbg[55] = wall_clock64();      // LVF_BACK_TIMING: the step is applied
    return;
  }
  unsigned long long* ldbg = (a.back.sp.dbg && (blockIdx.x == 2 || blockIdx.x == gridDim.x - 1) && threadIdx.x == 0) ? a.back.sp.dbg + (blockIdx.x == 2 ? 56 : 59) : nullptr;
  if (ldbg) ldbg[0] = wall_clock64();
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load((int*)a.back.pose_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > (unsigned long long)a.timeout_ticks) { atomicMax(a.fail, kFailHandover + 90000); break; }
    }
  }
  __syncthreads();
  if (a.fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (ldbg) ldbg[1] = wall_clock64();
  if (blockIdx.x == 1) {
    // the pose part of the step: candidate poses and the pose unknowns' share of the model cost change / step norm (the increments are read
    // at the coherence point into LDS; apply_step_body takes them from there)
    extern __shared__ double sdx_pose[];
    for (int i = threadIdx.x; i < T.dp; i += kBT) sdx_pose[i] = __hip_atomic_load(T.dxc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const double inv_radius = 1.0 / *T.radius;
    for (int vb = 0; vb * kBT < max(T.dp, T.n_kf); ++vb)
      apply_step_body<kBT, 1>(vb, T.n_kf, 0, T.s, sdx_pose, T.dxl, T.poses2, T.vel2, T.ba2, T.bg2, T.invd2, T.scal, T.d, T.ld, T.B, T.gc, inv_radius, T.pose_const, T.jac);
    return;
  }
  landmark_back_body<kBT, true>((int)blockIdx.x - 2, a.g_lm, T.n_lm, T.dp, T.ldE, T.E, T.C, T.Cd, T.gr, T.dxc, T.s.inv_depth, T.dxl, T.invd2, T.scal, T.kmin, T.kmax);
  if (ldbg) ldbg[2] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------ closing an iteration on device
// One workgroup per window: the step-quality test, the trust-region update, the commit of an accepted candidate (a copy of a few tens
// of kilobytes: the window's poses, velocities, biases and inverse depths) and the termination tests — what ceres::Solve's
// TrustRegionMinimizer does on the host between evaluations (declared semantics: oracle/lm.h).  The scalars arrive as 32-way striped
// sums (block_add); `rec` (optional, host-mapped) receives a copy of the control block so a waiting host sees progress without a copy.
struct DecideArgs {
  GP<const double> scal; GP<LmCtl> ctl; GP<LmCtl> rec; GP<int> ticket;
  int n_kf, n_lm;
  GP<double> poses, vel, ba, bg, invd;                 // the state
  GP<const double> poses2, vel2, ba2, bg2, invd2;      // the candidate
  GP<unsigned long long> dbg;                              // LVF_COST_TIMING=1: wall_clock64() stamps (100 MHz), else null
  GP<double> hist;                                         // LVF_LM_HISTORY=1: eight doubles per closed pass (64 passes), else null
};
constexpr int kDT = 256;
// COHERENT: the sums are read past the caches (the caller is the last workgroup of the launch that produced part of them)
template <bool COHERENT = false>
__device__ __forceinline__ void lm_decide_body(const DecideArgs& A) {
  __shared__ int s_commit, s_skip, s_iter, s_done;
  __shared__ double s_sum[8];
  LmCtl* c = A.ctl;
  if (A.dbg && threadIdx.x == 0) A.dbg[1] = wall_clock64();
  if (threadIdx.x == 0) s_skip = c->done;
  // the six striped sums: wave w adds the 32 stripes of slots w, w + 4 (lanes 32..63 contribute zero)
  {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int slot = wv; slot < 6; slot += kDT / 64) {
      double v = lane < kStripes ? (COHERENT ? __hip_atomic_load(A.scal + slot * kStripes + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : A.scal[slot * kStripes + lane]) : 0.0;
      if (slot == SC_GMAX / kStripes) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o)); }      // the gradient's max norm: a max over its stripes
      else v = wave_sum(v);
      if (lane == 0) s_sum[slot] = v;
    }
  }
  __syncthreads();
  if (s_skip) return;
  if (A.dbg && threadIdx.x == 0) A.dbg[2] = wall_clock64();
  if (threadIdx.x < kStripes) const_cast<double*>((const double*)A.scal)[SC_COST + threadIdx.x] = 0.0;      // read above; the next linearisation adds into it
  if (threadIdx.x == 0) {
    // the fields of the control block are read up front (independent requests, one wait) and written back once at the end: read and
    // written where the logic uses them they were 1.2 us of dependent traffic.  (A whole-struct copy goes through a scratch segment.)
    struct { double radius, decrease, last_radius, cost, initial_cost, cost_before, cost_after, model, dxnorm, xnorm, gmax;
             double function_tol, gradient_tol, parameter_tol, min_rel_decrease; int max_iters, iter, successes, invalid_run, accepted, solved, done, termination, why, rejected; } lc;
    lc.radius = c->radius; lc.decrease = c->decrease; lc.cost = c->cost; lc.initial_cost = c->initial_cost;
    lc.function_tol = c->function_tol; lc.gradient_tol = c->gradient_tol; lc.parameter_tol = c->parameter_tol; lc.min_rel_decrease = c->min_rel_decrease;
    lc.max_iters = c->max_iters; lc.iter = c->iter; lc.successes = c->successes; lc.invalid_run = c->invalid_run; lc.done = c->done; lc.termination = c->termination; lc.why = c->why; lc.rejected = c->rejected;
    const int hfail = *reinterpret_cast<const int*>(A.scal + SC_FAIL);
    const double cost_before = s_sum[SC_COST / kStripes], cost_new = s_sum[SC_COST_NEW / kStripes], model = -s_sum[SC_MODEL / kStripes];
    const double dxnorm = sqrt(s_sum[SC_DXNORM / kStripes]), xnorm = sqrt(s_sum[SC_XNORM / kStripes]);
    const double gmax = s_sum[SC_GMAX / kStripes];             // a max over the stripes (the stored bit patterns are the doubles')
    const bool solved = hfail == 0 && isfinite(cost_new) && isfinite(model);
    const int it = lc.iter;
    if (it == 0) { lc.initial_cost = cost_before; lc.cost = cost_before; }
    lc.cost_before = cost_before; lc.model = model; lc.dxnorm = dxnorm; lc.xnorm = xnorm; lc.gmax = gmax; lc.solved = solved ? 1 : 0;
    lc.last_radius = lc.radius;
    bool accepted = false, done = false;
    int termination = 1, why = LVF_WHY_MAX_ITERATIONS;
    // ceres::Solve's TrustRegionMinimizer, in its order (declared semantics + citations: oracle/lm.h lm_solve).
    // Top of the loop (FinalizeIterationAndCheckIfMinimizerCanContinue): the gradient at the point this pass linearised — it ends the
    // solve before a step is taken, so the pass is NOT an iteration; the smallest trust region likewise.
    // a chained sparse level gave up waiting for the level below (SpSrc): nothing about this step is judged — the loop stops where it is
    // (state, radius and counters untouched) and the host re-runs the iteration with un-chained launches
    if (hfail >= kFailHandover) { done = true; termination = 2; why = LVF_WHY_HANDOVER; }
    else if (gmax <= lc.gradient_tol) { done = true; termination = 0; why = LVF_WHY_GRADIENT; }
    else if (lc.radius < 1e-32) { done = true; termination = 0; why = LVF_WHY_MIN_RADIUS; }
    else {
      // (num_iterations = what Ceres records in Summary::iterations: accepted, rejected and invalid steps; a trial step that ends the solve
      // through the parameter / function tolerance returns before it is recorded)
      const bool valid = solved && model > 0.0;      // ComputeTrustRegionStep: solver failure or model_cost_change <= 0 = INVALID step
      if (!valid) {
        lc.iter = it + 1;
        lc.rejected += 1;
        lc.invalid_run += 1;
        if (lc.invalid_run >= 5) { done = true; termination = 2; why = LVF_WHY_INVALID_STEPS; }      // max_num_consecutive_invalid_steps
        else lc.radius *= 0.5;                       // LevenbergMarquardtStrategy::StepIsInvalid (decrease factor untouched)
      } else {
        lc.invalid_run = 0;
        // parameter tolerance, then function tolerance: both BEFORE the step-quality test, and neither takes the candidate
        if (dxnorm <= lc.parameter_tol * (xnorm + lc.parameter_tol)) { done = true; termination = 0; why = LVF_WHY_PARAMETER; }
        else if (fabs(cost_before - cost_new) <= lc.function_tol * cost_before) { done = true; termination = 0; why = LVF_WHY_FUNCTION; }
        else {
          lc.iter = it + 1;
          const double rho = (cost_before - cost_new) / model;
          if (rho > lc.min_rel_decrease) {
            accepted = true;
            const double t = 2.0 * rho - 1.0;
            lc.radius = fmin(lc.radius / fmax(1.0 / 3.0, 1.0 - t * t * t), 1e16);
            lc.decrease = 2.0;
            lc.successes += 1;
            lc.cost = cost_new;
          } else {
            lc.radius = lc.radius / lc.decrease;
            lc.decrease *= 2.0;
            lc.rejected += 1;
          }
        }
      }
      // after the step, Finalize's order again: the iteration cap first, then the smallest trust region (the gradient at an accepted point is
      // only known to the next pass)
      if (!done && lc.iter >= lc.max_iters) { done = true; termination = 1; why = LVF_WHY_MAX_ITERATIONS; }
      else if (!done && lc.radius < 1e-32) { done = true; termination = 0; why = LVF_WHY_MIN_RADIUS; }
    }
    lc.cost_after = accepted ? cost_new : (solved ? cost_new : cost_before);
    lc.accepted = accepted ? 1 : 0;
    if (done) { lc.termination = termination; lc.why = why; lc.done = 1; }
    s_commit = accepted ? 1 : 0;
    c->radius = lc.radius; c->decrease = lc.decrease; c->last_radius = lc.last_radius; c->cost = lc.cost; c->initial_cost = lc.initial_cost;
    c->cost_before = lc.cost_before; c->cost_after = lc.cost_after; c->model = lc.model; c->dxnorm = lc.dxnorm; c->xnorm = lc.xnorm; c->gmax = lc.gmax;
    c->iter = lc.iter; c->successes = lc.successes; c->invalid_run = lc.invalid_run; c->accepted = lc.accepted; c->solved = lc.solved;
    c->done = lc.done; c->termination = lc.termination; c->why = lc.why; c->rejected = lc.rejected;
    if (hfail < kFailHandover) c->jfrozen = 1;      // the Jacobi scaling of this solve is the first pass's (a pass that is re-run after a hand-over time-out takes it again)
    s_iter = lc.iter; s_done = lc.done;
    if (A.hist) {                                      // diagnostic: what this pass decided on
      double* h = A.hist + 8 * (it & 63);
      h[0] = (double)it; h[1] = cost_before; h[2] = cost_new; h[3] = model; h[4] = accepted ? 1.0 : 0.0; h[5] = (double)hfail; h[6] = lc.last_radius; h[7] = gmax;
    }
    if (A.dbg) A.dbg[3] = wall_clock64();
  }
  __syncthreads();
  if (s_commit) {
    // (eight 16-byte requests per thread in flight: as a load-store loop the 80 KB of inverse depths took 3.9 us of this one workgroup)
    const double2* __restrict__ p2 = reinterpret_cast<const double2*>((const double*)A.invd2);
    double2* __restrict__ q2 = reinterpret_cast<double2*>((double*)A.invd);
    const int n2 = A.n_lm / 2;
    for (int i0 = 0; i0 < n2; i0 += 8 * kDT) {      // (n2 > 0 inside)
      double vx[8], vy[8];             // (scalars: an array of double2 is not split into registers and lands in scratch)
#pragma unroll
      for (int u = 0; u < 8; ++u) { const double2 t = p2[min(i0 + u * kDT + (int)threadIdx.x, n2 - 1)]; vx[u] = t.x; vy[u] = t.y; }      // (unconditional, clamped)
#pragma unroll
      for (int u = 0; u < 8; ++u) q2[min(i0 + u * kDT + (int)threadIdx.x, n2 - 1)] = make_double2(vx[u], vy[u]);      // (past the end: the last element again, same value)
    }
    if ((A.n_lm & 1) && threadIdx.x == 0) A.invd[A.n_lm - 1] = A.invd2[A.n_lm - 1];
    for (int i = threadIdx.x; i < 7 * A.n_kf; i += kDT) A.poses[i] = A.poses2[i];
    for (int i = threadIdx.x; i < 3 * A.n_kf; i += kDT) { A.vel[i] = A.vel2[i]; A.ba[i] = A.ba2[i]; A.bg[i] = A.bg2[i]; }
  }
  if (A.dbg && threadIdx.x == 0) A.dbg[4] = wall_clock64();
  // the host only polls `iter` and `done` of its mirror (wait_for_iteration): two uncached stores to the pinned record instead of a
  // system-scope fence and a copy of the whole block; LAST, so that no load of this workgroup queues behind a write that crosses PCIe
  if (A.rec && threadIdx.x == 0) {
    __hip_atomic_store(&A.rec->why, c->why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);        // (ahead of `done`: a host that sees done also sees why the loop ended)
    __hip_atomic_store(&A.rec->done, s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&A.rec->iter, s_iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(kDT) void k_lm_decide(DecideArgs a) { lm_decide_body(a); }
__global__ __launch_bounds__(kDT) void k_lm_decide_b(const DecideArgs* __restrict__ t) { lm_decide_body(t[blockIdx.y]); }

// The candidate-cost pass and the decision in ONE launch: every workgroup adds its share of the candidate cost, takes a ticket, and the
// one that draws the last ticket closes the iteration (the sums live in atomics, so a plain completion wait orders them before the
// ticket; nothing else this launch writes is read by the decision).  Saves the k_lm_decide launch (~8 us of an iteration).
static_assert(kDT == kT, "the last workgroup of the cost pass runs the decision with its own threads");
__device__ __forceinline__ void cost_decide_body(const int b, const CostArgs& A, const DecideArgs& D, const int end_zero) {
  if (b >= A.nblocks) {
    // the accumulators of the linearisation (B, gc, C, g_rho, ...) were last read by k_step_tail: cleared here, beside the cost pass, for
    // the next iteration (never gated: a finished solve leaves them clean for the next one)
    if (end_zero && b - A.nblocks < A.zero_wgs) zero_list_share(A.zero, b - A.nblocks, A.zero_wgs);
    return;
  }
  if (A.done && *A.done) return;
  __shared__ double s_part[kT / 64];
  if (D.dbg && threadIdx.x == 0 && (b == 0 || b == A.g_imu)) D.dbg[b == 0 ? 8 : 12] = wall_clock64();
  {
    // the workgroup's sum goes out as a RETURNING atomic: its result can only come back once the add has been performed, and the
    // ticket below is drawn after it — so every add of this workgroup is in the sum before its ticket is drawn (a release fence here would write
    // the L2 back once per workgroup: measured 7 % slower for 8 windows than the separate decision launch)
    double c;
    if (b >= A.g_imu) c = cost_visual_value(b - A.g_imu, A);
    else {                                             // one ImuError factor at the candidate: 1/2 |sqrt_info r|^2
      __shared__ double sS[225];
      __shared__ double sP[OFF_COV];                     // sum_dt, linearisation biases, deltas and the 15 x 15 Jacobian of the pre-integration
      __shared__ double sr0[16];
      const int f = b;
      for (int k = threadIdx.x; k < 225; k += kT) sS[k] = A.imu.sqrt_info[(size_t)f * 225 + k];
      for (int k = threadIdx.x; k < OFF_COV; k += kT) sP[k] = A.imu.pre[(size_t)f * kPre + k];
      __syncthreads();
      // (from LDS the one lane's ~60 operand reads cost nothing to repeat, so the compiler does not hold them all in registers: this
      // path shares its kernel with the visual cost pass, whose occupancy it would otherwise set)
      if (D.dbg && threadIdx.x == 0 && b == 0) D.dbg[9] = wall_clock64();
      if (threadIdx.x == 0) imu_raw<false>(f, sP, A.imu.kf_i, A.imu.kf_j, A.s.poses, A.s.vel, A.s.ba, A.s.bg, sr0, nullptr);
      __syncthreads();
      if (D.dbg && threadIdx.x == 0 && b == 0) D.dbg[10] = wall_clock64();
      const double r = imu_weighted_residual(threadIdx.x, sS, sr0);      // rows 0..14 in lanes 0..14 of wave 0
      c = threadIdx.x < 15 ? 0.5 * r * r : 0.0;
    }
    // one add per WORKGROUP (the four wave sums meet in LDS first): the adds of a stripe serialise at their address
    const double v = wave_sum(c);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
  }
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < kT / 64; ++k) v += s_part[k];
    if (v != 0.0) {
      const double old = atomicAdd(A.cost + (b & (kStripes - 1)), v);
      asm volatile("" ::"v"(old) : "memory");
    }
    if (D.dbg && (b == 0 || b == A.g_imu)) D.dbg[b == 0 ? 11 : 13] = wall_clock64();
    const int t = atomicAdd(D.ticket, 1);
    s_last = t == A.nblocks - 1;
    if (s_last) atomicExch(D.ticket, 0);
  }
  __syncthreads();
  if (s_last) lm_decide_body<true>(D);
  if (s_last && D.dbg && threadIdx.x == 0) D.dbg[5] = wall_clock64();
}
// end_zero = 0 keeps the accumulators of this iteration's linearisation (per-call API: lvf_problem_download_reduced rebuilds the damped system from them)
__global__ __launch_bounds__(kT) void k_cost_decide(CostArgs a, DecideArgs d, int end_zero) { cost_decide_body(blockIdx.x, a, d, end_zero); }
__global__ __launch_bounds__(kT) void k_cost_decide_b(const CostArgs* __restrict__ t, const DecideArgs* __restrict__ d, int end_zero) { cost_decide_body(blockIdx.x, t[blockIdx.y], d[blockIdx.y], end_zero); }
__global__ __launch_bounds__(kT) void k_cost_decide_bt(const CostArgs* __restrict__ t, const DecideArgs* __restrict__ d, int end_zero) { cost_decide_body(blockIdx.y, t[blockIdx.x], d[blockIdx.x], end_zero); }

// ================================================================================================ host side
static StateP state_ptrs(const lvf_state* st) { return StateP{st->poses.p, st->vel.p, st->ba.p, st->bg.p, st->inv_depth.p, st->w_visual.p}; }
static inline int grid(int n) { return (n + kT - 1) / kT; }

// The argument blocks of ONE LM iteration of a window.  Nothing in them changes from iteration to iteration (the trust-region radius,
// the accept / reject state and the termination flag live in the device-resident LmCtl; an accepted candidate is COPIED into the state
// buffers by k_lm_decide), so they are built once per problem_configure and
//   * passed by value to the single-window launches, or
//   * stored as one entry per window in device tables, every launch of the chain then covering a whole batch of windows (blockIdx.y).
struct Chain {
  bool fast = false;            // merged linearisation (sorted TwoFrame work list) available
  bool batchable = false;       // every launch of the iteration has a table form (fast + band Schur merged with sparse level 0 + no priors)
  bool has_imu = false, has_prior = false, imu_in_cost = false;
  ZeroList zero_end{};
  ZeroList zero{};              // everything a linearisation accumulates into (explicit k_zero_multi when the accumulators are not known clean)
  ImuArgs imu_lin{}, imu_cost{};
  LinArgs lin{}; size_t lin_lds = 0;
  TfReduceArgs red{}; size_t red_lds = 0;     // compact mode: the slabs of the TwoFrame linearisation -> B, gc
  PrepArgs prep{};              // classic form (stores the whole lower triangle); also what the parity taps use
  // Early sparse levels.  The (v, ba, bg) columns only ever receive ImuError terms, the LM damping and the updates of lower levels, so a
  // level can form its columns from B itself (SpSrc) as soon as the linearisation launch is over: level 0 rides in the k_tf_reduce
  // launch, level 1 in k_prepare's, level 2 in the Schur complement's, and only what is left takes launches of its own (at 50 keyframes
  // two instead of four).  For that S is cleared with the accumulators and k_prepare ADDS the dense corner (prep_early).
  bool early = false;
  PrepArgs prep_early{}; size_t prep_lds = 0;
  int first_own_level = 0;      // sparse levels [first_own_level, n_levels) are launches of their own
  bool merged_level0 = false;
  SchurSp0Args ssp0{}; size_t ssp0_lds = 0;
  int n_levels = 0; SpArgs sp[kSpMaxLevels]; int sp_lds[kSpMaxLevels] = {0};
  CholArgs chol{};
  BackArgs back{}; size_t back_lds = 0;
  TailArgs tail{}; size_t tail_lds = 0;
  bool back_tail_merged = false; BackTailArgs bt{}; size_t bt_lds = 0;      // k_backsolve_tail (single-window chain, chained levels allowed)
  CostArgs cost{};
  DecideArgs dec{};
};

void stage_clock_free(StageClock* k);
void batch_orphan(lvf_problem_batch* b, lvf_problem* dying);
}  // namespace lvf
lvf_problem::~lvf_problem() {
  // a batch that still borrows this problem must never dereference it again: it is marked orphaned (its calls fail with LVF_ERR_STATE)
  // and forgets every member, so destroying it later touches nothing
  for (lvf_problem_batch* b : batches) lvf::batch_orphan(b, this);
  delete chain;
  lvf::stage_clock_free(clk);
  if (rec && !lvf::HostPinPool::get().give(rec, lvf::Pool::bucket(sizeof(lvf::LmCtl)))) (void)hipHostFree(rec);
  if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
  if (ev_band) (void)hipEventDestroy(ev_band);
}
namespace lvf {

static void fill_cost_visual(const lvf_problem* p, CostVisual& a) {
  a = CostVisual{};
  if (p->tc && p->tc->n) {
    a.n_tc = p->tc->n; a.tc_lo = (const double2*)p->tc->ob_a.p; a.tc_ro = (const double2*)p->tc->ob_b.p; a.tc_lm = p->tc->idx_a.p; a.tc_kf = p->tc->idx_b.p;
    a.tc_w = p->tc->wblk.n ? p->tc->wblk.p : nullptr; a.tc_left = p->tc->cam_a; a.tc_right = p->tc->cam_b;
  }
  if (p->tf && p->tf->n) {
    a.n_tf = p->tf->n; a.tf_fo = p->tf_fo(); a.tf_ob = p->tf_ob(); a.tf_lm = p->tf_lm(); a.tf_k1 = p->tf_k1();
    a.tf_k2 = p->tf_k2(); a.tf_left = p->tf->cam_a; a.tf_right = p->tf->cam_b;
  }
  if (p->po && p->po->n) {
    a.n_po = p->po->n; a.po_ob = (const double2*)p->po->ob_a.p; a.po_kf = p->po->idx_a.p; a.po_pwi = p->po->idx_b.p; a.po_pw = p->po->table.p; a.po_cam = p->po->cam_a;
  }
  a.g_tc = grid(a.n_tc); a.g_tf = grid(a.n_tf);
}

// a state-shaped VIEW of borrowed device pointers (for the launchers that take an lvf_state); never destroyed with live pointers
struct StateView {
  lvf_state v;
  StateView(lvf_ctx* ctx, int n_kf, int n_lm, double* poses, double* vel, double* ba, double* bg, double* invd, double* wv) {
    v.ctx = ctx; v.n_kf = n_kf; v.n_lm = n_lm;
    v.poses.p = poses; v.vel.p = vel; v.ba.p = ba; v.bg.p = bg; v.inv_depth.p = invd; v.w_visual.p = wv;
  }
  ~StateView() { v.poses.p = v.vel.p = v.ba.p = v.bg.p = v.inv_depth.p = v.w_visual.p = nullptr; }
};

// accumulates 1/2 sum rho into *cost_slot at the given state (residual-only pass); not gated by the LM control block
static int enqueue_cost(lvf_problem* p, const StateP& s, const lvf_state* imu_state_view, double huber, double* cost_slot) {
  hipStream_t q = p->ctx->stream;
  CostArgs c{};
  fill_cost_visual(p, c.a);
  c.n_kf = p->n_kf; c.s = s; c.huber = huber; c.cost = cost_slot; c.done = nullptr;
  c.nblocks = c.a.g_tc + c.a.g_tf + grid(c.a.n_po);
  if (c.nblocks > 0) hipLaunchKernelGGL(k_cost_visual, dim3(c.nblocks), dim3(kT), 0, q, c);
  if (p->imu && p->imu->n) {
    static_assert(kStripes == 32, "k_imu stripes its cost over 32 slots");
    LVF_TRY(launch_imu(p->imu, imu_state_view, false, cost_slot));      // residuals and their cost in one launch
  }
  if (p->prior && p->prior->n) {
    LVF_TRY(launch_pose_prior(p->prior, imu_state_view, false));
    hipLaunchKernelGGL(k_cost_sq, dim3(grid(6 * p->prior->n)), dim3(kT), 0, q, 6 * p->prior->n, p->prior->res.p, cost_slot);
  }
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

static void fill_back_args(lvf_problem* p, BackArgs& ba, size_t* lds_bytes) {
  SpBack sb{};
  sb.lv = p->sp_levels; sb.rows = p->sp_rows.p; sb.owner = p->sp_owner.p; sb.W = p->sp_W.p; sb.Linv = p->sp_L.p; sb.perm = p->perm.p;
  sb.off = p->off; sb.aug = p->aug; sb.d_total = p->d;
  sb.dbg = nullptr;
  int max_count = 0;
  for (int lv = 0; lv < p->sp_levels.n; ++lv) {
    max_count = std::max(max_count, p->sp_levels.count[lv]);
    sb.item0[lv] = p->sp_item0[lv]; sb.items[lv] = p->sp_items[lv];
  }
  const int n_nodes = p->sp_levels.n ? p->sp_levels.first[p->sp_levels.n - 1] + p->sp_levels.count[p->sp_levels.n - 1] : 0;
  sb.total_items = p->sp_levels.n ? p->sp_item0[p->sp_levels.n - 1] + p->sp_items[p->sp_levels.n - 1] : 0;
  sb.n_nodes = n_nodes; sb.max_count = max_count;
  sb.nodes = p->sp_nodes.p;
  static const bool big_lds = [] {        // up to 160 KB of LDS per workgroup on gfx950; the default cap for dynamic LDS is 64 KB
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_backsolve), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_backsolve_b), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) == hipSuccess;
  }();
  const size_t lds_cap = big_lds ? 156 * 1024 : 64 * 1024;
  size_t doubles = (size_t)p->off + (size_t)((p->ndense + 63) / 64) * 64 + (size_t)(kBParts + 1) * kNB + 9 * (size_t)max_count + kBT + (size_t)n_nodes + 2;
  sb.linv_in_lds = (doubles + 81 * (size_t)n_nodes) * sizeof(double) <= 48 * 1024 ? 1 : 0;
  if (sb.linv_in_lds) doubles += 81 * (size_t)n_nodes;
  int max_items = 0;
  for (int lv = 0; lv < p->sp_levels.n; ++lv) max_items = std::max(max_items, p->sp_items[lv]);
  sb.prod_items = ((doubles + 9 * (size_t)max_items) * sizeof(double) <= lds_cap) ? max_items : 0;
  doubles += 9 * (size_t)sb.prod_items;
  *lds_bytes = doubles * sizeof(double);
  ba.Sd = p->S.p + (size_t)p->off * (p->ld + 1); ba.ld = p->ld; ba.d = p->ndense; ba.Dinv = p->Dinv.p; ba.Ldiag = p->Ldiag.p; ba.xout = p->dxc.p; ba.sp = sb;
}

// the work list of the band Schur complement for the current rows-per-slice setting; its length is part of the launch grid
static inline int band_rows_clamped(int rows) { return std::min(kBandRowsMax, std::max(16, rows)); }
// work list of p's band Schur complement for `rows` landmark rows per slice, into buffers of the caller's (the problem's own list, or a
// batch's: a batch sums over wider slices and must not touch its members)
// `defer`: do not wait for the item count — an event is recorded behind its copy and await_band_work() collects it right before the first
// launch that needs it (the Schur complement's), by which time the linearisation launches enqueued in between have long kept the device busy
static int build_band_work(lvf_problem* p, int rows_in, DevBuf<int4>& work, int* n_work, bool defer = false) {
  hipStream_t q = p->ctx->stream;
  const int rows = band_rows_clamped(rows_in);
  const int n_slices = (p->n_lm + rows - 1) / rows;
  const int nt = p->ldE / 16, groups_max = (nt * (nt + 1) / 2 + kBandTilesPerGroup - 1) / kBandTilesPerGroup;
  LVF_TRY(work.ensure((size_t)n_slices * groups_max)); LVF_TRY(p->n_band_work_dev.ensure(1)); LVF_TRY(p->h_n_band_work.reserve(1));
  LVF_HIP(hipMemsetAsync(p->n_band_work_dev.p, 0, sizeof(int), q));
  hipLaunchKernelGGL(k_band_work, dim3(n_slices), dim3(64), 0, q, rows, p->dp, p->lm_nactive.p, p->lm_order.p, p->lm_kmin.p, p->lm_kmax.p, work.p, p->n_band_work_dev.p);
  LVF_HIP(hipGetLastError());
  LVF_HIP(hipMemcpyAsync(p->h_n_band_work.p, p->n_band_work_dev.p, sizeof(int), hipMemcpyDeviceToHost, q));
  if (defer) {
    if (!p->ev_band) LVF_HIP(hipEventCreateWithFlags(&p->ev_band, hipEventDisableTiming));
    LVF_HIP(hipEventRecord(p->ev_band, q));
    p->band_pending = true;
    *n_work = 0;
    return LVF_OK;
  }
  LVF_HIP(hipStreamSynchronize(q));
  *n_work = p->h_n_band_work[0];
  return LVF_OK;
}
static int ensure_band_work(lvf_problem* p) {
  if (!p->band_ready || p->n_lm == 0 || p->band_rows_built == p->band_rows) return LVF_OK;
  LVF_TRY(build_band_work(p, p->band_rows, p->band_work, &p->n_band_work, /*defer=*/true));
  p->band_rows_built = p->band_rows;
  return LVF_OK;
}

// (re)builds the argument blocks of an iteration from the problem's CURRENT buffers (call after problem_configure / set_pose_priors)
static int build_chain(lvf_problem* p) {
  if (!p->chain) p->chain = new Chain();
  Chain& c = *p->chain;
  c = Chain();
  p->accum_clean = false;           // buffers may have been re-allocated
  LVF_TRY(ensure_band_work(p));
  LVF_TRY(p->ctl.ensure(1));
  if (!p->rec) {
    void* h = HostPinPool::get().take(Pool::bucket(sizeof(LmCtl)));       // (pinned blocks are recycled: lvf_internal.hpp)
    if (!h) LVF_HIP(hipHostMalloc(&h, Pool::bucket(sizeof(LmCtl)), hipHostMallocDefault));
    p->rec = static_cast<LmCtl*>(h);
    std::memset(p->rec, 0, sizeof(LmCtl));
  }
  LmCtl* ctl = p->ctl.p;
  const int* done = &ctl->done;
  const double* radius = &ctl->radius;
  LVF_TRY(p->jh0.ensure((size_t)p->d + p->n_lm + 1));
  const JacobiDev jac{p->jh0.p, &ctl->jfrozen};
  const StateP s = state_ptrs(p->st);
  const StateP s2{p->poses2.p, p->vel2.p, p->ba2.p, p->bg2.p, p->invd2.p, p->st->w_visual.p};
  double* cost = p->scal.p + SC_COST;
  c.fast = p->tf && p->tf->n && p->tf_work.n && p->n_kf <= kMaxStagedKf;
  c.has_imu = p->imu && p->imu->n;
  c.has_prior = p->prior && p->prior->n;
  const size_t schur_lds = p->n_lm ? ((size_t)kSchurRows * (p->ldE + 16) + kSchurRows) * sizeof(double) + kBandRowsMax * sizeof(int) : 0;
  const bool schur_merged = p->n_lm && p->band_ready && schur_lds <= 64 * 1024 && p->sp_levels.n > 0 && (size_t)p->sp_shmem[0] <= 64 * 1024;
  {
    // LVF_EARLY_LEVELS=0: the classic order (A/B measurements); LVF_POISON_S fills S with NaN before the assembly, which only the classic form survives
    static const bool early_on = [] { const char* e = std::getenv("LVF_EARLY_LEVELS"); return !(e && e[0] == '0') && std::getenv("LVF_POISON_S") == nullptr; }();
    bool fits = true;
    for (int lv = 0; lv < std::min(3, p->sp_levels.n); ++lv) fits = fits && (size_t)p->sp_shmem[lv] <= 64 * 1024;
    c.early = early_on && c.fast && c.has_imu && schur_merged && fits && p->sp_levels.n >= 2;      // (one level: it already hides in the Schur launch)
  }
  // the merged back substitution + step tail (k_backsolve_tail) hands the pose increments over inside a launch: allowed where in-launch
  // hand-overs are allowed at all (a problem whose hand-over timed out keeps its launches apart: lvf_problem::no_chain); its flag lives in the
  // arrival-counter block, which is then cleared with the accumulators whether or not sparse levels are chained
  static const bool bt_merge_on = [] { const char* e = std::getenv("LVF_BACK_TAIL_MERGE"); return !(e && e[0] == '0'); }();
  static const bool bt_chain_on = [] { const char* e = std::getenv("LVF_CHAIN_LEVELS"); return !(e && std::atoi(e) <= 0); }();
  const bool bt_wanted = bt_merge_on && bt_chain_on && !p->no_chain && p->n_lm > 0;
  {
    int k = 0;
    static const bool tri_on = [] { const char* e = std::getenv("LVF_ZERO_TRI"); return !(e && e[0] == '0'); }();
    bool overflow = false;
    auto add = [&](double* ptr, size_t n, int tri = 0) {
      if (!(ptr && n)) return;
      if (k >= kZeroListMax) { overflow = true; return; }      // (the struct travels by value: never write past its arrays)
      c.zero.p[k] = ptr; c.zero.n[k] = n; c.zero.tri[k] = (tri_on && tri % 2 == 0) ? tri : 0; ++k;
    };
    add(p->B.p, (size_t)p->dpad * p->dpad, p->dpad); add(p->gc.p, p->dpad);
    if (p->n_lm) { if (!p->compact) add(p->E.p, (size_t)p->n_lm * p->ldE); add(p->C.p, p->n_lm); add(p->gr.p, p->n_lm); }
    if (c.early) add(p->S.p, (size_t)p->ld * p->ld, p->ld);      // early sparse levels add into S before k_prepare does
    if (c.early || bt_wanted) add(p->sp_sync.p, kSpMaxLevels);     // the arrival counters of chained levels / the pose hand-over flag
    c.zero_end = c.zero; c.zero_end.count = k;         // cleared at the END of an iteration, beside the cost pass (the scalars: by the decision itself)
    add(p->scal.p, SC_N);
    c.zero.count = k;
    LVF_REQUIRE(!overflow, "build_chain: more than %d accumulator arrays (raise kZeroListMax)", kZeroListMax);
  }
  if (c.has_imu) {
    fill_imu_args(p->imu, s.poses, s.vel, s.ba, s.bg, nullptr, nullptr, done, &c.imu_lin);
    fill_imu_args(p->imu, s2.poses, s2.vel, s2.ba, s2.bg, p->scal.p + SC_COST_NEW, nullptr, done, &c.imu_cost);
  }
  if (c.fast) {
    LinVisual& a = c.lin.v;
    a = LinVisual{};
    a.n_tfw = (int)p->tf_work.n; a.work = p->tf_work.p; a.tf_fo = p->tf_fo(); a.tf_ob = p->tf_ob();
    a.tf_lm = p->tf_lm(); a.tf_k1 = p->tf_k1(); a.tf_left = p->tf->cam_a; a.tf_right = p->tf->cam_b; a.unique_lk2 = p->tf_unique_lk2 ? 1 : 0;
    if (p->tc && p->tc->n) {
      a.n_tc = p->tc->n; a.tc_lo = (const double2*)p->tc->ob_a.p; a.tc_ro = (const double2*)p->tc->ob_b.p; a.tc_lm = p->tc->idx_a.p; a.tc_kf = p->tc->idx_b.p;
      a.tc_w = p->tc->wblk.n ? p->tc->wblk.p : nullptr; a.tc_left = p->tc->cam_a; a.tc_right = p->tc->cam_b;
    }
    if (p->po && p->po->n) {
      a.n_po = p->po->n; a.po_ob = (const double2*)p->po->ob_a.p; a.po_kf = p->po->idx_a.p; a.po_pwi = p->po->idx_b.p; a.po_pw = p->po->table.p; a.po_cam = p->po->cam_a;
    }
    a.g_tc = grid(a.n_tc); a.g_po = grid(a.n_po);
    if (c.has_imu) {
      a.n_imu = p->imu->n; a.imu_res = p->imu->res.p; a.imu_i = p->imu->idx_a.p; a.imu_j = p->imu->idx_b.p;
      for (int k = 0; k < 8; ++k) a.imu_J.j[k] = p->imu->jac[k].p;
      a.imu = ImuEvalArgs{p->imu->n, p->imu->pre.p, p->imu->sqrt_info.p, p->imu->idx_a.p, p->imu->idx_b.p};
    }
    static const int staged_on = [] { const char* e = std::getenv("LVF_STAGED"); return (e && e[0] == '0') ? 0 : 1; }();
    a.cp = TfCompact{0, nullptr, nullptr, nullptr, nullptr, staged_on};
    if (p->compact) {
      a.cp = TfCompact{1, p->tf_slot.p, p->slotB.p, p->slabP.p, p->slabQ.p, staged_on};
      TfReduceArgs& r = c.red;
      r.n_kf = p->n_kf; r.n_wg = a.n_tfw; r.run_first = p->run_first.p; r.slabP = p->slabP.p; r.slabQ = p->slabQ.p; r.B = p->B.p; r.ld = p->dpad; r.gc = p->gc.p;
      r.nblocks = p->n_kf * ((a.n_tfw + 63) / 64) + grid(p->n_kf * (p->n_kf - 1) / 2 * 36); r.done = done;
      r.own_blocks = r.nblocks; r.ride = SpArgs{}; r.ride.nblocks = 0;
    }
    c.lin.n_kf = p->n_kf; c.lin.s = s; c.lin.huber = 0.0; c.lin.pose_const = p->pose_const.p; c.lin.B = p->B.p; c.lin.ld = p->dpad; c.lin.gc = p->gc.p; c.lin.E = p->E.p;
    c.lin.ldE = p->ldE; c.lin.C = p->C.p; c.lin.gr = p->gr.p; c.lin.cost = cost; c.lin.done = done; c.lin.dbg = nullptr;
    c.lin.scal_reset = c.early ? p->scal.p : nullptr;
    c.lin.nblocks = a.n_tfw + a.g_tc + a.g_po + (a.imu.pre ? a.n_imu : (a.n_imu + 3) / 4);
    c.lin_lds = std::max((size_t)(sizeof(PoseD) / 8 + kAccSlots) * p->n_kf + 32 + 4 * (size_t)kStageWave, (size_t)std::max(kImuWaveLds, 1864 + 64)) * sizeof(double);
  }
  // damped system
  {
    PrepArgs& a = c.prep;
    a.ld = p->ld; a.dpad = p->dpad; a.iperm = p->iperm.p; a.B = p->B.p; a.gc = p->gc.p; a.radius = radius; a.S = p->S.p; a.jac = jac; a.jl0 = p->d;
    // (the lower triangle, folded: prepare_body)
    a.nS_blocks = (unsigned)(((size_t)((p->ld + 1) / 2) * (p->ld + 1) + kT - 1) / kT); a.n_lm = p->n_lm; a.dp = p->dp; a.ldE = p->ldE; a.C = p->C.p; a.gr = p->gr.p; a.Cd = p->Cd.p; a.E = p->E.p;
    a.eoff = p->lm_eoff.p; a.kmin = p->lm_kmin.p; a.kmax = p->lm_kmax.p; a.slotB = p->compact ? p->slotB.p : nullptr; a.Ct = p->Ct.p; a.grt = p->grt.p;
    a.scal = p->scal.p; a.nblocks = (int)a.nS_blocks + (p->n_lm ? grid(p->compact ? 8 * p->n_lm : p->n_lm) : 0); a.done = done;
    a.early = 0; a.off = p->off; a.own_blocks = a.nblocks; a.ride = SpArgs{}; a.ride.nblocks = 0;
    PrepArgs& e = c.prep_early;
    e = a;
    const int nn = p->ld - p->off;
    e.early = 1; e.scal = nullptr;                      // (the scalars are reset by the linearisation launch: LinArgs::scal_reset)
    e.nS_blocks = (unsigned)(((size_t)((nn + 1) / 2) * (nn + 1) + kT - 1) / kT);
    e.nblocks = e.own_blocks = (int)e.nS_blocks + (p->n_lm ? grid(p->compact ? 8 * p->n_lm : p->n_lm) : 0);
  }
  int* fail = reinterpret_cast<int*>(p->scal.p + SC_FAIL);
  c.n_levels = p->sp_levels.n;
  for (int lv = 0; lv < p->sp_levels.n; ++lv) {
    SpArgs& a = c.sp[lv];
    a.nodes = p->sp_nodes.p; a.first = p->sp_levels.first[lv]; a.tiles = p->sp_tiles[lv]; a.rows = p->sp_rows.p; a.S = p->S.p; a.ld = p->ld; a.W = p->sp_W.p;
    a.wstride = p->sp_wstride; a.Lout = p->sp_L.p; a.fail = fail; a.nblocks = p->sp_levels.count[lv] * p->sp_tiles[lv]; a.done = done;
    // (0.5 ms at 100 MHz before a chained level gives up on the level below — a hand-over normally takes microseconds, and a retry costs one iteration of 0.2 ms: round 4 waited 2 ms, ten iterations of latency on a shared GPU; LVF_CHAIN_TIMEOUT_US overrides; LVF_CHAIN_FENCE=0: relaxed hand-over, A/B only)
    static const unsigned chain_timeout = [] { const char* e = std::getenv("LVF_CHAIN_TIMEOUT_US"); return e ? (unsigned)std::max(1, std::atoi(e)) * 100u : 50000u; }();
    static const int chain_fenced = [] { const char* e = std::getenv("LVF_CHAIN_FENCE"); return (e && e[0] == '0') ? 0 : 1; }();
    a.src = c.early ? SpSrc{p->B.p, p->dpad, p->dp, p->gc.p, radius, p->sp_rows_nat.p, nullptr, 0, nullptr, chain_fenced, chain_timeout, p->off, std::getenv("LVF_CHAIN_RMW_READ") ? 1 : 0, nullptr, lv == 0 ? 1 : 0}
                    : SpSrc{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, chain_fenced, chain_timeout, p->off, 0, nullptr, 0};
    a.src.jac = jac;
    c.sp_lds[lv] = p->sp_shmem[lv];
  }
  c.merged_level0 = false;
  if (p->n_lm) {
    const int nt = p->ldE / 16, ntile = nt * (nt + 1) / 2;
    const size_t shb = schur_lds;
    // early form: the levels are dealt to the launches that exist anyway, in order
    int next_level = 0;
    // Where the levels go (measured, MI355X): the Schur launch hides three (one riding, two chained behind it); a level riding in
    // k_tf_reduce's launch costs ~0 us, in k_prepare's ~2 us, a launch of its own 7.6 us.  So the LAST three levels go to the Schur launch
    // and only what is left over rides in the two launches ahead (16 / 20 keyframes, three levels: 0.111 / 0.125 -> 0.098 / 0.118 ms
    // with the chain alone, 0.103 / 0.123 with riders in front of it).  LVF_RIDE_TF / LVF_RIDE_PREP = 0 | 1 override, LVF_CHAIN_LEVELS = 0..2.
    static const int ride_tf_env = [] { const char* e = std::getenv("LVF_RIDE_TF"); return e ? std::atoi(e) : -1; }();
    static const int ride_prep_env = [] { const char* e = std::getenv("LVF_RIDE_PREP"); return e ? std::atoi(e) : -1; }();
    static const int chain_n_env = [] { const char* e = std::getenv("LVF_CHAIN_LEVELS"); return e ? std::max(0, std::min(2, std::atoi(e))) : 2; }();
    const int chain_n = p->no_chain ? 0 : chain_n_env;      // (a hand-over that timed out once: every level in a launch of its own from then on)
    const int excess = std::max(0, c.n_levels - (1 + chain_n));
    const bool ride_tf = ride_tf_env >= 0 ? ride_tf_env != 0 : (p->compact && excess >= 1);
    const bool ride_prep = ride_prep_env >= 0 ? ride_prep_env != 0 : (excess >= 2 || (excess >= 1 && !(ride_tf && p->compact)));
    if (c.early) {
      if (!ride_tf) {}
      else if (p->compact && next_level < c.n_levels) { c.red.ride = c.sp[next_level]; c.red.nblocks = c.red.own_blocks + c.red.ride.nblocks; c.red_lds = (size_t)c.sp_lds[next_level]; ++next_level; }
      if (ride_prep && next_level < c.n_levels) { PrepArgs& e = c.prep_early; e.ride = c.sp[next_level]; e.nblocks = e.own_blocks + e.ride.nblocks; c.prep_lds = (size_t)c.sp_lds[next_level]; ++next_level; }
    }
    if (schur_merged) {
      SchurSp0Args& a = c.ssp0;
      a.rows = band_rows_clamped(p->band_rows);
      a.n_slices = (p->n_lm + a.rows - 1) / a.rows; a.n_groups = (ntile + kBandTilesPerGroup - 1) / kBandTilesPerGroup;
      a.dp = p->dp; a.ldE = p->ldE; a.E = p->E.p; a.Cd = p->Cd.p; a.order = p->lm_order.p;
      a.dbg = nullptr; a.n_active = p->lm_nactive.p; a.kmin = p->lm_kmin.p; a.kmax = p->lm_kmax.p;
      a.d_local = p->dp; a.ldS = p->ld; a.S_pose = p->S.p + (size_t)p->off_pose * (p->ld + 1);
      const int ride = c.early ? next_level : 0;       // classic: level 0 rides here
      a.sp = SpArgs{}; a.sp.nblocks = 0; a.sp_b = a.sp; a.sp_c = a.sp;
      c.ssp0_lds = shb;
      next_level = ride;
      if (ride < c.n_levels) { a.sp = c.sp[ride]; c.ssp0_lds = std::max(c.ssp0_lds, (size_t)p->sp_shmem[ride]); next_level = ride + 1; }
      // the Schur complement lasts ~20 us at this size, a level ~5: the next two levels wait for their predecessor INSIDE the launch
      if (c.early && chain_n > 0) {
        int* cnt = reinterpret_cast<int*>(p->sp_sync.p);
        SpArgs* slot[2] = {&a.sp_b, &a.sp_c};
        SpArgs* prev = &a.sp;
        for (int k = 0; k < std::min(2, chain_n) && next_level < c.n_levels && (size_t)p->sp_shmem[next_level] <= 64 * 1024; ++k) {
          *slot[k] = c.sp[next_level];
          prev->src.done_counter = cnt + 2 * (next_level - 1);
          slot[k]->src.wait_counter = cnt + 2 * (next_level - 1); slot[k]->src.wait_target = prev->nblocks;
          if (p->force_handover_timeouts > 0 && k == 0) { slot[k]->src.wait_target = prev->nblocks + 1; slot[k]->src.timeout_ticks = 2000u; }      // test hook: a producer that never arrives (20 us)
          c.ssp0_lds = std::max(c.ssp0_lds, (size_t)p->sp_shmem[next_level]);
          prev = slot[k];
          ++next_level;
        }
      }
      a.work = p->band_work.p; a.n_work = p->n_band_work;
      a.nblocks = a.n_work + a.sp.nblocks + a.sp_b.nblocks + a.sp_c.nblocks; a.done = done;
      c.merged_level0 = true;
      c.first_own_level = next_level;
    }
  }
  c.chol.Sd = p->S.p + (size_t)p->off * (p->ld + 1); c.chol.ld = p->ld; c.chol.nb = p->nb; c.chol.fail = fail; c.chol.Dinv = p->Dinv.p; c.chol.Ldiag = p->Ldiag.p; c.chol.done = done;
  c.chol.last_cols = p->ndense + 1 - kNB * (p->nb - 1);       // (the right-hand-side row is the last real one)
  fill_back_args(p, c.back, &c.back_lds);
  c.back.done = done;
  {
    TailArgs& a = c.tail;
    // (640 workgroups take the 10 000 landmarks of the BASELINE window in one pass of 16 per workgroup; measured 1-2 % of an iteration over a cap of 256)
    static const int tail_cap = [] { const char* e = std::getenv("LVF_TAIL_WGS"); return e ? std::atoi(e) : 640; }();
    a.g_lm = p->n_lm ? std::min(tail_cap, (p->n_lm + kT / 16 - 1) / (kT / 16)) : 0;
    a.n_lm = p->n_lm; a.dp = p->dp; a.ldE = p->ldE; a.E = p->E.p; a.C = p->compact ? p->Ct.p : p->C.p; a.Cd = p->Cd.p;
    a.gr = p->compact ? p->grt.p : p->gr.p; a.dxc = p->dxc.p; a.dxl = p->dxl.p; a.scal = p->scal.p;
    a.kmin = p->band_ready ? p->lm_kmin.p : nullptr; a.kmax = p->lm_kmax.p; a.n_kf = p->n_kf; a.s = s; a.poses2 = p->poses2.p; a.vel2 = p->vel2.p; a.ba2 = p->ba2.p;
    a.bg2 = p->bg2.p; a.invd2 = p->invd2.p; a.d = p->d; a.ld = p->dpad; a.B = p->B.p; a.gc = p->gc.p; a.radius = radius; a.nblocks = a.g_lm + grid(p->d); a.done = done; a.pose_const = p->pose_const.p; a.jac = jac;
    c.tail_lds = (size_t)p->ldE * sizeof(double);
  }
  {
    static const unsigned bt_timeout = [] { const char* e = std::getenv("LVF_CHAIN_TIMEOUT_US"); return e ? (unsigned)std::max(1, std::atoi(e)) * 100u : 50000u; }();
    static const int bt_fenced = [] { const char* e = std::getenv("LVF_CHAIN_FENCE"); return e ? std::atoi(e) : 1; }();
    static const bool bt_big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(k_backsolve_tail), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) == hipSuccess;
    c.back_tail_merged = bt_wanted && c.tail.g_lm > 0;
    if (c.back_tail_merged) {
      BackTailArgs& m = c.bt;
      m.back = c.back; m.tail = c.tail;
      m.back.pose_ready = reinterpret_cast<int*>(p->sp_sync.p) + 2 * kSpMaxLevels - 2;      // (the last int pair of the arrival-counter block: the levels use pairs 0 .. n_levels - 2)
      m.back.n_pose = p->dp;
      // (one round of workgroups: the launch's register and LDS footprint is the back substitution's, so about one workgroup fits a CU, and a
      // landmark workgroup that has to wait for a CU starts after the others are done)
      static const int bt_wgs = [] { const char* e = std::getenv("LVF_BACK_TAIL_WGS"); return e ? std::max(1, std::atoi(e)) : 224; }();
      m.g_lm = std::min(bt_wgs, (p->n_lm + kBT / 16 - 1) / (kBT / 16));
      m.fenced = bt_fenced == 2; m.back.pose_fenced = bt_fenced == 2; m.timeout_ticks = bt_timeout; m.fail = fail;
      if (p->force_handover_timeouts > 1) { m.back.pose_ready = reinterpret_cast<int*>(p->sp_sync.p) + 2 * kSpMaxLevels - 4; m.timeout_ticks = 2000u; }      // test hook (n >= 2): a flag nobody raises
      c.bt_lds = std::max(c.back_lds, c.tail_lds);
      if (c.bt_lds > 64 * 1024 && !bt_big_lds) c.back_tail_merged = false;
    }
    static const bool chain_info = std::getenv("LVF_CHAIN_INFO") != nullptr;
    if (chain_info) std::fprintf(stderr, "chain: n_kf %d n_lm %d fast %d has_imu %d early %d compact %d levels %d no_chain %d merged_level0 %d back_tail_merged %d (g_lm %d, lds %zu)\n", p->n_kf, p->n_lm, (int)c.fast, (int)c.has_imu,
                                 (int)c.early, (int)p->compact, c.n_levels, (int)p->no_chain, (int)c.merged_level0, (int)c.back_tail_merged, c.bt.g_lm, c.bt_lds);
  }
  fill_cost_visual(p, c.cost.a);
  c.cost.n_kf = p->n_kf; c.cost.s = s2; c.cost.huber = 0.0; c.cost.cost = p->scal.p + SC_COST_NEW; c.cost.done = done;
  c.cost.nblocks = c.cost.a.g_tc + c.cost.a.g_tf + grid(c.cost.a.n_po);
  {
    // two tiles of kT blocks per workgroup: half as many workgroups to dispatch ahead of the decision (measured -1 % of an iteration; three: same)
    static const int cost_tiles = [] { const char* e = std::getenv("LVF_COST_TILES"); return e ? std::atoi(e) : 2; }();
    if (cost_tiles > 1) {
      CostArgs& k = c.cost;
      const int per = kT * cost_tiles;
      k.tiles = cost_tiles;
      k.a.g_tc = (k.a.n_tc + per - 1) / per; k.a.g_tf = (k.a.n_tf + per - 1) / per;
      k.nblocks = k.a.g_tc + k.a.g_tf + (k.a.n_po + per - 1) / per;
    }
  }
  c.cost.g_imu = 0; c.cost.imu = ImuEvalArgs{};
  c.imu_in_cost = c.fast && c.has_imu && c.cost.nblocks > 0;         // the IMU cost rides in the merged cost + decision launch
  if (c.imu_in_cost) { c.cost.imu = ImuEvalArgs{p->imu->n, p->imu->pre.p, p->imu->sqrt_info.p, p->imu->idx_a.p, p->imu->idx_b.p}; c.cost.g_imu = p->imu->n; c.cost.nblocks += p->imu->n; }
  c.cost.zero = c.zero_end; c.cost.zero_wgs = c.fast ? kEndZeroWgs : 0;
  {
    DecideArgs& a = c.dec;
    a.scal = p->scal.p; a.ctl = ctl; a.rec = p->rec; a.ticket = reinterpret_cast<int*>(p->scal.p + SC_TICKET); a.n_kf = p->n_kf; a.n_lm = p->n_lm;
    a.poses = p->st->poses.p; a.vel = p->st->vel.p; a.ba = p->st->ba.p; a.bg = p->st->bg.p; a.invd = p->st->inv_depth.p;
    a.poses2 = p->poses2.p; a.vel2 = p->vel2.p; a.ba2 = p->ba2.p; a.bg2 = p->bg2.p; a.invd2 = p->invd2.p;
    static const bool lm_history = std::getenv("LVF_LM_HISTORY") != nullptr;
    a.hist = nullptr;
    if (lm_history) { LVF_TRY(p->dbg_hist.ensure(8 * 64)); a.hist = p->dbg_hist.p; }
  }
  c.batchable = c.fast && p->compact && c.has_imu && !c.has_prior && c.merged_level0 && c.lin.nblocks > 0 && c.cost.nblocks > 0;
  { const StateP sp = state_ptrs(p->st); std::memcpy(p->chain_state, &sp, sizeof(sp)); }
  p->chain_tcw = p->tc && p->tc->wblk.n ? p->tc->wblk.p : nullptr;
  p->chain_ready = true;
  return LVF_OK;
}
static bool chain_stale(const lvf_problem* p) {
  if (!p->chain_ready || !p->chain) return true;
  const StateP s = state_ptrs(p->st);
  static_assert(sizeof(StateP) == sizeof(p->chain_state), "StateP is six pointers");
  if (std::memcmp(&s, p->chain_state, sizeof(StateP)) != 0) return true;      // the state's buffers were re-allocated (window grew)
  // lvf_two_camera_set_block_weights after the problem was created: the argument blocks hold the old weight pointer (or none)
  const double* w = p->tc && p->tc->wblk.n ? p->tc->wblk.p : nullptr;
  return w != p->chain_tcw;
}

// the linearisation at the current state: cost, B, gc, E, C, gr.  `gated`: skipped on device once the LM loop has finished
// HIP events between the stages of one LM iteration (lvf_problem_stage_times): event 0 before the first launch, event k + 1 after stage k
enum { ST_IMU_LIN = 0, ST_LIN_VISUAL, ST_TF_REDUCE, ST_PREPARE, ST_SCHUR_SP0, ST_SP_LEVELS, ST_CHOL, ST_BACKSOLVE, ST_STEP_TAIL, ST_COST, ST_DECIDE, ST_N };
static const char* const kStageNames[ST_N] = {"k_zero_multi (only when the accumulators are not known clean)", "k_lin_visual", "k_tf_reduce (+ sparse level 0)", "k_prepare (+ sparse level 1)", "k_schur_sp0 (+ a sparse level)", "k_sp_eliminate (the levels left)",
                                             "k_chol_step (all block steps)", "k_chol_backsolve", "k_step_tail", "k_cost_decide (candidate cost incl. the ImuError factors; its last workgroup closes the iteration; + prior passes)", "k_lm_decide (windows without visual blocks)"};
// (one event set per timed iteration: the iterations are enqueued back to back and waited for ONCE, so every stage — the first one of an
// iteration included — starts behind a busy queue like in the device loop; with a wait per iteration the first stage absorbed the idle
// queue's start-up, ~6 us of k_lin_visual's figure)
constexpr int kClockReps = 16, kClockLaunches = 48;
// Two sources per timed iteration: ev[] — events between the STAGES on the stream (a span: kernels + the gaps between them + the marker's own
// cost) — and kstart[] / kstop[] — a start / stop event pair recorded WITH every kernel launch of the fast chain (hipExtLaunchKernelGGL: the
// dispatch packet's own timestamps, i.e. what rocprofv3 --kernel-trace reports as the kernel's duration).  Stage times are the sums of the
// second kind, so bench.py's roofline entries follow profiles/ for short stages too (the event-pair subtraction left k_lin_visual 18 % high).
struct StageClock {
  hipEvent_t ev[kClockReps][ST_N + 1]; int launches[ST_N]; bool on = false; int rep = 0;
  hipEvent_t kstart[kClockReps][kClockLaunches], kstop[kClockReps][kClockLaunches]; int kstage[kClockLaunches]; int nk = 0; bool kernel_events = false;
};
void stage_clock_free(StageClock* k) {
  if (!k) return;
  for (auto& r : k->ev) for (auto& e : r) (void)hipEventDestroy(e);
  if (k->kernel_events) { for (auto& r : k->kstart) for (auto& e : r) (void)hipEventDestroy(e); for (auto& r : k->kstop) for (auto& e : r) (void)hipEventDestroy(e); }
  delete k;
}
// a launch of the iteration's fast chain: plain, or — while lvf_problem_stage_times is recording — with its own start / stop events
#define LVF_CHAIN_LAUNCH(p_, stage_, kernel_, grid_, block_, lds_, q_, ...)                                                          \
  do {                                                                                                                              \
    StageClock* k__ = (p_)->clk;                                                                                                    \
    if (k__ && k__->on && k__->kernel_events && k__->nk < kClockLaunches) {                                                         \
      const int i__ = k__->nk++;                                                                                                    \
      k__->kstage[i__] = (stage_);                                                                                                  \
      hipExtLaunchKernelGGL(kernel_, grid_, block_, lds_, q_, k__->kstart[k__->rep][i__], k__->kstop[k__->rep][i__], 0, __VA_ARGS__); \
    } else hipLaunchKernelGGL(kernel_, grid_, block_, lds_, q_, __VA_ARGS__);                                                       \
  } while (0)
static inline void stage_mark(lvf_problem* p, int stage_done, int launches) {
  StageClock* k = p->clk;
  if (!k || !k->on) return;
  (void)hipEventRecord(k->ev[k->rep][stage_done + 1], p->ctx->stream);
  k->launches[stage_done] = launches;
}

// `iteration`: the launches belong to a full LM iteration (enqueue_iteration) — only then do the early sparse levels ride along and are the
// per-step scalars reset here; a stand-alone linearisation (gradient / cost taps) leaves S and the control block alone
static int enqueue_linearize(lvf_problem* p, double huber, bool gated, bool iteration = false) {
  hipStream_t q = p->ctx->stream;
  if (chain_stale(p)) LVF_TRY(build_chain(p));
  const Chain& c = *p->chain;
  const StateP s = state_ptrs(p->st);
  double* cost = p->scal.p + SC_COST;
  bool imu_done = false;
  // the accumulators are cleared at the END of every iteration (extra workgroups of the cost + decision launch); a launch of its own is
  // only needed when they are not known to be clean (first linearisation after a configure, stand-alone gradient / reduced-system taps)
  const bool clean = c.fast && p->accum_clean;
  p->accum_clean = false;
  if (p->clk && p->clk->on) (void)hipEventRecord(p->clk->ev[p->clk->rep][0], q);
  if (!clean) LVF_CHAIN_LAUNCH(p, ST_IMU_LIN, k_zero_multi, dim3(512, c.zero.count), dim3(kT), 0, q, c.zero);
  if (c.fast) {
    imu_done = true;                           // the ImuError factors are evaluated inside the merged launch below
    stage_mark(p, ST_IMU_LIN, clean ? 0 : 1);
    LinArgs la = c.lin;
    la.huber = huber;
    if (!gated) la.done = nullptr;
    if (!iteration) la.scal_reset = nullptr;
    static const bool lin_timing = std::getenv("LVF_LIN_TIMING") != nullptr;
    if (lin_timing) { LVF_TRY(p->dbg_lin.ensure((size_t)la.v.n_tfw * 24 + 8)); la.dbg = p->dbg_lin.p; }
    static const size_t lds_pad = [] { const char* e = std::getenv("LVF_LIN_LDS_PAD"); return e ? (size_t)std::atoi(e) : (size_t)0; }();      // experiment: fewer workgroups per CU
    LVF_CHAIN_LAUNCH(p, ST_LIN_VISUAL, k_lin_visual, dim3(la.nblocks), dim3(kT), c.lin_lds + lds_pad, q, la);
    stage_mark(p, ST_LIN_VISUAL, 1);
    if (p->compact) {
      TfReduceArgs ra = c.red;
      if (!gated) ra.done = nullptr;
      if (!iteration) { ra.nblocks = ra.own_blocks; ra.ride.nblocks = 0; }
      LVF_CHAIN_LAUNCH(p, ST_TF_REDUCE, k_tf_reduce, dim3(ra.nblocks), dim3(kT), ra.ride.nblocks > 0 ? c.red_lds : 0, q, ra);
    }
    stage_mark(p, ST_TF_REDUCE, p->compact ? 1 : 0);
    if (lin_timing) {
      std::vector<unsigned long long> t((size_t)la.v.n_tfw * 8);
      LVF_HIP(hipStreamSynchronize(q));
      LVF_HIP(hipMemcpy(t.data(), p->dbg_lin.p, t.size() * 8, hipMemcpyDeviceToHost));
      double ph[5] = {0, 0, 0, 0, 0}; unsigned long long first = ~0ull, last = 0, last_start = 0;
      for (int w = 0; w < la.v.n_tfw; ++w) {
        for (int k = 0; k < 5; ++k) ph[k] += (double)(t[(size_t)w * 8 + k + 1] - t[(size_t)w * 8 + k]) * 0.01;
        first = std::min(first, t[(size_t)w * 8]); last = std::max(last, t[(size_t)w * 8 + 5]); last_start = std::max(last_start, t[(size_t)w * 8]);
      }
      {
        double mx[5] = {0, 0, 0, 0, 0}; int slow = 0; double slow_t = 0;
        for (int w = 0; w < la.v.n_tfw; ++w) {
          for (int k = 0; k < 5; ++k) mx[k] = std::max(mx[k], (double)(t[(size_t)w * 8 + k + 1] - t[(size_t)w * 8 + k]) * 0.01);
          const double tot = (double)(t[(size_t)w * 8 + 5] - t[(size_t)w * 8]) * 0.01;
          if (tot > slow_t) { slow_t = tot; slow = w; }
        }
        std::vector<TfWork> hw((size_t)la.v.n_tfw);
        LVF_HIP(hipMemcpy(hw.data(), la.v.work, hw.size() * sizeof(TfWork), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "lin_tf: per-phase MAX over the workgroups (us): %.2f | %.2f | %.2f | %.2f | %.2f ; slowest workgroup %d (k2 = %d, %d blocks): %.2f us =", mx[0], mx[1], mx[2], mx[3], mx[4], slow, hw[slow].k2, hw[slow].count, slow_t);
        for (int k = 0; k < 5; ++k) std::fprintf(stderr, " %.2f", (double)(t[(size_t)slow * 8 + k + 1] - t[(size_t)slow * 8 + k]) * 0.01);
        // histogram of workgroup durations by current keyframe decile
        {
          unsigned long long u[16];
          LVF_HIP(hipMemcpy(u, p->dbg_lin.p + (size_t)la.v.n_tfw * 8 + 8 + (size_t)slow * 16, sizeof(u), hipMemcpyDeviceToHost));
          std::fprintf(stderr, " ; its waves (eval us, k1-sum us, groups):");
          for (int wv = 0; wv < 4; ++wv) std::fprintf(stderr, " [%.2f %.2f %llu]", (double)(u[4 * wv + 1] - u[4 * wv]) * 0.01, (double)(u[4 * wv + 2] - u[4 * wv + 1]) * 0.01, u[4 * wv + 3]);
        }
        std::fprintf(stderr, " ; mean duration by k2 decile:");
        const int nk = la.n_kf;
        for (int dcl = 0; dcl < 5; ++dcl) {
          double sum = 0; int cnt = 0;
          for (int w = 0; w < la.v.n_tfw; ++w) if (hw[w].k2 * 5 / std::max(nk, 1) == dcl) { sum += (double)(t[(size_t)w * 8 + 5] - t[(size_t)w * 8]) * 0.01; ++cnt; }
          std::fprintf(stderr, " %.1f(%d)", cnt ? sum / cnt : 0.0, cnt);
        }
        std::fprintf(stderr, "\n");
      }
      std::fprintf(stderr, "lin_tf: the TwoFrame workgroups START within %.2f us of each other; starts of workgroups 0, 1/4, 1/2, 3/4, last (us after the first): %.2f %.2f %.2f %.2f %.2f\n", (double)(last_start - first) * 0.01,
                   (double)(t[0] - first) * 0.01, (double)(t[(size_t)(la.v.n_tfw / 4) * 8] - first) * 0.01, (double)(t[(size_t)(la.v.n_tfw / 2) * 8] - first) * 0.01,
                   (double)(t[(size_t)(3 * la.v.n_tfw / 4) * 8] - first) * 0.01, (double)(t[(size_t)(la.v.n_tfw - 1) * 8] - first) * 0.01);
      std::fprintf(stderr, "lin_tf phases (us, mean over %d workgroups): stage %.2f | eval+landmark atomics %.2f | k1 sums %.2f | k2 sums %.2f | flush %.2f ; first start -> last end %.2f\n",
                   la.v.n_tfw, ph[0] / la.v.n_tfw, ph[1] / la.v.n_tfw, ph[2] / la.v.n_tfw, ph[3] / la.v.n_tfw, ph[4] / la.v.n_tfw, (double)(last - first) * 0.01);
      if (la.v.imu.pre) {
        unsigned long long u[5];
        LVF_HIP(hipMemcpy(u, p->dbg_lin.p + (size_t)la.v.n_tfw * 8, sizeof(u), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "lin_imu phases of factor 0 (us): stage %.2f | raw residual + pre-weighting Jacobian (one lane) %.2f | weight + to tangent %.2f | J^T J, J^T r %.2f ; start %.2f after the first TwoFrame workgroup, end %.2f before the last one's end\n",
                     (double)(u[1] - u[0]) * 0.01, (double)(u[2] - u[1]) * 0.01, (double)(u[3] - u[2]) * 0.01, (double)(u[4] - u[3]) * 0.01, ((double)u[0] - (double)first) * 0.01, ((double)last - (double)u[4]) * 0.01);
      }
    }
  } else {
    if (p->tc && p->tc->n)
      hipLaunchKernelGGL(k_lin_tc<false>, dim3(grid(p->tc->n)), dim3(kT), 0, q, p->tc->n, (const double2*)p->tc->ob_a.p, (const double2*)p->tc->ob_b.p,
                         p->tc->idx_a.p, p->tc->idx_b.p, p->tc->wblk.n ? p->tc->wblk.p : (const double*)nullptr, s, p->tc->cam_a, p->tc->cam_b, huber, p->C.p, p->gr.p, cost);
    if (p->tf && p->tf->n)
      hipLaunchKernelGGL(k_lin_tf<false>, dim3(grid(p->tf->n)), dim3(kT), 0, q, p->tf->n, p->n_kf, (const double2*)p->tf->ob_a.p, (const double2*)p->tf->ob_b.p,
                         p->tf->idx_a.p, p->tf->idx_b.p, p->tf->idx_c.p, s, p->tf->cam_a, p->tf->cam_b, huber, p->pose_const.p, p->B.p, p->dpad,
                         p->gc.p, p->E.p, p->ldE, p->C.p, p->gr.p, cost);
    if (p->po && p->po->n)
      hipLaunchKernelGGL(k_lin_po<false>, dim3(grid(p->po->n)), dim3(kT), 0, q, p->po->n, p->n_kf, (const double2*)p->po->ob_a.p, p->po->idx_a.p,
                         p->po->idx_b.p, p->po->table.p, s, p->po->cam_a, huber, p->pose_const.p, p->B.p, p->dpad, p->gc.p, cost);
  }
  if (c.has_imu && !imu_done) {
    LVF_TRY(launch_imu(p->imu, p->st, true));
    ImuJ J;
    for (int k = 0; k < 8; ++k) J.j[k] = p->imu->jac[k].p;
    hipLaunchKernelGGL(k_lin_imu, dim3(p->imu->n), dim3(64), 0, q, p->imu->n, p->n_kf, p->imu->res.p, J, p->imu->idx_a.p, p->imu->idx_b.p,
                       p->st->poses.p, p->pose_const.p, p->B.p, p->dpad, p->gc.p, cost);
  }
  if (c.has_prior) {
    const lvf_batch* pb = p->prior;
    const PriorArgs P{pb->n, pb->idx_a.p, pb->idx_b.p, pb->table.p, pb->ob_a.p, pb->ob_b.p};
    hipLaunchKernelGGL(k_prior_lin, dim3((pb->n + 63) / 64), dim3(64), 0, q, P, p->st->poses.p, p->pose_const.p, p->B.p, p->dpad, p->gc.p, cost);
  }
  LVF_HIP(hipGetLastError());
  p->linearized = true;
  return LVF_OK;
}

// the deferred item count of the band work list (build_band_work) -> the Schur launch's arguments
static int await_band_work(lvf_problem* p) {
  if (!p->band_pending) return LVF_OK;
  LVF_HIP(hipEventSynchronize(p->ev_band));
  p->band_pending = false;
  p->n_band_work = p->h_n_band_work[0];
  if (p->chain && p->chain->merged_level0) {
    SchurSp0Args& a = p->chain->ssp0;
    a.n_work = p->n_band_work;
    a.nblocks = a.n_work + a.sp.nblocks + a.sp_b.nblocks + a.sp_c.nblocks;
  }
  return LVF_OK;
}

// S (elimination order) = B + D - E^T Cd^-1 E, rhs row = -(gc - E^T Cd^-1 g_rho); radius read from `radius_dev`
static int enqueue_reduced_system(lvf_problem* p, const double* radius_dev, bool reset_scalars, bool gated, bool* level0_done) {
  hipStream_t q = p->ctx->stream;
  const Chain& c = *p->chain;
  // the parity tap (level0_done == nullptr: the damped system alone, nothing eliminated) always takes the classic assembly
  const bool early = c.early && level0_done != nullptr;
  PrepArgs pa = early ? c.prep_early : c.prep;
  pa.radius = radius_dev;
  if (!reset_scalars) pa.scal = nullptr;
  if (!gated) pa.done = nullptr;
  // LVF_POISON_S=1 (diagnostic): every byte of S is 0xff (NaN) before the assembly, so whatever the assembly does not write — the upper
  // triangle — stays NaN; results must not change (tests/test_gpu_solver.py runs the parity cases this way too)
  static const bool poison = std::getenv("LVF_POISON_S") != nullptr;
  if (poison) LVF_HIP(hipMemsetAsync(p->S.p, 0xff, (size_t)p->ld * p->ld * 8, q));
  LVF_CHAIN_LAUNCH(p, ST_PREPARE, k_prepare, dim3(pa.nblocks), dim3(kT), pa.ride.nblocks > 0 ? c.prep_lds : 0, q, pa);
  stage_mark(p, ST_PREPARE, 1);
  if (!early && c.early) p->accum_clean = false;       // the tap wrote S: the next iteration must clear it
  if (level0_done) *level0_done = false;
  if (p->n_lm) {
    LVF_TRY(await_band_work(p));                       // (patches c.ssp0: `c` refers to the problem's chain)
    if (c.merged_level0) {
      SchurSp0Args sa = c.ssp0;
      if (!gated) sa.done = nullptr;
      if (!level0_done) { sa.nblocks = sa.n_work; sa.sp.nblocks = 0; sa.sp_b.nblocks = 0; sa.sp_c.nblocks = 0; }       // the Schur complement alone (parity tap)
      static const bool schur_timing = std::getenv("LVF_SCHUR_TIMING") != nullptr;
      const int ns = sa.n_work;
      if (schur_timing) { LVF_TRY(p->dbg_lin.ensure((size_t)ns * 8 + 8)); LVF_HIP(hipMemsetAsync(p->dbg_lin.p, 0, (size_t)ns * 64, q)); sa.dbg = p->dbg_lin.p; }
      if (sa.nblocks > 0) LVF_CHAIN_LAUNCH(p, ST_SCHUR_SP0, k_schur_sp0, dim3(sa.nblocks), dim3(256), c.ssp0_lds, q, sa);
      stage_mark(p, ST_SCHUR_SP0, 1);
      if (schur_timing) {
        std::vector<unsigned long long> t((size_t)ns * 8);
        LVF_HIP(hipStreamSynchronize(q));
        LVF_HIP(hipMemcpy(t.data(), p->dbg_lin.p, t.size() * 8, hipMemcpyDeviceToHost));
        double ph[3] = {0, 0, 0}; int cnt = 0; unsigned long long first = ~0ull, last = 0;
        for (int w = 0; w < ns; ++w) {
          if (!t[(size_t)w * 8 + 3]) continue;
          ++cnt;
          for (int k = 0; k < 3; ++k) ph[k] += (double)(t[(size_t)w * 8 + k + 1] - t[(size_t)w * 8 + k]) * 0.01;
          first = std::min(first, t[(size_t)w * 8]); last = std::max(last, t[(size_t)w * 8 + 3]);
        }
        std::fprintf(stderr, "schur band phases (us, mean over %d of %d workgroups): setup + first fetch issue %.2f | chunks (stage + mfma) %.2f | output atomics %.2f ; first start -> last end %.2f\n",
                     cnt, ns, ph[0] / std::max(cnt, 1), ph[1] / std::max(cnt, 1), ph[2] / std::max(cnt, 1), (double)(last - first) * 0.01);
      }
      if (level0_done) *level0_done = true;
    } else {
      double* S_pose = p->S.p + (size_t)p->off_pose * (p->ld + 1);
      const LmBand band{p->band_ready ? p->lm_order.p : nullptr, p->lm_nactive.p, p->lm_kmin.p, p->lm_kmax.p};
      LVF_TRY(launch_schur(q, p->n_lm, p->dp, p->ldE, p->E.p, p->Cd.p, p->dp, p->ld, S_pose, band));
    }
  }
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

// one complete LM iteration of one window on its stream, closed on device by k_lm_decide; nothing is waited for
static int enqueue_iteration(lvf_problem* p, bool end_zero) {
  hipStream_t q = p->ctx->stream;
  if (chain_stale(p)) LVF_TRY(build_chain(p));
  const Chain& c = *p->chain;
  static const bool sp_timing = std::getenv("LVF_SP_TIMING") != nullptr;
  if (sp_timing && c.early) {
    // diagnostic: the levels' stamps (the chain is rebuilt with the debug pointer in every level's arguments; printed by the next call)
    const int n_nodes = p->sp_levels.n ? p->sp_levels.first[p->sp_levels.n - 1] + p->sp_levels.count[p->sp_levels.n - 1] : 0;
    if (p->dbg_sp.n == 0) {
      LVF_TRY(p->dbg_sp.ensure((size_t)n_nodes * 8 + 8)); p->dbg_sp.n = (size_t)n_nodes * 8;
      LVF_HIP(hipMemsetAsync(p->dbg_sp.p, 0, (size_t)n_nodes * 64, q));
      Chain& cw = *p->chain;
      for (int lv = 0; lv < cw.n_levels; ++lv) cw.sp[lv].src.dbg = p->dbg_sp.p;
      cw.red.ride.src.dbg = p->dbg_sp.p; cw.prep_early.ride.src.dbg = p->dbg_sp.p; cw.ssp0.sp.src.dbg = p->dbg_sp.p; cw.ssp0.sp_b.src.dbg = p->dbg_sp.p; cw.ssp0.sp_c.src.dbg = p->dbg_sp.p;
    } else {
      std::vector<unsigned long long> t((size_t)n_nodes * 8);
      LVF_HIP(hipStreamSynchronize(q));
      LVF_HIP(hipMemcpy(t.data(), p->dbg_sp.p, t.size() * 8, hipMemcpyDeviceToHost));
      for (int lv = 0; lv < p->sp_levels.n; ++lv) {
        const int f0 = p->sp_levels.first[lv], cnt = p->sp_levels.count[lv];
        double ph[7] = {0, 0, 0, 0, 0, 0, 0}; unsigned long long first = ~0ull, last = 0;
        for (int k = f0; k < f0 + cnt; ++k) {
          for (int j = 0; j < 7; ++j) ph[j] += (double)(t[(size_t)k * 8 + j + 1] - t[(size_t)k * 8 + j]) * 0.01 / cnt;
          first = std::min(first, t[(size_t)k * 8]); last = std::max(last, t[(size_t)k * 8 + 7]);
        }
        std::fprintf(stderr, "sparse level %d (%d blocks, us): requests by address %.2f | wait for the level below %.2f | S entries %.2f | factor %.2f | W + first adds (returning) %.2f | arrive %.2f | rest of the adds %.2f ; start %.2f after level 0's first start, span %.2f\n",
                     lv, cnt, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], (double)(first - t[0]) * 0.01, (double)(last - first) * 0.01);
      }
    }
  }
  LVF_TRY(enqueue_linearize(p, p->huber, true, true));
  bool level0_done = false;
  LVF_TRY(enqueue_reduced_system(p, &p->ctl.p->radius, true, true, &level0_done));
  const int own0 = level0_done ? c.first_own_level : 0;      // (levels below rode in the launches above)
  for (int lv = own0; lv < c.n_levels; ++lv)
    LVF_CHAIN_LAUNCH(p, ST_SP_LEVELS, k_sp_eliminate, dim3(c.sp[lv].nblocks), dim3(256), c.sp_lds[lv], q, c.sp[lv]);
  stage_mark(p, ST_SP_LEVELS, std::max(0, c.n_levels - own0));
  for (int kb = 0; kb < p->nb; ++kb) {
    CholArgs cha = c.chol;
    static const bool chol_timing = std::getenv("LVF_CHOL_TIMING") != nullptr;
    if (chol_timing) { LVF_TRY(p->dbg.ensure(64)); cha.dbg = p->dbg.p; }
    LVF_CHAIN_LAUNCH(p, ST_CHOL, k_chol_step, dim3(chol_step_grid(p->nb, kb)), dim3(kCT), 0, q, cha, kb);
  }
  {
    BackArgs ba = c.back;
    static const bool back_timing = std::getenv("LVF_BACK_TIMING") != nullptr;
    if (back_timing) { LVF_TRY(p->dbg.ensure(64)); ba.sp.dbg = p->dbg.p; }
    stage_mark(p, ST_CHOL, p->nb);
    if (c.back_tail_merged) {
      BackTailArgs bt = c.bt;
      if (back_timing) bt.back.sp.dbg = p->dbg.p;
      LVF_CHAIN_LAUNCH(p, ST_BACKSOLVE, k_backsolve_tail, dim3(2 + c.bt.g_lm), dim3(kBT), c.bt_lds, q, bt);
      stage_mark(p, ST_BACKSOLVE, 1);
      stage_mark(p, ST_STEP_TAIL, 0);
    } else {
      LVF_CHAIN_LAUNCH(p, ST_BACKSOLVE, k_chol_backsolve, dim3(1), dim3(kBT), c.back_lds, q, ba);
      stage_mark(p, ST_BACKSOLVE, 1);
      LVF_CHAIN_LAUNCH(p, ST_STEP_TAIL, k_step_tail, dim3(c.tail.nblocks), dim3(kT), c.tail_lds, q, c.tail);
      stage_mark(p, ST_STEP_TAIL, 1);
    }
  }
  // candidate cost: the small passes first, then the visual pass whose last workgroup closes the iteration
  CostArgs ca = c.cost;
  ca.huber = p->huber;
  if (c.has_imu && !c.imu_in_cost) LVF_TRY(launch_imu_args(q, c.imu_cost, false));
  if (c.has_prior) {
    const lvf_batch* pb = p->prior;
    const PriorArgs P{pb->n, pb->idx_a.p, pb->idx_b.p, pb->table.p, pb->ob_a.p, pb->ob_b.p};
    hipLaunchKernelGGL(k_prior_cost, dim3((pb->n + 63) / 64), dim3(64), 0, q, P, p->poses2.p, p->scal.p + SC_COST_NEW);
  }
  if (ca.nblocks > 0) {
    if (!end_zero) ca.zero_wgs = 0;
    DecideArgs da = c.dec;
    static const bool cost_timing = std::getenv("LVF_COST_TIMING") != nullptr;
    if (cost_timing) { LVF_TRY(p->dbg.ensure(64)); da.dbg = p->dbg.p; }
    LVF_CHAIN_LAUNCH(p, ST_COST, k_cost_decide, dim3(ca.nblocks + ca.zero_wgs), dim3(kT), 0, q, ca, da, end_zero ? 1 : 0);
    p->accum_clean = ca.zero_wgs > 0;
    if (p->accum_clean) p->linearized = false;         // the normal equations of this iteration are gone: no reduced-system tap
    stage_mark(p, ST_COST, 1 + (c.has_imu && !c.imu_in_cost ? 1 : 0) + (c.has_prior ? 1 : 0));
  } else {
    stage_mark(p, ST_COST, (c.has_imu ? 1 : 0) + (c.has_prior ? 1 : 0));
    LVF_CHAIN_LAUNCH(p, ST_DECIDE, k_lm_decide, dim3(1), dim3(kDT), 0, q, c.dec);
    stage_mark(p, ST_DECIDE, 1);
  }
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

static void ctl_from_options(const lvf_solver_options* o, double radius, double decrease, int max_iters, bool with_tolerances, LmCtl* c) {
  std::memset(c, 0, sizeof(*c));
  c->radius = radius; c->decrease = decrease; c->last_radius = radius;
  c->huber = o->huber_a; c->min_rel_decrease = o->min_relative_decrease;
  c->function_tol = with_tolerances ? o->function_tolerance : -1.0;
  c->gradient_tol = with_tolerances ? o->gradient_tolerance : -1.0;
  c->parameter_tol = with_tolerances ? o->parameter_tolerance : -1.0;
  c->max_iters = max_iters;
  c->termination = 1; c->why = LVF_WHY_MAX_ITERATIONS;
}
static int upload_ctl(lvf_problem* p, const LmCtl& c) {
  if (chain_stale(p)) LVF_TRY(build_chain(p));
  *p->rec = c;                               // the host-visible mirror starts from the same values
  LVF_TRY(p->h_ctl.reserve(1));
  p->h_ctl[0] = c;
  LVF_HIP(hipMemcpyAsync(p->ctl.p, p->h_ctl.p, sizeof(LmCtl), hipMemcpyHostToDevice, p->ctx->stream));
  return LVF_OK;
}
static int download_ctl(lvf_problem* p, LmCtl* out) {
  hipStream_t q = p->ctx->stream;
  LVF_TRY(p->h_ctl.reserve(2));
  LVF_HIP(hipMemcpyAsync(&p->h_ctl[1], p->ctl.p, sizeof(LmCtl), hipMemcpyDeviceToHost, q));
  LVF_HIP(hipStreamSynchronize(q));
  *out = p->h_ctl[1];
  return LVF_OK;
}

// The loop ended because a chained sparse level timed out waiting for the level below (LVF_WHY_HANDOVER; the step was neither taken nor
// counted).  The problem gives up chaining for good (its levels become launches of their own: the LVF_CHAIN_LEVELS=0 form), the control
// block is re-armed as the aborted iteration found it and the caller enqueues again.  Returns false when there is nothing to retry.
static bool handover_pending(const lvf_problem* p, const LmCtl& c) { return c.done && c.why == LVF_WHY_HANDOVER && !p->no_chain; }
static int rearm_after_handover(lvf_problem* p, LmCtl* c) {
  p->no_chain = true; p->unchained_solves = 0; p->chain_ready = false; p->handover_retries += 1;
  if (p->force_handover_timeouts > 0) p->force_handover_timeouts -= 1;
  c->done = 0; c->termination = 1; c->why = LVF_WHY_MAX_ITERATIONS;
  p->accum_clean = false;                    // (the aborted iteration's partial sums: cleared by an explicit launch before the re-run)
  return upload_ctl(p, *c);                  // rebuilds the chain
}

struct IterOut { double cost_before, cost_after, model, dxnorm, xnorm, gmax; bool accepted, solved; };

// exactly one LM iteration from the current state (no tolerance tests): the per-iteration parity point
static int lm_iteration(lvf_problem* p, const lvf_solver_options* o, double* radius, double* decrease, IterOut* out) {
  LmCtl c;
  ctl_from_options(o, *radius, *decrease, 1, false, &c);
  p->huber = o->huber_a;
  LVF_TRY(upload_ctl(p, c));
  LVF_TRY(enqueue_iteration(p, false));
  LVF_TRY(download_ctl(p, &c));
  if (handover_pending(p, c)) {              // a chained hand-over timed out: the same iteration again, un-chained
    LVF_TRY(rearm_after_handover(p, &c));
    LVF_TRY(enqueue_iteration(p, false));
    LVF_TRY(download_ctl(p, &c));
  }
  if (p->dbg.p && std::getenv("LVF_CHOL_TIMING")) {
    unsigned long long t[64];
    LVF_HIP(hipMemcpy(t, p->dbg.p, sizeof(t), hipMemcpyDeviceToHost));
    for (int kb = 0; kb < p->nb && kb < 8; ++kb) {
      const unsigned long long* u = t + 8 * kb;
      if (kb == 0) std::fprintf(stderr, "chol step 0 (us): loads %.2f | factor %.2f (%llu shader clocks) | store %.2f\n", (double)(u[3] - u[0]) * 0.01, (double)(u[4] - u[3]) * 0.01, u[7] - u[6], (double)(u[5] - u[4]) * 0.01);
      else if (kb + 2 < p->nb + 1) std::fprintf(stderr, "chol step %d (us): stage %.2f | mfma %.2f | relayout %.2f | factor %.2f | store %.2f ; since previous step's end %.2f\n", kb, (double)(u[1] - u[0]) * 0.01,
                        (double)(u[2] - u[1]) * 0.01, (double)(u[3] - u[2]) * 0.01, (double)(u[4] - u[3]) * 0.01, (double)(u[5] - u[4]) * 0.01, (double)(u[0] - u[-3]) * 0.01);
    }
  }
  if (p->dbg.p && std::getenv("LVF_COST_TIMING")) {
    unsigned long long t[64];
    LVF_HIP(hipMemcpy(t, p->dbg.p, sizeof(t), hipMemcpyDeviceToHost));
    auto us = [&](int a, int b) { return ((double)t[b] - (double)t[a]) * 0.01; };
    std::fprintf(stderr, "cost+decide (us): imu wg0 stage %.2f | raw (one lane) %.2f | weight+sum %.2f ; first visual wg %.2f (starts %.2f after imu wg0) ; decision starts %.2f after imu wg0's start: sums %.2f | logic %.2f | commit %.2f | record %.2f\n",
                 us(8, 9), us(9, 10), us(10, 11), us(12, 13), us(8, 12), us(8, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5));
  }
  if (p->dbg.p && std::getenv("LVF_BACK_TIMING")) {
    unsigned long long t[64];
    LVF_HIP(hipMemcpy(t, p->dbg.p, sizeof(t), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "backsolve phases (us):");
    for (unsigned long long k = 1; k < t[63] && k < 55; ++k) std::fprintf(stderr, " %.2f", (double)(t[k] - t[k - 1]) * 0.01);
    if (p->chain && p->chain->back_tail_merged)
      std::fprintf(stderr, " | merged launch, us after workgroup 0's start: step applied %.2f ; first landmark workgroup starts %.2f, sees the poses %.2f, done %.2f ; last one starts %.2f, sees %.2f, done %.2f",
                   (double)(t[55] - t[0]) * 0.01, (double)(t[56] - t[0]) * 0.01, (double)(t[57] - t[0]) * 0.01, (double)(t[58] - t[0]) * 0.01, (double)(t[59] - t[0]) * 0.01,
                   (double)(t[60] - t[0]) * 0.01, (double)(t[61] - t[0]) * 0.01);
    std::fprintf(stderr, "\n");
  }
  out->cost_before = c.cost_before; out->cost_after = c.cost_after; out->model = c.model; out->dxnorm = c.dxnorm; out->xnorm = c.xnorm; out->gmax = c.gmax;
  out->solved = c.solved != 0; out->accepted = c.accepted != 0;
  p->last_radius = c.last_radius;
  *radius = c.radius; *decrease = c.decrease;
  return LVF_OK;
}

// waits until the host-visible mirror shows at least `iter` closed iterations (or the loop finished); falls back to a stream
// synchronisation when the mirror does not move (e.g. device writes to host memory only becoming visible at kernel boundaries)
static int wait_for_iteration(lvf_problem* p, int iter) {
  const volatile LmCtl* r = p->rec;
  const auto t0 = std::chrono::steady_clock::now();
  while (r->iter < iter && !r->done) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05) {
      LmCtl c;
      LVF_TRY(download_ctl(p, &c));          // synchronises the stream
      *p->rec = c;
      break;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  return LVF_OK;
}

// Elimination plan (host, <= a few hundred nodes): which (v, ba, bg) blocks are factorised sparsely, in which level, with which
// neighbour rows; the S row of every unknown.  Rebuilt only when (n_kf, IMU index pairs) change.  LVF_SPARSE_VB=0 keeps every
// block in the dense corner (the round-1 layout, poses last) for A/B measurements.
static int build_elimination_plan(lvf_problem* p) {
  const int n = p->n_kf;
  std::vector<int32_t> key;
  key.push_back(n);
  const lvf_batch* imu = p->imu;
  const bool have_idx = imu && imu->n > 0 && (int)imu->host_kf1.size() == imu->n && (int)imu->host_kf2.size() == imu->n;
  if (imu && imu->n > 0) {
    key.push_back(have_idx ? 1 : 0);
    if (have_idx) { key.insert(key.end(), imu->host_kf1.begin(), imu->host_kf1.end()); key.insert(key.end(), imu->host_kf2.begin(), imu->host_kf2.end()); }
  }
  if (key == p->plan_key && p->ld > 0) return LVF_OK;
  static const bool sparse_on = [] { const char* e = std::getenv("LVF_SPARSE_VB"); return !(e && e[0] == '0'); }();
  std::vector<std::vector<char>> va(n, std::vector<char>(n, 0)), pa(n, std::vector<char>(n, 0));
  bool sparse = sparse_on && (!imu || imu->n == 0 || have_idx);
  if (have_idx)
    for (int f = 0; f < imu->n; ++f) {
      const int i = imu->host_kf1[f], j = imu->host_kf2[f];
      if (i < 0 || j < 0 || i >= n || j >= n || i == j) continue;
      va[i][j] = va[j][i] = 1;
      pa[i][i] = pa[i][j] = pa[j][i] = pa[j][j] = 1;
    }
  struct NodeInfo { int kf; std::vector<int> vb, pose; };
  std::vector<NodeInfo> nodes;
  std::vector<char> alive(n, 1);
  SpLevels lv{};
  // How many levels to eliminate sparsely.  Whatever is left rides in the dense corner, which is factorised in 64-column block steps:
  // a level beyond the first (level 0 rides in the Schur launch) costs a launch (~8.5 us), a block step ~18 us.  A dry run of the greedy
  // level construction gives the number of blocks left after each level; the cut-off minimises 8.5 (levels - 1) + 18 block steps (measured launch costs, us).
  // (At 50 keyframes: 5 levels and one block in the corner's padding instead of 6 levels; at 5: level 0 only.)
  int max_levels = kSpMaxLevels;
  if (sparse) {
    std::vector<std::vector<char>> va2 = va;
    std::vector<char> alive2(n, 1);
    std::vector<int> left_after;                      // blocks left after level l
    for (int l = 0; l < kSpMaxLevels; ++l) {
      std::vector<char> blocked(n, 0);
      std::vector<int> chosen;
      for (int k = 0; k < n; ++k) {
        if (!alive2[k] || blocked[k]) continue;
        chosen.push_back(k);
        for (int u = 0; u < n; ++u) if (va2[k][u]) blocked[u] = 1;
      }
      if (chosen.empty()) break;
      for (int b : chosen) {
        std::vector<int> nb_;
        for (int u = 0; u < n; ++u) if (va2[b][u] && alive2[u]) nb_.push_back(u);
        for (int u : nb_) { for (int w : nb_) if (w != u) va2[u][w] = 1; va2[u][b] = 0; }
        alive2[b] = 0;
      }
      int left = 0;
      for (int k = 0; k < n; ++k) left += alive2[k] ? 1 : 0;
      left_after.push_back(left);
    }
    double best = 1e300;
    for (size_t l = 0; l < left_after.size(); ++l) {
      const int steps = (9 * left_after[l] + p->dp + 1 + 63) / 64;
      const double cost = 8.5 * (double)l + 18.0 * steps;
      if (cost < best - 1e-9) { best = cost; max_levels = (int)l + 1; }
    }
    static const int force_levels = [] { const char* e = std::getenv("LVF_FORCE_LEVELS"); return e ? std::atoi(e) : 0; }();      // experiment
    if (force_levels > 0) max_levels = std::min((int)left_after.size(), force_levels);
  }
  while (sparse && lv.n < std::min(max_levels, kSpMaxLevels)) {
    std::vector<char> blocked(n, 0);
    std::vector<int> chosen;
    for (int k = 0; k < n; ++k) {
      if (!alive[k] || blocked[k]) continue;
      int m = 1;
      for (int u = 0; u < n; ++u) m += 9 * (va[k][u] && alive[u]) + 6 * pa[k][u];
      if (m > kSpMaxRows) continue;
      chosen.push_back(k);
      for (int u = 0; u < n; ++u) if (va[k][u]) blocked[u] = 1;
    }
    if (chosen.empty()) break;
    lv.first[lv.n] = (int)nodes.size(); lv.count[lv.n] = (int)chosen.size(); ++lv.n;
    for (int b : chosen) {
      NodeInfo ni; ni.kf = b;
      for (int u = 0; u < n; ++u) { if (va[b][u] && alive[u]) ni.vb.push_back(u); if (pa[b][u]) ni.pose.push_back(u); }
      nodes.push_back(std::move(ni));
    }
    for (size_t ci = 0; ci < chosen.size(); ++ci) {   // fill-in among the neighbours of an eliminated block
      const int b = chosen[ci];
      const NodeInfo& ni = nodes[lv.first[lv.n - 1] + (int)ci];
      for (int u : ni.vb) {
        for (int w : ni.vb) if (w != u) va[u][w] = 1;
        for (int k : ni.pose) pa[u][k] = 1;
        va[u][b] = 0;
      }
      alive[b] = 0;
    }
  }
  // S rows
  const int ns = (int)nodes.size();
  std::vector<int> vbcol(n, -1);
  for (int s_ = 0; s_ < ns; ++s_) vbcol[nodes[s_].kf] = 9 * s_;
  p->off = (9 * ns + 1) & ~1;
  int ndv = 0;
  for (int k = 0; k < n; ++k) if (alive[k]) vbcol[k] = p->off + 9 * ndv++;
  p->off_pose = p->off + 9 * ndv;
  p->ndense = 9 * ndv + p->dp;
  p->aug = p->off + p->ndense;
  p->nb = (p->ndense + 1 + 63) / 64;
  p->ld = p->off + 64 * p->nb;
  p->perm_h.assign(p->d, 0);
  std::vector<int> iperm(p->ld, -1);
  for (int i = 0; i < p->dp; ++i) p->perm_h[i] = p->off_pose + i;
  for (int k = 0; k < n; ++k) for (int c = 0; c < 9; ++c) p->perm_h[p->dp + 9 * k + c] = vbcol[k] + c;
  for (int i = 0; i < p->d; ++i) iperm[p->perm_h[i]] = i;
  iperm[p->aug] = -2;
  // device tables
  std::vector<SpNode> dn(ns);
  std::vector<int> rows, owner, rows_nat;
  p->sp_tiles.assign(lv.n, 1); p->sp_shmem.assign(lv.n, 0); p->sp_item0.assign(lv.n, 0); p->sp_items.assign(lv.n, 0);
  for (int l = 0; l < lv.n; ++l) {
    int mmax = 0;
    for (int s_ = lv.first[l]; s_ < lv.first[l] + lv.count[l]; ++s_) {
      const NodeInfo& ni = nodes[s_];
      dn[s_].col = 9 * s_; dn[s_].row_off = (int)rows.size(); dn[s_].id = ni.kf;
      std::vector<int> r;
      for (int u : ni.vb) for (int c = 0; c < 9; ++c) r.push_back(vbcol[u] + c);
      for (int k : ni.pose) for (int c = 0; c < 6; ++c) r.push_back(p->off_pose + 6 * k + c);
      r.push_back(p->aug);
      std::sort(r.begin(), r.end());
      dn[s_].m = (int)r.size();
      mmax = std::max(mmax, dn[s_].m);
      rows.insert(rows.end(), r.begin(), r.end());
      owner.insert(owner.end(), r.size(), s_);
    }
    p->sp_item0[l] = dn[lv.first[l]].row_off; p->sp_items[l] = (int)rows.size() - p->sp_item0[l];
    const int P = mmax * (mmax + 1) / 2;
    p->sp_tiles[l] = std::max(1, std::min(32, (P + 2047) / 2048));
    p->sp_shmem[l] = (9 * mmax + 81 + 9) * 8 + 2 * 4 * mmax + 16;
  }
  p->sp_levels = lv;
  hipStream_t q = p->ctx->stream;
  LVF_TRY(p->perm.assign(p->perm_h.data(), p->perm_h.size(), q)); LVF_TRY(p->iperm.assign(iperm.data(), iperm.size(), q));
  if (ns) {
    rows_nat.resize(rows.size());
    for (size_t i = 0; i < rows.size(); ++i) rows_nat[i] = iperm[rows[i]];
    LVF_TRY(p->sp_rows_nat.assign(rows_nat.data(), rows_nat.size(), q));
    LVF_TRY(p->sp_nodes.assign(dn.data(), dn.size(), q)); LVF_TRY(p->sp_rows.assign(rows.data(), rows.size(), q)); LVF_TRY(p->sp_owner.assign(owner.data(), owner.size(), q));
    LVF_TRY(p->sp_W.ensure(rows.size() * 9)); LVF_TRY(p->sp_L.ensure((size_t)ns * 81));
    p->sp_wstride = (int)rows.size();
  }
  LVF_HIP(hipStreamSynchronize(q));      // the host vectors above go out of scope
  p->plan_key = std::move(key);
  return LVF_OK;
}

int problem_configure(lvf_problem* p) {
  static const bool cfg_timing = std::getenv("LVF_CONFIGURE_TIMING") != nullptr;
  const auto cfg_t0 = std::chrono::steady_clock::now();
  auto cfg_mark = [&](const char* what) {
    if (cfg_timing) std::fprintf(stderr, "  problem_configure: %s at %.3f ms\n", what, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - cfg_t0).count());
  };
  lvf_state* st = p->st;
  lvf_ctx* ctx = p->ctx;
  p->n_kf = st->n_kf; p->n_lm = st->n_lm;
  p->d = 15 * p->n_kf; p->dp = 6 * p->n_kf;
  p->ldE = ((p->dp + 1 + 15) / 16) * 16;
  p->dpad = ((p->d + 1 + 63) / 64) * 64;
  LVF_TRY(build_elimination_plan(p));                 // sets ld, off, off_pose, ndense, aug, nb and the sparse levels
  cfg_mark("elimination plan");
  const size_t nS = (size_t)p->dpad * p->dpad;
  LVF_TRY(p->Dinv.ensure((size_t)p->nb * kNB * kNB)); LVF_TRY(p->Ldiag.ensure((size_t)p->nb * kNB * kNB));
  LVF_TRY(p->B.ensure(nS)); LVF_TRY(p->S.ensure((size_t)p->ld * p->ld)); LVF_TRY(p->gc.ensure(p->dpad)); LVF_TRY(p->dxc.ensure(p->dpad));
  LVF_TRY(p->C.ensure(p->n_lm)); LVF_TRY(p->gr.ensure(p->n_lm)); LVF_TRY(p->Cd.ensure(p->n_lm)); LVF_TRY(p->dxl.ensure(p->n_lm));
  LVF_TRY(p->scal.ensure(SC_ALLOC)); LVF_TRY(p->sp_sync.ensure(kSpMaxLevels));
  // candidate state x + dx (an accepted candidate is copied into the state by k_lm_decide)
  LVF_TRY(p->poses2.ensure(std::max(st->poses.cap, (size_t)7 * p->n_kf))); LVF_TRY(p->vel2.ensure(std::max(st->vel.cap, (size_t)3 * p->n_kf)));
  LVF_TRY(p->ba2.ensure(std::max(st->ba.cap, (size_t)3 * p->n_kf))); LVF_TRY(p->bg2.ensure(std::max(st->bg.cap, (size_t)3 * p->n_kf)));
  LVF_TRY(p->invd2.ensure(std::max(st->inv_depth.cap, (size_t)p->n_lm)));
  LVF_TRY(p->pose_const.ensure((size_t)p->n_kf + 8)); LVF_TRY(p->fail.ensure(1));      // (+8: cleared in 8-byte words)
  p->pose_const_h.assign(p->n_kf, 0);
  cfg_mark("buffers");
  p->tf_work.n = 0;
  p->tf_unique_lk2 = false; p->tf_k1_first = false; p->compact = false; p->tf_sorted_copy = false;
  lvf_batch* two_frame = p->tf;
  if (two_frame && two_frame->n && two_frame->sorted_by_kf && !two_frame->kf2_counts.empty() && two_frame->unique_lk2_known) {
    // the creator (the persistent window) vouches for the shape: sorted by current keyframe, k1 < k2, one block per (landmark, keyframe)
    size_t total = 0, nw = 0;
    for (int32_t c : two_frame->kf2_counts) { total += (size_t)c; nw += ((size_t)c + kT - 1) / kT; }
    if (total != (size_t)two_frame->n) { set_error("two-frame batch: per-keyframe counts do not add up to the number of blocks"); return LVF_ERR_INVALID; }
    LVF_TRY(p->h_tf_work.reserve(nw + 1));
    nw = 0;
    int at = 0;
    for (size_t k = 0; k < two_frame->kf2_counts.size(); ++k)
      for (int left = two_frame->kf2_counts[k]; left > 0; left -= kT) { const int c = std::min(left, kT); p->h_tf_work[nw++] = TfWork{at, c, (int)k}; at += c; }
    LVF_TRY(p->tf_work.assign(p->h_tf_work.p, nw, ctx->stream));
    p->tf_k1_first = true; p->tf_unique_lk2 = true;
  } else if (two_frame && two_frame->n && two_frame->sorted_by_kf && !two_frame->host_kf2.empty()) {
    // work list for the sorted fast path: runs of <= kT blocks sharing one current keyframe; k1 == k2 disables it.
    // ONE pass over the blocks gathers everything the host has to know about them (the list is 72 k entries at configs[3] and this
    // function is on adapt::Solve's path: seven separate passes were 0.45 ms of it):
    //   * k1 != k2 everywhere, k1 < k2 everywhere;
    //   * the current-keyframe runs (their starts) and, per run, how many blocks have which first keyframe (for the counting sort below);
    //   * how often the first keyframe steps DOWN inside a run (ids in creation order: almost never);
    //   * one block per (landmark, current keyframe) and ONE first keyframe per landmark, with two per-landmark tables: `seen_run[l]` =
    //     the last run landmark l appeared in (a repeat inside one run is a duplicate pair), `first_kf[l]` = its first keyframe.  The plain
    //     stores into E[l][k2 columns] must never meet the adds into E[l][k1 columns]; BuildProblem's blocks always satisfy this, a
    //     hand-made batch may not.
    const std::vector<int32_t>& k2 = two_frame->host_kf2; const std::vector<int32_t>& k1 = two_frame->host_kf1;
    const std::vector<int32_t>& lmh = two_frame->host_lm;
    const int n = two_frame->n, nkf = p->n_kf;
    static const bool k1sort_on = [] { const char* e = std::getenv("LVF_TF_K1SORT"); return !(e && e[0] == '0'); }();
    const bool want_hist = k1sort_on && nkf <= 256;
    bool ok = true, k1_first = true;
    bool uniq = two_frame->unique_lk2_known || lmh.size() == (size_t)n;
    const bool check_uniq = uniq && !two_frame->unique_lk2_known;
    const int32_t nl = (int32_t)p->n_lm;
    std::vector<int32_t> seen_run, first_kf;
    if (check_uniq) { seen_run.assign((size_t)nl, -1); first_kf.assign((size_t)nl, -1); }
    std::vector<int32_t> run_start;                 // first block of every current-keyframe run (+ n at the end)
    std::vector<int32_t> hist;                      // [run][first keyframe] block counts
    run_start.reserve((size_t)nkf + 2);
    if (want_hist) hist.reserve((size_t)(nkf + 1) * nkf);
    int descents = 0, cur = INT32_MIN, run = -1;
    int32_t* hrow = nullptr;
    for (int i = 0; i < n; ++i) {
      const int a = k1[i], c2 = k2[i];
      ok = ok && a != c2; k1_first = k1_first && a < c2;
      if (c2 != cur) {
        cur = c2; ++run; run_start.push_back(i);
        if (want_hist) { hist.resize(hist.size() + (size_t)nkf, 0); hrow = hist.data() + (size_t)run * nkf; }
      } else descents += a < k1[i - 1] ? 1 : 0;
      if (want_hist) ++hrow[std::min(std::max(a, 0), nkf - 1)];
      if (check_uniq && uniq) {
        const int32_t l = lmh[i];
        if (l < 0 || l >= nl) uniq = false;
        else {
          if (seen_run[l] == run) uniq = false;
          seen_run[l] = run;
          if (first_kf[l] < 0) first_kf[l] = a; else if (first_kf[l] != a) uniq = false;
        }
      }
    }
    run_start.push_back(n);
    p->tf_k1_first = ok && k1_first;
    if (ok) {
      // built straight into pinned staging owned by the problem: the upload is a real asynchronous copy and this function does not
      // have to wait for the stream before returning
      size_t nw = 0;
      for (size_t r = 0; r + 1 < run_start.size(); ++r) nw += (size_t)(run_start[r + 1] - run_start[r] + kT - 1) / kT;
      LVF_TRY(p->h_tf_work.reserve(nw + 1));
      nw = 0;
      for (size_t r = 0; r + 1 < run_start.size(); ++r)
        for (int i = run_start[r]; i < run_start[r + 1]; i += kT) p->h_tf_work[nw++] = TfWork{i, std::min(kT, run_start[r + 1] - i), k2[run_start[r]]};
      LVF_TRY(p->tf_work.assign(p->h_tf_work.p, nw, ctx->stream));
      // first keyframes out of order inside the runs (more than one block in eight steps DOWN): sorted copies for the linearisation.
      // BuildProblem's own order (landmark ids in creation order) passes untouched.
      if (want_hist && (size_t)descents * 8 > (size_t)n) {
        LVF_TRY(p->h_tfs_perm.reserve((size_t)n));
        int* perm = p->h_tfs_perm.p;
        for (size_t r = 0; r + 1 < run_start.size(); ++r) {     // counting sort of each run by first keyframe (stable): the histogram is there
          int32_t* cnt = hist.data() + r * (size_t)nkf;
          int32_t at = run_start[r];
          for (int k = 0; k < nkf; ++k) { const int32_t c = cnt[k]; cnt[k] = at; at += c; }
          for (int t = run_start[r]; t < run_start[r + 1]; ++t) perm[cnt[std::min(std::max(k1[t], 0), nkf - 1)]++] = t;
        }
        LVF_TRY(p->tfs_perm.assign(perm, (size_t)n, ctx->stream));
        LVF_TRY(p->tfs_fo.ensure(n)); LVF_TRY(p->tfs_ob.ensure(n)); LVF_TRY(p->tfs_lm.ensure(n)); LVF_TRY(p->tfs_k1.ensure(n)); LVF_TRY(p->tfs_k2.ensure(n));
        hipLaunchKernelGGL(k_tf_gather, dim3(grid(n)), dim3(kT), 0, ctx->stream, n, p->tfs_perm.p, (const double2*)two_frame->ob_a.p, (const double2*)two_frame->ob_b.p,
                           two_frame->idx_a.p, two_frame->idx_b.p, two_frame->idx_c.p, p->tfs_fo.p, p->tfs_ob.p, p->tfs_lm.p, p->tfs_k1.p, p->tfs_k2.p);
        LVF_HIP(hipGetLastError());
        p->tf_sorted_copy = true;
      }
      p->tf_unique_lk2 = uniq;
    }
  }
  cfg_mark("TwoFrame work list + shape checks");
  // landmark tracks -> band-limited Schur (device side: the TwoFrame indices already live there)
  p->band_ready = false;
  static const bool band_on = [] { const char* e = std::getenv("LVF_SCHUR_BAND"); return !(e && e[0] == '0'); }();
  const bool band_ok = band_on && p->n_lm > 0 && (size_t)(4 * p->n_kf + 1) * sizeof(int) <= 48 * 1024;
  // compact landmark layout + slabs (atomic-free TwoFrame linearisation): needs the sorted work list, one block per (landmark,
  // keyframe), the landmark's first keyframe ahead of its observations, and the merged band-Schur launch
  static const bool compact_on = [] { const char* e = std::getenv("LVF_COMPACT"); return !(e && e[0] == '0'); }();
  const size_t shb = ((size_t)kSchurRows * (p->ldE + 16) + kSchurRows) * sizeof(double) + kBandRowsMax * sizeof(int);
  const bool merged = shb <= 64 * 1024 && p->sp_levels.n > 0 && (size_t)p->sp_shmem[0] <= 64 * 1024;
  const bool want_compact = band_ok && compact_on && merged && p->dp <= 320 && two_frame && two_frame->n && p->tf_work.n && p->n_kf <= kMaxStagedKf && p->tf_unique_lk2 && p->tf_k1_first;
  LVF_TRY(p->E.ensure((size_t)p->n_lm * p->ldE));
  if (band_ok) { LVF_TRY(p->lm_kmin.ensure(p->n_lm)); LVF_TRY(p->lm_kmax.ensure(p->n_lm)); LVF_TRY(p->lm_order.ensure(p->n_lm)); LVF_TRY(p->lm_nactive.ensure(1)); }
  // atomic-free mode: E's non-zero pattern is fixed for the problem and fully overwritten by every linearisation — cleared once, here;
  // one launch for the clears AND the landmark tracks' initial values (a persistent window reconfigures every tick)
  {
    ZeroList z{};
    if (want_compact && p->n_lm) { z.p[z.count] = p->E.p; z.n[z.count] = (unsigned long long)p->n_lm * p->ldE; ++z.count; }
    z.p[z.count] = reinterpret_cast<double*>(p->pose_const.p); z.n[z.count] = (unsigned long long)(p->n_kf + 7) / 8; ++z.count;      // (the buffer's capacity is padded)
    z.p[z.count] = p->dxc.p; z.n[z.count] = (unsigned long long)p->dpad; ++z.count;
    hipLaunchKernelGGL(k_zero_multi_ranges, dim3(512, z.count + (band_ok ? 1 : 0)), dim3(kT), 0, ctx->stream, z, p->n_lm, band_ok ? p->lm_kmin.p : nullptr, band_ok ? p->lm_kmax.p : nullptr);
    LVF_HIP(hipGetLastError());
  }
  if (band_ok) {
    hipStream_t q = ctx->stream;
    if (two_frame && two_frame->n)
      hipLaunchKernelGGL(k_lm_range, dim3(grid(two_frame->n)), dim3(kT), 0, q, two_frame->n, two_frame->idx_a.p, two_frame->idx_b.p, two_frame->idx_c.p,
                         p->lm_kmin.p, p->lm_kmax.p);
    if (want_compact) {
      LVF_TRY(p->lm_eoff.ensure(p->n_lm)); LVF_TRY(p->n_slots.ensure(1));
      // (the slot offsets ride beside the sort: two independent one-workgroup chains, one launch)
      hipLaunchKernelGGL(k_lm_sort_offsets, dim3(2), dim3(1024), (size_t)(4 * p->n_kf + 1) * sizeof(int), q, p->n_lm, p->n_kf, p->lm_kmin.p, p->lm_kmax.p, p->lm_order.p,
                         p->lm_nactive.p, p->lm_eoff.p, p->n_slots.p);
    } else
    hipLaunchKernelGGL(k_lm_sort, dim3(1), dim3(1024), (size_t)(4 * p->n_kf + 1) * sizeof(int), q, p->n_lm, p->n_kf, p->lm_kmin.p, p->lm_kmax.p, p->lm_order.p,
                       p->lm_nactive.p);
    LVF_HIP(hipGetLastError());
    p->band_ready = true;
    p->band_rows_built = 0;         // the bands changed: the work list is rebuilt with the next chain
    p->band_epoch += 1;
    if (want_compact) {
      const size_t cap_slots = (size_t)p->n_lm * (size_t)std::max(p->n_kf - 1, 1);      // worst case: every landmark seen by every later keyframe
      const int n_wg = (int)p->tf_work.n;
      LVF_TRY(p->tf_slot.ensure(two_frame->n));
      LVF_TRY(p->slotB.ensure(cap_slots * 8));
      LVF_TRY(p->Ct.ensure(p->n_lm)); LVF_TRY(p->grt.ensure(p->n_lm));
      LVF_TRY(p->slabP.ensure((size_t)n_wg * p->n_kf * kSlabRow)); LVF_TRY(p->slabQ.ensure((size_t)n_wg * kSlabQ));
      {
        const int g_slots = grid(two_frame->n);
        hipLaunchKernelGGL(k_tf_slots_zero, dim3(g_slots + 256), dim3(kT), 0, q, two_frame->n, g_slots, p->tf_lm(), p->tf_k2(), p->lm_kmin.p, p->lm_eoff.p, p->tf_slot.p,
                           p->n_slots.p, p->slotB.p);
      }
      LVF_HIP(hipGetLastError());
      // run_first[k] = first workgroup of current keyframe k's run (the work list is sorted by k2); run_first[n_kf] = n_wg
      LVF_TRY(p->h_run_first.reserve((size_t)p->n_kf + 1));
      {
        int w = 0;
        for (int k = 0; k <= p->n_kf; ++k) {
          while (w < n_wg && p->h_tf_work[w].k2 < k) ++w;
          p->h_run_first[k] = w;
        }
      }
      LVF_TRY(p->run_first.assign(p->h_run_first.p, (size_t)p->n_kf + 1, q));
      p->compact = true;
    }
  }
  cfg_mark("device-side layout launches + E");
  // no stream wait here: every host source above is pinned and owned by the problem (or was waited for by the plan builder)
  p->linearized = false;
  p->chain_ready = false;
  return LVF_OK;
}

}  // namespace lvf

using namespace lvf;

// ------------------------------------------------------------------------------------------------ a batch of windows
// W independent windows advanced by ONE chain of launches per LM iteration: every kernel of the iteration takes blockIdx.y = window and
// reads that window's argument block from a device table.  A single window's iteration is a chain of ~23 small launches that leaves
// most of the 256 CUs idle (sequential pivots, 2 + k workgroups per panel step); a batch fills the same launches W times over — the
// shape of every independent-window client of the reference: RL environments (src/lvio_fusion/src/environment.cpp:18-115), loop-closure
// candidates (relocator.cpp:196-206), per-submap replays, and what ONE GPU of the 8-GPU sharding works on.
struct lvf_problem_batch {
  lvf_ctx* ctx = nullptr;
  std::vector<lvf_problem*> probs;
  bool tables = false;               // every member is batchable: table launches; otherwise the windows' own chains run back to back
  int W = 0, max_levels = 0, max_nb = 0;
  // per-stage argument tables [W] (device) and the launch shapes (max over the windows)
  lvf::DevBuf<lvf::ImuArgs> imu_lin, imu_cost; int g_imu_lin = 0, g_imu_cost = 0;
  lvf::DevBuf<lvf::LinArgs> lin; int g_lin = 0; size_t lds_lin = 0;
  lvf::DevBuf<lvf::TfReduceArgs> red; int g_red = 0; size_t lds_red = 0;
  lvf::DevBuf<lvf::PrepArgs> prep; int g_prep = 0; size_t lds_prep = 0;
  int first_own_level = 0;           // min over the windows: the first sparse level that is a launch of its own
  lvf::DevBuf<lvf::SchurSp0Args> ssp0; int g_ssp0 = 0; size_t lds_ssp0 = 0;
  lvf::DevBuf<lvf::SpArgs> sp[lvf::kSpMaxLevels]; int g_sp[lvf::kSpMaxLevels] = {0}; int lds_sp[lvf::kSpMaxLevels] = {0};
  lvf::DevBuf<lvf::CholArgs> chol;
  lvf::DevBuf<lvf::BackArgs> back; size_t lds_back = 0;
  lvf::DevBuf<lvf::TailArgs> tail; int g_tail = 0; size_t lds_tail = 0;
  lvf::DevBuf<lvf::CostArgs> cost; int g_cost = 0;
  lvf::DevBuf<lvf::DecideArgs> dec;
  lvf::DevBuf<lvf::ZeroList> zero;   // every window's full accumulator list (cleared in one launch when a window is not known clean)
  double huber_built = -1.0;
  // A batch of more than one window sums its Schur complements over wider landmark slices (fewer output atomics: LVF_BATCH_BAND_ROWS,
  // default 128).  The work lists for that width belong to the BATCH — a member's own list, slice width and chain are never touched, so a
  // window solved alone, then in a batch, then alone again runs the same arithmetic the first and the third time.
  std::vector<std::unique_ptr<lvf::DevBuf<int4>>> band_work; std::vector<int> n_band_work, band_epoch;
  bool orphaned = false;             // a member was destroyed before the batch: every later call fails with LVF_ERR_STATE
};

namespace lvf {

void batch_orphan(lvf_problem_batch* b, lvf_problem* dying) {
  b->orphaned = true;
  for (lvf_problem* p : b->probs)
    if (p != dying) p->batches.erase(std::remove(p->batches.begin(), p->batches.end(), b), p->batches.end());
  b->probs.clear();
}

template <typename T>
static int upload_table(DevBuf<T>& dst, const std::vector<T>& src, hipStream_t q) {
  LVF_TRY(dst.ensure(src.size()));
  // pageable source: the runtime stages the copy before returning, so `src` may go out of scope
  if (!src.empty()) LVF_HIP(hipMemcpyAsync(dst.p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, q));
  return LVF_OK;
}

static int batch_build_tables(lvf_problem_batch* b, double huber) {
  hipStream_t q = b->ctx->stream;
  const int W = b->W;
  bool all = true;
  static const int batch_rows = [] { const char* e = std::getenv("LVF_BATCH_BAND_ROWS"); return e ? std::atoi(e) : 128; }();
  if (b->orphaned) { set_error("lvf_problem_batch: a member problem was destroyed before the batch"); return LVF_ERR_STATE; }
  for (lvf_problem* p : b->probs) {
    if (chain_stale(p)) LVF_TRY(build_chain(p));
    LVF_TRY(await_band_work(p));             // (the member's own list: its count shares the pinned slot build_band_work reads below)
    all = all && p->chain->batchable;
  }
  b->tables = all;
  if (!all) return LVF_OK;
  const bool wide = W > 1 && batch_rows != 64;
  if (wide) {
    b->band_work.resize(W); b->n_band_work.resize(W, 0); b->band_epoch.resize(W, -1);
    for (int w = 0; w < W; ++w) {
      lvf_problem* p = b->probs[w];
      if (!b->band_work[w]) b->band_work[w].reset(new DevBuf<int4>());
      if (b->band_epoch[w] == p->band_epoch) continue;
      LVF_TRY(build_band_work(p, batch_rows, *b->band_work[w], &b->n_band_work[w]));
      b->band_epoch[w] = p->band_epoch;
    }
  }
  std::vector<ImuArgs> il(W), ic(W); std::vector<LinArgs> li(W); std::vector<TfReduceArgs> rd(W); std::vector<PrepArgs> pr(W); std::vector<SchurSp0Args> ss(W);
  std::vector<CholArgs> ch(W); std::vector<BackArgs> bk(W); std::vector<TailArgs> tl(W); std::vector<CostArgs> co(W); std::vector<DecideArgs> de(W); std::vector<ZeroList> zl(W);
  b->max_levels = 0; b->max_nb = 0;
  b->g_red = 0; b->lds_red = 0; b->lds_prep = 0; b->first_own_level = kSpMaxLevels;
  b->g_imu_lin = b->g_imu_cost = b->g_lin = b->g_prep = b->g_ssp0 = b->g_tail = b->g_cost = 0; b->lds_ssp0 = b->lds_back = b->lds_tail = b->lds_lin = 0;
  for (int w = 0; w < W; ++w) {
    const lvf_problem* p = b->probs[w];
    const Chain& c = *p->chain;
    il[w] = c.imu_lin; ic[w] = c.imu_cost; li[w] = c.lin; li[w].huber = huber; rd[w] = c.red; if (!p->compact) rd[w].nblocks = 0;
    b->g_red = std::max(b->g_red, rd[w].nblocks); pr[w] = c.early ? c.prep_early : c.prep; ss[w] = c.ssp0; ch[w] = c.chol; bk[w] = c.back; tl[w] = c.tail;
    if (wide) {                                          // the batch's own slice width and work list (the member's chain keeps its own)
      SchurSp0Args& a = ss[w];
      a.rows = band_rows_clamped(batch_rows); a.n_slices = (p->n_lm + a.rows - 1) / a.rows;
      a.work = b->band_work[w]->p; a.n_work = b->n_band_work[w];
      a.nblocks = a.n_work + a.sp.nblocks + a.sp_b.nblocks + a.sp_c.nblocks;
    }
    if (rd[w].nblocks > rd[w].own_blocks) b->lds_red = std::max(b->lds_red, c.red_lds);
    if (pr[w].nblocks > pr[w].own_blocks) b->lds_prep = std::max(b->lds_prep, c.prep_lds);
    b->first_own_level = std::min(b->first_own_level, c.first_own_level);
    co[w] = c.cost; co[w].huber = huber; de[w] = c.dec; zl[w] = c.zero;
    if (W >= 4) {                                        // fatter cost / zeroing workgroups in a batch (see cost_visual_value)
      CostArgs& k = co[w];
      const int t = 4, per = kT * t;
      k.tiles = t;
      k.a.g_tc = (k.a.n_tc + per - 1) / per; k.a.g_tf = (k.a.n_tf + per - 1) / per;
      k.nblocks = k.g_imu + k.a.g_tc + k.a.g_tf + (k.a.n_po + per - 1) / per;
      k.zero_wgs = std::min(k.zero_wgs, 48);
      TailArgs& ta = tl[w];                              // ... and fewer landmark workgroups per window (64 windows: 203 -> 170 us with 256 instead of 640)
      const int g2 = std::min(ta.g_lm, 256);
      ta.nblocks -= ta.g_lm - g2; ta.g_lm = g2;
    }
    b->g_imu_lin = std::max(b->g_imu_lin, c.imu_lin.n + c.imu_lin.zero_wgs); b->g_imu_cost = std::max(b->g_imu_cost, c.imu_cost.n + c.imu_cost.zero_wgs);
    b->g_lin = std::max(b->g_lin, c.lin.nblocks); b->lds_lin = std::max(b->lds_lin, c.lin_lds); b->g_prep = std::max(b->g_prep, pr[w].nblocks); b->g_ssp0 = std::max(b->g_ssp0, ss[w].nblocks);
    b->lds_ssp0 = std::max(b->lds_ssp0, c.ssp0_lds); b->lds_back = std::max(b->lds_back, c.back_lds); b->g_tail = std::max(b->g_tail, tl[w].nblocks);
    b->lds_tail = std::max(b->lds_tail, c.tail_lds); b->g_cost = std::max(b->g_cost, co[w].nblocks + co[w].zero_wgs);
    b->max_levels = std::max(b->max_levels, c.n_levels); b->max_nb = std::max(b->max_nb, p->nb);
  }
  LVF_TRY(upload_table(b->imu_lin, il, q)); LVF_TRY(upload_table(b->imu_cost, ic, q)); LVF_TRY(upload_table(b->lin, li, q)); LVF_TRY(upload_table(b->red, rd, q)); LVF_TRY(upload_table(b->prep, pr, q));
  LVF_TRY(upload_table(b->ssp0, ss, q)); LVF_TRY(upload_table(b->chol, ch, q)); LVF_TRY(upload_table(b->back, bk, q)); LVF_TRY(upload_table(b->tail, tl, q));
  LVF_TRY(upload_table(b->cost, co, q)); LVF_TRY(upload_table(b->dec, de, q)); LVF_TRY(upload_table(b->zero, zl, q));
  for (int lv = b->first_own_level; lv < b->max_levels; ++lv) {        // (the levels below ride in the launches ahead: Chain::first_own_level)
    std::vector<SpArgs> sp(W);
    b->g_sp[lv] = 0; b->lds_sp[lv] = 0;
    for (int w = 0; w < W; ++w) {
      const Chain& c = *b->probs[w]->chain;
      if (lv >= c.first_own_level && lv < c.n_levels) { sp[w] = c.sp[lv]; b->g_sp[lv] = std::max(b->g_sp[lv], c.sp[lv].nblocks); b->lds_sp[lv] = std::max(b->lds_sp[lv], c.sp_lds[lv]); }
      else { sp[w] = SpArgs{}; sp[w].nblocks = 0; }
    }
    LVF_TRY(upload_table(b->sp[lv], sp, q));
  }
  b->huber_built = huber;
  return LVF_OK;
}

// one LM iteration of every window of the batch; nothing is waited for
static int batch_enqueue_iteration(lvf_problem_batch* b, bool end_zero) {
  hipStream_t q = b->ctx->stream;
  if (!b->tables) {
    for (lvf_problem* p : b->probs) LVF_TRY(enqueue_iteration(p, end_zero));
    return LVF_OK;
  }
  const unsigned W = (unsigned)b->W;
  bool clean = true;
  for (lvf_problem* p : b->probs) { clean = clean && p->accum_clean; p->accum_clean = false; }
  if (!clean) hipLaunchKernelGGL(k_zero_table, dim3(512, W), dim3(kT), 0, q, b->zero.p);
  // LVF_BATCH_TRANSPOSE (bit mask, default 127: all; measured 8 windows 20.7k -> 22.1k it/s, 32 windows 28.9k -> 29.9k): which of lin (1), tf_reduce (2), prepare (4), Schur (8), block steps (16), tail (32), cost (64) are launched with blockIdx.x = window
  static const int tr = [] { const char* e = std::getenv("LVF_BATCH_TRANSPOSE"); return e ? std::atoi(e) : 127; }();
  const bool fits_y = std::max(std::max(b->g_lin, b->g_red), std::max(b->g_prep, b->g_ssp0)) <= 65535;
  if ((tr & 1) && fits_y) hipLaunchKernelGGL(k_lin_visual_bt, dim3(W, b->g_lin), dim3(kT), b->lds_lin, q, b->lin.p);
  else hipLaunchKernelGGL(k_lin_visual_b, dim3(b->g_lin, W), dim3(kT), b->lds_lin, q, b->lin.p);
  if (b->g_red > 0) {
    if ((tr & 2) && fits_y) hipLaunchKernelGGL(k_tf_reduce_bt, dim3(W, b->g_red), dim3(kT), b->lds_red, q, b->red.p);
    else hipLaunchKernelGGL(k_tf_reduce_b, dim3(b->g_red, W), dim3(kT), b->lds_red, q, b->red.p);
  }
  if ((tr & 4) && fits_y) hipLaunchKernelGGL(k_prepare_bt, dim3(W, b->g_prep), dim3(kT), b->lds_prep, q, b->prep.p);
  else hipLaunchKernelGGL(k_prepare_b, dim3(b->g_prep, W), dim3(kT), b->lds_prep, q, b->prep.p);
  if ((tr & 8) && fits_y) hipLaunchKernelGGL(k_schur_sp0_bt, dim3(W, b->g_ssp0), dim3(256), b->lds_ssp0, q, b->ssp0.p);
  else hipLaunchKernelGGL(k_schur_sp0_b, dim3(b->g_ssp0, W), dim3(256), b->lds_ssp0, q, b->ssp0.p);
  for (int lv = b->first_own_level; lv < b->max_levels; ++lv)
    if (b->g_sp[lv] > 0) hipLaunchKernelGGL(k_sp_eliminate_b, dim3(b->g_sp[lv], W), dim3(256), b->lds_sp[lv], q, b->sp[lv].p);
  for (int kb = 0; kb < b->max_nb; ++kb) {
    if (tr & 16) hipLaunchKernelGGL(k_chol_step_bt, dim3(W, chol_step_grid(b->max_nb, kb)), dim3(kCT), 0, q, b->chol.p, kb);
    else hipLaunchKernelGGL(k_chol_step_b, dim3(chol_step_grid(b->max_nb, kb), W), dim3(kCT), 0, q, b->chol.p, kb);
  }
  hipLaunchKernelGGL(k_chol_backsolve_b, dim3(1, W), dim3(kBT), b->lds_back, q, b->back.p);
  if ((tr & 32) && b->g_tail <= 65535) hipLaunchKernelGGL(k_step_tail_bt, dim3(W, b->g_tail), dim3(kT), b->lds_tail, q, b->tail.p);
  else hipLaunchKernelGGL(k_step_tail_b, dim3(b->g_tail, W), dim3(kT), b->lds_tail, q, b->tail.p);
  // (batchable windows always have visual blocks)
  if ((tr & 64) && b->g_cost <= 65535) hipLaunchKernelGGL(k_cost_decide_bt, dim3(W, b->g_cost), dim3(kT), 0, q, b->cost.p, b->dec.p, end_zero ? 1 : 0);
  else hipLaunchKernelGGL(k_cost_decide_b, dim3(b->g_cost, W), dim3(kT), 0, q, b->cost.p, b->dec.p, end_zero ? 1 : 0);
  LVF_HIP(hipGetLastError());
  for (lvf_problem* p : b->probs) { p->linearized = !end_zero; p->accum_clean = end_zero; }     // (batchable windows: the cost + decision launch clears them)
  return LVF_OK;
}

}  // namespace lvf

extern "C" {

void lvf_solver_options_default(lvf_solver_options* o) {
  if (!o) return;
  o->max_num_iterations = 50; o->max_solver_time_in_seconds = 0.0; o->huber_a = 1.0;
  o->initial_trust_region_radius = 1e4; o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8; o->min_relative_decrease = 1e-3;
}

int lvf_problem_create(lvf_ctx* ctx, lvf_state* st, lvf_batch* two_camera, lvf_batch* two_frame, lvf_batch* pose_only,
                       lvf_batch* imu, lvf_problem** out) {
  LVF_REQUIRE(ctx && st && out, "lvf_problem_create: null argument");
  LVF_REQUIRE(st->ctx == ctx, "lvf_problem_create: state belongs to another context");
  LVF_REQUIRE(!two_camera || two_camera->kind == LVF_K_TWO_CAMERA, "two_camera batch has the wrong kind");
  LVF_REQUIRE(!two_frame || two_frame->kind == LVF_K_TWO_FRAME, "two_frame batch has the wrong kind");
  LVF_REQUIRE(!pose_only || pose_only->kind == LVF_K_POSE_ONLY, "pose_only batch has the wrong kind");
  LVF_REQUIRE(!imu || imu->kind == LVF_K_IMU, "imu batch has the wrong kind");
  for (lvf_batch* b : {two_camera, two_frame, pose_only, imu})
    if (b) {
      LVF_REQUIRE(b->ctx == ctx, "batch belongs to another context");
      LVF_REQUIRE(b->min_n_kf <= st->n_kf && b->min_n_lm <= st->n_lm, "batch indices exceed the state (n_kf=%d n_lm=%d)", st->n_kf, st->n_lm);
    }
  LVF_REQUIRE(st->n_kf > 0, "lvf_problem_create: empty window");
  LVF_TRY(lvf::enter(ctx));
  auto* p = new lvf_problem();
  p->ctx = ctx; p->st = st; p->tc = two_camera; p->tf = two_frame; p->po = pose_only; p->imu = imu;
  const int rc = problem_configure(p);
  if (rc != LVF_OK) { delete p; return rc; }
  *out = p;
  return LVF_OK;
}
int lvf_problem_destroy(lvf_problem* p) { delete p; return LVF_OK; }

int lvf_problem_set_pose_priors(lvf_problem* p, lvf_batch* pose_priors) {
  LVF_REQUIRE(p, "lvf_problem_set_pose_priors: null problem");
  if (pose_priors) {
    LVF_REQUIRE(pose_priors->kind == LVF_K_POSE_PRIOR, "pose_priors batch has the wrong kind");
    LVF_REQUIRE(pose_priors->ctx == p->ctx, "batch belongs to another context");
    LVF_REQUIRE(pose_priors->min_n_kf <= p->n_kf, "pose-prior batch references keyframe %d but the window has %d", pose_priors->min_n_kf - 1, p->n_kf);
  }
  p->prior = pose_priors;
  p->linearized = false;
  p->chain_ready = false;
  return LVF_OK;
}

int lvf_problem_set_vbb_constant(lvf_problem* p, int kf, int v_constant, int ba_constant, int bg_constant) {
  LVF_REQUIRE(p, "null problem");
  LVF_REQUIRE(kf >= 0 && kf < p->n_kf, "keyframe %d out of range", kf);
  p->pose_const_h[kf] = (uint8_t)((p->pose_const_h[kf] & 1) | (v_constant ? 2 : 0) | (ba_constant ? 4 : 0) | (bg_constant ? 8 : 0));
  LVF_HIP(hipMemcpyAsync(p->pose_const.p, p->pose_const_h.data(), p->n_kf, hipMemcpyHostToDevice, p->ctx->stream));
  LVF_HIP(hipStreamSynchronize(p->ctx->stream));
  return LVF_OK;
}

int lvf_problem_set_pose_constant(lvf_problem* p, int kf, int is_constant) {
  LVF_REQUIRE(p, "null problem");
  LVF_REQUIRE(kf >= 0 && kf < p->n_kf, "keyframe %d out of range", kf);
  p->pose_const_h[kf] = (uint8_t)((p->pose_const_h[kf] & ~1) | (is_constant ? 1 : 0));
  LVF_HIP(hipMemcpyAsync(p->pose_const.p, p->pose_const_h.data(), p->n_kf, hipMemcpyHostToDevice, p->ctx->stream));
  LVF_HIP(hipStreamSynchronize(p->ctx->stream));
  return LVF_OK;
}

int lvf_problem_cost(lvf_problem* p, const lvf_solver_options* o, double* cost) {
  LVF_REQUIRE(p && o && cost, "lvf_problem_cost: null argument");
  LVF_TRY(lvf::enter(p->ctx));
  hipStream_t q = p->ctx->stream;
  LVF_HIP(hipMemsetAsync(p->scal.p, 0, SC_N * 8, q));
  LVF_TRY(enqueue_cost(p, state_ptrs(p->st), p->st, o->huber_a, p->scal.p + SC_COST));
  double hc[kStripes];
  LVF_HIP(hipMemcpyAsync(hc, p->scal.p + SC_COST, sizeof(hc), hipMemcpyDeviceToHost, q));
  // the stripes go back to zero: a following linearisation that trusts `accum_clean` (after a device-loop solve nothing else clears
  // SC_COST) adds its cost into them — solve -> cost -> solve would otherwise start from a doubled cost_before
  LVF_HIP(hipMemsetAsync(p->scal.p + SC_COST, 0, kStripes * 8, q));
  LVF_HIP(hipStreamSynchronize(q));
  *cost = stripe_sum(hc, 0);
  return LVF_OK;
}

// Problem::Evaluate's gradient: J^T r with the loss function's Corrector applied and pose blocks in tangent coordinates, at the current
// state — exactly what the linearisation accumulates.  gc [15 n_kf] in the reduced-system order (6 x n_kf pose tangents | 9 x n_kf (v, ba, bg)),
// gl [n_lm] (may be NULL) the inverse-depth entries.
int lvf_problem_stage_count(void) { return ST_N; }
const char* lvf_problem_stage_name(int stage) { return stage >= 0 && stage < ST_N ? kStageNames[stage] : ""; }

// `reps` LM iterations from the problem's current state (they ARE iterations: accepted steps move the state), HIP events on the
// library's stream between the stages; us[k] = average duration of stage k, launches[k] = kernel launches it consists of.
int lvf_problem_stage_times2(lvf_problem* p, const lvf_solver_options* o, double radius, int reps, double* us, double* spans_us, int* launches) {
  LVF_REQUIRE(p && o && us && reps > 0, "lvf_problem_stage_times: bad argument");
  LVF_TRY(lvf::enter(p->ctx));
  hipStream_t q = p->ctx->stream;
  if (!p->clk) {
    p->clk = new StageClock();
    for (auto& r : p->clk->ev) for (auto& e : r) LVF_HIP(hipEventCreate(&e));
    for (auto& r : p->clk->kstart) for (auto& e : r) LVF_HIP(hipEventCreate(&e));
    for (auto& r : p->clk->kstop) for (auto& e : r) LVF_HIP(hipEventCreate(&e));
    p->clk->kernel_events = true;
  }
  StageClock& k = *p->clk;
  for (int i = 0; i < ST_N; ++i) { us[i] = 0.0; k.launches[i] = 0; }
  std::vector<double> span(ST_N, 0.0), kern(ST_N, 0.0);
  std::vector<int> kcount(ST_N, 0);
  reps = std::min(reps, kClockReps);
  LmCtl c;
  ctl_from_options(o, radius, 2.0, reps + 2, false, &c);
  p->huber = o->huber_a;
  LVF_TRY(upload_ctl(p, c));
  LVF_TRY(enqueue_iteration(p, true));             // (un-timed: the timed iterations queue up behind it)
  int nk = 0;
  for (int r = 0; r < reps; ++r) {
    k.on = true; k.rep = r; k.nk = 0;
    const int rc = enqueue_iteration(p, true);
    k.on = false;
    nk = k.nk;
    LVF_TRY(rc);
  }
  LVF_HIP(hipStreamSynchronize(q));
  for (int r = 0; r < reps; ++r) {
    for (int i = 0; i < ST_N; ++i) {
      if (k.launches[i] == 0) continue;
      int prev = i;                                   // the event after the closest earlier stage that launched something (or event 0)
      while (prev > 0 && k.launches[prev - 1] == 0) --prev;
      float ms = 0.f;
      LVF_HIP(hipEventElapsedTime(&ms, k.ev[r][prev], k.ev[r][i + 1]));
      span[i] += 1e3 * (double)ms / reps;
    }
    for (int j = 0; j < nk; ++j) {                    // the kernels' own durations (dispatch timestamps)
      float ms = 0.f;
      LVF_HIP(hipEventElapsedTime(&ms, k.kstart[r][j], k.kstop[r][j]));
      kern[k.kstage[j]] += 1e3 * (double)ms / reps;
      if (r == 0) kcount[k.kstage[j]] += 1;
    }
  }
  // a stage whose launches all carried their own events reports the sum of the kernel durations; others (generic paths: stand-alone IMU /
  // prior passes) keep the between-stage span
  for (int i = 0; i < ST_N; ++i) us[i] = (k.launches[i] > 0 && kcount[i] == k.launches[i]) ? kern[i] : span[i];
  if (spans_us) for (int i = 0; i < ST_N; ++i) spans_us[i] = span[i];
  if (launches) for (int i = 0; i < ST_N; ++i) launches[i] = k.launches[i];
  return LVF_OK;
}
int lvf_problem_stage_times(lvf_problem* p, const lvf_solver_options* o, double radius, int reps, double* us, int* launches) {
  return lvf_problem_stage_times2(p, o, radius, reps, us, nullptr, launches);
}

int lvf_problem_gradient(lvf_problem* p, const lvf_solver_options* o, double* gc, double* gl) {
  LVF_REQUIRE(p && o && gc, "lvf_problem_gradient: null argument");
  LVF_TRY(lvf::enter(p->ctx));
  hipStream_t q = p->ctx->stream;
  LVF_TRY(enqueue_linearize(p, o->huber_a, false));
  const double* gr = p->gr.p;
  if (gl && p->n_lm && p->compact) {
    // atomic-free linearisation: a landmark's gradient entry is completed from its slot records by k_prepare (grt = gr + the slots)
    LmCtl c;
    ctl_from_options(o, o->initial_trust_region_radius, 2.0, 1, false, &c);
    LVF_TRY(upload_ctl(p, c));
    PrepArgs pa = p->chain->prep;
    pa.radius = &p->ctl.p->radius; pa.scal = nullptr; pa.done = nullptr;
    hipLaunchKernelGGL(k_prepare, dim3(pa.nblocks), dim3(kT), 0, q, pa);
    LVF_HIP(hipGetLastError());
    gr = p->grt.p;
  }
  LVF_HIP(hipMemcpyAsync(gc, p->gc.p, (size_t)p->d * 8, hipMemcpyDeviceToHost, q));
  if (gl && p->n_lm) LVF_HIP(hipMemcpyAsync(gl, gr, (size_t)p->n_lm * 8, hipMemcpyDeviceToHost, q));
  LVF_HIP(hipStreamSynchronize(q));
  return LVF_OK;
}

int lvf_problem_lm_iteration(lvf_problem* p, const lvf_solver_options* o, double* radius, double* decrease_factor,
                             double* cost_before, double* cost_after, int* accepted) {
  LVF_REQUIRE(p && o && radius && decrease_factor, "lvf_problem_lm_iteration: null argument");
  LVF_REQUIRE(*radius > 0.0 && *decrease_factor > 0.0, "radius and decrease_factor must be positive");
  LVF_TRY(lvf::enter(p->ctx));
  IterOut it;
  LVF_TRY(lm_iteration(p, o, radius, decrease_factor, &it));
  if (cost_before) *cost_before = it.cost_before;
  if (cost_after) *cost_after = it.cost_after;
  if (accepted) *accepted = it.accepted ? 1 : 0;
  return LVF_OK;
}

static void summary_from_ctl(const lvf_problem* p, const LmCtl& c, lvf_solver_summary* s) {
  std::memset(s, 0, sizeof(*s));
  s->num_residual_blocks = (p->tc ? p->tc->n : 0) + (p->tf ? p->tf->n : 0) + (p->po ? p->po->n : 0) + (p->imu ? p->imu->n : 0) + (p->prior ? p->prior->n : 0);
  s->initial_cost = c.initial_cost; s->final_cost = c.cost; s->num_iterations = c.iter; s->num_successful_steps = c.successes; s->termination = c.termination;
  s->num_unsuccessful_steps = c.rejected; s->termination_reason = c.why; s->hand_over_retries = p->handover_retries;
}


// The device LM loop: iterations are enqueued back to back, each closed on device (k_lm_decide); the host only watches a mirror of the
// control block to stop enqueueing once the loop has finished (an iteration enqueued after the end costs ~20 empty launches).
int lvf_problem_solve(lvf_problem* p, const lvf_solver_options* o, lvf_solver_summary* summary) { return lvf_problem_solve_then(p, o, summary, nullptr, nullptr); }

// lvf_problem_solve with a caller's launches enqueued BEHIND the last iteration and AHEAD of the wait that ends the solve (`tail(user)`,
// called once per pass of the hand-over retry loop, i.e. once in practice): the persistent window packs and copies its state back in
// the same stream wait instead of a second one (window.hip).
int lvf_problem_solve_then(lvf_problem* p, const lvf_solver_options* o, lvf_solver_summary* summary, int (*tail)(void*), void* user) {
  LVF_REQUIRE(p && o && summary, "lvf_problem_solve: null argument");
  LVF_TRY(lvf::enter(p->ctx));
  LmCtl c;
  ctl_from_options(o, o->initial_trust_region_radius, 2.0, o->max_num_iterations, true, &c);
  p->huber = o->huber_a;
  if (o->max_num_iterations <= 0) {          // nothing to iterate: report the cost at the start
    double cost = 0.0;
    LVF_TRY(lvf_problem_cost(p, o, &cost));
    c.initial_cost = c.cost = cost;
    summary_from_ctl(p, c, summary);
    if (tail) LVF_TRY(tail(user));
    return LVF_OK;
  }
  // a problem that lost its chained launches to a hand-over time-out gets them back after kUnchainedSolves solves (a time-out then simply sets it again)
  constexpr int kUnchainedSolves = 32;
  if (p->no_chain && ++p->unchained_solves > kUnchainedSolves && p->force_handover_timeouts == 0) { p->no_chain = false; p->unchained_solves = 0; p->chain_ready = false; }
  LVF_TRY(upload_ctl(p, c));
  const auto wall0 = std::chrono::steady_clock::now();
  bool timed_out = false;
  for (int first = 0;;) {
    for (int it = first; it < o->max_num_iterations; ++it) {
      LVF_TRY(enqueue_iteration(p, true));
      if (it >= first + 1) LVF_TRY(wait_for_iteration(p, it));          // iteration it-1 is closed; iteration `it` keeps the device busy meanwhile
      if (p->rec->done) break;
      if (o->max_solver_time_in_seconds > 0.0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() >= o->max_solver_time_in_seconds) { timed_out = true; break; }
    }
    // the caller's launches ride behind the last iteration — unless the host already KNOWS this pass ended in a hand-over time-out (the mirror
    // carries `why`): the state is not final then, the re-run's pass enqueues them.  (A time-out in the very last iteration enqueued is only
    // seen after the wait: the tail then runs twice, the second time on the final state.)
    const bool known_handover = p->rec->done && p->rec->why == LVF_WHY_HANDOVER && !p->no_chain;
    if (tail && !known_handover) LVF_TRY(tail(user));
    LVF_TRY(download_ctl(p, &c));
    if (!handover_pending(p, c)) break;
    LVF_TRY(rearm_after_handover(p, &c));     // a chained hand-over timed out: the loop goes on from the same point, un-chained
    first = c.iter;
  }
  p->last_radius = c.last_radius;
  summary_from_ctl(p, c, summary);
  if (timed_out && !c.done) summary->termination_reason = LVF_WHY_TIME;
  return LVF_OK;
}

int lvf_problem_reduced_dim(lvf_problem* p) { return p ? p->d : -1; }

// the DAMPED reduced system of the last lm_iteration, rebuilt (the factorisation overwrote S): S [d x d] symmetric, rhs [d]
int lvf_problem_download_reduced(lvf_problem* p, double* S, double* rhs) {
  LVF_REQUIRE(p && S && rhs, "lvf_problem_download_reduced: null argument");
  if (!p->linearized || !p->chain_ready) { set_error("no linearisation yet"); return LVF_ERR_STATE; }
  LVF_TRY(lvf::enter(p->ctx));
  hipStream_t q = p->ctx->stream;
  const size_t nS = (size_t)p->ld * p->ld;
  LVF_TRY(enqueue_reduced_system(p, &p->ctl.p->last_radius, false, false, nullptr));
  std::vector<double> h(nS);
  LVF_HIP(hipMemcpyAsync(h.data(), p->S.p, nS * 8, hipMemcpyDeviceToHost, q));
  LVF_HIP(hipStreamSynchronize(q));
  const int d = p->d, ld = p->ld;
  const std::vector<int>& pm = p->perm_h;          // natural unknown -> S row
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i; ++j) {
      const double v = h[(size_t)std::max(pm[i], pm[j]) * ld + std::min(pm[i], pm[j])];
      S[(size_t)i * d + j] = v; S[(size_t)j * d + i] = v;
    }
  for (int j = 0; j < d; ++j) rhs[j] = h[(size_t)p->aug * ld + pm[j]];
  return LVF_OK;
}

// ---- batch of windows
int lvf_problem_batch_create(lvf_ctx* ctx, lvf_problem* const* problems, int n, lvf_problem_batch** out) {
  LVF_REQUIRE(ctx && out && n >= 1 && problems, "lvf_problem_batch_create: bad arguments");
  for (int i = 0; i < n; ++i) {
    LVF_REQUIRE(problems[i], "lvf_problem_batch_create: problem %d is null", i);
    LVF_REQUIRE(problems[i]->ctx == ctx, "lvf_problem_batch_create: problem %d belongs to another context", i);
    for (int j = 0; j < i; ++j) LVF_REQUIRE(problems[j] != problems[i] && problems[j]->st != problems[i]->st, "lvf_problem_batch_create: windows %d and %d share state", j, i);
  }
  auto* b = new lvf_problem_batch();
  b->ctx = ctx; b->W = n; b->probs.assign(problems, problems + n);
  for (lvf_problem* p : b->probs) p->batches.push_back(b);
  *out = b;
  return LVF_OK;
}
// (members are only told that the batch is gone: nothing of theirs was changed by it)
int lvf_problem_batch_destroy(lvf_problem_batch* b) {
  if (b && !b->orphaned)
    for (lvf_problem* p : b->probs) p->batches.erase(std::remove(p->batches.begin(), p->batches.end(), b), p->batches.end());
  delete b;
  return LVF_OK;
}
int lvf_problem_batch_size(const lvf_problem_batch* b) { return b ? b->W : -1; }
// diagnostic (LVF_LM_HISTORY=1): {iteration, cost_before, cost_new, model, accepted, fail flag, radius, gradient max} of the passes of the last solve
int lvf_problem_debug_history(lvf_problem* p, double* out512) {
  LVF_REQUIRE(p && out512, "lvf_problem_debug_history: null argument");
  if (!p->dbg_hist.p) { set_error("lvf_problem_debug_history: LVF_LM_HISTORY is not set"); return LVF_ERR_STATE; }
  LVF_TRY(lvf::enter(p->ctx));
  LVF_HIP(hipMemcpyAsync(out512, p->dbg_hist.p, 512 * 8, hipMemcpyDeviceToHost, p->ctx->stream));
  LVF_HIP(hipStreamSynchronize(p->ctx->stream));
  return LVF_OK;
}
// test hook (see lvf.h)
int lvf_problem_debug_force_handover_timeout(lvf_problem* p, int n) {
  LVF_REQUIRE(p && n >= 0, "lvf_problem_debug_force_handover_timeout: bad argument");
  p->force_handover_timeouts = n; p->no_chain = false; p->chain_ready = false;
  return LVF_OK;
}
int lvf_problem_batch_uses_tables(lvf_problem_batch* b, const lvf_solver_options* o) {
  if (!b || !o || lvf::enter(b->ctx) != LVF_OK || batch_build_tables(b, o->huber_a) != LVF_OK) return -1;
  return b->tables ? 1 : 0;
}

// one LM iteration of every window (no tolerance tests); all arrays have one entry per window
int lvf_problem_batch_lm_iteration(lvf_problem_batch* b, const lvf_solver_options* o, double* radius, double* decrease_factor, double* cost_before,
                                   double* cost_after, int* accepted) {
  LVF_REQUIRE(b && o && radius && decrease_factor, "lvf_problem_batch_lm_iteration: null argument");
  LVF_TRY(lvf::enter(b->ctx));
  LVF_TRY(batch_build_tables(b, o->huber_a));
  for (int w = 0; w < b->W; ++w) {
    LVF_REQUIRE(radius[w] > 0.0 && decrease_factor[w] > 0.0, "radius and decrease_factor must be positive");
    LmCtl c;
    ctl_from_options(o, radius[w], decrease_factor[w], 1, false, &c);
    b->probs[w]->huber = o->huber_a;
    LVF_TRY(upload_ctl(b->probs[w], c));
  }
  LVF_TRY(batch_enqueue_iteration(b, false));
  {
    // windows whose chained hand-over timed out repeat the iteration un-chained (the others are done: their launches return at once)
    bool again = false;
    for (int w = 0; w < b->W; ++w) {
      LmCtl c;
      LVF_TRY(download_ctl(b->probs[w], &c));
      if (handover_pending(b->probs[w], c)) { LVF_TRY(rearm_after_handover(b->probs[w], &c)); again = true; }
    }
    if (again) { LVF_TRY(batch_build_tables(b, o->huber_a)); LVF_TRY(batch_enqueue_iteration(b, false)); }
  }
  for (int w = 0; w < b->W; ++w) {
    LmCtl c;
    LVF_TRY(download_ctl(b->probs[w], &c));
    b->probs[w]->last_radius = c.last_radius;
    radius[w] = c.radius; decrease_factor[w] = c.decrease;
    if (cost_before) cost_before[w] = c.cost_before;
    if (cost_after) cost_after[w] = c.cost_after;
    if (accepted) accepted[w] = c.accepted;
  }
  return LVF_OK;
}

int lvf_problem_batch_solve(lvf_problem_batch* b, const lvf_solver_options* o, lvf_solver_summary* summaries) {
  LVF_REQUIRE(b && o && summaries, "lvf_problem_batch_solve: null argument");
  LVF_TRY(lvf::enter(b->ctx));
  LVF_TRY(batch_build_tables(b, o->huber_a));
  if (o->max_num_iterations <= 0) {
    for (int w = 0; w < b->W; ++w) LVF_TRY(lvf_problem_solve(b->probs[w], o, &summaries[w]));
    return LVF_OK;
  }
  for (int w = 0; w < b->W; ++w) {
    LmCtl c;
    ctl_from_options(o, o->initial_trust_region_radius, 2.0, o->max_num_iterations, true, &c);
    b->probs[w]->huber = o->huber_a;
    LVF_TRY(upload_ctl(b->probs[w], c));
  }
  const auto wall0 = std::chrono::steady_clock::now();
  std::vector<LmCtl> cs((size_t)b->W);
  for (int round = 0;; ++round) {
    // (after a hand-over retry the windows stand at different iteration counts: the wait is for "one more than when this pass started")
    int base = o->max_num_iterations;
    for (int w = 0; w < b->W; ++w) if (!b->probs[w]->rec->done) base = std::min(base, (int)b->probs[w]->rec->iter);
    for (int it = base; it < o->max_num_iterations; ++it) {
      LVF_TRY(batch_enqueue_iteration(b, true));
      bool all_done = true;
      for (int w = 0; w < b->W; ++w) {
        if (it >= base + 1) LVF_TRY(wait_for_iteration(b->probs[w], it));
        all_done = all_done && b->probs[w]->rec->done;
      }
      if (all_done) break;
      if (o->max_solver_time_in_seconds > 0.0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() >= o->max_solver_time_in_seconds) break;
    }
    bool again = false;
    for (int w = 0; w < b->W; ++w) {
      LVF_TRY(download_ctl(b->probs[w], &cs[w]));
      if (handover_pending(b->probs[w], cs[w])) { LVF_TRY(rearm_after_handover(b->probs[w], &cs[w])); again = true; }
    }
    if (!again) break;
    LVF_TRY(batch_build_tables(b, o->huber_a));        // the re-armed windows' chains changed shape
  }
  for (int w = 0; w < b->W; ++w) {
    b->probs[w]->last_radius = cs[w].last_radius;
    summary_from_ctl(b->probs[w], cs[w], &summaries[w]);
  }
  return LVF_OK;
}

}  // extern "C"
