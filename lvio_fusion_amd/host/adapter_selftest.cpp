// adapter_selftest.cpp — drives the C++ host interface (include/lvf_ceres_adapter.hpp + host/adapt_problem.h) exactly the
// way the reference's callers do, on inputs written by tests/test_gpu_adapter.py, and dumps what came out.
//
//   adapter_selftest window <dir>   Backend::BuildProblem's loop shape (src/lvio_fusion/src/backend.cpp:96-183) over a
//                                   synthetic window, gpu::Evaluate, per-block CostFunction::Evaluate spot checks, then
//                                   adapt::Solve as Backend::Optimize issues it (:206-211)
//   adapter_selftest lidar <dir>    ScanToMapWithGround/Segmented's problem (association.cpp:270-384) + Mapping's solve
//                                   (mapping.cpp:153-163)
// Raw little-endian arrays: <dir>/<name>.f64 / .i32 in, <dir>/out_<name>.f64 out; a one-line JSON summary on stdout.
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "adapt_problem.h"

using namespace lvio_fusion;

template <typename T>
static std::vector<T> rd(const std::string& dir, const std::string& name) {
  std::ifstream f(dir + "/" + name, std::ios::binary | std::ios::ate);
  if (!f) { std::fprintf(stderr, "missing %s/%s\n", dir.c_str(), name.c_str()); std::exit(2); }
  const std::streamsize n = f.tellg();
  f.seekg(0);
  std::vector<T> v((size_t)n / sizeof(T));
  f.read(reinterpret_cast<char*>(v.data()), n);
  return v;
}
static void wr(const std::string& dir, const std::string& name, const std::vector<double>& v) {
  std::ofstream f(dir + "/" + name, std::ios::binary);
  f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(double)));
}
static void wri(const std::string& dir, const std::string& name, const std::vector<int>& v) {
  std::ofstream f(dir + "/" + name, std::ios::binary);
  f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(int)));
}
static lvf_camera cam_of(const std::vector<double>& c) {
  lvf_camera k;
  k.fx = c[0]; k.fy = c[1]; k.cx = c[2]; k.cy = c[3];
  for (int i = 0; i < 7; ++i) k.extrinsic[i] = c[4 + i];
  return k;
}

static int run_window(const std::string& dir) {
  auto meta = rd<int32_t>(dir, "meta.i32");   // n_kf, n_lm, max_iterations, weak_threshold, const_kf (-1 none)
  const int n_kf = meta[0], n_lm = meta[1], max_it = meta[2], weak_thr = meta[3], const_kf = meta[4];
  auto poses = rd<double>(dir, "poses.f64"), vel = rd<double>(dir, "vel.f64"), ba = rd<double>(dir, "ba.f64"), bg = rd<double>(dir, "bg.f64");
  auto invd = rd<double>(dir, "inv_depth.f64"), w_kf = rd<double>(dir, "w_kf.f64");
  const lvf_camera cam0 = cam_of(rd<double>(dir, "cam0.f64")), cam1 = cam_of(rd<double>(dir, "cam1.f64"));
  auto tc_l = rd<double>(dir, "tc_left_ob.f64"), tc_r = rd<double>(dir, "tc_right_ob.f64"); auto tc_lm = rd<int32_t>(dir, "tc_lm.i32"), tc_kf = rd<int32_t>(dir, "tc_kf.i32");
  auto tf_f = rd<double>(dir, "tf_first_ob.f64"), tf_o = rd<double>(dir, "tf_ob.f64");
  auto tf_lm = rd<int32_t>(dir, "tf_lm.i32"), tf_k1 = rd<int32_t>(dir, "tf_kf1.i32"), tf_k2 = rd<int32_t>(dir, "tf_kf2.i32");
  auto po_o = rd<double>(dir, "po_ob.f64"), po_pw = rd<double>(dir, "po_pw.f64"); auto po_kf = rd<int32_t>(dir, "po_kf.i32"), po_pi = rd<int32_t>(dir, "po_pw_idx.i32");
  auto pre = rd<double>(dir, "preint.f64"); auto imu_i = rd<int32_t>(dir, "imu_i.i32"), imu_j = rd<int32_t>(dir, "imu_j.i32");
  (void)n_lm;

  std::vector<ceres::CostFunction*> probe_cf;          // first block of each functor type, for Evaluate() spot checks
  std::vector<std::vector<double*>> probe_params;
  int n_prior = 0;
  // Backend::BuildProblem's loop (backend.cpp:96-183) into `problem`: one heap functor per block through the reference's Create signatures
  auto build_problem = [&](adapt::Problem& problem) {
  ceres::LossFunction* loss_function = new ceres::HuberLoss(1.0);
  ceres::LocalParameterization* local_parameterization =
      new ceres::ProductParameterization(new ceres::EigenQuaternionParameterization(), new ceres::IdentityParameterization(3));
  auto remember = [&](size_t want, ceres::CostFunction* cf, std::vector<double*> prm) {
    if (probe_cf.size() == want) { probe_cf.push_back(cf); probe_params.push_back(prm); }
  };
  size_t itc = 0, itf = 0, ipo = 0, iimu = 0;
  n_prior = 0;
  double* para_last_kf = nullptr;
  for (int k = 0; k < n_kf; ++k) {
    double* para_kf = &poses[7 * k];
    problem.AddParameterBlock(para_kf, 7, local_parameterization);
    // the frame's features in the order the python generator batched them (sorted by current keyframe within each type)
    while (itc < tc_kf.size() && tc_kf[itc] == k) {
      double* para_inv_depth = &invd[tc_lm[itc]];
      problem.AddParameterBlock(para_inv_depth, 1);
      auto* cf = gpu::TwoCameraReprojectionError::Create(&tc_l[2 * itc], &tc_r[2 * itc], cam0, cam1, 5 * w_kf[k]);
      problem.AddResidualBlock(ProblemType::Other, cf, loss_function, para_inv_depth);
      remember(0, cf, {para_inv_depth});
      ++itc;
    }
    while (ipo < po_kf.size() && po_kf[ipo] == k) {
      auto* cf = gpu::PoseOnlyReprojectionError::Create(&po_o[2 * ipo], &po_pw[3 * po_pi[ipo]], cam0, w_kf[k]);
      problem.AddResidualBlock(ProblemType::VisualError, cf, loss_function, para_kf);
      if (probe_cf.size() == 1) remember(1, cf, {para_kf});
      ++ipo;
    }
    while (itf < tf_k2.size() && tf_k2[itf] == k) {
      double* para_first_kf = &poses[7 * tf_k1[itf]];
      double* para_inv_depth = &invd[tf_lm[itf]];
      problem.AddParameterBlock(para_inv_depth, 1);
      auto* cf = gpu::TwoFrameReprojectionError::Create(&tf_f[2 * itf], &tf_o[2 * itf], cam0, cam1, w_kf[k]);
      problem.AddResidualBlock(ProblemType::VisualError, cf, loss_function, para_inv_depth, para_first_kf, para_kf);
      if (probe_cf.size() == 2) remember(2, cf, {para_inv_depth, para_first_kf, para_kf});
      ++itf;
    }
    if (!imu_j.empty()) {
      problem.AddParameterBlock(&vel[3 * k], 3); problem.AddParameterBlock(&ba[3 * k], 3); problem.AddParameterBlock(&bg[3 * k], 3);
      while (iimu < imu_j.size() && imu_j[iimu] == k) {
        const int i = imu_i[iimu];
        lvf_preint P;
        std::memcpy(&P, &pre[467 * iimu], sizeof(P));
        auto* cf = gpu::ImuError::Create(P);
        problem.AddResidualBlock(ProblemType::ImuError, cf, nullptr, &poses[7 * i], &vel[3 * i], &ba[3 * i], &bg[3 * i], para_kf, &vel[3 * k], &ba[3 * k], &bg[3 * k]);
        if (probe_cf.size() == 3) remember(3, cf, {&poses[7 * i], &vel[3 * i], &ba[3 * i], &bg[3 * i], para_kf, &vel[3 * k], &ba[3 * k], &bg[3 * k]});
        ++iimu;
      }
    }
    // weak-constraint check, backend.cpp:164-178 (threshold is a test knob; the reference uses 20)
    auto num_types = problem.GetTypes(para_kf);
    if (!num_types[ProblemType::ImuError] && num_types[ProblemType::VisualError] < weak_thr) {
      if (para_last_kf) {
        auto* cf = gpu::PoseGraphError::Create(para_last_kf, para_kf, 100, 0);
        problem.AddResidualBlock(ProblemType::Other, cf, nullptr, para_last_kf, para_kf);
      } else {
        auto* cf = gpu::PoseError::Create(para_kf, 100, 0);
        problem.AddResidualBlock(ProblemType::Other, cf, nullptr, para_kf);
      }
      ++n_prior;
    }
    para_last_kf = para_kf;
  }
  if (const_kf >= 0) problem.SetParameterBlockConstant(&poses[7 * const_kf]);
  };
  adapt::Problem problem;
  build_problem(problem);

  if (std::getenv("LVF_SELFTEST_HOSTONLY")) {      // host-side cost of the Ceres-surface path (no GPU needed): classify the blocks
    for (int rep = 0; rep < 3; ++rep) {
      ceres::Solver::Summary sm;
      const gpu::detail::Fail fail{&sm};
      std::vector<gpu::detail::BlockView> blocks;
      std::vector<double*> flat;
      const auto t0 = std::chrono::steady_clock::now();
      const bool ok1 = gpu::detail::collect(&problem, &blocks, &flat, fail);
      const auto t1 = std::chrono::steady_clock::now();
      gpu::detail::Window w;
      const bool ok2 = ok1 && gpu::detail::build_window(&problem, blocks, &w, fail);
      const auto t2 = std::chrono::steady_clock::now();
      std::fprintf(stderr, "host only: collect %.3f ms, build_window %.3f ms (%zu blocks, ok %d)\n", 1e3 * std::chrono::duration<double>(t1 - t0).count(),
                   1e3 * std::chrono::duration<double>(t2 - t1).count(), blocks.size(), (int)ok2);
      if (rep == 2) {
        // the window the recorder assembled while the blocks were being added must be the window the walk over the finished problem builds
        ceres::Solver::Summary sm2;
        const gpu::detail::Fail fail2{&sm2};
        gpu::detail::Window& r = problem.recorder.window();
        const bool usable = problem.recorder.usable(&problem);
        const bool ok3 = usable && gpu::detail::finish_window(&problem, &r, &problem.recorder.constant_blocks(), fail2);
        auto eqv = [](const auto& a, const auto& b) { return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(a[0])) == 0); };
        const bool same = ok2 && ok3 && eqv(r.pose_ptr, w.pose_ptr) && eqv(r.lm_ptr, w.lm_ptr) && eqv(r.v_ptr, w.v_ptr) && eqv(r.ba_ptr, w.ba_ptr) && eqv(r.bg_ptr, w.bg_ptr) &&
                          eqv(r.w_kf, w.w_kf) && eqv(r.pose_const, w.pose_const) && eqv(r.vbb_const, w.vbb_const) && eqv(r.tc_l, w.tc_l) && eqv(r.tc_r, w.tc_r) && eqv(r.tc_lm, w.tc_lm) && eqv(r.tc_w, w.tc_w) &&
                          eqv(r.tf_f, w.tf_f) && eqv(r.tf_o, w.tf_o) && eqv(r.tf_lm, w.tf_lm) && eqv(r.tf_k1, w.tf_k1) && eqv(r.tf_k2, w.tf_k2) && eqv(r.po_o, w.po_o) &&
                          eqv(r.po_pw, w.po_pw) && eqv(r.po_kf, w.po_kf) && eqv(r.po_pi, w.po_pi) && eqv(r.imu_pre, w.imu_pre) && eqv(r.imu_i, w.imu_i) && eqv(r.imu_j, w.imu_j) &&
                          eqv(r.pr_a, w.pr_a) && eqv(r.pr_b, w.pr_b) && eqv(r.pr_t, w.pr_t) && eqv(r.pr_w, w.pr_w) && eqv(r.pr_v, w.pr_v) && eqv(r.order_kind, w.order_kind) &&
                          eqv(r.order_idx, w.order_idx) && r.huber == w.huber && r.have_left == w.have_left && r.have_right == w.have_right &&
                          (!w.have_left || gpu::detail::same_cam(r.left, w.left)) && (!w.have_right || gpu::detail::same_cam(r.right, w.right));
        std::printf("{\"host_only\": 1, \"blocks\": %zu, \"walk_ok\": %d, \"recorder_usable\": %d, \"recorded_equals_walk\": %d, \"n_kf\": %zu, \"n_lm\": %zu, \"n_prior\": %zu}\n",
                    blocks.size(), (int)ok2, (int)usable, (int)same, w.pose_ptr.size(), w.lm_ptr.size(), w.pr_b.size());
      }
    }
    return 0;
  }

  // batched Problem::Evaluate at the initial point
  double cost0 = -1.0;
  std::vector<double> residuals;
  std::string err;
  if (!gpu::Evaluate(&problem, &cost0, &residuals, &err)) { std::fprintf(stderr, "Evaluate failed: %s\n", err.c_str()); return 1; }
  wr(dir, "out_residuals.f64", residuals);
  // the upstream-shaped call: robustified residuals, gradient and the CRS Jacobian in local coordinates
  int crs_rows = 0, crs_cols = 0;
  {
    ceres::Problem::EvaluateOptions eo;
    double cost1 = -1.0;
    std::vector<double> res1, grad;
    ceres::CRSMatrix jac;
    if (!gpu::Evaluate(&problem, eo, &cost1, &res1, &grad, &jac, &err)) { std::fprintf(stderr, "Evaluate (CRS) failed: %s\n", err.c_str()); return 1; }
    // (sums of ~10^5 blocks through atomics: the order, hence the last bits, differs from launch to launch)
    if (std::fabs(cost1 - cost0) > 1e-11 * std::fabs(cost0)) { std::fprintf(stderr, "Evaluate: cost differs between the two forms (%.17g vs %.17g)\n", cost0, cost1); return 1; }
    wr(dir, "out_eval_residuals.f64", res1); wr(dir, "out_eval_gradient.f64", grad); wr(dir, "out_crs_values.f64", jac.values);
    wri(dir, "out_crs_rows.i32", jac.rows); wri(dir, "out_crs_cols.i32", jac.cols);
    crs_rows = jac.num_rows; crs_cols = jac.num_cols;
    // which caller array every column block is: (kind, index, first column) with kind 0 pose, 1 vel, 2 ba, 3 bg, 4 inverse depth
    std::vector<double*> pbs;
    problem.GetParameterBlocks(&pbs);
    std::vector<int> colmap;
    int col = 0;
    auto inside = [](const std::vector<double>& a, const double* p) { return !a.empty() && p >= a.data() && p < a.data() + a.size(); };
    for (double* pb : pbs) {
      int kind = -1, index = -1, local = 0;
      if (inside(poses, pb)) { kind = 0; index = (int)(pb - poses.data()) / 7; local = 6; }
      else if (inside(vel, pb)) { kind = 1; index = (int)(pb - vel.data()) / 3; local = 3; }
      else if (inside(ba, pb)) { kind = 2; index = (int)(pb - ba.data()) / 3; local = 3; }
      else if (inside(bg, pb)) { kind = 3; index = (int)(pb - bg.data()) / 3; local = 3; }
      else if (inside(invd, pb)) { kind = 4; index = (int)(pb - invd.data()); local = 1; }
      colmap.push_back(kind); colmap.push_back(index); colmap.push_back(col);
      col += local;
    }
    if (col != crs_cols) { std::fprintf(stderr, "Evaluate: %d columns, expected %d\n", crs_cols, col); return 1; }
    wri(dir, "out_crs_colmap.i32", colmap);
    // a second call restricted to two parameter blocks, in reverse order: columns follow options.parameter_blocks, the rest is held constant
    ceres::Problem::EvaluateOptions sub;
    sub.parameter_blocks = {&poses[7 * (n_kf - 1)], &poses[7 * 1]};
    sub.apply_loss_function = false;
    std::vector<double> grad2;
    ceres::CRSMatrix jac2;
    if (!gpu::Evaluate(&problem, sub, nullptr, nullptr, &grad2, &jac2, &err)) { std::fprintf(stderr, "Evaluate (subset) failed: %s\n", err.c_str()); return 1; }
    if (jac2.num_cols != 12 || (int)grad2.size() != 12 || jac2.num_rows != jac.num_rows) { std::fprintf(stderr, "Evaluate (subset): bad shape\n"); return 1; }
    wr(dir, "out_sub_gradient.f64", grad2); wr(dir, "out_sub_values.f64", jac2.values); wri(dir, "out_sub_rows.i32", jac2.rows); wri(dir, "out_sub_cols.i32", jac2.cols);
  }

  // per-block CostFunction::Evaluate (Ceres calling convention) on the first block of each type
  std::vector<double> probe_out;
  for (size_t b = 0; b < probe_cf.size(); ++b) {
    const ceres::CostFunction* cf = probe_cf[b];
    const int R = cf->num_residuals();
    const auto& sizes = cf->parameter_block_sizes();
    std::vector<double> r(R);
    std::vector<std::vector<double>> J(sizes.size());
    std::vector<double*> Jp(sizes.size());
    for (size_t k = 0; k < sizes.size(); ++k) { J[k].assign((size_t)R * sizes[k], 0.0); Jp[k] = J[k].data(); }
    if (sizes.size() > 1) Jp[1] = nullptr;   // a NULL jacobians[i] must be honoured (constant block)
    if (!cf->Evaluate(probe_params[b].data(), r.data(), Jp.data())) { std::fprintf(stderr, "CostFunction::Evaluate failed on probe %zu: %s\n", b, lvf_last_error()); return 1; }
    std::vector<double> r2(R);
    if (!cf->Evaluate(probe_params[b].data(), r2.data(), nullptr) || r2 != r) { std::fprintf(stderr, "residual-only Evaluate differs on probe %zu\n", b); return 1; }
    probe_out.insert(probe_out.end(), r.begin(), r.end());
    for (size_t k = 0; k < sizes.size(); ++k) probe_out.insert(probe_out.end(), J[k].begin(), J[k].end());
  }
  wr(dir, "out_probe.f64", probe_out);

  ceres::Solver::Options options;
  options.linear_solver_type = ceres::SPARSE_SCHUR;
  options.max_num_iterations = max_it;
  options.num_threads = 4;
  ceres::Solver::Summary summary;
  adapt::Solve(options, &problem, &summary);
  std::fprintf(stderr, "adapt::Solve: %.3f ms for %d residual blocks (%d + %d LM steps)\n", 1e3 * summary.total_time_in_seconds, summary.num_residual_blocks_reduced,
               summary.num_successful_steps, summary.num_unsuccessful_steps);
  wr(dir, "out_poses.f64", poses); wr(dir, "out_vel.f64", vel); wr(dir, "out_ba.f64", ba); wr(dir, "out_bg.f64", bg); wr(dir, "out_inv_depth.f64", invd);
  if (std::getenv("LVF_SELFTEST_REPEAT")) {       // warm second call on the same thread (device allocator primed), outputs already written
    std::vector<double> p2 = poses, v2 = vel, a2 = ba, g2 = bg, d2 = invd;
    ceres::Solver::Summary s2;
    adapt::Solve(options, &problem, &s2);
    std::fprintf(stderr, "adapt::Solve (warm repeat): %.3f ms = preprocess %.3f + minimize %.3f + postprocess %.3f\n", 1e3 * s2.total_time_in_seconds,
                 1e3 * s2.preprocessor_time_in_seconds, 1e3 * s2.minimizer_time_in_seconds, 1e3 * s2.postprocessor_time_in_seconds);
    poses = p2; vel = v2; ba = a2; bg = g2; invd = d2;
  }
  if (std::getenv("LVF_SELFTEST_TICK")) {
    // One backend tick through the Ceres surface as the reference runs it (backend.cpp:203-211): a fresh adapt::Problem, BuildProblem's loop
    // (one X::Create heap functor + AddResidualBlock per block), adapt::Solve, and the Problem's destructor (it owns the functors).  The
    // state is restored between repeats; the ceres::Problem here is include/lvf_ceres_compat.h's stand-in (Ceres itself is not in the image).
    std::vector<double> p2 = poses, v2 = vel, a2 = ba, g2 = bg, d2 = invd;
    std::vector<std::array<double, 4>> tms;
    for (int rep = 0; rep < 5; ++rep) {
      poses = p2; vel = v2; ba = a2; bg = g2; invd = d2;
      const auto t0 = std::chrono::steady_clock::now();
      auto t1 = t0, t2 = t0;
      {
        adapt::Problem tick;
        build_problem(tick);
        t1 = std::chrono::steady_clock::now();
        ceres::Solver::Summary st;
        adapt::Solve(options, &tick, &st);
        t2 = std::chrono::steady_clock::now();
      }
      const auto t3 = std::chrono::steady_clock::now();
      auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return 1e3 * std::chrono::duration<double>(b - a).count(); };
      tms.push_back({ms(t0, t3), ms(t0, t1), ms(t1, t2), ms(t2, t3)});
    }
    std::sort(tms.begin(), tms.end());
    const auto& m = tms[tms.size() / 2];
    std::fprintf(stderr, "ceres-surface tick (median of 5): %.3f ms = build (Create + AddResidualBlock x %d) %.3f + adapt::Solve %.3f + ~Problem %.3f\n", m[0],
                 summary.num_residual_blocks_reduced, m[1], m[2], m[3]);
    poses = p2; vel = v2; ba = a2; bg = g2; invd = d2;
  }
  std::printf("{\"ok\": %d, \"message\": \"%s\", \"cost0\": %.17g, \"initial_cost\": %.17g, \"final_cost\": %.17g, \"successful\": %d, \"unsuccessful\": %d, "
              "\"num_residual_blocks\": %d, \"num_frames\": %d, \"n_prior\": %d, \"n_probe\": %zu, \"termination\": %d, \"crs_rows\": %d, \"crs_cols\": %d}\n",
              summary.termination_type != ceres::FAILURE, summary.message.c_str(), cost0, summary.initial_cost, summary.final_cost, summary.num_successful_steps,
              summary.num_unsuccessful_steps, summary.num_residual_blocks_reduced, problem.num_frames, n_prior, probe_cf.size(), (int)summary.termination_type, crs_rows,
              crs_cols);
  return summary.termination_type == ceres::FAILURE ? 1 : 0;
}

static int run_lidar(const std::string& dir) {
  auto meta = rd<int32_t>(dir, "meta.i32");   // mode, use_prior
  const int mode = meta[0], use_prior = meta[1];
  auto sc = rd<double>(dir, "scalars.f64");   // weight, huber_a (<=0: TrivialLoss), prior_weight
  auto p = rd<double>(dir, "p.f64"), pa = rd<double>(dir, "pa.f64"), pb = rd<double>(dir, "pb.f64"), pc = rd<double>(dir, "pc.f64");
  auto Twc1 = rd<double>(dir, "map_pose.f64"), rpy = rd<double>(dir, "rpyxyz.f64");
  const size_t n = p.size() / 3;
  double* rpyxyz = rpy.data();   // the live array of mapping.cpp:153
  adapt::Problem problem;
  ceres::LossFunction* loss = sc[1] > 0.0 ? static_cast<ceres::LossFunction*>(new ceres::HuberLoss(sc[1])) : new ceres::TrivialLoss();
  // association.cpp:273-275 / :331-333
  double *q0, *q1, *q2;
  if (mode == 0) { q0 = rpyxyz + 1; q1 = rpyxyz + 2; q2 = rpyxyz + 5; } else { q0 = rpyxyz + 0; q1 = rpyxyz + 3; q2 = rpyxyz + 4; }
  problem.AddParameterBlock(q0, 1); problem.AddParameterBlock(q1, 1); problem.AddParameterBlock(q2, 1);
  ceres::CostFunction* probe = nullptr;
  for (size_t i = 0; i < n; ++i) {
    ceres::CostFunction* cf = mode == 0 ? gpu::LidarPlaneErrorRPZ::Create(&p[3 * i], &pa[3 * i], &pb[3 * i], &pc[3 * i], Twc1.data(), rpyxyz, sc[0])
                                        : gpu::LidarPlaneErrorYXY::Create(&p[3 * i], &pa[3 * i], &pb[3 * i], &pc[3 * i], Twc1.data(), rpyxyz, sc[0]);
    problem.AddResidualBlock(ProblemType::LidarError, cf, loss, q0, q1, q2);
    if (!probe) probe = cf;
  }
  ceres::CostFunction* prior = nullptr;
  if (use_prior) {
    prior = mode == 0 ? gpu::PoseErrorRPZ::Create(rpyxyz, sc[2]) : gpu::PoseErrorYXY::Create(rpyxyz, sc[2]);
    problem.AddResidualBlock(ProblemType::PoseError, prior, nullptr, q0, q1, q2);
  }
  std::vector<double> probe_out;
  double* prm[3] = {q0, q1, q2};
  if (probe) {
    double r, j0, j1, j2; double* J[3] = {&j0, &j1, &j2};
    if (!probe->Evaluate(prm, &r, J)) { std::fprintf(stderr, "lidar Evaluate failed: %s\n", lvf_last_error()); return 1; }
    probe_out = {r, j0, j1, j2};
  }
  if (prior) {
    double x[3] = {*q0 + 0.01, *q1 - 0.02, *q2 + 0.03};
    double* xp[3] = {&x[0], &x[1], &x[2]};
    double r[3], j0[3], j1[3], j2[3]; double* J[3] = {j0, j1, j2};
    if (!prior->Evaluate(xp, r, J)) { std::fprintf(stderr, "prior Evaluate failed: %s\n", lvf_last_error()); return 1; }
    for (double v : r) probe_out.push_back(v);
    for (double* jj : J) for (int k = 0; k < 3; ++k) probe_out.push_back(jj[k]);
  }
  wr(dir, "out_probe.f64", probe_out);
  ceres::Solver::Options options;       // mapping.cpp:159-161
  options.linear_solver_type = ceres::DENSE_QR;
  options.max_num_iterations = 4;
  options.num_threads = 4;
  ceres::Solver::Summary summary;
  adapt::Solve(options, &problem, &summary);
  wr(dir, "out_rpyxyz.f64", rpy);
  std::printf("{\"ok\": %d, \"message\": \"%s\", \"initial_cost\": %.17g, \"final_cost\": %.17g, \"successful\": %d, \"unsuccessful\": %d, \"num_residual_blocks\": %d}\n",
              summary.termination_type != ceres::FAILURE, summary.message.c_str(), summary.initial_cost, summary.final_cost, summary.num_successful_steps,
              summary.num_unsuccessful_steps, summary.num_residual_blocks_reduced);
  return summary.termination_type == ceres::FAILURE ? 1 : 0;
}

// PoseGraph::BuildProblem + Optimize's solve (src/lvio_fusion/src/pose_graph.cpp:163-208): old and start frames constant, one pose
// per section start, PoseGraphError between consecutive poses (targets = the CURRENT relative poses, i.e. before the loop
// correction moved the start frame), RError on every section pose, default solver options.
static int run_posegraph(const std::string& dir) {
  auto poses = rd<double>(dir, "poses.f64");          // [n][7]: old frame, section starts ..., start frame (already relocated)
  auto start_before = rd<double>(dir, "start_before.f64");   // the start frame's pose before relocation (defines the last edge)
  const int n = (int)(poses.size() / 7);
  adapt::Problem problem;
  ceres::LocalParameterization* lp = new ceres::ProductParameterization(new ceres::EigenQuaternionParameterization(), new ceres::IdentityParameterization(3));
  double* para_old = &poses[0];
  double* para_start = &poses[7 * (n - 1)];
  problem.AddParameterBlock(para_old, 7, lp); problem.SetParameterBlockConstant(para_old);
  problem.AddParameterBlock(para_start, 7, lp); problem.SetParameterBlockConstant(para_start);
  double* last = para_old;
  for (int k = 1; k < n - 1; ++k) {
    double* para = &poses[7 * k];
    problem.AddParameterBlock(para, 7, lp);
    problem.AddResidualBlock(ProblemType::Other, gpu::PoseGraphError::Create(last, para), nullptr, last, para);
    problem.AddResidualBlock(ProblemType::Other, gpu::RError::Create(para), nullptr, para);
    last = para;
  }
  problem.AddResidualBlock(ProblemType::Other, gpu::PoseGraphError::Create(last, start_before.data()), nullptr, last, para_start);
  // RError::Evaluate spot check on a perturbed pose
  std::vector<double> probe;
  {
    ceres::CostFunction* cf = gpu::RError::Create(&poses[7], 3.0);
    double x[7]; for (int k = 0; k < 7; ++k) x[k] = poses[7 + k] + 0.01 * (k + 1);
    double* xp[1] = {x}; double r[4], J[28]; double* Jp[1] = {J};
    if (!cf->Evaluate(xp, r, Jp)) { std::fprintf(stderr, "RError Evaluate failed: %s\n", lvf_last_error()); return 1; }
    probe.insert(probe.end(), r, r + 4); probe.insert(probe.end(), J, J + 28);
    delete cf;
  }
  wr(dir, "out_probe.f64", probe);
  ceres::Solver::Options options;
  options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;
  ceres::Solver::Summary summary;
  adapt::Solve(options, &problem, &summary);
  wr(dir, "out_poses.f64", poses);
  std::printf("{\"ok\": %d, \"message\": \"%s\", \"initial_cost\": %.17g, \"final_cost\": %.17g, \"successful\": %d, \"unsuccessful\": %d, \"num_residual_blocks\": %d}\n",
              summary.termination_type != ceres::FAILURE, summary.message.c_str(), summary.initial_cost, summary.final_cost, summary.num_successful_steps,
              summary.num_unsuccessful_steps, summary.num_residual_blocks_reduced);
  return summary.termination_type == ceres::FAILURE ? 1 : 0;
}

// Relocator::UpdateNewSubmap's rotation solve (src/lvio_fusion/src/relocator.cpp:251-268), block for block: one quaternion parameter block under
// EigenQuaternionParameterization (identity to begin with), one RelocateRError per keyframe of the new sub-map, no loss, default options.
// gpu::Solve hands the problem to lvf_relocate_rotation_solve (one device launch); CostFunction::Evaluate is the batched functor on a batch of one.
static int run_relocate(const std::string& dir) {
  auto rel = rd<double>(dir, "relocated.f64"), un = rd<double>(dir, "unrelocated.f64");      // [n][7] each
  const int n = (int)(rel.size() / 7);
  double q[4] = {0.0, 0.0, 0.0, 1.0};
  std::vector<double> probe;
  int soft_fail = 0;
  {
    adapt::Problem problem;
    problem.AddParameterBlock(q, 4, new ceres::EigenQuaternionParameterization());
    for (int i = 0; i < n; ++i) problem.AddResidualBlock(ProblemType::Other, gpu::RelocateRError::Create(&rel[7 * i], &un[7 * i]), nullptr, q);
    {   // Evaluate spot check at a non-unit quaternion (the functor does not normalise: pose_error.hpp:199-212)
      ceres::CostFunction* cf = gpu::RelocateRError::Create(&rel[0], &un[0]);
      double x[4] = {0.02, -0.01, 0.03, 0.98}; double* xp[1] = {x}; double r[7], J[28]; double* Jp[1] = {J};
      if (!cf->Evaluate(xp, r, Jp)) { std::fprintf(stderr, "RelocateRError Evaluate failed: %s\n", lvf_last_error()); return 1; }
      double r2[7];
      if (!cf->Evaluate(xp, r2, nullptr) || std::memcmp(r, r2, sizeof(r)) != 0) { std::fprintf(stderr, "residual-only Evaluate differs\n"); return 1; }
      probe.insert(probe.end(), r, r + 7); probe.insert(probe.end(), J, J + 28);
      delete cf;
    }
    ceres::Solver::Options options;
    options.linear_solver_type = ceres::DENSE_QR;
    ceres::Solver::Summary summary;
    adapt::Solve(options, &problem, &summary);
    if (summary.termination_type == ceres::FAILURE) { std::fprintf(stderr, "rotation solve failed: %s\n", summary.message.c_str()); return 1; }
    probe.push_back(summary.initial_cost); probe.push_back(summary.final_cost); probe.push_back(summary.num_successful_steps); probe.push_back(summary.num_residual_blocks_reduced);
  }
  {   // a RelocateRError block mixed with another cost function is refused softly, the quaternion untouched
    adapt::Problem problem;
    double q2[4] = {0.0, 0.0, 0.0, 1.0}, pose[7] = {0, 0, 0, 1, 0, 0, 0};
    problem.AddParameterBlock(q2, 4, new ceres::EigenQuaternionParameterization());
    problem.AddParameterBlock(pose, 7, new ceres::ProductParameterization(new ceres::EigenQuaternionParameterization(), new ceres::IdentityParameterization(3)));
    problem.AddResidualBlock(ProblemType::Other, gpu::RelocateRError::Create(&rel[0], &un[0]), nullptr, q2);
    problem.AddResidualBlock(ProblemType::Other, gpu::PoseError::Create(pose), nullptr, pose);
    ceres::Solver::Options options; ceres::Solver::Summary summary;
    adapt::Solve(options, &problem, &summary);
    soft_fail = summary.termination_type == ceres::FAILURE && q2[0] == 0.0 && q2[1] == 0.0 && q2[2] == 0.0 && q2[3] == 1.0;
  }
  wr(dir, "out_probe.f64", probe);
  wr(dir, "out_q.f64", std::vector<double>(q, q + 4));
  std::printf("{\"ok\": 1, \"n\": %d, \"mixed_problem_refused_softly\": %d}\n", n, soft_fail);
  return 0;
}

// a cost function the adapter does not own must be refused softly (parameters untouched, FAILURE reported)
struct Foreign : ceres::SizedCostFunction<1, 1> {
  bool Evaluate(double const* const* p, double* r, double** J) const override { r[0] = p[0][0]; if (J && J[0]) J[0][0] = 1; return true; }
};
static int run_foreign() {
  adapt::Problem problem;
  double x = 3.0;
  problem.AddParameterBlock(&x, 1);
  problem.AddResidualBlock(ProblemType::Other, new Foreign(), nullptr, &x);
  ceres::Solver::Options options; ceres::Solver::Summary summary;
  adapt::Solve(options, &problem, &summary);
  std::printf("{\"ok\": %d, \"x\": %.17g, \"message\": \"%s\"}\n", summary.termination_type != ceres::FAILURE, x, summary.message.c_str());
  return 0;
}

// Environment::Optimize's visual + IMU problem (src/environment.cpp:18-75), block for block: ONE free pose (the environment's frame),
// PoseOnlyReprojectionError blocks on it under HuberLoss(1.0), and one ImuError to the previous keyframe whose pose, velocity and biases —
// like the frame's own velocity and biases — are registered and then held constant (:54-68).  The RL weight tuner calls this once per
// environment step (8 environments while training, 100 while testing: rl_fusion/td3.py:44-45).
static int run_environment(const std::string& dir) {
  auto meta = rd<int32_t>(dir, "meta.i32");   // n_blocks, max_iterations, with_imu
  const int n = meta[0], max_it = meta[1], with_imu = meta[2];
  auto pose = rd<double>(dir, "pose.f64"), last_pose = rd<double>(dir, "last_pose.f64");        // [7] each
  auto vbb = rd<double>(dir, "vbb.f64");                                                          // v, ba, bg of the frame, then of the last frame: [18]
  auto ob = rd<double>(dir, "ob.f64"), pw = rd<double>(dir, "pw.f64");
  const double weight = rd<double>(dir, "weight.f64")[0];
  const lvf_camera cam0 = cam_of(rd<double>(dir, "cam0.f64"));
  auto pre = rd<double>(dir, "preint.f64");
  adapt::Problem problem;
  ceres::LocalParameterization* local_parameterization =
      new ceres::ProductParameterization(new ceres::EigenQuaternionParameterization(), new ceres::IdentityParameterization(3));
  double* para = pose.data();
  problem.AddParameterBlock(para, 7, local_parameterization);
  ceres::LossFunction* loss_function = new ceres::HuberLoss(1.0);
  for (int i = 0; i < n; ++i)
    problem.AddResidualBlock(ProblemType::VisualError, gpu::PoseOnlyReprojectionError::Create(&ob[2 * i], &pw[3 * i], cam0, weight), loss_function, para);
  if (with_imu) {
    double *para_v = &vbb[0], *para_ba = &vbb[3], *para_bg = &vbb[6], *para_last_kf = last_pose.data(), *para_v_last = &vbb[9], *para_ba_last = &vbb[12], *para_bg_last = &vbb[15];
    problem.AddParameterBlock(para_v, 3); problem.AddParameterBlock(para_ba, 3); problem.AddParameterBlock(para_bg, 3);
    problem.AddParameterBlock(para_last_kf, 7);
    problem.AddParameterBlock(para_v_last, 3); problem.AddParameterBlock(para_ba_last, 3); problem.AddParameterBlock(para_bg_last, 3);
    problem.SetParameterBlockConstant(para_v); problem.SetParameterBlockConstant(para_ba); problem.SetParameterBlockConstant(para_bg);
    problem.SetParameterBlockConstant(para_last_kf);
    problem.SetParameterBlockConstant(para_v_last); problem.SetParameterBlockConstant(para_ba_last); problem.SetParameterBlockConstant(para_bg_last);
    lvf_preint P;
    std::memcpy(&P, pre.data(), sizeof(P));
    problem.AddResidualBlock(ProblemType::ImuError, gpu::ImuError::Create(P), nullptr, para_last_kf, para_v_last, para_ba_last, para_bg_last, para, para_v, para_ba, para_bg);
  }
  const std::vector<double> vbb0 = vbb, last0 = last_pose;
  ceres::Solver::Options options;
  options.linear_solver_type = ceres::DENSE_QR;
  options.num_threads = 1;
  options.max_num_iterations = max_it;
  ceres::Solver::Summary summary;
  adapt::Solve(options, &problem, &summary);
  wr(dir, "out_pose.f64", pose);
  const bool untouched = vbb == vbb0 && last_pose == last0;       // constant blocks are never written
  std::printf("{\"ok\": %d, \"message\": \"%s\", \"initial_cost\": %.17g, \"final_cost\": %.17g, \"successful\": %d, \"unsuccessful\": %d, \"num_residual_blocks\": %d, "
              "\"termination\": %d, \"constant_blocks_untouched\": %d, \"recorder_used\": %d}\n",
              summary.termination_type != ceres::FAILURE, summary.message.c_str(), summary.initial_cost, summary.final_cost, summary.num_successful_steps,
              summary.num_unsuccessful_steps, summary.num_residual_blocks_reduced, (int)summary.termination_type, (int)untouched, (int)problem.recorder.usable(&problem));
  return summary.termination_type == ceres::FAILURE ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && std::string(argv[1]) == "foreign") return run_foreign();
  if (argc < 3) { std::fprintf(stderr, "usage: %s window|lidar|posegraph|environment|relocate <dir> | foreign\n", argv[0]); return 2; }
  const std::string mode = argv[1];
  if (mode == "window") return run_window(argv[2]);
  if (mode == "lidar") return run_lidar(argv[2]);
  if (mode == "posegraph") return run_posegraph(argv[2]);
  if (mode == "environment") return run_environment(argv[2]);
  if (mode == "relocate") return run_relocate(argv[2]);
  return 2;
}
