// relocalize_driver.cpp — Relocator::CorrectLoop's candidate loop (src/lvio_fusion/src/relocator.cpp:186-233) sharded over GPUs from a
// C++ host, on the C-ABI only (include/lvf.h) — no Python, no torch in the process:
//
//   * inside one process: one worker THREAD per device slot, each with its own lvf_ctx (own HIP stream and allocator) on its device;
//     thread t evaluates candidates t, t + T, ... with lvf_scan_match (Mapping::Relocate: 4 outer iterations x {ground, surf}) and
//     writes their records (score - 20, relative_o_c[7], candidate id) into its slots of a shared table — no collective at all;
//   * across processes (one per GPU, the layout bench.py --gpus N uses): every rank does the above for its share r, r + W, ... and the
//     tables meet in ONE all-gather over RCCL (lvf_comm_allgather); rank 0 publishes the 128-byte id through a file;
//   * every rank then runs the same arg-max (`>=`: the LAST of equal scores; only scores > 0 qualify, relocator.cpp:198-204) and the
//     loop-correction tail on device: UpdateNewSubmap's rotation solve (lvf_relocate_rotation_solve, :251-267) and ForwardUpdate
//     (lvf_forward_update, pose_graph.cpp:245-252) over the keyframes that follow.
//
//   relocalize_driver <dir> <n_candidates> <threads> [--devices a,b,..] [--rank r --world w --idfile path] [--batched 0|1]
// (a worker's candidates go through ONE launch chain — lvf_scan_match_batch — unless --batched 0)
// Inputs (raw little-endian, written by tests/test_gpu_relocalize.py): <dir>/c<i>_{map,query}.f32 [n][4], c<i>_{map,query}_ground.u8,
// c<i>_poses.f64 = map_pose | last_pose | init_pose; optional <dir>/tail_{relocated,unrelocated,forward}.f64.  One JSON line on stdout.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "lvf.h"

template <typename T>
static std::vector<T> rd(const std::string& path, bool required = true) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) { if (required) { std::fprintf(stderr, "missing %s\n", path.c_str()); std::exit(2); } return {}; }
  const std::streamsize n = f.tellg();
  f.seekg(0);
  std::vector<T> v((size_t)n / sizeof(T));
  f.read(reinterpret_cast<char*>(v.data()), n);
  return v;
}

constexpr int kRecord = 9;            // score, relative_o_c[7], candidate id  (lvio_fusion_amd/relocalize.py uses the same layout)
constexpr int kRelocateBaseScore = 20;

struct Candidate {
  std::vector<float> map_ground, map_surf, query_ground, query_surf;      // [n][4]
  double map_pose[7], last_pose[7], init_pose[7];
};

static Candidate load_candidate(const std::string& dir, int i) {
  Candidate c;
  const std::string p = dir + "/c" + std::to_string(i) + "_";
  auto split = [](const std::vector<float>& pts, const std::vector<uint8_t>& g, std::vector<float>& a, std::vector<float>& b) {
    for (size_t k = 0; k < g.size(); ++k) { std::vector<float>& dst = g[k] ? a : b; dst.insert(dst.end(), &pts[4 * k], &pts[4 * k] + 4); }
  };
  split(rd<float>(p + "map.f32"), rd<uint8_t>(p + "map_ground.u8"), c.map_ground, c.map_surf);
  split(rd<float>(p + "query.f32"), rd<uint8_t>(p + "query_ground.u8"), c.query_ground, c.query_surf);
  const auto poses = rd<double>(p + "poses.f64");
  std::memcpy(c.map_pose, &poses[0], 56); std::memcpy(c.last_pose, &poses[7], 56); std::memcpy(c.init_pose, &poses[14], 56);
  return c;
}

// Mapping::Relocate for one candidate on one context; returns false with the library's message on failure
static bool evaluate(lvf_ctx* ctx, const Candidate& c, double* record, int cand_id, std::string* err) {
  lvf_scan_match_options o;
  lvf_scan_match_options_default(&o, 0.2);
  o.outer_iterations = 4; o.prior_weight = 0.0;
  lvf_map *mg = nullptr, *ms = nullptr;
  lvf_scan *sg = nullptr, *ss = nullptr;
  bool ok = true;
  auto chk = [&](int rc) { if (rc != LVF_OK && ok) { ok = false; *err = lvf_last_error(); } return rc == LVF_OK; };
  if (!c.map_ground.empty()) {
    chk(lvf_map_create(ctx, c.map_ground.data(), (int)c.map_ground.size() / 4, 4, o.thr_ground, &mg)) &&
        chk(lvf_scan_create(ctx, c.query_ground.data(), (int)c.query_ground.size() / 4, 4, &sg));
  }
  if (ok && !c.map_surf.empty()) {
    chk(lvf_map_create(ctx, c.map_surf.data(), (int)c.map_surf.size() / 4, 4, o.thr_surf, &ms)) &&
        chk(lvf_scan_create(ctx, c.query_surf.data(), (int)c.query_surf.size() / 4, 4, &ss));
  }
  lvf_scan_match_result r;
  if (ok && chk(lvf_scan_match(mg, sg, ms, ss, c.map_pose, c.init_pose, c.last_pose, &o, &r))) {
    record[0] = (double)(r.score - kRelocateBaseScore);
    std::memcpy(record + 1, r.relative_o_c, 56);
    record[8] = (double)cand_id;
  }
  if (sg) lvf_scan_destroy(sg);
  if (ss) lvf_scan_destroy(ss);
  if (mg) lvf_map_destroy(mg);
  if (ms) lvf_map_destroy(ms);
  return ok;
}

// The same for a worker's whole share in ONE launch chain (lvf_scan_match_batch): maps and scans are created per candidate, the solves of
// all candidates run side by side, the records come back with one read
static bool evaluate_batch(lvf_ctx* ctx, const std::vector<Candidate>& cands, const std::vector<int>& ids, const std::vector<size_t>& slots, double* table, std::string* err) {
  lvf_scan_match_options o;
  lvf_scan_match_options_default(&o, 0.2);
  o.outer_iterations = 4; o.prior_weight = 0.0;
  std::vector<lvf_scan_match_job> jobs(ids.size());
  bool ok = true;
  auto chk = [&](int rc) { if (rc != LVF_OK && ok) { ok = false; *err = lvf_last_error(); } return rc == LVF_OK; };
  // every candidate's map indices in ONE call: the host waits of an index build are shared between them (lvf_map_create_batch)
  std::vector<const float*> src; std::vector<int> M; std::vector<float> thr; std::vector<lvf_map**> dst;
  for (size_t k = 0; k < ids.size(); ++k) {
    const Candidate& c = cands[ids[k]];
    lvf_scan_match_job& j = jobs[k];
    std::memset(&j, 0, sizeof(j));
    if (!c.map_ground.empty()) { src.push_back(c.map_ground.data()); M.push_back((int)c.map_ground.size() / 4); thr.push_back(o.thr_ground); dst.push_back(&j.map_ground); }
    if (!c.map_surf.empty()) { src.push_back(c.map_surf.data()); M.push_back((int)c.map_surf.size() / 4); thr.push_back(o.thr_surf); dst.push_back(&j.map_surf); }
  }
  std::vector<lvf_map*> made(src.size(), nullptr);
  if (!src.empty() && chk(lvf_map_create_batch(ctx, (int)src.size(), src.data(), M.data(), 4, thr.data(), made.data())))
    for (size_t i = 0; i < made.size(); ++i) *dst[i] = made[i];
  for (size_t k = 0; k < ids.size() && ok; ++k) {
    const Candidate& c = cands[ids[k]];
    lvf_scan_match_job& j = jobs[k];
    if (j.map_ground) chk(lvf_scan_create(ctx, c.query_ground.data(), (int)c.query_ground.size() / 4, 4, &j.scan_ground));
    if (ok && j.map_surf) chk(lvf_scan_create(ctx, c.query_surf.data(), (int)c.query_surf.size() / 4, 4, &j.scan_surf));
    std::memcpy(j.map_pose, c.map_pose, 56); std::memcpy(j.frame_pose, c.init_pose, 56); std::memcpy(j.last_pose, c.last_pose, 56);
    j.has_last_pose = 1;
  }
  std::vector<lvf_scan_match_result> res(ids.size());
  if (ok && chk(lvf_scan_match_batch(ctx, jobs.data(), (int)jobs.size(), &o, kRelocateBaseScore, res.data(), nullptr)))
    for (size_t k = 0; k < ids.size(); ++k) {
      double* record = table + slots[k] * kRecord;
      record[0] = (double)(res[k].score - kRelocateBaseScore);
      std::memcpy(record + 1, res[k].relative_o_c, 56);
      record[8] = (double)ids[k];
    }
  for (auto& j : jobs) {
    if (j.scan_ground) lvf_scan_destroy(j.scan_ground);
    if (j.scan_surf) lvf_scan_destroy(j.scan_surf);
    if (j.map_ground) lvf_map_destroy(j.map_ground);
    if (j.map_surf) lvf_map_destroy(j.map_surf);
  }
  return ok;
}

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: relocalize_driver <dir> <n_candidates> <threads> [--devices a,b] [--rank r --world w --idfile f]\n"); return 2; }
  const std::string dir = argv[1];
  const int n = std::atoi(argv[2]), T = std::max(1, std::atoi(argv[3]));
  std::vector<int> devices{0};
  int rank = 0, world = 1;
  bool batched = true;                 // a worker's candidates in one launch chain (lvf_scan_match_batch); --batched 0: one lvf_scan_match each
  std::string idfile;
  for (int a = 4; a + 1 < argc; a += 2) {
    const std::string k = argv[a], v = argv[a + 1];
    if (k == "--devices") { devices.clear(); size_t p = 0; while (p < v.size()) { devices.push_back(std::atoi(v.c_str() + p)); p = v.find(',', p); if (p == std::string::npos) break; ++p; } }
    else if (k == "--rank") rank = std::atoi(v.c_str());
    else if (k == "--world") world = std::atoi(v.c_str());
    else if (k == "--idfile") idfile = v;
    else if (k == "--batched") batched = std::atoi(v.c_str()) != 0;
  }
  std::vector<Candidate> cands(n);
  for (int i = 0; i < n; ++i) cands[i] = load_candidate(dir, i);

  // this rank's candidates r, r + W, ...; thread t takes every T-th of them
  std::vector<int> mine;
  for (int i = rank; i < n; i += world) mine.push_back(i);
  const int slots = (n + world - 1) / world;
  std::vector<double> table((size_t)slots * kRecord, 0.0);
  for (int s = 0; s < slots; ++s) { table[(size_t)s * kRecord] = -INFINITY; table[(size_t)s * kRecord + 4] = 1.0; table[(size_t)s * kRecord + 8] = -1.0; }
  std::vector<std::string> errors(T);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      lvf_ctx* ctx = nullptr;
      if (lvf_ctx_create(devices[t % devices.size()], nullptr, &ctx) != LVF_OK) { errors[t] = lvf_last_error(); return; }
      if (batched) {
        std::vector<int> ids; std::vector<size_t> sl;
        for (size_t s = t; s < mine.size(); s += T) { ids.push_back(mine[s]); sl.push_back(s); }
        if (!ids.empty()) evaluate_batch(ctx, cands, ids, sl, table.data(), &errors[t]);
      } else
      for (size_t s = t; s < mine.size(); s += T)
        if (!evaluate(ctx, cands[mine[s]], &table[s * kRecord], mine[s], &errors[t])) break;     // a failed candidate keeps its "unused" slot
      lvf_ctx_destroy(ctx);
    });
  for (auto& x : th) x.join();
  const double eval_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::string err;
  for (const auto& e : errors) if (!e.empty()) err = e;

  // ---- the exchange (multi-process only) and the arg-max every rank repeats
  std::vector<double> all((size_t)slots * kRecord * world);
  lvf_ctx* ctx = nullptr;
  if (lvf_ctx_create(devices[0], nullptr, &ctx) != LVF_OK) { std::fprintf(stderr, "%s\n", lvf_last_error()); return 1; }
  lvf_comm* comm = nullptr;
  unsigned char id[LVF_COMM_ID_BYTES];
  const bool use_rccl = !idfile.empty();
  if (use_rccl) {
    if (rank == 0) {
      if (lvf_comm_get_unique_id(id) != LVF_OK) { std::fprintf(stderr, "%s\n", lvf_last_error()); return 1; }
      std::ofstream f(idfile + ".tmp", std::ios::binary); f.write(reinterpret_cast<const char*>(id), sizeof(id)); f.close();
      std::rename((idfile + ".tmp").c_str(), idfile.c_str());
    } else {
      for (int tries = 0;; ++tries) {
        auto v = rd<unsigned char>(idfile, false);
        if (v.size() == sizeof(id)) { std::memcpy(id, v.data(), sizeof(id)); break; }
        if (tries > 600) { std::fprintf(stderr, "rank %d: no id file after 60 s\n", rank); return 1; }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
      }
    }
  }
  if (lvf_comm_create(ctx, world, rank, use_rccl ? id : nullptr, &comm) != LVF_OK) { std::fprintf(stderr, "%s\n", lvf_last_error()); return 1; }
  if (lvf_comm_allgather(comm, table.data(), slots * kRecord, all.data()) != LVF_OK) { std::fprintf(stderr, "%s\n", lvf_last_error()); return 1; }
  int best = -1; double best_score = -1.0; double best_rel[7] = {0, 0, 0, 1, 0, 0, 0};
  for (int cid = 0; cid < n; ++cid)                       // candidate order, like the loop over new_submap_kfs
    for (size_t s = 0; s < (size_t)slots * world; ++s) {
      const double* r = &all[s * kRecord];
      if ((int)r[8] != cid || r[8] < 0) continue;
      if (r[0] > 0 && r[0] >= best_score) { best_score = r[0]; best = cid; std::memcpy(best_rel, r + 1, 56); }
    }

  // ---- loop-correction tail on device (optional inputs): rotation solve + forward update
  double q4[4] = {0, 0, 0, 1};
  lvf_solver_summary rs;
  std::memset(&rs, 0, sizeof(rs));
  const auto relocated = rd<double>(dir + "/tail_relocated.f64", false), unrelocated = rd<double>(dir + "/tail_unrelocated.f64", false);
  if (!relocated.empty()) {
    lvf_solver_options so;
    lvf_solver_options_default(&so);
    if (lvf_relocate_rotation_solve(ctx, (int)relocated.size() / 7, relocated.data(), unrelocated.data(), q4, &so, &rs) != LVF_OK) { std::fprintf(stderr, "%s\n", lvf_last_error()); return 1; }
  }
  auto forward = rd<double>(dir + "/tail_forward.f64", false);       // transform[7] | poses[m][7] | vw[m][3]
  if (forward.size() >= 7) {
    const int m = (int)(forward.size() - 7) / 10;
    if (lvf_forward_update(ctx, forward.data(), m, forward.data() + 7, forward.data() + 7 + (size_t)7 * m) != LVF_OK) { std::fprintf(stderr, "%s\n", lvf_last_error()); return 1; }
    std::ofstream f(dir + "/out_forward_r" + std::to_string(rank) + ".f64", std::ios::binary);
    f.write(reinterpret_cast<const char*>(forward.data() + 7), (std::streamsize)((forward.size() - 7) * 8));
  }
  {
    std::ofstream f(dir + "/out_records_r" + std::to_string(rank) + ".f64", std::ios::binary);
    f.write(reinterpret_cast<const char*>(all.data()), (std::streamsize)(all.size() * 8));
  }
  lvf_comm_destroy(comm);
  lvf_ctx_destroy(ctx);
  std::printf("{\"ok\": %d, \"error\": \"%s\", \"rank\": %d, \"world\": %d, \"threads\": %d, \"batched\": %d, \"rccl\": %d, \"eval_ms\": %.3f, \"best\": %d, \"best_score\": %.1f, "
              "\"best_rel\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g], \"q4\": [%.17g, %.17g, %.17g, %.17g], \"rot_iterations\": %d, \"rot_final_cost\": %.17g}\n",
              err.empty() ? 1 : 0, err.c_str(), rank, world, T, batched ? 1 : 0, use_rccl ? 1 : 0, eval_ms, best, best_score, best_rel[0], best_rel[1], best_rel[2], best_rel[3], best_rel[4],
              best_rel[5], best_rel[6], q4[0], q4[1], q4[2], q4[3], rs.num_iterations, rs.final_cost);
  return err.empty() ? 0 : 1;
}
