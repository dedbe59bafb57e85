// adapter_test.cc -- exercises the C++ drop-ins the way nautilus's solver.cc would
// (GetRelativeTransform, solver.cc:630-649; AddResidualBlock + ceres::Solve evaluation).
// Needs an MI355X: run by tests/test_adapters_gpu.py.  Prints ADAPTER_OK on success.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <array>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "CorrelativeScanMatcher.h"
#include "slam_residuals_hip.h"

using nautilus_hip::Vec2f;

static std::vector<Vec2f> Room(double ox, double oy, double oth) {
  // points on the walls of a 9 m x 6 m room with a pillar, seen from pose (ox, oy, oth)
  std::vector<Vec2f> world;
  for (double x = -4.5; x <= 4.5; x += 0.04) { world.emplace_back(x, -3.0); world.emplace_back(x, 3.0); }
  for (double y = -3.0; y <= 3.0; y += 0.04) { world.emplace_back(-4.5, y); world.emplace_back(4.5, y); }
  for (double t = 0; t < 1.0; t += 0.04) { world.emplace_back(1.0 + t, 0.5); world.emplace_back(1.0, 0.5 + 0.6 * t); }
  std::vector<Vec2f> out;
  const double c = std::cos(-oth), s = std::sin(-oth);
  for (const Vec2f &w : world) {
    const double dx = w(0) - ox, dy = w(1) - oy;
    out.emplace_back((float)(c * dx - s * dy), (float)(s * dx + c * dy));
  }
  return out;
}

#define REQUIRE(cond)                                                        \
  do {                                                                       \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

static void Dump(const char *dir, const char *name, const std::vector<Vec2f> &pc) {
  if (!dir) return;
  const std::string path = std::string(dir) + "/" + name;
  if (FILE *f = std::fopen(path.c_str(), "wb")) {
    std::fwrite(pc.data(), sizeof(Vec2f), pc.size(), f);
    std::fclose(f);
  }
}

// argv[1] (optional): directory that receives the point clouds and every matcher result as hex floats, so that
// tests/test_adapters_gpu.py can compare them with the oracle's restatement float for float.

// ---- one host thread per stream (what SURVEY section 8e describes for a C++ host: one thread per device, here two
// threads on the one device this box has): each matches its own pair list on its own stream, forced onto the split
// form in overlapped rounds (NHIP_BNB_SPLIT_BATCH: rounds of 3 pairs, candidates on a helper stream of the library's
// per-device pool).  A shared event ring or a stream of the wrong device would show as wrong records or an error.
struct ThreadList {
  std::vector<int32_t> src, slot;
  std::vector<double> rot0;  // (cos, sin) per pair
};

static bool MatchList(const float *d_xy, const int32_t *d_off, int32_t n_scans, const uint8_t *d_grids, const nhip_grid_spec_t &spec,
                      const nhip_search_t &search, const double *d_delta, const ThreadList &L, hipStream_t st,
                      std::vector<nhip_match_t> *out, int32_t info[8]) {
  const int32_t n = (int32_t)L.src.size();
  int32_t *d_src = nullptr, *d_slot = nullptr;
  double *d_rot0 = nullptr;
  uint64_t *d_keys = nullptr;
  nhip_match_t *d_out = nullptr;
  void *d_ws = nullptr;
  const int64_t ws = nhip_csm_workspace_bytes(n);
  bool ok = hipMalloc(&d_src, 4 * n) == hipSuccess && hipMalloc(&d_slot, 4 * n) == hipSuccess &&
            hipMalloc(&d_rot0, 16 * n) == hipSuccess && hipMalloc(&d_keys, 8 * n) == hipSuccess &&
            hipMalloc(&d_out, sizeof(nhip_match_t) * n) == hipSuccess && hipMalloc(&d_ws, (size_t)ws) == hipSuccess;
  ok = ok && hipMemcpyAsync(d_src, L.src.data(), 4 * n, hipMemcpyHostToDevice, st) == hipSuccess &&
       hipMemcpyAsync(d_slot, L.slot.data(), 4 * n, hipMemcpyHostToDevice, st) == hipSuccess &&
       hipMemcpyAsync(d_rot0, L.rot0.data(), 16 * n, hipMemcpyHostToDevice, st) == hipSuccess;
  ok = ok && nhip_csm_match_dev(d_xy, d_off, n_scans, d_grids, n_scans /* every scan is a target */, &spec, d_src, d_slot, d_rot0, d_delta, nullptr, n, &search, d_keys,
                                d_out, nullptr, d_ws, ws, st) == NHIP_OK;
  if (!ok) std::printf("MatchList: %s\n", nhip_last_error());
  ok = ok && nhip_csm_last_launch(info) == NHIP_OK;
  out->resize(n);
  ok = ok && hipMemcpyAsync(out->data(), d_out, sizeof(nhip_match_t) * n, hipMemcpyDeviceToHost, st) == hipSuccess &&
       hipStreamSynchronize(st) == hipSuccess;
  (void)hipFree(d_src); (void)hipFree(d_slot); (void)hipFree(d_rot0); (void)hipFree(d_keys); (void)hipFree(d_out); (void)hipFree(d_ws);
  return ok;
}

static bool TwoThreadsOnTwoStreams() {
  // eight clouds of the room from eight poses; every cloud is a target (grid) and a source
  const int NS = 8;
  std::vector<float> xy;
  std::vector<int32_t> off(1, 0), ids;
  for (int i = 0; i < NS; i++) {
    const auto pc = Room(0.25 * i - 0.9, 0.12 * (i % 3) - 0.1, 0.05 * i - 0.15);
    for (const Vec2f &p : pc) { xy.push_back(p(0)); xy.push_back(p(1)); }
    off.push_back((int32_t)(xy.size() / 2));
    ids.push_back(i);
  }
  const nhip_grid_spec_t spec = {30.0, 0.05, 2.0, 1e-10, 40, 16, 0, 0};
  const nhip_search_t search = {61, 81, 81, 0, M_PI / 180.0};
  float *d_xy = nullptr;
  int32_t *d_off = nullptr, *d_ids = nullptr;
  uint8_t *d_grids = nullptr;
  void *d_gws = nullptr;
  double *d_delta = nullptr;
  std::vector<double> delta(2 * 61);
  if (nhip_csm_delta_table(&search, delta.data()) != NHIP_OK) return false;
  const int64_t gbytes = nhip_grids_bytes(&spec, NS), gws = nhip_grid_workspace_bytes(&spec, NS);
  bool ok = hipMalloc(&d_xy, 4 * xy.size()) == hipSuccess && hipMalloc(&d_off, 4 * off.size()) == hipSuccess &&
            hipMalloc(&d_ids, 4 * NS) == hipSuccess && hipMalloc(&d_grids, (size_t)gbytes) == hipSuccess &&
            hipMalloc(&d_gws, (size_t)gws) == hipSuccess && hipMalloc(&d_delta, 8 * delta.size()) == hipSuccess;
  ok = ok && hipMemcpy(d_xy, xy.data(), 4 * xy.size(), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(d_off, off.data(), 4 * off.size(), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(d_ids, ids.data(), 4 * NS, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(d_delta, delta.data(), 8 * delta.size(), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemset(d_grids, 0, (size_t)gbytes) == hipSuccess;
  ok = ok && nhip_grid_build_dev(d_xy, d_off, NS, d_ids, NS, &spec, d_grids, d_gws, gws, nullptr) == NHIP_OK &&
       hipDeviceSynchronize() == hipSuccess;
  if (!ok) { std::printf("two threads: setup failed: %s\n", nhip_last_error()); return false; }
  ThreadList lists[2];
  for (int t = 0; t < 2; t++)
    for (int i = 0; i < 28; i++) {
      const int s_ = (3 * i + t) % NS, g_ = (5 * i + 2 * t + 1) % NS;
      const double th = 0.05 * (s_ - g_) + 0.01 * t;
      lists[t].src.push_back(s_);
      lists[t].slot.push_back(g_);
      lists[t].rot0.push_back(std::cos(th));
      lists[t].rot0.push_back(std::sin(th));
    }
  // reference: each list on its own, default form (28 pairs: one kernel per pair + the hand-over kernel)
  std::vector<nhip_match_t> want[2], got[2];
  int32_t info[2][8];
  for (int t = 0; t < 2; t++) ok = ok && MatchList(d_xy, d_off, NS, d_grids, spec, search, d_delta, lists[t], nullptr, &want[t], info[t]);
  ok = ok && info[0][0] == 0;
  // forced: the split form in rounds of 3 pairs, candidates of a round on a helper stream (tunables are read per launch
  // in a process started with NHIP_TUNABLES=1: main() set it before the library's first call)
  setenv("NHIP_BNB_KERNELS", "1", 1);
  setenv("NHIP_BNB_SPLIT", "1", 1);
  setenv("NHIP_BNB_SPLIT_BATCH", "3", 1);
  setenv("NHIP_BNB_SPLIT_MIN", "40", 1);
  hipStream_t st[2];
  ok = ok && hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking) == hipSuccess &&
       hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking) == hipSuccess;
  bool tok[2] = {false, false};
  for (int rep = 0; rep < 3 && ok; rep++) {
    std::thread th0([&] { tok[0] = MatchList(d_xy, d_off, NS, d_grids, spec, search, d_delta, lists[0], st[0], &got[0], info[0]); });
    std::thread th1([&] { tok[1] = MatchList(d_xy, d_off, NS, d_grids, spec, search, d_delta, lists[1], st[1], &got[1], info[1]); });
    th0.join();
    th1.join();
    ok = tok[0] && tok[1];
    for (int t = 0; t < 2 && ok; t++) {
      ok = info[t][0] == 3 && info[t][1] == 3 && info[t][3] == 10 &&
           std::memcmp(got[t].data(), want[t].data(), sizeof(nhip_match_t) * want[t].size()) == 0;
      if (!ok) std::printf("two threads: rep %d thread %d: form %d batch %d rounds %d, records %s\n", rep, t, info[t][0], info[t][1],
                           info[t][3], std::memcmp(got[t].data(), want[t].data(), sizeof(nhip_match_t) * want[t].size()) ? "DIFFER" : "equal");
    }
  }
  unsetenv("NHIP_BNB_KERNELS");
  unsetenv("NHIP_BNB_SPLIT");
  unsetenv("NHIP_BNB_SPLIT_BATCH");
  unsetenv("NHIP_BNB_SPLIT_MIN");
  // A stale id in a device-resident list (include/nautilus_hip.h, "Ids in device memory"): the call enqueues as usual,
  // the bad pair scores nothing, the others keep their records, and nhip_dev_status() says which id it was -- once.
  if (ok) {
    ThreadList bad = lists[0];
    bad.src[5] = NS + 3;
    std::vector<nhip_match_t> rec;
    int32_t inf[8], st4[4] = {0, 0, 0, 0};
    ok = nhip_dev_status(nullptr, nullptr) == NHIP_OK;  // (nothing pending from the runs above)
    ok = ok && MatchList(d_xy, d_off, NS, d_grids, spec, search, d_delta, bad, nullptr, &rec, inf);
    const int rc = nhip_dev_status(nullptr, st4);
    ok = ok && rc == NHIP_ERR_ARG && st4[1] == 2 && st4[2] == NS + 3 && st4[3] == 5 && std::strstr(nhip_last_error(), "d_pair_src") &&
         nhip_dev_status(nullptr, nullptr) == NHIP_OK;
    for (size_t i = 0; i < rec.size() && ok; i++) {
      if (i == 5) ok = rec[i].itheta == 0 && rec[i].ix == 0 && rec[i].iy == 0;
      else ok = std::memcmp(&rec[i], &want[0][i], sizeof(nhip_match_t)) == 0;
    }
    if (!ok) std::printf("stale id: rc %d status {%d %d %d %d}: %s\n", rc, st4[0], st4[1], st4[2], st4[3], nhip_last_error());
  }
  (void)hipStreamDestroy(st[0]); (void)hipStreamDestroy(st[1]);
  (void)hipFree(d_xy); (void)hipFree(d_off); (void)hipFree(d_ids); (void)hipFree(d_grids); (void)hipFree(d_gws); (void)hipFree(d_delta);
  return ok;
}

int main(int argc, char **argv) {
  setenv("NHIP_TUNABLES", "1", 1);  // (a test binary: the section on two host threads forces a form of the matcher)
  const char *dump = argc > 1 ? argv[1] : nullptr;
  FILE *results = nullptr;
  if (dump) results = std::fopen((std::string(dump) + "/results.txt").c_str(), "w");
  int ndev = 0;
  if (nhip_init(&ndev) != NHIP_OK) { std::printf("no GPU: %s\n", nhip_last_error()); return 2; }
  // ---- scan matcher, used exactly as solver.cc:633-644
  const double ax = 0.62, ay = -0.37, ath = 0.21;  // pose of A in B's frame (B at the origin)
  const std::vector<Vec2f> pc_b = Room(0, 0, 0), pc_a = Room(ax, ay, ath);
  CorrelativeScanMatcher scan_matcher(30, 2, 0.3, 0.01);
  auto trans_pair = scan_matcher.GetTransformation(pc_a, pc_b, ath + 0.05, 0.0, M_PI / 2);
  const float tx = trans_pair.second.first(0), ty = trans_pair.second.first(1), th = trans_pair.second.second;
  std::printf("csm: score %.4f  t = (%.3f, %.3f)  theta = %.4f   truth (%.3f, %.3f, %.4f)\n", trans_pair.first, tx, ty,
              th, ax, ay, ath);
  REQUIRE(std::fabs(tx - ax) < 0.03 && std::fabs(ty - ay) < 0.03 && std::fabs(th - ath) < 0.01);
  REQUIRE(trans_pair.first > -8.0 && trans_pair.first <= 0.0);
  Dump(dump, "pc_a.f32", pc_a);
  Dump(dump, "pc_b.f32", pc_b);
  if (results)
    std::fprintf(results, "get_transformation %a %a %a %a %a %a %a\n", ath + 0.05, 0.0, M_PI / 2, trans_pair.first,
                 (double)tx, (double)ty, (double)th);
  {  // a second call with the clouds swapped and a narrower restriction (another lattice shape)
    auto back = scan_matcher.GetTransformation(pc_b, pc_a, 0.0, ath - 0.02, M_PI / 6);
    if (results)
      std::fprintf(results, "get_transformation_swapped %a %a %a %a %a %a %a\n", 0.0, ath - 0.02, M_PI / 6, back.first,
                   (double)back.second.first(0), (double)back.second.first(1), (double)back.second.second);
  }
  // batched form, BASELINE lattice
  nhip_search_t search = {61, 81, 81, 0, M_PI / 180.0};
  CorrelativeScanMatcherBatch batch(30.0, 0.05, search);
  auto res = batch.Match({pc_a, pc_b}, {{0, 1}, {1, 0}}, {ath - 0.03, 0.0});
  std::printf("batch: (%.2f %.2f %.3f) (%.2f %.2f %.3f)\n", res[0].translation(0), res[0].translation(1), res[0].rotation,
              res[1].translation(0), res[1].translation(1), res[1].rotation);
  REQUIRE(std::fabs(res[0].translation(0) - ax) <= 0.051 && std::fabs(res[0].translation(1) - ay) <= 0.051);
  REQUIRE(std::fabs(res[0].rotation - ath) <= 0.0176);
  if (results)
    for (int i = 0; i < 2; i++)
      std::fprintf(results, "batch %d %a %a %a %a\n", i, res[i].score, (double)res[i].translation(0),
                   (double)res[i].translation(1), (double)res[i].rotation);

  // ---- residual blocks, used as solver.cc:277-295 + Ceres' evaluation loop would
  auto &B = nautilus_hip::ResidualBatcher::Instance();
  std::vector<Vec2f> sp, tp, sn, tn;
  for (int i = 0; i < 300; i++) {
    const float a = 0.01f * i;
    sp.emplace_back(2.f * std::cos(a), 2.f * std::sin(a));
    tp.emplace_back(2.05f * std::cos(a + 0.02f), 1.95f * std::sin(a + 0.02f));
    sn.emplace_back(std::cos(a), std::sin(a));
    tn.emplace_back(std::cos(a + 0.02f), std::sin(a + 0.02f));
  }
  double poses[3][3] = {{0.1, -0.2, 0.05}, {0.0, 0.0, 0.0}, {1.0, 0.5, -0.4}};
  auto *c0 = nautilus::LIDARNormalResidual::create(sp, tp, sn, tn);
  auto *c1 = nautilus::LIDARPointResidual::create(sp, tp, sn, tn);
  auto *c2 = nautilus::LIDARNormalResidual::create(tp, sp, tn, sn);
  REQUIRE(c0->num_residuals() == 600 && c0->parameter_block_sizes().size() == 2 && c0->parameter_block_sizes()[0] == 3);
  std::vector<double> r(600), j0(1800), j1(1800);
  double *jac[2] = {j0.data(), j1.data()};
  double const *params[2] = {poses[0], poses[1]};
  // No Bind, no PrepareForEvaluation: the block learns its parameter blocks from this call and is evaluated alone on
  // the GPU at the parameters passed in (what ceres::Problem::Evaluate or Covariance::Compute would trigger).
  REQUIRE(B.slow_path_calls() == 0 && B.live_blocks(nautilus_hip::kLidarNormal) == 2 && B.live_blocks(nautilus_hip::kLidarPoint) == 1);
  REQUIRE(c0->Evaluate(params, r.data(), jac));
  REQUIRE(B.slow_path_calls() == 1);
  const std::vector<double> r_one = r, j0_one = j0, j1_one = j1;
  {  // the one-line form of solver.cc:280-283 for the other two blocks
    struct FakeProblem {
      int added = 0;
      int AddResidualBlock(nautilus_hip::CostFunctionBase *, void *, double *, double *) { return ++added; }
    } problem;
    REQUIRE(nautilus_hip::AddResidualBlock(problem, c1, (void *)nullptr, poses[0], poses[1]) == 1);
    REQUIRE(nautilus_hip::AddResidualBlock(problem, c2, (void *)nullptr, poses[2], poses[0]) == 2);
  }
  B.PrepareForEvaluation(true, true);
  REQUIRE(c0->Evaluate(params, r.data(), jac));
  REQUIRE(B.slow_path_calls() == 1);  // served from the batch
  // batched (both Jacobians rebuilt on the host from q = S2T p_s and the block's constants: 32 B per correspondence over
  // PCIe) against single-block (both Jacobians from the kernel): residuals bit for bit, Jacobian entries to rounding -- the
  // host recovers u as q - t where the kernel had u before it added t
  REQUIRE(r == r_one);
  {
    double worst = 0;
    for (size_t i = 0; i < j0.size(); i++) {
      worst = std::fmax(worst, std::fabs(j0[i] - j0_one[i]) / (1.0 + std::fabs(j0_one[i])));
      worst = std::fmax(worst, std::fabs(j1[i] - j1_one[i]) / (1.0 + std::fabs(j1_one[i])));
    }
    REQUIRE(worst < 1e-13);
  }
  // central differences through the same path (residual-only evaluations)
  double maxerr = 0;
  for (int k = 0; k < 3; k++) {
    std::vector<double> rp(600), rm(600);
    const double eps = 1e-6, keep = poses[0][k];
    poses[0][k] = keep + eps; B.PrepareForEvaluation(false, true); REQUIRE(c0->Evaluate(params, rp.data(), nullptr));
    poses[0][k] = keep - eps; B.PrepareForEvaluation(false, true); REQUIRE(c0->Evaluate(params, rm.data(), nullptr));
    poses[0][k] = keep;
    for (int i = 0; i < 600; i++) maxerr = std::fmax(maxerr, std::fabs((rp[i] - rm[i]) / (2 * eps) - j0[3 * i + k]));
  }
  std::printf("resid: |J - finite difference|_max = %.3g\n", maxerr);
  REQUIRE(maxerr < 1e-6);
  B.PrepareForEvaluation(true, true);
  double *only_tgt[2] = {nullptr, j1.data()};  // constant first block: NULL jacobian (solver.cc:384-386)
  REQUIRE(c1->Evaluate(params, r.data(), only_tgt));
  REQUIRE(std::fabs(r[0] - (tp[0](0) - (std::cos(0.05) * sp[0](0) - std::sin(0.05) * sp[0](1) + 0.1))) < 1e-12);
  {
    // Evaluate() at a point OTHER than the prepared one (a rejected trial step, Covariance::Compute): must return the
    // values AT THAT POINT, not the cached ones.  Reference: prepare at the moved point and compare.
    const long before = B.slow_path_calls();
    double moved[3] = {poses[0][0] + 0.3, poses[0][1] - 0.2, poses[0][2] + 0.1};
    double const *at_moved[2] = {moved, poses[1]};
    std::vector<double> rs(600), js0(1800), js1(1800);
    double *jm[2] = {js0.data(), js1.data()};
    REQUIRE(c0->Evaluate(at_moved, rs.data(), jm));
    REQUIRE(B.slow_path_calls() == before + 1);
    REQUIRE(rs != r_one);
    const double keep[3] = {poses[0][0], poses[0][1], poses[0][2]};
    std::memcpy(poses[0], moved, sizeof(moved));
    B.PrepareForEvaluation(true, true);
    REQUIRE(c0->Evaluate(params, r.data(), jac));
    REQUIRE(r == rs);  // (Jacobians: the batch rebuilds them on the host from q, the single block gets the kernel's -- equal to rounding)
    for (size_t i = 0; i < j0.size(); i++)
      REQUIRE(std::fabs(j0[i] - js0[i]) <= 1e-13 * (1.0 + std::fabs(js0[i])) && std::fabs(j1[i] - js1[i]) <= 1e-13 * (1.0 + std::fabs(js1[i])));
    std::memcpy(poses[0], keep, sizeof(keep));
    B.PrepareForEvaluation(true, true);
  }
  {
    // Ceres calls Evaluate() from hardware_concurrency() threads (solver.cc:271): 16 threads, every block many times
    const long before = B.slow_path_calls();
    nautilus_hip::BatchedCost *blocks[3] = {c0, c1, c2};
    double const *bp[3][2] = {{poses[0], poses[1]}, {poses[0], poses[1]}, {poses[2], poses[0]}};
    std::vector<std::vector<double>> want_r(3, std::vector<double>(600)), want_j(3, std::vector<double>(3600));
    for (int q = 0; q < 3; q++) {
      double *jq[2] = {want_j[q].data(), want_j[q].data() + 1800};
      REQUIRE(blocks[q]->Evaluate(bp[q], want_r[q].data(), jq));
    }
    std::atomic<int> bad{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < 16; t++)
      pool.emplace_back([&, t]() {
        std::vector<double> rr(600), jj(3600);
        double *jq[2] = {jj.data(), jj.data() + 1800};
        for (int it = 0; it < 200; it++) {
          const int q = (t + it) % 3;
          if (!blocks[q]->Evaluate(bp[q], rr.data(), jq) || rr != want_r[q] || jj != want_j[q]) bad++;
        }
      });
    for (auto &th : pool) th.join();
    REQUIRE(bad == 0);
    REQUIRE(B.slow_path_calls() == before);  // all 3200 calls took the lock-free path
    std::printf("resid: 16 threads x 200 Evaluate() calls on the prepared batch: identical, lock-free\n");
  }
  if (dump) {
    // the three LIDAR blocks' inputs and what Evaluate() returned at the prepared point, for the oracle comparison
    // (tests/test_adapters_gpu.py: orc_lidar_block, the Jet<6> restatement of slam_residuals.h:65-89, 124-145)
    nautilus_hip::BatchedCost *blocks[3] = {c0, c1, c2};
    double const *bp[3][2] = {{poses[0], poses[1]}, {poses[0], poses[1]}, {poses[2], poses[0]}};
    const std::vector<Vec2f> *in[3][4] = {{&sp, &tp, &sn, &tn}, {&sp, &tp, &sn, &tn}, {&tp, &sp, &tn, &sn}};
    for (int q = 0; q < 3; q++) {
      std::vector<double> rr(600), ja(1800), jb(1800);
      double *jq[2] = {ja.data(), jb.data()};
      REQUIRE(blocks[q]->Evaluate(bp[q], rr.data(), jq));
      FILE *f = std::fopen((std::string(dump) + "/lidar_block_" + std::to_string(q) + ".bin").c_str(), "wb");
      REQUIRE(f != nullptr);
      const int32_t head[2] = {blocks[q]->family(), 300};
      std::fwrite(head, sizeof(head), 1, f);
      for (int v = 0; v < 4; v++) std::fwrite(in[q][v]->data(), sizeof(Vec2f), 300, f);
      std::fwrite(bp[q][0], sizeof(double), 3, f);
      std::fwrite(bp[q][1], sizeof(double), 3, f);
      std::fwrite(rr.data(), sizeof(double), 600, f);
      std::fwrite(ja.data(), sizeof(double), 1800, f);
      std::fwrite(jb.data(), sizeof(double), 1800, f);
      std::fclose(f);
    }
  }
  {
    // Two problems at once: a second "problem" (its own cost functions, its own parameter blocks) is built while the
    // first one lives, evaluated in the same pass, and survives the first one's destruction.
    double q0[3] = {0.3, 0.1, -0.2}, q1[3] = {-0.1, 0.0, 0.1};
    auto *d0 = nautilus::LIDARNormalResidual::create(sp, tp, sn, tn);
    B.Bind(d0, q0, q1);
    REQUIRE(B.live_blocks(nautilus_hip::kLidarNormal) == 3);
    const long before = B.slow_path_calls();
    B.PrepareForEvaluation(true, true);
    double const *qp[2] = {q0, q1};
    std::vector<double> rd(600), jd0(1800), jd1(1800), ra(600);
    double *jd[2] = {jd0.data(), jd1.data()};
    REQUIRE(d0->Evaluate(qp, rd.data(), jd) && c0->Evaluate(params, ra.data(), nullptr));
    REQUIRE(B.slow_path_calls() == before && ra == r_one && rd != r_one);
    delete c0; delete c1; delete c2;  // "problem A" is destroyed (ceres::Problem deletes its cost functions)
    REQUIRE(B.live_blocks(nautilus_hip::kLidarNormal) == 1 && B.live_blocks(nautilus_hip::kLidarPoint) == 0);
    B.PrepareForEvaluation(true, true);
    std::vector<double> rd2(600), jd20(1800), jd21(1800);
    double *jd2[2] = {jd20.data(), jd21.data()};
    REQUIRE(d0->Evaluate(qp, rd2.data(), jd2));
    REQUIRE(B.slow_path_calls() == before && rd2 == rd && jd20 == jd0 && jd21 == jd1);
    REQUIRE(B.compiled_rows(nautilus_hip::kLidarNormal) == 600);  // the batch holds the live block only
    delete d0;
    REQUIRE(B.live_blocks(nautilus_hip::kLidarNormal) == 0);
    std::printf("resid: two problems coexist; a destroyed problem leaves nothing behind\n");
  }
  {
    // Solver::OptimizeOverGrowingWindow (solver.cc:335-356): ten times ResetProblem() + AddOdomFactors +
    // BuildOptimizationOverWindow + Solve, with FEATURE-sized blocks (<= 20 planar LIDARNormal rows + <= 10 edge
    // LIDARPoint rows per (i, j), slam_types.h:66-67).  NO Reset() call, as in the reference's sources: the cost
    // functions' destructors (ceres::Problem owns them) take their blocks out.  Every window's batch holds exactly
    // that window's rows, no block is ever evaluated alone, and a rebuilt window reproduces its values bit for bit.
    const int n_nodes = 40;
    std::vector<std::array<double, 3>> P(n_nodes);
    for (int i = 0; i < n_nodes; i++) P[i] = {0.25 * i, 0.02 * i, 0.01 * i};
    auto feature = [&](int i, int j, int n, std::vector<Vec2f> *v) {
      for (int q = 0; q < 4; q++) v[q].clear();
      for (int t = 0; t < n; t++) {
        const float a = 0.1f * t + 0.01f * i - 0.02f * j;
        v[0].emplace_back(3.f * std::cos(a), 3.f * std::sin(a));
        v[1].emplace_back(3.02f * std::cos(a + 0.01f), 2.97f * std::sin(a + 0.01f));
        v[2].emplace_back(std::cos(a), std::sin(a));
        v[3].emplace_back(std::cos(a + 0.01f), std::sin(a + 0.01f));
      }
    };
    struct FakeProblem {
      std::vector<nautilus_hip::CostFunctionBase *> owned;
      int AddResidualBlock(nautilus_hip::CostFunctionBase *c, void *, double *, double *) { owned.push_back(c); return (int)owned.size(); }
      ~FakeProblem() { for (auto *c : owned) delete c; }  // ceres::Problem::~Problem with default ownership
    };
    struct Factor { Vec2f translation; float rotation; };
    const long slow0 = B.slow_path_calls(), built0 = B.batches_built();
    std::vector<double> first_w10;
    for (int pass = 0; pass < 2; pass++)
      for (int window = 1; window <= 10; window++) {
        FakeProblem problem;  // ceres_information.ResetProblem()
        for (int i = 1; i < n_nodes; i++) {  // AddOdomFactors (solver.cc:370-387)
          const Factor f{Vec2f(0.25f, 0.02f), 0.01f};
          nautilus_hip::AddResidualBlock(problem, nautilus::OdometryResidual::create(f, 1.0, 1.0), (void *)nullptr, P[i - 1].data(), P[i].data());
        }
        int64_t rows_n = 0, rows_p = 0;
        std::vector<Vec2f> v[4];
        for (int i = 1; i < n_nodes; i++)  // BuildOptimizationOverWindow (solver.cc:321-333) -> AddLidarResiduals (:297-318)
          for (int j = std::max(0, i - window); j < i; j++) {
            feature(i, j, 20 - (i % 3), v);
            nautilus_hip::AddResidualBlock(problem, nautilus::LIDARNormalResidual::create(v[0], v[1], v[2], v[3]), (void *)nullptr, P[i].data(), P[j].data());
            rows_n += 2 * (int64_t)v[0].size();
            feature(i, j, 10 - (j % 2), v);
            nautilus_hip::AddResidualBlock(problem, nautilus::LIDARPointResidual::create(v[0], v[1], v[2], v[3]), (void *)nullptr, P[i].data(), P[j].data());
            rows_p += 2 * (int64_t)v[0].size();
          }
        for (int it = 0; it < 3; it++) {  // ceres::Solve: evaluation points
          for (int i = 1; i < n_nodes; i++) P[i][0] = 0.25 * i + 1e-3 * it;
          B.PrepareForEvaluation(true, true);
          std::vector<double> all;
          for (auto *c : problem.owned) {
            const auto *bc = static_cast<const nautilus_hip::BatchedCost *>(c);
            std::vector<double> rr(c->num_residuals()), ja(3 * rr.size()), jb(3 * rr.size());
            double *jq[2] = {ja.data(), jb.data()};
            double const *pp[2] = {bc->block()->pa, bc->block()->pb};
            REQUIRE(c->Evaluate(pp, rr.data(), jq));
            if (window == 10 && it == 2) { all.insert(all.end(), rr.begin(), rr.end()); all.insert(all.end(), jb.begin(), jb.end()); }
          }
          if (window == 10 && it == 2) {
            if (pass == 0) first_w10 = all;
            else REQUIRE(all == first_w10);
          }
        }
        REQUIRE(B.compiled_rows(nautilus_hip::kLidarNormal) == rows_n && B.compiled_rows(nautilus_hip::kLidarPoint) == rows_p);
        REQUIRE(B.compiled_rows(nautilus_hip::kOdometry) == 3 * (n_nodes - 1));
      }
    REQUIRE(B.slow_path_calls() == slow0);              // no block was ever evaluated alone
    REQUIRE(B.batches_built() == built0 + 2 * 20);      // one device batch per family and problem build
    for (int f = 0; f < 4; f++) REQUIRE(B.live_blocks(f) == 0);
    std::printf("resid: 20 problem builds (windows 1..10 twice, FEATURE-sized blocks): constant batch size, no slow path\n");
  }

  // ---- odometry factors (solver.cc:378) and HITL line constraints (solver.cc:521,528) in the same problem
  struct Factor { Vec2f translation; float rotation; };   // slam_types::OdometryFactor2D's members used by create()
  struct Segment { Vec2f start, end; };                   // LineSegment<float>'s
  const Factor f0{Vec2f(0.5f, -0.1f), 0.2f}, f1{Vec2f(-0.3f, 0.4f), -3.0f};
  auto *o0 = nautilus::OdometryResidual::create(f0, 2.0, 5.0);
  auto *o1 = nautilus::OdometryResidual::create(f1, 1.0, 0.5);
  const Segment seg{Vec2f(-1.f, 0.f), Vec2f(1.f, 0.f)};
  const std::vector<Vec2f> lp = {Vec2f(0.f, 0.5f), Vec2f(2.f, 1.f), Vec2f(-3.f, -0.25f)};
  auto *l0 = nautilus::PointToLineResidual::create(seg, lp);
  REQUIRE(o0->num_residuals() == 3 && l0->num_residuals() == 3);
  double zero[3] = {0, 0, 0};
  B.Bind(o0, poses[0], poses[1]);
  B.Bind(l0, zero, zero);  // o1 stays unbound: it binds itself on its first Evaluate()
  B.PrepareForEvaluation(true, true);
  double ro[3], joi[9], joj[9];
  double *ojac[2] = {joi, joj};
  REQUIRE(o0->Evaluate(params, ro, ojac));
  REQUIRE(std::fabs(ro[0] - 2.0 * (0.1 + (double)0.5f - 0.0)) < 1e-14);
  REQUIRE(std::fabs(ro[1] - 2.0 * (-0.2 + (double)-0.1f - 0.0)) < 1e-14);
  REQUIRE(std::fabs(ro[2] - 5.0 * (0.05 + (double)0.2f)) < 1e-14);
  REQUIRE(joi[0] == 2.0 && joi[4] == 2.0 && std::fabs(joi[8] - 5.0) < 1e-14 && joj[0] == -2.0 && joi[1] == 0.0);
  double const *params12[2] = {poses[1], poses[2]};
  REQUIRE(o1->Evaluate(params12, ro, ojac));  // unbound: single-factor evaluation at the parameters passed in
  {  // angle wrap: 0 + (-3) - (-0.4) = -2.6 stays; weights differ per factor
    REQUIRE(std::fabs(ro[2] - 0.5 * (double)(0.0 + (double)-3.0f + 0.4)) < 1e-14);
    REQUIRE(std::fabs(ro[0] - (0.0 + (double)-0.3f - 1.0)) < 1e-14);
  }
  double rl[3], jl0[9], jl1[9];
  double *ljac[2] = {jl0, jl1};
  double const *zparams[2] = {zero, zero};
  REQUIRE(l0->Evaluate(zparams, rl, ljac));
  {  // the same block at another pose, outside the prepared point: shifted up by 0.25
    double up[3] = {0.0, 0.25, 0.0};
    double const *uparams[2] = {up, zero};
    double ru[3];
    REQUIRE(l0->Evaluate(uparams, ru, nullptr));
    REQUIRE(std::fabs(ru[0] - 0.75) < 1e-12);
  }
  // identity poses: distance to the segment itself: 0.5 above the middle, sqrt(2) past the end, sqrt(4+1/16)
  REQUIRE(std::fabs(rl[0] - 0.5) < 1e-12 && std::fabs(rl[1] - std::sqrt(2.0)) < 1e-12);
  REQUIRE(std::fabs(rl[2] - std::sqrt(4.0 + 0.0625)) < 1e-12);
  REQUIRE(std::fabs(jl0[1] - 1.0) < 1e-12 && std::fabs(jl1[1] + 1.0) < 1e-12);  // d r0 / d pose.y = +1, / d line.y = -1
  std::printf("odometry + point-to-line blocks: ok (r_line = %.4f %.4f %.4f)\n", rl[0], rl[1], rl[2]);
  delete o0; delete o1; delete l0;
  for (int f = 0; f < 4; f++) REQUIRE(B.live_blocks(f) == 0);
  // ---- the path's one collective through the C ABI: a C++ host owns the RCCL communicator
  // (one rank per GPU; this box has one, so world size 1) and hands it over as void*
  {
    ncclUniqueId id;
    ncclComm_t comm;
    REQUIRE(ncclGetUniqueId(&id) == ncclSuccess);
    REQUIRE(ncclCommInitRank(&comm, 1, id, 0) == ncclSuccess);
    std::vector<nhip_match_t> local(1000), all(1000);
    for (int i = 0; i < 1000; i++) local[i] = {i, 2 * i, 3 * i, -0.5f * i};
    nhip_match_t *d_local, *d_all;
    REQUIRE(hipMalloc(&d_local, sizeof(nhip_match_t) * 1000) == hipSuccess);
    REQUIRE(hipMalloc(&d_all, sizeof(nhip_match_t) * 1000) == hipSuccess);
    REQUIRE(hipMemcpy(d_local, local.data(), sizeof(nhip_match_t) * 1000, hipMemcpyHostToDevice) == hipSuccess);
    hipStream_t st;
    REQUIRE(hipStreamCreate(&st) == hipSuccess);
    REQUIRE(nhip_allgather_matches(comm, d_local, 1000, d_all, st) == NHIP_OK);
    REQUIRE(hipStreamSynchronize(st) == hipSuccess);
    REQUIRE(hipMemcpy(all.data(), d_all, sizeof(nhip_match_t) * 1000, hipMemcpyDeviceToHost) == hipSuccess);
    REQUIRE(std::memcmp(all.data(), local.data(), sizeof(nhip_match_t) * 1000) == 0);
    REQUIRE(nhip_allgather_matches(nullptr, d_local, 1000, d_all, st) == NHIP_ERR_ARG);
    (void)hipFree(d_local); (void)hipFree(d_all); (void)hipStreamDestroy(st);
    ncclCommDestroy(comm);
    std::printf("allgather of 1000 match records over RCCL (1 rank): ok\n");
  }
  // ---- LCMatcher::GetPossibleMatches (lc_matcher.cc:59-74) from a C++ host: one call for a source node's candidates
  {
    const double poses[4][3] = {{0.0, 0.0, 0.0}, {1.0, 0.0, 0.1}, {0.0, 30.0, 0.2}, {0.5, 0.5, 0.3}};
    const int32_t src[4] = {0, 0, 0, 0}, tgt[4] = {0, 1, 2, 3};
    const float cov[4][4] = {{1e-3f, 0, 0, 1e-3f}, {1e-3f, 0, 0, 1e-3f}, {1e-3f, 0, 0, 1e-3f}, {2e-3f, 1e-3f, 1e-3f, 2e-3f}};
    double scores[4];
    uint8_t flags[4];
    REQUIRE(nhip_lc_chi_square_gate(&poses[0][0], 4, src, tgt, &cov[0][0], 4, 5000.0, scores, flags) == NHIP_OK);
    REQUIRE(flags[0] == 0 && flags[1] == 1 && flags[2] == 0 && flags[3] == 1);      // self; 1000; 900000; 166.7
    REQUIRE(std::fabs(scores[1] - 1000.0) < 0.5 && std::fabs(scores[2] - 900000.0) < 500.0 && std::fabs(scores[3] - 166.667) < 0.1);
    REQUIRE(nhip_lc_chi_square_gate(&poses[0][0], 4, src, tgt, nullptr, 4, 5000.0, scores, flags) == NHIP_ERR_ARG);
    std::printf("chi-square gate of 4 candidates: ok\n");
  }
  REQUIRE(TwoThreadsOnTwoStreams());
  std::printf("two host threads on two streams, split form forced into overlapped rounds: records equal the one-thread runs\n");
  if (results) std::fclose(results);
  std::printf("ADAPTER_OK\n");
  return 0;
}
