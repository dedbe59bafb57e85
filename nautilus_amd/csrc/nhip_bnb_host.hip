// nhip_bnb_host.hip -- host side of the branch-and-bound matcher (nhip_bnb.hip holds the kernels): which form a pair
// list takes (one kernel per pair with hand-over lists / the split form in one round / in overlapped rounds), the
// workspace's layout, the helper streams of the overlapped rounds, the instrumentation's buffers.  Every form returns
// the same records; tests run them all (NHIP_TUNABLES=1 lets a process choose the form per launch).
#include <atomic>
#include <mutex>
#include <vector>

#include "nhip_bnb_params.h"

namespace nhip {

using namespace bnb;

namespace {

size_t bnb_lds_first(const GridLayout &L, bool pool_lds) {
  const size_t pool = pool_lds ? (size_t)L.pool_bytes : 0;
  return pool > (size_t)ORG_LDS ? pool : (size_t)ORG_LDS;
}
size_t bnb_lds_bytes(const GridLayout &L, const nhip_search_t *search, bool pool_lds) {
  return bnb_lds_first(L, pool_lds) + (size_t)search->n_theta * 128 * 4 + (size_t)QCAP * 8 + 64;
}
constexpr size_t LDS_MAX = 160 * 1024;

}  // namespace

bool bnb_fits(const GridLayout &L, const nhip_search_t *search) {
  const int nbx = (search->nx + BNB_B - 1) / BNB_B, nby = (search->ny + BNB_B - 1) / BNB_B;
  // (the pooled table goes to LDS when it fits beside the bounds; else it is read from global memory)
  return nbx <= NB && nby <= NB && bnb_lds_bytes(L, search, false) <= LDS_MAX && L.pool_bytes % 16 == 0 &&
         L.pool_bytes < (1ll << RUN_SHIFT) && L.S + 2 * L.pad < 65536 && search->n_theta <= MAX_ROT;
}

// Instrumentation buffers (NHIP_BNB_INSTRUMENT=1 only): process-wide, allocated on first use, guarded by g_instr_mu
static std::mutex g_instr_mu;
static unsigned long long *g_bnb_timeline = nullptr;
static unsigned long long *g_bnb_stats = nullptr;

constexpr int64_t BNB_WS_HEADER = 256;  // per XCD 32 bytes: {entries filled, next entry to work}
// Lists of fewer than SPLIT_MIN_PAIRS pairs: room for 16 handed-over rotations per pair on average (what does not fit is
// worked by the pair's own workgroup).  The split form: per pair its four counters, 1.5 entries of the candidates' work
// list and the rows of bounds of up to 64 rotations.  Lists of SPLIT_MIN_PAIRS .. SPLIT_PAIRS pairs take it in ONE round;
// longer lists in rounds of SPLIT_PAIRS with the candidates of a round on a helper stream beside the next round's bounds,
// which needs two rounds' state (8.6 GB at 61 rotations); with less workspace they stay fused.  Measured (match ms,
// one kernel per pair + hand-over / split; tools/r04_small_lists.sh, profiles/r04_small_lists.txt): 30 pairs 0.21 / 0.21,
// 100 pairs 0.23 / 0.28, 200 pairs 0.71 / 0.60, 300 pairs 1.09 / 0.79, 500 pairs 1.12 / 0.86, 1,000 pairs 1.60 / 1.21,
// 2,000 pairs 4.02 / 1.95, 3,000 pairs 4.08 / 2.63 (round 3's per-XCD work lists: 4.70), 10,000 pairs 8.1 / 6.4; round 3,
// 40,000 pairs 27.1 / 22.8, 1,000,000 pairs at 100 per target 606 fused, 663 in rounds of 65,536 without the helper
// stream (every round pays its own tail), 600 with it, 543 in rounds of 131,072 with it.
constexpr int64_t SPLIT_PAIRS = 131072, SPLIT_MIN_PAIRS = 192, SPLIT_RING = 16;
// Rounds of fewer pairs than this deal the additional workgroups of their heavy pairs over all eight XCDs' lists
// (csm_bnb_order_spread_kernel); longer ones keep them in the pair's home list (csm_bnb_order_kernel), where every XCD
// has heavy pairs of its own and the tables stay L2-resident.  Measured, match ms home / spread: 3,000 pairs 4.70 / 2.63,
// 4,500 pairs 3.39 / 3.46, 10,000 pairs 6.38 / 6.54 (profiles/r04_small_lists.txt).
constexpr int64_t SPREAD_BELOW_PAIRS = 4096;
int64_t split_bytes_per_pair(int32_t n_theta) { return 16 + 6 + (int64_t)n_theta * 512; }
constexpr int64_t SPLIT_SLOT_FIXED = 8 * 64 * 4 + 1024;  // per batch: the work lists' floor of 64 extra entries, alignment
int64_t bnb_workspace_bytes_lists(int32_t n_pairs) {  // (the hand-over lists alone: the one-kernel form)
  const int64_t n = n_pairs > 0 ? n_pairs : 0;
  return BNB_WS_HEADER + 8 * (((n + 7) / 8) * 16 + 64) * (int64_t)sizeof(RotEntry);
}
int64_t bnb_workspace_bytes(int32_t n_pairs) {
  const int64_t n = n_pairs > 0 ? n_pairs : 0;
  const int64_t lists = bnb_workspace_bytes_lists(n_pairs);
  const char *sp = tunable("NHIP_BNB_SPLIT");  // (=1: the split form for small batches too -- tests)
  const int64_t cap = SPLIT_PAIRS;
  const int64_t m = n <= cap ? n : 2 * cap;  // (a longer list: two rounds' state, so that the helper stream can be used)
  const bool forced = n > 0 && ((sp && sp[0] == '1') || tunable("NHIP_BNB_SPLIT_BATCH"));
  const int64_t split = n >= SPLIT_MIN_PAIRS || forced
                            ? BNB_WS_HEADER + (m / 512 + 4) * SPLIT_SLOT_FIXED + m * split_bytes_per_pair(64) : 0;
  return lists > split ? lists : split;
}

// The helper stream of the split form (the candidates of round i run beside the bounds of round i + 1) and the events
// that order the two.  One set per CALL IN FLIGHT, taken from a per-device pool: a host with one thread per device
// (nhip_set_device) gets a stream and events of ITS device, and two threads on one device never share
// events -- a wait binds to the event's latest record, so a shared ring would let one caller's candidates start on the
// other's bounds.  The pool's mutex is held only to take a set and to put it back, never across the enqueue: the set
// goes back as soon as the call has enqueued its work (the waits already issued stay bound to their records, and the
// helper stream runs in order, so the next user queues up behind).  Sets live until the process ends.
struct SplitSet {
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ea[SPLIT_RING], eb[SPLIT_RING];
};
static std::mutex g_split_mu;
static std::vector<SplitSet *> g_split_free;

static int split_set_acquire(SplitSet **out) {
  int dev = -1;
  NHIP_TRY_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lock(g_split_mu);
    for (size_t i = 0; i < g_split_free.size(); i++)
      if (g_split_free[i]->device == dev) {
        *out = g_split_free[i];
        g_split_free.erase(g_split_free.begin() + (long)i);
        return NHIP_OK;
      }
  }
  SplitSet *n = new SplitSet();
  n->device = dev;
  hipError_t e = hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking);
  for (int i = 0; i < SPLIT_RING && e == hipSuccess; i++) {
    e = hipEventCreateWithFlags(&n->ea[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&n->eb[i], hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    delete n;  // (what was created stays with the runtime: an allocation failure of streams / events is not a path to tidy)
    return hip_fail(e, "split form: helper stream / events", __FILE__, __LINE__);
  }
  *out = n;
  return NHIP_OK;
}
static void split_set_release(SplitSet *set) {
  if (!set) return;
  std::lock_guard<std::mutex> lock(g_split_mu);
  g_split_free.push_back(set);
}

// NHIP_BNB_INSTRUMENT=1 selects the instrumented build of the kernels; only then are NHIP_BNB_STATS and NHIP_BNB_TIMELINE
// read at all.
static bool instrumented() {
  const char *e = tunable("NHIP_BNB_INSTRUMENT");
  return e && e[0] == '1';
}

// What the calling thread's last launch_csm_bnb did (nhip_csm_last_launch: tests assert the form a list took).
static thread_local int32_t t_last_launch[8] = {0, 0, 0, 0, 0, 0, 0, 0};
void bnb_last_launch(int32_t out[8]) { memcpy(out, t_last_launch, sizeof(t_last_launch)); }

int launch_csm_bnb(const float *d_xy, const int32_t *d_offsets, const IdBounds &ids, const uint8_t *d_grids,
                   const nhip_grid_spec_t *spec, const GridLayout &L, const int32_t *d_pair_src,
                   const int32_t *d_pair_slot, const double *d_rot0_cs, const double *d_delta_cs,
                   const int32_t *d_pair_origin, int32_t n_pairs, const nhip_search_t *search,
                   uint64_t *d_keys, nhip_match_t *d_out, int32_t *d_sums, hipStream_t s, int *handled,
                   void *d_workspace, int64_t workspace_bytes, const int32_t *d_pair_kbase) {
  *handled = 0;
  if (!bnb_fits(L, search)) return NHIP_OK;
  *handled = 1;
  if (n_pairs == 0) return NHIP_OK;
  BnbParams P;
  memset(&P, 0, sizeof(P));
  P.xy = reinterpret_cast<const float2 *>(d_xy);
  P.offsets = d_offsets;
  P.grids = d_grids;
  P.pair_src = d_pair_src;
  P.pair_slot = d_pair_slot;
  P.ids = ids;
  P.rot0_cs = d_rot0_cs;
  P.delta_cs = d_delta_cs;
  P.pair_origin = d_pair_origin;
  P.pair_kbase = d_pair_kbase;
  P.keys = reinterpret_cast<unsigned long long *>(d_keys);
  P.pair_base = 0;
  P.n_pairs = n_pairs;
  P.n_theta = search->n_theta;
  P.nx = search->nx;
  P.ny = search->ny;
  P.hx = (search->nx - 1) / 2;
  P.hy = (search->ny - 1) / 2;
  P.nbx = (search->nx + BNB_B - 1) / BNB_B;
  P.nby = (search->ny + BNB_B - 1) / BNB_B;
  P.S = L.S;
  P.pad = L.pad;
  P.pitch = L.pitch;
  P.rows = L.S + 2 * L.pad;
  P.max_shift = spec->max_shift;
  P.pool_pitch = L.pool_pitch;
  P.pool_rows = L.pool_rows;
  P.pairs_per_xcd = (n_pairs + 7) / 8;
  P.grid_bytes = L.grid_bytes;
  P.skip_bytes = L.skip_bytes;
  P.slot_bytes = L.slot_bytes;
  P.pool_bytes = L.pool_bytes;
  P.pool4_bytes = L.pool4_bytes;
  P.pool4_pitch = L.pool4_pitch;
  P.hi_offset = L.grid_bytes + L.skip_bytes + L.pool_bytes + L.pool4_bytes;
  P.hi_bytes = L.hi_bytes;
  P.hi_pitch = L.hi_pitch;
  P.hi_tpr = L.hi_tpr;
  P.hi_copy_bytes = L.hi_copy_bytes;
  P.t16_bytes = L.t16_bytes;
  P.t16_tpr = L.t16_tpr;
  // Policies that never change the records (tests run the matcher in every form and compare): read per launch.
  const char *lv = tunable("NHIP_BNB_LEVELS");  // (1: without the sub-block bounds)
  P.levels = lv && lv[0] == '1' ? 1 : 2;
  const char *qe = tunable("NHIP_BNB_QUEUE");  // (the general path for every scan)
  P.general_all = qe && qe[0] == '1';
  static_assert(NHIP_SHORT_SCAN_POINTS == 64 * OCL, "the header's promise is the by-rotation form's limit");
  P.short_scans = (search->flags & NHIP_SEARCH_SHORT_SCANS) != 0 && !P.general_all && (uint32_t)P.rows < ORG_LIMIT;
  // (the general instantiation -- scans of more than 1088 points, NHIP_BNB_QUEUE=1 -- takes its exact sums on the row-major image)
  NHIP_REQUIRE(L.has_image || P.short_scans, "csm_match: grids built with NHIP_GRID_NO_IMAGE serve lists whose scans all have at most %d "
               "points, and the caller must say so (NHIP_SEARCH_SHORT_SCANS; the handle API sets it itself)", NHIP_SHORT_SCAN_POINTS);
  P.res = spec->res;
  P.inv_res = 1.0 / spec->res;
  P.inv_res_f = (float)P.inv_res;
  const bool instr = instrumented();
  if (instr) {
    std::lock_guard<std::mutex> lock(g_instr_mu);
    const char *st = tunable("NHIP_BNB_STATS");
    if (st && st[0] == '1') {
      if (!g_bnb_stats) {
        NHIP_TRY_HIP(hipMalloc(reinterpret_cast<void **>(&g_bnb_stats), 8 * (BNB_STATS_HEAD + (size_t)BNB_STATS_PAIRS)));
        NHIP_TRY_HIP(hipMemset(g_bnb_stats, 0, 8 * (BNB_STATS_HEAD + (size_t)BNB_STATS_PAIRS)));
      }
      P.stats = g_bnb_stats;
    }
    const char *tl = tunable("NHIP_BNB_TIMELINE");
    if (tl && tl[0] == '1') {
      if (!g_bnb_timeline) NHIP_TRY_HIP(hipMalloc(reinterpret_cast<void **>(&g_bnb_timeline), 48 * (size_t)BNB_STATS_PAIRS + 16));
      P.timeline = g_bnb_timeline;
      // (the candidates' launch of the split form: first start / last end per pair)
      NHIP_TRY_HIP(hipMemsetAsync(g_bnb_timeline + 4 * (size_t)BNB_STATS_PAIRS + 2, 0xff, 8 * (size_t)BNB_STATS_PAIRS, s));
      NHIP_TRY_HIP(hipMemsetAsync(g_bnb_timeline + 5 * (size_t)BNB_STATS_PAIRS + 2, 0, 8 * (size_t)BNB_STATS_PAIRS, s));
      const unsigned long long init[2] = {~0ull, 0ull};  // the second kernel's first start and last end
      NHIP_TRY_HIP(hipMemcpyAsync(g_bnb_timeline + 4 * (size_t)BNB_STATS_PAIRS, init, 16, hipMemcpyHostToDevice, s));
    }
  }
  // Work sharing.  A flat landscape leaves a pair thousands of candidates (the median pair: ~30): alone on the
  // chip its workgroup is busy for 3 ms (the median pair: 0.2 ms), and a batch that does not fill the chip many
  // times over waits for it.  Such a pair (>= heavy_min candidates after bounds and seeds) works only its first
  // keep_ranks rotations in best-first order (one per wave) itself and hands the others, with the masks of their
  // candidate blocks, to per-XCD lists in the caller's workspace; a second kernel works the lists with every wave of
  // the chip, sharing the pair's running best through keys[pair].  Measured (tools/bnb_heavy.py, bnb_quick.py;
  // profiles/r02_bnb_heavy.json): the heaviest pair alone 2.2 -> 0.94 ms; 30 pairs 0.80 -> 0.45 ms; 500 pairs + the
  // three heaviest 2.5 -> 1.3 ms.  From ~1000 pairs on the chip is full anyway and handing over only loses pruning
  // and L2 locality (2,000 pairs 4.6 -> 7.3 ms, 10,000 pairs unchanged), so large batches do not.  (Also tried:
  // letting the waves of finished workgroups take entries inside the first kernel, and persistent workgroups -- never
  // a gain.)
  // NHIP_BNB_KERNELS=1: never, =2: always; NHIP_BNB_HEAVY_MIN=<candidates>, NHIP_BNB_KEEP_RANKS=<n>.
  const char *force = tunable("NHIP_BNB_KERNELS");
  const char *hm = tunable("NHIP_BNB_HEAVY_MIN");
  const char *kr = tunable("NHIP_BNB_KEEP_RANKS");
  // (lists that take the split form -- SPLIT_MIN_PAIRS pairs and more -- do not hand rotations over: their candidates'
  //  launch shares the heavy pairs among several workgroups)
  // The form is decided ONCE, here, from the sizes: the split form's rounds as the workspace allows them, and the
  // hand-over lists only for lists that do not take the split form.  (Round 4 asked "is there room for 512 pairs' state"
  // at this point and sized the rounds further down.  nhip_csm_workspace_bytes(n) of a list of 192 .. 487 pairs at 61
  // rotations is LESS than 512 pairs' state, so for those lists the first test said no, the hand-over lists were set
  // up, and the split form -- which the header promises from 192 pairs -- was never taken.  Same records; slower.)
  const char *sp = tunable("NHIP_BNB_SPLIT");
  const char *sbat = tunable("NHIP_BNB_SPLIT_BATCH");
  const int64_t split_cap = SPLIT_PAIRS;
  int64_t split_batch = 0, split_slots = 0, slot_bytes = 0;
  if (d_workspace && !P.general_all && !(sp && sp[0] == '0') && !(force && force[0] == '2') &&
      (n_pairs >= SPLIT_MIN_PAIRS || (sp && sp[0] == '1'))) {
    split_batch = sbat && atoi(sbat) > 0 ? atoi(sbat) : split_cap;
    if (split_batch > n_pairs) split_batch = n_pairs;
    for (;;) {  // (a workspace too small for two batches in flight: smaller batches, down to 512 pairs)
      slot_bytes = (SPLIT_SLOT_FIXED + split_batch * split_bytes_per_pair(P.n_theta) + 511) & ~(int64_t)511;
      split_slots = (workspace_bytes - BNB_WS_HEADER - 512) / slot_bytes;
      const int64_t rounds = (n_pairs + split_batch - 1) / split_batch;
      if (split_slots >= (rounds < 2 ? rounds : 2) || split_batch <= 512) break;
      split_batch = split_batch / 2 > 512 ? split_batch / 2 : 512;
    }
    if (split_slots > SPLIT_RING) split_slots = SPLIT_RING;
    if (split_slots < 1) split_batch = 0;  // (no room: the fused form)
    // (several rounds pay off only with the helper stream, i.e. with two rounds' state, and in rounds that are long)
    if (!sbat && split_batch > 0 && n_pairs > split_batch && (split_slots < 2 || split_batch < split_cap)) split_batch = 0;
  }
  P.heavy_min = hm ? (uint32_t)atoi(hm) : (n_pairs <= 64 ? 1u : 384u);
  P.keep_ranks = kr ? (uint32_t)atoi(kr) : 8u;
  // (a pair keeps its first keep_ranks rotations: a search of no more rotations than that -- the seven per workgroup of
  //  GetTransformation's fine level -- can hand nothing over, and the second kernel's launch, 22 us of a 230 us call, is left out)
  const bool second = force ? force[0] == '2' : (n_pairs < 1024 && split_batch == 0 && (uint32_t)P.n_theta > P.keep_ranks);
  if (d_workspace && workspace_bytes >= BNB_WS_HEADER + 8 * (int64_t)sizeof(RotEntry) && second && !P.general_all) {
    P.rot_count = static_cast<uint32_t *>(d_workspace);
    P.rot_list = reinterpret_cast<RotEntry *>(static_cast<uint8_t *>(d_workspace) + BNB_WS_HEADER);
    const int64_t cap = (workspace_bytes - BNB_WS_HEADER) / (int64_t)sizeof(RotEntry) / 8;  // entries per XCD list
    P.rot_cap = (uint32_t)(cap < 0x0fffffffll ? cap : 0x0fffffffll);
    NHIP_TRY_HIP(hipMemsetAsync(d_workspace, 0, BNB_WS_HEADER, s));
  }
  const bool pool_lds = bnb_lds_bytes(L, search, true) <= LDS_MAX;
  size_t lds = bnb_lds_bytes(L, search, pool_lds);
  P.lds_first = (int32_t)bnb_lds_first(L, pool_lds);
  const int64_t blocks = (int64_t)P.pairs_per_xcd * 8;
  const bool second_kernel = P.rot_list != nullptr;
  // Large batches: the split form, in rounds of as many pairs as the workspace holds state for.
  // NHIP_BNB_SPLIT=0: never, =1: whenever the workspace allows; NHIP_BNB_SPLIT_MIN=<candidates per additional
  // workgroup of a pair>, NHIP_BNB_SPLIT_MAX=<workgroups per pair>.
  const char *smin = tunable("NHIP_BNB_SPLIT_MIN");
  const char *smax = tunable("NHIP_BNB_SPLIT_MAX");
  const char *sfront = tunable("NHIP_BNB_FRONT_MIN");  // (measurement: pairs with at least that many candidates first)
  const char *sov = tunable("NHIP_BNB_SPLIT_OVERLAP");
  if (P.rot_list) split_batch = 0;  // (hand-over lists in the workspace: one kernel per pair)
  // (an error return between timer_begin and timer_end closes the open slot)
  struct TimerScope {
    int id;
    hipStream_t s;
    bool open = true;
    TimerScope(int i, hipStream_t st) : id(i), s(st) { timer_begin(id, s); }
    void end() {
      if (open) timer_end(id, s);
      open = false;
    }
    ~TimerScope() { end(); }
  };
  TimerScope t_all(NHIP_TIMER_CSM, s);
  {
    const int64_t rounds = split_batch > 0 ? (n_pairs + split_batch - 1) / split_batch : 1;
    const bool ov = split_batch > 0 && !(sov && sov[0] == '0') && split_slots >= 2 && n_pairs > split_batch;
    const int32_t info[8] = {split_batch > 0 ? (ov ? 3 : (rounds > 1 ? 2 : 1)) : 0, (int32_t)split_batch, (int32_t)split_slots,
                             (int32_t)rounds, P.short_scans, second_kernel ? 1 : 0, instr ? 1 : 0, n_pairs};
    memcpy(t_last_launch, info, sizeof(info));
  }
  if (split_batch > 0) {
    // Candidates (bound by the L1's lookups) beside the next batch's bounds (bound by the vector ALUs): the first part
    // of every batch on the caller's stream, the second on the helper stream, each batch's state in its own slot of the
    // workspace.  NHIP_BNB_SPLIT_OVERLAP=0: everything on the caller's stream.
    const bool overlap = !(sov && sov[0] == '0') && split_slots >= 2 && n_pairs > split_batch;
    SplitSet *set = nullptr;
    if (overlap) {
      const int rc = split_set_acquire(&set);
      if (rc) return rc;
    }
    struct SetGuard {
      SplitSet *p;
      ~SetGuard() { split_set_release(p); }
    } set_guard{set};
    hipStream_t s2 = overlap ? set->stream : s;
    uint8_t *base = static_cast<uint8_t *>(d_workspace) + BNB_WS_HEADER;
    base += (512 - (reinterpret_cast<uintptr_t>(base) & 511)) & 511;
    int64_t round = 0;
    for (int64_t b0 = 0; b0 < n_pairs; b0 += split_batch, round++) {
      const int32_t nb = (int32_t)(n_pairs - b0 < split_batch ? n_pairs - b0 : split_batch);
      BnbParams Q = P;
      Q.pair_src += b0;
      Q.pair_slot += b0;
      Q.rot0_cs += 2 * b0;
      if (Q.pair_origin) Q.pair_origin += 2 * b0;
      Q.keys += b0;
      Q.pair_base = (int32_t)b0;  // (what nhip_dev_status names is an index into the CALLER's arrays)
      Q.n_pairs = nb;
      Q.pairs_per_xcd = (nb + 7) / 8;
      Q.ps_work_stride = Q.pairs_per_xcd + (Q.pairs_per_xcd / 2 > 64 ? Q.pairs_per_xcd / 2 : 64);
      Q.split_min = smin ? (uint32_t)(atoi(smin) > 0 ? atoi(smin) : 1) : 300u;
      // (workgroups per pair at most.  With the heavy pairs at the front of the list -- front_min below -- their workgroups
      //  start with the launch, and rounds of 8,192 pairs and more are long enough for four of them to finish the heaviest
      //  pair inside it: 8 / 6 / 5 / 4 / 3 per pair measured 2.87 / 2.84 / 2.80 / 2.79 / 2.94 ms on configs[1], 16-bit cells,
      //  2.86 / 3.00 / 2.79 / 2.72 / 2.79 with 8-bit cells: profiles/r06_front_min.txt)
      Q.split_max = smax ? (uint32_t)(atoi(smax) > 0 ? atoi(smax) : 1) : (nb < SPREAD_BELOW_PAIRS ? 16u : (nb >= 8192 ? 4u : 8u));
      // pairs with at least 64 candidates left (a quarter of a configs[1] list, most of its work) go to the front of their
      // XCD's list: profiles/r06_front_min.txt -- 0 / 40 / 70 / 100 / 150: 3.01 / 2.90 / 2.87 / 2.92 / 2.96 ms per 10,000 pairs
      Q.front_min = sfront ? (uint32_t)(atoi(sfront) > 0 ? atoi(sfront) : 0) : 64u;
      uint8_t *w = base + (round % split_slots) * slot_bytes;
      Q.ps_count = reinterpret_cast<uint32_t *>(w);
      Q.ps_live = Q.ps_count + nb;
      Q.ps_next = Q.ps_live + nb;
      Q.ps_nw = Q.ps_next + nb;
      // (spread form: the ticket counter of the additional workgroups, zeroed with the four arrays before it)
      const bool spread = nb < SPREAD_BELOW_PAIRS;
      Q.ps_ticket = spread ? Q.ps_nw + nb : nullptr;
      Q.ps_work = reinterpret_cast<int32_t *>(Q.ps_nw + nb + 4);
      const uintptr_t rows = (reinterpret_cast<uintptr_t>(Q.ps_work + 8 * (size_t)Q.ps_work_stride) + 511) & ~(uintptr_t)511;
      Q.ps_rows = reinterpret_cast<uint32_t *>(rows);
      NHIP_REQUIRE((int64_t)(rows - reinterpret_cast<uintptr_t>(w)) + (int64_t)nb * P.n_theta * 512 <= slot_bytes &&
                       w + slot_bytes <= static_cast<uint8_t *>(d_workspace) + workspace_bytes,
                   "csm_bnb: workspace accounting");
      // (the slot's previous batch must be through its candidates)
      if (overlap && round >= split_slots) NHIP_TRY_HIP(hipStreamWaitEvent(s, set->eb[(round - split_slots) % SPLIT_RING], 0));
      NHIP_TRY_HIP(hipMemsetAsync(w, 0, 16 * (size_t)nb + 16, s));
      if (spread) NHIP_TRY_HIP(hipMemsetAsync(Q.ps_work, 0xff, 32 * (size_t)Q.ps_work_stride, s));  // (-1: no pair)
      const int64_t blocks_b = (int64_t)Q.pairs_per_xcd * 8;
      {
        TimerScope t_a(NHIP_TIMER_CSM_BOUNDS, s);
        const int rc = instr ? bnb::launch_bnb_split_a_instr(Q, L.cb, pool_lds, lds, blocks_b, s)
                             : bnb::launch_bnb_split_a(Q, L.cb, pool_lds, lds, blocks_b, s);
        if (rc) return rc;
      }
      if (overlap) {
        NHIP_TRY_HIP(hipEventRecord(set->ea[round % SPLIT_RING], s));
        NHIP_TRY_HIP(hipStreamWaitEvent(s2, set->ea[round % SPLIT_RING], 0));
      }
      {
        TimerScope t_b(NHIP_TIMER_CSM_CAND, s2);
        const int rc = instr ? bnb::launch_bnb_split_b_instr(Q, L.cb, s2) : bnb::launch_bnb_split_b(Q, L.cb, s2);
        if (rc) return rc;
      }
      if (overlap) NHIP_TRY_HIP(hipEventRecord(set->eb[round % SPLIT_RING], s2));
    }
    // (the helper stream works in order: its last batch done, all are)
    if (overlap) NHIP_TRY_HIP(hipStreamWaitEvent(s, set->eb[(round - 1) % SPLIT_RING], 0));
    t_all.end();
    NHIP_TRY_HIP(hipGetLastError());
    launch_csm_finalize(d_keys, d_pair_src, d_offsets, ids.n_scans, n_pairs, P.nx, P.ny, L, d_out, d_sums, s);
    NHIP_TRY_HIP(hipGetLastError());
    return NHIP_OK;
  }
  // (Tried and removed: the batch as K launches on K streams, so that one hardware queue's in-order dispatch does not
  //  keep free slots empty -- 2 / 4 / 8 queues took 10 / 30 / 45 % longer, profiles/r03_matcher_experiments.txt.)
  const int rc = instr ? bnb::launch_bnb_kernels_instr(P, L.cb, pool_lds, lds, blocks, second_kernel, s)
                       : bnb::launch_bnb_kernels(P, L.cb, pool_lds, lds, blocks, second_kernel, s);
  if (rc) return rc;
  t_all.end();
  NHIP_TRY_HIP(hipGetLastError());
  launch_csm_finalize(d_keys, d_pair_src, d_offsets, ids.n_scans, n_pairs, P.nx, P.ny, L, d_out, d_sums, s);
  NHIP_TRY_HIP(hipGetLastError());
  return NHIP_OK;
}

// NHIP_BNB_STATS=1: (blocks evaluated exactly, blocks in all) since the last call; resets the counters
int bnb_stats_per_pair(unsigned long long *out, int32_t n) {
  if (!g_bnb_stats || n <= 0) return NHIP_OK;
  NHIP_TRY_HIP(hipMemcpy(out, g_bnb_stats + BNB_STATS_HEAD, 8 * (size_t)(n < BNB_STATS_PAIRS ? n : BNB_STATS_PAIRS), hipMemcpyDeviceToHost));
  return NHIP_OK;
}

int bnb_timeline_read(unsigned long long *out, int32_t n) {
  if (!g_bnb_timeline || n <= 0) return NHIP_OK;
  if (n > BNB_STATS_PAIRS) n = BNB_STATS_PAIRS;
  NHIP_TRY_HIP(hipMemcpy(out, g_bnb_timeline, 32 * (size_t)n, hipMemcpyDeviceToHost));
  // (the last pair's slot is followed by the second kernel's first start / last end)
  NHIP_TRY_HIP(hipMemcpy(out + 4 * (size_t)n, g_bnb_timeline + 4 * (size_t)BNB_STATS_PAIRS, 16, hipMemcpyDeviceToHost));
  return NHIP_OK;
}

int bnb_timeline_cand_read(unsigned long long *out, int32_t n) {  // out[0..n): first start, out[n..2n): last end
  if (!g_bnb_timeline || n <= 0) return NHIP_OK;
  if (n > BNB_STATS_PAIRS) n = BNB_STATS_PAIRS;
  NHIP_TRY_HIP(hipMemcpy(out, g_bnb_timeline + 4 * (size_t)BNB_STATS_PAIRS + 2, 8 * (size_t)n, hipMemcpyDeviceToHost));
  NHIP_TRY_HIP(hipMemcpy(out + n, g_bnb_timeline + 5 * (size_t)BNB_STATS_PAIRS + 2, 8 * (size_t)n, hipMemcpyDeviceToHost));
  return NHIP_OK;
}

int bnb_stats_read(unsigned long long out[16]) {
  for (int i = 0; i < BNB_STATS_HEAD; i++) out[i] = 0;
  if (!g_bnb_stats) return NHIP_OK;
  NHIP_TRY_HIP(hipMemcpy(out, g_bnb_stats, 8 * BNB_STATS_HEAD, hipMemcpyDeviceToHost));
  NHIP_TRY_HIP(hipMemset(g_bnb_stats, 0, 8 * BNB_STATS_HEAD));
  return NHIP_OK;
}


}  // namespace nhip
