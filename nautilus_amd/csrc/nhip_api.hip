// nhip_api.hip -- extern "C" shim of libnautilus_hip (include/nautilus_hip.h): argument
// checking, host-side spec tables, handle objects, in-stream kernel timing.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstddef>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>

#include "nhip_common.h"

namespace nhip {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
  return NHIP_ERR_HIP;
}

int require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device visible (hipGetDeviceCount -> %d, n = %d): this library has no CPU path",
              (int)e, n);
    return NHIP_ERR_NODEV;
  }
  return NHIP_OK;
}

// The status words of a device (nhip_common.h, "ids that live in device memory"): 16 bytes of device memory per device,
// kept to the end of the process.  They are allocated and zeroed by dev_status_prepare() -- called from nhip_init (every
// visible device) and nhip_set_device, i.e. OUTSIDE any launch path -- so that dev_status(), which every `_dev` launcher
// calls, is a lookup: no hipMalloc, no null-stream memset, nothing a stream capture could trip over.  A process that hands
// the library a device it never named to nhip_init / nhip_set_device gets the words on that device's first launch (the
// lazy path below; it synchronises once, and must not be the first thing inside a capture: include/nautilus_hip.h says so).
namespace {
constexpr int MAX_DEV = 64;
std::mutex g_status_mu;
std::atomic<uint32_t *> g_status_words[MAX_DEV];
}  // namespace

uint32_t *dev_status_prepare(int dev) {
  if (dev < 0 || dev >= MAX_DEV) return nullptr;
  if (uint32_t *w = g_status_words[dev].load(std::memory_order_acquire)) return w;
  std::lock_guard<std::mutex> lock(g_status_mu);
  if (uint32_t *w = g_status_words[dev].load(std::memory_order_acquire)) return w;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (cur != dev && hipSetDevice(dev) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  void *p = nullptr;
  bool ok = hipMalloc(&p, sizeof(uint32_t) * DEV_STATUS_WORDS) == hipSuccess;
  if (ok && hipMemset(p, 0, sizeof(uint32_t) * DEV_STATUS_WORDS) != hipSuccess) {
    (void)hipFree(p);  // (not leaked, and not retried with the same pointer)
    ok = false;
  }
  if (!ok) (void)hipGetLastError();
  if (cur != dev) (void)hipSetDevice(cur);
  if (!ok) return nullptr;
  g_status_words[dev].store(static_cast<uint32_t *>(p), std::memory_order_release);
  return static_cast<uint32_t *>(p);
}

uint32_t *dev_status() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (uint32_t *w = g_status_words[dev].load(std::memory_order_acquire)) return w;
  return dev_status_prepare(dev);
}

const char *tunable(const char *name) {
  static std::once_flag once;
  static bool on = false;
  std::call_once(once, [] {
    const char *e = getenv("NHIP_TUNABLES");
    on = e && e[0] == '1';
  });
  return on ? getenv(name) : nullptr;
}

// ---------------------------------------------------------------- spec tables (host)
int make_layout(const nhip_grid_spec_t *spec, GridLayout *L) {
  NHIP_REQUIRE(spec != nullptr && L != nullptr, "grid spec: null pointer");
  NHIP_REQUIRE(spec->range > 0 && spec->res > 0, "grid spec: range and res must be > 0");
  NHIP_REQUIRE(spec->sigma > 0 && spec->sigma <= 16.0 / 3.0, "grid spec: sigma must be in (0, 5.33] cells");
  NHIP_REQUIRE(spec->floor_p > 0 && spec->floor_p < 1, "grid spec: floor_p must be in (0, 1)");
  NHIP_REQUIRE(spec->max_shift >= 0 && spec->max_shift <= 4096, "grid spec: max_shift out of range");
  NHIP_REQUIRE(spec->cell_bits == 0 || spec->cell_bits == 8 || spec->cell_bits == 16,
               "grid spec: cell_bits must be 8 or 16 (0 = 16), got %d", spec->cell_bits);
  NHIP_REQUIRE((spec->flags & ~(NHIP_GRID_SKIP_MAP | NHIP_GRID_NO_IMAGE)) == 0 && spec->reserved == 0,
               "grid spec: unknown flags %d / reserved %d", spec->flags, spec->reserved);
  NHIP_REQUIRE((spec->flags & (NHIP_GRID_SKIP_MAP | NHIP_GRID_NO_IMAGE)) != (NHIP_GRID_SKIP_MAP | NHIP_GRID_NO_IMAGE),
               "grid spec: a skip map (the every-add kernels') needs the image NHIP_GRID_NO_IMAGE leaves out");
  const double side = floor((spec->range * 2.0) / spec->res);  // cimg_debug.h:21-22
  NHIP_REQUIRE(side >= 1 && side <= 16384, "grid spec: side %g out of range [1, 16384]", side);
  L->S = (int32_t)side;
  L->cb = spec->cell_bits == 8 ? 1 : 2;  // (0 = the default: 16-bit cells, in every struct of the ABI)
  L->levels = L->cb == 2 ? 65535 : 255;
  L->pad = ((2 * spec->max_shift + 16) + 3) & ~3;
  L->pitch = ((L->S + 2 * L->pad) * L->cb + 15) & ~15;
  L->R = (int32_t)ceil(3.0 * spec->sigma);
  L->plain_bytes = (int64_t)L->pitch * (int64_t)(L->S + 2 * L->pad);
  L->has_image = !(spec->flags & NHIP_GRID_NO_IMAGE);
  L->grid_bytes = L->has_image ? L->plain_bytes : 0;
  L->skip_bytes = L->has_image ? (((int64_t)skip_pitch(L->pitch) * (int64_t)(L->S + 2 * L->pad)) + 15) & ~15ll : 0;
  // pooled table: one byte per 8 x 8 stored cells, + BNB_MAX_NB rows / + BNB_MAX_NB + 5 columns of zeros so that a
  // window origin anywhere in the stored image can read its 11 x 16-byte rows without bounds checks
  L->pool_rows = (L->S + 2 * L->pad + BNB_B - 1) / BNB_B + BNB_MAX_NB + 1;
  L->pool_pitch = (((L->S + 2 * L->pad + BNB_B - 1) / BNB_B + BNB_MAX_NB + 5) + 15) & ~15;
  L->pool_bytes = (int64_t)L->pool_rows * L->pool_pitch;
  // second-level table: per 4 x 4 stored cells a PAIR of bytes {P4[i][j], P4[i + 1][j]} (the two sub-block rows of a
  // block in one read); a block's sub-blocks reach 2 * BNB_MAX_NB entries past the origin's, and a strip of three
  // blocks is read as 16 bytes from a 4-byte-aligned offset
  L->pool4_rows = (L->S + 2 * L->pad + BNB_B4 - 1) / BNB_B4 + 2 * BNB_MAX_NB + 2;
  L->pool4_pitch = ((2 * ((L->S + 2 * L->pad + BNB_B4 - 1) / BNB_B4 + 2 * BNB_MAX_NB + 2) + 16) + 15) & ~15;
  L->pool4_bytes = (int64_t)L->pool4_rows * L->pool4_pitch;
  // The matcher's planes behind the two tables are stored ONE CACHE LINE PER TILE (128 bytes): they must start on a line
  // boundary in EVERY slot, or each tile read touches two lines.  (Round 5 appended the hit raster, whose size is not a
  // multiple of 128: slots 1, 2, ... started 16, 32, ... bytes off a line, and the candidates kernel's L2 fetch went from 0.86
  // to 1.39 GB per 10,000 pairs -- profiles/r06_cand_traffic_bisect.txt.)  So the second table is padded to the next line
  // boundary of the slot here, and the raster -- the slot's last part -- below.
  L->pool4_bytes += (128 - ((L->grid_bytes + L->skip_bytes + L->pool_bytes + L->pool4_bytes) & 127)) & 127;
  // 16-bit cells: the plane of their high bytes, one byte per cell at the 8-bit pitch.  The matcher sums exact 8 x 8
  // and 4 x 4 blocks on this plane at the cost of 8-bit cells (256 * sum(hi) + 255 * points bounds a pose's sum from
  // above) and reads 16-bit cells only for the poses that bound still admits (nhip_bnb.hip)
  // (8-bit cells: the same two tiled copies hold the cells themselves -- the image's bytes)
  L->hi_pitch = ((L->S + 2 * L->pad) + 15) & ~15;
  L->hi_tpr = L->hi_pitch / 16 + 1;  // (+ 1: the shifted copy's last tile)
  L->hi_copy_bytes = (int64_t)((L->S + 2 * L->pad + 7) / 8) * L->hi_tpr * (int64_t)HI_TILE_BYTES;
  L->t16_tpr = L->cb == 2 ? L->hi_pitch / 8 : 0;
  L->t16_bytes = (int64_t)((L->S + 2 * L->pad + 7) / 8) * L->t16_tpr * (int64_t)HI_TILE_BYTES;
  L->hi_bytes = 2 * L->hi_copy_bytes + L->t16_bytes;  // (the matcher's private planes, all three)
  // the hit raster, one bit per cell + a zero border of HIT_PAD cells: bit rows of whole dwords
  L->hits_pitch = ((L->S + 2 * HIT_PAD + 31) / 32) * 4;
  L->hits_bytes = (((int64_t)L->hits_pitch * (L->S + 2 * HIT_PAD) + 8) + 15) & ~15ll;  // (+ 8: a row's last 64-bit window)
  L->slot_bytes = L->grid_bytes + L->skip_bytes + L->pool_bytes + L->pool4_bytes + L->hi_bytes + L->hits_bytes;
  L->hits_bytes += (128 - (L->slot_bytes & 127)) & 127;  // (every slot starts on a line boundary: see pool4_bytes above)
  if (const char *sp = tunable("NHIP_GRID_SLOT_PAD")) L->hits_bytes += (int64_t)(atoi(sp) > 0 ? atoi(sp) : 0) * 128;  // (measurement: slot stride vs. channels)
  L->slot_bytes = L->grid_bytes + L->skip_bytes + L->pool_bytes + L->pool4_bytes + L->hi_bytes + L->hits_bytes;
  L->Lf = log(spec->floor_p);
  L->step = -L->Lf / (double)L->levels;
  // integer taps: round(16384 * g_i / sum g)
  double g[129], tot = 0.0;
  for (int i = -L->R; i <= L->R; i++) {
    g[i + L->R] = exp(-((double)i * (double)i) / (2.0 * spec->sigma * spec->sigma));
    tot += g[i + L->R];
  }
  L->K = 0;
  for (int i = 0; i <= 2 * L->R; i++) L->K += (int64_t)floor(16384.0 * g[i] / tot + 0.5);
  return NHIP_OK;
}

// The quantiser of the spec, evaluated directly (used only to build the threshold table).
static uint32_t quantise_direct(uint64_t V, int64_t K, double floor_p, int32_t levels) {
  double v = (double)V / ((double)K * (double)K);
  if (v < floor_p) v = floor_p;
  const double Lf = log(floor_p);
  const double step = -Lf / (double)levels;
  double q = floor((log(v) - Lf) / step + 0.5);
  if (q < 0.0) q = 0.0;
  if (q > (double)levels) q = (double)levels;
  return (uint32_t)q;
}

// thr[k] = smallest integer V in [0, K*K] whose quantised value is >= k (0xffffffff: level never reached).
// q is non-decreasing in V, so a binary search per level is exact; it starts from a bracket around the
// closed-form inverse so that the 65535 levels of the 16-bit table cost a few log() calls each.
static void make_thresholds(int64_t K, double floor_p, int32_t levels, uint32_t *thr) {
  const uint64_t vmax = (uint64_t)K * (uint64_t)K;
  const double Lf = log(floor_p), step = -Lf / (double)levels;
  auto q = [&](uint64_t V) { return quantise_direct(V, K, floor_p, levels); };
  const uint32_t q0 = q(0), qmax = q(vmax);
  thr[0] = 0;
  for (int32_t k = 1; k <= levels; k++) {
    if (qmax < (uint32_t)k) { thr[k] = 0xffffffffu; continue; }
    if (q0 >= (uint32_t)k) { thr[k] = 0; continue; }
    // invariant: q(lo) < k <= q(hi)
    uint64_t lo = 0, hi = vmax;
    const double est = (double)vmax * exp(Lf + ((double)k - 0.5) * step);
    if (est > 2.0 && est < (double)vmax) {
      const uint64_t e = (uint64_t)est, w = e / 1000000 + 2;  // ~1e-6 relative bracket
      const uint64_t a = e > w ? e - w : 0, b = e + w < vmax ? e + w : vmax;
      if (q(a) < (uint32_t)k) lo = a;
      if (q(b) >= (uint32_t)k) hi = b;
    }
    while (hi - lo > 1) {
      const uint64_t mid = lo + (hi - lo) / 2;
      if (q(mid) >= (uint32_t)k) hi = mid; else lo = mid;
    }
    thr[k] = (uint32_t)hi;
  }
}

int make_tables(const nhip_grid_spec_t *spec, const GridLayout &L, GridTables *T) {
  memset(T, 0, sizeof(*T));
  double g[129], tot = 0.0;
  for (int i = -L.R; i <= L.R; i++) {
    g[i + L.R] = exp(-((double)i * (double)i) / (2.0 * spec->sigma * spec->sigma));
    tot += g[i + L.R];
  }
  for (int i = 0; i <= 2 * L.R; i++) T->taps[i] = (int32_t)floor(16384.0 * g[i] / tot + 0.5);
  if (L.cb == 1) {
    make_thresholds(L.K, spec->floor_p, 255, T->thr);
    return NHIP_OK;
  }
  // 16-bit cells: 65536 entries, computed once per (tap sum, floor) and kept for the life of the process
  static std::mutex mu;
  static std::map<std::pair<int64_t, double>, std::vector<uint32_t>> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto &v = cache[{L.K, spec->floor_p}];
  if (v.empty()) {
    v.resize(65536);
    make_thresholds(L.K, spec->floor_p, 65535, v.data());
  }
  T->thr16 = v.data();
  return NHIP_OK;
}

// ---------------------------------------------------------------- in-stream timing
namespace {
struct TimerSlot {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  hipEvent_t open = nullptr;
};
std::mutex g_tmu;
bool g_timing = false;
TimerSlot g_slots[NHIP_TIMER_COUNT];
}  // namespace

void timer_begin(int id, hipStream_t s) {
  if (!g_timing) return;
  std::lock_guard<std::mutex> lk(g_tmu);
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  g_slots[id].open = e;
}

void timer_end(int id, hipStream_t s) {
  if (!g_timing) return;
  std::lock_guard<std::mutex> lk(g_tmu);
  if (!g_slots[id].open) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, s);
  g_slots[id].ev.emplace_back(g_slots[id].open, e);
  g_slots[id].open = nullptr;
}

// ---------------------------------------------------------------- host phases of the handle API
// Wall-clock seconds of the calling thread's LAST handle-API call (nhip_scans_upload, nhip_grids_build, nhip_csm_match,
// the *_free calls), by what the host was waiting for: nhip_host_phases().  Round 4's bench saw one call in five of the
// host-buffer route take 4 s instead of 12 ms and could not say where (a median hid it); the clocks cost two
// steady_clock reads per phase.
enum { PH_ALLOC = 0, PH_UPLOAD, PH_ENQUEUE, PH_WAIT, PH_DOWNLOAD, PH_FREE, PH_HOST, PH_COUNT };
static thread_local double t_phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
struct PhaseClock {
  int id;
  std::chrono::steady_clock::time_point t0;
  explicit PhaseClock(int i) : id(i), t0(std::chrono::steady_clock::now()) {}
  ~PhaseClock() { t_phase[id] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
static void phases_reset() {
  for (double &v : t_phase) v = 0.0;
}

// ---------------------------------------------------------------- device buffers of the handle API
// The handle entry points own their device memory and used to hipMalloc / hipFree it per call.  Measured in round 5
// (tools/r05_host_api_stall.py, profiles/r05_host_api_stall.txt): on a quiet device the pair costs microseconds, but
// hipFree is a device-wide synchronisation whose cost is the driver's -- 0.2 to 8 ms per nhip_csm_match call for its 328 MB
// of workspace, and 0.33 s PER hipFree of the 12 GB of tables for seconds after another client of the process (torch's
// caching allocator) had released 130 GB; round 4's bench saw one host-buffer call in five take 4 s.  So released buffers
// are kept, per device, up to a byte cap (nhip_device_pool_configure; least recently released out first) and handed to
// the next allocation they fit (at most twice the size asked for): a loop of build / match / free touches the driver's
// allocator once.  Contents are never assumed -- every user initialises what it reads -- with one exception, stated on the
// entries themselves: a table buffer and its build workspace released together (PoolEntry::key).
namespace {
struct PoolEntry {
  void *p;
  size_t bytes;
  int device;
  // A table buffer and the workspace of the build that filled it may come back TOGETHER with their contents known: both carry
  // the key of that release (0: contents unknown) and a digest of (spec, targets, which of the two).  While both are still in
  // the pool nobody has written to either, so a build of the same shape takes the pair and rebuilds incrementally -- it clears
  // the lines the previous build wrote instead of zero-filling gigabytes (pool_take_pair).
  uint64_t key = 0, meta = 0;
};
std::mutex g_pool_mu;
std::vector<PoolEntry> &g_pool = *new std::vector<PoolEntry>();  // oldest first (never destroyed: see the drop-in cache below)
// PER DEVICE.  4 GB by default: enough for the workspace of a 10,000-pair match (0.33 GB), the tables of ~450 targets and the
// small per-call buffers -- what a drop-in host that called the library once can defend holding.  A host that cycles
// larger tables (bench.py's host-buffer leg: 8.3 GB) raises it with nhip_device_pool_configure.
int64_t g_pool_cap = 4ll << 30;
constexpr size_t POOL_MAX_ENTRIES = 32;

int current_device() {
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return d;
}
void *pool_take(size_t n, size_t *got) {
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(g_pool_mu);
  // best fit; a buffer whose contents are known (half of a kept pair, see PoolEntry) only when nothing else fits
  size_t best = g_pool.size();
  for (int keyed = 0; keyed < 2 && best == g_pool.size(); keyed++)
    for (size_t i = 0; i < g_pool.size(); i++)
      if (g_pool[i].device == dev && (g_pool[i].key != 0) == (keyed == 1) && g_pool[i].bytes >= n &&
          g_pool[i].bytes <= 2 * n + (1u << 20) && (best == g_pool.size() || g_pool[i].bytes < g_pool[best].bytes))
        best = i;
  if (best == g_pool.size()) return nullptr;
  void *p = g_pool[best].p;
  *got = g_pool[best].bytes;
  g_pool.erase(g_pool.begin() + (long)best);
  return p;
}
// true: the pool keeps the buffer; `evict` receives what it lets go of for it (freed by the caller, outside the lock)
// `dev`: the device the buffer was ALLOCATED on (DevBuf records it) -- not the device that happens to be current when it
// is released: a handle built on device 0 and freed after nhip_set_device(1) stays a device-0 buffer.  Cap and entry limit
// are per device: releases on one GPU never evict what another GPU's callers keep.
bool pool_put(void *p, size_t bytes, int dev, std::vector<void *> *evict, uint64_t key = 0, uint64_t meta = 0) {
  std::lock_guard<std::mutex> lock(g_pool_mu);
  if (dev < 0 || (int64_t)bytes > g_pool_cap) return false;
  g_pool.push_back({p, bytes, dev, key, meta});
  int64_t tot = 0;
  size_t cnt = 0;
  for (auto &e : g_pool)
    if (e.device == dev) {
      tot += (int64_t)e.bytes;
      cnt++;
    }
  for (size_t i = 0; i < g_pool.size() && (tot > g_pool_cap || cnt > POOL_MAX_ENTRIES);) {
    if (g_pool[i].device != dev) {
      i++;
      continue;
    }
    tot -= (int64_t)g_pool[i].bytes;  // (oldest of this device first)
    cnt--;
    evict->push_back(g_pool[i].p);
    g_pool.erase(g_pool.begin() + (long)i);
  }
  return true;
}
// the two buffers of one keyed release (metas `meta_a`, `meta_b`), if both are still here
bool pool_take_pair(uint64_t meta_a, uint64_t meta_b, size_t need_a, size_t need_b, void **pa, size_t *ba, void **pb, size_t *bb) {
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(g_pool_mu);
  for (size_t i = g_pool.size(); i-- > 0;) {  // (newest first)
    if (g_pool[i].device != dev || g_pool[i].key == 0 || g_pool[i].meta != meta_a || g_pool[i].bytes < need_a) continue;
    for (size_t j = 0; j < g_pool.size(); j++) {
      if (j == i || g_pool[j].device != dev || g_pool[j].key != g_pool[i].key || g_pool[j].meta != meta_b || g_pool[j].bytes < need_b)
        continue;
      *pa = g_pool[i].p; *ba = g_pool[i].bytes;
      *pb = g_pool[j].p; *bb = g_pool[j].bytes;
      g_pool.erase(g_pool.begin() + (long)(i > j ? i : j));
      g_pool.erase(g_pool.begin() + (long)(i > j ? j : i));
      return true;
    }
  }
  return false;
}
std::atomic<uint64_t> g_pool_key{1};
void pool_drain(std::vector<void *> *out, int device /* -1: every device */) {
  std::lock_guard<std::mutex> lock(g_pool_mu);
  for (size_t i = 0; i < g_pool.size();)
    if (device < 0 || g_pool[i].device == device) {
      out->push_back(g_pool[i].p);
      g_pool.erase(g_pool.begin() + (long)i);
    } else {
      i++;
    }
}
}  // namespace

// Work this thread's current handle call has enqueued may still be running: set by the entry points that launch kernels
// on buffers they own (InFlight), cleared once they have synchronised.  DevBuf::free used to be a hipFree -- an implicit
// device-wide synchronisation -- and is now a hand-over to the pool: on an error path (a failed launch after earlier ones
// were enqueued, a failed round of the split form with the helper stream still busy) the buffers would return to the pool
// while kernels still read or write them.  So the first release under the flag waits for the device.
static thread_local bool t_inflight = false;
struct InFlight {
  InFlight() { t_inflight = true; }
  static void done() { t_inflight = false; }  // (the caller has synchronised)
};

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int device = -1;  // the device the memory lives on
  int alloc(size_t n) {
    free();
    if (n == 0) n = 16;
    PhaseClock pc(PH_ALLOC);
    size_t got = 0;
    device = current_device();
    if (void *q = pool_take(n, &got)) {
      p = q;
      bytes = got;
      return NHIP_OK;
    }
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) {  // (what the pool holds may be what is missing: let go of it and ask once more)
      (void)hipGetLastError();
      std::vector<void *> drop;
      pool_drain(&drop, current_device());
      for (void *d : drop) (void)hipFree(d);
      e = drop.empty() ? e : hipMalloc(&p, n);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
      set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
      return NHIP_ERR_ALLOC;
    }
    bytes = n;
    return NHIP_OK;
  }
  void free(uint64_t key = 0, uint64_t meta = 0) {  // (key, meta: the contents stay known to the pool, see PoolEntry)
    if (p) {
      PhaseClock pc(PH_FREE);
      if (t_inflight) {  // an error path: nothing that was enqueued may outlive its buffers' release
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
        t_inflight = false;
      }
      std::vector<void *> evict;
      if (!pool_put(p, bytes, device, &evict, key, meta)) (void)hipFree(p);
      for (void *d : evict) (void)hipFree(d);
    }
    p = nullptr;
    bytes = 0;
  }
  void adopt(void *q, size_t n) {  // (a buffer pool_take_pair returned: filed under the current device)
    free();
    p = q;
    bytes = n;
    device = current_device();
  }
  ~DevBuf() { free(); }
};

}  // namespace nhip

struct nhip_scans {
  nhip::DevBuf xy, offsets;
  int32_t n_scans = 0;
  int64_t n_points = 0;
  std::vector<int32_t> h_offsets;
};

struct nhip_grids {
  nhip::DevBuf grids;
  nhip::DevBuf ws;         // the build's workspace: its tile list and line masks describe what `grids` holds
  uint64_t shape = 0;      // digest of (spec as built, targets): what a later build must equal to rebuild into these buffers
  bool dirty = false;      // something besides the build wrote into `grids` (a late skip-map build): contents no longer the list's
  bool rebuilt = false;    // this handle's build was an incremental rebuild into a kept pair
  nhip_grid_spec_t spec;
  nhip::GridLayout L;
  int32_t n = 0;
  std::mutex mu;  // (the late skip-map build)
};

struct nhip_resid_batch {
  nhip::DevBuf corr, corr_block, block_src, block_tgt, consts, poses, res, jsrc, jtgt, jtt;  // (jtt also holds q: nhip_resid_batch_eval_q)
  nhip::DevBuf one_poses, one_consts, one_idx;  // single-block evaluation: 2 poses, 8 constants, {0, 1}
  std::vector<int32_t> h_offsets;
  std::mutex one_mu;
  int kind = 0;
  int32_t n_blocks = 0, n_poses = 0;
  int64_t n_corr = 0;
};

using namespace nhip;

// digest of what a table buffer's layout and build depend on: the spec as handed in, the number of targets
static uint64_t grids_shape(const nhip_grid_spec_t *spec, int32_t n_targets) {
  uint64_t h = 0xcbf29ce484222325ull;
  const unsigned char *b = reinterpret_cast<const unsigned char *>(spec);
  for (size_t i = 0; i < sizeof(*spec); i++) h = (h ^ b[i]) * 0x100000001b3ull;
  for (int i = 0; i < 4; i++) h = (h ^ (uint64_t)((uint32_t)n_targets >> (8 * i) & 0xffu)) * 0x100000001b3ull;
  return h >> 1;  // (the low bit is the pool's: which buffer of the pair)
}

extern "C" {

const char *nhip_last_error(void) { return g_err.c_str(); }
const char *nhip_version(void) { return "nautilus_hip 0.1 (gfx950)"; }

int nhip_init(int *n_devices) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  if (n_devices) *n_devices = n;
  if (n <= 0) {
    set_error("no HIP device visible");
    return NHIP_ERR_NODEV;
  }
  int cur = 0;
  if (hipGetDevice(&cur) == hipSuccess) {
    (void)dev_status_prepare(cur);  // (the other devices' words: nhip_set_device, before that device's first launch)
  } else {
    (void)hipGetLastError();
  }
  return NHIP_OK;
}

int nhip_set_device(int device) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_TRY_HIP(hipSetDevice(device));
  (void)dev_status_prepare(device);
  return NHIP_OK;
}

int nhip_grid_layout(const nhip_grid_spec_t *spec, nhip_grid_layout_t *out) {
  GridLayout L;
  int rc = make_layout(spec, &L);
  if (rc) return rc;
  NHIP_REQUIRE(out != nullptr, "grid_layout: null out");
  out->side = L.S;
  out->pad = L.pad;
  out->pitch = L.pitch;
  out->rows = L.S + 2 * L.pad;
  out->blur_radius = L.R;
  out->cell_bytes = L.cb;
  out->tap_sum = L.K;
  out->grid_bytes = L.grid_bytes;
  out->score_floor = L.Lf;
  out->score_step = L.step;
  out->skip_bytes = L.skip_bytes;
  out->slot_bytes = L.slot_bytes;
  out->pool_bytes = L.pool_bytes;
  out->pool_pitch = L.pool_pitch;
  out->pool_rows = L.pool_rows;
  out->pool4_bytes = L.pool4_bytes;
  out->pool4_pitch = L.pool4_pitch;
  out->pool4_rows = L.pool4_rows;
  out->hi_bytes = L.hi_bytes;
  out->hi_pitch = L.hi_pitch;
  out->hits_pitch = L.hits_pitch;
  out->hits_bytes = L.hits_bytes;
  return NHIP_OK;
}

int64_t nhip_grids_bytes(const nhip_grid_spec_t *spec, int64_t n_grids) {
  GridLayout L;
  if (make_layout(spec, &L)) return -1;
  return n_grids * L.slot_bytes + 256;
}

int64_t nhip_grid_workspace_bytes(const nhip_grid_spec_t *spec, int32_t chunk) {
  GridLayout L;
  if (make_layout(spec, &L)) return -1;
  if (chunk < 1) chunk = 1;
  // header | [16-bit threshold table] | tile occupancy | tile list
  return GRID_WS_HEADER + (L.cb == 2 ? GRID_WS_THR16 : 0) + 4 + (int64_t)chunk * grid_ws_per_target(L.S);
}

int nhip_grid_tables(const nhip_grid_spec_t *spec, int32_t *taps, uint32_t *thresholds) {
  GridLayout L;
  int rc = make_layout(spec, &L);
  if (rc) return rc;
  GridTables T;
  rc = make_tables(spec, L, &T);
  if (rc) return rc;
  if (taps) memcpy(taps, T.taps, sizeof(int32_t) * (2 * L.R + 1));
  if (thresholds && L.cb == 1) memcpy(thresholds, T.thr, sizeof(T.thr));
  if (thresholds && L.cb == 2) memcpy(thresholds, T.thr16, sizeof(uint32_t) * 65536);
  return NHIP_OK;
}

int nhip_csm_rot0(const double *rot_a, const double *rot_b, int32_t n, double *cs_out) {
  NHIP_REQUIRE(rot_a && cs_out && n >= 0, "csm_rot0: bad arguments");
  for (int32_t i = 0; i < n; i++) {
    // math_util.h:81-89: AngleDiff(a0, a1) = AngleMod(a0 - a1), AngleMod: a -= 2pi*rint(a / 2pi)
    double a = rot_a[i] - (rot_b ? rot_b[i] : 0.0);
    a -= (2.0 * M_PI) * rint(a / (2.0 * M_PI));
    cs_out[2 * i] = cos(a);
    cs_out[2 * i + 1] = sin(a);
  }
  return NHIP_OK;
}

int nhip_csm_delta_table(const nhip_search_t *search, double *cs_out) {
  NHIP_REQUIRE(search && cs_out && search->n_theta >= 1, "csm_delta_table: bad arguments");
  for (int32_t k = 0; k < search->n_theta; k++) {
    const double d = (double)(k - (search->n_theta - 1) / 2) * search->theta_step;
    cs_out[2 * k] = cos(d);
    cs_out[2 * k + 1] = sin(d);
  }
  return NHIP_OK;
}

int nhip_match_to_transform(const nhip_match_t *m, const nhip_grid_spec_t *spec,
                            const nhip_search_t *search, double theta0, int32_t origin_x,
                            int32_t origin_y, float *tx, float *ty, float *theta) {
  NHIP_REQUIRE(m && spec && search, "match_to_transform: null argument");
  if (tx) *tx = (float)((double)(origin_x + m->ix - (search->nx - 1) / 2) * spec->res);
  if (ty) *ty = (float)((double)(origin_y + m->iy - (search->ny - 1) / 2) * spec->res);
  if (theta) *theta = (float)(theta0 + (double)(m->itheta - (search->n_theta - 1) / 2) * search->theta_step);
  return NHIP_OK;
}

double nhip_score_from_sum(const nhip_grid_spec_t *spec, int64_t sum, int32_t n_points) {
  const double Lf = log(spec->floor_p), step = -Lf / (spec->cell_bits == 8 ? 255.0 : 65535.0);
  if (n_points <= 0) return Lf;
  const double t = step * (double)sum;
  const double u = t / (double)n_points;
  return Lf + u;
}

// ---------------------------------------------------------------- device-pointer API
int nhip_grid_build_dev(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, const int32_t *d_target_ids,
                        int32_t n_targets, const nhip_grid_spec_t *spec, uint8_t *d_grids,
                        void *d_workspace, int64_t workspace_bytes, void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(d_xy && d_offsets && d_target_ids && d_grids && d_workspace, "grid_build_dev: null pointer");
  NHIP_REQUIRE(n_targets >= 0 && n_scans >= 0, "grid_build_dev: n_targets %d / n_scans %d < 0", n_targets, n_scans);
  GridLayout L;
  rc = make_layout(spec, &L);
  if (rc) return rc;
  if (n_targets == 0) return NHIP_OK;
  return launch_grid_build(d_xy, d_offsets, n_scans, d_target_ids, n_targets, spec, L, d_grids, d_workspace,
                           workspace_bytes, static_cast<hipStream_t>(stream));
}

int nhip_grid_rebuild_dev(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, const int32_t *d_target_ids,
                          int32_t n_targets, const nhip_grid_spec_t *spec, uint8_t *d_grids,
                          void *d_workspace, int64_t workspace_bytes, void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(d_xy && d_offsets && d_target_ids && d_grids && d_workspace, "grid_rebuild_dev: null pointer");
  NHIP_REQUIRE(n_targets >= 0 && n_scans >= 0, "grid_rebuild_dev: n_targets %d / n_scans %d < 0", n_targets, n_scans);
  GridLayout L;
  rc = make_layout(spec, &L);
  if (rc) return rc;
  if (n_targets == 0) return NHIP_OK;
  return launch_grid_build(d_xy, d_offsets, n_scans, d_target_ids, n_targets, spec, L, d_grids, d_workspace,
                           workspace_bytes, static_cast<hipStream_t>(stream), true);
}

int nhip_csm_match_dev(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, const uint8_t *d_grids,
                       int32_t n_grids, const nhip_grid_spec_t *spec, const int32_t *d_pair_src,
                       const int32_t *d_pair_slot, const double *d_rot0_cs,
                       const double *d_delta_cs, const int32_t *d_pair_origin, int32_t n_pairs,
                       const nhip_search_t *search, uint64_t *d_keys, nhip_match_t *d_out,
                       int32_t *d_sums, void *d_workspace, int64_t workspace_bytes, void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(d_xy && d_offsets && d_grids && d_pair_src && d_pair_slot && d_rot0_cs && d_delta_cs &&
                   d_keys && d_out && search,
               "csm_match_dev: null pointer");
  NHIP_REQUIRE(workspace_bytes >= 0 && (d_workspace || workspace_bytes == 0), "csm_match_dev: bad workspace");
  NHIP_REQUIRE(n_pairs >= 0 && n_scans >= 0 && n_grids >= 0, "csm_match_dev: negative count (n_pairs %d, n_scans %d, n_grids %d)",
               n_pairs, n_scans, n_grids);
  GridLayout L;
  rc = make_layout(spec, &L);
  if (rc) return rc;
  const IdBounds ids = {n_scans, n_grids, dev_status()};
  nhip_search_t pub = *search;
  pub.flags &= ~(SEARCH_I_KEYS_ZERO | SEARCH_I_NO_FINALIZE);  // (the library's own bits: never a caller's)
  return launch_csm_match(d_xy, d_offsets, ids, d_grids, spec, L, d_pair_src, d_pair_slot, d_rot0_cs,
                          d_delta_cs, d_pair_origin, n_pairs, &pub, d_keys, d_out, d_sums,
                          static_cast<hipStream_t>(stream), d_workspace, workspace_bytes);
}

int64_t nhip_csm_workspace_bytes(int32_t n_pairs) { return bnb_workspace_bytes(n_pairs); }

int nhip_dev_status(void *stream, int32_t info[4]) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_TRY_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  uint32_t w[DEV_STATUS_WORDS] = {0u, 0u, 0u, 0u};
  uint32_t *st = dev_status();
  if (st) NHIP_TRY_HIP(hipMemcpy(w, st, sizeof(w), hipMemcpyDeviceToHost));
  if (info)
    for (int i = 0; i < 4; i++) info[i] = (int32_t)w[i];
  if (w[0] == 0u) return NHIP_OK;
  // cleared in the order of the stream that was asked about (a null-stream memset would race with kernels that are
  // flagging ids on other streams); the words are ONE set per device: see the header on what that means for several clients
  NHIP_TRY_HIP(hipMemsetAsync(st, 0, sizeof(w), static_cast<hipStream_t>(stream)));
  NHIP_TRY_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  const char *what = w[1] == BAD_TARGET_ID  ? "target scan id (nhip_grid_build_dev / nhip_grid_rebuild_dev: d_target_ids)"
                     : w[1] == BAD_PAIR_SRC  ? "source scan id (nhip_csm_match_dev: d_pair_src)"
                     : w[1] == BAD_PAIR_SLOT ? "grid slot (nhip_csm_match_dev: d_pair_slot)"
                     : w[1] == BAD_BLOCK_ID  ? "block id (d_corr_block)"
                     : w[1] == BAD_POSE_ID   ? "pose index (d_block_src / d_block_tgt / pose arrays)"
                     : w[1] == BAD_SCAN_ID   ? "scan id (d_block_src / d_block_tgt of the correspondence search)"
                                             : "id";
  set_error("an id read from device memory was out of range: %s = %d at index %d (kinds seen since the last check: 0x%x); "
            "the kernels treated every such entry as empty", what, (int32_t)w[2], (int32_t)w[3], w[0]);
  return NHIP_ERR_ARG;
}

int nhip_device_pool_configure(int64_t max_bytes) {
  NHIP_REQUIRE(max_bytes >= 0, "device_pool_configure: negative size");
  std::vector<void *> drop;
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    g_pool_cap = max_bytes;
    std::map<int, int64_t> tot;
    for (auto &e : g_pool) tot[e.device] += (int64_t)e.bytes;
    for (size_t i = 0; i < g_pool.size();) {  // (oldest first, each device against the cap on its own)
      if (tot[g_pool[i].device] > g_pool_cap) {
        tot[g_pool[i].device] -= (int64_t)g_pool[i].bytes;
        drop.push_back(g_pool[i].p);
        g_pool.erase(g_pool.begin() + (long)i);
      } else {
        i++;
      }
    }
  }
  for (void *d : drop) (void)hipFree(d);
  return NHIP_OK;
}

int nhip_device_pool_release(void) {
  std::vector<void *> drop;
  pool_drain(&drop, -1);
  for (void *d : drop) (void)hipFree(d);
  return NHIP_OK;
}

int nhip_device_pool_stats(int64_t *entries, int64_t *bytes) {
  std::lock_guard<std::mutex> lock(g_pool_mu);
  int64_t tot = 0;
  for (auto &e : g_pool) tot += (int64_t)e.bytes;
  if (entries) *entries = (int64_t)g_pool.size();
  if (bytes) *bytes = tot;
  return NHIP_OK;
}

int nhip_host_phases(double out[8]) {
  NHIP_REQUIRE(out != nullptr, "host_phases: null out");
  for (int i = 0; i < 8; i++) out[i] = t_phase[i];
  return NHIP_OK;
}

int nhip_csm_last_launch(int32_t out[8]) {
  NHIP_REQUIRE(out != nullptr, "csm_last_launch: null out");
  bnb_last_launch(out);
  return NHIP_OK;
}

int nhip_csm_scores_dev(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, const uint8_t *d_grids,
                        int32_t n_grids, const nhip_grid_spec_t *spec, int32_t src, int32_t slot,
                        const double *d_rot0_cs, const double *d_delta_cs, int32_t origin_x,
                        int32_t origin_y, const nhip_search_t *search, int32_t *d_sums,
                        void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(d_xy && d_offsets && d_grids && d_rot0_cs && d_delta_cs && d_sums && search,
               "csm_scores_dev: null pointer");
  NHIP_REQUIRE(src >= 0 && src < n_scans && slot >= 0 && slot < n_grids, "csm_scores_dev: scan %d of %d / grid slot %d of %d out of range",
               src, n_scans, slot, n_grids);
  GridLayout L;
  rc = make_layout(spec, &L);
  if (rc) return rc;
  return launch_csm_scores(d_xy, d_offsets, d_grids, spec, L, src, slot, d_rot0_cs, d_delta_cs,
                           origin_x, origin_y, search, d_sums, static_cast<hipStream_t>(stream));
}

int nhip_resid_lidar_dev(int kind, const float *d_corr, const int32_t *d_corr_block,
                         int64_t n_corr, const int32_t *d_block_src, const int32_t *d_block_tgt,
                         int32_t n_blocks, const double *d_poses, int32_t n_poses,
                         double *d_block_consts, double *d_residuals, double *d_jac_src,
                         double *d_jac_tgt, void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(d_corr && d_corr_block && d_block_src && d_block_tgt && d_poses && d_block_consts &&
                   d_residuals,
               "resid_lidar_dev: null pointer");
  return launch_resid_lidar(kind, d_corr, d_corr_block, n_corr, d_block_src, d_block_tgt, n_blocks,
                            d_poses, n_poses, d_block_consts, d_residuals, d_jac_src, d_jac_tgt,
                            static_cast<hipStream_t>(stream));
}

int nhip_resid_lidar_normal_eq_dev(int kind, const float *d_corr, const int32_t *d_block_offsets,
                                   const int32_t *d_block_src, const int32_t *d_block_tgt,
                                   int32_t n_blocks, const double *d_poses, int32_t n_poses,
                                   double *d_block_consts, double *d_out, void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(d_corr && d_block_offsets && d_block_src && d_block_tgt && d_poses && d_block_consts && d_out,
               "resid_lidar_normal_eq_dev: null pointer");
  NHIP_REQUIRE(n_poses >= 0, "resid_lidar_normal_eq_dev: negative size");
  return launch_resid_normal_eq(kind, d_corr, d_block_offsets, d_block_src, d_block_tgt, n_blocks, d_poses, n_poses,
                                d_block_consts, d_out, static_cast<hipStream_t>(stream));
}

int nhip_pose_affines(const double *poses, int32_t n, float *out) {
  NHIP_REQUIRE(poses && out && n >= 0, "pose_affines: bad arguments");
  for (int32_t i = 0; i < n; i++) {
    // entries of PoseArrayToAffine<double>(pose).cast<float>() (slam_util.h:20-28, 37-40)
    out[4 * i + 0] = (float)cos(poses[3 * i + 2]);
    out[4 * i + 1] = (float)sin(poses[3 * i + 2]);
    out[4 * i + 2] = (float)poses[3 * i + 0];
    out[4 * i + 3] = (float)poses[3 * i + 1];
  }
  return NHIP_OK;
}

int nhip_corr_search_dev(const float *d_xy, const float *d_normals, const int32_t *d_offsets, int32_t n_scans,
                         const int32_t *d_block_src, const int32_t *d_block_tgt, int32_t n_blocks,
                         const float *d_pose_aff, float outlier_threshold,
                         const int64_t *d_cap_offsets, float *d_corr_padded, int32_t *d_counts,
                         void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(d_xy && d_normals && d_offsets && d_block_src && d_block_tgt && d_pose_aff && d_cap_offsets &&
                   d_corr_padded && d_counts,
               "corr_search_dev: null pointer");
  NHIP_REQUIRE(n_blocks >= 0 && n_scans >= 0 && outlier_threshold > 0, "corr_search_dev: bad size or threshold");
  return launch_corr_search(d_xy, d_normals, d_offsets, n_scans, d_block_src, d_block_tgt, n_blocks, d_pose_aff,
                            outlier_threshold, 0.f, false, d_cap_offsets, d_corr_padded, d_counts,
                            static_cast<hipStream_t>(stream));
}

int nhip_corr_search_normals_dev(const float *d_xy, const float *d_normals, const int32_t *d_offsets, int32_t n_scans,
                                 const int32_t *d_block_src, const int32_t *d_block_tgt, int32_t n_blocks,
                                 const float *d_pose_aff, float outlier_threshold, float min_abs_cosine,
                                 const int64_t *d_cap_offsets, float *d_corr_padded, int32_t *d_counts,
                                 void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(d_xy && d_normals && d_offsets && d_block_src && d_block_tgt && d_pose_aff && d_cap_offsets &&
                   d_corr_padded && d_counts,
               "corr_search_normals_dev: null pointer");
  NHIP_REQUIRE(n_blocks >= 0 && n_scans >= 0 && outlier_threshold > 0, "corr_search_normals_dev: bad size or threshold");
  NHIP_REQUIRE(min_abs_cosine >= 0.f && min_abs_cosine <= 1.f, "corr_search_normals_dev: min_abs_cosine outside [0, 1]");
  return launch_corr_search(d_xy, d_normals, d_offsets, n_scans, d_block_src, d_block_tgt, n_blocks, d_pose_aff,
                            outlier_threshold, min_abs_cosine, true, d_cap_offsets, d_corr_padded, d_counts,
                            static_cast<hipStream_t>(stream));
}

int nhip_corr_compact_dev(const float *d_corr_padded, const int64_t *d_cap_offsets,
                          const int32_t *d_counts, int32_t n_blocks, int32_t *d_block_offsets,
                          float *d_corr, int32_t *d_corr_block, void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(d_corr_padded && d_cap_offsets && d_counts && d_block_offsets && d_corr && d_corr_block,
               "corr_compact_dev: null pointer");
  NHIP_REQUIRE(n_blocks >= 0, "corr_compact_dev: n_blocks < 0");
  return launch_corr_compact(d_corr_padded, d_cap_offsets, d_counts, n_blocks, d_block_offsets, d_corr,
                             d_corr_block, static_cast<hipStream_t>(stream));
}

int nhip_resid_point_to_line_dev(const float *d_segments, const float *d_points,
                                 const int32_t *d_point_block, int64_t n_points,
                                 const int32_t *d_block_pose, const int32_t *d_block_line,
                                 int32_t n_blocks, const double *d_poses, int32_t n_poses,
                                 const double *d_line_poses, int32_t n_line_poses, double *d_residuals,
                                 double *d_jac_pose, double *d_jac_line, void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(d_segments && d_points && d_point_block && d_block_pose && d_block_line && d_poses &&
                   d_line_poses && d_residuals,
               "resid_point_to_line_dev: null pointer");
  return launch_resid_point_to_line(d_segments, d_points, d_point_block, n_points, d_block_pose,
                                    d_block_line, n_blocks, d_poses, n_poses, d_line_poses, n_line_poses, d_residuals,
                                    d_jac_pose, d_jac_line, static_cast<hipStream_t>(stream));
}

int nhip_resid_odometry_dev(const float *d_t_odom, const float *d_r_odom, const int32_t *d_pose_i,
                            const int32_t *d_pose_j, int32_t n_factors, double translation_weight,
                            double rotation_weight, const double *d_poses, int32_t n_poses, double *d_residuals,
                            double *d_jac_i, double *d_jac_j, void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(d_t_odom && d_r_odom && d_pose_i && d_pose_j && d_poses && d_residuals,
               "resid_odometry_dev: null pointer");
  return launch_resid_odometry(d_t_odom, d_r_odom, d_pose_i, d_pose_j, n_factors,
                               translation_weight, rotation_weight, d_poses, n_poses, d_residuals, d_jac_i,
                               d_jac_j, static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------- handle API
int nhip_scans_upload(const float *xy, const int32_t *offsets, int32_t n_scans, nhip_scans_t **out) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(offsets && out && n_scans >= 0, "scans_upload: bad arguments");
  phases_reset();
  NHIP_REQUIRE(offsets[0] == 0, "scans_upload: offsets[0] must be 0");
  for (int32_t i = 0; i < n_scans; i++)
    NHIP_REQUIRE(offsets[i + 1] >= offsets[i], "scans_upload: offsets not monotone at scan %d", i);
  const int64_t n_points = offsets[n_scans];
  NHIP_REQUIRE(n_points == 0 || xy, "scans_upload: null xy");
  nhip_scans *s = new nhip_scans();
  s->n_scans = n_scans;
  s->n_points = n_points;
  s->h_offsets.assign(offsets, offsets + n_scans + 1);
  if ((rc = s->xy.alloc(sizeof(float) * 2 * (size_t)n_points)) ||
      (rc = s->offsets.alloc(sizeof(int32_t) * (size_t)(n_scans + 1)))) {
    delete s;
    return rc;
  }
  hipError_t e = hipSuccess;
  {
    PhaseClock pc(PH_UPLOAD);
    if (n_points) e = hipMemcpy(s->xy.p, xy, sizeof(float) * 2 * (size_t)n_points, hipMemcpyHostToDevice);
    if (e == hipSuccess)
      e = hipMemcpy(s->offsets.p, offsets, sizeof(int32_t) * (size_t)(n_scans + 1), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    delete s;
    return hip_fail(e, "scans_upload memcpy", __FILE__, __LINE__);
  }
  *out = s;
  return NHIP_OK;
}

int nhip_scans_free(nhip_scans_t *scans) {
  phases_reset();
  delete scans;
  return NHIP_OK;
}

int nhip_grids_build(const nhip_scans_t *scans, const int32_t *target_ids, int32_t n_targets,
                     const nhip_grid_spec_t *spec, nhip_grids_t **out) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(scans && out && n_targets >= 0 && (target_ids || n_targets == 0), "grids_build: bad arguments");
  phases_reset();
  for (int32_t i = 0; i < n_targets; i++)
    NHIP_REQUIRE(target_ids[i] >= 0 && target_ids[i] < scans->n_scans,
                 "grids_build: target id %d out of range", target_ids[i]);
  GridLayout L;
  rc = make_layout(spec, &L);
  if (rc) return rc;
  nhip_grids *g = new nhip_grids();
  g->spec = *spec;
  g->L = L;
  g->n = n_targets;
  g->shape = grids_shape(spec, n_targets);
  DevBuf ids;
  DevBuf &ws = g->ws;
  const int32_t chunk = n_targets > 0 ? n_targets : 1;  // workspace is ~25 bytes per 64x64 tile: all targets in one pass
  const int64_t ws_bytes = nhip_grid_workspace_bytes(spec, chunk);
  const size_t grid_bytes = (size_t)n_targets * L.slot_bytes + 256;
  // the buffers of the last build of this shape, if a handle released them and nobody has touched them since: the build
  // clears what that build wrote (its tile list and line masks are in the workspace, vouched for by a tag the clearing
  // kernel checks on the device) instead of zero-filling the slots -- 1.3 ms per 1000 targets at 1200 x 1200
  {
    void *pg = nullptr, *pw = nullptr;
    size_t bg = 0, bw = 0;
    if (n_targets > 0 && pool_take_pair(g->shape << 1, (g->shape << 1) | 1u, grid_bytes, (size_t)ws_bytes, &pg, &bg, &pw, &bw)) {
      g->grids.adopt(pg, bg);
      ws.adopt(pw, bw);
      g->rebuilt = true;
    }
  }
  if ((!g->rebuilt && ((rc = g->grids.alloc(grid_bytes)) || (rc = ws.alloc((size_t)ws_bytes)))) ||
      (rc = ids.alloc(sizeof(int32_t) * (size_t)(n_targets > 0 ? n_targets : 1)))) {
    delete g;
    return rc;
  }
  hipError_t e = hipSuccess;
  {
    PhaseClock pc(PH_UPLOAD);
    // (a fresh build zero-fills its n_targets slots itself, on its stream: launch_grid_build; what it does not reach is the
    //  256-byte tail past the last slot, which loads of the last slot's planes may run into)
    if (!g->rebuilt) e = hipMemset(static_cast<uint8_t *>(g->grids.p) + (size_t)n_targets * L.slot_bytes, 0, 256);
    if (e == hipSuccess && n_targets)
      e = hipMemcpy(ids.p, target_ids, sizeof(int32_t) * (size_t)n_targets, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    delete g;
    return hip_fail(e, "grids_build setup", __FILE__, __LINE__);
  }
  if (n_targets) {
    InFlight inflight;
    {
      PhaseClock pc(PH_ENQUEUE);
      rc = launch_grid_build(static_cast<const float *>(scans->xy.p),
                             static_cast<const int32_t *>(scans->offsets.p), scans->n_scans,
                             static_cast<const int32_t *>(ids.p), n_targets, spec, L,
                             static_cast<uint8_t *>(g->grids.p), ws.p, ws_bytes, nullptr, g->rebuilt);
    }
    if (rc == NHIP_OK) {
      PhaseClock pc(PH_WAIT);
      e = hipDeviceSynchronize();
      if (e != hipSuccess) rc = hip_fail(e, "grids_build sync", __FILE__, __LINE__);
      InFlight::done();
    }
    if (rc) {
      g->dirty = true;  // (a failed build: contents unknown)
      delete g;
      return rc;
    }
  }
  *out = g;
  return NHIP_OK;
}

int nhip_grids_free(nhip_grids_t *grids) {
  phases_reset();
  if (grids && grids->n > 0 && !grids->dirty && grids->grids.p && grids->ws.p) {
    // the pair goes back with its contents known (see PoolEntry): the next build of this shape may rebuild into it
    const uint64_t key = g_pool_key.fetch_add(1);
    grids->grids.free(key, grids->shape << 1);
    grids->ws.free(key, (grids->shape << 1) | 1u);
  }
  delete grids;
  return NHIP_OK;
}

int nhip_grids_was_rebuilt(const nhip_grids_t *grids) { return grids && grids->rebuilt ? 1 : 0; }

int nhip_grids_download(const nhip_grids_t *grids, int32_t slot, uint8_t *out) {
  NHIP_REQUIRE(grids && out && slot >= 0 && slot < grids->n, "grids_download: bad arguments");
  NHIP_REQUIRE(grids->L.has_image, "grids_download: the grids were built with NHIP_GRID_NO_IMAGE (nhip_grids_download_tiled16 / "
               "_hi_plane return the matcher's copies of the cells)");
  NHIP_TRY_HIP(hipMemcpy(out, static_cast<const uint8_t *>(grids->grids.p) + (size_t)slot * grids->L.slot_bytes,
                         (size_t)grids->L.grid_bytes, hipMemcpyDeviceToHost));
  return NHIP_OK;
}

int nhip_grids_download_hi_plane_copy(const nhip_grids_t *grids, int32_t slot, int32_t copy, uint8_t *out) {
  NHIP_REQUIRE(grids && out && slot >= 0 && slot < grids->n && (copy == 0 || copy == 1), "grids_download_hi_plane: bad arguments");
  const GridLayout &L = grids->L;
  std::vector<uint8_t> raw((size_t)L.hi_bytes);
  NHIP_TRY_HIP(hipMemcpy(raw.data(), static_cast<const uint8_t *>(grids->grids.p) + (size_t)slot * L.slot_bytes + L.grid_bytes +
                                         L.skip_bytes + L.pool_bytes + L.pool4_bytes,
                         (size_t)L.hi_bytes, hipMemcpyDeviceToHost));
  const int32_t rows = L.S + 2 * L.pad;
  for (int32_t r = 0; r < rows; r++)
    for (int32_t c = 0; c < L.hi_pitch; c++)
      out[(size_t)r * L.hi_pitch + c] = raw[hi_tiled((uint32_t)r, (uint32_t)c, (uint32_t)copy, (uint32_t)L.hi_tpr, (uint32_t)L.hi_copy_bytes)];
  return NHIP_OK;
}

int nhip_grids_download_tiled16(const nhip_grids_t *grids, int32_t slot, uint8_t *out) {
  NHIP_REQUIRE(grids && out && slot >= 0 && slot < grids->n, "grids_download_tiled16: bad arguments");
  const GridLayout &L = grids->L;
  NHIP_REQUIRE(L.t16_bytes > 0, "grids_download_tiled16: 8-bit grids have no tiled 16-bit copy");
  std::vector<uint8_t> raw((size_t)L.t16_bytes);
  NHIP_TRY_HIP(hipMemcpy(raw.data(), static_cast<const uint8_t *>(grids->grids.p) + (size_t)slot * L.slot_bytes + L.grid_bytes +
                                         L.skip_bytes + L.pool_bytes + L.pool4_bytes + 2 * L.hi_copy_bytes,
                         (size_t)L.t16_bytes, hipMemcpyDeviceToHost));
  const int32_t rows = L.S + 2 * L.pad;
  memset(out, 0, (size_t)L.plain_bytes);
  for (int32_t r = 0; r < rows; r++)
    for (int32_t c = 0; c < rows; c++)  // (square image: `rows` cells per row; the plain pitch may end before hi_pitch cells)
      memcpy(out + (size_t)r * L.pitch + 2 * (size_t)c, raw.data() + t16_tiled((uint32_t)r, (uint32_t)c, (uint32_t)L.t16_tpr), 2);
  return NHIP_OK;
}

int nhip_grids_download_hi_plane(const nhip_grids_t *grids, int32_t slot, uint8_t *out) {
  return nhip_grids_download_hi_plane_copy(grids, slot, 0, out);
}

int nhip_grids_download_skip_map(const nhip_grids_t *grids, int32_t slot, uint8_t *out) {
  NHIP_REQUIRE(grids && out && slot >= 0 && slot < grids->n, "grids_download_skip_map: bad arguments");
  NHIP_REQUIRE(grids->L.has_image, "grids_download_skip_map: grids built with NHIP_GRID_NO_IMAGE carry no skip map");
  const GridLayout &L = grids->L;
  NHIP_TRY_HIP(hipMemcpy(out, static_cast<const uint8_t *>(grids->grids.p) + (size_t)slot * L.slot_bytes + L.grid_bytes,
                         (size_t)L.skip_bytes, hipMemcpyDeviceToHost));
  return NHIP_OK;
}

int nhip_grids_download_pool(const nhip_grids_t *grids, int32_t slot, uint8_t *out) {
  NHIP_REQUIRE(grids && out && slot >= 0 && slot < grids->n, "grids_download_pool: bad arguments");
  const GridLayout &L = grids->L;
  NHIP_TRY_HIP(hipMemcpy(out, static_cast<const uint8_t *>(grids->grids.p) + (size_t)slot * L.slot_bytes + L.grid_bytes + L.skip_bytes,
                         (size_t)L.pool_bytes, hipMemcpyDeviceToHost));
  return NHIP_OK;
}

int nhip_grids_download_hits(const nhip_grids_t *grids, int32_t slot, uint8_t *out) {
  NHIP_REQUIRE(grids && out && slot >= 0 && slot < grids->n, "grids_download_hits: bad arguments");
  const GridLayout &L = grids->L;
  NHIP_TRY_HIP(hipMemcpy(out, static_cast<const uint8_t *>(grids->grids.p) + (size_t)slot * L.slot_bytes + L.grid_bytes + L.skip_bytes +
                                  L.pool_bytes + L.pool4_bytes + L.hi_bytes,
                         (size_t)L.hits_bytes, hipMemcpyDeviceToHost));
  return NHIP_OK;
}

int nhip_grids_download_pool4(const nhip_grids_t *grids, int32_t slot, uint8_t *out) {
  NHIP_REQUIRE(grids && out && slot >= 0 && slot < grids->n, "grids_download_pool4: bad arguments");
  const GridLayout &L = grids->L;
  // (rows x pitch: pool4_bytes may hold padding behind the table)
  NHIP_TRY_HIP(hipMemcpy(out, static_cast<const uint8_t *>(grids->grids.p) + (size_t)slot * L.slot_bytes + L.grid_bytes + L.skip_bytes + L.pool_bytes,
                         (size_t)L.pool4_rows * (size_t)L.pool4_pitch, hipMemcpyDeviceToHost));
  return NHIP_OK;
}

// 16-bit grids are built without skip maps unless their spec asks (the branch-and-bound matcher never reads them).
// The first search on a handle that takes the kernel that performs every add builds them, once.
static int ensure_skip_maps(const nhip_grids_t *grids, const nhip_search_t *search, int32_t n_pairs = 0x7fffffff) {
  // (writes to the handle -- the maps, then the flag -- under the handle's mutex, which is taken before the flag is looked
  //  at: concurrent nhip_csm_match calls on one handle are ordered, the loser finds the maps built.  L and n never change.)
  nhip_grids *g = const_cast<nhip_grids *>(grids);
  if (g->L.cb != 2 || g->n == 0 || !g->L.has_image) return NHIP_OK;
  NHIP_REQUIRE(search->n_theta >= 1 && search->nx >= 1 && search->ny >= 1, "search: empty lattice");
  if (!csm_takes_exhaustive(g->L, search)) return NHIP_OK;
  // (the kernel whose lanes are poses reads no skip map)
  if (csm_small_plane_fits(search) || ((search->flags & NHIP_SEARCH_LATENCY) && csm_small_tiled_fits(search, n_pairs, nullptr, nullptr)))
    return NHIP_OK;
  std::lock_guard<std::mutex> lock(g->mu);
  if (g->spec.flags & NHIP_GRID_SKIP_MAP) return NHIP_OK;
  g->dirty = true;  // (maps the build's tile list does not know of)
  int rc = launch_skipmap_build(static_cast<uint8_t *>(g->grids.p), g->n, g->L, nullptr);
  if (rc) return rc;
  NHIP_TRY_HIP(hipStreamSynchronize(nullptr));
  g->spec.flags |= NHIP_GRID_SKIP_MAP;
  return NHIP_OK;
}

int nhip_csm_match(const nhip_scans_t *scans, const nhip_grids_t *grids, const int32_t *pair_src,
                   const int32_t *pair_slot, const double *theta0, const int32_t *pair_origin,
                   int32_t n_pairs, const nhip_search_t *search, nhip_match_t *out,
                   int32_t *out_sums) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(scans && grids && search && n_pairs >= 0, "csm_match: bad arguments");
  NHIP_REQUIRE(n_pairs == 0 || (pair_src && pair_slot && theta0 && out), "csm_match: null array");
  phases_reset();
  PhaseClock pc_host(PH_HOST);  // (the whole call; the phases below are inside it)
  for (int32_t i = 0; i < n_pairs; i++) {
    NHIP_REQUIRE(pair_src[i] >= 0 && pair_src[i] < scans->n_scans, "csm_match: pair %d source %d out of range", i, pair_src[i]);
    NHIP_REQUIRE(pair_slot[i] >= 0 && pair_slot[i] < grids->n, "csm_match: pair %d grid slot %d out of range", i, pair_slot[i]);
    if (pair_origin)
      NHIP_REQUIRE(abs(pair_origin[2 * i]) + (search->nx - 1) / 2 <= grids->spec.max_shift &&
                       abs(pair_origin[2 * i + 1]) + (search->ny - 1) / 2 <= grids->spec.max_shift,
                   "csm_match: pair %d search centre (%d, %d) exceeds the grids' max_shift %d", i,
                   pair_origin[2 * i], pair_origin[2 * i + 1], grids->spec.max_shift);
  }
  if (n_pairs == 0) return NHIP_OK;
  // sums are reported as int32: the longest scan whose largest possible sum fits
  const int64_t max_pts = 0x7fffffffll / (grids->L.cb == 2 ? 65535 : 255);
  for (int32_t i = 0; i < n_pairs; i++) {
    const int64_t n_i = (int64_t)scans->h_offsets[pair_src[i] + 1] - scans->h_offsets[pair_src[i]];
    NHIP_REQUIRE(n_i <= max_pts, "csm_match: pair %d: scan %d has %lld points; with %d-bit cells at most %lld fit the "
                 "int32 sums", i, pair_src[i], (long long)n_i, 8 * grids->L.cb, (long long)max_pts);
  }
  if ((rc = ensure_skip_maps(grids, search))) return rc;
  nhip_grid_spec_t spec_now;  // (the flags may be written by a concurrent call's ensure_skip_maps: read them under the same lock)
  {
    std::lock_guard<std::mutex> lock(const_cast<nhip_grids *>(grids)->mu);
    spec_now = grids->spec;
  }
  // the host knows the scan lengths: when every source fits the by-rotation form the general kernel is not launched
  nhip_search_t search_now = *search;
  search_now.flags &= ~(SEARCH_I_KEYS_ZERO | SEARCH_I_NO_FINALIZE);  // (the library's own bits: never a caller's)
  {
    bool all_short = true;
    for (int32_t i = 0; i < n_pairs && all_short; i++)
      all_short = scans->h_offsets[pair_src[i] + 1] - scans->h_offsets[pair_src[i]] <= NHIP_SHORT_SCAN_POINTS;
    if (all_short) search_now.flags |= NHIP_SEARCH_SHORT_SCANS;
  }
  search = &search_now;
  std::vector<double> rot0(2 * (size_t)n_pairs), delta(2 * (size_t)search->n_theta);
  if ((rc = nhip_csm_rot0(theta0, nullptr, n_pairs, rot0.data()))) return rc;
  if ((rc = nhip_csm_delta_table(search, delta.data()))) return rc;
  DevBuf d_src, d_slot, d_rot0, d_delta, d_keys, d_out, d_sums, d_org, d_ws;
  // The split form's workspace is 32 KB per pair (4.3 GB at 131,072 pairs, 8.6 GB beyond): on a GPU that cannot spare
  // it the list still matches -- with the hand-over lists alone, in the one-kernel form (same records).
  int64_t ws_bytes = bnb_workspace_bytes(n_pairs);
  if (d_ws.alloc((size_t)ws_bytes) != NHIP_OK) {
    (void)hipGetLastError();
    ws_bytes = bnb_workspace_bytes_lists(n_pairs);
    if ((rc = d_ws.alloc((size_t)ws_bytes))) return rc;
  }
  if (pair_origin) {
    if ((rc = d_org.alloc(sizeof(int32_t) * 2 * (size_t)n_pairs))) return rc;
    NHIP_TRY_HIP(hipMemcpy(d_org.p, pair_origin, sizeof(int32_t) * 2 * (size_t)n_pairs, hipMemcpyHostToDevice));
  }
  if ((rc = d_src.alloc(sizeof(int32_t) * (size_t)n_pairs)) || (rc = d_slot.alloc(sizeof(int32_t) * (size_t)n_pairs)) ||
      (rc = d_rot0.alloc(sizeof(double) * rot0.size())) || (rc = d_delta.alloc(sizeof(double) * delta.size())) ||
      (rc = d_keys.alloc(sizeof(uint64_t) * (size_t)n_pairs)) || (rc = d_out.alloc(sizeof(nhip_match_t) * (size_t)n_pairs)) ||
      (rc = d_sums.alloc(sizeof(int32_t) * (size_t)n_pairs)))
    return rc;
  {
    PhaseClock pc(PH_UPLOAD);
    NHIP_TRY_HIP(hipMemcpy(d_src.p, pair_src, sizeof(int32_t) * (size_t)n_pairs, hipMemcpyHostToDevice));
    NHIP_TRY_HIP(hipMemcpy(d_slot.p, pair_slot, sizeof(int32_t) * (size_t)n_pairs, hipMemcpyHostToDevice));
    NHIP_TRY_HIP(hipMemcpy(d_rot0.p, rot0.data(), sizeof(double) * rot0.size(), hipMemcpyHostToDevice));
    NHIP_TRY_HIP(hipMemcpy(d_delta.p, delta.data(), sizeof(double) * delta.size(), hipMemcpyHostToDevice));
  }
  const IdBounds idb = {scans->n_scans, grids->n, dev_status()};  // (checked on the host above; the kernels check again)
  PhaseClock pc_enq(PH_ENQUEUE);  // (to the end of the call minus the phases inside it; nhip_host_phases subtracts nothing:
                                  //  read it as "enqueue + wait + download + frees")
  InFlight inflight;  // (a failure from here on: the DevBufs above wait for the device before they return to the pool)
  rc = launch_csm_match(static_cast<const float *>(scans->xy.p), static_cast<const int32_t *>(scans->offsets.p), idb,
                        static_cast<const uint8_t *>(grids->grids.p), &spec_now, grids->L,
                        static_cast<const int32_t *>(d_src.p), static_cast<const int32_t *>(d_slot.p),
                        static_cast<const double *>(d_rot0.p), static_cast<const double *>(d_delta.p),
                        pair_origin ? static_cast<const int32_t *>(d_org.p) : nullptr, n_pairs, search, static_cast<uint64_t *>(d_keys.p), static_cast<nhip_match_t *>(d_out.p),
                        static_cast<int32_t *>(d_sums.p), nullptr, d_ws.p, ws_bytes);
  if (rc) return rc;
  {
    PhaseClock pc(PH_WAIT);  // (the kernels; the downloads below find them done)
    NHIP_TRY_HIP(hipStreamSynchronize(nullptr));
    InFlight::done();
  }
  {
    PhaseClock pc(PH_DOWNLOAD);
    NHIP_TRY_HIP(hipMemcpy(out, d_out.p, sizeof(nhip_match_t) * (size_t)n_pairs, hipMemcpyDeviceToHost));
    if (out_sums) NHIP_TRY_HIP(hipMemcpy(out_sums, d_sums.p, sizeof(int32_t) * (size_t)n_pairs, hipMemcpyDeviceToHost));
  }
  return NHIP_OK;
}

int nhip_csm_scores(const nhip_scans_t *scans, const nhip_grids_t *grids, int32_t src, int32_t slot,
                    double theta0, int32_t origin_x, int32_t origin_y, const nhip_search_t *search,
                    int32_t *out_sums) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(scans && grids && search && out_sums, "csm_scores: bad arguments");
  NHIP_REQUIRE(src >= 0 && src < scans->n_scans && slot >= 0 && slot < grids->n, "csm_scores: index out of range");
  {
    nhip_search_t ex = *search;
    ex.flags |= NHIP_SEARCH_EXHAUSTIVE;  // (the volume always comes from the kernel that performs every add)
    if ((rc = ensure_skip_maps(grids, &ex))) return rc;
  }
  double rot0[2];
  std::vector<double> delta(2 * (size_t)search->n_theta);
  if ((rc = nhip_csm_rot0(&theta0, nullptr, 1, rot0))) return rc;
  if ((rc = nhip_csm_delta_table(search, delta.data()))) return rc;
  const size_t vol = (size_t)search->n_theta * search->nx * search->ny;
  DevBuf d_rot0, d_delta, d_vol;
  if ((rc = d_rot0.alloc(sizeof(rot0))) || (rc = d_delta.alloc(sizeof(double) * delta.size())) ||
      (rc = d_vol.alloc(sizeof(int32_t) * vol)))
    return rc;
  NHIP_TRY_HIP(hipMemcpy(d_rot0.p, rot0, sizeof(rot0), hipMemcpyHostToDevice));
  NHIP_TRY_HIP(hipMemcpy(d_delta.p, delta.data(), sizeof(double) * delta.size(), hipMemcpyHostToDevice));
  InFlight inflight;
  rc = launch_csm_scores(static_cast<const float *>(scans->xy.p), static_cast<const int32_t *>(scans->offsets.p),
                         static_cast<const uint8_t *>(grids->grids.p), &grids->spec, grids->L, src, slot,
                         static_cast<const double *>(d_rot0.p), static_cast<const double *>(d_delta.p),
                         origin_x, origin_y, search, static_cast<int32_t *>(d_vol.p), nullptr);
  if (rc) return rc;
  NHIP_TRY_HIP(hipMemcpy(out_sums, d_vol.p, sizeof(int32_t) * vol, hipMemcpyDeviceToHost));  // (synchronises the null stream)
  InFlight::done();
  return NHIP_OK;
}

int nhip_lc_scatter_scores_dev(const float *d_xy, const int32_t *d_offsets, int32_t n_scans, double *d_scores,
                               void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(n_scans >= 0 && (n_scans == 0 || (d_xy && d_offsets && d_scores)), "lc_scatter_scores_dev: bad arguments");
  return launch_lc_scatter_scores(d_xy, d_offsets, n_scans, d_scores, static_cast<hipStream_t>(stream));
}

int nhip_lc_pair_gate_dev(const double *d_poses, int32_t n_poses, const int32_t *d_candidates, int32_t n_candidates,
                          double max_range, int32_t min_separation, uint8_t *d_flags, void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(n_candidates >= 0 && n_poses >= 0 && (n_candidates == 0 || (d_poses && d_candidates && d_flags)),
               "lc_pair_gate_dev: bad arguments");
  return launch_lc_pair_gate(d_poses, n_poses, d_candidates, n_candidates, max_range, min_separation, d_flags,
                             static_cast<hipStream_t>(stream));
}

int nhip_lc_chi_square_gate_dev(const double *d_poses, int32_t n_poses, const int32_t *d_pair_src, const int32_t *d_pair_tgt,
                                const float *d_cov, int32_t n_pairs, double max_score, double *d_scores,
                                uint8_t *d_flags, void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(n_pairs >= 0 && n_poses >= 0 && (n_pairs == 0 || (d_poses && d_pair_src && d_pair_tgt && d_cov && d_scores && d_flags)),
               "lc_chi_square_gate_dev: bad arguments");
  NHIP_REQUIRE((reinterpret_cast<uintptr_t>(d_cov) & 15) == 0, "lc_chi_square_gate_dev: d_cov must be 16-byte aligned");
  return launch_lc_chi_square(d_poses, n_poses, d_pair_src, d_pair_tgt, d_cov, n_pairs, max_score, d_scores, d_flags,
                              static_cast<hipStream_t>(stream));
}

int nhip_lc_chi_square_gate(const double *poses, int32_t n_poses, const int32_t *pair_src, const int32_t *pair_tgt,
                            const float *cov, int32_t n, double max_score, double *scores, uint8_t *flags) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(n >= 0 && n_poses >= 0 && (n == 0 || (poses && pair_src && pair_tgt && cov && scores && flags)),
               "lc_chi_square_gate: bad arguments");
  for (int32_t i = 0; i < n; i++)
    NHIP_REQUIRE(pair_src[i] >= 0 && pair_src[i] < n_poses && pair_tgt[i] >= 0 && pair_tgt[i] < n_poses,
                 "lc_chi_square_gate: pair %d (%d, %d) out of range", i, pair_src[i], pair_tgt[i]);
  if (n == 0) return NHIP_OK;
  const size_t N = (size_t)n;
  DevBuf dp, ds, dt, dc, dsc, df;
  if ((rc = dp.alloc(sizeof(double) * 3 * (size_t)n_poses)) || (rc = ds.alloc(4 * N)) || (rc = dt.alloc(4 * N)) ||
      (rc = dc.alloc(16 * N)) || (rc = dsc.alloc(8 * N)) || (rc = df.alloc(N)))
    return rc;
  NHIP_TRY_HIP(hipMemcpy(dp.p, poses, sizeof(double) * 3 * (size_t)n_poses, hipMemcpyHostToDevice));
  NHIP_TRY_HIP(hipMemcpy(ds.p, pair_src, 4 * N, hipMemcpyHostToDevice));
  NHIP_TRY_HIP(hipMemcpy(dt.p, pair_tgt, 4 * N, hipMemcpyHostToDevice));
  NHIP_TRY_HIP(hipMemcpy(dc.p, cov, 16 * N, hipMemcpyHostToDevice));
  InFlight inflight;
  rc = launch_lc_chi_square(static_cast<const double *>(dp.p), n_poses, static_cast<const int32_t *>(ds.p),
                            static_cast<const int32_t *>(dt.p), static_cast<const float *>(dc.p), n, max_score,
                            static_cast<double *>(dsc.p), static_cast<uint8_t *>(df.p), nullptr);
  if (rc) return rc;
  NHIP_TRY_HIP(hipMemcpy(scores, dsc.p, 8 * N, hipMemcpyDeviceToHost));
  NHIP_TRY_HIP(hipMemcpy(flags, df.p, N, hipMemcpyDeviceToHost));
  InFlight::done();  // (the downloads above synchronised the null stream)
  return NHIP_OK;
}

int nhip_lc_scatter_scores(const nhip_scans_t *scans, double *scores) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(scans && (scores || scans->n_scans == 0), "lc_scatter_scores: bad arguments");
  if (scans->n_scans == 0) return NHIP_OK;
  DevBuf d;
  if ((rc = d.alloc(sizeof(double) * (size_t)scans->n_scans))) return rc;
  InFlight inflight;
  rc = launch_lc_scatter_scores(static_cast<const float *>(scans->xy.p), static_cast<const int32_t *>(scans->offsets.p),
                                scans->n_scans, static_cast<double *>(d.p), nullptr);
  if (rc) return rc;
  NHIP_TRY_HIP(hipMemcpy(scores, d.p, sizeof(double) * (size_t)scans->n_scans, hipMemcpyDeviceToHost));
  InFlight::done();  // (the downloads above synchronised the null stream)
  return NHIP_OK;
}

int nhip_lc_pair_gate(const double *poses, int32_t n_poses, const int32_t *candidates, int32_t n, double max_range,
                      int32_t min_separation, uint8_t *flags) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(n >= 0 && n_poses >= 0 && (n == 0 || (poses && candidates && flags)), "lc_pair_gate: bad arguments");
  for (int32_t i = 0; i < n; i++)
    NHIP_REQUIRE(candidates[i] >= 0 && candidates[i] < n_poses, "lc_pair_gate: candidate %d out of range", candidates[i]);
  if (n == 0) return NHIP_OK;
  DevBuf dp, dc, df;
  if ((rc = dp.alloc(sizeof(double) * 3 * (size_t)n_poses)) || (rc = dc.alloc(sizeof(int32_t) * (size_t)n)) ||
      (rc = df.alloc((size_t)n * n)))
    return rc;
  NHIP_TRY_HIP(hipMemcpy(dp.p, poses, sizeof(double) * 3 * (size_t)n_poses, hipMemcpyHostToDevice));
  NHIP_TRY_HIP(hipMemcpy(dc.p, candidates, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
  InFlight inflight;
  rc = launch_lc_pair_gate(static_cast<const double *>(dp.p), n_poses, static_cast<const int32_t *>(dc.p), n, max_range,
                           min_separation, static_cast<uint8_t *>(df.p), nullptr);
  if (rc) return rc;
  NHIP_TRY_HIP(hipMemcpy(flags, df.p, (size_t)n * n, hipMemcpyDeviceToHost));
  InFlight::done();  // (the downloads above synchronised the null stream)
  return NHIP_OK;
}

// ---- the reference-shaped single-pair call, with the tables of the last targets kept
// Solver::SolveAutoLC -> GetRelativeTransform (solver.cc:630-649, 676-700) calls GetTransformation once per candidate pair,
// and a target scan is matched against many sources in a row.  A call used to zero and build 77 MB + 161 MB of tables
// for a target it had seen a call ago (1.4 ms, 708 calls/s); now the two grid handles of the last targets stay, keyed by
// the target cloud ITSELF (length + 64-bit hash to find it, the bytes compared to be sure) and the constructor's
// parameters, least recently used out first under a byte cap (nhip_csm_cache_configure / nhip_csm_cache_clear).
// The fine grid of a cached target is built for the largest reach any coarse optimum can ask for (its layout no longer
// depends on the source), which changes no result: the border is zeros either way.
namespace {

struct CachedTarget {
  std::vector<float> cloud;   // the target's points, for the exact comparison
  uint64_t hash = 0;
  nhip_csm_params_t params;
  int device = -1;
  nhip_grids_t *g1 = nullptr, *g2 = nullptr;
  nhip_grid_spec_t spec1, spec2;
  int64_t bytes = 0;
  ~CachedTarget() {
    if (g1) nhip_grids_free(g1);
    if (g2) nhip_grids_free(g2);
  }
};

std::mutex g_cache_mu;
// most recently used first.  (Heap-allocated and never destroyed: at process exit the HIP runtime may be gone before
// static destructors run, and freeing device memory then is not safe; nhip_csm_cache_clear() frees it while it is.)
std::vector<std::shared_ptr<CachedTarget>> &g_cache = *new std::vector<std::shared_ptr<CachedTarget>>();
int64_t g_cache_cap = 3ll << 30;
int64_t g_cache_hits = 0, g_cache_misses = 0;

uint64_t hash_bytes(const void *p, size_t n) {  // FNV-1a over 8-byte words (+ tail): finds the entry, never decides equality
  const uint8_t *b = static_cast<const uint8_t *>(p);
  uint64_t h = 1469598103934665603ull;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, b + i, 8);
    h = (h ^ w) * 1099511628211ull;
  }
  for (; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

bool same_params(const nhip_csm_params_t &a, const nhip_csm_params_t &b) {
  return a.scanner_range == b.scanner_range && a.trans_range == b.trans_range && a.low_res == b.low_res &&
         a.high_res == b.high_res && a.sigma == b.sigma && a.floor_p == b.floor_p &&
         (a.cell_bits == 0 ? 16 : a.cell_bits) == (b.cell_bits == 0 ? 16 : b.cell_bits);
}

// The calling thread's scratch for one pair (device buffers that live as long as the thread's library use: a call is
// two launches and four small copies, no allocation).
constexpr int DROPIN_PARTS_MAX = 8;  // "pairs" (workgroups) one search's rotations may be dealt over
struct DropInScratch {
  int device = -1;
  // par: one block of per-call parameters, uploaded in ONE copy {scan offsets int32[2] @0, (cos, sin) theta0 per part
  // double[16] @16, search centre per part int32[16] @144, rotation base per part int32[8] @208}; res: the records
  // nhip_match_t[8] @0 and their sums int32[8] @128, downloaded in one copy
  DevBuf xy, par, idx, keys, res, delta1, delta2, ws;
  // the chained form (both levels enqueued behind one another, ONE synchronisation per call): the fine level's parameter
  // block is written on the device by dropin_bridge_kernel from the coarse record and a table of the (cos, sin) every
  // coarse rotation would hand to the fine level (libm values, computed by the host per call); its keys, workspace and
  // records are its own, the host's blocks travel through pinned memory
  DevBuf par2, rot1, keys2, ws2;
  void *pin = nullptr;  // pinned host staging: upload block (256 B of parameters + the table) | download block (512 B)
  size_t xy_cap = 0;
  int32_t n_theta1 = -1, n_theta2 = -1;
  double step1 = 0, step2 = 0;
};
// (a few KB of device memory per calling thread, freed when the thread ends: thread-local destructors -- the main thread's
//  too -- run before the process's static destructors, i.e. while the HIP runtime is still there)
static thread_local double t_dropin_info[4] = {0, 0, 0, 0};
constexpr int DROPIN_CHAIN_ROT_MAX = 512;               // coarse rotations the chained form's table holds
constexpr size_t DROPIN_UP_BYTES = 256 + 16 * (size_t)DROPIN_CHAIN_ROT_MAX, DROPIN_DOWN_BYTES = 512;
struct ScratchHolder {
  DropInScratch *p = nullptr;
  ~ScratchHolder() {
    if (p && p->pin) (void)hipHostFree(p->pin);
    delete p;
  }
};
thread_local ScratchHolder t_scratch;

int scratch_for(int device, int32_t n_a, const nhip_search_t &s1, const nhip_search_t &s2, DropInScratch **out) {
  if (!t_scratch.p) t_scratch.p = new DropInScratch();
  DropInScratch &S = *t_scratch.p;
  int rc;
  // (a failure below leaves the scratch EMPTY -- device -1 -- so that the thread's next call sets it up again instead of
  //  finding the device it asked for and null buffers behind it)
  auto reset = [&S]() {
    if (S.pin) (void)hipHostFree(S.pin);
    S.~DropInScratch();
    new (&S) DropInScratch();
  };
  if (S.device != device) {
    reset();
    const int32_t zeros[2 * DROPIN_PARTS_MAX] = {0};
    constexpr size_t G = DROPIN_PARTS_MAX;
    if ((rc = S.par.alloc(256)) || (rc = S.idx.alloc(8 * G)) || (rc = S.keys.alloc(8 * G)) || (rc = S.res.alloc(DROPIN_DOWN_BYTES)) ||
        (rc = S.ws.alloc((size_t)bnb_workspace_bytes_lists((int32_t)G))) || (rc = S.par2.alloc(256)) ||
        (rc = S.rot1.alloc(DROPIN_UP_BYTES)) || (rc = S.keys2.alloc(8 * G)) ||
        (rc = S.ws2.alloc((size_t)bnb_workspace_bytes_lists((int32_t)G)))) {
      reset();
      return rc;
    }
    if (hipHostMalloc(&S.pin, DROPIN_UP_BYTES + DROPIN_DOWN_BYTES, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      S.pin = nullptr;  // (no pinned memory: the call falls back to the form with one synchronisation per level)
    }
    const hipError_t e = hipMemcpy(S.idx.p, zeros, 8 * G, hipMemcpyHostToDevice);  // source scan 0, grid slot 0 for every part
    if (e != hipSuccess) {
      reset();
      return hip_fail(e, "csm_get_transformation scratch", __FILE__, __LINE__);
    }
    S.device = device;
  }
  if ((size_t)n_a > S.xy_cap) {
    const size_t cap = std::max<size_t>((size_t)n_a, 2048);
    S.xy_cap = 0;  // (DevBuf::alloc frees first: a failed allocation leaves no buffer, and no capacity that says otherwise)
    if ((rc = S.xy.alloc(sizeof(float) * 2 * cap))) return rc;
    S.xy_cap = cap;
  }
  auto table = [&](DevBuf &d, int32_t &n_have, double &step_have, const nhip_search_t &s) -> int {
    if (n_have == s.n_theta && step_have == s.theta_step) return NHIP_OK;
    // (+ DROPIN_PARTS_MAX copies of the last rotation: a search dealt over several workgroups in equal parts reads past
    //  the table's end; a copy's poses tie with the original's and lose the tie by their larger index)
    std::vector<double> t(2 * (size_t)(s.n_theta + DROPIN_PARTS_MAX));
    int r = nhip_csm_delta_table(&s, t.data());
    if (r) return r;
    for (int e = 0; e < DROPIN_PARTS_MAX; e++) {
      t[2 * (size_t)(s.n_theta + e)] = t[2 * (size_t)(s.n_theta - 1)];
      t[2 * (size_t)(s.n_theta + e) + 1] = t[2 * (size_t)(s.n_theta - 1) + 1];
    }
    if ((r = d.alloc(sizeof(double) * t.size()))) return r;
    NHIP_TRY_HIP(hipMemcpy(d.p, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
    n_have = s.n_theta;
    step_have = s.theta_step;
    return NHIP_OK;
  };
  if ((rc = table(S.delta1, S.n_theta1, S.step1, s1)) || (rc = table(S.delta2, S.n_theta2, S.step2, s2))) return rc;
  *out = &S;
  return NHIP_OK;
}

// One pair (scan 0 of the scratch against slot 0 of `g`) on the null stream; the record comes back to the host.
// The branch-and-bound matcher computes a pair's bounds in the pair's ONE workgroup, eight rotations at a time: a search
// of 21 rotations is three rounds on one CU while 255 idle.  So the rotations are dealt over `parts` workgroups -- to the
// kernels they are `parts` pairs of the same scan and table whose rotation 0 is entry kbase of the rotation table
// (BnbParams::pair_kbase) -- and the host takes the best of their records: the larger sum, on a tie the smaller index
// ((k * nx + ix) * ny + iy with the part's rotations counted from the search's first), which is the one-workgroup result.
struct DropInPar {  // the per-call parameter block as the kernels read it (device copy: DropInScratch::par / par2)
  int32_t off[4];   // scan offsets {0, n_a}
  double cs[2 * DROPIN_PARTS_MAX];   // (cos, sin) theta0 per part
  int32_t org[2 * DROPIN_PARTS_MAX], kb[DROPIN_PARTS_MAX];  // search centre per part, rotation base per part
};
static_assert(sizeof(DropInPar) == 240 && offsetof(DropInPar, cs) == 16 && offsetof(DropInPar, org) == 144 &&
              offsetof(DropInPar, kb) == 208, "DropInPar layout");
struct DropInRes {
  nhip_match_t rec[DROPIN_PARTS_MAX];
  int32_t sums[DROPIN_PARTS_MAX];
};
static_assert(sizeof(DropInRes) == 160 && offsetof(DropInRes, sums) == 128, "DropInRes layout");

// parts: the fewest (2 .. 8) that give every workgroup an odd number (the lattice's rule) of at most 8 rotations
void dropin_parts(const nhip_grids_t *g, const nhip_search_t *search, int *parts, int *per) {
  *parts = 1;
  *per = search->n_theta;
  const char *one = tunable("NHIP_DROPIN_PARTS");  // (measurement: "1" keeps the search in one workgroup)
  const int q0 = one && atoi(one) >= 2 ? atoi(one) : 2;  // (measurement: at least that many parts)
  if (!csm_takes_exhaustive(g->L, search) && search->n_theta > 8 && !(one && one[0] == '1'))
    for (int q = q0; q <= DROPIN_PARTS_MAX; q++) {
      const int r = (search->n_theta + q - 1) / q;
      if ((r & 1) && r <= 8) {
        *parts = q;
        *per = r;
        break;
      }
    }
}

// enqueue one level: scan 0 of the scratch against slot 0 of `g`, parameters in the device block `d_par` (DropInPar),
// records into the device block `d_res` (DropInRes)
int dropin_enqueue(DropInScratch &S, int32_t n_a, nhip_grids_t *g, const nhip_grid_spec_t &spec_now, const nhip_search_t *search,
                   const void *d_delta, const void *d_par, bool with_origin, int parts, int per, void *d_keys, void *d_ws,
                   size_t ws_bytes, void *d_res) {
  const uint8_t *dp = static_cast<const uint8_t *>(d_par);
  uint8_t *dr = static_cast<uint8_t *>(d_res);
  nhip_search_t part = *search;
  part.n_theta = per;
  // (the host holds the cloud: a source that fits the matcher's by-rotation form saves the launch of the other instantiation)
  if (n_a <= NHIP_SHORT_SCAN_POINTS) part.flags |= NHIP_SEARCH_SHORT_SCANS;
  const IdBounds idb = {1, g->n, dev_status()};  // (the scratch holds one scan; every part reads scan 0, slot 0)
  return launch_csm_match(static_cast<const float *>(S.xy.p), reinterpret_cast<const int32_t *>(dp), idb,
                          static_cast<const uint8_t *>(g->grids.p), &spec_now, g->L, static_cast<const int32_t *>(S.idx.p),
                          static_cast<const int32_t *>(S.idx.p) + DROPIN_PARTS_MAX, reinterpret_cast<const double *>(dp + 16),
                          static_cast<const double *>(d_delta), with_origin ? reinterpret_cast<const int32_t *>(dp + 144) : nullptr,
                          parts, &part, static_cast<uint64_t *>(d_keys), reinterpret_cast<nhip_match_t *>(dr),
                          reinterpret_cast<int32_t *>(dr + 128), nullptr, d_ws, (int64_t)ws_bytes,
                          parts > 1 ? reinterpret_cast<const int32_t *>(dp + 208) : nullptr);
}

// the best of the parts' records
void dropin_pick(DropInRes &res, int parts, int per, const nhip_search_t *search, nhip_match_t *m) {
  int best = -1;
  int64_t best_lin = 0;
  for (int q = 0; q < parts; q++) {
    // (a copy of the last rotation past the table's end IS the last rotation)
    res.rec[q].itheta = std::min(res.rec[q].itheta + q * per, search->n_theta - 1);
    const int64_t lin = ((int64_t)res.rec[q].itheta * search->nx + res.rec[q].ix) * search->ny + res.rec[q].iy;
    if (best < 0 || res.sums[q] > res.sums[best] || (res.sums[q] == res.sums[best] && lin < best_lin)) {
      best = q;
      best_lin = lin;
    }
  }
  *m = res.rec[best];
}

int spec_under_lock(nhip_grids_t *g, nhip_grid_spec_t *out) {
  std::lock_guard<std::mutex> lock(g->mu);
  *out = g->spec;
  return NHIP_OK;
}

int match_one(DropInScratch &S, int32_t n_a, nhip_grids_t *g, const nhip_search_t *search, const void *d_delta, double theta0,
              const int32_t *origin, nhip_match_t *m) {
  int parts, per;
  dropin_parts(g, search, &parts, &per);
  int rc = ensure_skip_maps(g, search, parts);
  if (rc) return rc;
  nhip_grid_spec_t spec_now;
  spec_under_lock(g, &spec_now);
  DropInPar par;
  memset(&par, 0, sizeof(par));
  par.off[1] = n_a;
  if ((rc = nhip_csm_rot0(&theta0, nullptr, 1, par.cs))) return rc;
  for (int q = 0; q < parts; q++) {
    par.cs[2 * q] = par.cs[0];
    par.cs[2 * q + 1] = par.cs[1];
    par.org[2 * q] = origin ? origin[0] : 0;
    par.org[2 * q + 1] = origin ? origin[1] : 0;
    par.kb[q] = q * per;
  }
  NHIP_TRY_HIP(hipMemcpyAsync(S.par.p, &par, sizeof(par), hipMemcpyHostToDevice, nullptr));
  if ((rc = dropin_enqueue(S, n_a, g, spec_now, search, d_delta, S.par.p, origin != nullptr, parts, per, S.keys.p, S.ws.p, S.ws.bytes, S.res.p)))
    return rc;
  DropInRes res;
  NHIP_TRY_HIP(hipMemcpy(&res, S.res.p, sizeof(res), hipMemcpyDeviceToHost));
  dropin_pick(res, parts, per, search, m);
  return NHIP_OK;
}

// The fine level's parameter block from the coarse level's record, on the device (one thread): the same arithmetic as
// nhip_match_to_transform + the host code between the two levels -- products and quotients of doubles, individually
// rounded (the library is compiled with -ffp-contract=off), conversions to float by round-to-nearest, lround -- so the host,
// which repeats it on the downloaded coarse record to report the transform, arrives at the same origin.  The (cos, sin) of
// the fine search's centre angle are NOT computed here: device and host libm differ in the last place; the host tabulates
// its own values for every coarse rotation and the kernel picks the winner's.
__global__ void dropin_bridge_kernel(const nhip_match_t *rec1, const double *rot1, int32_t hx1, int32_t hy1, double low_res,
                                     double high_res, int32_t n_a, int32_t parts2, int32_t per2, DropInPar *par2, int32_t *info,
                                     const unsigned long long *keys1, int32_t nx1, int32_t ny1, double Lf1, double step1,
                                     nhip_match_t *rec1_out, int32_t *sums1_out, unsigned long long *keys2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  nhip_match_t m;
  if (keys1) {
    // the coarse search left its keys undecoded (SEARCH_I_NO_FINALIZE): csm_finalize_kernel's decoding, here, and the fine
    // search's keys zeroed for it (SEARCH_I_KEYS_ZERO) -- two small launches and a memset fewer per call
    const unsigned long long key = keys1[0];
    const uint32_t sum = (uint32_t)(key >> 32), lin = 0xffffffffu - (uint32_t)key;
    m.iy = (int32_t)(lin % (uint32_t)ny1);
    m.ix = (int32_t)((lin / (uint32_t)ny1) % (uint32_t)nx1);
    m.itheta = (int32_t)(lin / ((uint32_t)ny1 * (uint32_t)nx1));
    double sc = Lf1;
    if (n_a > 0) sc = __dadd_rn(Lf1, __ddiv_rn(__dmul_rn(step1, (double)sum), (double)n_a));
    m.score = __double2float_rn(sc);
    rec1_out[0] = m;
    sums1_out[0] = (int32_t)sum;
    for (int q = 0; q < DROPIN_PARTS_MAX; q++) keys2[q] = 0ull;
  } else {
    m = rec1[0];
  }
  const float tx1 = __double2float_rn(__dmul_rn((double)(m.ix - hx1), low_res));
  const float ty1 = __double2float_rn(__dmul_rn((double)(m.iy - hy1), low_res));
  const int32_t ox = (int32_t)lround(__ddiv_rn((double)tx1, high_res)), oy = (int32_t)lround(__ddiv_rn((double)ty1, high_res));
  par2->off[0] = 0;
  par2->off[1] = n_a;
  par2->off[2] = par2->off[3] = 0;
  const int32_t k = m.itheta < 0 ? 0 : (m.itheta >= DROPIN_CHAIN_ROT_MAX ? DROPIN_CHAIN_ROT_MAX - 1 : m.itheta);
  const double c = rot1[2 * k], s_ = rot1[2 * k + 1];
  for (int q = 0; q < DROPIN_PARTS_MAX; q++) {
    par2->cs[2 * q] = c;
    par2->cs[2 * q + 1] = s_;
    par2->org[2 * q] = ox;
    par2->org[2 * q + 1] = oy;
    par2->kb[q] = q < parts2 ? q * per2 : 0;
  }
  info[0] = ox;
  info[1] = oy;
}

// Both levels behind one another on the null stream, ONE synchronisation: upload (source cloud; coarse parameters + the
// table of fine-centre rotations, one pinned block), coarse search, bridge, fine search (+ exact score), download of both
// levels' records in one pinned block.  Same records as two match_one calls (tests/test_csm_gpu.py, test_adapters_gpu.py
// compare the call with the oracle's two-level search float for float).  Returns NHIP_ERR_STATE when the form does not
// apply (no pinned staging, more coarse rotations than the table holds): the caller takes the two-synchronisation form.
int match_chained(DropInScratch &S, const float *pc_a, int32_t n_a, CachedTarget &T, const nhip_search_t &s1, const nhip_search_t &s2,
                  const nhip_grid_spec_t &spec1, double theta0, nhip_match_t *m1, nhip_match_t *m2) {
  if (!S.pin || s1.n_theta > DROPIN_CHAIN_ROT_MAX) return NHIP_ERR_STATE;
  int rc;
  int parts1, per1, parts2, per2;
  dropin_parts(T.g1, &s1, &parts1, &per1);
  dropin_parts(T.g2, &s2, &parts2, &per2);
  if ((rc = ensure_skip_maps(T.g1, &s1, parts1)) || (rc = ensure_skip_maps(T.g2, &s2, parts2))) return rc;
  nhip_grid_spec_t spec1_now, spec2_now;
  spec_under_lock(T.g1, &spec1_now);
  spec_under_lock(T.g2, &spec2_now);
  if (parts1 != 1) return NHIP_ERR_STATE;  // (the bridge reads ONE coarse record; the coarse lattice is the every-add kernels')
  uint8_t *up = static_cast<uint8_t *>(S.pin), *down = up + DROPIN_UP_BYTES;
  DropInPar *par1 = reinterpret_cast<DropInPar *>(up);
  memset(par1, 0, sizeof(*par1));
  par1->off[1] = n_a;
  if ((rc = nhip_csm_rot0(&theta0, nullptr, 1, par1->cs))) return rc;
  // what the host would hand the fine level for each coarse rotation k: theta1 = (double)(float)(theta0 + (k - half) * step)
  double *rot = reinterpret_cast<double *>(up + 256);
  const int32_t half1 = (s1.n_theta - 1) / 2;
  for (int32_t k = 0; k < s1.n_theta; k++) {
    const double theta1 = (double)(float)(theta0 + (double)(k - half1) * s1.theta_step);
    if ((rc = nhip_csm_rot0(&theta1, nullptr, 1, rot + 2 * k))) return rc;
  }
  const size_t up_bytes = 256 + 16 * (size_t)s1.n_theta;
  if (n_a) NHIP_TRY_HIP(hipMemcpyAsync(S.xy.p, pc_a, sizeof(float) * 2 * (size_t)n_a, hipMemcpyHostToDevice, nullptr));
  NHIP_TRY_HIP(hipMemcpyAsync(S.rot1.p, up, up_bytes, hipMemcpyHostToDevice, nullptr));
  uint8_t *dres = static_cast<uint8_t *>(S.res.p);
  // both levels through the kernel whose lanes are poses (the default): the bridge decodes the coarse keys and zeroes the fine
  // ones, the exact-score pass decodes the fine keys -- no finalize launches, no second memset
  const bool fused = csm_takes_exhaustive(T.g1->L, &s1) && csm_small_plane_fits(&s1) && csm_takes_exhaustive(T.g2->L, &s2) &&
                     (csm_small_plane_fits(&s2) || ((s2.flags & NHIP_SEARCH_LATENCY) && csm_small_tiled_fits(&s2, parts2, nullptr, nullptr))) &&
                     (s2.flags & NHIP_SEARCH_EXACT_SCORE) && parts2 == 1 && !tunable("NHIP_CSM_SMALL") && !tunable("NHIP_DROPIN_UNFUSED");
  nhip_search_t s1c = s1, s2c = s2;
  if (fused) {
    s1c.flags |= SEARCH_I_NO_FINALIZE;
    s2c.flags |= SEARCH_I_KEYS_ZERO | SEARCH_I_NO_FINALIZE;
  }
  if ((rc = dropin_enqueue(S, n_a, T.g1, spec1_now, &s1c, S.delta1.p, S.rot1.p, false, 1, per1, S.keys.p, S.ws.p, S.ws.bytes, dres))) return rc;
  hipLaunchKernelGGL(dropin_bridge_kernel, dim3(1), dim3(64), 0, nullptr, reinterpret_cast<const nhip_match_t *>(dres),
                     reinterpret_cast<const double *>(static_cast<uint8_t *>(S.rot1.p) + 256), (s1.nx - 1) / 2, (s1.ny - 1) / 2,
                     spec1.res, T.spec2.res, n_a, parts2, per2, static_cast<DropInPar *>(S.par2.p),
                     reinterpret_cast<int32_t *>(dres + 480),
                     fused ? static_cast<const unsigned long long *>(S.keys.p) : nullptr, s1.nx, s1.ny, T.g1->L.Lf, T.g1->L.step,
                     reinterpret_cast<nhip_match_t *>(dres), reinterpret_cast<int32_t *>(dres + 128),
                     static_cast<unsigned long long *>(S.keys2.p));
  NHIP_TRY_HIP(hipGetLastError());
  if ((rc = dropin_enqueue(S, n_a, T.g2, spec2_now, &s2c, S.delta2.p, S.par2.p, true, parts2, per2, S.keys2.p, S.ws2.p, S.ws2.bytes, dres + 256)))
    return rc;
  NHIP_TRY_HIP(hipMemcpyAsync(down, dres, DROPIN_DOWN_BYTES, hipMemcpyDeviceToHost, nullptr));
  NHIP_TRY_HIP(hipStreamSynchronize(nullptr));
  DropInRes r1, r2;
  memcpy(&r1, down, sizeof(r1));
  memcpy(&r2, down + 256, sizeof(r2));
  dropin_pick(r1, 1, per1, &s1, m1);
  dropin_pick(r2, parts2, per2, &s2, m2);
  return NHIP_OK;
}

}  // namespace

int nhip_csm_cache_configure(int64_t max_bytes) {
  NHIP_REQUIRE(max_bytes >= 0, "csm_cache_configure: negative size");
  std::vector<std::shared_ptr<CachedTarget>> drop;  // (freed outside the lock)
  std::lock_guard<std::mutex> lock(g_cache_mu);
  g_cache_cap = max_bytes;
  int64_t tot = 0;
  size_t keep = 0;
  for (; keep < g_cache.size(); keep++) {
    if (tot + g_cache[keep]->bytes > g_cache_cap) break;
    tot += g_cache[keep]->bytes;
  }
  drop.assign(g_cache.begin() + (long)keep, g_cache.end());
  g_cache.resize(keep);
  return NHIP_OK;
}

int nhip_csm_cache_clear(void) {
  std::vector<std::shared_ptr<CachedTarget>> drop;
  std::lock_guard<std::mutex> lock(g_cache_mu);
  drop.swap(g_cache);
  return NHIP_OK;
}

int nhip_csm_cache_stats(int64_t *entries, int64_t *bytes, int64_t *hits, int64_t *misses) {
  std::lock_guard<std::mutex> lock(g_cache_mu);
  int64_t tot = 0;
  for (auto &e : g_cache) tot += e->bytes;
  if (entries) *entries = (int64_t)g_cache.size();
  if (bytes) *bytes = tot;
  if (hits) *hits = g_cache_hits;
  if (misses) *misses = g_cache_misses;
  return NHIP_OK;
}

int nhip_csm_get_transformation_info(double out[4]) {
  NHIP_REQUIRE(out != nullptr, "csm_get_transformation_info: null out");
  for (int i = 0; i < 4; i++) out[i] = t_dropin_info[i];
  return NHIP_OK;
}

int nhip_csm_get_transformation(const nhip_csm_params_t *p, const float *pc_a, int32_t n_a, const float *pc_b,
                                int32_t n_b, double rot_a, double rot_b, double rot_restriction, double *score,
                                float *tx, float *ty, float *theta) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(p && score && tx && ty && theta && n_a >= 0 && n_b >= 0 && (pc_a || n_a == 0) && (pc_b || n_b == 0),
               "csm_get_transformation: bad arguments");
  NHIP_REQUIRE(p->low_res > 0 && p->high_res > 0 && p->low_res >= p->high_res && p->trans_range >= 0 && rot_restriction >= 0,
               "csm_get_transformation: bad search parameters");
  const int32_t bits = p->cell_bits == 0 ? 16 : p->cell_bits;
  // (sums are int32, as in nhip_csm_match: the longest source cloud whose largest possible sum fits)
  NHIP_REQUIRE((int64_t)n_a <= 0x7fffffffll / (bits == 16 ? 65535 : 255), "csm_get_transformation: pc_a has %d points; with %d-bit "
               "cells at most %lld fit the int32 sums", n_a, bits, (long long)(0x7fffffffll / (bits == 16 ? 65535 : 255)));
  int device = 0;
  NHIP_TRY_HIP(hipGetDevice(&device));
  double theta0 = rot_a - rot_b;  // math_util.h:81-89 AngleDiff
  theta0 -= (2.0 * M_PI) * rint(theta0 / (2.0 * M_PI));
  const double coarse_step = M_PI / 180.0;
  // level 1: low_res grid, whole translation range, +-rot_restriction
  const int32_t h1 = (int32_t)floor(p->trans_range / p->low_res);
  const nhip_grid_spec_t spec1 = {p->scanner_range, p->low_res, p->sigma, p->floor_p, h1, bits, 0, 0};
  // (The coarse lattice is many rotations of few translations -- 181 x 13 x 13 at the reference's constants: the kernel that
  //  performs every add spreads its rotations over the whole chip, where the branch-and-bound matcher would compute the
  //  bounds of all of them in the pair's ONE workgroup, 23 rounds of its eight waves.  Same records either way.)
  const char *l1 = tunable("NHIP_DROPIN_COARSE");  // (measurement: "bnb" keeps the branch-and-bound matcher)
  const bool coarse_every_add = !(l1 && l1[0] == 'b') && (2 * h1 + 1) <= 21;
  const nhip_search_t s1 = {2 * (int32_t)floor(rot_restriction / coarse_step) + 1, 2 * h1 + 1, 2 * h1 + 1,
                            coarse_every_add ? NHIP_SEARCH_EXHAUSTIVE : 0, coarse_step};
  // level 2: high_res grid, +-low_res around the coarse optimum, +-1 coarse step in 0.1 steps.  Its tables are built for
  // the largest search centre a coarse optimum can produce (+ the fine half-width), so that they serve every source.
  const int32_t ratio = (int32_t)lround(p->low_res / p->high_res);
  const int32_t reach_max = (int32_t)lround((double)h1 * p->low_res / p->high_res) + ratio + 2;
  // (the score the call returns is the fine optimum's on the UNQUANTISED table -- NHIP_SEARCH_EXACT_SCORE: the reference's
  //  table holds doubles, cimg_debug.h:19; both searches run on the quantised tables)
  // The fine level performs EVERY add, in the kernel whose lanes are poses (NHIP_SEARCH_LATENCY: 21 x 16 tiles of four rows of
  // the 61 x 61 plane, ~45 us whatever the clouds).  Round 6 measured the alternatives (profiles/r06_dropin_fine_level.txt):
  // the branch-and-bound matcher takes 110 us where the clouds match and 0.8-10 ms where they do not -- a flat landscape
  // leaves it thousands of candidate blocks on a table whose pooled level does not fit LDS; the strip kernels 0.16-0.48 ms.
  // NHIP_DROPIN_FINE=bnb / strips (under NHIP_TUNABLES=1) select those.  Same records in every form.
  const char *l2 = tunable("NHIP_DROPIN_FINE");
  const int32_t fine_flags = (l2 && l2[0] == 'b') ? 0 : ((l2 && l2[0] == 's') ? NHIP_SEARCH_EXHAUSTIVE : (NHIP_SEARCH_EXHAUSTIVE | NHIP_SEARCH_LATENCY));
  const nhip_search_t s2 = {21, 2 * ratio + 1, 2 * ratio + 1, NHIP_SEARCH_EXACT_SCORE | fine_flags, coarse_step / 10.0};
  const bool cacheable = reach_max <= 4096 && n_b > 0;

  // ---- the target's tables: from the cache, or built now
  std::shared_ptr<CachedTarget> T;
  const size_t b_bytes = sizeof(float) * 2 * (size_t)n_b;
  const uint64_t h = cacheable ? hash_bytes(pc_b, b_bytes) ^ (uint64_t)n_b : 0;
  if (cacheable) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    for (size_t i = 0; i < g_cache.size(); i++) {
      CachedTarget &c = *g_cache[i];
      if (c.hash == h && c.device == device && c.cloud.size() == 2 * (size_t)n_b && same_params(c.params, *p) &&
          memcmp(c.cloud.data(), pc_b, b_bytes) == 0) {
        T = g_cache[i];
        g_cache.erase(g_cache.begin() + (long)i);
        g_cache.insert(g_cache.begin(), T);
        g_cache_hits++;
        break;
      }
    }
    if (!T) g_cache_misses++;
  }
  nhip_match_t m1;
  float tx1, ty1, th1;
  if (!T) {
    T = std::make_shared<CachedTarget>();
    T->hash = h;
    T->params = *p;
    T->device = device;
    if (n_b) T->cloud.assign(pc_b, pc_b + 2 * (size_t)n_b);
    const int32_t off[2] = {0, n_b}, target = 0;
    nhip_scans_t *bs = nullptr;
    if ((rc = nhip_scans_upload(pc_b, off, 1, &bs))) return rc;
    T->spec1 = spec1;
    rc = nhip_grids_build(bs, &target, 1, &spec1, &T->g1);
    if (rc == NHIP_OK && cacheable) {
      T->spec2 = {p->scanner_range, p->high_res, p->sigma, p->floor_p, reach_max, bits, 0, 0};
      rc = nhip_grids_build(bs, &target, 1, &T->spec2, &T->g2);
    }
    if (rc) {
      nhip_scans_free(bs);
      return rc;
    }
    if (!cacheable) {
      // (no common reach, or an empty target: the fine grid is sized by this call's coarse optimum, below; nothing is kept.
      //  The target's scan table must outlive the coarse match: matched here.)
      DropInScratch *S0 = nullptr;
      if ((rc = scratch_for(device, n_a, s1, s2, &S0))) { nhip_scans_free(bs); return rc; }
      hipError_t e = hipSuccess;
      if (n_a) e = hipMemcpy(S0->xy.p, pc_a, sizeof(float) * 2 * (size_t)n_a, hipMemcpyHostToDevice);
      if (e != hipSuccess) { nhip_scans_free(bs); return hip_fail(e, "csm_get_transformation upload", __FILE__, __LINE__); }
      rc = match_one(*S0, n_a, T->g1, &s1, S0->delta1.p, theta0, nullptr, &m1);
      if (rc == NHIP_OK) rc = nhip_match_to_transform(&m1, &spec1, &s1, theta0, 0, 0, &tx1, &ty1, &th1);
      if (rc) { nhip_scans_free(bs); return rc; }
      const int32_t origin[2] = {(int32_t)lround((double)tx1 / p->high_res), (int32_t)lround((double)ty1 / p->high_res)};
      const int32_t reach = std::max(abs(origin[0]), abs(origin[1])) + ratio;
      T->spec2 = {p->scanner_range, p->high_res, p->sigma, p->floor_p, reach, bits, 0, 0};
      rc = nhip_grids_build(bs, &target, 1, &T->spec2, &T->g2);
      nhip_scans_free(bs);
      if (rc) return rc;
      nhip_match_t m2;
      const double theta1 = th1;
      if ((rc = match_one(*S0, n_a, T->g2, &s2, S0->delta2.p, theta1, origin, &m2))) return rc;
      if ((rc = nhip_match_to_transform(&m2, &T->spec2, &s2, theta1, origin[0], origin[1], tx, ty, theta))) return rc;
      *score = (double)m2.score;
      return NHIP_OK;
    }
    nhip_scans_free(bs);
    T->bytes = (int64_t)T->g1->grids.bytes + (int64_t)T->g2->grids.bytes;
    std::vector<std::shared_ptr<CachedTarget>> drop;  // (evicted entries are freed outside the lock; a thread still matching
    {                                                  //  against one keeps it alive through its own shared_ptr)
      std::lock_guard<std::mutex> lock(g_cache_mu);
      // (two threads that missed on the same target at once both built it: the second finds the first's entry and keeps
      //  its own tables for this call only, so that no target counts twice against the cap)
      bool have = false;
      for (auto &c : g_cache)
        have = have || (c->hash == h && c->device == device && c->cloud.size() == T->cloud.size() && same_params(c->params, *p) &&
                        memcmp(c->cloud.data(), T->cloud.data(), b_bytes) == 0);
      if (!have && T->bytes <= g_cache_cap) {
        g_cache.insert(g_cache.begin(), T);
        int64_t tot = 0;
        size_t keep = 0;
        for (; keep < g_cache.size(); keep++) {
          if (tot + g_cache[keep]->bytes > g_cache_cap) break;
          tot += g_cache[keep]->bytes;
        }
        drop.assign(g_cache.begin() + (long)keep, g_cache.end());
        g_cache.resize(keep);
      }
    }
  }
  // ---- the two searches of this source against the target's tables, chained on the device: upload, coarse search, bridge
  // kernel (the fine level's parameter block from the coarse record), fine search, exact score, ONE download, ONE
  // synchronisation.  (No pinned staging, a coarse search of more rotations than the chain's table holds, or
  // NHIP_DROPIN_CHAIN=0: one level after the other with the host in between.)
  DropInScratch *S = nullptr;
  if ((rc = scratch_for(device, n_a, s1, s2, &S))) return rc;
  nhip_match_t m2;
  const char *ch = tunable("NHIP_DROPIN_CHAIN");
  rc = (ch && ch[0] == '0') ? NHIP_ERR_STATE : match_chained(*S, pc_a, n_a, *T, s1, s2, spec1, theta0, &m1, &m2);
  const bool chained = rc == NHIP_OK;
  if (rc != NHIP_OK && rc != NHIP_ERR_STATE) return rc;
  if (!chained) {
    if (n_a) NHIP_TRY_HIP(hipMemcpyAsync(S->xy.p, pc_a, sizeof(float) * 2 * (size_t)n_a, hipMemcpyHostToDevice, nullptr));
    if ((rc = match_one(*S, n_a, T->g1, &s1, S->delta1.p, theta0, nullptr, &m1))) return rc;
  }
  if ((rc = nhip_match_to_transform(&m1, &spec1, &s1, theta0, 0, 0, &tx1, &ty1, &th1))) return rc;
  const int32_t origin[2] = {(int32_t)lround((double)tx1 / p->high_res), (int32_t)lround((double)ty1 / p->high_res)};
  NHIP_REQUIRE(std::max(abs(origin[0]), abs(origin[1])) + ratio <= reach_max, "csm_get_transformation: coarse optimum (%d, %d) beyond "
               "the fine tables' reach %d", origin[0], origin[1], reach_max);
  const double theta1 = th1;
  t_dropin_info[0] = (double)m1.score;
  t_dropin_info[1] = (s2.flags & NHIP_SEARCH_LATENCY) ? 2.0 : ((s2.flags & NHIP_SEARCH_EXHAUSTIVE) ? 1.0 : 0.0);
  t_dropin_info[2] = chained ? 1.0 : 0.0;
  t_dropin_info[3] = (double)m1.itheta;
  if (!chained && (rc = match_one(*S, n_a, T->g2, &s2, S->delta2.p, theta1, origin, &m2))) return rc;
  if ((rc = nhip_match_to_transform(&m2, &T->spec2, &s2, theta1, origin[0], origin[1], tx, ty, theta))) return rc;
  *score = (double)m2.score;
  return NHIP_OK;
}

int nhip_resid_batch_create(int kind, const float *corr, const int32_t *block_offsets,
                            const int32_t *block_src, const int32_t *block_tgt, int32_t n_blocks,
                            int32_t n_poses, nhip_resid_batch_t **out) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(kind == NHIP_LIDAR_NORMAL || kind == NHIP_LIDAR_POINT, "resid_batch_create: bad kind %d", kind);
  NHIP_REQUIRE(block_offsets && block_src && block_tgt && out && n_blocks >= 0 && n_poses >= 0,
               "resid_batch_create: bad arguments");
  NHIP_REQUIRE(block_offsets[0] == 0, "resid_batch_create: block_offsets[0] must be 0");
  for (int32_t b = 0; b < n_blocks; b++) {
    // slam_residuals.h:109 CHECK_GT(source_points.size(), 0)
    NHIP_REQUIRE(block_offsets[b + 1] > block_offsets[b], "resid_batch_create: block %d is empty", b);
    NHIP_REQUIRE(block_src[b] >= 0 && block_src[b] < n_poses && block_tgt[b] >= 0 && block_tgt[b] < n_poses,
                 "resid_batch_create: block %d pose index out of range", b);
  }
  const int64_t n_corr = block_offsets[n_blocks];
  NHIP_REQUIRE(n_corr == 0 || corr, "resid_batch_create: null corr");
  std::vector<int32_t> cb((size_t)n_corr);
  for (int32_t b = 0; b < n_blocks; b++)
    for (int32_t i = block_offsets[b]; i < block_offsets[b + 1]; i++) cb[i] = b;
  nhip_resid_batch *B = new nhip_resid_batch();
  B->kind = kind;
  B->n_blocks = n_blocks;
  B->n_poses = n_poses;
  B->n_corr = n_corr;
  B->h_offsets.assign(block_offsets, block_offsets + n_blocks + 1);
  if ((rc = B->jtt.alloc(sizeof(double) * 2 * (size_t)n_corr)) || (rc = B->one_poses.alloc(sizeof(double) * 6)) ||
      (rc = B->one_consts.alloc(sizeof(double) * 8)) || (rc = B->one_idx.alloc(sizeof(int32_t) * 2))) {
    delete B;
    return rc;
  }
  {
    const int32_t idx[2] = {0, 1};
    hipError_t e0 = hipMemcpy(B->one_idx.p, idx, sizeof(idx), hipMemcpyHostToDevice);
    if (e0 != hipSuccess) {
      delete B;
      return hip_fail(e0, "resid_batch_create memcpy", __FILE__, __LINE__);
    }
  }
  if ((rc = B->corr.alloc(sizeof(float) * 8 * (size_t)n_corr)) || (rc = B->corr_block.alloc(sizeof(int32_t) * (size_t)n_corr)) ||
      (rc = B->block_src.alloc(sizeof(int32_t) * (size_t)n_blocks)) || (rc = B->block_tgt.alloc(sizeof(int32_t) * (size_t)n_blocks)) ||
      (rc = B->consts.alloc(sizeof(double) * 8 * (size_t)n_blocks)) || (rc = B->poses.alloc(sizeof(double) * 3 * (size_t)n_poses)) ||
      (rc = B->res.alloc(sizeof(double) * 2 * (size_t)n_corr)) || (rc = B->jsrc.alloc(sizeof(double) * 6 * (size_t)n_corr)) ||
      (rc = B->jtgt.alloc(sizeof(double) * 6 * (size_t)n_corr))) {
    delete B;
    return rc;
  }
  hipError_t e = hipSuccess;
  if (n_corr) {
    e = hipMemcpy(B->corr.p, corr, sizeof(float) * 8 * (size_t)n_corr, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(B->corr_block.p, cb.data(), sizeof(int32_t) * (size_t)n_corr, hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && n_blocks) {
    e = hipMemcpy(B->block_src.p, block_src, sizeof(int32_t) * (size_t)n_blocks, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(B->block_tgt.p, block_tgt, sizeof(int32_t) * (size_t)n_blocks, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    delete B;
    return hip_fail(e, "resid_batch_create memcpy", __FILE__, __LINE__);
  }
  *out = B;
  return NHIP_OK;
}

int nhip_resid_batch_eval(nhip_resid_batch_t *B, const double *poses, double *residuals,
                          double *jac_src, double *jac_tgt) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(B && poses && residuals, "resid_batch_eval: bad arguments");
  if (B->n_corr == 0) return NHIP_OK;
  NHIP_TRY_HIP(hipMemcpy(B->poses.p, poses, sizeof(double) * 3 * (size_t)B->n_poses, hipMemcpyHostToDevice));
  rc = launch_resid_lidar(B->kind, static_cast<const float *>(B->corr.p), static_cast<const int32_t *>(B->corr_block.p),
                          B->n_corr, static_cast<const int32_t *>(B->block_src.p),
                          static_cast<const int32_t *>(B->block_tgt.p), B->n_blocks,
                          static_cast<const double *>(B->poses.p), B->n_poses, static_cast<double *>(B->consts.p),
                          static_cast<double *>(B->res.p), jac_src ? static_cast<double *>(B->jsrc.p) : nullptr,
                          jac_tgt ? static_cast<double *>(B->jtgt.p) : nullptr, nullptr);
  if (rc) return rc;
  NHIP_TRY_HIP(hipMemcpy(residuals, B->res.p, sizeof(double) * 2 * (size_t)B->n_corr, hipMemcpyDeviceToHost));
  if (jac_src) NHIP_TRY_HIP(hipMemcpy(jac_src, B->jsrc.p, sizeof(double) * 6 * (size_t)B->n_corr, hipMemcpyDeviceToHost));
  if (jac_tgt) NHIP_TRY_HIP(hipMemcpy(jac_tgt, B->jtgt.p, sizeof(double) * 6 * (size_t)B->n_corr, hipMemcpyDeviceToHost));
  return NHIP_OK;
}

int nhip_resid_batch_eval_compact(nhip_resid_batch_t *B, const double *poses, double *residuals, double *jac_src,
                                  double *jac_tgt_theta) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(B && poses && residuals && jac_src && jac_tgt_theta, "resid_batch_eval_compact: bad arguments");
  if (B->n_corr == 0) return NHIP_OK;
  NHIP_TRY_HIP(hipMemcpy(B->poses.p, poses, sizeof(double) * 3 * (size_t)B->n_poses, hipMemcpyHostToDevice));
  rc = launch_resid_lidar(B->kind, static_cast<const float *>(B->corr.p), static_cast<const int32_t *>(B->corr_block.p),
                          B->n_corr, static_cast<const int32_t *>(B->block_src.p),
                          static_cast<const int32_t *>(B->block_tgt.p), B->n_blocks,
                          static_cast<const double *>(B->poses.p), B->n_poses, static_cast<double *>(B->consts.p),
                          static_cast<double *>(B->res.p), static_cast<double *>(B->jsrc.p), nullptr, nullptr,
                          static_cast<double *>(B->jtt.p), 0);
  if (rc) return rc;
  // three copies on the stream the kernel ran on; into pinned memory (nhip_host_alloc) they run at the PCIe rate
  NHIP_TRY_HIP(hipMemcpyAsync(residuals, B->res.p, sizeof(double) * 2 * (size_t)B->n_corr, hipMemcpyDeviceToHost, nullptr));
  NHIP_TRY_HIP(hipMemcpyAsync(jac_src, B->jsrc.p, sizeof(double) * 6 * (size_t)B->n_corr, hipMemcpyDeviceToHost, nullptr));
  NHIP_TRY_HIP(hipMemcpyAsync(jac_tgt_theta, B->jtt.p, sizeof(double) * 2 * (size_t)B->n_corr, hipMemcpyDeviceToHost, nullptr));
  NHIP_TRY_HIP(hipStreamSynchronize(nullptr));
  return NHIP_OK;
}

int nhip_resid_batch_eval_q(nhip_resid_batch_t *B, const double *poses, double *residuals, double *q, double *block_consts) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(B && poses && residuals && q && block_consts, "resid_batch_eval_q: bad arguments");
  if (B->n_corr == 0) return NHIP_OK;
  NHIP_TRY_HIP(hipMemcpy(B->poses.p, poses, sizeof(double) * 3 * (size_t)B->n_poses, hipMemcpyHostToDevice));
  // (residual-only instantiation + one more 16-byte store per correspondence; q goes where the compact form keeps J_tgt's
  //  theta column: the two forms of one batch are not in flight together -- the handle's calls synchronise)
  rc = launch_resid_lidar(B->kind, static_cast<const float *>(B->corr.p), static_cast<const int32_t *>(B->corr_block.p),
                          B->n_corr, static_cast<const int32_t *>(B->block_src.p),
                          static_cast<const int32_t *>(B->block_tgt.p), B->n_blocks,
                          static_cast<const double *>(B->poses.p), B->n_poses, static_cast<double *>(B->consts.p),
                          static_cast<double *>(B->res.p), nullptr, nullptr, nullptr, nullptr, 0, static_cast<double *>(B->jtt.p));
  if (rc) return rc;
  NHIP_TRY_HIP(hipMemcpyAsync(residuals, B->res.p, sizeof(double) * 2 * (size_t)B->n_corr, hipMemcpyDeviceToHost, nullptr));
  NHIP_TRY_HIP(hipMemcpyAsync(q, B->jtt.p, sizeof(double) * 2 * (size_t)B->n_corr, hipMemcpyDeviceToHost, nullptr));
  NHIP_TRY_HIP(hipMemcpyAsync(block_consts, B->consts.p, sizeof(double) * 8 * (size_t)B->n_blocks, hipMemcpyDeviceToHost, nullptr));
  NHIP_TRY_HIP(hipStreamSynchronize(nullptr));
  return NHIP_OK;
}

int nhip_resid_jacobians_from_q(int kind, const float *corr, const double *q, const double *block_consts, int64_t n,
                                double *jac_src, double *jac_tgt) {
  NHIP_REQUIRE(kind == NHIP_LIDAR_NORMAL || kind == NHIP_LIDAR_POINT, "resid_jacobians_from_q: bad kind %d", kind);
  NHIP_REQUIRE(n >= 0 && (n == 0 || (corr && q && block_consts)), "resid_jacobians_from_q: bad arguments");
  // the closed forms of resid_lidar_kernel (nhip_resid.hip) with u = q - t; consts = {l00, l01, l10, l11, tx, ty, i00, i01}
  const double tx = block_consts[4], ty = block_consts[5], i00 = block_consts[6], i01 = block_consts[7], i10 = -i01, i11 = i00;
  for (int64_t i = 0; i < n; i++) {
    const double qx = q[2 * i], qy = q[2 * i + 1], ux = qx - tx, uy = qy - ty;
    double js[6], jt[6];
    if (kind == NHIP_LIDAR_NORMAL) {
      const double nsx = corr[8 * i + 4], nsy = corr[8 * i + 5], ntx = corr[8 * i + 6], nty = corr[8 * i + 7];
      js[0] = ntx * i00 + nty * i10;
      js[1] = ntx * i01 + nty * i11;
      js[2] = ntx * (-uy) + nty * ux;
      js[3] = -(nsx * i00 + nsy * i10);
      js[4] = -(nsx * i01 + nsy * i11);
      js[5] = -(nsx * (-uy) + nsy * ux);
      jt[0] = -js[0];
      jt[1] = -js[1];
      jt[2] = ntx * qy - nty * qx;
      jt[3] = -js[3];
      jt[4] = -js[4];
      jt[5] = -(nsx * qy - nsy * qx);
    } else {
      js[0] = -i00; js[1] = -i01; js[2] = uy;
      js[3] = -i10; js[4] = -i11; js[5] = -ux;
      jt[0] = i00;  jt[1] = i01;  jt[2] = -qy;
      jt[3] = i10;  jt[4] = i11;  jt[5] = qx;
    }
    if (jac_src) memcpy(jac_src + 6 * i, js, sizeof(js));
    if (jac_tgt) memcpy(jac_tgt + 6 * i, jt, sizeof(jt));
  }
  return NHIP_OK;
}

int nhip_resid_batch_eval_block(nhip_resid_batch_t *B, int32_t block, const double *source_pose,
                                const double *target_pose, double *residuals, double *jac_src, double *jac_tgt) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(B && source_pose && target_pose && residuals, "resid_batch_eval_block: bad arguments");
  NHIP_REQUIRE(block >= 0 && block < B->n_blocks, "resid_batch_eval_block: block %d out of range", block);
  const int64_t o = B->h_offsets[block], n = B->h_offsets[block + 1] - o;
  double two[6];
  memcpy(two, source_pose, 3 * sizeof(double));
  memcpy(two + 3, target_pose, 3 * sizeof(double));
  std::lock_guard<std::mutex> lk(B->one_mu);  // one scratch set per batch: callers on several threads take turns
  NHIP_TRY_HIP(hipMemcpy(B->one_poses.p, two, sizeof(two), hipMemcpyHostToDevice));
  // the block's slice of the batch; its rows carry block id `block`, the one set of constants sits at index 0
  rc = launch_resid_lidar(B->kind, static_cast<const float *>(B->corr.p) + 8 * o,
                          static_cast<const int32_t *>(B->corr_block.p) + o, n,
                          static_cast<const int32_t *>(B->one_idx.p), static_cast<const int32_t *>(B->one_idx.p) + 1, 1,
                          static_cast<const double *>(B->one_poses.p), 2, static_cast<double *>(B->one_consts.p),
                          static_cast<double *>(B->res.p) + 2 * o, jac_src ? static_cast<double *>(B->jsrc.p) + 6 * o : nullptr,
                          jac_tgt ? static_cast<double *>(B->jtgt.p) + 6 * o : nullptr, nullptr, nullptr, block);
  if (rc) return rc;
  NHIP_TRY_HIP(hipMemcpy(residuals, static_cast<double *>(B->res.p) + 2 * o, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost));
  if (jac_src) NHIP_TRY_HIP(hipMemcpy(jac_src, static_cast<double *>(B->jsrc.p) + 6 * o, sizeof(double) * 6 * (size_t)n, hipMemcpyDeviceToHost));
  if (jac_tgt) NHIP_TRY_HIP(hipMemcpy(jac_tgt, static_cast<double *>(B->jtgt.p) + 6 * o, sizeof(double) * 6 * (size_t)n, hipMemcpyDeviceToHost));
  return NHIP_OK;
}

int nhip_host_alloc(size_t bytes, void **out) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(out, "host_alloc: null out");
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault);
  if (e != hipSuccess) {
    *out = nullptr;
    set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return NHIP_ERR_ALLOC;
  }
  return NHIP_OK;
}

int nhip_host_free(void *p) {
  if (p) NHIP_TRY_HIP(hipHostFree(p));
  return NHIP_OK;
}

int nhip_resid_batch_free(nhip_resid_batch_t *batch) {
  delete batch;
  return NHIP_OK;
}

namespace {
// upload n bytes (n may be 0)
int up(DevBuf &d, const void *h, size_t n) {
  int rc = d.alloc(n);
  if (rc) return rc;
  if (n) NHIP_TRY_HIP(hipMemcpy(d.p, h, n, hipMemcpyHostToDevice));
  return NHIP_OK;
}
}  // namespace

int nhip_resid_odometry(const float *t_odom, const float *r_odom, const int32_t *pose_i,
                        const int32_t *pose_j, int32_t n, double tw, double rw, const double *poses,
                        int32_t n_poses, double *residuals, double *jac_i, double *jac_j) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(n >= 0 && n_poses >= 0, "resid_odometry: negative size");
  if (n == 0) return NHIP_OK;
  NHIP_REQUIRE(t_odom && r_odom && pose_i && pose_j && poses && residuals, "resid_odometry: null pointer");
  for (int32_t f = 0; f < n; f++)
    NHIP_REQUIRE(pose_i[f] >= 0 && pose_i[f] < n_poses && pose_j[f] >= 0 && pose_j[f] < n_poses,
                 "resid_odometry: factor %d pose index out of range", f);
  DevBuf dt, dr, di, dj, dp, res, ji, jj;
  if ((rc = up(dt, t_odom, sizeof(float) * 2 * (size_t)n)) || (rc = up(dr, r_odom, sizeof(float) * (size_t)n)) ||
      (rc = up(di, pose_i, sizeof(int32_t) * (size_t)n)) || (rc = up(dj, pose_j, sizeof(int32_t) * (size_t)n)) ||
      (rc = up(dp, poses, sizeof(double) * 3 * (size_t)n_poses)) || (rc = res.alloc(sizeof(double) * 3 * (size_t)n)) ||
      (rc = ji.alloc(sizeof(double) * 9 * (size_t)n)) || (rc = jj.alloc(sizeof(double) * 9 * (size_t)n)))
    return rc;
  InFlight inflight;
  rc = launch_resid_odometry(static_cast<const float *>(dt.p), static_cast<const float *>(dr.p),
                             static_cast<const int32_t *>(di.p), static_cast<const int32_t *>(dj.p), n, tw, rw,
                             static_cast<const double *>(dp.p), n_poses, static_cast<double *>(res.p),
                             jac_i ? static_cast<double *>(ji.p) : nullptr,
                             jac_j ? static_cast<double *>(jj.p) : nullptr, nullptr);
  if (rc) return rc;
  NHIP_TRY_HIP(hipMemcpy(residuals, res.p, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost));
  if (jac_i) NHIP_TRY_HIP(hipMemcpy(jac_i, ji.p, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost));
  if (jac_j) NHIP_TRY_HIP(hipMemcpy(jac_j, jj.p, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost));
  InFlight::done();  // (the downloads above synchronised the null stream)
  return NHIP_OK;
}

int nhip_resid_point_to_line(const float *segments, const float *points, const int32_t *point_block,
                             int64_t n_points, const int32_t *block_pose, const int32_t *block_line,
                             int32_t n_blocks, const double *poses, int32_t n_poses,
                             const double *line_poses, int32_t n_line_poses, double *residuals,
                             double *jac_pose, double *jac_line) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(n_points >= 0 && n_blocks >= 0 && n_poses >= 0 && n_line_poses >= 0, "resid_point_to_line: negative size");
  if (n_points == 0) return NHIP_OK;
  NHIP_REQUIRE(segments && points && point_block && block_pose && block_line && poses && line_poses && residuals,
               "resid_point_to_line: null pointer");
  for (int32_t b = 0; b < n_blocks; b++)
    NHIP_REQUIRE(block_pose[b] >= 0 && block_pose[b] < n_poses && block_line[b] >= 0 && block_line[b] < n_line_poses,
                 "resid_point_to_line: block %d parameter index out of range", b);
  for (int64_t i = 0; i < n_points; i++)
    NHIP_REQUIRE(point_block[i] >= 0 && point_block[i] < n_blocks, "resid_point_to_line: point %lld block out of range",
                 (long long)i);
  DevBuf ds, dpt, dpb, dbp, dbl, dp, dl, res, j0, j1;
  if ((rc = up(ds, segments, sizeof(float) * 4 * (size_t)n_blocks)) || (rc = up(dpt, points, sizeof(float) * 2 * (size_t)n_points)) ||
      (rc = up(dpb, point_block, sizeof(int32_t) * (size_t)n_points)) || (rc = up(dbp, block_pose, sizeof(int32_t) * (size_t)n_blocks)) ||
      (rc = up(dbl, block_line, sizeof(int32_t) * (size_t)n_blocks)) || (rc = up(dp, poses, sizeof(double) * 3 * (size_t)n_poses)) ||
      (rc = up(dl, line_poses, sizeof(double) * 3 * (size_t)n_line_poses)) || (rc = res.alloc(sizeof(double) * (size_t)n_points)) ||
      (rc = j0.alloc(sizeof(double) * 3 * (size_t)n_points)) || (rc = j1.alloc(sizeof(double) * 3 * (size_t)n_points)))
    return rc;
  InFlight inflight;
  rc = launch_resid_point_to_line(static_cast<const float *>(ds.p), static_cast<const float *>(dpt.p),
                                  static_cast<const int32_t *>(dpb.p), n_points, static_cast<const int32_t *>(dbp.p),
                                  static_cast<const int32_t *>(dbl.p), n_blocks, static_cast<const double *>(dp.p), n_poses,
                                  static_cast<const double *>(dl.p), n_line_poses, static_cast<double *>(res.p),
                                  jac_pose ? static_cast<double *>(j0.p) : nullptr,
                                  jac_line ? static_cast<double *>(j1.p) : nullptr, nullptr);
  if (rc) return rc;
  NHIP_TRY_HIP(hipMemcpy(residuals, res.p, sizeof(double) * (size_t)n_points, hipMemcpyDeviceToHost));
  if (jac_pose) NHIP_TRY_HIP(hipMemcpy(jac_pose, j0.p, sizeof(double) * 3 * (size_t)n_points, hipMemcpyDeviceToHost));
  if (jac_line) NHIP_TRY_HIP(hipMemcpy(jac_line, j1.p, sizeof(double) * 3 * (size_t)n_points, hipMemcpyDeviceToHost));
  InFlight::done();  // (the downloads above synchronised the null stream)
  return NHIP_OK;
}

// ---------------------------------------------------------------- multi-GPU all-gather
namespace {
// ncclAllGather(sendbuff, recvbuff, sendcount, datatype, comm, stream); ncclInt8 = 0 (rccl.h)
typedef int (*nccl_allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef const char *(*nccl_errstr_fn)(int);
nccl_allgather_fn g_allgather = nullptr;
nccl_errstr_fn g_errstr = nullptr;
std::once_flag g_rccl_once;

void bind_rccl() {
  // RTLD_NOLOAD first: a process that already holds an RCCL (PyTorch ships its own copy) must
  // keep using that one, since the communicator came from it
  void *h = nullptr;
  for (const char *name : {"librccl.so.1", "librccl.so"}) {
    h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    if (h) break;
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) return;
  g_allgather = reinterpret_cast<nccl_allgather_fn>(dlsym(h, "ncclAllGather"));
  g_errstr = reinterpret_cast<nccl_errstr_fn>(dlsym(h, "ncclGetErrorString"));
}
}  // namespace

int nhip_allgather_matches(void *comm, const nhip_match_t *d_local, int32_t n_local, nhip_match_t *d_all,
                           void *stream) {
  int rc = require_device();
  if (rc) return rc;
  NHIP_REQUIRE(n_local >= 0, "allgather_matches: negative count");
  if (n_local == 0) return NHIP_OK;
  NHIP_REQUIRE(comm && d_local && d_all, "allgather_matches: null pointer");
  std::call_once(g_rccl_once, bind_rccl);
  if (!g_allgather) {
    const char *why = dlerror();
    set_error("allgather_matches: librccl.so could not be loaded (%s)", why ? why : "no ncclAllGather symbol");
    return NHIP_ERR_STATE;
  }
  const int st = g_allgather(d_local, d_all, sizeof(nhip_match_t) * (size_t)n_local, /*ncclInt8=*/0, comm,
                             static_cast<hipStream_t>(stream));
  if (st != 0) {
    set_error("allgather_matches: ncclAllGather failed: %s", g_errstr ? g_errstr(st) : "unknown");
    return NHIP_ERR_HIP;
  }
  return NHIP_OK;
}

int nhip_bnb_stats_per_pair(uint64_t *evaluated, int32_t n_pairs) {
  NHIP_REQUIRE(evaluated && n_pairs >= 0, "bnb_stats_per_pair: bad arguments");
  return bnb_stats_per_pair(reinterpret_cast<unsigned long long *>(evaluated), n_pairs);
}

int nhip_bnb_timeline(uint64_t *ticks, int32_t n_pairs) {
  NHIP_REQUIRE(ticks && n_pairs >= 0, "bnb_timeline: bad arguments");
  return bnb_timeline_read(reinterpret_cast<unsigned long long *>(ticks), n_pairs);
}

int nhip_bnb_timeline_candidates(uint64_t *ticks, int32_t n_pairs) {
  NHIP_REQUIRE(ticks && n_pairs >= 0, "bnb_timeline_candidates: bad arguments");
  return bnb_timeline_cand_read(reinterpret_cast<unsigned long long *>(ticks), n_pairs);
}

int nhip_bnb_stats(uint64_t *evaluated, uint64_t *total) {
  unsigned long long v[16];
  int rc = bnb_stats_read(v);
  if (rc) return rc;
  if (evaluated) *evaluated = v[0] + (v[3] + 3) / 4;  // in blocks: four sub-blocks = one block
  if (total) *total = v[1];
  return NHIP_OK;
}

int nhip_bnb_stats_levels(uint64_t out[16]) {
  NHIP_REQUIRE(out != nullptr, "bnb_stats_levels: null out");
  unsigned long long v[16];
  int rc = bnb_stats_read(v);
  if (rc) return rc;
  for (int i = 0; i < 16; i++) out[i] = v[i];
  return NHIP_OK;
}

// ---------------------------------------------------------------- timing
int nhip_timing_enable(int on) {
  std::lock_guard<std::mutex> lk(g_tmu);
  g_timing = on != 0;
  return NHIP_OK;
}

int nhip_timing_reset(void) {
  std::lock_guard<std::mutex> lk(g_tmu);
  for (auto &s : g_slots) {
    for (auto &p : s.ev) {
      (void)hipEventDestroy(p.first);
      (void)hipEventDestroy(p.second);
    }
    s.ev.clear();
    if (s.open) (void)hipEventDestroy(s.open);
    s.open = nullptr;
  }
  return NHIP_OK;
}

int nhip_timing_get(int id, double *total_ms, int32_t *launches) {
  NHIP_REQUIRE(id >= 0 && id < NHIP_TIMER_COUNT, "timing_get: bad id %d", id);
  std::lock_guard<std::mutex> lk(g_tmu);
  double tot = 0.0;
  for (auto &p : g_slots[id].ev) {
    NHIP_TRY_HIP(hipEventSynchronize(p.second));
    float ms = 0.f;
    NHIP_TRY_HIP(hipEventElapsedTime(&ms, p.first, p.second));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (int32_t)g_slots[id].ev.size();
  return NHIP_OK;
}

}  // extern "C"
