// fleet_capi.hip -- host side of the C ABI declared in include/fleet_hip.h (libfleet_hip.so).
//
// Owns: device copies of the pre-staged tables (re-packed into the rows the kernels read), the SoA state of the
// env batch, one HIP stream, staging buffers for the host-pointer entry points, a cached hipGraph for tape replays.
// There is no CPU path in this library: without a HIP device fleet_create fails with FLEET_ERR_NODEVICE.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <unistd.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fleet_device.h"
#include "fleet_direct.h"

namespace {

thread_local std::string g_create_error;

// Host-side copies of the host-pointer path (observations from the pinned landing buffer to a pageable destination): one core
// reads memory the device has just written at ~20 GB/s, so the pieces of a transfer are copied by a few worker threads while
// the calling thread waits for the next piece to land.  One pool per process, created at the first use and never torn down
// (its threads sleep on a condition variable); a forked child makes its own.
class CopyPool {
 public:
  static CopyPool* get() {
    static std::mutex mk;
    static CopyPool* pool = nullptr;
    std::lock_guard<std::mutex> g(mk);
    if (!pool || pool->pid_ != getpid()) pool = new CopyPool(3);  // (a fork leaves the parent's object behind: no threads in it)
    return pool;
  }
  void submit(void* dst, const void* src, size_t n) {
    pending_.fetch_add(1, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> g(m_);
      q_.push_back({dst, src, n});
    }
    cv_.notify_one();
  }
  void wait() {  // all submitted copies done (the caller's own: calls on one handle are serialised, pools are per process)
    while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  }

 private:
  struct Task { void* dst; const void* src; size_t n; };
  explicit CopyPool(int workers) : pid_(getpid()) {
    for (int i = 0; i < workers; ++i) std::thread([this] { run(); }).detach();
  }
  void run() {
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !q_.empty(); });
        t = q_.front();
        q_.pop_front();
      }
      memcpy(t.dst, t.src, t.n);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  pid_t pid_;
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<Task> q_;
  std::atomic<int> pending_{0};
};

struct Batch {
  FleetParams p{};
  FleetDev d{};
  int device = 0;
  hipStream_t stream = nullptr;      // the stream launches go to: the handle's own one, or an adopted one (fleet_set_stream)
  hipStream_t own_stream = nullptr;  // created with the handle, destroyed with it; never handed out of the library's control
  std::string error;
  std::vector<void*> allocs;
  // staging for the *_host entry points
  void* st_actions = nullptr;
  float* st_obs = nullptr;
  float* st_term = nullptr;
  double* st_reward = nullptr;
  uint8_t* st_done = nullptr;
  uint8_t* st_mask = nullptr;
  // host path: {reward f64[E], finished-episode returns f64[E], count i32 (+pad), idx i32[E], lengths i32[E], done u8[E]} in
  // ONE device block = one transfer; the
  // compacted terminal rows; pinned host mirrors (PCIe at full rate, no pageable staging by the runtime)
  char* st_small = nullptr;
  size_t small_bytes = 0, small_off_count = 0, small_off_idx = 0, small_off_done = 0, small_off_ret = 0, small_off_len = 0;
  float* st_term_compact = nullptr;
  char* pin_small = nullptr;
  void* pin_actions = nullptr;
  float* pin_term = nullptr;
  // observations to a pageable destination: pinned landing buffer (allocated at the first such step) + one event per piece
  char* pin_obs = nullptr;
  static constexpr int kObsPieces = 3;  // every transfer on the link costs ~15 us of its own: 8 pieces halve the link rate (measured)
  hipEvent_t obs_piece_ev[kObsPieces] = {};
  bool host_step_has_episodes = false;
  uint32_t last_step_err = 0;  // OR of the device error bits as of the last fleet_step_host
  double* st_dist = nullptr;
  int32_t* dev_sched = nullptr;
  FleetCold cold_host{};
  FleetCold* cold_dev = nullptr;
  FleetDev* self_dev = nullptr;
  void* st_field = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  std::vector<hipEvent_t> region_events;  // fleet_time_regions_begin / _read
  // cached tape graph
  hipGraphExec_t graph_exec = nullptr;
  const void* graph_tape = nullptr;
  int graph_len = 0, graph_dtype = 0;
  float* graph_obs = nullptr;
  double* graph_reward = nullptr;
  uint8_t* graph_done = nullptr;
  // direct AQL submission of tape runs (fleet_direct.hip; FLEET_LAUNCH_DIRECT): the handle's own HSA queue, what its prepared
  // argument blocks describe, the spans of the timed runs waited for so far
  FleetDirect* direct = nullptr;
  const void* dq_tape = nullptr;
  int dq_len = 0, dq_dtype = 0, dq_mode = 0;
  float* dq_obs = nullptr;
  float* dq_term = nullptr;
  double* dq_reward = nullptr;
  uint8_t* dq_done = nullptr;
  uint64_t dq_gen = 0;   // the handle's generation the prepared argument blocks were made in
  int dq_timed = 0;      // 0 / 1 (first and last packet of a run) / 2 (every packet): fleet_direct_submit
  std::vector<double> dq_spans_us;
  // Every call that changes what a launch's argument block embeds (the handle's streams, its start schedule, its policy parameters:
  // anything a later version may move into FleetDev) bumps the generation: argument blocks prepared before it are never reused.
  uint64_t gen = 1;
};

// A run submitted to the handle's own queue is not on its HIP stream: every entry point that touches the handle waits for it first.
static int direct_drain(Batch* h) {
  if (!h || !h->direct) return FLEET_OK;
  return fleet_direct_wait(h->direct, &h->dq_spans_us, &h->error);
}
#define FLEET_ENTER(h)                          \
  do {                                          \
    const int _rc = direct_drain(h);            \
    if (_rc != FLEET_OK) return _rc;            \
  } while (0)

#define HIP_TRY(b, expr)                                                                         \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      (b)->error = std::string(#expr) + ": " + hipGetErrorString(_e);                            \
      return FLEET_ERR_HIP;                                                                      \
    }                                                                                            \
  } while (0)

template <typename T>
int dev_alloc(Batch* b, T** out, size_t count, bool zero = true) {
  void* ptr = nullptr;
  const size_t bytes = (count ? count : 1) * sizeof(T);
  HIP_TRY(b, hipMalloc(&ptr, bytes));
  b->allocs.push_back(ptr);
  if (zero) HIP_TRY(b, hipMemsetAsync(ptr, 0, bytes, b->stream));
  *out = static_cast<T*>(ptr);
  return FLEET_OK;
}

template <typename T>
int dev_upload(Batch* b, const T** out, const T* host, size_t count) {
  T* ptr = nullptr;
  int rc = dev_alloc(b, &ptr, count, false);
  if (rc) return rc;
  HIP_TRY(b, hipMemcpyAsync(ptr, host, count * sizeof(T), hipMemcpyHostToDevice, b->stream));
  *out = ptr;
  return FLEET_OK;
}

int obs_dim_of(const FleetParams* p) {
  const int N = p->num_cars, L = p->price_lookahead, B = p->bl_pv_lookahead;
  int dim = 2 * N + (L + 1) * 2;
  if (p->include_building && p->include_pv)
    dim += 2 * (B + 1);
  else if (p->include_building || p->include_pv)
    dim += B + 1;
  if (p->aux) {
    dim += 5 * N + 1 + 6;
    if (p->include_building) dim += 3;
  }
  return dim;
}

const char* validate(const FleetParams* p, const FleetTables* t) {
  if (!p || !t) return "null params/tables";
  if (p->abi_version != FLEET_ABI_VERSION) return "abi_version mismatch";
  if (p->struct_bytes != (int)sizeof(FleetParams)) return "FleetParams size mismatch";
  if (p->num_envs < 1 || p->num_cars < 1 || p->table_rows < 2) return "num_envs/num_cars/table_rows out of range";
  if (p->num_cars > 65535) return "num_cars: at most 65535 EVs per env";  // (the step kernel's packed argument, fleet_kernels.hip `p_N`)
  if (p->episode_steps < 1 || p->steps_per_hour < 1) return "episode_steps/steps_per_hour out of range";
  if (p->episode_steps >= FLEET_MAX_EPISODE_STEPS) return "episode_steps exceeds 2^29 - 1 (the env head's sample count is 29 bits wide)";
  if (p->price_lookahead < 0 || p->bl_pv_lookahead < 0) return "negative look-ahead";
  if (p->deg_mode < FLEET_DEG_NONE || p->deg_mode > FLEET_DEG_RAINFLOW) return "unknown deg_mode";
  if (p->deg_mode == FLEET_DEG_RAINFLOW && p->init_soh != 1.0)
    return "rainflow/SEI degradation needs init_soh == 1.0 (the reference's used-battery branch is ill-defined, quirk Q4)";
  // the rainflow stack size travels in a 26-bit field of the hot record (fleet_device.h HOT_PACK)
  if (p->deg_mode == FLEET_DEG_RAINFLOW && p->episode_steps > FLEET_MAX_STACK_ROWS - 3)
    return "rainflow/SEI degradation: episode_steps exceeds 67 million (the packed rainflow stack size is 26 bits wide)";
  // ... and the kernels address an EV's rainflow row as (its env's rows) + a 32-bit byte offset
  if (p->deg_mode == FLEET_DEG_RAINFLOW && (uint64_t)p->num_cars * ((uint64_t)p->episode_steps + 24) * 8ull >= (1ull << 32))
    return "rainflow/SEI degradation: num_cars x episode_steps too large (the rainflow rows of one env exceed 4 GiB)";
  if (p->table_rows >= 0x3FFFFFFF) return "table_rows exceeds 2^30 - 1 (segment ends are 30 bits wide)";
  if (t->finish_row)
    for (int r = 0; r < p->table_rows; ++r) {
      if (t->finish_row[r] >= p->table_rows) return "finish_row entry outside the table";
      if (p->deg_mode == FLEET_DEG_RAINFLOW && t->finish_row[r] - r > FLEET_MAX_STACK_ROWS - 3)
        return "rainflow/SEI degradation: an episode spans more than 67 million rows (the packed rainflow stack size is 26 bits wide)";
    }
  if (t->lookahead_row)
    for (size_t k = 0; k < (size_t)p->table_rows * (size_t)t->lookahead_cols; ++k)
      if (t->lookahead_row[k] >= p->table_rows) return "lookahead_row entry outside the table";
  if (p->normalize && p->include_pv && !p->include_building)
    return "normalize with pv but without building load crashes in the reference (quirk Q4); unsupported";
  if ((t->dt_row != nullptr) != (t->finish_row != nullptr) || (t->dt_row && (!t->lookahead_row || t->lookahead_cols < 1)))
    return "irregular-grid tables must be given together (dt_row, finish_row, lookahead_row)";
  if (t->dt_row && !p->real_time) return "an irregular time grid needs real_time = 1";
  if (t->lookahead_row && (t->lookahead_cols < p->price_lookahead || t->lookahead_cols < p->bl_pv_lookahead))
    return "lookahead_cols smaller than a look-ahead";
  if (p->log_capacity < 0) return "negative log_capacity";
  if (t->pick_rows && t->n_pick_rows < 1) return "empty pick_rows";
  if (p->start_lo < 0 || p->start_hi < p->start_lo || p->start_hi > (t->pick_rows ? t->n_pick_rows : p->table_rows) - 1)
    return "start range outside the table";
  if (t->pick_rows)
    for (int i = 0; i < t->n_pick_rows; ++i)
      if (t->pick_rows[i] < 0 || t->pick_rows[i] > p->table_rows - 1) return "pick_rows entry outside the table";
  if (!t->there || !t->time_left || !t->soc_on_return || !t->delu || !t->tariff || !t->prc || !t->trc || !t->load ||
      !t->pv || !t->hour || !t->minute || !t->month || !t->weekday)
    return "a required table pointer is null";
  return nullptr;
}

// hourly look-ahead row: `resample("H").first()` of the slice starting at t (observer_bl_pv.py:50-80):
// bucket 0 = row t, bucket k>=1 = first row of clock hour floor_hour(t)+k.
inline int lookahead_row(const FleetParams& p, const FleetTables& tb, int t, int k) {
  if (k == 0) return t;
  if (tb.lookahead_row) {  // irregular grid: tabulated by date on the host
    const int r = tb.lookahead_row[(size_t)t * tb.lookahead_cols + (k - 1)];
    return r < 0 ? p.table_rows - 1 : r;
  }
  int r = ((t + p.hour_phase) / p.steps_per_hour + k) * p.steps_per_hour - p.hour_phase;
  return r > p.table_rows - 1 ? p.table_rows - 1 : r;
}

// Env-level observation blocks are a pure function of the table row: assemble (and normalise) them once, in
// float64 with the reference's operation order, and store the float32 words the reference would emit.
//   block A: price[L+1] | tariff[L+1] | building_load[B+1]* | pv[B+1]*      (observer_bl_pv.py:50-80)
//   block B: evse | grid_cap† | avail_grid_cap† | possible_avg_action† | month/week/hour sin,cos  (:92-107)
void build_tail_rows(const FleetParams& p, const FleetTables& t, int tail_a, int tail_b, int stride, std::vector<float>& out) {
  const int T = p.table_rows, L = p.price_lookahead, B = p.bl_pv_lookahead, N = p.num_cars;
  const bool norm = p.normalize != 0;
  out.assign((size_t)T * stride, 0.0f);
  const double two_pi = 2 * M_PI;
  for (int r = 0; r < T; ++r) {
    float* o = out.data() + (size_t)r * stride;
    int k0 = 0;
    for (int k = 0; k <= L; ++k) {
      double v = (t.delu[lookahead_row(p, t, r, k)] + p.fixed_markup) * p.variable_multiplier;
      if (norm) v = (v - p.min_price) / (p.max_price - p.min_price);
      o[k0++] = (float)v;
    }
    for (int k = 0; k <= L; ++k) {
      double v = t.tariff[lookahead_row(p, t, r, k)] * (1 - p.feed_in_deduction);
      if (norm) v = (v - p.min_tariff) / (p.max_tariff - p.min_tariff);
      o[k0++] = (float)v;
    }
    double load0 = 0.0, pv0 = 0.0;
    if (p.include_building) {
      load0 = t.load[r];
      for (int k = 0; k <= B; ++k) {
        double v = t.load[lookahead_row(p, t, r, k)];
        if (norm) v = v / p.max_building;
        o[k0++] = (float)v;
      }
    }
    if (p.include_pv) {
      pv0 = t.pv[r];
      for (int k = 0; k <= B; ++k) {
        double v = t.pv[lookahead_row(p, t, r, k)];
        if (norm) v = v / p.max_pv;
        o[k0++] = (float)v;
      }
    }
    if (!p.aux) continue;
    o[k0++] = (float)(norm ? p.evse_power / p.max_evse : p.evse_power);
    if (p.include_building) {
      const double grid_cap = p.grid_connection;
      const double avail = grid_cap - load0 + pv0;
      const double q = avail / (N * p.evse_power);
      const double pavg = q < 1 ? q : 1;
      o[k0++] = (float)(norm ? grid_cap / p.max_grid : grid_cap);
      o[k0++] = (float)(norm ? avail / p.max_grid : avail);
      o[k0++] = (float)pavg;
    }
    if (t.time_feat) {
      for (int k = 0; k < 6; ++k) o[k0++] = t.time_feat[(size_t)r * 6 + k];
    } else {
      o[k0++] = (float)std::sin(two_pi * t.month[r] / 12);
      o[k0++] = (float)std::cos(two_pi * t.month[r] / 12);
      o[k0++] = (float)std::sin(two_pi * t.weekday[r] / 7);
      o[k0++] = (float)std::cos(two_pi * t.weekday[r] / 7);
      o[k0++] = (float)std::sin(two_pi * t.hour[r] / 24);
      o[k0++] = (float)std::cos(two_pi * t.hour[r] / 24);
    }
    (void)tail_a; (void)tail_b;
  }
}

// Physics rows: only combinations the reference itself evaluates on per-time scalars, same float64 operations
// in the same order, so the stored doubles are bit-identical to what the reference computes per step.
void build_phys_rows(const FleetParams& p, const FleetTables& t, std::vector<PhysRow>& phys, std::vector<uint8_t>& flags) {
  const int T = p.table_rows, N = p.num_cars;
  phys.resize(T);
  flags.resize(T);
  const double spot_offset = p.fixed_markup / 1000;  // ev_charger.py:34
  for (int r = 0; r < T; ++r) {
    PhysRow& q = phys[r];
    q.k_cost = (t.delu[r] / 1000.0 + spot_offset) * p.variable_multiplier;  // (current_spot + spot_offset) * spot_multiplier :145-149
    q.k_rev = -1 * p.discharging_eff * t.tariff[r] / 1000 * (1 - p.feed_in_deduction);  // :196-199 without the energy factor
    q.k_charge = -1 * p.price_multiplier * t.prc[r] / 1000;       // :154-155
    q.k_discharge = -1 * p.price_multiplier * t.trc[r] / 1000;    // :204-205
    q.load = p.include_building ? t.load[r] : 0.0;
    q.pv = p.include_pv ? t.pv[r] : 0.0;
    // connected_cars = max(sum(There[t]), 1) is a function of the time row alone (:138-140), so
    // current_pv_energy / connected_cars (:134,142) can be tabulated with the reference's own two operations
    long connected = 0;
    for (int c = 0; c < N; ++c) connected += t.there[(size_t)r * N + c];
    if (connected < 1) connected = 1;
    const double pv_energy = p.include_pv ? t.pv[r] * (t.dt_row ? t.dt_row[r] : p.dt) : 0.0;
    q.pv_share = pv_energy / (double)connected;
    q.pad = 0;
    q.dt = t.dt_row ? t.dt_row[r] : p.dt;
    uint8_t f = 0;
    if (t.hour[r] == 14 && t.minute[r] == 45) f |= FLEET_TFLAG_DEG;
    if (t.hour[r] > 11 && t.hour[r] < 15) f |= FLEET_TFLAG_LUNCH;
    flags[r] = f;
  }
  for (int r = 0; r < T; ++r) phys[r].flags_next = flags[r + 1 < T ? r + 1 : T - 1];
}

// Per-(t, EV) schedule records in run-length form (struct SegRec in fleet_device.h): consecutive rows of an EV with the same
// There, the same SOC_on_return (bit for bit) and a time_left that counts down to the same departure row form a segment and
// share one record.  Whether a row's float32 time_left is what the kernels derive from the departure row is checked here
// with the kernels' own expression (seg_tl); a row where it is not -- an irregular time grid, a hand-made table -- becomes
// a one-row segment that carries its time_left verbatim.
void build_seg_rows(const FleetParams& p, const FleetTables& t, std::vector<SegRec>& seg) {
  const int T = p.table_rows, N = p.num_cars;
  seg.resize((size_t)T * N);
  std::vector<uint8_t> raw((size_t)T);
  std::vector<uint32_t> dep((size_t)T);
  for (int c = 0; c < N; ++c) {
    for (int r = 0; r < T; ++r) {
      const float tl = t.time_left[(size_t)r * N + c];
      raw[r] = 0;
      dep[r] = 0;  // time_left == 0: no departure ahead
      if (tl != 0.0f) {
        const double k = (double)tl / p.dt;
        const long long kk = std::llround(k);
        SegRec probe;
        probe.sor = 0.0;
        probe.tlx = (uint32_t)(r + kk);
        probe.se = 0;
        if (!t.dt_row && kk >= 1 && (long long)r + kk < 0x3FFFFFFFll && seg_tl(probe, r, p.dt) == tl)
          dep[r] = (uint32_t)(r + kk);
        else
          raw[r] = 1;
      }
    }
    uint32_t end = (uint32_t)T;
    for (int r = T - 1; r >= 0; --r) {
      const size_t k = (size_t)r * N + c;
      if (r < T - 1) {
        const size_t k1 = k + N;
        uint64_t s0, s1;
        memcpy(&s0, &t.soc_on_return[k], 8);
        memcpy(&s1, &t.soc_on_return[k1], 8);
        const bool same = !raw[r] && !raw[r + 1] && t.there[k] == t.there[k1] && s0 == s1 && dep[r] == dep[r + 1];
        if (!same) end = (uint32_t)(r + 1);
      }
      SegRec& x = seg[k];
      x.sor = t.soc_on_return[k];
      if (raw[r]) {
        memcpy(&x.tlx, &t.time_left[k], 4);
      } else {
        x.tlx = dep[r];
      }
      x.se = end | (raw[r] ? SEG_RAW : 0u) | (t.there[k] ? 0x80000000u : 0u);
    }
  }
}

int create_impl(const FleetParams* p, const FleetTables* t, int device, Batch* b) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    b->error = "no HIP device visible; libfleet_hip has no CPU fallback";
    return FLEET_ERR_NODEVICE;
  }
  if (device < 0 || device >= ndev) {
    b->error = "device index out of range";
    return FLEET_ERR_INVALID;
  }
  b->p = *p;
  b->device = device;
  HIP_TRY(b, hipSetDevice(device));
  HIP_TRY(b, hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking));
  b->stream = b->own_stream;
  HIP_TRY(b, hipEventCreate(&b->ev_start));
  HIP_TRY(b, hipEventCreate(&b->ev_stop));

  FleetDev& d = b->d;
  const int E = p->num_envs, N = p->num_cars, T = p->table_rows, L = p->price_lookahead, B = p->bl_pv_lookahead;
  d.E = E; d.N = N; d.T = T;
  d.obs_dim = obs_dim_of(p);
  d.episode_steps = p->episode_steps;
  // rows of the rainflow stack workspace: pushes <= logged samples; on an irregular grid an episode spans as many rows as its
  // finish row says, not episode_steps
  int max_rows = p->episode_steps;
  if (t->finish_row)
    for (int r = 0; r < T; ++r)
      if (t->finish_row[r] - r > max_rows) max_rows = t->finish_row[r] - r;
  d.stack_cap = max_rows + 3;
  d.tail_a_len = 2 * (L + 1) + (p->include_building ? B + 1 : 0) + (p->include_pv ? B + 1 : 0);
  d.tail_b_len = p->aux ? (1 + (p->include_building ? 3 : 0) + 6) : 0;
  d.tail_stride = ((d.tail_a_len + d.tail_b_len + 3) / 4) * 4;
  d.aux = p->aux; d.normalize = p->normalize; d.is_caretaker = p->is_caretaker; d.deg_mode = p->deg_mode;
  d.auto_reset = p->auto_reset;
  d.real_time = p->real_time ? 1 : 0;
  d.carry_run = (N <= fleet_max_evs_per_lane_group()) ? 1 : 0;
  d.dt = p->dt; d.evse_power = p->evse_power;
  d.p_avail = p->obc_max_power < p->evse_power ? p->obc_max_power : p->evse_power;  // min([obc, evse]) ev_charger.py:95
  d.init_cap = p->init_battery_cap; d.grid_connection = p->grid_connection;
  d.eta_c = p->charging_eff; d.eta_d = p->discharging_eff;
  // stress_temp (rainflow_sei_degradation.py:72-73) is a constant of the batch
  d.stress_temp = std::exp(6.93E-2 * (p->temperature - 25.0) * ((25.0 + 273.15) / (p->temperature + 273.15)));
  d.penalty_invalid = p->penalty_invalid_action; d.penalty_oc = p->penalty_overcharging; d.clip_oc = p->clip_overcharging;
  d.penalty_overload = p->penalty_overloading; d.fully_charged_reward = p->fully_charged_reward;
  d.target_soc = p->target_soc; d.target_soc_lunch = p->target_soc_lunch; d.eps = p->eps;
  d.max_time_left = p->max_time_left;
  // auxiliary observation slots: divisions by constants become multiplications by the correctly rounded quotient / reciprocal
  d.hn_scale = p->batt_cap_nominal / (p->evse_power * p->charging_eff);
  d.inv_eta_c = 1.0 / p->charging_eff;

  FleetCold& cd = b->cold_host;
  cd.min_laxity = p->min_laxity; cd.def_soc = p->def_soc; cd.init_soh = p->init_soh; cd.temperature = p->temperature;
  cd.dt = p->dt; cd.batt_cap_nominal = p->batt_cap_nominal; cd.hn_denominator = p->evse_power * p->charging_eff;
  cd.max_soc = p->max_soc; cd.max_hours_needed = p->max_hours_needed; cd.max_laxity = p->max_laxity;
  cd.inv_max_soc = p->normalize ? 1.0 / p->max_soc : 1.0;
  cd.inv_max_hours_needed = p->normalize ? 1.0 / p->max_hours_needed : 1.0;
  cd.inv_max_laxity = p->normalize ? 1.0 / p->max_laxity : 1.0;
  cd.seed = p->seed; cd.picker_mode = p->picker_mode; cd.start_lo = p->start_lo; cd.start_hi = p->start_hi;
  cd.env_id_offset = p->env_id_offset; cd.sched_n = 0; cd.normalize = p->normalize; cd.sched = nullptr;

  // ---- tables ---------------------------------------------------------------------------------------
  int rc;
  {
    std::vector<PhysRow> phys;
    std::vector<uint8_t> flags;
    build_phys_rows(*p, *t, phys, flags);
    std::vector<float> tail;
    build_tail_rows(*p, *t, d.tail_a_len, d.tail_b_len, d.tail_stride, tail);
    std::vector<SegRec> seg;
    build_seg_rows(*p, *t, seg);
    if ((rc = dev_upload(b, &d.seg, seg.data(), seg.size()))) return rc;
    if ((rc = dev_upload(b, &d.tab_phys, phys.data(), phys.size()))) return rc;
    if ((rc = dev_upload(b, &d.tab_flags, flags.data(), flags.size()))) return rc;
    // the last degradation row at or before every row: where an episode's rainflow count may stop (EnvRec::rf_until)
    std::vector<int32_t> last_deg((size_t)T);
    int32_t last = -1;
    for (int r = 0; r < T; ++r) {
      if (flags[r] & FLEET_TFLAG_DEG) last = r;
      last_deg[r] = last;
    }
    if ((rc = dev_upload(b, &cd.tab_last_deg, last_deg.data(), last_deg.size()))) return rc;
    if ((rc = dev_upload(b, &d.tab_tail, tail.data(), tail.size()))) return rc;
    HIP_TRY(b, hipStreamSynchronize(b->stream));  // host vectors go out of scope here
  }
  {
    // night-charging policy (benchmarking/night_charging.py:81-98): clock of every table row + per-env window state
    std::vector<uint16_t> hm((size_t)T);
    for (int i = 0; i < T; ++i)
      hm[i] = (uint16_t)((t->hour[i] << 8) | t->minute[i] | ((t->second && t->second[i]) ? 0x8000 : 0));  // bit 15: off the minute
    if ((rc = dev_upload(b, &cd.tab_hm, hm.data(), hm.size()))) return rc;
    std::vector<int32_t> idle((size_t)E, FLEET_NIGHT_IDLE);
    if ((rc = dev_alloc(b, &cd.night_start, (size_t)E, false))) return rc;
    HIP_TRY(b, hipMemcpyAsync(cd.night_start, idle.data(), idle.size() * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
    cd.night_hour = -1; cd.night_minute = 0; cd.night_limit_s = 0;
    if ((rc = dev_alloc(b, &cd.last_len, (size_t)E))) return rc;  // (zeroed)
    cd.rf_count_all = 0;
    cd.step_s = (int)std::llround(p->dt * 3600.0);
    HIP_TRY(b, hipStreamSynchronize(b->stream));
  }
  if (t->pick_rows)
    if ((rc = dev_upload(b, &cd.pick_rows, t->pick_rows, (size_t)t->n_pick_rows))) return rc;
  if (t->finish_row) {  // irregular time grid (real_time): episode-end row by date (the per-row step length is in PhysRow)
    if ((rc = dev_upload(b, &d.tab_finish, t->finish_row, (size_t)T))) return rc;
    HIP_TRY(b, hipStreamSynchronize(b->stream));
  }
  if ((rc = dev_alloc(b, &b->cold_dev, 1))) return rc;
  HIP_TRY(b, hipMemcpyAsync(b->cold_dev, &cd, sizeof(FleetCold), hipMemcpyHostToDevice, b->stream));
  d.cold = b->cold_dev;

  // ---- state ----------------------------------------------------------------------------------------
  const size_t EN = (size_t)E * N;
  if ((rc = dev_alloc(b, &d.hot, EN))) return rc;
  if ((rc = dev_alloc(b, &d.run, EN))) return rc;
  if ((rc = dev_alloc(b, &d.soh, EN))) return rc;
  if ((rc = dev_alloc(b, &d.soc_deg, EN))) return rc;
  if ((rc = dev_alloc(b, &d.sei, EN))) return rc;
  if (p->log_data) {  // device-side data log: ring of log_capacity rows per env (default: two episodes incl. their reset rows)
    d.log_cap = p->log_capacity > 0 ? p->log_capacity : 2 * (p->episode_steps + 1);
    const size_t rows = (size_t)d.log_cap * E;
    if ((rc = dev_alloc(b, &d.log_pos, (size_t)E))) return rc;
    if ((rc = dev_alloc(b, &d.log_row, rows))) return rc;
    if ((rc = dev_alloc(b, &d.log_env, rows * 4))) return rc;
    if ((rc = dev_alloc(b, &d.log_ev, rows * 4 * N))) return rc;
    if ((rc = dev_alloc(b, &d.log_obs, rows * (size_t)d.obs_dim))) return rc;
  }
  if ((rc = dev_alloc(b, &d.env, E))) return rc;
  if (p->deg_mode == FLEET_DEG_RAINFLOW) {
    d.rf_row_stride = ((RF_HDR_WORDS + d.stack_cap + 15) / 16) * 16;  // RfHdr + stack, rounded to whole 128-byte lines
    // the kernels address an EV's row as (the env's rows, a scalar base) + a 32-bit byte offset
    if ((uint64_t)N * (uint64_t)d.rf_row_stride * 8ull >= (1ull << 32)) {
      b->error = "num_cars x episode length: the rainflow rows of one env exceed 4 GiB";
      return FLEET_ERR_INVALID;
    }
    if ((rc = dev_alloc(b, &d.rf_rows, EN * (size_t)d.rf_row_stride, false))) return rc;
    // headers: rainflow_length = 1 (rainflow_sei_degradation.py:57), everything else 0
    RfHdr h0;
    memset(&h0, 0, sizeof h0);
    h0.rf_len = 1;
    std::vector<RfHdr> hdrs(EN, h0);
    HIP_TRY(b, hipMemcpy2DAsync(d.rf_rows, (size_t)d.rf_row_stride * 8, hdrs.data(), sizeof(RfHdr), sizeof(RfHdr), EN,
                                hipMemcpyHostToDevice, b->stream));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
  }
  {
    // persistent degradation state (RainflowSeiDegradation.__init__, rainflow_sei_degradation.py:24-66), the initial
    // SoH and the (cleared) sticky target flags (fleet_environment.py:263)
    std::vector<double> soh(EN, p->init_soh);  // the hot records themselves are zero (no sticky target flag yet)
    std::vector<SeiRec> sei(EN);
    for (auto& q : sei) { q.fd_cyc = 0; q.fd_cal = 0; q.sei_soh = p->init_soh; q.sei_l = 1.0 - p->init_soh; }
    HIP_TRY(b, hipMemcpyAsync(d.soh, soh.data(), EN * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(b, hipMemcpyAsync(d.sei, sei.data(), EN * sizeof(SeiRec), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
  }
  if ((rc = dev_alloc(b, &b->self_dev, 1))) return rc;
  d.self = b->self_dev;
  // ---- staging for host entry points -------------------------------------------------------------------------------
  const size_t OD = (size_t)E * d.obs_dim;
  if ((rc = dev_alloc(b, (char**)&b->st_actions, EN * 8))) return rc;
  if ((rc = dev_alloc(b, &b->st_obs, OD))) return rc;
  if ((rc = dev_alloc(b, &b->st_term, OD))) return rc;
  {
    b->small_off_ret = (size_t)E * 8;
    b->small_off_count = b->small_off_ret + (size_t)E * 8;
    b->small_off_idx = b->small_off_count + 8;
    b->small_off_len = b->small_off_idx + (size_t)E * 4;
    b->small_off_done = b->small_off_len + (size_t)E * 4;
    b->small_bytes = b->small_off_done + (size_t)E;
    if ((rc = dev_alloc(b, &b->st_small, b->small_bytes))) return rc;
    b->st_reward = reinterpret_cast<double*>(b->st_small);
    b->st_done = reinterpret_cast<uint8_t*>(b->st_small + b->small_off_done);
    d.err_any = reinterpret_cast<uint32_t*>(b->st_small + b->small_off_count + 4);  // the word beside the episode count
    if ((rc = dev_alloc(b, &b->st_term_compact, OD))) return rc;
    HIP_TRY(b, hipHostMalloc((void**)&b->pin_small, b->small_bytes, hipHostMallocDefault));
    HIP_TRY(b, hipHostMalloc(&b->pin_actions, EN * 8, hipHostMallocDefault));
    HIP_TRY(b, hipHostMalloc((void**)&b->pin_term, OD * sizeof(float), hipHostMallocDefault));
  }
  if ((rc = dev_alloc(b, &b->st_mask, E))) return rc;
  if ((rc = dev_alloc(b, &b->st_dist, EN))) return rc;
  if ((rc = dev_alloc(b, (char**)&b->st_field, (EN > 2 * (size_t)E ? EN : 2 * (size_t)E) * 8))) return rc;  // a field, or the [2, E] gather block
  // device-resident copy of the (now complete) argument block for the out-of-line rare paths (reset, daily degradation)
  HIP_TRY(b, hipMemcpyAsync(b->self_dev, &d, sizeof(FleetDev), hipMemcpyHostToDevice, b->stream));
  HIP_TRY(b, hipStreamSynchronize(b->stream));
  return FLEET_OK;
}

// Name the first env that carries device error bits (the reference raises at the offending line: fleet_environment.py:610,
// rainflow_sei_degradation.py:164-167,179-180,209-210; running off the table is a KeyError of its `db.loc[...]`).
const char* deverr_names(uint32_t bits, char* buf, size_t n) {
  snprintf(buf, n, "%s%s%s%s%s%s%s",
           (bits & FLEET_DEVERR_PLACEMENT) ? " placement: a workgroup of a run on the library's own queue ran on another die than probed, the run's results are void;" : "",
           (bits & FLEET_DEVERR_INTERNAL) ? " internal: inconsistent launch arguments;" : "",
           (bits & FLEET_DEVERR_OBS_FORMAT) ? " observation format not recognized;" : "",
           (bits & FLEET_DEVERR_NEG_LIFE) ? " life degradation is negative;" : "",
           (bits & FLEET_DEVERR_SOH_MISMATCH) ? " degradation calculation is not correct;" : "",
           (bits & FLEET_DEVERR_DOD_RANGE) ? " DoD too large;" : "",
           (bits & FLEET_DEVERR_TABLE_END) ? " the episode runs past the last table row;" : "");
  return buf;
}

// RCCL, bound at run time: a process that never gathers across GPUs does not need librccl, and a process that has PyTorch in
// it gets the copy PyTorch has already mapped (same soname) instead of a second one.  The handful of NCCL-API declarations the
// binding needs are restated here (their ABI is fixed: rccl/rccl.h `ncclUniqueId` = 128 opaque bytes, `ncclSuccess` = 0,
// `ncclDouble` = 8), so that building this library does not need RCCL's headers either.
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[FLEET_RCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclDouble = 8;
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};
void rccl_bind(RcclApi& api);
RcclApi& rccl() {  // bound once, whichever thread asks first
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, rccl_bind, std::ref(api));
  return api;
}
void rccl_bind(RcclApi& api) {
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
    api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (api.lib) break;
  }
  if (!api.lib) {
    const char* e = dlerror();
    api.why = std::string("librccl not found: ") + (e ? e : "");
    return;
  }
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.lib, "ncclAllGather"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) api.why = "librccl lacks an expected symbol";
}

void drop_graph(Batch* b) {
  if (b->graph_exec) {
    (void)hipGraphExecDestroy(b->graph_exec);
    b->graph_exec = nullptr;
  }
}

}  // namespace

struct FleetEnvBatch : Batch {};

extern "C" {

int fleet_obs_dim(const FleetParams* p) {
  if (!p || p->num_cars < 1) return -1;
  return obs_dim_of(p);
}

int fleet_create(const FleetParams* p, const FleetTables* t, int device, fleet_handle* out) {
  if (out) *out = nullptr;
  if (const char* why = validate(p, t)) {
    g_create_error = why;
    return FLEET_ERR_INVALID;
  }
  if (!out) {
    g_create_error = "null output handle";
    return FLEET_ERR_INVALID;
  }
  FleetEnvBatch* b = new FleetEnvBatch();
  int rc = create_impl(p, t, device, b);
  if (rc != FLEET_OK) {
    g_create_error = b->error;
    fleet_destroy(b);
    return rc;
  }
  *out = b;
  return FLEET_OK;
}

int fleet_destroy(fleet_handle h) {
  if (!h) return FLEET_OK;
  (void)hipSetDevice(h->device);
  if (h->direct) fleet_direct_close(h->direct);  // waits for a run in flight
  h->direct = nullptr;
  (void)hipStreamSynchronize(h->stream);
  drop_graph(h);
  for (void* ptr : h->allocs) (void)hipFree(ptr);
  for (void* ptr : {(void*)h->pin_small, h->pin_actions, (void*)h->pin_term, (void*)h->pin_obs})
    if (ptr) (void)hipHostFree(ptr);
  for (auto& e : h->obs_piece_ev)
    if (e) (void)hipEventDestroy(e);
  if (h->dev_sched) (void)hipFree(h->dev_sched);
  for (auto& e : h->region_events)
    if (e) (void)hipEventDestroy(e);
  if (h->ev_start) (void)hipEventDestroy(h->ev_start);
  if (h->ev_stop) (void)hipEventDestroy(h->ev_stop);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return FLEET_OK;
}

const char* fleet_last_error(fleet_handle h) { return h ? h->error.c_str() : g_create_error.c_str(); }

int fleet_set_stream(fleet_handle h, void* hip_stream) {
  FLEET_ENTER(h);
  if (!h) return FLEET_ERR_INVALID;
  h->gen += 1;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // what was enqueued on the stream in use so far is finished before the switch
  drop_graph(h);
  // the handle's own stream is kept (fleet_set_stream(h, fleet_own_stream) or a later fleet_use_own_stream goes back to it);
  // an adopted stream is only borrowed: the caller keeps it alive while the handle uses it
  h->stream = static_cast<hipStream_t>(hip_stream);
  return FLEET_OK;
}

int fleet_use_own_stream(fleet_handle h) {
  FLEET_ENTER(h);
  if (!h) return FLEET_ERR_INVALID;
  h->gen += 1;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  drop_graph(h);
  h->stream = h->own_stream;
  return FLEET_OK;
}

int fleet_synchronize(fleet_handle h) {
  FLEET_ENTER(h);
  if (!h) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FLEET_OK;
}

int fleet_stream_query(fleet_handle h) {
  if (!h) return FLEET_ERR_INVALID;
  if (h->direct && fleet_direct_busy(h->direct)) return -1;
  const hipError_t e = hipStreamQuery(h->stream);
  if (e == hipSuccess) return FLEET_OK;
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    return -1;
  }
  h->error = std::string("hipStreamQuery: ") + hipGetErrorString(e);
  return FLEET_ERR_HIP;
}

int fleet_set_start_schedule(fleet_handle h, const int32_t* starts, int n_episodes) {
  FLEET_ENTER(h);
  if (!h || n_episodes < 0 || (n_episodes > 0 && !starts)) return FLEET_ERR_INVALID;
  h->gen += 1;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->dev_sched) {
    (void)hipFree(h->dev_sched);
    h->dev_sched = nullptr;
  }
  h->cold_host.sched = nullptr;
  h->cold_host.sched_n = 0;
  if (n_episodes > 0) {
    const size_t n = (size_t)n_episodes * h->d.E;
    for (size_t i = 0; i < n; ++i)
      if (starts[i] < 0 || starts[i] > h->d.T - 1) {
        h->error = "start row outside the table";
        return FLEET_ERR_INVALID;
      }
    HIP_TRY(h, hipMalloc((void**)&h->dev_sched, n * sizeof(int32_t)));
    HIP_TRY(h, hipMemcpy(h->dev_sched, starts, n * sizeof(int32_t), hipMemcpyHostToDevice));
    h->cold_host.sched = h->dev_sched;
    h->cold_host.sched_n = n_episodes;
  }
  // the cold block lives in device memory, so captured graphs stay valid
  HIP_TRY(h, hipMemcpy(h->cold_dev, &h->cold_host, sizeof(FleetCold), hipMemcpyHostToDevice));
  return FLEET_OK;
}

int fleet_reset_dev(fleet_handle h, const uint8_t* mask, float* obs) {
  FLEET_ENTER(h);
  if (!h || !obs) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, fleet_launch_reset(h->d, mask, obs, h->stream));
  return FLEET_OK;
}

int fleet_step_dev(fleet_handle h, const void* actions, int act_dtype, float* obs, double* reward, uint8_t* done,
                   float* terminal_obs) {
  FLEET_ENTER(h);
  if (!h || !actions || !obs || !reward || !done || (act_dtype != FLEET_ACT_F32 && act_dtype != FLEET_ACT_F64)) {
    if (h) h->error = "fleet_step_dev: null buffer or bad action dtype";
    return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, fleet_launch_step(h->d, actions, act_dtype, 1, obs, reward, done, terminal_obs, nullptr, h->stream));
  return FLEET_OK;
}

int fleet_step_many_dev(fleet_handle h, int K, const void* actions, int act_dtype, float* obs, double* reward_sum,
                        int32_t* done_count) {
  FLEET_ENTER(h);
  if (!h || K < 1 || !actions || !obs || !reward_sum || (act_dtype != FLEET_ACT_F32 && act_dtype != FLEET_ACT_F64)) {
    if (h) h->error = "fleet_step_many_dev: bad argument";
    return FLEET_ERR_INVALID;
  }
  if (!h->d.auto_reset) {
    h->error = "fleet_step_many_dev needs auto_reset = 1";
    return FLEET_ERR_INVALID;
  }
  if (h->d.real_time) {
    h->error = "fleet_step_many_dev is not available with real_time = 1 (each launch already spans a variable number of rows)";
    return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  // K == 1 without done_count is the single-step kernel (it writes the per-step reward = the sum of one, and the done flag
  // to the staging buffer); with done_count the launcher takes the multi-step kernel, which counts episode ends
  HIP_TRY(h, fleet_launch_step(h->d, actions, act_dtype, K, obs, reward_sum, h->st_done, nullptr, done_count, h->stream));
  return FLEET_OK;
}

int fleet_set_night_policy(fleet_handle h, int charging_hour, int charging_minute, int max_hours) {
  FLEET_ENTER(h);
  if (!h) return FLEET_ERR_INVALID;
  // charging_hour may be 24 (the reference's own edge when the window would open exactly at midnight: never opens)
  if (charging_hour < 0 || charging_hour > 24 || charging_minute < 0 || charging_minute > 59 || max_hours < 0 ||
      max_hours > 24 * 365) {
    h->error = "fleet_set_night_policy: argument out of range";
    return FLEET_ERR_INVALID;
  }
  h->gen += 1;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->cold_host.night_hour = charging_hour;
  h->cold_host.night_minute = charging_minute;
  h->cold_host.night_limit_s = 3600 * max_hours;
  std::vector<int32_t> idle((size_t)h->d.E, FLEET_NIGHT_IDLE);
  HIP_TRY(h, hipMemcpy(h->cold_host.night_start, idle.data(), idle.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(h->cold_dev, &h->cold_host, sizeof(FleetCold), hipMemcpyHostToDevice));
  return FLEET_OK;
}

int fleet_set_rainflow_count_all(fleet_handle h, int on) {
  FLEET_ENTER(h);
  if (!h) return FLEET_ERR_INVALID;
  h->gen += 1;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->cold_host.rf_count_all = on ? 1 : 0;
  HIP_TRY(h, hipMemcpy(h->cold_dev, &h->cold_host, sizeof(FleetCold), hipMemcpyHostToDevice));
  return FLEET_OK;
}

int fleet_rollout_policy_dev(fleet_handle h, int policy, int K, float* obs, double* reward_sum, int32_t* done_count) {
  FLEET_ENTER(h);
  if (!h || K < 1 || !obs || !reward_sum ||
      (policy != FLEET_ACT_POLICY_UNCONTROLLED && policy != FLEET_ACT_POLICY_DISTRIBUTED && policy != FLEET_ACT_POLICY_NIGHT)) {
    if (h) h->error = "fleet_rollout_policy_dev: bad argument";
    return FLEET_ERR_INVALID;
  }
  if (policy == FLEET_ACT_POLICY_NIGHT && h->cold_host.night_hour < 0) {
    h->error = "fleet_rollout_policy_dev: FLEET_ACT_POLICY_NIGHT needs fleet_set_night_policy first";
    return FLEET_ERR_INVALID;
  }
  if (!h->d.auto_reset) {
    h->error = "fleet_rollout_policy_dev needs auto_reset = 1";
    return FLEET_ERR_INVALID;
  }
  if (h->d.real_time) {
    h->error = "fleet_rollout_policy_dev is not available with real_time = 1";
    return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  // K >= 1 always takes the multi-step kernel (policies are only compiled into it); done_count may be NULL
  HIP_TRY(h, fleet_launch_step(h->d, nullptr, policy, K, obs, reward_sum, h->st_done, nullptr, done_count, h->stream));
  return FLEET_OK;
}

int fleet_reset_host(fleet_handle h, const uint8_t* mask, float* obs) {
  FLEET_ENTER(h);
  if (!h || !obs) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t OD = (size_t)h->d.E * h->d.obs_dim * sizeof(float);
  if (mask) {
    HIP_TRY(h, hipMemcpyAsync(h->st_mask, mask, h->d.E, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->st_obs, obs, OD, hipMemcpyHostToDevice, h->stream));  // keep unmasked rows as they were
  }
  HIP_TRY(h, fleet_launch_reset(h->d, mask ? h->st_mask : nullptr, h->st_obs, h->stream));
  HIP_TRY(h, hipMemcpyAsync(obs, h->st_obs, OD, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FLEET_OK;
}

int fleet_step_host(fleet_handle h, const void* actions, int act_dtype, float* obs, double* reward, uint8_t* done,
                    float* terminal_obs) {
  FLEET_ENTER(h);
  if (!h || !actions || !obs || !reward || !done || (act_dtype != FLEET_ACT_F32 && act_dtype != FLEET_ACT_F64)) {
    if (h) h->error = "fleet_step_host: null buffer or bad action dtype";
    return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  const int E = h->d.E;
  const size_t EN = (size_t)E * h->d.N;
  const size_t row = (size_t)h->d.obs_dim * sizeof(float);
  const size_t OD = (size_t)E * row;
  const size_t abytes = EN * (act_dtype == FLEET_ACT_F64 ? 8 : 4);
  // Action buffers from fleet_host_alloc are pinned: the transfer runs straight out of them.  Anything else goes through the
  // handle's pinned mirror (one small memcpy on the host instead of the runtime's pageable staging).
  hipPointerAttribute_t attr;
  auto pinned = [&](const void* p) {
    return hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeHost;
  };
  const bool act_pinned = pinned(actions);
  (void)hipGetLastError();  // hipPointerGetAttributes on a pageable pointer leaves an error code behind
  const void* asrc = actions;
  if (!act_pinned) {
    memcpy(h->pin_actions, actions, abytes);
    asrc = h->pin_actions;
  }
  HIP_TRY(h, hipMemcpyAsync(h->st_actions, asrc, abytes, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, fleet_launch_step(h->d, h->st_actions, act_dtype, 1, h->st_obs, h->st_reward, h->st_done,
                               terminal_obs ? h->st_term : nullptr, nullptr, h->stream));
  if (terminal_obs)
    HIP_TRY(h, fleet_launch_term_compact(h->d, h->st_done, h->st_term, reinterpret_cast<int32_t*>(h->st_small + h->small_off_idx),
                                         reinterpret_cast<int32_t*>(h->st_small + h->small_off_count),
                                         reinterpret_cast<double*>(h->st_small + h->small_off_ret),
                                         reinterpret_cast<int32_t*>(h->st_small + h->small_off_len), h->st_term_compact, h->stream));
  h->host_step_has_episodes = terminal_obs != nullptr;
  HIP_TRY(h, hipMemcpyAsync(h->pin_small, h->st_small, h->small_bytes, hipMemcpyDeviceToHost, h->stream));
  // The observations: straight into the caller's buffer at the link rate if it is pinned.  A pageable destination (a fresh
  // array per step, what the reference's env returns) gets them in pieces through a pinned landing buffer of the handle: each
  // piece is copied to its destination by this thread as soon as it has landed, while the following pieces are still on the
  // link -- the host copy (the slower of the two at 6 MB per step) hides the transfer instead of following it.
  const bool obs_pinned = pinned(obs);
  (void)hipGetLastError();
  if (obs_pinned || OD < (size_t)1 << 18) {
    HIP_TRY(h, hipMemcpyAsync(obs, h->st_obs, OD, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  } else {
    if (!h->pin_obs) {
      HIP_TRY(h, hipHostMalloc((void**)&h->pin_obs, OD, hipHostMallocDefault));
      for (auto& e : h->obs_piece_ev) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // the first piece is small (the host copy starts early), the rest equal; boundaries on 4 KiB
    size_t cut[Batch::kObsPieces + 1];
    constexpr int P = Batch::kObsPieces;
    cut[0] = 0;
    const size_t first = (OD / (2 * (size_t)P)) & ~(size_t)4095;
    for (int c = 1; c < P; ++c) cut[c] = (first + (OD - first) * (size_t)(c - 1) / (size_t)(P - 1)) & ~(size_t)4095;
    cut[P] = OD;
    const char* src = reinterpret_cast<const char*>(h->st_obs);
    for (int c = 0; c < P; ++c) {
      if (cut[c + 1] > cut[c])
        HIP_TRY(h, hipMemcpyAsync(h->pin_obs + cut[c], src + cut[c], cut[c + 1] - cut[c], hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipEventRecord(h->obs_piece_ev[c], h->stream));
    }
    char* dst = reinterpret_cast<char*>(obs);
    CopyPool* pool = CopyPool::get();
    hipError_t werr = hipSuccess;
    constexpr int kSplit = 4;  // host copies per piece: three workers + (for the last piece) this thread
    for (int c = 0; c < P && werr == hipSuccess; ++c) {
      werr = hipEventSynchronize(h->obs_piece_ev[c]);
      if (werr != hipSuccess || cut[c + 1] <= cut[c]) continue;
      const size_t lo = cut[c], n = cut[c + 1] - cut[c];
      const bool last = (c + 1 == P);
      const int parts = last ? kSplit : kSplit - 1;
      for (int k = 0; k < parts; ++k) {
        const size_t a0 = lo + ((n * (size_t)k / parts) & ~(size_t)63);
        const size_t a1 = (k + 1 == parts) ? lo + n : lo + ((n * (size_t)(k + 1) / parts) & ~(size_t)63);
        if (last && k + 1 == parts) memcpy(dst + a0, h->pin_obs + a0, a1 - a0);  // nothing left to wait for: this thread copies too
        else pool->submit(dst + a0, h->pin_obs + a0, a1 - a0);
      }
    }
    pool->wait();
    HIP_TRY(h, werr);
  }
  memcpy(reward, h->pin_small, (size_t)E * 8);
  memcpy(done, h->pin_small + h->small_off_done, (size_t)E);
  h->last_step_err = *reinterpret_cast<const uint32_t*>(h->pin_small + h->small_off_count + 4);  // came with the rewards
  if (terminal_obs) {
    // terminal observations only exist for the envs that finished in this step: only those rows cross PCIe; rows of envs
    // that did not finish are left untouched
    const int n = *reinterpret_cast<const int32_t*>(h->pin_small + h->small_off_count);
    if (n > 0) {
      const int32_t* idx = reinterpret_cast<const int32_t*>(h->pin_small + h->small_off_idx);
      HIP_TRY(h, hipMemcpyAsync(h->pin_term, h->st_term_compact, (size_t)n * row, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      for (int k = 0; k < n; ++k) memcpy(terminal_obs + (size_t)idx[k] * h->d.obs_dim, h->pin_term + (size_t)k * h->d.obs_dim, row);
    }
  }
  // Device error bits raised by this step (or left by an earlier one: they are sticky) are reported by this very call, like the
  // reference raises inside step(); the outputs above are complete.  No extra launch or transfer on the clean path.
  if (h->last_step_err) {
    (void)fleet_check_errors(h);  // names the env in fleet_last_error
    return FLEET_ERR_STATE;
  }
  return FLEET_OK;
}

int fleet_last_step_error_bits(fleet_handle h, uint32_t* bits) {
  FLEET_ENTER(h);
  if (!h || !bits) return FLEET_ERR_INVALID;
  *bits = h->last_step_err;
  return FLEET_OK;
}

int fleet_last_step_episodes(fleet_handle h, int32_t* n, const int32_t** env_idx, const double** ep_return, const int32_t** ep_len) {
  FLEET_ENTER(h);
  if (!h || !n) return FLEET_ERR_INVALID;
  if (!h->host_step_has_episodes) {
    h->error = "fleet_last_step_episodes: needs a preceding fleet_step_host with a terminal_obs buffer";
    return FLEET_ERR_INVALID;
  }
  *n = *reinterpret_cast<const int32_t*>(h->pin_small + h->small_off_count);
  if (env_idx) *env_idx = reinterpret_cast<const int32_t*>(h->pin_small + h->small_off_idx);
  if (ep_return) *ep_return = reinterpret_cast<const double*>(h->pin_small + h->small_off_ret);
  if (ep_len) *ep_len = reinterpret_cast<const int32_t*>(h->pin_small + h->small_off_len);
  return FLEET_OK;
}

int fleet_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return FLEET_ERR_INVALID;
  *out = nullptr;
  return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? FLEET_OK : FLEET_ERR_HIP;
}

int fleet_host_free(void* p) {
  if (!p) return FLEET_OK;
  return hipHostFree(p) == hipSuccess ? FLEET_OK : FLEET_ERR_HIP;
}

static size_t field_bytes(const FleetDev& d, int field) {
  const size_t E = d.E, EN = (size_t)d.E * d.N;
  size_t bytes = 0;
  switch (field) {
    case FLEET_F_SOC: case FLEET_F_SOH: case FLEET_F_SOC_DEG: case FLEET_F_TARGET_SOC: case FLEET_F_FD_CYC:
    case FLEET_F_FD_CAL: case FLEET_F_SEI_L: bytes = EN * 8; break;
    case FLEET_F_HOURS_LEFT: case FLEET_F_RF_LEN: case FLEET_F_RF_CYCLES: case FLEET_F_RF_STACK: bytes = EN * 4; break;
    case FLEET_F_CASHFLOW: case FLEET_F_EP_RETURN: case FLEET_F_LAST_EP_RETURN: case FLEET_F_PENALTY_RECORD:
    case FLEET_F_LAST_EP_LEN_F64:
    bytes = E * 8; break;
    case FLEET_F_TIME_IDX: case FLEET_F_START_IDX: case FLEET_F_EP_LEN: case FLEET_F_LAST_EP_LEN: case FLEET_F_ERROR_BITS:
    case FLEET_F_EPISODES: case FLEET_F_RF_UNTIL: bytes = E * 4; break;
    case FLEET_F_DONE: bytes = E; break;
    default: break;
  }
  return bytes;
}

int fleet_get_dev(fleet_handle h, int field, void* out_dev) {
  FLEET_ENTER(h);
  if (!h || !out_dev) return FLEET_ERR_INVALID;
  if (!field_bytes(h->d, field)) {
    h->error = "fleet_get_dev: unknown field";
    return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, fleet_launch_gather_field(h->d, field, out_dev, h->stream));
  return FLEET_OK;
}

int fleet_get(fleet_handle h, int field, void* out) {
  FLEET_ENTER(h);
  if (!h || !out) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t bytes = field_bytes(h->d, field);
  if (!bytes) {
    h->error = "fleet_get: unknown field";
    return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, fleet_launch_gather_field(h->d, field, h->st_field, h->stream));
  HIP_TRY(h, hipMemcpyAsync(out, h->st_field, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FLEET_OK;
}

int fleet_get_dist_factor(fleet_handle h, double* out) {
  FLEET_ENTER(h);
  if (!h || !out) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, fleet_launch_dist_factor(h->d, h->st_dist, h->stream));
  HIP_TRY(h, hipMemcpyAsync(out, h->st_dist, (size_t)h->d.E * h->d.N * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FLEET_OK;
}

int fleet_rccl_unique_id(void* id128) {
  if (!id128) return FLEET_ERR_INVALID;
  RcclApi& r = rccl();
  if (!r.why.empty()) { g_create_error = r.why; return FLEET_ERR_HIP; }
  static_assert(sizeof(ncclUniqueId) == FLEET_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  if (r.GetUniqueId(&id) != ncclSuccess) { g_create_error = "ncclGetUniqueId failed"; return FLEET_ERR_HIP; }
  memcpy(id128, &id, sizeof id);
  return FLEET_OK;
}

int fleet_rccl_comm_create(int device, int world_size, int rank, const void* id128, void** comm) {
  if (!id128 || !comm || world_size < 1 || rank < 0 || rank >= world_size) return FLEET_ERR_INVALID;
  RcclApi& r = rccl();
  if (!r.why.empty()) { g_create_error = r.why; return FLEET_ERR_HIP; }
  if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return FLEET_ERR_HIP; }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclComm_t c = nullptr;
  const ncclResult_t rc = r.CommInitRank(&c, world_size, id, rank);
  if (rc != ncclSuccess) {
    g_create_error = std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "failed");
    return FLEET_ERR_HIP;
  }
  *comm = c;
  return FLEET_OK;
}

int fleet_rccl_comm_destroy(void* comm) {
  if (!comm) return FLEET_OK;
  RcclApi& r = rccl();
  if (!r.why.empty()) return FLEET_ERR_HIP;
  return r.CommDestroy(static_cast<ncclComm_t>(comm)) == ncclSuccess ? FLEET_OK : FLEET_ERR_HIP;
}

int fleet_gather_episode_stats_rccl(fleet_handle h, void* comm, int world_size, double* out_dev) {
  FLEET_ENTER(h);
  if (!h || !comm || world_size < 1 || !out_dev) {
    if (h) h->error = "fleet_gather_episode_stats_rccl: bad argument";
    return FLEET_ERR_INVALID;
  }
  RcclApi& r = rccl();
  if (!r.why.empty()) { h->error = r.why; return FLEET_ERR_HIP; }
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t E = (size_t)h->d.E;
  // [2, E] float64 in the handle's staging block: returns, then lengths (exact in float64)
  double* send = static_cast<double*>(h->st_field);
  HIP_TRY(h, fleet_launch_gather_field(h->d, FLEET_F_LAST_EP_RETURN, send, h->stream));
  HIP_TRY(h, fleet_launch_gather_field(h->d, FLEET_F_LAST_EP_LEN_F64, send + E, h->stream));
  const ncclResult_t rc = r.AllGather(send, out_dev, 2 * E, ncclDouble, static_cast<ncclComm_t>(comm), h->stream);
  if (rc != ncclSuccess) {
    h->error = std::string("ncclAllGather: ") + (r.GetErrorString ? r.GetErrorString(rc) : "failed");
    return FLEET_ERR_HIP;
  }
  return FLEET_OK;
}

int fleet_log_capacity(fleet_handle h) { return (h && h->d.log_pos) ? h->d.log_cap : 0; }

int fleet_log_dropped(fleet_handle h, int64_t* rows) {
  FLEET_ENTER(h);
  if (!h || !rows || !h->d.log_pos) {
    if (h) h->error = "fleet_log_dropped: the data log is off or a null pointer";
    return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  std::vector<int32_t> pos((size_t)h->d.E);
  HIP_TRY(h, hipMemcpyAsync(pos.data(), h->d.log_pos, pos.size() * 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  int64_t n = 0;
  for (int32_t p : pos) n += p > h->d.log_cap ? p - h->d.log_cap : 0;
  *rows = n;
  return FLEET_OK;
}

int fleet_log_read(fleet_handle h, int32_t* pos, int32_t* row, double* env, double* ev, float* obs) {
  FLEET_ENTER(h);
  if (!h || !h->d.log_pos) {
    if (h) h->error = "fleet_log_read: the data log is off (FleetParams.log_data = 0)";
    return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  const FleetDev& d = h->d;
  const size_t rows = (size_t)d.log_cap * d.E;
  if (pos) HIP_TRY(h, hipMemcpyAsync(pos, d.log_pos, (size_t)d.E * 4, hipMemcpyDeviceToHost, h->stream));
  if (row) HIP_TRY(h, hipMemcpyAsync(row, d.log_row, rows * 4, hipMemcpyDeviceToHost, h->stream));
  if (env) HIP_TRY(h, hipMemcpyAsync(env, d.log_env, rows * 4 * 8, hipMemcpyDeviceToHost, h->stream));
  if (ev) HIP_TRY(h, hipMemcpyAsync(ev, d.log_ev, rows * 4 * d.N * 8, hipMemcpyDeviceToHost, h->stream));
  if (obs) HIP_TRY(h, hipMemcpyAsync(obs, d.log_obs, rows * (size_t)d.obs_dim * 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FLEET_OK;
}

int fleet_log_clear(fleet_handle h) {
  FLEET_ENTER(h);
  if (!h || !h->d.log_pos) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemsetAsync(h->d.log_pos, 0, (size_t)h->d.E * 4, h->stream));
  return FLEET_OK;
}

int fleet_get_stream(fleet_handle h, void** hip_stream) {
  FLEET_ENTER(h);
  if (!h || !hip_stream) return FLEET_ERR_INVALID;
  *hip_stream = static_cast<void*>(h->stream);
  return FLEET_OK;
}

int fleet_check_errors(fleet_handle h) {
  FLEET_ENTER(h);
  if (!h) return FLEET_ERR_INVALID;
  std::vector<uint32_t> e(h->d.E);
  int rc = fleet_get(h, FLEET_F_ERROR_BITS, e.data());
  if (rc) return rc;
  for (int i = 0; i < h->d.E; ++i)
    if (e[i]) {
      std::vector<int32_t> t(h->d.E);
      (void)fleet_get(h, FLEET_F_TIME_IDX, t.data());
      char names[400], buf[640];
      snprintf(buf, sizeof buf, "device error bits 0x%x on env %d at table row %d of %d:%s (FLEET_DEVERR_* in fleet_hip.h)", e[i], i, t[i],
               h->d.T, deverr_names(e[i], names, sizeof names));
      h->error = buf;
      return FLEET_ERR_STATE;
    }
  return FLEET_OK;
}

int fleet_timer_start(fleet_handle h) {
  FLEET_ENTER(h);
  if (!h) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipEventRecord(h->ev_start, h->stream));
  return FLEET_OK;
}

int fleet_timer_stop(fleet_handle h, float* elapsed_ms) {
  FLEET_ENTER(h);
  if (!h || !elapsed_ms) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipEventRecord(h->ev_stop, h->stream));
  HIP_TRY(h, hipEventSynchronize(h->ev_stop));
  HIP_TRY(h, hipEventElapsedTime(elapsed_ms, h->ev_start, h->ev_stop));
  return FLEET_OK;
}

int fleet_timer_mark(fleet_handle h) {
  FLEET_ENTER(h);
  if (!h) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipEventRecord(h->ev_stop, h->stream));
  return FLEET_OK;
}

int fleet_timer_read(fleet_handle h, float* elapsed_ms) {
  FLEET_ENTER(h);
  if (!h || !elapsed_ms) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipEventSynchronize(h->ev_stop));
  HIP_TRY(h, hipEventElapsedTime(elapsed_ms, h->ev_start, h->ev_stop));
  return FLEET_OK;
}

// What every submission to the library's own queues has in common: the queue exists (opened at the first use: code object, source
// hash, placement probe), the launch's argument blocks describe exactly these buffers on the handle as it is now, and everything the
// HIP stream was given before has completed.  `tape`: tape_len rows of actions (a closed-loop step: one row).
static int direct_ready(fleet_handle h, const void* tape, int tape_len, int act_dtype, float* obs, double* reward, uint8_t* done,
                        float* terminal_obs, int mode) {
  if (h->stream != h->own_stream) {
    // a run is not ordered on a HIP stream: ops queued on a borrowed stream (torch's) after the call would read its outputs too early
    // and nothing could tell them.  The handle's own stream is never handed to anybody else's ops.
    h->error = "launches through the library's own queue are not ordered on a borrowed HIP stream: fleet_use_own_stream first "
               "(the call itself waits for the stream's earlier work; fleet_wait_step / fleet_synchronize order what follows)";
    return FLEET_ERR_INVALID;
  }
  const bool stale = !h->direct || h->dq_gen != h->gen || h->dq_tape != tape || h->dq_len != tape_len || h->dq_dtype != act_dtype ||
                     h->dq_obs != obs || h->dq_term != terminal_obs || h->dq_reward != reward || h->dq_done != done || h->dq_mode != mode;
  if (stale) {
    // (the launches in flight read the argument blocks that are about to be replaced)
    int rc = h->direct ? fleet_direct_wait(h->direct, &h->dq_spans_us, &h->error) : FLEET_OK;
    if (rc != FLEET_OK) return rc;
    if (!h->direct) {
      rc = fleet_direct_open(h->device, &h->direct, &h->error);
      if (rc != FLEET_OK) return rc;
    }
    FleetStepLaunch L;
    const hipError_t e = fleet_describe_step(h->d, tape, act_dtype, obs, reward, done, terminal_obs, &L);
    if (e != hipSuccess) {
      h->error = "direct submission serves single-step launches only (no real_time, no data log)";
      return FLEET_ERR_INVALID;
    }
    // a batch of more wavefronts than are resident at once (256 CUs x 4 SIMDs x 5 of this kernel = 5120) runs as two ranges of
    // workgroups on two queues; one wavefront per env or less only (the wider groups were not measured to gain)
#ifndef FLEET_DIRECT_SPLIT_WAVES
#define FLEET_DIRECT_SPLIT_WAVES 6144
#endif
    const bool split = mode == FLEET_LAUNCH_DIRECT && h->d.N <= 64 && (size_t)L.grid * (L.block / 64) >= FLEET_DIRECT_SPLIT_WAVES;
    h->dq_tape = nullptr;  // whatever happens below, the old key describes nothing any more
    h->dq_len = 0;
    const size_t row = (size_t)h->d.E * h->d.N * (act_dtype == FLEET_ACT_F64 ? 8 : 4);
    rc = fleet_direct_prepare(h->direct, L, tape, tape_len, row, split, &h->error);
    if (rc != FLEET_OK) return rc;
    h->dq_mode = mode; h->dq_gen = h->gen;
    h->dq_tape = tape; h->dq_len = tape_len; h->dq_dtype = act_dtype; h->dq_obs = obs; h->dq_term = terminal_obs; h->dq_reward = reward; h->dq_done = done;
  }
  // what the stream was given before (a reset, a copy of actions ...) has completed before the first packet is written
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FLEET_OK;
}

int fleet_run_tape_dev(fleet_handle h, int steps, const void* tape, int tape_len, int act_dtype, float* obs,
                       double* reward, uint8_t* done, int use_graph) {
  if (!h || steps < 0 || !tape || tape_len < 1 || !obs || !reward || !done ||
      (act_dtype != FLEET_ACT_F32 && act_dtype != FLEET_ACT_F64)) {
    if (h) h->error = "fleet_run_tape_dev: bad argument";
    return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t row = (size_t)h->d.E * h->d.N * (act_dtype == FLEET_ACT_F64 ? 8 : 4);
  const char* base = static_cast<const char*>(tape);
  if (use_graph == FLEET_LAUNCH_DIRECT || use_graph == FLEET_LAUNCH_DIRECT_ONE_QUEUE) {
    // the library's own AQL packets: the launches of the run keep their state in the dies' L2s (fleet_direct.hip), the last one
    // writes it back.  Asynchronous like the other forms; not on the HIP stream -- the next call on the handle waits for the run.
    if (steps == 0) return FLEET_OK;
    const int rc = direct_ready(h, tape, tape_len, act_dtype, obs, reward, done, nullptr, use_graph);
    if (rc != FLEET_OK) return rc;
    return fleet_direct_submit(h->direct, steps, h->dq_timed, &h->error);
  }
  FLEET_ENTER(h);
  int i = 0;
  // the captured graph holds a whole number of tape cycles and at least 64 launches, however short the tape (a short tape must not
  // turn the replay into many short graphs: every hipGraphLaunch costs the host ~10 us)
  const int glen = tape_len * ((64 + tape_len - 1) / tape_len);
  if (use_graph && steps >= glen) {
    const bool stale = !h->graph_exec || h->graph_tape != tape || h->graph_len != tape_len || h->graph_dtype != act_dtype ||
                       h->graph_obs != obs || h->graph_reward != reward || h->graph_done != done;
    if (stale) {
      drop_graph(h);
      hipGraph_t graph = nullptr;
      // capture is not allowed on the legacy null stream (what torch's default stream is): record on the handle's own
      // stream then; the graph itself is launched on the stream in use
      hipStream_t cap = h->stream ? h->stream : h->own_stream;
      HIP_TRY(h, hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < glen; ++k) {
        hipError_t e = fleet_launch_step(h->d, base + (size_t)(k % tape_len) * row, act_dtype, 1, obs, reward, done, nullptr, nullptr, cap);
        if (e != hipSuccess) {
          (void)hipStreamEndCapture(cap, &graph);
          if (graph) (void)hipGraphDestroy(graph);
          h->error = std::string("capture: ") + hipGetErrorString(e);
          return FLEET_ERR_HIP;
        }
      }
      HIP_TRY(h, hipStreamEndCapture(cap, &graph));
      hipError_t e = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) {
        h->graph_exec = nullptr;
        h->error = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
        return FLEET_ERR_HIP;
      }
      (void)hipGraphUpload(h->graph_exec, h->stream);  // best effort: the first replay does not pay for the upload
      h->graph_tape = tape; h->graph_len = tape_len; h->graph_dtype = act_dtype;
      h->graph_obs = obs; h->graph_reward = reward; h->graph_done = done;
    }
    for (; i + glen <= steps; i += glen) HIP_TRY(h, hipGraphLaunch(h->graph_exec, h->stream));
  }
  for (; i < steps; ++i)
    HIP_TRY(h, fleet_launch_step(h->d, base + (size_t)(i % tape_len) * row, act_dtype, 1, obs, reward, done, nullptr, nullptr,
                                 h->stream));
  return FLEET_OK;
}

int fleet_direct_queues(fleet_handle h) {
  if (!h) return FLEET_ERR_INVALID;
  return h->direct ? fleet_direct_parts(h->direct) : 0;
}

int fleet_direct_placement(fleet_handle h, int32_t map8[8], int32_t* num_xcc, int32_t* any_grid) {
  if (!h || !map8) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  if (!h->direct) {
    const int rc = fleet_direct_open(h->device, &h->direct, &h->error);
    if (rc != FLEET_OK) return rc;
  }
  int m[8], nx = 0, ag = 0;
  const int rc = fleet_direct_probed(h->direct, m, &nx, &ag);
  for (int j = 0; j < 8; ++j) map8[j] = m[j];
  if (num_xcc) *num_xcc = nx;
  if (any_grid) *any_grid = ag;
  return rc;
}

int fleet_direct_split_plan(uint32_t grid_workgroups, int split, uint32_t part_grid[2]) {
  if (!part_grid) return FLEET_ERR_INVALID;
  unsigned pg[2];
  const int parts = fleet_direct_plan(grid_workgroups, split != 0, pg);
  part_grid[0] = pg[0];
  part_grid[1] = pg[1];
  return parts;
}

int fleet_debug_direct_fault(fleet_handle h, int kind, int tape_row) {
  if (!h || !h->direct) {
    if (h) h->error = "fleet_debug_direct_fault: no prepared run (run one through the library's own queue first)";
    return FLEET_ERR_INVALID;
  }
  FLEET_ENTER(h);
  return fleet_direct_fault(h->direct, kind, tape_row, &h->error);
}

int fleet_time_regions_begin(fleet_handle h, int regions, int steps, const void* tape, int tape_len, int act_dtype, float* obs,
                              double* reward, uint8_t* done, int use_graph) {
  if (!h || regions < 1 || regions > 256) return FLEET_ERR_INVALID;
  FLEET_ENTER(h);
  HIP_TRY(h, hipSetDevice(h->device));
  if (use_graph == FLEET_LAUNCH_DIRECT || use_graph == FLEET_LAUNCH_DIRECT_ONE_QUEUE) {  // the runs' own dispatch timestamps: start of the first launch -> end of the last
    h->dq_spans_us.clear();
    h->dq_timed = 1;
    int rc = FLEET_OK;
    for (int r = 0; r < regions && rc == FLEET_OK; ++r) rc = fleet_run_tape_dev(h, steps, tape, tape_len, act_dtype, obs, reward, done, use_graph);
    h->dq_timed = 0;
    return rc;
  }
  for (auto& e : h->region_events)
    if (e) (void)hipEventDestroy(e);
  h->region_events.assign(2 * (size_t)regions, nullptr);
  for (auto& e : h->region_events) HIP_TRY(h, hipEventCreate(&e));
  for (int r = 0; r < regions; ++r) {
    HIP_TRY(h, hipEventRecord(h->region_events[2 * r], h->stream));
    const int rc = fleet_run_tape_dev(h, steps, tape, tape_len, act_dtype, obs, reward, done, use_graph);
    if (rc != FLEET_OK) return rc;
    HIP_TRY(h, hipEventRecord(h->region_events[2 * r + 1], h->stream));
  }
  return FLEET_OK;
}

int fleet_time_regions_read(fleet_handle h, float* region_ms) {
  FLEET_ENTER(h);
  if (h && region_ms && h->region_events.empty() && !h->dq_spans_us.empty()) {  // (FLEET_ENTER has waited for the runs)
    bool ok = true;
    for (size_t r = 0; r < h->dq_spans_us.size(); ++r) {
      region_ms[r] = (float)(h->dq_spans_us[r] * 1e-3);
      ok = ok && h->dq_spans_us[r] >= 0.0;  // -1: a packet without dispatch timestamps (a queue whose profiling could not be enabled)
    }
    h->dq_spans_us.clear();
    if (!ok) {
      h->error = "fleet_time_regions_read: a run carries no dispatch timestamps";
      return FLEET_ERR_HIP;
    }
    return FLEET_OK;
  }
  if (!h || !region_ms || h->region_events.empty()) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipEventSynchronize(h->region_events.back()));
  for (size_t r = 0; r < h->region_events.size() / 2; ++r)
    HIP_TRY(h, hipEventElapsedTime(&region_ms[r], h->region_events[2 * r], h->region_events[2 * r + 1]));
  for (auto& e : h->region_events)
    if (e) (void)hipEventDestroy(e);
  h->region_events.clear();
  return FLEET_OK;
}

int fleet_time_steps_dev(fleet_handle h, int steps, const void* tape, int tape_len, int act_dtype, float* obs,
                         double* reward, uint8_t* done, float* per_launch_ms) {
  FLEET_ENTER(h);
  if (!h || steps < 1 || !tape || tape_len < 1 || !obs || !reward || !done || !per_launch_ms ||
      (act_dtype != FLEET_ACT_F32 && act_dtype != FLEET_ACT_F64)) {
    if (h) h->error = "fleet_time_steps_dev: bad argument";
    return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t row = (size_t)h->d.E * h->d.N * (act_dtype == FLEET_ACT_F64 ? 8 : 4);
  const char* base = static_cast<const char*>(tape);
  std::vector<hipEvent_t> ev(2 * (size_t)steps, nullptr);
  int rc = FLEET_OK;
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) rc = FLEET_ERR_HIP;
  if (rc == FLEET_OK) {
    for (int i = 0; i < steps && rc == FLEET_OK; ++i) {
      if (hipEventRecord(ev[2 * i], h->stream) != hipSuccess) rc = FLEET_ERR_HIP;
      if (fleet_launch_step(h->d, base + (size_t)(i % tape_len) * row, act_dtype, 1, obs, reward, done, nullptr, nullptr,
                            h->stream) != hipSuccess)
        rc = FLEET_ERR_HIP;
      if (hipEventRecord(ev[2 * i + 1], h->stream) != hipSuccess) rc = FLEET_ERR_HIP;
    }
    if (hipStreamSynchronize(h->stream) != hipSuccess) rc = FLEET_ERR_HIP;
    for (int i = 0; i < steps && rc == FLEET_OK; ++i)
      if (hipEventElapsedTime(&per_launch_ms[i], ev[2 * i], ev[2 * i + 1]) != hipSuccess) rc = FLEET_ERR_HIP;
  }
  for (auto& e : ev)
    if (e) (void)hipEventDestroy(e);
  if (rc != FLEET_OK) h->error = "fleet_time_steps_dev: a HIP call failed";
  return rc;
}

}  // extern "C"

int fleet_selftest_stress(int device, uint64_t n_samples, uint64_t seed, double* max_rel_err) {
  if (!max_rel_err || n_samples == 0) return FLEET_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FLEET_ERR_NODEVICE;
  if (device < 0 || device >= ndev) return FLEET_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return FLEET_ERR_HIP;
  unsigned long long* worst = nullptr;
  if (hipMalloc(&worst, sizeof(unsigned long long)) != hipSuccess) return FLEET_ERR_HIP;
  int rc = FLEET_OK;
  unsigned long long host = 0;
  if (hipMemset(worst, 0, sizeof host) != hipSuccess || fleet_launch_selftest_stress(n_samples, seed, worst, nullptr) != hipSuccess ||
      hipMemcpy(&host, worst, sizeof host, hipMemcpyDeviceToHost) != hipSuccess)
    rc = FLEET_ERR_HIP;
  (void)hipFree(worst);
  memcpy(max_rel_err, &host, sizeof host);
  return rc;
}

int fleet_selftest_division(int device, uint64_t n_pairs, uint64_t seed, uint64_t* mismatches) {
  if (!mismatches || n_pairs == 0) return FLEET_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FLEET_ERR_NODEVICE;
  if (device < 0 || device >= ndev) return FLEET_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return FLEET_ERR_HIP;
  unsigned long long* bad = nullptr;
  if (hipMalloc(&bad, 2 * sizeof(unsigned long long)) != hipSuccess) return FLEET_ERR_HIP;
  int rc = FLEET_OK;
  unsigned long long host[2] = {0, 0};
  if (hipMemset(bad, 0, sizeof host) != hipSuccess || fleet_launch_selftest_division(n_pairs, seed, bad, nullptr) != hipSuccess ||
      hipMemcpy(host, bad, sizeof host, hipMemcpyDeviceToHost) != hipSuccess)
    rc = FLEET_ERR_HIP;
  (void)hipFree(bad);
  mismatches[0] = host[0];
  mismatches[1] = host[1];
  return rc;
}
