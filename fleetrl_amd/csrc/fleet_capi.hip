// fleet_capi.hip -- host side of the C ABI declared in include/fleet_hip.h (libfleet_hip.so).
//
// Owns: device copies of the pre-staged tables (re-packed into the rows the kernels read), the SoA state of the
// env batch, one HIP stream, staging buffers for the host-pointer entry points, a cached hipGraph for tape replays.
// There is no CPU path in this library: without a HIP device fleet_create fails with FLEET_ERR_NODEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fleet_device.h"

namespace {

thread_local std::string g_create_error;

struct Batch {
  FleetParams p{};
  FleetDev d{};
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string error;
  std::vector<void*> allocs;
  // staging for the *_host entry points
  void* st_actions = nullptr;
  float* st_obs = nullptr;
  float* st_term = nullptr;
  double* st_reward = nullptr;
  uint8_t* st_done = nullptr;
  uint8_t* st_mask = nullptr;
  double* st_dist = nullptr;
  int32_t* dev_sched = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  // cached tape graph
  hipGraphExec_t graph_exec = nullptr;
  const void* graph_tape = nullptr;
  int graph_len = 0, graph_dtype = 0;
  float* graph_obs = nullptr;
  double* graph_reward = nullptr;
  uint8_t* graph_done = nullptr;
};

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
  if (p->episode_steps < 1 || p->steps_per_hour < 1) return "episode_steps/steps_per_hour out of range";
  if (p->price_lookahead < 0 || p->bl_pv_lookahead < 0) return "negative look-ahead";
  if (p->deg_mode < FLEET_DEG_NONE || p->deg_mode > FLEET_DEG_RAINFLOW) return "unknown deg_mode";
  if (p->deg_mode == FLEET_DEG_RAINFLOW && p->init_soh != 1.0)
    return "rainflow/SEI degradation needs init_soh == 1.0 (the reference's used-battery branch is ill-defined, quirk Q4)";
  if (p->normalize && p->include_pv && !p->include_building)
    return "normalize with pv but without building load crashes in the reference (quirk Q4); unsupported";
  if (p->start_lo < 0 || p->start_hi < p->start_lo || p->start_hi > p->table_rows - 1) return "start range outside the table";
  if (!t->there || !t->time_left || !t->soc_on_return || !t->delu || !t->tariff || !t->prc || !t->trc || !t->load ||
      !t->pv || !t->hour || !t->minute || !t->month || !t->weekday)
    return "a required table pointer is null";
  return nullptr;
}

// hourly look-ahead row: `resample("H").first()` of the slice starting at t (observer_bl_pv.py:50-80):
// bucket 0 = row t, bucket k>=1 = first row of clock hour floor_hour(t)+k.
inline int lookahead_row(const FleetParams& p, int t, int k) {
  if (k == 0) return t;
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
      double v = (t.delu[lookahead_row(p, r, k)] + p.fixed_markup) * p.variable_multiplier;
      if (norm) v = (v - p.min_price) / (p.max_price - p.min_price);
      o[k0++] = (float)v;
    }
    for (int k = 0; k <= L; ++k) {
      double v = t.tariff[lookahead_row(p, r, k)] * (1 - p.feed_in_deduction);
      if (norm) v = (v - p.min_tariff) / (p.max_tariff - p.min_tariff);
      o[k0++] = (float)v;
    }
    double load0 = 0.0, pv0 = 0.0;
    if (p.include_building) {
      load0 = t.load[r];
      for (int k = 0; k <= B; ++k) {
        double v = t.load[lookahead_row(p, r, k)];
        if (norm) v = v / p.max_building;
        o[k0++] = (float)v;
      }
    }
    if (p.include_pv) {
      pv0 = t.pv[r];
      for (int k = 0; k <= B; ++k) {
        double v = t.pv[lookahead_row(p, r, k)];
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
  const int T = p.table_rows;
  phys.resize(T);
  flags.resize(T);
  const double spot_offset = p.fixed_markup / 1000;  // ev_charger.py:34
  for (int r = 0; r < T; ++r) {
    PhysRow& q = phys[r];
    q.spot_plus_offset = t.delu[r] / 1000.0 + spot_offset;        // (current_spot + self.spot_offset) :145,149
    q.tariff = t.tariff[r];
    q.k_charge = -1 * p.price_multiplier * t.prc[r] / 1000;       // :154-155
    q.k_discharge = -1 * p.price_multiplier * t.trc[r] / 1000;    // :204-205
    q.load = p.include_building ? t.load[r] : 0.0;
    q.pv = p.include_pv ? t.pv[r] : 0.0;
    q.pv_energy = p.include_pv ? t.pv[r] * p.dt : 0.0;            // :134-136
    q.reserved = 0.0;
    uint8_t f = 0;
    if (t.hour[r] == 14 && t.minute[r] == 45) f |= FLEET_TFLAG_DEG;
    if (t.hour[r] > 11 && t.hour[r] < 15) f |= FLEET_TFLAG_LUNCH;
    flags[r] = f;
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
  HIP_TRY(b, hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  b->own_stream = true;
  HIP_TRY(b, hipEventCreate(&b->ev_start));
  HIP_TRY(b, hipEventCreate(&b->ev_stop));

  FleetDev& d = b->d;
  const int E = p->num_envs, N = p->num_cars, T = p->table_rows, L = p->price_lookahead, B = p->bl_pv_lookahead;
  d.E = E; d.N = N; d.T = T;
  d.obs_dim = obs_dim_of(p);
  d.episode_steps = p->episode_steps;
  d.hist_cap = p->episode_steps + 2;
  d.tail_a_len = 2 * (L + 1) + (p->include_building ? B + 1 : 0) + (p->include_pv ? B + 1 : 0);
  d.tail_b_len = p->aux ? (1 + (p->include_building ? 3 : 0) + 6) : 0;
  d.tail_stride = ((d.tail_a_len + d.tail_b_len + 3) / 4) * 4;
  d.aux = p->aux; d.normalize = p->normalize; d.is_caretaker = p->is_caretaker; d.deg_mode = p->deg_mode;
  d.auto_reset = p->auto_reset; d.picker_mode = p->picker_mode; d.start_lo = p->start_lo; d.start_hi = p->start_hi;
  d.env_id_offset = p->env_id_offset; d.sched_n = 0; d.seed = p->seed;
  d.dt = p->dt; d.evse_power = p->evse_power;
  d.p_avail = p->obc_max_power < p->evse_power ? p->obc_max_power : p->evse_power;  // min([obc, evse]) ev_charger.py:95
  d.batt_cap_nominal = p->batt_cap_nominal; d.init_cap = p->init_battery_cap; d.grid_connection = p->grid_connection;
  d.eta_c = p->charging_eff; d.eta_d = p->discharging_eff; d.variable_multiplier = p->variable_multiplier;
  d.one_minus_fee = 1 - p->feed_in_deduction;
  d.penalty_invalid = p->penalty_invalid_action; d.penalty_oc = p->penalty_overcharging; d.clip_oc = p->clip_overcharging;
  d.penalty_overload = p->penalty_overloading; d.fully_charged_reward = p->fully_charged_reward;
  d.target_soc = p->target_soc; d.target_soc_lunch = p->target_soc_lunch; d.eps = p->eps; d.def_soc = p->def_soc;
  d.min_laxity = p->min_laxity; d.init_soh = p->init_soh; d.temperature = p->temperature;
  d.hn_denominator = p->evse_power * p->charging_eff;
  d.max_time_left = p->max_time_left; d.max_soc = p->max_soc; d.max_hours_needed = p->max_hours_needed;
  d.max_laxity = p->max_laxity;

  // ---- tables ---------------------------------------------------------------------------------------
  int rc;
  const size_t TN = (size_t)T * N;
  if ((rc = dev_upload(b, &d.tab_there, t->there, TN))) return rc;
  if ((rc = dev_upload(b, &d.tab_tl, t->time_left, TN))) return rc;
  if ((rc = dev_upload(b, &d.tab_sor, t->soc_on_return, TN))) return rc;
  std::vector<PhysRow> phys;
  std::vector<uint8_t> flags;
  build_phys_rows(*p, *t, phys, flags);
  std::vector<float> tail;
  build_tail_rows(*p, *t, d.tail_a_len, d.tail_b_len, d.tail_stride, tail);
  if ((rc = dev_upload(b, &d.tab_phys, phys.data(), phys.size()))) return rc;
  if ((rc = dev_upload(b, &d.tab_flags, flags.data(), flags.size()))) return rc;
  if ((rc = dev_upload(b, &d.tab_tail, tail.data(), tail.size()))) return rc;
  HIP_TRY(b, hipStreamSynchronize(b->stream));  // host vectors go out of scope below

  // ---- state ----------------------------------------------------------------------------------------
  const size_t EN = (size_t)E * N;
  if ((rc = dev_alloc(b, &d.soc, EN))) return rc;
  if ((rc = dev_alloc(b, &d.hl, EN))) return rc;
  if ((rc = dev_alloc(b, &d.soc_deg, EN))) return rc;
  if ((rc = dev_alloc(b, &d.soh, EN))) return rc;
  if ((rc = dev_alloc(b, &d.tgt090, EN))) return rc;
  if ((rc = dev_alloc(b, &d.rf_len, EN))) return rc;
  if ((rc = dev_alloc(b, &d.fd_cyc, EN))) return rc;
  if ((rc = dev_alloc(b, &d.fd_cal, EN))) return rc;
  if ((rc = dev_alloc(b, &d.sei_l, EN))) return rc;
  if ((rc = dev_alloc(b, &d.sei_soh, EN))) return rc;
  if (p->deg_mode != FLEET_DEG_NONE) {
    if ((rc = dev_alloc(b, &d.hist, EN * (size_t)d.hist_cap, false))) return rc;
  }
  if (p->deg_mode == FLEET_DEG_RAINFLOW) {
    if ((rc = dev_alloc(b, &d.rf_stack, EN * (size_t)(d.hist_cap + 1), false))) return rc;
  }
  if ((rc = dev_alloc(b, &d.t_idx, E))) return rc;
  if ((rc = dev_alloc(b, &d.t_end, E))) return rc;
  if ((rc = dev_alloc(b, &d.start_idx, E))) return rc;
  if ((rc = dev_alloc(b, &d.hist_len, E))) return rc;
  if ((rc = dev_alloc(b, &d.episodes, E))) return rc;
  if ((rc = dev_alloc(b, &d.ep_len, E))) return rc;
  if ((rc = dev_alloc(b, &d.last_ep_len, E))) return rc;
  if ((rc = dev_alloc(b, &d.ep_return, E))) return rc;
  if ((rc = dev_alloc(b, &d.last_ep_return, E))) return rc;
  if ((rc = dev_alloc(b, &d.cashflow, E))) return rc;
  if ((rc = dev_alloc(b, &d.penalty_record, E))) return rc;
  if ((rc = dev_alloc(b, &d.err, E))) return rc;
  if ((rc = dev_alloc(b, &d.done_flag, E))) return rc;
  {
    // persistent degradation state (RainflowSeiDegradation.__init__, rainflow_sei_degradation.py:24-66) and the
    // initial SoH / target flags (fleet_environment.py:263)
    std::vector<double> v(EN, p->init_soh), l(EN, 1.0 - p->init_soh);
    std::vector<int32_t> one(EN, 1);
    std::vector<uint8_t> t9(EN, 0);
    HIP_TRY(b, hipMemcpyAsync(d.soh, v.data(), EN * 8, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(b, hipMemcpyAsync(d.sei_soh, v.data(), EN * 8, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(b, hipMemcpyAsync(d.sei_l, l.data(), EN * 8, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(b, hipMemcpyAsync(d.rf_len, one.data(), EN * 4, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(b, hipMemcpyAsync(d.tgt090, t9.data(), EN, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
  }
  // ---- staging for host entry points -------------------------------------------------------------------------------
  const size_t OD = (size_t)E * d.obs_dim;
  if ((rc = dev_alloc(b, (char**)&b->st_actions, EN * 8))) return rc;
  if ((rc = dev_alloc(b, &b->st_obs, OD))) return rc;
  if ((rc = dev_alloc(b, &b->st_term, OD))) return rc;
  if ((rc = dev_alloc(b, &b->st_reward, E))) return rc;
  if ((rc = dev_alloc(b, &b->st_done, E))) return rc;
  if ((rc = dev_alloc(b, &b->st_mask, E))) return rc;
  if ((rc = dev_alloc(b, &b->st_dist, EN))) return rc;
  HIP_TRY(b, hipStreamSynchronize(b->stream));
  return FLEET_OK;
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
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  drop_graph(h);
  for (void* ptr : h->allocs) (void)hipFree(ptr);
  if (h->dev_sched) (void)hipFree(h->dev_sched);
  if (h->ev_start) (void)hipEventDestroy(h->ev_start);
  if (h->ev_stop) (void)hipEventDestroy(h->ev_stop);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return FLEET_OK;
}

const char* fleet_last_error(fleet_handle h) { return h ? h->error.c_str() : g_create_error.c_str(); }

int fleet_set_stream(fleet_handle h, void* hip_stream) {
  if (!h) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  drop_graph(h);
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  h->stream = static_cast<hipStream_t>(hip_stream);
  h->own_stream = false;
  return FLEET_OK;
}

int fleet_synchronize(fleet_handle h) {
  if (!h) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FLEET_OK;
}

int fleet_set_start_schedule(fleet_handle h, const int32_t* starts, int n_episodes) {
  if (!h || n_episodes < 0 || (n_episodes > 0 && !starts)) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  drop_graph(h);
  if (h->dev_sched) {
    (void)hipFree(h->dev_sched);
    h->dev_sched = nullptr;
  }
  h->d.sched = nullptr;
  h->d.sched_n = 0;
  if (n_episodes > 0) {
    const size_t n = (size_t)n_episodes * h->d.E;
    for (size_t i = 0; i < n; ++i)
      if (starts[i] < 0 || starts[i] > h->d.T - 1) {
        h->error = "start row outside the table";
        return FLEET_ERR_INVALID;
      }
    HIP_TRY(h, hipMalloc((void**)&h->dev_sched, n * sizeof(int32_t)));
    HIP_TRY(h, hipMemcpy(h->dev_sched, starts, n * sizeof(int32_t), hipMemcpyHostToDevice));
    h->d.sched = h->dev_sched;
    h->d.sched_n = n_episodes;
  }
  return FLEET_OK;
}

int fleet_reset_dev(fleet_handle h, const uint8_t* mask, float* obs) {
  if (!h || !obs) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, fleet_launch_reset(h->d, mask, obs, h->stream));
  return FLEET_OK;
}

int fleet_step_dev(fleet_handle h, const void* actions, int act_dtype, float* obs, double* reward, uint8_t* done,
                   float* terminal_obs) {
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
  if (!h || K < 1 || !actions || !obs || !reward_sum || (act_dtype != FLEET_ACT_F32 && act_dtype != FLEET_ACT_F64)) {
    if (h) h->error = "fleet_step_many_dev: bad argument";
    return FLEET_ERR_INVALID;
  }
  if (!h->d.auto_reset) {
    h->error = "fleet_step_many_dev needs auto_reset = 1";
    return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  if (K == 1) {
    // K == 1 writes per-step reward/done; keep the many-step contract (sum / count) by using the staging done buffer
    HIP_TRY(h, fleet_launch_step(h->d, actions, act_dtype, 1, obs, reward_sum, h->st_done, nullptr, nullptr, h->stream));
    if (done_count) {
      h->error = "fleet_step_many_dev: done_count needs K >= 2";
      return FLEET_ERR_INVALID;
    }
    return FLEET_OK;
  }
  HIP_TRY(h, fleet_launch_step(h->d, actions, act_dtype, K, obs, reward_sum, h->st_done, nullptr, done_count, h->stream));
  return FLEET_OK;
}

int fleet_reset_host(fleet_handle h, const uint8_t* mask, float* obs) {
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
  if (!h || !actions || !obs || !reward || !done || (act_dtype != FLEET_ACT_F32 && act_dtype != FLEET_ACT_F64)) {
    if (h) h->error = "fleet_step_host: null buffer or bad action dtype";
    return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t EN = (size_t)h->d.E * h->d.N;
  const size_t OD = (size_t)h->d.E * h->d.obs_dim * sizeof(float);
  HIP_TRY(h, hipMemcpyAsync(h->st_actions, actions, EN * (act_dtype == FLEET_ACT_F64 ? 8 : 4), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, fleet_launch_step(h->d, h->st_actions, act_dtype, 1, h->st_obs, h->st_reward, h->st_done,
                               terminal_obs ? h->st_term : nullptr, nullptr, h->stream));
  HIP_TRY(h, hipMemcpyAsync(obs, h->st_obs, OD, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(reward, h->st_reward, h->d.E * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(done, h->st_done, h->d.E, hipMemcpyDeviceToHost, h->stream));
  if (terminal_obs) HIP_TRY(h, hipMemcpyAsync(terminal_obs, h->st_term, OD, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FLEET_OK;
}

int fleet_get(fleet_handle h, int field, void* out) {
  if (!h || !out) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  const FleetDev& d = h->d;
  const size_t E = d.E, EN = (size_t)d.E * d.N;
  const void* src = nullptr;
  size_t bytes = 0;
  switch (field) {
    case FLEET_F_SOC: src = d.soc; bytes = EN * 8; break;
    case FLEET_F_HOURS_LEFT: src = d.hl; bytes = EN * 4; break;
    case FLEET_F_SOH: src = d.soh; bytes = EN * 8; break;
    case FLEET_F_SOC_DEG: src = d.soc_deg; bytes = EN * 8; break;
    case FLEET_F_TIME_IDX: src = d.t_idx; bytes = E * 4; break;
    case FLEET_F_START_IDX: src = d.start_idx; bytes = E * 4; break;
    case FLEET_F_CASHFLOW: src = d.cashflow; bytes = E * 8; break;
    case FLEET_F_EP_RETURN: src = d.ep_return; bytes = E * 8; break;
    case FLEET_F_EP_LEN: src = d.ep_len; bytes = E * 4; break;
    case FLEET_F_LAST_EP_RETURN: src = d.last_ep_return; bytes = E * 8; break;
    case FLEET_F_LAST_EP_LEN: src = d.last_ep_len; bytes = E * 4; break;
    case FLEET_F_RF_LEN: src = d.rf_len; bytes = EN * 4; break;
    case FLEET_F_FD_CYC: src = d.fd_cyc; bytes = EN * 8; break;
    case FLEET_F_FD_CAL: src = d.fd_cal; bytes = EN * 8; break;
    case FLEET_F_SEI_L: src = d.sei_l; bytes = EN * 8; break;
    case FLEET_F_ERROR_BITS: src = d.err; bytes = E * 4; break;
    case FLEET_F_DONE: src = d.done_flag; bytes = E; break;
    case FLEET_F_EPISODES: src = d.episodes; bytes = E * 4; break;
    case FLEET_F_PENALTY_RECORD: src = d.penalty_record; bytes = E * 8; break;
    case FLEET_F_TARGET_SOC: {
      std::vector<uint8_t> f(EN);
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      HIP_TRY(h, hipMemcpy(f.data(), d.tgt090, EN, hipMemcpyDeviceToHost));
      double* o = static_cast<double*>(out);
      for (size_t i = 0; i < EN; ++i) o[i] = f[i] ? 0.9 : d.target_soc;
      return FLEET_OK;
    }
    default:
      h->error = "fleet_get: unknown field";
      return FLEET_ERR_INVALID;
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
  return FLEET_OK;
}

int fleet_get_dist_factor(fleet_handle h, double* out) {
  if (!h || !out) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, fleet_launch_dist_factor(h->d, h->st_dist, h->stream));
  HIP_TRY(h, hipMemcpyAsync(out, h->st_dist, (size_t)h->d.E * h->d.N * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FLEET_OK;
}

int fleet_check_errors(fleet_handle h) {
  if (!h) return FLEET_ERR_INVALID;
  std::vector<uint32_t> e(h->d.E);
  int rc = fleet_get(h, FLEET_F_ERROR_BITS, e.data());
  if (rc) return rc;
  for (int i = 0; i < h->d.E; ++i)
    if (e[i]) {
      char buf[160];
      snprintf(buf, sizeof buf, "device error bits 0x%x on env %d (see FLEET_DEVERR_* in fleet_hip.h)", e[i], i);
      h->error = buf;
      return FLEET_ERR_STATE;
    }
  return FLEET_OK;
}

int fleet_timer_start(fleet_handle h) {
  if (!h) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipEventRecord(h->ev_start, h->stream));
  return FLEET_OK;
}

int fleet_timer_stop(fleet_handle h, float* elapsed_ms) {
  if (!h || !elapsed_ms) return FLEET_ERR_INVALID;
  HIP_TRY(h, hipEventRecord(h->ev_stop, h->stream));
  HIP_TRY(h, hipEventSynchronize(h->ev_stop));
  HIP_TRY(h, hipEventElapsedTime(elapsed_ms, h->ev_start, h->ev_stop));
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
  int i = 0;
  if (use_graph && steps >= tape_len) {
    const bool stale = !h->graph_exec || h->graph_tape != tape || h->graph_len != tape_len || h->graph_dtype != act_dtype ||
                       h->graph_obs != obs || h->graph_reward != reward || h->graph_done != done;
    if (stale) {
      drop_graph(h);
      hipGraph_t graph = nullptr;
      HIP_TRY(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < tape_len; ++k) {
        hipError_t e = fleet_launch_step(h->d, base + (size_t)k * row, act_dtype, 1, obs, reward, done, nullptr, nullptr, h->stream);
        if (e != hipSuccess) {
          (void)hipStreamEndCapture(h->stream, &graph);
          if (graph) (void)hipGraphDestroy(graph);
          h->error = std::string("capture: ") + hipGetErrorString(e);
          return FLEET_ERR_HIP;
        }
      }
      HIP_TRY(h, hipStreamEndCapture(h->stream, &graph));
      hipError_t e = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) {
        h->graph_exec = nullptr;
        h->error = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
        return FLEET_ERR_HIP;
      }
      h->graph_tape = tape; h->graph_len = tape_len; h->graph_dtype = act_dtype;
      h->graph_obs = obs; h->graph_reward = reward; h->graph_done = done;
    }
    for (; i + tape_len <= steps; i += tape_len) HIP_TRY(h, hipGraphLaunch(h->graph_exec, h->stream));
  }
  for (; i < steps; ++i)
    HIP_TRY(h, fleet_launch_step(h->d, base + (size_t)(i % tape_len) * row, act_dtype, 1, obs, reward, done, nullptr, nullptr,
                                 h->stream));
  return FLEET_OK;
}

}  // extern "C"
