// fleet_device.h -- device-side view of one env batch (kernel argument block) shared by the kernels
// (fleet_kernels.hip) and the host side of the C ABI (fleet_capi.hip).  gfx950 only.
//
// Layout rules (DESIGN.md "Data layout in HBM"):
//   * everything a lane reads AND writes every step sits in one dense 16-byte record per (env, EV): one 16-byte load /
//     store per lane, consecutive lanes = consecutive records, whole cache lines; what changes rarely (soh, the schedule
//     record of the next row, the rainflow stack top) sits in planes of its own that are only written when it changes;
//   * nothing a lane needs to start its arithmetic depends on the env's time row: the schedule columns of the next row
//     travel with the state in run-length form (`run`, struct SegRec), so the per-(row, EV) table is only touched when
//     an EV crosses a schedule event, and then for the row AFTER next -- off the step's critical path;
//   * everything a group needs per env and step sits in ONE 64-byte record (16-byte head + leader statistics);
//   * the env-level observation blocks and the physics scalars of a time row sit in contiguous rows;
//   * rarely touched state (rainflow accumulators + stack, SEI model) sits in per-EV 128-byte-aligned rows / 32-byte
//     records so an event touches one cache line and the hot path none.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fleet_hip.h"

// ---- read-only tables ---------------------------------------------------------------------------------------
// Physics row of time row t (72 B): per-time scalars pre-combined on the host (fleet_capi.hip build_phys_rows()).
// k_charge / k_discharge / pv_share use exactly the reference's float64 operations (bit-identical); k_cost / k_rev
// re-associate two multiplications of the money terms (cashflow differs from the reference by <= 1 ulp per EV).
struct PhysRow {
  double k_cost;        // (DELU[t]/1000.0 + fixed_markup/1000) * variable_multiplier : EUR per kWh drawn (ev_charger.py:145-149)
  double k_rev;         // -1 * discharging_eff * tariff[t] / 1000 * (1 - fee)        : EUR per kWh of (negative) energy (:196-199)
  double k_charge;      // -1*price_multiplier*prc[t]/1000                              (:154-155)
  double k_discharge;   // -1*price_multiplier*trc[t]/1000                              (:204-205)
  double load;          // building load [kW] or 0                                      (fleet_environment.py:480-483)
  double pv;            // pv [kW] or 0                                                 (:485-488)
  double pv_share;      // pv[t]*dt / max(sum(There[t]),1)                              (ev_charger.py:134-142)
  uint32_t flags_next;  // FLEET_TFLAG_* of time row t+1 (the row the step advances to)
  uint32_t pad;
  double dt;            // hours from row t to row t+1: `get_next_dt` (fleet_environment.py:455, :994-1008); the model's
                        // constant step on a regular grid, per row on an irregular one (real_time)
};

// Schedule record of (time row t, EV c), 16 B: the three schedule columns of the row -- db["There"], db["time_left"],
// db["SOC_on_return"] -- in run-length form.  Between two schedule events of an EV (departure, arrival, a change of
// SOC_on_return at an hour boundary of the caretaker's lunch rule) `There` and `SOC_on_return` are constant and `time_left`
// falls by dt per row, so consecutive rows form a SEGMENT and every row of a segment holds the SAME record: the two constant
// columns, the departure row `time_left` counts down to, and the first row after the segment.  A lane that holds the record
// of one row therefore holds the record of every row up to `seg_end` and only has to touch the table when it crosses a
// segment boundary (about four times per EV and day).  Built by fleet_create (fleet_capi.hip build_seg_rows), which checks
// row by row that the float32 `time_left` the table holds is exactly what `seg_tl` derives; a row where it is not (irregular
// time grids, hand-made tables) becomes a one-row segment that carries its `time_left` verbatim (SEG_RAW).
struct SegRec {
  double sor;    // db["SOC_on_return"] of every row of the segment
  uint32_t tlx;  // departure row `dep`: time_left(r) = (dep - r) * dt for r < dep, else 0; SEG_RAW: the float32 time_left itself
  uint32_t se;   // [29:0] first row after the segment, [30] SEG_RAW, [31] db["There"]
};
#define SEG_RAW 0x40000000u
#define SEG_END(se) ((int)((se) & 0x3FFFFFFFu))
#define SEG_THERE(se) ((se) >> 31)
// db["time_left"] of row r from the record of r's segment (exact float32: checked row by row when the table was built)
__host__ __device__ inline float seg_tl(const SegRec& s, int r, double dt) {
  union { uint32_t u; float f; } v;
  v.u = s.tlx;
  const int left = (int)s.tlx - r;
  const float reg = left > 0 ? (float)((double)left * dt) : 0.0f;
  return (s.se & SEG_RAW) ? v.f : reg;  // (a select: the RAW case is a table property, the lanes of a wavefront may differ)
}

#define FLEET_TFLAG_DEG 1u    // hour == 14 && minute == 45   (fleet_environment.py:665)
#define FLEET_TFLAG_LUNCH 2u  // 11 < hour < 15               (:538)

// ---- state ---------------------------------------------------------------------------------------------------
// Hot state of (env e, EV c): ONE dense 16-byte record read and written every step (consecutive lanes = consecutive
// records = whole cache lines).  episode.soc and episode.soc_deg (the last logged SOC sample) share its float64 field:
//   * while the EV is plugged in they are equal after every step (fleet_environment.py:621-623): x = soc = soc_deg;
//   * while it is away (hours_left == 0) soc_deg keeps its last value and soc is the table's SOC_on_return of an empty
//     slot, i.e. +0.0: FROZEN, x = soc_deg and soc = 0.0 is implied;
//   * the one combination that does not fit -- away-like state with a non-zero soc (the rows after an EV's last departure
//     of the table, hand-made tables) or a reset that starts from soc == -0.0 -- keeps x = soc and puts soc_deg into the
//     soc_deg plane: FROZEN | INPLANE (read through a dependent load; it does not occur inside the reference's episodes).
// Everything else that changes rarely has planes of its own that are only written when it changes: `soh` (daily), the
// schedule record of the next row `run` (at schedule events), the rainflow stack top `rf_top` (when a reversal is pushed).
struct Hot {
  double x;        // episode.soc and / or episode.soc_deg, see above
  float hl;        // episode.hours_left (multiple of dt, exact in f32)
  uint32_t bits;   // [25:0] rainflow stack size (the stack always starts at slot 0), [26] INPLANE, [28:27] sign of the last
                   // SOC slope (0 none, 1 up, 2 down), [29] FROZEN, [30] There at the current time row (carried so the
                   // step needs no table read for it), [31] sticky "target_soc = 0.9" flag (quirk Q7)
};
#define HOT_TAIL(b) ((int)((b) & 0x3FFFFFFu))
#define HOT_INPLANE(b) ((((b) >> 26) & 1u) != 0u)
#define HOT_SGN(b) ((int)(((b) >> 27) & 3u))
#define HOT_FROZEN(b) ((((b) >> 29) & 1u) != 0u)
#define HOT_THERE(b) (((b) >> 30) & 1u)
#define HOT_T090(b) (((b) >> 31) != 0u)
#define FLEET_MAX_STACK_ROWS 0x3FFFFFF  // 26-bit stack size: 67 million samples per episode
#define HOT_PACK(tail, sgn, frozen, inplane, there, t090)                                                                 \
  (((uint32_t)(tail) & 0x3FFFFFFu) | ((inplane) ? 0x4000000u : 0u) | (((uint32_t)(sgn) & 3u) << 27) |                    \
   ((frozen) ? 0x20000000u : 0u) | (((uint32_t)(there) & 1u) << 30) | ((t090) ? 0x80000000u : 0u))
// episode.soc / episode.soc_deg of a hot record (`plane` = the EV's soc_deg plane entry, only read when INPLANE)
#define HOT_SOC(h) ((HOT_FROZEN((h).bits) && !HOT_INPLANE((h).bits)) ? 0.0 : (h).x)

// What the step arithmetic needs of a physics row: the first 64 bytes of PhysRow.
struct PhysHot {
  double k_cost, k_rev, k_charge, k_discharge, load, pv, pv_share;
  uint32_t flags_next, pad;
};
static_assert(sizeof(PhysHot) == 64 && sizeof(PhysRow) == 72, "PhysHot is the head of PhysRow");

// Env record, 64 B: the 16-byte head every lane of the group needs (wave-uniform for G == 64), followed by the episode
// statistics only the group's leader lane touches.  One pointer, one line per env and step.
// The head also carries the FLEET_TFLAG_* bits of the row AFTER the current one (`flags_next` of the current row's physics
// record), left there by the launch that advanced to the current row: the state machine needs them (lunch target, daily
// degradation row) before the money terms need the physics record itself, and with them in the head nothing the step
// needs early depends on the time row -- the physics record is requested when the head arrives and consumed last.
struct EnvHead {
  int32_t t;         // current table row (episode.time)
  int32_t t_end;     // finish row (episode.finish_time)
  int32_t nsamp;     // [28:0] len(LogDataDeg.soc_log), [29] the sample the NEXT step logs is still counted (t < EnvRec::rf_until),
                     // [31:30] FLEET_TFLAG_* of row t + 1
  int32_t episodes;  // finished (or abandoned) episodes: start-schedule index / Philox counter
};
#define HEAD_NSAMP(x) ((int32_t)((uint32_t)(x) & 0x1FFFFFFFu))
#define HEAD_LIVE(x) ((((uint32_t)(x) >> 29) & 1u) != 0u)
#define HEAD_FLAGS(x) ((uint32_t)(x) >> 30)
#define HEAD_PACK(nsamp, flags, live) \
  ((int32_t)(((uint32_t)(nsamp) & 0x1FFFFFFFu) | ((live) ? 0x20000000u : 0u) | ((uint32_t)(flags) << 30)))
#define FLEET_MAX_EPISODE_STEPS 0x1FFFFFFF  // 29-bit sample count
struct EnvRec {
  EnvHead h;
  int32_t ep_len;          // steps taken in the running episode
  int32_t rf_until;        // the last row of the running episode on which the degradation model is evaluated (14:45 rows,
                           // fleet_environment.py:665), -1 if there is none: SOC samples logged AFTER it are never counted -- the
                           // reference's reset() clears the log (:338-339) before anybody reads them -- so the streaming count
                           // stops there (a quarter of all EV-steps with 48 h episodes).  INT32_MAX: count everything
                           // (fleet_set_rainflow_count_all, diagnostics).  Set by reset; head bit 29 caches `t < rf_until`.
  uint32_t err;            // FLEET_DEVERR_* bits
  int32_t start_done;      // [30:0] row the running episode started on (episode.start_time), [31] episode.done
  double ep_return;        // episode.cumulative_reward
  double last_ep_return;   // return of the last finished episode
  double cashflow;         // episode.current_charging_expense (last step)
  double penalty_record;   // episode.penalty_record
};
static_assert(sizeof(EnvRec) == 64, "one 64-byte record per env");

// Rainflow row of (env e, EV c): a 48-byte header followed by the reversal stack, 128-byte aligned, so that everything a
// push touches -- accumulators, the two newest stack entries, the entries right below them -- sits in ONE cache line for
// the usual stack depths.  Nothing of it is read by a step that pushes no reversal point (three steps in four): whether
// a step pushes is decided from the hot record alone (sign of the last slope), and only then is the row requested.
//   * `s2` is the ONLY copy of the newest stack entry: the stack words hold the entries below it (stack[0 .. tail-2];
//     `s1` caches the last of them).  A push that closes no cycle therefore writes one stack word (the displaced old
//     top), and a push that closes a full cycle writes none (the two popped points vanish, the new point stays in s2).
//   * what every closure reads and writes sits in the first 16 bytes, the stack top in the next 16.
struct RfHdr {
  double mean_sum;  // sum of cycle means over the closed cycles of this episode
  int32_t nc;       // closed cycles this episode
  int32_t rf_len;   // RainflowSeiDegradation.rainflow_length (persists across episodes, quirk Q6)
  double s1;        // stack[tail-2]
  double s2;        // stack[tail-1]
  double csum;      // stress sum of the closed cycles with index >= rainflow_length-1 (rarely non-zero, see rf_finish)
  double pad;
};
struct RfAccHead {  // bytes 0..15 of RfHdr
  double mean_sum;
  int32_t nc;
  int32_t rf_len;
};
struct RfTop {  // bytes 16..31 of RfHdr
  double s1;
  double s2;
};
#define RF_HDR_WORDS 6  // doubles of the header; the stack follows
static_assert(sizeof(RfHdr) == 8 * RF_HDR_WORDS && sizeof(RfAccHead) == 16 && sizeof(RfTop) == 16, "RfHdr layout");
// SEI model state of (env e, EV c), 32 B, touched on the daily row only (persists across episodes, quirk Q6).
struct SeiRec {
  double fd_cyc;   // RainflowSeiDegradation.fd_cyc
  double fd_cal;   // .fd_cal
  double sei_l;    // .l
  double sei_soh;  // .soh (the model's own copy, only used by its consistency check)
};
// Scalars only the rare paths need (reset, daily degradation, slow observation path).  Lives in device memory and
// is read through a pointer so that the hot path's scalar-register budget is not spent on it.
struct FleetCold {
  double min_laxity, def_soc, init_soh, temperature, dt;
  double batt_cap_nominal, hn_denominator, max_soc, max_hours_needed, max_laxity;
  double inv_max_soc, inv_max_hours_needed, inv_max_laxity;  // oracle_normalization.py:127-131 as multiplications
  unsigned long long seed;
  int picker_mode, start_lo, start_hi, env_id_offset;
  int sched_n;
  int normalize;
  const int32_t* sched;  // [sched_n, E]
  const int32_t* pick_rows;  // candidate start rows of the pickers (start_lo / start_hi index into it), or nullptr
  // FLEET_ACT_POLICY_NIGHT (benchmarking/night_charging.py:50-98)
  const uint16_t* tab_hm;  // [T] hour << 8 | minute of the table row
  int32_t* night_start;    // [E] row at which the env's charging window opened, FLEET_NIGHT_IDLE when closed
  int night_hour, night_minute;  // charging_hour / charging_minute; night_hour < 0: not configured
  int night_limit_s;       // 3600 * int(max_time_needed)
  int step_s;              // seconds per table row
  const int32_t* tab_last_deg;  // [T] the last row <= r that carries FLEET_TFLAG_DEG, -1 before the first one (EnvRec::rf_until)
  int32_t* last_len;       // [E] length of the env's last finished episode
  int rf_count_all;        // diagnostics: keep the rainflow count running to the end of every episode (EnvRec::rf_until)
};
#define FLEET_NIGHT_IDLE INT32_MIN

struct FleetDev {
  // ---- sizes / flags --------------------------------------------------------------------------------
  int E, N, T;
  int obs_dim;
  int episode_steps;
  int stack_cap;    // rows of the rainflow stack workspace (episode_steps + 3)
  int tail_a_len;   // price|tariff|load|pv look-ahead block  (obs offset 2N)
  int tail_b_len;   // evse|grid|avail|pavg|6 time features   (obs offset 2N + tail_a_len + 5N), 0 if !aux
  int tail_stride;  // floats per tail row (tail_a then tail_b, padded to a multiple of 4)
  int aux, normalize, is_caretaker, deg_mode, auto_reset;
  int real_time;    // event-skipping step (multi-step kernel only)
  int carry_run;    // single-step launches read the carried schedule records `run` (one EV per lane: N <= fleet_max_evs_per_lane_group()),
                    // so every kernel that advances the batch leaves them behind -- also the K-step kernels of a batch with N > 64
  // ---- hot scalars (FleetParams) ------------------------------------------------------------------------
  double dt, p_avail, init_cap, eta_c, eta_d, penalty_invalid, penalty_oc, clip_oc, target_soc, target_soc_lunch, eps,
      fully_charged_reward, evse_power, grid_connection, penalty_overload, max_time_left, stress_temp;
  // auxiliary observation slots (observer_*.py:85-91), computed per lane from the carried schedule record:
  // hn_scale = nominal capacity / (evse * eta_c); the normaliser's reciprocals are in FleetCold
  double hn_scale;
  double inv_eta_c;  // 1 / eta_c, correctly rounded (host): `need / eta_c` as div_rcp (fleet_kernels.hip)
  // ---- read-only tables ---------------------------------------------------------------------------------
  const SegRec* seg;          // [T,N] schedule records in run-length form
  const PhysRow* tab_phys;    // [T]
  const uint8_t* tab_flags;   // [T]
  const float* tab_tail;      // [T,tail_stride]
  const int32_t* tab_finish;  // [T] episode-end row of an episode starting on row t (-1: never), or nullptr (t + episode_steps)
  const FleetCold* cold;
  const struct FleetDev* self;  // device-resident copy of this block: the out-of-line rare paths read it from memory
  // ---- state ------------------------------------------------------------------------------------------
  Hot* hot;           // [E,N]
  SegRec* run;        // [E,N] schedule record of row t+1 (the row the next step advances to); rewritten when t+2 crosses a
                      // segment boundary, i.e. at schedule events only
  double* soh;        // [E,N]
  double* soc_deg;    // [E,N] (valid where the INPLANE bit is set)
  SeiRec* sei;        // [E,N]
  EnvRec* env;        // [E]
  uint32_t* err_any;  // one word: OR of every FLEET_DEVERR_* bit any env has raised (lives in the block the host-pointer step copies
                      // back with rewards and dones, so a failing step is reported by that very step at no extra cost); the
                      // kernels read the pointer from the device-resident copy of this block (`self`), on the error path only
  // device-side data log (FleetParams.log_data; utils/data_logger/data_logger.py:21-68): a ring of `log_cap` rows per env,
  // written by the kernels in every mode (single step, K-step, policy rollout, reset); env e's next row goes to slot
  // log_pos[e] % log_cap.  All nullptr / 0 when log_data is off.
  int log_cap;
  int32_t* log_pos;   // [E] rows written so far
  int32_t* log_row;   // [log_cap][E] table row of episode.time; bit 31: the row reset() writes (fleet_environment.py:420-432)
  double* log_env;    // [log_cap][E][4] reward, cashflow, overload_amount, cum_soc_missing (:659-661)
  double* log_ev;     // [log_cap][E][4][N] action, (dis)charging energy (ev_charger.py:114,174), degradation, soh
  float* log_obs;     // [log_cap][E][obs_dim] the (normalised) observation of the row
  double* rf_rows;    // [E*N][rf_row_stride] per-EV rainflow row: RfHdr (6 doubles) followed by the reversal stack, EV-major
                      // and 128-byte aligned so that a push / cycle closure touches ONE cache line (accumulators, stack top
                      // and the entries below it) instead of one line per field / stack level
  int rf_row_stride;  // doubles per row (multiple of 16)
};

// launchers implemented in fleet_kernels.hip
hipError_t fleet_launch_reset(const FleetDev& d, const uint8_t* mask, float* obs, hipStream_t s);
int fleet_max_evs_per_lane_group();  // up to this many EVs per env the single-step kernel gives every EV a lane (fleet_kernels.hip kMaxGroup)
hipError_t fleet_launch_step(const FleetDev& d, const void* actions, int act_dtype, int K, float* obs, double* reward,
                             uint8_t* done, float* terminal_obs, int32_t* done_count, hipStream_t s);
// A single-step launch written down instead of issued (direct AQL submission, fleet_direct.hip): the host-side kernel symbol (its
// name resolves the kernel in the code object), the launch geometry and the kernel-argument block; `actions_offset` = where the two
// copies of the action pointer sit in it (a tape replay patches them per step).  host_fn == nullptr / hipErrorNotSupported: not a
// single-step configuration (real_time, data log).
struct FleetStepLaunch {
  const void* host_fn;
  unsigned grid, block, args_bytes;
  unsigned actions_offset[2];
  unsigned packed_n_offset;  // where `p_N` sits: EVs per env | first workgroup of the grid << 16 (a run split over two queues)
  unsigned guard_offset;     // where the eight bytes of the placement record sit (zeros = no check: every launch through HIP)
  unsigned rec_offset;       // where {blocks pointer, rows, rotate} of a run's FIRST launch sit (the launch that writes the record)
  alignas(8) unsigned char args[640];  // (also the distance between two argument blocks of a run: fleet_direct.hip, the step kernel's record)
};
hipError_t fleet_describe_step(const FleetDev& d, const void* actions, int act_dtype, float* obs, double* reward, uint8_t* done,
                               float* terminal_obs, FleetStepLaunch* out);
// compact the terminal observations of the envs with done[e] != 0 (env order): idx[k], *count, compact[k, obs_dim]
hipError_t fleet_launch_term_compact(const FleetDev& d, const uint8_t* done, const float* term, int32_t* idx, int32_t* count,
                                     double* ep_ret, int32_t* ep_len, float* compact, hipStream_t s);
hipError_t fleet_launch_dist_factor(const FleetDev& d, double* out, hipStream_t s);
// unpack state planes for fleet_get: field ids of include/fleet_hip.h -> contiguous device buffer `out`
hipError_t fleet_launch_gather_field(const FleetDev& d, int field, void* out, hipStream_t s);
// div_rcp (the charge arithmetic's divisions by a reciprocal) against the IEEE sequence on n random operand pairs: bad_dev[0..1]
// cycle_stress against library double precision on n random triples: worst_dev[0] = bits of the largest relative difference
hipError_t fleet_launch_selftest_stress(unsigned long long n, unsigned long long seed, unsigned long long* worst_dev, hipStream_t s);
const char* fleet_kernels_src_sha();  // FLEET_SRC_SHA as compiled into the library's kernels (fleet_kernels.hip)
hipError_t fleet_launch_selftest_division(unsigned long long n, unsigned long long seed, unsigned long long* bad_dev, hipStream_t s);
