/*
 * fleet_hip.h -- C ABI of the MI355X-native batched FleetRL step ("libfleet_hip.so").
 *
 * Drop-in boundary.  The reference has no native code; what this ABI replaces is the
 * Python object protocol of `fleetrl.fleet_env.fleet_environment.FleetEnv`
 *   __init__  /root/reference/fleetrl/fleet_env/fleet_environment.py:76-328
 *   reset     :330-434
 *   step      :436-702
 *   getters   :741-799
 * batched over `num_envs` independent environments (what SB3's SubprocVecEnv does with one OS process per
 * env).  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every entry point returns an int status (FLEET_OK = 0); nothing throws, nothing aborts;
 *     `fleet_last_error(h)` returns a human-readable message for the last non-zero status.
 *   - plain pointers and sizes only; buffers are caller-owned.  `*_host` entry points take host
 *     pointers and are synchronous; `*_dev` entry points take device pointers (e.g. torch-ROCm tensor
 *     data_ptr()) and are asynchronous on the handle's HIP stream.
 *   - one handle = one device + one stream; calls on one handle must be serialised by the caller.
 *   - layouts are row-major:  actions [E,N], obs [E,obs_dim] f32, reward [E] f64, done [E] u8.
 */
#ifndef FLEET_HIP_H
#define FLEET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLEET_ABI_VERSION 10

/* status codes */
#define FLEET_OK 0
#define FLEET_ERR_INVALID 1   /* bad argument / unsupported configuration (SURVEY.md Q4 matrix)            */
#define FLEET_ERR_HIP 2       /* a HIP runtime call failed                                               */
#define FLEET_ERR_STATE 3     /* device-side error word set (impossible state, rainflow stack overflow)  */
#define FLEET_ERR_NODEVICE 4  /* no HIP device: this library has no CPU fallback by design              */
#define FLEET_ERR_UNSUPPORTED 5 /* the platform does not offer what the requested launch mode relies on (FLEET_LAUNCH_DIRECT:
                                 no HSA agent at the HIP device's PCI address, a workgroup -> die placement that is not stable);
                                 use another mode                                                         */

/* FleetParams.deg_mode -- which battery-degradation model runs on the daily 14:45 step                   */
#define FLEET_DEG_NONE 0      /* calculate_degradation = False                                           */
#define FLEET_DEG_LINEAR 1    /* EmpiricalDegradation  (utils/battery_degradation/empirical_degradation.py:29-99), quirk Q1 */
#define FLEET_DEG_RAINFLOW 2  /* RainflowSeiDegradation (rainflow_sei_degradation.py:91-212)            */

/* FleetParams.picker_mode -- episode start row (utils/time_picker/)                                      */
#define FLEET_PICK_STATIC 0   /* always start_lo                                                         */
#define FLEET_PICK_RANDOM 1   /* uniform integer in [start_lo, start_hi], counter-based Philox4x32-10   */
#define FLEET_PICK_EVAL 2     /* same sampler, caller passes the validation range                       */

/* action element type */
#define FLEET_ACT_F32 0
#define FLEET_ACT_F64 1
/* built-in open-loop policies (fleet_rollout_policy_dev): the action rules of the reference's benchmark harnesses */
#define FLEET_ACT_POLICY_UNCONTROLLED 2 /* all ones           (benchmarking/uncontrolled_charging.py:51-54) */
#define FLEET_ACT_POLICY_DISTRIBUTED 3  /* clip(get_dist_factor(), 0, 1) (benchmarking/distributed_charging.py:50-54) */
#define FLEET_ACT_POLICY_NIGHT 4        /* time-window rule, stateful per env (benchmarking/night_charging.py:81-98);
                                           configure with fleet_set_night_policy first */

/* device-side error bits (per env, OR-ed into one word; see fleet_get(FLEET_F_ERROR_BITS))               */
#define FLEET_DEVERR_OBS_FORMAT 1u     /* the reference's `raise TypeError("Observation format not recognized")` :610 */
#define FLEET_DEVERR_NEG_LIFE 2u       /* "Life degradation is negative" rainflow_sei_degradation.py:179-180         */
#define FLEET_DEVERR_SOH_MISMATCH 4u   /* "Degradation calculation is not correct" :209-210                         */
#define FLEET_DEVERR_DOD_RANGE 8u      /* "DoD too large" :164-167                                                   */
#define FLEET_DEVERR_TABLE_END 16u     /* episode ran past the last table row                                        */
#define FLEET_DEVERR_INTERNAL 32u      /* a kernel found its launch arguments inconsistent (a build problem, not a data one) */
#define FLEET_DEVERR_PLACEMENT 64u     /* a launch of a run on the library's own queue (FLEET_LAUNCH_DIRECT ...) found one of its
                                          workgroups on another die than the library's probe of that queue says: the run's state
                                          may be stale, its results are void (a platform problem, not a data one)                */

/*
 * Scalars of one env group (all envs of a handle share tables and parameters).
 * Field meaning follows the reference's config classes; see fleetrl_amd/config.py for how each is
 * derived from the reference's config dict.
 */
typedef struct FleetParams {
  int32_t abi_version;      /* FLEET_ABI_VERSION */
  int32_t struct_bytes;     /* sizeof(FleetParams) as seen by the caller */
  int32_t num_envs;         /* E */
  int32_t num_cars;         /* N  (db["ID"].max()+1, fleet_environment.py:260) */
  int32_t table_rows;       /* T */
  int32_t episode_steps;    /* episode_length[h] * steps_per_hour; finish = start + episode_steps (:355).  Rainflow mode:
                               < 2^26 - 3, and num_cars * (episode_steps + 24) * 8 bytes < 4 GiB */
  int32_t price_lookahead;  /* L  (time_config.py:14) */
  int32_t bl_pv_lookahead;  /* B  (time_config.py:15) */
  int32_t steps_per_hour;   /* 60 / minutes */
  int32_t hour_phase;       /* table row 0 lies `hour_phase` steps after a full clock hour (0 for shipped data) */
  int32_t include_building; /* flags, fleet_environment.py:139-145 */
  int32_t include_pv;
  int32_t aux;
  int32_t normalize;        /* normalize_in_env -> OracleNormalization, else UnitNormalization */
  int32_t is_caretaker;     /* CompanyType.Caretaker: lunch-break target SOC (:536-554) */
  int32_t deg_mode;         /* FLEET_DEG_* */
  int32_t picker_mode;      /* FLEET_PICK_* */
  int32_t start_lo;         /* inclusive */
  int32_t start_hi;         /* inclusive */
  int32_t auto_reset;       /* 1: VecEnv semantics (done envs are reset inside the step, terminal obs reported)
                               0: gymnasium.Env semantics (obs of a done env is its terminal obs)           */
  int32_t env_id_offset;    /* global index of env 0 of this handle (multi-GPU sharding keeps RNG streams env-stable) */
  int32_t log_data;         /* 1: device-side data log -- every row the reference's DataLogger would get
                               (utils/data_logger/data_logger.py:21-68; call sites fleet_environment.py:420-432, 659-690) is
                               written to a per-env ring by the kernels, in every mode (single step, K steps per launch,
                               policy rollouts, resets); read with fleet_log_read.  0: nothing is logged */
  int32_t real_time;        /* 1: event-skipping step (fleet_environment.py:453,692-699, event_manager.py:16-31): the same
                               action is applied row after row until a relevant event (departure, arrival, penalty,
                               overload, episode end, clock minute 15); irregular time grids need FleetTables.dt_row etc. */
  int32_t log_capacity;     /* rows per env of the data-log ring; 0 = 2 * (episode_steps + 1).  Older rows are overwritten */
  uint64_t seed;            /* Philox key for the random/eval picker */

  double dt;                /* hours per step (time_config.py:24) */
  double evse_power;        /* load_calculation.evse_max_power */
  double obc_max_power;     /* ev_config.obc_max_power; possible_power = min(obc, evse) (ev_charger.py:95) */
  double batt_cap_nominal;  /* load_calculation.batt_cap, used by the aux observations only (observer_*.py:88) */
  double init_battery_cap;  /* ev_config.init_battery_cap */
  double grid_connection;   /* load_calculation.grid_connection */
  double charging_eff;
  double discharging_eff;
  double fixed_markup;      /* EUR/MWh (obs) ; spot_offset = fixed_markup/1000 (ev_charger.py:34) */
  double variable_multiplier;
  double feed_in_deduction;
  double price_multiplier;  /* already scaled by max_batt_cap/init_cap (:194-195) and zeroed by ignore_price_reward */
  double penalty_invalid_action;
  double penalty_overcharging;
  double clip_overcharging;
  double penalty_overloading;
  double fully_charged_reward;
  double target_soc;
  double target_soc_lunch;
  double eps;               /* 0.005 (:230) */
  double def_soc;
  double min_laxity;
  double init_soh;
  double temperature;
  /* OracleNormalization.__init__ constants (oracle_normalization.py:34-54); ignored unless `normalize` */
  double max_time_left;
  double max_price, min_price;
  double max_tariff, min_tariff;
  double max_building;
  double max_pv;
  double max_soc;
  double max_hours_needed;
  double max_laxity;
  double max_evse;
  double max_grid;
} FleetParams;

/* Pre-staged tables (HOST pointers; fleet_create uploads / re-packs them).  Row t, EV c -> [t*N + c].     */
typedef struct FleetTables {
  const uint8_t* there;         /* [T,N]  db["There"]                          */
  const float* time_left;       /* [T,N]  db["time_left"], multiples of dt     */
  const double* soc_on_return;  /* [T,N]  db["SOC_on_return"]                  */
  const double* delu;           /* [T]    EUR/MWh                              */
  const double* tariff;         /* [T]                                         */
  const double* prc;            /* [T]    price_reward_curve                   */
  const double* trc;            /* [T]    tariff_reward_curve                  */
  const double* load;           /* [T]    kW (zeros if !include_building)      */
  const double* pv;             /* [T]    kW (zeros if !include_pv)            */
  const uint8_t* hour;          /* [T]                                         */
  const uint8_t* minute;        /* [T]                                         */
  const uint8_t* month;         /* [T]  1..12                                  */
  const uint8_t* weekday;       /* [T]  Monday = 0                             */
  const float* time_feat;       /* [T,6] month/week/hour sin,cos as float32, or NULL (library computes with libm) */
  /* Irregular time grids (real_time only; all NULL / 0 for a regular grid, where the library derives them from
   * FleetParams.dt / steps_per_hour / hour_phase).  The reference reads the step length off the data
   * (`get_next_dt`, fleet_environment.py:994-1022), ends the episode on the row whose date equals start + episode_length
   * (:355, :627) and takes the hourly look-ahead values from the first row of each clock hour
   * (`resample("H").first()`, observer_bl_pv.py:53-79). */
  const double* dt_row;         /* [T]  hours from row t to row t+1 (last row: any positive value)                   */
  const int32_t* finish_row;    /* [T]  row whose date is date[t] + episode_length hours, or -1 (the episode never ends) */
  const int32_t* lookahead_row; /* [T,lookahead_cols]  column k-1: first row of clock hour floor_hour(date[t]) + k, k >= 1;
                                   -1: no such row                                                                     */
  int32_t lookahead_cols;       /* >= max(price_lookahead, bl_pv_lookahead)                                           */
  int32_t reserved0;
  const uint8_t* second;        /* [T]  seconds of the row's clock time (the clock-minute-15 event needs second == 0) */
  /* Candidate start rows of the random / eval time pickers, or NULL (= every row of [start_lo, start_hi]).  The
   * reference draws from a `date_range` at the model frequency (random_time_picker.py:25-28), which on an irregular
   * grid is a subset of the rows; with this list FleetParams.start_lo / start_hi index INTO it. */
  const int32_t* pick_rows;     /* [n_pick_rows] ascending row numbers */
  int32_t n_pick_rows;
  int32_t reserved1;
} FleetTables;

typedef struct FleetEnvBatch* fleet_handle;

/* fields for fleet_get (all copied to a HOST buffer) */
#define FLEET_F_SOC 0            /* f64 [E,N] */
#define FLEET_F_HOURS_LEFT 1     /* f32 [E,N] */
#define FLEET_F_SOH 2            /* f64 [E,N] */
#define FLEET_F_SOC_DEG 3        /* f64 [E,N] */
#define FLEET_F_TARGET_SOC 4     /* f64 [E,N] */
#define FLEET_F_TIME_IDX 5       /* i32 [E]   */
#define FLEET_F_START_IDX 6      /* i32 [E]   */
#define FLEET_F_CASHFLOW 7       /* f64 [E]   last step's cashflow (episode.current_charging_expense) */
#define FLEET_F_EP_RETURN 8      /* f64 [E]   running episode return (episode.cumulative_reward)      */
#define FLEET_F_EP_LEN 9         /* i32 [E]   */
#define FLEET_F_LAST_EP_RETURN 10 /* f64 [E]  return of the last finished episode */
#define FLEET_F_LAST_EP_LEN 11   /* i32 [E]   */
#define FLEET_F_RF_LEN 12        /* i32 [E,N] sei_deg.rainflow_length */
#define FLEET_F_FD_CYC 13        /* f64 [E,N] */
#define FLEET_F_FD_CAL 14        /* f64 [E,N] */
#define FLEET_F_SEI_L 15         /* f64 [E,N] */
#define FLEET_F_ERROR_BITS 16    /* u32 [E]   */
#define FLEET_F_DONE 17          /* u8  [E]   episode.done */
#define FLEET_F_EPISODES 18      /* i32 [E]   finished-episode counter */
#define FLEET_F_PENALTY_RECORD 19 /* f64 [E]  episode.penalty_record */
#define FLEET_F_LAST_EP_LEN_F64 20 /* f64 [E] length of the last finished episode as float64 (fleet_get_dev / the RCCL gather) */
#define FLEET_F_RF_CYCLES 21     /* i32 [E,N] rainflow cycles closed so far in the running episode (0 without rainflow degradation) */
#define FLEET_F_RF_STACK 22      /* i32 [E,N] reversal points on the EV's rainflow stack (bench.py derives the share of EV-steps that
                                    push a reversal point / close a cycle from the two: the workload's invariants)               */
#define FLEET_F_RF_UNTIL 23      /* i32 [E]   the last table row of the running episode on which the degradation model is evaluated
                                    (-1: none; INT32_MAX after fleet_set_rainflow_count_all): SOC samples logged after it are not
                                    counted, see fleet_set_rainflow_count_all                                                   */

/* ---- lifetime ------------------------------------------------------------------------------------- */
int fleet_obs_dim(const FleetParams* p);  /* detect_dim_and_bounds, fleet_environment.py:854-949; <0 on invalid flags */
int fleet_create(const FleetParams* p, const FleetTables* t, int device, fleet_handle* out);
int fleet_destroy(fleet_handle h);
const char* fleet_last_error(fleet_handle h);  /* h may be NULL: error of the last failed fleet_create */
/* Launch on an external hipStream_t from now on (e.g. torch's current stream, so that launches are ordered with the torch ops
 * around them).  The stream is borrowed: the caller keeps it alive while the handle uses it.  The handle's own stream is
 * kept; fleet_use_own_stream goes back to it.  Both synchronise the stream in use so far and drop a cached tape graph. */
int fleet_set_stream(fleet_handle h, void* hip_stream);
int fleet_use_own_stream(fleet_handle h);
int fleet_get_stream(fleet_handle h, void** hip_stream); /* the hipStream_t the handle launches on (to order another stream
                                                            against it with events) */
int fleet_synchronize(fleet_handle h);
/* non-blocking: FLEET_OK when everything enqueued on the handle's stream has finished, -1 while it has not (hipStreamQuery) */
int fleet_stream_query(fleet_handle h);

/* Inject episode start rows (parity tests / `set_start_time`): `starts` is HOST [n_episodes,E]; episode k of
 * env e starts at starts[(k % n_episodes)*E + e].  n_episodes = 0 clears the schedule (picker resumes). */
int fleet_set_start_schedule(fleet_handle h, const int32_t* starts, int n_episodes);

/* ---- reset / step, device pointers, asynchronous on the handle's stream --------------------------------
 * mask: u8 [E] or NULL (= all).  obs: f32 [E,obs_dim] (rows of unmasked envs are left untouched).        */
int fleet_reset_dev(fleet_handle h, const uint8_t* mask, float* obs);
int fleet_step_dev(fleet_handle h, const void* actions, int act_dtype, float* obs, double* reward,
                   uint8_t* done, float* terminal_obs /* [E,obs_dim] or NULL */);
/* K consecutive steps in ONE launch for open-loop rollouts (actions known up front): actions [K,E,N];
 * obs = observation after the last step, reward_sum[E] = sum of the K rewards, done_count[E] (or NULL) = number of
 * episode ends among them.  auto_reset must be 1. */
int fleet_step_many_dev(fleet_handle h, int K, const void* actions, int act_dtype, float* obs,
                        double* reward_sum, int32_t* done_count);

/* K consecutive steps in ONE launch with a built-in policy (FLEET_ACT_POLICY_*) evaluated on the device instead of an
 * action tape: the reference's `benchmarking/` harnesses without a host round trip per step.  Outputs as
 * fleet_step_many_dev.  auto_reset must be 1. */
int fleet_rollout_policy_dev(fleet_handle h, int policy, int K, float* obs, double* reward_sum, int32_t* done_count);
/* Parameters of FLEET_ACT_POLICY_NIGHT, as the reference's harness derives them before its loop
 * (benchmarking/night_charging.py:50-73): charging starts when `charging_hour <= hour && charging_minute <= minute`
 * of the env's current row (the reference's own, non-lexicographic test), runs until more than `max_hours`
 * (= int(max_time_needed)) have passed since it started, and on a caretaker fleet the rows of hours 11..14 use the
 * distributed rule instead.  Each env keeps its own "charging since" state, which -- like the reference's loop
 * variable -- survives episode resets; this call clears it.  Not callable while a captured graph is replaying. */
int fleet_set_night_policy(fleet_handle h, int charging_hour, int charging_minute, int max_hours);
/* The rainflow count stops at the episode's last degradation row.  The reference appends one SOC sample per EV and step to
 * LogDataDeg (fleet_environment.py:655) and runs rainflow.extract_cycles on that log only on the 14:45 rows (:665,
 * rainflow_sei_degradation.py:132); reset() clears the log (:338-339).  What is logged between an episode's last 14:45 row and
 * its end is therefore never read by anybody -- with 48 h episodes a quarter of all samples -- and the kernels, which count
 * while they log, stop counting there (FLEET_F_RF_UNTIL; state of health, fd_cyc, rainflow_length, observations, rewards:
 * exactly as with the full count, tests/test_rf_tail_gpu.py).  `on` != 0 keeps the count running to the end of every episode
 * from each env's next reset on (FLEET_F_RF_CYCLES / FLEET_F_RF_STACK then describe the whole series: diagnostics, the
 * adversarial count tests).  Default: off.  Not callable while a captured graph is replaying. */
int fleet_set_rainflow_count_all(fleet_handle h, int on);

/* ---- reset / step, host pointers, synchronous ------------------------------------------------------------ */
int fleet_reset_host(fleet_handle h, const uint8_t* mask, float* obs);
/* terminal_obs (or NULL): rows of envs that finished in this step are written; all other rows are left untouched.
 * `obs` / `actions` may be any host memory: buffers from fleet_host_alloc (pinned) are transferred to / from directly; a
 * pageable `obs` receives the observations in pieces through a pinned buffer of the handle, each piece copied to it while
 * the next ones are still on the link. */
int fleet_step_host(fleet_handle h, const void* actions, int act_dtype, float* obs, double* reward,
                    uint8_t* done, float* terminal_obs);

/* The episodes that ended in the last fleet_step_host call (the one with a terminal_obs buffer): their env indices in
 * ascending order, returns (episode.cumulative_reward) and lengths -- what SB3's Monitor wrapper would put into
 * info["episode"].  The pointers refer to the handle's own host buffer and stay valid until the next step. */
int fleet_last_step_episodes(fleet_handle h, int32_t* n, const int32_t** env_idx, const double** ep_return, const int32_t** ep_len);

/* Pinned (page-locked) host memory for the *_host entry points: observations / actions in such buffers cross PCIe straight
 * out of / into them at the full link rate; any other host pointer is staged through the handle's own pinned mirrors (one
 * extra memcpy).  Independent of any handle; free with fleet_host_free. */
int fleet_host_alloc(size_t bytes, void** out);
int fleet_host_free(void* p);

/* ---- state access ------------------------------------------------------------------------------------- */
int fleet_get(fleet_handle h, int field, void* out_host);
/* same, into a DEVICE buffer and asynchronous on the handle's stream (e.g. the episode returns that a multi-GPU run
 * all-gathers for logging, without a host round trip) */
int fleet_get_dev(fleet_handle h, int field, void* out_dev);
/* `FleetEnv.get_dist_factor` (:782-799): hours_needed / (hours_left + 0.001) from a fresh observation, f64 [E,N] */
int fleet_get_dist_factor(fleet_handle h, double* out_host);
/* ---- multi-GPU logging gather ---------------------------------------------------------------------
 * Envs are independent: one process per GPU, each with a handle for its own contiguous env range (FleetParams.env_id_offset),
 * nothing exchanged on the data path.  The one collective is the gather of the finished episodes' returns / lengths for
 * logging -- ONE RCCL all-gather over xGMI, enqueued on the handle's stream, no PyTorch in the process needed (the Python
 * package does the same through torch.distributed, fleetrl_amd/distributed.py).  librccl is opened at run time.
 *   fleet_rccl_unique_id       rank 0 makes the 128-byte id; the caller's launcher hands it to the other ranks (file, env, MPI ...)
 *   fleet_rccl_comm_create     every rank: ncclCommInitRank on `device`; *comm is an ncclComm_t
 *   fleet_gather_episode_stats_rccl   out_dev: DEVICE f64 [world_size, 2, E]: per rank, last_ep_return[E] then last_ep_len[E]
 *                              (as float64) of that rank's envs; every rank must have the same E.  Asynchronous on the
 *                              handle's stream (fleet_synchronize before reading). */
#define FLEET_RCCL_UNIQUE_ID_BYTES 128
int fleet_rccl_unique_id(void* id128);
int fleet_rccl_comm_create(int device, int world_size, int rank, const void* id128, void** comm);
int fleet_rccl_comm_destroy(void* comm);
int fleet_gather_episode_stats_rccl(fleet_handle h, void* comm, int world_size, double* out_dev);

/* ---- device-side data log (FleetParams.log_data = 1) ---------------------------------------------------------------
 * What `FleetEnv.get_log()` (:741-748) returns is rebuilt from this ring: row k of env e (k counted since creation or the
 * last fleet_log_clear; the ring holds the last `capacity` of them, row k in slot k % capacity) is
 *   row [slot,E]      i32  table row of episode.time; bit 31 set: the row reset() writes (zeros + observation + SoH)
 *   env [slot,E,4]    f64  reward, cashflow, overload_amount [kW], cum_soc_missing     (Penalties = reward - cashflow *
 *                          price_multiplier and the absolute values are the caller's, :659-661)
 *   ev  [slot,E,4,N]  f64  action, (dis)charging energy per EV [kWh] (ev_charger.py:114,174), degradation, SoH
 *   obs [slot,E,obs_dim] f32  the observation logged with the row
 * The step that ends an episode is not logged (:679); with auto-reset the next row is the reset row of the next episode.
 * With real_time = 1 every table row the skipping loop passes gets its row, like in the reference (:677-690). */
int fleet_log_capacity(fleet_handle h);  /* rows per env; 0 when the log is off */
/* rows the ring has already overwritten, summed over the envs (sum of max(pos - capacity, 0)): what a caller that wants every
 * row -- like the reference's unbounded DataLogger -- has lost; size FleetParams.log_capacity so that this stays 0 */
int fleet_log_dropped(fleet_handle h, int64_t* rows);
/* copy the ring to HOST buffers (any of them may be NULL); pos [E] = rows written so far per env; synchronous */
int fleet_log_read(fleet_handle h, int32_t* pos, int32_t* row, double* env, double* ev, float* obs);
int fleet_log_clear(fleet_handle h);     /* forget all rows (asynchronous on the handle's stream) */

/* Device-side errors -- the conditions under which the reference raises inside step(): `TypeError("Observation format not
 * recognized")` fleet_environment.py:610, `TypeError("DoD too large.")` / `TypeError("Life degradation is negative")` /
 * `RuntimeError("Degradation calculation is not correct")` rainflow_sei_degradation.py:164-167,179-180,209-210, and a table
 * lookup past the last row -- set FLEET_DEVERR_* bits per env (sticky).
 *   - fleet_step_host returns FLEET_ERR_STATE from the very call whose step raised them (its outputs are still complete): the OR
 *     of all bits travels in the block that brings rewards and dones back, no extra launch or transfer.
 *   - the *_dev entry points are asynchronous: poll with fleet_check_errors (one small launch + copy). */
/* FLEET_ERR_STATE if any env has device error bits set; fleet_last_error names the first such env, its table row and the bits */
int fleet_check_errors(fleet_handle h);
/* the OR of all envs' error bits as of the last fleet_step_host (no device work) */
int fleet_last_step_error_bits(fleet_handle h, uint32_t* bits);

/* ---- measurement helpers (bench.py): HIP events on the handle's stream ------------------------------- */
int fleet_timer_start(fleet_handle h);
int fleet_timer_stop(fleet_handle h, float* elapsed_ms);  /* synchronises on the stop event */
/* the same in two halves, for several handles whose streams run concurrently: record every handle's stop event first
 * (asynchronous), then read them (each read synchronises on its own stop event) */
int fleet_timer_mark(fleet_handle h);
int fleet_timer_read(fleet_handle h, float* elapsed_ms);
/* launch `steps` single-step launches back to back from a device-resident action tape [tape_len,E,N]
 * (step i uses tape row i % tape_len).  `use_graph`: how the launches reach the GPU --
 *   FLEET_LAUNCH_EAGER   one hipLaunchKernel per step on the handle's stream
 *   FLEET_LAUNCH_GRAPH   a captured hipGraph of whole tape cycles (>= 64 launches), replayed; shorter remainders eagerly
 *   FLEET_LAUNCH_DIRECT  AQL dispatch packets written by the library into an HSA queue of the handle's own, with the cache
 *                        actions HIP attaches to every kernel boundary reduced to what a run of steps needs: every launch still
 *                        invalidates the per-CU caches, only the LAST launch of the run writes the L2s back: no launch
 *                        waits for the previous one's write-back (workgroup w of every launch of a run stays on one die, so a die
 *                        only reads state it wrote itself; probed when the queue is opened, recorded at the start of every run and
 *                        checked by every launch: FLEET_DEVERR_PLACEMENT, fleet_direct_placement).
 *                        Semantics: asynchronous like the others, but NOT on the HIP stream -- the run starts after everything
 *                        the stream holds has completed (the call waits for that), nothing of it is visible before it has
 *                        completed, and every later call on the handle (fleet_synchronize, a step, a get ...) waits for it first;
 *                        fleet_stream_query reports it.  Single-step configurations only (no real_time, no data log);
 *                        needs libfleet_hip.gfx950.hsaco beside the library (fleetrl_amd.build).  */
#define FLEET_LAUNCH_EAGER 0
#define FLEET_LAUNCH_GRAPH 1
#define FLEET_LAUNCH_DIRECT 2
/* FLEET_LAUNCH_DIRECT covers a batch of more wavefronts than the device holds at once (>= 6144, envs of up to 64 EVs) with two
 * ranges of workgroups on TWO queues, each an in-order chain of its own (the halves drift apart and overlap: 16384 x 50 -17 % per
 * step); FLEET_LAUNCH_DIRECT_ONE_QUEUE never does.  fleet_direct_queues: how the handle's last direct run was laid out (0: none yet). */
#define FLEET_LAUNCH_DIRECT_ONE_QUEUE 3
int fleet_run_tape_dev(fleet_handle h, int steps, const void* tape, int tape_len, int act_dtype,
                       float* obs, double* reward, uint8_t* done, int use_graph);

int fleet_direct_queues(fleet_handle h);

/* What FLEET_LAUNCH_DIRECT relies on, as probed when the handle opened its queue (opens it if need be):
 * map8[k] = the die (HW_REG_XCC_ID) that chain of probe launches found workgroups w with (w & 7) == k on; num_xcc = dies of the
 * device; any_grid = 1 if the map also held across launches whose grids are not multiples of 8 workgroups.  The map is information:
 * the die a queue deals from moves whenever a queue is created or destroyed in the process, so every run records the map afresh on
 * the device (its first launch) and every step launch checks the die it runs on against that record (FLEET_DEVERR_PLACEMENT).  FLEET_ERR_UNSUPPORTED: the probe
 * found the placement not periodic or not stable from launch to launch, and the mode is refused on this platform. */
int fleet_direct_placement(fleet_handle h, int32_t map8[8], int32_t* num_xcc, int32_t* any_grid);
/* How a grid of `grid_workgroups` is laid over the handle's queues (a pure function, no device needed): returns 1 (part_grid[0] = the
 * whole grid) or 2 (two ranges of workgroups: part_grid[0] a multiple of 8 that fits the kernel's 16-bit first-workgroup field). */
int fleet_direct_split_plan(uint32_t grid_workgroups, int split, uint32_t part_grid[2]);
/* TEST HOOK for the placement guard (the handle must have run through its own queue before) --
 * kind 1: the handle's NEXT run gets a placement record shifted by one workgroup: what its launches would see if the queue's first
 *         die had moved in the middle of the run;
 * kind 2: in the prepared argument block of tape row `tape_row`, the grid's first workgroup shifted by one (every workgroup steps its
 *         neighbour's envs: the state IS corrupted).
 * The next run (through that row) must raise FLEET_DEVERR_PLACEMENT. */
int fleet_debug_direct_fault(fleet_handle h, int kind, int tape_row);

/* `regions` timed regions of exactly `steps` launches each (as fleet_run_tape_dev), enqueued back to back on the handle's stream
 * with a HIP event before and after each: the kernels' own time per region, without the host's gaps between regions.
 * _begin only enqueues (several handles' streams can be filled before any is read); _read waits and returns the per-region
 * device durations in milliseconds (HOST array [regions]). */
/* (FLEET_LAUNCH_DIRECT: the regions are the runs' own dispatch timestamps -- start of the first launch to end of the last) */
int fleet_time_regions_begin(fleet_handle h, int regions, int steps, const void* tape, int tape_len, int act_dtype,
                             float* obs, double* reward, uint8_t* done, int use_graph);
int fleet_time_regions_read(fleet_handle h, float* region_ms);

/* like fleet_run_tape_dev without a graph, but brackets EVERY launch with its own HIP event pair on the handle's
 * stream and returns the per-launch device durations in milliseconds (HOST array [steps]); synchronous. */
int fleet_time_steps_dev(fleet_handle h, int steps, const void* tape, int tape_len, int act_dtype, float* obs,
                         double* reward, uint8_t* done, float* per_launch_ms);

/* ---- self-test --------------------------------------------------------------------------------------- */
/* The charge arithmetic (EvCharger.charge, ev_charger.py:114,128,189) divides by eta_c and by the battery capacity; the kernels
 * form both quotients from a reciprocal with a residual correction instead of the IEEE division sequence.  This entry runs both
 * forms on `n_pairs` pseudo-random operand pairs of the charge arithmetic's ranges on the device and returns how many quotients
 * differ in any bit: mismatches[0] for `need / eta_c`, mismatches[1] for `energy / cap` (expected: 0 and 0). */
int fleet_selftest_division(int device, uint64_t n_pairs, uint64_t seed, uint64_t* mismatches);
/* The stress of a closed rainflow cycle (deg_rate_cycle, rainflow_sei_degradation.py:68-80: (kd1 * dod^kd2 + kd3)^-1 * e^(k_sigma *
 * (soc - sigma_ref)) * stress_temp) is evaluated with a hardware float32 logarithm inside dod^-0.501 and polynomial exponentials instead
 * of the library's pow / exp.  This entry evaluates both forms on `n_samples` pseudo-random (depth of discharge, mean SOC, cycle weight)
 * triples of the reachable domain on the device and returns the largest relative difference (expected: < 1e-8; the degradation it
 * feeds is a 1e-5-sized correction to the state of health). */
int fleet_selftest_stress(int device, uint64_t n_samples, uint64_t seed, double* max_rel_err);

#ifdef __cplusplus
}
#endif
#endif /* FLEET_HIP_H */
