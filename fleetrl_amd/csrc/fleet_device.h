// fleet_device.h -- device-side view of one env batch (kernel argument block) shared by the kernels
// (fleet_kernels.hip) and the host side of the C ABI (fleet_capi.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fleet_hip.h"

// Physics row: everything the charge/overload arithmetic needs at time row t, 64 B, one row per t.
// Values are pre-combined on the host ONLY where the reference applies the very same float64 operations
// to per-time scalars (so the result is bit-identical): see fleet_capi.hip build_phys_rows().
struct PhysRow {
  double spot_plus_offset;  // DELU[t]/1000.0 + fixed_markup/1000            (ev_charger.py:145,149)
  double tariff;            // tariff[t]                                     (:194)
  double k_charge;          // -1*price_multiplier*prc[t]/1000               (:154-155)
  double k_discharge;       // -1*price_multiplier*trc[t]/1000               (:204-205)
  double load;              // building load [kW] or 0                       (fleet_environment.py:480-483)
  double pv;                // pv [kW] or 0                                  (:485-488)
  double pv_energy;         // pv[t]*dt [kWh] or 0                           (ev_charger.py:134)
  double reserved;
};

#define FLEET_TFLAG_DEG 1u    // hour == 14 && minute == 45   (fleet_environment.py:665)
#define FLEET_TFLAG_LUNCH 2u  // 11 < hour < 15               (:538)

struct FleetDev {
  // ---- sizes / flags --------------------------------------------------------------------------------
  int E, N, T;
  int obs_dim;
  int episode_steps;
  int hist_cap;     // episode_steps + 2 rows
  int tail_a_len;   // price|tariff|load|pv look-ahead block  (obs offset 2N)
  int tail_b_len;   // evse|grid|avail|pavg|6 time features   (obs offset 2N + tail_a_len + 5N), 0 if !aux
  int tail_stride;  // floats per tail row (tail_a then tail_b, padded to a multiple of 4)
  int aux, normalize, is_caretaker, deg_mode, auto_reset;
  int picker_mode, start_lo, start_hi, env_id_offset;
  int sched_n;
  unsigned long long seed;
  // ---- scalars (FleetParams) ------------------------------------------------------------------------
  double dt, evse_power, p_avail, batt_cap_nominal, init_cap, grid_connection;
  double eta_c, eta_d, variable_multiplier, one_minus_fee, penalty_invalid, penalty_oc, clip_oc, penalty_overload,
      fully_charged_reward, target_soc, target_soc_lunch, eps, def_soc, min_laxity, init_soh, temperature;
  double hn_denominator;  // evse_power * charging_eff  (observer_*.py:88)
  double max_time_left, max_soc, max_hours_needed, max_laxity;
  // ---- read-only tables ---------------------------------------------------------------------------------
  const uint8_t* tab_there;  // [T,N]
  const float* tab_tl;       // [T,N]
  const double* tab_sor;     // [T,N]
  const PhysRow* tab_phys;   // [T]
  const uint8_t* tab_flags;  // [T]
  const float* tab_tail;     // [T,tail_stride]
  const int32_t* sched;           // [sched_n,E] injected start rows or nullptr
  // ---- per-(env,EV) state -----------------------------------------------------------------------------
  double* soc;       // [E,N]
  float* hl;         // [E,N] hours_left (multiples of dt: exact in f32)
  double* soc_deg;   // [E,N]
  double* soh;       // [E,N]
  uint8_t* tgt090;   // [E,N] sticky "target_soc = 0.9" flag (quirk Q7)
  int32_t* rf_len;   // [E,N] RainflowSeiDegradation.rainflow_length
  double* fd_cyc;    // [E,N]
  double* fd_cal;    // [E,N]
  double* sei_l;     // [E,N]
  double* sei_soh;   // [E,N]
  double* hist;      // [hist_cap, E*N]  LogDataDeg.soc_log, time-major so lanes stay coalesced
  double* rf_stack;  // [hist_cap+1, E*N] reversal stack workspace of the rainflow replay
  // ---- per-env state ----------------------------------------------------------------------------------
  int32_t* t_idx;
  int32_t* t_end;
  int32_t* start_idx;
  int32_t* hist_len;
  int32_t* episodes;
  int32_t* ep_len;
  int32_t* last_ep_len;
  double* ep_return;
  double* last_ep_return;
  double* cashflow;
  double* penalty_record;
  uint32_t* err;
  uint8_t* done_flag;
};

// launchers implemented in fleet_kernels.hip
hipError_t fleet_launch_reset(const FleetDev& d, const uint8_t* mask, float* obs, hipStream_t s);
hipError_t fleet_launch_step(const FleetDev& d, const void* actions, int act_dtype, int K, float* obs, double* reward,
                             uint8_t* done, float* terminal_obs, int32_t* done_count, hipStream_t s);
hipError_t fleet_launch_dist_factor(const FleetDev& d, double* out, hipStream_t s);
