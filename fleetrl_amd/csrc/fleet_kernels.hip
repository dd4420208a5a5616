// fleet_kernels.hip -- the FleetEnv step / reset hot path as hand-written HIP for gfx950 (MI355X, CDNA4).
//
// What runs here (reference: /root/reference/fleetrl, all float64 in the reference's operation order):
//   EvCharger.charge            utils/ev_charging/ev_charger.py:39-231
//   LoadCalculation.check_violation + ScoreConfig.overloading_penalty
//                               utils/load_calculation/load_calculation.py:83-94, fleet_env/config/score_config.py:33-41
//   arrival/departure state machine + ScoreConfig.soc_violation_penalty
//                               fleet_env/fleet_environment.py:528-623, score_config.py:26-30
//   Observer*.get_obs + Unit/OracleNormalization.normalize_obs
//                               utils/observation/observer_*.py, utils/normalization/*.py
//   LogDataDeg.log_soc, RainflowSeiDegradation / EmpiricalDegradation.calculate_degradation
//                               utils/battery_degradation/*.py
//   FleetEnv.reset (incl. the vec-env auto-reset)  fleet_environment.py:330-434
//
// Mapping.  One *group* of G lanes owns one env (G = smallest power of two >= min(N,64)); a 64-lane wavefront
// holds 64/G envs and a 256-thread workgroup 256/G.  Lane g of a group owns EVs g, g+G, ...  Per-env sums
// (cost, revenue, reward, sum(action*there)) are reduced inside the wavefront with DPP row shifts / row
// broadcasts -- no LDS round trip, no atomics; "connected cars" is a popcount of a wave ballot.
// Every lane of a group tracks the per-env scalars (time row, episode end, history length) redundantly in
// registers, so nothing written by one lane is ever re-read by another inside a launch.
// Tables are read with the EV index fastest ([T,N] rows) and state with [E,N] rows, so a wavefront's
// accesses are contiguous runs of N elements.  No MFMA: there is no contraction on this path.
#include "fleet_device.h"

namespace {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------------------
// wavefront helpers
// ---------------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // old = 0 and bound_ctrl = 1: lanes whose source is out of range (or whose row is masked off) add 0.0
  int l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true);
  int h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
  return v + __hiloint2double(h2, l2);
}

// Sum over the G lanes of an aligned group; the result is valid in the LAST lane of the group.
// row_shr:1/2/4/8 (0x111..0x118) scan inside a 16-lane row, row_bcast:15 (0x142) and row_bcast:31 (0x143)
// carry row totals across rows.  All 64 lanes must execute this (uniform control flow).
template <int G>
__device__ __forceinline__ double group_sum_to_last(double v) {
  if (G >= 2) v = dpp_add<0x111, 0xF>(v);
  if (G >= 4) v = dpp_add<0x112, 0xF>(v);
  if (G >= 8) v = dpp_add<0x114, 0xF>(v);
  if (G >= 16) v = dpp_add<0x118, 0xF>(v);
  if (G >= 32) v = dpp_add<0x142, 0xA>(v);
  if (G >= 64) v = dpp_add<0x143, 0xC>(v);
  return v;
}

template <int G>
__device__ __forceinline__ int group_count(bool pred, int lane) {
  unsigned long long m = __ballot(pred);
  if (G == 64) return __popcll(m);
  const int base = lane & ~(G - 1);
  return __popcll((m >> base) & ((1ull << G) - 1ull));
}

// Philox4x32-10 start-row sampler; same specification as the oracle's (counter = (global env, episode, 0, 0)).
__device__ __forceinline__ uint32_t philox_start(unsigned long long seed, uint32_t env, uint32_t episode) {
  uint32_t c0 = env, c1 = episode, c2 = 0, c3 = 0;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c0;
}

__device__ __forceinline__ int choose_start(const FleetDev& d, int e, int episode) {
  if (d.sched_n > 0) return d.sched[(size_t)(episode % d.sched_n) * d.E + e];
  if (d.picker_mode == FLEET_PICK_STATIC) return d.start_lo;
  const uint32_t range = (uint32_t)(d.start_hi - d.start_lo + 1);
  const uint32_t x = philox_start(d.seed, (uint32_t)(d.env_id_offset + e), (uint32_t)episode);
  return d.start_lo + (int)__umulhi(x, range);
}

// ScoreConfig.soc_violation_penalty (score_config.py:26-30)
__device__ __forceinline__ double soc_violation_penalty(double missing) {
  return -500.0 / (1.0 + exp(-16.48461585 * (missing - 0.29229767))) + 1.0;
}

// ScoreConfig.overloading_penalty (score_config.py:33-41)
__device__ __forceinline__ double overloading_penalty(double rel, double scale) {
  const double pen = (rel < 1.1) ? 0.0 : -700.0 / (1.0 + exp(-15.77350877 * (rel - 1.33298382)));
  return pen * scale;
}

// ---------------------------------------------------------------------------------------------------------
// observation assembly (observer_*.py + normalization/*.py); layout: DESIGN.md "Observation row"
// ---------------------------------------------------------------------------------------------------------
// per-EV slots of EV c at table row t: soc, hours_left from live state; the five aux slots from the TABLE row
// (quirk Q10: not from live state), observer_bl_pv.py:85-91.
__device__ __forceinline__ void write_obs_ev(const FleetDev& d, float* __restrict__ row, int c, int t, double soc,
                                             float hl, double tgt) {
  const int N = d.N;
  row[c] = (float)soc;
  row[N + c] = d.normalize ? (float)((double)hl / d.max_time_left) : hl;
  if (!d.aux) return;
  const size_t ti = (size_t)t * N + c;
  const double th = (double)d.tab_there[ti];
  const double tgt_th = tgt * th;                        // target_soc * there
  const double cl = tgt_th - d.tab_sor[ti];              // charging_left
  const double hn = cl * d.batt_cap_nominal / d.hn_denominator;  // hours_needed
  double lax = ((double)d.tab_tl[ti] / (hn + 0.001) - 1.0) * th;
  lax = lax < 0.0 ? 0.0 : (lax > 5.0 ? 5.0 : lax);       // np.clip(laxity, 0, 5)
  float* a = row + 2 * N + d.tail_a_len;
  if (d.normalize) {
    a[c] = (float)th;
    a[N + c] = (float)(tgt_th / d.max_soc);
    a[2 * N + c] = (float)(cl / d.max_soc);
    a[3 * N + c] = (float)(hn / d.max_hours_needed);
    a[4 * N + c] = (float)(lax / d.max_laxity);
  } else {
    a[c] = (float)th;
    a[N + c] = (float)tgt_th;
    a[2 * N + c] = (float)cl;
    a[3 * N + c] = (float)hn;
    a[4 * N + c] = (float)lax;
  }
}

// env-level blocks: a pure function of the table row, pre-assembled (and pre-normalised) on the host.
template <int G>
__device__ __forceinline__ void write_obs_tail(const FleetDev& d, float* __restrict__ row, int t, int g) {
  const float* __restrict__ src = d.tab_tail + (size_t)t * d.tail_stride;
  float* a = row + 2 * d.N;
  for (int j = g; j < d.tail_a_len; j += G) a[j] = src[j];
  float* b = row + 2 * d.N + d.tail_a_len + 5 * d.N;
  for (int j = g; j < d.tail_b_len; j += G) b[j] = src[d.tail_a_len + j];
}

// ---------------------------------------------------------------------------------------------------------
// battery degradation (daily, on the 14:45 row)
// ---------------------------------------------------------------------------------------------------------
// RainflowSeiDegradation.calculate_degradation for one EV (rainflow_sei_degradation.py:91-212), with the
// rainflow.extract_cycles replay fused in as a single streaming pass over the episode's SOC history:
//   * reversal stack in HBM workspace (column i of rf_stack), bounded by the history length;
//   * cycles are consumed the moment they are emitted: sum of means (-> mean_soc_cal), count (-> len),
//     and the stress sum over the slice [rainflow_length-1, len-1), which needs a one-cycle delay because
//     the slice excludes the LAST emitted cycle.
__device__ double sei_degradation(const FleetDev& d, size_t i, int n, uint32_t& err) {
  const size_t EN = (size_t)d.E * d.N;
  const double* h = d.hist + i;
  double* stk = d.rf_stack + i;
  const int L = d.rf_len[i];
  const double k_sigma = 1.04, sigma_ref = 0.5, k_temp = 6.93E-2, temp_ref = 25.0;
  const double stress_temp = exp(k_temp * (d.temperature - temp_ref) * ((temp_ref + 273.15) / (d.temperature + 273.15)));

  int head = 0, tail = 0, nc = 0;
  double mean_sum = 0.0, slice_sum = 0.0, pend = 0.0, max_dod = 0.0;
  bool has_pend = false;

  auto emit = [&](double xa, double xb, double count) {
    if (has_pend) slice_sum += pend;  // the previous cycle is not the last one -> inside [L-1, len-1) if flagged
    has_pend = false;
    const double rng = fabs(xa - xb), mean = 0.5 * (xa + xb);
    if (nc >= L - 1) {
      double eff = rng * count;
      eff = eff < 0.0 ? 0.0 : (eff > 1.0 ? 1.0 : eff);
      const double s_dod = 1.0 / (1.4E5 * pow(eff, -5.01E-1) + -1.23E5);
      const double s_soc = exp(k_sigma * (mean - sigma_ref));
      pend = s_dod * s_soc * stress_temp;
      has_pend = true;
      max_dod = rng > max_dod ? rng : max_dod;
    }
    mean_sum += mean;
    ++nc;
  };
  auto push = [&](double x) {
    stk[(size_t)tail * EN] = x;
    ++tail;
    while (tail - head >= 3) {
      const double x1 = stk[(size_t)(tail - 3) * EN], x2 = stk[(size_t)(tail - 2) * EN], x3 = stk[(size_t)(tail - 1) * EN];
      const double X = fabs(x3 - x2), Y = fabs(x2 - x1);
      if (X < Y) break;
      if (tail - head == 3) {
        emit(x1, x2, 0.5);
        ++head;
      } else {
        emit(x1, x2, 1.0);
        stk[(size_t)(tail - 3) * EN] = x3;
        tail -= 2;
      }
    }
  };

  if (n >= 2) {
    double x_last = h[0], x = h[EN];
    double d_last = x - x_last;
    push(x_last);
    double x_next = 0.0;
    for (int k = 2; k < n; ++k) {
      x_next = h[(size_t)k * EN];
      if (x_next == x) continue;
      const double d_next = x_next - x;
      if (d_last * d_next < 0.0) push(x);
      x = x_next;
      d_last = d_next;
    }
    if (n > 2) push(x_next);
    while (tail - head > 1) {
      emit(stk[(size_t)head * EN], stk[(size_t)(head + 1) * EN], 0.5);
      ++head;
    }
  }

  double degradation = 0.0;
  if (nc > 0 && nc > L) {
    if (max_dod > 5.0) err |= FLEET_DEVERR_DOD_RANGE;
    const double battery_age = (double)(n - 1) * d.dt * 3600.0;  // max(End) is always the last sample
    const double mean_soc_cal = mean_sum / (double)nc;
    const double fd_cyc = d.fd_cyc[i] + slice_sum;
    const double fd_cal = (4.14E-10 * battery_age) * exp(k_sigma * (mean_soc_cal - sigma_ref)) * stress_temp;
    const double fd = fd_cyc + fd_cal;
    const double alpha = 5.75E-2, beta = 121.0;
    const double new_l = 1.0 - alpha * exp(-beta * fd) - (1.0 - alpha) * exp(-fd);
    if (new_l < 0.0) err |= FLEET_DEVERR_NEG_LIFE;
    degradation = new_l - d.sei_l[i];
    d.fd_cyc[i] = fd_cyc;
    d.fd_cal[i] = fd_cal;
    d.sei_l[i] = new_l;
    d.rf_len[i] = nc;
  }
  const double s = d.sei_soh[i] - degradation;
  d.sei_soh[i] = s;
  if (fabs(s - (1.0 - d.sei_l[i])) > 0.0001) err |= FLEET_DEVERR_SOH_MISMATCH;
  return degradation;
}

// EmpiricalDegradation.calculate_degradation for one EV (empirical_degradation.py:29-99; quirks Q1, Q5).
__device__ __forceinline__ double linear_degradation(const FleetDev& d, size_t i, int n) {
  const size_t EN = (size_t)d.E * d.N;
  const double old_soc = d.hist[(size_t)(n - 2) * EN + i], new_soc = d.hist[(size_t)(n - 1) * EN + i];
  const double avg = (old_soc + new_soc) / 2.0;
  // nearest of {0, 40, 90} to a SOC in [0,1] scale -- replicated literally (argmin, first wins ties)
  int best = 0;
  double bd = fabs(0.0 - avg);
  if (fabs(40.0 - avg) < bd) { best = 1; bd = fabs(40.0 - avg); }
  if (fabs(90.0 - avg) < bd) best = 2;
  const double cal = (best == 0 ? 0.0065 : best == 1 ? 0.0293 : 0.065) * d.dt / 8760.0;
  const double dod = fabs(new_soc - old_soc);
  const double cyc = (d.evse_power <= 22.0) ? dod * 0.000125 / 2.0 : dod * 0.000167 / 2.0;
  return cal + cyc;
}

// ---------------------------------------------------------------------------------------------------------
// reset of one env by its group (FleetEnv.reset, fleet_environment.py:330-434)
// ---------------------------------------------------------------------------------------------------------
struct EnvRegs {  // per-env scalars, tracked redundantly by every lane of the group
  int t, t_end, hist_len, episodes;
};

template <int G>
__device__ __forceinline__ void reset_env(const FleetDev& d, int e, int g, bool leader, EnvRegs& r, float* __restrict__ obs_row) {
  const int N = d.N;
  const size_t EN = (size_t)d.E * N;
  const int start = choose_start(d, e, r.episodes);
  r.t = start;
  r.t_end = start + d.episode_steps;
  r.hist_len = (d.deg_mode != FLEET_DEG_NONE) ? 1 : 0;
  for (int c = g; c < N; c += G) {
    const size_t i = (size_t)e * N + c, ti = (size_t)start * N + c;
    const double soh = 1.0 * d.init_soh;
    const double cap = soh * d.init_cap;
    double soc = d.tab_sor[ti];
    const float hl = d.tab_tl[ti];
    const double tgt = d.tgt090[i] ? 0.9 : d.target_soc;  // target_soc survives reset (quirk Q7)
    const double time_needed = (tgt - soc) * cap / d.p_avail;               // :384
    if ((hl > 0.0f) && (d.min_laxity * time_needed > (double)hl))           // :388
      soc = tgt - (time_needed * d.p_avail / cap) / d.min_laxity;           // :389-390
    const double soc_deg = (soc == 0.0) ? d.def_soc : soc;                  // :395-399
    d.soc[i] = soc;
    d.hl[i] = hl;
    d.soc_deg[i] = soc_deg;
    d.soh[i] = soh;
    if (d.deg_mode != FLEET_DEG_NONE) d.hist[i] = soc_deg;                  // :417-418 (row 0)
    if (obs_row) write_obs_ev(d, obs_row, c, start, soc, hl, tgt);
  }
  if (obs_row) write_obs_tail<G>(d, obs_row, start, g);
  if (leader) {
    d.t_idx[e] = start;
    d.t_end[e] = r.t_end;
    d.start_idx[e] = start;
    d.hist_len[e] = r.hist_len;
    d.ep_return[e] = 0.0;
    d.ep_len[e] = 0;
    d.penalty_record[e] = 0.0;
    d.done_flag[e] = 0;
    if (r.t_end > d.T - 1) d.err[e] |= FLEET_DEVERR_TABLE_END;
  }
}

template <int G>
__global__ __launch_bounds__(kBlock) void fleet_reset_kernel(FleetDev d, const uint8_t* __restrict__ mask, float* __restrict__ obs) {
  const int g = threadIdx.x % G;
  const int e = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (e >= d.E) return;
  if (mask && !mask[e]) return;
  EnvRegs r;
  r.episodes = d.episodes[e];
  // an explicit reset of an episode that is in progress abandons it: count it so the next start row differs
  if (d.ep_len[e] > 0 && !d.done_flag[e]) {
    r.episodes += 1;
    if (g == G - 1) d.episodes[e] = r.episodes;
  }
  reset_env<G>(d, e, g, g == G - 1, r, obs ? obs + (size_t)e * d.obs_dim : nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// the step (FleetEnv.step, fleet_environment.py:436-702), K consecutive steps per launch
// ---------------------------------------------------------------------------------------------------------
template <int G, int DEG, typename ActT>
__global__ __launch_bounds__(kBlock, 2) void fleet_step_kernel(FleetDev d, const ActT* __restrict__ actions, int K,
                                                            float* __restrict__ obs, double* __restrict__ reward,
                                                            uint8_t* __restrict__ done, float* __restrict__ terminal_obs,
                                                            int32_t* __restrict__ done_count) {
  const int N = d.N;
  const int lane = threadIdx.x & 63;
  const int g = threadIdx.x % G;
  const bool leader = (g == G - 1);
  const int e_raw = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  const bool env_ok = e_raw < d.E;  // surplus groups of the last block run the arithmetic on env E-1 but store nothing
  const int e = env_ok ? e_raw : d.E - 1;
  const size_t EN = (size_t)d.E * N;

  EnvRegs r;
  r.t = d.t_idx[e];
  r.t_end = d.t_end[e];
  r.hist_len = d.hist_len[e];
  r.episodes = d.episodes[e];
  double ep_return = d.ep_return[e], penalty_record = d.penalty_record[e];
  int ep_len = d.ep_len[e];
  uint32_t err = 0;
  double reward_sum = 0.0;
  int n_done = 0;
  float* const obs_row = obs + (size_t)e * d.obs_dim;
  float* const term_row = terminal_obs ? terminal_obs + (size_t)e * d.obs_dim : nullptr;

  for (int k = 0; k < K; ++k) {
    const int t = r.t;
    int t1 = t + 1;  // :508
    if (t1 > d.T - 1) { t1 = d.T - 1; err |= FLEET_DEVERR_TABLE_END; }
    const bool is_done = (t + 1 == r.t_end);  // :627-628
    const bool resets = is_done && d.auto_reset;
    // where this step's observation goes: with vec-env auto-reset the terminal observation is reported aside
    float* const step_row = resets ? term_row : obs_row;
    const bool write_step_obs = env_ok && (step_row != nullptr);

    // ---- connected cars = sum(There[t]) (ev_charger.py:138-140) ----------------------------------------------
    int connected = 0;
    for (int c0 = 0; c0 < N; c0 += G) {
      const int c = c0 + g;
      const bool th = (c < N) && (d.tab_there[(size_t)t * N + c] != 0);
      connected += group_count<G>(th, lane);
    }
    connected = connected < 1 ? 1 : connected;

    const PhysRow ph = d.tab_phys[t];
    const double pv_share = ph.pv_energy / (double)connected;  // current_pv_energy / connected_cars (:142)
    const uint8_t flags1 = d.tab_flags[t1];
    const bool lunch = d.is_caretaker && (flags1 & FLEET_TFLAG_LUNCH);
    const ActT* __restrict__ act = actions + ((size_t)k * d.E + e) * N;

    double cost = 0.0, rev = 0.0, rew = 0.0, asum = 0.0, penrec = 0.0;
    for (int c = g; c < N; c += G) {
      const size_t i = (size_t)e * N + c;
      const size_t ti = (size_t)t * N + c, ti1 = (size_t)t1 * N + c;
      const double a = (double)act[c];
      const int th = d.tab_there[ti];
      double soc = d.soc[i];
      float hl = d.hl[i];
      const double soh = d.soh[i];
      const double cap = soh * d.init_cap;
      bool t090 = d.tgt090[i] != 0;
      double tgt = t090 ? 0.9 : d.target_soc;

      // ---- EvCharger.charge (ev_charger.py:89-222) ---------------------------------------------------------
      if (a >= 0.0) {
        const double need = (tgt - soc) * cap;    // :100
        const double dem = d.p_avail * a * d.dt;  // :101
        if (dem * d.eta_c > need) {               // :104-107 (applied whether or not the EV is there, quirk Q9)
          const double x = dem - need;
          double pen = d.penalty_oc * (x * x);
          pen = pen > d.clip_oc ? pen : d.clip_oc;
          rew += pen;
        }
        double en;
        if (th == 1) {
          const double lim = need / d.eta_c;  // :114
          en = lim < dem ? lim : dem;
        } else {
          en = 0.0;
          if (fabs(a) > 0.05) rew += d.penalty_invalid * (a * a);  // :120-122
        }
        soc = soc + en * d.eta_c / cap;  // :128
        double grid_e = en - pv_share;   // :142
        grid_e = grid_e > 0.0 ? grid_e : 0.0;
        cost += grid_e * ph.spot_plus_offset * d.variable_multiplier;  // :149
        rew += ph.k_charge * grid_e;                                   // :154-156
      } else {
        const double left = -1.0 * soc * cap;     // :161
        const double dem = d.p_avail * a * d.dt;  // :162
        if ((dem * d.eta_d < left) && (th != 0)) {  // :165-167 (no clip, needs presence)
          const double x = left - dem;
          rew += d.penalty_oc * (x * x);
        }
        double en;
        if (th == 1) {
          en = left > dem ? left : dem;  // :174
        } else {
          en = 0.0;
          if (fabs(a) > 0.05) rew += d.penalty_invalid * (a * a);  // :180-182
        }
        soc = soc + en / cap;                                                 // :189
        rev += -1.0 * en * d.eta_d * ph.tariff / 1000.0 * d.one_minus_fee;    // :196-199
        rew += ph.k_discharge * en;                                           // :204-206
      }
      asum += a * (double)th;  // corrected_actions = actions * there (fleet_environment.py:491)

      // ---- arrival / departure state machine (fleet_environment.py:528-623) ----------------------------------
      const float ntl = d.tab_tl[ti1];
      if ((hl != 0.0f) && (ntl == 0.0f)) {  // a car just left :531
        const double target = lunch ? d.target_soc_lunch : tgt;  // :536-557
        const double missing = target - soc;
        if (missing > d.eps) {
          const double pen = soc_violation_penalty(missing);
          rew += pen;
          penrec += pen;  // episode.penalty_record (:549,566,584)
        } else {
          rew += d.fully_charged_reward;
        }
      }
      if ((ntl != 0.0f) && (hl != 0.0f)) {  // still charging :593-594
        hl = (float)((double)hl - d.dt);
      } else if (ntl == 0.0f) {  // no car in the next step :597-599
        hl = ntl;
        soc = d.tab_sor[ti1];
      } else {  // new arrival :602-606  (hl == 0 && ntl != 0; the reference's `else: raise` is unreachable)
        hl = ntl;
        soc = d.tab_sor[ti1];
      }
      if (soh <= 0.9) {  // :613-614 sticky target (quirk Q7)
        t090 = true;
        tgt = 0.9;
      }
      double soc_deg = d.soc_deg[i];
      if (hl != 0.0f) soc_deg = soc;  // :621-623

      if (env_ok) {
        d.soc[i] = soc;
        d.hl[i] = hl;
        d.soc_deg[i] = soc_deg;
        if (t090) d.tgt090[i] = 1;
      }
      // ---- observation of the advanced time row (fleet_environment.py:511-518, 645-652) ----------------------------
      if (write_step_obs) write_obs_ev(d, step_row, c, t1, soc, hl, tgt);

      // ---- SOC log + daily degradation (:655-673) ---------------------------------------------------------------
      if (DEG != FLEET_DEG_NONE) {
        if (r.hist_len < d.hist_cap) {
          if (env_ok) d.hist[(size_t)r.hist_len * EN + i] = soc_deg;
        } else {
          err |= FLEET_DEVERR_TABLE_END;
        }
        if ((flags1 & FLEET_TFLAG_DEG) && env_ok) {
          const int n = r.hist_len < d.hist_cap ? r.hist_len + 1 : d.hist_cap;
          const double deg = (DEG == FLEET_DEG_RAINFLOW) ? sei_degradation(d, i, n, err) : linear_degradation(d, i, n);
          d.soh[i] = soh - deg;  // :671 ; battery_cap = soh * init_cap is recomputed from soh on use (:673)
        }
      }
    }
    if (write_step_obs) write_obs_tail<G>(d, step_row, t1, g);
    if (DEG != FLEET_DEG_NONE && r.hist_len < d.hist_cap) r.hist_len += 1;

    // ---- per-env reductions; totals land in the leader lane ---------------------------------------------------
    cost = group_sum_to_last<G>(cost);
    rev = group_sum_to_last<G>(rev);
    rew = group_sum_to_last<G>(rew);
    asum = group_sum_to_last<G>(asum);
    penrec = group_sum_to_last<G>(penrec);
    r.t = t1;
    if (leader) {
      const double cashflow = -1.0 * cost + rev;  // ev_charger.py:225
      penalty_record += penrec;
      // LoadCalculation.check_violation (load_calculation.py:93) and the sigmoid penalty (:496-502)
      const double head = d.grid_connection - ph.load - asum * d.evse_power + ph.pv;
      const double over = fabs(head < 0.0 ? head : 0.0);
      if (over > 0.0) {
        const double pen = overloading_penalty(over / d.grid_connection + 1.0, d.penalty_overload);
        rew += pen;
        penalty_record += pen;
      }
      ep_return += rew;  // :637
      ep_len += 1;
      reward_sum += rew;
      if (env_ok) {
        d.cashflow[e] = cashflow;
        if (K == 1) {
          reward[e] = rew;
          done[e] = is_done ? 1 : 0;
        }
      }
    }
    if (is_done) {
      n_done += 1;
      if (leader && env_ok) {
        d.last_ep_return[e] = ep_return;
        d.last_ep_len[e] = ep_len;
        d.done_flag[e] = 1;
      }
      r.episodes += 1;
      if (resets) {
        if (env_ok) {
          reset_env<G>(d, e, g, leader, r, obs_row);
        } else {  // surplus group: keep its registers moving without touching memory
          r.t = choose_start(d, e, r.episodes);
          r.t_end = r.t + d.episode_steps;
          r.hist_len = (DEG != FLEET_DEG_NONE) ? 1 : 0;
        }
        ep_return = 0.0;
        ep_len = 0;
        penalty_record = 0.0;
      }
    }
  }

  if (leader && env_ok) {
    d.t_idx[e] = r.t;
    d.hist_len[e] = r.hist_len;
    d.episodes[e] = r.episodes;
    d.ep_return[e] = ep_return;
    d.ep_len[e] = ep_len;
    d.penalty_record[e] = penalty_record;
    if (K != 1) {
      reward[e] = reward_sum;
      if (done_count) done_count[e] = n_done;
    }
  }
  if (err && env_ok) atomicOr(&d.err[e], err);
}

// FleetEnv.get_dist_factor (fleet_environment.py:782-799)
__global__ void fleet_dist_factor_kernel(FleetDev d, double* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)d.E * d.N) return;
  const int e = (int)(i / d.N), c = (int)(i % d.N);
  const size_t ti = (size_t)d.t_idx[e] * d.N + c;
  const double th = (double)d.tab_there[ti];
  const double tgt = d.tgt090[i] ? 0.9 : d.target_soc;
  const double cl = tgt * th - d.tab_sor[ti];
  const double hn = cl * d.batt_cap_nominal / d.hn_denominator;
  out[i] = hn / ((double)d.tab_tl[ti] + 0.001);
}

int group_size(int N) {
  int G = 1;
  while (G < N && G < 64) G <<= 1;
  return G;
}

template <int G, int DEG>
hipError_t launch_step_gd(const FleetDev& d, const void* actions, int act_dtype, int K, float* obs, double* reward,
                          uint8_t* done, float* terminal_obs, int32_t* done_count, hipStream_t s) {
  const int epb = kBlock / G;
  const dim3 grid((d.E + epb - 1) / epb), block(kBlock);
  if (act_dtype == FLEET_ACT_F64)
    hipLaunchKernelGGL((fleet_step_kernel<G, DEG, double>), grid, block, 0, s, d, (const double*)actions, K, obs, reward,
                       done, terminal_obs, done_count);
  else
    hipLaunchKernelGGL((fleet_step_kernel<G, DEG, float>), grid, block, 0, s, d, (const float*)actions, K, obs, reward,
                       done, terminal_obs, done_count);
  return hipGetLastError();
}

template <int G>
hipError_t launch_step_g(const FleetDev& d, const void* actions, int act_dtype, int K, float* obs, double* reward,
                         uint8_t* done, float* terminal_obs, int32_t* done_count, hipStream_t s) {
  switch (d.deg_mode) {
    case FLEET_DEG_NONE: return launch_step_gd<G, FLEET_DEG_NONE>(d, actions, act_dtype, K, obs, reward, done, terminal_obs, done_count, s);
    case FLEET_DEG_LINEAR: return launch_step_gd<G, FLEET_DEG_LINEAR>(d, actions, act_dtype, K, obs, reward, done, terminal_obs, done_count, s);
    default: return launch_step_gd<G, FLEET_DEG_RAINFLOW>(d, actions, act_dtype, K, obs, reward, done, terminal_obs, done_count, s);
  }
}

template <int G>
hipError_t launch_reset_g(const FleetDev& d, const uint8_t* mask, float* obs, hipStream_t s) {
  const int epb = kBlock / G;
  hipLaunchKernelGGL((fleet_reset_kernel<G>), dim3((d.E + epb - 1) / epb), dim3(kBlock), 0, s, d, mask, obs);
  return hipGetLastError();
}

}  // namespace

#define FLEET_DISPATCH_G(N, CALL)          \
  switch (group_size(N)) {                 \
    case 1: return CALL(1);                \
    case 2: return CALL(2);                \
    case 4: return CALL(4);                \
    case 8: return CALL(8);                \
    case 16: return CALL(16);              \
    case 32: return CALL(32);              \
    default: return CALL(64);              \
  }

hipError_t fleet_launch_reset(const FleetDev& d, const uint8_t* mask, float* obs, hipStream_t s) {
#define CALL(Gv) launch_reset_g<Gv>(d, mask, obs, s)
  FLEET_DISPATCH_G(d.N, CALL)
#undef CALL
}

hipError_t fleet_launch_step(const FleetDev& d, const void* actions, int act_dtype, int K, float* obs, double* reward,
                             uint8_t* done, float* terminal_obs, int32_t* done_count, hipStream_t s) {
#define CALL(Gv) launch_step_g<Gv>(d, actions, act_dtype, K, obs, reward, done, terminal_obs, done_count, s)
  FLEET_DISPATCH_G(d.N, CALL)
#undef CALL
}

hipError_t fleet_launch_dist_factor(const FleetDev& d, double* out, hipStream_t s) {
  const size_t n = (size_t)d.E * d.N;
  hipLaunchKernelGGL(fleet_dist_factor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, out);
  return hipGetLastError();
}
