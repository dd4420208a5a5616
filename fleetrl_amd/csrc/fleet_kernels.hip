// fleet_kernels.hip -- the FleetEnv step / reset hot path as hand-written HIP for gfx950 (MI355X, CDNA4).
//
// What runs here (reference: /root/reference/fleetrl, float64 in the reference's operation order):
//   EvCharger.charge            utils/ev_charging/ev_charger.py:39-231
//   LoadCalculation.check_violation + ScoreConfig.overloading_penalty
//                               utils/load_calculation/load_calculation.py:83-94, fleet_env/config/score_config.py:33-41
//   arrival/departure state machine + ScoreConfig.soc_violation_penalty
//                               fleet_env/fleet_environment.py:528-623, score_config.py:26-30
//   Observer*.get_obs + Unit/OracleNormalization.normalize_obs
//                               utils/observation/observer_*.py, utils/normalization/*.py
//   LogDataDeg.log_soc, RainflowSeiDegradation / EmpiricalDegradation.calculate_degradation
//                               utils/battery_degradation/*.py  (+ third-party rainflow.extract_cycles)
//   FleetEnv.reset (incl. the vec-env auto-reset)  fleet_environment.py:330-434
//
// Mapping.  One *group* of G lanes owns one env.  Up to 256 EVs per env every EV has a lane of its own: G = the smallest power
// of two >= N up to 64 (a 64-lane wavefront holds 64/G envs, a 256-thread workgroup 256/G), and two or four whole wavefronts of
// one workgroup for 64 < N <= 256 (kMaxGroup); beyond that lane g of a one-wavefront group walks EVs g, g+64, ...  Per-env sums
// (cashflow, reward, sum(action*there), penalty record) are reduced inside the wavefront -- lane-swap folds of four quantities at
// once for a whole wavefront, DPP row shifts / row broadcasts for smaller groups -- and, for an env of several wavefronts, the
// per-wavefront partial sums meet in the LDS behind one workgroup barrier; no atomics.  Every lane of a group tracks the per-env
// scalars (time row, episode end, sample count) redundantly in registers, so nothing written by one lane is re-read by another
// inside a launch.  No MFMA: there is no contraction on this path.
//
// Schedule columns in run-length form.  What a step needs of the (time row, EV) table -- There, time_left, SOC_on_return of
// the row it advances to -- travels with the state (`run`, struct SegRec in fleet_device.h): a lane holds the record of row
// t+1 when the launch starts, so nothing it needs to start its arithmetic depends on the env's time row, and it only touches
// the table when row t+2 crosses a schedule event of its EV (about 4 % of the EV-steps), for the NEXT launch.  The row
// flags the state machine needs travel in the env head; the physics record of the time row is requested when the head
// arrives and consumed late (money terms); the four table-derived auxiliary observation slots are computed per lane from the
// carried record (one reciprocal, no division; write_obs_ev).
//
// Rainflow without a history replay.  The reference re-runs rainflow over the whole episode history every
// simulated day.  Three-point rainflow is a streaming algorithm, so the kernel keeps its state per EV (a row in HBM:
// closed-cycle count, sum of cycle means, the two newest stack entries, stress sum of the closed cycles that fall into the
// reference's slice, reversal stack; slope sign and stack size in the hot record) and feeds it ONE sample per step; the row
// is only touched by a step that pushes a reversal point, requested in the middle of the step and consumed at its end.  On the daily 14:45 row
// the forced last point and the residual half cycles are evaluated on a *virtual* copy of the stack (registers
// only), which reproduces the reference's full recount, including its cross-episode bookkeeping
// (rainflow_length, quirk Q6), at O(stack depth) instead of O(history).
#include "fleet_device.h"
#include <cstddef>
#include <cstring>

#ifdef FLEET_STAMPS
// Diagnostic build only (tools/stamps.py): s_memtime stamps of every wave (the first 4096) at fixed points of the step,
// written to a buffer nothing else reads.  Never compiled into the product library.
__device__ unsigned long long fleet_stamp_buf[4096 * 32];
#define FLEET_STAMP(k)                                                                                   \
  do {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    unsigned long long _t;                                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                           \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    const unsigned _w = blockIdx.x * (FLEET_KBLOCK / 64) + threadIdx.x / 64;                             \
    if ((threadIdx.x & 63) == 0 && _w < 4096) fleet_stamp_buf[_w * 32 + (k)] = _t;                       \
  } while (0)
// wall-clock stamps (s_memrealtime: 100 MHz, one counter for the whole chip) at a wave's entry (slot 9) and exit (slot 10):
// the launch's timeline across dies, which the per-die shader-clock stamps cannot give
#define FLEET_STAMP_RT(k)                                                                                \
  do {                                                                                                   \
    const unsigned long long _t = __builtin_amdgcn_s_memrealtime();                                      \
    const unsigned _w = blockIdx.x * (FLEET_KBLOCK / 64) + threadIdx.x / 64;                             \
    if ((threadIdx.x & 63) == 0 && _w < 4096) fleet_stamp_buf[_w * 32 + (k)] = _t;                       \
  } while (0)
// where the wavefront runs (slot 16: HW_REG_HW_ID = wave / SIMD / CU / SH / SE ids; slot 17: HW_REG_XCC_ID): does the tail of slow
// wavefronts belong to a die, a CU, a SIMD?
#define FLEET_STAMP_WHERE()                                                                              \
  do {                                                                                                   \
    const unsigned _hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));                         \
    const unsigned _xc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));                        \
    const unsigned _w = blockIdx.x * (FLEET_KBLOCK / 64) + threadIdx.x / 64;                             \
    if ((threadIdx.x & 63) == 0 && _w < 4096) {                                                          \
      fleet_stamp_buf[_w * 32 + 16] = _hw;                                                               \
      fleet_stamp_buf[_w * 32 + 17] = _xc;                                                               \
    }                                                                                                    \
  } while (0)
extern "C" int fleet_debug_read_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fleet_stamp_buf), sizeof(fleet_stamp_buf));
}
#else
#define FLEET_STAMP(k) do {} while (0)
#define FLEET_STAMP_RT(k) do {} while (0)
#define FLEET_STAMP_WHERE() do {} while (0)
#endif

namespace {

// Minimum workgroups per CU the kernels are compiled for (= waves per SIMD; register budget 512 / this).  The single-step
// kernel needs 97 VGPRs.  The multi-step kernel wants ~150; its wavefronts advance independently and are bound by their own
// dependent round trips, so what counts is that all of a 4096-env batch's wavefronts are resident at once: it is compiled for
// four per SIMD (128 VGPRs, a few dozen bytes of spills) -- +21 % env-steps/s over the three its natural register count
// allows (profiles/r03_experiments/ab_multiwaves.log).  With several EVs per lane it stays at two.
constexpr int kSingleWaves = 4, kMultiWaves = 4, kMultiWideWaves = 2;
#ifndef FLEET_KBLOCK
#define FLEET_KBLOCK 256  // (a macro only because the diagnostic stamp code above indexes its buffer with it)
#endif
constexpr int kBlock = FLEET_KBLOCK;  // threads per workgroup
// Largest lane group of one env in the single-step kernel: up to this many EVs every EV has a lane of its own.  Groups above 64
// lanes are two or four WAVEFRONTS of one workgroup: every wavefront reduces its lanes' terms as a one-wavefront env does, the
// per-wavefront partial sums meet in the LDS behind one workgroup barrier, and the group's last lane adds them up (round 5: the
// c5 shard's 200-EV envs as 4 wavefronts x 1 EV per lane instead of 1 wavefront x 4 EVs per lane walked one after the other).
constexpr int kMaxGroup = 256;
static_assert(kMaxGroup <= kBlock && kBlock % 64 == 0, "a lane group is a whole number of a workgroup's wavefronts");

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
// true in every lane of the group if `v` holds in any of its lanes (real_time event test)
template <int G>
__device__ __forceinline__ bool group_any(bool v) {
  const unsigned long long m = __ballot(v);
  if (G >= 64) return m != 0ull;  // (groups of several wavefronts never run the event-skipping loop: launch_step_gd)
  const int base = (int)(threadIdx.x % 64) & ~(G - 1);
  return ((m >> base) & ((1ull << (G % 64)) - 1ull)) != 0ull;
}

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

// Four per-env sums at once for one env per wavefront (G == 64); the totals are valid in the LAST lane.  The quantities are
// folded pairwise with the gfx950 lane-swap instructions -- after `v_permlane32_swap` one register holds the lower half's
// values of a AND b, the other the upper half's, so ONE add folds two quantities from 64 to 32 lanes; `v_permlane16_swap`
// does the same from 32 to 16 -- which leaves each quantity spread over one 16-lane row; four DPP row shifts finish the rows
// and three lane reads bring the other rows' totals to the last lane: 27 vector instructions instead of 18 per quantity.
__device__ __forceinline__ double swap_fold32(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);  // lanes 0-31: a folded, 32-63: b folded
}
__device__ __forceinline__ double swap_fold16(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);  // rows: x.r0+x.r1, y.r0+y.r1, x.r2+x.r3, y.r2+y.r3
}
__device__ __forceinline__ double lane_read(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ void wave_sum4_to_last(double& a, double& b, double& c, double& dd) {
  double z = swap_fold16(swap_fold32(a, b), swap_fold32(c, dd));  // row 0: a, row 1: c, row 2: b, row 3: dd (16 partial sums each)
  z = dpp_add<0x111, 0xF>(z);
  z = dpp_add<0x112, 0xF>(z);
  z = dpp_add<0x114, 0xF>(z);
  z = dpp_add<0x118, 0xF>(z);  // row totals in lanes 15, 31, 47, 63
  a = lane_read(z, 15);
  c = lane_read(z, 31);
  b = lane_read(z, 47);
  dd = z;  // the last lane's own row
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

__device__ __forceinline__ int choose_start(const FleetCold* cd, int E, int e, int episode) {
  if (cd->sched_n > 0) return cd->sched[(size_t)(episode % cd->sched_n) * E + e];
  int k = cd->start_lo;
  if (cd->picker_mode != FLEET_PICK_STATIC) {
    const uint32_t range = (uint32_t)(cd->start_hi - cd->start_lo + 1);
    const uint32_t x = philox_start(cd->seed, (uint32_t)(cd->env_id_offset + e), (uint32_t)episode);
    k += (int)__umulhi(x, range);
  }
  return cd->pick_rows ? cd->pick_rows[k] : k;  // candidate list of the pickers' date_range on an irregular grid
}

// 1 / x for the auxiliary slots' one division: hardware reciprocal seed (v_rcp_f64, ~26 good bits) + two Newton steps = full
// float64 accuracy (<= 1 ulp) in five instructions, against ~14 of the IEEE division sequence with its special-case handling.
__device__ __forceinline__ double rcp_newton(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
// The same with ONE Newton step: relative error <= ~2^-44 (the seed is good to ~2^-23).  Enough wherever the result is rounded
// to float32 afterwards or feeds div_rcp's residual correction (which squares the reciprocal's error once more).
__device__ __forceinline__ double rcp_newton1(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
}
// x / c, IEEE-correctly rounded, from a reciprocal rc ~ 1 / c (Markstein's residual correction): q0 = x * rc is within a few
// ulp of the quotient, e = x - q0 * c is EXACT in one fma, and q0 + e * rc is the quotient to a relative 2 * |rc * c - 1|^2
// (2^-104 for a correctly rounded rc, 2^-87 for rcp_newton1) before the final rounding -- i.e. the correctly rounded quotient
// unless x / c lies that close to a rounding boundary, which no pair of float64 operands of these magnitudes does in practice
// (tests/test_capi_gpu.py::test_division_by_reciprocal_is_bit_exact: 2^30 operand pairs of the charge arithmetic's ranges
// against the IEEE sequence, 0 differences).  `v_div_fixup` restores what the three fmas lose at the edges (x = +-0, inf, NaN,
// c = 0): 5 vector instructions instead of the 11 of the IEEE division sequence, none of them quarter-rate.
__device__ __forceinline__ double div_rcp(double x, double c, double rc) {
  const double q0 = x * rc;
  const double e = fma(-q0, c, x);
  return __builtin_amdgcn_div_fixup(fma(e, rc, q0), c, x);
}

// ScoreConfig.soc_violation_penalty (score_config.py:26-30)
__device__ __forceinline__ double soc_violation_penalty(double missing) {
  return -500.0 * rcp_newton(1.0 + exp(-16.48461585 * (missing - 0.29229767))) + 1.0;  // (<= 2 ulp of the quotient)
}

// ScoreConfig.overloading_penalty (score_config.py:33-41)
__device__ __forceinline__ double overloading_penalty(double rel, double scale) {
  const double pen = (rel < 1.1) ? 0.0 : -700.0 / (1.0 + exp(-15.77350877 * (rel - 1.33298382)));
  return pen * scale;
}

// ---------------------------------------------------------------------------------------------------------
// observation assembly (observer_*.py + normalization/*.py); layout: DESIGN.md "Observation row"
// ---------------------------------------------------------------------------------------------------------
typedef float fleet_v4f __attribute__((ext_vector_type(4)));
// Observation rows are written once and read by nobody on the chip: non-temporal stores (-1.5 % per launch, r03 ab_nt.log).
// Everything else is stored plain: write-through (`sc1`) and non-temporal state stores were measured on every class of store
// and lose everywhere (profiles/r03_experiments/ab_stores.log).
__device__ __forceinline__ void st_obs(float* p, float v) { __builtin_nontemporal_store(v, p); }
template <typename T>
__device__ __forceinline__ void st_rec16(T* p, const T& v) {  // a 16-byte record as ONE store
  static_assert(sizeof(T) == 16, "16-byte record");
  fleet_v4f w;
  __builtin_memcpy(&w, &v, 16);
  *reinterpret_cast<fleet_v4f*>(p) = w;
}
// base + 32-bit byte offset.  The offset is made opaque at every use: its 64-bit zero-extension must be formed in the basic
// block of the access for the instruction selector to see "uniform base + 32-bit lane offset" (scalar-base addressing); a
// zero-extension hoisted into an earlier block arrives as an anonymous 64-bit vector value and costs a 64-bit vector add.
// (in place: the caller's variable is the one register all its uses share)
template <typename T>
__device__ __forceinline__ T* at_off(T* base, unsigned& byte_off) {
  asm volatile("" : "+v"(byte_off));
  return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ const T* at_off(const T* base, unsigned& byte_off) {
  asm volatile("" : "+v"(byte_off));
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ void st_plain(T* p, const T& v) { *p = v; }
__device__ __forceinline__ void st_obs_at(float* base, unsigned& byte_off, float v) { st_obs(at_off(base, byte_off), v); }

// One EV of one env: the planes [E, N] are addressed as (plane + e * N) + c -- the first part is wave-uniform when a
// wavefront is one env (G == 64) and lives in scalar registers.
struct EvIx {
  size_t eN;   // e * N
  unsigned c;  // EV of the env
  __device__ __forceinline__ size_t flat() const { return eN + c; }
};
template <typename T>
__device__ __forceinline__ T* ev_at(T* plane, const EvIx& ix) {
  unsigned off = ix.c * (unsigned)sizeof(T);
  return at_off(plane + ix.eN, off);
}
// the EV's rainflow row (row stride in float64 words; one env's rows stay below 4 GiB: fleet_create checks)
__device__ __forceinline__ double* rf_row_of(const FleetDev& d, const EvIx& ix, unsigned word = 0) {
  unsigned off = (ix.c * (unsigned)d.rf_row_stride + word) * 8u;
  return at_off(d.rf_rows + ix.eN * (size_t)d.rf_row_stride, off);
}

// The three schedule columns of one (row, EV), decoded from the record of the row's segment.
struct RowRec {
  double sor;      // db["SOC_on_return"]
  float tl;        // db["time_left"]
  uint32_t there;  // db["There"]
};
__device__ __forceinline__ RowRec seg_row(const SegRec& s, int r, double dt) {
  RowRec o;
  o.sor = s.sor;
  o.tl = seg_tl(s, r, dt);
  o.there = SEG_THERE(s.se);
  return o;
}

// Per-EV slots of EV c.  soc / hours_left come from live state; the five auxiliary slots from the TABLE row the step
// advanced to (quirk Q10): there | target_soc * there | charging_left | hours_needed | laxity (observer_bl_pv.py:85-91), each
// divided by the normaliser's constant when normalize_in_env (oracle_normalization.py:127-131).  They are computed per lane
// from the carried schedule record in float64 and rounded to float32 like the reference's; `cl * cap / (evse * eta)` and the
// normaliser's divisions are multiplications by the correctly rounded quotient / reciprocal and `time_left / (hours_needed +
// 0.001)` uses rcp_newton: <= 2 ulp of float64 before the rounding to float32, i.e. the float32 word is the reference's
// except when the float64 value lies within ~2e-16 relative of a rounding boundary (tests/test_hip_parity.py reports the
// exact-match fraction; the north-star tolerance is 1e-5).  Round 3 read these four words from a [T, N] table: 16 bytes per EV
// and step, 7 of a wavefront's 44 line requests; the two float64 divisions that had made the per-lane form lose in round 3
// (ab_seg3.log) are gone.
__device__ __forceinline__ void write_obs_ev(const FleetDev& d, float* __restrict__ row, int c, double soc, float hl, double tgt,
                                             const RowRec& tb) {
  const int N = d.N;
  // one 32-bit lane offset for all seven slots; the slot arrays' bases are wave-uniform when a wavefront is one env (scalar
  // registers, `global_store ... s[base]` addressing: no 64-bit vector address per slot)
  unsigned o4 = (unsigned)c * 4u;
  st_obs_at(row, o4, (float)soc);
  st_obs_at(row + N, o4, d.normalize ? (float)((double)hl / d.self->max_time_left) : hl);
  if (!d.aux) return;
  float* a = row + 2 * N + d.tail_a_len;
  const double th = (double)tb.there;
  const double tgt_th = tgt * th;
  const double cl = tgt_th - tb.sor;
  const double hn = cl * d.hn_scale;
  double lax = ((double)tb.tl * rcp_newton1(hn + 0.001) - 1.0) * th;
  lax = lax < 0.0 ? 0.0 : (lax > 5.0 ? 5.0 : lax);  // np.clip(., 0, 5): keeps -0.0 (an absent EV) and NaN like numpy does
  st_obs_at(a, o4, (float)tb.there);
  if (d.normalize) {
    const FleetCold* cd = d.self->cold;
    st_obs_at(a + N, o4, (float)(tgt_th * cd->inv_max_soc));
    st_obs_at(a + 2 * N, o4, (float)(cl * cd->inv_max_soc));
    st_obs_at(a + 3 * N, o4, (float)(hn * cd->inv_max_hours_needed));
    st_obs_at(a + 4 * N, o4, (float)(lax * cd->inv_max_laxity));
  } else {
    st_obs_at(a + N, o4, (float)tgt_th);
    st_obs_at(a + 2 * N, o4, (float)cl);
    st_obs_at(a + 3 * N, o4, (float)hn);
    st_obs_at(a + 4 * N, o4, (float)lax);
  }
}

// env-level blocks: a pure function of the table row, pre-assembled (and pre-normalised) on the host.  Lane j copies
// tail float j to its slot (block A right after the 2N state slots, block B after the 5N auxiliary slots).  The load
// is issued early (with the other time-row loads) and the store late: `tail_load` / `tail_store`; for the usual
// sizes (<= G floats) that is one predicated load and one store per lane, no loop.
template <int G>
__device__ __forceinline__ float tail_load(const FleetDev& d, int t, int g) {
  const int total = d.tail_a_len + d.tail_b_len;
  return (g < total) ? d.tab_tail[(size_t)t * d.tail_stride + g] : 0.0f;
}

template <int G>
__device__ __forceinline__ void tail_store(const FleetDev& d, float* __restrict__ row, int t, int g, float first) {
  const float* __restrict__ src = d.tab_tail + (size_t)t * d.tail_stride;
  const int na = d.tail_a_len, total = d.tail_a_len + d.tail_b_len;
  const unsigned base_a = 2u * (unsigned)d.N, base_b = 7u * (unsigned)d.N;  // block B: 2N + na + 5N + (j - na) = 7N + j
  int j = g;
  if (j < total) st_obs(row + ((j < na ? base_a : base_b) + (unsigned)j), first);
  if (total > G) {
    for (j += G; j < total; j += G) row[(j < na ? base_a : base_b) + (unsigned)j] = src[j];
  }
}

template <int G>
__device__ __forceinline__ void write_obs_tail(const FleetDev& d, float* __restrict__ row, int t, int g) {
  tail_store<G>(d, row, t, g, tail_load<G>(d, t, g));
}

// ---------------------------------------------------------------------------------------------------------
// battery degradation
// ---------------------------------------------------------------------------------------------------------
// exp(z) by its Taylor polynomial of degree 14: truncation < 2e-15 relative for |z| <= 0.55 (a mean SOC in [0, 1]) and
// < 8e-13 for |z| <= 1; beyond that -- a mean SOC far outside [0, 1], which the reference does not clip (quirk Q9) and a
// schedule whose trips use more than a battery charge can produce -- the library exp takes over.
__device__ __forceinline__ double exp_small(double z) {
  if (fabs(z) > 1.0) return exp(z);
  double r = 1.0 / 87178291200.0;  // 1/14!
  r = fma(r, z, 1.0 / 6227020800.0);
  r = fma(r, z, 1.0 / 479001600.0);
  r = fma(r, z, 1.0 / 39916800.0);
  r = fma(r, z, 1.0 / 3628800.0);
  r = fma(r, z, 1.0 / 362880.0);
  r = fma(r, z, 1.0 / 40320.0);
  r = fma(r, z, 1.0 / 5040.0);
  r = fma(r, z, 1.0 / 720.0);
  r = fma(r, z, 1.0 / 120.0);
  r = fma(r, z, 1.0 / 24.0);
  r = fma(r, z, 1.0 / 6.0);
  r = fma(r, z, 0.5);
  r = fma(r, z, 1.0);
  r = fma(r, z, 1.0);
  return r;
}

// x^(-0.501) for 0 < x <= 1, as x^(-1/2) * exp(-0.001 * ln x):
//   x^(-1/2): hardware reciprocal-square-root seed + two Newton steps (full float64 accuracy);
//   ln x    : hardware float32 log2 (relative error ~1e-7, i.e. <= 4e-6 absolute for x >= 1e-17); multiplied by
//             0.001 that leaves <= 4e-9 relative error in the result; exp of an argument <= 0.04 by Taylor.
// The library pow() would be exact to 1 ulp but costs several hundred instructions inside a divergent branch.
__device__ __forceinline__ double pow_m0501(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  const double lnx = (double)(__builtin_amdgcn_logf((float)x)) * 0.6931471805599453;  // log2 -> ln
  const double z = -0.001 * lnx;
  double r = 1.0 / 720.0;
  r = fma(r, z, 1.0 / 120.0);
  r = fma(r, z, 1.0 / 24.0);
  r = fma(r, z, 1.0 / 6.0);
  r = fma(r, z, 0.5);
  r = fma(r, z, 1.0);
  r = fma(r, z, 1.0);
  return y * r;
}

// stress of one rainflow cycle: deg_rate_cycle(dod, avg_soc, temp) (rainflow_sei_degradation.py:68-80) for
// effective_dod = clip(range*count, 0, 1) (:170).  Relative accuracy ~1e-8 (see pow_m0501), which moves SoH by
// < 1e-12 relative (the degradation is a 1e-5-sized correction to 1.0); DESIGN.md "Numerics".
__device__ __forceinline__ double cycle_stress(double rng, double mean, double count, double stress_temp) {
  double eff = rng * count;
  eff = eff > 1.0 ? 1.0 : eff;
  if (!(eff > 0.0)) return 0.0;  // pow(0, -0.501) = inf -> 1/inf = 0
  const double s_dod = 1.0 / (1.4E5 * pow_m0501(eff) + -1.23E5);   // (kd1 * dod**kd2 + kd3) ** -1
  const double s_soc = exp_small(1.04 * (mean - 0.5));              // e ** (k_sigma * (soc - sigma_ref)), |arg| <= 0.55
  return s_dod * s_soc * stress_temp;
}

// A real reversal point `p` arrives (rainflow.reversals yielded it): push it and close every cycle the
// three-point rule allows (rainflow.extract_cycles, the `while len(points) >= 3` loop).
// The stack of the EV always starts at slot 0 (`tail` = its size; when the three-point rule drops the FIRST point -- the
// stack is exactly [a, b, p] then -- the survivor below the top is rewritten to slot 0, so no head index exists and the size
// alone describes it).  Its newest entry lives in the row header only (s2; s1 caches the one below), the entries below it in
// the stack words behind the header (struct RfHdr in fleet_device.h).
// The push is split in two so that its memory round trip hides behind the rest of the step: `rf_begin`, right after the
// state machine, knows the new sample and therefore whether a reversal point is pushed, and REQUESTS the EV's row (header
// head, stack top, the two entries below the top two: three 16-byte loads of one cache line); `rf_finish`, after the
// observation stores and the money terms, consumes it.  A step that pushes nothing -- three in four -- never touches the row.
struct RfReq {
  double p;        // the reversal point to push
  RfAccHead acc;   // requested when a point is pushed
  RfTop top;       // stack[tail-2], stack[tail-1]
  double w0, w1;   // stack[tail-3], [tail-4] (before the push)
  bool push;
  bool win;        // w0 / w1 were requested (else the pops read the stack words)
};
// `early`: the row's header and the entries below the top two were already requested at the start of the EV's step (K steps
// per launch: the same row lines serve all K steps of the launch from the cache, and a wavefront that advances on its own is
// bound by its own dependent round trips, which this removes from every step that pushes).
__device__ __forceinline__ void rf_request(const FleetDev& d, const EvIx& i, int tail, RfReq& q) {
  const double* row = rf_row_of(d, i);
  q.acc = *reinterpret_cast<const RfAccHead*>(row);
  q.top = *reinterpret_cast<const RfTop*>(row + 2);
  // stack[tail-4], stack[tail-3]; for a shallow stack they fall into the row's own header (never used: `nwin`)
  const double* w = rf_row_of(d, i, (unsigned)(RF_HDR_WORDS + tail - 4));  // tail >= 1
  q.w1 = w[0];
  q.w0 = w[1];
  q.win = true;
}
__device__ __forceinline__ void rf_begin(const FleetDev& d, const EvIx& i, double old_deg, double soc_deg, int tail, int& sgn, RfReq& q,
                                         bool early = false) {
  q.push = false;
  q.p = old_deg;
  // rainflow.reversals, one sample per step: equal samples are skipped, a strict slope sign change makes the previous
  // sample a reversal point
  if (soc_deg != old_deg) {
    const int s_next = (soc_deg > old_deg) ? 1 : 2;
    q.push = (sgn != 0 && sgn != s_next);
    sgn = s_next;
  }
  if (q.push && !early) rf_request(d, i, tail, q);
}
// `top`: the stack top after the push (only written when a point was pushed)
// `acc_out`: the accumulator head after the push (only written when the push closed a cycle)
// Shape (round 5): the push that closes nothing, the half cycle and the FIRST full cycle are one straight line of selects -- the
// only memory they need is what rf_request brought (top two entries, the two below, the accumulators) -- and only a second closure
// of the same push (the new top against what lies below: rare) enters a loop that reads stack words.  The general loop of rounds
// 2-4 walked every push through its loop control and the first-point test: 8.3 -> 7.9 us per 4096x50 launch for the same
// algorithm (profiles/r05_experiments/ab7_rf_finish_peeled.log).
__device__ __forceinline__ void rf_finish(const FleetDev& d, const EvIx& i, const RfReq& q, int& tail, RfTop& top, RfAccHead& acc_out,
                                          uint32_t& err) {
  if (!q.push) return;
  double* row = rf_row_of(d, i);
  double* stk = row + RF_HDR_WORDS;
  if (tail >= d.stack_cap) {  // cannot happen (pushes <= samples < stack_cap); refuse instead of overrunning
    err |= FLEET_DEVERR_TABLE_END;
    return;
  }
  const double p = q.p;
  const double a0 = q.top.s1, b0 = q.top.s2;  // stack[tail-2] (also in the stack words), stack[tail-1] (only in the header)
  const int size = tail + 1;                  // points on the stack with p pushed
  const bool closes = (size >= 3) && !(fabs(p - b0) < fabs(b0 - a0));
  const bool half = closes && (size == 3);    // Y contains the starting point: half cycle, the first point is dropped
  // ONE store for b0: it joins the stack words when nothing closes (slot tail-1) and is rewritten to slot 0 when the first point
  // is dropped; a full cycle leaves the stack words as they are
  if (!closes || half) st_plain(rf_row_of(d, i, (unsigned)(RF_HDR_WORDS + (half ? 0 : tail - 1))), b0);
  const int L = q.acc.rf_len;
  int nc = q.acc.nc;
  double mean_sum = q.acc.mean_sum, dcsum = 0.0;
  bool has_csum = false;
  double a = a0, b = b0;
  int t = size;
  if (closes) {
    if (nc >= L - 1) {  // only the closed cycles beyond the last evaluation's count carry stress: none in the steady state
      dcsum = cycle_stress(fabs(a0 - b0), 0.5 * (a0 + b0), half ? 0.5 : 1.0, d.self->stress_temp);
      has_csum = true;
    }
    mean_sum += 0.5 * (a0 + b0);
    nc += 1;
    if (half) {
      t = 2;  // stack = [b0, p]
    } else {  // full cycle: its two points vanish, p lives in s2, the entries below come from the request
      t = size - 2;
      const int nwin = q.win ? (tail - 2 > 2 ? 2 : tail - 2) : 0;
      b = (nwin >= 1) ? q.w0 : stk[t - 2];
      a = (t >= 3) ? ((nwin >= 2) ? q.w1 : stk[t - 3]) : 0.0;
      while (t >= 3) {  // further closures of the new top against what lies below: rare, from the stack words
        if (fabs(p - b) < fabs(b - a)) break;
        if (nc >= L - 1) {
          dcsum += cycle_stress(fabs(a - b), 0.5 * (a + b), (t == 3) ? 0.5 : 1.0, d.self->stress_temp);
          has_csum = true;
        }
        mean_sum += 0.5 * (a + b);
        nc += 1;
        if (t == 3) {
          stk[0] = b;
          t = 2;
        } else {
          t -= 2;
          b = stk[t - 2];
          a = (t >= 3) ? stk[t - 3] : 0.0;
        }
      }
    }
  }
  tail = t;
  top.s1 = b;  // stack[tail-2]
  top.s2 = p;  // stack[tail-1]
  st_plain(reinterpret_cast<RfTop*>(row + 2), top);
  if (closes) {
    RfAccHead out;
    out.mean_sum = mean_sum;
    out.nc = nc;
    out.rf_len = L;
    acc_out = out;
    st_plain(reinterpret_cast<RfAccHead*>(row), out);
    if (has_csum) reinterpret_cast<RfHdr*>(row)->csum += dcsum;
  }
}
// RainflowSeiDegradation.calculate_degradation for one EV on the daily row (rainflow_sei_degradation.py:91-212).
// `v` = the sample just logged (forced last reversal), `n` = number of logged samples.  The forced point and the
// residual half cycles are evaluated on a virtual stack (vt, vh, registers a/b); nothing of the streaming state
// is modified except rainflow_length / fd_cyc / fd_cal / l / csum when the reference would update them.
// `top` / `have_top`: the stack top when this step's push has just written it (registers are newer than the row).
__device__ __forceinline__ double sei_evaluate(const FleetDev& d, const EvIx& ix, double v, int n, int tail, const RfTop& top, bool have_top,
                                             uint32_t& err, double dt_hours, int* new_len = nullptr) {
  const size_t i = ix.flat();
  double* row = d.rf_rows + i * (size_t)d.rf_row_stride;
  const double* stk = row + RF_HDR_WORDS;
  // everything this needs from memory is requested up front (one round trip)
  const RfHdr hd = *reinterpret_cast<const RfHdr*>(row);
  SeiRec sr = d.sei[i];
  const int L = hd.rf_len;
  const int nc = hd.nc;
  const double mean_sum0 = hd.mean_sum, csum0 = hd.csum, fd_cyc0 = sr.fd_cyc, sei_l0 = sr.sei_l, sei_soh0 = sr.sei_soh;
  const double st = d.stress_temp;
#ifdef FLEET_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  FLEET_STAMP(11);  // records arrived

  // The walk below emits at most one cycle per stack entry (`tail` entries and the forced last point make at most `tail`
  // ranges), and the model is only updated when the cycle count passes rainflow_length (`len > L` below): an EV whose closed
  // cycles plus stack entries stay within it -- typically the first 14:45 row of an episode, whose rainflow_length still is the
  // previous episode's (quirk Q6) -- gets no update whatever the walk finds: degradation 0, records as they are.  Nothing to
  // walk, nothing to store.  (These wavefronts end their launch: -1.3 % per launch at 4096 x 50, -16 % at 2048 x 50.)
  if (nc + tail <= L) return 0.0;
  int nv = 0;
  double vmean = 0.0, vsum = 0.0, pend = 0.0, max_dod = 0.0;
  bool has_pend = false;
  auto emit = [&](double x1, double x2, double count) {
    if (has_pend) vsum += pend;  // the previous cycle is not the last one
    has_pend = false;
    const double rng = fabs(x1 - x2), mean = 0.5 * (x1 + x2);
    if (nc + nv >= L - 1) {
      pend = cycle_stress(rng, mean, count, st);
      has_pend = true;
      max_dod = rng > max_dod ? rng : max_dod;
    }
    vmean += mean;
    nv += 1;
  };
  if (n >= 3) {  // with two samples rainflow.reversals yields only the first point: no cycle at all
    int vt = tail, vh = 0;
    int size = vt - vh + 1;
    double a = have_top ? top.s1 : hd.s1, b = have_top ? top.s2 : hd.s2;
    while (size >= 3) {
      const double X = fabs(v - b), Y = fabs(b - a);
      if (X < Y) break;
      emit(a, b, (size == 3) ? 0.5 : 1.0);
      if (size == 3) {
        vh += 1;
        size = 2;
      } else {
        vt -= 2;
        size -= 2;
        b = stk[vt - 1];
        a = (size >= 3) ? stk[vt - 2] : 0.0;
      }
    }
    // remaining ranges are half cycles: stack[vh..vt) followed by the forced point
    double prev = (vt - vh >= 2) ? stk[vh] : b;
    for (int j = vh + 1; j < vt; ++j) {
      const double cur = (j == vt - 1) ? b : stk[j];
      emit(prev, cur, 0.5);
      prev = cur;
    }
    emit(b, v, 0.5);
  }

  FLEET_STAMP(12);  // stack walked, cycle stresses evaluated
  double degradation = 0.0;
  double sei_l = sei_l0;
  const int len = nc + nv;
  if (len > 0 && len > L) {
    if (max_dod > 5.0) err |= FLEET_DEVERR_DOD_RANGE;
    const double battery_age = (double)(n - 1) * dt_hours * 3600.0;  // max(End) is always the last sample's index
    const double mean_soc_cal = (mean_sum0 + vmean) / (double)len;
    const double fd_cyc = fd_cyc0 + (csum0 + vsum);
    const double fd_cal = (4.14E-10 * battery_age) * exp(1.04 * (mean_soc_cal - 0.5)) * st;
    const double fd = fd_cyc + fd_cal;
    const double alpha = 5.75E-2, beta = 121.0;
    sei_l = 1.0 - alpha * exp(-beta * fd) - (1.0 - alpha) * exp(-fd);
    if (sei_l < 0.0) err |= FLEET_DEVERR_NEG_LIFE;
    degradation = sei_l - sei_l0;
    sr.fd_cyc = fd_cyc;
    sr.fd_cal = fd_cal;
    sr.sei_l = sei_l;
    RfAccHead out;  // rainflow_length moves on; every closed cycle so far now lies below the new rainflow_length-1
    out.mean_sum = mean_sum0;
    out.nc = nc;
    out.rf_len = len;
    *reinterpret_cast<RfAccHead*>(row) = out;
    reinterpret_cast<RfHdr*>(row)->csum = 0.0;
    if (new_len) *new_len = len;
  }
  FLEET_STAMP(13);  // SEI model evaluated
  const double s = sei_soh0 - degradation;
  sr.sei_soh = s;
  d.sei[i] = sr;
  if (fabs(s - (1.0 - sei_l)) > 0.0001) err |= FLEET_DEVERR_SOH_MISMATCH;
  return degradation;
}

// EmpiricalDegradation.calculate_degradation for one EV (empirical_degradation.py:29-99; quirks Q1, Q5):
// the last two log entries are the SOC sample before and after this step.
__device__ __forceinline__ double linear_degradation(const FleetDev& d, double old_soc, double new_soc, double dt_hours) {
  const double avg = (old_soc + new_soc) / 2.0;
  // nearest of {0, 40, 90} to a SOC on a [0,1] scale -- replicated literally (argmin, first wins ties)
  int best = 0;
  double bd = fabs(0.0 - avg);
  if (fabs(40.0 - avg) < bd) { best = 1; bd = fabs(40.0 - avg); }
  if (fabs(90.0 - avg) < bd) best = 2;
  const double cal = (best == 0 ? 0.0065 : best == 1 ? 0.0293 : 0.065) * dt_hours / 8760.0;
  const double dod = fabs(new_soc - old_soc);
  const double cyc = (d.evse_power <= 22.0) ? dod * 0.000125 / 2.0 : dod * 0.000167 / 2.0;
  return cal + cyc;
}

// The hot record of an EV whose soc / soc_deg / hours_left are given (struct Hot in fleet_device.h): the shared float64
// field, the FROZEN / INPLANE flags, and the soc_deg plane entry in the one case that needs it.  `plane_has` = the
// plane already holds this soc_deg (the EV was INPLANE before and soc_deg has not changed since).
__device__ __forceinline__ Hot hot_encode(const FleetDev& d, const EvIx& i, double soc, double soc_deg, float hl, int tail, int sgn,
                                          uint32_t there, bool t090, bool plane_has) {
  Hot h;
  h.hl = hl;
  bool frozen = false, inplane = false;
  h.x = soc;
  if (__double_as_longlong(soc_deg) != __double_as_longlong(soc)) {
    frozen = true;
    if (__double_as_longlong(soc) == 0ll) {
      h.x = soc_deg;  // soc == +0.0 is implied
    } else {
      inplane = true;
      if (!plane_has) d.soc_deg[i.flat()] = soc_deg;
    }
  }
  h.bits = HOT_PACK(tail, sgn, frozen, inplane, there, t090);
  return h;
}

// ---------------------------------------------------------------------------------------------------------
// reset of one env by its group (FleetEnv.reset, fleet_environment.py:330-434)
// ---------------------------------------------------------------------------------------------------------
// FleetEnv.reset in two parts -- what each EV of the env does for itself, and what is done once per env -- so that both lane
// parts can be placed independently (reset_env below runs them for a group of G lanes; round 5's flat-mapping experiment ran the
// env's part on another thread than its EVs').
// The EV's part (fleet_environment.py:345-399): state of health, SOC / hours_left from the start row, laxity fix-up, first SOC
// sample, the carried schedule record, the observation slots.  `log_obs_row` / `log_ev_soh`: the data log's row reset() writes.
__device__ __forceinline__ void reset_ev(const FleetDev& d, int e, int c, int start, float* __restrict__ obs_row,
                                         float* __restrict__ log_obs_row, double* __restrict__ log_ev) {
  const int N = d.N;
  const FleetCold* cd = d.cold;
  const int next = start + 1 > d.T - 1 ? d.T - 1 : start + 1;
  const EvIx ix = {(size_t)e * N, (unsigned)c};
  const size_t i = ix.flat();
  const SegRec s0 = d.seg[(size_t)start * N + c];
  d.run[i] = d.seg[(size_t)next * N + c];  // the record the first step of the episode advances to
  const RowRec tb = seg_row(s0, start, d.dt);
  const bool t090 = HOT_T090(d.hot[i].bits);  // target_soc survives reset (quirk Q7)
  const double soh = 1.0 * cd->init_soh;
  const double cap = soh * d.init_cap;
  double soc = tb.sor;
  const float hl = tb.tl;
  const double tgt = t090 ? 0.9 : d.target_soc;
  const double time_needed = (tgt - soc) * cap / d.p_avail;              // :384
  if ((hl > 0.0f) && (cd->min_laxity * time_needed > (double)hl))        // :388
    soc = tgt - (time_needed * d.p_avail / cap) / cd->min_laxity;        // :389-390
  const double soc_deg = (soc == 0.0) ? cd->def_soc : soc;               // :395-399
  d.hot[i] = hot_encode(d, ix, soc, soc_deg, hl, 1, 0, tb.there, t090, false);  // rainflow: the first sample is the first reversal point
  d.soh[i] = soh;
  if (d.deg_mode == FLEET_DEG_RAINFLOW) {  // LogDataDeg restarts; the SEI bookkeeping does NOT (quirk Q6)
    RfHdr* hp = reinterpret_cast<RfHdr*>(d.rf_rows + i * (size_t)d.rf_row_stride);
    RfHdr hd = *hp;  // rainflow_length survives
    hd.mean_sum = 0.0;
    hd.csum = 0.0;
    hd.nc = 0;
    hd.s1 = 0.0;
    hd.s2 = soc_deg;  // the stack is [soc_deg]: its only entry lives in the header
    *hp = hd;
  }
  if (obs_row) write_obs_ev(d, obs_row, c, soc, hl, tgt, tb);
  if (log_obs_row) {
    write_obs_ev(d, log_obs_row, c, soc, hl, tgt, tb);
    double* lev = log_ev + c;
    lev[0] = 0.0;
    lev[N] = 0.0;
    lev[2 * N] = 0.0;
    lev[3 * N] = soh;
  }
}
// Start row, finish row and sample count of the env's next episode (time pickers, :351-355) -- every lane of the env computes
// them for itself (registers, no exchange).
// `rf_until`: the last row of the new episode on which the degradation model runs (EnvRec::rf_until).
__device__ __forceinline__ int reset_times(const FleetDev& d, int e, EnvHead& r, int& rf_until) {
  const FleetCold* cd = d.cold;
  const int start = choose_start(cd, d.E, e, r.episodes);
  r.t = start;
  r.t_end = d.tab_finish ? d.tab_finish[start] : start + d.episode_steps;  // :355 (exact date match on an irregular grid)
  r.nsamp = (d.deg_mode != FLEET_DEG_NONE) ? 1 : 0;
  // the degradation model is evaluated on the rows (start, t_end] that carry FLEET_TFLAG_DEG; what is logged after the last of
  // them is cleared by the next reset() unread
  const int last = cd->tab_last_deg[r.t_end > d.T - 1 ? d.T - 1 : r.t_end];
  rf_until = cd->rf_count_all ? INT32_MAX : (last > start ? last : -1);
  return start;
}
// The env's part: its record (episode counters zeroed :402-404, the head with the row flags the episode's first step needs).
__device__ __forceinline__ void reset_head(const FleetDev& d, int e, const EnvHead& r, int start, int rf_until) {
  EnvRec* er = d.env + e;
  EnvHead hd = r;
  hd.nsamp = HEAD_PACK(r.nsamp, d.tab_phys[start].flags_next, start < rf_until);
  er->h = hd;
  er->rf_until = rf_until;
  er->ep_return = 0.0;
  er->ep_len = 0;
  er->penalty_record = 0.0;
  er->start_done = start;  // bit 31 (episode.done) cleared
  // (an episode whose finish row lies beyond the table is legal until a step leaves the table: FLEET_DEVERR_TABLE_END is raised
  // there, like the KeyError of the reference's `db.loc[...]`)
}

// `lp`: the env's data-log cursor (rows written so far; only used when the log is on), advanced by the row reset() writes.
template <int G, bool LOG>
__device__ __forceinline__ void reset_env(const FleetDev& d, int e, int g, bool leader, EnvHead& r, float* __restrict__ obs_row, int& lp,
                                          int& rf_until) {
  const bool log_on = LOG && (d.log_pos != nullptr);
  const int N = d.N;
  const int start = reset_times(d, e, r, rf_until);
  // data log: the row reset() writes -- time, observation and SoH, zeros for everything else (:420-432)
  const size_t lrow = log_on ? (size_t)(lp % d.log_cap) * d.E + e : 0;
  float* const log_obs_row = log_on ? d.log_obs + lrow * d.obs_dim : nullptr;
  double* const log_ev = log_on ? d.log_ev + lrow * 4 * N : nullptr;
  for (int c = g; c < N; c += G) reset_ev(d, e, c, start, obs_row, log_obs_row, log_ev);
  if (obs_row) write_obs_tail<G>(d, obs_row, start, g);
  if (log_on) {
    write_obs_tail<G>(d, log_obs_row, start, g);
    if (leader) {
      d.log_row[lrow] = (int32_t)((uint32_t)start | 0x80000000u);
      double* le = d.log_env + lrow * 4;
      le[0] = le[1] = le[2] = le[3] = 0.0;
    }
    lp += 1;
  }
  if (leader) reset_head(d, e, r, start, rf_until);
}

template <int G>
__global__ __launch_bounds__(kBlock) void fleet_reset_kernel(FleetDev d, const uint8_t* __restrict__ mask, float* __restrict__ obs) {
  const int g = threadIdx.x % G;
  const int e = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (e >= d.E) return;
  if (mask && !mask[e]) return;
  EnvHead r = d.env[e].h;
  r.nsamp = HEAD_NSAMP(r.nsamp);
  // an explicit reset of an episode that is in progress abandons it: count it so the next start row differs
  if (d.env[e].ep_len > 0 && d.env[e].start_done >= 0) r.episodes += 1;
  int lp = d.log_pos ? d.log_pos[e] : 0;
  int rf_until;
  reset_env<G, true>(d, e, g, g == G - 1, r, obs ? obs + (size_t)e * d.obs_dim : nullptr, lp, rf_until);
  if (d.log_pos && g == G - 1) d.log_pos[e] = lp;
}

// The tail of an EV's step: the rainflow push (second half), the linear model's daily update, the data-log row, and the
// stores of the state records that changed.
template <int DEG, bool WIDE>
__device__ __forceinline__ void ev_finish(const FleetDev& d, const EvIx& i, int c, int N, bool env_ok, bool deg_row, double dt_step, const RfReq& rq,
                                          int tail, int sgn, double soc, double soc_deg, double old_deg, float hl, uint32_t there1,
                                          bool t090, bool inplane, bool crosses, const SegRec& nr, double soh0, double a, double en, bool logs,
                                          size_t lrow, const Hot& h_in, uint32_t& err, double& sei_sample, double& sei_soh, int& sei_tail,
                                          RfTop& sei_top, bool& sei_have_top, RfAccHead& acc_c, RfTop& top_c, bool carry) {
  double soh = soh0;
  RfTop top = top_c;
  if (DEG == FLEET_DEG_RAINFLOW && env_ok) rf_finish(d, i, rq, tail, top, acc_c, err);
  const bool pushed = rq.push;
  if (carry && pushed) top_c = top;
  if (DEG == FLEET_DEG_LINEAR && deg_row) soh = soh - linear_degradation(d, old_deg, soc_deg, dt_step);
  if (DEG == FLEET_DEG_RAINFLOW && !WIDE) {
    sei_sample = soc_deg;
    sei_soh = soh0;
    sei_tail = tail;
    sei_top = top;
    sei_have_top = pushed || carry;
  }
  if (logs) {  // action, energy, degradation, SoH (rainflow: the daily pass below overwrites the last two on its row)
    double* lev = d.log_ev + lrow * 4 * N + c;
    lev[0] = a;
    lev[N] = en;
    lev[2 * N] = soh0 - soh;
    lev[3 * N] = soh;
  }
  FLEET_STAMP(5);
  if (env_ok) {
    // soc_deg == soc whenever the EV has hours left; otherwise it keeps its previous value, which shares the record's float64
    // field with an empty slot's soc == 0.  An EV that is away and stays away leaves its record as it was: no store (what a
    // launch leaves dirty in the L2 is written back before it ends; a third of a caretaker fleet's EV-steps).
    const Hot h_out = hot_encode(d, i, soc, soc_deg, hl, tail, sgn, there1, t090, inplane);
    const bool same = (__double_as_longlong(h_out.x) == __double_as_longlong(h_in.x)) &&
                      (__float_as_uint(h_out.hl) == __float_as_uint(h_in.hl)) && (h_out.bits == h_in.bits);
    if (!same) st_rec16(ev_at(d.hot, i), h_out);
    if (crosses) st_rec16(ev_at(d.run, i), nr);  // the next launch advances into another segment of the EV's schedule
    if (DEG == FLEET_DEG_LINEAR && deg_row) *ev_at(d.soh, i) = soh;  // battery_cap = soh * init_cap is recomputed on use (:673)
  }
}

// ---------------------------------------------------------------------------------------------------------
// One EV's share of a step: EvCharger.charge (ev_charger.py:89-222) and the arrival / departure state machine
// (fleet_environment.py:528-623), without the money terms (they wait for the time row's physics record).  Shared by every
// step kernel.  `hb` = the EV's hot record, `old_deg` = its last logged SOC sample, `tb1` = the schedule columns of the row the
// step advances to, `soh0` = its state of health, `a` = its action.
// ---------------------------------------------------------------------------------------------------------
struct EvPhys {
  double soc, soc_deg;  // episode.soc / episode.soc_deg after the step
  double en;            // charged (> 0) / discharged (< 0) energy; 0 for an absent EV (:114 / :174)
  double rew;           // penalties and rewards of the EV except the price terms
  double penrec, miss;  // episode.penalty_record / cum_soc_missing contributions (:544-584)
  double a_th;          // action * there (fleet_environment.py:491)
  float hl;             // episode.hours_left
  bool pos, t090;       // action >= 0; sticky "target_soc = 0.9" (quirk Q7)
  bool event;           // EVENTS: something the reference counts into episode.events (real_time)
};
template <bool EVENTS>
__device__ __forceinline__ EvPhys ev_physics(const FleetDev& d, const Hot& hb, double old_deg, const RowRec& tb1, double soh0, double a,
                                             double dt_step, bool lunch) {
  EvPhys o;
  double rew = 0.0, penrec = 0.0, miss_sum = 0.0;
  bool ev_lane = false;
  const uint32_t th = HOT_THERE(hb.bits);  // There at the current time row, carried from the previous step / reset
  double soc = HOT_SOC(hb);
  float hl = hb.hl;
  const double cap = soh0 * d.init_cap;
  bool t090 = HOT_T090(hb.bits);
  const double tgt = t090 ? 0.9 : d.target_soc;
  const bool present = (th == 1u);

  // ---- EvCharger.charge (ev_charger.py:89-222) -----------------------------------------------------------------
  // Both action signs in ONE straight-line flow: every quantity of both branches is computed unconditionally and
  // merged with selects / min / max, so a wavefront whose lanes hold both signs (the normal case) does not walk two
  // masked branches, and there is no exec-mask bookkeeping on the hot path.
  const bool pos = (a >= 0.0);
  const double dem = d.p_avail * a * dt_step;   // demanded (dis)charge energy :101 / :162
  const double need = (tgt - soc) * cap;     // ev_total_energy_demand :100
  const double left = -1.0 * soc * cap;      // ev_total_energy_left :161
  // overcharging / over-discharging penalty :104-107 (applied even to an absent EV, clipped; quirk Q9) and
  // :165-167 (needs presence, not clipped)
  // (both products computed: a short-circuit would cost two masked regions for one multiplication)
  const bool viol_c = dem * d.eta_c > need;
  const bool viol_d = (dem * d.eta_d < left) & (th != 0u);
  const bool viol = pos ? viol_c : viol_d;
  const double x = pos ? (dem - need) : (left - dem);
  const double pen_raw = d.penalty_oc * (x * x);
  const double pen_oc = pos ? fmax(pen_raw, d.clip_oc) : pen_raw;
  rew += viol ? pen_oc : 0.0;
  const double lim = div_rcp(need, d.eta_c, d.inv_eta_c);  // need / eta_c, correctly rounded :114
  const double en_p = pos ? fmin(lim, dem) : fmax(left, dem);  // :114 / :174
  const double en = present ? en_p : 0.0;
  rew += (!present && fabs(a) > 0.05) ? d.penalty_invalid * (a * a) : 0.0;  // :120-122 / :180-182
  if (EVENTS) ev_lane = ev_lane || viol || (!present && fabs(a) > 0.05);     // episode.events :108,123,168,183
  soc = soc + div_rcp(pos ? en * d.eta_c : en, cap, rcp_newton1(cap));  // soc + energy / cap, the quotient correctly rounded :128 / :189
  o.a_th = a * (double)th;  // corrected_actions = actions * there (fleet_environment.py:491)

  // ---- arrival / departure state machine (fleet_environment.py:528-623) ----------------------------------
  const float ntl = tb1.tl;
  // departure :532, arrival :603, low state of health :615 are events too
  if (EVENTS) ev_lane = ev_lane || ((hl != 0.0f) != (ntl != 0.0f)) || (soh0 <= 0.9);
  if ((hl != 0.0f) && (ntl == 0.0f)) {  // a car just left :531
    const double target = lunch ? d.target_soc_lunch : tgt;  // :536-557
    const double missing = target - soc;
    if (missing > d.eps) {
      const double pen = soc_violation_penalty(missing);
      rew += pen;
      penrec += pen;  // episode.penalty_record (:549,566,584)
      miss_sum += missing;  // cum_soc_missing (:544,561,579), only reported through the data log
    } else {
      rew += d.fully_charged_reward;
    }
  }
  {
    const bool staying = (ntl != 0.0f) && (hl != 0.0f);  // still charging :593-594; otherwise no car in the next
    hl = staying ? (float)((double)hl - dt_step) : ntl;     // step :597-599 or a new arrival :602-606 (the reference's
    soc = staying ? soc : tb1.sor;                       // `else: raise` is unreachable)
  }
  if (soh0 <= 0.9) t090 = true;  // :613-614 sticky target (quirk Q7)
  o.soc_deg = (hl != 0.0f) ? soc : old_deg;  // :621-623
  o.soc = soc;
  o.hl = hl;
  o.en = en;
  o.rew = rew;
  o.penrec = penrec;
  o.miss = miss_sum;
  o.pos = pos;
  o.t090 = t090;
  o.event = ev_lane;
  return o;
}

// ---------------------------------------------------------------------------------------------------------
// the step (FleetEnv.step, fleet_environment.py:436-702)
//   MULTI = false: exactly one step per launch (the drop-in path: an observation is needed before the next action)
//   MULTI = true : K consecutive steps per launch from an action tape (open-loop rollouts); waves advance
//                  independently, so there is no per-step grid-wide synchronisation at all.
// For G == 64 a wavefront is one env: the env index, its time row and everything derived from them are made
// wave-uniform (readfirstlane), which moves row addressing and the table-row loads to the scalar unit.
// ---------------------------------------------------------------------------------------------------------
// The leading arguments of fleet_step_kernel as the kernel-argument segment lays them out (natural alignment, declaration
// order): where the argument block `d_arg` starts in the segment (`late_args`).
struct StepKernargPrefix {
  const Hot* p_hot;
  const SegRec* p_run;
  const double* p_soh;
  const void* p_actions;
  int p_E, p_N;
  EnvRec* p_env;
  FleetDev d_arg;
};
static_assert(offsetof(StepKernargPrefix, d_arg) == 48 && alignof(FleetDev) == 8, "twelve preloaded dwords, then the argument block");
// ... and the whole argument list: what a launch that does not go through hipLaunchKernel (fleet_describe_step: AQL packets written by
// the library itself, fleet_direct.hip) puts into the kernel-argument segment.  Checked against the code object's metadata at load.
struct StepKernargs {
  const Hot* p_hot;
  const SegRec* p_run;
  const double* p_soh;
  const void* p_actions;
  int p_E, p_N;
  EnvRec* p_env;
  FleetDev d_arg;
  const void* actions;
  int act_mode, K;
  float* obs;
  double* reward;
  uint8_t* done;
  float* terminal_obs;
  int32_t* done_count;
  unsigned long long guard_bytes;  // placement record of the run this launch belongs to (see the kernel, "Placement guard"): byte k =
                                   // 0x80 | die of workgroups w with (w & 7) == k; 0 = no check (every launch through HIP)
  unsigned char* rec_blocks;       // the FIRST launch of a run on the library's own queue: the argument blocks of the run's other launches
  int rec_rows, rec_rotate;        // (rec_rows blocks, FleetStepLaunch::kBlockBytes apart), into which it writes that record; else nullptr
};
static_assert(offsetof(StepKernargs, d_arg) == offsetof(StepKernargPrefix, d_arg) && sizeof(StepKernargs) <= sizeof(FleetStepLaunch::args),
              "the argument block of a described launch");
thread_local FleetStepLaunch* t_describe = nullptr;  // set by fleet_describe_step around fleet_launch_step

// What a K-step instance carries besides the action tape (MULTI only; round 5): the built-in policies and the event-skipping loop of
// real_time each cost the tape rollout scalar registers it spills and branches it never takes -- compiled per use, the tape-only
// instance runs 9 % faster (9.35e8 -> 1.02e9 env-steps/s at 4096x50, profiles/r05_experiments/ab11_kstep_kernel_per_mode.log).
// kModeAll = everything behind run-time tests (the data-log instances and the small groups, where instances are not multiplied).
constexpr int kModeAll = 0, kModeTape = 1, kModePolicy = 2, kModeRt = 3;

template <int G, int DEG, bool MULTI, bool WIDE, bool LOG = false, bool A64 = false, int MODE = kModeAll>
__global__ __launch_bounds__(kBlock, MULTI ? (WIDE ? kMultiWideWaves : kMultiWaves) : kSingleWaves) void fleet_step_kernel(
    // The first twelve argument dwords are preloaded into scalar registers at wave launch (-amdgpu-kernarg-preload-count,
    // fleetrl_amd/build.py; twelve is what fits beside the other user registers): what the first loads of a wavefront need --
    // its lanes' state records and action, E and N for their addresses -- is passed here once more, ahead of the argument
    // block, so that those loads do not wait for an argument fetch.
    // (no __restrict__ on the state pointers: the same kernel stores to these arrays through the argument block)
    const Hot* p_hot, const SegRec* p_run, const double* p_soh, const void* __restrict__ p_actions, int p_E, int p_N, EnvRec* p_env,
    FleetDev d_arg, const void* __restrict__ actions, int act_mode, int K,
                                                               float* __restrict__ obs, double* __restrict__ reward,
                                                               uint8_t* __restrict__ done, float* __restrict__ terminal_obs,
                                                               int32_t* __restrict__ done_count, unsigned long long guard_bytes,
                                                               unsigned char* __restrict__ rec_blocks, int rec_rows, int rec_rotate) {
  FLEET_STAMP_RT(9);
  FLEET_STAMP(0);
  // The argument block is ~150 dwords of scalars for ~100 scalar registers.  One step per launch, one EV per lane: what the END
  // of the step needs of it (the tail store's geometry, the leader lane's constants and pointers, the head bookkeeping) is
  // re-read from the kernel-argument segment after the lane's EV is done -- scalar loads through the constant cache, behind a
  // laundered pointer so that they are not hoisted back to the entry -- instead of being fetched at entry and carried over the
  // whole step in lanes of a vector register (24 v_writelane + 22 v_readlane on every wavefront's path before; none now).
  // Re-reading in the MIDDLE of the EV's step instead stalls on those loads (+0.3 us at 2048 envs) and re-reading everything at
  // several points costs +0.7 us (profiles/r04_experiments/args_reloaded_*).  `late_args_ok` guards the hard-wired offset.
  const FleetDev& d = d_arg;
  auto late_args = [&]() -> const FleetDev& {
    if constexpr (!MULTI && !WIDE) {
      typedef const __attribute__((address_space(4))) char* karg_ptr;
      karg_ptr kp = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(StepKernargPrefix, d_arg);
      asm volatile("" : "+s"(kp));
      return *(const FleetDev*)(const __attribute__((address_space(4))) FleetDev*)kp;
    } else {
      return d_arg;
    }
  };
  constexpr bool kPol = MULTI && (MODE == kModeAll || MODE == kModePolicy);  // the built-in policies are compiled in
  constexpr bool kRt = MULTI && (MODE == kModeAll || MODE == kModeRt);       // the event-skipping loop is compiled in
  // `p_N`: EVs per env in the low half; in the high half the workgroup this launch's grid starts at -- 0 for every launch through HIP;
  // a run on the library's own queues may cover the batch with two grids on two queues (fleet_direct.hip), and the second grid's
  // workgroups continue the first one's numbering.  (Here because it is needed before the first load: p_N is a preloaded argument.)
  const int N = p_N & 0xffff, E_ = p_E;
  const int wg_base = (int)((unsigned)p_N >> 16);
  const int g = threadIdx.x % G;
  const bool leader = (g == G - 1);
  int e_raw = ((int)blockIdx.x + wg_base) * (kBlock / G) + threadIdx.x / G;
  if (G >= 64) e_raw = __builtin_amdgcn_readfirstlane(e_raw);
  const bool env_ok = e_raw < E_;  // surplus groups of the last block run the arithmetic on env E-1 but store nothing
  const int e = env_ok ? e_raw : E_ - 1;

  // One launch = one step and one EV per lane (N <= G): everything the lane needs to start its arithmetic -- its state
  // records, the schedule record of the row the step advances to, its action -- has an address that does not depend on
  // the env's time row, so it is requested before the env record is even read.
  constexpr bool kEarly = !MULTI && !WIDE;
  Hot h_pre = {0.0, 0.0f, 0u};
  SegRec run_pre = {0.0, 0u, 0u};
  double soh_pre = 0.0;
  float a32_pre = 0.0f;
  double a64_pre = 0.0;
  if (kEarly) {
    // unconditional, straight-line requests (surplus lanes of the group re-read the env's last EV and drop it): no
    // exec-mask region for the compiler to close with a wait before the env record is even requested
    const EvIx i0 = {(size_t)e * N, (unsigned)(g < N ? g : N - 1)};
    // in the order the step consumes them (the memory system serves the chip-wide burst of these requests roughly first
    // come, first served, and a wavefront's wait counts its loads in issue order): what the charge arithmetic needs first
    // (full 64-bit lane addresses here: with scalar bases the four requests leave ~1 % later -- each base is a dependent
    // chain on the one scalar unit --, profiles/r03_experiments/ab_saddr.log)
    const size_t f0 = i0.flat();
    h_pre = p_hot[f0];
    soh_pre = p_soh[f0];
    if (A64) a64_pre = ((const double*)p_actions)[f0];
    else a32_pre = ((const float*)p_actions)[f0];
    run_pre = p_run[f0];
  }

  EnvHead r = p_env[e].h;
  if (G >= 64) {
    r.t = __builtin_amdgcn_readfirstlane(r.t);
    r.t_end = __builtin_amdgcn_readfirstlane(r.t_end);
    r.nsamp = __builtin_amdgcn_readfirstlane(r.nsamp);
    r.episodes = __builtin_amdgcn_readfirstlane(r.episodes);
  }
  const uint32_t head_flags = HEAD_FLAGS(r.nsamp);  // FLEET_TFLAG_* of row t + 1, left by the previous launch
  // is the sample this step logs still counted?  (EnvRec::rf_until: nothing reads what is logged after the episode's last
  // degradation row.)  One step per launch: bit 29 of the head, left by the previous launch's leader, who alone holds the row
  // itself; K steps per launch: every lane holds the row and compares per step.
  const bool head_live = HEAD_LIVE(r.nsamp);
  r.nsamp = HEAD_NSAMP(r.nsamp);
  double ep_return = 0.0, penalty_record = 0.0;
  int ep_len = 0;
  int rf_until = -1;
  if (leader) {
    const EnvRec* er = d.env + e;
    ep_return = er->ep_return;
    penalty_record = er->penalty_record;
    ep_len = er->ep_len;
    if (DEG == FLEET_DEG_RAINFLOW && !MULTI) rf_until = er->rf_until;
  }
  if (DEG == FLEET_DEG_RAINFLOW && MULTI) {
    rf_until = p_env[e].rf_until;
    if (G >= 64) rf_until = __builtin_amdgcn_readfirstlane(rf_until);
  }
  uint32_t err = 0;
  double reward_sum = 0.0;
  int n_done = 0;
  float* const obs_row = obs + (size_t)e * d.obs_dim;
  float* const term_row = terminal_obs ? terminal_obs + (size_t)e * d.obs_dim : nullptr;
  const int steps = MULTI ? K : 1;
  const int vzero = (int)__builtin_amdgcn_mbcnt_lo(0u, 0u);  // 0 in every lane, opaque to the uniformity analysis
  // night-charging policy: the env's "charging since" row travels in a register over the K steps
  int night_st = FLEET_NIGHT_IDLE;
  if (kPol && act_mode == FLEET_ACT_POLICY_NIGHT) {
    night_st = d.cold->night_start[e];
    if (G >= 64) night_st = __builtin_amdgcn_readfirstlane(night_st);
  }

  // data log cursor of the env (rows written so far), carried in a register over the launch's steps.  The logging code lives
  // in an instance of its own (LOG, multi-step form; the launcher routes every launch of a logging batch to it), so the
  // kernels of the hot path pay nothing for it -- neither instructions nor registers.
  static_assert(!LOG || MULTI, "the data log is compiled into the multi-step kernel only");
  constexpr bool log_on = LOG;
  int lp = log_on ? d.log_pos[e] : 0;
  if (G >= 64) lp = __builtin_amdgcn_readfirstlane(lp);

  // real_time (event-skipping, fleet_environment.py:453,692-699): the launch repeats the step with the same action until
  // a relevant event happened; it reports the LAST pass's observation / reward / done.  Multi-step kernel, K == 1.
  const bool rt = kRt && (d.real_time != 0);
  // K steps per launch, one EV per lane: the head of the EV's rainflow row (closed-cycle count, sum of means, rainflow_length,
  // the two newest stack entries) is read ONCE per launch and carried in registers over the K steps -- a push updates the
  // registers and stores to the row, nothing re-reads it; the stack words are only read when a closure pops into them
  constexpr bool kRfCarry = MULTI && !WIDE && DEG == FLEET_DEG_RAINFLOW;
  // (Carrying the EV's state record, its state of health and the schedule record the same way was measured and is NOT done:
  // -14 % K-step rate -- the kernel is at the 128-register limit of four resident wavefronts per SIMD, the six extra live
  // registers spill, and those loads overlap with other wavefronts' arithmetic anyway;
  // profiles/r03_experiments/ab_stcarry.log.)
  RfAccHead acc_c = {0.0, 0, 0};
  RfTop top_c = {0.0, 0.0};
  auto carry_load = [&]() {
    const EvIx i0 = {(size_t)e * N, (unsigned)(g < N ? g : N - 1)};
    const double* row = rf_row_of(d, i0);
    acc_c = *reinterpret_cast<const RfAccHead*>(row);
    top_c = *reinterpret_cast<const RfTop*>(row + 2);
  };
  if (kRfCarry) carry_load();
  double last_rew = 0.0;
  bool last_done = false;
  uint32_t head_after = 0;   // single step: FLEET_TFLAG_* of the row after the one the launch advances to
  bool head_reset = false;   // the env was reset in this launch: its head describes the new start row
  for (int k = 0; rt || k < steps; ++k) {
    const int t = r.t;
    int t1 = t + 1;  // :508
    if (t1 > d.T - 1) { t1 = d.T - 1; err |= FLEET_DEVERR_TABLE_END; }
    const int t2 = t1 + 1 > d.T - 1 ? d.T - 1 : t1 + 1;  // the row the NEXT step advances to
    const bool is_done = (t + 1 == r.t_end);  // :627-628
    const bool resets = is_done && d.auto_reset;
    const bool rf_live = MULTI ? (t < rf_until) : head_live;  // the sample of row t + 1 is counted
    // where this step's observation goes: with vec-env auto-reset the terminal observation is reported aside
    float* const step_row = resets ? term_row : obs_row;
    // intermediate steps of a K-step launch only need their observation when the episode ends (terminal observation)
    const bool write_step_obs = env_ok && (step_row != nullptr) && (!MULTI || rt || resets || k == steps - 1);

    FLEET_STAMP(1);
    // ---- loads that depend on the time row: the row's physics scalars and observation tail -----------------------------
    // The 72-byte row is wave-uniform for G == 64, but keeping it in scalar registers for the whole lane loop costs 16
    // of the ~100 SGPRs (spills); a deliberately lane-indexed (vzero == 0) load puts it in vector registers instead.
    // (not in rainflow mode, where vector registers are the scarcer resource)
    // One step per launch: requested as a lane-indexed (vector) load when the head arrives and consumed last -- by the money
    // terms, after the state machine and the observation stores -- so that no wait on the scalar side stands between the
    // arrival of the state records and the charge arithmetic; the row flags the state machine needs come with the head.
    const PhysHot ph = *reinterpret_cast<const PhysHot*>(d.tab_phys + (t + ((kEarly || DEG != FLEET_DEG_RAINFLOW) ? vzero : 0)));
    // flags of the row after t1, for the head the next launch reads
    uint32_t flags_after = 0;
    if (kEarly) flags_after = reinterpret_cast<const PhysHot*>(d.tab_phys + (t1 + vzero))->flags_next;
    // hours this step spans: `get_next_dt` (:455, :994-1008) -- a constant unless the grid is irregular (real_time only)
    const double dt_step = rt ? d.tab_phys[t].dt : d.dt;
    const uint32_t flags1 = kEarly ? head_flags : ph.flags_next;
    const bool lunch = d.is_caretaker && (flags1 & FLEET_TFLAG_LUNCH);
    const bool deg_row = (DEG != FLEET_DEG_NONE) && (flags1 & FLEET_TFLAG_DEG);
    // the few wavefronts with extra work after the step (daily evaluation, episode end + reset) finish last and set the
    // launch's duration: they get issue priority over their SIMD's other wavefronts for the step itself (-3 % per launch)
    if (!MULTI && G >= 64 && ((DEG == FLEET_DEG_RAINFLOW && deg_row) || is_done)) __builtin_amdgcn_s_setprio(3);
    const size_t abase = ((size_t)(rt ? 0 : k) * d.E + e) * N;

    // data log: the step's row (not written for the step that ends the episode, :679) -- its observation goes to the log's own
    // buffer, so K-step launches log every step although they only return the last observation
    const bool logs = log_on && env_ok && !is_done;
    const size_t lrow = logs ? (size_t)(lp % d.log_cap) * d.E + e : 0;
    float* const log_obs_row = logs ? d.log_obs + lrow * d.obs_dim : nullptr;

    const float tail_first = (write_step_obs || logs) ? tail_load<G>(d, t1, g) : 0.0f;  // consumed after the lane loop

    // Multi-step launches: an opaque per-iteration zero keeps the compiler from hoisting every lane address of the step
    // body out of the K loop (60 extra live vector registers = half the resident wavefronts); recomputing them each
    // step costs a handful of integer instructions.
    int kz = 0;
    if (MULTI) asm volatile("" : "+v"(kz));

    // Which action rule applies to this env in this step (policies only; FLEET_ACT_POLICY_NIGHT resolves to one of
    // "all zeros" / "all ones" / the distributed rule per step, benchmarking/night_charging.py:81-98)
    int pol = act_mode;
    if (kPol && act_mode == FLEET_ACT_POLICY_NIGHT) {
      const FleetCold* cd = d.cold;
      const int hm = cd->tab_hm[t];
      const int hour = hm >> 8, minute = hm & 255;
      if (d.is_caretaker && hour >= 11 && hour <= 14) {
        pol = FLEET_ACT_POLICY_DISTRIBUTED;  // :85-88, `continue`: the window bookkeeping below is skipped
      } else {
        bool charging = (night_st != FLEET_NIGHT_IDLE);
        if ((cd->night_hour <= hour && cd->night_minute <= minute) || charging) {  // :90
          if (!charging) night_st = t;  // charging_start = copy(time) :91-92
          charging = true;
          pol = FLEET_ACT_POLICY_UNCONTROLLED;  // np.ones :94
        } else {
          pol = -1;  // np.zeros :96
        }
        // :97-98  (time - charging_start).total_seconds() / 3600 > int(max_time_needed); rows are a regular grid
        if (charging && (t - night_st) * cd->step_s > cd->night_limit_s) night_st = FLEET_NIGHT_IDLE;
      }
    }

    double cash = 0.0, rew = 0.0, asum = 0.0, penrec = 0.0, miss_sum = 0.0;
    // what the daily SEI pass needs of the lane's EV, carried in registers when a lane owns one EV (no reload round trip for
    // the few wavefronts on the 14:45 row, which otherwise finish last and set the launch's duration)
    double sei_sample = 0.0, sei_soh = 0.0;
    int sei_tail = 0;
    RfTop sei_top = {0.0, 0.0};
    bool sei_have_top = false;
    bool ev_lane = false;  // real_time: something the reference counts into episode.events happened to this lane's EVs
    // Several EVs per lane, one step per launch (N > 64): the lane's NEXT EV's records are requested before the current EV is
    // worked on (software pipelining of the lane loop) -- otherwise every turn of the loop starts with a memory round trip
    constexpr bool kPipe = WIDE && !MULTI;
    Hot hb_n = {0.0, 0.0f, 0u};
    SegRec rr_n = {0.0, 0u, 0u};
    double soh_n = 0.0, act_n = 0.0;
    auto request_ev = [&](int cn) {
      const int cc = cn < N ? cn : N - 1;  // past the end: a harmless re-read of the last EV (no exec-mask region)
      const EvIx in = {(size_t)e * N, (unsigned)cc}, ia = {abase, (unsigned)cc}, it = {(size_t)t1 * N, (unsigned)cc};
      hb_n = *ev_at(d.hot, in);
      soh_n = *ev_at(d.soh, in);
      act_n = (act_mode == FLEET_ACT_F64) ? *ev_at((const double*)actions, ia) : (double)*ev_at((const float*)actions, ia);
      rr_n = *ev_at(d.seg, it);
    };
    if (kPipe) request_ev(g);
    for (int c = g + kz; c < N; c += G) {
      const EvIx i = {(size_t)e * N, (unsigned)c};
      // all loads of this EV are issued before anything is consumed
      const Hot hb = kEarly ? h_pre : (kPipe ? hb_n : *ev_at(d.hot, i));
      // schedule record of row t1: carried with the state (one step per launch) or read from the table (K steps / several EVs
      // per lane: the time row is in registers there, the table read is not on anybody's critical path, and the carried
      // record is rewritten once, when the launch ends)
      const EvIx it1 = {(size_t)t1 * N, (unsigned)c};  // the table row the step advances to
      const SegRec rr = kEarly ? run_pre : (kPipe ? rr_n : *ev_at(d.seg, it1));
      const double soh0 = kEarly ? soh_pre : (kPipe ? soh_n : *ev_at(d.soh, i));
      const double act_cur = act_n;
      if (kPipe) request_ev(c + G);
      // last logged SOC sample: shares the record's float64 field with the SOC (struct Hot); the soc_deg plane only holds
      // it in a combination that does not occur inside the reference's episodes (dependent load, INPLANE)
      const bool inplane = HOT_INPLANE(hb.bits);
      double old_deg = hb.x;
      if (inplane) old_deg = d.soc_deg[i.flat()];
      RfReq rq;
      rq.push = false;
      constexpr bool kRfEarly = MULTI && DEG == FLEET_DEG_RAINFLOW;
      rq.win = false;
      if (kRfCarry) {
        rq.acc = acc_c;
        rq.top = top_c;
      } else if (kRfEarly && env_ok && rf_live) {
        rf_request(d, i, HOT_TAIL(hb.bits), rq);
      }
      double a;
      if (kPol && act_mode >= FLEET_ACT_POLICY_UNCONTROLLED) {
        // built-in open-loop policies of the reference's benchmark harnesses, evaluated in place of an action tape
        if (pol == FLEET_ACT_POLICY_UNCONTROLLED) {
          a = 1.0;  // benchmarking/uncontrolled_charging.py:51-54: np.ones(n_evs)
        } else if (pol < 0) {
          a = 0.0;
        } else {
          // benchmarking/distributed_charging.py:50-54: clip(get_dist_factor(), 0, 1), get_dist_factor =
          // hours_needed / (hours_left + 0.001) from the TABLE row of the current time (fleet_environment.py:782-799)
          const RowRec tb0 = seg_row(d.seg[(size_t)t * N + c], t, d.dt);
          const FleetCold* cd = d.cold;
          const double th0 = (double)tb0.there;
          const double cl0 = (HOT_T090(hb.bits) ? 0.9 : d.target_soc) * th0 - tb0.sor;
          const double hn0 = cl0 * cd->batt_cap_nominal / cd->hn_denominator;
          const double f = hn0 / ((double)tb0.tl + 0.001);
          a = f < 0.0 ? 0.0 : (f > 1.0 ? 1.0 : f);
        }
      } else if (kEarly) {
        // the widening must stay here: hoisted into the early-load block it would wait for every outstanding load
        float a32 = a32_pre;
        asm volatile("" : "+v"(a32));
        a = A64 ? a64_pre : (double)a32;
      } else if (kPipe) {
        a = act_cur;
      } else {
        const EvIx ia = {abase, (unsigned)c};
        a = (act_mode == FLEET_ACT_F64) ? *ev_at((const double*)actions, ia) : (double)*ev_at((const float*)actions, ia);
      }
      // the schedule record of the row AFTER next, for the next launch: only when that row starts a new segment of the EV's
      // schedule (a departure, an arrival, ...).  Nothing in this step waits for it except the store at its very end.
      const bool crosses = kEarly && (t1 + 1 >= SEG_END(rr.se));
      SegRec nr = rr;
      const EvIx it2 = {(size_t)t2 * N, (unsigned)c};
      if (crosses) nr = *ev_at(d.seg, it2);
      const RowRec tb1 = seg_row(rr, t1, d.dt);
      FLEET_STAMP(2);
      const EvPhys ph_ev = ev_physics<kRt>(d, hb, old_deg, tb1, soh0, a, dt_step, lunch);
      double soc = ph_ev.soc;
      float hl = ph_ev.hl;
      const bool t090 = ph_ev.t090, pos = ph_ev.pos;
      const double en = ph_ev.en, soc_deg = ph_ev.soc_deg;
      rew += ph_ev.rew;
      penrec += ph_ev.penrec;
      miss_sum += ph_ev.miss;
      asum += ph_ev.a_th;
      if (kRt) ev_lane = ev_lane || ph_ev.event;
      // ---- SOC log (:655): the new sample of the streaming rainflow; what a cycle closure needs of the EV's row is requested
      // here and consumed after the observation stores and the money terms
      int tail = HOT_TAIL(hb.bits), sgn = HOT_SGN(hb.bits);
      // (K steps per launch: request and consumption stay together -- the registers the request holds across the observation
      // stores would cost the multi-step kernel a resident wavefront per SIMD)
      constexpr bool kSplitRf = !MULTI;
      if (kSplitRf && DEG == FLEET_DEG_RAINFLOW && env_ok && rf_live) rf_begin(d, i, old_deg, soc_deg, tail, sgn, rq);

      FLEET_STAMP(3);
      // ---- observation of the advanced time row (fleet_environment.py:511-518, 645-652) ------------------------
      const double tgt_obs = t090 ? 0.9 : d.target_soc;  // the target the observer sees: after this step's sticky update
      if (write_step_obs) write_obs_ev(d, step_row, c, soc, hl, tgt_obs, tb1);
      if (logs) write_obs_ev(d, log_obs_row, c, soc, hl, tgt_obs, tb1);

      // ---- money terms of EvCharger.charge: the only consumers of the time row's physics record, which was requested when
      // the env head arrived and has had the charge arithmetic, the state machine and the observation stores to get here
      {
        const double grid_e = fmax(en - ph.pv_share, 0.0);  // :142 (charging only)
        cash += pos ? -(grid_e * ph.k_cost) : en * ph.k_rev;       // -charging_cost :149 / +discharging_revenue :196-199
        rew += pos ? ph.k_charge * grid_e : ph.k_discharge * en;   // :154-156 / :204-206
      }

      FLEET_STAMP(4);
      // ---- SOC log + daily degradation (:655-673) -------------------------------------------------------------
      if (!kSplitRf && DEG == FLEET_DEG_RAINFLOW && env_ok && rf_live) rf_begin(d, i, old_deg, soc_deg, tail, sgn, rq, kRfEarly);
      ev_finish<DEG, WIDE>(d, i, c, N, env_ok, deg_row, dt_step, rq, tail, sgn, soc, soc_deg, old_deg, hl, tb1.there, t090, inplane,
                           crosses, nr, soh0, a, en, logs, lrow, hb, err, sei_sample, sei_soh, sei_tail, sei_top, sei_have_top, acc_c,
                           top_c, kRfCarry);
      if (!WIDE) break;  // N <= G: a single pass, and no loop for the compiler to hoist rare-path constants out of
    }
    {  // ---- the rest of the step reads the argument block afresh (see `late_args`) ----
    const FleetDev& d = late_args();
    if (!MULTI && !WIDE && (d.T != d_arg.T || d.E != p_E)) err |= FLEET_DEVERR_INTERNAL;  // the block is not where it is assumed to be
    if (write_step_obs) tail_store<G>(d, step_row, t1, g, tail_first);
    if (logs) tail_store<G>(d, log_obs_row, t1, g, tail_first);
    if (DEG != FLEET_DEG_NONE) r.nsamp += 1;

    FLEET_STAMP(6);
    // ---- per-env reductions; totals land in the leader lane ---------------------------------------------------
    if (G >= 64) {
      wave_sum4_to_last(cash, rew, asum, penrec);  // this wavefront's lanes: totals in its last lane
      if (G > 64) {
        // an env of several wavefronts: their partial sums meet in the LDS and the group's last lane adds them in wavefront order
        // (K steps per launch: the env's wavefronts meet here once per step -- every wavefront of the workgroup runs the same K
        // steps, so the barrier is uniform; two buffers in turn, so that a wavefront already in the next step does not overwrite
        // what the leader is still adding up.  The event-skipping loop of real_time has env-dependent trip counts: never here.)
        __shared__ double s_part[2][kBlock / 64][4];
        double(*part)[4] = s_part[MULTI ? (k & 1) : 0];
        const int w = (int)threadIdx.x / 64;
        if ((threadIdx.x & 63) == 63) {
          part[w][0] = cash;
          part[w][1] = rew;
          part[w][2] = asum;
          part[w][3] = penrec;
        }
        __syncthreads();
        if (leader) {
          const int w0 = ((int)threadIdx.x / G) * (G / 64);
          double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
          for (int j = 0; j < G / 64; ++j) {
            s0 += part[w0 + j][0];
            s1 += part[w0 + j][1];
            s2 += part[w0 + j][2];
            s3 += part[w0 + j][3];
          }
          cash = s0;
          rew = s1;
          asum = s2;
          penrec = s3;
        }
      }
    } else {
      cash = group_sum_to_last<G>(cash);
      rew = group_sum_to_last<G>(rew);
      asum = group_sum_to_last<G>(asum);
      if (__any(penrec != 0.0)) penrec = group_sum_to_last<G>(penrec);  // wave-uniform branch; rare
    }
    if (log_on) miss_sum = group_sum_to_last<G>(miss_sum);  // kernel-argument-uniform branch (log_data only)
    r.t = t1;
    if (leader) {
      penalty_record += penrec;
      // LoadCalculation.check_violation (load_calculation.py:93) and the sigmoid penalty (:496-502)
      const double head_room = d.grid_connection - ph.load - asum * d.evse_power + ph.pv;
      const double over = fabs(head_room < 0.0 ? head_room : 0.0);
      if (over > 0.0) {
        const double pen = overloading_penalty(over / d.grid_connection + 1.0, d.penalty_overload);
        rew += pen;
        penalty_record += pen;
        if (kRt) ev_lane = true;  // :499
      }
      if (logs) {
        d.log_row[lrow] = t1;  // episode.time
        double* le = d.log_env + lrow * 4;
        le[0] = rew;
        le[1] = cash;
        le[2] = over;      // grid = abs(overload_amount) (:660)
        le[3] = miss_sum;  // soc_v = abs(cum_soc_missing) (:661)
      }
      ep_return += rew;  // :637
      ep_len += 1;
      reward_sum += rew;
      last_rew = rew;
      if (env_ok) {
        d.env[e].cashflow = cash;  // cashflow = -charging_cost + discharging_revenue (ev_charger.py:225)
        if (!MULTI) {
          reward[e] = rew;
          done[e] = is_done ? 1 : 0;
        }
      }
    }
    FLEET_STAMP(7);
    // ---- daily SEI evaluation (:666-671) ---------------------------------------------------------------------
    // Runs in a second pass over the group's EVs, after the per-step arithmetic has retired, so that its temporaries
    // (transcendentals, accumulators) never coexist with the hot path's registers.  One step in 96, and wave-uniform for
    // G == 64.
    if (DEG == FLEET_DEG_RAINFLOW && deg_row && env_ok) {
      for (int c = g; c < N; c += G) {
        const EvIx ix = {(size_t)e * N, (unsigned)c};
        const size_t i = ix.flat();
        double deg, soh_new;
        if (!WIDE) {
          // (its scalars from the argument block the end of the step has just re-read, not from the device-resident copy: one
          // dependent round trip less on the wavefronts that end the launch, -2 % per launch at 4096 x 50)
          deg = sei_evaluate(d, ix, sei_sample, r.nsamp, sei_tail, sei_top, sei_have_top, err, dt_step,
                             kRfCarry ? &acc_c.rf_len : nullptr);
          soh_new = sei_soh - deg;
        } else {  // several EVs per lane: re-read the few words from the records this lane has just stored
          const Hot hb = d.hot[i];
          const double sample = HOT_INPLANE(hb.bits) ? d.soc_deg[i] : hb.x;
          const RfTop none = {0.0, 0.0};
          deg = sei_evaluate(*d.self, ix, sample, r.nsamp, HOT_TAIL(hb.bits), none, false, err, dt_step);
          soh_new = d.soh[i] - deg;
        }
        d.soh[i] = soh_new;
        if (logs) {
          double* lev = d.log_ev + lrow * 4 * N + c;
          lev[2 * N] = deg;
          lev[3 * N] = soh_new;
        }
        if (!WIDE) break;
      }
    }
    if (logs) lp += 1;
    head_after = flags_after;
    if (is_done) {
      n_done += 1;
      if (leader && env_ok) {
        EnvRec* er = d.env + e;
        er->last_ep_return = ep_return;
        d.cold->last_len[e] = ep_len;
        er->start_done |= (int32_t)0x80000000u;  // episode.done
      }
      r.episodes += 1;
      if (resets) {
        head_reset = true;
        if (env_ok) {
          reset_env<G, LOG>(*d.self, e, g, leader, r, obs_row, lp, rf_until);
          if (kRfCarry) carry_load();  // the reset rewrote the row's head (same lane, same addresses: program order holds)
        } else {  // surplus group: keep its registers moving without touching memory
          r.t = choose_start(d.cold, d.E, e, r.episodes);
          r.t_end = d.tab_finish ? d.tab_finish[r.t] : r.t + d.episode_steps;
          r.nsamp = (DEG != FLEET_DEG_NONE) ? 1 : 0;
        }
        ep_return = 0.0;
        ep_len = 0;
        penalty_record = 0.0;
      }
    }
    if (rt) {
      // EventManager.check_event (event_manager.py:16-31): the advanced row's clock minute == 15 is an event of its own;
      // the end of the episode is one (:629); running off the table ends the loop (the reference would raise there)
      last_done = is_done;
      const int hm1 = d.cold->tab_hm[t1];
      const bool minute15 = ((hm1 & 255) == 15) && !(hm1 & 0x8000);  // minute == 15 and second == 0
      if (group_any<G>(ev_lane) || is_done || minute15 || (t + 1 > d.T - 1)) break;
    }
    }  // late_args scope
  }

  {  // ---- after the steps: again through the freshly read block ----
  const FleetDev& d = late_args();
  // (only where a single-step launch that reads them can follow on the same handle: `carry_run`, i.e. up to kMaxGroup EVs per env)
  if (!kEarly && (!WIDE || d.carry_run) && env_ok) {  // the carried schedule records: the row the NEXT launch advances to
    const int rn = r.t + 1 > d.T - 1 ? d.T - 1 : r.t + 1;
    for (int c = g; c < N; c += G) d.run[(size_t)e * N + c] = d.seg[(size_t)rn * N + c];
  }
  if (leader && env_ok) {
    EnvRec* er = d.env + e;
    // the head carries the row flags the next launch's state machine needs (struct EnvHead)
    uint32_t nflags = head_after;
    if (!kEarly || head_reset) nflags = d.tab_phys[r.t].flags_next;
    r.nsamp = HEAD_PACK(r.nsamp, nflags, r.t < rf_until);
    er->h = r;
    er->ep_return = ep_return;
    er->ep_len = ep_len;
    er->penalty_record = penalty_record;
    if (log_on) d.log_pos[e] = lp;
    if (MULTI) {
      reward[e] = rt ? last_rew : reward_sum;
      if (done && (rt || steps == 1)) done[e] = (rt ? last_done : (n_done != 0)) ? 1 : 0;  // one agent step: its done flag
      if (done_count) done_count[e] = n_done;
      if (kPol && act_mode == FLEET_ACT_POLICY_NIGHT) d.cold->night_start[e] = night_st;
    }
  }
  // Placement guard (single-step launches on the library's own queue, fleet_direct.hip).  Such launches carry no release fence,
  // which is only correct while workgroup w of every launch of a run runs on the die (XCC) that ran workgroup w of the previous
  // one -- a die's L2 is the only place the env's newest state lives.  The hardware deals the workgroups of a dispatch to the dies
  // round-robin from a die that belongs to the QUEUE, but that die is not a constant: it moves by one whenever a queue is created
  // or destroyed in the process (measured: tools/ubench/xcc_map.cpp) and the platform promises nothing (MI355X_MICROARCH.md,
  // "Workgroup dispatch, XCD placement").  So the FIRST launch of every run -- which reads state that the previous run's release
  // made everybody's, and may therefore sit anywhere -- writes the dies of its first eight workgroups into the argument blocks of
  // the run's other launches (byte k of `guard_bytes` there = 0x80 | die of workgroup k; written through and drained), and every
  // other launch compares the die it finds itself on with its slot of that record -- here, at the very end, from the
  // kernel-argument segment itself: one scalar load that hits the constant cache, one s_getreg, a few scalar instructions, nothing
  // held across the step (a record carried from the entry cost 1-3 % per launch in spilled scalars, one fetched by a device-scope
  // vector load 4 %: profiles/r06_experiments/placement_guard_cost.log).  A mismatch raises FLEET_DEVERR_PLACEMENT (sticky; the
  // run's results are void, fleet_check_errors tells the caller).  Launches through HIP carry a record of zeros: no check.
  if constexpr (!MULTI) {
    struct Tail { unsigned long long rec; unsigned char* blocks; int rows, rotate; };
    static_assert(offsetof(StepKernargs, rec_blocks) == offsetof(StepKernargs, guard_bytes) + 8, "one 24-byte piece of the argument block");
    typedef const __attribute__((address_space(4))) char* karg_ptr;
    karg_ptr kp = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(StepKernargs, guard_bytes);
    asm volatile("" : "+s"(kp));  // (not hoisted to the entry: see late_args)
    const Tail& tl = *(const Tail*)(const __attribute__((address_space(4))) Tail*)kp;
    const unsigned w = blockIdx.x + (unsigned)wg_base;
    const unsigned have = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID, bits [3:0]
    const unsigned slot = (unsigned)(tl.rec >> (8u * (w & 7u))) & 0xffu;
    if ((slot & 0x80u) && (slot & 0xfu) != have) err |= FLEET_DEVERR_PLACEMENT;
    if (tl.blocks != nullptr && blockIdx.x < 8 && threadIdx.x < 64) {  // the run's first launch: its first eight workgroups record
      // (the blocks are FleetStepLaunch::kBlockBytes apart; `rotate`: 0 -- or, test hook of the guard's negative test, the record
      // shifted by that many workgroups)
      unsigned char* at = tl.blocks + offsetof(StepKernargs, guard_bytes) + ((w + (unsigned)tl.rotate) & 7u);
      for (int r = (int)threadIdx.x; r < tl.rows; r += 64)
        __hip_atomic_store(at + (size_t)r * sizeof(FleetStepLaunch::args), (unsigned char)(0x80u | have), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    (void)guard_bytes; (void)rec_blocks; (void)rec_rows; (void)rec_rotate;
  }
  if (err && env_ok) {  // FLEET_DEVERR_*: per env, and OR-ed into the one word the host-pointer step brings back with its results
    atomicOr(&d.env[e].err, err);
    atomicOr(d.self->err_any, err);
  }
  }  // late_args scope
  FLEET_STAMP(8);
  FLEET_STAMP_RT(10);
  FLEET_STAMP_WHERE();
}

// FleetEnv.get_dist_factor (fleet_environment.py:782-799)
__global__ void fleet_dist_factor_kernel(FleetDev d, double* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)d.E * d.N) return;
  const int e = (int)(i / d.N), c = (int)(i % d.N);
  const int t = d.env[e].h.t;
  const RowRec tb = seg_row(d.seg[(size_t)t * d.N + c], t, d.dt);
  const double th = (double)tb.there;
  const double tgt = HOT_T090(d.hot[i].bits) ? 0.9 : d.target_soc;
  const double cl = tgt * th - tb.sor;
  const double hn = cl * d.cold->batt_cap_nominal / d.cold->hn_denominator;
  out[i] = hn / ((double)tb.tl + 0.001);
}

// fleet_get: unpack one field into a contiguous buffer (types as documented in include/fleet_hip.h)
__global__ void fleet_gather_field_kernel(FleetDev d, int field, void* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t E = d.E, EN = (size_t)d.E * d.N;
  const bool per_car = field == FLEET_F_SOC || field == FLEET_F_HOURS_LEFT || field == FLEET_F_SOH || field == FLEET_F_SOC_DEG ||
                       field == FLEET_F_TARGET_SOC || field == FLEET_F_RF_LEN || field == FLEET_F_RF_CYCLES || field == FLEET_F_RF_STACK ||
                       field == FLEET_F_FD_CYC ||
                       field == FLEET_F_FD_CAL || field == FLEET_F_SEI_L;
  if (i >= (per_car ? EN : E)) return;
  switch (field) {
    case FLEET_F_SOC: ((double*)out)[i] = HOT_SOC(d.hot[i]); break;
    case FLEET_F_HOURS_LEFT: ((float*)out)[i] = d.hot[i].hl; break;
    case FLEET_F_SOH: ((double*)out)[i] = d.soh[i]; break;
    case FLEET_F_SOC_DEG: ((double*)out)[i] = HOT_INPLANE(d.hot[i].bits) ? d.soc_deg[i] : d.hot[i].x; break;
    case FLEET_F_TARGET_SOC: ((double*)out)[i] = HOT_T090(d.hot[i].bits) ? 0.9 : d.target_soc; break;
    case FLEET_F_RF_LEN:
      ((int32_t*)out)[i] = d.rf_rows ? reinterpret_cast<const RfHdr*>(d.rf_rows + i * (size_t)d.rf_row_stride)->rf_len : 1;
      break;
    case FLEET_F_RF_CYCLES:
      ((int32_t*)out)[i] = d.rf_rows ? reinterpret_cast<const RfHdr*>(d.rf_rows + i * (size_t)d.rf_row_stride)->nc : 0;
      break;
    case FLEET_F_RF_STACK: ((int32_t*)out)[i] = d.rf_rows ? HOT_TAIL(d.hot[i].bits) : 0; break;
    case FLEET_F_FD_CYC: ((double*)out)[i] = d.sei[i].fd_cyc; break;
    case FLEET_F_FD_CAL: ((double*)out)[i] = d.sei[i].fd_cal; break;
    case FLEET_F_SEI_L: ((double*)out)[i] = d.sei[i].sei_l; break;
    case FLEET_F_TIME_IDX: ((int32_t*)out)[i] = d.env[i].h.t; break;
    case FLEET_F_START_IDX: ((int32_t*)out)[i] = d.env[i].start_done & 0x7FFFFFFF; break;
    case FLEET_F_CASHFLOW: ((double*)out)[i] = d.env[i].cashflow; break;
    case FLEET_F_EP_RETURN: ((double*)out)[i] = d.env[i].ep_return; break;
    case FLEET_F_EP_LEN: ((int32_t*)out)[i] = d.env[i].ep_len; break;
    case FLEET_F_LAST_EP_RETURN: ((double*)out)[i] = d.env[i].last_ep_return; break;
    case FLEET_F_LAST_EP_LEN: ((int32_t*)out)[i] = d.cold->last_len[i]; break;
    case FLEET_F_LAST_EP_LEN_F64: ((double*)out)[i] = (double)d.cold->last_len[i]; break;
    case FLEET_F_RF_UNTIL: ((int32_t*)out)[i] = d.env[i].rf_until; break;
    case FLEET_F_ERROR_BITS: ((uint32_t*)out)[i] = d.env[i].err; break;
    case FLEET_F_DONE: ((uint8_t*)out)[i] = (uint8_t)(d.env[i].start_done < 0); break;
    case FLEET_F_EPISODES: ((int32_t*)out)[i] = d.env[i].h.episodes; break;
    case FLEET_F_PENALTY_RECORD: ((double*)out)[i] = d.env[i].penalty_record; break;
    default: break;
  }
}

// Host path: the terminal observations of the envs that finished in this step, compacted (fleet_step_host moves only these
// rows over PCIe instead of the whole [E, obs_dim] buffer).  One workgroup: a serial-over-chunks scan of the done flags in
// env order (deterministic), then the rows are copied by the whole launch.
__global__ __launch_bounds__(1024) void fleet_term_scan_kernel(const uint8_t* __restrict__ done, int E, int32_t* __restrict__ idx,
                                                               int32_t* __restrict__ count, const EnvRec* __restrict__ env,
                                                               const FleetCold* __restrict__ cold, double* __restrict__ ep_ret,
                                                               int32_t* __restrict__ ep_len) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int e0 = 0; e0 < E; e0 += 1024) {
    const int e = e0 + (int)threadIdx.x;
    const bool f = (e < E) && done[e] != 0;
    const unsigned long long m = __ballot(f);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    if (f) {  // the finished episode's return / length travel with the index (what SB3's Monitor would report)
      idx[off + before] = e;
      ep_ret[off + before] = env[e].last_ep_return;
      ep_len[off + before] = cold->last_len[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += s_wave[w];
      s_base += tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = s_base;
}

__global__ void fleet_term_gather_kernel(const float* __restrict__ term, int obs_dim, const int32_t* __restrict__ idx,
                                         const int32_t* __restrict__ count, float* __restrict__ compact) {
  const int n = *count;
  for (int k = blockIdx.x; k < n; k += gridDim.x) {
    const float* src = term + (size_t)idx[k] * obs_dim;
    float* dst = compact + (size_t)k * obs_dim;
    for (int j = threadIdx.x; j < obs_dim; j += blockDim.x) dst[j] = src[j];
  }
}

// Self-test of div_rcp (fleet_selftest_division): operand pairs drawn the way the charge arithmetic forms them, the IEEE division
// sequence beside the reciprocal form, bit for bit.  case 0: need / eta_c with the host's correctly rounded 1 / eta_c;
// case 1: energy / cap with rcp_newton1(cap).  A few lanes in a thousand carry the edge values (+-0, a denormal-sized residue).
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(unsigned long long h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }
__global__ void fleet_selftest_division_kernel(unsigned long long n, unsigned long long seed, unsigned long long* __restrict__ bad) {
  unsigned long long b0 = 0, b1 = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long h = mix64(seed + i * 4ull);
    const double soc = -0.25 + 1.5 * u01(h), soh = 0.8 + 0.2 * u01(mix64(h)), init_cap = 10.0 + 90.0 * u01(mix64(h + 1));
    const double eta = 0.5 + 0.5 * u01(mix64(h + 2)), tgt = (h & 1) ? 0.85 : 0.9;
    const double a = 2.0 * u01(mix64(h + 3)) - 1.0, p_avail = 2.0 + 20.0 * u01(mix64(h + 4));
    const double cap = soh * init_cap;
    double need = (tgt - soc) * cap;
    double en = p_avail * a * 0.25;
    const unsigned sel = (unsigned)(h >> 40) % 1000u;
    if (sel == 0) need = 0.0;
    if (sel == 1) need = -0.0;
    if (sel == 2) en = -0.0;
    if (sel == 3) en = 7e-18 * cap;
    const double inv_eta = 1.0 / eta;  // IEEE: correctly rounded, like the host's
    const double q0 = need / eta, r0 = div_rcp(need, eta, inv_eta);
    const double x1 = (a >= 0.0) ? en * eta : en;
    const double q1 = x1 / cap, r1 = div_rcp(x1, cap, rcp_newton1(cap));
    b0 += (__double_as_longlong(q0) != __double_as_longlong(r0));
    b1 += (__double_as_longlong(q1) != __double_as_longlong(r1));
  }
  if (b0) atomicAdd(bad, b0);
  if (b1) atomicAdd(bad + 1, b1);
}

// cycle_stress (hardware float32 log2 inside x^-0.501, Taylor exp) against the same expression in library double precision, on n
// pseudo-random (range, mean, weight) triples of the reachable domain: worst[0] = largest relative difference as the bits of a double
__global__ void fleet_selftest_stress_kernel(unsigned long long n, unsigned long long seed, unsigned long long* __restrict__ worst) {
  double w = 0.0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long h = mix64(seed + i * 3ull);
    // depth of discharge: half of the samples log-uniform over 1e-9 ... 1 (tiny cycles are the common ones), half uniform
    const double u = u01(h), v = u01(mix64(h + 1));
    const double rng = (h & 1) ? exp(-20.7232658 * u) : u;
    const double mean = -0.25 + 1.5 * v;  // a mean SOC a little outside [0, 1] too (quirk Q9: the reference does not clip it)
    const double count = (h & 2) ? 1.0 : 0.5;
    const double st = 0.9 + 0.2 * u01(mix64(h + 2));
    const double got = cycle_stress(rng, mean, count, st);
    double eff = rng * count;
    eff = eff > 1.0 ? 1.0 : eff;
    const double want = (eff > 0.0) ? (1.0 / (1.4E5 * pow(eff, -0.501) + -1.23E5)) * exp(1.04 * (mean - 0.5)) * st : 0.0;
    const double rel = (want != 0.0) ? fabs(got - want) / fabs(want) : fabs(got);
    w = rel > w ? rel : w;
  }
  atomicMax(worst, (unsigned long long)__double_as_longlong(w));  // non-negative doubles order like their bit patterns
}

int group_size(int N) {
  int G = 1;
  while (G < N && G < 64) G <<= 1;
  return G;
}

// A K-step launch without the data log: the instance that carries what the launch uses -- the tape only, the built-in policies, or the
// event-skipping loop (groups of 32 lanes and more; smaller groups keep ONE instance with everything behind run-time tests).
template <int G, int DEG, bool WIDE>
void launch_many(const FleetDev& d, dim3 grid, dim3 block, const void* actions, int act_mode, int K, float* obs, double* reward,
                 uint8_t* done, float* terminal_obs, int32_t* done_count, hipStream_t s) {
#define FLEET_PRE_ARGS d.hot, d.run, d.soh, actions, d.E, d.N, d.env,  /* the leading arguments (12 dwords, preloaded) */
#define FLEET_MANY(MODE)                                                                                                          \
  hipLaunchKernelGGL((fleet_step_kernel<G, DEG, true, WIDE, false, false, MODE>), grid, block, 0, s, FLEET_PRE_ARGS d, actions, act_mode, K, \
                     obs, reward, done, terminal_obs, done_count, 0ull, nullptr, 0, 0)
  if (G < 32) FLEET_MANY(kModeAll);
  else if (d.real_time) FLEET_MANY((G < 32 ? kModeAll : kModeRt));
  else if (act_mode >= FLEET_ACT_POLICY_UNCONTROLLED) FLEET_MANY((G < 32 ? kModeAll : kModePolicy));
  else FLEET_MANY((G < 32 ? kModeAll : kModeTape));
#undef FLEET_MANY
}

// A single-step launch goes to the HIP stream -- or, when fleet_describe_step asks, is written down instead: kernel, grid and the
// argument block, for the AQL packets the library writes itself (fleet_direct.hip).
inline void describe_launch(FleetStepLaunch* L, const void* host_fn, dim3 grid, dim3 block, const FleetDev& d, const void* actions,
                            int act_mode, float* obs, double* reward, uint8_t* done, float* terminal_obs, int32_t* done_count) {
  StepKernargs a{};
  a.p_hot = d.hot; a.p_run = d.run; a.p_soh = d.soh; a.p_actions = actions; a.p_E = d.E; a.p_N = d.N; a.p_env = d.env;
  a.d_arg = d; a.actions = actions; a.act_mode = act_mode; a.K = 1;
  a.obs = obs; a.reward = reward; a.done = done; a.terminal_obs = terminal_obs; a.done_count = done_count;
  L->host_fn = host_fn; L->grid = grid.x; L->block = block.x; L->args_bytes = (unsigned)sizeof a;
  L->actions_offset[0] = (unsigned)offsetof(StepKernargs, p_actions); L->actions_offset[1] = (unsigned)offsetof(StepKernargs, actions);
  L->packed_n_offset = (unsigned)offsetof(StepKernargs, p_N);
  L->guard_offset = (unsigned)offsetof(StepKernargs, guard_bytes);
  L->rec_offset = (unsigned)offsetof(StepKernargs, rec_blocks);  // (then rec_rows and rec_rotate)
  static_assert(offsetof(StepKernargs, rec_rows) == offsetof(StepKernargs, rec_blocks) + 8 && offsetof(StepKernargs, rec_rotate) == offsetof(StepKernargs, rec_rows) + 4,
                "fleet_direct_prepare fills the three as one 16-byte piece");
  memcpy(L->args, &a, sizeof a);
}
#define FLEET_LAUNCH_SINGLE(KERNEL, GRID)                                                                                              \
  do {                                                                                                                                 \
    if (t_describe) describe_launch(t_describe, (const void*)(KERNEL), GRID, block, d, actions, f64, obs, reward, done, terminal_obs, done_count); \
    else hipLaunchKernelGGL(KERNEL, GRID, block, 0, s, FLEET_PRE_ARGS d, actions, f64, 1, obs, reward, done, terminal_obs, done_count, 0ull, nullptr, 0, 0); \
  } while (0)

template <int G, int DEG>
hipError_t launch_step_gd(const FleetDev& d, const void* actions, int act_dtype, int K, float* obs, double* reward,
                          uint8_t* done, float* terminal_obs, int32_t* done_count, hipStream_t s) {
  const int epb = kBlock / G;
  const dim3 grid((d.E + epb - 1) / epb), block(kBlock);
  const int f64 = act_dtype;  // FLEET_ACT_F32 / FLEET_ACT_F64 / FLEET_ACT_POLICY_* (policies: MULTI kernel only)
  // the single-step kernel carries neither the policies, nor the event-skipping loop, nor the data-log code
  const bool single = (K == 1 && !done_count && act_dtype < FLEET_ACT_POLICY_UNCONTROLLED && !d.real_time && !d.log_pos);
  if (t_describe && !single) return hipErrorNotSupported;  // only single-step launches are described
  // K steps per launch from a tape or a built-in policy keep one EV per lane too (not the event-skipping loop, not the data log)
  const bool many_grouped = (!single && !d.real_time && !d.log_pos);
  if (G == 64 && (single || many_grouped) && d.N > 64 && d.N <= kMaxGroup) {  // one EV per lane, two or four wavefronts per env
    constexpr int GG2 = (G == 64) ? 128 : G, GG4 = (G == 64) ? 256 : G;  // (only instantiated behind G == 64)
    if (many_grouped) {
      if (d.N <= 128) launch_many<GG2, DEG, false>(d, dim3((d.E + 1) / 2), block, actions, f64, K, obs, reward, done, terminal_obs, done_count, s);
      else launch_many<GG4, DEG, false>(d, dim3(d.E), block, actions, f64, K, obs, reward, done, terminal_obs, done_count, s);
      return hipGetLastError();
    }
    if (d.N <= 128) {
      const dim3 g2((d.E + 1) / 2);
      if (f64 == FLEET_ACT_F64)
        FLEET_LAUNCH_SINGLE((fleet_step_kernel<GG2, DEG, false, false, false, true>), g2);
      else
        FLEET_LAUNCH_SINGLE((fleet_step_kernel<GG2, DEG, false, false>), g2);
    } else {
      const dim3 g4(d.E);
      if (f64 == FLEET_ACT_F64)
        FLEET_LAUNCH_SINGLE((fleet_step_kernel<GG4, DEG, false, false, false, true>), g4);
      else
        FLEET_LAUNCH_SINGLE((fleet_step_kernel<GG4, DEG, false, false>), g4);
    }
    return hipGetLastError();
  }
  if (G == 64 && d.N > G) {  // more EVs than lanes: every lane walks several EVs
    if (single)
      FLEET_LAUNCH_SINGLE((fleet_step_kernel<G, DEG, false, (G == 64)>), grid);
    else if (d.log_pos)
      hipLaunchKernelGGL((fleet_step_kernel<G, DEG, true, (G == 64), true>), grid, block, 0, s, FLEET_PRE_ARGS d, actions, f64, K, obs, reward,
                         done, terminal_obs, done_count, 0ull, nullptr, 0, 0);
    else
      launch_many<G, DEG, (G == 64)>(d, grid, block, actions, f64, K, obs, reward, done, terminal_obs, done_count, s);
  } else {
    if (single && f64 == FLEET_ACT_F64)
      FLEET_LAUNCH_SINGLE((fleet_step_kernel<G, DEG, false, false, false, true>), grid);
    else if (single)
      FLEET_LAUNCH_SINGLE((fleet_step_kernel<G, DEG, false, false>), grid);
    else if (d.log_pos)
      hipLaunchKernelGGL((fleet_step_kernel<G, DEG, true, false, true>), grid, block, 0, s, FLEET_PRE_ARGS d, actions, f64, K, obs, reward, done,
                         terminal_obs, done_count, 0ull, nullptr, 0, 0);
    else
      launch_many<G, DEG, false>(d, grid, block, actions, f64, K, obs, reward, done, terminal_obs, done_count, s);
  }
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

// What fleet_direct_open launches on its queue before it accepts the mode: every workgroup writes down the die it runs on
// (HW_REG_XCC_ID, all bits).  A symbol with C linkage: resolved by name in the code object the HSA loader holds.
extern "C" __global__ void fleet_probe_xcc_kernel(uint32_t* __restrict__ out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
}
// Which sources this code was compiled from (fleetrl_amd/build.py passes the hash of the sources and flags to BOTH artefacts):
// fleet_direct_open reads the code object's copy through the HSA loader and refuses a code object that is not the library's twin.
#ifndef FLEET_SRC_SHA
#define FLEET_SRC_SHA "unversioned"
#endif
extern "C" __device__ __attribute__((used)) const char fleet_src_sha[32] = FLEET_SRC_SHA;
const char* fleet_kernels_src_sha() { return FLEET_SRC_SHA; }

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

// Up to this many EVs per env a single-step launch gives every EV a lane of its own (groups of 1 ... 4 wavefronts per env) and reads
// the carried schedule records; beyond it the lanes walk several EVs each and read the table.
int fleet_max_evs_per_lane_group() { return kMaxGroup; }

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

hipError_t fleet_describe_step(const FleetDev& d, const void* actions, int act_dtype, float* obs, double* reward, uint8_t* done,
                               float* terminal_obs, FleetStepLaunch* out) {
  out->host_fn = nullptr;
  t_describe = out;
  const hipError_t e = fleet_launch_step(d, actions, act_dtype, 1, obs, reward, done, terminal_obs, nullptr, nullptr);
  t_describe = nullptr;
  if (e != hipSuccess) return e;
  return out->host_fn ? hipSuccess : hipErrorNotSupported;
}

hipError_t fleet_launch_term_compact(const FleetDev& d, const uint8_t* done, const float* term, int32_t* idx, int32_t* count,
                                     double* ep_ret, int32_t* ep_len, float* compact, hipStream_t s) {
  hipLaunchKernelGGL(fleet_term_scan_kernel, dim3(1), dim3(1024), 0, s, done, d.E, idx, count, d.env, d.cold, ep_ret, ep_len);
  hipLaunchKernelGGL(fleet_term_gather_kernel, dim3(256), dim3(256), 0, s, term, d.obs_dim, idx, count, compact);
  return hipGetLastError();
}

hipError_t fleet_launch_dist_factor(const FleetDev& d, double* out, hipStream_t s) {
  const size_t n = (size_t)d.E * d.N;
  hipLaunchKernelGGL(fleet_dist_factor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, out);
  return hipGetLastError();
}

hipError_t fleet_launch_gather_field(const FleetDev& d, int field, void* out, hipStream_t s) {
  const size_t n = (size_t)d.E * d.N;
  hipLaunchKernelGGL(fleet_gather_field_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, field, out);
  return hipGetLastError();
}

hipError_t fleet_launch_selftest_stress(unsigned long long n, unsigned long long seed, unsigned long long* worst_dev, hipStream_t s) {
  hipLaunchKernelGGL(fleet_selftest_stress_kernel, dim3(2048), dim3(256), 0, s, n, seed, worst_dev);
  return hipGetLastError();
}

hipError_t fleet_launch_selftest_division(unsigned long long n, unsigned long long seed, unsigned long long* bad_dev, hipStream_t s) {
  hipLaunchKernelGGL(fleet_selftest_division_kernel, dim3(2048), dim3(256), 0, s, n, seed, bad_dev);
  return hipGetLastError();
}
