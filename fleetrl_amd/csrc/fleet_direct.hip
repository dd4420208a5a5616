// Direct AQL submission: a run of single-step launches written into HSA queues (one, or two for large batches) by the library itself
// (host code only).
//
// Why: HIP gives every kernel packet an agent-scope acquire AND release fence.  On a part whose eight dies have an L2 each, not
// coherent with one another, the release is a write-back of the die's dirty lines at the end of EVERY launch -- the next launch waits
// for it to drain and then fetches those lines again.  Between two steps of the same batch none of that is needed: workgroup w -- and
// the hardware hands workgroup w of a grid to die w mod 8 -- owns the same envs in every launch, so every byte of state a die reads
// was last written by itself (or by nobody: tables, actions).  A run submitted here keeps the acquire (the per-CU vector caches and the
// scalar caches ARE invalidated at every launch: a wavefront of env e runs on another CU of its die next time) and drops the release
// on all packets but the last, which releases at system scope: after the run every result is where any reader expects it.
// tools/ubench/aql_fence.cpp is the microbenchmark of the effect (read-modify-write of 16 MB by 4096 workgroups: 5.3 -> 2.9 us per
// launch, empty launch 1.56 -> 1.44 us; results identical over 2000 launches); tests/test_direct_gpu.py holds the step kernel to
// bit-identical state, observations and rewards against the HIP-stream path.  What the step kernel gains (in-kernel stamps,
// profiles/r05_experiments/direct_queue_stamps_by_batch.log): up to ~1000 envs x 50 EVs the state records really are served by the
// die's L2 (first loads back after 750 instead of 1140 cycles); from 2048 envs on a launch turns over more lines per die than the L2
// holds and the gain is the shorter launch floor plus the write-back that no longer has to drain between launches
// (4096 x 50: 8.0 -> 6.45 us per step).
//
// What a caller may rely on: nothing of the run is visible before it has completed (fleet_synchronize / the next call on the handle
// waits for it), everything after.  What the library relies on: the single-step kernels touch an env's state only from that env's
// own workgroup, and bytes of different envs that share a cache line are merged by the L2's byte masks -- as inside any one launch.
#include <dlfcn.h>
#include <fcntl.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <unistd.h>

#include <cstring>
#include <map>

#include "fleet_direct.h"

namespace {

struct KernelObject {
  uint64_t object = 0;
  uint32_t kernarg_bytes = 0, lds_bytes = 0, scratch_bytes = 0;
};

std::string hsa_err(const char* what, hsa_status_t s) {
  const char* m = nullptr;
  hsa_status_string(s, &m);
  return std::string(what) + ": " + (m ? m : "HSA error");
}

#define HSA_TRY(err, expr)                         \
  do {                                             \
    const hsa_status_t _s = (expr);                \
    if (_s != HSA_STATUS_SUCCESS) {                \
      if (err) *(err) = hsa_err(#expr, _s);        \
      return FLEET_ERR_HIP;                        \
    }                                              \
  } while (0)

struct AgentSearch {
  uint32_t domain, bus, dev;
  int ordinal, seen;
  hsa_agent_t by_bdf, by_ordinal;
  bool have_bdf, have_ordinal;
};

hsa_status_t find_agent(hsa_agent_t a, void* data) {
  AgentSearch* s = static_cast<AgentSearch*>(data);
  hsa_device_type_t t;
  if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
  uint32_t bdf = 0, domain = 0;
  (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
  (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
  if (!s->have_bdf && domain == s->domain && ((bdf >> 8) & 0xff) == s->bus && ((bdf >> 3) & 0x1f) == s->dev) {
    s->by_bdf = a;
    s->have_bdf = true;
  }
  if (s->seen == s->ordinal) {
    s->by_ordinal = a;
    s->have_ordinal = true;
  }
  s->seen += 1;
  return HSA_STATUS_SUCCESS;
}

// <library without ".so">.gfx950.hsaco: the code object fleetrl_amd.build compiles from the same source beside every library it builds
std::string code_object_path() {
  Dl_info info;
  std::string p = "libfleet_hip.so";
  if (dladdr(reinterpret_cast<const void*>(&fleet_direct_open), &info) && info.dli_fname) p = info.dli_fname;
  if (p.size() > 3 && p.compare(p.size() - 3, 3, ".so") == 0) p.resize(p.size() - 3);
  return p + ".gfx950.hsaco";
}

}  // namespace

struct FleetDirect {
  int device = 0;
  hsa_agent_t agent{};
  hsa_queue_t* queue[2] = {nullptr, nullptr};  // the second one is created when a run is split (fleet_direct_prepare)
  hsa_executable_t exe{};
  hsa_code_object_reader_t reader{};
  bool have_exe = false, have_reader = false, hsa_up = false;
  int fd = -1;
  uint64_t tick_hz = 0;
  std::map<std::string, KernelObject> kernels;
  // the prepared launch
  KernelObject kernel;
  unsigned block = 0;
  int parts = 1;              // 1: the whole grid on queue 0; 2: the grid as two ranges of workgroups, one per queue
  unsigned part_grid[2] = {0, 0};
  char* kargs_dev = nullptr;  // parts x tape_len blocks of kBlockBytes (part-major)
  size_t kargs_cap = 0;
  int tape_len = 0;
  static constexpr size_t kBlockBytes = 512;
  // signals: a pool; the ones handed out since the last wait; per timed run and queue (first, last) among them; per queue the newest
  // run's last
  std::vector<hsa_signal_t> pool, pending;
  struct Mark { hsa_signal_t first[2], last[2]; int parts; };
  std::vector<Mark> marks;
  hsa_signal_t last[2] = {};
  bool in_flight = false;
};

#ifdef FLEET_STAMPS
static FleetDirect* g_last_direct = nullptr;  // diagnostic build only: whose code object fleet_debug_read_stamps_direct reads
#endif

static hsa_signal_t take_signal(FleetDirect* q) {
  hsa_signal_t s{};
  if (!q->pool.empty()) {
    s = q->pool.back();
    q->pool.pop_back();
    hsa_signal_store_relaxed(s, 1);
    return s;
  }
  if (hsa_signal_create(1, 0, nullptr, &s) != HSA_STATUS_SUCCESS) s.handle = 0;
  return s;
}

static int open_queue(FleetDirect* q, int k, std::string* err) {
  if (q->queue[k]) return FLEET_OK;
  uint32_t qmax = 0;
  (void)hsa_agent_get_info(q->agent, HSA_AGENT_INFO_QUEUE_MAX_SIZE, &qmax);
  uint32_t qsize = 16384;
  while (qmax && qsize > qmax) qsize >>= 1;
  hsa_status_t st = hsa_queue_create(q->agent, qsize, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q->queue[k]);
  if (st != HSA_STATUS_SUCCESS) {
    q->queue[k] = nullptr;
    if (err) *err = hsa_err("hsa_queue_create", st);
    return FLEET_ERR_HIP;
  }
  // dispatch timestamps in the completion signals (read for the packets that carry one: the first and last of a timed run).  Enabled
  // before the queue's first packet: the switch is not seen by a queue that is already running.
  if ((st = hsa_amd_profiling_set_profiler_enabled(q->queue[k], 1)) != HSA_STATUS_SUCCESS) {
    if (err) *err = hsa_err("hsa_amd_profiling_set_profiler_enabled", st);
    return FLEET_ERR_HIP;
  }
  return FLEET_OK;
}

int fleet_direct_open(int hip_device, FleetDirect** out, std::string* err) {
  *out = nullptr;
  FleetDirect* q = new FleetDirect();
  q->device = hip_device;
  auto fail = [&](int rc) {
    fleet_direct_close(q);
    return rc;
  };
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, hip_device) != hipSuccess) {
    if (err) *err = "fleet_direct_open: hipGetDeviceProperties failed";
    return fail(FLEET_ERR_HIP);
  }
  hsa_status_t st = hsa_init();
  if (st != HSA_STATUS_SUCCESS) {
    if (err) *err = hsa_err("hsa_init", st);
    return fail(FLEET_ERR_HIP);
  }
  q->hsa_up = true;
  AgentSearch s{};
  s.domain = (uint32_t)prop.pciDomainID; s.bus = (uint32_t)prop.pciBusID; s.dev = (uint32_t)prop.pciDeviceID;
  s.ordinal = hip_device;
  st = hsa_iterate_agents(find_agent, &s);
  if (st != HSA_STATUS_SUCCESS || !(s.have_bdf || s.have_ordinal)) {
    if (err) *err = "fleet_direct_open: no HSA agent for the HIP device";
    return fail(FLEET_ERR_HIP);
  }
  q->agent = s.have_bdf ? s.by_bdf : s.by_ordinal;
  char isa[64] = {0};
  (void)hsa_agent_get_info(q->agent, HSA_AGENT_INFO_NAME, isa);
  if (strncmp(isa, "gfx950", 6) != 0) {
    if (err) *err = std::string("fleet_direct_open: the agent is ") + isa + ", the code object is gfx950";
    return fail(FLEET_ERR_INVALID);
  }
  (void)hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &q->tick_hz);
  const int rcq = open_queue(q, 0, err);
  if (rcq != FLEET_OK) return fail(rcq);
  const std::string path = code_object_path();
  q->fd = open(path.c_str(), O_RDONLY);
  if (q->fd < 0) {
    if (err) *err = "fleet_direct_open: cannot read " + path + " (fleetrl_amd.build builds it beside the library)";
    return fail(FLEET_ERR_INVALID);
  }
  if ((st = hsa_code_object_reader_create_from_file(q->fd, &q->reader)) != HSA_STATUS_SUCCESS) {
    if (err) *err = hsa_err("hsa_code_object_reader_create_from_file", st);
    return fail(FLEET_ERR_HIP);
  }
  q->have_reader = true;
  if ((st = hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &q->exe)) != HSA_STATUS_SUCCESS) {
    if (err) *err = hsa_err("hsa_executable_create_alt", st);
    return fail(FLEET_ERR_HIP);
  }
  q->have_exe = true;
  if ((st = hsa_executable_load_agent_code_object(q->exe, q->agent, q->reader, nullptr, nullptr)) != HSA_STATUS_SUCCESS ||
      (st = hsa_executable_freeze(q->exe, nullptr)) != HSA_STATUS_SUCCESS) {
    if (err) *err = hsa_err("loading the step kernels' code object", st);
    return fail(FLEET_ERR_HIP);
  }
#ifdef FLEET_STAMPS
  g_last_direct = q;
#endif
  *out = q;
  return FLEET_OK;
}

void fleet_direct_close(FleetDirect* q) {
  if (!q) return;
  if (q->in_flight) (void)fleet_direct_wait(q, nullptr, nullptr);
#ifdef FLEET_STAMPS
  if (g_last_direct == q) g_last_direct = nullptr;
#endif
  for (hsa_queue_t* hq : q->queue)
    if (hq) (void)hsa_queue_destroy(hq);
  for (hsa_signal_t s : q->pool) (void)hsa_signal_destroy(s);
  if (q->have_exe) (void)hsa_executable_destroy(q->exe);
  if (q->have_reader) (void)hsa_code_object_reader_destroy(q->reader);
  if (q->fd >= 0) close(q->fd);
  if (q->kargs_dev) (void)hipFree(q->kargs_dev);
  if (q->hsa_up) (void)hsa_shut_down();
  delete q;
}

#ifdef FLEET_STAMPS
// Diagnostic build only (tools/stamps.py): the stamp buffer of the code object THIS queue's kernels run from (the HSA-loaded copy has
// a buffer of its own, which hipMemcpyFromSymbol does not see)
extern "C" int fleet_debug_read_stamps_direct(unsigned long long* out, size_t bytes) {
  if (!g_last_direct) return -1;
  hsa_executable_symbol_t sym;
  uint64_t addr = 0;
  if (hsa_executable_get_symbol_by_name(g_last_direct->exe, "fleet_stamp_buf", &g_last_direct->agent, &sym) != HSA_STATUS_SUCCESS ||
      hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_VARIABLE_ADDRESS, &addr) != HSA_STATUS_SUCCESS)
    return -2;
  return (int)hipMemcpy(out, reinterpret_cast<const void*>(addr), bytes, hipMemcpyDeviceToHost);
}
#endif

int fleet_direct_prepare(FleetDirect* q, const FleetStepLaunch& L, const void* tape, int tape_len, size_t row_bytes, bool split,
                         std::string* err) {
  if (!q || !L.host_fn || tape_len < 1 || L.args_bytes > FleetDirect::kBlockBytes) return FLEET_ERR_INVALID;
  if (q->in_flight) {
    if (err) *err = "fleet_direct_prepare: a run is in flight";
    return FLEET_ERR_STATE;
  }
  const char* name = hipKernelNameRefByPtr(L.host_fn, nullptr);
  if (!name) {
    if (err) *err = "fleet_direct_prepare: the kernel has no name";
    return FLEET_ERR_HIP;
  }
  auto it = q->kernels.find(name);
  if (it == q->kernels.end()) {
    hsa_executable_symbol_t sym;
    KernelObject k;
    const std::string kd = std::string(name) + ".kd";
    HSA_TRY(err, hsa_executable_get_symbol_by_name(q->exe, kd.c_str(), &q->agent, &sym));
    HSA_TRY(err, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    HSA_TRY(err, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_bytes));
    HSA_TRY(err, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.lds_bytes));
    HSA_TRY(err, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.scratch_bytes));
    it = q->kernels.emplace(name, k).first;
  }
  if (it->second.kernarg_bytes != L.args_bytes) {  // the code object beside the library is not the one this library was built with
    if (err) *err = "fleet_direct_prepare: the code object's argument segment is " + std::to_string(it->second.kernarg_bytes) +
                    " bytes, the library's " + std::to_string(L.args_bytes);
    return FLEET_ERR_STATE;
  }
  q->kernel = it->second;
  q->block = L.block;
  // one grid on one queue -- or, for a batch of more wavefronts than are resident at once, two ranges of workgroups on two queues,
  // each an in-order chain of its own: the two halves drift apart and one's loads run under the other's arithmetic and stores
  // (16384 x 50: 24.5 -> 20.9 us per step; no gain at 4096 x 50, profiles/r05_experiments/direct_queue_two_handles.log).  The second
  // range's workgroups continue the numbering of the first (the kernel's packed `p_N`), so env e stays workgroup e / (256 / G) and
  // keeps its die; the first range is a multiple of 8 workgroups.
  q->parts = (split && L.grid >= 16) ? 2 : 1;
  if (q->parts == 2 && open_queue(q, 1, nullptr) != FLEET_OK) q->parts = 1;  // no second queue to be had: one chain, as for small batches
  if (q->parts == 2) {
    q->part_grid[0] = ((L.grid / 2 + 7) / 8) * 8;
    q->part_grid[1] = L.grid - q->part_grid[0];
  } else {
    q->part_grid[0] = L.grid;
    q->part_grid[1] = 0;
  }
  const size_t need = (size_t)q->parts * tape_len * FleetDirect::kBlockBytes;
  if (need > q->kargs_cap) {
    if (q->kargs_dev) (void)hipFree(q->kargs_dev);
    q->kargs_dev = nullptr;
    q->kargs_cap = 0;
    if (hipMalloc(reinterpret_cast<void**>(&q->kargs_dev), need) != hipSuccess) {
      if (err) *err = "fleet_direct_prepare: out of device memory";
      return FLEET_ERR_HIP;
    }
    q->kargs_cap = need;
  }
  std::vector<unsigned char> host(need, 0);
  for (int part = 0; part < q->parts; ++part)
    for (int k = 0; k < tape_len; ++k) {
      unsigned char* b = host.data() + ((size_t)part * tape_len + k) * FleetDirect::kBlockBytes;
      memcpy(b, L.args, L.args_bytes);
      const void* row = static_cast<const char*>(tape) + (size_t)k * row_bytes;
      memcpy(b + L.actions_offset[0], &row, sizeof row);
      memcpy(b + L.actions_offset[1], &row, sizeof row);
      if (part == 1) {
        int32_t packed;
        memcpy(&packed, b + L.packed_n_offset, 4);
        packed = (int32_t)(((uint32_t)packed & 0xffffu) | (q->part_grid[0] << 16));
        memcpy(b + L.packed_n_offset, &packed, 4);
      }
    }
  if (hipMemcpy(q->kargs_dev, host.data(), need, hipMemcpyHostToDevice) != hipSuccess) {
    if (err) *err = "fleet_direct_prepare: argument upload failed";
    return FLEET_ERR_HIP;
  }
  q->tape_len = tape_len;
  return FLEET_OK;
}

int fleet_direct_submit(FleetDirect* q, int steps, bool timed, std::string* err) {
  if (!q || !q->queue[0] || q->tape_len < 1 || steps < 1) return FLEET_ERR_INVALID;
  if (q->pending.size() > 4096) {  // a caller that submits run after run without ever waiting (more than any timed series: <= 256 regions): a wait recycles the signals
    const int rc = fleet_direct_wait(q, nullptr, err);
    if (rc != FLEET_OK) return rc;
  }
  FleetDirect::Mark m{};
  m.parts = q->parts;
  for (int part = 0; part < q->parts; ++part) {
    m.last[part] = take_signal(q);
    m.first[part] = timed ? take_signal(q) : hsa_signal_t{};
    if (!m.last[part].handle || (timed && !m.first[part].handle)) {
      if (err) *err = "fleet_direct_submit: hsa_signal_create failed";
      return FLEET_ERR_HIP;
    }
    q->pending.push_back(m.last[part]);
    if (timed) q->pending.push_back(m.first[part]);
  }
  for (int i = 0; i < steps; ++i)
    for (int part = 0; part < q->parts; ++part) {  // step by step, queue by queue: both chains get going at once
      hsa_queue_t* hq = q->queue[part];
      const uint64_t idx = hsa_queue_add_write_index_relaxed(hq, 1);
      while (idx - hsa_queue_load_read_index_scacquire(hq) >= hq->size) {}
      hsa_kernel_dispatch_packet_t* p = static_cast<hsa_kernel_dispatch_packet_t*>(hq->base_address) + (idx & (hq->size - 1));
      p->workgroup_size_x = (uint16_t)q->block; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
      p->grid_size_x = q->part_grid[part] * q->block; p->grid_size_y = 1; p->grid_size_z = 1;
      p->private_segment_size = q->kernel.scratch_bytes;
      p->group_segment_size = q->kernel.lds_bytes;
      p->kernel_object = q->kernel.object;
      p->kernarg_address = q->kargs_dev + ((size_t)part * q->tape_len + (size_t)(i % q->tape_len)) * FleetDirect::kBlockBytes;
      p->reserved2 = 0;
      hsa_signal_t none{};
      p->completion_signal = (i == steps - 1) ? m.last[part] : ((timed && i == 0) ? m.first[part] : none);
      // the first packet of a run acquires at system scope (whatever the host or another queue wrote before the run), the others at
      // agent scope; only the last one releases
      const int acq = (i == 0) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
      const int rel = (i == steps - 1) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE;
      const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                         (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
      const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
      __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
      hsa_signal_store_screlease(hq->doorbell_signal, (hsa_signal_value_t)idx);
    }
  if (timed) {
    if (steps == 1)  // (one packet cannot carry two signals: its own start and end are the span then)
      for (int part = 0; part < q->parts; ++part) m.first[part] = m.last[part];
    q->marks.push_back(m);
  }
  for (int part = 0; part < 2; ++part) q->last[part] = part < q->parts ? m.last[part] : hsa_signal_t{};
  q->in_flight = true;
  return FLEET_OK;
}

int fleet_direct_parts(FleetDirect* q) { return q ? q->parts : 0; }

bool fleet_direct_busy(FleetDirect* q) {
  if (!q || !q->in_flight) return false;
  for (hsa_signal_t sgn : q->last)
    if (sgn.handle && hsa_signal_load_scacquire(sgn) >= 1) return true;
  return false;
}

int fleet_direct_wait(FleetDirect* q, std::vector<double>* spans_us, std::string* err) {
  if (!q) return FLEET_ERR_INVALID;
  if (q->in_flight) {
    // spin for the first millisecond (a caller that polls -- fleet_stream_query -- never gets here before the run is done; one that
    // synchronises right after a short run wants it back within microseconds), then sleep on the signal's interrupt.
    // 60 s in all: a run of 16384 launches of the largest batch is ~1 s
    const uint64_t hz = q->tick_hz ? q->tick_hz : 100000000ull;
    for (hsa_signal_t sgn : q->last) {
      if (!sgn.handle) continue;
      if (hsa_signal_wait_scacquire(sgn, HSA_SIGNAL_CONDITION_LT, 1, hz / 1000, HSA_WAIT_STATE_ACTIVE) >= 1 &&
          hsa_signal_wait_scacquire(sgn, HSA_SIGNAL_CONDITION_LT, 1, hz * 60ull, HSA_WAIT_STATE_BLOCKED) >= 1) {
        if (err) *err = "fleet_direct_wait: the run did not complete within 60 s";
        return FLEET_ERR_HIP;
      }
    }
    q->in_flight = false;
  }
  for (const FleetDirect::Mark& m : q->marks) {  // a timed run: from the earlier start of its first launches to the later end of its last
    uint64_t t0 = UINT64_MAX, t1 = 0;
    bool ok = true;
    for (int part = 0; part < m.parts; ++part) {
      hsa_amd_profiling_dispatch_time_t a{}, b{};
      ok = ok && hsa_amd_profiling_get_dispatch_time(q->agent, m.first[part], &a) == HSA_STATUS_SUCCESS &&
           hsa_amd_profiling_get_dispatch_time(q->agent, m.last[part], &b) == HSA_STATUS_SUCCESS;
      t0 = a.start < t0 ? a.start : t0;
      t1 = b.end > t1 ? b.end : t1;
    }
    if (spans_us) spans_us->push_back(ok && q->tick_hz && t1 > t0 ? (double)(t1 - t0) * 1e6 / (double)q->tick_hz : -1.0);
  }
  q->marks.clear();
  for (hsa_signal_t sgn : q->pending) q->pool.push_back(sgn);
  q->pending.clear();
  q->last[0].handle = q->last[1].handle = 0;
  return FLEET_OK;
}
