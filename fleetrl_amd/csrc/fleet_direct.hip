// Direct AQL submission: single-step launches written into HSA queues (one, or two for large batches) by the library itself
// (host code only).
//
// Why: HIP gives every kernel packet an agent-scope acquire AND release fence.  On a part whose eight dies have an L2 each, not
// coherent with one another, the release is a write-back of the die's dirty lines at the end of EVERY launch -- the next launch waits
// for it to drain and then fetches those lines again.  Between two steps of the same batch none of that is needed: workgroup w owns
// the same envs in every launch and the hardware hands workgroup w of every grid to the same die, so every byte of state a die reads
// was last written by itself (or by nobody: tables, actions).  A run submitted here keeps the acquire (the per-CU vector caches and
// the scalar caches ARE invalidated at every launch: a wavefront of env e runs on another CU of its die next time) and drops the
// release on all packets but the last, which releases at system scope: after the run every result is where any reader expects it.
// tools/ubench/aql_fence.cpp is the microbenchmark of the effect (read-modify-write of 16 MB by 4096 workgroups: 5.3 -> 2.9 us per
// launch, empty launch 1.56 -> 1.44 us; results identical over 2000 launches); tests/test_direct_gpu.py holds the step kernel to
// bit-identical state, observations and rewards against the HIP-stream path.
//
// THE ASSUMPTION, AND WHAT GUARDS IT.  "Workgroup w of every launch runs on the same die" is how the hardware is observed to deal
// workgroups (round-robin over the dies, dispatch after dispatch from the same die of the queue: tools/ubench/xcc_map.cpp) -- it is not a
// documented contract (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").  So:
//   * fleet_direct_open PROBES it on the queue it has just created -- a chain of launches, with and without a multiple of 8
//     workgroups, each workgroup writing down HW_REG_XCC_ID -- and refuses the mode (FLEET_ERR_UNSUPPORTED; callers fall back to HIP's
//     launches) unless the map workgroup -> die is periodic in 8 and identical from launch to launch;
//   * the die a queue starts dealing from is NOT a constant of the queue: it moves by one whenever a queue is created or destroyed in
//     the process (found by this library's own tests: the second queue of a split run moved the first one's; then measured,
//     tools/ubench/xcc_map.cpp).  So the FIRST launch of every run -- a chain of launches that ends with the release; its first
//     launch reads what the previous release made everybody's and may sit anywhere -- writes the dies of its first eight workgroups,
//     as they are then, into the argument blocks of the run's other launches, and every other launch compares the die it finds
//     itself on with its slot of that record
//     (fleet_kernels.hip, "Placement guard": one scalar load from the argument segment and one s_getreg at the end of the step): a
//     launch that lands elsewhere raises FLEET_DEVERR_PLACEMENT, which fleet_check_errors / the host step report as a hard error.
//     A run whose queue moved in its middle is void and says so; between runs a move is harmless;
//   * the agent is the one with the HIP device's PCI address -- never "the n-th GPU agent" (HIP and HSA enumerate differently under
//     *_VISIBLE_DEVICES);
//   * the code object must carry the source hash the library was compiled with.
//
// What a caller may rely on: nothing of the run is visible before it has completed (fleet_synchronize / the next call on the handle
// waits for it), everything after.  What the library relies on beside the placement: the single-step kernels touch an env's state only
// from that env's own workgroup, and bytes of different envs that share a cache line are merged by the L2's byte masks -- as inside
// any one launch.
//
// What was tried on top and is NOT here (round 6, profiles/r06_experiments/closed_loop_write_through.log): a CLOSED-LOOP step on this
// queue -- no release, the observations / rewards / done flags of every step stored write-through so that a policy could consume them
// while the state stays in the L2s.  Built to green parity (a torch policy reading every step's observation, 4096 x 50, against the
// stream launches bit for bit and against the CPU oracle) and measured: write-through stores run at the memory side's ~1.3 TB/s, the
// 6.3 MB of observations per launch cost 2.2 us (6.5 -> 8.7 us) -- more than HIP's whole release (8.0 us per step).  A step whose
// outputs must be visible before the next one starts pays for making them visible; the cheapest way to do that is the release fence,
// i.e. fleet_step_dev.
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
  hsa_agent_t by_bdf;
  int matches, gpus;
};

hsa_status_t find_agent(hsa_agent_t a, void* data) {
  AgentSearch* s = static_cast<AgentSearch*>(data);
  hsa_device_type_t t;
  if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
  s->gpus += 1;
  uint32_t bdf = 0, domain = 0;
  if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS ||
      hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain) != HSA_STATUS_SUCCESS)
    return HSA_STATUS_SUCCESS;
  if (domain == s->domain && ((bdf >> 8) & 0xff) == s->bus && ((bdf >> 3) & 0x1f) == s->dev) {
    if (s->matches == 0) s->by_bdf = a;
    s->matches += 1;
  }
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

constexpr unsigned kProbeLaunches = 6, kProbeMaxGrid = 2048;
// a chain shaped like a run: whole multiples of 8 workgroups, then grids that are not, then multiples again
constexpr unsigned kProbeGrids[kProbeLaunches] = {1024, 1024, 1027, 1027, 1024, 1024};

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
  // placement (see the header comment)
  KernelObject probe_kernel;       // fleet_probe_xcc_kernel: the open-time probe
  uint32_t* probe_out = nullptr;   // device: kProbeLaunches x kProbeMaxGrid words
  char* probe_kargs = nullptr;     // device: one 64-byte argument block per probe launch
  uint32_t guard[2] = {0, 0};      // what the probe found per queue (bit 31 + eight 3-bit dies): information, see fleet_direct_probed
  bool probed[2] = {false, false};
  bool any_grid = false;           // the map also held across grids that are not multiples of 8 workgroups (both queues)
  uint32_t num_xcc = 0;
  int fault_rotate = 0;            // test hook: the next run's placement record is shifted by one workgroup
  // the prepared launch
  KernelObject kernel;
  unsigned block = 0;
  int parts = 1;              // 1: the whole grid on queue 0; 2: the grid as two ranges of workgroups, one per queue
  unsigned part_grid[2] = {0, 0};
  char* kargs_dev = nullptr;  // parts x tape_len blocks of kBlockBytes (part-major), then per part the block of the run's FIRST launch
                              // (tape row 0's with the recording arguments: it writes the placement record into the others)
  size_t kargs_cap = 0;
  int tape_len = 0;
  std::vector<unsigned char> kargs_host;  // what was uploaded (the fault hook patches a block of it)
  unsigned packed_n_offset = 0, guard_offset = 0, rec_offset = 0;
  static constexpr size_t kBlockBytes = sizeof(FleetStepLaunch::args);
  // signals: a pool; the ones handed out since the last wait; per timed run what its spans are read from
  std::vector<hsa_signal_t> pool, pending;
  struct Mark { hsa_signal_t first[2], last[2]; int parts; std::vector<hsa_signal_t> each; };
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

static void write_packet(hsa_queue_t* hq, const KernelObject& k, unsigned block, unsigned grid_wgs, void* kernarg, int acq, int rel,
                         hsa_signal_t done) {
  const uint64_t idx = hsa_queue_add_write_index_relaxed(hq, 1);
  while (idx - hsa_queue_load_read_index_scacquire(hq) >= hq->size) {}
  hsa_kernel_dispatch_packet_t* p = static_cast<hsa_kernel_dispatch_packet_t*>(hq->base_address) + (idx & (hq->size - 1));
  p->workgroup_size_x = (uint16_t)block; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->reserved0 = 0;
  p->grid_size_x = grid_wgs * block; p->grid_size_y = 1; p->grid_size_z = 1;
  p->private_segment_size = k.scratch_bytes;
  p->group_segment_size = k.lds_bytes;
  p->kernel_object = k.object;
  p->kernarg_address = kernarg;
  p->reserved2 = 0;
  p->completion_signal = done;
  // barrier bit: launch n+1 starts when launch n has completed
  const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                     (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
  const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
  __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
  hsa_signal_store_screlease(hq->doorbell_signal, (hsa_signal_value_t)idx);
}

static int kernel_by_name(FleetDirect* q, const std::string& name, KernelObject* out, std::string* err) {
  auto it = q->kernels.find(name);
  if (it == q->kernels.end()) {
    hsa_executable_symbol_t sym;
    KernelObject k;
    const std::string kd = name + ".kd";
    HSA_TRY(err, hsa_executable_get_symbol_by_name(q->exe, kd.c_str(), &q->agent, &sym));
    HSA_TRY(err, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    HSA_TRY(err, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_bytes));
    HSA_TRY(err, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.lds_bytes));
    HSA_TRY(err, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.scratch_bytes));
    it = q->kernels.emplace(name, k).first;
  }
  *out = it->second;
  return FLEET_OK;
}

static int open_queue(FleetDirect* q, int k, std::string* err) {
  if (q->queue[k]) return FLEET_OK;
  uint32_t qmax = 0;
  (void)hsa_agent_get_info(q->agent, HSA_AGENT_INFO_QUEUE_MAX_SIZE, &qmax);
  uint32_t qsize = 16384;
  while (qmax && qsize > qmax) qsize >>= 1;
  hsa_queue_t* hq = nullptr;
  hsa_status_t st = hsa_queue_create(q->agent, qsize, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &hq);
  if (st != HSA_STATUS_SUCCESS) {
    if (err) *err = hsa_err("hsa_queue_create", st);
    return FLEET_ERR_HIP;
  }
  // dispatch timestamps in the completion signals (read for the packets that carry one).  Enabled before the queue's first packet:
  // the switch is not seen by a queue that is already running.  A queue without them is not kept (its spans would read as -1).
  if ((st = hsa_amd_profiling_set_profiler_enabled(hq, 1)) != HSA_STATUS_SUCCESS) {
    (void)hsa_queue_destroy(hq);
    if (err) *err = hsa_err("hsa_amd_profiling_set_profiler_enabled", st);
    return FLEET_ERR_HIP;
  }
  q->queue[k] = hq;
  return FLEET_OK;
}

// The placement probe of queue k (header comment): kProbeLaunches dependent launches shaped like a run, every workgroup writing down
// HW_REG_XCC_ID.  FLEET_OK: q->guard[k] holds the map; FLEET_ERR_UNSUPPORTED: the placement is not what the mode needs.
static int probe_queue(FleetDirect* q, int k, std::string* err) {
  const size_t words = (size_t)kProbeLaunches * kProbeMaxGrid;
  if (!q->probe_out) {
    if (hipMalloc(reinterpret_cast<void**>(&q->probe_out), words * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&q->probe_kargs), kProbeLaunches * 64) != hipSuccess) {
      if (err) *err = "fleet_direct_open: out of device memory (placement probe)";
      return FLEET_ERR_HIP;
    }
    std::vector<unsigned char> hk(kProbeLaunches * 64, 0);
    for (unsigned i = 0; i < kProbeLaunches; ++i) {
      uint32_t* p = q->probe_out + (size_t)i * kProbeMaxGrid;
      memcpy(hk.data() + (size_t)i * 64, &p, sizeof p);
    }
    if (hipMemcpy(q->probe_kargs, hk.data(), hk.size(), hipMemcpyHostToDevice) != hipSuccess) {
      if (err) *err = "fleet_direct_open: argument upload failed (placement probe)";
      return FLEET_ERR_HIP;
    }
  }
  if (hipMemset(q->probe_out, 0xff, words * 4) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    if (err) *err = "fleet_direct_open: hipMemset failed (placement probe)";
    return FLEET_ERR_HIP;
  }
  hsa_signal_t done = take_signal(q);
  if (!done.handle) {
    if (err) *err = "fleet_direct_open: hsa_signal_create failed";
    return FLEET_ERR_HIP;
  }
  for (unsigned i = 0; i < kProbeLaunches; ++i) {
    const bool last = (i + 1 == kProbeLaunches);
    write_packet(q->queue[k], q->probe_kernel, 256, kProbeGrids[i], q->probe_kargs + (size_t)i * 64,
                 i == 0 ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT, last ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE,
                 last ? done : hsa_signal_t{});
  }
  const uint64_t hz = q->tick_hz ? q->tick_hz : 100000000ull;
  const bool timeout = hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, hz * 5ull, HSA_WAIT_STATE_BLOCKED) >= 1;
  q->pool.push_back(done);
  if (timeout) {
    if (err) *err = "fleet_direct_open: the placement probe did not complete within 5 s";
    return FLEET_ERR_HIP;
  }
  std::vector<uint32_t> host(words);
  if (hipMemcpy(host.data(), q->probe_out, words * 4, hipMemcpyDeviceToHost) != hipSuccess) {
    if (err) *err = "fleet_direct_open: reading the placement probe failed";
    return FLEET_ERR_HIP;
  }
  unsigned map8[8];
  bool multiples_ok = true, odd_ok = true;
  std::string what;
  for (unsigned i = 0; i < kProbeLaunches; ++i) {
    const uint32_t* o = host.data() + (size_t)i * kProbeMaxGrid;
    const bool multiple = (kProbeGrids[i] % 8 == 0);
    bool ok = true;
    for (unsigned w = 0; w < kProbeGrids[i] && ok; ++w) {
      const unsigned x = o[w] & 0xfu;
      if (o[w] == 0xffffffffu || x > 7u) ok = false;                     // the workgroup did not run / an id the guard word cannot hold
      else if (i == 0 && w < 8) map8[w] = x;
      else if (x != map8[w & 7]) ok = false;                              // not periodic in 8, or not the first launch's map
    }
    if (!ok) {
      (multiple ? multiples_ok : odd_ok) = false;
      what += (what.empty() ? "launch " : ", ") + std::to_string(i) + " (" + std::to_string(kProbeGrids[i]) + " workgroups)";
    }
  }
  if (!multiples_ok) {
    if (err) *err = "fleet_direct_open: workgroup w of consecutive launches on queue " + std::to_string(k) +
                    " does not stay on one die (placement probe: " + what + "): the mode is not safe on this platform";
    return FLEET_ERR_UNSUPPORTED;
  }
  uint32_t g = 0x80000000u;
  for (unsigned j = 0; j < 8; ++j) g |= (map8[j] & 7u) << (3 * j);
  q->guard[k] = g;
  q->probed[k] = true;
  q->any_grid = (k == 0 ? true : q->any_grid) && odd_ok;
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
  // the agent with the HIP device's PCI address -- and only that one: HIP and HSA number the devices differently as soon as
  // HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES re-map them, so "the n-th GPU agent" could be another GPU
  AgentSearch s{};
  s.domain = (uint32_t)prop.pciDomainID; s.bus = (uint32_t)prop.pciBusID; s.dev = (uint32_t)prop.pciDeviceID;
  st = hsa_iterate_agents(find_agent, &s);
  if (st != HSA_STATUS_SUCCESS || s.matches != 1) {
    char bdf[64];
    snprintf(bdf, sizeof bdf, "%04x:%02x:%02x", s.domain, s.bus, s.dev);
    if (err) *err = std::string("fleet_direct_open: ") + (s.matches == 0 ? "no" : "more than one") + " HSA GPU agent at the HIP device's PCI address " +
                    bdf + " (" + std::to_string(s.gpus) + " GPU agents visible)";
    return fail(FLEET_ERR_UNSUPPORTED);
  }
  q->agent = s.by_bdf;
  char isa[64] = {0};
  (void)hsa_agent_get_info(q->agent, HSA_AGENT_INFO_NAME, isa);
  if (strncmp(isa, "gfx950", 6) != 0) {
    if (err) *err = std::string("fleet_direct_open: the agent is ") + isa + ", the code object is gfx950";
    return fail(FLEET_ERR_UNSUPPORTED);
  }
  (void)hsa_agent_get_info(q->agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_NUM_XCC, &q->num_xcc);
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
  {
    // the code object must be the library's twin: both carry the hash of the sources and flags they were compiled from
    hsa_executable_symbol_t sym;
    uint64_t addr = 0;
    char sha[32] = {0};
    if (hsa_executable_get_symbol_by_name(q->exe, "fleet_src_sha", &q->agent, &sym) != HSA_STATUS_SUCCESS ||
        hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_VARIABLE_ADDRESS, &addr) != HSA_STATUS_SUCCESS ||
        hipMemcpy(sha, reinterpret_cast<const void*>(addr), sizeof sha, hipMemcpyDeviceToHost) != hipSuccess) {
      if (err) *err = "fleet_direct_open: " + path + " carries no source hash (built from older sources than the library)";
      return fail(FLEET_ERR_STATE);
    }
    sha[sizeof sha - 1] = 0;
    if (strcmp(sha, fleet_kernels_src_sha()) != 0) {
      if (err) *err = "fleet_direct_open: " + path + " was compiled from other sources (" + sha + ") than the library (" +
                      fleet_kernels_src_sha() + "); rebuild both with fleetrl_amd.build";
      return fail(FLEET_ERR_STATE);
    }
  }
  int rc = kernel_by_name(q, "fleet_probe_xcc_kernel", &q->probe_kernel, err);
  if (rc != FLEET_OK) return fail(rc);
  if ((rc = probe_queue(q, 0, err)) != FLEET_OK) return fail(rc);
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
  if (q->probe_out) (void)hipFree(q->probe_out);
  if (q->probe_kargs) (void)hipFree(q->probe_kargs);
  if (q->hsa_up) (void)hsa_shut_down();
  delete q;
}

int fleet_direct_probed(FleetDirect* q, int map[8], int* num_xcc, int* any_grid) {
  if (!q) return FLEET_ERR_INVALID;
  for (int j = 0; j < 8; ++j) map[j] = q->probed[0] ? (int)((q->guard[0] >> (3 * j)) & 7u) : -1;
  if (num_xcc) *num_xcc = (int)q->num_xcc;
  if (any_grid) *any_grid = q->any_grid ? 1 : 0;
  return FLEET_OK;
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

int fleet_direct_plan(unsigned grid, bool split, unsigned part_grid[2]) {
  part_grid[0] = grid;
  part_grid[1] = 0;
  if (!split || grid < 16) return 1;
  // The second range's workgroups continue the numbering of the first (the kernel's packed `p_N`: EVs per env in the low half, the
  // range's first workgroup in the high half), so env e stays workgroup e / (256 / G) and keeps its die; the first range is a multiple
  // of 8 workgroups.  A first range that does not fit the 16-bit field is not split at all (grids above ~131 000 workgroups
  // would have wrapped it and stepped the first half twice).
  const unsigned first = ((grid / 2 + 7) / 8) * 8;
  if (first > 0xffffu || first >= grid) return 1;
  part_grid[0] = first;
  part_grid[1] = grid - first;
  return 2;
}

int fleet_direct_prepare(FleetDirect* q, const FleetStepLaunch& L, const void* tape, int tape_len, size_t row_bytes, bool split,
                         std::string* err) {
  if (!q || !L.host_fn || tape_len < 1 || L.args_bytes > FleetDirect::kBlockBytes) return FLEET_ERR_INVALID;
  if (q->in_flight) {
    if (err) *err = "fleet_direct_prepare: launches are in flight";
    return FLEET_ERR_STATE;
  }
  const char* name = hipKernelNameRefByPtr(L.host_fn, nullptr);
  if (!name) {
    if (err) *err = "fleet_direct_prepare: the kernel has no name";
    return FLEET_ERR_HIP;
  }
  KernelObject kernel;
  int rc = kernel_by_name(q, name, &kernel, err);
  if (rc != FLEET_OK) return rc;
  if (kernel.kernarg_bytes != L.args_bytes) {  // (cannot happen behind the source-hash check of fleet_direct_open)
    if (err) *err = "fleet_direct_prepare: the code object's argument segment is " + std::to_string(kernel.kernarg_bytes) +
                    " bytes, the library's " + std::to_string(L.args_bytes);
    return FLEET_ERR_STATE;
  }
  // one grid on one queue -- or, for a batch of more wavefronts than are resident at once, two ranges of workgroups on two queues,
  // each an in-order chain of its own: the two halves drift apart and one's loads run under the other's arithmetic and stores
  // (16384 x 50: 24.5 -> 20.9 us per step; no gain at 4096 x 50, profiles/r05_experiments/direct_queue_two_handles.log)
  unsigned part_grid[2];
  int parts = fleet_direct_plan(L.grid, split, part_grid);
  if (parts == 2) {
    // no second queue to be had, or one whose placement probe fails: one chain, as for small batches
    if (open_queue(q, 1, nullptr) != FLEET_OK) parts = 1;
    else if (!q->probed[1] && probe_queue(q, 1, nullptr) != FLEET_OK) parts = 1;
    if (parts == 1) {
      part_grid[0] = L.grid;
      part_grid[1] = 0;
    }
  }
  if (!q->any_grid)
    for (int part = 0; part < parts; ++part)
      if (part_grid[part] % 8 != 0) {
        if (err) *err = "fleet_direct_prepare: on this platform the placement only holds for grids of whole multiples of 8 workgroups (this one: " +
                        std::to_string(part_grid[part]) + ")";
        return FLEET_ERR_UNSUPPORTED;
      }
  const size_t blocks_bytes = (size_t)parts * tape_len * FleetDirect::kBlockBytes;
  const size_t need = blocks_bytes + (size_t)parts * FleetDirect::kBlockBytes;
  std::vector<unsigned char> host(need, 0);
  for (int part = 0; part < parts; ++part)
    for (int k = 0; k < tape_len; ++k) {
      unsigned char* b = host.data() + ((size_t)part * tape_len + k) * FleetDirect::kBlockBytes;
      memcpy(b, L.args, L.args_bytes);
      const void* row = static_cast<const char*>(tape) + (size_t)k * row_bytes;
      memcpy(b + L.actions_offset[0], &row, sizeof row);
      memcpy(b + L.actions_offset[1], &row, sizeof row);
      if (part == 1) {
        int32_t packed;
        memcpy(&packed, b + L.packed_n_offset, 4);
        packed = (int32_t)(((uint32_t)packed & 0xffffu) | (part_grid[0] << 16));
        memcpy(b + L.packed_n_offset, &packed, 4);
      }
      memset(b + L.guard_offset, 0, 8);  // the placement record: written by the run's first launch
      memset(b + L.rec_offset, 0, 16);
    }
  char* dev = q->kargs_dev;
  if (need > q->kargs_cap) {
    dev = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&dev), need) != hipSuccess) {
      if (err) *err = "fleet_direct_prepare: out of device memory";
      return FLEET_ERR_HIP;
    }
  }
  for (int part = 0; part < parts; ++part) {  // the first launch of a run: tape row 0's block + where the part's other blocks are
    unsigned char* f = host.data() + blocks_bytes + (size_t)part * FleetDirect::kBlockBytes;
    memcpy(f, host.data() + (size_t)part * tape_len * FleetDirect::kBlockBytes, FleetDirect::kBlockBytes);
    struct { char* blocks; int rows, rotate; } ra = {dev + (size_t)part * tape_len * FleetDirect::kBlockBytes, tape_len, 0};
    static_assert(sizeof ra == 16, "the recording arguments");
    memcpy(f + L.rec_offset, &ra, sizeof ra);
  }
  if (hipMemcpy(dev, host.data(), need, hipMemcpyHostToDevice) != hipSuccess) {
    if (dev != q->kargs_dev) (void)hipFree(dev);
    if (err) *err = "fleet_direct_prepare: argument upload failed";
    return FLEET_ERR_HIP;
  }
  // ---- nothing can fail from here on: commit ----
  if (dev != q->kargs_dev) {
    if (q->kargs_dev) (void)hipFree(q->kargs_dev);
    q->kargs_dev = dev;
    q->kargs_cap = need;
  }
  q->kargs_host.swap(host);
  q->kernel = kernel;
  q->block = L.block;
  q->parts = parts;
  q->part_grid[0] = part_grid[0];
  q->part_grid[1] = part_grid[1];
  q->tape_len = tape_len;
  q->packed_n_offset = L.packed_n_offset;
  q->guard_offset = L.guard_offset;
  q->rec_offset = L.rec_offset;
  return FLEET_OK;
}

int fleet_direct_fault(FleetDirect* q, int kind, int tape_row, std::string* err) {
  if (!q || q->tape_len < 1 || tape_row < 0 || tape_row >= q->tape_len || (kind != 1 && kind != 2)) return FLEET_ERR_INVALID;
  if (q->in_flight) {
    if (err) *err = "fleet_direct_fault: launches are in flight";
    return FLEET_ERR_STATE;
  }
  if (kind == 1) {  // (takes effect at the next chain's start: fleet_direct_submit)
    q->fault_rotate = 1;
    return FLEET_OK;
  }
  for (int part = 0; part < q->parts; ++part) {  // the grid's first workgroup shifted by one
    const size_t off = ((size_t)part * q->tape_len + tape_row) * FleetDirect::kBlockBytes;
    unsigned char* b = q->kargs_host.data() + off;
    uint32_t packed;
    memcpy(&packed, b + q->packed_n_offset, 4);
    packed += 1u << 16;
    memcpy(b + q->packed_n_offset, &packed, 4);
    if (hipMemcpy(q->kargs_dev + off, b, FleetDirect::kBlockBytes, hipMemcpyHostToDevice) != hipSuccess) {
      if (err) *err = "fleet_direct_fault: argument upload failed";
      return FLEET_ERR_HIP;
    }
  }
  return FLEET_OK;
}

int fleet_direct_submit(FleetDirect* q, int steps, int timed, std::string* err) {
  if (!q || !q->queue[0] || q->tape_len < 1 || steps < 1) return FLEET_ERR_INVALID;
  if (q->pending.size() > 4096) {  // a caller that submits run after run without ever waiting (more than any timed series): a wait recycles the signals
    const int rc = fleet_direct_wait(q, nullptr, err);
    if (rc != FLEET_OK) return rc;
  }
  FleetDirect::Mark m{};
  m.parts = q->parts;
  auto need_signal = [&]() {
    hsa_signal_t s = take_signal(q);
    if (s.handle) q->pending.push_back(s);
    return s;
  };
  for (int part = 0; part < q->parts; ++part) {
    m.last[part] = need_signal();
    m.first[part] = (timed == 1) ? need_signal() : hsa_signal_t{};
    if (!m.last[part].handle || (timed == 1 && !m.first[part].handle)) {
      if (err) *err = "fleet_direct_submit: hsa_signal_create failed";
      return FLEET_ERR_HIP;
    }
  }
  if (timed == 2) {  // every packet carries a signal of its own (the last one of each part: m.last)
    m.each.resize((size_t)steps * q->parts);
    for (int i = 0; i < steps; ++i)
      for (int part = 0; part < q->parts; ++part) {
        hsa_signal_t s = (i == steps - 1) ? m.last[part] : need_signal();
        if (!s.handle) {
          if (err) *err = "fleet_direct_submit: hsa_signal_create failed";
          return FLEET_ERR_HIP;
        }
        m.each[(size_t)i * q->parts + part] = s;
      }
  }
  // the run's first launch goes out with the block that has it write the placement record into the others (header comment)
  const size_t first_blocks = (size_t)q->parts * q->tape_len * FleetDirect::kBlockBytes;
  if (q->fault_rotate) {  // test hook: a record shifted by one workgroup
    const int one = 1;
    for (int part = 0; part < q->parts; ++part)
      if (hipMemcpy(q->kargs_dev + first_blocks + (size_t)part * FleetDirect::kBlockBytes + q->rec_offset + 12, &one, 4, hipMemcpyHostToDevice) != hipSuccess)
        return FLEET_ERR_HIP;
  }
  for (int i = 0; i < steps; ++i)
    for (int part = 0; part < q->parts; ++part) {  // step by step, queue by queue: both chains get going at once
      hsa_signal_t sig{};
      if (timed == 2) sig = m.each[(size_t)i * q->parts + part];
      else if (i == steps - 1) sig = m.last[part];
      else if (timed == 1 && i == 0) sig = m.first[part];
      // the first packet after a release acquires at system scope (whatever the host or another queue wrote meanwhile), the others
      // at agent scope; a packet releases only when it is the last of a run that asks for it
      const int acq = (i == 0) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
      const int rel = (i == steps - 1) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE;
      write_packet(q->queue[part], q->kernel, q->block, q->part_grid[part],
                   i == 0 ? q->kargs_dev + first_blocks + (size_t)part * FleetDirect::kBlockBytes
                          : q->kargs_dev + ((size_t)part * q->tape_len + (size_t)(i % q->tape_len)) * FleetDirect::kBlockBytes,
                   acq, rel, sig);
    }
  if (timed) {
    if (timed == 1 && steps == 1)  // (one packet cannot carry two signals: its own start and end are the span then)
      for (int part = 0; part < q->parts; ++part) m.first[part] = m.last[part];
    q->marks.push_back(std::move(m));
  }
  for (int part = 0; part < 2; ++part) q->last[part] = part < q->parts ? (timed ? q->marks.back().last[part] : m.last[part]) : hsa_signal_t{};
  q->in_flight = true;
  if (q->fault_rotate) {  // (the hook corrupts ONE run: the next one records as ever)
    q->fault_rotate = 0;
    const int rc = fleet_direct_wait(q, nullptr, err);
    if (rc != FLEET_OK) return rc;
    const int zero = 0;
    for (int part = 0; part < q->parts; ++part)
      if (hipMemcpy(q->kargs_dev + first_blocks + (size_t)part * FleetDirect::kBlockBytes + q->rec_offset + 12, &zero, 4, hipMemcpyHostToDevice) != hipSuccess)
        return FLEET_ERR_HIP;
    return FLEET_OK;
  }
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
    // 20 s in all: a run of 16384 launches of the largest batch is ~1 s
    const uint64_t hz = q->tick_hz ? q->tick_hz : 100000000ull;
    for (hsa_signal_t sgn : q->last) {
      if (!sgn.handle) continue;
      if (hsa_signal_wait_scacquire(sgn, HSA_SIGNAL_CONDITION_LT, 1, hz / 1000, HSA_WAIT_STATE_ACTIVE) >= 1 &&
          hsa_signal_wait_scacquire(sgn, HSA_SIGNAL_CONDITION_LT, 1, hz * 20ull, HSA_WAIT_STATE_BLOCKED) >= 1) {
        if (err) *err = "fleet_direct_wait: the launches did not complete within 20 s";
        return FLEET_ERR_HIP;
      }
    }
    q->in_flight = false;
  }
  auto us = [&](uint64_t t0, uint64_t t1, bool ok) { return ok && q->tick_hz && t1 > t0 ? (double)(t1 - t0) * 1e6 / (double)q->tick_hz : -1.0; };
  for (const FleetDirect::Mark& m : q->marks) {
    if (!m.each.empty()) {  // every launch's own duration; with two queues a step lasts from its earlier start to its later end
      for (size_t i = 0; i < m.each.size() / m.parts; ++i) {
        uint64_t t0 = UINT64_MAX, t1 = 0;
        bool ok = true;
        for (int part = 0; part < m.parts; ++part) {
          hsa_amd_profiling_dispatch_time_t a{};
          ok = ok && hsa_amd_profiling_get_dispatch_time(q->agent, m.each[i * m.parts + part], &a) == HSA_STATUS_SUCCESS;
          t0 = a.start < t0 ? a.start : t0;
          t1 = a.end > t1 ? a.end : t1;
        }
        if (spans_us) spans_us->push_back(us(t0, t1, ok));
      }
      continue;
    }
    // a timed run: from the earlier start of its first launches to the later end of its last
    uint64_t t0 = UINT64_MAX, t1 = 0;
    bool ok = true;
    for (int part = 0; part < m.parts; ++part) {
      hsa_amd_profiling_dispatch_time_t a{}, b{};
      ok = ok && hsa_amd_profiling_get_dispatch_time(q->agent, m.first[part], &a) == HSA_STATUS_SUCCESS &&
           hsa_amd_profiling_get_dispatch_time(q->agent, m.last[part], &b) == HSA_STATUS_SUCCESS;
      t0 = a.start < t0 ? a.start : t0;
      t1 = b.end > t1 ? b.end : t1;
    }
    if (spans_us) spans_us->push_back(us(t0, t1, ok));
  }
  q->marks.clear();
  for (hsa_signal_t sgn : q->pending) q->pool.push_back(sgn);
  q->pending.clear();
  q->last[0].handle = q->last[1].handle = 0;
  return FLEET_OK;
}
