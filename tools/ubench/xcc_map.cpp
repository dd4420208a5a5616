// Microbenchmark (GPU box): which die (XCC) does workgroup w of a launch run on -- and is that the SAME die in the next launch of the
// same queue?  fleet_direct.hip drops the release fence between the launches of a run and therefore relies on "workgroup w of every
// launch runs on the die workgroup w of the previous launch ran on".  The platform does not promise it (MI355X_MICROARCH.md,
// "Workgroup dispatch, XCD placement": observed round-robin, the die of workgroup 0 not fixed), so the library probes it when it opens
// its queue and the step kernel checks it in every launch; this tool is the exploration behind that probe: sequences of launches
// with grids that are and are not multiples of 8 workgroups, on two HSA queues, interleaved, with other workgroup sizes, and
// through a HIP stream.
// build + run: tools/ubench/xcc_map.sh
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#define HSA_OK(x) do { hsa_status_t _s = (x); if (_s != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(_s, &m); \
  fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m); exit(2); } } while (0)
#define HIP_OK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(_e)); exit(2); } } while (0)

static hsa_agent_t g_gpu; static bool g_have = false;
static hsa_status_t pick(hsa_agent_t a, void*) {
  hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have) { g_gpu = a; g_have = true; }
  return HSA_STATUS_SUCCESS;
}
struct Kern { uint64_t obj; uint32_t karg, lds, scratch; };
static void put(hsa_queue_t* q, const Kern& k, void* kargs, uint32_t blocks, uint32_t block, int acq, int rel, hsa_signal_t done) {
  const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
  while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {}
  hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
  p->workgroup_size_x = (uint16_t)block; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
  p->grid_size_x = blocks * block; p->grid_size_y = 1; p->grid_size_z = 1;
  p->private_segment_size = k.scratch; p->group_segment_size = k.lds;
  p->kernel_object = k.obj; p->kernarg_address = kargs; p->reserved2 = 0; p->completion_signal = done;
  const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                          (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
  const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
  __atomic_store_n((uint32_t*)p, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
  hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
}

constexpr int kMaxGrid = 8192;
struct Launch { int queue; unsigned grid, block; };  // queue 0 / 1: HSA queues of the tool's own; 2: a HIP stream

static void report(const char* title, const std::vector<Launch>& seq, const std::vector<unsigned>& host) {
  printf("---- %s\n", title);
  unsigned first8[8] = {0};
  for (size_t i = 0; i < seq.size(); ++i) {
    const unsigned* o = host.data() + i * 2 * kMaxGrid;
    bool periodic = true, same = true;
    int per_die[16] = {0};
    for (unsigned w = 0; w < seq[i].grid; ++w) {
      const unsigned x = o[2 * w] & 0xf;
      per_die[x] += 1;
      periodic = periodic && (x == (o[2 * (w & 7)] & 0xf) || seq[i].grid < 8);
      if (i > 0 && w < 8) same = same && (x == first8[w]);
    }
    if (i == 0) for (unsigned w = 0; w < 8 && w < seq[i].grid; ++w) first8[w] = o[2 * w] & 0xf;
    printf("  launch %2zu on %s grid %5u x %4u  raw XCC_ID[wg0] 0x%08x  dies of wg 0..15:", i,
           seq[i].queue == 2 ? "HIP stream" : (seq[i].queue ? "queue B   " : "queue A   "), seq[i].grid, seq[i].block, o[0]);
    for (unsigned w = 0; w < 16 && w < seq[i].grid; ++w) printf(" %u", o[2 * w] & 0xf);
    printf("  periodic(8) %s  first8 == launch 0's %s  workgroups per die:", periodic ? "yes" : "NO", (i == 0 || same) ? "yes" : "NO");
    for (int x = 0; x < 8; ++x) printf(" %d", per_die[x]);
    printf("\n");
  }
}

int main(int argc, char** argv) {
  const char* hsaco = argc > 1 ? argv[1] : "/tmp/xcc_map_kernels.hsaco";
  HIP_OK(hipSetDevice(0));
  HSA_OK(hsa_init());
  HSA_OK(hsa_iterate_agents(pick, nullptr));
  if (!g_have) { fprintf(stderr, "no GPU agent\n"); return 2; }
  uint32_t num_xcc = 0;

  (void)hsa_agent_get_info(g_gpu, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_NUM_XCC, &num_xcc);

  uint32_t cus = 0; (void)hsa_agent_get_info(g_gpu, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_COMPUTE_UNIT_COUNT, &cus);
  printf("agent: %u compute units, NUM_XCC %u\n", cus, num_xcc);
  hsa_queue_t* q[2];
  for (auto& x : q) HSA_OK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &x));
  const int fd = open(hsaco, O_RDONLY);
  if (fd < 0) { perror(hsaco); return 2; }
  hsa_code_object_reader_t rd; HSA_OK(hsa_code_object_reader_create_from_file(fd, &rd));
  hsa_executable_t exe; HSA_OK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  HSA_OK(hsa_executable_load_agent_code_object(exe, g_gpu, rd, nullptr, nullptr));
  HSA_OK(hsa_executable_freeze(exe, nullptr));
  hsa_executable_symbol_t s; Kern k{};
  HSA_OK(hsa_executable_get_symbol_by_name(exe, "k_where.kd", &g_gpu, &s));
  HSA_OK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.obj));
  hipModule_t mod; HIP_OK(hipModuleLoad(&mod, hsaco));
  hipFunction_t f; HIP_OK(hipModuleGetFunction(&f, mod, "k_where"));
  hipStream_t st; HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hsa_signal_t done; HSA_OK(hsa_signal_create(1, 0, nullptr, &done));

  auto run = [&](const char* title, const std::vector<Launch>& seq) {
    unsigned* out; HIP_OK(hipMalloc(&out, seq.size() * 2 * kMaxGrid * 4));
    HIP_OK(hipMemset(out, 0xff, seq.size() * 2 * kMaxGrid * 4));
    char* kargs; HIP_OK(hipMalloc(&kargs, seq.size() * 64));
    std::vector<char> hk(seq.size() * 64, 0);
    for (size_t i = 0; i < seq.size(); ++i) { unsigned* p = out + i * 2 * kMaxGrid; memcpy(hk.data() + i * 64, &p, 8); }
    HIP_OK(hipMemcpy(kargs, hk.data(), hk.size(), hipMemcpyHostToDevice));
    HIP_OK(hipDeviceSynchronize());
    for (size_t i = 0; i < seq.size(); ++i) {
      // every launch is waited for before the next is written: the order of the sequence is the order on the device also across queues
      if (seq[i].queue == 2) {
        unsigned* p = out + i * 2 * kMaxGrid; size_t sz = 8;
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        HIP_OK(hipModuleLaunchKernel(f, seq[i].grid, 1, 1, seq[i].block, 1, 1, 0, st, nullptr, cfg));
        HIP_OK(hipStreamSynchronize(st));
      } else {
        hsa_signal_store_relaxed(done, 1);
        put(q[seq[i].queue], k, kargs + i * 64, seq[i].grid, seq[i].block, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM, done);
        if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 5ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_ACTIVE) != 0) {
          fprintf(stderr, "timeout\n"); exit(3);
        }
      }
    }
    std::vector<unsigned> host(seq.size() * 2 * kMaxGrid);
    HIP_OK(hipMemcpy(host.data(), out, host.size() * 4, hipMemcpyDeviceToHost));
    report(title, seq, host);
    HIP_OK(hipFree(out)); HIP_OK(hipFree(kargs));
  };
  // back-to-back submission (no host wait between the packets): what a run of steps looks like
  auto run_chain = [&](const char* title, int queue, const std::vector<unsigned>& grids) {
    std::vector<Launch> seq;
    for (unsigned g : grids) seq.push_back({queue, g, 256});
    unsigned* out; HIP_OK(hipMalloc(&out, seq.size() * 2 * kMaxGrid * 4));
    HIP_OK(hipMemset(out, 0xff, seq.size() * 2 * kMaxGrid * 4));
    char* kargs; HIP_OK(hipMalloc(&kargs, seq.size() * 64));
    std::vector<char> hk(seq.size() * 64, 0);
    for (size_t i = 0; i < seq.size(); ++i) { unsigned* p = out + i * 2 * kMaxGrid; memcpy(hk.data() + i * 64, &p, 8); }
    HIP_OK(hipMemcpy(kargs, hk.data(), hk.size(), hipMemcpyHostToDevice));
    HIP_OK(hipDeviceSynchronize());
    hsa_signal_store_relaxed(done, 1);
    for (size_t i = 0; i < seq.size(); ++i) {
      hsa_signal_t sg{}; if (i + 1 == seq.size()) sg = done;
      put(q[queue], k, kargs + i * 64, seq[i].grid, 256, i == 0 ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT,
          i + 1 == seq.size() ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE, sg);
    }
    if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 5ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_ACTIVE) != 0) { fprintf(stderr, "timeout\n"); exit(3); }
    std::vector<unsigned> host(seq.size() * 2 * kMaxGrid);
    HIP_OK(hipMemcpy(host.data(), out, host.size() * 4, hipMemcpyDeviceToHost));
    report(title, seq, host);
    HIP_OK(hipFree(out)); HIP_OK(hipFree(kargs));
  };

  const std::vector<unsigned> grids = {1024, 1024, 1027, 1027, 1024, 13, 13, 5, 3, 1024, 4096, 4093, 4096};
  for (int qi = 0; qi < 2; ++qi) {
    std::vector<Launch> seq;
    for (unsigned g : grids) seq.push_back({qi, g, 256});
    run(qi ? "queue B, one launch at a time" : "queue A, one launch at a time", seq);
  }
  run("queues A and B interleaved", {{0, 1024, 256}, {1, 1027, 256}, {0, 1024, 256}, {1, 1024, 256}, {0, 1027, 256}, {1, 1024, 256}, {0, 1024, 256}});
  run("other workgroup sizes (queue A)", {{0, 1024, 256}, {0, 1024, 64}, {0, 1027, 64}, {0, 1024, 64}, {0, 1024, 1024}, {0, 1021, 1024}, {0, 1024, 1024}, {0, 1024, 256}});
  {
    std::vector<Launch> seq;
    for (unsigned g : grids) seq.push_back({2, g, 256});
    run("HIP stream, one launch at a time", seq);
  }
  run("HIP stream between the queues' launches", {{0, 1024, 256}, {2, 1027, 256}, {0, 1024, 256}, {2, 13, 256}, {0, 1024, 256}});
  run_chain("queue A, a chain of packets without host waits (the shape of a run)", 0, {1024, 1024, 1024, 1027, 1027, 1027, 1024, 1024, 13, 13, 13, 1024});
  run_chain("queue B, a chain of packets without host waits", 1, {4096, 4096, 4093, 4093, 4096, 4096});
  // ---- does the die a queue deals from move when OTHER queues come and go?  (it does: the library records it per chain)
  {
    printf("---- queue A while other queues are created and destroyed\n");
    auto first_die = [&](const char* when) {
      unsigned* out; HIP_OK(hipMalloc(&out, 2 * kMaxGrid * 4));
      char* kargs; HIP_OK(hipMalloc(&kargs, 64));
      HIP_OK(hipMemcpy(kargs, &out, 8, hipMemcpyHostToDevice));
      HIP_OK(hipDeviceSynchronize());
      hsa_signal_store_relaxed(done, 1);
      put(q[0], k, kargs, 64, 256, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM, done);
      if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 5ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_ACTIVE) != 0) { fprintf(stderr, "timeout\n"); exit(3); }
      unsigned h[16];
      HIP_OK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
      printf("  %-58s queue A deals workgroups 0..7 to dies:", when);
      for (int w = 0; w < 8; ++w) printf(" %u", h[2 * w] & 0xf);
      printf("\n");
      HIP_OK(hipFree(out)); HIP_OK(hipFree(kargs));
    };
    first_die("as it is (queues A, B and one HIP stream exist)");
    std::vector<hsa_queue_t*> extra;
    for (int i = 0; i < 3; ++i) {
      hsa_queue_t* x; HSA_OK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &x));
      extra.push_back(x);
      first_die(("after creating HSA queue #" + std::to_string(i + 3)).c_str());
    }
    hipStream_t st2[3];
    for (int i = 0; i < 3; ++i) {
      HIP_OK(hipStreamCreateWithFlags(&st2[i], hipStreamNonBlocking));
      unsigned* p = nullptr; HIP_OK(hipMalloc(&p, 2 * kMaxGrid * 4)); size_t sz = 8;
      void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
      HIP_OK(hipModuleLaunchKernel(f, 8, 1, 1, 256, 1, 1, 0, st2[i], nullptr, cfg));
      HIP_OK(hipStreamSynchronize(st2[i]));
      HIP_OK(hipFree(p));
      first_die(("after a launch on a new HIP stream #" + std::to_string(i + 2)).c_str());
    }
    for (size_t i = 0; i < extra.size(); ++i) {
      hsa_queue_destroy(extra[i]);
      first_die(("after destroying HSA queue #" + std::to_string(i + 3)).c_str());
    }
    for (int i = 0; i < 40; ++i) {  // more queues than the hardware has slots for
      hsa_queue_t* x;
      if (hsa_queue_create(g_gpu, 256, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &x) != HSA_STATUS_SUCCESS) break;
      extra.push_back(x);
    }
    first_die("with 40 more (idle) HSA queues");
    first_die("again");
  }
  hsa_queue_destroy(q[0]); hsa_queue_destroy(q[1]);
  return 0;
}
