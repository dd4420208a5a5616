// Microbenchmark (GPU box): what the cache actions at a kernel boundary cost, and whether state survives in a die's L2 from one
// launch to the next when the boundary carries none.  HIP gives every kernel packet agent-scope acquire + release fences (the eight
// L2s of the part are not coherent with each other, so that is an L2 write-back + invalidate per launch).  The step kernel does not
// need them between two steps: workgroup w -- hence die w mod 8 -- owns the same envs in every launch.  This tool writes AQL
// dispatch packets into an HSA queue of its own with the fence scopes of its choice:
//   agent/agent on every packet (what HIP does)  |  none/none between the launches, agent acquire on the first, system release on the last
// for an empty kernel, a read-modify-write of 4 KB per workgroup and the same behind a dependent chain of loads; checks the
// result (n launches -> every word == n) and prints microseconds per launch.  The same kernels through a HIP graph for reference.
// build: see aql_fence.sh
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
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
static Kern symbol(hsa_executable_t exe, const char* name) {
  hsa_executable_symbol_t s; Kern k{};
  HSA_OK(hsa_executable_get_symbol_by_name(exe, (std::string(name) + ".kd").c_str(), &g_gpu, &s));
  HSA_OK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.obj));
  HSA_OK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.karg));
  HSA_OK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.lds));
  HSA_OK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.scratch));
  return k;
}
static void put(hsa_queue_t* q, const Kern& k, void* kargs, uint32_t blocks, int acq, int rel, hsa_signal_t done, int barrier = 1) {
  const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
  while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {}
  hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
  p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
  p->grid_size_x = blocks * 256; p->grid_size_y = 1; p->grid_size_z = 1;
  p->private_segment_size = k.scratch; p->group_segment_size = k.lds;
  p->kernel_object = k.obj; p->kernarg_address = kargs; p->reserved2 = 0; p->completion_signal = done;
  const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (barrier << HSA_PACKET_HEADER_BARRIER) |
                          (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
  const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
  __atomic_store_n((uint32_t*)p, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
  hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
}

int main(int argc, char** argv) {
  const char* hsaco = argc > 1 ? argv[1] : "tools/ubench/aql_fence_kernels.hsaco";
  const int blocks = argc > 2 ? atoi(argv[2]) : 1024, words = argc > 3 ? atoi(argv[3]) : 512, n = argc > 4 ? atoi(argv[4]) : 2000;
  HIP_OK(hipSetDevice(0));
  double* buf; HIP_OK(hipMalloc(&buf, (size_t)blocks * words * 8));
  struct Args { double* buf; int words; int pad; };
  char* kargs; HIP_OK(hipMalloc(&kargs, 4096));
  HSA_OK(hsa_init());
  HSA_OK(hsa_iterate_agents(pick, nullptr));
  if (!g_have) { fprintf(stderr, "no GPU agent\n"); return 2; }
  hsa_queue_t* q;
  HSA_OK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
  const int fd = open(hsaco, O_RDONLY);
  if (fd < 0) { perror(hsaco); return 2; }
  hsa_code_object_reader_t rd; HSA_OK(hsa_code_object_reader_create_from_file(fd, &rd));
  hsa_executable_t exe; HSA_OK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  HSA_OK(hsa_executable_load_agent_code_object(exe, g_gpu, rd, nullptr, nullptr));
  HSA_OK(hsa_executable_freeze(exe, nullptr));
  const char* names[3] = {"k_empty", "k_touch", "k_chain"};
  Kern ks[3];
  for (int i = 0; i < 3; ++i) { ks[i] = symbol(exe, names[i]); printf("%s kernarg %u lds %u scratch %u\n", names[i], ks[i].karg, ks[i].lds, ks[i].scratch); }
  std::vector<char> hk(4096, 0);
  Args a{buf, words, 0}; memcpy(hk.data(), &a, sizeof a);
  HIP_OK(hipMemcpy(kargs, hk.data(), 4096, hipMemcpyHostToDevice));
  hsa_signal_t done; HSA_OK(hsa_signal_create(1, 0, nullptr, &done));
  std::vector<double> host((size_t)blocks * words);
  printf("%d workgroups x %d B, %d launches per figure\n", blocks, words * 8, n);
  for (int rep = 0; rep < 1; ++rep)
  for (int ki = 0; ki < 3; ++ki) {
    for (int mode = 0; mode < 3; ++mode) {
      // mode 0: agent/agent everywhere; 1: none/none inside the batch; 2: agent acquire, no release inside the batch
      HIP_OK(hipMemset(buf, 0, (size_t)blocks * words * 8));
      HIP_OK(hipDeviceSynchronize());
      hsa_signal_store_relaxed(done, 1);
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < n; ++i) {
        int acq = HSA_FENCE_SCOPE_AGENT, rel = HSA_FENCE_SCOPE_AGENT;
        if (mode == 1) { acq = (i == 0) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE; rel = HSA_FENCE_SCOPE_NONE; }
        if (mode == 2) { rel = HSA_FENCE_SCOPE_NONE; }
        if (i == n - 1) rel = HSA_FENCE_SCOPE_SYSTEM;
        hsa_signal_t s{}; if (i == n - 1) s = done;
        put(q, ks[ki], kargs, blocks, acq, rel, s);
      }
      if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 20ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_ACTIVE) != 0) {
        fprintf(stderr, "timeout waiting for the batch\n"); return 3;
      }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
      HIP_OK(hipMemcpy(host.data(), buf, host.size() * 8, hipMemcpyDeviceToHost));
      size_t bad = 0; const double want = ki == 0 ? 0.0 : (double)n;
      for (double v : host) bad += (v != want);
      printf("own queue  %-8s %-28s %7.3f us per launch   wrong words %zu\n", names[ki],
             mode == 0 ? "agent/agent" : mode == 1 ? "none/none inside the batch" : "agent acquire, no release", us, bad);
    }
    // the same kernel through HIP: a captured graph of 100 launches, replayed
    hipModule_t mod; HIP_OK(hipModuleLoad(&mod, hsaco));
    hipFunction_t f; HIP_OK(hipModuleGetFunction(&f, mod, names[ki]));
    hipStream_t st; HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIP_OK(hipMemset(buf, 0, (size_t)blocks * words * 8));
    HIP_OK(hipDeviceSynchronize());
    Args ha{buf, words, 0}; size_t sz = sizeof ha;
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ha, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    hipGraph_t g; hipGraphExec_t ge;
    HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 100; ++i) HIP_OK(hipModuleLaunchKernel(f, blocks, 1, 1, 256, 1, 1, 0, st, nullptr, cfg));
    HIP_OK(hipStreamEndCapture(st, &g));
    HIP_OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HIP_OK(hipGraphLaunch(ge, st)); HIP_OK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n / 100; ++i) HIP_OK(hipGraphLaunch(ge, st));
    HIP_OK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (n / 100 * 100);
    printf("HIP graph  %-8s %-28s %7.3f us per launch\n", names[ki], "", us);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(st); (void)hipModuleUnload(mod);
  }
  // ---- dependent launches without the barrier bit (see the *_dep kernels): per-launch argument blocks carry the generation
  {
    struct DArgs { double* buf; int words; int pad; int* done; int gen; int flavor; int* err; };
    const char* dn[3] = {"k_empty_dep", "k_touch_dep", "k_chain_dep"};
    int* flags; HIP_OK(hipMalloc(&flags, (size_t)(blocks + 1) * 4));
    char* dk; HIP_OK(hipMalloc(&dk, (size_t)n * 64));
    for (int rep = 0; rep < 1; ++rep)
    for (int ki = 0; ki < 3; ++ki) {
      const Kern k = symbol(exe, dn[ki]);
      for (int flavor = 1; flavor < 5; ++flavor)
      for (int barrier = 0; barrier < 2; ++barrier)
      for (int acqm = 0; acqm < 1; ++acqm) {
        std::vector<char> hb((size_t)n * 64, 0);
        for (int i = 0; i < n; ++i) { DArgs d{buf, words, 0, flags, i, flavor, flags + blocks}; memcpy(hb.data() + (size_t)i * 64, &d, sizeof d); }
        HIP_OK(hipMemcpy(dk, hb.data(), hb.size(), hipMemcpyHostToDevice));
        HIP_OK(hipMemset(buf, 0, (size_t)blocks * words * 8));
        HIP_OK(hipMemset(flags, 0, (size_t)(blocks + 1) * 4));
        HIP_OK(hipDeviceSynchronize());
        hsa_signal_store_relaxed(done, 1);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {
          const int acq = (i == 0) ? HSA_FENCE_SCOPE_SYSTEM : (acqm ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE);
          const int rel = (i == n - 1) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE;
          hsa_signal_t s{}; if (i == n - 1) s = done;
          put(q, k, dk + (size_t)i * 64, blocks, acq, rel, s, (i == 0 || barrier) ? 1 : 0);
        }
        if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 20ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_ACTIVE) != 0) {
          fprintf(stderr, "timeout waiting for the batch\n"); return 3;
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        HIP_OK(hipMemcpy(host.data(), buf, host.size() * 8, hipMemcpyDeviceToHost));
        int errs = 0; HIP_OK(hipMemcpy(&errs, flags + blocks, 4, hipMemcpyDeviceToHost));
        size_t bad = 0; const double want = ki == 0 ? 0.0 : (double)n;
        for (double v : host) bad += (v != want);
        const char* fl[5] = {"agent-scope flag, buffer_inv", "L2 flag, buffer_inv sc1", "L2 flag, state loads sc1", "L2 flag, state loads sc0 sc1", "L2 flag, plain loads (control)"};
        printf("%-10s %-12s %-30s acquire %-5s %7.3f us per launch   wrong words %zu  gave up %d\n", barrier ? "barrier" : "no barrier", dn[ki],
               fl[flavor], acqm ? "agent" : "none", us, bad, errs);
      }
    }
  }
  hsa_queue_destroy(q);
  return 0;
}
