// Microbenchmark (GPU box): how much of a launch's footprint a die's L2 carries over to the next launch when the kernel boundary has
// no release fence -- 1024 workgroups (128 per die) read-modify-write (k_touch) or read (k_read) a chunk each, the chunk size swept so
// that the footprint per die goes from 0.5 to 6 MB; per setting the time per launch with HIP's fences and with none.  Built like
// aql_fence.cpp (shares its kernels' code object).
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <fcntl.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#define HSA_OK(x) do { hsa_status_t _s = (x); if (_s != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(_s, &m); fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m); exit(2); } } while (0)
#define HIP_OK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(_e)); exit(2); } } while (0)
static hsa_agent_t g_gpu; static bool g_have = false;
static hsa_status_t pick(hsa_agent_t a, void*) { hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t); if (t == HSA_DEVICE_TYPE_GPU && !g_have) { g_gpu = a; g_have = true; } return HSA_STATUS_SUCCESS; }
struct Kern { uint64_t obj; };
int main(int argc, char** argv) {
  const char* hsaco = argc > 1 ? argv[1] : "/tmp/aql_fence_kernels.hsaco";
  const int blocks = 1024, n = 1000;
  HIP_OK(hipSetDevice(0));
  const size_t max_words = 8192;
  double* buf; HIP_OK(hipMalloc(&buf, (size_t)blocks * max_words * 8));
  char* kargs; HIP_OK(hipMalloc(&kargs, 4096));
  HSA_OK(hsa_init()); HSA_OK(hsa_iterate_agents(pick, nullptr));
  hsa_queue_t* q; HSA_OK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
  const int fd = open(hsaco, O_RDONLY); if (fd < 0) { perror(hsaco); return 2; }
  hsa_code_object_reader_t rd; HSA_OK(hsa_code_object_reader_create_from_file(fd, &rd));
  hsa_executable_t exe; HSA_OK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  HSA_OK(hsa_executable_load_agent_code_object(exe, g_gpu, rd, nullptr, nullptr)); HSA_OK(hsa_executable_freeze(exe, nullptr));
  const char* names[2] = {"k_touch", "k_read"};
  uint64_t obj[2];
  for (int i = 0; i < 2; ++i) { hsa_executable_symbol_t s; HSA_OK(hsa_executable_get_symbol_by_name(exe, (std::string(names[i]) + ".kd").c_str(), &g_gpu, &s)); HSA_OK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &obj[i])); }
  hsa_signal_t done; HSA_OK(hsa_signal_create(1, 0, nullptr, &done));
  printf("1024 workgroups, %d launches per figure; us per launch: HIP's fences / no fence inside the batch\n", n);
  const int sweep[] = {512, 1024, 1536, 2048, 3072, 4096, 6144, 8192};
  for (int ki = 0; ki < 2; ++ki)
    for (int words : sweep) {
      struct Args { double* buf; int words; int pad; } a{buf, words, 0};
      std::vector<char> hk(4096, 0); memcpy(hk.data(), &a, sizeof a);
      HIP_OK(hipMemcpy(kargs, hk.data(), 4096, hipMemcpyHostToDevice));
      double us[2];
      for (int mode = 0; mode < 2; ++mode) {
        HIP_OK(hipMemset(buf, 0, (size_t)blocks * max_words * 8)); HIP_OK(hipDeviceSynchronize());
        hsa_signal_store_relaxed(done, 1);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {
          int acq = HSA_FENCE_SCOPE_AGENT, rel = HSA_FENCE_SCOPE_AGENT;
          if (mode == 1) { acq = (i == 0) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT; rel = HSA_FENCE_SCOPE_NONE; }
          if (i == n - 1) rel = HSA_FENCE_SCOPE_SYSTEM;
          const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
          while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {}
          hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
          p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->grid_size_x = blocks * 256; p->grid_size_y = 1; p->grid_size_z = 1;
          p->private_segment_size = 0; p->group_segment_size = 0; p->kernel_object = obj[ki]; p->kernarg_address = kargs; p->reserved2 = 0;
          hsa_signal_t s{}; if (i == n - 1) s = done; p->completion_signal = s;
          const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) | (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
          __atomic_store_n((uint32_t*)p, (uint32_t)header | ((uint32_t)(1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16), __ATOMIC_RELEASE);
          hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
        }
        if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 20ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_ACTIVE) != 0) { fprintf(stderr, "timeout\n"); return 3; }
        us[mode] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
      }
      const double mb = (double)blocks * words * 8 / 1e6;
      printf("%-8s %6.1f MB (%4.2f MB per die)  %7.3f / %7.3f us   %5.2f / %5.2f TB/s of footprint per launch\n", names[ki], mb, mb / 8, us[0], us[1],
             mb / us[0], mb / us[1]);
    }
  hsa_queue_destroy(q);
  return 0;
}
