// Microbenchmark (GPU box): what a launch pays for the bytes it leaves dirty in the write-back L2 -- 4096 wavefronts each
// read 3 KB and write 3 KB (the step kernel's shape: 12 MB in, 12 MB out), with plain / nt / sc1 (write-through) / sc0 sc1
// stores, 16 bytes or 4 bytes per lane.  Reports the time per launch of a back-to-back sequence (HIP events).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE, int WIDTH>
__global__ __launch_bounds__(256) void k(const v4f* __restrict__ in, v4f* __restrict__ out, int per_lane16) {
  const size_t wave = (size_t)blockIdx.x * 4 + threadIdx.x / 64;
  const int lane = threadIdx.x & 63;
  const size_t base = wave * (size_t)per_lane16 * 64;
  for (int j = 0; j < per_lane16; ++j) {
    v4f v = in[base + (size_t)j * 64 + lane];
    v.x += 1.0f;
    v4f* p = out + base + (size_t)j * 64 + lane;
    if (WIDTH == 16) {
      if (MODE == 0) *p = v;
      if (MODE == 1) __builtin_nontemporal_store(v, p);
      if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
      if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
      if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
    } else {  // four 4-byte stores per lane at a 256-byte stride: the shape of the observation slots
      float* q = reinterpret_cast<float*>(out + base + (size_t)j * 64) + lane;
      for (int w = 0; w < 4; ++w) {
        float f = v[w];
        if (MODE == 0) q[w * 64] = f;
        if (MODE == 1) __builtin_nontemporal_store(f, q + w * 64);
        if (MODE == 2) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(q + w * 64), "v"(f) : "memory");
        if (MODE == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(q + w * 64), "v"(f) : "memory");
        if (MODE == 4) asm volatile("global_store_dword %0, %1, off sc1 nt" : : "v"(q + w * 64), "v"(f) : "memory");
      }
    }
  }
}
__global__ void empty_k(int* p) { if (p == nullptr) return; }
template <int MODE, int WIDTH>
float run(const v4f* in, v4f* out, int waves, int per_lane16, int reps) {
  // a captured graph of 64 launches, replayed: no host launch cost in the figure (as bench.py runs the step kernel)
  hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 64; ++i) hipLaunchKernelGGL((k<MODE, WIDTH>), dim3(waves / 4), dim3(256), 0, st, in, out, per_lane16);
  (void)hipStreamEndCapture(st, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 10; ++i) (void)hipGraphLaunch(ge, st);
  (void)hipStreamSynchronize(st);
  (void)hipEventRecord(a, st);
  for (int i = 0; i < reps / 64; ++i) (void)hipGraphLaunch(ge, st);
  (void)hipEventRecord(b, st); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(st);
  return ms * 1000.0f / (reps / 64 * 64);
}
int main() {
  const int waves = 4096;
  const char* names[] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt"};
  for (int per_lane16 : {0, 1, 2, 3, 6}) {  // 1 KB, 3 KB, 6 KB per wave each way
    const size_t n = (size_t)waves * (per_lane16 ? per_lane16 : 1) * 64;
    v4f *in, *out;
    (void)hipMalloc(&in, n * 16); (void)hipMalloc(&out, n * 16);
    (void)hipMemset(in, 0, n * 16);
    printf("---- %d waves, %.1f MB read + %.1f MB written per launch\n", waves, n * 16 / 1e6, n * 16 / 1e6);
    float t;
#define RUN(M, W) t = run<M, W>(in, out, waves, per_lane16, 4096); printf("  %-8s %2d B/lane stores: %.2f us per launch\n", names[M], W, t);
    RUN(0, 16) RUN(1, 16) RUN(2, 16) RUN(4, 16) RUN(0, 4) RUN(1, 4) RUN(2, 4)
    (void)hipFree(in); (void)hipFree(out);
  }
  {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty_k, dim3(1024), dim3(256), 0, 0, nullptr);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("empty kernel, 1024 workgroups: %.2f us per launch\n", ms * 1000 / 2000);
  }
  return 0;
}
