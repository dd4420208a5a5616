// Microbenchmark (GPU box): what the control flow of a divergent region costs per wavefront -- exec-mask save / restore with a
// not-taken and a taken skip branch, uniform scalar branches, and s_waitcnt with nothing outstanding.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 32
#define OUTER 64
template <int OP>
__global__ void k(double* out, unsigned long long* cyc, int lanes_on) {
  float f0 = threadIdx.x, f1 = 1.0f;
  const int lane = threadIdx.x & 63;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int o = 0; o < OUTER; ++o) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      if (OP == 0) {  // divergent region, some lanes active: not-taken execz branch
        asm volatile("v_cmp_gt_i32 vcc, %2, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_f32 %0, %0, %0\n1:\n s_or_b64 exec, exec, s[20:21]"
                     : "+v"(f0) : "v"(lane), "s"(lanes_on) : "vcc", "s20", "s21");
      }
      if (OP == 1) {  // the same with no lane active: the skip branch is taken
        asm volatile("v_cmp_gt_i32 vcc, 0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n1:\n s_or_b64 exec, exec, s[20:21]"
                     : "+v"(f0) : "v"(lane) : "vcc", "s20", "s21");
      }
      if (OP == 2) {  // uniform scalar branch, not taken
        asm volatile("s_cmp_eq_u32 %1, 12345\n s_cbranch_scc1 1f\n v_add_f32 %0, %0, %0\n1:" : "+v"(f0) : "s"(lanes_on) : "scc");
      }
      if (OP == 3) {  // uniform scalar branch, taken over 4 instructions
        asm volatile("s_cmp_lg_u32 %1, 12345\n s_cbranch_scc1 1f\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n1:" : "+v"(f0) : "s"(lanes_on) : "scc");
      }
      if (OP == 4) { asm volatile("v_add_f32 %0, %0, %0\n s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(f0)); }
      if (OP == 5) { asm volatile("v_add_f32 %0, %0, %0\n s_nop 0" : "+v"(f0)); }
      if (OP == 6) { asm volatile("v_readfirstlane_b32 s20, %0\n v_add_f32 %0, %0, %0" : "+v"(f0) : : "s20"); }
      if (OP == 7) {  // taken branch over a long region (64 instructions): instruction fetch redirect
        asm volatile("s_cmp_lg_u32 %1, 12345\n s_cbranch_scc1 1f\n .rept 64\n v_add_f32 %0, %0, %0\n .endr\n1:" : "+v"(f0) : "s"(lanes_on) : "scc");
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1;
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int OP>
void run(const char* name) {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4096 * 256 * 8); (void)hipMalloc(&cyc, 4096 * 4 * 8);
  for (int wps : {1, 4}) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 12);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 12);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    s /= h.size();
    printf("%-46s waves/SIMD %d: %.1f cycles per region (wave view), %.1f per SIMD\n", name, wps, s / (REP * OUTER), s / (REP * OUTER) / wps);
  }
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  run<0>("cmp+saveexec+execz(not taken)+1 valu+restore"); run<1>("cmp+saveexec+execz(TAKEN over 4)+restore");
  run<2>("s_cmp+scc branch not taken+1 valu"); run<3>("s_cmp+scc branch TAKEN over 4 valu"); run<4>("1 valu + s_waitcnt(0) idle");
  run<5>("1 valu + s_nop"); run<6>("readfirstlane + 1 valu"); run<7>("s_cmp+scc branch TAKEN over 64 valu");
  return 0;
}
