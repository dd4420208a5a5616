// Microbenchmark (GPU box), second set: selects / compares with scalar masks, float64 ops with scalar or literal operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define OUTER 64
template <int OP>
__global__ void k(double* out, unsigned long long* cyc, double seed) {
  double a = seed + threadIdx.x * 1e-3, b = 1.000001, c = 0.5, x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
  float f0 = (float)a, f1 = 1.0f, f2 = 2.0f, f3 = 3.0f;
  int i0 = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int o = 0; o < OUTER; ++o) {
#pragma unroll
    for (int r = 0; r < REP / 4; ++r) {
      if (OP == 0) { asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(i0) : "vcc"); }
      if (OP == 1) { asm volatile("s_mov_b64 s[20:21], 0x55\n v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(i0) : "s20", "s21"); }
      if (OP == 2) { asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %5, %5, %4, vcc" : : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(f0), "v"(f1) : "vcc"); }
      if (OP == 3) { asm volatile("v_cmp_lt_f64_e64 s[20:21], %0, %1\n v_cmp_lt_f64_e64 s[22:23], %1, %2\n v_cmp_lt_f64_e64 s[20:21], %2, %3\n v_cmp_lt_f64_e64 s[22:23], %3, %0" : : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "s20", "s21", "s22", "s23"); }
      if (OP == 4) { asm volatile("s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3ff00000\n v_mul_f64 %0, s[20:21], %0\n v_mul_f64 %1, s[20:21], %1\n v_mul_f64 %2, s[20:21], %2\n v_mul_f64 %3, s[20:21], %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "s20", "s21"); }
      if (OP == 5) { asm volatile("s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3ff00000\n v_fma_f64 %0, %0, s[20:21], 1.0\n v_fma_f64 %1, %1, s[20:21], 1.0\n v_fma_f64 %2, %2, s[20:21], 1.0\n v_fma_f64 %3, %3, s[20:21], 1.0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "s20", "s21"); }
      if (OP == 6) { asm volatile("s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]" : : : "s20", "s21", "s22", "s23"); }
      if (OP == 7) { asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
      if (OP == 8) { asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %4\n v_bfe_u32 %2, %4, 3, 5\n v_and_b32 %3, 63, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(i0)); }
      if (OP == 9) { asm volatile("v_exp_f32 %0, %0\n v_log_f32 %1, %1\n v_rsq_f64 %2, %2\n v_rcp_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(x2), "+v"(f3)); }
      if (OP == 10) { asm volatile("v_ldexp_f64 %0, %0, %4\n v_rndne_f64 %1, %1\n v_cvt_i32_f64 %5, %2\n v_cvt_f64_i32 %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(i0), "+v"(f1)); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + f0 + f1 + f2 + f3;
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int OP>
void run(const char* name) {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4096 * 256 * 8); (void)hipMalloc(&cyc, 4096 * 4 * 8);
  for (int wps : {1, 4}) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    s /= h.size();
    printf("%-28s waves/SIMD %d: %.2f cycles per wave-instruction, %.2f per instruction per SIMD\n", name, wps, s / (REP * OUTER), s / (REP * OUTER) / wps);
  }
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  run<0>("v_cndmask vcc (4 indep)"); run<1>("v_cndmask_e64 sgpr mask"); run<2>("v_cmp_f64->vcc + cndmask"); run<3>("v_cmp_f64_e64 -> sgpr");
  run<4>("v_mul_f64 sgpr operand"); run<5>("v_fma_f64 sgpr + inline const"); run<7>("v_mov_b64");
  run<8>("cvt_f32_u32 / bfe / and"); run<9>("exp_f32 log_f32 rsq_f64 rcp_f32"); run<10>("ldexp rndne cvt_i32 cvt_f64_i32");
  return 0;
}
