// Microbenchmark (GPU box): issue cost of the float64 / select / convert instructions the step kernel is made of, one wave
// per SIMD and four, so that instruction counts of the kernel can be priced.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define OUTER 64
template <int OP>
__global__ void k(double* out, unsigned long long* cyc, double seed) {
  double a = seed + threadIdx.x * 1e-3, b = 1.000001, c = 0.5, x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
  float f0 = (float)a, f1 = 1.0f;
  int i0 = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int o = 0; o < OUTER; ++o) {
#pragma unroll
    for (int r = 0; r < REP / 4; ++r) {
      if (OP == 0) { asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c)); }
      if (OP == 1) { asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b)); }
      if (OP == 2) { asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b)); }
      if (OP == 3) { asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
      if (OP == 4) { asm volatile("v_div_scale_f64 %0, vcc, %0, %4, %0\n v_div_scale_f64 %1, vcc, %1, %4, %1\n v_div_scale_f64 %2, vcc, %2, %4, %2\n v_div_scale_f64 %3, vcc, %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b) : "vcc"); }
      if (OP == 5) { asm volatile("v_div_fmas_f64 %0, %0, %4, %5\n v_div_fmas_f64 %1, %1, %4, %5\n v_div_fmas_f64 %2, %2, %4, %5\n v_div_fmas_f64 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c) : "vcc"); }
      if (OP == 6) { asm volatile("v_div_fixup_f64 %0, %0, %4, %5\n v_div_fixup_f64 %1, %1, %4, %5\n v_div_fixup_f64 %2, %2, %4, %5\n v_div_fixup_f64 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c)); }
      if (OP == 7) { asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(f0), "+v"(f1) : "v"(i0) : "vcc"); }
      if (OP == 8) { asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %2\n v_cmp_lt_f64 vcc, %2, %3\n v_cmp_lt_f64 vcc, %3, %0" : : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc"); }
      if (OP == 9) { asm volatile("v_cvt_f32_f64 %0, %2\n v_cvt_f32_f64 %1, %3\n v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5" : "+v"(f0), "+v"(f1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3)); }
      if (OP == 10) { asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(f0), "v"(f1)); }
      if (OP == 11) { asm volatile("v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_min_f64 %2, %2, %4\n v_min_f64 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b)); }
      if (OP == 12) { asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(f0), "+v"(f1) : "v"(f1), "v"(f0)); }
      if (OP == 13) { asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %0" : "+v"(f0), "+v"(f1)); }
      if (OP == 14) { asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f0), "+v"(f1)); }
      if (OP == 15) { asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b)); }
      if (OP == 16) { asm volatile("s_mov_b32 s20, 0x12345\n s_mov_b32 s21, 0x12345\n s_add_u32 s20, s20, s21\n s_mul_i32 s21, s20, s21" : : : "s20", "s21"); }
      if (OP == 17) { asm volatile("v_readlane_b32 s20, %0, 3\n v_writelane_b32 %0, s20, 5\n v_readlane_b32 s21, %0, 4\n v_writelane_b32 %0, s21, 6" : "+v"(f0) : : "s20", "s21"); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + f0 + f1;
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int OP>
void run(const char* name) {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 4096 * 256 * 8); hipMalloc(&cyc, 4096 * 4 * 8);
  for (int wps : {1, 2, 4}) {  // waves per SIMD: blocks of 256 threads = 4 waves = one per SIMD; wps blocks per CU
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    s /= h.size();
    printf("%-16s waves/SIMD %d: %.2f cycles per wave-instruction (wave view), %.2f per instruction per SIMD\n", name, wps, s / (REP * OUTER), s / (REP * OUTER) / wps);
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("v_fma_f64"); run<1>("v_mul_f64"); run<2>("v_add_f64"); run<3>("v_rcp_f64"); run<4>("v_div_scale_f64");
  run<5>("v_div_fmas_f64"); run<6>("v_div_fixup_f64"); run<7>("v_cndmask_b32"); run<8>("v_cmp_lt_f64"); run<9>("v_cvt_f32_f64");
  run<10>("v_cvt_f64_f32"); run<11>("v_max/min_f64"); run<12>("v_fma_f32"); run<13>("v_mov_b32"); run<14>("v_mov_b32_dpp");
  run<15>("v_lshl_add_u64"); run<16>("s_alu"); run<17>("v_read/writelane");
  return 0;
}
