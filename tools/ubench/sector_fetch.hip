// Microbenchmark (GPU box): does a load that needs one 64-byte half of a 128-byte line fetch the half or the line?
// One row per lane, a row = one 128-byte line of its own (the rainflow row of a pushing EV).  Kernels:
//   lo16      16 bytes at +0                      hi16      16 bytes at +64
//   lo48      16 bytes at +0, +16, +32            both      16 bytes at +0 and at +64
//   pair64    rows of 64 bytes (two lanes per line), 48 bytes per row
// and the same with a 16-byte store back into the row (`_w`): what a read-modify-write of 16 bytes moves.
// Run under rocprofv3 --pmc with the L2's request-size counters (TCC_EA0_RDREQ_32B / _64B / _128B, TCC_EA0_WRREQ / _64B):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/sector_fetch.hip -o tools/ubench/sector_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE, bool WR>
__global__ __launch_bounds__(256) void k(v4f* __restrict__ rows, float* __restrict__ sink) {
  const size_t lane = (size_t)blockIdx.x * 256 + threadIdx.x;
  v4f* r = rows + lane * (MODE == 4 ? 4 : 8);  // 16-byte units: 64-byte or 128-byte rows
  v4f v = {0, 0, 0, 0};
  if (MODE == 0) v = r[0];
  if (MODE == 1) v = r[4];
  if (MODE == 2) v = r[0] + r[1] + r[2];
  if (MODE == 3) v = r[0] + r[4];
  if (MODE == 4) v = r[0] + r[1] + r[2];
  if (WR) { v.x += 1.0f; r[MODE == 1 ? 4 : 0] = v; }
  else if (v.x == 12345.0f) sink[lane] = v.y;  // never true
}
template <int MODE, bool WR>
void run(const char* name, v4f* rows, float* sink, int lanes) {
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, WR>), dim3(lanes / 256), dim3(256), 0, 0, rows, sink);
  hipDeviceSynchronize();
  printf("%s done\n", name);
}
int main() {
  const int lanes = 1 << 20;  // 1 Mi rows: 128 MiB of 128-byte rows
  v4f* rows; float* sink;
  hipMalloc(&rows, (size_t)lanes * 128); hipMalloc(&sink, (size_t)lanes * 4);
  hipMemset(rows, 0, (size_t)lanes * 128);
  hipDeviceSynchronize();
  run<0, false>("lo16", rows, sink, lanes);   run<1, false>("hi16", rows, sink, lanes);
  run<2, false>("lo48", rows, sink, lanes);   run<3, false>("both", rows, sink, lanes);
  run<4, false>("pair64", rows, sink, lanes);
  run<0, true>("lo16_w", rows, sink, lanes);  run<2, true>("lo48_w", rows, sink, lanes);
  run<3, true>("both_w", rows, sink, lanes);  run<4, true>("pair64_w", rows, sink, lanes);
  return 0;
}
