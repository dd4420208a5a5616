// Device side of tools/ubench/aql_fence.cpp (compiled to a code object of its own: hipcc --genco).  Every workgroup owns the same
// chunk of the buffer in every launch -- the step kernel's situation: env e is always workgroup e / 4, i.e. always the same die.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void k_empty(double* buf, int words) {}
// read-modify-write of the workgroup's chunk: after n launches every word must hold n (a stale read loses an increment)
extern "C" __global__ __launch_bounds__(256) void k_touch(double* buf, int words) {
  double* c = buf + (size_t)blockIdx.x * words;
  for (int j = threadIdx.x; j < words; j += 256) c[j] += 1.0;
}
// the same plus a dependent chain of eight loads per lane inside the chunk before the update: exposes the latency of where the
// chunk is found (the die's L2 / beyond it)
extern "C" __global__ __launch_bounds__(256) void k_chain(double* buf, int words) {
  double* c = buf + (size_t)blockIdx.x * words;
  int at = threadIdx.x;
  double acc = 0.0;
  for (int r = 0; r < 8; ++r) {
    const double v = c[at];
    acc += v;
    at = (at * 5 + 17 + ((int)v & 1) * 32) & (words - 1);  // words: a power of two
  }
  for (int j = threadIdx.x; j < words; j += 256) c[j] += 1.0;
  if (acc < 0.0) c[0] = acc;  // never
}
