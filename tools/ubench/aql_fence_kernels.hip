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

// ---- dependent launches WITHOUT the barrier bit: workgroup w of launch g waits for workgroup w of launch g - 1 (a flag per
// workgroup, written and polled inside the die's L2) instead of the whole launch waiting for the whole previous launch
__device__ __forceinline__ int poll_flag(const int* p, int flavor) {
  int v;
  if (flavor == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void wait_turn(const int* done, int gen, int flavor, int* err) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    int n = 0;
    while (poll_flag(done + blockIdx.x, flavor) != gen && n < (1 << 14)) { __builtin_amdgcn_s_sleep(2); ++n; }
    ok = n < (1 << 14);
    if (!ok) atomicAdd(err, 1);
  }
  __syncthreads();
  // the CU's vector cache may hold lines of this chunk from an earlier launch: invalidate it (flavors 0, 1) -- or let the loads of
  // the chunk go past it (flavors 2, 3: ld_state) -- or do nothing (flavor 4: must lose updates, the control)
  if (flavor <= 1) asm volatile("buffer_inv sc1" ::: "memory");
}
__device__ __forceinline__ double ld_state(const double* p, int flavor) {
  double v;
  if (flavor == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (flavor == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else v = *p;
  return v;
}
__device__ __forceinline__ void pass_turn(int* done, int gen, int flavor) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's stores have reached the L2
  __syncthreads();
  if (threadIdx.x == 0) {
    if (flavor == 0) __hip_atomic_store(done + blockIdx.x, gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else asm volatile("global_store_dword %0, %1, off sc0" : : "v"(done + blockIdx.x), "v"(gen + 1) : "memory");
  }
}
extern "C" __global__ __launch_bounds__(256) void k_empty_dep(double* buf, int words, int* done, int gen, int flavor, int* err) {
  wait_turn(done, gen, flavor, err);
  pass_turn(done, gen, flavor);
}
extern "C" __global__ __launch_bounds__(256) void k_touch_dep(double* buf, int words, int* done, int gen, int flavor, int* err) {
  wait_turn(done, gen, flavor, err);
  double* c = buf + (size_t)blockIdx.x * words;
  for (int j = threadIdx.x; j < words; j += 256) c[j] = ld_state(c + j, flavor) + 1.0;
  pass_turn(done, gen, flavor);
}
extern "C" __global__ __launch_bounds__(256) void k_chain_dep(double* buf, int words, int* done, int gen, int flavor, int* err) {
  wait_turn(done, gen, flavor, err);
  double* c = buf + (size_t)blockIdx.x * words;
  int at = threadIdx.x;
  double acc = 0.0;
  for (int r = 0; r < 8; ++r) {
    const double v = ld_state(c + at, flavor);
    acc += v;
    at = (at * 5 + 17 + ((int)v & 1) * 32) & (words - 1);
  }
  for (int j = threadIdx.x; j < words; j += 256) c[j] = ld_state(c + j, flavor) + 1.0;
  if (acc < 0.0) c[0] = acc;
  pass_turn(done, gen, flavor);
}

// read the workgroup's chunk, write one word: how much CLEAN data a die's L2 carries from launch to launch (k_touch: dirty data)
extern "C" __global__ __launch_bounds__(256) void k_read(double* buf, int words) {
  double* c = buf + (size_t)blockIdx.x * words;
  double acc = 0.0;
  for (int j = threadIdx.x; j < words - 256; j += 256) acc += c[j];
  c[words - 256 + threadIdx.x] = acc * 0.0 + c[words - 256 + threadIdx.x] + 1.0;
}
