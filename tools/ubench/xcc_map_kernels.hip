// Device side of tools/ubench/xcc_map.cpp: every workgroup writes down where it runs (HW_REG_XCC_ID, HW_REG_HW_ID).
#include <hip/hip_runtime.h>
extern "C" __global__ void k_where(unsigned* out) {
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));      // HW_REG_XCC_ID, all 32 bits
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
  }
}
