// hipcc -O3 --offload-arch=gfx950 -S: k2 adds the FIRST result of v_permlane16_swap to itself (v_add_f32 v1, v1, v1), k3 (both results
// through an empty asm statement) adds the two (v_add_f32 v1, v1, v2).  ROCm 7.2.0.  Found in round 6 (csrc/i2p_dense.hip).
#include <hip/hip_runtime.h>
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__global__ void k2(float* out) {
  float x = out[threadIdx.x];
  float y = out[threadIdx.x + 64];
  u2v r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
  out[threadIdx.x] = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__global__ void k3(float* out) {
  float x = out[threadIdx.x];
  float y = out[threadIdx.x + 64];
  u2v r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
  unsigned a = r[0], b = r[1];
  asm volatile("" : "+v"(a), "+v"(b));
  out[threadIdx.x] = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
