// Shared device/host helpers for libdeepinteraction_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/deepinteraction_hip.h"

namespace di {
// Device word ADDED to the attention-dropout seed of the pillar attention (forward and backward) when set: a captured
// training step bakes the host seed into its graph, the device word is rewritten before every replay (di_i2p_set_seed_ptr).
const unsigned long long *i2p_seed_ptr();


// ---- host side: error plumbing (no exceptions across the C ABI)
void set_error(const char *fmt, ...);
int check_launch(const char *what);
// Launch state that belongs to a DEVICE, not to the process (one process may drive several): the CU count of the current
// device (<= 0: error set), and "the dynamic-LDS limit of this kernel has been raised on the current device" (one bit
// per device ordinal; idempotent, a race only repeats the runtime call).
int device_cus();
struct LdsRaised {
  unsigned long long done = 0;
};
int ensure_lds(LdsRaised &state, const void *kernel, int bytes);
// Measurement: events BOUND TO A DISPATCH (hipExtLaunchKernelGGL's start / stop events carry the kernel's own begin / end time
// stamps, what rocprofv3 reports; events recorded on the stream around a launch add the 2-3 us of their own packets).
// di_timed_begin arms the calling thread; the next launch site that supports it takes the pair (and clears the slot).
bool take_launch_events(hipEvent_t &start, hipEvent_t &stop);

#define DI_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      di::set_error(__VA_ARGS__);        \
      return DI_ERR_ARG;                 \
    }                                    \
  } while (0)

// ---- device side
// A wavefront is 64 lanes; the gather kernels give 16 lanes to one texel
// (8 channels = 16 B fp16 / 32 B fp32 per lane -> one 256/512 B coalesced row).
constexpr int kLanesPerTexel = 16;
constexpr int kChPerLane = 8;

template <typename T>
struct Pack8;  // 8 consecutive channels as stored in memory
template <>
struct Pack8<float> {
  float4 a, b;
};
template <>
struct Pack8<__half> {
  uint4 r;
};

__device__ __forceinline__ Pack8<float> ld8(const float *p) {
  Pack8<float> v;
  v.a = reinterpret_cast<const float4 *>(p)[0];
  v.b = reinterpret_cast<const float4 *>(p)[1];
  return v;
}
__device__ __forceinline__ Pack8<__half> ld8(const __half *p) {
  Pack8<__half> v;
  v.r = *reinterpret_cast<const uint4 *>(p);
  return v;
}
__device__ __forceinline__ void st8(float *p, const Pack8<float> &v) {
  reinterpret_cast<float4 *>(p)[0] = v.a;
  reinterpret_cast<float4 *>(p)[1] = v.b;
}
__device__ __forceinline__ void st8(__half *p, const Pack8<__half> &v) {
  *reinterpret_cast<uint4 *>(p) = v.r;
}
template <typename T>
__device__ __forceinline__ Pack8<T> zero8();
template <>
__device__ __forceinline__ Pack8<float> zero8<float>() {
  Pack8<float> v;
  v.a = make_float4(0, 0, 0, 0);
  v.b = v.a;
  return v;
}
template <>
__device__ __forceinline__ Pack8<__half> zero8<__half>() {
  Pack8<__half> v;
  v.r = make_uint4(0, 0, 0, 0);
  return v;
}

__device__ __forceinline__ void unpack8(const Pack8<float> &v, float (&f)[8]) {
  f[0] = v.a.x; f[1] = v.a.y; f[2] = v.a.z; f[3] = v.a.w;
  f[4] = v.b.x; f[5] = v.b.y; f[6] = v.b.z; f[7] = v.b.w;
}
__device__ __forceinline__ void unpack8(const Pack8<__half> &v, float (&f)[8]) {
  const __half2 *h = reinterpret_cast<const __half2 *>(&v.r);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float2 t = __half22float2(h[i]);
    f[2 * i] = t.x;
    f[2 * i + 1] = t.y;
  }
}
__device__ __forceinline__ Pack8<float> pack8f(const float (&f)[8], float) {
  Pack8<float> v;
  v.a = make_float4(f[0], f[1], f[2], f[3]);
  v.b = make_float4(f[4], f[5], f[6], f[7]);
  return v;
}
__device__ __forceinline__ Pack8<__half> pack8f(const float (&f)[8], __half) {
  Pack8<__half> v;
  __half2 *h = reinterpret_cast<__half2 *>(&v.r);
#pragma unroll
  for (int i = 0; i < 4; ++i) h[i] = __floats2half2_rn(f[2 * i], f[2 * i + 1]);
  return v;
}

// <a,b> over the 8 channels a lane owns, fp32 accumulate.
__device__ __forceinline__ float dot8(const Pack8<float> &a, const Pack8<float> &b, float acc) {
  acc = fmaf(a.a.x, b.a.x, acc); acc = fmaf(a.a.y, b.a.y, acc);
  acc = fmaf(a.a.z, b.a.z, acc); acc = fmaf(a.a.w, b.a.w, acc);
  acc = fmaf(a.b.x, b.b.x, acc); acc = fmaf(a.b.y, b.b.y, acc);
  acc = fmaf(a.b.z, b.b.z, acc); acc = fmaf(a.b.w, b.b.w, acc);
  return acc;
}
typedef _Float16 di_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot8(const Pack8<__half> &a, const Pack8<__half> &b, float acc) {
  // v_dot2_f32_f16: two fp16 products accumulated in fp32
  const di_h2 *x = reinterpret_cast<const di_h2 *>(&a.r);
  const di_h2 *y = reinterpret_cast<const di_h2 *>(&b.r);
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2(x[i], y[i], acc, false);
  return acc;
}

// DPP cross-lane within a row of 16 lanes (no LDS traffic).
template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// every lane of a 16-lane row ends with the row's sum / max
__device__ __forceinline__ float row16_sum(float x) {
  x += dpp<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp<0x141>(x);  // row_half_mirror
  x += dpp<0x140>(x);  // row_mirror
  return x;
}
__device__ __forceinline__ float row16_max(float x) {
  x = fmaxf(x, dpp<0xB1>(x));
  x = fmaxf(x, dpp<0x4E>(x));
  x = fmaxf(x, dpp<0x141>(x));
  x = fmaxf(x, dpp<0x140>(x));
  return x;
}

// Attention dropout of the pillar attention (training): keep/drop decision of key slot `slot` of pillar `p`,
// a counter-based hash of (seed, p, slot) so that forward and backward regenerate the same mask.
__device__ __forceinline__ bool di_keep(unsigned long long seed, int p, int slot, float drop_p) {
  unsigned long long x = seed + (((unsigned long long)(unsigned)p << 8) | (unsigned)slot);
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return (float)(x >> 40) * (1.f / 16777216.f) >= drop_p;
}

// XCD-aware block remap: the dispatcher round-robins consecutive block ids over the
// 8 XCDs (private L2 each); give every XCD one contiguous chunk of the tile list so
// neighbouring tiles (which share halo texels) hit the same L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int nx = 8;
  const int q = nblk / nx, r = nblk % nx;
  const int xcd = bid % nx, i = bid / nx;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + i;
}

}  // namespace di
