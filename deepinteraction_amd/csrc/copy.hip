// Device-to-device copy of one large buffer in ONE launch (the `load()` of the next sample into the captured input arena:
// ~110 MB per sample at the benched shape).  torch's `copy_` hands this to the runtime's blit path, which runs it as 6-7
// kernels of ~10 us (67 us per sample, 1.6 TB/s of copy rate); a plain streaming kernel with eight independent 16-byte loads
// per lane in flight does the same bytes at the chip's stream rate.
#include "di_common.h"

namespace di {

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void copy_kernel(const u4 *__restrict__ src, u4 *__restrict__ dst, long long n16) {
  const long long stride = (long long)gridDim.x * 256 * UNROLL;
  for (long long base = ((long long)blockIdx.x * UNROLL) * 256 + threadIdx.x; base < n16; base += stride) {
    u4 r[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long e = base + (long long)u * 256;
      if (e < n16) r[u] = __builtin_nontemporal_load(src + e);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long e = base + (long long)u * 256;
      if (e < n16) __builtin_nontemporal_store(r[u], dst + e);
    }
  }
}

__global__ void copy_tail_kernel(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, long long n) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) dst[e] = src[e];
}

}  // namespace di

extern "C" int di_copy_d2d(void *dst, const void *src, long long bytes, void *stream) {
  DI_REQUIRE(bytes >= 0, "negative size %lld", bytes);
  if (bytes == 0) return DI_OK;
  DI_REQUIRE(dst != nullptr && src != nullptr, "null buffer");
  DI_REQUIRE(((size_t)dst & 15) == 0 && ((size_t)src & 15) == 0, "buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const long long n16 = bytes / 16;
  if (n16 > 0) {
    const int n_cu = di::device_cus();
    if (n_cu <= 0) return DI_ERR_LAUNCH;
    long long blocks = (n16 + 256 * 8 - 1) / (256 * 8);
    if (blocks > (long long)n_cu * 8) blocks = (long long)n_cu * 8;
    hipLaunchKernelGGL(di::copy_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, (const di::u4 *)src, (di::u4 *)dst, n16);
  }
  const long long tail = bytes - n16 * 16;
  if (tail > 0)
    hipLaunchKernelGGL(di::copy_tail_kernel, dim3(1), dim3(64), 0, s, (const unsigned char *)src + n16 * 16,
                       (unsigned char *)dst + n16 * 16, tail);
  return di::check_launch("copy_d2d");
}
