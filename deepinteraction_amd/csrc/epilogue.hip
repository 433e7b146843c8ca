// Bias (+ residual) (+ ReLU) IN PLACE on a channels-last fp16 map: the epilogue of the library convolutions of the frozen
// backbones (round 6).  torch's `F.conv2d(x, w, b)` on ROCm is MIOpen's convolution followed by a separate bias kernel, and
// `relu_` / `add_` are two more passes over the map - in the frozen ResNet-50 + FPN of `--from-images` (reference config
// Fusion_0075_refactor.py:120-145: `img_backbone` / `img_neck`, `detectors/deepinteraction.py:100-118`) these passes took
// about as long as the convolutions themselves.  One pass instead of two (conv-bias-ReLU) or three (a bottleneck's
// conv-bias, + identity, ReLU):   y[p][c] = act(y[p][c] + bias[c] (+ z[p][c])).
#include <math.h>

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "di_common.h"

namespace di {
namespace ep {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void bias_act_kernel(__half *__restrict__ y, const float *__restrict__ bias,
                                                       const __half *__restrict__ z, long long n8, int C, int relu) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n8; t += stride) {
    const int c = (int)((t * 8) % C);
    const uint4 raw = reinterpret_cast<const uint4 *>(y)[t];
    const h8 v = __builtin_bit_cast(h8, raw);
    const float4 b0 = *reinterpret_cast<const float4 *>(bias + c), b1 = *reinterpret_cast<const float4 *>(bias + c + 4);
    float f[8] = {(float)v[0] + b0.x, (float)v[1] + b0.y, (float)v[2] + b0.z, (float)v[3] + b0.w,
                  (float)v[4] + b1.x, (float)v[5] + b1.y, (float)v[6] + b1.z, (float)v[7] + b1.w};
    if (z) {
      const h8 r = __builtin_bit_cast(h8, reinterpret_cast<const uint4 *>(z)[t]);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
    }
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)(relu ? fmaxf(f[e], 0.f) : f[e]);
    reinterpret_cast<uint4 *>(y)[t] = __builtin_bit_cast(uint4, o);
  }
}

// FPN top-down step in place: lo[n][y][x][:] += hi[n][sy(y)][sx(x)][:], nearest neighbour as `F.interpolate(mode='nearest')` picks it
// (source index = min(floor(dst * in / out), in - 1), the scale in float32) - torch materialises the up-sampled map (one pass) and
// adds it (another).
__global__ __launch_bounds__(256) void upsample_add_kernel(__half *__restrict__ lo, const __half *__restrict__ hi, int n, int Hl, int Wl,
                                                           int Hh, int Wh, int C8, float sy, float sx) {
  const long long total = (long long)n * Hl * Wl * C8, stride = (long long)gridDim.x * 256;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
    const int c = (int)(t % C8);
    long long p = t / C8;
    const int x = (int)(p % Wl);
    p /= Wl;
    const int y = (int)(p % Hl), img = (int)(p / Hl);
    const int ys = min((int)floorf(y * sy), Hh - 1), xs = min((int)floorf(x * sx), Wh - 1);
    const h8 a = __builtin_bit_cast(h8, reinterpret_cast<const uint4 *>(lo)[t]);
    const h8 b = __builtin_bit_cast(h8, reinterpret_cast<const uint4 *>(hi)[((long long)(img * Hh + ys) * Wh + xs) * C8 + c]);
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)a[e] + (float)b[e]);
    reinterpret_cast<uint4 *>(lo)[t] = __builtin_bit_cast(uint4, o);
  }
}

}  // namespace ep
}  // namespace di

extern "C" int di_bias_act_inplace(void *y, const float *bias, const void *residual, long long npix, int C, int relu, void *stream) {
  DI_REQUIRE(npix >= 0 && C > 0 && C % 8 == 0, "map of %lld texels x %d channels (a multiple of 8)", npix, C);
  if (npix == 0) return DI_OK;
  const long long n8 = npix * (C / 8);
  const int n_cu = di::device_cus();
  if (n_cu <= 0) return DI_ERR_LAUNCH;
  long long grid = (n8 + 255) / 256;
  if (grid > (long long)n_cu * 16) grid = (long long)n_cu * 16;
  hipLaunchKernelGGL(di::ep::bias_act_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (__half *)y, bias,
                     (const __half *)residual, n8, C, relu);
  return di::check_launch("bias_act_inplace");
}

extern "C" int di_upsample_add_inplace(void *lo, const void *hi, int n, int Hl, int Wl, int Hh, int Wh, int C, void *stream) {
  DI_REQUIRE(n > 0 && Hl > 0 && Wl > 0 && Hh > 0 && Wh > 0 && C > 0 && C % 8 == 0, "maps %d x (%d x %d) += (%d x %d), %d channels", n, Hl,
             Wl, Hh, Wh, C);
  const long long total = (long long)n * Hl * Wl * (C / 8);
  const int n_cu = di::device_cus();
  if (n_cu <= 0) return DI_ERR_LAUNCH;
  long long grid = (total + 255) / 256;
  if (grid > (long long)n_cu * 16) grid = (long long)n_cu * 16;
  hipLaunchKernelGGL(di::ep::upsample_add_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (__half *)lo, (const __half *)hi,
                     n, Hl, Wl, Hh, Wh, C / 8, (float)Hh / (float)Hl, (float)Wh / (float)Wl);
  return di::check_launch("upsample_add_inplace");
}
