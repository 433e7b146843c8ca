// Training-mode BatchNorm2d (+ ReLU) of the necks' ConvBNReLU blocks on channels-last maps, forward and backward
// (reference encoder_utils.py:11-34: nn.Conv2d -> nn.BatchNorm2d -> nn.ReLU in train() mode, 38 instances per step).
// The library path is 3 + 1 launches per direction (MIOpen spatial BatchNorm with two finalisation kernels of 19-29 us that
// move no data, then the activation / its mask); here a direction is
//   partial sums per workgroup  ->  finalise (deterministic: a fixed-order sum over the workgroups; coefficients, saved
//   statistics, running statistics)  ->  apply (normalise + ReLU, or the input gradient with the ReLU mask recomputed)
// and the map is read once per pass in 16-byte pieces.  A map is (pixels, C), C a multiple of 8, <= 256; statistics float32
// whatever the map's type; biased variance for the normalisation, unbiased for the running estimate (torch semantics).
#include "di_common.h"

namespace di {
namespace bn {

constexpr int kMaxBlocks = 512;

// every 16-byte piece of the map belongs to (pixel, channel group cg = 8 channels); a thread keeps ONE channel group
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void partial_kernel(const T *__restrict__ x, const T *__restrict__ dy,
                                                      const float *__restrict__ coef,   // BWD: [mean | rstd | gamma | beta] x C
                                                      long long npix, int C, int relu, float *__restrict__ part) {
  __shared__ float red[2][256][8];
  const int lpr = C >> 3, rows = 256 / lpr;          // lanes per pixel row, pixel rows per workgroup iteration
  const int cg = threadIdx.x % lpr, r = threadIdx.x / lpr;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = b[e] = 0.f;
  float mean[8], rstd[8], gam[8], bet[8];
  if (BWD) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mean[e] = coef[cg * 8 + e];
      rstd[e] = coef[C + cg * 8 + e];
      gam[e] = coef[2 * C + cg * 8 + e];
      bet[e] = coef[3 * C + cg * 8 + e];
    }
  }
  if (r < rows)
    for (long long p = (long long)blockIdx.x * rows + r; p < npix; p += (long long)gridDim.x * rows) {
      float v[8];
      unpack8(ld8(x + p * C + cg * 8), v);
      if (!BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          a[e] += v[e];
          b[e] = fmaf(v[e], v[e], b[e]);
        }
      } else {
        float g[8];
        unpack8(ld8(dy + p * C + cg * 8), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (v[e] - mean[e]) * rstd[e];
          const float gg = (relu && fmaf(xh, gam[e], bet[e]) <= 0.f) ? 0.f : g[e];
          a[e] += gg;
          b[e] = fmaf(gg, xh, b[e]);
        }
      }
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][threadIdx.x][e] = a[e];
    red[1][threadIdx.x][e] = b[e];
  }
  __syncthreads();
  // thread t < 2C: quantity t / C, channel t % C - the sum over the workgroup's rows, in row order
  for (int t = threadIdx.x; t < 2 * C; t += 256) {
    const int qn = t / C, c = t - qn * C;
    float s = 0.f;
    for (int rr = 0; rr < rows; ++rr) s += red[qn][rr * lpr + (c >> 3)][c & 7];
    part[(long long)blockIdx.x * 2 * C + t] = s;
  }
}

// sum over the workgroups' partial results of (quantity qn, channel c): lane l adds blocks l, l + 64, ... in order, then a
// fixed butterfly over the lanes - the same order in every run
__device__ __forceinline__ float block_sum(const float *__restrict__ part, int nblk, int C, int col) {
  float s = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 64) s += part[(long long)b * 2 * C + col];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  return s;
}

// forward: sums -> coef = [scale | shift] for the apply pass, saved [mean | rstd | gamma | beta], running statistics
// (one 64-lane workgroup per channel)
__global__ __launch_bounds__(64) void finalize_fwd_kernel(const float *__restrict__ part, int nblk, int C, long long npix,
                                    const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                    float momentum, float *__restrict__ running_mean, float *__restrict__ running_var,
                                    long long *__restrict__ num_batches, float *__restrict__ coef, float *__restrict__ saved) {
  const int c = blockIdx.x;
  const float s = block_sum(part, nblk, C, c), q = block_sum(part, nblk, C, C + c);
  if (threadIdx.x != 0) return;
  if (c == 0 && num_batches) *num_batches += 1;
  const float n = (float)npix;
  const float mean = s / n;
  const float var = fmaxf(q / n - mean * mean, 0.f);
  const float rstd = rsqrtf(var + eps);
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  coef[c] = rstd * g;
  coef[C + c] = bt - mean * rstd * g;
  saved[c] = mean;                                   // [mean | rstd | gamma | beta]: the backward's coefficients
  saved[C + c] = rstd;
  saved[2 * C + c] = g;
  saved[3 * C + c] = bt;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void apply_fwd_kernel(const T *__restrict__ x, const float *__restrict__ coef,
                                                        long long npix, int C, int relu, T *__restrict__ y) {
  const int lpr = C >> 3, rows = 256 / lpr;
  const int cg = threadIdx.x % lpr, r = threadIdx.x / lpr;
  if (r >= rows) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = coef[cg * 8 + e];
    sh[e] = coef[C + cg * 8 + e];
  }
  for (long long p = (long long)blockIdx.x * rows + r; p < npix; p += (long long)gridDim.x * rows) {
    float v[8];
    unpack8(ld8(x + p * C + cg * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = fmaf(v[e], sc[e], sh[e]);
      if (relu) v[e] = fmaxf(v[e], 0.f);
    }
    st8(y + p * C + cg * 8, pack8f(v, T()));
  }
}

// backward: sums -> gradients of the affine parameters and the per-channel terms of dx
__global__ __launch_bounds__(64) void finalize_bwd_kernel(const float *__restrict__ part, int nblk, int C, long long npix,
                                    float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ terms) {
  const int c = blockIdx.x;
  const float s = block_sum(part, nblk, C, c), q = block_sum(part, nblk, C, C + c);
  if (threadIdx.x != 0) return;
  if (dgamma) dgamma[c] = q;
  if (dbeta) dbeta[c] = s;
  const float n = (float)npix;
  terms[c] = s / n;                                  // mean of dy
  terms[C + c] = q / n;                              // mean of dy * xhat
}

template <typename T>
__global__ __launch_bounds__(256) void apply_bwd_kernel(const T *__restrict__ x, const T *__restrict__ dy,
                                                        const float *__restrict__ coef, const float *__restrict__ terms,
                                                        long long npix, int C, int relu, T *__restrict__ dx) {
  const int lpr = C >> 3, rows = 256 / lpr;
  const int cg = threadIdx.x % lpr, r = threadIdx.x / lpr;
  if (r >= rows) return;
  float mean[8], rstd[8], gam[8], bet[8], m1[8], m2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mean[e] = coef[cg * 8 + e];
    rstd[e] = coef[C + cg * 8 + e];
    gam[e] = coef[2 * C + cg * 8 + e];
    bet[e] = coef[3 * C + cg * 8 + e];
    m1[e] = terms[cg * 8 + e];
    m2[e] = terms[C + cg * 8 + e];
  }
  for (long long p = (long long)blockIdx.x * rows + r; p < npix; p += (long long)gridDim.x * rows) {
    float v[8], g[8];
    unpack8(ld8(x + p * C + cg * 8), v);
    unpack8(ld8(dy + p * C + cg * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (v[e] - mean[e]) * rstd[e];
      const float gg = (relu && fmaf(xh, gam[e], bet[e]) <= 0.f) ? 0.f : g[e];
      v[e] = gam[e] * rstd[e] * (gg - m1[e] - xh * m2[e]);
    }
    st8(dx + p * C + cg * 8, pack8f(v, T()));
  }
}

static int grid_for(long long npix, int C) {
  const int rows = 256 / (C >> 3);
  const long long need = (npix + rows - 1) / rows;
  return (int)(need < kMaxBlocks ? need : kMaxBlocks);
}

static int check(long long npix, int C, int dtype) {
  DI_REQUIRE(npix > 0 && C >= 8 && C % 8 == 0 && C <= 256, "BatchNorm map of %lld x %d (C a multiple of 8, <= 256)", npix, C);
  DI_REQUIRE(dtype == DI_F16 || dtype == DI_F32, "unsupported dtype %d", dtype);
  return DI_OK;
}

}  // namespace bn
}  // namespace di

extern "C" {

int di_bn_workspace_floats(int C) { return di::bn::kMaxBlocks * 2 * C + 4 * C; }

int di_bn_train_fwd(const void *x, long long npix, int C, int dtype, const float *gamma, const float *beta, float eps,
                    float momentum, float *running_mean, float *running_var, long long *num_batches, int relu, void *y,
                    float *saved, float *workspace, void *stream) {
  using namespace di::bn;
  if (int rc = check(npix, C, dtype)) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = grid_for(npix, C);
  float *part = workspace, *coef = workspace + (size_t)kMaxBlocks * 2 * C;
  if (dtype == DI_F16)
    hipLaunchKernelGGL((partial_kernel<__half, false>), dim3(nblk), dim3(256), 0, s, (const __half *)x, nullptr, nullptr,
                       npix, C, 0, part);
  else
    hipLaunchKernelGGL((partial_kernel<float, false>), dim3(nblk), dim3(256), 0, s, (const float *)x, nullptr, nullptr, npix,
                       C, 0, part);
  hipLaunchKernelGGL(finalize_fwd_kernel, dim3(C), dim3(64), 0, s, part, nblk, C, npix, gamma, beta, eps, momentum,
                     running_mean, running_var, num_batches, coef, saved);
  if (dtype == DI_F16)
    hipLaunchKernelGGL(apply_fwd_kernel<__half>, dim3(nblk), dim3(256), 0, s, (const __half *)x, coef, npix, C, relu,
                       (__half *)y);
  else
    hipLaunchKernelGGL(apply_fwd_kernel<float>, dim3(nblk), dim3(256), 0, s, (const float *)x, coef, npix, C, relu,
                       (float *)y);
  return di::check_launch("bn_train_fwd");
}

int di_bn_train_bwd(const void *x, const void *grad_y, long long npix, int C, int dtype, const float *coef, int relu,
                    void *grad_x, float *grad_gamma, float *grad_beta, float *workspace, void *stream) {
  using namespace di::bn;
  if (int rc = check(npix, C, dtype)) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = grid_for(npix, C);
  float *part = workspace, *terms = workspace + (size_t)kMaxBlocks * 2 * C;
  if (dtype == DI_F16)
    hipLaunchKernelGGL((partial_kernel<__half, true>), dim3(nblk), dim3(256), 0, s, (const __half *)x, (const __half *)grad_y,
                       coef, npix, C, relu, part);
  else
    hipLaunchKernelGGL((partial_kernel<float, true>), dim3(nblk), dim3(256), 0, s, (const float *)x, (const float *)grad_y,
                       coef, npix, C, relu, part);
  hipLaunchKernelGGL(finalize_bwd_kernel, dim3(C), dim3(64), 0, s, part, nblk, C, npix, grad_gamma, grad_beta, terms);
  if (dtype == DI_F16)
    hipLaunchKernelGGL(apply_bwd_kernel<__half>, dim3(nblk), dim3(256), 0, s, (const __half *)x, (const __half *)grad_y, coef,
                       terms, npix, C, relu, (__half *)grad_x);
  else
    hipLaunchKernelGGL(apply_bwd_kernel<float>, dim3(nblk), dim3(256), 0, s, (const float *)x, (const float *)grad_y, coef,
                       terms, npix, C, relu, (float *)grad_x);
  return di::check_launch("bn_train_bwd");
}

}  // extern "C"
