// Sparse -> dense depth completion on the GPU (gfx950).
//
// Re-statement of ip_basic `fill_in_multiscale(extrapolate=False, blur_type='bilateral')`
// (reference models/utils/ip_basic/depth_map_utils.py:134-287), which the reference
// runs on the HOST with OpenCV in the middle of every encoder layer
// (encoder_utils.py:175-182: GPU -> CPU -> 6 x cv2 pipelines -> GPU).  Here the whole
// pipeline stays on the device as a chain of tiny stencil kernels over the
// (n_views, Hi, Wi) maps (134k pixels at the Fusion_0075 shape - L2 resident), and the
// caller runs it once per sample instead of once per layer (it does not depend on
// features).  Exact float compare/min/max/median semantics; the bilateral weights follow
// OpenCV's 4096-bin interpolated exp LUT.
#include <float.h>

#include "di_common.h"

namespace di {

__device__ __forceinline__ float inv_depth(float d) { return d > 0.1f ? 100.f - d : d; }  // :171-174

// first valid row of a column (dc_col_first_kernel); H = "no valid pixel",
// which np.argmax reports as row 0 (:209-213, :228)
__device__ __forceinline__ int first_row(const int32_t *__restrict__ first, int idx, int H) {
  const int f = first[idx];
  return f >= H ? 0 : f;
}

// ---- stage 1 (:166-196): three depth-binned cross dilations of the inverted map
template <int R>
__device__ __forceinline__ float cross_dilate_bin(const float *__restrict__ d, int H, int W, int y,
                                                  int x, float lo, float hi) {
  float m = -INFINITY;
#pragma unroll
  for (int o = -R; o <= R; ++o) {
    const int yy = y + o;
    if (yy >= 0 && yy < H) {
      const float v = d[yy * W + x];
      m = fmaxf(m, (v > lo && v <= hi) ? inv_depth(v) : 0.f);
    }
    const int xx = x + o;
    if (o != 0 && xx >= 0 && xx < W) {
      const float v = d[y * W + xx];
      m = fmaxf(m, (v > lo && v <= hi) ? inv_depth(v) : 0.f);
    }
  }
  return m;
}

__global__ __launch_bounds__(256) void dc_multiscale_kernel(const float *__restrict__ in,
                                                            float *__restrict__ out, int32_t *__restrict__ first_a,
                                                            int32_t *__restrict__ first_b, int V, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V * H * W) return;
  const int v = i / (H * W), r = i - v * H * W, y = r / W, x = r - y * W;
  const float *d = in + (size_t)v * H * W;
  const float far_ = cross_dilate_bin<1>(d, H, W, y, x, 30.0f, INFINITY);   // CROSS_KERNEL_3
  const float med = cross_dilate_bin<2>(d, H, W, y, x, 15.0f, 30.0f);       // CROSS_KERNEL_5
  const float near_ = cross_dilate_bin<3>(d, H, W, y, x, 0.1f, 15.0f);      // CROSS_KERNEL_7
  float s2 = inv_depth(d[r]);
  if (far_ > 0.1f) s2 = far_;
  if (med > 0.1f) s2 = med;
  if (near_ > 0.1f) s2 = near_;
  out[i] = s2;
}

// ---- full-kernel dilate / erode with OpenCV's default border (ignored)
template <int R, bool IS_MAX>
__device__ __forceinline__ float box_extreme(const float *__restrict__ d, int H, int W, int y, int x) {
  float m = IS_MAX ? -INFINITY : INFINITY;
  for (int dy = -R; dy <= R; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int dx = -R; dx <= R; ++dx) {
      const int xx = x + dx;
      if (xx < 0 || xx >= W) continue;
      const float v = d[yy * W + xx];
      m = IS_MAX ? fmaxf(m, v) : fminf(m, v);
    }
  }
  return m;
}

template <int R, bool IS_MAX>
__global__ __launch_bounds__(256) void dc_box_kernel(const float *__restrict__ in,
                                                     float *__restrict__ out, int V, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V * H * W) return;
  const int v = i / (H * W), r = i - v * H * W, y = r / W, x = r - y * W;
  out[i] = box_extreme<R, IS_MAX>(in + (size_t)v * H * W, H, W, y, x);
}

// ---- the morphological close (5x5 dilate then 5x5 erode, :198-200) in ONE launch: a 16 x 64 tile with a halo of 4; texels
// outside the map are -inf for the dilation and +inf for the erosion (OpenCV's default border: ignored).  Same float
// max / min over the same values as dc_box_kernel<2, true> followed by dc_box_kernel<2, false>: bit-identical.
constexpr int kTH = 16, kTW = 64;
__global__ __launch_bounds__(256) void dc_close_kernel(const float *__restrict__ in, float *__restrict__ out, int V, int H,
                                                       int W) {
  __shared__ float a[kTH + 8][kTW + 8 + 1], b[kTH + 4][kTW + 4 + 1];
  const int v = blockIdx.z, y0 = blockIdx.y * kTH, x0 = blockIdx.x * kTW, tid = threadIdx.x;
  const float *d = in + (size_t)v * H * W;
  for (int e = tid; e < (kTH + 8) * (kTW + 8); e += 256) {
    const int ly = e / (kTW + 8), lx = e - ly * (kTW + 8), gy = y0 - 4 + ly, gx = x0 - 4 + lx;
    a[ly][lx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? d[gy * W + gx] : -INFINITY;
  }
  __syncthreads();
  for (int e = tid; e < (kTH + 4) * (kTW + 4); e += 256) {
    const int ly = e / (kTW + 4), lx = e - ly * (kTW + 4), gy = y0 - 2 + ly, gx = x0 - 2 + lx;
    float m = INFINITY;                                      // outside the map: ignored by the erosion
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      m = -INFINITY;
#pragma unroll
      for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) m = fmaxf(m, a[ly + dy][lx + dx]);
    }
    b[ly][lx] = m;
  }
  __syncthreads();
  for (int e = tid; e < kTH * kTW; e += 256) {
    const int ly = e / kTW, lx = e - ly * kTW, gy = y0 + ly, gx = x0 + lx;
    if (gy >= H || gx >= W) continue;
    float m = INFINITY;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) m = fminf(m, b[ly + dy][lx + dx]);
    out[(size_t)v * H * W + gy * W + gx] = m;
  }
}

// ---- the six hole-filling passes of :241-245 (`dc_fill_kernel<2, true>` six times) in ONE launch: a 16 x 64 tile with a
// halo of 12 in two LDS buffers; pass k is evaluated on the region that still has valid neighbours (it shrinks by 2 per
// pass), texels outside the map are -inf (ignored by the dilation) and never filled.  The same comparisons and maxima over
// the same values: bit-identical.
template <int ITERS>
__global__ __launch_bounds__(256) void dc_fill_iter_kernel(const float *__restrict__ in, const int32_t *__restrict__ first,
                                                           float *__restrict__ out, int V, int H, int W) {
  constexpr int HALO = 2 * ITERS, LH = kTH + 2 * HALO, LW = kTW + 2 * HALO;
  __shared__ float buf[2][LH][LW + 1];
  __shared__ int s_first[LW];
  const int v = blockIdx.z, y0 = blockIdx.y * kTH - HALO, x0 = blockIdx.x * kTW - HALO, tid = threadIdx.x;
  const float *d = in + (size_t)v * H * W;
  for (int e = tid; e < LH * LW; e += 256) {
    const int ly = e / LW, lx = e - ly * LW, gy = y0 + ly, gx = x0 + lx;
    buf[0][ly][lx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? d[gy * W + gx] : -INFINITY;
  }
  for (int lx = tid; lx < LW; lx += 256) {
    const int gx = x0 + lx;
    s_first[lx] = (gx >= 0 && gx < W) ? first_row(first, v * W + gx, H) : H;
  }
  __syncthreads();
  int cur = 0;
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    const int m = 2 * (it + 1), rh = LH - 2 * m, rw = LW - 2 * m;
    for (int e = tid; e < rh * rw; e += 256) {
      const int ly = m + e / rw, lx = m + e % rw, gy = y0 + ly, gx = x0 + lx;
      float o = -INFINITY;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const float c = buf[cur][ly][lx];
        o = c;
        if (c < 0.1f && gy >= s_first[lx]) {
          float mx = -INFINITY;
#pragma unroll
          for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) mx = fmaxf(mx, buf[cur][ly + dy][lx + dx]);
          o = mx;
        }
      }
      buf[cur ^ 1][ly][lx] = o;
    }
    __syncthreads();
    cur ^= 1;
  }
  for (int e = tid; e < kTH * kTW; e += 256) {
    const int ly = HALO + e / kTW, lx = HALO + e % kTW, gy = y0 + ly, gx = x0 + lx;
    if (gy < H && gx < W) out[(size_t)v * H * W + gy * W + gx] = buf[cur][ly][lx];
  }
}

// ---- 5x5 median, BORDER_REPLICATE (cv2.medianBlur for CV_32F)
__device__ __forceinline__ float median25(const float *__restrict__ d, int H, int W, int y, int x) {
  float a[25];
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy) {
    const int yy = min(max(y + dy, 0), H - 1);
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      const int xx = min(max(x + dx, 0), W - 1);
      a[(dy + 2) * 5 + dx + 2] = d[yy * W + xx];
    }
  }
  // 13 bubble passes: the 12 largest settle in a[13..24], a[12] is the median
#pragma unroll
  for (int p = 0; p < 13; ++p) {
#pragma unroll
    for (int j = 0; j < 24 - p; ++j) {
      const float lo = fminf(a[j], a[j + 1]), hi = fmaxf(a[j], a[j + 1]);
      a[j] = lo;
      a[j + 1] = hi;
    }
  }
  return a[12];
}

// First valid row of every column of a map (H = none): one thread per column, plain stores.  (Round 2 accumulated these
// with global atomicMin from the producing kernels after a plain-store initialisation by an earlier kernel of the chain -
// the pattern that gave wrong results from the second hipGraph replay on in the top-k kernels; two 3-us launches on the
// side stream cost nothing, the chain is hidden under the shared convolutions.)
__global__ __launch_bounds__(256) void dc_col_first_kernel(const float *__restrict__ in, int32_t *__restrict__ first, int V,
                                                           int H, int W) {
  // a block = 64 columns x 4 row quarters: coalesced row reads, every load independent (a serial scan of one column
  // per thread was 112 dependent strided loads: 32 us)
  __shared__ int part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), yq = threadIdx.x >> 6;
  const int hq = (H + 3) >> 2, y0 = yq * hq, y1 = min(y0 + hq, H);
  int f = H;
  if (col < V * W) {
    const int v = col / W, x = col - v * W;
    const float *d = in + (size_t)v * H * W + x;
    for (int yb = y0; yb < y1; yb += 8) {
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = d[(size_t)min(yb + j, y1 - 1) * W];
#pragma unroll
      for (int j = 7; j >= 0; --j) f = (yb + j < y1 && t[j] > 0.1f && yb + j < f) ? yb + j : f;
    }
  }
  part[yq][threadIdx.x & 63] = f;
  __syncthreads();
  if (yq == 0 && col < V * W) first[col] = min(min(part[0][threadIdx.x], part[1][threadIdx.x]), min(part[2][threadIdx.x], part[3][threadIdx.x]));
}

// :203-206  s4 = s3 > 0.1 ? median(s3) : s3
__global__ __launch_bounds__(256) void dc_median_valid_kernel(const float *__restrict__ in,
                                                              float *__restrict__ out, int32_t *__restrict__ first,
                                                              int V, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V * H * W) return;
  const int v = i / (H * W), r = i - v * H * W, y = r / W, x = r - y * W;
  const float c = in[i];
  const float o = c > 0.1f ? median25(in + (size_t)v * H * W, H, W, y, x) : c;
  out[i] = o;
}

// :216-222  empty = !(s4 > 0.1) & top_mask ; s5 = empty ? dilate9x9(s4) : s4
// :241-245  empty = (s7 < 0.1) & top_mask  ; s7 = empty ? dilate5x5(s7) : s7
template <int R, bool STRICT_LT>
__global__ __launch_bounds__(256) void dc_fill_kernel(const float *__restrict__ in,
                                                      const int32_t *__restrict__ first,
                                                      float *__restrict__ out, int32_t *__restrict__ first_out,
                                                      int V, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V * H * W) return;
  const int v = i / (H * W), r = i - v * H * W, y = r / W, x = r - y * W;
  const float c = in[i];
  const bool top = y >= first_row(first, v * W + x, H);
  const bool empty = (STRICT_LT ? (c < 0.1f) : !(c > 0.1f)) && top;
  const float o = empty ? box_extreme<R, true>(in + (size_t)v * H * W, H, W, y, x) : c;
  out[i] = o;
}

// :248-250  valid = (s7 > 0.1) & top_mask ; s7 = valid ? median(s7) : s7   (valid kept for :260)
// grid (blocks per view, views).  Also the block's min / max of the result (cv::minMaxLoc inside bilateralFilter_32f
// needs the per-view extremes): plain stores of per-block partials - contended global atomics cost ~70 ns each here.
__global__ __launch_bounds__(256) void dc_median_top_kernel(const float *__restrict__ in,
                                                            const int32_t *__restrict__ first,
                                                            float *__restrict__ out,
                                                            float *__restrict__ valid, float *__restrict__ mm_part,
                                                            int V, int H, int W) {
  __shared__ float smin[4], smax[4];
  const int v = blockIdx.y, r0 = blockIdx.x * 256 + threadIdx.x;
  const bool in_map = r0 < H * W;
  const int r = in_map ? r0 : H * W - 1;
  const int y = r / W, x = r - y * W;
  const size_t i = (size_t)v * H * W + r;
  const float c = in[i];
  const bool ok = c > 0.1f && y >= first_row(first, v * W + x, H);
  const float o = ok ? median25(in + (size_t)v * H * W, H, W, y, x) : c;
  if (in_map) {
    valid[i] = ok ? 1.f : 0.f;
    out[i] = o;
  }
  float mn = o, mx = o;
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, sft));
    mx = fmaxf(mx, __shfl_xor(mx, sft));
  }
  if ((threadIdx.x & 63) == 0) {
    smin[threadIdx.x >> 6] = mn;
    smax[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float *p = mm_part + ((size_t)v * gridDim.x + blockIdx.x) * 2;
    p[0] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    p[1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  }
}

__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return min(max(i, 0), n - 1);
}

// :259-266  blurred = bilateralFilter(s7, 5, 0.5, 2.0); s7[valid] = blurred[valid]; invert back.
__global__ __launch_bounds__(256) void dc_bilateral_invert_kernel(const float *__restrict__ in,
                                                                  const float *__restrict__ valid,
                                                                  const float *__restrict__ mm_part,
                                                                  float *__restrict__ out, int V,
                                                                  int H, int W) {
  // grid (blocks per view, views); first the view's extremes from the per-block partials of the previous kernel
  __shared__ float s_mm[2];
  const int v = blockIdx.y;
  if (threadIdx.x < 64) {
    float mn_ = INFINITY, mx_ = -INFINITY;
    for (int p = threadIdx.x; p < (int)gridDim.x; p += 64) {
      mn_ = fminf(mn_, mm_part[((size_t)v * gridDim.x + p) * 2]);
      mx_ = fmaxf(mx_, mm_part[((size_t)v * gridDim.x + p) * 2 + 1]);
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
      mn_ = fminf(mn_, __shfl_xor(mn_, sft));
      mx_ = fmaxf(mx_, __shfl_xor(mx_, sft));
    }
    if (threadIdx.x == 0) {
      s_mm[0] = mn_;
      s_mm[1] = mx_;
    }
  }
  __syncthreads();
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= H * W) return;
  const int i = v * H * W + r, y = r / W, x = r - y * W;
  const float *d = in + (size_t)v * H * W;
  const float c = d[r];
  float res = c;
  const float mn = s_mm[0], mx = s_mm[1];
  if (valid[i] != 0.f && !(fabsf(mn - mx) < FLT_EPSILON)) {
    const double gcc = -0.5 / (0.5 * 0.5);   // sigma_color 0.5
    const double gsc = -0.5 / (2.0 * 2.0);   // sigma_space 2.0
    const float scale_index = (float)(4096.0 / ((double)mx - (double)mn));
    float sum = 0.f, wsum = 0.f;
    for (int dy = -2; dy <= 2; ++dy) {
      for (int dx = -2; dx <= 2; ++dx) {
        const int rr = dy * dy + dx * dx;
        if (rr > 4) continue;  // circular support r <= 2
        const float sw = (float)exp((double)rr * gsc);
        const float val = d[reflect101(y + dy, H) * W + reflect101(x + dx, W)];
        float alpha = fabsf(val - c) * scale_index;
        int idx = (int)floorf(alpha);
        alpha -= (float)idx;
        idx = min(max(idx, 0), 4096);
        const double v0 = (double)idx / (double)scale_index, v1 = (double)(idx + 1) / (double)scale_index;
        const float e0 = (float)exp(v0 * v0 * gcc), e1 = (float)exp(v1 * v1 * gcc);
        const float w = sw * (e0 + alpha * (e1 - e0));
        sum += val * w;
        wsum += w;
      }
    }
    res = sum / wsum;
  }
  out[i] = res > 0.1f ? 100.f - res : res;  // :263-266
}

}  // namespace di

extern "C" int di_depth_complete(const float *sparse, float *dense, float *scratch, int32_t *iscratch,
                                 int n_views, int Hi, int Wi, void *stream) {
  using namespace di;
  DI_REQUIRE(n_views > 0 && Hi > 2 && Wi > 2, "bad depth map shape V=%d H=%d W=%d", n_views, Hi, Wi);
  hipStream_t s = (hipStream_t)stream;
  const int V = n_views, H = Hi, W = Wi;
  const int n = V * H * W;
  const dim3 g((n + 255) / 256), b(256);
  float *A = scratch, *B = scratch + (size_t)n, *valid = scratch + 2 * (size_t)n;
  float *mm_part = scratch + 3 * (size_t)n;                  // per-block min / max partials: 2 * V * blocks-per-view
  int32_t *first_a = iscratch, *first_b = iscratch + (size_t)V * W;
  // 9 launches (round 5: the close and the six filling passes are one launch each; rounds 2-4: 15), no global atomics: the per-column "first valid row" by a column scan, the per-view min / max as
  // per-block partials of the kernel that produces the map
  hipLaunchKernelGGL(dc_multiscale_kernel, g, b, 0, s, sparse, A, first_a, first_b, V, H, W);   // s2
  const dim3 gt((W + kTW - 1) / kTW, (H + kTH - 1) / kTH, V);
  hipLaunchKernelGGL(dc_close_kernel, gt, b, 0, s, A, B, V, H, W);                       // close: dilate + erode -> s3
  { float *t = A; A = B; B = t; }
  const dim3 gc((V * W + 63) / 64);
  hipLaunchKernelGGL(dc_median_valid_kernel, g, b, 0, s, A, B, first_a, V, H, W);        // s4
  hipLaunchKernelGGL(dc_col_first_kernel, gc, b, 0, s, B, first_a, V, H, W);             // its top rows
  hipLaunchKernelGGL((dc_fill_kernel<4, false>), g, b, 0, s, B, first_a, A, first_b, V, H, W);      // s5
  hipLaunchKernelGGL(dc_col_first_kernel, gc, b, 0, s, A, first_b, V, H, W);             // its top rows
  float *src = B, *dst = A;
  hipLaunchKernelGGL((dc_fill_iter_kernel<6>), gt, b, 0, s, A, first_b, B, V, H, W);     // s7: the six filling passes
  const dim3 gv((H * W + 255) / 256, V);
  hipLaunchKernelGGL(dc_median_top_kernel, gv, b, 0, s, src, first_b, dst, valid, mm_part, V, H, W);
  hipLaunchKernelGGL(dc_bilateral_invert_kernel, gv, b, 0, s, dst, valid, mm_part, dense, V, H, W);
  return check_launch("depth_complete");
}
