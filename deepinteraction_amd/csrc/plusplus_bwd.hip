// Backward of the DeepInteraction++ samplers (training of row a20).  In the reference these gradients come from
// mmcv's `ms_deform_attn` backward CUDA op and torch autograd of `F.grid_sample` (fusion_transformerv4.py:574, :630);
// the sampling GEOMETRY of the polar attention carries no gradient there either (`apply_3d_transformation(...)
// .detach()`, :560, :595; projections of constant grids), so:
//
//   ms_deform_attn_bwd     d(value) (float32 atomics), d(offsets), d(logits) through the fused softmax
//   grid_gather_bwd        d(feat)  += w_corner * d(out[point])
//   polar_bev_sample_bwd   d(polar) += w_corner / n_seeing * d(out[cell])       (d(bev) = d(out): identity, host side)
//
// Each re-derives locations exactly as its forward kernel (same expressions, same rounding).
#include "di_common.h"

namespace di {
namespace pp {

constexpr int kMaxLevelsB = 4;
struct LevelsB {
  int n;
  int h[kMaxLevelsB], w[kMaxLevelsB], start[kMaxLevelsB];
};

template <int N>
__device__ __forceinline__ void ldvecb(const float *p, float (&f)[N]) {
  const float4 *q = reinterpret_cast<const float4 *>(p);
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    const float4 v = q[i];
    f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
  }
}
template <int N>
__device__ __forceinline__ void ldvecb(const __half *p, float (&f)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) f[i] = __half2float(p[i]);
}
__device__ __forceinline__ void stf(float *p, float v) { *p = v; }
__device__ __forceinline__ void stf(__half *p, float v) { *p = __float2half(v); }

template <typename T>
__device__ __forceinline__ bool corner_ld(const T *__restrict__ map, int H, int W, int sy, int sx, int y, int x,
                                          float (&f)[8]) {
  const bool ok = y >= 0 && y < H && x >= 0 && x < W;
  if (ok) {
    unpack8(ld8(map + (size_t)y * sy + (size_t)x * sx), f);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = 0.f;
  }
  return ok;
}
__device__ __forceinline__ void corner_add(float *__restrict__ gmap, int sy, int sx, int y, int x, float wgt,
                                           const float (&g)[8]) {
  float *p = gmap + (size_t)y * sy + (size_t)x * sx;
#pragma unroll
  for (int i = 0; i < 8; ++i) atomicAdd(p + i, wgt * g[i]);
}
// scatter wgt * g to the (up to) four valid corners of (px, py); zeros padding, texel centres at integers
__device__ __forceinline__ void bilinear_scatter8(float *__restrict__ gmap, int H, int W, int sy, int sx, float px,
                                                  float py, float wgt, const float (&g)[8]) {
  if (!(px > -1.f && px < (float)W && py > -1.f && py < (float)H)) return;
  const float fx = floorf(px), fy = floorf(py);
  const int x0 = (int)fx, y0 = (int)fy;
  const float ax = px - fx, ay = py - fy;
  if (y0 >= 0) {
    if (x0 >= 0) corner_add(gmap, sy, sx, y0, x0, wgt * (1.f - ay) * (1.f - ax), g);
    if (x0 + 1 < W) corner_add(gmap, sy, sx, y0, x0 + 1, wgt * (1.f - ay) * ax, g);
  }
  if (y0 + 1 < H) {
    if (x0 >= 0) corner_add(gmap, sy, sx, y0 + 1, x0, wgt * ay * (1.f - ax), g);
    if (x0 + 1 < W) corner_add(gmap, sy, sx, y0 + 1, x0 + 1, wgt * ay * ax, g);
  }
}

// sum over the two lanes that share a head (adjacent lanes): quad_perm [1,0,3,2]
__device__ __forceinline__ float pair_sum(float x) { return x + dpp<0xB1>(x); }

// lane = (query row, head, half).  gproj rows hold [d offsets (8,L,P,2) | d logits (8,L*P)] like the forward's
// packed projection; gvalue (bs, S, 128) float32, zero-filled by the caller.
template <typename T, int L, int P>
__global__ __launch_bounds__(256) void ms_deform_attn_bwd_kernel(
    const T *__restrict__ value, const T *__restrict__ off, int off_rs, const T *__restrict__ logit, int logit_rs,
    const float *__restrict__ ref, int ref_shared, const T *__restrict__ gout, float *__restrict__ gvalue,
    T *__restrict__ gproj, int gproj_rs, int bs, int nq, int S, LevelsB lv) {
  constexpr int LP = L * P;
  const int l16 = threadIdx.x & 15;
  const int head = l16 >> 1, half = l16 & 1;
  const long long total = (long long)bs * nq;
  // every XCD (private L2) takes one contiguous range of queries: with the default round-robin of consecutive
  // workgroups over the 8 XCDs each L2 ends up fetching the WHOLE value map (measured: 8x the map per launch)
  const long long row = (long long)xcd_remap(blockIdx.x, gridDim.x) * 16 + (threadIdx.x >> 4);
  if (row >= total) return;                         // whole 16-lane groups leave together: DPP pairs stay intact
  const int b = (int)(row / nq), q = (int)(row - (long long)b * nq);
  float w[LP], ofs[LP * 2], dA[LP], dox[LP], doy[LP];
  ldvecb<LP>(logit + (size_t)row * logit_rs + head * LP, w);
  ldvecb<LP * 2>(off + (size_t)row * off_rs + head * LP * 2, ofs);
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < LP; ++i) m = fmaxf(m, w[i]);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) {
    w[i] = __expf(w[i] - m);
    sum += w[i];
  }
  const float inv = 1.f / sum;
  float g[8];
  unpack8(ld8(gout + (size_t)row * 128 + head * 16 + half * 8), g);
  const float *rf = ref + ((size_t)(ref_shared ? 0 : b) * nq + q) * L * 2;
  const size_t vbase = (size_t)b * S * 128 + head * 16 + half * 8;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int H = lv.h[l], W = lv.w[l];
    const T *map = value + vbase + (size_t)lv.start[l] * 128;
    float *gmap = gvalue + vbase + (size_t)lv.start[l] * 128;
    const float rx = rf[l * 2], ry = rf[l * 2 + 1];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = l * P + p;
      const float A = w[i] * inv;
      const float px = (rx + ofs[i * 2] / (float)W) * (float)W - 0.5f;
      const float py = (ry + ofs[i * 2 + 1] / (float)H) * (float)H - 0.5f;
      float da = 0.f, dx = 0.f, dy = 0.f;
      if (px > -1.f && px < (float)W && py > -1.f && py < (float)H) {
        const float fx = floorf(px), fy = floorf(py);
        const int x0 = (int)fx, y0 = (int)fy;
        const float ax = px - fx, ay = py - fy;
        float v00[8], v01[8], v10[8], v11[8];
        const bool k00 = corner_ld(map, H, W, W * 128, 128, y0, x0, v00);
        const bool k01 = corner_ld(map, H, W, W * 128, 128, y0, x0 + 1, v01);
        const bool k10 = corner_ld(map, H, W, W * 128, 128, y0 + 1, x0, v10);
        const bool k11 = corner_ld(map, H, W, W * 128, 128, y0 + 1, x0 + 1, v11);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float top = v00[c] + ax * (v01[c] - v00[c]), bot = v10[c] + ax * (v11[c] - v10[c]);
          da = fmaf(g[c], top + ay * (bot - top), da);
          dx = fmaf(g[c], (1.f - ay) * (v01[c] - v00[c]) + ay * (v11[c] - v10[c]), dx);
          dy = fmaf(g[c], bot - top, dy);
        }
        if (k00) corner_add(gmap, W * 128, 128, y0, x0, A * (1.f - ay) * (1.f - ax), g);
        if (k01) corner_add(gmap, W * 128, 128, y0, x0 + 1, A * (1.f - ay) * ax, g);
        if (k10) corner_add(gmap, W * 128, 128, y0 + 1, x0, A * ay * (1.f - ax), g);
        if (k11) corner_add(gmap, W * 128, 128, y0 + 1, x0 + 1, A * ay * ax, g);
      }
      dA[i] = pair_sum(da);
      dox[i] = pair_sum(dx) * A;                   // d px / d off_x = 1 (pixel units)
      doy[i] = pair_sum(dy) * A;
    }
  }
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) dot = fmaf(w[i] * inv, dA[i], dot);
  if (half == 0) {
    T *go = gproj + (size_t)row * gproj_rs + head * LP * 2;
    T *gl = gproj + (size_t)row * gproj_rs + 8 * LP * 2 + head * LP;
#pragma unroll
    for (int i = 0; i < LP; ++i) {
      stf(go + 2 * i, dox[i]);
      stf(go + 2 * i + 1, doy[i]);
      stf(gl + i, w[i] * inv * (dA[i] - dot));
    }
  }
}

// gfeat (Bf,H,W,C) float32 zero-filled; gout (Bg,N,C)
template <typename T>
__global__ __launch_bounds__(256) void grid_gather_bwd_kernel(const float *__restrict__ grid,
                                                              const T *__restrict__ gout, float *__restrict__ gfeat,
                                                              int Bg, int N, int per_feat, int H, int W, int C) {
  const int l16 = threadIdx.x & 15, ch0 = l16 * kChPerLane;
  const long long total = (long long)Bg * N;
  const long long idx = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (idx >= total || ch0 >= C) return;
  const int gi = (int)(idx / N);
  const float gx = grid[idx * 2], gy = grid[idx * 2 + 1];
  const float px = ((gx + 1.f) * (float)W - 1.f) * 0.5f, py = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
  float g[8];
  unpack8(ld8(gout + (size_t)idx * C + ch0), g);
  bilinear_scatter8(gfeat + (size_t)(gi / per_feat) * H * W * C + ch0, H, W, W * C, C, px, py, 1.f, g);
}

// gpolar (B,V,Wp,R,C) float32 zero-filled (ray-major, as the forward reads it); gout (B,Hb,Wb,C)
template <typename T>
__global__ __launch_bounds__(256) void polar_bev_sample_bwd_kernel(const T *__restrict__ gout,
                                                                   const float *__restrict__ proj,
                                                                   const float *__restrict__ aug_rev,
                                                                   const float *__restrict__ cam_xy,
                                                                   const float *__restrict__ par,
                                                                   float *__restrict__ gpolar, int B, int V, int R,
                                                                   int Wp, int Hb, int Wb, int C) {
  constexpr int ZS = 10, VMAX = 8;
  const int l16 = threadIdx.x & 15, ch0 = l16 * kChPerLane;
  const long long total = (long long)B * Hb * Wb;
  const long long cell = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (cell >= total || ch0 >= C) return;
  const int b = (int)(cell / (Hb * Wb)), ij = (int)(cell - (long long)b * Hb * Wb);
  const int i = ij / Wb, j = ij - i * Wb;
  const float x0 = par[0], y0 = par[1], z0 = par[2], x1 = par[3], y1 = par[4], z1 = par[5];
  const float in_h = par[6], in_w = par[7], r0 = par[8], Rf = par[9];
  const float bx = ((float)j + 0.5f) / (float)Hb * (x1 - x0) + x0;
  const float by = ((float)i + 0.5f) / (float)Wb * (y1 - y0) + y0;
  const float *A = aug_rev + b * 12;
  float fxs[VMAX], fys[VMAX];
  bool anys[VMAX];
  int vis = 0;
#pragma unroll
  for (int v = 0; v < VMAX; ++v) {
    anys[v] = false;
    fxs[v] = fys[v] = 0.f;
    if (v >= V) continue;
    const float *M = proj + ((size_t)b * V + v) * 16;
    const float cx = cam_xy[((size_t)b * V + v) * 2], cy = cam_xy[((size_t)b * V + v) * 2 + 1];
    float su = 0.f, sr = 0.f;
    bool any = false;
#pragma unroll
    for (int k = 0; k < ZS; ++k) {
      const float bz = ((float)k + 0.5f) / (float)ZS * (z1 - z0) + z0;
      const float px = bx * A[0] + by * A[3] + bz * A[6] + A[9];
      const float py = bx * A[1] + by * A[4] + bz * A[7] + A[10];
      const float pz = bx * A[2] + by * A[5] + bz * A[8] + A[11];
      const float xc = M[0] * px + M[1] * py + M[2] * pz + M[3];
      const float yc = M[4] * px + M[5] * py + M[6] * pz + M[7];
      const float zc = M[8] * px + M[9] * py + M[10] * pz + M[11];
      const float zd = fmaxf(zc, 1e-5f);
      const float u = 2.f * (xc / zd / in_w) - 1.f, vv = 2.f * (yc / zd / in_h) - 1.f;
      any |= (zc > 1e-5f) && u > -1.f && u < 1.f && vv > -1.f && vv < 1.f;
      su += u;
      const float dx = px - cx, dy = py - cy;
      sr += fminf(fmaxf(2.f * (sqrtf(dx * dx + dy * dy) - r0) / Rf - 1.f, -1.f), 1.f);
    }
    anys[v] = any;
    vis += any ? 1 : 0;
    fxs[v] = ((su / (float)ZS + 1.f) * (float)Wp - 1.f) * 0.5f;
    fys[v] = ((sr / (float)ZS + 1.f) * (float)R - 1.f) * 0.5f;
  }
  if (vis == 0) return;
  float g[8];
  unpack8(ld8(gout + (size_t)cell * C + ch0), g);
  const float inv = 1.f / (float)vis;
#pragma unroll
  for (int v = 0; v < VMAX; ++v) {
    if (v >= V || !anys[v]) continue;
    bilinear_scatter8(gpolar + ((size_t)b * V + v) * R * Wp * C + ch0, R, Wp, C, R * C, fxs[v], fys[v], inv, g);
  }
}

}  // namespace pp
}  // namespace di

extern "C" {

int di_ms_deform_attn_bwd(const void *value, const void *offsets, int off_row_stride, const void *logits,
                          int logit_row_stride, const float *ref, int ref_shared, const void *grad_out,
                          float *grad_value, void *grad_proj, int grad_proj_row_stride, int bs, int nq, int n_levels,
                          int n_points, const int32_t *level_hw, int dtype, void *stream) {
  DI_REQUIRE(bs > 0 && nq > 0, "bad deformable attention shape");
  DI_REQUIRE((n_levels == 1 || n_levels == 2) && n_points == 4, "levels %d / points %d unsupported (1|2 levels, 4 points)",
             n_levels, n_points);
  DI_REQUIRE(grad_proj_row_stride >= 8 * n_levels * n_points * 3, "grad_proj rows too short");
  di::pp::LevelsB lv;
  lv.n = n_levels;
  int S = 0;
  for (int l = 0; l < n_levels; ++l) {
    lv.h[l] = level_hw[2 * l];
    lv.w[l] = level_hw[2 * l + 1];
    DI_REQUIRE(lv.h[l] > 0 && lv.w[l] > 0, "bad level shape");
    lv.start[l] = S;
    S += lv.h[l] * lv.w[l];
  }
  const long long rows = (long long)bs * nq;
  const dim3 grid((unsigned)((rows + 15) / 16)), blk(256);
  hipStream_t s = (hipStream_t)stream;
#define DI_MSDB(TT, LL)                                                                                              \
  hipLaunchKernelGGL((di::pp::ms_deform_attn_bwd_kernel<TT, LL, 4>), grid, blk, 0, s, (const TT *)value,             \
                     (const TT *)offsets, off_row_stride, (const TT *)logits, logit_row_stride, ref, ref_shared,     \
                     (const TT *)grad_out, grad_value, (TT *)grad_proj, grad_proj_row_stride, bs, nq, S, lv)
  if (dtype == DI_F16) { if (n_levels == 1) DI_MSDB(__half, 1); else DI_MSDB(__half, 2); }
  else if (dtype == DI_F32) { if (n_levels == 1) DI_MSDB(float, 1); else DI_MSDB(float, 2); }
  else { di::set_error("unsupported dtype %d", dtype); return DI_ERR_ARG; }
#undef DI_MSDB
  return di::check_launch("ms_deform_attn_bwd");
}

int di_grid_gather_bwd(const float *grid, const void *grad_out, float *grad_feat, int n_grids, int n_points,
                       int grids_per_feat, int H, int W, int C, int dtype, void *stream) {
  DI_REQUIRE(n_grids > 0 && n_points > 0 && grids_per_feat > 0 && H > 0 && W > 0, "bad grid gather shape");
  DI_REQUIRE(C > 0 && C % 8 == 0 && C <= 128, "C=%d must be a multiple of 8, <= 128", C);
  const long long total = (long long)n_grids * n_points;
  const dim3 g((unsigned)((total + 15) / 16)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DI_F16)
    hipLaunchKernelGGL((di::pp::grid_gather_bwd_kernel<__half>), g, blk, 0, s, grid, (const __half *)grad_out,
                       grad_feat, n_grids, n_points, grids_per_feat, H, W, C);
  else if (dtype == DI_F32)
    hipLaunchKernelGGL((di::pp::grid_gather_bwd_kernel<float>), g, blk, 0, s, grid, (const float *)grad_out,
                       grad_feat, n_grids, n_points, grids_per_feat, H, W, C);
  else { di::set_error("unsupported dtype %d", dtype); return DI_ERR_ARG; }
  return di::check_launch("grid_gather_bwd");
}

int di_polar_bev_sample_bwd(const void *grad_out, const float *proj, const float *aug_rev, const float *cam_xy,
                            const float *params, float *grad_polar, int B, int V, int R, int Wp, int Hb, int Wb,
                            int C, int dtype, void *stream) {
  DI_REQUIRE(B > 0 && V > 0 && V <= 8 && R > 0 && Wp > 0 && Hb > 0 && Wb > 0, "bad polar sample shape");
  DI_REQUIRE(Hb == Wb, "square BEV maps only");
  DI_REQUIRE(C > 0 && C % 8 == 0 && C <= 128, "C=%d must be a multiple of 8, <= 128", C);
  const long long total = (long long)B * Hb * Wb;
  const dim3 g((unsigned)((total + 15) / 16)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DI_F16)
    hipLaunchKernelGGL((di::pp::polar_bev_sample_bwd_kernel<__half>), g, blk, 0, s, (const __half *)grad_out, proj,
                       aug_rev, cam_xy, params, grad_polar, B, V, R, Wp, Hb, Wb, C);
  else if (dtype == DI_F32)
    hipLaunchKernelGGL((di::pp::polar_bev_sample_bwd_kernel<float>), g, blk, 0, s, (const float *)grad_out, proj,
                       aug_rev, cam_xy, params, grad_polar, B, V, R, Wp, Hb, Wb, C);
  else { di::set_error("unsupported dtype %d", dtype); return DI_ERR_ARG; }
  return di::check_launch("polar_bev_sample_bwd");
}

}  // extern "C"
