// Geometry shared by the BEV -> image gather (cross_modal.hip) and its fused form inside the 1x1 projection kernel
// (pointwise.hip): the augmentation affine, torch's bilinear grid_sample of one 8-channel slice, and the un-projection of a
// feature pixel through its completed depth (reference encoder_utils.py:185-196).
#pragma once
#include "di_common.h"

namespace di {

struct Affine {  // p' = p @ A + t   (row-vector convention of mmdet3d's LiDARPoints.rotate)
  float a[9], t[3];
};
__device__ __forceinline__ void apply_affine(const Affine &f, float &x, float &y, float &z) {
  const float nx = x * f.a[0] + y * f.a[3] + z * f.a[6] + f.t[0];
  const float ny = x * f.a[1] + y * f.a[4] + z * f.a[7] + f.t[1];
  const float nz = x * f.a[2] + y * f.a[5] + z * f.a[8] + f.t[2];
  x = nx; y = ny; z = nz;
}
__device__ __forceinline__ Affine load_affine(const float *p) {
  Affine f;
#pragma unroll
  for (int i = 0; i < 9; ++i) f.a[i] = p[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) f.t[i] = p[9 + i];
  return f;
}

// torch grid_sample(bilinear, zeros, align_corners=False) of one texel row slice (8 channels).
template <typename T>
__device__ __forceinline__ void bilinear8(const T *__restrict__ map, int Hm, int Wm, int C, float ix,
                                          float iy, int ch0, float (&o)[8]) {
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float ax = ix - fx, ay = iy - fy;
  const float w00 = (1.f - ax) * (1.f - ay), w01 = ax * (1.f - ay), w10 = (1.f - ax) * ay, w11 = ax * ay;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = 0.f;
  const bool xl = x0 >= 0 && x0 < Wm, xh = x0 + 1 >= 0 && x0 + 1 < Wm;
  const bool yl = y0 >= 0 && y0 < Hm, yh = y0 + 1 >= 0 && y0 + 1 < Hm;
  float f[8];
  if (yl && xl) {
    unpack8(ld8(map + ((size_t)y0 * Wm + x0) * C + ch0), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaf(w00, f[i], o[i]);
  }
  if (yl && xh) {
    unpack8(ld8(map + ((size_t)y0 * Wm + x0 + 1) * C + ch0), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaf(w01, f[i], o[i]);
  }
  if (yh && xl) {
    unpack8(ld8(map + ((size_t)(y0 + 1) * Wm + x0) * C + ch0), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaf(w10, f[i], o[i]);
  }
  if (yh && xh) {
    unpack8(ld8(map + ((size_t)(y0 + 1) * Wm + x0 + 1) * C + ch0), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaf(w11, f[i], o[i]);
  }
}

// The same sample with every load UNCONDITIONAL (corner addresses clamped into the map, the weights of corners outside it -
// and of a masked pixel, `live` false - set to zero): no branch between the loads, so the 4 x 4 corner rows of a pixel group
// fly together.  Adding w * f with w = +0 leaves the running sum bit for bit, and the order of the four terms is bilinear8's:
// identical results.
struct Bilin4 {
  int o00, o01, o10, o11;        // texel indices (y * Wm + x) of the four corners, clamped
  float w00, w01, w10, w11;
};
__device__ __forceinline__ Bilin4 bilinear_setup(int Hm, int Wm, float ix, float iy, bool live) {
  const float fx = floorf(ix), fy = floorf(iy);
  const float ax = ix - fx, ay = iy - fy;
  // a masked pixel may carry any coordinates (inf / nan): keep the integer conversions defined
  const int x0 = live ? (int)fminf(fmaxf(fx, -2.f), (float)Wm + 1.f) : 0, y0 = live ? (int)fminf(fmaxf(fy, -2.f), (float)Hm + 1.f) : 0;
  const bool xl = live && x0 >= 0 && x0 < Wm, xh = live && x0 + 1 >= 0 && x0 + 1 < Wm;
  const bool yl = y0 >= 0 && y0 < Hm, yh = y0 + 1 >= 0 && y0 + 1 < Hm;
  const int cx0 = min(max(x0, 0), Wm - 1), cx1 = min(max(x0 + 1, 0), Wm - 1);
  const int cy0 = min(max(y0, 0), Hm - 1), cy1 = min(max(y0 + 1, 0), Hm - 1);
  Bilin4 b;
  b.o00 = cy0 * Wm + cx0; b.o01 = cy0 * Wm + cx1; b.o10 = cy1 * Wm + cx0; b.o11 = cy1 * Wm + cx1;
  b.w00 = (yl && xl) ? (1.f - ax) * (1.f - ay) : 0.f;
  b.w01 = (yl && xh) ? ax * (1.f - ay) : 0.f;
  b.w10 = (yh && xl) ? (1.f - ax) * ay : 0.f;
  b.w11 = (yh && xh) ? ax * ay : 0.f;
  return b;
}
template <typename T>
__device__ __forceinline__ void bilinear8_nb(const T *__restrict__ map, int C, const Bilin4 &b, int ch0, float (&o)[8]) {
  const Pack8<T> p00 = ld8(map + (size_t)b.o00 * C + ch0), p01 = ld8(map + (size_t)b.o01 * C + ch0);
  const Pack8<T> p10 = ld8(map + (size_t)b.o10 * C + ch0), p11 = ld8(map + (size_t)b.o11 * C + ch0);
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = 0.f;
  unpack8(p00, f);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = fmaf(b.w00, f[i], o[i]);
  unpack8(p01, f);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = fmaf(b.w01, f[i], o[i]);
  unpack8(p10, f);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = fmaf(b.w10, f[i], o[i]);
  unpack8(p11, f);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = fmaf(b.w11, f[i], o[i]);
}

// Where feature pixel `pix` = (view v, row yy, column xx) of the (V, Hi, Wi) image maps samples the BEV map: un-project through
// the completed depth, re-apply the augmentation, strict range test, normalised grid -> texel coordinates
// (encoder_utils.py:185-196).  False: the pixel lifts outside the point-cloud range and reads zeros.
struct WarpGeom {
  const float *depth, *img2lidar, *xs, *ys;
  Affine A;
  float r0, r1, r2, r3, r4, r5;
  int Hi, Wi, Hb, Wb;
};
__device__ __forceinline__ WarpGeom load_warp_geom(const float *depth, const float *img2lidar, const float *aug, const float *xs,
                                                   const float *ys, const float *pc_range, int Hi, int Wi, int Hb, int Wb) {
  WarpGeom g;
  g.depth = depth; g.img2lidar = img2lidar; g.xs = xs; g.ys = ys;
  g.A = load_affine(aug);
  g.r0 = pc_range[0]; g.r1 = pc_range[1]; g.r2 = pc_range[2];
  g.r3 = pc_range[3]; g.r4 = pc_range[4]; g.r5 = pc_range[5];
  g.Hi = Hi; g.Wi = Wi; g.Hb = Hb; g.Wb = Wb;
  return g;
}
__device__ __forceinline__ bool warp_position(const WarpGeom &g, int pix, float &ix, float &iy) {
  const int v = pix / (g.Hi * g.Wi);
  const int rem = pix - v * g.Hi * g.Wi;
  const int yy = rem / g.Wi, xx = rem - yy * g.Wi;
  const float d = g.depth[pix];
  const float X = g.xs[xx] * d, Y = g.ys[yy] * d;  // [x*d, y*d, d, 1] (:185-187)
  const float *M = g.img2lidar + v * 16;
  float x = M[0] * X + M[1] * Y + M[2] * d + M[3];
  float y = M[4] * X + M[5] * Y + M[6] * d + M[7];
  float z = M[8] * X + M[9] * Y + M[10] * d + M[11];
  apply_affine(g.A, x, y, z);  // re-apply the augmentation (:189)
  const bool lift = x > g.r0 && y > g.r1 && z > g.r2 && x < g.r3 && y < g.r4 && z < g.r5;  // strict (:191-192)
  const float gx = ((x - g.r0) / (g.r3 - g.r0) - 0.5f) * 2.f;  // x -> BEV width (:193-194)
  const float gy = ((y - g.r1) / (g.r4 - g.r1) - 0.5f) * 2.f;  // y -> BEV height
  ix = ((gx + 1.f) * g.Wb - 1.f) * 0.5f;
  iy = ((gy + 1.f) * g.Hb - 1.f) * 0.5f;
  return lift;
}

}  // namespace di
