// Feature gradients of the three gather operators of the hot path (the training
// step).  In the reference these come from torch / detectron2 autograd of `F.grid_sample`
// (encoder_utils.py:195, :297) and `ROIAlignV2` (decoder_utils.py:739-741, 822-823); sampling coordinates are
// functions of points, metas and box predictions that carry no gradient in the reference either
// (`res_layer` is detached through bbox_coder.decode on deep copies, decoder_utils.py:672-679), so only the
// FEATURE gradients exist:
//
//   bevwarp_gather_bwd   d(bev)   += w_corner * d(warped[pixel])             (4 bilinear corners)
//   roi_align_bwd        d(feat)  += w_corner / 4 * d(out[roi, bin])         (2x2 samples x 4 corners)
//   i2p_attn_bwd         d(img), d(qfold) of  ctx = sum_j softmax_j(<qfold, s_j>) s_j,  s_j = bilinear(img, uv_j)
//
// All three re-derive the geometry exactly as their forward kernels do (same code, same rounding) and
// accumulate with float32 atomics into float32 gradient maps (the caller casts); 16 lanes own one texel
// (8 channels each), as in the forward kernels.
#include <limits.h>
#include <stdlib.h>

#include "di_common.h"

namespace di {

struct AffineB {
  float a[9], t[3];
};
__device__ __forceinline__ void apply_affine_b(const AffineB &f, float &x, float &y, float &z) {
  const float nx = x * f.a[0] + y * f.a[3] + z * f.a[6] + f.t[0];
  const float ny = x * f.a[1] + y * f.a[4] + z * f.a[7] + f.t[1];
  const float nz = x * f.a[2] + y * f.a[5] + z * f.a[8] + f.t[2];
  x = nx; y = ny; z = nz;
}
__device__ __forceinline__ AffineB load_affine_b(const float *p) {
  AffineB f;
#pragma unroll
  for (int i = 0; i < 9; ++i) f.a[i] = p[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) f.t[i] = p[9 + i];
  return f;
}

// the same projection as cross_modal.hip::project_point (encoder_utils.py:157-170 / :281-295)
__device__ __forceinline__ bool project_point_b(const float *__restrict__ M, float x, float y, float z,
                                                float ori_H, float ori_W, float &nx, float &ny) {
  const float cx = M[0] * x + M[1] * y + M[2] * z + M[3];
  const float cy = M[4] * x + M[5] * y + M[6] * z + M[7];
  const float cz = M[8] * x + M[9] * y + M[10] * z + M[11];
  const float eps = 1e-5f;
  const float den = fmaxf(cz, eps);
  const float u = cx / den, v = cy / den;
  nx = (u / ori_W - 0.5f) * 2.f;
  ny = (v / ori_H - 0.5f) * 2.f;
  return cz > eps && nx > -1.f && nx < 1.f && ny > -1.f && ny < 1.f;
}

struct Bilin {
  int x0, y0;
  float w00, w01, w10, w11;
  bool v00, v01, v10, v11;
};
// torch grid_sample(bilinear, zeros, align_corners=False) corner set of an un-normalised coordinate
__device__ __forceinline__ Bilin bilin_setup(float ix, float iy, int Hm, int Wm) {
  Bilin b;
  const float fx = floorf(ix), fy = floorf(iy);
  b.x0 = (int)fx;
  b.y0 = (int)fy;
  const float ax = ix - fx, ay = iy - fy;
  b.w00 = (1.f - ax) * (1.f - ay);
  b.w01 = ax * (1.f - ay);
  b.w10 = (1.f - ax) * ay;
  b.w11 = ax * ay;
  const bool xl = b.x0 >= 0 && b.x0 < Wm, xh = b.x0 + 1 >= 0 && b.x0 + 1 < Wm;
  const bool yl = b.y0 >= 0 && b.y0 < Hm, yh = b.y0 + 1 >= 0 && b.y0 + 1 < Hm;
  b.v00 = yl && xl;
  b.v01 = yl && xh;
  b.v10 = yh && xl;
  b.v11 = yh && xh;
  return b;
}
template <typename T>
__device__ __forceinline__ void bilin_gather8(const T *__restrict__ map, int Wm, int C, const Bilin &b, int ch0,
                                              float (&o)[8]) {
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = 0.f;
  if (b.v00) {
    unpack8(ld8(map + ((size_t)b.y0 * Wm + b.x0) * C + ch0), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaf(b.w00, f[i], o[i]);
  }
  if (b.v01) {
    unpack8(ld8(map + ((size_t)b.y0 * Wm + b.x0 + 1) * C + ch0), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaf(b.w01, f[i], o[i]);
  }
  if (b.v10) {
    unpack8(ld8(map + ((size_t)(b.y0 + 1) * Wm + b.x0) * C + ch0), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaf(b.w10, f[i], o[i]);
  }
  if (b.v11) {
    unpack8(ld8(map + ((size_t)(b.y0 + 1) * Wm + b.x0 + 1) * C + ch0), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaf(b.w11, f[i], o[i]);
  }
}
// Float32 atomics want 64 CONSECUTIVE bytes per 16-lane instruction: with the forward's layout (lane l = channels 8 l ..
// 8 l + 7, the lanes of one instruction 32 B apart) the L2 atomic units retire 37 G adds / s, with lane l = channels l,
// l + 16, ... (each instruction covers one contiguous 64-B piece) 329 G / s (tools/micro/atomic_pattern.hip).  The gradient
// rows are therefore transposed inside their 16-lane group through a 512-B LDS slot before they are scattered:
// in g[e] = channel 8 l16 + e, out t[k] = channel l16 + 16 k.  (DS operations of one wave are served in issue order, so the
// reads see every lane's writes; the wave barrier only keeps the compiler from reordering them.)
constexpr int kSlotFloats = 128;
__device__ __forceinline__ void to_lane_major(const float (&g)[8], float (&t)[8], float *slot, int l16) {
  *reinterpret_cast<float4 *>(slot + 8 * l16) = make_float4(g[0], g[1], g[2], g[3]);
  *reinterpret_cast<float4 *>(slot + 8 * l16 + 4) = make_float4(g[4], g[5], g[6], g[7]);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < 8; ++k) t[k] = slot[l16 + 16 * k];
  __builtin_amdgcn_wave_barrier();
}
// row += w * t for the channels l16 + 16 k < C of one texel row
__device__ __forceinline__ void add_row(float *__restrict__ row, int C, int l16, float w, const float (&t)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (l16 + 16 * k < C) atomicAdd(row + l16 + 16 * k, w * t[k]);
}
__device__ __forceinline__ void scatter_rows(float *__restrict__ gmap, int Wm, int C, const Bilin &b, int l16,
                                             const float (&t)[8]) {
  if (b.v00) add_row(gmap + ((size_t)b.y0 * Wm + b.x0) * C, C, l16, b.w00, t);
  if (b.v01) add_row(gmap + ((size_t)b.y0 * Wm + b.x0 + 1) * C, C, l16, b.w01, t);
  if (b.v10) add_row(gmap + ((size_t)(b.y0 + 1) * Wm + b.x0) * C, C, l16, b.w10, t);
  if (b.v11) add_row(gmap + ((size_t)(b.y0 + 1) * Wm + b.x0 + 1) * C, C, l16, b.w11, t);
}

// ---------------------------------------------------------------- BEV -> image gather, backward
// One 16-lane group walks a RUN of consecutive pixels of one image row and keeps the four corner sums of
// the BEV cell it is currently hitting in registers, flushing them with atomics only when the cell
// changes: neighbouring pixels un-project to the same 0.6 m cell most of the time, and every pixel without
// depth un-projects to the camera centre - tens of thousands of texels on ONE cell (15 ms of serialised
// atomics per launch when every pixel adds on its own).
constexpr int kRunB = 16;

template <typename T>
__global__ __launch_bounds__(256) void bevwarp_gather_bwd_kernel(
    const T *__restrict__ grad_out, const float *__restrict__ depth, const float *__restrict__ img2lidar,
    const float *__restrict__ aug, const float *__restrict__ xs, const float *__restrict__ ys,
    const float *__restrict__ pc_range, float *__restrict__ grad_bev, int V, int Hi, int Wi, int Hb, int Wb,
    int C, int dbg) {
  __shared__ float slots[16 * kSlotFloats];            // one transposition slot per 16-lane group
  const int l16 = threadIdx.x & 15;
  float *slot = slots + (threadIdx.x >> 4) * kSlotFloats;
  const bool ch_ok = l16 * kChPerLane < C;
  const int ch0 = l16 * kChPerLane;
  const AffineB A = load_affine_b(aug);
  const float r0 = pc_range[0], r1 = pc_range[1], r2 = pc_range[2];
  const float r3 = pc_range[3], r4 = pc_range[4], r5 = pc_range[5];
  const int runs_per_row = (Wi + kRunB - 1) / kRunB;
  const int total = V * Hi * runs_per_row;
  // What a group still holds at the END of its run is merged over the workgroup's 16 groups (16 consecutive runs = 256
  // consecutive pixels) before it goes out: the cells every depth-less pixel lands on (the camera centre: half of the pixels
  // of a synthetic sample, whole image rows) then receive one set of atomics per workgroup round instead of one per run -
  // 502 -> ~60 us per launch (26 us without any atomics).
  __shared__ float mbuf[16][4][128];
  __shared__ int mid[16][2];
  const int grp = threadIdx.x >> 4;
  const int ngrp = gridDim.x * 16;
  for (int base = blockIdx.x * 16; base < total; base += ngrp) {
    const int run = min(base + grp, total - 1);
    const bool run_ok = base + grp < total;
    const int row = run / runs_per_row, x_beg = (run - row * runs_per_row) * kRunB;
    const int v = row / Hi, yy = row - v * Hi;
    const float *M = img2lidar + v * 16;
    int cx0 = INT_MIN, cy0 = INT_MIN;
    float a00[8], a01[8], a10[8], a11[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a00[i] = a01[i] = a10[i] = a11[i] = 0.f;
    auto flush = [&]() {
      if (cx0 == INT_MIN || (dbg & 1)) return;          // (uniform over the 16-lane group; dbg & 1: measurement, no atomics)
      const bool xl = cx0 >= 0 && cx0 < Wb, xh = cx0 + 1 >= 0 && cx0 + 1 < Wb;
      const bool yl = cy0 >= 0 && cy0 < Hb, yh = cy0 + 1 >= 0 && cy0 + 1 < Hb;
      float *p = grad_bev + ((long long)cy0 * Wb + cx0) * C;
      float t[8];
      to_lane_major(a00, t, slot, l16);
      if (yl && xl) add_row(p, C, l16, 1.f, t);
      to_lane_major(a01, t, slot, l16);
      if (yl && xh) add_row(p + C, C, l16, 1.f, t);
      to_lane_major(a10, t, slot, l16);
      if (yh && xl) add_row(p + (long long)Wb * C, C, l16, 1.f, t);
      to_lane_major(a11, t, slot, l16);
      if (yh && xh) add_row(p + (long long)Wb * C + C, C, l16, 1.f, t);
#pragma unroll
      for (int i = 0; i < 8; ++i) a00[i] = a01[i] = a10[i] = a11[i] = 0.f;
    };
    for (int xx = x_beg; xx < (run_ok ? min(x_beg + kRunB, Wi) : x_beg); ++xx) {
      const int pix = (v * Hi + yy) * Wi + xx;
      const float d = depth[pix];
      const float X = xs[xx] * d, Y = ys[yy] * d;
      float x = M[0] * X + M[1] * Y + M[2] * d + M[3];
      float y = M[4] * X + M[5] * Y + M[6] * d + M[7];
      float z = M[8] * X + M[9] * Y + M[10] * d + M[11];
      apply_affine_b(A, x, y, z);
      const bool lift = x > r0 && y > r1 && z > r2 && x < r3 && y < r4 && z < r5;
      if (!lift) continue;
      const float gx = ((x - r0) / (r3 - r0) - 0.5f) * 2.f;
      const float gy = ((y - r1) / (r4 - r1) - 0.5f) * 2.f;
      const float ix = ((gx + 1.f) * Wb - 1.f) * 0.5f;
      const float iy = ((gy + 1.f) * Hb - 1.f) * 0.5f;
      const Bilin b = bilin_setup(ix, iy, Hb, Wb);
      if (b.x0 != cx0 || b.y0 != cy0) {
        flush();
        cx0 = b.x0;
        cy0 = b.y0;
      }
      float g[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) g[i] = 0.f;
      if (ch_ok && !(dbg & 2)) unpack8(ld8(grad_out + (size_t)pix * C + ch0), g);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a00[i] = fmaf(b.w00, g[i], a00[i]);
        a01[i] = fmaf(b.w01, g[i], a01[i]);
        a10[i] = fmaf(b.w10, g[i], a10[i]);
        a11[i] = fmaf(b.w11, g[i], a11[i]);
      }
    }
    // ---- merge over the workgroup: the lowest group holding a cell collects what the others hold for it
    if (l16 == 0) {
      mid[grp][0] = cx0;
      mid[grp][1] = cy0;
    }
    if (ch_ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        mbuf[grp][0][ch0 + i] = a00[i];
        mbuf[grp][1][ch0 + i] = a01[i];
        mbuf[grp][2][ch0 + i] = a10[i];
        mbuf[grp][3][ch0 + i] = a11[i];
      }
    }
    __syncthreads();
    bool leader = cx0 != INT_MIN;
    for (int g2 = 0; g2 < grp; ++g2) leader = leader && !(mid[g2][0] == cx0 && mid[g2][1] == cy0);
    if (leader) {
      for (int g2 = grp + 1; g2 < 16; ++g2)
        if (mid[g2][0] == cx0 && mid[g2][1] == cy0 && ch_ok) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            a00[i] += mbuf[g2][0][ch0 + i];
            a01[i] += mbuf[g2][1][ch0 + i];
            a10[i] += mbuf[g2][2][ch0 + i];
            a11[i] += mbuf[g2][3][ch0 + i];
          }
        }
      flush();
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- RoIAlign, backward
__device__ __forceinline__ void roi_scatter_rows(float *__restrict__ gmap, int H, int W, int C, float y, float x,
                                                 int l16, const float (&t)[8], float wgt) {
  if (y < -1.f || y > (float)H || x < -1.f || x > (float)W) return;
  y = fmaxf(y, 0.f);
  x = fmaxf(x, 0.f);
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
  const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
  add_row(gmap + ((size_t)yl * W + xl) * C, C, l16, hy * hx * wgt, t);
  add_row(gmap + ((size_t)yl * W + xh) * C, C, l16, hy * lx * wgt, t);
  add_row(gmap + ((size_t)yh * W + xl) * C, C, l16, ly * hx * wgt, t);
  add_row(gmap + ((size_t)yh * W + xh) * C, C, l16, ly * lx * wgt, t);
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const T *__restrict__ grad_out,
                                                            const float *__restrict__ rois,
                                                            float *__restrict__ grad_feat, int R, int N, int H,
                                                            int W, int C, float scale) {
  constexpr int PB = 7, G = 2;
  __shared__ float slots[16 * kSlotFloats];            // one transposition slot per 16-lane group
  const int l16 = threadIdx.x & 15;
  float *slot = slots + (threadIdx.x >> 4) * kSlotFloats;
  const bool ch_ok = l16 * kChPerLane < C;
  const int ch0 = l16 * kChPerLane;
  const int total = R * PB * PB;
  const int ngrp = gridDim.x * (blockDim.x >> 4);
  for (int g = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4); g < total; g += ngrp) {
    const int r = g / (PB * PB), bin = g - r * PB * PB;
    const int ph = bin / PB, pw = bin - ph * PB;
    const float *roi = rois + (size_t)r * 5;
    int n = (int)roi[0];
    n = min(max(n, 0), N - 1);
    const float sw = roi[1] * scale - 0.5f, sh = roi[2] * scale - 0.5f;
    const float ew = roi[3] * scale - 0.5f, eh = roi[4] * scale - 0.5f;
    const float bw = (ew - sw) / PB, bh = (eh - sh) / PB;
    float go[8], gt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) go[i] = 0.f;
    if (ch_ok) unpack8(ld8(grad_out + (size_t)g * C + ch0), go);
    to_lane_major(go, gt, slot, l16);
    float *gmap = grad_feat + (size_t)n * H * W * C;
#pragma unroll
    for (int iy = 0; iy < G; ++iy) {
      const float y = sh + ph * bh + (iy + 0.5f) * bh / G;
#pragma unroll
      for (int ix = 0; ix < G; ++ix) {
        const float x = sw + pw * bw + (ix + 0.5f) * bw / G;
        roi_scatter_rows(gmap, H, W, C, y, x, l16, gt, 1.f / (G * G));
      }
    }
  }
}

// ---------------------------------------------------------------- image -> BEV pillar attention, backward
// One wavefront per pillar, as the forward.  Pass A re-runs the forward of the pillar (valid-key list,
// running max / sum, context) to get (m, l, ctx); pass B walks the keys again:
//   p_j = exp(score_j - m) / l,   dscore_j = p_j (<g, s_j> - <g, ctx>),
//   d(s_j) = p_j g + dscore_j qfold   -> scattered into d(img) through the bilinear corners,
//   d(qfold) = sum_j dscore_j s_j     -> one 256 B row per pillar.
constexpr int kMaxSlotsB = 128;
struct KeyEntB {
  float ix, iy;
  int cam;
};

template <typename T>
__global__ __launch_bounds__(256) void i2p_attn_bwd_kernel(
    const T *__restrict__ img, const T *__restrict__ qfold, const T *__restrict__ grad_ctx, const T *__restrict__ grad_mass,
    const float *__restrict__ pillars, const int32_t *__restrict__ coors, const int32_t *__restrict__ num_points,
    const float *__restrict__ proj, const float *__restrict__ aug, float *__restrict__ grad_img,
    float *__restrict__ grad_qfold, int P, int Tp, int D, int V, int Hi, int Wi, int Hb, int Wb, int C,
    float ori_H, float ori_W, float drop_p, unsigned long long seed, const unsigned long long *__restrict__ seed_add) {
  if (seed_add != nullptr) seed += *seed_add;
  __shared__ KeyEntB s_list[4][kMaxSlotsB];
  __shared__ float slots[16 * kSlotFloats];            // one transposition slot per 16-lane group
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  const int l16 = lane & 15, sub = lane >> 4;
  float *slot = slots + (tid >> 4) * kSlotFloats;
  const bool ch_ok = l16 * kChPerLane < C;
  const int ch0 = l16 * kChPerLane;
  KeyEntB *list = s_list[wib];
  const AffineB A = load_affine_b(aug);
  const int nslots = Tp * V;
  const int nwaves = gridDim.x * 4;
  for (int p = blockIdx.x * 4 + wib; p < P; p += nwaves) {
    const int np = num_points[p];
    const int cy = coors[p * 4 + 2], cx = coors[p * 4 + 3];
    int count = 0;
    for (int base = 0; base < nslots; base += 64) {
      const int slot = base + lane;
      const int pt = slot / V, cam = slot - pt * V;
      bool ok = slot < nslots && pt < np;
      float ix = 0.f, iy = 0.f;
      if (ok) {
        const float *pp = pillars + ((size_t)p * Tp + pt) * D;
        float x = pp[0], y = pp[1], z = pp[2];
        apply_affine_b(A, x, y, z);
        float nx, ny;
        ok = project_point_b(proj + cam * 16, x, y, z, ori_H, ori_W, nx, ny);
        ix = ((nx + 1.f) * Wi - 1.f) * 0.5f;
        iy = ((ny + 1.f) * Hi - 1.f) * 0.5f;
      }
      const unsigned long long mask = __ballot(ok);
      if (ok) {
        const int rank = count + __popcll(mask & ((1ull << lane) - 1ull));
        list[rank].ix = ix;
        list[rank].iy = iy;
        list[rank].cam = slot;          // slot id: camera = slot % V; dropout hash key
      }
      count += __popcll(mask);
    }
    __builtin_amdgcn_wave_barrier();
    if (count == 0) continue;   // the cell's context was 0 and carries no gradient path

    float qf[8], g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qf[i] = g[i] = 0.f;
    if (ch_ok) {
      unpack8(ld8(qfold + ((size_t)cy * Wb + cx) * C + ch0), qf);
      unpack8(ld8(grad_ctx + ((size_t)cy * Wb + cx) * C + ch0), g);
    }
    // ---- pass A: (m, l, ctx) exactly as the forward
    float m = -INFINITY, l = 0.f, ms = 0.f, acc[8];   // ms: the kept probability mass sum_j d_j p_j (before the 1 / l)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int e = sub; e < count; e += 4) {
      const KeyEntB k = list[e];
      float s8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) s8[i] = 0.f;
      if (ch_ok) bilin_gather8(img + (size_t)(k.cam % V) * Hi * Wi * C, Wi, C, bilin_setup(k.ix, k.iy, Hi, Wi), ch0, s8);
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) part = fmaf(qf[i], s8[i], part);
      const float sc = row16_sum(part);
      const float mn = fmaxf(m, sc);
      const float a = __expf(m - mn), pe = __expf(sc - mn);
      l = l * a + pe;
      const float pv = drop_p > 0.f ? (di_keep(seed, p, k.cam, drop_p) ? pe / (1.f - drop_p) : 0.f) : pe;
      ms = ms * a + pv;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = acc[i] * a + pv * s8[i];
      m = mn;
    }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float mo = __shfl_xor(m, off), lo = __shfl_xor(l, off);
      const float mn = fmaxf(m, mo);
      const float a = (m == mn) ? 1.f : __expf(m - mn);
      const float b = (mo == mn) ? 1.f : __expf(mo - mn);
      l = l * a + lo * b;
      ms = ms * a + __shfl_xor(ms, off) * b;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = acc[i] * a + __shfl_xor(acc[i], off) * b;
      m = mn;
    }
    const float inv = 1.f / l;
    float gc = 0.f;   // <g, ctx>
#pragma unroll
    for (int i = 0; i < 8; ++i) gc = fmaf(g[i], acc[i] * inv, gc);
    gc = row16_sum(gc);
    // the output also carries mass * (folded value bias): d out / d p_j gains d_j * gm, <g, ctx> gains gm * mass
    const float gm = grad_mass != nullptr ? (float)grad_mass[(size_t)cy * Wb + cx] : 0.f;
    gc = fmaf(gm, ms * inv, gc);
    // ---- pass B
    float gq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) gq[i] = 0.f;
    for (int e = sub; e < count; e += 4) {
      const KeyEntB k = list[e];
      const Bilin b = bilin_setup(k.ix, k.iy, Hi, Wi);
      float s8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) s8[i] = 0.f;
      if (ch_ok) bilin_gather8(img + (size_t)(k.cam % V) * Hi * Wi * C, Wi, C, b, ch0, s8);
      float part = 0.f, gs = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        part = fmaf(qf[i], s8[i], part);
        gs = fmaf(g[i], s8[i], gs);
      }
      const float sc = row16_sum(part);
      gs = row16_sum(gs);
      const float pj = __expf(sc - m) * inv;
      const float dj = drop_p > 0.f ? (di_keep(seed, p, k.cam, drop_p) ? 1.f / (1.f - drop_p) : 0.f) : 1.f;
      const float ds = pj * (dj * (gs + gm) - gc);      // ctx = sum_j d_j p_j s_j, mass = sum_j d_j p_j
      float gsj[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        gsj[i] = dj * pj * g[i] + ds * qf[i];
        gq[i] = fmaf(ds, s8[i], gq[i]);
      }
      float gst[8];
      to_lane_major(gsj, gst, slot, l16);
      scatter_rows(grad_img + (size_t)(k.cam % V) * Hi * Wi * C, Wi, C, b, l16, gst);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      gq[i] += __shfl_xor(gq[i], 16);
      gq[i] += __shfl_xor(gq[i], 32);
    }
    if (sub == 0 && ch_ok) {
      float *dst = grad_qfold + ((size_t)cy * Wb + cx) * C + ch0;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = gq[i];     // one pillar per cell: a plain store
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace di

extern "C" {

int di_bevwarp_gather_bwd(const void *grad_out, const float *depth, const float *img2lidar, const float *aug_fwd,
                          const float *xs, const float *ys, const float *pc_range, float *grad_bev, int n_views,
                          int Hi, int Wi, int Hb, int Wb, int C, int dtype, void *stream) {
  DI_REQUIRE(n_views > 0 && Hi > 0 && Wi > 0 && Hb > 0 && Wb > 0, "bad gather shape");
  DI_REQUIRE(C > 0 && C % 8 == 0 && C <= 128, "C=%d must be a multiple of 8, <= 128", C);
  const int total = n_views * Hi * ((Wi + di::kRunB - 1) / di::kRunB);   // one 16-lane group per run of pixels
  const int blocks = min((total + 15) / 16, 256 * 16);
  hipStream_t s = (hipStream_t)stream;
  static const int dbg = getenv("DI_BW_DBG") ? atoi(getenv("DI_BW_DBG")) : 0;   // measurement: 1 = no atomics, 2 = no gradient loads
  if (dtype == DI_F16)
    hipLaunchKernelGGL(di::bevwarp_gather_bwd_kernel<__half>, dim3(blocks), dim3(256), 0, s, (const __half *)grad_out,
                       depth, img2lidar, aug_fwd, xs, ys, pc_range, grad_bev, n_views, Hi, Wi, Hb, Wb, C, dbg);
  else if (dtype == DI_F32)
    hipLaunchKernelGGL(di::bevwarp_gather_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float *)grad_out,
                       depth, img2lidar, aug_fwd, xs, ys, pc_range, grad_bev, n_views, Hi, Wi, Hb, Wb, C, dbg);
  else {
    di::set_error("unsupported dtype %d", dtype);
    return DI_ERR_ARG;
  }
  return di::check_launch("bevwarp_gather_bwd");
}

int di_roi_align_bwd(const void *grad_out, const float *rois, float *grad_feat, int R, int N, int H, int W, int C,
                     float spatial_scale, int dtype, void *stream) {
  DI_REQUIRE(R >= 0 && N > 0 && H > 0 && W > 0, "bad roi_align shape");
  DI_REQUIRE(C > 0 && C % 8 == 0 && C <= 128, "C=%d must be a multiple of 8, <= 128", C);
  if (R == 0) return DI_OK;
  const int total = R * 49;
  const int blocks = min((total + 15) / 16, 256 * 16);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DI_F16)
    hipLaunchKernelGGL(di::roi_align_bwd_kernel<__half>, dim3(blocks), dim3(256), 0, s, (const __half *)grad_out, rois,
                       grad_feat, R, N, H, W, C, spatial_scale);
  else if (dtype == DI_F32)
    hipLaunchKernelGGL(di::roi_align_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float *)grad_out, rois,
                       grad_feat, R, N, H, W, C, spatial_scale);
  else {
    di::set_error("unsupported dtype %d", dtype);
    return DI_ERR_ARG;
  }
  return di::check_launch("roi_align_bwd");
}

int di_i2p_attn_bwd_mass(const void *img, const void *qfold, const void *grad_ctx, const void *grad_mass, const float *pillars,
                    const int32_t *coors, const int32_t *num_points, const float *proj, const float *aug_rev,
                    float *grad_img, float *grad_qfold, int P, int T, int D, int n_views, int Hi, int Wi, int Hb,
                    int Wb, int C, float ori_H, float ori_W, float dropout_p, unsigned long long seed, int dtype,
                    void *stream) {
  DI_REQUIRE(P >= 0 && T > 0 && D >= 3 && n_views > 0, "bad pillar shape P=%d T=%d D=%d V=%d", P, T, D, n_views);
  DI_REQUIRE(T * n_views <= di::kMaxSlotsB, "T*n_views=%d exceeds %d key slots", T * n_views, di::kMaxSlotsB);
  DI_REQUIRE(C > 0 && C % 8 == 0 && C <= 128, "C=%d must be a multiple of 8, <= 128", C);
  if (P == 0) return DI_OK;
  const int blocks = min((P + 3) / 4, 256 * 8);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DI_F16)
    hipLaunchKernelGGL(di::i2p_attn_bwd_kernel<__half>, dim3(blocks), dim3(256), 0, s, (const __half *)img,
                       (const __half *)qfold, (const __half *)grad_ctx, (const __half *)grad_mass, pillars, coors, num_points, proj, aug_rev,
                       grad_img, grad_qfold, P, T, D, n_views, Hi, Wi, Hb, Wb, C, ori_H, ori_W, dropout_p, seed, di::i2p_seed_ptr());
  else if (dtype == DI_F32)
    hipLaunchKernelGGL(di::i2p_attn_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float *)img,
                       (const float *)qfold, (const float *)grad_ctx, (const float *)grad_mass, pillars, coors, num_points, proj, aug_rev,
                       grad_img, grad_qfold, P, T, D, n_views, Hi, Wi, Hb, Wb, C, ori_H, ori_W, dropout_p, seed, di::i2p_seed_ptr());
  else {
    di::set_error("unsupported dtype %d", dtype);
    return DI_ERR_ARG;
  }
  return di::check_launch("i2p_attn_bwd");
}


int di_i2p_attn_bwd(const void *img, const void *qfold, const void *grad_ctx, const float *pillars,
                    const int32_t *coors, const int32_t *num_points, const float *proj, const float *aug_rev,
                    float *grad_img, float *grad_qfold, int P, int T, int D, int n_views, int Hi, int Wi, int Hb,
                    int Wb, int C, float ori_H, float ori_W, float dropout_p, unsigned long long seed, int dtype,
                    void *stream) {
  return di_i2p_attn_bwd_mass(img, qfold, grad_ctx, nullptr, pillars, coors, num_points, proj, aug_rev, grad_img, grad_qfold, P,
                              T, D, n_views, Hi, Wi, Hb, Wb, C, ori_H, ori_W, dropout_p, seed, dtype, stream);
}

}  // extern "C"
