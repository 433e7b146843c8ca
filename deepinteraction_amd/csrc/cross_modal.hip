// Cross-modal gather kernels for gfx950:
//   image -> BEV  pillar-projection attention   (reference encoder_utils.py:257-320, MMRI_I2P)
//   BEV -> image  depth scatter + bilinear gather (reference encoder_utils.py:142-199, BEVWarp)
//
// Both are HBM/L2-bound gathers over channels-last maps.  16 lanes own one texel
// (8 channels each), so every bilinear corner is one coalesced 256 B (fp16) row and a
// wavefront has four gathers in flight.  Geometry (projection, masks, sampling
// coordinates) is recomputed in registers in fp32 - it is a few dozen flops per key,
// cheaper than caching it in HBM.
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

// lidar point -> camera; the mask and normalisation of encoder_utils.py:157-170 / :281-295.
// Returns false when behind the camera or not strictly inside the image.
__device__ __forceinline__ bool project_point(const float *__restrict__ M /*4x4 row-major*/,
                                              float x, float y, float z, float ori_H, float ori_W,
                                              float &u, float &v, float &depth, float &nx, float &ny) {
  const float cx = M[0] * x + M[1] * y + M[2] * z + M[3];
  const float cy = M[4] * x + M[5] * y + M[6] * z + M[7];
  const float cz = M[8] * x + M[9] * y + M[10] * z + M[11];
  const float eps = 1e-5f;
  depth = cz;
  const float den = fmaxf(cz, eps);
  u = cx / den;
  v = cy / den;
  nx = (u / ori_W - 0.5f) * 2.f;
  ny = (v / ori_H - 0.5f) * 2.f;
  return cz > eps && nx > -1.f && nx < 1.f && ny > -1.f && ny < 1.f;
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

// ---------------------------------------------------------------------------------
// image -> BEV: one wavefront per pillar.
//   1. the 64 lanes project the pillar's T*6 (point, camera) slots (two rounds), ballot the
//      valid ones and compact their sampling coordinates into a per-wave LDS list;
//   2. the four 16-lane groups each take every 4th valid key: bilinear-gather its 128
//      channels (4 x 256 B rows), dot with the folded query (DPP row reduction), online
//      softmax (running max / sum / weighted sum in registers);
//   3. the four partial states are merged with two cross-row exchanges and one group
//      writes the 256 B context row of the pillar's BEV cell.
// ---------------------------------------------------------------------------------
constexpr int kMaxSlots = 128;

struct KeyEnt {
  float ix, iy;
  int cam;
};

template <typename T, bool FULLC>
__global__ __launch_bounds__(256) void i2p_attn_fwd_kernel(
    const T *__restrict__ img, const T *__restrict__ qfold, const float *__restrict__ pillars,
    const int32_t *__restrict__ coors, const int32_t *__restrict__ num_points,
    const float *__restrict__ proj, const float *__restrict__ aug, T *__restrict__ ctx,
    T *__restrict__ valid_out, int P, int Tp, int D, int V, int Hi, int Wi, int Hb, int Wb, int C,
    float ori_H, float ori_W, float drop_p, unsigned long long seed) {
  __shared__ KeyEnt s_list[4][kMaxSlots];
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  const int l16 = lane & 15, sub = lane >> 4;
  const bool ch_ok = FULLC || l16 * kChPerLane < C;
  const int ch0 = l16 * kChPerLane;
  KeyEnt *list = s_list[wib];
  const Affine A = load_affine(aug);
  const int nslots = Tp * V;
  const int nwaves = gridDim.x * 4;
  for (int p = blockIdx.x * 4 + wib; p < P; p += nwaves) {
    const int np = num_points[p];
    const int cy = coors[p * 4 + 2], cx = coors[p * 4 + 3];
    int count = 0;
    for (int base = 0; base < nslots; base += 64) {
      const int slot = base + lane;
      const int pt = slot / V, cam = slot - pt * V;  // slot = point*6 + cam (:298,:309-310)
      bool ok = slot < nslots && pt < np;            // slots >= num_points masked (:303-307)
      float ix = 0.f, iy = 0.f;
      if (ok) {
        const float *pp = pillars + ((size_t)p * Tp + pt) * D;
        float x = pp[0], y = pp[1], z = pp[2];
        apply_affine(A, x, y, z);
        float u, v, dep, nx, ny;
        ok = project_point(proj + cam * 16, x, y, z, ori_H, ori_W, u, v, dep, nx, ny);
        ix = ((nx + 1.f) * Wi - 1.f) * 0.5f;  // grid_sample un-normalise, align_corners=False
        iy = ((ny + 1.f) * Hi - 1.f) * 0.5f;
      }
      const unsigned long long mask = __ballot(ok);
      if (ok) {
        const int rank = count + __popcll(mask & ((1ull << lane) - 1ull));
        list[rank].ix = ix;
        list[rank].iy = iy;
        list[rank].cam = slot;          // the slot id: camera = slot % V, and the dropout hash key
      }
      count += __popcll(mask);
    }
    __builtin_amdgcn_wave_barrier();
    if (count == 0) continue;  // no valid key: the cell stays 0 (:314-315)

    float qf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qf[i] = 0.f;
    if (ch_ok) unpack8(ld8(qfold + ((size_t)cy * Wb + cx) * C + ch0), qf);

    float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int e = sub; e < count; e += 4) {
      const KeyEnt k = list[e];
      float s8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) s8[i] = 0.f;
      if (ch_ok) bilinear8(img + (size_t)(k.cam % V) * Hi * Wi * C, Hi, Wi, C, k.ix, k.iy, ch0, s8);
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) part = fmaf(qf[i], s8[i], part);
      const float sc = row16_sum(part);
      const float mn = fmaxf(m, sc);
      const float a = __expf(m - mn);  // first key: exp(-inf) = 0
      const float pe = __expf(sc - mn);
      l = l * a + pe;
      // attention dropout (training, nn.MultiheadAttention dropout on the probabilities): a dropped key
      // stays in the softmax denominator, its value term vanishes, kept ones are scaled by 1 / (1 - p)
      const float pv = drop_p > 0.f ? (di_keep(seed, p, k.cam, drop_p) ? pe / (1.f - drop_p) : 0.f) : pe;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = acc[i] * a + pv * s8[i];
      m = mn;
    }
    // merge the four groups' online-softmax states (lanes with equal l16)
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float mo = __shfl_xor(m, off), lo = __shfl_xor(l, off);
      const float mn = fmaxf(m, mo);
      const float a = (m == mn) ? 1.f : __expf(m - mn);
      const float b = (mo == mn) ? 1.f : __expf(mo - mn);
      l = l * a + lo * b;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = acc[i] * a + __shfl_xor(acc[i], off) * b;
      m = mn;
    }
    if (sub == 0 && ch_ok) {
      const float inv = 1.f / l;
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = acc[i] * inv;
      st8(ctx + ((size_t)cy * Wb + cx) * C + ch0, pack8f(o, T()));
    }
    if (lane == 0) valid_out[(size_t)cy * Wb + cx] = (T)1.f;
    __builtin_amdgcn_wave_barrier();  // list is reused by the next pillar
  }
}

// ---------------------------------------------------------------------------------
// BEV -> image, step 1: sparse depth scatter.  One thread per raw point; the
// (point index, depth) pair is packed into 64 bits so a single atomicMax implements
// "the highest point index wins" deterministically (the reference leaves duplicates
// to write order, encoder_utils.py:174).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void depth_scatter_kernel(
    const float *__restrict__ pts, int n_pts, int stride, const float *__restrict__ proj,
    const float *__restrict__ aug, unsigned long long *__restrict__ packed, int V, int Hi, int Wi,
    float ori_H, float ori_W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pts) return;
  const Affine A = load_affine(aug);
  float x = pts[(size_t)i * stride], y = pts[(size_t)i * stride + 1], z = pts[(size_t)i * stride + 2];
  apply_affine(A, x, y, z);
  for (int v = 0; v < V; ++v) {
    float u, w, dep, nx, ny;
    if (!project_point(proj + v * 16, x, y, z, ori_H, ori_W, u, w, dep, nx, ny)) continue;
    int r = (int)(w / ori_H * (float)Hi);  // .long() truncation of :174
    int c = (int)(u / ori_W * (float)Wi);
    r = min(max(r, 0), Hi - 1);
    c = min(max(c, 0), Wi - 1);
    const unsigned long long key =
        ((unsigned long long)(unsigned)(i + 1) << 32) | (unsigned long long)__float_as_uint(dep);
    atomicMax(packed + ((size_t)v * Hi + r) * Wi + c, key);
  }
}

__global__ __launch_bounds__(256) void depth_unpack_kernel(const unsigned long long *__restrict__ packed,
                                                           float *__restrict__ depth, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) depth[i] = __uint_as_float((unsigned)(packed[i] & 0xffffffffull));
}

// ---------------------------------------------------------------------------------
// BEV -> image, step 3: un-project each feature pixel through its completed depth and
// bilinear-gather the BEV map.  One 16-lane group per output texel.
// ---------------------------------------------------------------------------------
template <typename T, bool FULLC>
__global__ __launch_bounds__(256) void bevwarp_gather_kernel(
    const T *__restrict__ bev, const float *__restrict__ depth, const float *__restrict__ img2lidar,
    const float *__restrict__ aug, const float *__restrict__ xs, const float *__restrict__ ys,
    const float *__restrict__ pc_range, T *__restrict__ out, int V, int Hi, int Wi, int Hb, int Wb,
    int C) {
  const int l16 = threadIdx.x & 15;
  const bool ch_ok = FULLC || l16 * kChPerLane < C;
  const int ch0 = l16 * kChPerLane;
  const Affine A = load_affine(aug);
  const float r0 = pc_range[0], r1 = pc_range[1], r2 = pc_range[2];
  const float r3 = pc_range[3], r4 = pc_range[4], r5 = pc_range[5];
  const int total = V * Hi * Wi;
  const int ngrp = gridDim.x * (blockDim.x >> 4);
  for (int pix = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4); pix < total; pix += ngrp) {
    const int v = pix / (Hi * Wi);
    const int rem = pix - v * Hi * Wi;
    const int yy = rem / Wi, xx = rem - yy * Wi;
    const float d = depth[pix];
    const float X = xs[xx] * d, Y = ys[yy] * d;  // [x*d, y*d, d, 1] (:185-187)
    const float *M = img2lidar + v * 16;
    float x = M[0] * X + M[1] * Y + M[2] * d + M[3];
    float y = M[4] * X + M[5] * Y + M[6] * d + M[7];
    float z = M[8] * X + M[9] * Y + M[10] * d + M[11];
    apply_affine(A, x, y, z);  // re-apply the augmentation (:189)
    const bool lift = x > r0 && y > r1 && z > r2 && x < r3 && y < r4 && z < r5;  // strict (:191-192)
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    if (lift && ch_ok) {
      const float gx = ((x - r0) / (r3 - r0) - 0.5f) * 2.f;  // x -> BEV width (:193-194)
      const float gy = ((y - r1) / (r4 - r1) - 0.5f) * 2.f;  // y -> BEV height
      const float ix = ((gx + 1.f) * Wb - 1.f) * 0.5f;
      const float iy = ((gy + 1.f) * Hb - 1.f) * 0.5f;
      bilinear8(bev, Hb, Wb, C, ix, iy, ch0, o);
    }
    if (ch_ok) st8(out + (size_t)pix * C + ch0, pack8f(o, T()));  // masked texels are 0 (:196)
  }
}

template <typename T>
static int run_i2p(const void *img, const void *qfold, const float *pillars, const int32_t *coors,
                   const int32_t *num_points, const float *proj, const float *aug_rev, void *ctx,
                   void *valid, int P, int Tp, int D, int V, int Hi, int Wi, int Hb, int Wb, int C,
                   float ori_H, float ori_W, float drop_p, unsigned long long seed, hipStream_t stream) {
  if (P == 0) return DI_OK;
  const int blocks = min((P + 3) / 4, 256 * 8);
#define DI_I2P(FULL)                                                                              \
  hipLaunchKernelGGL((i2p_attn_fwd_kernel<T, FULL>), dim3(blocks), dim3(256), 0, stream,          \
                     (const T *)img, (const T *)qfold, pillars, coors, num_points, proj, aug_rev, \
                     (T *)ctx, (T *)valid, P, Tp, D, V, Hi, Wi, Hb, Wb, C, ori_H, ori_W, drop_p, seed)
  if (C == 128) DI_I2P(true); else DI_I2P(false);
#undef DI_I2P
  return check_launch("i2p_attn_fwd");
}

template <typename T>
static int run_gather(const void *bev, const float *depth, const float *img2lidar, const float *aug,
                      const float *xs, const float *ys, const float *pc_range, void *out, int V,
                      int Hi, int Wi, int Hb, int Wb, int C, hipStream_t stream) {
  const int total = V * Hi * Wi;
  const int blocks = min((total + 15) / 16, 256 * 16);
#define DI_GAT(FULL)                                                                               \
  hipLaunchKernelGGL((bevwarp_gather_kernel<T, FULL>), dim3(blocks), dim3(256), 0, stream,         \
                     (const T *)bev, depth, img2lidar, aug, xs, ys, pc_range, (T *)out, V, Hi, Wi, \
                     Hb, Wb, C)
  if (C == 128) DI_GAT(true); else DI_GAT(false);
#undef DI_GAT
  return check_launch("bevwarp_gather_fwd");
}

}  // namespace di

extern "C" {

int di_i2p_attn_fwd_ex(const void *img, const void *qfold, const float *pillars, const int32_t *coors,
                       const int32_t *num_points, const float *proj, const float *aug_rev, void *ctx,
                       void *valid, int P, int T, int D, int n_views, int Hi, int Wi, int Hb, int Wb,
                       int C, float ori_H, float ori_W, float dropout_p, unsigned long long seed, int dtype,
                       void *stream) {
  DI_REQUIRE(P >= 0 && T > 0 && D >= 3 && n_views > 0, "bad pillar shape P=%d T=%d D=%d V=%d", P, T, D, n_views);
  DI_REQUIRE(T * n_views <= di::kMaxSlots, "T*n_views=%d exceeds %d key slots", T * n_views, di::kMaxSlots);
  DI_REQUIRE(C > 0 && C % 8 == 0 && C <= 128, "C=%d must be a multiple of 8, <= 128", C);
  DI_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p=%f out of [0,1)", (double)dropout_p);
  if (dtype == DI_F16)
    return di::run_i2p<__half>(img, qfold, pillars, coors, num_points, proj, aug_rev, ctx, valid, P, T,
                               D, n_views, Hi, Wi, Hb, Wb, C, ori_H, ori_W, dropout_p, seed, (hipStream_t)stream);
  if (dtype == DI_F32)
    return di::run_i2p<float>(img, qfold, pillars, coors, num_points, proj, aug_rev, ctx, valid, P, T,
                              D, n_views, Hi, Wi, Hb, Wb, C, ori_H, ori_W, dropout_p, seed, (hipStream_t)stream);
  di::set_error("unsupported dtype %d", dtype);
  return DI_ERR_ARG;
}

int di_i2p_attn_fwd(const void *img, const void *qfold, const float *pillars, const int32_t *coors,
                    const int32_t *num_points, const float *proj, const float *aug_rev, void *ctx,
                    void *valid, int P, int T, int D, int n_views, int Hi, int Wi, int Hb, int Wb,
                    int C, float ori_H, float ori_W, int dtype, void *stream) {
  return di_i2p_attn_fwd_ex(img, qfold, pillars, coors, num_points, proj, aug_rev, ctx, valid, P, T, D, n_views,
                            Hi, Wi, Hb, Wb, C, ori_H, ori_W, 0.f, 0ull, dtype, stream);
}

int di_depth_scatter(const float *pts, int n_pts, int pt_stride, const float *proj,
                     const float *aug_rev, unsigned long long *packed, float *depth, int n_views,
                     int Hi, int Wi, float ori_H, float ori_W, void *stream) {
  DI_REQUIRE(n_pts >= 0 && pt_stride >= 3 && n_views > 0 && Hi > 0 && Wi > 0, "bad scatter shape");
  hipStream_t s = (hipStream_t)stream;
  if (n_pts > 0) {
    hipLaunchKernelGGL(di::depth_scatter_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, s, pts, n_pts,
                       pt_stride, proj, aug_rev, packed, n_views, Hi, Wi, ori_H, ori_W);
    int rc = di::check_launch("depth_scatter");
    if (rc) return rc;
  }
  const int n = n_views * Hi * Wi;
  hipLaunchKernelGGL(di::depth_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, packed, depth, n);
  return di::check_launch("depth_unpack");
}

int di_bevwarp_gather_fwd(const void *bev, const float *depth, const float *img2lidar,
                          const float *aug_fwd, const float *xs, const float *ys,
                          const float *pc_range, void *out, int n_views, int Hi, int Wi, int Hb,
                          int Wb, int C, int dtype, void *stream) {
  DI_REQUIRE(n_views > 0 && Hi > 0 && Wi > 0 && Hb > 0 && Wb > 0, "bad gather shape");
  DI_REQUIRE(C > 0 && C % 8 == 0 && C <= 128, "C=%d must be a multiple of 8, <= 128", C);
  if (dtype == DI_F16)
    return di::run_gather<__half>(bev, depth, img2lidar, aug_fwd, xs, ys, pc_range, out, n_views, Hi,
                                  Wi, Hb, Wb, C, (hipStream_t)stream);
  if (dtype == DI_F32)
    return di::run_gather<float>(bev, depth, img2lidar, aug_fwd, xs, ys, pc_range, out, n_views, Hi,
                                 Wi, Hb, Wb, C, (hipStream_t)stream);
  di::set_error("unsupported dtype %d", dtype);
  return DI_ERR_ARG;
}

}  // extern "C"
