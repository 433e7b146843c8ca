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
#include "warp_common.h"
#include "i2p_common.h"

namespace di {

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

// ---------------------------------------------------------------------------------
// image -> BEV (MMRI_I2P), two kernels.  The pillar attention is VALU-bound, not memory-bound: what
// matters is the instruction count per key, so everything that depends on the geometry only is done ONCE per sample.
//
//  i2p_keys_kernel   geometry only - once per sample and forward, shared by every encoder layer.  One wavefront per
//      pillar: the 64 lanes project the pillar's T*V (point, camera) slots (two rounds), ballot the valid ones and
//      write them, compacted and READY TO GATHER, into the KEY TABLE row of the pillar's BEV cell:
//          cnt[cell] | pillar[cell] | key[cell][T*V] = {pixel index of the upper-left corner | corner steps | slot,
//                                                       the four bilinear weights (0 for corners outside the map)}
//      (cells without a pillar keep cnt = 0 from the clear kernel that runs first; plain stores only).
//
//  i2p_attn_kernel   one wavefront per BEV CELL (not per pillar: empty cells get their zero row here, so the output
//      maps need no fill).  The cell's key row arrives with two coalesced 16-B-per-lane loads; the four 16-lane
//      groups each take every 4th key:  four corner rows (256 B each), score = sum_k w_k <q, f_k> with
//      v_dot2_f32_f16 on the raw fp16 rows (no conversions), online softmax with ONE running maximum per cell (shared
//      by the four groups: their partial sums then merge by plain addition), value update acc += (p w_k) f_k as
//      v_fma_mix_f32 (fp16 operand, fp32 accumulate).  ~110 VALU instructions per round of four keys; the first
//      version (projection of all 120 slots per pillar and layer, fp32 blend after 32 conversions per key, 64-bit
//      address arithmetic, per-group maxima) spent ~4x that.
// ---------------------------------------------------------------------------------
// kMaxSlots, KeyEnt: i2p_common.h (shared with the matrix-core attention pass, i2p_dense.hip)

__global__ __launch_bounds__(256) void i2p_clear_kernel(int *__restrict__ cnt, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) cnt[i] = 0;
}

__global__ __launch_bounds__(256) void i2p_keys_kernel(
    const float *__restrict__ pillars, const int32_t *__restrict__ coors, const int32_t *__restrict__ num_points,
    const float *__restrict__ proj, const float *__restrict__ aug, int *__restrict__ cnt, int *__restrict__ pil,
    KeyEnt *__restrict__ keys, int P, int Tp, int D, int V, int Hi, int Wi, int Hb, int Wb, float ori_H, float ori_W) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  const Affine A = load_affine(aug);
  const int nslots = Tp * V;
  const int np = num_points[p];
  const int cy = coors[p * 4 + 2], cx = coors[p * 4 + 3];
  if (np <= 0 || cy < 0 || cy >= Hb || cx < 0 || cx >= Wb) return;    // padding rows of a fixed-size pillar buffer
  const int cell = cy * Wb + cx;
  KeyEnt *row = keys + (size_t)cell * nslots;
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
      // torch grid_sample(bilinear, padding zeros, align_corners=False): corners outside the map contribute 0 - here
      // a zero weight on a clamped (always readable) pixel
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fx, y0 = (int)fy;
      const float ax = ix - fx, ay = iy - fy;
      const bool xl = x0 >= 0 && x0 < Wi, xh = x0 + 1 >= 0 && x0 + 1 < Wi;
      const bool yl = y0 >= 0 && y0 < Hi, yh = y0 + 1 >= 0 && y0 + 1 < Hi;
      const int xa = min(max(x0, 0), Wi - 1), xb = min(max(x0 + 1, 0), Wi - 1);
      const int ya = min(max(y0, 0), Hi - 1), yb = min(max(y0 + 1, 0), Hi - 1);
      KeyEnt k;
      k.pix = (cam * Hi + ya) * Wi + xa;
      k.info = (xb - xa) | ((yb - ya) << 1) | (slot << 8);
      k.w00 = (yl && xl) ? (1.f - ax) * (1.f - ay) : 0.f;
      k.w01 = (yl && xh) ? ax * (1.f - ay) : 0.f;
      k.w10 = (yh && xl) ? (1.f - ax) * ay : 0.f;
      k.w11 = (yh && xh) ? ax * ay : 0.f;
      k.pad0 = k.pad1 = 0;
      float4 *dst = reinterpret_cast<float4 *>(row + rank);
      dst[0] = reinterpret_cast<const float4 *>(&k)[0];
      dst[1] = reinterpret_cast<const float4 *>(&k)[1];
    }
    count += __popcll(mask);
  }
  if (lane == 0) {
    cnt[cell] = count;       // 0: no valid key, the cell stays 0 (:314-315)
    pil[cell] = p;
  }
}

template <typename T> struct Vec8;
template <> struct Vec8<__half> { typedef _Float16 type __attribute__((ext_vector_type(8))); };
template <> struct Vec8<float> { typedef float type __attribute__((ext_vector_type(8))); };

__device__ __forceinline__ float qdot(const Vec8<__half>::type &q, const Vec8<__half>::type &f) {
  float d = 0.f;
  d = __builtin_amdgcn_fdot2(__builtin_shufflevector(q, q, 0, 1), __builtin_shufflevector(f, f, 0, 1), d, false);
  d = __builtin_amdgcn_fdot2(__builtin_shufflevector(q, q, 2, 3), __builtin_shufflevector(f, f, 2, 3), d, false);
  d = __builtin_amdgcn_fdot2(__builtin_shufflevector(q, q, 4, 5), __builtin_shufflevector(f, f, 4, 5), d, false);
  d = __builtin_amdgcn_fdot2(__builtin_shufflevector(q, q, 6, 7), __builtin_shufflevector(f, f, 6, 7), d, false);
  return d;
}
__device__ __forceinline__ float qdot(const Vec8<float>::type &q, const Vec8<float>::type &f) {
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) d = fmaf(q[i], f[i], d);
  return d;
}

// acc[i] += c * f[i]: v_fma_mix_f32 takes the fp16 row element as is (fp32 accumulate, no conversion instruction, no
// temporaries); the compiler would otherwise convert all 32 elements first (v_cvt + v_pk_fma_f32, 32 more live VGPRs)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void axpy8(float (&acc)[8], float c, const Vec8<__half>::type &f) {
  const u32x4 u = __builtin_bit_cast(u32x4, f);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * j]) : "v"(u[j]), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * j + 1]) : "v"(u[j]), "v"(c));
  }
}
__device__ __forceinline__ void axpy8(float (&acc)[8], float c, const Vec8<float>::type &f) {
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = fmaf(c, f[i], acc[i]);
}

// x summed / maximised over the four 16-lane rows of the wave (lanes with equal lane % 16); every lane gets the result
__device__ __forceinline__ float rows_sum(float x) {
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}
__device__ __forceinline__ float rows_max(float x) {
  x = fmaxf(x, __shfl_xor(x, 16));
  x = fmaxf(x, __shfl_xor(x, 32));
  return x;
}

// one key of one 16-lane group: the four corner rows (in flight) and their bilinear weights
template <typename T>
struct KeyRows {
  typename Vec8<T>::type f00, f01, f10, f11;
  float w00, w01, w10, w11;
  int slot;
};

// MASS (training with attention dropout): also write the kept probability mass sum_j d_j p_j of every cell (1 without
// dropout): the folded value bias enters the output scaled by it.
template <typename T, bool FULLC, bool MASS = false>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 5 : 4) void i2p_attn_kernel(
    const T *__restrict__ img, const T *__restrict__ qfold, const int *__restrict__ cnt_tab,
    const int *__restrict__ pil_tab, const KeyEnt *__restrict__ keys, const int *__restrict__ order,
    T *__restrict__ ctx, T *__restrict__ valid_out, int ncell, int nslots, int Wi, int C_, float drop_p,
    unsigned long long seed, T *__restrict__ mass_out = nullptr, const unsigned long long *__restrict__ seed_add = nullptr) {
  typedef typename Vec8<T>::type V8;
  if (seed_add != nullptr) seed += *seed_add;
  const int C = FULLC ? 128 : C_;                // 128 channels: the row offsets are shifts
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  const int l16 = lane & 15, sub = lane >> 4;
  const bool ch_ok = l16 * kChPerLane < C;      // C < 128: the upper lanes of a group read channel 0.. (a valid
  const int ch0 = ch_ok ? l16 * kChPerLane : 0;  // address), carry no score and never store
  // Which cells: `order` lists the BEV cells sorted by azimuth around the ego vehicle (then radius).  XCD x (the blocks
  // with blockIdx % 8 == x: consecutive block ids go round-robin over the eight XCDs, each with a private 4 MB L2)
  // takes the x-th eighth of that list and its waves sweep it together (wave w: positions w, w + W, w + 2W, ...): at
  // any moment an XCD works on one narrow wedge of the scene = a band of columns in one or two cameras.
  // A wave has <= 64 positions (the host sizes the grid for that); lane j holds cell j and its key count.
  const int xcd = blockIdx.x & 7, W = (gridDim.x >> 3) * 4, lw = (blockIdx.x >> 3) * 4 + wib;
  const int chunk = (ncell + 7) >> 3, lo = xcd * chunk, hi = min(lo + chunk, ncell);
  const int pos = lo + lw + lane * W;
  const int cellv = pos < hi ? (order != nullptr ? order[pos] : pos) : -1;
  const int cntv = cellv >= 0 ? cnt_tab[cellv] : 0;
  unsigned long long todo = __ballot(cntv > 0 && cntv <= 64);
  unsigned long long crowded = __ballot(cntv > 64);
  unsigned long long empty = __ballot(cellv >= 0 && cntv == 0);
  while (empty) {                // no pillar / no valid key: the cell is 0 (:314-315)
    const int j = __builtin_ctzll(empty);
    empty &= empty - 1;
    const int cell = __builtin_amdgcn_readlane(cellv, j);
    if (sub == 0 && ch_ok) {
      V8 z;
#pragma unroll
      for (int i = 0; i < 8; ++i) z[i] = 0;
      *reinterpret_cast<V8 *>(ctx + (size_t)cell * C + ch0) = z;
    }
    if (lane == 0) valid_out[cell] = (T)0.f;
    if (MASS && lane == 0) mass_out[cell] = (T)0.f;
  }

  // corner rows of one key: 32-bit BYTE offsets from the (uniform) map base (global_load with an SGPR base)
  const unsigned stepx = (unsigned)C * sizeof(T), stepy = (unsigned)Wi * stepx, lane_off = (unsigned)ch0 * sizeof(T);
  const char *img_b = reinterpret_cast<const char *>(img);
  auto fetch_key = [&](KeyRows<T> &r, int pix, int info, float w00, float w01, float w10, float w11) {
    r.w00 = w00; r.w01 = w01; r.w10 = w10; r.w11 = w11;
    r.slot = info >> 8;
    const unsigned o00 = (unsigned)pix * stepx + lane_off;
    const unsigned dx = (info & 1) ? stepx : 0u, dy = (info & 2) ? stepy : 0u;
    r.f00 = *reinterpret_cast<const V8 *>(img_b + o00);
    r.f01 = *reinterpret_cast<const V8 *>(img_b + (o00 + dx));
    r.f10 = *reinterpret_cast<const V8 *>(img_b + (o00 + dy));
    r.f11 = *reinterpret_cast<const V8 *>(img_b + (o00 + dx + dy));
  };

  // online-softmax state of the cell being reduced: ONE running maximum for the whole cell, partial sums per group
  float m = -INFINITY, l = 0.f, ms = 0.f, acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const float keep_scale = 1.f / (1.f - drop_p);
  auto consume = [&](const KeyRows<T> &k, const V8 &q, bool live, int cell) {
    float part = k.w00 * qdot(q, k.f00) + k.w01 * qdot(q, k.f01) + k.w10 * qdot(q, k.f10) + k.w11 * qdot(q, k.f11);
    if (!ch_ok) part = 0.f;        // C < 128: lanes beyond the channels
    const float sc = live ? row16_sum(part) : -INFINITY;
    const float mn = fmaxf(m, rows_max(sc));             // >= one live key per round: finite
    const float a = __expf(m - mn);                      // first round: exp(-inf) = 0
    const float pe = __expf(sc - mn);                    // dead group: exp(-inf) = 0
    l = l * a + pe;
    // attention dropout (training, nn.MultiheadAttention dropout on the probabilities): a dropped key
    // stays in the softmax denominator, its value term vanishes, kept ones are scaled by 1 / (1 - p)
    float pv = pe;
    if (drop_p > 0.f) pv = di_keep(seed, pil_tab[cell], k.slot, drop_p) ? pe * keep_scale : 0.f;
    if (MASS) ms = ms * a + pv;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] *= a;
    axpy8(acc, pv * k.w00, k.f00);
    axpy8(acc, pv * k.w01, k.f01);
    axpy8(acc, pv * k.w10, k.f10);
    axpy8(acc, pv * k.w11, k.f11);
    m = mn;
  };
  auto finish = [&](int cell) {
    // the four groups hold partial sums under the same maximum: add them (lanes with equal l16)
    l = rows_sum(l);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = rows_sum(acc[i]);
    if (sub == 0 && ch_ok) {
      const float inv = 1.f / l;
      V8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = acc[i] * inv;
      *reinterpret_cast<V8 *>(ctx + (size_t)cell * C + ch0) = o;
    }
    if (lane == 0) valid_out[cell] = (T)1.f;
    if (MASS) {
      ms = rows_sum(ms);
      if (lane == 0) mass_out[cell] = (T)(ms / l);
      ms = 0.f;
    }
    m = -INFINITY;
    l = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  };

  // cells with more than 64 keys (their key row does not fit one register set per lane): plain loop, keys from the table
  while (crowded) {
    const int j = __builtin_ctzll(crowded);
    crowded &= crowded - 1;
    const int cell = __builtin_amdgcn_readlane(cellv, j), count = __builtin_amdgcn_readlane(cntv, j);
    const V8 q = *reinterpret_cast<const V8 *>(qfold + (size_t)cell * C + ch0);
    for (int e0 = 0; e0 < count; e0 += 4) {
      const float4 *kp = reinterpret_cast<const float4 *>(keys + (size_t)cell * nslots + min(e0 + sub, count - 1));
      const float4 k0 = kp[0], k1 = kp[1];
      KeyRows<T> rows;
      fetch_key(rows, __float_as_int(k0.x), __float_as_int(k0.y), k0.z, k0.w, k1.x, k1.y);
      consume(rows, q, e0 + sub < count, cell);
    }
    finish(cell);
  }
  if (!todo) return;

  // The other cells.  Every load below is UNCONDITIONAL (clamped addresses, a dummy re-read at the end): hipcc's
  // waitcnt insertion merges the counter states of all paths into a join, and a branch that loads would turn the
  // waits after it into "wait for everything in flight".
  struct Cell {
    int j, cell, count;
    float4 k0;     // lane e: key e of the cell = {pix, info, w00, w01 | w10, w11}
    float2 k1;
    V8 q;
  };
  auto open_cell = [&](Cell &c, int j) {
    c.j = j;
    c.cell = __builtin_amdgcn_readlane(cellv, j);
    c.count = __builtin_amdgcn_readlane(cntv, j);
    const float4 *kp = reinterpret_cast<const float4 *>(keys + (size_t)c.cell * nslots + min(lane, c.count - 1));
    c.k0 = kp[0];
    c.k1 = *reinterpret_cast<const float2 *>(kp + 1);
    c.q = *reinterpret_cast<const V8 *>(qfold + (size_t)c.cell * C + ch0);
  };
  auto fetch_round = [&](KeyRows<T> &rows, const Cell &c, int r) {
    const int e = min(4 * r + sub, c.count - 1);        // ragged last round: re-read the last key (masked later)
    fetch_key(rows, __float_as_int(__shfl(c.k0.x, e)), __float_as_int(__shfl(c.k0.y, e)), __shfl(c.k0.z, e),
              __shfl(c.k0.w, e), __shfl(c.k1.x, e), __shfl(c.k1.y, e));
  };
  auto next_cell = [&]() {
    if (!todo) return -1;
    const int j = __builtin_ctzll(todo);
    todo &= todo - 1;
    return j;
  };

  // Only the next cell's key row and query are requested ahead; the gathers themselves are hidden by the other waves
  // of the SIMD.  (A version that streamed the rounds through two row buffers - next round's rows in flight while the
  // current one is reduced, 115 VGPRs, 4 waves - measured the same time: the kernel is bound by instruction issue.)
  Cell cc, cn;
  open_cell(cc, next_cell());
  while (true) {
    const int jn = next_cell();
    open_cell(cn, jn >= 0 ? jn : cc.j);      // unconditional (a re-read at the end): see the note on waitcnt above
    const int rounds = (cc.count + 3) >> 2;
    for (int r = 0; r < rounds; ++r) {
      KeyRows<T> rows;
      fetch_round(rows, cc, r);
      consume(rows, cc.q, 4 * r + sub < cc.count, cc.cell);
    }
    finish(cc.cell);
    if (jn < 0) break;
    cc = cn;
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
  const WarpGeom G = load_warp_geom(depth, img2lidar, aug, xs, ys, pc_range, Hi, Wi, Hb, Wb);
  const int total = V * Hi * Wi;
  const int ngrp = gridDim.x * (blockDim.x >> 4);
  for (int pix = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4); pix < total; pix += ngrp) {
    float ix, iy;
    const bool lift = warp_position(G, pix, ix, iy);
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    if (lift && ch_ok) bilinear8(bev, Hb, Wb, C, ix, iy, ch0, o);
    if (ch_ok) st8(out + (size_t)pix * C + ch0, pack8f(o, T()));  // masked texels are 0 (:196)
  }
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

long long di_i2p_key_table_bytes(int Hb, int Wb, int T, int n_views) {
  return (long long)Hb * Wb * (8 + (long long)T * n_views * (long long)sizeof(di::KeyEnt));
}

// Geometry pass of the pillar attention: fills the key table (di_i2p_key_table_bytes) of ONE sample.
int di_i2p_build_keys(const float *pillars, const int32_t *coors, const int32_t *num_points, const float *proj,
                      const float *aug_rev, void *key_table, int P, int T, int D, int n_views, int Hi, int Wi, int Hb,
                      int Wb, float ori_H, float ori_W, void *stream) {
  DI_REQUIRE(P >= 0 && T > 0 && D >= 3 && n_views > 0, "bad pillar shape P=%d T=%d D=%d V=%d", P, T, D, n_views);
  DI_REQUIRE(T * n_views <= di::kMaxSlots, "T*n_views=%d exceeds %d key slots", T * n_views, di::kMaxSlots);
  DI_REQUIRE(Hb > 0 && Wb > 0 && Hi > 0 && Wi > 0, "bad map shape");
  const int ncell = Hb * Wb;
  int *cnt = reinterpret_cast<int *>(key_table);
  int *pil = cnt + ncell;
  di::KeyEnt *keys = reinterpret_cast<di::KeyEnt *>(pil + ncell);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(di::i2p_clear_kernel, dim3((ncell + 255) / 256), dim3(256), 0, s, cnt, ncell);
  if (P > 0)
    hipLaunchKernelGGL(di::i2p_keys_kernel, dim3((P + 3) / 4), dim3(256), 0, s, pillars, coors, num_points, proj, aug_rev,
                       cnt, pil, keys, P, T, D, n_views, Hi, Wi, Hb, Wb, ori_H, ori_W);
  return di::check_launch("i2p_build_keys");
}

// Attention pass: writes EVERY cell of ctx (Hb*Wb, C) and valid (Hb*Wb) - zeros where the table has no key.
// `mass` (optional, Hb*Wb like valid): the kept probability mass of every cell under attention dropout.
int di_i2p_attn_fwd_mass(const void *img, const void *qfold, const void *key_table, const int32_t *cell_order, void *ctx,
                         void *valid, void *mass, int T, int n_views, int Hi, int Wi, int Hb, int Wb, int C, float dropout_p,
                         unsigned long long seed, int dtype, void *stream) {
  DI_REQUIRE(T > 0 && n_views > 0 && T * n_views <= di::kMaxSlots, "T*n_views=%d exceeds %d key slots", T * n_views,
             di::kMaxSlots);
  DI_REQUIRE(C > 0 && C % 8 == 0 && C <= 128, "C=%d must be a multiple of 8, <= 128", C);
  DI_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p=%f out of [0,1)", (double)dropout_p);
  DI_REQUIRE((long long)n_views * Hi * Wi * C * 4 < (1ll << 32), "image map too large for 32-bit byte offsets");
  const int ncell = Hb * Wb;
  const int *cnt = reinterpret_cast<const int *>(key_table);
  const int *pil = cnt + ncell;
  const di::KeyEnt *keys = reinterpret_cast<const di::KeyEnt *>(pil + ncell);
  // a multiple of 8 blocks (one share per XCD); a wave walks <= 64 cells (their counts live one per lane)
  static const int want_blocks = getenv("DI_I2P_BLOCKS") ? atoi(getenv("DI_I2P_BLOCKS")) : 2048;
  int blocks = std::max(std::min((ncell + 3) / 4, want_blocks), (ncell + 255) / 256);
  blocks = (blocks + 7) / 8 * 8;
  hipStream_t s = (hipStream_t)stream;
#define DI_I2P_GO(TT, FULL)                                                                              \
  do {                                                                                                   \
    if (mass)                                                                                            \
      hipLaunchKernelGGL((di::i2p_attn_kernel<TT, FULL, true>), dim3(blocks), dim3(256), 0, s, (const TT *)img, \
                         (const TT *)qfold, cnt, pil, keys, cell_order, (TT *)ctx, (TT *)valid, ncell, T * n_views, \
                         Wi, C, dropout_p, seed, (TT *)mass, di::i2p_seed_ptr());                                          \
    else                                                                                                 \
      hipLaunchKernelGGL((di::i2p_attn_kernel<TT, FULL, false>), dim3(blocks), dim3(256), 0, s, (const TT *)img, \
                         (const TT *)qfold, cnt, pil, keys, cell_order, (TT *)ctx, (TT *)valid, ncell, T * n_views, \
                         Wi, C, dropout_p, seed, (TT *)nullptr, di::i2p_seed_ptr());                                         \
  } while (0)
  if (dtype == DI_F16) {
    if (C == 128) DI_I2P_GO(__half, true); else DI_I2P_GO(__half, false);
  } else if (dtype == DI_F32) {
    if (C == 128) DI_I2P_GO(float, true); else DI_I2P_GO(float, false);
  } else {
    di::set_error("unsupported dtype %d", dtype);
    return DI_ERR_ARG;
  }
#undef DI_I2P_GO
  return di::check_launch("i2p_attn_fwd");
}

int di_i2p_attn_fwd(const void *img, const void *qfold, const void *key_table, const int32_t *cell_order, void *ctx,
                    void *valid, int T, int n_views, int Hi, int Wi, int Hb, int Wb, int C, float dropout_p,
                    unsigned long long seed, int dtype, void *stream) {
  return di_i2p_attn_fwd_mass(img, qfold, key_table, cell_order, ctx, valid, nullptr, T, n_views, Hi, Wi, Hb, Wb, C, dropout_p,
                              seed, dtype, stream);
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
