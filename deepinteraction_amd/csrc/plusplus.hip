// DeepInteraction++ operators for gfx950: multi-scale deformable attention, the polar
// ray attention's two samplers, and a small multi-head attention for short sequences.
//
//   di_ms_deform_attn_fwd     mmcv `MultiScaleDeformableAttention` core (8 heads x 16 ch) with the softmax over the
//                             L*P logits and the offset -> sampling location arithmetic fused in; replaces the
//                             mmcv CUDA op `ms_deform_attn` reached from fusion_transformerv4.py:170-178 (self
//                             attention over 2 levels) and :238 (MMRI_P2I over the warped BEV map)
//   di_grid_gather_fwd        bilinear `grid_sample` of a channels-last map at an explicit normalised grid (+ an
//                             additive term): the polar ray queries, fusion_transformerv4.py:574-575
//   di_polar_bev_sample_fwd   fusion_transformerv4.py:581-640 for ALL cameras in one launch: every BEV cell lifts
//                             its 10 height samples into each camera, averages the sampling location, reads the
//                             camera's polar map, averages over the cameras that see it and adds the residual
//   di_mha_small_fwd          softmax(QK^T/sqrt(16))V for many short sequences (rays x image columns),
//                             flash-attn's role at fusion_transformerv4.py:697-700
//
// All are gather / HBM-bound: 16 lanes own one 128-channel texel row (16 B per lane), fp32 accumulation.
#include <type_traits>

#include <hip/hip_ext.h>

#include "di_common.h"

namespace di {
namespace pp {

constexpr int kMaxLevels = 4;
struct Levels {
  int n;
  int h[kMaxLevels], w[kMaxLevels], start[kMaxLevels];
};

// One corner of a bilinear footprint, zero outside the map.
template <typename T>
__device__ __forceinline__ void corner8(const T *__restrict__ map, int H, int W, int sy, int sx, int y, int x,
                                        float wgt, float (&acc)[8]) {
  if (y < 0 || y >= H || x < 0 || x >= W) return;
  float f[8];
  unpack8(ld8(map + (size_t)y * sy + (size_t)x * sx), f);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = fmaf(wgt, f[i], acc[i]);
}

// Bilinear sample at pixel coordinates (px, py) (texel centres at integers), zeros padding; `map` already points
// at the lane's 8 channels; (sy, sx) = elements between vertically / horizontally neighbouring texels.
template <typename T>
__device__ __forceinline__ void bilinear8(const T *__restrict__ map, int H, int W, int sy, int sx, float px, float py,
                                          float wgt, float (&acc)[8]) {
  if (!(px > -1.f && px < (float)W && py > -1.f && py < (float)H)) return;      // also rejects NaN / huge
  const float fx = floorf(px), fy = floorf(py);
  const int x0 = (int)fx, y0 = (int)fy;
  const float ax = px - fx, ay = py - fy;
  corner8(map, H, W, sy, sx, y0, x0, wgt * (1.f - ay) * (1.f - ax), acc);
  corner8(map, H, W, sy, sx, y0, x0 + 1, wgt * (1.f - ay) * ax, acc);
  corner8(map, H, W, sy, sx, y0 + 1, x0, wgt * ay * (1.f - ax), acc);
  corner8(map, H, W, sy, sx, y0 + 1, x0 + 1, wgt * ay * ax, acc);
}

// N consecutive elements (N in {4, 8, 16}, N*sizeof(T)-byte aligned up to 16 B) as floats, in the widest loads.
template <int N>
__device__ __forceinline__ void ldvec(const float *p, float (&f)[N]) {
  const float4 *q = reinterpret_cast<const float4 *>(p);
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    const float4 v = q[i];
    f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
  }
}
template <int N>
__device__ __forceinline__ void ldvec(const __half *p, float (&f)[N]) {
  if constexpr (N == 4) {
    const uint2 r = *reinterpret_cast<const uint2 *>(p);
    const __half2 *h = reinterpret_cast<const __half2 *>(&r);
    const float2 a = __half22float2(h[0]), b = __half22float2(h[1]);
    f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y;
  } else {
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
      float t[8];
      unpack8(ld8(p + 8 * i), t);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[8 * i + j] = t[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Deformable attention.  value (bs, S, 128) [S = sum H_l W_l, 8 heads x 16 ch]; off (bs*nq rows, row stride
// off_rs) holding (8, L, P, 2) raw offsets; logit (rows, stride logit_rs) holding (8, L*P); ref (ref_bs in {1,bs},
// nq, L, 2) fp32 in [0,1]; out (bs, nq, 128).  lane = (query, head, half of the head's 16 channels).
template <typename T, int L, int P>
__global__ __launch_bounds__(256) void ms_deform_attn_kernel(const T *__restrict__ value, const T *__restrict__ off,
                                                             int off_rs, const T *__restrict__ logit, int logit_rs,
                                                             const float *__restrict__ ref, int ref_shared,
                                                             T *__restrict__ out, int bs, int nq, int S, Levels lv) {
  constexpr int LP = L * P;
  const int l16 = threadIdx.x & 15;
  const int head = l16 >> 1, half = l16 & 1;
  const long long total = (long long)bs * nq;
  // every XCD (private L2) takes one contiguous range of queries: with the default round-robin of consecutive
  // workgroups over the 8 XCDs each L2 ends up fetching the WHOLE value map (measured: 8x the map per launch)
  const long long row = (long long)xcd_remap(blockIdx.x, gridDim.x) * 16 + (threadIdx.x >> 4);
  if (row >= total) return;
  const int b = (int)(row / nq), q = (int)(row - (long long)b * nq);
  // softmax over the head's L*P logits
  float w[LP], ofs[LP * 2];
  ldvec<LP>(logit + (size_t)row * logit_rs + head * LP, w);
  ldvec<LP * 2>(off + (size_t)row * off_rs + head * LP * 2, ofs);
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
  const float *rf = ref + ((size_t)(ref_shared ? 0 : b) * nq + q) * L * 2;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const T *vb = value + (size_t)b * S * 128 + head * 16 + half * 8;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int H = lv.h[l], W = lv.w[l];
    const T *map = vb + (size_t)lv.start[l] * 128;
    const float rx = rf[l * 2], ry = rf[l * 2 + 1];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float ox = ofs[(l * P + p) * 2], oy = ofs[(l * P + p) * 2 + 1];
      // loc = ref + off / (W, H);  pixel = loc * size - 0.5   (grid_sample, align_corners=False)
      const float px = (rx + ox / (float)W) * (float)W - 0.5f;
      const float py = (ry + oy / (float)H) * (float)H - 0.5f;
      bilinear8(map, H, W, W * 128, 128, px, py, w[l * P + p] * inv, acc);
    }
  }
  st8(out + (size_t)row * 128 + head * 16 + half * 8, pack8f(acc, T()));
}

// ------------------------------------------------------------------------------------------------
// Deformable attention over a HEAD-MAJOR value map (fp16 inference, round 5):  value (bs, 8, S, 16) - what the value
// projection writes when asked to (di_pointwise_chain_hm_fwd / di_pointwise_multi_warp_hm_fwd).
//
// Why: the kernel above is bound by the texture addresser, not by bytes (PMC, round 3: traffic 1.05 x algorithmic at 0.15
// of the HBM roofline).  A head's 16 channels are 32 B of a 256-B channels-last texel, so one wave-level load touches 32
// distinct 32-B pieces - one per clock - and a query needs 8 heads x L*P points x 4 corners = 256 of them.  Head-major, the
// two corners of a footprint ROW are 64 contiguous bytes: four lanes (corner x, channel half) fetch them as ONE piece,
// half as many addresser cycles per query.  Lane = (query, head, corner column cx, channel half): 32 lanes per query; a
// lane accumulates the left OR the right corners of every sample and the two are added at the end (DPP).  The geometry
// of a sample (location, floor, bounds, the four weights times the soft-max probability) is evaluated ONCE per quad -
// lane q of a quad owns the points q, q + 4 - and handed to the other lanes by quad broadcasts (DPP, no LDS); values go
// into the fp32 accumulators as v_fma_mix_f32 (fp16 operand as is, no conversions).
template <int SRC>
__device__ __forceinline__ int quad_bcast(int v) {
  constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
  return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xF, 0xF, true);
}
template <int SRC>
__device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, quad_bcast<SRC>(__builtin_bit_cast(int, v)));
}
template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int kQuadSwap1 = 0xB1;   // quad_perm [1, 0, 3, 2]
constexpr int kQuadSwap2 = 0x4E;   // quad_perm [2, 3, 0, 1]

typedef unsigned msda_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void axpy8_mix(float (&acc)[8], float c, const msda_u4 &u) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * j]) : "v"(u[j]), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * j + 1]) : "v"(u[j]), "v"(c));
  }
}

template <int B, int E, class F>
__device__ __forceinline__ void msda_static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    msda_static_for<B + 1, E>(f);
  }
}

template <int L, int P>
__global__ __launch_bounds__(256) void ms_deform_attn_hm_kernel(const __half *__restrict__ value, const __half *__restrict__ off,
                                                                int off_rs, const __half *__restrict__ logit, int logit_rs,
                                                                const float *__restrict__ ref, int ref_shared,
                                                                __half *__restrict__ out, int bs, int nq, int S, Levels lv) {
  static_assert(P == 4 && (L == 1 || L == 2), "a quad's four lanes own the four points of a level");
  constexpr int NS = L;                        // points per lane: point s * 4 + q4 (level s)
  const int l32 = threadIdx.x & 31, head = l32 >> 2, q4 = l32 & 3, cx = q4 >> 1, half = q4 & 1;
  const long long total = (long long)bs * nq;
  const long long row = (long long)xcd_remap(blockIdx.x, gridDim.x) * 8 + (threadIdx.x >> 5);
  if (row >= total) return;                    // (whole 32-lane groups: the DPP exchanges stay inside a quad)
  const int b = (int)(row / nq), q = (int)(row - (long long)b * nq);
  const _Float16 *lg = reinterpret_cast<const _Float16 *>(logit) + (size_t)row * logit_rs + head * (L * P);
  const _Float16 *of = reinterpret_cast<const _Float16 *>(off) + (size_t)row * off_rs + head * (L * P * 2);
  // soft-max over the head's L*P logits: every lane holds NS of them
  float e[NS], m = -INFINITY;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    e[s] = (float)lg[s * 4 + q4];
    m = fmaxf(m, e[s]);
  }
  m = fmaxf(m, quad_perm<kQuadSwap1>(m));
  m = fmaxf(m, quad_perm<kQuadSwap2>(m));
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    e[s] = __expf(e[s] - m);
    sum += e[s];
  }
  sum += quad_perm<kQuadSwap1>(sum);
  sum += quad_perm<kQuadSwap2>(sum);
  const float inv = 1.f / sum;
  const float *rf = ref + ((size_t)(ref_shared ? 0 : b) * nq + q) * L * 2;
  // geometry of this lane's points: the clamped upper-left texel (index inside the head's plane), the steps to the right /
  // lower corner (0 at the border: the weight is 0 there), the four weights (x the probability; 0 = zero padding)
  unsigned info[NS];
  float w00[NS], w01[NS], w10[NS], w11[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int H = lv.h[s], W = lv.w[s];
    const float ox = (float)of[(s * 4 + q4) * 2], oy = (float)of[(s * 4 + q4) * 2 + 1];
    // loc = ref + off / (W, H);  pixel = loc * size - 0.5   (grid_sample, align_corners=False)
    const float px = (rf[s * 2] + ox / (float)W) * (float)W - 0.5f;
    const float py = (rf[s * 2 + 1] + oy / (float)H) * (float)H - 0.5f;
    const bool live = px > -1.f && px < (float)W && py > -1.f && py < (float)H;      // also rejects NaN / huge
    const float fx = live ? floorf(px) : 0.f, fy = live ? floorf(py) : 0.f;
    const int x0 = (int)fx, y0 = (int)fy;
    const float ax = live ? px - fx : 0.f, ay = live ? py - fy : 0.f;
    const bool xl = live && x0 >= 0, xh = live && x0 + 1 < W, yl = y0 >= 0, yh = y0 + 1 < H;
    const int xa = max(x0, 0), xb = min(x0 + 1, W - 1), ya = max(y0, 0), yb = min(y0 + 1, H - 1);
    info[s] = (unsigned)(lv.start[s] + ya * W + xa) | ((unsigned)(xb - xa) << 30) | ((unsigned)(yb - ya) << 31);
    const float a = e[s] * inv;
    w00[s] = (yl && xl) ? a * (1.f - ay) * (1.f - ax) : 0.f;
    w01[s] = (yl && xh) ? a * (1.f - ay) * ax : 0.f;
    w10[s] = (yh && xl) ? a * ay * (1.f - ax) : 0.f;
    w11[s] = (yh && xh) ? a * ay * ax : 0.f;
  }
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const unsigned char *plane = reinterpret_cast<const unsigned char *>(value) + ((size_t)(b * 8 + head) * S) * 32 + half * 16;
  msda_static_for<0, L * P>([&](auto pc) {
    constexpr int p = decltype(pc)::value, src = p & 3, s = p >> 2;
    const unsigned inf_ = (unsigned)quad_bcast<src>((int)info[s]);
    const float a0 = quad_bcast<src>(w00[s]), a1 = quad_bcast<src>(w01[s]);
    const float b0 = quad_bcast<src>(w10[s]), b1 = quad_bcast<src>(w11[s]);
    const float wt = cx ? a1 : a0, wb = cx ? b1 : b0;
    const unsigned it = (inf_ & 0x3FFFFFFFu) + (cx ? ((inf_ >> 30) & 1u) : 0u);
    const unsigned ib = it + ((inf_ >> 31) ? (unsigned)lv.w[s] : 0u);
    const msda_u4 top = *reinterpret_cast<const msda_u4 *>(plane + (size_t)it * 32);
    const msda_u4 bot = *reinterpret_cast<const msda_u4 *>(plane + (size_t)ib * 32);
    axpy8_mix(acc, wt, top);
    axpy8_mix(acc, wb, bot);
  });
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] += quad_perm<kQuadSwap2>(acc[i]);      // left + right corners
  if (cx == 0) st8(out + (size_t)row * 128 + head * 16 + half * 8, pack8f(acc, __half()));
}

// ------------------------------------------------------------------------------------------------
// out[g, n, :] = bilinear(feat[g / per_feat], grid[g, n]) (+ add[n, :]);  feat (Bf,H,W,C), grid (Bg,N,2) in [-1,1].
template <typename T>
__global__ __launch_bounds__(256) void grid_gather_kernel(const T *__restrict__ feat, const float *__restrict__ grid,
                                                          const T *__restrict__ add, T *__restrict__ out, int Bg,
                                                          int N, int per_feat, int H, int W, int C) {
  const int l16 = threadIdx.x & 15, ch0 = l16 * kChPerLane;
  const long long total = (long long)Bg * N;
  const long long idx = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (idx >= total || ch0 >= C) return;
  const int g = (int)(idx / N), n = (int)(idx - (long long)g * N);
  const float gx = grid[idx * 2], gy = grid[idx * 2 + 1];
  const float px = ((gx + 1.f) * (float)W - 1.f) * 0.5f, py = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  bilinear8(feat + (size_t)(g / per_feat) * H * W * C + ch0, H, W, W * C, C, px, py, 1.f, acc);
  if (add != nullptr) {
    float a[8];
    unpack8(ld8(add + (size_t)n * C + ch0), a);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += a[i];
  }
  st8(out + (size_t)idx * C + ch0, pack8f(acc, T()));
}

// ------------------------------------------------------------------------------------------------
// polar (B, V, Wp, R, C) polar maps, RAY-major (texel (r, w) at (w*R + r)*C: the layout the ray transformer emits); bev (B, Hb, Wb, C) residual; proj (B, V, 4, 4); aug_rev (B, 12)
// [A row-major, t: p' = p A + t]; cam_xy (B, V, 2); par = pc_range(6), input H, input W, r0, R.
// One 16-lane group per BEV cell.  The geometry of a cell - 10 height samples projected into each of the V cameras: two
// divides and a square root per (camera, sample) - does not depend on the channel: the group's lanes SHARE it (lane 2c + h
// takes half h of the samples of camera c, eight cameras per pass) and exchange the per-camera results by lane shuffles.
// (Round 1-4: every lane evaluated all V x 10 projections - ~3 600 VALU instructions per lane for 35 MB of traffic, 70 us.)
template <typename T>
__global__ __launch_bounds__(256) void polar_bev_sample_kernel(const T *__restrict__ polar, const T *__restrict__ bev,
                                                               const float *__restrict__ proj,
                                                               const float *__restrict__ aug_rev,
                                                               const float *__restrict__ cam_xy,
                                                               const float *__restrict__ par, T *__restrict__ out,
                                                               int B, int V, int R, int Wp, int Hb, int Wb, int C) {
  constexpr int ZS = 10;
  const int l16 = threadIdx.x & 15;
  const bool ch_ok = l16 * kChPerLane < C;                   // C < 128: the upper lanes only help with the geometry
  const int ch0 = ch_ok ? l16 * kChPerLane : 0;
  const long long total = (long long)B * Hb * Wb;
  const long long cell = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (cell >= total) return;                                 // (whole groups: the shuffles below stay inside a group)
  const int b = (int)(cell / (Hb * Wb)), ij = (int)(cell - (long long)b * Hb * Wb);
  const int i = ij / Wb, j = ij - i * Wb;
  const float x0 = par[0], y0 = par[1], z0 = par[2], x1 = par[3], y1 = par[4], z1 = par[5];
  const float in_h = par[6], in_w = par[7], r0 = par[8], Rf = par[9];
  // cell centre: the reference divides the column index by the ROW count and vice versa (:585-586); square maps
  const float bx = ((float)j + 0.5f) / (float)Hb * (x1 - x0) + x0;
  const float by = ((float)i + 0.5f) / (float)Wb * (y1 - y0) + y0;
  const float *A = aug_rev + b * 12;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  int vis = 0;
  for (int v0 = 0; v0 < V; v0 += 8) {
    const int v = v0 + (l16 >> 1), half = l16 & 1;
    float su = 0.f, sr = 0.f;
    int any = 0;
    if (v < V) {
      const float *M = proj + ((size_t)b * V + v) * 16;
      const float cx = cam_xy[((size_t)b * V + v) * 2], cy = cam_xy[((size_t)b * V + v) * 2 + 1];
#pragma unroll
      for (int kk = 0; kk < ZS / 2; ++kk) {
        const int k = half * (ZS / 2) + kk;
        const float bz = ((float)k + 0.5f) / (float)ZS * (z1 - z0) + z0;
        const float px = bx * A[0] + by * A[3] + bz * A[6] + A[9];
        const float py = bx * A[1] + by * A[4] + bz * A[7] + A[10];
        const float pz = bx * A[2] + by * A[5] + bz * A[8] + A[11];
        const float xc = M[0] * px + M[1] * py + M[2] * pz + M[3];
        const float yc = M[4] * px + M[5] * py + M[6] * pz + M[7];
        const float zc = M[8] * px + M[9] * py + M[10] * pz + M[11];
        const float zd = fmaxf(zc, 1e-5f);
        const float u = 2.f * (xc / zd / in_w) - 1.f, vv = 2.f * (yc / zd / in_h) - 1.f;
        any |= (int)((zc > 1e-5f) && u > -1.f && u < 1.f && vv > -1.f && vv < 1.f);
        su += u;
        const float dx = px - cx, dy = py - cy;
        const float rad = sqrtf(dx * dx + dy * dy);
        sr += fminf(fmaxf(2.f * (rad - r0) / Rf - 1.f, -1.f), 1.f);
      }
    }
    su += __shfl_xor(su, 1);                                  // the two halves of a camera's samples
    sr += __shfl_xor(sr, 1);
    any |= __shfl_xor(any, 1);
    const int nv = min(8, V - v0);
    for (int c = 0; c < nv; ++c) {                            // every lane walks the cameras of this pass
      const int any_c = __shfl(any, 2 * c, 16);
      const float su_c = __shfl(su, 2 * c, 16), sr_c = __shfl(sr, 2 * c, 16);
      if (!any_c) continue;
      ++vis;
      const float lx = su_c / (float)ZS, ly = sr_c / (float)ZS;
      const float fx = ((lx + 1.f) * (float)Wp - 1.f) * 0.5f, fy = ((ly + 1.f) * (float)R - 1.f) * 0.5f;
      if (ch_ok) bilinear8(polar + ((size_t)b * V + v0 + c) * R * Wp * C + ch0, R, Wp, C, R * C, fx, fy, 1.f, acc);   // ray-major
    }
  }
  if (!ch_ok) return;
  const float inv = 1.f / (float)(vis > 0 ? vis : 1);
  float res[8];
  unpack8(ld8(bev + (size_t)cell * C + ch0), res);
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = fmaf(acc[c], inv, res[c]);
  st8(out + (size_t)cell * C + ch0, pack8f(acc, T()));
}

// ------------------------------------------------------------------------------------------------
// q (N, Tq, q_rs) / k, v (N, S, kv_rs) / out (N, Tq, E): E = heads*16 columns starting at the given pointers (so a
// packed [Q|K|V] GEMM output can be passed with strides).  Workgroup = (sequence, group of 4 heads); a wave owns one
// head, its K/V rows sit in LDS (every lane reads the same row: broadcast) and a lane runs the online softmax of
// one query row, 4 keys per rescale.
template <typename T>
__global__ __launch_bounds__(256) void mha_small_kernel(const T *__restrict__ q, int q_rs, const T *__restrict__ k,
                                                        const T *__restrict__ v, int kv_rs, T *__restrict__ out,
                                                        int out_rs, int Tq, int S, int heads, float scale_log2) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const int n = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int head = blockIdx.y * 4 + wave;
  if (head >= heads) return;
  T *ks = smem + (size_t)wave * S * 32, *vs = ks + (size_t)S * 16;
  const T *kb = k + (size_t)n * S * kv_rs + head * 16, *vb = v + (size_t)n * S * kv_rs + head * 16;
  for (int t = lane; t < S * 2; t += 64) {                       // S rows x two 8-channel halves
    const int s = t >> 1, h8 = (t & 1) * 8;
    st8(ks + s * 16 + h8, ld8(kb + (size_t)s * kv_rs + h8));
    st8(vs + s * 16 + h8, ld8(vb + (size_t)s * kv_rs + h8));
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();                               // the wave only reads what it wrote itself
  for (int t0 = 0; t0 < Tq; t0 += 64) {
    const int t = t0 + lane;
    if (t >= Tq) break;
    float qv[16], acc[16];
    {
      float a[8], b8[8];
      const T *qp = q + ((size_t)n * Tq + t) * q_rs + head * 16;
      unpack8(ld8(qp), a);
      unpack8(ld8(qp + 8), b8);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        qv[i] = a[i] * scale_log2;
        qv[8 + i] = b8[i] * scale_log2;
        acc[i] = acc[8 + i] = 0.f;
      }
    }
    float m = -INFINITY, l = 0.f;
    for (int s0 = 0; s0 < S; s0 += 4) {
      float sc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = min(s0 + u, S - 1);
        float a[8], b8[8];
        unpack8(ld8(ks + s * 16), a);
        unpack8(ld8(ks + s * 16 + 8), b8);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) d = fmaf(qv[i], a[i], fmaf(qv[8 + i], b8[i], d));
        sc[u] = (s0 + u < S) ? d : -INFINITY;
      }
      const float mn = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), m);
      const float corr = exp2f(m - mn);
      l *= corr;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] *= corr;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = min(s0 + u, S - 1);
        const float p = exp2f(sc[u] - mn);
        l += p;
        float a[8], b8[8];
        unpack8(ld8(vs + s * 16), a);
        unpack8(ld8(vs + s * 16 + 8), b8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = fmaf(p, a[i], acc[i]);
          acc[8 + i] = fmaf(p, b8[i], acc[8 + i]);
        }
      }
      m = mn;
    }
    const float inv = 1.f / l;
    float o0[8], o1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o0[i] = acc[i] * inv;
      o1[i] = acc[8 + i] * inv;
    }
    T *op = out + ((size_t)n * Tq + t) * out_rs + head * 16;
    st8(op, pack8f(o0, T()));
    st8(op + 8, pack8f(o1, T()));
  }
}

// ------------------------------------------------------------------------------------------------
// Matrix-core form of the short-sequence attention (fp16, S <= 16 * NT16 <= 128): workgroup = (sequence, 4 heads),
// wave = head.  The head's K and V rows are staged in LDS ([head][key][K16 | V16], 64-B rows); per 16 queries:
//   S^T = K . Q^T   one 16x16x16 MFMA per 16-key tile (A = K rows from LDS, B = Q^T straight from global),
//   softmax per query = per lane column (two cross-row exchanges), scores never leave registers,
//   O^T = V^T . P^T with the exp registers as the B operand and V^T from ds_read_b64_tr_b16.
typedef _Float16 sh4 __attribute__((ext_vector_type(4)));
typedef __fp16 shv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float sf4 __attribute__((ext_vector_type(4)));

template <int NT16>
__global__ __launch_bounds__(256) void mha_small_mfma_kernel(const __half *__restrict__ q, int q_rs,
                                                             const __half *__restrict__ k,
                                                             const __half *__restrict__ v, int kv_rs,
                                                             __half *__restrict__ out, int out_rs, int Tq, int S,
                                                             int heads, float scale_log2) {
  constexpr int KC = NT16 * 16;
  __shared__ __align__(16) unsigned char lds[4 * KC * 64];
  const int n = blockIdx.x, head0 = blockIdx.y * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  // stage: 16-B piece p of a key: p >> 3 = K | V, (p >> 1) & 3 = local head, p & 1 = half of the head's 16 dims
  for (int e = tid; e < KC * 16; e += 256) {
    const int key = e >> 4, p = e & 15;
    const int isv = p >> 3, hh = (p >> 1) & 3, half8 = p & 1;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (key < S && head0 + hh < heads)
      val = *reinterpret_cast<const uint4 *>((isv ? v : k) + ((size_t)n * S + key) * kv_rs + (head0 + hh) * 16 + half8 * 8);
    *reinterpret_cast<uint4 *>(lds + ((hh * KC + key) * 64 + isv * 32 + half8 * 16)) = val;
  }
  __syncthreads();
  const int h = head0 + wave;
  if (h >= heads) return;
  const unsigned char *hb = lds + (size_t)wave * KC * 64;
  const int nqg = (Tq + 15) / 16;
  for (int qg = 0; qg < nqg; ++qg) {
    const int qi = qg * 16 + i;
    const int qc = qi < Tq ? qi : Tq - 1;
    const sh4 qf = *reinterpret_cast<const sh4 *>(q + ((size_t)n * Tq + qc) * q_rs + h * 16 + 4 * g);
    sf4 sc[NT16];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
      const sh4 kf = *reinterpret_cast<const sh4 *>(hb + (16 * t + i) * 64 + g * 8);
      sf4 c = __builtin_amdgcn_mfma_f32_16x16x16f16(kf, qf, sf4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = (16 * t + 4 * g + r < S) ? c[r] * scale_log2 : -INFINITY;   // key 4g + r of the tile
        c[r] = x;
        m = fmaxf(m, x);
      }
      sc[t] = c;
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
    sf4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
      sh4 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = exp2f(sc[t][r] - m);
        l += e;
        pf[r] = (_Float16)e;
      }
      const shv4 vt = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (shv4 __attribute__((address_space(3))) *)(hb + (16 * t + 4 * g + (i >> 2)) * 64 + 32 + (i & 3) * 8));
      sh4 vf;
      vf[0] = vt[0]; vf[1] = vt[1]; vf[2] = vt[2]; vf[3] = vt[3];
      acc = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf, acc, 0, 0, 0);
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (qi < Tq) {
      const float inv = 1.f / l;
      sh4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (_Float16)(acc[r] * inv);               // O^T[dim 4g + r][query i]
      *reinterpret_cast<sh4 *>(out + ((size_t)n * Tq + qi) * out_rs + h * 16 + 4 * g) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// out = LayerNorm(x (+ res)) * gamma + beta over C <= 128 channels; one 16-lane group per token, statistics in
// fp32 (mean, then centred variance), DPP row reductions.  Fuses the residual add that precedes every post-norm.
template <typename T, bool HAS_RES>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const T *__restrict__ x, const T *__restrict__ res,
                                                            const T *__restrict__ gamma, const T *__restrict__ beta,
                                                            T *__restrict__ out, long long n_tokens, int C, float eps) {
  const int l16 = threadIdx.x & 15, ch0 = l16 * kChPerLane;
  const bool ok = ch0 < C;
  float g[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) g[i] = b[i] = 0.f;
  if (ok) {
    unpack8(ld8(gamma + ch0), g);
    unpack8(ld8(beta + ch0), b);
  }
  const float invC = 1.f / (float)C;
  const long long stride = (long long)gridDim.x * 16;
  for (long long t = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); t < n_tokens; t += stride) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
    if (ok) {
      unpack8(ld8(x + t * C + ch0), v);
      if (HAS_RES) {
        float r[8];
        unpack8(ld8(res + t * C + ch0), r);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += r[i];
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    const float mean = row16_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float d = ok ? v[i] - mean : 0.f;
      q = fmaf(d, d, q);
    }
    const float rstd = rsqrtf(row16_sum(q) * invC + eps);
    if (ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaf((v[i] - mean) * rstd, g[i], b[i]);
      st8(out + t * C + ch0, pack8f(v, T()));
    }
  }
}

}  // namespace pp
}  // namespace di

extern "C" {

int di_add_layernorm_fwd(const void *x, const void *res, const void *gamma, const void *beta, void *out,
                         long long n_tokens, int C, float eps, int dtype, void *stream) {
  DI_REQUIRE(n_tokens >= 0 && C > 0 && C % 8 == 0 && C <= 128, "C=%d must be a multiple of 8, <= 128", C);
  if (n_tokens == 0) return DI_OK;
  const long long want = (n_tokens + 15) / 16;
  const dim3 g((unsigned)(want < 256 * 32 ? want : 256 * 32)), blk(256);
  hipStream_t s = (hipStream_t)stream;
#define DI_LN(TT, RR)                                                                                             \
  hipLaunchKernelGGL((di::pp::add_layernorm_kernel<TT, RR>), g, blk, 0, s, (const TT *)x, (const TT *)res,          \
                     (const TT *)gamma, (const TT *)beta, (TT *)out, n_tokens, C, eps)
  if (dtype == DI_F16) { if (res) DI_LN(__half, true); else DI_LN(__half, false); }
  else if (dtype == DI_F32) { if (res) DI_LN(float, true); else DI_LN(float, false); }
  else { di::set_error("unsupported dtype %d", dtype); return DI_ERR_ARG; }
#undef DI_LN
  return di::check_launch("add_layernorm_fwd");
}

int di_ms_deform_attn_fwd(const void *value, const void *offsets, int off_row_stride, const void *logits,
                          int logit_row_stride, const float *ref, int ref_shared, void *out, int bs, int nq,
                          int n_levels, int n_points, const int32_t *level_hw, int dtype, void *stream) {
  DI_REQUIRE(bs > 0 && nq > 0, "bad deformable attention shape");
  DI_REQUIRE((n_levels == 1 || n_levels == 2) && n_points == 4, "levels %d / points %d unsupported (1|2 levels, 4 points)",
             n_levels, n_points);
  {   // the offsets / logits of one head are fetched with 8- / 16-byte loads
    const size_t esz = dtype == DI_F16 ? 2 : 4;
    const size_t la = (size_t)n_levels * 4 * esz < 16 ? (size_t)n_levels * 4 * esz : 16;
    DI_REQUIRE(((uintptr_t)offsets % 16) == 0 && (off_row_stride * esz) % 16 == 0 && ((uintptr_t)logits % la) == 0 &&
                   (logit_row_stride * esz) % la == 0,
               "offsets / logits rows must be 16-byte aligned (packed projection of 8*L*P*3 columns)");
  }
  di::pp::Levels lv;
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
#define DI_MSDA(TT, LL)                                                                                              \
  hipLaunchKernelGGL((di::pp::ms_deform_attn_kernel<TT, LL, 4>), grid, blk, 0, s, (const TT *)value,                 \
                     (const TT *)offsets, off_row_stride, (const TT *)logits, logit_row_stride, ref, ref_shared,     \
                     (TT *)out, bs, nq, S, lv)
  if (dtype == DI_F16) { if (n_levels == 1) DI_MSDA(__half, 1); else DI_MSDA(__half, 2); }
  else if (dtype == DI_F32) { if (n_levels == 1) DI_MSDA(float, 1); else DI_MSDA(float, 2); }
  else { di::set_error("unsupported dtype %d", dtype); return DI_ERR_ARG; }
#undef DI_MSDA
  return di::check_launch("ms_deform_attn_fwd");
}

int di_ms_deform_attn_hm_fwd(const void *value_hm, const void *offsets, int off_row_stride, const void *logits,
                             int logit_row_stride, const float *ref, int ref_shared, void *out, int bs, int nq,
                             int n_levels, int n_points, const int32_t *level_hw, void *stream) {
  DI_REQUIRE(bs > 0 && nq > 0, "bad deformable attention shape");
  DI_REQUIRE((n_levels == 1 || n_levels == 2) && n_points == 4, "levels %d / points %d unsupported (1|2 levels, 4 points)",
             n_levels, n_points);
  DI_REQUIRE(((uintptr_t)offsets % 4) == 0 && (off_row_stride % 2) == 0 && value_hm && logits && ref && out,
             "offset rows must be 4-byte aligned");
  di::pp::Levels lv;
  lv.n = n_levels;
  int S = 0;
  for (int l = 0; l < n_levels; ++l) {
    lv.h[l] = level_hw[2 * l];
    lv.w[l] = level_hw[2 * l + 1];
    DI_REQUIRE(lv.h[l] > 0 && lv.w[l] > 0, "bad level shape");
    lv.start[l] = S;
    S += lv.h[l] * lv.w[l];
  }
  DI_REQUIRE(S < (1 << 30), "%d texels per map exceed the 30-bit index", S);
  const long long rows = (long long)bs * nq;
  const dim3 grid((unsigned)((rows + 7) / 8)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  const bool timed = di::take_launch_events(ev0, ev1);       // measurement: the dispatch's own begin / end time stamps
#define DI_MSDA_HM(LL)                                                                                                         \
  do {                                                                                                                         \
    if (timed)                                                                                                                 \
      hipExtLaunchKernelGGL((di::pp::ms_deform_attn_hm_kernel<LL, 4>), grid, blk, 0, s, ev0, ev1, 0, (const __half *)value_hm, \
                            (const __half *)offsets, off_row_stride, (const __half *)logits, logit_row_stride, ref,           \
                            ref_shared, (__half *)out, bs, nq, S, lv);                                                         \
    else                                                                                                                       \
      hipLaunchKernelGGL((di::pp::ms_deform_attn_hm_kernel<LL, 4>), grid, blk, 0, s, (const __half *)value_hm,                 \
                         (const __half *)offsets, off_row_stride, (const __half *)logits, logit_row_stride, ref, ref_shared,   \
                         (__half *)out, bs, nq, S, lv);                                                                        \
  } while (0)
  if (n_levels == 1) DI_MSDA_HM(1);
  else DI_MSDA_HM(2);
#undef DI_MSDA_HM
  return di::check_launch("ms_deform_attn_hm_fwd");
}

int di_grid_gather_fwd(const void *feat, const float *grid, const void *add, void *out, int n_grids, int n_points,
                       int grids_per_feat, int H, int W, int C, int dtype, void *stream) {
  DI_REQUIRE(n_grids > 0 && n_points > 0 && grids_per_feat > 0 && H > 0 && W > 0, "bad grid gather shape");
  DI_REQUIRE(C > 0 && C % 8 == 0 && C <= 128, "C=%d must be a multiple of 8, <= 128", C);
  const long long total = (long long)n_grids * n_points;
  const dim3 g((unsigned)((total + 15) / 16)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DI_F16)
    hipLaunchKernelGGL((di::pp::grid_gather_kernel<__half>), g, blk, 0, s, (const __half *)feat, grid,
                       (const __half *)add, (__half *)out, n_grids, n_points, grids_per_feat, H, W, C);
  else if (dtype == DI_F32)
    hipLaunchKernelGGL((di::pp::grid_gather_kernel<float>), g, blk, 0, s, (const float *)feat, grid,
                       (const float *)add, (float *)out, n_grids, n_points, grids_per_feat, H, W, C);
  else { di::set_error("unsupported dtype %d", dtype); return DI_ERR_ARG; }
  return di::check_launch("grid_gather_fwd");
}

int di_polar_bev_sample_fwd(const void *polar, const void *bev, const float *proj, const float *aug_rev,
                            const float *cam_xy, const float *params, void *out, int B, int V, int R, int Wp, int Hb,
                            int Wb, int C, int dtype, void *stream) {
  DI_REQUIRE(B > 0 && V > 0 && R > 0 && Wp > 0 && Hb > 0 && Wb > 0, "bad polar sample shape");
  DI_REQUIRE(Hb == Wb, "square BEV maps only (the reference swaps the axis sizes, fusion_transformerv4.py:585-586)");
  DI_REQUIRE(C > 0 && C % 8 == 0 && C <= 128, "C=%d must be a multiple of 8, <= 128", C);
  const long long total = (long long)B * Hb * Wb;
  const dim3 g((unsigned)((total + 15) / 16)), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DI_F16)
    hipLaunchKernelGGL((di::pp::polar_bev_sample_kernel<__half>), g, blk, 0, s, (const __half *)polar,
                       (const __half *)bev, proj, aug_rev, cam_xy, params, (__half *)out, B, V, R, Wp, Hb, Wb, C);
  else if (dtype == DI_F32)
    hipLaunchKernelGGL((di::pp::polar_bev_sample_kernel<float>), g, blk, 0, s, (const float *)polar,
                       (const float *)bev, proj, aug_rev, cam_xy, params, (float *)out, B, V, R, Wp, Hb, Wb, C);
  else { di::set_error("unsupported dtype %d", dtype); return DI_ERR_ARG; }
  return di::check_launch("polar_bev_sample_fwd");
}

int di_mha_small_fwd(const void *q, int q_row_stride, const void *k, const void *v, int kv_row_stride, void *out,
                     int out_row_stride, int n_seq, int Tq, int S, int num_heads, int head_dim, float scale, int dtype,
                     void *stream) {
  DI_REQUIRE(head_dim == 16, "head_dim %d unsupported (16 only)", head_dim);
  DI_REQUIRE(n_seq > 0 && Tq > 0 && S > 0 && num_heads > 0, "bad attention shape");
  DI_REQUIRE(q_row_stride % 8 == 0 && kv_row_stride % 8 == 0 && out_row_stride % 8 == 0, "row strides must be 16-B aligned");
  const size_t esz = dtype == DI_F16 ? 2 : 4;
  const size_t lds = (size_t)4 * S * 32 * esz;
  DI_REQUIRE(lds <= 160 * 1024, "S=%d too long for the short-sequence kernel (K/V of 4 heads must fit the 160 KB LDS)", S);
  if (lds > 64 * 1024) {   // above the default dynamic-LDS cap: opt in (idempotent, no stream interaction)
    const void *fn = dtype == DI_F16 ? (const void *)di::pp::mha_small_kernel<__half>
                                     : (const void *)di::pp::mha_small_kernel<float>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      di::set_error("cannot reserve %zu B of LDS", lds);
      return DI_ERR_ARG;
    }
  }
  const dim3 g(n_seq, (num_heads + 3) / 4), blk(256);
  hipStream_t s = (hipStream_t)stream;
  const float sl2 = scale * 1.4426950408889634f;
  if (dtype == DI_F16 && S <= 128) {                 // matrix-core kernel, K/V of 4 heads in static LDS
    if (S <= 64)
      hipLaunchKernelGGL((di::pp::mha_small_mfma_kernel<4>), g, blk, 0, s, (const __half *)q, q_row_stride,
                         (const __half *)k, (const __half *)v, kv_row_stride, (__half *)out, out_row_stride, Tq, S,
                         num_heads, sl2);
    else
      hipLaunchKernelGGL((di::pp::mha_small_mfma_kernel<8>), g, blk, 0, s, (const __half *)q, q_row_stride,
                         (const __half *)k, (const __half *)v, kv_row_stride, (__half *)out, out_row_stride, Tq, S,
                         num_heads, sl2);
    return di::check_launch("mha_small_fwd");
  }
  if (dtype == DI_F16)
    hipLaunchKernelGGL((di::pp::mha_small_kernel<__half>), g, blk, lds, s, (const __half *)q, q_row_stride,
                       (const __half *)k, (const __half *)v, kv_row_stride, (__half *)out, out_row_stride, Tq, S,
                       num_heads, sl2);
  else if (dtype == DI_F32)
    hipLaunchKernelGGL((di::pp::mha_small_kernel<float>), g, blk, lds, s, (const float *)q, q_row_stride,
                       (const float *)k, (const float *)v, kv_row_stride, (float *)out, out_row_stride, Tq, S,
                       num_heads, sl2);
  else { di::set_error("unsupported dtype %d", dtype); return DI_ERR_ARG; }
  return di::check_launch("mha_small_fwd");
}

}  // extern "C"
