// MMPI decoder kernels for gfx950 (reference models/dense_heads/deepinteraction_decoder.py:201-313
// and models/utils/decoder_utils.py:632-841):
//   heatmap_nms      sigmoid-average of the two dense heat maps + 3x3 local-max mask (:225-238)
//   query_geometry   box decode (transfusion_bbox_coder.py:39-91) + 8 corners + projection of
//                    centre/corners into every camera + on-image mask + circumscribed rectangles
//                    (decoder_utils.py:666-738, :804-819) - one thread per (sample, query), all
//                    views at once, no host loop, no device->host sync
//   roi_align        detectron2 ROIAlign(aligned=True, 7x7, 2x2 samples) on channels-last maps,
//                    all (sample, view, query) RoIs in one launch, output (R,49,C)
//   mha_decode       the 200 x 32400 multi-head cross attention (decoder_utils.py:101-103) as a
//                    split-KV online-softmax pass + a combine pass; scores never touch HBM and
//                    the reference's averaged (B,200,32400) weight output is not produced
#include "di_common.h"
#include <type_traits>

namespace di {

// ---------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T>
__global__ __launch_bounds__(256) void heatmap_nms_kernel(const T *__restrict__ a,
                                                          const T *__restrict__ b,
                                                          float *__restrict__ out, int B, int Cc, int H,
                                                          int W, int ksize, unsigned k1_mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Cc * H * W) return;
  const int x = i % W, y = (i / W) % H, c = (i / (W * H)) % Cc;
  const size_t plane = (size_t)(i / (W * H)) * H * W;
  auto heat = [&](int yy, int xx) {
    const size_t j = plane + (size_t)yy * W + xx;
    return (sigmoidf_((float)a[j]) + sigmoidf_((float)b[j])) * 0.5f;
  };
  const float v = heat(y, x);
  float res = v;
  if (!((k1_mask >> c) & 1u)) {
    const int pad = ksize / 2;
    if (y < pad || y >= H - pad || x < pad || x >= W - pad) {
      res = 0.f;  // local_max is only filled in the interior (:230): v == 0 never holds for a sigmoid
    } else {
      float m = v;
      for (int dy = -pad; dy <= pad; ++dy)
        for (int dx = -pad; dx <= pad; ++dx) m = fmaxf(m, heat(y + dy, x + dx));
      res = (v == m) ? v : 0.f;
    }
  }
  out[i] = res;
}

// ---------------------------------------------------------------------------------
struct QGeomParams {
  float cell;         // out_size_factor * voxel_size[0]
  float x0, y0;       // pc_range[0], pc_range[1]
  float bev_cell;     // voxel_size[0] * out_size_factor of the bbox coder (:810)
  float dim_scale;    // 1 (image block) or 2 (point block, :807)
  int ld;             // row stride of the (B,k,ld) prediction tensors (>= Q: a column window of a wider tensor)
};

// res tensors are (B,k,Q) float32.  per_sample: [w, h, flip, orig_w, crop_x, crop_y] x B.
__global__ __launch_bounds__(256) void query_geometry_kernel(
    const float *__restrict__ center, const float *__restrict__ height, const float *__restrict__ dim,
    const float *__restrict__ rot, const float *__restrict__ proj, const float *__restrict__ aug_rev,
    const float *__restrict__ per_sample, int32_t *__restrict__ on_img, float *__restrict__ rect_img,
    float *__restrict__ rect_bev, int B, int Q, int V, QGeomParams P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Q) return;
  const int b = i / Q, q = i - b * Q;
  const float cx = center[(b * 2 + 0) * P.ld + q] * P.cell + P.x0;   // :666, coder :61-62
  const float cy = center[(b * 2 + 1) * P.ld + q] * P.cell + P.y0;
  const float hz = height[b * P.ld + q];
  const float dx = __expf(dim[(b * 3 + 0) * P.ld + q]) , dy = __expf(dim[(b * 3 + 1) * P.ld + q]);
  const float dz = __expf(dim[(b * 3 + 2) * P.ld + q]);
  const float zb = hz - dz * 0.5f;                                  // gravity -> bottom centre (:68)
  const float yaw = atan2f(rot[(b * 2 + 0) * P.ld + q], rot[(b * 2 + 1) * P.ld + q]);
  const float sn = sinf(yaw), cs = cosf(yaw);
  float px[9], py[9], pz[9];
  px[0] = cx; py[0] = cy; pz[0] = hz;                               // query centre (gravity height, :667)
  float bxmin = INFINITY, bymin = INFINITY, bxmax = -INFINITY, bymax = -INFINITY;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // LiDARInstance3DBoxes.corners (mmdet3d 0.17.1): unit cube minus (0.5,0.5,0), scaled,
    // rotated about z with p @ [[c,-s,0],[s,c,0],[0,0,1]], translated.  Order is irrelevant
    // to the min/max consumers.
    const float ux = ((k >> 2) & 1) - 0.5f, uy = ((k >> 1) & 1) - 0.5f, uz = (float)(k & 1);
    const float lx = ux * dx * P.dim_scale, ly = uy * dy * P.dim_scale, lz = uz * dz * P.dim_scale;
    const float wx = lx * cs + ly * sn + cx;
    const float wy = -lx * sn + ly * cs + cy;
    px[k + 1] = wx; py[k + 1] = wy; pz[k + 1] = lz + zb;
    const float gx = (wx - P.x0) / P.bev_cell, gy = (wy - P.x0) / P.bev_cell;   // :810 (pc_range[0] for both)
    bxmin = fminf(bxmin, gx); bxmax = fmaxf(bxmax, gx);
    bymin = fminf(bymin, gy); bymax = fmaxf(bymax, gy);
  }
  if (rect_bev) {
    float *r = rect_bev + (size_t)i * 4;
    r[0] = bxmin; r[1] = bymin; r[2] = bxmax; r[3] = bymax;
  }
  if (!rect_img) return;
  const float *A = aug_rev + b * 12;
  const float *S = per_sample + b * 6;
  const float w = S[0], h = S[1], flip = S[2], orig_w = S[3], crop_x = S[4], crop_y = S[5];
#pragma unroll
  for (int k = 0; k < 9; ++k) {                                     // un-augment (:692)
    const float x = px[k], y = py[k], z = pz[k];
    px[k] = x * A[0] + y * A[3] + z * A[6] + A[9];
    py[k] = x * A[1] + y * A[4] + z * A[7] + A[10];
    pz[k] = x * A[2] + y * A[5] + z * A[8] + A[11];
  }
  for (int v = 0; v < V; ++v) {
    const float *M = proj + ((size_t)b * V + v) * 16;
    float xmin = INFINITY, ymin = INFINITY, xmax = -INFINITY, ymax = -INFINITY;
    int on = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float X = M[0] * px[k] + M[1] * py[k] + M[2] * pz[k] + M[3];
      const float Y = M[4] * px[k] + M[5] * py[k] + M[6] * pz[k] + M[7];
      const float Z = fmaxf(M[8] * px[k] + M[9] * py[k] + M[10] * pz[k] + M[11], 1e-5f);   // :699
      float u = X / Z - crop_x, t = Y / Z - crop_y;                 // scale factor 1, crop offset (:705-706)
      if (flip != 0.f) u = orig_w - u;                              // :710-714
      if (k == 0) {
        on = (u > 0.f && u < w && t > 0.f && t < h) ? 1 : 0;        // strict (:723)
      } else {
        xmin = fminf(xmin, u); xmax = fmaxf(xmax, u);
        ymin = fminf(ymin, t); ymax = fmaxf(ymax, t);
      }
    }
    const size_t o = ((size_t)b * V + v) * Q + q;
    on_img[o] = on;
    float *r = rect_img + o * 4;
    r[0] = xmin; r[1] = ymin; r[2] = xmax; r[3] = ymax;             // unclipped, input pixels (:730-738)
  }
}

// ---------------------------------------------------------------------------------
// ROIAlign(aligned=True).  One 16-lane group per (roi, bin): 2x2 samples x 4 corners of
// 256 B coalesced rows.  rois (R,5) = [map index, x0, y0, x1, y1] in input pixels.
template <typename T>
__device__ __forceinline__ void roi_sample8(const T *__restrict__ map, int H, int W, int C, float y,
                                            float x, int ch0, float (&acc)[8]) {
  if (!(y >= -1.f && y <= (float)H && x >= -1.f && x <= (float)W)) return;   // also rejects NaN/inf
  y = fmaxf(y, 0.f);
  x = fmaxf(x, 0.f);
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
  const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
  const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  float f[8];
  unpack8(ld8(map + ((size_t)yl * W + xl) * C + ch0), f);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = fmaf(w1, f[i], acc[i]);
  unpack8(ld8(map + ((size_t)yl * W + xh) * C + ch0), f);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = fmaf(w2, f[i], acc[i]);
  unpack8(ld8(map + ((size_t)yh * W + xl) * C + ch0), f);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = fmaf(w3, f[i], acc[i]);
  unpack8(ld8(map + ((size_t)yh * W + xh) * C + ch0), f);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = fmaf(w4, f[i], acc[i]);
}

struct HLOut {};      // output tag: split fp16 pair
template <typename T, bool FULLC, typename TO = T>
__global__ __launch_bounds__(256) void roi_align_kernel(const T *__restrict__ feat,
                                                        const float *__restrict__ rois,
                                                        typename std::conditional<std::is_same<TO, HLOut>::value, __half, TO>::type *__restrict__ out, int R, int N, int H, int W,
                                                        int C, float scale) {
  constexpr int PB = 7, G = 2;
  const int l16 = threadIdx.x & 15;
  const bool ch_ok = FULLC || l16 * kChPerLane < C;
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
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    if (ch_ok) {
      const T *map = feat + (size_t)n * H * W * C;
#pragma unroll
      for (int iy = 0; iy < G; ++iy) {
        const float y = sh + ph * bh + (iy + 0.5f) * bh / G;
#pragma unroll
        for (int ix = 0; ix < G; ++ix) {
          const float x = sw + pw * bw + (ix + 0.5f) * bw / G;
          roi_sample8(map, H, W, C, y, x, ch0, acc);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] *= 1.f / (G * G);
      if constexpr (std::is_same<TO, HLOut>::value) {        // [hi C | lo C] fp16: x = hi + lo / 2048 (csrc/token32.hip)
        __half *o = reinterpret_cast<__half *>(out) + (size_t)g * 2 * C + ch0;
        Pack8<__half> ph, pl;
        __half2 *hh = reinterpret_cast<__half2 *>(&ph.r), *hl = reinterpret_cast<__half2 *>(&pl.r);
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
          const __half a0 = __float2half(acc[2 * i2]), a1 = __float2half(acc[2 * i2 + 1]);
          hh[i2] = __halves2half2(a0, a1);
          hl[i2] = __floats2half2_rn((acc[2 * i2] - __half2float(a0)) * 2048.f, (acc[2 * i2 + 1] - __half2float(a1)) * 2048.f);
        }
        st8(o, ph);
        st8(o + C, pl);
      } else {
        st8(out + (size_t)g * C + ch0, pack8f(acc, TO()));
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Multi-head decode attention, head dim 16.  q (B,Q,E), kv (B,S,2E) = [K | V], E = H*16.
// Pass 1: grid (chunks, H, B); the workgroup stages its KV chunk of one head in LDS
// (every thread then reads the same key: LDS broadcast) and each thread runs the online
// softmax of one query over the chunk.  Pass 2 merges the chunk states.
constexpr int kHD = 16;       // head dim
constexpr int kChunk = 512;   // keys per workgroup

template <typename T>
__global__ __launch_bounds__(256) void mha_decode_partial_kernel(const T *__restrict__ q,
                                                                 const T *__restrict__ kv,
                                                                 float *__restrict__ part, int B, int Q,
                                                                 int S, int Hh, float scale) {
  __shared__ float sk[kChunk][kHD];
  __shared__ float sv[kChunk][kHD];
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int nchunk = gridDim.x;
  const int E = Hh * kHD;
  const int s0 = chunk * kChunk;
  const int ns = min(kChunk, S - s0);
  // stage: thread t copies 8 channels of K or V of key t/4 .. (coalesced 16 B fp16 / 32 B fp32)
  for (int i = threadIdx.x; i < ns * 4; i += blockDim.x) {
    const int key = i >> 2, part4 = i & 3;            // part4: 0,1 -> K halves; 2,3 -> V halves
    const int isv = part4 >> 1, half = part4 & 1;
    const T *src = kv + ((size_t)b * S + s0 + key) * 2 * E + isv * E + h * kHD + half * 8;
    float f[8];
    unpack8(ld8(src), f);
    float *dst = (isv ? sv[key] : sk[key]) + half * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = f[j];
  }
  __syncthreads();
  for (int qi = threadIdx.x; qi < Q; qi += blockDim.x) {
    float qv[kHD];
    {
      const T *src = q + ((size_t)b * Q + qi) * E + h * kHD;
      float f[8];
      unpack8(ld8(src), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) qv[j] = f[j] * scale;
      unpack8(ld8(src + 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) qv[8 + j] = f[j] * scale;
    }
    float m = -INFINITY, l = 0.f, acc[kHD];
#pragma unroll
    for (int j = 0; j < kHD; ++j) acc[j] = 0.f;
    for (int s = 0; s < ns; s += 4) {
      float sc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float d = 0.f;
        if (s + u < ns) {
#pragma unroll
          for (int j = 0; j < kHD; ++j) d = fmaf(qv[j], sk[s + u][j], d);
        } else {
          d = -INFINITY;
        }
        sc[u] = d;
      }
      const float mn = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), m);
      const float a = __expf(m - mn);
      l *= a;
#pragma unroll
      for (int j = 0; j < kHD; ++j) acc[j] *= a;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float p = __expf(sc[u] - mn);   // padded slots: exp(-inf) = 0
        l += p;
        if (s + u < ns) {
#pragma unroll
          for (int j = 0; j < kHD; ++j) acc[j] = fmaf(p, sv[s + u][j], acc[j]);
        }
      }
      m = mn;
    }
    float *dst = part + ((((size_t)b * Hh + h) * nchunk + chunk) * Q + qi) * (kHD + 2);
    dst[0] = m;
    dst[1] = l;
#pragma unroll
    for (int j = 0; j < kHD; ++j) dst[2 + j] = acc[j];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mha_decode_combine_kernel(const float *__restrict__ part,
                                                                 T *__restrict__ out, int B, int Q,
                                                                 int Hh, int nchunk) {
  // one 16-lane group per (b, h, q); lane = channel of the head
  const int g = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
  const int d = threadIdx.x & 15;
  if (g >= B * Hh * Q) return;
  const int qi = g % Q, h = (g / Q) % Hh, b = g / (Q * Hh);
  const float *base = part + (((size_t)b * Hh + h) * nchunk) * Q * (kHD + 2) + (size_t)qi * (kHD + 2);
  float M = -INFINITY;
  for (int c = 0; c < nchunk; ++c) M = fmaxf(M, base[(size_t)c * Q * (kHD + 2)]);
  float L = 0.f, O = 0.f;
  for (int c = 0; c < nchunk; ++c) {
    const float *p = base + (size_t)c * Q * (kHD + 2);
    const float w = __expf(p[0] - M);
    L += p[1] * w;
    O += p[2 + d] * w;
  }
  out[((size_t)b * Q + qi) * Hh * kHD + h * kHD + d] = (T)(O / L);
}


// ---------------------------------------------------------------------------------
// fp16 matrix-core form (head dim 16).  grid (key ranges, query splits, B), 16 wavefronts (two per head):
//   * the workgroup walks its key range in STAGES of 128 keys, double buffered: the K and V rows of ALL heads of the
//     next stage are in flight (registers) while the current stage is multiplied; LDS layout [head][key][K16 | V16]
//     in 80-B rows (conflict-free 8-B fragment reads), coalesced 512-B key rows from HBM;
//   * a wavefront owns one head and up to two query groups of 16 (four waves per SIMD hide the LDS and MFMA latencies;
//     the query splits share the groups); per stage and
//     group:  S^T = K . Q^T  one 16x16x16 MFMA per 16-key tile (A = K rows from LDS, B = Q^T in registers), scores in
//     the exp2 domain (scale * log2 e folded into one multiply), running maximum per query = per lane column (two
//     cross-row exchanges per stage), O^T += V^T . P^T with the exp registers as B operand and V^T from
//     ds_read_b64_tr_b16; the (m, l, O) state of a (head, query) stays in registers across the stages;
//   * one partial state per (head, query, key RANGE) - 64 ranges at 32 400 keys, a quarter of the one-state-per-256-
//     keys layout this replaces - stored range-contiguous so that the combine pass reads 4.6 KB rows.
namespace mh {
constexpr int KS = 128;        // keys per stage
constexpr int NT16 = KS / 16;  // 16-key tiles per stage
constexpr int KROW = 48;       // LDS bytes per (head, key) of K: 32 + 16 pad - conflict-free 8-B fragment reads
constexpr int VROW = 32;       // ... of V: the transposing reads take 8 rows x 32 B = all 64 banks, no padding
constexpr int ROWB = KROW + VROW;
constexpr int WPH = 2;         // wavefronts per head
constexpr int MAXQG = 2;       // query groups of 16 per wavefront (WPH * MAXQG per workgroup and head)
constexpr int NTH = 1024;
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f4 __attribute__((ext_vector_type(4)));
}  // namespace mh

// max of three without the canonicalisation (v_max x, x) that fmaxf puts in front of every operand - in a kernel that
// is bound by VALU issue the running maximum would cost 8 instructions per 16 x 16 tile instead of 2
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
template <int HH>   // 8: the head count as a constant (index arithmetic of the staging becomes shifts); 0: run time
__global__ __launch_bounds__(mh::NTH) void mha_decode_mfma_kernel(const __half *__restrict__ q,
                                                                  const __half *__restrict__ kv,
                                                                  float *__restrict__ part, int B, int Q, int S,
                                                                  int Hh_, float scale2, int range_keys) {
  using namespace mh;
  const int Hh = HH > 0 ? HH : Hh_;
  extern __shared__ __align__(16) unsigned char lds[];   // 2 x Hh * KS * ROWB bytes
  const int range = blockIdx.x, b = blockIdx.z, nrange = gridDim.x;
  const int E = Hh * kHD;
  const int r0 = range * range_keys, r1 = min(r0 + range_keys, S);
  const int nstage = (r1 - r0 + KS - 1) / KS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int stage_bytes = Hh * KS * ROWB;
  // staging: 16-B piece p of a key row: p / 2Hh = K|V, (p >> 1) % Hh = head, p & 1 = half of the head's 16 dims
  const int ppk = 4 * Hh;                                // pieces per key row (2E halfs / 8)
  constexpr int NPT = KS * 4 * 8 / NTH;                  // pieces per thread at 8 heads (the host caps Hh at 8)
  const int npiece = KS * ppk;
  uint4 stage_regs[NPT];
  auto fetch = [&](int st) {
    const int s0 = r0 + st * KS;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int e = tid + j * NTH;
      const int key = e / ppk, p = e - key * ppk;          // HH = 8: shifts
      uint4 val = make_uint4(0, 0, 0, 0);
      if (e < npiece && s0 + key < r1)
        val = *reinterpret_cast<const uint4 *>(kv + ((size_t)b * S + s0 + key) * 2 * E + p * 8);
      stage_regs[j] = val;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int e = tid + j * NTH;
      if (e < npiece) {
        const int key = e / ppk, p = e - key * ppk;
        const int isv = p / (2 * Hh), hh = (p >> 1) % Hh, half8 = p & 1;
        unsigned char *dst = lds + buf * stage_bytes + (isv ? Hh * KS * KROW + (hh * KS + key) * VROW
                                                             : (hh * KS + key) * KROW);
        *reinterpret_cast<uint4 *>(dst + half8 * 16) = stage_regs[j];
      }
    }
  };
  fetch(0);
  commit(0);
  __syncthreads();

  const int nqg = (Q + 15) / 16;
  // this wave's query groups (same for every head it owns): blockIdx.y, + gridDim.y, ...  (<= MAXQG, host-checked)
  {
    const int h = wave % Hh, part_of_head = wave / Hh;   // Hh * WPH <= 16 wavefronts (host-checked)
    const int qg0 = blockIdx.y + part_of_head * gridDim.y, qgstep = WPH * gridDim.y;
    h4 qf[MAXQG];
    f4 acc[MAXQG], lacc[MAXQG];
    float m[MAXQG];
    const h4 ones = {(_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1};
#pragma unroll
    for (int j = 0; j < MAXQG; ++j) {
      const int qg = qg0 + j * qgstep;
      const int qi = min(qg * 16 + i, Q - 1);
      qf[j] = part_of_head < WPH ? *reinterpret_cast<const h4 *>(q + ((size_t)b * Q + qi) * E + h * kHD + 4 * g) : h4{0, 0, 0, 0};
      acc[j] = f4{0.f, 0.f, 0.f, 0.f};
      lacc[j] = f4{0.f, 0.f, 0.f, 0.f};
      m[j] = -INFINITY;
    }
    for (int st = 0; st < nstage; ++st) {
      if (st + 1 < nstage) fetch(st + 1);
      const int ns = min(KS, r1 - (r0 + st * KS));       // keys of this stage (ragged only at the very end)
      if (part_of_head < WPH) {
        const unsigned char *kb = lds + (st & 1) * stage_bytes + (size_t)h * KS * KROW;
        const unsigned char *vb = lds + (st & 1) * stage_bytes + (size_t)Hh * KS * KROW + (size_t)h * KS * VROW;
        {
#pragma unroll
          for (int j = 0; j < MAXQG; ++j) {
            if (qg0 + j * qgstep >= nqg) break;            // wave-uniform
            f4 sc[NT16];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
              const h4 kf = *reinterpret_cast<const h4 *>(kb + (16 * t + i) * KROW + g * 8);
              f4 c = __builtin_amdgcn_mfma_f32_16x16x16f16(kf, qf[j], f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);   // unscaled
              if (ns < KS) {                               // ragged stage (the very last one): keys past the end score -inf
                asm volatile("" ::: "memory");             // keep the wave-uniform test a BRANCH (as selects it costs
#pragma unroll                                             // 8 VALU instructions per tile in every stage)
                for (int r = 0; r < 4; ++r)
                  if (16 * t + 4 * g + r >= ns) c[r] = -INFINITY;
              }
              mx = max3_raw(c[2], c[3], max3_raw(c[0], c[1], mx));
              sc[t] = c;
            }
            mx = max3_raw(mx, __shfl_xor(mx, 16), mx);
            mx = max3_raw(mx, __shfl_xor(mx, 32), mx) * scale2;            // scale2 > 0: max(s c) = s max(c)
            const float mn = max3_raw(m[j], mx, mx);                        // finite: a stage has at least one key
            const float a = __builtin_amdgcn_exp2f(m[j] - mn);
            lacc[j] *= a;
            acc[j] *= a;
            m[j] = mn;
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
              h4 pf;
#pragma unroll
              for (int r = 0; r < 4; ++r) pf[r] = (_Float16)__builtin_amdgcn_exp2f(fmaf(sc[t][r], scale2, -mn));
              // the denominator on the matrix core too: ones . P^T sums the 16 keys of the tile for every query
              // (every row of the result is the sum: no cross-row reduction at the end), of the SAME fp16-rounded
              // probabilities as the numerator
              lacc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pf, lacc[j], 0, 0, 0);
              const hv4 vt = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                  (hv4 __attribute__((address_space(3))) *)(vb + (16 * t + 4 * g + (i >> 2)) * VROW + (i & 3) * 8));
              h4 vf;
              vf[0] = vt[0]; vf[1] = vt[1]; vf[2] = vt[2]; vf[3] = vt[3];
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf, acc[j], 0, 0, 0);
            }
          }
        }
      }
      if (st + 1 < nstage) commit((st + 1) & 1);
      __syncthreads();
    }
    if (part_of_head < WPH) {
#pragma unroll
      for (int j = 0; j < MAXQG; ++j) {
        const int qg = qg0 + j * qgstep;
        const int qi = qg * 16 + i;
        const float lj = lacc[j][0];
        if (qg < nqg && qi < Q) {
          float *dst = part + ((((size_t)b * Hh + h) * Q + qi) * nrange + range) * (kHD + 2);
          if (g == 0) {
            dst[0] = m[j];        // exp2 domain
            dst[1] = lj;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[2 + 4 * g + r] = acc[j][r];   // O^T[dim 4g + r][query i]
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mha_decode_combine4_kernel(const float *__restrict__ part,
                                                                  T *__restrict__ out, int B, int Q, int Hh,
                                                                  int nrange) {
  // one wavefront per (b, h, q): lane = (range slice cs of 4, head channel d); the ranges of a (b, h, q) are contiguous
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= B * Hh * Q) return;
  const int lane = threadIdx.x & 63, d = lane & 15, cs = lane >> 4;
  const int qi = w % Q, h = (w / Q) % Hh, b = w / (Q * Hh);
  const float *base = part + (((size_t)b * Hh + h) * Q + qi) * nrange * (kHD + 2);
  float M = -INFINITY;
  for (int c = cs; c < nrange; c += 4) M = fmaxf(M, base[(size_t)c * (kHD + 2)]);
  M = fmaxf(M, __shfl_xor(M, 16));
  M = fmaxf(M, __shfl_xor(M, 32));
  float L = 0.f, O = 0.f;
  for (int c = cs; c < nrange; c += 4) {
    const float *p = base + (size_t)c * (kHD + 2);
    const float wgt = __builtin_amdgcn_exp2f(p[0] - M);
    L += p[1] * wgt;
    O += p[2 + d] * wgt;
  }
  L += __shfl_xor(L, 16);
  L += __shfl_xor(L, 32);
  O += __shfl_xor(O, 16);
  O += __shfl_xor(O, 32);
  if (cs == 0) out[((size_t)b * Q + qi) * Hh * kHD + h * kHD + d] = (T)(O / L);
}

// launch geometry of the matrix-core path: query splits so that a wave has <= MAXQG groups, key ranges (multiples of
// the stage size) so that ~256 workgroups exist
static void mha_plan(int B, int Q, int S, int &qsplit, int &nrange, int &range_keys) {
  const int nqg = (Q + 15) / 16;
  qsplit = (nqg + mh::MAXQG * mh::WPH - 1) / (mh::MAXQG * mh::WPH);
  const int nstage_total = (S + mh::KS - 1) / mh::KS;
  int want = (256 + qsplit * B - 1) / (qsplit * B);
  want = std::max(1, std::min(want, nstage_total));
  const int stages_per_range = (nstage_total + want - 1) / want;
  range_keys = stages_per_range * mh::KS;
  nrange = (S + range_keys - 1) / range_keys;
}

}  // namespace di

extern "C" {

int di_heatmap_nms(const void *a, const void *b, float *out, int B, int num_classes, int H, int W,
                   int nms_kernel, unsigned k1_class_mask, int dtype, void *stream) {
  DI_REQUIRE(B > 0 && num_classes > 0 && num_classes <= 32 && H > 0 && W > 0, "bad heatmap shape");
  DI_REQUIRE(nms_kernel % 2 == 1 && nms_kernel >= 1, "nms kernel %d must be odd", nms_kernel);
  const int n = B * num_classes * H * W;
  const dim3 g((n + 255) / 256), blk(256);
  if (dtype == DI_F16)
    hipLaunchKernelGGL(di::heatmap_nms_kernel<__half>, g, blk, 0, (hipStream_t)stream, (const __half *)a,
                       (const __half *)b, out, B, num_classes, H, W, nms_kernel, k1_class_mask);
  else if (dtype == DI_F32)
    hipLaunchKernelGGL(di::heatmap_nms_kernel<float>, g, blk, 0, (hipStream_t)stream, (const float *)a,
                       (const float *)b, out, B, num_classes, H, W, nms_kernel, k1_class_mask);
  else {
    di::set_error("unsupported dtype %d", dtype);
    return DI_ERR_ARG;
  }
  return di::check_launch("heatmap_nms");
}

int di_query_geometry_ld(const float *center, const float *height, const float *dim, const float *rot,
                         const float *proj, const float *aug_rev, const float *per_sample, int32_t *on_img,
                         float *rect_img, float *rect_bev, int B, int Q, int ld, int n_views, float cell, float pc_x0,
                         float pc_y0, float bev_cell, float dim_scale, void *stream);

int di_query_geometry(const float *center, const float *height, const float *dim, const float *rot,
                      const float *proj, const float *aug_rev, const float *per_sample, int32_t *on_img,
                      float *rect_img, float *rect_bev, int B, int Q, int n_views, float cell, float pc_x0,
                      float pc_y0, float bev_cell, float dim_scale, void *stream) {
  return di_query_geometry_ld(center, height, dim, rot, proj, aug_rev, per_sample, on_img, rect_img, rect_bev, B, Q, Q,
                              n_views, cell, pc_x0, pc_y0, bev_cell, dim_scale, stream);
}

int di_query_geometry_ld(const float *center, const float *height, const float *dim, const float *rot,
                         const float *proj, const float *aug_rev, const float *per_sample, int32_t *on_img,
                         float *rect_img, float *rect_bev, int B, int Q, int ld, int n_views, float cell, float pc_x0,
                         float pc_y0, float bev_cell, float dim_scale, void *stream) {
  DI_REQUIRE(B > 0 && Q > 0 && ld >= Q, "bad query shape B=%d Q=%d ld=%d", B, Q, ld);
  DI_REQUIRE((rect_img == nullptr) == (on_img == nullptr), "rect_img and on_img go together");
  di::QGeomParams P{cell, pc_x0, pc_y0, bev_cell, dim_scale, ld};
  hipLaunchKernelGGL(di::query_geometry_kernel, dim3((B * Q + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, center, height, dim, rot, proj, aug_rev, per_sample, on_img,
                     rect_img, rect_bev, B, Q, n_views, P);
  return di::check_launch("query_geometry");
}

int di_roi_align_fwd(const void *feat, const float *rois, void *out, int R, int N, int H, int W, int C,
                     float spatial_scale, int dtype, void *stream) {
  return di_roi_align_x_fwd(feat, rois, out, R, N, H, W, C, spatial_scale, dtype, dtype, stream);
}

int di_roi_align_x_fwd(const void *feat, const float *rois, void *out, int R, int N, int H, int W, int C,
                       float spatial_scale, int dtype, int out_dtype, void *stream) {
  DI_REQUIRE(R >= 0 && N > 0 && H > 0 && W > 0, "bad roi_align shape");
  DI_REQUIRE(out_dtype == dtype || out_dtype == DI_F32 || (out_dtype == DI_F16_HL && dtype == DI_F16 && C == 128),
             "roi_align writes the map's type, float32, or (fp16 map, C = 128) the split hi | lo pair");
  DI_REQUIRE(C > 0 && C % 8 == 0 && C <= 128, "C=%d must be a multiple of 8, <= 128", C);
  if (R == 0) return DI_OK;
  const int total = R * 49;
  const int blocks = min((total + 15) / 16, 256 * 16);
  hipStream_t s = (hipStream_t)stream;
#define DI_ROI(TT, FULL)                                                                              \
  hipLaunchKernelGGL((di::roi_align_kernel<TT, FULL>), dim3(blocks), dim3(256), 0, s, (const TT *)feat, \
                     rois, (TT *)out, R, N, H, W, C, spatial_scale)
  if (out_dtype == DI_F16_HL) {                            // fp16 map -> RoI features with float32 accuracy as [hi | lo] fp16
    hipLaunchKernelGGL((di::roi_align_kernel<__half, true, di::HLOut>), dim3(blocks), dim3(256), 0, s, (const __half *)feat,
                       rois, (__half *)out, R, N, H, W, C, spatial_scale);
  } else if (dtype == DI_F16 && out_dtype == DI_F32) {    // fp16 map, float32 RoI features
    if (C == 128)
      hipLaunchKernelGGL((di::roi_align_kernel<__half, true, float>), dim3(blocks), dim3(256), 0, s, (const __half *)feat,
                         rois, (float *)out, R, N, H, W, C, spatial_scale);
    else
      hipLaunchKernelGGL((di::roi_align_kernel<__half, false, float>), dim3(blocks), dim3(256), 0, s, (const __half *)feat,
                         rois, (float *)out, R, N, H, W, C, spatial_scale);
  } else if (dtype == DI_F16) { if (C == 128) DI_ROI(__half, true); else DI_ROI(__half, false); }
  else if (dtype == DI_F32) { if (C == 128) DI_ROI(float, true); else DI_ROI(float, false); }
  else { di::set_error("unsupported dtype %d", dtype); return DI_ERR_ARG; }
#undef DI_ROI
  return di::check_launch("roi_align_fwd");
}

int di_mha_decode_scratch_floats(int B, int Q, int S, int num_heads) {
  int qsplit, nrange, range_keys;
  di::mha_plan(B, Q, S, qsplit, nrange, range_keys);
  const int nchunk = std::max(nrange, (S + di::kChunk - 1) / di::kChunk);   // the larger of the two paths' state counts
  return B * num_heads * nchunk * Q * (di::kHD + 2);
}

int di_mha_decode_fwd(const void *q, const void *kv, void *out, float *scratch, int B, int Q, int S,
                      int num_heads, int head_dim, float scale, int dtype, void *stream) {
  DI_REQUIRE(head_dim == di::kHD, "head_dim %d unsupported (16 only)", head_dim);
  DI_REQUIRE(B > 0 && Q > 0 && S > 0 && num_heads > 0 && scale > 0.f, "bad attention shape (scale must be positive)");
  const int nchunk = (S + di::kChunk - 1) / di::kChunk;
  const dim3 g1(nchunk, num_heads, B), blk(256);
  const dim3 g2((B * num_heads * Q + 15) / 16);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DI_F16 && num_heads <= 8) {
    // matrix-core path: all heads of a key range per workgroup
    int qsplit, nrange, range_keys;
    di::mha_plan(B, Q, S, qsplit, nrange, range_keys);
    const int lds = 2 * num_heads * di::mh::KS * di::mh::ROWB;
    static di::LdsRaised raised8, raised0;
    if (int rc = di::ensure_lds(raised8, (const void *)di::mha_decode_mfma_kernel<8>, 160 * 1024)) return rc;
    if (int rc = di::ensure_lds(raised0, (const void *)di::mha_decode_mfma_kernel<0>, 160 * 1024)) return rc;
    if (num_heads == 8)
      hipLaunchKernelGGL(di::mha_decode_mfma_kernel<8>, dim3(nrange, qsplit, B), dim3(di::mh::NTH), lds, s,
                         (const __half *)q, (const __half *)kv, scratch, B, Q, S, num_heads,
                         scale * 1.4426950408889634f, range_keys);
    else
      hipLaunchKernelGGL(di::mha_decode_mfma_kernel<0>, dim3(nrange, qsplit, B), dim3(di::mh::NTH), lds, s,
                         (const __half *)q, (const __half *)kv, scratch, B, Q, S, num_heads,
                         scale * 1.4426950408889634f, range_keys);
    hipLaunchKernelGGL(di::mha_decode_combine4_kernel<__half>, dim3((B * num_heads * Q + 3) / 4), blk, 0, s,
                       scratch, (__half *)out, B, Q, num_heads, nrange);
  } else if (dtype == DI_F16) {
    hipLaunchKernelGGL(di::mha_decode_partial_kernel<__half>, g1, blk, 0, s, (const __half *)q,
                       (const __half *)kv, scratch, B, Q, S, num_heads, scale);
    hipLaunchKernelGGL(di::mha_decode_combine_kernel<__half>, g2, blk, 0, s, scratch, (__half *)out, B, Q,
                       num_heads, nchunk);
  } else if (dtype == DI_F32) {
    hipLaunchKernelGGL(di::mha_decode_partial_kernel<float>, g1, blk, 0, s, (const float *)q,
                       (const float *)kv, scratch, B, Q, S, num_heads, scale);
    hipLaunchKernelGGL(di::mha_decode_combine_kernel<float>, g2, blk, 0, s, scratch, (float *)out, B, Q,
                       num_heads, nchunk);
  } else {
    di::set_error("unsupported dtype %d", dtype);
    return DI_ERR_ARG;
  }
  return di::check_launch("mha_decode_fwd");
}

}  // extern "C"
