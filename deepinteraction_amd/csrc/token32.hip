// Token-level kernels of the MMPI decoder in FLOAT32 for gfx950 (round 3).
//
// The decoder works on B*Q <= ~1000 query tokens of 128 channels (reference decoder_utils.py:35-113 decoder layer,
// :498-581 prediction FFN, :584-629 DynamicConv, :632-841 RoI blocks).  The work is tiny (a few MFLOP per linear layer)
// and numerically touchy: with random-init weights the 200 x 32 400 cross attention has logits of magnitude ~500 and
// the DynamicConv chain amplifies a 2^-11 perturbation of any operand to 1e-2 of the box outputs
// (tests/tools/fp16_error_budget.py).  So the token state, the RoI features, the generated DynamicConv parameters and
// every weight of this path are float32, multiplied on the matrix cores with the float32 MFMA (16x16x4, full fp32
// products and accumulation); only the BEV / image MAPS the tokens gather from stay fp16.
//
// Launch count is what the path is bound by (5-15 us of latency per launch for < 1 us of work), so the token-parallel
// parts are ONE kernel per dependency level: `program_kernel` gives a workgroup 16 token rows in LDS and runs a short
// PROGRAM of steps on them (load / self attention among the sample's queries / merge of the cross attention's partial
// states / linear / residual + LayerNorm / store / prediction heads), every linear layer reading its weights straight
// from L2.  What cannot be row-parallel stays its own kernel: the DynamicConv parameter generator (weight stationary,
// 128 -> 32 768 per query), the DynamicConv core (one workgroup per RoI) and the split-K out_layer (6 272 -> 128).
//
// All GEMMs run TRANSPOSED, Y^T = W . X^T: the A operand of `v_mfma_f32_16x16x4_f32` is one weight row per lane, the
// B operand one token row per lane, both read as float4 (4 consecutive k, consumed by 4 MFMAs whose k index g maps to
// k = 16c + 4g + t), and a lane ends with 4 consecutive output channels of one token.
#include <string.h>

#include "di_common.h"

namespace di {
namespace t32 {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ f4 mfma4(f4 a, f4 b, f4 c) {
#pragma unroll
  for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], c, 0, 0, 0);
  return c;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

constexpr int TM = 16;          // token rows per workgroup
constexpr int NW = 8;           // wavefronts per workgroup
constexpr int NTH = NW * 64;
constexpr int LDW = 516;        // floats per LDS row: 512 + 4 (float4 reads of 16 rows spread over the banks)
constexpr int NBUF = 3;
constexpr int LDS_BYTES = NBUF * TM * LDW * 4;

// ------------------------------------------------------------------------------------------------------------
// The steps of a program.  `src` / `dst` / `aux` name LDS row buffers (0..2); widths are in floats.
// ------------------------------------------------------------------------------------------------------------
struct Step {           // mirrors di_tok_step of include/deepinteraction_hip.h
  int kind, src, dst, aux;
  int K, N, a, b;
  float f;
  int pad;
  const void *p0, *p1, *p2, *p3;
  long long ld0, ld1;
};
struct Heads {          // mirrors di_tok_heads
  const float *w2, *b2, *qpos;
  const unsigned char *keep;
  float *pos_out;
  float *out[DI_TOK_MAX_HEADS];
  const float *first[DI_TOK_MAX_HEADS];
  int cls[DI_TOK_MAX_HEADS];
  int nheads, center_head, ldo, col0;
};
struct Program {
  int n;
  Step s[DI_TOK_MAX_STEPS];
};

// K_LOAD: dst[r][a + c] = p0[m * ld0 + c] (+ p1[m * ld1 + c]),  c < K
__device__ __forceinline__ void step_load(const Step &s, float *buf, long long m0, int rows, int tid) {
  const float *x = (const float *)s.p0, *p = (const float *)s.p1;
  float *d = buf + s.dst * TM * LDW + s.a;
  const int k4 = s.K >> 2;
  for (int e = tid; e < TM * k4; e += NTH) {
    const int r = e / k4, c = (e - r * k4) * 4;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      v = ld4(x + (m0 + r) * s.ld0 + c);
      if (p != nullptr) v += ld4(p + (m0 + r) * s.ld1 + c);
    }
    *reinterpret_cast<f4 *>(d + r * LDW + c) = v;
  }
}

// K_LOAD_PARTS: dst[r][c] = sum_{s < a} p0[(s * b + m) * 128 + c] + p1[c]   (split-K partial sums, b = total rows)
__device__ __forceinline__ void step_load_parts(const Step &s, float *buf, long long m0, int rows, int tid) {
  const float *part = (const float *)s.p0, *bias = (const float *)s.p1;
  float *d = buf + s.dst * TM * LDW;
  const int r = tid >> 5, c = (tid & 31) * 4;
  f4 v = bias != nullptr ? ld4(bias + c) : f4{0.f, 0.f, 0.f, 0.f};
  if (r < rows) {
    for (int sl = 0; sl < s.a; ++sl) v += ld4(part + ((size_t)sl * s.b + m0 + r) * 128 + c);
  } else {
    v = f4{0.f, 0.f, 0.f, 0.f};
  }
  *reinterpret_cast<f4 *>(d + r * LDW + c) = v;
}

// K_ATTN: self attention among the Q tokens of the sample for this workgroup's 16 queries (reference
// decoder_utils.py:743-746 / :824-826 and :91-95): wave = head (8 heads x 16 dims), packed projection p0 = [q | k | v]
// rows of ld0 floats.  Optional visibility (image RoI block: the attention runs among the queries of ONE view): key k
// is visible to query q when bit view[q] of member[k] is set, or when view[q] < 0.  f = scale * log2(e).
__device__ __forceinline__ void step_attn(const Step &s, float *buf, int b, int q0, int Q, int lane, int wave) {
  const float *base = (const float *)s.p0 + (size_t)b * Q * s.ld0;
  const unsigned char *member = (const unsigned char *)s.p1;
  const signed char *view = (const signed char *)s.p2;
  const int ld = (int)s.ld0, h = wave, i = lane & 15, g = lane >> 4;
  const int qc = min(q0 + i, Q - 1);
  const f4 qv = ld4(base + (size_t)qc * ld + h * 16 + 4 * g);
  const int vq = member != nullptr ? (int)view[(size_t)b * Q + qc] : -1;
  const float sl2 = s.f;
  float m = -INFINITY, l = 0.f;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const int ntile = (Q + 15) >> 4;
  // tile t: K row of key 16t + i (dims 4g..4g+3), V[key 16t + 4g + r][dim i], membership bytes of keys 16t + 4g + r
  auto fetch = [&](int t, f4 &kk, f4 &vv, unsigned &mem) {
    const int ki = min(16 * t + i, Q - 1);
    kk = ld4(base + (size_t)ki * ld + 128 + h * 16 + 4 * g);
    mem = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kr = min(16 * t + 4 * g + r, Q - 1);
      vv[r] = base[(size_t)kr * ld + 256 + h * 16 + i];
      if (member != nullptr) mem = (mem & ~(0xFFu << (8 * r))) | ((unsigned)member[(size_t)b * Q + kr] << (8 * r));
    }
  };
  f4 kk, vv, kn, vn;
  unsigned mem, memn;
  fetch(0, kk, vv, mem);
  for (int t = 0; t < ntile; ++t) {
    if (t + 1 < ntile) fetch(t + 1, kn, vn, memn);
    f4 c = mfma4(kk, qv, f4{0.f, 0.f, 0.f, 0.f});          // S^T[key 4g + r][query i]
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * t + 4 * g + r;
      const bool vis = key < Q && (vq < 0 || ((mem >> (8 * r + vq)) & 1u));
      c[r] = vis ? c[r] * sl2 : -INFINITY;
      mx = fmaxf(mx, c[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float ms = mn == -INFINITY ? 0.f : mn;             // nothing visible so far: keep everything at zero
    const float a = exp2f(m - ms);
    l *= a;
    acc *= a;
    m = mn;
    f4 p;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = exp2f(c[r] - ms);
      l += p[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[r], p[r], acc, 0, 0, 0);   // O^T[dim 4g+r'][query i]
    kk = kn; vv = vn; mem = memn;
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  *reinterpret_cast<f4 *>(buf + s.dst * TM * LDW + i * LDW + h * 16 + 4 * g) = acc * inv;
}

// K_COMBINE: merge of the cross attention's partial softmax states (csrc/decoder.hip mha_decode_*: one state
// [m (exp2 domain), l, O[16]] per (sample, head, query, key range), range-contiguous): wave = head.  a = ranges.
__device__ __forceinline__ void step_combine(const Step &s, float *buf, int b, int q0, int Q, int lane, int wave) {
  const int h = wave, i = lane & 15, g = lane >> 4, nrange = s.a;
  const int qc = min(q0 + i, Q - 1);
  const float *base = (const float *)s.p0 + (((size_t)b * NW + h) * Q + qc) * nrange * 18;
  float M = -INFINITY, L = 0.f;
  f4 O = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < nrange; c0 += 8) {
    float mc[8], lc[8];
    f4 oc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float *p = base + (size_t)min(c0 + j, nrange - 1) * 18;
      const float2 ml = *reinterpret_cast<const float2 *>(p);
      const float2 o0 = *reinterpret_cast<const float2 *>(p + 2 + 4 * g);
      const float2 o1 = *reinterpret_cast<const float2 *>(p + 4 + 4 * g);
      mc[j] = c0 + j < nrange ? ml.x : -INFINITY;
      lc[j] = ml.y;
      oc[j] = f4{o0.x, o0.y, o1.x, o1.y};
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float mn = fmaxf(M, mc[j]);
      const float ms = mn == -INFINITY ? 0.f : mn;
      const float a = exp2f(M - ms), w = exp2f(mc[j] - ms);
      L = L * a + lc[j] * w;
      O = O * a + oc[j] * w;
      M = mn;
    }
  }
  const float inv = L > 0.f ? 1.f / L : 0.f;
  *reinterpret_cast<f4 *>(buf + s.dst * TM * LDW + i * LDW + h * 16 + 4 * g) = O * inv;
}

// K_LINEAR: dst[r][n] = act(sum_k src[r][k] * W[n][k] + bias[n]),  n < N, k < K (multiples of 16; W row-major (N,K))
__device__ __forceinline__ void step_linear(const Step &s, float *buf, int lane, int wave) {
  const float *W = (const float *)s.p0, *bias = (const float *)s.p1;
  const float *x = buf + s.src * TM * LDW;
  float *d = buf + s.dst * TM * LDW;
  const int i = lane & 15, g = lane >> 4, K = s.K;
  for (int t = wave; t < (s.N >> 4); t += NW) {
    const float *wr = W + (size_t)(16 * t + i) * K + 4 * g;
    const float *xr = x + i * LDW + 4 * g;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 128) {                 // 8 weight fragments in flight per lane
      f4 a[8];
      const int nc = min(8, (K - k0) >> 4);
#pragma unroll
      for (int c = 0; c < 8; ++c) a[c] = c < nc ? ld4(wr + k0 + 16 * c) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < nc) acc = mfma4(a[c], ld4(xr + k0 + 16 * c), acc);
    }
    const int n = 16 * t + 4 * g;
    if (bias != nullptr) acc += ld4(bias + n);
    if (s.a == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
    } else if (s.a == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = gelu_erf(acc[r]);
    }
    *reinterpret_cast<f4 *>(d + i * LDW + n) = acc;       // Y^T[channel 4g + r][token i]
  }
}

// K_ROWOP on 128-wide rows: v = src (+ buf[aux] if aux >= 0); v = LayerNorm(v) * p0 + p1 if p0; relu if b & 1;
// rows with p2[m] == 0 zeroed.  16 lanes per row, 8 channels per lane.
__device__ __forceinline__ void step_rowop(const Step &s, float *buf, long long m0, int rows, int tid) {
  if (tid >= 256) return;
  const int r = tid >> 4, c0 = (tid & 15) * 8;
  const float *x = buf + s.src * TM * LDW + r * LDW + c0;
  float v[8];
  unpack8(ld8(x), v);
  if (s.aux >= 0) {
    float a[8];
    unpack8(ld8(buf + s.aux * TM * LDW + r * LDW + c0), a);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += a[j];
  }
  if (s.p0 != nullptr) {
    float sum = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
    const float mean = row16_sum(sum) * (1.f / 128.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = v[j] - mean;
      ss += d * d;
    }
    const float inv = rsqrtf(row16_sum(ss) * (1.f / 128.f) + s.f);
    float w[8], bb[8];
    unpack8(ld8((const float *)s.p0 + c0), w);
    unpack8(ld8((const float *)s.p1 + c0), bb);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (v[j] - mean) * inv * w[j] + bb[j];
  }
  if (s.b & 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
  }
  if (s.p2 != nullptr && r < rows && !((const unsigned char *)s.p2)[m0 + r]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  }
  st8(buf + s.dst * TM * LDW + r * LDW + c0, pack8f(v, 0.f));
}

// K_STORE: p0[m * ld0 + c] = src[r][a + c],  c < N
__device__ __forceinline__ void step_store(const Step &s, const float *buf, long long m0, int rows, int tid) {
  float *y = (float *)s.p0;
  const float *x = buf + s.src * TM * LDW + s.a;
  const int n4 = s.N >> 2;
  for (int e = tid; e < rows * n4; e += NTH) {
    const int r = e / n4, c = (e - r * n4) * 4;
    *reinterpret_cast<f4 *>(y + (m0 + r) * s.ld0 + c) = ld4(x + r * LDW + c);
  }
}

// K_HEADS: the second layers of the prediction heads (reference decoder_utils.py:498-581: per head Conv1d(C -> 64) +
// BN + ReLU - a K_LINEAR step into `src`, BatchNorm folded, heads stacked - then Conv1d(64 -> classes)), and what
// follows every call in deepinteraction_decoder.py: `center += query_pos` (:265,:288), the on-the-image merge with the
// first stage's result (:292-295), the placement at column col0 of the (B, classes, ldo) output tensors (:304-311).
__device__ __forceinline__ void step_heads(const Step &s, const Heads &ho, const float *buf, int b, int q0, int Q, int tid) {
  const float *hid = buf + s.src * TM * LDW;
  int total = 0;
#pragma unroll
  for (int h = 0; h < DI_TOK_MAX_HEADS; ++h) total += h < ho.nheads ? ho.cls[h] : 0;
  for (int e = tid; e < TM * total; e += NTH) {
    const int r = e / total, o = e - r * total;
    const int q = q0 + r;
    if (q >= Q) continue;
    int h = 0, row0 = 0;
    while (h + 1 < ho.nheads && o >= row0 + ho.cls[h]) row0 += ho.cls[h++];
    const int cidx = o - row0, ncls = ho.cls[h];
    const float *wr = ho.w2 + (size_t)o * 64;
    const float *hr = hid + r * LDW + h * 64;
    float a = ho.b2[o];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const f4 w = ld4(wr + 4 * j), x = ld4(hr + 4 * j);
      a = fmaf(w[0], x[0], a);
      a = fmaf(w[1], x[1], a);
      a = fmaf(w[2], x[2], a);
      a = fmaf(w[3], x[3], a);
    }
    const size_t bq = (size_t)b * Q + q;
    if (h == ho.center_head) a += ho.qpos[bq * 2 + cidx];
    if (ho.keep != nullptr && !ho.keep[bq]) a = ho.first[h][((size_t)b * ncls + cidx) * Q + q];
    ho.out[h][((size_t)b * ncls + cidx) * ho.ldo + ho.col0 + q] = a;
    if (h == ho.center_head && ho.pos_out != nullptr) ho.pos_out[bq * 2 + cidx] = a;
  }
}

__global__ __launch_bounds__(NTH) void program_kernel(Program prog, Heads heads, int Q) {
  extern __shared__ __align__(16) float buf[];          // NBUF x TM x LDW
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, q0 = blockIdx.x * TM;
  const int rows = min(TM, Q - q0);
  const long long m0 = (long long)b * Q + q0;
  for (int si = 0; si < prog.n; ++si) {
    const Step &s = prog.s[si];
    switch (s.kind) {
      case DI_TOK_LOAD: step_load(s, buf, m0, rows, tid); break;
      case DI_TOK_LOAD_PARTS: step_load_parts(s, buf, m0, rows, tid); break;
      case DI_TOK_ATTN: step_attn(s, buf, b, q0, Q, lane, wave); break;
      case DI_TOK_COMBINE: step_combine(s, buf, b, q0, Q, lane, wave); break;
      case DI_TOK_LINEAR: step_linear(s, buf, lane, wave); break;
      case DI_TOK_ROWOP: step_rowop(s, buf, m0, rows, tid); break;
      case DI_TOK_STORE: step_store(s, buf, m0, rows, tid); break;
      case DI_TOK_HEADS: step_heads(s, heads, buf, b, q0, Q, tid); break;
      default: break;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------
// Wide linear (N >> M, K == 128), weight stationary: DynamicConv's parameter generator (decoder_utils.py:608:
// Linear 128 -> 2*128*128 per query).  A workgroup owns 128 output columns, a wave 32 of them with its 16 weight
// fragments in registers, and walks all the token rows.  Rows of W map to MFMA rows so that a lane holds 8
// consecutive output columns (32-B stores, 128 B per token row and wave quarter).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wide_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ W,
                                                   const float *__restrict__ bias, float *__restrict__ Y,
                                                   long long ldy, int M, int N) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 128 + wave * 32;
  // MFMA row i of fragment nb <-> output column n0 + 8 (i >> 2) + 4 nb + (i & 3)
  f4 a[2][8];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = n0 + 8 * (i >> 2) + 4 * nb + (i & 3);
#pragma unroll
    for (int c = 0; c < 8; ++c) a[nb][c] = ld4(W + (size_t)n * 128 + 16 * c + 4 * g);
  }
  f4 bs[2];
  bs[0] = bias ? ld4(bias + n0 + 8 * g) : f4{0.f, 0.f, 0.f, 0.f};
  bs[1] = bias ? ld4(bias + n0 + 8 * g + 4) : f4{0.f, 0.f, 0.f, 0.f};
  f4 xb[8], xn[8];
  auto fetch = [&](int m0, f4 (&x)[8]) {
    const int mr = min(m0 + i, M - 1);
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = ld4(X + (size_t)mr * ldx + 16 * c + 4 * g);
  };
  fetch(0, xb);
  for (int m0 = 0; m0 < M; m0 += 16) {
    if (m0 + 16 < M) fetch(m0 + 16, xn);
    f4 acc[2] = {bs[0], bs[1]};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      acc[0] = mfma4(a[0][c], xb[c], acc[0]);
      acc[1] = mfma4(a[1][c], xb[c], acc[1]);
    }
    if (m0 + i < M) {
      float *y = Y + (size_t)(m0 + i) * ldy + n0 + 8 * g;
      *reinterpret_cast<f4 *>(y) = acc[0];
      *reinterpret_cast<f4 *>(y + 4) = acc[1];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) xb[c] = xn[c];
  }
}

// ------------------------------------------------------------------------------------------------------------
// Split-K linear (K >> 128, N == 128): DynamicConv's out_layer (decoder_utils.py:624: Linear 49*128 -> 128 on the
// flattened RoI feature).  grid (row blocks of 16, K slices); a wave owns 32 output columns; partial sums go to a
// float32 workspace (slice, M, 128), summed by a K_LOAD_PARTS step of the program that follows.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splitk_kernel(const float *__restrict__ X, long long ldx,
                                                     const float *__restrict__ W, float *__restrict__ part, int M,
                                                     int K, int kslice) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const long long m0 = (long long)blockIdx.x * 16;
  const long long mr = m0 + i < M ? m0 + i : M - 1;
  const int kb = blockIdx.y * kslice, ke = min(kb + kslice, K);
  const float *xr = X + mr * ldx + 4 * g;
  const float *w0 = W + (size_t)(wave * 32 + i) * K + 4 * g, *w1 = w0 + (size_t)16 * K;
  f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
  for (int k0 = kb; k0 < ke; k0 += 64) {                  // 4 chunks of 16 in flight
    f4 xa[4], wa[4], wb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = min(k0 + 16 * c, ke - 16);
      xa[c] = ld4(xr + k);
      wa[c] = ld4(w0 + k);
      wb[c] = ld4(w1 + k);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (k0 + 16 * c < ke) {
        acc[0] = mfma4(wa[c], xa[c], acc[0]);
        acc[1] = mfma4(wb[c], xa[c], acc[1]);
      }
  }
  if (m0 + i < M) {
    float *dst = part + ((size_t)blockIdx.y * M + m0 + i) * 128 + wave * 32 + 4 * g;
    *reinterpret_cast<f4 *>(dst) = acc[0];
    *reinterpret_cast<f4 *>(dst + 16) = acc[1];
  }
}

// ------------------------------------------------------------------------------------------------------------
// DynamicConv core (decoder_utils.py:617-622), one workgroup per RoI:
//     F1 = relu(LN1(roi (49x128) . p1 (128x128)));   F2 = relu(LN2(F1 . p2))         -> F2 (49x128)
// computed transposed, F1^T = p1^T . roi^T: the accumulators of the first product (4 consecutive channels d of one
// spatial position per lane) ARE the B operands of the second one (MFMA (nb, r) consumes k <-> d = 16nb + 4g + r).
// The generated parameters arrive as params[q] = [ p1^T (d, c) | p2^T (e, d) ]: the rows of the generating Linear are
// permuted once on the host, so both A operands are plain float4 loads.  A wave owns 16 spatial positions.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dynconv_kernel(const float *__restrict__ roi, const float *__restrict__ params,
                                                      const float *__restrict__ n1w, const float *__restrict__ n1b,
                                                      const float *__restrict__ n2w, const float *__restrict__ n2b,
                                                      float *__restrict__ out, float eps) {
  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int s = wave * 16 + i;                          // spatial position of this lane's column (49 valid)
  const int sr = s < 49 ? s : 48;
  const float *rq = roi + ((size_t)q * 49 + sr) * 128 + 4 * g;
  const float *p1 = params + (size_t)q * 32768 + 4 * g, *p2 = p1 + 16384;
  f4 xb[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) xb[c] = ld4(rq + 16 * c);
  f4 acc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    f4 a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = ld4(p1 + (16 * nb + i) * 128 + 16 * c);
    f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; ++c) v = mfma4(a[c], xb[c], v);
    acc[nb] = v;                                        // F1^T[d = 16nb + 4g + r][position i]
  }
  // LayerNorm over the 128 channels of position s: lane holds d = 16nb + 4g + r; the other 96 live in lanes i + 16g'
  auto ln_relu = [&](f4 (&v)[8], const float *w, const float *b) {
    float sum = 0.f;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) sum += v[nb][0] + v[nb][1] + v[nb][2] + v[nb][3];
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.f / 128.f);
    float ss = 0.f;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = v[nb][r] - mean;
        ss += d * d;
      }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    const float inv = rsqrtf(ss * (1.f / 128.f) + eps);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const f4 wv = ld4(w + 16 * nb + 4 * g), bv = ld4(b + 16 * nb + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[nb][r] = fmaxf((v[nb][r] - mean) * inv * wv[r] + bv[r], 0.f);
    }
  };
  ln_relu(acc, n1w, n1b);
  f4 acc2[8];
#pragma unroll
  for (int ne = 0; ne < 8; ++ne) {
    f4 a[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) a[nb] = ld4(p2 + (16 * ne + i) * 128 + 16 * nb);   // p2^T[e = 16ne + i][d = 16nb + 4g + r]
    f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) v = mfma4(a[nb], acc[nb], v);
    acc2[ne] = v;
  }
  ln_relu(acc2, n2w, n2b);
  if (s < 49) {
    float *o = out + ((size_t)q * 49 + s) * 128 + 4 * g;
#pragma unroll
    for (int ne = 0; ne < 8; ++ne) *reinterpret_cast<f4 *>(o + 16 * ne) = acc2[ne];
  }
}

// ------------------------------------------------------------------------------------------------------------
// Query initialisation (deepinteraction_decoder.py:242-253) + the learned positional embedding of the proposals
// (decoder_utils.py:16-32, BatchNorm folded): one workgroup of 128 threads per query.
//   feat = bev[cell] + class_encoding[:, label] + bias;  pos = (cell % W + .5, cell // W + .5);
//   pe = W2 . relu(W1 . pos + b1) + b2
// The BEV map is fp16 (a feature MAP), everything written is float32.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void query_init_kernel(const __half *__restrict__ bev /*(B,H,W,128)*/,
                                                         const long long *__restrict__ top, const float *__restrict__ ce_w /*(128, ncls)*/,
                                                         const float *__restrict__ ce_b, const float *__restrict__ w1,
                                                         const float *__restrict__ b1, const float *__restrict__ w2,
                                                         const float *__restrict__ b2, float *__restrict__ feat,
                                                         float *__restrict__ pe_out, float *__restrict__ pos_out,
                                                         long long *__restrict__ labels, int Q, int HW, int Wb, int ncls) {
  __shared__ float hid[128];
  const int q = blockIdx.x, b = blockIdx.y, c = threadIdx.x;
  const long long t = top[(size_t)b * Q + q];
  const int cls = (int)(t / HW), cell = (int)(t % HW);
  const float px = (float)(cell % Wb) + 0.5f, py = (float)(cell / Wb) + 0.5f;
  feat[((size_t)b * Q + q) * 128 + c] = __half2float(bev[((size_t)b * HW + cell) * 128 + c]) + ce_w[c * ncls + cls] + ce_b[c];
  hid[c] = fmaxf(w1[2 * c] * px + w1[2 * c + 1] * py + b1[c], 0.f);
  __syncthreads();
  float a = b2[c];
  const float *wr = w2 + (size_t)c * 128;
#pragma unroll 8
  for (int j = 0; j < 32; ++j) {
    const f4 w = ld4(wr + 4 * j);
    a = fmaf(w[0], hid[4 * j], a);
    a = fmaf(w[1], hid[4 * j + 1], a);
    a = fmaf(w[2], hid[4 * j + 2], a);
    a = fmaf(w[3], hid[4 * j + 3], a);
  }
  pe_out[((size_t)b * Q + q) * 128 + c] = a;
  if (c == 0) {
    pos_out[((size_t)b * Q + q) * 2] = px;
    pos_out[((size_t)b * Q + q) * 2 + 1] = py;
    labels[(size_t)b * Q + q] = cls;
  }
}

// ------------------------------------------------------------------------------------------------------------
// RoI bookkeeping of the two RoI blocks, on the device (no host synchronisation).
//   image block (decoder_utils.py:681-759): a view with <= 1 centre on it is skipped (:726), a later view overwrites
//   an earlier one (:728,:759) => query q keeps the output of its LAST valid view v*(q); its self attention runs
//   among the queries of that view.  Out: view[q] = v*(q) or -1, member[q] = bit v set when q is on valid view v,
//   rois[q] = (b*V + v*(q), rect of q on v*(q)), keep[q] = v*(q) >= 0, on_img[q] = float(v*(q)).
//   point block (:804-823): rois[q] = (b, BEV rect of q).
// One workgroup per sample.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roi_select_kernel(const int *__restrict__ on /*(B,V,Q)*/,
                                                         const float *__restrict__ rect /*(B,V,Q,4) or (B,Q,4)*/,
                                                         float *__restrict__ rois /*(B*Q,5)*/, signed char *__restrict__ view,
                                                         unsigned char *__restrict__ member, unsigned char *__restrict__ keep,
                                                         float *__restrict__ on_img, int V, int Q) {
  __shared__ int cnt[8];
  const int b = blockIdx.x, t = threadIdx.x;
  if (on == nullptr) {                                     // point block
    for (int q = t; q < Q; q += 256) {
      float *r = rois + ((size_t)b * Q + q) * 5;
      const float *s = rect + ((size_t)b * Q + q) * 4;
      r[0] = (float)b; r[1] = s[0]; r[2] = s[1]; r[3] = s[2]; r[4] = s[3];
    }
    return;
  }
  if (t < 8) cnt[t] = 0;
  __syncthreads();
  for (int e = t; e < V * Q; e += 256)
    if (on[(size_t)b * V * Q + e]) atomicAdd(&cnt[e / Q], 1);
  __syncthreads();
  for (int q = t; q < Q; q += 256) {
    int last = -1;
    unsigned bits = 0;
    for (int v = 0; v < V; ++v)
      if (on[((size_t)b * V + v) * Q + q] && cnt[v] > 1) {
        last = v;
        bits |= 1u << v;
      }
    const int lc = last < 0 ? 0 : last;
    const size_t o = (size_t)b * Q + q;
    view[o] = (signed char)last;
    member[o] = (unsigned char)bits;
    keep[o] = last >= 0;
    on_img[o] = (float)last;
    const float *s = rect + (((size_t)b * V + lc) * Q + q) * 4;
    float *r = rois + o * 5;
    r[0] = (float)(b * V + lc); r[1] = s[0]; r[2] = s[1]; r[3] = s[2]; r[4] = s[3];
  }
}

static int check_program(const di_tok_step *steps, int n, const di_tok_heads *heads) {
  DI_REQUIRE(steps != nullptr && n > 0 && n <= DI_TOK_MAX_STEPS, "a token program has 1..%d steps, got %d", DI_TOK_MAX_STEPS, n);
  for (int i = 0; i < n; ++i) {
    const di_tok_step &s = steps[i];
    auto okbuf = [](int b) { return b >= 0 && b < NBUF; };
    switch (s.kind) {
      case DI_TOK_LOAD:
        DI_REQUIRE(okbuf(s.dst) && s.p0 && s.K > 0 && s.K % 4 == 0 && s.a >= 0 && s.a % 4 == 0 && s.a + s.K <= 512,
                   "step %d (load): bad buffer / width", i);
        break;
      case DI_TOK_LOAD_PARTS:
        DI_REQUIRE(okbuf(s.dst) && s.p0 && s.a > 0 && s.b > 0, "step %d (load parts): bad arguments", i);
        break;
      case DI_TOK_ATTN:
        DI_REQUIRE(okbuf(s.dst) && s.p0 && s.ld0 >= 384 && s.ld0 % 4 == 0 && (s.p1 == nullptr) == (s.p2 == nullptr),
                   "step %d (attention): needs the packed [q|k|v] projection (8 heads x 16) and member + view together", i);
        break;
      case DI_TOK_COMBINE:
        DI_REQUIRE(okbuf(s.dst) && s.p0 && s.a > 0, "step %d (combine): bad arguments", i);
        break;
      case DI_TOK_LINEAR:
        DI_REQUIRE(okbuf(s.src) && okbuf(s.dst) && s.src != s.dst && s.p0 && s.K > 0 && s.K % 16 == 0 && s.K <= 512 &&
                       s.N > 0 && s.N % 16 == 0 && s.N <= 512 && s.a >= 0 && s.a <= 2,
                   "step %d (linear): K, N multiples of 16 up to 512, src != dst", i);
        break;
      case DI_TOK_ROWOP:
        DI_REQUIRE(okbuf(s.src) && okbuf(s.dst) && s.aux < NBUF && (s.p0 == nullptr) == (s.p1 == nullptr),
                   "step %d (row op): bad buffers / LayerNorm needs weight and bias", i);
        break;
      case DI_TOK_STORE:
        DI_REQUIRE(okbuf(s.src) && s.p0 && s.N > 0 && s.N % 4 == 0 && s.a >= 0 && s.a % 4 == 0 && s.a + s.N <= 512,
                   "step %d (store): bad buffer / width", i);
        break;
      case DI_TOK_HEADS:
        DI_REQUIRE(okbuf(s.src) && heads != nullptr && heads->nheads > 0 && heads->nheads <= DI_TOK_MAX_HEADS &&
                       heads->w2 && heads->b2 && heads->qpos && (heads->keep == nullptr || heads->first[0] != nullptr),
                   "step %d (heads): needs the head description (the on-the-image merge needs the first stage's outputs)", i);
        break;
      default:
        set_error("step %d: unknown kind %d", i, s.kind);
        return DI_ERR_ARG;
    }
  }
  return DI_OK;
}

}  // namespace t32
}  // namespace di

extern "C" {

int di_token_program(const di_tok_step *steps, int nsteps, const di_tok_heads *heads, int B, int Q, void *stream) {
  using namespace di::t32;
  static_assert(sizeof(Step) == sizeof(di_tok_step) && sizeof(Heads) == sizeof(di_tok_heads), "C ABI structs");
  DI_REQUIRE(B > 0 && Q > 0, "bad token program shape B=%d Q=%d", B, Q);
  if (int rc = check_program(steps, nsteps, heads)) return rc;
  Program prog;
  prog.n = nsteps;
  memcpy(prog.s, steps, sizeof(Step) * nsteps);
  Heads h;
  memset(&h, 0, sizeof(h));
  if (heads != nullptr) memcpy(&h, heads, sizeof(h));
  static di::LdsRaised raised;
  if (int rc = di::ensure_lds(raised, (const void *)program_kernel, LDS_BYTES)) return rc;
  hipLaunchKernelGGL(program_kernel, dim3((Q + TM - 1) / TM, B), dim3(NTH), LDS_BYTES, (hipStream_t)stream, prog, h, Q);
  return di::check_launch("token_program");
}

int di_token_wide(const float *x, int ldx, const float *w, const float *bias, float *y, long long ldy, int M, int N,
                  void *stream) {
  DI_REQUIRE(x && w && y && M > 0 && N > 0 && N % 128 == 0, "bad wide linear shape M=%d N=%d (N multiple of 128, K = 128)", M, N);
  hipLaunchKernelGGL(di::t32::wide_kernel, dim3(N / 128), dim3(256), 0, (hipStream_t)stream, x, ldx, w, bias, y, ldy, M, N);
  return di::check_launch("token_wide");
}

long long di_token_splitk_workspace_bytes(int M, int K) { return (long long)((K + 447) / 448) * M * 128 * 4; }

int di_token_splitk(const float *x, long long ldx, const float *w, float *workspace, int M, int K, int *nslices,
                    void *stream) {
  DI_REQUIRE(x && w && workspace && M > 0 && K > 0 && K % 16 == 0, "bad split-K shape M=%d K=%d (N = 128)", M, K);
  const int kslice = 448, ns = (K + kslice - 1) / kslice;
  hipLaunchKernelGGL(di::t32::splitk_kernel, dim3((M + 15) / 16, ns), dim3(256), 0, (hipStream_t)stream, x, ldx, w,
                     workspace, M, K, kslice);
  if (nslices) *nslices = ns;
  return di::check_launch("token_splitk");
}

int di_dynconv_fwd(const float *roi, const float *params, const float *n1w, const float *n1b, const float *n2w,
                   const float *n2b, float *out, int R, float eps, void *stream) {
  DI_REQUIRE(R > 0 && roi && params && out && n1w && n1b && n2w && n2b, "bad DynamicConv call R=%d", R);
  hipLaunchKernelGGL(di::t32::dynconv_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, roi, params, n1w, n1b, n2w, n2b,
                     out, eps);
  return di::check_launch("dynconv_fwd");
}

int di_roi_select(const int *on, const float *rect, float *rois, void *view, void *member, void *keep, float *on_img,
                  int B, int V, int Q, void *stream) {
  DI_REQUIRE(B > 0 && Q > 0 && rect && rois, "bad roi_select shape");
  DI_REQUIRE(on == nullptr || (V > 0 && V <= 8 && view && member && keep && on_img), "image mode needs V <= 8 and all outputs");
  hipLaunchKernelGGL(di::t32::roi_select_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, on, rect, rois,
                     (signed char *)view, (unsigned char *)member, (unsigned char *)keep, on_img, V, Q);
  return di::check_launch("roi_select");
}

int di_query_init(const void *bev, const long long *top, const float *ce_w, const float *ce_b, const float *w1,
                  const float *b1, const float *w2, const float *b2, float *feat, float *pe, float *pos, long long *labels,
                  int B, int Q, int Hb, int Wb, int ncls, void *stream) {
  DI_REQUIRE(B > 0 && Q > 0 && Hb > 0 && Wb > 0 && ncls > 0, "bad query_init shape");
  hipLaunchKernelGGL(di::t32::query_init_kernel, dim3(Q, B), dim3(128), 0, (hipStream_t)stream, (const __half *)bev, top,
                     ce_w, ce_b, w1, b1, w2, b2, feat, pe, pos, labels, Q, Hb * Wb, Wb, ncls);
  return di::check_launch("query_init");
}

}  // extern "C"
