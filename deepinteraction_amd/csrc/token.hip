// Token-level kernels of the MMPI decoder (fp16 inference form) for gfx950.
//
// The decoder works on B*Q <= ~1000 query tokens of 128 channels (reference decoder_utils.py:35-113 decoder layer,
// :498-581 prediction FFN, :584-629 DynamicConv, :632-841 RoI blocks).  As library calls that is ~300 launches of
// tiny GEMMs, casts, adds, LayerNorms, soft-maxes: the step is launch-count bound.  Here every linear layer is ONE
// kernel including what surrounds it - the positional-embedding add on its input, bias, ReLU/GELU, the residual
// add and up to two LayerNorms on its output - and the other pieces (self attention among the queries, the two
// per-query 49x128x128 products of DynamicConv with their LayerNorms, query initialisation, the six prediction heads
// with the centre offset / on-the-image merge) are one kernel each.
//
// All GEMMs run on the matrix cores TRANSPOSED, Y^T = W . X^T (16x16x32 f16 MFMA, fp32 accumulate): the A operand is
// 8 consecutive k of one weight row, the B operand 8 consecutive k of one token row - both single 16-B loads from the
// row-major arrays - and a lane ends up with 4 consecutive output channels of one token.
#include "di_common.h"

namespace di {
namespace tok {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ h8 ld_h8(const __half *p) { return __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(p)); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// What follows the accumulation of a 128-wide output row (all optional, in this order):
//   v += bias;  v = act1(v);  v = LN1(v + res1);  v = relu(v) (act2);  v = LN2(v + res2)
struct Epilogue {
  const float *bias;        // (N) or null
  int act1;                 // 0 none, 1 relu, 2 gelu (erf)
  const __half *res1;       // (M, ldr1) or null
  int ldr1;
  const __half *ln1_w, *ln1_b;   // (128) or null: LayerNorm over the 128 outputs
  int act2;                 // relu after LN1
  const __half *res2;
  int ldr2;
  const __half *ln2_w, *ln2_b;
  float eps;
  const unsigned char *keep;     // (M) or null: rows with keep[m] == 0 are written as zeros (decoder_utils.py:665)
};

// The epilogue's memory operands of one (row, 8-channel) slot, fetched at the START of the kernel so that their round
// trip overlaps the K loop (these kernels are a chain of dependent round trips; every one removed is ~1 us).
struct EpiOperands {
  Pack8<__half> r1, r2, w1, b1, w2, b2;
  float bias[8];
};
__device__ __forceinline__ EpiOperands epilogue_prefetch(const Epilogue &e, long long m, int c0, bool row_ok) {
  EpiOperands o;
  o.r1 = o.r2 = o.w1 = o.b1 = o.w2 = o.b2 = zero8<__half>();
  if (e.res1 && e.ln1_w && row_ok) o.r1 = ld8(e.res1 + m * e.ldr1 + c0);
  if (e.res2 && e.ln2_w && row_ok) o.r2 = ld8(e.res2 + m * e.ldr2 + c0);
  if (e.ln1_w) {
    o.w1 = ld8(e.ln1_w + c0);
    o.b1 = ld8(e.ln1_b + c0);
  }
  if (e.ln2_w) {
    o.w2 = ld8(e.ln2_w + c0);
    o.b2 = ld8(e.ln2_b + c0);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) o.bias[j] = e.bias ? e.bias[c0 + j] : 0.f;
  return o;
}

__device__ __forceinline__ void layer_norm8(float (&v)[8], const Pack8<__half> &w, const Pack8<__half> &b, float eps) {
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j];
  s = row16_sum(s);
  const float mean = s * (1.f / 128.f);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float d = v[j] - mean;
    ss += d * d;
  }
  ss = row16_sum(ss);
  const float inv = rsqrtf(ss * (1.f / 128.f) + eps);
  float wf[8], bf[8];
  unpack8(w, wf);
  unpack8(b, bf);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (v[j] - mean) * inv * wf[j] + bf[j];
}

// 16 consecutive lanes own one token row of 128 outputs, 8 channels each (c0 = 8 * (lane & 15)): needs N == 128 when
// a LayerNorm is present.
__device__ __forceinline__ void epilogue8(float (&v)[8], const Epilogue &e, const EpiOperands &o, long long m, bool row_ok) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] += o.bias[j];
  if (e.act1 == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
  } else if (e.act1 == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = gelu_erf(v[j]);
  }
  if (e.ln1_w) {
    float r[8];
    unpack8(o.r1, r);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += r[j];
    layer_norm8(v, o.w1, o.b1, e.eps);
  }
  if (e.act2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
  }
  if (e.ln2_w) {
    float r[8];
    unpack8(o.r2, r);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += r[j];
    layer_norm8(v, o.w2, o.b2, e.eps);
  }
  if (e.keep && row_ok && !e.keep[m]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Row-block linear: a workgroup owns 16 token rows x 128 output columns (grid.y blocks of 128 columns); its four
// waves split K, partial sums meet in LDS, then 16 lanes per row run the epilogue.  K % 128 == 0, N % 128 == 0.
//   Y[m][n] = epilogue( sum_k (X[m][k] + P[m][k]) * W[n][k] )
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tl_rowblock_kernel(const __half *__restrict__ X, int ldx,
                                                          const __half *__restrict__ X2, int ldx2, int K1,
                                                          const __half *__restrict__ P, int ldp,
                                                          const __half *__restrict__ W, Epilogue ep,
                                                          __half *__restrict__ Y, int ldy, int M, int N, int K) {
  __shared__ __align__(16) float part[4][16][132];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const long long m0 = (long long)blockIdx.x * 16;
  const int n0 = blockIdx.y * 128;
  const long long mr = m0 + i < M ? m0 + i : M - 1;            // ragged tail: clamped read, masked store
  const int kq = K / 4;                                        // this wave's K range
  const int em = tid >> 4, ec0 = (tid & 15) * 8;               // this thread's epilogue slot: row em, channels ec0..+7
  const bool eok = m0 + em < M;
  Epilogue e = ep;
  if (e.bias) e.bias += n0;
  const EpiOperands eo = epilogue_prefetch(e, m0 + em, ec0, eok);
  f4 acc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = wave * kq; k0 < (wave + 1) * kq; k0 += 32) {
    const int kk = k0 + g * 8;
    // columns [0, K1) come from X, [K1, K) from X2 (a channel concat that is never materialised); K1 % 32 == 0
    h8 b = kk < K1 ? ld_h8(X + mr * ldx + kk) : ld_h8(X2 + mr * ldx2 + (kk - K1));
    if (P != nullptr) b = b + ld_h8(P + mr * ldp + kk);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const h8 a = ld_h8(W + (size_t)(n0 + 16 * nb + i) * K + kk);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[nb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) *reinterpret_cast<f4 *>(&part[wave][i][16 * nb + 4 * g]) = acc[nb];
  __syncthreads();
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = part[0][em][ec0 + j] + part[1][em][ec0 + j] + part[2][em][ec0 + j] + part[3][em][ec0 + j];
  epilogue8(v, e, eo, m0 + em, eok);
  if (eok) st8(Y + (m0 + em) * ldy + n0 + ec0, pack8f(v, __half()));
}

// ------------------------------------------------------------------------------------------------------------
// Wide linear (N >> M, K == 128): weight stationary.  A workgroup owns 128 output columns, a wave 32 of them with
// its 8 weight fragments in registers, and walks all the token rows.  DynamicConv's parameter generator
// (decoder_utils.py:608: Linear 128 -> 2*128*128 per query).  Rows of W map to MFMA rows so that a lane holds 8
// consecutive output columns (one 16-B store).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tl_wide_kernel(const __half *__restrict__ X, int ldx,
                                                      const __half *__restrict__ W, const float *__restrict__ bias,
                                                      __half *__restrict__ Y, long long ldy, int M, int N) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 128 + wave * 32;
  // MFMA row i = 4g' + r of fragment nb <-> output column n0 + 8g' + 4nb + r
  h8 a[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = n0 + 8 * (i >> 2) + 4 * nb + (i & 3);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) a[nb][kk] = ld_h8(W + (size_t)n * 128 + kk * 32 + g * 8);
  }
  float bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bs[j] = bias ? bias[n0 + 8 * g + j] : 0.f;
  for (int m0 = 0; m0 < M; m0 += 16) {
    const int mr = m0 + i < M ? m0 + i : M - 1;
    f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const h8 b = ld_h8(X + (size_t)mr * ldx + kk * 32 + g * 8);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[nb][kk], b, acc[nb], 0, 0, 0);
    }
    if (m0 + i < M) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[0][r] + bs[r];
        v[4 + r] = acc[1][r] + bs[4 + r];
      }
      st8(Y + (size_t)(m0 + i) * ldy + n0 + 8 * g, pack8f(v, __half()));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Split-K linear (K >> 128, N == 128): DynamicConv's out_layer (decoder_utils.py:624: Linear 49*128 -> 128 on the
// flattened RoI feature).  grid (row blocks of 16, K slices); a wave owns 32 output columns; partial sums go to a
// float32 workspace (slice, M, 128), summed by tl_finish_kernel which also runs the epilogue.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tl_splitk_kernel(const __half *__restrict__ X, long long ldx,
                                                        const __half *__restrict__ W, float *__restrict__ part, int M,
                                                        int K, int kslice) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const long long m0 = (long long)blockIdx.x * 16;
  const long long mr = m0 + i < M ? m0 + i : M - 1;
  const int kb = blockIdx.y * kslice, ke = min(kb + kslice, K);
  f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
  for (int k0 = kb; k0 < ke; k0 += 32) {
    const h8 b = ld_h8(X + mr * ldx + k0 + g * 8);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const h8 a = ld_h8(W + (size_t)(wave * 32 + 16 * nb + i) * K + k0 + g * 8);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[nb], 0, 0, 0);
    }
  }
  if (m0 + i < M) {
    float *dst = part + ((size_t)blockIdx.y * M + m0 + i) * 128 + wave * 32 + 4 * g;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) *reinterpret_cast<f4 *>(dst + 16 * nb) = acc[nb];
  }
}

__global__ __launch_bounds__(256) void tl_finish_kernel(const float *__restrict__ part, int nslice, Epilogue ep,
                                                        __half *__restrict__ Y, int ldy, int M) {
  const long long m = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c0 = (threadIdx.x & 15) * 8;
  const bool ok = m < M;
  const long long mr = ok ? m : M - 1;
  const EpiOperands eo = epilogue_prefetch(ep, m, c0, ok);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  for (int s = 0; s < nslice; ++s) {
    const float4 *p = reinterpret_cast<const float4 *>(part + ((size_t)s * M + mr) * 128 + c0);
    const float4 a = p[0], b = p[1];
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
    v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  epilogue8(v, ep, eo, m, ok);
  if (ok) st8(Y + m * ldy + c0, pack8f(v, __half()));
}

// ------------------------------------------------------------------------------------------------------------
// Self attention among the Q <= 512 queries of a sample (8 heads x 16 dims), from the packed q|k|v projection.
// grid (B, heads / 4, query groups of 16); wave = head.  Optional membership mask of the image RoI block
// (decoder_utils.py:745: the self attention runs among the queries of ONE view): key k is visible to query q when
// bit view[q] of member[k] is set, or when view[q] < 0 (a query no camera sees attends to everything; its output
// is discarded by the caller).
// ------------------------------------------------------------------------------------------------------------
typedef _Float16 sh4 __attribute__((ext_vector_type(4)));
typedef __fp16 shv4 __attribute__((ext_vector_type(4)));

template <int NT16>
__global__ __launch_bounds__(256) void tok_mha_kernel(const __half *__restrict__ qkv, int ld,
                                                      const unsigned char *__restrict__ member,
                                                      const signed char *__restrict__ view, __half *__restrict__ out,
                                                      int ldo, int Q, int heads, float scale_log2) {
  constexpr int KC = NT16 * 16;
  extern __shared__ __align__(16) unsigned char lds[];             // [4 heads][KC keys][K16 | V16] + member[KC]
  const int E = heads * 16;
  const int b = blockIdx.x, head0 = blockIdx.y * 4, qg = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const __half *base = qkv + (size_t)b * Q * ld;
  unsigned char *lmem = lds + 4 * KC * 64;
  for (int e = tid; e < KC * 16; e += 256) {
    const int key = e >> 4, p = e & 15;
    const int isv = p >> 3, hh = (p >> 1) & 3, half8 = p & 1;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (key < Q && head0 + hh < heads)
      val = *reinterpret_cast<const uint4 *>(base + (size_t)key * ld + (isv ? 2 * E : E) + (head0 + hh) * 16 + half8 * 8);
    *reinterpret_cast<uint4 *>(lds + ((hh * KC + key) * 64 + isv * 32 + half8 * 16)) = val;
  }
  if (member != nullptr)
    for (int e = tid; e < KC; e += 256) lmem[e] = e < Q ? member[(size_t)b * Q + e] : 0;
  __syncthreads();
  const int h = head0 + wave;
  if (h >= heads) return;
  const unsigned char *hb = lds + (size_t)wave * KC * 64;
  const int qi = qg * 16 + i;
  const int qc = qi < Q ? qi : Q - 1;
  const sh4 qf = *reinterpret_cast<const sh4 *>(base + (size_t)qc * ld + h * 16 + 4 * g);
  const int vq = (member != nullptr) ? (int)view[(size_t)b * Q + qc] : -1;
  f4 sc[NT16];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT16; ++t) {
    const sh4 kf = *reinterpret_cast<const sh4 *>(hb + (16 * t + i) * 64 + g * 8);
    f4 c = __builtin_amdgcn_mfma_f32_16x16x16f16(kf, qf, f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    unsigned mem4 = 0xFFFFFFFFu;
    if (member != nullptr && vq >= 0) mem4 = *reinterpret_cast<const unsigned *>(lmem + 16 * t + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * t + 4 * g + r;
      const bool vis = key < Q && (vq < 0 || ((mem4 >> (8 * r + vq)) & 1u));
      const float x = vis ? c[r] * scale_log2 : -INFINITY;
      c[r] = x;
      mx = fmaxf(mx, x);
    }
    sc[t] = c;
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float l = 0.f;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT16; ++t) {
    sh4 pf;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = exp2f(sc[t][r] - mx);
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
  if (qi < Q) {
    const float inv = 1.f / l;
    sh4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (_Float16)(acc[r] * inv);                 // O^T[dim 4g + r][query i]
    *reinterpret_cast<sh4 *>(out + ((size_t)b * Q + qi) * ldo + h * 16 + 4 * g) = o;
  }
}

// ------------------------------------------------------------------------------------------------------------
// DynamicConv core (decoder_utils.py:617-622), one workgroup per RoI:
//     F1 = relu(LN1(roi (49x128) . p1 (128x128)));   F2 = relu(LN2(F1 . p2))         -> F2 (49x128)
// computed transposed, F1^T = p1^T . roi^T, so that the accumulators of the first product (4 consecutive
// channels d of one spatial position per lane) are the B operand of the second one with the MFMA k index mapped to
// d = 32kk + 16t + 4g + r (k = 8g + 4t + r).  The generated parameters arrive in the layout that makes both A
// operands plain 16-B loads: params[q] = [ p1t (d, c) | p2t (e, permuted d) ] - the rows of the generating Linear
// are permuted once on the host (decoder_utils.DynamicConv.fused_params).  A wave owns 16 spatial positions.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dynconv_kernel(const __half *__restrict__ roi, const __half *__restrict__ params,
                                                      const __half *__restrict__ n1w, const __half *__restrict__ n1b,
                                                      const __half *__restrict__ n2w, const __half *__restrict__ n2b,
                                                      __half *__restrict__ out, float eps) {
  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int s = wave * 16 + i;                          // spatial position of this lane's column (49 valid)
  const int sr = s < 49 ? s : 48;
  const __half *rq = roi + ((size_t)q * 49 + sr) * 128;
  const __half *p1 = params + (size_t)q * 32768, *p2 = p1 + 16384;
  h8 xb[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) xb[kk] = ld_h8(rq + kk * 32 + g * 8);
  f4 acc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_h8(p1 + (16 * nb + i) * 128 + kk * 32 + g * 8), xb[kk], acc[nb], 0, 0, 0);
  }
  // LayerNorm over the 128 channels of position s: lane holds d = 16nb + 4g + r; the other 96 live in lanes i+16g'
  auto ln_relu = [&](const __half *w, const __half *b) {
    float sum = 0.f;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += acc[nb][r];
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.f / 128.f);
    float ss = 0.f;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[nb][r] - mean;
        ss += d * d;
      }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    const float inv = rsqrtf(ss * (1.f / 128.f) + eps);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const h4 wv = *reinterpret_cast<const h4 *>(w + 16 * nb + 4 * g);
      const h4 bv = *reinterpret_cast<const h4 *>(b + 16 * nb + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[nb][r] = fmaxf((acc[nb][r] - mean) * inv * (float)wv[r] + (float)bv[r], 0.f);
    }
  };
  ln_relu(n1w, n1b);
  h8 hb[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    h8 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      t[r] = (_Float16)acc[2 * kk][r];
      t[4 + r] = (_Float16)acc[2 * kk + 1][r];
    }
    hb[kk] = t;
  }
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_h8(p2 + (16 * nb + i) * 128 + kk * 32 + g * 8), hb[kk], acc[nb], 0, 0, 0);
  }
  ln_relu(n2w, n2b);
  if (s < 49) {
    __half *o = out + ((size_t)q * 49 + s) * 128;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      h4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (_Float16)acc[nb][r];
      *reinterpret_cast<h4 *>(o + 16 * nb + 4 * g) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Query initialisation (deepinteraction_decoder.py:242-253) + the learned positional embedding of the proposals
// (decoder_utils.py:16-32, BatchNorm folded, float32): one workgroup of 128 threads per query.
//   feat = bev[cell] + class_encoding[:, label] + bias;  pos = (cell % W + .5, cell // W + .5);
//   pe = W2 . relu(W1 . pos + b1) + b2
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pos_embed(const float px, const float py, const float *__restrict__ w1,
                                          const float *__restrict__ b1, const float *__restrict__ w2,
                                          const float *__restrict__ b2, float *hid /*LDS 128*/, int c, float &pe) {
  hid[c] = fmaxf(w1[2 * c] * px + w1[2 * c + 1] * py + b1[c], 0.f);
  __syncthreads();
  float a = b2[c];
  const float4 *wr = reinterpret_cast<const float4 *>(w2 + (size_t)c * 128);
#pragma unroll 8
  for (int j = 0; j < 32; ++j) {
    const float4 w = wr[j];
    a = fmaf(w.x, hid[4 * j], a);
    a = fmaf(w.y, hid[4 * j + 1], a);
    a = fmaf(w.z, hid[4 * j + 2], a);
    a = fmaf(w.w, hid[4 * j + 3], a);
  }
  pe = a;
}

__global__ __launch_bounds__(128) void query_init_kernel(const __half *__restrict__ bev /*(B,H,W,128)*/,
                                                         const long long *__restrict__ top, const __half *__restrict__ ce_w /*(128, ncls)*/,
                                                         const __half *__restrict__ ce_b, const float *__restrict__ w1,
                                                         const float *__restrict__ b1, const float *__restrict__ w2,
                                                         const float *__restrict__ b2, __half *__restrict__ feat,
                                                         __half *__restrict__ pe_out, float *__restrict__ pos_out,
                                                         long long *__restrict__ labels, int Q, int HW, int Wb, int ncls) {
  __shared__ float hid[128];
  const int q = blockIdx.x, b = blockIdx.y, c = threadIdx.x;
  const long long t = top[(size_t)b * Q + q];
  const int cls = (int)(t / HW), cell = (int)(t % HW);
  const float px = (float)(cell % Wb) + 0.5f, py = (float)(cell / Wb) + 0.5f;
  const float f = __half2float(bev[((size_t)b * HW + cell) * 128 + c]) + __half2float(ce_w[c * ncls + cls]) + __half2float(ce_b[c]);
  feat[((size_t)b * Q + q) * 128 + c] = __float2half(f);
  float pe;
  pos_embed(px, py, w1, b1, w2, b2, hid, c, pe);
  pe_out[((size_t)b * Q + q) * 128 + c] = __float2half(pe);
  if (c == 0) {
    pos_out[((size_t)b * Q + q) * 2] = px;
    pos_out[((size_t)b * Q + q) * 2 + 1] = py;
    labels[(size_t)b * Q + q] = cls;
  }
}

// ------------------------------------------------------------------------------------------------------------
// The prediction heads of one decoder stage (decoder_utils.py:498-581: per head Conv1d(C -> 64) + BN + ReLU,
// Conv1d(64 -> classes)), BatchNorm folded, all heads at once: hidden = relu(W1 . [x1 ; x2] + b1) (NH*64 wide, on
// the matrix cores), then the tiny per-head second layers in float32.  Also what follows every call in
// deepinteraction_decoder.py: `center += query_pos` (:265,:288), the on-the-image merge with the first stage's
// result (:292-295) and the placement at column offset l*Q of the (B, classes, L*Q) output tensors (:304-311).
// A workgroup owns 16 queries.  Outputs are float32 (box geometry).
// ------------------------------------------------------------------------------------------------------------
constexpr int kMaxHeads = 8;
struct HeadOut {
  float *out[kMaxHeads];          // (B, cls_h, ldo) each; this stage writes columns [col0, col0 + Q)
  const float *first[kMaxHeads];  // (B, cls_h, Q) of the first stage, or null
  int cls[kMaxHeads];
  int row0[kMaxHeads];            // first row of head h in the stacked second layer
  int nheads, center_head;
};

// grid (query blocks of 16, B, heads): a workgroup evaluates ONE head for 16 queries (wave w = 16 of its 64 hidden
// channels), so a stage is ~80 small workgroups instead of 13 long ones.
__global__ __launch_bounds__(256) void pred_head_kernel(const __half *__restrict__ x1, const __half *__restrict__ x2, int K,
                                                        const __half *__restrict__ w1, const float *__restrict__ b1,
                                                        const float *__restrict__ w2 /*(rows, 64)*/, const float *__restrict__ b2,
                                                        const float *__restrict__ qpos /*(B,Q,2)*/,
                                                        const unsigned char *__restrict__ keep /*(B,Q) or null*/,
                                                        HeadOut ho, float *__restrict__ pos_out /*(B,Q,2) or null*/,
                                                        int Q, int ldo, int col0) {
  __shared__ float hid[16][68];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, q0 = blockIdx.x * 16, h = blockIdx.z;
  const int qr = q0 + i < Q ? q0 + i : Q - 1;
  const size_t row = (size_t)b * Q + qr;
  const int n0 = h * 64 + wave * 16;                       // this wave's 16 hidden channels (rows of the stacked W1)
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 32) {
    const int kk = k0 + g * 8;
    const h8 bx = kk < 128 ? ld_h8(x1 + row * 128 + kk) : ld_h8(x2 + row * 128 + kk - 128);
    const h8 a = ld_h8(w1 + (size_t)(n0 + i) * K + kk);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bx, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) hid[i][wave * 16 + 4 * g + r] = fmaxf(acc[r] + b1[n0 + 4 * g + r], 0.f);
  __syncthreads();
  // second layer of this head: thread -> (query, class row)
  const int ncls = ho.cls[h], r0 = ho.row0[h];
  for (int e = tid; e < 16 * ncls; e += 256) {
    const int m = e / ncls, cidx = e - m * ncls;
    const int q = q0 + m;
    if (q >= Q) continue;
    const int o = r0 + cidx;
    const float4 *wr = reinterpret_cast<const float4 *>(w2 + (size_t)o * 64);
    float a = b2[o];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float4 w = wr[j];
      a = fmaf(w.x, hid[m][4 * j], a);
      a = fmaf(w.y, hid[m][4 * j + 1], a);
      a = fmaf(w.z, hid[m][4 * j + 2], a);
      a = fmaf(w.w, hid[m][4 * j + 3], a);
    }
    if (h == ho.center_head) a += qpos[((size_t)b * Q + q) * 2 + cidx];
    if (keep != nullptr && !keep[(size_t)b * Q + q]) a = ho.first[h][((size_t)b * ncls + cidx) * Q + q];
    ho.out[h][((size_t)b * ncls + cidx) * ldo + col0 + q] = a;
    if (h == ho.center_head && pos_out != nullptr) pos_out[((size_t)b * Q + q) * 2 + cidx] = a;
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

static int check_epilogue(const Epilogue &e, int N) {
  if ((e.ln1_w || e.ln2_w) && N != 128) {
    set_error("a LayerNorm epilogue needs N == 128, got %d", N);
    return DI_ERR_ARG;
  }
  return DI_OK;
}

}  // namespace tok
}  // namespace di

extern "C" {

/* Y = epilogue((X [; X2] + P) . W^T): see include/deepinteraction_hip.h */
int di_token_linear(const void *x, int ldx, const void *x2, int ldx2, int k1, const void *p, int ldp, const void *w,
                    const float *bias, int act1, const void *res1, int ldr1, const void *ln1_w, const void *ln1_b,
                    int act2, const void *res2, int ldr2, const void *ln2_w, const void *ln2_b, float eps,
                    const void *keep, void *y, int ldy, int M, int N, int K, void *workspace, void *stream) {
  using namespace di::tok;
  DI_REQUIRE(M > 0 && N > 0 && K > 0 && x && w && y, "bad linear shape M=%d N=%d K=%d", M, N, K);
  DI_REQUIRE(N % 128 == 0 && K % 128 == 0, "N=%d and K=%d must be multiples of 128", N, K);
  DI_REQUIRE(x2 == nullptr || (k1 > 0 && k1 < K && k1 % 32 == 0), "bad concat split k1=%d", k1);
  DI_REQUIRE((ln1_w == nullptr) == (ln1_b == nullptr) && (ln2_w == nullptr) == (ln2_b == nullptr), "LayerNorm needs weight and bias");
  Epilogue e{bias, act1, (const __half *)res1, ldr1, (const __half *)ln1_w, (const __half *)ln1_b, act2,
             (const __half *)res2, ldr2, (const __half *)ln2_w, (const __half *)ln2_b, eps, (const unsigned char *)keep};
  int rc = check_epilogue(e, N);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const bool plain = !x2 && !p && !act1 && !res1 && !ln1_w && !act2 && !res2 && !ln2_w && !keep;
  if (K == 128 && N >= 2048 && plain) {            // weight-stationary: DynamicConv's parameter generator
    hipLaunchKernelGGL(tl_wide_kernel, dim3(N / 128), dim3(256), 0, s, (const __half *)x, ldx, (const __half *)w, bias,
                       (__half *)y, (long long)ldy, M, N);
    return di::check_launch("token_linear(wide)");
  }
  if (K >= 2048 && N == 128 && !x2 && !p) {        // split-K: DynamicConv's out_layer
    DI_REQUIRE(workspace != nullptr, "split-K linear needs a workspace of di_token_linear_workspace_bytes()");
    const int kslice = 448;
    const int ns = (K + kslice - 1) / kslice;
    hipLaunchKernelGGL(tl_splitk_kernel, dim3((M + 15) / 16, ns), dim3(256), 0, s, (const __half *)x, (long long)ldx,
                       (const __half *)w, (float *)workspace, M, K, kslice);
    hipLaunchKernelGGL(tl_finish_kernel, dim3((M + 15) / 16), dim3(256), 0, s, (const float *)workspace, ns, e,
                       (__half *)y, ldy, M);
    return di::check_launch("token_linear(split-K)");
  }
  hipLaunchKernelGGL(tl_rowblock_kernel, dim3((M + 15) / 16, N / 128), dim3(256), 0, s, (const __half *)x, ldx,
                     (const __half *)x2, ldx2, x2 ? k1 : K, (const __half *)p, ldp, (const __half *)w, e, (__half *)y, ldy,
                     M, N, K);
  return di::check_launch("token_linear");
}

long long di_token_linear_workspace_bytes(int M, int N, int K) {
  if (K >= 2048 && N == 128) return (long long)((K + 447) / 448) * M * 128 * 4;
  return 0;
}

int di_token_mha(const void *qkv, int ld, const void *member, const void *view, void *out, int ldo, int B, int Q,
                 int heads, float scale, void *stream) {
  using namespace di::tok;
  DI_REQUIRE(B > 0 && Q > 0 && Q <= 512 && heads > 0 && heads % 4 == 0, "bad attention shape B=%d Q=%d heads=%d (Q <= 512)", B, Q, heads);
  DI_REQUIRE((member == nullptr) == (view == nullptr), "mask needs member and view");
  const float sl2 = scale * 1.4426950408889634f;
  const dim3 grid(B, heads / 4, (Q + 15) / 16), blk(256);
  hipStream_t s = (hipStream_t)stream;
#define DI_TMHA(NT)                                                                                                  \
  do {                                                                                                               \
    constexpr int lds = 4 * NT * 16 * 64 + NT * 16;                                                                  \
    static di::LdsRaised raised;                                                                                     \
    if (int rc = di::ensure_lds(raised, (const void *)tok_mha_kernel<NT>, lds)) return rc;                           \
    hipLaunchKernelGGL((tok_mha_kernel<NT>), grid, blk, lds, s, (const __half *)qkv, ld, (const unsigned char *)member, \
                       (const signed char *)view, (__half *)out, ldo, Q, heads, sl2);                                \
  } while (0)
  if (Q <= 208) DI_TMHA(13);
  else if (Q <= 400) DI_TMHA(25);
  else DI_TMHA(32);
#undef DI_TMHA
  return di::check_launch("token_mha");
}

int di_dynconv_fwd(const void *roi, const void *params, const void *n1w, const void *n1b, const void *n2w, const void *n2b,
                   void *out, int R, float eps, void *stream) {
  DI_REQUIRE(R > 0 && roi && params && out, "bad DynamicConv call R=%d", R);
  hipLaunchKernelGGL(di::tok::dynconv_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const __half *)roi,
                     (const __half *)params, (const __half *)n1w, (const __half *)n1b, (const __half *)n2w,
                     (const __half *)n2b, (__half *)out, eps);
  return di::check_launch("dynconv_fwd");
}

int di_roi_select(const int *on, const float *rect, float *rois, void *view, void *member, void *keep, float *on_img,
                  int B, int V, int Q, void *stream) {
  DI_REQUIRE(B > 0 && Q > 0 && rect && rois, "bad roi_select shape");
  DI_REQUIRE(on == nullptr || (V > 0 && V <= 8 && view && member && keep && on_img), "image mode needs V <= 8 and all outputs");
  hipLaunchKernelGGL(di::tok::roi_select_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, on, rect, rois,
                     (signed char *)view, (unsigned char *)member, (unsigned char *)keep, on_img, V, Q);
  return di::check_launch("roi_select");
}

int di_query_init(const void *bev, const long long *top, const void *ce_w, const void *ce_b, const float *w1,
                  const float *b1, const float *w2, const float *b2, void *feat, void *pe, float *pos, long long *labels,
                  int B, int Q, int Hb, int Wb, int ncls, void *stream) {
  DI_REQUIRE(B > 0 && Q > 0 && Hb > 0 && Wb > 0 && ncls > 0, "bad query_init shape");
  hipLaunchKernelGGL(di::tok::query_init_kernel, dim3(Q, B), dim3(128), 0, (hipStream_t)stream, (const __half *)bev, top,
                     (const __half *)ce_w, (const __half *)ce_b, w1, b1, w2, b2, (__half *)feat, (__half *)pe, pos, labels, Q,
                     Hb * Wb, Wb, ncls);
  return di::check_launch("query_init");
}

int di_pred_heads(const void *x1, const void *x2, const void *w1, const float *b1, const float *w2, const float *b2,
                  const float *qpos, const void *keep, float *const *out, const float *const *first, const int *cls,
                  int nheads, int center_head, float *pos_out, int B, int Q, int ldo, int col0, void *stream) {
  using namespace di::tok;
  DI_REQUIRE(nheads > 0 && nheads <= kMaxHeads, "unsupported number of heads %d", nheads);
  HeadOut ho;
  int rows = 0;
  for (int h = 0; h < kMaxHeads; ++h) {
    ho.out[h] = h < nheads ? out[h] : nullptr;
    ho.first[h] = (h < nheads && first) ? first[h] : nullptr;
    ho.cls[h] = h < nheads ? cls[h] : 0;
    ho.row0[h] = rows;
    if (h < nheads) rows += cls[h];
  }
  ho.nheads = nheads;
  ho.center_head = center_head;
  DI_REQUIRE(keep == nullptr || first != nullptr, "the on-the-image merge needs the first stage's outputs");
  const int K = x2 ? 256 : 128;
  const dim3 grid((Q + 15) / 16, B, nheads), blk(256);
  hipLaunchKernelGGL(pred_head_kernel, grid, blk, 0, (hipStream_t)stream, (const __half *)x1, (const __half *)x2, K,
                     (const __half *)w1, b1, w2, b2, qpos, (const unsigned char *)keep, ho, pos_out, Q, ldo, col0);
  return di::check_launch("pred_heads");
}

}  // extern "C"
