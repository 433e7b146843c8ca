// The decoder layer's 200 x 32 400 cross attention with float32-accurate LOGITS on the fp16 matrix cores (round 3).
//
// Reference: decoder_utils.py:96-103 (`multihead_attn(query + query_pos, key + key_pos, key + key_pos)`) and
// :421-485 (in-projection, q k^T, softmax, . v).  With the positional embeddings in, the logits of this attention reach
// |s| ~ 500 (random-init weights: std 90), so a 2^-11 relative rounding of K or q moves a logit by 0.1 and the soft-max
// weights by 10 % - measured with the oracle (tests/tools/fp16_error_budget.py): K or q in fp16 alone puts 5 % of the
// box outputs beyond 1e-3, while V and the probabilities in fp16 stay below 5e-4.  So:
//
//   * `kv_project_kernel`: K = Wk x + (Wk kpe + bk), V = Wv x + (Wv kpe + bv) from the fp16 BEV map x with FLOAT32
//     weights, as two fp16 MFMA passes over W = Whi + 2^-11 Wlo (both halves exact fp16 numbers, the low half pre-scaled
//     so that it stays out of the fp16 subnormals), fp32 accumulation, the position term (the BEV grid is constant:
//     computed once in float32 on the host side) added as a per-key bias; K leaves as the pair Khi + 2^-11 Klo, V as fp16:
//     rows [Khi | Klo | V] of 3E halfs.
//   * `mha_decode_x_kernel`: S^T = Khi qhi + 2^-11 (Klo qhi + Khi qlo) - three 16x16x16 MFMAs per 16-key tile instead of
//     one, the float32 query split in registers - then the online soft-max and O^T += V^T P^T exactly as the fp16 form
//     (csrc/decoder.hip mha_decode_mfma_kernel); one partial state [m, l, O[16]] per (sample, head, query, key range),
//     merged by a DI_TOK_COMBINE step of the token program that follows (csrc/token32.hip).
#include "di_common.h"

namespace di {
namespace xa {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

__device__ __forceinline__ h8 ld_h8(const __half *p) { return __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(p)); }
__device__ __forceinline__ f4 ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }

// ------------------------------------------------------------------------------------------------------------
// K / V projection of the BEV tokens.  X (B*S, 128) fp16 (the channels-last map); wp = the (256, 128) float32 weight
// (rows 0..127: K, 128..255: V) as hi / lo fp16 in MFMA fragment order (`ops.pack_kv_weight`); kbias, vbias (S, 128)
// float32; out (B*S, 384) fp16.  A workgroup owns 128 rows; WAVE w computes output channels [64w, 64w + 64) for all of
// them, so its 32 KiB of weight fragments are read once (contiguous KiB loads) and stay in registers, and the rows
// stream through as B operands (the first version had every wave walk all 256 channels of 32 rows: 128 KiB of
// row-strided weight loads per wave, 512 KiB per CU through a 64 B/clk texture pipe).  The rows of a tile pair map to
// MFMA rows so that a lane ends with 8 consecutive channels (16-B stores).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kv_project_kernel(const __half *__restrict__ X, const __half *__restrict__ wp,
                                                         const float *__restrict__ kbias, const float *__restrict__ vbias,
                                                         __half *__restrict__ out, long long rows, int S) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const long long row0 = (long long)blockIdx.x * 128;
  h8 a[2][2][8];                                          // [pt][nb][2kk + h]
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const __half *blk = wp + (size_t)(2 * (2 * wave + pl) + nb) * 4096 + lane * 8;
#pragma unroll
      for (int q = 0; q < 8; ++q) a[pl][nb][q] = ld_h8(blk + q * 512);
    }
  const bool isk = wave < 2;
  const float *bias = isk ? kbias : vbias;
  const int cb = (isk ? 64 * wave : 64 * (wave - 2)) + 8 * g;         // first of the lane's 8 channels (pl = 0) inside K or V
  h8 xb[4], xn[4];
  f4 bb[4], bn[4];
  auto fetch = [&](int gi, h8 (&x)[4], f4 (&bq)[4]) {
    const long long r = min(row0 + 16 * gi + i, rows - 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) x[kk] = ld_h8(X + r * 128 + 32 * kk + 8 * g);
    const float *bp = bias + (size_t)(r % S) * 128 + cb;
    bq[0] = ld4(bp); bq[1] = ld4(bp + 4); bq[2] = ld4(bp + 32); bq[3] = ld4(bp + 36);
  };
  fetch(0, xb, bb);
  for (int gi = 0; gi < 8; ++gi) {
    if (row0 + 16 * gi >= rows) break;
    if (gi + 1 < 8) fetch(gi + 1, xn, bn);
    const long long row = row0 + 16 * gi + i;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      f4 hi[2], lo[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        hi[nb] = f4{0.f, 0.f, 0.f, 0.f};
        lo[nb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          lo[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[pl][nb][2 * kk + 1], xb[kk], lo[nb], 0, 0, 0);
          hi[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[pl][nb][2 * kk], xb[kk], hi[nb], 0, 0, 0);
        }
      }
      if (row < rows) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fmaf(lo[0][r], kLoInv, hi[0][r]) + bb[2 * pl][r];
          v[4 + r] = fmaf(lo[1][r], kLoInv, hi[1][r]) + bb[2 * pl + 1][r];
        }
        h8 oh, ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          oh[j] = (_Float16)v[j];
          ol[j] = (_Float16)((v[j] - (float)oh[j]) * kLoScale);
        }
        __half *o = out + row * 384 + cb + 32 * pl;
        if (isk) {
          *reinterpret_cast<h8 *>(o) = oh;
          *reinterpret_cast<h8 *>(o + 128) = ol;
        } else {
          *reinterpret_cast<h8 *>(o + 256) = oh;
        }
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      xb[kk] = xn[kk];
      bb[kk] = bn[kk];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Attention over key ranges.  grid (key ranges, query splits, B), 16 wavefronts (two per head); stages of 64 keys,
// double buffered in LDS as [head][key][Khi16] | [head][key][Klo16] (48-B rows: conflict-free 8-B fragment reads) |
// [head][key][V16] (32-B rows for the transposing reads); the rows of the next stage are in flight in registers while
// the current one is multiplied.  q (B,Q,E) float32 UNSCALED; kx (B,S,3E) fp16 = [Khi | Klo | V].
// ------------------------------------------------------------------------------------------------------------
constexpr int kHD = 16;
constexpr int KS = 64;         // keys per stage
constexpr int NT16 = KS / 16;
constexpr int KROW = 48, VROW = 32;
constexpr int WPH = 2;         // wavefronts per head
constexpr int MAXQG = 2;       // query groups of 16 per wavefront
constexpr int NTH = 1024;
constexpr int HEADS = 8;
constexpr int STAGE_BYTES = HEADS * KS * (2 * KROW + VROW);
constexpr int OFF_LO = HEADS * KS * KROW, OFF_V = 2 * HEADS * KS * KROW;

__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__global__ __launch_bounds__(NTH) void mha_decode_x_kernel(const float *__restrict__ q, const __half *__restrict__ kx,
                                                           float *__restrict__ part, int B, int Q, int S,
                                                           float scale2, int range_keys) {
  extern __shared__ __align__(16) unsigned char lds[];   // 2 x STAGE_BYTES
  constexpr int Hh = HEADS, E = HEADS * kHD;
  const int range = blockIdx.x, b = blockIdx.z, nrange = gridDim.x;
  const int r0 = range * range_keys, r1 = min(r0 + range_keys, S);
  const int nstage = (r1 - r0 + KS - 1) / KS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  // staging: 16-B piece p of a key row (48 pieces): p / 16 = Khi | Klo | V, (p >> 1) & 7 = head, p & 1 = half of 16 dims
  constexpr int PPK = 6 * Hh;
  constexpr int NPT = KS * PPK / NTH;                    // 3 pieces per thread
  uint4 stage_regs[NPT];
  auto fetch = [&](int st) {
    const int s0 = r0 + st * KS;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int e = tid + j * NTH;
      const int key = e / PPK, p = e - key * PPK;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (s0 + key < r1) val = *reinterpret_cast<const uint4 *>(kx + ((size_t)b * S + s0 + key) * 3 * E + p * 8);
      stage_regs[j] = val;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int e = tid + j * NTH;
      const int key = e / PPK, p = e - key * PPK;
      const int seg = p >> 4, hh = (p >> 1) & 7, half8 = p & 1;
      unsigned char *dst = lds + buf * STAGE_BYTES +
                           (seg == 2 ? OFF_V + (hh * KS + key) * VROW : seg * OFF_LO + (hh * KS + key) * KROW);
      *reinterpret_cast<uint4 *>(dst + half8 * 16) = stage_regs[j];
    }
  };
  fetch(0);
  commit(0);
  __syncthreads();

  const int nqg = (Q + 15) / 16;
  const int h = wave % Hh, part_of_head = wave / Hh;       // 8 heads x 2 waves
  const int qg0 = blockIdx.y + part_of_head * gridDim.y, qgstep = WPH * gridDim.y;
  h4 qhi[MAXQG], qlo[MAXQG];
  f4 acc[MAXQG], lacc[MAXQG];
  float m[MAXQG];
  const h4 ones = {(_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1};
#pragma unroll
  for (int j = 0; j < MAXQG; ++j) {
    const int qg = qg0 + j * qgstep;
    const int qi = min(qg * 16 + i, Q - 1);
    const f4 qf = ld4(q + ((size_t)b * Q + qi) * E + h * kHD + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      qhi[j][r] = (_Float16)qf[r];
      qlo[j][r] = (_Float16)((qf[r] - (float)qhi[j][r]) * kLoScale);
    }
    acc[j] = f4{0.f, 0.f, 0.f, 0.f};
    lacc[j] = f4{0.f, 0.f, 0.f, 0.f};
    m[j] = -INFINITY;
  }
  for (int st = 0; st < nstage; ++st) {
    if (st + 1 < nstage) fetch(st + 1);
    const int ns = min(KS, r1 - (r0 + st * KS));           // keys of this stage (ragged only at the very end)
    const unsigned char *kb = lds + (st & 1) * STAGE_BYTES + (size_t)h * KS * KROW;
    const unsigned char *vb = lds + (st & 1) * STAGE_BYTES + OFF_V + (size_t)h * KS * VROW;
#pragma unroll
    for (int j = 0; j < MAXQG; ++j) {
      if (qg0 + j * qgstep >= nqg) break;                  // wave-uniform
      f4 sc[NT16];
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < NT16; ++t) {
        const h4 kh = *reinterpret_cast<const h4 *>(kb + (16 * t + i) * KROW + g * 8);
        const h4 kl = *reinterpret_cast<const h4 *>(kb + OFF_LO + (16 * t + i) * KROW + g * 8);
        f4 lo = __builtin_amdgcn_mfma_f32_16x16x16f16(kl, qhi[j], f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_16x16x16f16(kh, qlo[j], lo, 0, 0, 0);
        f4 c = __builtin_amdgcn_mfma_f32_16x16x16f16(kh, qhi[j], f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = fmaf(lo[r], kLoInv, c[r]);      // unscaled logit, float32-accurate
        if (ns < KS) {
          asm volatile("" ::: "memory");
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * t + 4 * g + r >= ns) c[r] = -INFINITY;
        }
        mx = max3_raw(c[2], c[3], max3_raw(c[0], c[1], mx));
        sc[t] = c;
      }
      mx = max3_raw(mx, __shfl_xor(mx, 16), mx);
      mx = max3_raw(mx, __shfl_xor(mx, 32), mx) * scale2;                  // scale2 > 0: max(s c) = s max(c)
      const float mn = max3_raw(m[j], mx, mx);                              // finite: a stage has at least one key
      const float a = __builtin_amdgcn_exp2f(m[j] - mn);
      lacc[j] *= a;
      acc[j] *= a;
      m[j] = mn;
#pragma unroll
      for (int t = 0; t < NT16; ++t) {
        h4 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) pf[r] = (_Float16)__builtin_amdgcn_exp2f(fmaf(sc[t][r], scale2, -mn));
        lacc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pf, lacc[j], 0, 0, 0);
        const hv4 vt = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (hv4 __attribute__((address_space(3))) *)(vb + (16 * t + 4 * g + (i >> 2)) * VROW + (i & 3) * 8));
        h4 vf;
        vf[0] = vt[0]; vf[1] = vt[1]; vf[2] = vt[2]; vf[3] = vt[3];
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf, acc[j], 0, 0, 0);
      }
    }
    if (st + 1 < nstage) commit((st + 1) & 1);
    __syncthreads();
  }
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

// Merge of the key-range states: one wavefront per (sample, head, query); lane = (range slice of 4, head channel);
// out (B*Q, 128) float32.  (Inside the 13-workgroup token program this merge cost 39 us: 590 KB of states per
// workgroup; as its own launch every CU takes a share.)
__global__ __launch_bounds__(256) void combine_kernel(const float *__restrict__ part, float *__restrict__ out, int B,
                                                      int Q, int nrange) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= B * HEADS * Q) return;
  const int lane = threadIdx.x & 63, d = lane & 15, cs = lane >> 4;
  const int qi = w % Q, h = (w / Q) % HEADS, b = w / (Q * HEADS);
  const float *base = part + (((size_t)b * HEADS + h) * Q + qi) * nrange * (kHD + 2);
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
  if (cs == 0) out[((size_t)b * Q + qi) * HEADS * kHD + h * kHD + d] = O / L;
}

static void plan(int B, int Q, int S, int &qsplit, int &nrange, int &range_keys) {
  const int nqg = (Q + 15) / 16;
  qsplit = (nqg + MAXQG * WPH - 1) / (MAXQG * WPH);
  const int nstage_total = (S + KS - 1) / KS;
  int want = (256 + qsplit * B - 1) / (qsplit * B);
  want = want < 1 ? 1 : (want > nstage_total ? nstage_total : want);
  const int stages_per_range = (nstage_total + want - 1) / want;
  range_keys = stages_per_range * KS;
  nrange = (S + range_keys - 1) / range_keys;
}

}  // namespace xa
}  // namespace di

extern "C" {

int di_kv_project_fwd(const void *x, const void *w_packed, const float *kbias, const float *vbias, void *out, int B, int S,
                      void *stream) {
  DI_REQUIRE(x && w_packed && kbias && vbias && out && B > 0 && S > 0, "bad kv_project call B=%d S=%d", B, S);
  const long long rows = (long long)B * S;
  hipLaunchKernelGGL(di::xa::kv_project_kernel, dim3((unsigned)((rows + 127) / 128)), dim3(256), 0, (hipStream_t)stream,
                     (const __half *)x, (const __half *)w_packed, kbias, vbias, (__half *)out, rows, S);
  return di::check_launch("kv_project_fwd");
}

int di_mha_decode_x_ranges(int B, int Q, int S) {
  int qsplit, nrange, range_keys;
  di::xa::plan(B, Q, S, qsplit, nrange, range_keys);
  return nrange;
}

int di_mha_decode_x_fwd(const float *q, const void *kx, float *scratch, float *out, int B, int Q, int S, float scale,
                        void *stream) {
  using namespace di::xa;
  DI_REQUIRE(q && kx && scratch && B > 0 && Q > 0 && S > 0 && scale > 0.f, "bad attention shape (scale must be positive)");
  int qsplit, nrange, range_keys;
  plan(B, Q, S, qsplit, nrange, range_keys);
  static di::LdsRaised raised;
  if (int rc = di::ensure_lds(raised, (const void *)mha_decode_x_kernel, 2 * STAGE_BYTES)) return rc;
  hipLaunchKernelGGL(mha_decode_x_kernel, dim3(nrange, qsplit, B), dim3(NTH), 2 * STAGE_BYTES, (hipStream_t)stream, q,
                     (const __half *)kx, scratch, B, Q, S, scale * 1.4426950408889634f, range_keys);
  if (out != nullptr)
    hipLaunchKernelGGL(combine_kernel, dim3((B * HEADS * Q + 3) / 4), dim3(256), 0, (hipStream_t)stream, scratch, out, B, Q, nrange);
  return di::check_launch("mha_decode_x_fwd");
}

}  // extern "C"
