// Token-level kernels of the MMPI decoder with FLOAT32 accuracy for gfx950 (round 3).
//
// The decoder works on B*Q <= ~1000 query tokens of 128 channels (reference decoder_utils.py:35-113 decoder layer,
// :498-581 prediction FFN, :584-629 DynamicConv, :632-841 RoI blocks).  The work is tiny (a few MFLOP per linear layer)
// and numerically touchy: with random-init weights the 200 x 32 400 cross attention has logits of magnitude ~500 and
// the DynamicConv chain amplifies a 2^-11 perturbation of any operand to 1e-2 of the box outputs
// (tests/tools/fp16_error_budget.py).  So the token state, the RoI features, the generated DynamicConv parameters and
// every weight of this path carry float32 accuracy; only the BEV / image MAPS the tokens gather from stay fp16.
//
// Matrix products: the float32 MFMA (16x16x4) runs at 1/16 of the fp16 rate and made the first version of these
// kernels MFMA-bound (program 41 us, generator 25 us, DynamicConv 25 us).  Every operand is therefore SPLIT into two
// fp16 numbers, x = hi + lo / 2048 (hi = fp16(x), lo = fp16((x - hi) * 2048): the low half pre-scaled so that it never
// lands in the fp16 subnormals), and a product runs as three fp16 MFMAs (16x16x32) with float32 accumulation:
//     a . b  =  ahi . bhi  +  (ahi . blo + alo . bhi) / 2048          (+ alo . blo / 2^22, dropped: 2^-22 relative)
// - 16/3 times the float32 MFMA rate, weights pre-split once on the host, activations split in registers, products
// that feed another kernel's matrix operand (generated parameters, DynamicConv output) written as hi / lo pairs.
//
// Launch count and dependent round trips are what the path is bound by, so the token-parallel parts are ONE kernel
// per dependency level: `program_kernel` gives a workgroup 16 token rows in LDS and runs a short PROGRAM of steps on
// them (load / self attention among the sample's queries / merge of the cross attention's partial states / linear /
// residual + LayerNorm / store / prediction heads); a linear step issues ALL weight fragments of up to four (tile,
// K-chunk) items before its first MFMA, and the first group of the NEXT linear step before the barrier that ends the
// current step; consecutive loads are one round trip; the attention keeps four key tiles in flight.  What cannot be
// row-parallel stays its own kernel: the DynamicConv parameter generator (weight stationary, 128 -> 32 768 per
// query), the DynamicConv core (one workgroup per RoI) and the split-K out_layer (6 272 -> 128).
//
// All GEMMs run TRANSPOSED, Y^T = W . X^T: the A operand is 8 consecutive k of one weight row per lane, the B operand
// 8 consecutive k of one token row, and a lane ends with 4 consecutive output channels of one token.
#include <stdlib.h>
#include <string.h>

#include "di_common.h"

namespace di {
namespace t32 {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

__device__ __forceinline__ f4 ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ h8 ld_h8(const __half *p) { return __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(p)); }
__device__ __forceinline__ f4 mfma4(f4 a, f4 b, f4 c) {       // float32 MFMA, 4 consecutive k (the attention's 16-dim heads)
#pragma unroll
  for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], c, 0, 0, 0);
  return c;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// 8 floats -> the hi / lo fp16 pair
struct HL {
  h8 hi, lo;
};
__device__ __forceinline__ void split1(float x, _Float16 &hi, _Float16 &lo) {
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * kLoScale);
}
__device__ __forceinline__ HL split8(f4 a, f4 b) {
  HL r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    _Float16 h, l;
    split1(a[j], h, l);
    r.hi[j] = h; r.lo[j] = l;
    split1(b[j], h, l);
    r.hi[4 + j] = h; r.lo[4 + j] = l;
  }
  return r;
}
// one k-step (32) of the three-pass product: (ah, al) += (whi, wlo) x (b.hi, b.lo)
__device__ __forceinline__ void mfma3(h8 whi, h8 wlo, const HL &b, f4 &ah, f4 &al) {
  al = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, b.lo, al, 0, 0, 0);
  al = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, b.hi, al, 0, 0, 0);
  ah = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, b.hi, ah, 0, 0, 0);
}

constexpr int TM = 16;          // token rows per workgroup
constexpr int NW = 8;           // wavefronts per workgroup
constexpr int NTH = NW * 64;
constexpr int LDW = 516;        // floats per LDS row: 512 + 4 (float4 reads of 16 rows spread over the banks)
constexpr int NBUF = 3;
constexpr int LDS_BYTES = NBUF * TM * LDW * 4 + 2 * TM * (512 + 8) * 2;   // row buffers + the hi / lo image of a linear step's input

// ------------------------------------------------------------------------------------------------------------
// The steps of a program.  `src` / `dst` / `aux` name LDS row buffers (0..2); widths are in floats.
// ------------------------------------------------------------------------------------------------------------
struct Step {           // mirrors di_tok_step of include/deepinteraction_hip.h
  int kind, src, dst, aux;
  int K, N, a, b;
  float f;
  int role_lo, role_hi;   // the step runs in the workgroups whose role (blockIdx.z) lies in [role_lo, role_hi]
  int rt, rc, nch;        // LINEAR: tiles / K-chunks added per role to the packed weight's block index; chunks of the weight
  const void *p0, *p1, *p2, *p3;
  long long ld0, ld1;
  long long roff;         // elements added to p0 per role (LOAD / LOAD_PARTS / STORE)
};
struct Heads {          // mirrors di_tok_heads
  const float *w2, *b2, *qpos;
  const unsigned char *keep;
  float *pos_out;
  float *out[DI_TOK_MAX_HEADS];
  const float *first[DI_TOK_MAX_HEADS];
  int cls[DI_TOK_MAX_HEADS];
  int nheads, center_head, ldo, col0;
  const float *qpos2;     // DeepInteraction++ look-forward update: pos2_out = (centre head's raw output) + qpos2
  float *pos2_out;
};
struct Program {
  int n;
  Step s[DI_TOK_MAX_STEPS];
};

// The program arrives as a kernel argument.  The kernarg segment is host memory: the first version indexed it per
// step (`prog.s[si].field` = scalar loads over the fabric) and paid 1 500 shader cycles of fixed cost per step, 7 000
// per linear step (measured with the stamps below).  So the whole argument block is copied to LDS once, by all
// threads in one round trip, and a step's fields are pulled into SGPRs (readfirstlane) when the step starts.
__device__ __forceinline__ Step fetch_step(const Step *ls) {
  Step s;
  const unsigned *src = reinterpret_cast<const unsigned *>(ls);
  unsigned *dst = reinterpret_cast<unsigned *>(&s);
#pragma unroll
  for (int j = 0; j < (int)(sizeof(Step) / 4); ++j) dst[j] = __builtin_amdgcn_readfirstlane(src[j]);
  return s;
}

// K_LOAD: dst[r][a + c] = (p0 + role * roff)[m * ld0 + c] (+ p1[m * ld1 + c]),  c < K.  Up to three CONSECUTIVE load
// steps run as one phase: every global load of all of them is issued before the first LDS write (one round trip).
constexpr int kMaxLoadRun = 3;
__device__ __forceinline__ void steps_load(const Program &prog, int si, int nrun, float *buf, long long m0, int rows, int tid,
                                           int role) {
  Step s[kMaxLoadRun];
  bool on[kMaxLoadRun];
#pragma unroll
  for (int q = 0; q < kMaxLoadRun; ++q) {
    s[q] = fetch_step(&prog.s[min(si + q, DI_TOK_MAX_STEPS - 1)]);
    on[q] = q < nrun && role >= s[q].role_lo && role <= s[q].role_hi;
  }
  f4 v[kMaxLoadRun][4];
#pragma unroll
  for (int q = 0; q < kMaxLoadRun; ++q) {
    if (!on[q]) continue;
    const float *x = (const float *)s[q].p0 + (role - s[q].role_lo) * s[q].roff, *p = (const float *)s[q].p1;
    const int k4 = s[q].K >> 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + j * NTH;
      const int r = e / k4, c = (e - r * k4) * 4;
      v[q][j] = f4{0.f, 0.f, 0.f, 0.f};
      if (e < TM * k4 && r < rows) {
        v[q][j] = ld4(x + (m0 + r) * s[q].ld0 + c);
        if (p != nullptr) v[q][j] += ld4(p + (m0 + r) * s[q].ld1 + c);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kMaxLoadRun; ++q) {
    if (!on[q]) continue;
    float *d = buf + s[q].dst * TM * LDW + s[q].a;
    const int k4 = s[q].K >> 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + j * NTH;
      const int r = e / k4, c = (e - r * k4) * 4;
      if (e < TM * k4) *reinterpret_cast<f4 *>(d + r * LDW + c) = v[q][j];
    }
  }
}

// K_LOAD_PARTS: dst[r][c] = sum_{s < a} p0[(s * b + m) * 128 + c] + p1[c]   (split-K partial sums, b = total rows)
__device__ __forceinline__ void step_load_parts(const Step &s, float *buf, long long m0, int rows, int tid, int rr) {
  const float *part = (const float *)s.p0 + rr * s.roff, *bias = (const float *)s.p1;
  float *d = buf + s.dst * TM * LDW;
  const int r = tid >> 5, c = (tid & 31) * 4;
  f4 v = bias != nullptr ? ld4(bias + c) : f4{0.f, 0.f, 0.f, 0.f};
  if (r < rows) {
    const float *p = part + (size_t)(m0 + r) * 128 + c;
    const size_t st = (size_t)s.b * 128;
    for (int s0 = 0; s0 < s.a; s0 += 8) {                  // 8 slices in flight
      f4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = s0 + j < s.a ? ld4(p + (size_t)(s0 + j) * st) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) v += t[j];
    }
  } else {
    v = f4{0.f, 0.f, 0.f, 0.f};
  }
  *reinterpret_cast<f4 *>(d + r * LDW + c) = v;
}

// K_ATTN: self attention among the Q tokens of the sample for this workgroup's 16 queries (reference
// decoder_utils.py:743-746 / :824-826 and :91-95): wave = head (8 heads x 16 dims), packed projection p0 = [q | k | v]
// rows of ld0 floats [q | k] + the transposed values p3 = V^T (B, 128, Qp = b).  Optional visibility (image RoI block: the attention runs among the queries of ONE view): key k
// is visible to query q when bit view[q] of member[k] is set, or when view[q] < 0.  f = scale * log2(e).
// Key tiles go in chunks of 8: all loads of a chunk are issued first; the 8 score tiles are independent products, ONE
// maximum exchange per chunk (the first version did the online soft-max tile by tile: a dependent chain of MFMA -> 2
// cross-lane exchanges -> exp -> MFMA per tile), then the probabilities and O^T += V^T P^T.  Both products use the
// FLOAT32 MFMA (16x16x4: the head dim 16 = four of them per tile): the three-pass fp16 form measured 32 000 cycles per
// step here - the split of K, V and P costs 5 VALU operations per element, a wave64 VALU operation takes 4 cycles and
// two waves share a SIMD, so the step was bound by the conversions, not by the matrix cores.
struct HL4 {
  h4 hi, lo;
};
__device__ __forceinline__ HL4 split4(f4 a) {
  HL4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    _Float16 h, l;
    split1(a[j], h, l);
    r.hi[j] = h;
    r.lo[j] = l;
  }
  return r;
}
__device__ __forceinline__ void mfma3s(const HL4 &a, const HL4 &b, f4 &ch, f4 &cl) {
  cl = __builtin_amdgcn_mfma_f32_16x16x16f16(a.lo, b.hi, cl, 0, 0, 0);
  cl = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.lo, cl, 0, 0, 0);
  ch = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.hi, ch, 0, 0, 0);
}
__device__ __forceinline__ void step_attn(const Step &s, float *buf, int b, int q0, int Q, int lane, int wave) {
  const float *base = (const float *)s.p0 + (size_t)b * Q * s.ld0;          // rows [q | k] of ld0 floats
  const unsigned char *member = (const unsigned char *)s.p1;
  const signed char *view = (const signed char *)s.p2;
  const int ld = (int)s.ld0, h = wave, i = lane & 15, g = lane >> 4, Qp = s.b;
  const float *vt = (const float *)s.p3 + ((size_t)b * 128 + h * 16 + i) * Qp;   // V^T[dim i of head h][key]
  const int qc = min(q0 + i, Q - 1);
  const f4 qv = ld4(base + (size_t)qc * ld + h * 16 + 4 * g);                 // B[k = dim 4g + j][col = query i]
  const int vq = member != nullptr ? (int)view[(size_t)b * Q + qc] : -1;
  const float sl2 = s.f;
  float m = -INFINITY, l = 0.f;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const int ntile = (Q + 15) >> 4;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  const bool mem4 = member != nullptr && (Q & 3) == 0;     // membership bytes of 4 keys as one aligned dword
  for (int t0 = 0; t0 < ntile; t0 += 8) {
    // tile t: K row of key 16t + i (dims 4g..4g+3), V^T[dim i][keys 16t + 4g .. + 3] (the transposed copy the producer
    // stored: one 16-B load where the row-major layout needed four 4-B loads), membership bytes of keys 16t + 4g + r;
    // keys past the end are masked below
    f4 kk[8], vv[8];
    unsigned mem[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = min(t0 + j, ntile - 1), k4 = 16 * t + 4 * g;
      kk[j] = ld4(base + (size_t)min(16 * t + i, Q - 1) * ld + 128 + h * 16 + 4 * g);
      vv[j] = ld4(vt + k4);
      mem[j] = 0xFFFFFFFFu;
      if (mem4) {
        mem[j] = k4 < Q ? *reinterpret_cast<const unsigned *>(member + (size_t)b * Q + k4) : 0u;
      } else if (member != nullptr) {
        unsigned mm = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) mm |= (unsigned)member[(size_t)b * Q + min(k4 + r, Q - 1)] << (8 * r);
        mem[j] = mm;
      }
    }
    f4 sc[8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f4 c = mfma4(kk[j], qv, zero);                   // S^T[key 4g + r][query i]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * (t0 + j) + 4 * g + r;
        const bool vis = key < Q && (vq < 0 || ((mem[j] >> (8 * r + vq)) & 1u));
        const float x = vis ? c[r] * sl2 : -INFINITY;
        sc[j][r] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float ms = mn == -INFINITY ? 0.f : mn;             // nothing visible so far: keep everything at zero
    const float a = __builtin_amdgcn_exp2f(m - ms);
    l *= a;
    acc *= a;
    m = mn;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f4 p;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = __builtin_amdgcn_exp2f(sc[j][r] - ms);
        l += p[r];
      }
      acc = mfma4(vv[j], p, acc);                            // O^T[dim 4g + r'][query i]: A = V^T[dim i][key 4g + r]
    }
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  *reinterpret_cast<f4 *>(buf + s.dst * TM * LDW + i * LDW + h * 16 + 4 * g) = acc * inv;
}

// K_COMBINE: merge of the cross attention's partial softmax states (csrc/cross_attn.hip mha_decode_x_kernel: one state
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

// K_LINEAR: dst[r][n] = act(sum_k src[r][k] * W[n][k] + bias[n]),  n < N (multiple of 16), k < K (multiple of 128);
// W = hi + lo / 2048 packed in fragment order (p0, see linear_issue).  Work items = (16-channel tile of this wave, 128-wide K
// chunk), tile-major; the weight fragments of FOUR items (8 x 16 B per lane each) are issued before the first MFMA of
// the group.  (Holding the next step's first group in registers across the barrier made the compiler spill 529
// VGPRs; warming L2 with one 4-byte LDS-DMA request per weight line in a prologue cost as much texture-pipe time as
// it saved.)
struct WFrag {
  h8 w[8];              // k-step kk: w[2kk] = hi, w[2kk + 1] = lo
  f4 bias;
};
// The weight arrives PACKED in fragment order (ops.pack_linear): block (tile t, chunk c) = 8 pieces of 1 KiB,
// piece 2kk + h = [lane 16g + i][8 halfs] = W_h[16t + i][128c + 32kk + 8g .. + 7] - one load instruction of a wave reads
// one contiguous KiB (row-major weights made every instruction touch 16 half cache lines: 25 GB/s per workgroup).
__device__ __forceinline__ void linear_issue(const Step &s, int it0, WFrag (&A)[4], int lane, int wave, int rr) {
  const __half *wp = (const __half *)s.p0;
  const float *bias = (const float *)s.p1;
  const int g = lane >> 4, nch = s.K >> 7, nchw = s.nch > 0 ? s.nch : nch;
  const int ntile = s.N >> 4;
  const int mine = wave < ntile ? (ntile - wave + NW - 1) / NW : 0;
  const int total = mine * nch;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int it = it0 + j;
    if (it < total) {                                        // wave-uniform (every load instruction costs the CU's one
      const int tt = it / nch, c = it - tt * nch, t = wave + NW * tt + rr * s.rt;       // texture pipe 16 cycles)
      const __half *blk = wp + ((size_t)(t * nchw + c + rr * s.rc) * 8) * 512 + lane * 8;
#pragma unroll
      for (int q = 0; q < 8; ++q) A[j].w[q] = ld_h8(blk + q * 512);
      A[j].bias = bias != nullptr ? ld4(bias + 16 * t + 4 * g) : f4{0.f, 0.f, 0.f, 0.f};
    }
  }
}
// The B operands: the 16 token rows of `src`, split into hi / lo ONCE per step by the whole workgroup into an LDS image
// (rows of KIMG halfs: hi then lo), not by every wave for itself (8 x the conversions: the step was VALU bound).
constexpr int KIMG = 512 + 8;                           // halfs per image row and half (padding: 16-B reads of 16 rows spread over the banks)
__device__ __forceinline__ void linear_image(const Step &s, const float *buf, __half *img, int tid) {
  const float *x = buf + s.src * TM * LDW;
  const int k8 = s.K >> 3;
  for (int e = tid; e < TM * k8; e += NTH) {
    const int r = e / k8, c = (e - r * k8) * 8;
    const HL v = split8(ld4(x + r * LDW + c), ld4(x + r * LDW + c + 4));
    *reinterpret_cast<h8 *>(img + r * KIMG + c) = v.hi;
    *reinterpret_cast<h8 *>(img + TM * KIMG + r * KIMG + c) = v.lo;
  }
}
__device__ __forceinline__ void step_linear(const Step &s, float *buf, const __half *img, int lane, int wave, int rr) {
  float *d = buf + s.dst * TM * LDW;
  const int i = lane & 15, g = lane >> 4, nch = s.K >> 7;
  const int ntile = s.N >> 4;
  const int mine = wave < ntile ? (ntile - wave + NW - 1) / NW : 0;
  const int total = mine * nch;
  if (total == 0) return;
  const __half *xr = img + i * KIMG + 8 * g;
  f4 ah = {0.f, 0.f, 0.f, 0.f}, al = {0.f, 0.f, 0.f, 0.f};
  WFrag A[4];
  for (int it0 = 0; it0 < total; it0 += 4) {
    linear_issue(s, it0, A, lane, wave, rr);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int it = it0 + j;
      if (it >= total) break;                               // wave-uniform
      const int tt = it / nch, c = it - tt * nch;
      if (c == 0) {
        ah = f4{0.f, 0.f, 0.f, 0.f};
        al = f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        HL b;
        b.hi = *reinterpret_cast<const h8 *>(xr + 128 * c + 32 * kk);
        b.lo = *reinterpret_cast<const h8 *>(xr + TM * KIMG + 128 * c + 32 * kk);
        mfma3(A[j].w[2 * kk], A[j].w[2 * kk + 1], b, ah, al);
      }
      if (c == nch - 1) {
        const int n = 16 * (wave + NW * tt) + 4 * g;
        f4 v = A[j].bias;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += fmaf(al[r], kLoInv, ah[r]);
        if (s.a == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (s.a == 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
        }
        *reinterpret_cast<f4 *>(d + i * LDW + n) = v;       // Y^T[channel 4g + r][token i]
      }
    }
  }
}

// K_ROWOP on 128-wide rows: v = src (+ buf[aux] if aux >= 0); v = LayerNorm(v) * p0 + p1 if p0; relu if b & 1;
// rows with p2[m] == 0 zeroed.  16 lanes per row, 8 channels per lane.
__device__ __forceinline__ void step_rowop(const Step &s, float *buf, long long m0, int rows, int tid) {
  if (tid >= 256) return;
  const int r = tid >> 4, c0 = (tid & 15) * 8;
  float w[8], bb[8];
  if (s.p0 != nullptr) {                                    // issued first: their round trip overlaps the row reads
    unpack8(ld8((const float *)s.p0 + c0), w);
    unpack8(ld8((const float *)s.p1 + c0), bb);
  }
  const bool kept = s.p2 == nullptr || r >= rows || ((const unsigned char *)s.p2)[m0 + r] != 0;
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
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (v[j] - mean) * inv * w[j] + bb[j];
  }
  if (s.b & 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
  }
  if (!kept) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  }
  st8(buf + s.dst * TM * LDW + r * LDW + c0, pack8f(v, 0.f));
}

// K_STORE: (p0 + role * roff)[m * ld0 + c] = src[r][a + c],  c < N.  With b == 1 the store is TRANSPOSED:
// p0[(sample * N + c) * ld0 + q] = src[q - q0][a + c] - the layout the self attention reads its values in (ld0 = Qp).
__device__ __forceinline__ void step_store(const Step &s, const float *buf, long long m0, int rows, int tid, int rr, int b,
                                           int q0) {
  float *y = (float *)s.p0 + rr * s.roff;
  const float *x = buf + s.src * TM * LDW + s.a;
  if (s.b == 1) {
    for (int e = tid; e < s.N * 4; e += NTH) {
      const int c = e >> 2, r4 = (e & 3) * 4;
      f4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = x[(r4 + r) * LDW + c];
      *reinterpret_cast<f4 *>(y + ((size_t)b * s.N + c) * s.ld0 + q0 + r4) = v;
    }
    return;
  }
  if (s.b == 2) {                                         // split store: p0 fp16 rows [hi N | lo N] (the B operand of wide_kernel)
    __half *yh = (__half *)s.p0;
    const int n8 = s.N >> 3;
    for (int e = tid; e < rows * n8; e += NTH) {
      const int r = e / n8, c = (e - r * n8) * 8;
      const HL v = split8(ld4(x + r * LDW + c), ld4(x + r * LDW + c + 4));
      *reinterpret_cast<h8 *>(yh + (m0 + r) * s.ld0 + c) = v.hi;
      *reinterpret_cast<h8 *>(yh + (m0 + r) * s.ld0 + s.N + c) = v.lo;
    }
    return;
  }
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
__device__ __forceinline__ void step_heads(const Step &s, const Heads &ho, const float *buf, int b, int q0, int Q, int tid, int rr) {
  // s.a == 1: this workgroup evaluates head `rr` only, its 64 hidden channels at columns 0..63 of src
  const float *hid = buf + s.src * TM * LDW;
  int total = 0, first_row = 0;
#pragma unroll
  for (int h = 0; h < DI_TOK_MAX_HEADS; ++h) {
    const int n = h < ho.nheads ? ho.cls[h] : 0;
    if (s.a == 1) {
      if (h < rr) first_row += n;
      if (h == rr) total = n;
    } else {
      total += n;
    }
  }
  for (int e = tid; e < TM * total; e += NTH) {
    const int r = e / total, o = first_row + e - r * total;
    const int q = q0 + r;
    if (q >= Q) continue;
    int h = 0, row0 = 0;
    while (h + 1 < ho.nheads && o >= row0 + ho.cls[h]) row0 += ho.cls[h++];
    const int cidx = o - row0, ncls = ho.cls[h];
    const float *wr = ho.w2 + (size_t)o * 64;
    const float *hr = hid + r * LDW + (s.a == 1 ? 0 : h * 64);
    const size_t bq = (size_t)b * Q + q;
    f4 wv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) wv[j] = ld4(wr + 4 * j);
    float a = ho.b2[o];
    const float qp = h == ho.center_head ? ho.qpos[bq * 2 + cidx] : 0.f;
    const bool merge = ho.keep != nullptr && !ho.keep[bq];
    const float fv = merge ? ho.first[h][((size_t)b * ncls + cidx) * Q + q] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const f4 x = ld4(hr + 4 * j);
      a = fmaf(wv[j][0], x[0], a);
      a = fmaf(wv[j][1], x[1], a);
      a = fmaf(wv[j][2], x[2], a);
      a = fmaf(wv[j][3], x[3], a);
    }
    // DeepInteraction++ (deepinteractionplusplus_decoder.py:291-294): the NEXT stage's look-forward centre is this
    // stage's raw offset + the previous stage's centre, whatever the on-the-image merge does to the stored value
    if (h == ho.center_head && ho.pos2_out != nullptr) ho.pos2_out[bq * 2 + cidx] = a + ho.qpos2[bq * 2 + cidx];
    a += qp;
    if (merge) a = fv;
    ho.out[h][((size_t)b * ncls + cidx) * ho.ldo + ho.col0 + q] = a;
    if (h == ho.center_head && ho.pos_out != nullptr) ho.pos_out[bq * 2 + cidx] = a;
  }
}

__global__ __launch_bounds__(NTH) void program_kernel(Program prog, Heads heads, int Q, int touch, unsigned long long *stamps) {
  extern __shared__ __align__(16) float buf[];          // NBUF x TM x LDW floats, then the hi / lo image (2 x TM x KIMG halfs)
  __shared__ Program lp;
  __shared__ Heads lh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (round 6, measured and not kept: role r on XCD r % 8, so that a role's 13 token groups share one L2 for its <= 128 KB of
  //  weights - HBM traffic per launch 6.62 -> 6.56 MB, time unchanged: the weights are not what this kernel fetches from HBM)
  const int b = blockIdx.y, q0 = blockIdx.x * TM;
  const int rows = min(TM, Q - q0);
  const long long m0 = (long long)b * Q + q0;
  const bool stamp = stamps != nullptr && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
  if (stamp) stamps[0] = __builtin_readcyclecounter();
  {
    const unsigned *src = reinterpret_cast<const unsigned *>(&prog);
    unsigned *dst = reinterpret_cast<unsigned *>(&lp);
    for (int j = tid; j < (int)(sizeof(Program) / 4); j += NTH) dst[j] = src[j];
    const unsigned *hs = reinterpret_cast<const unsigned *>(&heads);
    unsigned *hd = reinterpret_cast<unsigned *>(&lh);
    if (tid < (int)(sizeof(Heads) / 4)) hd[tid] = hs[tid];
  }
  __syncthreads();
  const int nsteps = __builtin_amdgcn_readfirstlane(lp.n);
  const int role = blockIdx.z;
  __half *img = reinterpret_cast<__half *>(buf + NBUF * TM * LDW);
  (void)touch;
  for (int si = 0; si < nsteps; ++si) {
    const Step s = fetch_step(&lp.s[si]);
    const bool mine = role >= s.role_lo && role <= s.role_hi;
    const int rr = role - s.role_lo;
    if (s.kind == DI_TOK_LOAD) {                           // (a run of loads checks the roles step by step)
      int nrun = 1;
      while (nrun < kMaxLoadRun && si + nrun < nsteps && __builtin_amdgcn_readfirstlane(lp.s[si + nrun].kind) == DI_TOK_LOAD) ++nrun;
      steps_load(lp, si, nrun, buf, m0, rows, tid, role);
      si += nrun - 1;
    } else if (mine) {
      switch (s.kind) {
        case DI_TOK_LOAD_PARTS: step_load_parts(s, buf, m0, rows, tid, rr); break;
        case DI_TOK_ATTN: step_attn(s, buf, b, q0, Q, lane, wave); break;
        case DI_TOK_COMBINE: step_combine(s, buf, b, q0, Q, lane, wave); break;
        case DI_TOK_LINEAR:
          linear_image(s, buf, img, tid);
          __syncthreads();
          step_linear(s, buf, img, lane, wave, rr);
          break;
        case DI_TOK_ROWOP: step_rowop(s, buf, m0, rows, tid); break;
        case DI_TOK_STORE: step_store(s, buf, m0, rows, tid, rr, b, q0); break;
        case DI_TOK_HEADS: step_heads(s, lh, buf, b, q0, Q, tid, rr); break;
        default: break;
      }
    }
    __syncthreads();
    if (stamp) stamps[si + 1] = __builtin_readcyclecounter();      // profiling aid (di_token_program_timed)
  }
}

// ------------------------------------------------------------------------------------------------------------
// DynamicConv, three kernels (decoder_utils.py:608-624).  Everything here is bound by how fast a CU takes in matrix
// operands (~64 B/clk at best), so every operand is laid out for ONE contiguous KiB per wave-level load and no byte is
// read twice by a workgroup:
//
//   wide_kernel     params = Linear 128 -> 2*128*128 of the query feature, weight stationary: a wave owns 64 output
//                   values (4 fragments x hi/lo = 32 KiB of packed weight, read once) and walks the 13 row tiles.
//   dynconv_kernel  F2 = relu(LN2(relu(LN1(roi . p1)) . p2)), one workgroup per RoI: each WAVE owns 32 of the 128
//                   output channels for all 64 (49 valid) positions, so the RoI's 256 KiB of generated parameters are
//                   read exactly once; the LayerNorm statistics and F1 cross the waves through LDS.
//   splitk_kernel   partial sums of out_layer (6 272 -> 128) over 14 K slices.
//
// Layouts (host side: decoder_fused._dyn_layout):
//   params[m] (65 536 halfs) = fragments (product p, channel block nb, k-step kk, half h) of 512 halfs =
//       [lane 16g + i][8] :  p = 0: p1^T[d = 16nb + i][c = 32kk + 8g + j];  p = 1: p2^T[e = 16nb + i][d = 32kk + 8g + j]
//     (the rows of the generating Linear are permuted so that its 32 768 outputs come out in this order: value index
//      V = fragment * 512 + lane * 8 + j without the half bit);
//   F2 leaves as f2p[k-step ks = 4s + e/32][m][hi 32 | lo 32] (k = s*128 + e of the flattened RoI feature): the B
//     operand of the split-K kernel, 16 rows x 128 B per (row tile, k-step);
//   out_layer's weight in k-step order: [tile t][k-step ks][half h][lane][8] (ops.pack_ksteps).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wide_kernel(const __half *__restrict__ X /*(M,256) = [hi | lo]*/, int ldx,
                                                      const __half *__restrict__ wp, const float *__restrict__ bias,
                                                      __half *__restrict__ Y, int M) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  // a wave = 64 values x a quarter of the row tiles: 2048 waves = two per SIMD (the kernel is a chain of MFMA and
  // conversion latencies per row tile; one wave per SIMD left the CUs busy 3x longer than their MFMA + VALU work).
  // The token rows arrive already split (the producing program stores hi | lo): every wave splitting the same 200 rows
  // for itself cost more VALU time than the MFMAs.
  const int wv = blockIdx.x, quarter = wave;              // the workgroup's 64 values V0 .. V0 + 63; a wave = a quarter of the rows
  const int ntile = (M + 15) >> 4, per = (ntile + 3) >> 2, t_lo = quarter * per, t_hi = min(t_lo + per, ntile);
  const int V0 = wv * 64;
  // packed weight: tile T = 4 wv + nb, row i' <-> value V0 + 16 (i' >> 2) + 4 nb + (i' & 3): a lane ends with the 16
  // consecutive values V0 + 16 g + 4 nb + r
  h8 a[4][8];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const __half *blk = wp + (size_t)(4 * wv + nb) * 4096 + lane * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) a[nb][q] = ld_h8(blk + q * 512);
  }
  f4 bs[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) bs[nb] = bias ? ld4(bias + V0 + 16 * g + 4 * nb) : f4{0.f, 0.f, 0.f, 0.f};
  HL xb[4], xn[4];
  auto fetch = [&](int m0, HL (&x)[4]) {
    const __half *p = X + (size_t)min(m0 + i, M - 1) * ldx + 8 * g;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      x[kk].hi = ld_h8(p + 32 * kk);
      x[kk].lo = ld_h8(p + 128 + 32 * kk);
    }
  };
  if (t_lo >= t_hi) return;
  fetch(16 * t_lo, xb);
  const size_t ooff = ((size_t)(V0 >> 9) * 2) * 512 + (V0 & 511) + 16 * g;
  for (int m0 = 16 * t_lo; m0 < 16 * t_hi; m0 += 16) {
    if (m0 + 16 < 16 * t_hi) fetch(m0 + 16, xn);
    f4 hi[4], lo[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      hi[nb] = bs[nb];
      lo[nb] = f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) mfma3(a[nb][2 * kk], a[nb][2 * kk + 1], xb[kk], hi[nb], lo[nb]);
    if (m0 + i < M) {
      const HL o0 = split8(hi[0] + lo[0] * kLoInv, hi[1] + lo[1] * kLoInv);
      const HL o1 = split8(hi[2] + lo[2] * kLoInv, hi[3] + lo[3] * kLoInv);
      __half *y = Y + (size_t)(m0 + i) * 65536 + ooff;
      *reinterpret_cast<h8 *>(y) = o0.hi;
      *reinterpret_cast<h8 *>(y + 8) = o1.hi;
      *reinterpret_cast<h8 *>(y + 512) = o0.lo;
      *reinterpret_cast<h8 *>(y + 520) = o1.lo;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xb[kk] = xn[kk];
  }
}

constexpr int DC_LD = 136;      // halfs per LDS row and half of the (64 positions x 128 channels) operand image
__global__ __launch_bounds__(256) void dynconv_kernel(const __half *__restrict__ roi /*(R,49,256) = [hi | lo]*/,
                                                      const __half *__restrict__ params, const float *__restrict__ n1w,
                                                      const float *__restrict__ n1b, const float *__restrict__ n2w,
                                                      const float *__restrict__ n2b, __half *__restrict__ f2p, int M, float eps) {
  // the B operand of both products as an hi / lo image: the RoI feature (RoIAlign writes it split), then F1 (every wave
  // writes its 32 channels split).  The first version kept float32 rows here and every wave split all of them for
  // itself: 4x the conversions, the kernel was bound by them.
  __shared__ __align__(16) __half img[2 * 64 * DC_LD];
  __shared__ float red[4][64];                            // per (wave, position) partial sums
  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const __half *pq = params + (size_t)q * 65536 + lane * 8;
  // the wave's parameter fragments of product p: channel blocks nb = 2 wave + nbl, k-steps kk, halves h
  auto frags = [&](int prod, h8 (&a)[2][8]) {
#pragma unroll
    for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) a[nbl][qq] = ld_h8(pq + (size_t)(((prod * 8 + 2 * wave + nbl) * 4) * 2 + qq) * 512);
  };
  h8 a[2][8];
  frags(0, a);
  {                                                       // RoI feature -> LDS (rows >= 49 zero)
    const __half *rq = roi + (size_t)q * 49 * 256;
    const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = tid; e < 64 * 32; e += 256) {
      const int s = e >> 5, h = (e >> 4) & 1, c = (e & 15) * 8;
      *reinterpret_cast<h8 *>(&img[(h * 64 + s) * DC_LD + c]) = s < 49 ? ld_h8(rq + s * 256 + h * 128 + c) : zero;
    }
  }
  __syncthreads();
  f4 acc[2][4];                                           // [channel block][position group]: channel 32 wave + 16 nbl + 4g + r, position 16 pg + i
  auto product = [&]() {
    f4 lo[2][4];
#pragma unroll
    for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
      for (int pg = 0; pg < 4; ++pg) {
        acc[nbl][pg] = f4{0.f, 0.f, 0.f, 0.f};
        lo[nbl][pg] = f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) {
      const __half *xr = &img[(16 * pg + i) * DC_LD + 8 * g];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        HL b;
        b.hi = *reinterpret_cast<const h8 *>(xr + 32 * kk);
        b.lo = *reinterpret_cast<const h8 *>(xr + 64 * DC_LD + 32 * kk);
#pragma unroll
        for (int nbl = 0; nbl < 2; ++nbl) mfma3(a[nbl][2 * kk], a[nbl][2 * kk + 1], b, acc[nbl][pg], lo[nbl][pg]);
      }
    }
#pragma unroll
    for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
      for (int pg = 0; pg < 4; ++pg) acc[nbl][pg] += lo[nbl][pg] * kLoInv;
  };
  // LayerNorm over the 128 channels of every position + ReLU: this wave holds 32 channels of each position
  auto ln_relu = [&](const float *w, const float *b) {
    float mean[4], inv[4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int pg = 0; pg < 4; ++pg) {
        float sum = 0.f;
#pragma unroll
        for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = pass == 0 ? acc[nbl][pg][r] : acc[nbl][pg][r] - mean[pg];
            sum += pass == 0 ? d : d * d;
          }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (g == 0) red[wave][16 * pg + i] = sum;
      }
      __syncthreads();
#pragma unroll
      for (int pg = 0; pg < 4; ++pg) {
        const int pp = 16 * pg + i;
        const float t = (red[0][pp] + red[1][pp]) + (red[2][pp] + red[3][pp]);
        if (pass == 0) mean[pg] = t * (1.f / 128.f);
        else inv[pg] = rsqrtf(t * (1.f / 128.f) + eps);
      }
      __syncthreads();
    }
#pragma unroll
    for (int nbl = 0; nbl < 2; ++nbl) {
      const int ch = 32 * wave + 16 * nbl + 4 * g;
      const f4 wv = ld4(w + ch), bv = ld4(b + ch);
#pragma unroll
      for (int pg = 0; pg < 4; ++pg)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[nbl][pg][r] = fmaxf((acc[nbl][pg][r] - mean[pg]) * inv[pg] * wv[r] + bv[r], 0.f);
    }
  };
  product();                                              // F1^T
  frags(1, a);                                            // (in flight under the LayerNorm)
  ln_relu(n1w, n1b);                                      // its last barrier: every wave is done reading the RoI image
#pragma unroll
  for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) {
      const HL4 v = split4(acc[nbl][pg]);
      __half *o = &img[(16 * pg + i) * DC_LD + 32 * wave + 16 * nbl + 4 * g];
      *reinterpret_cast<h4 *>(o) = v.hi;
      *reinterpret_cast<h4 *>(o + 64 * DC_LD) = v.lo;
    }
  __syncthreads();
  product();                                              // F2^T
  ln_relu(n2w, n2b);
  // f2p[ks = 4 s + wave][q][h * 32 + 16 nbl + 4 g + r]
#pragma unroll
  for (int pg = 0; pg < 4; ++pg) {
    const int sp = 16 * pg + i;
    if (sp < 49) {
      __half *o = f2p + ((size_t)(4 * sp + wave) * M + q) * 64 + 4 * g;
#pragma unroll
      for (int nbl = 0; nbl < 2; ++nbl) {
        const HL4 v = split4(acc[nbl][pg]);
        *reinterpret_cast<h4 *>(o + 16 * nbl) = v.hi;
        *reinterpret_cast<h4 *>(o + 32 + 16 * nbl) = v.lo;
      }
    }
  }
}

// Split-K out_layer: grid (row blocks of 16, K slices); a wave owns 32 output columns (2 tiles); k-steps of 32.
__global__ __launch_bounds__(256) void splitk_kernel(const __half *__restrict__ f2p, const __half *__restrict__ wp,
                                                     float *__restrict__ part, int M, int nks, int ks_per_slice) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  // XCD-aware (round 6): consecutive block ids go round-robin over the 8 XCDs, each with its own L2.  With a (row block,
  // slice) grid the 13 row blocks of ONE slice landed on all 8 XCDs and every XCD fetched that slice's weights for itself:
  // 32 MB of HBM traffic per launch for 8.4 MB of operands (profiles/r06_pmc_forward.json).  Now slice s belongs to XCD s % 8:
  // its row blocks share one L2, the weights cross the fabric once.
  const int nrb = (M + 15) >> 4, nsl = (nks + ks_per_slice - 1) / ks_per_slice;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int slice = xcd + 8 * (j / nrb), rb = j % nrb;
  if (slice >= nsl) return;
  const int m0 = rb * 16;
  const int mr = min(m0 + i, M - 1);
  const int kb = slice * ks_per_slice, ke = min(kb + ks_per_slice, nks);
  f4 ah[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}, al[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
  const __half *w0 = wp + (size_t)(2 * wave) * nks * 1024 + lane * 8, *w1 = w0 + (size_t)nks * 1024;
  for (int k0 = kb; k0 < ke; k0 += 7) {                    // 7 k-steps in flight (42 x 16 B per lane)
    HL xb[7];
    h8 a0h[7], a0l[7], a1h[7], a1l[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      const int ks = min(k0 + c, ke - 1);
      const __half *xr = f2p + ((size_t)ks * M + mr) * 64 + 8 * g;
      xb[c].hi = ld_h8(xr);
      xb[c].lo = ld_h8(xr + 32);
      a0h[c] = ld_h8(w0 + (size_t)ks * 1024);
      a0l[c] = ld_h8(w0 + (size_t)ks * 1024 + 512);
      a1h[c] = ld_h8(w1 + (size_t)ks * 1024);
      a1l[c] = ld_h8(w1 + (size_t)ks * 1024 + 512);
    }
#pragma unroll
    for (int c = 0; c < 7; ++c)
      if (k0 + c < ke) {
        mfma3(a0h[c], a0l[c], xb[c], ah[0], al[0]);
        mfma3(a1h[c], a1l[c], xb[c], ah[1], al[1]);
      }
  }
  if (m0 + i < M) {
    float *dst = part + ((size_t)slice * M + m0 + i) * 128 + wave * 32 + 4 * g;
    *reinterpret_cast<f4 *>(dst) = ah[0] + al[0] * kLoInv;
    *reinterpret_cast<f4 *>(dst + 16) = ah[1] + al[1] * kLoInv;
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
    DI_REQUIRE(s.role_lo >= 0 && s.role_hi >= s.role_lo && s.role_hi < 16, "step %d: bad role range [%d, %d]", i, s.role_lo, s.role_hi);
    switch (s.kind) {
      case DI_TOK_LOAD:
        DI_REQUIRE(okbuf(s.dst) && s.p0 && s.K > 0 && s.K % 4 == 0 && s.a >= 0 && s.a % 4 == 0 && s.a + s.K <= 512,
                   "step %d (load): bad buffer / width", i);
        break;
      case DI_TOK_LOAD_PARTS:
        DI_REQUIRE(okbuf(s.dst) && s.p0 && s.a > 0 && s.b > 0, "step %d (load parts): bad arguments", i);
        break;
      case DI_TOK_ATTN:
        DI_REQUIRE(okbuf(s.dst) && s.p0 && s.p3 && s.ld0 >= 256 && s.ld0 % 4 == 0 && s.b >= 16 && s.b % 16 == 0 &&
                       (s.p1 == nullptr) == (s.p2 == nullptr),
                   "step %d (attention): needs rows [q|k] (8 heads x 16), V^T with a key stride multiple of 16, member + view together", i);
        break;
      case DI_TOK_COMBINE:
        DI_REQUIRE(okbuf(s.dst) && s.p0 && s.a > 0, "step %d (combine): bad arguments", i);
        break;
      case DI_TOK_LINEAR:
        DI_REQUIRE(okbuf(s.src) && okbuf(s.dst) && s.src != s.dst && s.p0 && s.K > 0 && s.K % 128 == 0 &&
                       s.K <= 512 && s.N > 0 && s.N % 16 == 0 && s.N <= 512 && s.a >= 0 && s.a <= 2,
                   "step %d (linear): K multiple of 128, N of 16, both <= 512, src != dst, packed weight in p0", i);
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

static int launch_program(const di_tok_step *steps, int nsteps, const di_tok_heads *heads, int B, int Q,
                          unsigned long long *stamps, void *stream) {
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
  static const int touch = getenv("DI_TOK_TOUCH") ? atoi(getenv("DI_TOK_TOUCH")) : 1;
  int roles = 1;
  for (int i = 0; i < nsteps; ++i) roles = steps[i].role_hi + 1 > roles ? steps[i].role_hi + 1 : roles;
  hipLaunchKernelGGL(program_kernel, dim3((Q + TM - 1) / TM, B, roles), dim3(NTH), LDS_BYTES, (hipStream_t)stream, prog, h, Q,
                     touch, stamps);
  return di::check_launch("token_program");
}

int di_token_program(const di_tok_step *steps, int nsteps, const di_tok_heads *heads, int B, int Q, void *stream) {
  return launch_program(steps, nsteps, heads, B, Q, nullptr, stream);
}

int di_token_program_timed(const di_tok_step *steps, int nsteps, const di_tok_heads *heads, int B, int Q,
                           unsigned long long *stamps, void *stream) {
  DI_REQUIRE(stamps != nullptr, "di_token_program_timed needs nsteps + 1 stamps");
  return launch_program(steps, nsteps, heads, B, Q, stamps, stream);
}

int di_token_wide(const void *x_hl, int ldx, const void *w_packed, const float *bias, void *params, int M, void *stream) {
  DI_REQUIRE(x_hl && w_packed && params && M > 0 && ldx >= 256, "bad parameter generator call M=%d ldx=%d", M, ldx);
  hipLaunchKernelGGL(di::t32::wide_kernel, dim3(32768 / 64), dim3(256), 0, (hipStream_t)stream, (const __half *)x_hl, ldx,
                     (const __half *)w_packed, bias, (__half *)params, M);
  return di::check_launch("token_wide");
}

long long di_token_splitk_workspace_bytes(int M, int K) { return (long long)((K / 32 + 13) / 14) * M * 128 * 4; }

int di_token_splitk(const void *f2p, const void *w_packed, float *workspace, int M, int K, int *nslices, void *stream) {
  DI_REQUIRE(f2p && w_packed && workspace && M > 0 && K >= 32 && K % 32 == 0,
             "bad split-K shape M=%d K=%d (N = 128, K multiple of 32)", M, K);
  const int nks = K / 32, per = 14, ns = (nks + per - 1) / per;
  hipLaunchKernelGGL(di::t32::splitk_kernel, dim3(8 * ((M + 15) / 16) * ((ns + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const __half *)f2p, (const __half *)w_packed, workspace, M, nks, per);
  if (nslices) *nslices = ns;
  return di::check_launch("token_splitk");
}

int di_dynconv_fwd(const void *roi_hl, const void *params, const float *n1w, const float *n1b, const float *n2w,
                   const float *n2b, void *f2p, int R, float eps, void *stream) {
  DI_REQUIRE(R > 0 && roi_hl && params && f2p && n1w && n1b && n2w && n2b, "bad DynamicConv call R=%d", R);
  hipLaunchKernelGGL(di::t32::dynconv_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const __half *)roi_hl,
                     (const __half *)params, n1w, n1b, n2w, n2b, (__half *)f2p, R, eps);
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
