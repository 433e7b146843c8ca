// Fused chains of 1x1 convolutions (+ folded BatchNorm, + ReLU, + channel concat) on channels-last fp16
// maps, on the gfx950 matrix cores.
//
// The MMRI encoder is full of them (reference encoder_utils.py:92-117 query/key/value projections of
// LocalContextAttentionBlock = 2 or 1 x {Conv1x1 + BN + ReLU}; deepinteraction_encoder.py:13-19,26-32
// the out_proj / integration pairs = Conv1x1(cat(a, b)) + BN twice).  As library GEMMs each link of a
// chain reads and writes a whole 34 MB map (and a separate ReLU pass re-reads it); here a chain is one
// pass over the pixels:
//
//     h = act1( W1 . [x1 ; x2] + b1 )          (x2 optional: the concat is never materialised)
//     y = act2( W2 . [h  ; x3] + b2 )          (second link optional; x3 optional)
//
// One wavefront owns 32 pixels at a time.  Everything is computed TRANSPOSED - H^T = W1 . X^T - so
// that (i) the B operand of the 16x16x32 MFMA is 8 consecutive channels of one pixel = one 16-B global
// load per lane straight from the channels-last map, (ii) the accumulator of link 1 (lane = pixel,
// registers = 4 consecutive output channels) IS, after ReLU and conversion to fp16, the B operand of
// link 2, with the MFMA k index mapped to hidden channel 32kk + 16t + 4g + r (k = 8g + 4t + r): the
// hidden activations never leave registers, and W2's columns are permuted accordingly when it is staged.
// Weights live in LDS (XOR-swizzled 16-B chunks: conflict-free ds_read_b128 fragments), staged once per
// persistent workgroup.  fp32 accumulation, bias add in fp32.
#include <hip/hip_ext.h>

#include "di_common.h"
#include "warp_common.h"
#include <type_traits>

namespace di {
namespace pw {

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int NW = 8, NT = NW * 64;   // wavefronts / threads per workgroup
constexpr int PG = 2;                 // pixel groups of 16 per wavefront pass

// LDS image of a (128 x K) fp16 weight matrix: row r at r*K*2 bytes, 16-B chunk c of the row stored at
// position (c & ~15) | ((c & 15) ^ (r & 15)).
template <int K>
__device__ __forceinline__ int w_off(int r, int c) {
  return r * K * 2 + (((c & ~15) | ((c & 15) ^ (r & 15))) << 4);
}

// stage W (128 x K, row-major fp16 in global) into LDS; PERM: the first 128 columns are re-ordered for
// the register-resident hidden operand: LDS column 32kk + 8g + 4t + r <- global column 32kk + 16t + 4g + r
// channel held by LDS row 16nb + 4g + r of the LAST link of a chain: 32(nb/2) + 8g + 4(nb%2) + r - a lane's fragment
// pair (2p, 2p+1) is then 8 consecutive output channels = one 16-B store (8-B stores are store-issue bound)
__device__ __forceinline__ int out_row(int rl) {
  const int nb = rl >> 4, gq = (rl >> 2) & 3, rq = rl & 3;
  return 32 * (nb >> 1) + 8 * gq + 4 * (nb & 1) + rq;
}

template <int K, bool PERM, bool ROWPERM = false>
__device__ __forceinline__ void stage_w(const __half *__restrict__ w, unsigned char *lds, int tid) {
  constexpr int CH = K / 8;            // 16-B chunks per row
  constexpr int N = 128 * CH / NT;     // chunks per thread
  static_assert(128 * CH % NT == 0, "whole rounds of chunks");
  // ALL loads of the thread first, then the LDS writes: one L2 round trip instead of N dependent ones
  uint2 lo[N], hi[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const int e = tid + j * NT;
    const int rl = e / CH, c = e - rl * CH;
    const int r = ROWPERM ? out_row(rl) : rl;                // source row
    // PERM: destination chunk c = 4kk + g (c < 16) holds columns 32kk + 8g + (4t + r'), i.e. two 8-B pieces of the
    // source (t = 0, 1); every other chunk is 16 contiguous bytes
    const bool perm = PERM && c < 16;
    const int kk = c >> 2, g = c & 3;
    const int c_lo = perm ? 32 * kk + 4 * g : c * 8, c_hi = perm ? 32 * kk + 16 + 4 * g : c * 8 + 4;
    lo[j] = *reinterpret_cast<const uint2 *>(w + (size_t)r * K + c_lo);
    hi[j] = *reinterpret_cast<const uint2 *>(w + (size_t)r * K + c_hi);
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const int e = tid + j * NT;
    const int rl = e / CH, c = e - rl * CH;
    *reinterpret_cast<uint4 *>(lds + w_off<K>(rl, c)) = make_uint4(lo[j].x, lo[j].y, hi[j].x, hi[j].y);
  }
}

template <int K1, int K2>
__global__ __launch_bounds__(NT, 2) void pointwise_chain_kernel(
    const __half *__restrict__ x1, const __half *__restrict__ x2, const __half *__restrict__ x3,
    const __half *__restrict__ w1, const float *__restrict__ b1, const __half *__restrict__ w2,
    const float *__restrict__ b2, __half *__restrict__ y, long long M, int relu1, int relu2,
    const __half *__restrict__ mask, const float *__restrict__ bm, int hm_S = 0) {
  // hm_S > 0 (single-link chains): the output is written HEAD-MAJOR - token t of map b, channels 16h .. 16h + 15 at
  // ((b * 8 + h) * hm_S + t) * 16 - the layout ms_deform_attn_hm_kernel (csrc/plusplus.hip) gathers from
  extern __shared__ __align__(16) unsigned char lds[];
  unsigned char *lw1 = lds;
  unsigned char *lw2 = lds + 128 * K1 * 2;
  float *lb = reinterpret_cast<float *>(lds + 128 * K1 * 2 + 128 * K2 * 2);   // b1[128], b2[128], bm[128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;

  constexpr int KK1 = K1 / 32;
  // weights > 80 KB: one workgroup per CU (2 waves per SIMD, 256 VGPRs each) - room to read the weight fragments of the
  // next row block under the MFMAs of the current one; the small variants run two workgroups per CU at <= 128 VGPRs
  constexpr bool PIPE = 128 * K1 * 2 + 128 * K2 * 2 + 1536 > 80 * 1024;
  const long long nchunk = (M + 16 * PG - 1) / (16 * PG);
  // wave-major numbering: the ragged last round of chunks lands on wave 0 of many workgroups, not on all waves of a few
  const long long ch0 = (long long)wave * gridDim.x + blockIdx.x, chstep = (long long)gridDim.x * NW;
  // B operands of link 1: pixel i of each group, channels 32kk + 8g .. +7 (x1 then x2).  The FIRST chunk's loads are
  // issued before the weights are staged (a launch gives a wave one or two chunks: its HBM round trip then overlaps
  // the 64 KB weight staging instead of following it).
  h8 xb[PG][KK1];
  _Float16 xm[PG];                                              // per-pixel mask of the masked link-1 bias (optional)
  auto load_x = [&](long long ch) {
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      const long long p = ch * (16 * PG) + pg * 16 + i;
      const long long pc = p < M ? p : M - 1;                   // ragged tail: clamped read, no store
      xm[pg] = mask != nullptr ? reinterpret_cast<const _Float16 *>(mask)[pc] : (_Float16)0;
#pragma unroll
      for (int kk = 0; kk < KK1; ++kk) {
        const __half *src = kk < 4 ? x1 : x2;
        xb[pg][kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(src + pc * 128 + (kk & 3) * 32 + g * 8));
      }
    }
  };
  if (ch0 < nchunk) load_x(ch0);

  // biases first (their loads then share the weights' L2 round trip instead of adding two of their own)
  float bv1 = 0.f, bv2 = 0.f, bvm = 0.f;
  if (tid < 128) {
    bv1 = b1[K2 == 0 ? out_row(tid) : tid];
    if (K2 > 0) bv2 = b2[out_row(tid)];
    if (bm != nullptr) bvm = bm[K2 == 0 ? out_row(tid) : tid];
  }
  stage_w<K1, false, K2 == 0>(w1, lw1, tid);                  // the rows of the LAST link are permuted (16-B stores)
  if (K2 > 0) stage_w<(K2 > 0 ? K2 : 128), true, true>(w2, lw2, tid);
  if (tid < 128) {
    lb[tid] = bv1;
    lb[128 + tid] = bv2;
    lb[256 + tid] = bvm;
  }
  __syncthreads();

  // ReLU as a floor (0 or -inf): no branch inside the MFMA loops
  const float fl1 = relu1 ? 0.f : -INFINITY, fl2 = relu2 ? 0.f : -INFINITY;
  // bias: the accumulators START from it (the C operand of a block's first MFMA: no add); ReLU: a packed fp16 max on the
  // rounded values (max(round(x), 0) = round(max(x, 0))): 8 VALU instructions per 8 outputs instead of 20 - next to the
  // 32 MFMAs of a block the float32 add / max / convert epilogue cost as many issue slots as the MFMAs themselves
  const _Float16 fh1 = (_Float16)fl1, fh2 = (_Float16)fl2;
  const h8 floor1 = {fh1, fh1, fh1, fh1, fh1, fh1, fh1, fh1}, floor2 = {fh2, fh2, fh2, fh2, fh2, fh2, fh2, fh2};
  for (long long ch = ch0; ch < nchunk; ch += chstep) {
    const long long p0 = ch * (16 * PG);
    long long pix[PG];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) pix[pg] = p0 + pg * 16 + i;
    // ---- link 1: H^T[oc][px] = W1 . X^T
    // the weight fragments of row block nb + 1 are read while the MFMAs of row block nb run (two blocks live at a time)
    f4 acc[PG][8];
    {
      h8 a[KK1], an[KK1];
      if constexpr (PIPE) {
#pragma unroll
        for (int kk = 0; kk < KK1; ++kk) a[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw1 + w_off<K1>(i, 4 * kk + g)));
      }
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        if constexpr (PIPE) {
          if (nb < 7) {
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk)
              an[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw1 + w_off<K1>(16 * (nb + 1) + i, 4 * kk + g)));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        {
          const f4 bias = *reinterpret_cast<const f4 *>(lb + 16 * nb + 4 * g);
#pragma unroll
          for (int pg = 0; pg < PG; ++pg) acc[pg][nb] = bias;
        }
#pragma unroll
        for (int kk = 0; kk < KK1; ++kk) {
          if constexpr (!PIPE) a[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw1 + w_off<K1>(16 * nb + i, 4 * kk + g)));
#pragma unroll
          for (int pg = 0; pg < PG; ++pg) acc[pg][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk], xb[pg][kk], acc[pg][nb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PIPE) {
#pragma unroll
          for (int kk = 0; kk < KK1; ++kk) a[kk] = an[kk];
        }
      }
    }
    // lane holds output channels 16nb + 4g + r of pixel i (bias already inside)
    if (mask != nullptr) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        const f4 mb = *reinterpret_cast<const f4 *>(lb + 256 + 16 * nb + 4 * g);
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) acc[pg][nb] += mb * (float)xm[pg];   // bias that applies to the masked pixels only
      }
    }
    if (K2 == 0) {
#pragma unroll
      for (int pg = 0; pg < PG; ++pg)
        if (pix[pg] < M) {
          // channels 32 p2 + 8 g .. + 7 = half (g & 1) of head 2 p2 + (g >> 1)
          long long base = pix[pg] * 128 + 8 * g, step = 32;
          if (hm_S > 0) {
            const int bmap = (int)pix[pg] / hm_S, t = (int)pix[pg] - bmap * hm_S;
            base = (((long long)bmap * 8 + (g >> 1)) * hm_S + t) * 16 + (g & 1) * 8;
            step = 2ll * hm_S * 16;
          }
#pragma unroll
          for (int p2 = 0; p2 < 4; ++p2) {
            h8 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              o[r] = (_Float16)acc[pg][2 * p2][r];
              o[4 + r] = (_Float16)acc[pg][2 * p2 + 1][r];
            }
            *reinterpret_cast<h8 *>(y + base + p2 * step) = __builtin_elementwise_max(o, floor1);
          }
        }
      if (ch + chstep < nchunk) load_x(ch + chstep);
      continue;
    }
    // ---- link 2: B operand = the hidden activations in registers (k = 8g + 4t + r <-> channel 32kk + 16t + 4g + r),
    // then (K2 = 256) the channels of x3
    constexpr int KK2 = (K2 > 0 ? K2 : 128) / 32;
    h8 hb[PG][4];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        h8 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t[r] = (_Float16)acc[pg][2 * kk][r];
          t[4 + r] = (_Float16)acc[pg][2 * kk + 1][r];
        }
        hb[pg][kk] = __builtin_elementwise_max(t, floor1);
      }
    if (K2 == 256) {
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) {
        const long long pc = pix[pg] < M ? pix[pg] : M - 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)   // xb is dead: reuse its registers for x3
          xb[pg][kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(x3 + pc * 128 + kk * 32 + g * 8));
      }
    }
    constexpr int KW2 = K2 > 0 ? K2 : 128;
    h8 a2[KK2], an2[KK2];
    if constexpr (PIPE) {
#pragma unroll
      for (int kk = 0; kk < KK2; ++kk) a2[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw2 + w_off<KW2>(i, 4 * kk + g)));
    }
#pragma unroll
    for (int p2 = 0; p2 < 4; ++p2) {
      f4 o2[2][PG];                                            // [fragment of the pair][pixel group]
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int nb = 2 * p2 + e;
        if constexpr (PIPE) {
          if (nb < 7) {
#pragma unroll
            for (int kk = 0; kk < KK2; ++kk)
              an2[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw2 + w_off<KW2>(16 * (nb + 1) + i, 4 * kk + g)));
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < KK2; ++kk) a2[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw2 + w_off<KW2>(16 * nb + i, 4 * kk + g)));
        }
        {
          const f4 bias = *reinterpret_cast<const f4 *>(lb + 128 + 16 * nb + 4 * g);
#pragma unroll
          for (int pg = 0; pg < PG; ++pg) o2[e][pg] = bias;
        }
#pragma unroll
        for (int kk = 0; kk < KK2; ++kk)
#pragma unroll
          for (int pg = 0; pg < PG; ++pg)
            o2[e][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[kk], kk < 4 ? hb[pg][kk] : xb[pg][kk & 3], o2[e][pg], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PIPE) {
#pragma unroll
          for (int kk = 0; kk < KK2; ++kk) a2[kk] = an2[kk];
        }
      }
#pragma unroll
      for (int pg = 0; pg < PG; ++pg)
        if (pix[pg] < M) {
          h8 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = (_Float16)o2[0][pg][r];
            o[4 + r] = (_Float16)o2[1][pg][r];
          }
          *reinterpret_cast<h8 *>(y + pix[pg] * 128 + 32 * p2 + 8 * g) = __builtin_elementwise_max(o, floor2);   // channels 32p + 8g + 0..7
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch + chstep < nchunk) load_x(ch + chstep);
  }
}

template <int K1, int K2>
static int launch(const void *x1, const void *x2, const void *x3, const void *w1, const float *b1,
                  const void *w2, const float *b2, void *y, long long M, int relu1, int relu2,
                  const void *mask, const float *bm, hipStream_t stream, int hm_S = 0) {
  constexpr int LDS = 128 * K1 * 2 + 128 * K2 * 2 + 1536;
  static LdsRaised lds_raised;
  if (int rc = ensure_lds(lds_raised, (const void *)pointwise_chain_kernel<K1, K2>, LDS)) return rc;
  const int n_cu = device_cus();
  if (n_cu <= 0) return DI_ERR_LAUNCH;
  const long long nchunk = (M + 16 * PG - 1) / (16 * PG);
  const int per_cu = LDS <= 80 * 1024 ? 2 : 1;
  long long grid = (long long)n_cu * per_cu;
  if (grid * NW > nchunk) grid = (nchunk + NW - 1) / NW;
  hipEvent_t ev0, ev1;
  if (take_launch_events(ev0, ev1))                          // measurement: the dispatch's own begin / end time stamps
    hipExtLaunchKernelGGL((pointwise_chain_kernel<K1, K2>), dim3((unsigned)grid), dim3(NT), LDS, stream, ev0, ev1, 0,
                          (const __half *)x1, (const __half *)x2, (const __half *)x3, (const __half *)w1, b1,
                          (const __half *)w2, b2, (__half *)y, M, relu1, relu2, (const __half *)mask, bm, hm_S);
  else
    hipLaunchKernelGGL((pointwise_chain_kernel<K1, K2>), dim3((unsigned)grid), dim3(NT), LDS, stream,
                       (const __half *)x1, (const __half *)x2, (const __half *)x3, (const __half *)w1, b1,
                       (const __half *)w2, b2, (__half *)y, M, relu1, relu2, (const __half *)mask, bm, hm_S);
  return check_launch("pointwise_chain");
}

// ------------------------------------------------------------------------------------------------------------
// SEVERAL chains over ONE input map in one launch: the query / key / value projections of a
// LocalContextAttentionBlock (and the query projection of the P2I block, which reads the same image map) - reference
// encoder_utils.py:92-117, 127-131 - read their input once.  The input pixels of a wave (up to NG groups of 16) stay in
// registers as MFMA B operands while the workgroup walks the chains: stage the chain's weights in LDS, multiply,
// store that chain's output map.  Per chain the arithmetic is exactly pointwise_chain_kernel<128, 0 | 128>'s.
// ------------------------------------------------------------------------------------------------------------
constexpr int kMaxChains = 4;
constexpr int kChainImage = 2 * 128 * 128 * 2 + 1024;   // LDS image of a chain: W1 | W2 (swizzled, W2 k-permuted) | b1 | b2
struct Chain {
  const unsigned char *img;  // kChainImage bytes, prepared once on the host (ops.chain_image)
  __half *y;
  int relu1, relu2, two;
  int hm;                    // > 0 (single-link chains): head-major output, `hm` tokens per map (see pointwise_chain_kernel)
};
struct MultiArgs {
  Chain c[kMaxChains];
  int n;
};
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// The chain images are moved by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land as one contiguous KiB, no
// VGPRs, asynchronous) into one of TWO LDS buffers: the next chain's weights arrive while the current chain is
// multiplied.
// WARP: the input map is not read but GATHERED - pixel p of the (V, Hi, Wi) image maps takes the BEV map's bilinear sample at
// the position its completed depth un-projects to (BEVWarp, reference encoder_utils.py:185-196; the arithmetic of
// bevwarp_gather_kernel, rounded to fp16 as that kernel stores it): the key / value projections of the P2I block read the
// warped map straight from the BEV map and the 34 MB intermediate is never written.
struct WarpArgs {
  const __half *bev;
  const float *depth, *img2lidar, *aug, *xs, *ys, *pc_range;
  int Hi, Wi, Hb, Wb;
};

template <int NG, bool WARP = false>
__global__ __launch_bounds__(NT, 2) void pointwise_multi_kernel(const __half *__restrict__ x, MultiArgs A, long long M,
                                                                WarpArgs Wp = WarpArgs{}) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int Mi = (int)M;                                       // < 2^24 pixels (checked by the host): 32-bit offsets
  const int ngroups = (Mi + 15) / 16;
  // wave-major numbering: the ragged last round of groups (wg < ngroups mod stride) lands on wave 0 of MANY workgroups
  // instead of on all eight waves of a few
  const int wg = wave * gridDim.x + blockIdx.x, stride = gridDim.x * NW;

  auto dma_chain = [&](const Chain &ch, unsigned char *buf) {
    const unsigned char *src = ch.img + lane * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int chunk = wave * 4 + j;                         // W1: KiB chunks 0..31
      __builtin_amdgcn_global_load_lds((gptr_t)(src + chunk * 1024), (lptr_t)(buf + chunk * 1024), 16, 0, 0);
    }
    if (ch.two) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int chunk = 32 + wave * 4 + j;                  // W2: chunks 32..63
        __builtin_amdgcn_global_load_lds((gptr_t)(src + chunk * 1024), (lptr_t)(buf + chunk * 1024), 16, 0, 0);
      }
    }
    if (wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)(src + 64 * 1024), (lptr_t)(buf + 64 * 1024), 16, 0, 0);   // biases
  };
  dma_chain(A.c[0], lds);

  // this wave's pixels: group j = wg + j * stride; B operands (pixel i, channels 32kk + 8g .. +7) loaded ONCE
  h8 xb[NG][4];
  int pix[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int grp = wg + j * stride;
    pix[j] = grp < ngroups ? grp * 16 + i : Mi;                // M: "no pixel" (never stored)
    if constexpr (!WARP) {
      const int pc = pix[j] < Mi ? pix[j] : Mi - 1;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        xb[j][kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(x + (size_t)pc * 128 + kk * 32 + g * 8));
    }
  }
  // WARP (at most two chains: the host checks): BOTH weight images are resident in the two LDS buffers, so the groups go
  // through the chains one PAIR at a time and a pair is gathered right before its products - only 2 x 16 operand registers
  // are live beside the 64 of a group's 16 corner rows, which therefore leave together, without a branch between them
  // (bilinear8_nb).  Gathered up front for all five groups (80 operand registers) the rows were fetched one load -> blend
  // round trip at a time: 17 us of the launch with the matrix pipe idle.
  auto gather = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int pc = pix[j] < Mi ? pix[j] : Mi - 1;
    const WarpGeom G = load_warp_geom(Wp.depth, Wp.img2lidar, Wp.aug, Wp.xs, Wp.ys, Wp.pc_range, Wp.Hi, Wp.Wi, Wp.Hb, Wp.Wb);
    float ix, iy;
    const bool lift = warp_position(G, pc, ix, iy);
    const Bilin4 bl = bilinear_setup(Wp.Hb, Wp.Wb, ix, iy, lift);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float o[8];
      bilinear8_nb(Wp.bev, 128, bl, kk * 32 + g * 8, o);
      xb[j][kk] = __builtin_bit_cast(h8, pack8f(o, __half()));
    }
  };
  if constexpr (WARP) {
    if (A.n > 1) dma_chain(A.c[1], lds + kChainImage);
  }

  // chain c on the pair(s) of groups `only` (-1: all of them)
  auto run_chain = [&](int c, auto onlyc) {
    constexpr int only = decltype(onlyc)::value;
    const Chain ch = A.c[c];
    const bool two = ch.two != 0;
    unsigned char *buf = lds + (c & 1) * kChainImage;
    const unsigned char *lw1 = buf, *lw2 = buf + 128 * 128 * 2;
    const float *lb = reinterpret_cast<const float *>(buf + 2 * 128 * 128 * 2);
    if constexpr (!WARP) {
      // chain c's image has landed (issued a whole chain ago), and everybody is done with the other buffer
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (c + 1 < A.n) dma_chain(A.c[c + 1], lds + ((c + 1) & 1) * kChainImage);
    }

    // one step = the chain on NP (2, or 1 for the odd last one) pixel groups of this wave, sharing every weight fragment
    // ReLU as a floor (0 or -inf): no branch inside the MFMA loops
    const float fl1 = ch.relu1 ? 0.f : -INFINITY, fl2 = ch.relu2 ? 0.f : -INFINITY;
    const _Float16 fh1 = (_Float16)fl1, fh2 = (_Float16)fl2;   // (bias as the accumulators' start, ReLU as a packed fp16 max: see the chain kernel)
    const h8 floor1 = {fh1, fh1, fh1, fh1, fh1, fh1, fh1, fh1}, floor2 = {fh2, fh2, fh2, fh2, fh2, fh2, fh2, fh2};
    auto step = [&](auto J0, auto NPc) {
      constexpr int j0 = decltype(J0)::value, NP = decltype(NPc)::value;
      // ---- link 1, one PAIR of row blocks at a time: the pair (2p, 2p + 1) is stored (one-link chains) or becomes k-step p
      // of link 2's B operand at once, so only 2 x NP accumulators are live and NP can be 4 - a weight fragment read from
      // LDS then feeds four MFMAs instead of two (round 5: the 4-chain image launch issued 192 fragment reads per wave and
      // chain, now 64-128; LDS reads and MFMAs cost the same number of clocks at two MFMAs per read).
      // The weight fragments of row block nb + 1 are read while the MFMAs of row block nb run.
      h8 hb[NP][4];
      h8 a[4], an[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) a[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw1 + w_off<128>(i, 4 * kk + g)));
      unsigned obase[NP], ostp = 32;                          // element offsets: < 2^24 pixels x 128 channels
      if (!two) {
#pragma unroll
        for (int pg = 0; pg < NP; ++pg) {
          obase[pg] = (unsigned)pix[j0 + pg] * 128u + 8u * g;
          if (ch.hm > 0) {
            const int bmap = pix[j0 + pg] / ch.hm, t = pix[j0 + pg] - bmap * ch.hm;
            obase[pg] = (((unsigned)bmap * 8u + (g >> 1)) * ch.hm + t) * 16u + (g & 1) * 8u;
          }
        }
        if (ch.hm > 0) ostp = 2u * ch.hm * 16u;
      }
#pragma unroll
      for (int p2 = 0; p2 < 4; ++p2) {
        f4 acc[2][NP];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int nb = 2 * p2 + e;
          // (NP <= 2: the whole next row block is read ahead into a second fragment set; NP >= 3: a fragment feeds 3-4
          // MFMAs = 48-64 clocks of matrix work, the next FRAGMENT is read under them - one register set less)
          if constexpr (NP <= 2) {
            if (nb < 7) {
#pragma unroll
              for (int kk = 0; kk < 4; ++kk)
                an[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw1 + w_off<128>(16 * (nb + 1) + i, 4 * kk + g)));
              __builtin_amdgcn_sched_barrier(0);                   // reads first: they fly under this block's MFMAs
            }
          }
          {
            const f4 bias = *reinterpret_cast<const f4 *>(lb + 16 * nb + 4 * g);
#pragma unroll
            for (int pg = 0; pg < NP; ++pg) acc[e][pg] = bias;
          }
          if constexpr (NP <= 2) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
              for (int pg = 0; pg < NP; ++pg) acc[e][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk], xb[j0 + pg][kk], acc[e][pg], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a[kk] = an[kk];
          } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              // the fragment after this one: (nb, kk + 1), or (nb + 1, 0)
              const int nnb = kk < 3 ? nb : nb + 1, nkk = kk < 3 ? kk + 1 : 0;
              h8 nxt = a[0];
              if (nnb < 8) nxt = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw1 + w_off<128>(16 * nnb + i, 4 * nkk + g)));
#pragma unroll
              for (int pg = 0; pg < NP; ++pg) acc[e][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], xb[j0 + pg][kk], acc[e][pg], 0, 0, 0);
              a[0] = nxt;
            }
          }
        }
        // output rows of the LAST link are permuted in the image: fragment pair (2p, 2p+1) of lane group g holds
        // channels 32p + 8g + 0..7 -> one 16-B store per pair (8-B stores are store-issue bound: ~7 B/clk/CU)
#pragma unroll
        for (int pg = 0; pg < NP; ++pg) {
          h8 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = (_Float16)acc[0][pg][r];
            o[4 + r] = (_Float16)acc[1][pg][r];
          }
          o = __builtin_elementwise_max(o, floor1);
          if (two) hb[pg][p2] = o;
          else if (pix[j0 + pg] < Mi) *reinterpret_cast<h8 *>(ch.y + (size_t)(obase[pg] + p2 * ostp)) = o;
        }
      }
      if (!two) return;
      // ---- link 2: the hidden activations in registers are the B operand (k = 8g + 4t + r <-> channel 32kk + 16t + 4g + r)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) a[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw2 + w_off<128>(i, 4 * kk + g)));
#pragma unroll
      for (int p2 = 0; p2 < 4; ++p2) {
        f4 o2[2][NP];                                          // [fragment of the pair][pixel group]
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int nb = 2 * p2 + e;
          if constexpr (NP <= 2) {
            if (nb < 7) {
#pragma unroll
              for (int kk = 0; kk < 4; ++kk)
                an[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw2 + w_off<128>(16 * (nb + 1) + i, 4 * kk + g)));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          {
            const f4 bias = *reinterpret_cast<const f4 *>(lb + 128 + 16 * nb + 4 * g);
#pragma unroll
            for (int pg = 0; pg < NP; ++pg) o2[e][pg] = bias;
          }
          if constexpr (NP <= 2) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
              for (int pg = 0; pg < NP; ++pg) o2[e][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk], hb[pg][kk], o2[e][pg], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a[kk] = an[kk];
          } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const int nnb = kk < 3 ? nb : nb + 1, nkk = kk < 3 ? kk + 1 : 0;
              h8 nxt = a[0];
              if (nnb < 8) nxt = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw2 + w_off<128>(16 * nnb + i, 4 * nkk + g)));
#pragma unroll
              for (int pg = 0; pg < NP; ++pg) o2[e][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], hb[pg][kk], o2[e][pg], 0, 0, 0);
              a[0] = nxt;
            }
          }
        }
#pragma unroll
        for (int pg = 0; pg < NP; ++pg)
          if (pix[j0 + pg] < Mi) {
            h8 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              o[r] = (_Float16)o2[0][pg][r];
              o[4 + r] = (_Float16)o2[1][pg][r];
            }
            *reinterpret_cast<h8 *>(ch.y + (size_t)pix[j0 + pg] * 128 + 32 * p2 + 8 * g) = __builtin_elementwise_max(o, floor2);   // channels 32p + 8g + 0..7
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // plain form: up to FOUR groups share every weight fragment (then the fifth alone / the rest in a pair); all conditions
    // are wave-uniform and there is no barrier inside a step
    if constexpr (!WARP) {
      int nv = 0;
#pragma unroll
      for (int j = 0; j < NG; ++j) nv += (wg + j * stride < ngroups) ? 1 : 0;
      using IC0 = std::integral_constant<int, 0>;
      if constexpr (NG >= 4) {
        if (nv >= 4) {
          step(IC0{}, std::integral_constant<int, 4>{});
          if constexpr (NG == 5) {
            if (nv == 5) step(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
          }
          return;
        }
        if (nv == 3) {
          step(IC0{}, std::integral_constant<int, 3>{});
          return;
        }
      }
      if (nv == 2) step(IC0{}, std::integral_constant<int, 2>{});
      else if (nv == 1) step(IC0{}, std::integral_constant<int, 1>{});
      return;
    }
    // gathered form: groups in pairs; a wave with an odd number of groups runs its last one alone (half the MFMA work)
    static_for<0, (NG + 1) / 2>([&](auto S) {
      constexpr int j0 = 2 * decltype(S)::value;
      if constexpr (only < 0 || only == decltype(S)::value) {
        if (wg + j0 * stride < ngroups) {
          if constexpr (j0 + 1 < NG) {
            if (wg + (j0 + 1) * stride < ngroups) step(std::integral_constant<int, j0>{}, std::integral_constant<int, 2>{});
            else step(std::integral_constant<int, j0>{}, std::integral_constant<int, 1>{});
          } else {
            step(std::integral_constant<int, j0>{}, std::integral_constant<int, 1>{});
          }
        }
      }
    });
  };
  if constexpr (WARP) {
    static_for<0, (NG + 1) / 2>([&](auto S) {
      constexpr int j0 = 2 * decltype(S)::value;
      gather(std::integral_constant<int, j0>{});
      if constexpr (j0 + 1 < NG) gather(std::integral_constant<int, j0 + 1>{});
      if constexpr (j0 == 0) {                               // both images have landed (every wave's pieces)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      for (int c = 0; c < A.n; ++c) run_chain(c, S);
    });
  } else {
    for (int c = 0; c < A.n; ++c) run_chain(c, std::integral_constant<int, -1>{});
  }
}

// ------------------------------------------------------------------------------------------------------------
// Transformer FFN + post-norm in one pass over the tokens (DeepInteraction++ layers, mmcv FFN + the following
// LayerNorm: fusion_transformerv4.py operation_order (..., 'ffn', 'norm')):
//     y = LayerNorm(x + W2 . relu(W1 . x + b1) + b2),      x: (M, 128) fp16, hidden = 128 * n_chunks
// The hidden layer is walked in chunks of 128 channels: chunk c is a two-link chain (W1[c] : 128 x 128, then the columns
// c of W2 : 128 x 128) whose second link ACCUMULATES into the output registers - the (M, hidden) activation never exists
// in memory (274 MB of traffic per call at 134 400 tokens and hidden 512), ReLU / bias / residual / LayerNorm run on
// the fp32 accumulators.  Same machinery as pointwise_multi_kernel: a wave keeps its 2 x 16 tokens in registers as MFMA
// B operands, the chunk images (ops.chain_image of the chunk) arrive by LDS-DMA in a double buffer.
// ------------------------------------------------------------------------------------------------------------
struct FfnArgs {
  const unsigned char *img[8];   // chunk images (kChainImage bytes each); b2 lives in chunk 0's image, zeros elsewhere
  int n;
  const __half *ln_w, *ln_b;
  float eps;
  // single-link form  y = LayerNorm(res + W . x + b)  (the output projection of an attention block + its post-norm):
  // one image holding a one-link chain (ops.chain_image(w, b)), the residual from `res` instead of x
  const __half *res;
  int single;
  // optional second output: the PRE-normalisation sum  res + W . x + b  (fp16) - a DeepInteraction++ layer keeps the
  // un-normalised self-attention output next to its LayerNorm (fusion_transformerv4.py:187-190, `self_feat`)
  __half *presum;
};

__global__ __launch_bounds__(NT, 2) void ffn_ln_kernel(const __half *__restrict__ x, __half *__restrict__ y, FfnArgs A, long long M) {
  // PERSISTENT since round 5: one workgroup per CU walks the token pairs in passes (pair = 2 groups of 16 tokens per wave;
  // pass p gives wave w of workgroup b the pair p * waves + w * workgroups + b - wave-major, so a ragged last pass lands
  // on wave 0 of many workgroups and costs a fraction of a pass).  Before: one workgroup per 256 tokens - 134 400 tokens
  // = 525 workgroups = 2.05 rounds on 256 CUs ran as 3 (72 000 tokens: 1.1 as 2).  The chunk images are re-streamed per
  // pass (they never fit: 4 x 66.5 KB); the double buffer runs straight through the pass boundaries.
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int Mi = (int)M;
  const int ngroups = (Mi + 15) / 16, npairs = (ngroups + 1) / 2;
  const int waves = (int)gridDim.x * NW, npass = (npairs + waves - 1) / waves;
  const int nch = A.single ? 1 : A.n, total = npass * nch;
  auto dma_chunk = [&](const unsigned char *img, unsigned char *buf) {
    const unsigned char *src = img + lane * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {                              // W1 | W2: KiB chunks 0..63, 8 per wave
      const int chunk = wave * 8 + j;
      __builtin_amdgcn_global_load_lds((gptr_t)(src + chunk * 1024), (lptr_t)(buf + chunk * 1024), 16, 0, 0);
    }
    if (wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)(src + 64 * 1024), (lptr_t)(buf + 64 * 1024), 16, 0, 0);   // biases
  };
  dma_chunk(A.img[0], lds);
  int cs = 0;                                                  // chunk images streamed so far (buffer = cs & 1)
  for (int pass = 0; pass < npass; ++pass) {
    const int pidx = pass * waves + wave * (int)gridDim.x + (int)blockIdx.x;
    const bool active = pidx < npairs;                         // wave-uniform; idle waves still stream weights and meet the barriers
    h8 xb[2][4];
    int pix[2];
#pragma unroll
    for (int pg = 0; pg < 2; ++pg) {
      const int grp = 2 * pidx + pg;
      pix[pg] = (active && grp < ngroups) ? grp * 16 + i : Mi;
      const int pc = pix[pg] < Mi ? pix[pg] : Mi - 1;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        xb[pg][kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(x + (size_t)pc * 128 + kk * 32 + g * 8));
    }
    f4 out[2][8];
#pragma unroll
    for (int pg = 0; pg < 2; ++pg)
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) out[pg][nb] = f4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < nch; ++c, ++cs) {
      unsigned char *buf = lds + (cs & 1) * kChainImage;
      const unsigned char *lw1 = buf, *lw2 = buf + 128 * 128 * 2;
      const float *lb = reinterpret_cast<const float *>(buf + 2 * 128 * 128 * 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this chunk's image has landed (issued a whole chunk ago)
      __syncthreads();
      if (cs + 1 < total) dma_chunk(A.img[c + 1 < nch ? c + 1 : 0], lds + ((cs + 1) & 1) * kChainImage);
      if (!active) continue;
      // ---- link 1: hidden chunk = relu(W1[c] . x + b1[c]), one PAIR of row blocks at a time: the pair becomes k-step p2 of
      // link 2's operand at once (or, single-link form, two blocks of the output), so only 2 x 2 accumulators are live
      h8 hb[2][4];
#pragma unroll
      for (int p2 = 0; p2 < 4; ++p2) {
        f4 acc[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int nb = 2 * p2 + e;
#pragma unroll
          for (int pg = 0; pg < 2; ++pg) acc[e][pg] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const h8 a = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw1 + w_off<128>(16 * nb + i, 4 * kk + g)));
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) acc[e][pg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[pg][kk], acc[e][pg], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        const f4 ba = *reinterpret_cast<const f4 *>(lb + 16 * (2 * p2) + 4 * g), bb = *reinterpret_cast<const f4 *>(lb + 16 * (2 * p2 + 1) + 4 * g);
        if (A.single) {                                        // the link's accumulators (+ bias) ARE the output
#pragma unroll
          for (int pg = 0; pg < 2; ++pg) {
            out[pg][2 * p2] = acc[0][pg] + ba;
            out[pg][2 * p2 + 1] = acc[1][pg] + bb;
          }
        } else {
#pragma unroll
          for (int pg = 0; pg < 2; ++pg) {
            h8 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              t[r] = (_Float16)fmaxf(acc[0][pg][r] + ba[r], 0.f);
              t[4 + r] = (_Float16)fmaxf(acc[1][pg][r] + bb[r], 0.f);
            }
            hb[pg][p2] = t;
          }
        }
      }
      if (A.single) continue;
      // ---- link 2: out += W2[:, chunk c] . hidden chunk  (+ the chunk image's b2: the real bias in chunk 0, zeros after)
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const h8 a = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(lw2 + w_off<128>(16 * nb + i, 4 * kk + g)));
#pragma unroll
          for (int pg = 0; pg < 2; ++pg) out[pg][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, hb[pg][kk], out[pg][nb], 0, 0, 0);
        }
        const f4 bias = *reinterpret_cast<const f4 *>(lb + 128 + 16 * nb + 4 * g);
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) out[pg][nb] += bias;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!active) continue;
    // ---- epilogue: residual + LayerNorm over the 128 channels of a token (32 per lane, 4 lanes per token), 16-B stores.
    // Image row 16nb + 4g + r of the last link is channel 32(nb/2) + 8g + 4(nb%2) + r = element 4(nb%2) + r of xb[.][nb/2].
#pragma unroll
    for (int pg = 0; pg < 2; ++pg) {
      if (A.single) {                                          // residual from its own tensor, same channel layout
        const int pc = pix[pg] < Mi ? pix[pg] : Mi - 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          xb[pg][kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(A.res + (size_t)pc * 128 + kk * 32 + g * 8));
      }
      float sum = 0.f;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = out[pg][nb][r] + (float)xb[pg][nb >> 1][4 * (nb & 1) + r];
          out[pg][nb][r] = v;
          sum += v;
        }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      const float mean = sum * (1.f / 128.f);
      float var = 0.f;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = out[pg][nb][r] - mean;
          var += d * d;
        }
      var += __shfl_xor(var, 16);
      var += __shfl_xor(var, 32);
      const float rstd = rsqrtf(var * (1.f / 128.f) + A.eps);
      if (A.presum != nullptr && pix[pg] < Mi) {
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
          h8 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = (_Float16)out[pg][2 * p2][r];
            o[4 + r] = (_Float16)out[pg][2 * p2 + 1][r];
          }
          *reinterpret_cast<h8 *>(A.presum + (size_t)pix[pg] * 128 + 32 * p2 + 8 * g) = o;
        }
      }
      if (pix[pg] < Mi) {
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
          const h8 lw = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(A.ln_w + 32 * p2 + 8 * g));
          const h8 lbv = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(A.ln_b + 32 * p2 + 8 * g));
          h8 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = (_Float16)((out[pg][2 * p2][r] - mean) * rstd * (float)lw[r] + (float)lbv[r]);
            o[4 + r] = (_Float16)((out[pg][2 * p2 + 1][r] - mean) * rstd * (float)lw[4 + r] + (float)lbv[4 + r]);
          }
          *reinterpret_cast<h8 *>(y + (size_t)pix[pg] * 128 + 32 * p2 + 8 * g) = o;
        }
      }
    }
  }
}

template <int NG, bool WARP>
static int launch_multi(const void *x, const MultiArgs &A, long long M, long long grid, hipStream_t stream,
                        const WarpArgs &Wp = WarpArgs{}) {
  constexpr int LDS = 2 * kChainImage;
  static LdsRaised lds_raised;
  if (int rc = ensure_lds(lds_raised, (const void *)pointwise_multi_kernel<NG, WARP>, LDS)) return rc;
  hipEvent_t ev0, ev1;
  if (take_launch_events(ev0, ev1))                          // measurement: the dispatch's own begin / end time stamps
    hipExtLaunchKernelGGL((pointwise_multi_kernel<NG, WARP>), dim3((unsigned)grid), dim3(NT), LDS, stream, ev0, ev1, 0,
                          (const __half *)x, A, M, Wp);
  else
    hipLaunchKernelGGL((pointwise_multi_kernel<NG, WARP>), dim3((unsigned)grid), dim3(NT), LDS, stream, (const __half *)x, A, M, Wp);
  return check_launch("pointwise_multi");
}

// chains of one launch + the split of the map over the workgroups (shared by the plain and the gathered form)
static int multi_setup(MultiArgs &A, int n_chains, const void *const *image, void *const *y, const int *relu1, const int *relu2,
                       const int *two_links, long long n_pixels, long long &grid, int &ng, const int *hm = nullptr) {
  DI_REQUIRE(n_chains >= 1 && n_chains <= kMaxChains, "1..%d chains, got %d", kMaxChains, n_chains);
  A.n = n_chains;
  for (int c = 0; c < kMaxChains; ++c) {
    if (c < n_chains) {
      DI_REQUIRE(image[c] && y[c], "chain %d: image and y are required", c);
      A.c[c] = Chain{(const unsigned char *)image[c], (__half *)y[c], relu1[c], relu2[c], two_links[c], hm ? hm[c] : 0};
      DI_REQUIRE(A.c[c].hm == 0 || (!two_links[c] && A.c[c].hm > 0 && n_pixels % A.c[c].hm == 0),
                 "chain %d: a head-major output takes a single-link chain and whole maps (%d tokens per map)", c, A.c[c].hm);
    } else {
      A.c[c] = Chain{nullptr, nullptr, 0, 0, 0, 0};
    }
  }
  const int n_cu = di::device_cus();
  if (n_cu <= 0) return DI_ERR_LAUNCH;
  // every wave keeps its pixel groups in registers: NG in {2, 4, 5} groups per wave, one workgroup per CU when the map
  // allows it (more workgroups than CUs only beyond 5 groups per wave)
  const long long ngroups = (n_pixels + 15) / 16;
  grid = n_cu;
  if (ngroups < (long long)grid * NW) grid = (ngroups + NW - 1) / NW;        // small map: one group per wave, every CU busy
  ng = (int)((ngroups + grid * NW - 1) / (grid * NW));
  if (ng > 5) {
    grid = (ngroups + NW * 5 - 1) / (NW * 5);
    ng = 5;
  }
  return DI_OK;
}

}  // namespace pw
}  // namespace di

extern "C" int di_ffn_ln_fwd_ex(const void *x, int n_chunks, const void *const *image, const void *residual, const void *ln_w,
                                const void *ln_b, float eps, void *y, void *presum, long long n_tokens, void *stream);

extern "C" int di_ffn_ln_fwd(const void *x, int n_chunks, const void *const *image, const void *residual, const void *ln_w,
                             const void *ln_b, float eps, void *y, long long n_tokens, void *stream) {
  return di_ffn_ln_fwd_ex(x, n_chunks, image, residual, ln_w, ln_b, eps, y, nullptr, n_tokens, stream);
}

extern "C" int di_ffn_ln_fwd_ex(const void *x, int n_chunks, const void *const *image, const void *residual, const void *ln_w,
                                const void *ln_b, float eps, void *y, void *presum, long long n_tokens, void *stream) {
  using namespace di::pw;
  DI_REQUIRE(n_tokens > 0 && n_tokens < (1ll << 24) && x && y && ln_w && ln_b, "bad token count (1 .. 2^24 - 1)");
  DI_REQUIRE(n_chunks >= 1 && n_chunks <= 8 && image, "hidden width = 128 x (1 .. 8)");
  FfnArgs A;
  for (int c = 0; c < 8; ++c) A.img[c] = c < n_chunks ? (const unsigned char *)image[c] : nullptr;
  for (int c = 0; c < n_chunks; ++c) DI_REQUIRE(A.img[c], "chunk image %d missing", c);
  A.n = n_chunks;
  A.ln_w = (const __half *)ln_w;
  A.ln_b = (const __half *)ln_b;
  A.eps = eps;
  A.res = (const __half *)residual;
  A.single = residual != nullptr;
  A.presum = (__half *)presum;
  DI_REQUIRE(!A.single || n_chunks == 1, "the single-link form takes one image");
  constexpr int LDS = 2 * kChainImage;
  static di::LdsRaised raised;
  if (int rc = di::ensure_lds(raised, (const void *)ffn_ln_kernel, LDS)) return rc;
  const long long groups = (n_tokens + 15) / 16, pairs = (groups + 1) / 2;
  const int n_cu = di::device_cus();
  if (n_cu <= 0) return DI_ERR_LAUNCH;
  long long grid = (pairs + NW - 1) / NW;                      // persistent: at most one workgroup per CU
  if (grid > n_cu) grid = n_cu;
  hipLaunchKernelGGL(ffn_ln_kernel, dim3((unsigned)grid), dim3(NT), LDS, (hipStream_t)stream, (const __half *)x, (__half *)y, A,
                     n_tokens);
  return di::check_launch("ffn_ln");
}

extern "C" int di_pointwise_multi_fwd(const void *x, int n_chains, const void *const *image, void *const *y,
                                      const int *relu1, const int *relu2, const int *two_links, long long n_pixels,
                                      void *stream) {
  using namespace di::pw;
  DI_REQUIRE(n_pixels > 0 && n_pixels < (1ll << 24) && x, "bad map size (1 .. 2^24 - 1 pixels)");
  MultiArgs A;
  long long grid;
  int ng;
  if (int rc = multi_setup(A, n_chains, image, y, relu1, relu2, two_links, n_pixels, grid, ng)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (ng <= 2) return launch_multi<2, false>(x, A, n_pixels, grid, s);
  if (ng <= 4) return launch_multi<4, false>(x, A, n_pixels, grid, s);
  return launch_multi<5, false>(x, A, n_pixels, grid, s);
}

extern "C" int di_pointwise_multi_warp_fwd(const void *bev, const float *depth, const float *img2lidar, const float *aug_fwd,
                                           const float *xs, const float *ys, const float *pc_range, int n_views, int Hi,
                                           int Wi, int Hb, int Wb, int n_chains, const void *const *image, void *const *y,
                                           const int *relu1, const int *relu2, const int *two_links, void *stream) {
  using namespace di::pw;
  const long long n_pixels = (long long)n_views * Hi * Wi;
  DI_REQUIRE(n_views > 0 && Hi > 0 && Wi > 0 && Hb > 0 && Wb > 0 && n_pixels < (1ll << 24), "bad gather shape");
  DI_REQUIRE(bev && depth && img2lidar && aug_fwd && xs && ys && pc_range, "null geometry");
  DI_REQUIRE(n_chains <= 2, "the gathered form keeps both weight images resident: at most 2 chains, got %d", n_chains);
  MultiArgs A;
  long long grid;
  int ng;
  if (int rc = multi_setup(A, n_chains, image, y, relu1, relu2, two_links, n_pixels, grid, ng)) return rc;
  const WarpArgs Wp{(const __half *)bev, depth, img2lidar, aug_fwd, xs, ys, pc_range, Hi, Wi, Hb, Wb};
  hipStream_t s = (hipStream_t)stream;
  if (ng <= 2) return launch_multi<2, true>(nullptr, A, n_pixels, grid, s, Wp);
  if (ng <= 4) return launch_multi<4, true>(nullptr, A, n_pixels, grid, s, Wp);
  return launch_multi<5, true>(nullptr, A, n_pixels, grid, s, Wp);
}

extern "C" int di_pointwise_chain_hm_fwd(const void *x, const void *w, const float *b, void *y_hm, long long n_tokens,
                                         int tokens_per_map, int relu, void *stream) {
  DI_REQUIRE(x && w && b && y_hm && n_tokens > 0 && n_tokens < (1ll << 31), "bad arguments");
  DI_REQUIRE(tokens_per_map > 0 && n_tokens % tokens_per_map == 0, "whole maps of %d tokens", tokens_per_map);
  return di::pw::launch<128, 0>(x, nullptr, nullptr, w, b, nullptr, nullptr, y_hm, n_tokens, relu, 0, nullptr, nullptr,
                                (hipStream_t)stream, tokens_per_map);
}

extern "C" int di_pointwise_multi_warp_hm_fwd(const void *bev, const float *depth, const float *img2lidar, const float *aug_fwd,
                                              const float *xs, const float *ys, const float *pc_range, int n_views, int Hi,
                                              int Wi, int Hb, int Wb, int n_chains, const void *const *image, void *const *y,
                                              const int *relu1, const int *relu2, const int *two_links, const int *hm_tokens,
                                              void *stream) {
  using namespace di::pw;
  const long long n_pixels = (long long)n_views * Hi * Wi;
  DI_REQUIRE(n_views > 0 && Hi > 0 && Wi > 0 && Hb > 0 && Wb > 0 && n_pixels < (1ll << 24), "bad gather shape");
  DI_REQUIRE(bev && depth && img2lidar && aug_fwd && xs && ys && pc_range, "null geometry");
  DI_REQUIRE(n_chains <= 2, "the gathered form keeps both weight images resident: at most 2 chains, got %d", n_chains);
  MultiArgs A;
  long long grid;
  int ng;
  if (int rc = multi_setup(A, n_chains, image, y, relu1, relu2, two_links, n_pixels, grid, ng, hm_tokens)) return rc;
  const WarpArgs Wp{(const __half *)bev, depth, img2lidar, aug_fwd, xs, ys, pc_range, Hi, Wi, Hb, Wb};
  hipStream_t s = (hipStream_t)stream;
  if (ng <= 2) return launch_multi<2, true>(nullptr, A, n_pixels, grid, s, Wp);
  if (ng <= 4) return launch_multi<4, true>(nullptr, A, n_pixels, grid, s, Wp);
  return launch_multi<5, true>(nullptr, A, n_pixels, grid, s, Wp);
}

extern "C" int di_pointwise_chain_masked_fwd(const void *x1, const void *x2, const void *x3, const void *w1,
                                             const float *b1, const void *w2, const float *b2, void *y,
                                             long long n_pixels, int k1, int k2, int relu1, int relu2,
                                             const void *mask, const float *bm, void *stream) {
  DI_REQUIRE(n_pixels > 0, "empty map");
  DI_REQUIRE((mask == nullptr) == (bm == nullptr), "mask and its bias come together");
  DI_REQUIRE(x1 && w1 && b1 && y, "x1, w1, b1, y are required");
  DI_REQUIRE((k1 == 128 && !x2) || (k1 == 256 && x2), "k1 = 128 (x1) or 256 (x1 ; x2), got %d", k1);
  DI_REQUIRE(k2 == 0 || (w2 && b2), "second link needs w2 and b2");
  DI_REQUIRE((k2 == 0 && !x3) || (k2 == 128 && !x3) || (k2 == 256 && x3), "k2 = 0, 128 (h) or 256 (h ; x3), got %d", k2);
  hipStream_t s = (hipStream_t)stream;
#define DI_PW(A, B) \
  if (k1 == A && k2 == B) return di::pw::launch<A, B>(x1, x2, x3, w1, b1, w2, b2, y, n_pixels, relu1, relu2, mask, bm, s)
  DI_PW(128, 0);
  DI_PW(128, 128);
  DI_PW(256, 0);
  DI_PW(256, 256);
  DI_PW(128, 256);
  DI_PW(256, 128);
#undef DI_PW
  di::set_error("unsupported chain k1=%d k2=%d", k1, k2);
  return DI_ERR_ARG;
}

extern "C" int di_pointwise_chain_fwd(const void *x1, const void *x2, const void *x3, const void *w1,
                                      const float *b1, const void *w2, const float *b2, void *y,
                                      long long n_pixels, int k1, int k2, int relu1, int relu2, void *stream) {
  return di_pointwise_chain_masked_fwd(x1, x2, x3, w1, b1, w2, b2, y, n_pixels, k1, k2, relu1, relu2, nullptr, nullptr,
                                       stream);
}
