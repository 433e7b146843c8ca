// Fused 9x9 local-window attention on the gfx950 matrix cores (fp16 maps, C = 128).
//
// Banded attention recast as dense 16x16x32 MFMA tiles - one wavefront owns a strip of 16
// query pixels of one image row; for each of the 9 window rows the 24 candidate keys
// (x0-4 .. x0+19) are covered by two 16-key tiles:
//
//   S^T = K . Q^T      A = K tile  [key i = lane%16][ch 8*(lane/16)+j]   (16 B rows of the LDS halo)
//                      B = Q^T     [ch 8*(lane/16)+j][query lane%16]     (16 B straight from global)
//                      C -> lane holds, for query lane%16, the keys 4*(lane/16)+r of the tile
//   softmax            per query = per lane column: in-lane over (row, tile, r) + two cross-row
//                      exchanges; band mask |dx| <= 4 is a per-lane 8-bit constant; out-of-image
//                      keys are zero texels -> score 0, kept in the softmax (reference semantics)
//   O^T = V^T . P^T    A = V^T     [ch lane%16][key]  via ds_read_b64_tr_b16 (hardware transpose
//                                  of a [4 keys][16 ch] block of the row-major V halo)
//                      B = P^T     = the softmax registers of the lane, converted to fp16, with
//                                  the MFMA k index permuted as k = 8*(lane/16) + 4*tile + r so
//                                  that NO cross-lane movement is needed between the two GEMMs
//                      C -> lane holds, for query lane%16, channels 16n + 4*(lane/16) + r
//
// A workgroup is 4 wavefronts = a 16 x 4 pixel tile; its 24 x 12 texel halo (72 KiB) is staged
// once for K and once for V in LDS (two workgroups per CU), 32-B segments XOR-swizzled by the
// key column so that both the ds_read_b128 K fragments and the transposed V reads are
// bank-conflict free.  Per strip: 72 + 72 MFMAs, ~400 VALU ops of softmax; scores and weights
// never leave registers.
#include "di_common.h"

namespace di {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f4 __attribute__((ext_vector_type(4)));

namespace mf {
constexpr int TW = 16, TH = 4;          // tile: 16 queries per strip, 4 strips (rows) per workgroup
constexpr int KS = 9, R = 4;            // window 9x9
constexpr int HC = TW + KS - 1;         // 24 halo columns
constexpr int HR = TH + KS - 1;         // 12 halo rows
constexpr int PITCH = HC + 1;           // +1 all-zero texel per row (keys 24..31 of the second tile)
constexpr int TEXEL = 256;              // bytes: 128 ch x fp16
constexpr int ROWB = PITCH * TEXEL;     // 6400 B
constexpr int LDS_BYTES = HR * ROWB;    // 76 800 B -> two workgroups per CU
}  // namespace mf

// Stage the halo of `src` (zero outside the image) into the swizzled LDS image.
__device__ __forceinline__ void mf_stage(const __half *__restrict__ src, unsigned char *lds, int img,
                                         int y0, int x0, int H, int W, int tid) {
  using namespace mf;
  const int p = tid & 15;              // 16-B chunk of the texel
  const int e0 = tid >> 4;             // 16 texels per pass
  constexpr int NT = HR * HC;          // 288 texels -> 18 passes, issued 6 loads deep
#pragma unroll
  for (int base = 0; base < NT; base += 16 * 6) {
    uint4 v[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int e = base + u * 16 + e0;
      const int hr = e / HC, hc = e - hr * HC;
      const int gy = y0 - R + hr, gx = x0 - R + hc;
      v[u] = make_uint4(0, 0, 0, 0);
      if (e < NT && gy >= 0 && gy < H && gx >= 0 && gx < W)
        v[u] = *reinterpret_cast<const uint4 *>(src + ((size_t)(img * H + gy) * W + gx) * 128 + p * 8);
    }
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int e = base + u * 16 + e0;
      const int hr = e / HC, hc = e - hr * HC;
      if (e < NT) {
        const int off = hr * ROWB + hc * TEXEL + ((((p >> 1) ^ (hc & 7)) << 5) | ((p & 1) << 4));
        *reinterpret_cast<uint4 *>(lds + off) = v[u];
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void local_attn_mfma_kernel(
    const __half *__restrict__ q, const __half *__restrict__ k, const __half *__restrict__ v,
    __half *__restrict__ out, int n, int H, int W, float scale, int tiles_x, int tiles_y) {
  using namespace mf;
  extern __shared__ __align__(16) unsigned char lds[];
  const int per_img = tiles_x * tiles_y;
  const int bid = xcd_remap(blockIdx.x, n * per_img);
  const int img = bid / per_img;
  const int rr = bid - img * per_img;
  const int ty = rr / tiles_x;
  const int y0 = ty * TH, x0 = (rr - ty * tiles_x) * TW;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wrow = tid >> 6;        // wave w owns query row y0 + w
  const int i = lane & 15, g = lane >> 4;
  const int gy = y0 + wrow, gx = x0 + i;
  const bool pix_ok = gy < H && gx < W;

  // the zero texel of every halo row (never overwritten by mf_stage)
  if (tid < HR * 16) *reinterpret_cast<uint4 *>(lds + (tid >> 4) * ROWB + HC * TEXEL + (tid & 15) * 16) =
      make_uint4(0, 0, 0, 0);

  // Q^T fragments straight from global: query i, channels kk*32 + 8g .. +7
  h8 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    uint4 raw = make_uint4(0, 0, 0, 0);
    if (pix_ok)
      raw = *reinterpret_cast<const uint4 *>(q + ((size_t)(img * H + gy) * W + gx) * 128 + kk * 32 + g * 8);
    qf[kk] = __builtin_bit_cast(h8, raw);
  }

  mf_stage(k, lds, img, y0, x0, H, W, tid);
  __syncthreads();

  // ---- S^T = K . Q^T : s[dy][t] holds keys c = 16t + 4g + r for query i
  f4 s[KS][2];
  {
    // K fragment address of (tile t, channel group kk): key column hc = 16t + i (zero texel when >= 24)
    const int hc1 = (16 + i < HC) ? 16 + i : HC;
    const int b0 = wrow * ROWB + i * TEXEL, b1 = wrow * ROWB + hc1 * TEXEL;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = ((((2 * kk + (g >> 1)) ^ (i & 7)) << 5) | ((g & 1) << 4));
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const uint4 raw = *reinterpret_cast<const uint4 *>(lds + (t ? b1 : b0) + koff[kk] + dy * ROWB);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, raw), qf[kk], acc, 0, 0, 0);
        }
        s[dy][t] = acc;
      }
    }
  }

  // ---- softmax over the 81 window slots of query i (valid band: i <= c <= i + 8)
  unsigned band = 0;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * t + 4 * g + r;
      if (c >= i && c <= i + 8) band |= 1u << (4 * t + r);
    }
  float m = -INFINITY;
#pragma unroll
  for (int dy = 0; dy < KS; ++dy)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = (band >> (4 * t + r)) & 1u ? s[dy][t][r] * scale : -INFINITY;
        s[dy][t][r] = x;
        m = fmaxf(m, x);
      }
  m = fmaxf(m, __shfl_xor(m, 16));
  m = fmaxf(m, __shfl_xor(m, 32));
  float sum = 0.f;
  h8 pf[KS];  // P^T fragments: k = 8g + 4t + r  <->  key c = 16t + 4g + r
#pragma unroll
  for (int dy = 0; dy < KS; ++dy) {
    h8 pk;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[dy][t][r] - m);   // masked slots: exp(-inf) = 0
        sum += e;
        pk[4 * t + r] = (_Float16)e;
      }
    pf[dy] = pk;
  }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);

  __syncthreads();   // every wave is done with the K halo
  mf_stage(v, lds, img, y0, x0, H, W, tid);
  __syncthreads();

  // ---- O^T = V^T . P^T : acc[n] holds channels 16n + 4g + r of query i
  f4 acc[8];
#pragma unroll
  for (int nn = 0; nn < 8; ++nn) acc[nn] = f4{0.f, 0.f, 0.f, 0.f};
  {
    // transposed V read: lane supplies row (key) 4g + i/4 of the tile, columns 4*(i%4) of the 16-ch block
    const int kc0 = 4 * g + (i >> 2);
    const int kc1 = (16 + kc0 < HC) ? 16 + kc0 : HC;
    const int vb0 = wrow * ROWB + kc0 * TEXEL + (i & 3) * 8;
    const int vb1 = wrow * ROWB + kc1 * TEXEL + (i & 3) * 8;
    const int sw0 = kc0 & 7, sw1 = kc1 & 7;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
      for (int nn = 0; nn < 8; ++nn) {
        const hv4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (hv4 __attribute__((address_space(3))) *)(lds + vb0 + ((nn ^ sw0) << 5) + dy * ROWB));
        const hv4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (hv4 __attribute__((address_space(3))) *)(lds + vb1 + ((nn ^ sw1) << 5) + dy * ROWB));
        h8 a;
        a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3];
        a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
        acc[nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pf[dy], acc[nn], 0, 0, 0);
      }
    }
  }

  if (pix_ok) {
    const float inv = 1.f / sum;
    __half *dst = out + ((size_t)(img * H + gy) * W + gx) * 128 + 4 * g;
#pragma unroll
    for (int nn = 0; nn < 8; ++nn) {
      h4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (_Float16)(acc[nn][r] * inv);
      *reinterpret_cast<h4 *>(dst + 16 * nn) = o;     // 8 B per lane; the 4 lanes of a query write 32 B
    }
  }
}

int launch_local_attn_mfma(const void *q, const void *k, const void *v, void *out, int n, int H, int W,
                           float scale, hipStream_t stream) {
  using namespace mf;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  static bool attr_set = false;   // idempotent; a race only repeats the call
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)local_attn_mfma_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DI_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(local_attn_mfma_kernel, dim3(n * tiles_x * tiles_y), dim3(256), LDS_BYTES, stream,
                     (const __half *)q, (const __half *)k, (const __half *)v, (__half *)out, n, H, W, scale,
                     tiles_x, tiles_y);
  return check_launch("local_attn_mfma");
}

}  // namespace di
