// Local-window attention for gfx950: the fused forward of
// LocalContextAttentionBlock (reference encoder_utils.py:132-134) and the three
// window kernels behind the five locatt_ops entry points (reference
// locatt_ops/kernels.cuh:4-119), re-designed for CDNA4:
//
//   * channels-last maps: one texel = C contiguous channels; 16 lanes own one
//     texel (8 channels = 16 B fp16 per lane), so a wavefront moves four adjacent
//     texels = 1 KiB per instruction, fully coalesced.
//   * one workgroup (16 wavefronts = 64 texel groups) owns a 16 x 4 pixel tile, ONE pixel
//     per 16-lane group (so the 81 softmax weights of a pixel are 6 registers per lane
//     with static indices and the code stays small), and stages the
//     (TW+kW-1) x (TH+kH-1) halo of K (then V) ONCE in LDS; the 81 window reads per
//     pixel are 16-B ds_read_b128, conflict-free (a 16-lane group reads one
//     contiguous 256 B texel).  Out-of-image halo texels are zero-filled, which
//     reproduces the reference's "OOB slot scores 0 and stays in the softmax".
//   * channel reduction over the 16 lanes of a texel with DPP row ops (no LDS);
//     softmax weights never leave registers (ds_bpermute broadcast in the PV pass),
//     so the (n,H,W,81) weight tensor of the reference never touches HBM.
//   * fp32 accumulation for both element types (v_dot2_f32_f16 for fp16).
//   * block->tile map is XCD-aware (neighbouring tiles share halos in one L2).
#include <stdlib.h>

#include "di_common.h"

namespace di {

constexpr int kTW = 16;   // tile width  (pixels)
constexpr int kTH = 4;    // tile height (pixels)
constexpr int kGroups = kTW * kTH;             // 64 texel groups of 16 lanes
constexpr int kThreads = kGroups * kLanesPerTexel;  // 1024 threads = 16 wavefronts

template <int KH, int KW>
struct LaCfg {
  static constexpr int HW = kTW + KW - 1;   // halo width
  static constexpr int HH = kTH + KH - 1;   // halo height
  static constexpr int K = KH * KW;
  // window slot (dy,dx) is kept in lane dx (of the texel's 16 lanes), register dy: the dy loop
  // is unrolled (static register index), the dx loop is rolled (small code, <= 64 VGPRs).
  static_assert(KW <= 16, "window row must fit the 16 lanes of a texel group");
};

struct Tile {
  int img, x0, y0;
};
__device__ __forceinline__ Tile tile_of_block(int n, int tiles_x, int tiles_y) {
  const int per_img = tiles_x * tiles_y;
  const int bid = xcd_remap(blockIdx.x, n * per_img);
  Tile t;
  t.img = bid / per_img;
  const int r = bid - t.img * per_img;
  const int ty = r / tiles_x;
  t.y0 = ty * kTH;
  t.x0 = (r - ty * tiles_x) * kTW;
  return t;
}

// Stage the halo of `src` for a tile into LDS, zero-filling outside the image.
template <typename T, int HH, int HW, int RH, int RW, bool FULLC>
__device__ __forceinline__ void stage_halo(const T *__restrict__ src, T *halo, const Tile &t, int H,
                                           int W, int C, int grp, int l16) {
  const bool ch_ok = FULLC || l16 * kChPerLane < C;
  for (int e = grp; e < HH * HW; e += kGroups) {
    const int hy = e / HW, hx = e - hy * HW;
    const int gy = t.y0 - RH + hy, gx = t.x0 - RW + hx;
    Pack8<T> val = zero8<T>();
    if (ch_ok && gy >= 0 && gy < H && gx >= 0 && gx < W)
      val = ld8(src + ((size_t)(t.img * H + gy) * W + gx) * C + l16 * kChPerLane);
    if (ch_ok) st8(halo + (size_t)e * C + l16 * kChPerLane, val);
  }
}

__device__ __forceinline__ float bcast16(float x, int src_lane_in_row, int row_base) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((row_base + src_lane_in_row) << 2,
                                                               __builtin_bit_cast(int, x)));
}

// ---------------------------------------------------------------------------------
// Fused forward: out = softmax_k(<q, k_k> * scale) . v_k
// ---------------------------------------------------------------------------------
template <typename T, int KH, int KW, int MINW, bool FULLC>
__global__ __launch_bounds__(kThreads, MINW) void local_attn_fwd_kernel(
    const T *__restrict__ q, const T *__restrict__ k, const T *__restrict__ v, T *__restrict__ out,
    int n, int H, int W, int C, float scale, int tiles_x, int tiles_y) {
  using Cfg = LaCfg<KH, KW>;
  constexpr int HW = Cfg::HW, HH = Cfg::HH;
  extern __shared__ __align__(16) unsigned char smem[];
  T *halo = reinterpret_cast<T *>(smem);

  const Tile t = tile_of_block(n, tiles_x, tiles_y);
  const int tid = threadIdx.x;
  const int l16 = tid & 15, grp = tid >> 4;
  const int px = grp & (kTW - 1), py = grp / kTW;  // a wavefront = 4 adjacent pixels of one row
  const int gx = t.x0 + px, gy = t.y0 + py;
  const bool ch_ok = FULLC || l16 * kChPerLane < C;  // FULLC: C == 128, every lane owns channels
  const bool pix_ok = gx < W && gy < H;
  const size_t pix_off = ((size_t)(t.img * H + gy) * W + gx) * C + l16 * kChPerLane;

  Pack8<T> qv = zero8<T>();
  if (pix_ok && ch_ok) qv = ld8(q + pix_off);
  stage_halo<T, HH, HW, KH / 2, KW / 2, FULLC>(k, halo, t, H, W, C, grp, l16);
  __syncthreads();

  const T *win = halo + (size_t)(py * HW + px) * C + l16 * kChPerLane;  // window origin texel
  float s[KH];
#pragma unroll
  for (int dy = 0; dy < KH; ++dy) {
    s[dy] = -INFINITY;
#pragma unroll(KW % 3 == 0 ? 3 : 1)
    for (int dx = 0; dx < KW; ++dx) {
      float part = 0.f;
      if (ch_ok) part = dot8(qv, ld8(win + (size_t)(dy * HW + dx) * C), 0.f);
      const float tot = row16_sum(part) * scale;
      if (l16 == dx) s[dy] = tot;
    }
  }
  float m = s[0];
#pragma unroll
  for (int i = 1; i < KH; ++i) m = fmaxf(m, s[i]);
  m = row16_max(m);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < KH; ++i) {
    s[i] = __expf(s[i] - m);  // lanes >= KW hold -inf: exp = 0
    sum += s[i];
  }
  sum = row16_sum(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int i = 0; i < KH; ++i) s[i] *= inv;

  __syncthreads();  // every group is done with the K halo
  stage_halo<T, HH, HW, KH / 2, KW / 2, FULLC>(v, halo, t, H, W, C, grp, l16);
  __syncthreads();

  const int row_base = (tid & 63) & ~15;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
  for (int dy = 0; dy < KH; ++dy) {
#pragma unroll 1
    for (int dx = 0; dx < KW; ++dx) {
      const float w = bcast16(s[dy], dx, row_base);
      if (ch_ok) {
        float f[8];
        unpack8(ld8(win + (size_t)(dy * HW + dx) * C), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(w, f[i], acc[i]);
      }
    }
  }
  if (pix_ok && ch_ok) st8(out + pix_off, pack8f(acc, T()));
}

// ---------------------------------------------------------------------------------
// cc2k (kernels.cuh:4-42): y[p,k] = <a[p], b[p+off_k]>, 0 for OOB slots (still written)
// ---------------------------------------------------------------------------------
template <typename T, int KH, int KW, int MINW, bool FULLC>
__global__ __launch_bounds__(kThreads, MINW) void cc2k_kernel(const T *__restrict__ a,
                                                              const T *__restrict__ b,
                                                              float *__restrict__ y, int n, int H,
                                                              int W, int C, int tiles_x, int tiles_y) {
  using Cfg = LaCfg<KH, KW>;
  constexpr int HW = Cfg::HW, HH = Cfg::HH, K = Cfg::K;
  extern __shared__ __align__(16) unsigned char smem[];
  T *halo = reinterpret_cast<T *>(smem);
  const Tile t = tile_of_block(n, tiles_x, tiles_y);
  const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
  const int px = grp & (kTW - 1), py = grp / kTW;
  const int gx = t.x0 + px, gy = t.y0 + py;
  const bool ch_ok = FULLC || l16 * kChPerLane < C;  // FULLC: C == 128, every lane owns channels
  const bool pix_ok = gx < W && gy < H;
  const size_t pix = (size_t)(t.img * H + gy) * W + gx;
  Pack8<T> av = zero8<T>();
  if (pix_ok && ch_ok) av = ld8(a + pix * C + l16 * kChPerLane);
  stage_halo<T, HH, HW, KH / 2, KW / 2, FULLC>(b, halo, t, H, W, C, grp, l16);
  __syncthreads();
  const T *win = halo + (size_t)(py * HW + px) * C + l16 * kChPerLane;
  float s[KH];
#pragma unroll
  for (int dy = 0; dy < KH; ++dy) {
    s[dy] = 0.f;
#pragma unroll(KW % 3 == 0 ? 3 : 1)
    for (int dx = 0; dx < KW; ++dx) {
      float part = 0.f;
      if (ch_ok) part = dot8(av, ld8(win + (size_t)(dy * HW + dx) * C), 0.f);
      const float tot = row16_sum(part);
      if (l16 == dx) s[dy] = tot;
    }
  }
  if (pix_ok && l16 < KW) {
    float *dst = y + pix * K + l16;
#pragma unroll
    for (int dy = 0; dy < KH; ++dy) dst[dy * KW] = s[dy];
  }
}

// ---------------------------------------------------------------------------------
// ck2c_ori (kernels.cuh:44-80): y[p,:] = sum_k w[p,k] * x[p+off_k,:]
// ---------------------------------------------------------------------------------
template <typename T, int KH, int KW, int MINW, bool FULLC>
__global__ __launch_bounds__(kThreads, MINW) void ck2c_ori_kernel(const T *__restrict__ x,
                                                                  const float *__restrict__ w,
                                                                  T *__restrict__ y, int n, int H,
                                                                  int W, int C, int tiles_x,
                                                                  int tiles_y) {
  using Cfg = LaCfg<KH, KW>;
  constexpr int HW = Cfg::HW, HH = Cfg::HH, K = Cfg::K;
  extern __shared__ __align__(16) unsigned char smem[];
  T *halo = reinterpret_cast<T *>(smem);
  const Tile t = tile_of_block(n, tiles_x, tiles_y);
  const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
  const int px = grp & (kTW - 1), py = grp / kTW;
  const int gx = t.x0 + px, gy = t.y0 + py;
  const bool ch_ok = FULLC || l16 * kChPerLane < C;  // FULLC: C == 128, every lane owns channels
  const bool pix_ok = gx < W && gy < H;
  const size_t pix = (size_t)(t.img * H + gy) * W + gx;
  float wreg[KH];  // weight of slot (dy,dx) in lane dx, register dy
#pragma unroll
  for (int dy = 0; dy < KH; ++dy) wreg[dy] = (pix_ok && l16 < KW) ? w[pix * K + dy * KW + l16] : 0.f;
  stage_halo<T, HH, HW, KH / 2, KW / 2, FULLC>(x, halo, t, H, W, C, grp, l16);
  __syncthreads();
  const T *win = halo + (size_t)(py * HW + px) * C + l16 * kChPerLane;
  const int row_base = (tid & 63) & ~15;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
  for (int dy = 0; dy < KH; ++dy) {
#pragma unroll 1
    for (int dx = 0; dx < KW; ++dx) {
      const float wk = bcast16(wreg[dy], dx, row_base);
      if (ch_ok) {
        float f[8];
        unpack8(ld8(win + (size_t)(dy * HW + dx) * C), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(wk, f[i], acc[i]);
      }
    }
  }
  if (pix_ok && ch_ok) st8(y + pix * C + l16 * kChPerLane, pack8f(acc, T()));
}

// ---------------------------------------------------------------------------------
// ck2c_loc (kernels.cuh:82-119): y[p,:] = sum_k x[p-off_k,:] * w[p-off_k, k]
// (the source pixel s = p - off_k must lie inside the image).  The halo of x covers the
// mirrored window; the weight is read per source texel (uniform over the 16 lanes).
// ---------------------------------------------------------------------------------
template <typename T, int KH, int KW, int MINW, bool FULLC>
__global__ __launch_bounds__(kThreads, MINW) void ck2c_loc_kernel(const T *__restrict__ x,
                                                                  const float *__restrict__ w,
                                                                  T *__restrict__ y, int n, int H,
                                                                  int W, int C, int tiles_x,
                                                                  int tiles_y) {
  using Cfg = LaCfg<KH, KW>;
  constexpr int HW = Cfg::HW, HH = Cfg::HH, K = Cfg::K;
  constexpr int RH = KH / 2, RW = KW / 2;
  extern __shared__ __align__(16) unsigned char smem[];
  T *halo = reinterpret_cast<T *>(smem);
  const Tile t = tile_of_block(n, tiles_x, tiles_y);
  const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
  const int px = grp & (kTW - 1), py = grp / kTW;
  const int gx = t.x0 + px, gy = t.y0 + py;
  const bool ch_ok = FULLC || l16 * kChPerLane < C;  // FULLC: C == 128, every lane owns channels
  const bool pix_ok = gx < W && gy < H;
  stage_halo<T, HH, HW, RH, RW, FULLC>(x, halo, t, H, W, C, grp, l16);
  __syncthreads();
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll 1
  for (int kk = 0; kk < K; ++kk) {
    const int dy = kk / KW - RH, dx = kk % KW - RW;
    const int sy = gy - dy, sx = gx - dx;  // source pixel
    float wk = 0.f;
    if (pix_ok && sy >= 0 && sy < H && sx >= 0 && sx < W)
      wk = w[((size_t)(t.img * H + sy) * W + sx) * K + kk];
    if (ch_ok) {
      float f[8];
      unpack8(ld8(halo + (size_t)((py + RH - dy) * HW + (px + RW - dx)) * C + l16 * kChPerLane), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(wk, f[i], acc[i]);
    }
  }
  if (pix_ok && ch_ok)
    st8(y + ((size_t)(t.img * H + gy) * W + gx) * C + l16 * kChPerLane, pack8f(acc, T()));
}

// ---------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------
enum LaOp { OP_FUSED, OP_CC2K, OP_CK2C_ORI, OP_CK2C_LOC };

struct LaArgs {
  const void *a, *b, *c;
  void *out;
  int n, H, W, C;
  float scale;
  hipStream_t stream;
};

template <typename K>
static hipError_t allow_lds(K kern, size_t lds) {
  if (lds <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <typename T, int KH, int KW, int MINW, bool FULLC>
static int launch_la(LaOp op, const LaArgs &A) {
  using Cfg = LaCfg<KH, KW>;
  const int tiles_x = (A.W + kTW - 1) / kTW, tiles_y = (A.H + kTH - 1) / kTH;
  const size_t lds = (size_t)Cfg::HH * Cfg::HW * A.C * sizeof(T);
  if (lds > 160 * 1024) {
    set_error("local attention tile needs %zu B of LDS", lds);
    return DI_ERR_LDS;
  }
  const dim3 grid(A.n * tiles_x * tiles_y), block(kThreads);
  hipError_t e = hipSuccess;
  switch (op) {
    case OP_FUSED: {
      auto kern = local_attn_fwd_kernel<T, KH, KW, MINW, FULLC>;
      if ((e = allow_lds(kern, lds)) != hipSuccess) break;
      hipLaunchKernelGGL(kern, grid, block, lds, A.stream, (const T *)A.a, (const T *)A.b,
                         (const T *)A.c, (T *)A.out, A.n, A.H, A.W, A.C, A.scale, tiles_x, tiles_y);
      break;
    }
    case OP_CC2K: {
      auto kern = cc2k_kernel<T, KH, KW, MINW, FULLC>;
      if ((e = allow_lds(kern, lds)) != hipSuccess) break;
      hipLaunchKernelGGL(kern, grid, block, lds, A.stream, (const T *)A.a, (const T *)A.b,
                         (float *)A.out, A.n, A.H, A.W, A.C, tiles_x, tiles_y);
      break;
    }
    case OP_CK2C_ORI: {
      auto kern = ck2c_ori_kernel<T, KH, KW, MINW, FULLC>;
      if ((e = allow_lds(kern, lds)) != hipSuccess) break;
      hipLaunchKernelGGL(kern, grid, block, lds, A.stream, (const T *)A.a, (const float *)A.b,
                         (T *)A.out, A.n, A.H, A.W, A.C, tiles_x, tiles_y);
      break;
    }
    case OP_CK2C_LOC: {
      auto kern = ck2c_loc_kernel<T, KH, KW, MINW, FULLC>;
      if ((e = allow_lds(kern, lds)) != hipSuccess) break;
      hipLaunchKernelGGL(kern, grid, block, lds, A.stream, (const T *)A.a, (const float *)A.b,
                         (T *)A.out, A.n, A.H, A.W, A.C, tiles_x, tiles_y);
      break;
    }
  }
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
    return DI_ERR_LAUNCH;
  }
  return check_launch("local attention");
}

template <typename T, int MINW>
static int dispatch_window(LaOp op, int kH, int kW, const LaArgs &A) {
#define DI_WIN(h, w)                                            \
  if (kH == h && kW == w)                                       \
    return A.C == 128 ? launch_la<T, h, w, MINW, true>(op, A)   \
                      : launch_la<T, h, w, MINW, false>(op, A);
  DI_WIN(9, 9) DI_WIN(7, 7) DI_WIN(5, 5) DI_WIN(3, 3) DI_WIN(3, 5) DI_WIN(5, 3)
#undef DI_WIN
  set_error("unsupported window %dx%d (supported: 3x3 5x5 7x7 9x9 3x5 5x3)", kH, kW);
  return DI_ERR_ARG;
}

int launch_local_attn_mfma(const void *q, const void *k, const void *v, void *out, int n, int H, int W,
                           float scale, hipStream_t stream);   // local_attn_mfma.hip
int launch_local_attn_mfma2(const void *q, const void *k, const void *v, void *out, int n, int H, int W,
                            float scale, int cfg, hipStream_t stream);   // local_attn_mfma2.hip
int launch_local_attn_ring(const void *q, const void *k, const void *v, void *out, int n, int H, int W,
                           float scale, int cfg, hipStream_t stream);    // local_attn_ring.hip
int ring_timeouts(unsigned *host_out, hipStream_t stream);
int ring_timeouts_async(unsigned *host_out, hipStream_t stream);
int ring_stamps(unsigned long long *host_out, hipStream_t stream);

static int run_la(LaOp op, int dtype, int kH, int kW, const LaArgs &A) {
  DI_REQUIRE(A.n > 0 && A.H > 0 && A.W > 0, "empty feature map n=%d H=%d W=%d", A.n, A.H, A.W);
  DI_REQUIRE(A.C > 0 && A.C % 8 == 0 && A.C <= 128, "C=%d must be a multiple of 8, <= 128", A.C);
  // 9x9 halo of a 16x4 tile: fp16 24x12x256 B = 72 KiB -> two 16-wave workgroups per CU
  // (8 waves/SIMD, <= 64 VGPRs); fp32 144 KiB -> one workgroup per CU (4 waves/SIMD).
  if (dtype == DI_F16) return dispatch_window<__half, 8>(op, kH, kW, A);
  if (dtype == DI_F32) return dispatch_window<float, 4>(op, kH, kW, A);
  set_error("unsupported dtype %d", dtype);
  return DI_ERR_ARG;
}

}  // namespace di

extern "C" {

int di_local_attn_fwd_ex(const void *q, const void *k, const void *v, void *out, int n, int H, int W,
                         int C, int kH, int kW, float scale, int dtype, int variant, void *stream) {
  di::LaArgs A{q, k, v, out, n, H, W, C, scale, (hipStream_t)stream};
  const bool mfma_ok = dtype == DI_F16 && C == 128 && kH == 9 && kW == 9 && n > 0 && H > 0 && W > 0 &&
                       (long long)n * H * W * 256 < (1ll << 31);   // 32-bit byte offsets inside the kernel
  if (mfma_ok && variant == DI_LA_AUTO) {
    static const int auto_env = getenv("DI_LA_AUTO") ? atoi(getenv("DI_LA_AUTO")) : 0;   // measurement: what AUTO launches
    if (auto_env > 0) variant = auto_env;
  }
  if (variant >= DI_LA_MFMA && variant < DI_LA_MFMA + 5) {
    if (!mfma_ok) {
      di::set_error("MFMA local attention needs fp16, C=128, 9x9, < 2^23 pixels (got dtype=%d C=%d %dx%d)", dtype, C, kH, kW);
      return DI_ERR_ARG;
    }
    return di::launch_local_attn_mfma2(q, k, v, out, n, H, W, scale, variant - DI_LA_MFMA, (hipStream_t)stream);
  }
  if (variant >= DI_LA_RING && variant < DI_LA_RING + 12) {
    if (!mfma_ok) {
      di::set_error("MFMA local attention needs fp16, C=128, 9x9, < 2^23 pixels (got dtype=%d C=%d %dx%d)", dtype, C, kH, kW);
      return DI_ERR_ARG;
    }
    return di::launch_local_attn_ring(q, k, v, out, n, H, W, scale, variant - DI_LA_RING, (hipStream_t)stream);
  }
  if (mfma_ok && variant == DI_LA_AUTO) {
    // Round 4, cold inputs, 6 x 112 x 200: the ring generation 37.1-38.8 us against 40.0-40.5 for the register-staged second
    // generation (8 x 8 tiles, 2 workgroups per CU); inside the forward both take ~40 us per image-side launch, the ring form
    // gives the better step (953.6 against 947.9 samples/s, one sample at a time 1.415 against 1.452 ms).  Maps with fewer
    // than two 16 x 8 tiles per CU (the 180 x 180 BEV map: 276 tiles) stay on the second generation (16.0 against 16.3 us).
    const long long ring_tiles = (long long)n * ((W + 15) / 16) * ((H + 7) / 8);
    const int cus = di::device_cus();                        // (the ring form runs one workgroup per CU)
    if (cus > 0 && ring_tiles >= 2ll * cus) return di::launch_local_attn_ring(q, k, v, out, n, H, W, scale, 0, (hipStream_t)stream);
    return di::launch_local_attn_mfma2(q, k, v, out, n, H, W, scale, 1, (hipStream_t)stream);
  }
  return di::run_la(di::OP_FUSED, dtype, kH, kW, A);
}

int di_local_attn_ring_stamps(void *host_out, void *stream) {
  return di::ring_stamps((unsigned long long *)host_out, (hipStream_t)stream);
}

int di_local_attn_ring_timeouts_async(void *host_word, void *stream) {
  if (host_word == nullptr) {
    di::set_error("null host word");
    return DI_ERR_ARG;
  }
  return di::ring_timeouts_async((unsigned *)host_word, (hipStream_t)stream);
}

int di_local_attn_ring_timeouts(void *stream) {
  unsigned n = 0;
  if (di::ring_timeouts(&n, (hipStream_t)stream) != DI_OK) return -1;
  return (int)n;
}

int di_local_attn_fwd(const void *q, const void *k, const void *v, void *out, int n, int H, int W,
                      int C, int kH, int kW, float scale, int dtype, void *stream) {
  return di_local_attn_fwd_ex(q, k, v, out, n, H, W, C, kH, kW, scale, dtype, DI_LA_AUTO, stream);
}

int di_locatt_similar_fwd(const void *x_ori, const void *x_loc, float *out_w, int n, int H, int W,
                          int C, int kH, int kW, int dtype, void *stream) {
  di::LaArgs A{x_ori, x_loc, nullptr, out_w, n, H, W, C, 1.f, (hipStream_t)stream};
  return di::run_la(di::OP_CC2K, dtype, kH, kW, A);
}

// similar.cu:62-89: is_ori -> ck2c_ori(x, grad) ; else ck2c_loc(x, grad)
int di_locatt_similar_bwd(const void *x, const float *grad_w, void *grad_in, int n, int H, int W,
                          int C, int kH, int kW, int is_ori, int dtype, void *stream) {
  di::LaArgs A{x, grad_w, nullptr, grad_in, n, H, W, C, 1.f, (hipStream_t)stream};
  return di::run_la(is_ori ? di::OP_CK2C_ORI : di::OP_CK2C_LOC, dtype, kH, kW, A);
}

int di_locatt_weighting_fwd(const void *x_ori, const float *x_weight, void *out, int n, int H, int W,
                            int C, int kH, int kW, int dtype, void *stream) {
  di::LaArgs A{x_ori, x_weight, nullptr, out, n, H, W, C, 1.f, (hipStream_t)stream};
  return di::run_la(di::OP_CK2C_ORI, dtype, kH, kW, A);
}

// weighting.cu:64-78: ck2c_loc(grad_out, x_weight)
int di_locatt_weighting_bwd_ori(const float *x_weight, const void *grad_out, void *grad_ori, int n,
                                int H, int W, int C, int kH, int kW, int dtype, void *stream) {
  di::LaArgs A{grad_out, x_weight, nullptr, grad_ori, n, H, W, C, 1.f, (hipStream_t)stream};
  return di::run_la(di::OP_CK2C_LOC, dtype, kH, kW, A);
}

// weighting.cu:105-119: cc2k(grad_out, x_ori)
int di_locatt_weighting_bwd_weight(const void *x_ori, const void *grad_out, float *grad_w, int n,
                                   int H, int W, int C, int kH, int kW, int dtype, void *stream) {
  di::LaArgs A{grad_out, x_ori, nullptr, grad_w, n, H, W, C, 1.f, (hipStream_t)stream};
  return di::run_la(di::OP_CC2K, dtype, kH, kW, A);
}

}  // extern "C"
