// 3x3 convolutions of the hot path (stride 1, pad 1) on channels-last fp16 maps, as an implicit GEMM on the gfx950
// matrix cores: the two shared convolutions of the MMRI encoder (reference necks/deepinteraction_encoder.py:45-62,
// 256 -> 128 on the 6x112x200 image maps, 512 -> 128 on the 180x180 BEV map: 79 + 38 GFLOP per sample) and the heat-map
// heads of the decoder (dense_heads/deepinteraction_decoder.py:96-119: 128 -> 128 + BN + ReLU, 128 -> classes).
// Owning them removes the dependence of the step on MIOpen's solver selection / find-db state.
//
//   y[p][n] = act( sum_{tap, c} w[n][tap][c] * x[p + off(tap)][c] + b[n] )          (BatchNorm folded by the caller)
//
// A workgroup (4 waves) owns a tile of TH x 16 output pixels and all output channels.  The input halo tile
// ((TH+2) x 18 pixels) of one 32-channel chunk is staged in LDS once and read nine times - once per tap, as the shifted
// B operand of the 16x16x32 MFMA (lane = pixel, 8 consecutive channels = one ds_read_b128; 16-B slots swizzled so that
// every read is conflict free).  The next chunk is fetched into registers while the current one is multiplied, then
// written to the other LDS buffer: one barrier per chunk.  Weights (A operand, pre-packed (Cout, 9, Cin)) come straight
// from L2, one tap ahead of the MFMAs that use them.  fp32 accumulate, bias / ReLU on the accumulators.
#include "di_common.h"
#include <type_traits>

namespace di {
namespace cv {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int TW = 16, CK = 32;

// LDS image of a halo chunk: pixel P (row-major in the (TH+2) x 18 halo) at P * 64 bytes, its four 16-B channel slots
// rotated: slot g of pixel P lives at ((g + 2 * (P >> 2)) & 3).
__device__ __forceinline__ int lds_off(int P, int g) { return P * 64 + (((g + 2 * (P >> 2)) & 3) << 4); }

// TH: tile rows; WN: waves along the output channels (WM = 4 / WN along the rows); NTW: 16-channel fragments per wave.
template <int TH, int WN, int NTW>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(const __half *__restrict__ x, const __half *__restrict__ w,
                                                         const float *__restrict__ bias, __half *__restrict__ y,
                                                         int H, int W, int Cin, int Cout, int relu, int nchw,
                                                         int tiles_x, int tiles_y) {
  constexpr int WM = 4 / WN, RW = TH / WM;                 // rows of the tile per wave
  constexpr int HP = (TH + 2) * (TW + 2);                  // halo pixels
  constexpr int NLD = (HP * 4 + 255) / 256;                // 16-B pieces per thread and chunk
  __shared__ __align__(16) unsigned char lds[2][HP * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, img = t / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const __half *xi = x + (size_t)img * H * W * Cin;

  // ---- staging: piece e = (halo pixel P, slot s); out-of-image pixels are zeros
  uint4 stage[NLD];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int e = tid + j * 256;
      const int P = e >> 2, s = e & 3;
      const int hy = P / (TW + 2), hx = P - hy * (TW + 2);
      const int yy = y0 + hy - 1, xx = x0 + hx - 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < HP * 4 && yy >= 0 && yy < H && xx >= 0 && xx < W)
        v = *reinterpret_cast<const uint4 *>(xi + ((size_t)yy * W + xx) * Cin + c0 + s * 8);
      stage[j] = v;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int e = tid + j * 256;
      if (e < HP * 4) *reinterpret_cast<uint4 *>(&lds[buf][lds_off(e >> 2, e & 3)]) = stage[j];
    }
  };

  f4 acc[RW][NTW];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[r][n] = f4{0.f, 0.f, 0.f, 0.f};

  const int nrow = wn * NTW * 16 + i;                      // this lane's weight row of fragment 0
  const __half *wl = w + (size_t)nrow * 9 * Cin + g * 8;
  auto load_a = [&](h8 (&a)[NTW], int tap, int c0) {
#pragma unroll
    for (int n = 0; n < NTW; ++n)
      a[n] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(wl + (size_t)n * 16 * 9 * Cin + tap * Cin + c0));
  };

  fetch(0);
  commit(0);
  __syncthreads();
  const int nchunk = Cin / CK;
  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1, c0 = ch * CK;
    if (ch + 1 < nchunk) fetch(c0 + CK);                   // global loads in flight during the MFMAs below
    h8 a[2][NTW];
    load_a(a[0], 0, c0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) load_a(a[(tap + 1) & 1], tap + 1, c0);
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int P = (wm * RW + r + ky) * (TW + 2) + i + kx;
        const h8 b = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(&lds[buf][lds_off(P, g)]));
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[tap & 1][n], b, acc[r][n], 0, 0, 0);
      }
    }
    if (ch + 1 < nchunk) commit(buf ^ 1);                  // the other buffer: its last readers passed the barrier below
    __syncthreads();
  }

  // ---- epilogue: lane holds channels n0 + 4g + r of pixel (row, x0 + i)
  const int xx = x0 + i;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int yy = y0 + wm * RW + r;
    if (yy >= H || xx >= W) continue;
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int n0 = (wn * NTW + n) * 16 + 4 * g;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = acc[r][n][q] + (n0 + q < Cout ? bias[n0 + q] : 0.f);
        if (relu) v[q] = fmaxf(v[q], 0.f);
      }
      if (!nchw) {
        if (n0 + 3 < Cout) {
          h4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (_Float16)v[q];
          *reinterpret_cast<h4 *>(y + (((size_t)img * H + yy) * W + xx) * Cout + n0) = o;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (n0 + q < Cout) y[(((size_t)img * H + yy) * W + xx) * Cout + n0 + q] = __float2half(v[q]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (n0 + q < Cout) {
            const size_t o = (((size_t)img * Cout + n0 + q) * H + yy) * W + xx;
            if (nchw == 2) reinterpret_cast<float *>(y)[o] = v[q];      // float32 logits (heat-map heads)
            else y[o] = __float2half(v[q]);
          }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------
// The class head's last convolution (128 -> classes <= 16; dense_heads/deepinteraction_decoder.py:109-118), 180 x 180: 1.9 GFLOP
// on 8.3 MB - a launch of `conv3x3_kernel<4, 1, 1>` took 17 us (0.5 TB/s): 540 tiles in 2.1 rounds, and per tile FOUR dependent
// round trips to memory (one per 32-channel chunk, each with 9 MFMAs per wave behind it).  Here: 8-row tiles (276: one round
// at two workgroups per CU) and the halo chunks fetched FOUR ahead - for Cin = 128 every load of the tile is issued before
// the first LDS write: 17.2 -> 14.8 us.  Same arithmetic order as conv3x3_kernel.
// ------------------------------------------------------------------------------------------------------------
template <int TH>
__global__ __launch_bounds__(256, 2) void conv3x3_small_kernel(const __half *__restrict__ x, const __half *__restrict__ w,
                                                               const float *__restrict__ bias, __half *__restrict__ y,
                                                               int H, int W, int Cin, int Cout, int relu, int nchw,
                                                               int tiles_x, int tiles_y) {
  constexpr int WM = 4, RW = TH / WM, PF = 4;
  constexpr int HP = (TH + 2) * (TW + 2);
  constexpr int NLD = (HP * 4 + 255) / 256;
  __shared__ __align__(16) unsigned char lds[2][HP * 64];
  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, img = t / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const __half *xi = x + (size_t)img * H * W * Cin;

  // piece e = (halo pixel P, slot s) of this thread: its address (or none) is the same for every chunk
  const __half *src[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int e = tid + j * 256;
    const int P = e >> 2, s4 = e & 3;
    const int hy = P / (TW + 2), hx = P - hy * (TW + 2);
    const int yy = y0 + hy - 1, xx = x0 + hx - 1;
    src[j] = (e < HP * 4 && yy >= 0 && yy < H && xx >= 0 && xx < W) ? xi + ((size_t)yy * W + xx) * Cin + s4 * 8 : nullptr;
  }
  uint4 stage[PF][NLD];
  auto fetch = [&](uint4 (&st)[NLD], int c0) {
#pragma unroll
    for (int j = 0; j < NLD; ++j)
      st[j] = src[j] != nullptr ? *reinterpret_cast<const uint4 *>(src[j] + c0) : make_uint4(0, 0, 0, 0);
  };
  auto commit = [&](const uint4 (&st)[NLD], int buf) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int e = tid + j * 256;
      if (e < HP * 4) *reinterpret_cast<uint4 *>(&lds[buf][lds_off(e >> 2, e & 3)]) = st[j];
    }
  };

  f4 acc[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) acc[r] = f4{0.f, 0.f, 0.f, 0.f};
  const __half *wl = w + (size_t)i * 9 * Cin + g * 8;       // weight row i (rows >= Cout are zero in the packing)

  const int nchunk = Cin / CK;                              // a multiple of PF (host)
#pragma unroll
  for (int q = 0; q < PF; ++q) fetch(stage[q], q * CK);
  commit(stage[0], 0);
  __syncthreads();
  for (int c4 = 0; c4 < nchunk; c4 += PF) {
    // (all 36 weight fragments of the group in flight at once instead of one tap ahead: 15.7 us, no gain - the launch is at its
    // latency floor: one round of 276 workgroups = one memory round trip + ~2 us of products + the stores)
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int ch = c4 + q, buf = q & 1, c0 = ch * CK;      // (PF is even: chunk ch lives in buffer ch & 1)
      // slot q has been committed: refill it with the chunk PF ahead
      if (ch + PF < nchunk) fetch(stage[q], (ch + PF) * CK);
      h8 a[2];
      a[0] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(wl + c0));
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) a[(tap + 1) & 1] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(wl + (tap + 1) * Cin + c0));
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          const int P = (wm * RW + r + ky) * (TW + 2) + i + kx;
          const h8 b = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(&lds[buf][lds_off(P, g)]));
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[tap & 1], b, acc[r], 0, 0, 0);
        }
      }
      if (ch + 1 < nchunk) commit(stage[(q + 1) % PF], buf ^ 1);
      __syncthreads();
    }
  }

  const int xx = x0 + i;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int yy = y0 + wm * RW + r;
    if (yy >= H || xx >= W) continue;
    const int n0 = 4 * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (n0 + q >= Cout) continue;
      float v = acc[r][q] + bias[n0 + q];
      if (relu) v = fmaxf(v, 0.f);
      if (!nchw) {
        y[(((size_t)img * H + yy) * W + xx) * Cout + n0 + q] = __float2half(v);
      } else {
        const size_t o = (((size_t)img * Cout + n0 + q) * H + yy) * W + xx;
        if (nchw == 2) reinterpret_cast<float *>(y)[o] = v;      // float32 logits (heat-map heads)
        else y[o] = __float2half(v);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------
// Second kernel, for Cout == 128: the weights go through LDS as well.  The K loop runs in STAGES = (32-channel chunk,
// kernel row ky): a stage multiplies the three taps of one kernel row.  Its weight tile (3 taps x 128 rows x 32
// channels = 24 KB, contiguous in the (chunk, ky, kx, n, c) packing) and - once per chunk - the next halo chunk are
// fetched into registers TWO stages (one chunk) ahead and written to the other LDS buffers one stage ahead: the L2 /
// HBM latency is covered by a whole stage of MFMAs (48 per wave) of both resident workgroups, and the A fragments
// become conflict-free ds_read_b128 like the B fragments (same slot rotation by row).
// ------------------------------------------------------------------------------------------------------------
template <int TH>
__global__ __launch_bounds__(256, 2) void conv3x3_lds_kernel(const __half *__restrict__ x, const __half *__restrict__ wst,
                                                             const float *__restrict__ bias, __half *__restrict__ y,
                                                             int H, int W, int Cin, int relu, int tiles_x, int tiles_y) {
  constexpr int WN = 2, WM = 2, NTW = 4, RW = TH / WM;
  constexpr int HP = (TH + 2) * (TW + 2);
  constexpr int NLD = (HP * 4 + 255) / 256;                // halo pieces per thread and chunk (2 or 3)
  constexpr int ASZ = 3 * 128 * 64;                        // weight tile of a stage: 6 pieces per thread
  static_assert(NLD <= 3, "halo of at most 3 pieces per thread");
  __shared__ __align__(16) unsigned char lA[2][ASZ];
  __shared__ __align__(16) unsigned char lB[2][HP * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, img = t / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const __half *xi = x + (size_t)img * H * W * Cin;

  // The prefetched tiles live in NAMED registers: arrays that stay live across the back edge of the stage loop are
  // not promoted to registers by hipcc (they went to scratch, and every scratch store waited for its load - the
  // whole prefetch was lost: 2.7k cycles per stage instead of ~1k).
  uint4 a0, a1, a2, a3, a4, a5, b0, b1, b2;
  b0 = b1 = b2 = make_uint4(0, 0, 0, 0);
  // halo piece j of this thread: pixel offset in the image (in elements, -1: outside) and LDS offset
  int bsrc0, bsrc1, bsrc2, bdst0, bdst1, bdst2;
  {
    auto piece = [&](int j, int &src, int &dst) {
      const int e = tid + j * 256;
      const int P = e >> 2, s4 = e & 3;
      const int hy = P / (TW + 2), hx = P - hy * (TW + 2);
      const int yy = y0 + hy - 1, xx = x0 + hx - 1;
      const bool ok = e < HP * 4 && yy >= 0 && yy < H && xx >= 0 && xx < W;
      src = ok ? (yy * W + xx) * Cin + s4 * 8 : -1;
      dst = e < HP * 4 ? lds_off(P, s4) : -1;
    };
    piece(0, bsrc0, bdst0);
    piece(1, bsrc1, bdst1);
    piece(2, bsrc2, bdst2);
  }
  // the staged weights are stored in LDS order already (the slot rotation of lds_off applied by the host packing)
  const int adst0 = tid * 16, adst1 = (tid + 256) * 16, adst2 = (tid + 512) * 16, adst3 = (tid + 768) * 16;
  const int adst4 = (tid + 1024) * 16, adst5 = (tid + 1280) * 16;
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
#define DI_FETCH_A(st)                                                                                                  \
  do {                                                                                                                  \
    const uint4 *src_ = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(wst) + (size_t)(st) * ASZ) + tid; \
    a0 = src_[0]; a1 = src_[256]; a2 = src_[512]; a3 = src_[768]; a4 = src_[1024]; a5 = src_[1280];                     \
  } while (0)
#define DI_COMMIT_A(buf)                                                                                                \
  do {                                                                                                                  \
    unsigned char *d_ = lA[buf];                                                                                        \
    *reinterpret_cast<uint4 *>(d_ + adst0) = a0; *reinterpret_cast<uint4 *>(d_ + adst1) = a1;                           \
    *reinterpret_cast<uint4 *>(d_ + adst2) = a2; *reinterpret_cast<uint4 *>(d_ + adst3) = a3;                           \
    *reinterpret_cast<uint4 *>(d_ + adst4) = a4; *reinterpret_cast<uint4 *>(d_ + adst5) = a5;                           \
  } while (0)
#define DI_FETCH_B(c0)                                                                                                  \
  do {                                                                                                                  \
    b0 = bsrc0 >= 0 ? *reinterpret_cast<const uint4 *>(xi + bsrc0 + (c0)) : zero4;                                      \
    b1 = bsrc1 >= 0 ? *reinterpret_cast<const uint4 *>(xi + bsrc1 + (c0)) : zero4;                                      \
    if (NLD > 2) b2 = bsrc2 >= 0 ? *reinterpret_cast<const uint4 *>(xi + bsrc2 + (c0)) : zero4;                         \
  } while (0)
#define DI_COMMIT_B(buf)                                                                                                \
  do {                                                                                                                  \
    unsigned char *d_ = lB[buf];                                                                                        \
    if (bdst0 >= 0) *reinterpret_cast<uint4 *>(d_ + bdst0) = b0;                                                        \
    if (bdst1 >= 0) *reinterpret_cast<uint4 *>(d_ + bdst1) = b1;                                                        \
    if (NLD > 2 && bdst2 >= 0) *reinterpret_cast<uint4 *>(d_ + bdst2) = b2;                                             \
  } while (0)

  f4 acc[RW][NTW];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[r][n] = f4{0.f, 0.f, 0.f, 0.f};

  // Pipeline: the weight tile of stage st+2 and the halo chunk ch+1 are in flight (in registers) while stage st is
  // multiplied; a tile is written to LDS at the START of the stage before its use - a full stage after its loads were
  // issued - into the buffer whose last readers passed the previous barrier.  The loads are unconditional (the tail
  // re-reads the last tile) so that no control flow surrounds them.
  const int nstage = (Cin / CK) * 3, nchunk = Cin / CK;
  DI_FETCH_B(0);
  DI_FETCH_A(0);
  DI_COMMIT_B(0);
  DI_COMMIT_A(0);
  DI_FETCH_A(nstage > 1 ? 1 : 0);
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int ch = st / 3, ky = st - ch * 3;
    const int ab = st & 1, bb = ch & 1;
    DI_COMMIT_A(ab ^ 1);                                   // stage st+1, fetched one stage ago
    DI_FETCH_A(st + 2 < nstage ? st + 2 : nstage - 1);
    if (ky == 0) DI_FETCH_B((ch + 1 < nchunk ? ch + 1 : ch) * CK);
    if (ky == 2) DI_COMMIT_B(bb ^ 1);                      // two stages after its (HBM) loads were issued
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      h8 a[NTW];
#pragma unroll
      for (int n = 0; n < NTW; ++n)
        a[n] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(&lA[ab][lds_off(kx * 128 + wn * 64 + n * 16 + i, g)]));
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int P = (wm * RW + r + ky) * (TW + 2) + i + kx;
        const h8 b = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(&lB[bb][lds_off(P, g)]));
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[n], b, acc[r][n], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#undef DI_FETCH_A
#undef DI_COMMIT_A
#undef DI_FETCH_B
#undef DI_COMMIT_B

  // epilogue.  The staged weight rows are permuted on the host (row 16nb + 4g + r = output channel 32(nb/2) + 8g +
  // 4(nb%2) + r): a lane's fragment pair is 8 consecutive channels of pixel (row, x0 + i) = one 16-B store.
  const int xx = x0 + i;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int yy = y0 + wm * RW + r;
    if (yy >= H || xx >= W) continue;
#pragma unroll
    for (int p2 = 0; p2 < NTW / 2; ++p2) {
      const int c0 = 32 * ((wn * NTW) / 2 + p2) + 8 * g;
      h8 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v0 = acc[r][2 * p2][q] + bias[c0 + q], v1 = acc[r][2 * p2 + 1][q] + bias[c0 + 4 + q];
        if (relu) {
          v0 = fmaxf(v0, 0.f);
          v1 = fmaxf(v1, 0.f);
        }
        o[q] = (_Float16)v0;
        o[4 + q] = (_Float16)v1;
      }
      *reinterpret_cast<h8 *>(y + (((size_t)img * H + yy) * W + xx) * 128 + c0) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Third kernel, for Cout == 128 and maps of at least 16 rows: 16 x 16-pixel tiles, 8 waves (one workgroup per CU, two
// waves per SIMD as before), and the weight tiles travel L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs,
// asynchronous) into a ring of THREE buffers, two stages ahead of their use.  Against the 8 x 16 kernel: half the
// weight bytes per output pixel through L2 and LDS (a 24 KB tile per stage now serves 256 pixels: weights were
// 590 KB per 128-pixel tile, 6x the input halo), and a prefetch distance that does not cost registers.
// Per stage every wave issues [halo loads of the next chunk (ky == 0 only)], then 3 DMA instructions at the START of the
// stage; they have the stage's 48 MFMAs per wave to land.
// ------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// TH: tile rows, 12, 16 or 20 (the host picks the one with the fewest rounds x rows for the map); NWV: 8 waves (4 x 2: a
// wave owns TH / 4 rows x 64 output channels) or 16 (4 x 4: 32 output channels per wave, 4 waves per SIMD at <= 128 VGPRs)
template <int B_, int E_, class F_>
__device__ __forceinline__ void static_for_cv(F_ &&f) {
  if constexpr (B_ < E_) {
    f(std::integral_constant<int, B_>{});
    static_for_cv<B_ + 1, E_>(f);
  }
}

// PIPE: the fragment reads of a stage are issued by hand, a fixed distance ahead of the MFMAs that use them (see `multiply`)
template <int TH, int NWV, bool PIPE, bool TS = false>   // TS: measurement build, phase timestamps of workgroup 0 into y
__global__ __launch_bounds__(NWV * 64, 1) void conv3x3_dma_kernel(const __half *__restrict__ x, const __half *__restrict__ wst,
                                                             const float *__restrict__ bias, __half *__restrict__ y,
                                                             int H, int W, int Cin, int relu, int tiles_x, int tiles_y) {
  constexpr int NT = NWV * 64, WM = 4, WN = NWV / WM, NTW = 8 / WN, RW = TH / WM;
  static_assert(NWV == 8 || NWV == 16, "8 or 16 waves");
  static_assert(TH % WM == 0, "rows per wave");
  constexpr int HP = (TH + 2) * (TW + 2);                  // halo pixels
  constexpr int NLD = (HP * 4 + NT - 1) / NT;              // halo pieces per thread and chunk: 1 .. 4
  constexpr int ASZ = 3 * 128 * 64;                        // weight tile of a stage: 24 KB = 24 DMA chunks of 1 KB
  // the ring buffers are SEPARATE objects: stage 3 ch + k always uses buffer k, so every access names its buffer at
  // compile time and the compiler's wait-count insertion can tell a DMA into one buffer from reads of another (with
  // one dynamic array it forces vmcnt(0) - every DMA in flight - in front of each stage's fragment reads)
  __shared__ __align__(16) unsigned char lA0[ASZ], lA1[ASZ], lA2[ASZ];
  __shared__ __align__(16) unsigned char lB0[HP * 64], lB1[HP * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, img = t / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const __half *xi = x + (size_t)img * H * W * Cin;

  uint4 b0, b1, b2, b3;                                    // named registers (see conv3x3_lds_kernel)
  b1 = b2 = b3 = make_uint4(0, 0, 0, 0);
  int bsrc0, bsrc1, bsrc2, bsrc3, bdst0, bdst1, bdst2, bdst3;
  {
    auto piece = [&](int j, int &src, int &dst) {
      const int e = tid + j * NT;
      const int P = e >> 2, s4 = e & 3;
      const int hy = P / (TW + 2), hx = P - hy * (TW + 2);
      const int yy = y0 + hy - 1, xx = x0 + hx - 1;
      const bool ok = e < HP * 4 && yy >= 0 && yy < H && xx >= 0 && xx < W;
      src = ok ? (yy * W + xx) * Cin + s4 * 8 : -1;
      dst = e < HP * 4 ? lds_off(P, s4) : -1;
    };
    piece(0, bsrc0, bdst0);
    piece(1, bsrc1, bdst1);
    piece(2, bsrc2, bdst2);
    piece(3, bsrc3, bdst3);
  }
  static_assert(NLD >= 1 && NLD <= 4, "one to four halo pieces per thread");
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  const int nchunk = Cin / CK, nstage = nchunk * 3;
  auto dma_a = [&](int st, unsigned char *buf) {           // the 24 chunks of 1 KB of the tile: 3 (8 waves) or 1-2 per wave
    const int sc = st < nstage ? st : nstage - 1;          // past the end: a re-read (keeps the vmcnt arithmetic uniform)
    const unsigned char *src = reinterpret_cast<const unsigned char *>(wst) + (size_t)sc * ASZ + lane * 16;
#pragma unroll
    for (int j = 0; j < (24 + NWV - 1) / NWV; ++j) {
      const int c = NWV == 8 ? wave * 3 + j : wave + j * NWV;
      if (NWV == 8 || c < 24) __builtin_amdgcn_global_load_lds((gptr_t)(src + c * 1024), (lptr_t)(buf + c * 1024), 16, 0, 0);
    }
  };
#define DI_FETCH_B(c0)                                                                                                  \
  do {                                                                                                                  \
    b0 = bsrc0 >= 0 ? *reinterpret_cast<const uint4 *>(xi + bsrc0 + (c0)) : zero4;                                      \
    if (NLD > 1) b1 = bsrc1 >= 0 ? *reinterpret_cast<const uint4 *>(xi + bsrc1 + (c0)) : zero4;                         \
    if (NLD > 2) b2 = bsrc2 >= 0 ? *reinterpret_cast<const uint4 *>(xi + bsrc2 + (c0)) : zero4;                         \
    if (NLD > 3) b3 = bsrc3 >= 0 ? *reinterpret_cast<const uint4 *>(xi + bsrc3 + (c0)) : zero4;                         \
  } while (0)
#define DI_COMMIT_B(d_)                                                                                                 \
  do {                                                                                                                  \
    if (bdst0 >= 0) *reinterpret_cast<uint4 *>(d_ + bdst0) = b0;                                                        \
    if (NLD > 1 && bdst1 >= 0) *reinterpret_cast<uint4 *>(d_ + bdst1) = b1;                                             \
    if (NLD > 2 && bdst2 >= 0) *reinterpret_cast<uint4 *>(d_ + bdst2) = b2;                                             \
    if (NLD > 3 && bdst3 >= 0) *reinterpret_cast<uint4 *>(d_ + bdst3) = b3;                                             \
  } while (0)

  // the same requests one at a time, for the hand-scheduled stages: piece j of a weight tile, halo register k
  constexpr int NPIECE = (24 + NWV - 1) / NWV;
  auto dma_piece = [&](int st, unsigned char *buf, int j) {
    const int sc = st < nstage ? st : nstage - 1;
    const int c = NWV == 8 ? wave * 3 + j : wave + j * NWV;
    const unsigned char *src = reinterpret_cast<const unsigned char *>(wst) + (size_t)sc * ASZ + lane * 16;
    if (NWV == 8 || c < 24) __builtin_amdgcn_global_load_lds((gptr_t)(src + c * 1024), (lptr_t)(buf + c * 1024), 16, 0, 0);
  };
  auto fetch_one = [&](int k, int c0) {
    if (k == 0) b0 = bsrc0 >= 0 ? *reinterpret_cast<const uint4 *>(xi + bsrc0 + c0) : zero4;
    if (k == 1 && NLD > 1) b1 = bsrc1 >= 0 ? *reinterpret_cast<const uint4 *>(xi + bsrc1 + c0) : zero4;
    if (k == 2 && NLD > 2) b2 = bsrc2 >= 0 ? *reinterpret_cast<const uint4 *>(xi + bsrc2 + c0) : zero4;
    if (k == 3 && NLD > 3) b3 = bsrc3 >= 0 ? *reinterpret_cast<const uint4 *>(xi + bsrc3 + c0) : zero4;
  };
  auto commit_one = [&](int k, unsigned char *d_) {
    if (k == 0 && bdst0 >= 0) *reinterpret_cast<uint4 *>(d_ + bdst0) = b0;
    if (k == 1 && NLD > 1 && bdst1 >= 0) *reinterpret_cast<uint4 *>(d_ + bdst1) = b1;
    if (k == 2 && NLD > 2 && bdst2 >= 0) *reinterpret_cast<uint4 *>(d_ + bdst2) = b2;
    if (k == 3 && NLD > 3 && bdst3 >= 0) *reinterpret_cast<uint4 *>(d_ + bdst3) = b3;
  };

  f4 acc[RW][NTW];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[r][n] = f4{0.f, 0.f, 0.f, 0.f};

  // One stage = 3 RW steps (kx, r); step s multiplies weight set kx (NTW fragments) with the shifted pixel row B(kx, r).
  // Left to the compiler the stage came out as bursts - read 8, wait, 8 MFMAs, read 2, wait, ... (a dozen exposed LDS
  // latencies per stage, both waves of a SIMD in the same phase after the barrier: the matrix pipe idled half the time).
  // Here every read is issued D steps (>= 8 MFMAs = 128 clk) before its use: B fragments through a ring of D + 1
  // registers, the next weight set spread over the current set's steps; sched_barriers pin the order.
  auto multiply_pipe = [&](const unsigned char *A, int ky, const unsigned char *B, auto &&hook) {
    constexpr int NS = 3 * RW, D = NTW >= 4 ? 2 : 4;
    h8 a[2][NTW], b[D + 1];
    auto rd_a = [&](int kx, int n) {
      return __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(A + lds_off(kx * 128 + wn * NTW * 16 + n * 16 + i, g)));
    };
    auto rd_b = [&](int st) {
      const int kx = st / RW, r = st - kx * RW;
      return __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(B + lds_off((wm * RW + r + ky) * (TW + 2) + i + kx, g)));
    };
#pragma unroll
    for (int n = 0; n < NTW; ++n) a[0][n] = rd_a(0, n);
#pragma unroll
    for (int d = 0; d < D; ++d) b[d] = rd_b(d);
    __builtin_amdgcn_sched_barrier(0);
    static_for_cv<0, NS>([&](auto sc) {
      constexpr int st = decltype(sc)::value, kx = st / RW, r = st % RW;
      if constexpr (st + D < NS) b[(st + D) % (D + 1)] = rd_b(st + D);
      if constexpr (kx + 1 < 3) {                          // fragment n of the next weight set leaves at row n * RW / NTW
        static_for_cv<0, NTW>([&](auto nc) {
          constexpr int n = decltype(nc)::value;
          if constexpr (n * RW / NTW == r) a[(kx + 1) & 1][n] = rd_a(kx + 1, n);
        });
      }
      hook(sc);                                            // this step's share of the stage's memory requests
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NTW; ++n)
        acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kx & 1][n], b[st % (D + 1)], acc[r][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto multiply = [&](const unsigned char *A, int ky, const unsigned char *B) {
    if constexpr (PIPE) {
      multiply_pipe(A, ky, B, [](auto) {});
      return;
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      h8 a[NTW];
#pragma unroll
      for (int n = 0; n < NTW; ++n)
        a[n] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(A + lds_off(kx * 128 + wn * NTW * 16 + n * 16 + i, g)));
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int P = (wm * RW + r + ky) * (TW + 2) + i + kx;
        const h8 b = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(B + lds_off(P, g)));
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[n], b, acc[r][n], 0, 0, 0);
      }
    }
  };

  // end of a stage: this wave's DMA requests (the next stage's tile, and the one after it that left at the start of this
  // stage - a whole stage of MFMAs ago) have landed, its LDS writes and reads are done, then the workgroup barrier.
  // A counted wait (vmcnt(3): all but the youngest tile) measured the same, so the plain one stays.  s_barrier, not
  // __syncthreads(): its release fence would be placed by the compiler where it also stalls the fragment reads.
#define DI_STAGE_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
  __shared__ unsigned long long ts_lds[TS ? NWV * 16 : 1];
  int tsi = 0;
  bool ts_on = false;
#define DI_TS()                                                                                                          \
  do {                                                                                                                   \
    if (TS && ts_on && lane == 0 && tsi < 16) ts_lds[wave * 16 + tsi++] = __builtin_readcyclecounter();                  \
  } while (0)
  // prologue: halo chunk 0 and weight tiles 0, 1
  DI_FETCH_B(0);
  dma_a(0, lA0);
  dma_a(1, lA1);
  DI_COMMIT_B(lB0);                                        // (the compiler waits for the halo registers here)
  DI_STAGE_BARRIER();
  // one chunk = stages 3 ch .. 3 ch + 2 = weight buffers 0, 1, 2; halo buffers alternate per chunk (loop unrolled by 2)
  auto chunk = [&](int ch, unsigned char *Bcur, unsigned char *Bnext) {
    const int st = ch * 3;
    ts_on = TS && blockIdx.x == 0 && ch == 3;
    DI_TS();                                               // 0: start of the chunk
    if constexpr (PIPE) {
      // Measured with the phase timestamps of the TS build: issued in a burst at the start of a stage, the 4 halo loads + 3
      // DMA pieces of every wave blocked the waves for 1 500 (waves 0-3) to 3 900 ticks (waves 4-7: the texture path takes
      // the 56 KB of requests of the eight waves in order) before the first MFMA, and the early waves then waited as long at
      // the barrier: a chunk of three stages took 11 800 ticks for 3 x 1 450 ticks of products.  Here every request is
      // one step's hook: the eight waves send 8 KB at a time, under the MFMAs of the other steps.
      constexpr int NS = 3 * RW;
      const int c_next = (ch + 1 < nchunk ? ch + 1 : ch) * CK;
      multiply_pipe(lA0, 0, Bcur, [&](auto sc) {           // ky = 0: next halo chunk (registers), weight tile st + 2
        constexpr int q = decltype(sc)::value;
        if constexpr (q < 2 * NLD && q % 2 == 0) fetch_one(q / 2, c_next);
        if constexpr (q % 3 == 1 && q / 3 < NPIECE) dma_piece(st + 2, lA2, q / 3);
      });
      DI_TS();
      DI_STAGE_BARRIER();
      DI_TS();
      multiply_pipe(lA1, 1, Bcur, [&](auto sc) {           // ky = 1: weight tile st + 3
        constexpr int q = decltype(sc)::value;
        if constexpr (q % 3 == 1 && q / 3 < NPIECE) dma_piece(st + 3, lA0, q / 3);
      });
      DI_TS();
      DI_STAGE_BARRIER();
      DI_TS();
      multiply_pipe(lA2, 2, Bcur, [&](auto sc) {           // ky = 2: the halo registers go to the other buffer, weight tile st + 4
        constexpr int q = decltype(sc)::value;
        if constexpr (q < 2 * NLD && q % 2 == 0) commit_one(q / 2, Bnext);
        if constexpr (q % 3 == 1 && q / 3 < NPIECE) dma_piece(st + 4, lA1, q / 3);
      });
      DI_TS();
      DI_STAGE_BARRIER();
      DI_TS();
      static_assert(NS >= 9, "nine steps hold the requests of a stage");
      return;
    }
    // ky = 0: the next halo chunk starts its trip, then the weight tile two stages ahead
    DI_FETCH_B((ch + 1 < nchunk ? ch + 1 : ch) * CK);
    dma_a(st + 2, lA2);
    __builtin_amdgcn_sched_barrier(0);                     // the requests leave at the START of the stage
    DI_TS();                                               // 1: requests issued
    multiply(lA0, 0, Bcur);
    DI_TS();                                               // 2: products of ky = 0 issued
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DI_TS();                                               // 3: LDS idle
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DI_TS();                                               // 4: DMA / loads landed
    DI_STAGE_BARRIER();
    DI_TS();                                               // 5: past the barrier
    // ky = 1
    dma_a(st + 3, lA0);
    __builtin_amdgcn_sched_barrier(0);
    multiply(lA1, 1, Bcur);
    DI_TS();                                               // 6
    DI_STAGE_BARRIER();
    DI_TS();                                               // 7
    // ky = 2: the halo of the next chunk goes to the other buffer (its last readers passed a barrier a chunk ago)
    DI_COMMIT_B(Bnext);                                    // first: the wait for the halo registers (loaded two stages
    __builtin_amdgcn_sched_barrier(0);                     // ago) must not cover the DMA issued next
    DI_TS();                                               // 8: halo committed
    dma_a(st + 4, lA1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(lA2, 2, Bcur);
    DI_TS();                                               // 9
    DI_STAGE_BARRIER();
    DI_TS();                                               // 10
  };
  for (int ch = 0; ch < nchunk; ch += 2) {
    chunk(ch, lB0, lB1);
    if (ch + 1 < nchunk) chunk(ch + 1, lB1, lB0);
  }
#undef DI_FETCH_B
#undef DI_COMMIT_B
#undef DI_STAGE_BARRIER
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no DMA may outlive the workgroup's LDS
  if (TS && blockIdx.x == 0) {
    __syncthreads();
    if (tid == 0) {
      unsigned long long *dump = reinterpret_cast<unsigned long long *>(y);
      for (int e = 0; e < NWV * 16; ++e) dump[e] = ts_lds[e];
    }
    return;
  }

  // epilogue (row permutation of the staged weights: a lane's fragment pair = 8 consecutive channels = one 16-B store)
  const int xx = x0 + i;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int yy = y0 + wm * RW + r;
    if (yy >= H || xx >= W) continue;
#pragma unroll
    for (int p2 = 0; p2 < NTW / 2; ++p2) {
      const int c0 = 32 * ((wn * NTW) / 2 + p2) + 8 * g;
      h8 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v0 = acc[r][2 * p2][q] + bias[c0 + q], v1 = acc[r][2 * p2 + 1][q] + bias[c0 + 4 + q];
        if (relu) {
          v0 = fmaxf(v0, 0.f);
          v1 = fmaxf(v1, 0.f);
        }
        o[q] = (_Float16)v0;
        o[4 + q] = (_Float16)v1;
      }
      *reinterpret_cast<h8 *>(y + (((size_t)img * H + yy) * W + xx) * 128 + c0) = o;
    }
  }
}


// ------------------------------------------------------------------------------------------------------------
// Fourth kernel: PRODUCER / CONSUMER wavefronts.  What the phase timestamps of the kernel above showed
// (tools/debug/conv_ts.py): a stage's 60 MFMAs per wave take 1 450 ticks, but every wave also issues the stage's memory
// requests, and a wave is BLOCKED while the texture path accepts them - 600 ticks for its 3 weight DMA pieces (L2 hits,
// ~40 B/clk per CU), 1 500 - 3 900 ticks in the stage that also loads the next halo chunk (first-touch lines from HBM /
// Infinity Cache: ~10-15 B/clk per CU, the eight waves served in order) - and the early waves then wait as long at the
// barrier: 11 800 ticks per chunk of three stages for 4 350 ticks of products.  Spreading the requests over the steps of the
// stage did not help (the blocking moves, the total stays).
// Here the eight MFMA wavefronts (consumers) never touch the vector-memory path: four more wavefronts (producers, one
// per SIMD; 12 waves = 3 per SIMD at <= 168 VGPRs) issue everything as LDS-DMA - the weight tiles as before and ALSO the
// input halo (1 KB pieces of 16 pixels x 64 B, the slot rotation of `lds_off` applied to the address each lane fetches,
// pixels outside the map read a zero line) - and only they block.  One s_barrier per stage for all twelve waves.
// Lessons of the bring-up (timestamps of the TS build, tools/debug/conv_ts.py): (i) a SPILLED consumer is a consumer on
// the vector-memory path - four spilled registers reloaded in a stage's first steps queued behind the producers' requests
// and made that stage 2.5x slower: the consumers keep ONE set of weight fragments (re-read right after their last MFMA of
// a set) and compute the pixel-row addresses inside the loop (hoisted they cost 45 registers); (ii) staging the halo
// through the producers' registers instead of LDS-DMA is slower (their LDS writes and the wait in front of them sit in
// front of the chunk's last barrier: 128x180x180 16.7 -> 27.4 us); (iii) what is left: the two consumers of a SIMD
// need ~1 900 ticks for a 12-row stage whose 72 MFMAs take 1 152 (address arithmetic and LDS waits between the MFMAs),
// plus ~250 ticks of barrier per stage.
// ------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(64))) unsigned int cv_zero_line[16];

template <int TH, bool TS = false>   // TS: measurement build, phase timestamps of workgroup 0 into y
__global__ __launch_bounds__(768, 1) void conv3x3_pc_kernel(const __half *__restrict__ x, const __half *__restrict__ wst,
                                                            const float *__restrict__ bias, __half *__restrict__ y, int H, int W,
                                                            int Cin, int relu, int tiles_x, int tiles_y) {
  constexpr int WM = 4, WN = 2, NTW = 4, RW = TH / WM, NPROD = 4;
  static_assert(TH % WM == 0, "rows per wave");
  constexpr int HP = (TH + 2) * (TW + 2);                  // halo pixels
  constexpr int NPH = (HP + 15) / 16;                      // halo pieces of 1 KB (16 pixels x 64 B) per chunk
  constexpr int KH = (NPH + NPROD - 1) / NPROD;            // ... per producer
  constexpr int KH0 = (KH + 1) / 2;                        // issued in the ky = 0 stage (the rest in ky = 1)
  constexpr int ASZ = 3 * 128 * 64;                        // weight tile of a stage: 24 KB = 24 DMA pieces
  constexpr int WPP = 24 / NPROD;                          // weight pieces per producer and stage
  __shared__ __align__(16) unsigned char lA0[ASZ], lA1[ASZ], lA2[ASZ];
  __shared__ __align__(16) unsigned char lB0[NPH * 1024], lB1[NPH * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y, img = t / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int nchunk = Cin / CK, nstage = nchunk * 3;
  __shared__ unsigned long long ts_lds[TS ? 12 * 16 : 1];
  int tsi = 0;
  bool ts_on = false;
#define DI_TS()                                                                                                          \
  do {                                                                                                                   \
    if (TS && ts_on && lane == 0 && tsi < 16) ts_lds[wave * 16 + tsi++] = __builtin_readcyclecounter();                  \
  } while (0)
  auto ts_dump = [&]() {
    if (TS && blockIdx.x == 0) {
      __syncthreads();
      if (tid == 0) {
        unsigned long long *dump = reinterpret_cast<unsigned long long *>(y);
        for (int e = 0; e < 12 * 16; ++e) dump[e] = ts_lds[e];
      }
    }
  };

  if (wave >= 8) {
    // ------------------------------------------------------------------ producer
    const int pw = wave - 8;
    const unsigned char *xi = reinterpret_cast<const unsigned char *>(x + (size_t)img * H * W * Cin);
    const int pos = lane & 3;
    const unsigned char *zsrc = reinterpret_cast<const unsigned char *>(cv_zero_line) + pos * 16;
    // piece q = min(pw + 4 k, NPH - 1) (the clamp repeats the last piece: same data to the same place, uniform counts);
    // lane l of a piece = pixel 16 q + l / 4, slot position l % 4 <- channel slot s4 with lds_off's rotation
    long long hsrc[KH];                                    // byte offset of the lane's 16 B inside the image's map, -1 = zeros
    int hq[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int q = min(pw + NPROD * k, NPH - 1);
      const int P = q * 16 + (lane >> 2);
      const int s4 = (pos - 2 * (P >> 2)) & 3;
      const int hy = P / (TW + 2), hx = P - hy * (TW + 2);
      const int yy = y0 + hy - 1, xx = x0 + hx - 1;
      const bool ok = P < HP && yy >= 0 && yy < H && xx >= 0 && xx < W;
      hsrc[k] = ok ? ((long long)(yy * W + xx) * Cin + s4 * 8) * 2 : -1;
      hq[k] = q;
    }
    auto dma_w = [&](int st, unsigned char *buf) {
      const int sc = st < nstage ? st : nstage - 1;        // past the end: a re-read (uniform counts)
      const unsigned char *src = reinterpret_cast<const unsigned char *>(wst) + (size_t)sc * ASZ + lane * 16;
#pragma unroll
      for (int j = 0; j < WPP; ++j) {
        const int c = pw * WPP + j;
        __builtin_amdgcn_global_load_lds((gptr_t)(src + c * 1024), (lptr_t)(buf + c * 1024), 16, 0, 0);
      }
    };
    auto dma_h = [&](int chunk, unsigned char *buf, int k_lo, int k_hi) {
      const int c0 = (chunk < nchunk ? chunk : nchunk - 1) * CK * 2;
#pragma unroll
      for (int k = 0; k < KH; ++k)
        if (k >= k_lo && k < k_hi) {
          const unsigned char *p = hsrc[k] >= 0 ? xi + hsrc[k] + c0 : zsrc;
          __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(buf + hq[k] * 1024), 16, 0, 0);
        }
    };
#define DI_PROD_BARRIER(N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory")
    dma_h(0, lB0, 0, KH);
    dma_w(0, lA0);
    dma_w(1, lA1);
    DI_PROD_BARRIER(0);
    for (int ch = 0; ch < nchunk; ++ch) {
      unsigned char *Bnext = (ch & 1) ? lB0 : lB1;
      const int st = ch * 3;
      // whatever a stage issues may still be in flight at its end; everything older has landed
      ts_on = TS && blockIdx.x == 0 && ch == 3;
      DI_TS();                                             // 0
      dma_w(st + 2, lA2);
      dma_h(ch + 1, Bnext, 0, KH0);
      DI_TS();                                             // 1: ky0 requests issued
      DI_PROD_BARRIER(WPP + KH0);
      DI_TS();                                             // 3: barrier
      dma_w(st + 3, lA0);
      dma_h(ch + 1, Bnext, KH0, KH);
      DI_TS();                                             // 4
      DI_PROD_BARRIER(WPP + KH - KH0);
      DI_TS();                                             // 5
      dma_w(st + 4, lA1);
      DI_TS();                                             // 6
      DI_PROD_BARRIER(WPP);
      DI_TS();                                             // 7
    }
#undef DI_PROD_BARRIER
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA may outlive the workgroup's LDS
    ts_dump();
    return;
  }

  // ------------------------------------------------------------------ consumer
  const int i = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  f4 acc[RW][NTW];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[r][n] = f4{0.f, 0.f, 0.f, 0.f};
  int brow[8];
  {
    const int q = wm * RW * (TW + 2) + i;
#pragma unroll
    for (int c = 0; c < 8; ++c) brow[c] = q * 64 + (((g + 2 * ((((q & 7) + c) >> 2) & 1)) & 3) << 4);
  }
  // one stage: every fragment read D steps ahead of its MFMAs (see conv3x3_dma_kernel's multiply_pipe)
  auto multiply = [&](const unsigned char *A, int ky, const unsigned char *B) {
    constexpr int NS = 3 * RW, D = 2;
    h8 a[NTW], b[D + 1];
    auto rd_a = [&](int kx, int n) {
      return __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(A + lds_off(kx * 128 + wn * NTW * 16 + n * 16 + i, g)));
    };
    // pixel row of step (kx, r): pixel P = q + ct with q = wm RW 18 + i (this lane) and ct = (r + ky) 18 + kx (compile time);
    // lds_off(P, g) = 64 q + rot + 64 ct, and the slot rotation only needs bit 2 of P = bit 2 of (q & 7) + (ct & 7): one of
    // EIGHT per-lane addresses (brow[ct & 7], set up once) plus an immediate - no address arithmetic between the MFMAs
    // (computed per read it was ~7 VALU instructions x 15 reads per stage; hoisted by the compiler, 45 registers and spills)
    auto rd_b = [&](int st) {
      const int kx = st / RW, r = st - kx * RW;
      const int ct = (r + ky) * (TW + 2) + kx;
      return __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(B + brow[ct & 7] + ct * 64));
    };
#pragma unroll
    for (int n = 0; n < NTW; ++n) a[n] = rd_a(0, n);
#pragma unroll
    for (int d = 0; d < D; ++d) b[d] = rd_b(d);
    __builtin_amdgcn_sched_barrier(0);
    // ONE set of weight fragments (a second set does not fit 168 registers without spilling): in the last row of a weight
    // set every fragment is re-read for the next set right after its last MFMA has issued
    static_for_cv<0, NS>([&](auto sc) {
      constexpr int st = decltype(sc)::value, kx = st / RW, r = st % RW;
      if constexpr (st + D < NS) b[(st + D) % (D + 1)] = rd_b(st + D);
      __builtin_amdgcn_sched_barrier(0);
      static_for_cv<0, NTW>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[n], b[st % (D + 1)], acc[r][n], 0, 0, 0);
        if constexpr (r == RW - 1 && kx + 1 < 3) {
          a[n] = rd_a(kx + 1, n);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };
#define DI_CONS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  DI_CONS_BARRIER();
  for (int ch = 0; ch < nchunk; ++ch) {
    const unsigned char *Bcur = (ch & 1) ? lB1 : lB0;
    ts_on = TS && blockIdx.x == 0 && ch == 3;
    DI_TS();                                               // 0
    multiply(lA0, 0, Bcur);
    DI_TS();                                               // 1: ky0 products issued
    DI_CONS_BARRIER();
    DI_TS();                                               // 3: barrier
    multiply(lA1, 1, Bcur);
    DI_TS();                                               // 4
    DI_CONS_BARRIER();
    DI_TS();                                               // 5
    multiply(lA2, 2, Bcur);
    DI_TS();                                               // 6
    DI_CONS_BARRIER();
    DI_TS();                                               // 7
  }
#undef DI_CONS_BARRIER
  if (TS) {
    ts_dump();
    if (blockIdx.x == 0) return;
  }

  // epilogue (row permutation of the staged weights: a lane's fragment pair = 8 consecutive channels = one 16-B store)
  const int xx = x0 + i;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int yy = y0 + wm * RW + r;
    if (yy >= H || xx >= W) continue;
#pragma unroll
    for (int p2 = 0; p2 < NTW / 2; ++p2) {
      const int c0 = 32 * ((wn * NTW) / 2 + p2) + 8 * g;
      h8 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v0 = acc[r][2 * p2][q] + bias[c0 + q], v1 = acc[r][2 * p2 + 1][q] + bias[c0 + 4 + q];
        if (relu) {
          v0 = fmaxf(v0, 0.f);
          v1 = fmaxf(v1, 0.f);
        }
        o[q] = (_Float16)v0;
        o[4 + q] = (_Float16)v1;
      }
      *reinterpret_cast<h8 *>(y + (((size_t)img * H + yy) * W + xx) * 128 + c0) = o;
    }
  }
}

}  // namespace cv
}  // namespace di

extern "C" int di_conv3x3_fwd(const void *x, const void *w_packed, const void *w_staged, const float *bias, void *y,
                              int n, int H, int W, int Cin, int Cout, int relu, int out_nchw, void *stream) {
  using namespace di::cv;
  DI_REQUIRE(n > 0 && H > 0 && W > 0 && x && w_packed && bias && y, "bad conv shape");
  DI_REQUIRE(Cin % CK == 0, "Cin=%d must be a multiple of %d", Cin, CK);
  DI_REQUIRE(Cout == 128 || Cout <= 16, "Cout=%d: 128 or <= 16 (weights padded to 16 rows)", Cout);
  hipStream_t s = (hipStream_t)stream;
  const int tiles_x = (W + TW - 1) / TW;
#define DI_CV(TH, WN, NTW)                                                                                          \
  do {                                                                                                              \
    const int tiles_y = (H + TH - 1) / TH;                                                                          \
    hipLaunchKernelGGL((conv3x3_kernel<TH, WN, NTW>), dim3(tiles_x * tiles_y * n), dim3(256), 0, s,                 \
                       (const __half *)x, (const __half *)w_packed, bias, (__half *)y, H, W, Cin, Cout, relu,      \
                       out_nchw, tiles_x, tiles_y);                                                                 \
  } while (0)
  if (Cout == 128 && w_staged != nullptr && !out_nchw) {
    // weights through LDS.  8-row tiles whenever the map has at least 8 rows: the kernel is latency bound per
    // stage, so fewer, larger tiles that fit the resident slots in one round beat more, smaller ones
    // (180 x 180: 276 tiles of 8 x 16 in one round instead of 540 of 4 x 16 in two)
    static const int no_dma = getenv("DI_CONV_NO_DMA") ? atoi(getenv("DI_CONV_NO_DMA")) : 0;
    if (H >= 12 && !no_dma) {
      // one workgroup per CU: the tile height with the fewest (rounds of tiles over the CUs) x (rows per tile) -
      // 6 x 112 x 200: 20 rows = 468 tiles = 2 rounds (16 rows: 546 tiles = 3 rounds); 180 x 180: 12 rows = 180 tiles
      const int cus = di::device_cus();
      if (cus <= 0) return DI_ERR_LAUNCH;
      static const int force_th = getenv("DI_CONV_TH") ? atoi(getenv("DI_CONV_TH")) : 0;
      int best = 0;
      long long best_cost = 0;
      for (int th : {12, 16, 20}) {
        if (force_th ? th != force_th : th > H + 3) continue;
        const long long tiles = (long long)tiles_x * ((H + th - 1) / th) * n;
        const long long cost = ((tiles + cus - 1) / cus) * th;
        if (best == 0 || cost < best_cost) best = th, best_cost = cost;
      }
      if (best == 0) best = 12;
      // 16-wave workgroups for the 12- and 16-row tiles (measured against 8 waves: 512x180x180 60.4 -> 53.8 us, 128x180x180
      // 19.5 -> 18.1 on random maps, 32.3 -> 31.3 inside the forward); the 20-row image tile stays at 8 waves (inside the
      // forward 90.4 us against 102.4 with 16, although 111.7 -> 109.0 on random maps).  DI_CONV_W16=0 / 1 forces one form.
      static const int w16_env = getenv("DI_CONV_W16") ? atoi(getenv("DI_CONV_W16")) : -1;
      const int w16 = w16_env >= 0 ? w16_env : (best != 20);
      // the producer / consumer kernel for the 12- and 20-row tiles (measured against conv3x3_dma_kernel on random maps: 6x256x112x200
      // 111.4 -> 90.7 us, 512x180x180 51.4 -> 47.8, 128x180x180 18.4 -> 16.7; the benched forward 885 -> 905 samples/s);
      // DI_CONV_PC=0 selects the older kernel for A/B runs
      static const int pc = getenv("DI_CONV_PC") ? atoi(getenv("DI_CONV_PC")) : 1;
      if (pc && (best == 12 || best == 20)) {
#define DI_PC(TH_)                                                                                                  \
  hipLaunchKernelGGL((conv3x3_pc_kernel<TH_>), dim3(tiles_x * ((H + TH_ - 1) / TH_) * n), dim3(768), 0, s,             \
                     (const __half *)x, (const __half *)w_staged, bias, (__half *)y, H, W, Cin, relu, tiles_x,       \
                     (H + TH_ - 1) / TH_)
        static const int pcts = getenv("DI_CONV_TS") ? atoi(getenv("DI_CONV_TS")) : 0;
        if (pcts && best == 12)
          hipLaunchKernelGGL((conv3x3_pc_kernel<12, true>), dim3(tiles_x * ((H + 11) / 12) * n), dim3(768), 0, s, (const __half *)x,
                             (const __half *)w_staged, bias, (__half *)y, H, W, Cin, relu, tiles_x, (H + 11) / 12);
        else if (pcts && best == 20)
          hipLaunchKernelGGL((conv3x3_pc_kernel<20, true>), dim3(tiles_x * ((H + 19) / 20) * n), dim3(768), 0, s, (const __half *)x,
                             (const __half *)w_staged, bias, (__half *)y, H, W, Cin, relu, tiles_x, (H + 19) / 20);
        else if (best == 12) DI_PC(12); else DI_PC(20);
#undef DI_PC
        return di::check_launch("conv3x3_fwd");
      }
      static const int ts = getenv("DI_CONV_TS") ? atoi(getenv("DI_CONV_TS")) : 0;   // measurement build (20-row tiles, 8 waves)
      if (ts) {
        hipLaunchKernelGGL((conv3x3_dma_kernel<20, 8, true, true>), dim3(tiles_x * ((H + 19) / 20) * n), dim3(512), 0, s,
                           (const __half *)x, (const __half *)w_staged, bias, (__half *)y, H, W, Cin, relu, tiles_x, (H + 19) / 20);
        return di::check_launch("conv3x3_fwd");
      }
      static const int pipe = getenv("DI_CONV_PIPE") ? atoi(getenv("DI_CONV_PIPE")) : 1;   // 0: compiler-scheduled stages (A/B)
#define DI_DMA2(TH_, NWV_, P_)                                                                                    \
  hipLaunchKernelGGL((conv3x3_dma_kernel<TH_, NWV_, P_>), dim3(tiles_x * ((H + TH_ - 1) / TH_) * n), dim3(NWV_ * 64), 0, s, \
                     (const __half *)x, (const __half *)w_staged, bias, (__half *)y, H, W, Cin, relu, tiles_x,     \
                     (H + TH_ - 1) / TH_)
#define DI_DMA(TH_, NWV_) do { if (pipe) DI_DMA2(TH_, NWV_, true); else DI_DMA2(TH_, NWV_, false); } while (0)
      if (w16) {
        if (best == 12) DI_DMA(12, 16); else if (best == 16) DI_DMA(16, 16); else DI_DMA(20, 16);
      } else {
        if (best == 12) DI_DMA(12, 8); else if (best == 16) DI_DMA(16, 8); else DI_DMA(20, 8);
      }
#undef DI_DMA2
#undef DI_DMA
    } else if (H >= 8) {
      const int tiles_y = (H + 7) / 8;
      hipLaunchKernelGGL((conv3x3_lds_kernel<8>), dim3(tiles_x * tiles_y * n), dim3(256), 0, s, (const __half *)x,
                         (const __half *)w_staged, bias, (__half *)y, H, W, Cin, relu, tiles_x, tiles_y);
    } else {
      const int tiles_y = (H + 3) / 4;
      hipLaunchKernelGGL((conv3x3_lds_kernel<4>), dim3(tiles_x * tiles_y * n), dim3(256), 0, s, (const __half *)x,
                         (const __half *)w_staged, bias, (__half *)y, H, W, Cin, relu, tiles_x, tiles_y);
    }
  } else if (Cout == 128) {
    if ((long long)n * ((H + 7) / 8) * tiles_x >= 768) DI_CV(8, 2, 4);
    else DI_CV(4, 2, 4);
  } else if (Cin % (4 * CK) == 0 && H >= 8) {
    static const int small = getenv("DI_CONV_SMALL") ? atoi(getenv("DI_CONV_SMALL")) : 1;   // 0: conv3x3_kernel<4, 1, 1> (A/B)
    if (small) {
      const int tiles_y = (H + 7) / 8;
      hipLaunchKernelGGL((conv3x3_small_kernel<8>), dim3(tiles_x * tiles_y * n), dim3(256), 0, s, (const __half *)x,
                         (const __half *)w_packed, bias, (__half *)y, H, W, Cin, Cout, relu, out_nchw, tiles_x, tiles_y);
    } else {
      DI_CV(4, 1, 1);
    }
  } else {
    DI_CV(4, 1, 1);
  }
#undef DI_CV
  return di::check_launch("conv3x3_fwd");
}
