// Fused 9x9 local-window attention on the gfx950 matrix cores, third generation (fp16 maps, C = 128):
// the second generation's row-pair MFMA tiles (local_attn_mfma2.hip) with the halo staged by LDS-DMA.
//
// What the second generation measured (tools/la_floor.py, tools/debug/la_grid_sweep.sh): the time of a launch is
// (rounds of tiles per resident workgroup) x (8.5-9 us per tile), and the tile time does NOT move with the number of
// resident workgroups (352 .. 512): the launch is bound by the latency of one tile's dependent phases, so its
// throughput is the number of tiles in flight.  That number was capped at 2 workgroups per CU by the 64 staging
// registers per lane (250 VGPRs) and 64 KB of LDS per workgroup.  Here
//   * the halo travels L2 -> LDS with global_load_lds_dwordx4 (1 KB per wave instruction = one halo row of a 32-channel
//     unit): no staging registers, no address arithmetic per chunk, no commit pass through the VGPR -> LDS path;
//     texels outside the map read a 64-byte zero line instead (one select per instruction);
//   * units are 16 KB: the tile's 64 queries, then 32 channels of the 16 x 16 K halo (4 units), then V likewise: 9 units
//     through a ring of THREE LDS buffers; the DMA of unit X + 2 leaves at the start of pass X, right after the one
//     barrier per unit (a counted wait: only unit X has to have landed);
//   * <= 168 VGPRs and 48 KB of LDS: three workgroups per CU (2 100 tiles over 768 slots = 2.7 rounds against 4.1 rounds
//     over 512, and the 529 tiles of a 180 x 180 BEV map run in ONE round).
// The LDS image is the second generation's (texel slices XOR-swizzled in 32-B segments): the DMA writes lane l at
// base + 16 l, so the swizzle is applied to the global address each lane fetches.
#include <stdlib.h>
#include <type_traits>

#include "di_common.h"

namespace di {
namespace m3 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int WY = 4, NT = WY * 64;                  // 4 waves: 8 x 2 queries each
constexpr int TW = 8, TH = 2 * WY;                   // tile of query pixels
constexpr int HC = TW + 8, HR = TH + 8;              // halo columns / rows
constexpr int CU = 32, S = CU * 2;                   // channels of a unit, bytes of a texel slice
constexpr int ROWB = HC * S, UNITB = HR * ROWB;      // 1 KB per halo row = one DMA instruction; 16 KB per unit
constexpr int NU = 128 / CU;                         // units per operand
static_assert(ROWB == 1024 && HR == 4 * WY, "one DMA instruction per halo row, four rows per wave");

__device__ __attribute__((aligned(64))) unsigned int zero_line[16];   // what a texel outside the map reads

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// 16-B chunk `c16` of the 64-B slice of halo column `hc` sits at this byte offset of the slice
__device__ __forceinline__ int swz(int hc, int c16) { return (((c16 >> 1) ^ ((hc >> 2) & 1)) << 5) | ((c16 & 1) << 4); }

struct TileCoord {
  int img, y0, x0;
};
__device__ __forceinline__ TileCoord decode_tile(int tile, int tiles_x, int per_img) {
  TileCoord t;
  t.img = tile / per_img;
  const int r = tile - t.img * per_img;
  const int ty = r / tiles_x;
  t.y0 = ty * TH;
  t.x0 = (r - ty * tiles_x) * TW;
  return t;
}

// One LDS-DMA instruction: lane l moves 16 B from its own global address to LDS byte lds_addr + 16 l.  Inline assembly on
// purpose: the compiler's wait-count insertion puts s_waitcnt vmcnt(0) - every DMA in flight, the one just issued
// included - in front of every LDS read that follows a DMA it knows about (its alias test between the two buffers does
// not survive the transposing reads); the kernel orders DMA against reads itself, with one vmcnt(0) per unit barrier.
__device__ __forceinline__ void dma16(const void *gp, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gp), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr_of(const void *p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}

// Start of pass X: this wave's LDS reads are done and everything it sent to memory has landed EXCEPT the four youngest
// requests - the DMA pieces of unit X + 1, always the last vector-memory instructions of a pass - then the barrier: unit X
// is complete in LDS for every wave, and the buffer of unit X - 1 = unit X + 2 is free.
#define DI_UNIT_BARRIER() asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory")

// A tile is 9 units of 16 KB: Q (64 queries x 256 B), K channels 0-31 .. 96-127 of the 16 x 16 halo, then V likewise.
// Unit X lives in buffer X % 3; its DMA leaves two passes ahead.
constexpr int NUNIT = 1 + 2 * NU, NBUF = 3;
static_assert(NUNIT % NBUF == 0, "the buffer of a unit must not depend on the tile");

template <int WPS>
__global__ __launch_bounds__(NT, WPS) void local_attn_m3_kernel(
    const __half *__restrict__ q, const __half *__restrict__ k, const __half *__restrict__ v,
    __half *__restrict__ out, int n, int H, int W, float scale, int tiles_x, int tiles_y) {
  __shared__ __align__(1024) unsigned char bufs[NBUF * UNITB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wy = wave;
  const int i = lane & 15, g = lane >> 4;
  const int j = i & 7, qrow = i >> 3;
  const unsigned lds0 = lds_addr_of(bufs);

  // ---- DMA constants of this lane.  K / V unit: LDS position lane * 16 of a halo row = texel lane / 4, slice position
  // lane % 4.  Q unit: a piece is 4 queries x 256 B; position p of query t holds its chunk p ^ (t & 15)
  const int d_hc = lane >> 2, d_pos = lane & 3;
  const int d_c16 = (((d_pos >> 1) ^ ((d_hc >> 2) & 1)) << 1) | (d_pos & 1);   // the chunk that belongs there
  const int d_off = d_hc * 256 + d_c16 * 16;
  const unsigned char *const zsrc = reinterpret_cast<const unsigned char *>(zero_line) + d_pos * 16;
  // ---- fragment constants
  const int koff = wy * 2 * ROWB + i * S + swz(i, g);       // K fragment: key column i, channels 8 g .. 8 g + 7 of the unit
  const int kcv = 4 * g + (i >> 2);                         // V^T fragment: key column addressed by this lane
  const int vsw = (kcv >> 2) & 1;
  const int vbase = wy * 2 * ROWB + kcv * S + (i & 3) * 8;
  // additive softmax masks: 0 where key c = 4g + r lies in the band of query column j (j <= c <= j + 8)
  // and the key row belongs to the window of the query's row, -inf elsewhere
  const float cs = scale * 1.44269504088896f;               // scores in log2 units
  f4 nm_mid, nm_first, nm_last;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool in_band = 4 * g + r >= j && 4 * g + r <= j + 8;
    nm_mid[r] = in_band ? 0.f : -INFINITY;
    nm_first[r] = (in_band && qrow == 0) ? 0.f : -INFINITY;   // key row 0: only the upper query row
    nm_last[r] = (in_band && qrow == 1) ? 0.f : -INFINITY;    // key row 9: only the lower query row
  }

  // ---- tiles: XCD x (workgroups with blockIdx % 8 == x share an L2) owns the contiguous range
  // [T*x/8, T*(x+1)/8) and walks it `gxw` tiles per round
  const int per_img = tiles_x * tiles_y, ntiles = n * per_img;
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
  const int gxw = ((int)gridDim.x - xcd + 7) >> 3;          // workgroups of this XCD
  const int t_end = (int)(((long long)ntiles * (xcd + 1)) >> 3);
  int tile = (int)(((long long)ntiles * xcd) >> 3) + wl;
  if (tile >= t_end) return;

  // rows 4 wave .. 4 wave + 3 of the halo unit (operand `src`, channels cu0 .. cu0 + 31) of tile t -> LDS byte `dst`
  auto dma_halo = [&](const __half *__restrict__ src, const TileCoord &t, int cu0, unsigned dst) {
    const long long tile_off = ((long long)(t.img * H + t.y0 - 4) * W + (t.x0 - 4)) * 256 + cu0 * 2;
    const unsigned char *base = reinterpret_cast<const unsigned char *>(src) + tile_off;
    const int gx = t.x0 - 4 + d_hc;
    const bool x_ok = gx >= 0 && gx < W;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int hr = wave * 4 + r, gy = t.y0 - 4 + hr;
      const bool ok = x_ok && gy >= 0 && gy < H;
      const unsigned char *p = ok ? base + (long long)hr * W * 256 + d_off : zsrc;
      dma16(p, __builtin_amdgcn_readfirstlane(dst + hr * ROWB));
    }
  };
  // the Q unit: piece 4 wave + r = queries (row 2 wave + r / 2, columns 4 (r % 2) .. + 3); queries beyond the map edge
  // (ragged tiles) read a clamped texel, their results are never stored
  auto dma_q = [&](const TileCoord &t, unsigned dst) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tq = (2 * wave + (r >> 1)) * 8 + 4 * (r & 1) + (lane >> 4);      // query of the tile, 0 .. 63
      const int gy = min(t.y0 + (tq >> 3), H - 1), gx = min(t.x0 + (tq & 7), W - 1);
      const unsigned char *p = reinterpret_cast<const unsigned char *>(q) + ((long long)((t.img * H + gy) * W + gx) << 8) +
                               (((lane & 15) ^ (tq & 15)) << 4);
      dma16(p, __builtin_amdgcn_readfirstlane(dst + (4 * wave + r) * 1024));
    }
  };
  // unit X (0 .. NUNIT + 1: the last two belong to the next tile) -> its buffer
  auto dma_unit = [&](auto xc, const TileCoord &cur, const TileCoord &nxt) {
    constexpr int X = decltype(xc)::value;
    constexpr int x = X % NUNIT;
    const TileCoord &t = X < NUNIT ? cur : nxt;
    const unsigned dst = lds0 + (x % NBUF) * UNITB;
    if constexpr (x == 0) dma_q(t, dst);
    else if constexpr (x <= NU) dma_halo(k, t, (x - 1) * CU, dst);
    else dma_halo(v, t, (x - 1 - NU) * CU, dst);
  };
  // finished output channels of a V unit wait here and are stored at the start of the next pass
  h4 pend[2];
  __half *pend_dst = nullptr;
  bool pend_ok = false;
  auto st_pend = [&]() {
    if (pend_ok) {
      *reinterpret_cast<h4 *>(pend_dst) = pend[0];
      *reinterpret_cast<h4 *>(pend_dst + 16) = pend[1];
    }
  };

  TileCoord cur = decode_tile(tile, tiles_x, per_img);
  dma_unit(std::integral_constant<int, 0>{}, cur, cur);
  dma_unit(std::integral_constant<int, 1>{}, cur, cur);

  for (;;) {
    // past the last tile the pipeline re-reads this tile's first units (nobody reads them): uniform request counts
    const bool has_next = tile + gxw < t_end;
    TileCoord nxt = cur;
    if (has_next) nxt = decode_tile(tile + gxw, tiles_x, per_img);

    // ---------------- unit 0: the Q^T fragments (query i, channels kk*32 + 8g .. +7)
    h8 qf[4];
    {
      DI_UNIT_BARRIER();
      st_pend();                                             // the last V unit of the previous tile
      dma_unit(std::integral_constant<int, 2>{}, cur, nxt);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char *buf = bufs + ((2 * wy + qrow) * 8 + j) * 256;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        qf[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(buf + (((kk * 4 + g) ^ i) << 4)));
    }

    // ---------------- S^T = K . Q^T over the K units
    f4 s[10];
#pragma unroll
    for (int rr = 0; rr < 10; ++rr) s[rr] = f4{0.f, 0.f, 0.f, 0.f};
    static_for<0, NU>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      constexpr int X = 1 + u;
      DI_UNIT_BARRIER();
      dma_unit(std::integral_constant<int, X + 2>{}, cur, nxt);
      __builtin_amdgcn_sched_barrier(0);                     // the requests leave at the START of the pass
      const unsigned char *buf = bufs + (X % NBUF) * UNITB;
#pragma unroll
      for (int rr = 0; rr < 10; ++rr) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(buf + koff + rr * ROWB);
        s[rr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, raw), qf[u], s[rr], 0, 0, 0);
      }
    });

    // ---------------- softmax over the 81 window slots of query i, in log2 units: y = s*cs + mask
    h8 pf[5];
    float sum;
    {
      float m = -INFINITY;
#pragma unroll
      for (int rr = 0; rr < 10; ++rr) {
        const f4 nm = rr == 0 ? nm_first : (rr == 9 ? nm_last : nm_mid);
        s[rr] = s[rr] * cs + nm;
        m = fmaxf(m, fmaxf(fmaxf(s[rr][0], s[rr][1]), fmaxf(s[rr][2], s[rr][3])));
      }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      f2 sum2 = {0.f, 0.f};
#pragma unroll
      for (int pr = 0; pr < 5; ++pr) {
        h8 pk;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f4 d = s[2 * pr + t] - m;
          f4 e;
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(d[r]);   // masked slots: exp2(-inf) = 0
          sum2 += f2{e[0], e[1]} + f2{e[2], e[3]};
#pragma unroll
          for (int r = 0; r < 4; ++r) pk[4 * t + r] = (_Float16)e[r];
        }
        pf[pr] = pk;
      }
      sum = sum2[0] + sum2[1];
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
    }

    // ---------------- O^T = V^T . P^T over the V units, each unit finishes CU output channels
    const float inv = 1.f / sum;
    const int gy = cur.y0 + 2 * wy + qrow, gx = cur.x0 + j;
    const bool pix_ok = gy < H && gx < W;
    __half *dst = out + ((long long)(cur.img * H + gy) * W + gx) * 128 + 4 * g;
    static_for<0, NU>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      constexpr int X = 1 + NU + u;                          // unit index in the tile
      DI_UNIT_BARRIER();
      if constexpr (u > 0) st_pend();                        // the previous V unit
      dma_unit(std::integral_constant<int, X + 2>{}, cur, nxt);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char *buf = bufs + (X % NBUF) * UNITB;
      f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int pr = 0; pr < 5; ++pr) {
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) {
          const unsigned char *p0 = buf + vbase + ((nl ^ vsw) << 5) + 2 * pr * ROWB;
          const hv4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(p0));
          const hv4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(p0 + ROWB));
          h8 a;
          a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3];
          a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
          acc[nl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pf[pr], acc[nl], 0, 0, 0);
        }
      }
#pragma unroll
      for (int nl = 0; nl < 2; ++nl) {
        const f4 o = acc[nl] * inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) pend[nl][r] = (_Float16)o[r];
      }
      pend_dst = dst + u * CU;
      pend_ok = pix_ok;
    });
    if (!has_next) break;
    cur = nxt;
    tile += gxw;
  }
  st_pend();                                                 // the last V unit of the last tile
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no DMA may outlive the workgroup's LDS
}
#undef DI_UNIT_BARRIER


// ------------------------------------------------------------------------------------------------------------
// Fourth generation: PRODUCER / CONSUMER wavefronts (the lesson of the 3x3 convolution, conv3x3.hip: a wave that runs
// MFMAs must not issue vector-memory requests - it is blocked while the texture path accepts them - and must not spill).
// A workgroup = 4 consumer waves (the 8 x 8 tile: LDS reads, MFMAs, soft-max; their only memory instructions are the
// output stores) + 2 producer waves that issue every LDS-DMA piece (8 of the 16 pieces of a unit each), three units ahead
// through a ring of FOUR 16 KB buffers (the ring position is a run-time counter: 9 units per tile).  One s_barrier per unit
// for the six waves: the producers arrive after a counted wait (unit U has landed, two younger units may be in flight), the
// consumers after their LDS reads of unit U - 1, whose buffer the producers then refill with unit U + 3.
// ------------------------------------------------------------------------------------------------------------
constexpr int NBUF4 = 4, NPROD4 = 2, NT4 = (WY + NPROD4) * 64;

__global__ __launch_bounds__(NT4, 2) void local_attn_m4_kernel(
    const __half *__restrict__ q, const __half *__restrict__ k, const __half *__restrict__ v,
    __half *__restrict__ out, int n, int H, int W, float scale, int tiles_x, int tiles_y) {
  __shared__ __align__(1024) unsigned char bufs[NBUF4 * UNITB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned lds0 = lds_addr_of(bufs);

  // ---- tiles: XCD x owns the contiguous range [T*x/8, T*(x+1)/8) and walks it `gxw` tiles per round
  const int per_img = tiles_x * tiles_y, ntiles = n * per_img;
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
  const int gxw = ((int)gridDim.x - xcd + 7) >> 3;
  const int t_end = (int)(((long long)ntiles * (xcd + 1)) >> 3);
  int tile = (int)(((long long)ntiles * xcd) >> 3) + wl;
  if (tile >= t_end) return;
  TileCoord cur = decode_tile(tile, tiles_x, per_img);

  if (wave >= WY) {
    // ------------------------------------------------------------------ producer
    const int pw = wave - WY;
    const int d_hc = lane >> 2, d_pos = lane & 3;
    const int d_c16 = (((d_pos >> 1) ^ ((d_hc >> 2) & 1)) << 1) | (d_pos & 1);
    const int d_off = d_hc * 256 + d_c16 * 16;
    const unsigned char *const zsrc = reinterpret_cast<const unsigned char *>(zero_line) + d_pos * 16;
    // halo rows 8 pw .. 8 pw + 7 of a K / V unit (operand `src`, channels cu0 .. cu0 + 31) of tile t -> LDS byte `dst`
    auto dma_halo = [&](const __half *__restrict__ src, const TileCoord &t, int cu0, unsigned dst) {
      const long long tile_off = ((long long)(t.img * H + t.y0 - 4) * W + (t.x0 - 4)) * 256 + cu0 * 2;
      const unsigned char *base = reinterpret_cast<const unsigned char *>(src) + tile_off;
      const int gx = t.x0 - 4 + d_hc;
      const bool x_ok = gx >= 0 && gx < W;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int hr = pw * 8 + r, gy = t.y0 - 4 + hr;
        const bool ok = x_ok && gy >= 0 && gy < H;
        const unsigned char *p = ok ? base + (long long)hr * W * 256 + d_off : zsrc;
        dma16(p, __builtin_amdgcn_readfirstlane(dst + hr * ROWB));
      }
    };
    // pieces 8 pw .. 8 pw + 7 of the Q unit: piece e = queries (row e / 2, columns 4 (e % 2) .. + 3)
    auto dma_q = [&](const TileCoord &t, unsigned dst) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int e = pw * 8 + r;
        const int tq = (e >> 1) * 8 + 4 * (e & 1) + (lane >> 4);
        const int gy = min(t.y0 + (tq >> 3), H - 1), gx = min(t.x0 + (tq & 7), W - 1);
        const unsigned char *p = reinterpret_cast<const unsigned char *>(q) + ((long long)((t.img * H + gy) * W + gx) << 8) +
                                 (((lane & 15) ^ (tq & 15)) << 4);
        dma16(p, __builtin_amdgcn_readfirstlane(dst + e * 1024));
      }
    };
    auto dma_unit = [&](auto xc, const TileCoord &t, unsigned dst) {
      constexpr int x = decltype(xc)::value;
      if constexpr (x == 0) dma_q(t, dst);
      else if constexpr (x <= NU) dma_halo(k, t, (x - 1) * CU, dst);
      else dma_halo(v, t, (x - 1 - NU) * CU, dst);
    };
    // unit U has landed (the 16 pieces of the two younger units may be in flight), then the barrier of unit U
#define DI_PROD_BARRIER() asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory")
    int ring = 0;                                            // buffer of the unit whose barrier comes next
    dma_unit(std::integral_constant<int, 0>{}, cur, lds0);
    dma_unit(std::integral_constant<int, 1>{}, cur, lds0 + UNITB);
    dma_unit(std::integral_constant<int, 2>{}, cur, lds0 + 2 * UNITB);
    for (;;) {
      // past the last tile the pipeline re-reads this tile's first units into free buffers (nobody reads them)
      const bool has_next = tile + gxw < t_end;
      TileCoord nxt = cur;
      if (has_next) nxt = decode_tile(tile + gxw, tiles_x, per_img);
      static_for<0, NUNIT>([&](auto xc) {
        constexpr int x = decltype(xc)::value;
        DI_PROD_BARRIER();                                   // unit x of this tile; buffer (ring - 1) & 3 is free now
        const unsigned dst = lds0 + ((ring + NBUF4 - 1) & (NBUF4 - 1)) * UNITB;
        if constexpr (x + NBUF4 - 1 < NUNIT) dma_unit(std::integral_constant<int, x + NBUF4 - 1>{}, cur, dst);
        else dma_unit(std::integral_constant<int, x + NBUF4 - 1 - NUNIT>{}, nxt, dst);
        ring = (ring + 1) & (NBUF4 - 1);
      });
      if (!has_next) break;
      cur = nxt;
      tile += gxw;
    }
#undef DI_PROD_BARRIER
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no DMA may outlive the workgroup's LDS
    return;
  }

  // ------------------------------------------------------------------ consumer
  const int wy = wave;
  const int i = lane & 15, g = lane >> 4;
  const int j = i & 7, qrow = i >> 3;
  const int koff = wy * 2 * ROWB + i * S + swz(i, g);
  const int kcv = 4 * g + (i >> 2);
  const int vsw = (kcv >> 2) & 1;
  const int vbase = wy * 2 * ROWB + kcv * S + (i & 3) * 8;
  const float cs = scale * 1.44269504088896f;
  f4 nm_mid, nm_first, nm_last;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool in_band = 4 * g + r >= j && 4 * g + r <= j + 8;
    nm_mid[r] = in_band ? 0.f : -INFINITY;
    nm_first[r] = (in_band && qrow == 0) ? 0.f : -INFINITY;
    nm_last[r] = (in_band && qrow == 1) ? 0.f : -INFINITY;
  }
  h4 pend[2];
  __half *pend_dst = nullptr;
  bool pend_ok = false;
  auto st_pend = [&]() {
    if (pend_ok) {
      *reinterpret_cast<h4 *>(pend_dst) = pend[0];
      *reinterpret_cast<h4 *>(pend_dst + 16) = pend[1];
    }
  };
#define DI_CONS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  int ring = 0;
  for (;;) {
    const bool has_next = tile + gxw < t_end;
    // ---------------- unit 0: the Q^T fragments (query i, channels kk*32 + 8g .. +7)
    h8 qf[4];
    {
      DI_CONS_BARRIER();
      st_pend();                                             // the last V unit of the previous tile
      const unsigned char *buf = bufs + ring * UNITB + ((2 * wy + qrow) * 8 + j) * 256;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        qf[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(buf + (((kk * 4 + g) ^ i) << 4)));
      ring = (ring + 1) & (NBUF4 - 1);
    }
    // ---------------- S^T = K . Q^T over the K units
    f4 s[10];
#pragma unroll
    for (int rr = 0; rr < 10; ++rr) s[rr] = f4{0.f, 0.f, 0.f, 0.f};
    static_for<0, NU>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      DI_CONS_BARRIER();
      const unsigned char *buf = bufs + ring * UNITB + koff;
#pragma unroll
      for (int rr = 0; rr < 10; ++rr) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(buf + rr * ROWB);
        s[rr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, raw), qf[u], s[rr], 0, 0, 0);
      }
      ring = (ring + 1) & (NBUF4 - 1);
    });
    // ---------------- softmax over the 81 window slots of query i, in log2 units: y = s*cs + mask
    h8 pf[5];
    float sum;
    {
      float m = -INFINITY;
#pragma unroll
      for (int rr = 0; rr < 10; ++rr) {
        const f4 nm = rr == 0 ? nm_first : (rr == 9 ? nm_last : nm_mid);
        s[rr] = s[rr] * cs + nm;
        m = fmaxf(m, fmaxf(fmaxf(s[rr][0], s[rr][1]), fmaxf(s[rr][2], s[rr][3])));
      }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      f2 sum2 = {0.f, 0.f};
#pragma unroll
      for (int pr = 0; pr < 5; ++pr) {
        h8 pk;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f4 d = s[2 * pr + t] - m;
          f4 e;
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(d[r]);   // masked slots: exp2(-inf) = 0
          sum2 += f2{e[0], e[1]} + f2{e[2], e[3]};
#pragma unroll
          for (int r = 0; r < 4; ++r) pk[4 * t + r] = (_Float16)e[r];
        }
        pf[pr] = pk;
      }
      sum = sum2[0] + sum2[1];
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
    }
    // ---------------- O^T = V^T . P^T over the V units, each unit finishes CU output channels
    const float inv = 1.f / sum;
    const int gy = cur.y0 + 2 * wy + qrow, gx = cur.x0 + j;
    const bool pix_ok = gy < H && gx < W;
    __half *dst = out + ((long long)(cur.img * H + gy) * W + gx) * 128 + 4 * g;
    static_for<0, NU>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      DI_CONS_BARRIER();
      if constexpr (u > 0) st_pend();                        // the previous V unit
      const unsigned char *buf = bufs + ring * UNITB + vbase;
      f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int pr = 0; pr < 5; ++pr) {
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) {
          const unsigned char *p0 = buf + ((nl ^ vsw) << 5) + 2 * pr * ROWB;
          const hv4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(p0));
          const hv4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(p0 + ROWB));
          h8 a;
          a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3];
          a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
          acc[nl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pf[pr], acc[nl], 0, 0, 0);
        }
      }
#pragma unroll
      for (int nl = 0; nl < 2; ++nl) {
        const f4 o = acc[nl] * inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) pend[nl][r] = (_Float16)o[r];
      }
      pend_dst = dst + u * CU;
      pend_ok = pix_ok;
      ring = (ring + 1) & (NBUF4 - 1);
    });
    if (!has_next) break;
    tile += gxw;
    cur = decode_tile(tile, tiles_x, per_img);
  }
#undef DI_CONS_BARRIER
  st_pend();                                                 // the last V unit of the last tile
}

static int launch_m4(const void *q, const void *k, const void *v, void *out, int n, int H, int W, float scale, hipStream_t stream) {
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const long long ntiles = (long long)n * tiles_x * tiles_y;
  DI_REQUIRE((long long)n * H * W * 256 < (1ll << 31), "map of %d x %d x %d texels exceeds the 2 GiB offset range", n, H, W);
  const int n_cu = device_cus();
  if (n_cu <= 0) return DI_ERR_LAUNCH;
  long long grid = (long long)n_cu * 2;
  if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
  static const int grid_env = getenv("DI_LA_GRID") ? atoi(getenv("DI_LA_GRID")) : 0;
  if (grid_env > 0 && grid_env < grid) grid = grid_env / 8 * 8;
  hipLaunchKernelGGL(local_attn_m4_kernel, dim3((unsigned)grid), dim3(NT4), 0, stream, (const __half *)q, (const __half *)k,
                     (const __half *)v, (__half *)out, n, H, W, scale, tiles_x, tiles_y);
  return check_launch("local_attn_m4");
}

template <int WPS>
static int launch(const void *q, const void *k, const void *v, void *out, int n, int H, int W, float scale, hipStream_t stream) {
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const long long ntiles = (long long)n * tiles_x * tiles_y;
  DI_REQUIRE((long long)n * H * W * 256 < (1ll << 31), "map of %d x %d x %d texels exceeds the 2 GiB offset range", n, H, W);
  const int n_cu = device_cus();
  if (n_cu <= 0) return DI_ERR_LAUNCH;
  // one workgroup per resident slot, a multiple of the 8 XCDs; never more than one per tile
  long long grid = (long long)n_cu * WPS;
  if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
  static const int grid_env = getenv("DI_LA_GRID") ? atoi(getenv("DI_LA_GRID")) : 0;   // measurement: workgroups of the launch
  if (grid_env > 0 && grid_env < grid) grid = grid_env / 8 * 8;
  hipLaunchKernelGGL(local_attn_m3_kernel<WPS>, dim3((unsigned)grid), dim3(NT), 0, stream, (const __half *)q,
                     (const __half *)k, (const __half *)v, (__half *)out, n, H, W, scale, tiles_x, tiles_y);
  return check_launch("local_attn_m3");
}

}  // namespace m3

// cfg 0: three workgroups per CU; 1: two (the second generation's occupancy, for A/B measurements); 2: the producer /
// consumer form (fourth generation)
int launch_local_attn_mfma3(const void *q, const void *k, const void *v, void *out, int n, int H, int W, float scale,
                            int cfg, hipStream_t stream) {
  switch (cfg) {
    case 0: return m3::launch<3>(q, k, v, out, n, H, W, scale, stream);
    case 1: return m3::launch<2>(q, k, v, out, n, H, W, scale, stream);
    case 2: return m3::launch_m4(q, k, v, out, n, H, W, scale, stream);      // producer / consumer waves
  }
  set_error("unknown local_attn_mfma3 configuration %d", cfg);
  return DI_ERR_ARG;
}

}  // namespace di
