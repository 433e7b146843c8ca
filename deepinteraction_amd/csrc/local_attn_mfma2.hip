// Fused 9x9 local-window attention on the gfx950 matrix cores, second generation:
// persistent workgroups, software-pipelined staging, row-PAIR MFMA tiles (fp16 maps, C = 128).
//
// What limits the first matrix-core kernel (local_attn_mfma.hip, 0.24 of the HBM roofline) is not
// HBM traffic (1.04x algorithmic, profiles/r01b_pmc_hbm.csv) but the serial
// {stage halo -> barrier -> MFMA} rounds of a one-tile workgroup: every workgroup of the launch
// loads at the same time, then every workgroup computes with nothing in flight.  Here
//
//   * a wavefront owns 8 x 2 query pixels.  Its 16 MFMA columns are the 8 queries of row y and the
//     8 queries of row y+1; one 16-key tile (columns x0-4 .. x0+11 of ONE key row) then serves both
//     query rows, so the band |dx| <= 4 fills 9/16 of a tile instead of 9/32 and a 10-row halo
//     covers both windows: 40 + 40 MFMAs per 16 queries instead of 72 + 72, and the softmax runs
//     over 40 instead of 72 registers;
//   * S^T = K . Q^T  (A = K keys x channels from LDS, B = Q^T from registers), softmax per query =
//     per lane column, O^T = V^T . P^T with the MFMA k index = (row of the pair, key) so the
//     softmax registers ARE the B operand (no cross-lane movement); V^T fragments come from
//     ds_read_b64_tr_b16 (hardware transpose) - as in the first kernel;
//   * the halo is staged in channel UNITS of CU channels (K units then V units, 128/CU each): the
//     S accumulation over channels and the independent output channels make every unit a
//     self-contained {LDS image, MFMA pass}.  Two LDS buffers and two register sets: the global
//     loads of unit s+2 are issued DURING the MFMA pass of unit s - one load every few MFMAs, so
//     the texture path (64 B/clk per CU) drains them in the background instead of stalling the
//     wave - and written to LDS after the pass of unit s+1 (one barrier per unit).  The output
//     stores of a V unit and the next tile's Q fragments ride in the same way inside the next pass;
//   * the workgroup is PERSISTENT: every XCD owns one contiguous range of tiles and walks it round
//     by round, so a round's halo rows are the previous round's rows of the SAME L2;
//   * LDS texel slices are XOR-swizzled in 32-B segments by the key column so that the
//     ds_read_b128 K fragments, the transposed V reads and the ds_write_b128 staging stores are
//     all bank-conflict free for slices of 64, 128 or 256 bytes (SQ_LDS_BANK_CONFLICT = 0).
#include <stdlib.h>
#include <type_traits>

#include <hip/hip_ext.h>

#include "di_common.h"

namespace di {
namespace m2 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

template <int WX_, int WY_, int CU_, int WPS_, int DBG_ = 0>
struct Cfg {
  static constexpr int WX = WX_, WY = WY_, CU = CU_, WPS = WPS_;
  static constexpr int DBG = DBG_;                   // measurement only: 4 = every load from one tile (L2 hits),
                                                     // 6 = per-wave phase timestamps, 7 = both
  static constexpr bool TS = DBG == 6 || DBG == 7, ONE_TILE = DBG == 4 || DBG == 7;
  static constexpr int NW = WX * WY, NT = NW * 64;   // wavefronts / threads per workgroup
  static constexpr int TW = 8 * WX, TH = 2 * WY;     // tile of query pixels
  static constexpr int HC = TW + 8, HR = TH + 8;     // halo columns / rows
  static constexpr int S = CU * 2;                   // bytes of one texel slice
  static constexpr int CPT = S / 16;                 // 16-B chunks per slice
  static constexpr int NSEG = S / 32;                // 32-B segments per slice
  static constexpr int TPR = 256 / S;                // slices per 256-B LDS bank row
  static constexpr int ROWB = HC * S;
  static constexpr int UNITB = HR * ROWB;            // one staged unit
  static constexpr int NCHUNK = HR * HC * CPT;
  static constexpr int NLD = (NCHUNK + NT - 1) / NT; // 16-B loads per lane per unit
  static constexpr int NU = 128 / CU;                // units per operand (K, V)
  static constexpr int KK = CU / 32;                 // S-phase MFMA k-steps per unit
  static constexpr int NN = CU / 16;                 // O-phase 16-channel blocks per unit
  static constexpr int LDS_BYTES = 2 * UNITB;
  static_assert(CU == 32 || CU == 64, "unit = 32 or 64 channels (two units per operand at least)");
};

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// byte offset of 16-B chunk `c16` inside the slice of halo column `hc`
template <class G>
__device__ __forceinline__ int swz(int hc, int c16) {
  const int f = (hc / G::TPR) % G::NSEG;
  return ((((c16 >> 1) ^ f)) << 5) | ((c16 & 1) << 4);
}

__device__ __forceinline__ void lds_barrier() {
  // LDS traffic only: outstanding global loads (the next units, in registers) stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// vector-memory instructions may not be scheduled across this point; everything else may
__device__ __forceinline__ void pin_vmem() { __builtin_amdgcn_sched_barrier(0x0381); }

struct TileCoord {
  int img, y0, x0;
};
__device__ __forceinline__ TileCoord decode_tile(int tile, int tiles_x, int per_img, int TH, int TW) {
  TileCoord t;
  t.img = tile / per_img;
  const int r = tile - t.img * per_img;
  const int ty = r / tiles_x;
  t.y0 = ty * TH;
  t.x0 = (r - ty * tiles_x) * TW;
  return t;
}

template <class G>
__global__ __launch_bounds__(G::NT, G::WPS) void local_attn_m2_kernel(
    const __half *__restrict__ q, const __half *__restrict__ k, const __half *__restrict__ v,
    __half *__restrict__ out, int n, int H, int W, float scale, int tiles_x, int tiles_y) {
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr int NLD = G::NLD, ROWB = G::ROWB, S = G::S, NU = G::NU;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wx = wave % G::WX, wy = wave / G::WX;
  const int i = lane & 15, g = lane >> 4;
  const int j = i & 7, qrow = i >> 3;

  // ---- staging constants of this lane: slot s moves chunk e = s*NT + tid of the unit (the tail of a
  // partial last slot repeats the unit's last chunk: same data to the same address)
  int st_lds[NLD], st_pk[NLD];
  unsigned st_go[NLD];                              // byte offset of the chunk from the halo's first texel
#pragma unroll
  for (int s = 0; s < NLD; ++s) {
    const int e = min(s * G::NT + tid, G::NCHUNK - 1);
    const int tex = e / G::CPT, c16 = e % G::CPT;
    const int hr = tex / G::HC, hc = tex - hr * G::HC;
    st_lds[s] = hr * ROWB + hc * S + swz<G>(hc, c16);
    st_pk[s] = hr | (hc << 8);
    st_go[s] = (unsigned)(hr * W + hc) * 256u + c16 * 16;
  }
  // ---- fragment constants
  const int hcq = wx * 8 + i;                       // K fragment: key c = i of the wave's 16 columns
  int koff[G::KK];
#pragma unroll
  for (int kl = 0; kl < G::KK; ++kl) koff[kl] = wy * 2 * ROWB + hcq * S + swz<G>(hcq, kl * 4 + g);
  const int kcv = wx * 8 + 4 * g + (i >> 2);        // V^T fragment: key row addressed by this lane
  const int vsw = (kcv / G::TPR) % G::NSEG;
  const int vbase = wy * 2 * ROWB + kcv * S + (i & 3) * 8;
  // additive softmax masks: 0 where key c = 4g + r lies in the band of query column j (j <= c <= j + 8)
  // and the key row belongs to the window of the query's row, -inf elsewhere
  const float cs = scale * 1.44269504088896f;       // scores in log2 units
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
  const int gxw = ((int)gridDim.x - xcd + 7) >> 3;  // workgroups of this XCD
  const int t_end = (int)(((long long)ntiles * (xcd + 1)) >> 3);
  int tile = (int)(((long long)ntiles * xcd) >> 3) + wl;
  if (tile >= t_end) return;

  // measurement only: lane 0 of every wave of workgroup 0 samples the shader clock at phase boundaries
  __shared__ unsigned long long ts_lds[G::TS ? 16 * 48 : 1];
  int tsi = 0;
#define DI_TS()                                                                                        \
  do {                                                                                                 \
    if (G::TS && blockIdx.x == 0 && lane == 0 && tsi < 48) ts_lds[wave * 48 + tsi++] = __builtin_readcyclecounter(); \
  } while (0)
  DI_TS();

  // ---- staging: two register sets; prep() turns (operand, tile, unit) into a scalar base and per-lane
  // offsets (interior tiles: the lane's constant offsets; border tiles: out-of-image texels get a
  // clamped valid address and are zeroed when the registers are written to LDS), ld() moves one slot.
  uint4 RA[NLD], RB[NLD];
  unsigned okA = ~0u, okB = ~0u;
  unsigned off[NLD];
  const unsigned char *gbase = nullptr;
  auto prep = [&](unsigned &okbits, const __half *__restrict__ src, const TileCoord &t, int cu0) {
    long long tile_off = ((long long)(t.img * H + t.y0 - 4) * W + (t.x0 - 4)) * 256 + cu0 * 2;
    if (G::ONE_TILE) tile_off = ((long long)(16 - 4) * W + (16 - 4)) * 256 + cu0 * 2;
    gbase = reinterpret_cast<const unsigned char *>(src) + tile_off;
    const bool interior =
        G::ONE_TILE || (t.y0 >= 4 && t.x0 >= 4 && t.y0 - 4 + G::HR <= H && t.x0 - 4 + G::HC <= W);
    if (interior) {
      okbits = ~0u;
#pragma unroll
      for (int s = 0; s < NLD; ++s) off[s] = st_go[s];
    } else {
      okbits = 0;
#pragma unroll
      for (int s = 0; s < NLD; ++s) {
        const int gy = t.y0 - 4 + (st_pk[s] & 255), gx = t.x0 - 4 + (st_pk[s] >> 8);
        const int dy = min(max(gy, 0), H - 1) - gy, dx = min(max(gx, 0), W - 1) - gx;
        okbits |= (unsigned)((dy | dx) == 0) << s;
        off[s] = st_go[s] + (unsigned)(__mul24(dy, W) + dx) * 256u;
      }
    }
  };
  auto ld = [&](uint4 (&R)[NLD], int s) { R[s] = *reinterpret_cast<const uint4 *>(gbase + off[s]); };
  auto commit = [&](const uint4 (&R)[NLD], unsigned okbits, int buf) {
    if (okbits == ~0u) {
#pragma unroll
      for (int s = 0; s < NLD; ++s) *reinterpret_cast<uint4 *>(lds + buf * G::UNITB + st_lds[s]) = R[s];
    } else {
#pragma unroll
      for (int s = 0; s < NLD; ++s) {
        uint4 val = R[s];
        if (!((okbits >> s) & 1u)) val = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(lds + buf * G::UNITB + st_lds[s]) = val;
      }
    }
  };
  // Q^T fragments (query i, channels kk*32 + 8g .. +7) straight from global; queries beyond the map
  // edge (ragged tiles) read a clamped texel, their results are never stored
  h8 qf[4];
  const unsigned char *qbase = nullptr;
  auto prep_q = [&](const TileCoord &t) {
    const int gy = min(t.y0 + 2 * wy + qrow, H - 1), gx = min(t.x0 + 8 * wx + j, W - 1);
    qbase = reinterpret_cast<const unsigned char *>(q) + ((unsigned)((t.img * H + gy) * W + gx) << 8) + g * 16;
  };
  auto ld_q = [&](int kk) { qf[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(qbase + kk * 64)); };
  // finished output channels of a V unit wait here and are stored during the next pass
  // (after a v_permlane16_swap between the blocks of a pair a lane owns 8 consecutive channels: block pair pr is ONE 16-byte
  // store per lane, 64 contiguous bytes per pixel, instead of two 8-byte stores in 32-byte pieces)
  static_assert(G::NN % 2 == 0, "output blocks are stored in pairs");
  constexpr int NP = G::NN / 2;
  unsigned pend[G::NN][2];
  __half *pend_dst = nullptr;
  bool pend_ok = false;
  auto st_pend = [&](int pr) {
    if (pend_ok)
      *reinterpret_cast<uint4 *>(pend_dst + 32 * pr) =
          make_uint4(pend[2 * pr][0], pend[2 * pr][1], pend[2 * pr + 1][0], pend[2 * pr + 1][1]);
  };

  // ---- prologue: unit u of a tile (K units 0..NU-1, V units NU..2NU-1) uses LDS buffer u & 1 and
  // register set u & 1
  TileCoord cur = decode_tile(tile, tiles_x, per_img, G::TH, G::TW);
  prep(okA, k, cur, 0);
#pragma unroll
  for (int s = 0; s < NLD; ++s) ld(RA, s);
  prep(okB, k, cur, G::CU);
#pragma unroll
  for (int s = 0; s < NLD; ++s) ld(RB, s);
  prep_q(cur);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) ld_q(kk);
  commit(RA, okA, 0);
  lds_barrier();
  DI_TS();

  for (;;) {
    const bool has_next = tile + gxw < t_end;
    TileCoord nxt = cur;
    if (has_next) nxt = decode_tile(tile + gxw, tiles_x, per_img, G::TH, G::TW);

    // ---------------- S^T = K . Q^T over the K units
    f4 s[10];
#pragma unroll
    for (int rr = 0; rr < 10; ++rr) s[rr] = f4{0.f, 0.f, 0.f, 0.f};
    float sum = 0.f;
    h8 pf[5];
    static_for<0, NU>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      // background work of this pass: the loads of unit u + 2 (a K or V unit of this tile) and, in
      // the first pass, the output stores left over from the previous tile
      constexpr int un = u + 2;
      if constexpr (u & 1) prep(okB, un < NU ? k : v, cur, (un % NU) * G::CU);
      else prep(okA, un < NU ? k : v, cur, (un % NU) * G::CU);
      constexpr int nbg = NLD + (u == 0 ? NP : 0);
      constexpr int per = (nbg + 9) / 10;
      DI_TS();
      const unsigned char *buf = lds + (u & 1) * G::UNITB;
      static_for<0, 10>([&](auto rc) {
        constexpr int rr = decltype(rc)::value;
        static_for<0, per>([&](auto bc) {
          constexpr int b = rr * per + decltype(bc)::value;
          if constexpr (b < NLD) {
            if constexpr (u & 1) ld(RB, b);
            else ld(RA, b);
          } else if constexpr (b < nbg) {
            st_pend(b - NLD);
          }
        });
#pragma unroll
        for (int kl = 0; kl < G::KK; ++kl) {
          const uint4 raw = *reinterpret_cast<const uint4 *>(buf + koff[kl] + rr * ROWB);
          s[rr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, raw), qf[u * G::KK + kl], s[rr], 0, 0, 0);
        }
        pin_vmem();
      });
      DI_TS();
      if constexpr (u == NU - 1) {
        // ---- softmax over the 81 window slots of query i, in log2 units: y = s*cs + mask
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
      DI_TS();
      if constexpr (u & 1) commit(RA, okA, 0);    // unit u + 1
      else commit(RB, okB, 1);
      DI_TS();
      lds_barrier();
      DI_TS();
    });

    // ---------------- O^T = V^T . P^T over the V units, each unit finishes CU output channels
    const float inv = 1.f / sum;
    const int gy = cur.y0 + 2 * wy + qrow, gx = cur.x0 + 8 * wx + j;
    const bool pix_ok = gy < H && gx < W;
    __half *dst = out + ((long long)(cur.img * H + gy) * W + gx) * 128 + (g & 1) * 16 + (g >> 1) * 8;
    static_for<0, NU>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      constexpr int U = NU + u;                    // unit index in the tile
      constexpr int un = U + 2;                    // the unit loaded during this pass
      constexpr bool next_tile = un >= 2 * NU;     // ... K unit un - 2NU (0 or 1) of the next tile
      constexpr int uk = un - 2 * NU;
      const bool more = (u + 1 < NU) || has_next;
      const bool do_ld = !next_tile || has_next;
      if (do_ld) {
        const __half *src = next_tile ? k : v;
        const int cu0 = (next_tile ? uk : un - NU) * G::CU;
        if constexpr (U & 1) prep(okB, src, next_tile ? nxt : cur, cu0);
        else prep(okA, src, next_tile ? nxt : cur, cu0);
      }
      constexpr bool with_q = next_tile && uk == 0;
      if (with_q && do_ld) prep_q(nxt);
      constexpr int nst = u > 0 ? NP : 0;          // stores of the previous V unit
      constexpr int nbg = NLD + (with_q ? 4 : 0) + nst;
      constexpr int per = (nbg + 4) / 5;
      DI_TS();
      const unsigned char *buf = lds + (U & 1) * G::UNITB;
      f4 acc[G::NN];
#pragma unroll
      for (int nl = 0; nl < G::NN; ++nl) acc[nl] = f4{0.f, 0.f, 0.f, 0.f};
      static_for<0, 5>([&](auto pc) {
        constexpr int pr = decltype(pc)::value;
        static_for<0, per>([&](auto bc) {
          constexpr int b = pr * per + decltype(bc)::value;
          if constexpr (b < NLD) {
            if (do_ld) {
              if constexpr (U & 1) ld(RB, b);
              else ld(RA, b);
            }
          } else if constexpr (with_q && b < NLD + 4) {
            if (do_ld) ld_q(b - NLD);              // qf is dead since the last K unit
          } else if constexpr (b < nbg) {
            st_pend(b - NLD - (with_q ? 4 : 0));
          }
        });
#pragma unroll
        for (int nl = 0; nl < G::NN; ++nl) {
          const unsigned char *p0 = buf + vbase + ((nl ^ vsw) << 5) + 2 * pr * ROWB;
          const hv4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(p0));
          const hv4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(p0 + ROWB));
          h8 a;
          a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3];
          a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
          acc[nl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pf[pr], acc[nl], 0, 0, 0);
        }
        pin_vmem();
      });
      DI_TS();
#pragma unroll
      for (int nl = 0; nl < G::NN; ++nl) {
        const f4 o = acc[nl] * inv;
        h4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (_Float16)o[r];
        const uint2 raw = __builtin_bit_cast(uint2, ov);
        pend[nl][0] = raw.x;
        pend[nl][1] = raw.y;
      }
#pragma unroll
      for (int pr = 0; pr < NP; ++pr)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const u2v sw = __builtin_amdgcn_permlane16_swap(pend[2 * pr][d], pend[2 * pr + 1][d], false, false);
          pend[2 * pr][d] = sw[0];
          pend[2 * pr + 1][d] = sw[1];
        }
      pend_dst = dst + u * G::CU;
      pend_ok = pix_ok;
      DI_TS();
      if (more) {
        if constexpr (U & 1) commit(RA, okA, 0);   // unit U + 1
        else commit(RB, okB, 1);
        DI_TS();
        lds_barrier();
        DI_TS();
      }
    });
    if (!has_next) break;
    cur = nxt;
    tile += gxw;
  }
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) st_pend(pr);      // the last V unit of the last tile

  if (G::TS && blockIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      unsigned long long *dump = reinterpret_cast<unsigned long long *>(out);
      dump[0] = G::NW * 48;
      for (int e = 0; e < G::NW * 48; ++e) dump[1 + e] = ts_lds[e];
    }
  }
#undef DI_TS
}

template <class G>
static int launch(const void *q, const void *k, const void *v, void *out, int n, int H, int W, float scale,
                  int wg_per_cu, hipStream_t stream) {
  const int tiles_x = (W + G::TW - 1) / G::TW, tiles_y = (H + G::TH - 1) / G::TH;
  const long long ntiles = (long long)n * tiles_x * tiles_y;
  DI_REQUIRE((long long)n * H * W * 256 < (1ll << 31), "map of %d x %d x %d texels exceeds the 2 GiB offset range", n, H, W);
  static LdsRaised lds_raised;
  if (int rc = ensure_lds(lds_raised, (const void *)local_attn_m2_kernel<G>, G::LDS_BYTES)) return rc;
  const int n_cu = device_cus();
  if (n_cu <= 0) return DI_ERR_LAUNCH;
  // one workgroup per resident slot, a multiple of the 8 XCDs; never more than one per tile
  long long grid = (long long)n_cu * wg_per_cu;
  if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
  static const int grid_env = getenv("DI_LA_GRID") ? atoi(getenv("DI_LA_GRID")) : 0;   // measurement: workgroups of the launch
  if (grid_env > 0 && grid_env < grid) grid = grid_env / 8 * 8;
  hipEvent_t ev0, ev1;
  if (take_launch_events(ev0, ev1))                          // measurement: the dispatch's own begin / end time stamps
    hipExtLaunchKernelGGL(local_attn_m2_kernel<G>, dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, stream, ev0, ev1, 0,
                          (const __half *)q, (const __half *)k, (const __half *)v, (__half *)out, n, H, W, scale, tiles_x,
                          tiles_y);
  else
    hipLaunchKernelGGL(local_attn_m2_kernel<G>, dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, stream,
                       (const __half *)q, (const __half *)k, (const __half *)v, (__half *)out, n, H, W, scale,
                       tiles_x, tiles_y);
  return check_launch("local_attn_m2");
}

}  // namespace m2

// cfg: 0 = 16x8 tile / 64-ch units / 1 workgroup per CU;   2 = 16x4 / 64-ch / 2 per CU (the default of rounds 1-2);
//      1 = 8x8 / 64-ch / 2 per CU (the default since round 3: halo 16x16 = 4.0 texels staged per query instead of 4.5,
//      200 = 25 x 8 columns and 112 = 14 x 8 rows leave no ragged tile - 2100 tiles instead of 2184);  3 = 8x16 / 1 per CU;
//      4 = measurement build of 0 (phase timestamps).  (32-channel units and 24x8 tiles spill at the
//      VGPR caps their occupancy needs and were dropped.)
int launch_local_attn_mfma2(const void *q, const void *k, const void *v, void *out, int n, int H, int W,
                            float scale, int cfg, hipStream_t stream) {
  switch (cfg) {
    case 0: return m2::launch<m2::Cfg<2, 4, 64, 2>>(q, k, v, out, n, H, W, scale, 1, stream);
    case 1: return m2::launch<m2::Cfg<1, 4, 64, 2>>(q, k, v, out, n, H, W, scale, 2, stream);   // 8 x 8 tiles
    case 3: return m2::launch<m2::Cfg<1, 8, 64, 2>>(q, k, v, out, n, H, W, scale, 1, stream);   // 8 x 16 tiles
    case 2: return m2::launch<m2::Cfg<2, 2, 64, 2>>(q, k, v, out, n, H, W, scale, 2, stream);
    case 4: return m2::launch<m2::Cfg<2, 4, 64, 2, 6>>(q, k, v, out, n, H, W, scale, 1, stream);
  }
  set_error("unknown local_attn_mfma2 configuration %d", cfg);
  return DI_ERR_ARG;
}

}  // namespace di
