// Fused 9x9 window attention FOR TRAINING on the gfx950 matrix cores (fp16 maps, C = 128): a forward that keeps the
// log-sum-exp of every query, and a backward that RECOMPUTES the soft-max from it - the (n, H, W, 81) weight tensor of
// the reference (similarFunction -> F.softmax -> weightingFunction, encoder_utils.py:36-81,132-134; kernels
// similar.cu:43-92, weighting.cu:44-122) never exists.  Mixed-precision training path (torch.autocast): operands and
// results fp16, every accumulation float32.
//
// One kernel template, four unit programs.  The data flow is the second inference generation's
// (local_attn_mfma2.hip): persistent workgroups of 4 wavefronts, 8 x 8 CENTRE pixels per workgroup (a wavefront owns
// 8 x 2 = 16 MFMA columns), the 16 x 16 HALO staged through two LDS buffers in 64-channel units, the loads of unit u + 2
// in flight during the MFMA pass of unit u, XCD-contiguous tile ranges, swizzled LDS.  A unit is one of
//   S0 / S1 : acc[halo pixel, centre] += <halo unit, centre fragments b0 / b1>          (MFMA A = halo rows from LDS)
//   O0 / O1 : out[channel, centre]    = sum over halo pixels  halo^T . P  /  halo^T . dS  (A = ds_read_tr of the unit)
// and the window relation is symmetric (pixel a is in the window of b iff b is in the window of a), so the SAME band
// masks serve a query-centred and a key-centred pass:
//   FWD    centre = queries (b0 = Q), halo = K, V:   S0 S0 | soft-max, L = log-sum-exp out | O0 O0 -> out
//   BWD_Q  centre = queries (b0 = Q, b1 = dO), halo = K, V:
//          S0 S0 | P = exp(S - L) | S1 S1 (dP - D = <V, dO> - D: the accumulators start at -D) | dS = P (dP - D) | O1 O1 (K^T dS) -> dQ
//   BWD_V  centre = KEYS (b0 = K), halo = Q, dO, with L of the halo pixels in an LDS table:
//          S0 S0 | P | O0 O0 (dO^T P) -> dV
//   BWD_K  centre = KEYS (b0 = K, b1 = V), halo = Q, dO, L and D of the halo pixels in LDS tables:
//          S0 S0 | P | S1 S1 | dS | O1 O1 (Q^T dS) -> dK
// (one key-centred program of 8 units needs P and dS live together and spills 60-117 registers at 2 workgroups per CU)
// with D = <dO, O> per query (rowdot128_kernel).  Zero padding as in the reference: a key beyond the map edge has k = v =
// 0 and TAKES PART in the soft-max (logit 0); it receives no gradient; a query beyond the edge does not exist (its L is
// +inf in the table, so P = 0).  dS is packed to fp16 after a per-centre normalisation by its largest magnitude (the
// same trick as the forward's exp(s - max)), so gradient magnitudes never meet the fp16 range.
#include <math.h>
#include <type_traits>

#include "di_common.h"

namespace di {
namespace lt {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

enum { FWD = 0, BWD_Q = 1, BWD_V = 2, BWD_K = 3 };
constexpr bool key_centred(int mode) { return mode >= BWD_V; }
constexpr bool short_prog(int mode) { return mode == FWD || mode == BWD_V; }
enum { S0 = 0, S1 = 1, O0 = 2, O1 = 3 };

constexpr int n_units(int mode) { return short_prog(mode) ? 4 : 6; }
constexpr int unit_kind(int mode, int U) {
  if (U < 2) return S0;
  if (short_prog(mode)) return O0;
  return U < 4 ? S1 : O1;
}
constexpr bool is_s(int kind) { return kind == S0 || kind == S1; }
constexpr int src_of(int kind) { return (kind == S0 || kind == O1) ? 0 : 1; }   // halo operand 0 or 1

template <int WX_, int WY_, int MODE_>
struct Cfg {
  static constexpr int WX = WX_, WY = WY_, MODE = MODE_, CU = 64, WPS = 2;
  static constexpr int NUNITS = n_units(MODE_);
  static constexpr int NW = WX * WY, NT = NW * 64;
  static constexpr int TW = 8 * WX, TH = 2 * WY;     // tile of centre pixels
  static constexpr int HC = TW + 8, HR = TH + 8;     // halo columns / rows
  static constexpr int S = CU * 2;                   // bytes of one texel slice
  static constexpr int CPT = S / 16, NSEG = S / 32, TPR = 256 / S;
  static constexpr int ROWB = HC * S, UNITB = HR * ROWB;
  static constexpr int NCHUNK = HR * HC * CPT;
  static constexpr int NLD = (NCHUNK + NT - 1) / NT;
  static constexpr int KK = CU / 32, NN = CU / 16;
  static constexpr int LDS_BYTES = 2 * UNITB;
  static constexpr int NB = short_prog(MODE_) ? 4 : 8;    // centre fragments (b0, b1)
  static_assert(HR * HC <= NT, "one thread per halo pixel for the L / D table");
};

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

template <class G>
__device__ __forceinline__ int swz(int hc, int c16) {
  const int f = (hc / G::TPR) % G::NSEG;
  return ((((c16 >> 1) ^ f)) << 5) | ((c16 & 1) << 4);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
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
__global__ __launch_bounds__(G::NT, G::WPS) void window_train_kernel(
    const __half *__restrict__ h0, const __half *__restrict__ h1, const __half *__restrict__ c0,
    const __half *__restrict__ c1, float *__restrict__ lse, const float *__restrict__ dsum,
    __half *__restrict__ out0, __half *__restrict__ out1, int n, int H, int W, float scale, int tiles_x, int tiles_y) {
  extern __shared__ __align__(16) unsigned char lds[];
  __shared__ __align__(16) float ltab[key_centred(G::MODE) ? G::HR * G::HC : 4];
  __shared__ __align__(16) float dtab[G::MODE == BWD_K ? G::HR * G::HC : 4];
  constexpr int NLD = G::NLD, ROWB = G::ROWB, S = G::S, MODE = G::MODE, NUNITS = G::NUNITS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wx = wave % G::WX, wy = wave / G::WX;
  const int i = lane & 15, g = lane >> 4;
  const int j = i & 7, qrow = i >> 3;

  // staging constants of this lane: slot s moves 16-B chunk e = s * 256 + tid of the unit, i.e. (HC = 16, 8 chunks per texel)
  // halo row 2 s + (tid >> 7), halo column (tid >> 3) & 15, chunk tid & 7 - everything but the row is the lane's own constant
  static_assert(G::HC == 16 && G::NT == 256 && G::CPT == 8 && G::NCHUNK == NLD * G::NT, "8 x 8 centre tiles, 4 wavefronts");
  const int hr_l = tid >> 7, hc_l = (tid >> 3) & 15, c16_l = tid & 7;
  const int lds0 = hr_l * ROWB + hc_l * S + swz<G>(hc_l, c16_l);
  const unsigned go0 = (unsigned)(hr_l * W + hc_l) * 256u + c16_l * 16;
  const unsigned row2 = (unsigned)W * 512u;          // two map rows
  const int wrow = W << 8;
  const int hcq = wx * 8 + i;
  int koff[G::KK];
#pragma unroll
  for (int kl = 0; kl < G::KK; ++kl) koff[kl] = wy * 2 * ROWB + hcq * S + swz<G>(hcq, kl * 4 + g);
  const int kcv = wx * 8 + 4 * g + (i >> 2);
  const int vsw = (kcv / G::TPR) % G::NSEG;
  const int vbase = wy * 2 * ROWB + kcv * S + (i & 3) * 8;
  const float cs = scale * 1.44269504088896f;
  // additive soft-max mask of key / halo column c = 4g + r for centre column j: 0 inside the band j <= c <= j + 8; halo row 0
  // belongs to the window of the wave's upper centre row only, halo row 9 to the lower one
  f4 nm_mid;
#pragma unroll
  for (int r = 0; r < 4; ++r) nm_mid[r] = (4 * g + r >= j && 4 * g + r <= j + 8) ? 0.f : -INFINITY;
  auto mask_of = [&](int rr) {
    const float edge = (rr == 0 ? qrow == 0 : (rr == 9 ? qrow == 1 : true)) ? 0.f : -INFINITY;
    return nm_mid + edge;
  };

  const int per_img = tiles_x * tiles_y, ntiles = n * per_img;
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
  const int gxw = ((int)gridDim.x - xcd + 7) >> 3;
  const int t_end = (int)(((long long)ntiles * (xcd + 1)) >> 3);
  int tile = (int)(((long long)ntiles * xcd) >> 3) + wl;
  if (tile >= t_end) return;

  // ---- halo staging (register sets A / B, see local_attn_mfma2.hip)
  uint4 RA[NLD], RB[NLD];
  unsigned okA = ~0u, okB = ~0u;
  const unsigned char *gbase = nullptr;
  int py0 = 0;                                       // map row of this lane's slot 0, may be outside the map
  unsigned dxo = 0;                                  // byte offset that clamps the lane's column into the map
  // prep() turns (operand, tile, unit) into a scalar base + the lane's clamping terms; out-of-image texels get a clamped
  // valid address and are zeroed when the registers are written to LDS
  auto prep = [&](unsigned &okbits, const __half *__restrict__ src, const TileCoord &t, int cu0) {
    const long long tile_off = ((long long)(t.img * H + t.y0 - 4) * W + (t.x0 - 4)) * 256 + cu0 * 2;
    gbase = reinterpret_cast<const unsigned char *>(src) + tile_off;
    py0 = t.y0 - 4 + hr_l;
    const int gx = t.x0 - 4 + hc_l;
    const int dx = min(max(gx, 0), W - 1) - gx;
    dxo = (unsigned)(dx * 256);
    const bool interior = t.y0 >= 4 && t.x0 >= 4 && t.y0 - 4 + G::HR <= H && t.x0 - 4 + G::HC <= W;
    if (interior) {
      okbits = ~0u;
    } else {
      okbits = 0;
#pragma unroll
      for (int s = 0; s < NLD; ++s) {
        const int gy = py0 + 2 * s;
        okbits |= (unsigned)(dx == 0 && gy >= 0 && gy < H) << s;
      }
    }
  };
  auto ld = [&](uint4 (&R)[NLD], int s) {
    const int gy = py0 + 2 * s;
    const int dy = min(max(gy, 0), H - 1) - gy;
    R[s] = *reinterpret_cast<const uint4 *>(gbase + (go0 + s * row2 + (unsigned)__mul24(dy, wrow) + dxo));
  };
  auto commit = [&](const uint4 (&R)[NLD], unsigned okbits, int buf) {
    if (okbits == ~0u) {
#pragma unroll
      for (int s = 0; s < NLD; ++s) *reinterpret_cast<uint4 *>(lds + buf * G::UNITB + lds0 + s * 2 * ROWB) = R[s];
    } else {
#pragma unroll
      for (int s = 0; s < NLD; ++s) {
        uint4 val = R[s];
        if (!((okbits >> s) & 1u)) val = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(lds + buf * G::UNITB + lds0 + s * 2 * ROWB) = val;
      }
    }
  };
  // ---- centre fragments (pixel i of the wave's 8 x 2, channels kk * 32 + 8g .. + 7) and the centre's scalars
  h8 bf[G::NB];
  float Lc = 0.f, Dc = 0.f;
  unsigned coff = 0, cpix = 0;
  auto prep_c = [&](const TileCoord &t) {
    const int gy = min(t.y0 + 2 * wy + qrow, H - 1), gx = min(t.x0 + 8 * wx + j, W - 1);
    cpix = (unsigned)((t.img * H + gy) * W + gx);
    coff = (cpix << 8) + g * 16;
  };
  auto ld_c = [&](int b) {
    const unsigned char *base = reinterpret_cast<const unsigned char *>(b < 4 ? c0 : c1);
    bf[b] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(base + coff + (b & 3) * 64));
  };
  auto ld_scalars = [&]() {
    if constexpr (MODE == BWD_Q) {
      Lc = lse[cpix];
      Dc = dsum[cpix];
    }
  };
  // (output blocks are stored in pairs: after a v_permlane16_swap a lane owns 8 consecutive channels - one 16-byte store)
  constexpr int NP = G::NN / 2;
  unsigned pend[G::NN][2];
  __half *pend_dst = nullptr;
  bool pend_ok = false;
  auto st_pend = [&](int pr) {
    if (pend_ok)
      *reinterpret_cast<uint4 *>(pend_dst + 32 * pr) =
          make_uint4(pend[2 * pr][0], pend[2 * pr][1], pend[2 * pr + 1][0], pend[2 * pr + 1][1]);
  };
  auto halo_src = [&](int kind) { return src_of(kind) ? h1 : h0; };

  // ---- prologue
  TileCoord cur = decode_tile(tile, tiles_x, per_img, G::TH, G::TW);
  prep(okA, h0, cur, 0);
#pragma unroll
  for (int s = 0; s < NLD; ++s) ld(RA, s);
  prep(okB, h0, cur, G::CU);
#pragma unroll
  for (int s = 0; s < NLD; ++s) ld(RB, s);
  prep_c(cur);
#pragma unroll
  for (int b = 0; b < 4; ++b) ld_c(b);
  ld_scalars();
  commit(RA, okA, 0);
  lds_barrier();

  for (;;) {
    const bool has_next = tile + gxw < t_end;
    TileCoord nxt = cur;
    if (has_next) nxt = decode_tile(tile + gxw, tiles_x, per_img, G::TH, G::TW);

    f4 s[10];
    h8 pf[5], pg[5];
    float inv0 = 1.f, inv1 = 1.f;
    const int gy = cur.y0 + 2 * wy + qrow, gx = cur.x0 + 8 * wx + j;
    const bool pix_ok = gy < H && gx < W;
    const long long pix = (long long)(cur.img * H + gy) * W + gx;
    float tl = INFINITY, td = 0.f;                   // key-centred passes: L, D of halo pixel `tid`

    static_for<0, NUNITS>([&](auto uc) {
      constexpr int U = decltype(uc)::value;
      constexpr int kind = unit_kind(MODE, U);
      constexpr int un = U + 2;
      constexpr bool next_tile = un >= NUNITS;
      constexpr int uw = next_tile ? un - NUNITS : un;          // the unit loaded during this pass
      constexpr int ukind = unit_kind(MODE, uw);
      const bool more = (U + 1 < NUNITS) || has_next;
      const bool do_ld = !next_tile || has_next;
      if (do_ld) {
        if constexpr (U & 1) prep(okB, halo_src(ukind), next_tile ? nxt : cur, (uw & 1) * G::CU);
        else prep(okA, halo_src(ukind), next_tile ? nxt : cur, (uw & 1) * G::CU);
      }
      // centre fragments ride as background loads where the registers are dead: b0 of the NEXT tile in the last-but-one
      // pass, b1 of THIS tile in its first pass (first use: unit 2)
      constexpr bool with_c = U == NUNITS - 2;
      constexpr bool with_b1 = U == 0 && G::NB > 4;
      if (with_c && do_ld) prep_c(nxt);
      constexpr int ncl = (with_c || with_b1) ? 4 : 0;
      constexpr int cfirst = with_b1 ? 4 : 0;
      constexpr int pkind = unit_kind(MODE, U == 0 ? NUNITS - 1 : U - 1);
      constexpr int nst = is_s(pkind) ? 0 : NP;                 // stores of the previous O unit
      constexpr int nbg = NLD + ncl + nst;
      constexpr int steps = is_s(kind) ? 10 : 5;
      constexpr int per = (nbg + steps - 1) / steps;
      const unsigned char *buf = lds + (U & 1) * G::UNITB;
      auto background = [&](auto bc) {
        constexpr int b = decltype(bc)::value;
        if constexpr (b < NLD) {
          if (do_ld) {
            if constexpr (U & 1) ld(RB, b);
            else ld(RA, b);
          }
        } else if constexpr (b < NLD + ncl) {
          if (with_b1 || do_ld) ld_c(cfirst + b - NLD);
        } else if constexpr (b < nbg) {
          st_pend(b - NLD - ncl);
        }
      };
      if constexpr (key_centred(MODE) && U == 0) {
        if (tid < G::HR * G::HC) {
          const int hr = tid / G::HC, hc = tid - hr * G::HC;
          const int ty = cur.y0 - 4 + hr, tx = cur.x0 - 4 + hc;
          if (ty >= 0 && ty < H && tx >= 0 && tx < W) {
            const long long p = (long long)(cur.img * H + ty) * W + tx;
            tl = lse[p];
            if constexpr (MODE == BWD_K) td = dsum[p];
          }
        }
      }

      if constexpr (is_s(kind)) {
        if constexpr (U == 0) {
#pragma unroll
          for (int rr = 0; rr < 10; ++rr) s[rr] = f4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (U == 2) {                      // dP - D: the accumulators start at -D of their halo pixel / centre
#pragma unroll
          for (int rr = 0; rr < 10; ++rr) {
            if constexpr (MODE == BWD_K)
              s[rr] = -*reinterpret_cast<const f4 *>(&dtab[(wy * 2 + rr) * G::HC + wx * 8 + 4 * g]);
            else
              s[rr] = f4{-Dc, -Dc, -Dc, -Dc};
          }
        }
        constexpr int b0 = (kind == S0 ? 0 : 4) + (U & 1) * G::KK;
        static_for<0, 10>([&](auto rc) {
          constexpr int rr = decltype(rc)::value;
          static_for<0, per>([&](auto bc) { background(std::integral_constant<int, rr * per + decltype(bc)::value>{}); });
#pragma unroll
          for (int kl = 0; kl < G::KK; ++kl) {
            const uint4 raw = *reinterpret_cast<const uint4 *>(buf + koff[kl] + rr * ROWB);
            s[rr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, raw), bf[b0 + kl], s[rr], 0, 0, 0);
          }
          pin_vmem();
        });
        if constexpr (U == 1) {
          // ---- probabilities of the 81 window slots, in log2 units: y = s * cs + mask
          if constexpr (MODE == FWD) {
            float m = -INFINITY;
#pragma unroll
            for (int rr = 0; rr < 10; ++rr) {
              const f4 nm = mask_of(rr);
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
                for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(d[r]);
                sum2 += f2{e[0], e[1]} + f2{e[2], e[3]};
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[4 * t + r] = (_Float16)e[r];
              }
              pf[pr] = pk;
            }
            float sum = sum2[0] + sum2[1];
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            inv0 = 1.f / sum;
            if (g == 0 && pix_ok) lse[pix] = m + __builtin_amdgcn_logf(sum);      // v_log_f32 = log2
          } else {
#pragma unroll
            for (int pr = 0; pr < 5; ++pr) {
              h8 pk;
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                const int rr = 2 * pr + t;
                const f4 nm = mask_of(rr);
                f4 l4 = f4{Lc, Lc, Lc, Lc};
                if constexpr (key_centred(MODE))
                  l4 = *reinterpret_cast<const f4 *>(&ltab[(wy * 2 + rr) * G::HC + wx * 8 + 4 * g]);
                const f4 d = s[rr] * cs + nm - l4;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[4 * t + r] = (_Float16)__builtin_amdgcn_exp2f(d[r]);
              }
              pf[pr] = pk;
            }
          }
        }
        if constexpr (U == 3) {
          // ---- dS = P (dP - D), normalised per centre by its largest magnitude before the fp16 packing
          float amax = 0.f;
#pragma unroll
          for (int rr = 0; rr < 10; ++rr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = (float)pf[rr >> 1][4 * (rr & 1) + r];
              const float ds = p * s[rr][r];
              s[rr][r] = ds;
              amax = fmaxf(amax, fabsf(ds));
            }
          }
          amax = fmaxf(amax, __shfl_xor(amax, 16));
          amax = fmaxf(amax, __shfl_xor(amax, 32));
          const float ia = amax > 0.f ? 1.f / amax : 0.f;
#pragma unroll
          for (int pr = 0; pr < 5; ++pr) {
            h8 pk;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) pk[4 * t + r] = (_Float16)(s[2 * pr + t][r] * ia);
            pg[pr] = pk;
          }
          inv1 = amax * scale;
        }
      } else {
        f4 acc[G::NN];
#pragma unroll
        for (int nl = 0; nl < G::NN; ++nl) acc[nl] = f4{0.f, 0.f, 0.f, 0.f};
        static_for<0, 5>([&](auto pc) {
          constexpr int pr = decltype(pc)::value;
          static_for<0, per>([&](auto bc) { background(std::integral_constant<int, pr * per + decltype(bc)::value>{}); });
#pragma unroll
          for (int nl = 0; nl < G::NN; ++nl) {
            const unsigned char *p0 = buf + vbase + ((nl ^ vsw) << 5) + 2 * pr * ROWB;
            const hv4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(p0));
            const hv4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(p0 + ROWB));
            h8 a;
            a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3];
            a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
            acc[nl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, kind == O0 ? pf[pr] : pg[pr], acc[nl], 0, 0, 0);
          }
          pin_vmem();
        });
        const float os = kind == O0 ? inv0 : inv1;
#pragma unroll
        for (int nl = 0; nl < G::NN; ++nl) {
          const f4 o = acc[nl] * os;
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
        pend_dst = (kind == O0 ? out0 : out1) + pix * 128 + (g & 1) * 16 + (g >> 1) * 8 + (U & 1) * G::CU;
        pend_ok = pix_ok;
      }
      if constexpr (with_c) {
        if (do_ld) ld_scalars();
      }
      if constexpr (key_centred(MODE) && U == 0) {
        if (tid < G::HR * G::HC) {
          ltab[tid] = tl;
          if constexpr (MODE == BWD_K) dtab[tid] = td;
        }
      }
      if (more) {
        if constexpr (U & 1) commit(RA, okA, 0);
        else commit(RB, okB, 1);
        lds_barrier();
      }
    });
    if (!has_next) break;
    cur = nxt;
    tile += gxw;
  }
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) st_pend(pr);
}

// D[p] = <a[p, :], b[p, :]> over 128 fp16 channels (16 lanes per pixel)
__global__ __launch_bounds__(256) void rowdot128_kernel(const __half *__restrict__ a, const __half *__restrict__ b,
                                                        float *__restrict__ out, long long npix) {
  const long long p = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  if (p >= npix) return;
  const Pack8<__half> x = ld8(a + p * 128 + l * 8), y = ld8(b + p * 128 + l * 8);
  const float d = row16_sum(dot8(x, y, 0.f));
  if (l == 0) out[p] = d;
}

template <class G>
static int launch(const void *h0, const void *h1, const void *c0, const void *c1, float *lse, const float *dsum,
                  void *out0, void *out1, int n, int H, int W, float scale, hipStream_t stream, const char *what) {
  const int tiles_x = (W + G::TW - 1) / G::TW, tiles_y = (H + G::TH - 1) / G::TH;
  const long long ntiles = (long long)n * tiles_x * tiles_y;
  static LdsRaised lds_raised;
  if (int rc = ensure_lds(lds_raised, (const void *)window_train_kernel<G>, G::LDS_BYTES)) return rc;
  const int n_cu = device_cus();
  if (n_cu <= 0) return DI_ERR_LAUNCH;
  long long grid = (long long)n_cu * G::WPS;
  if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
  hipLaunchKernelGGL(window_train_kernel<G>, dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, stream,
                     (const __half *)h0, (const __half *)h1, (const __half *)c0, (const __half *)c1, lse, dsum,
                     (__half *)out0, (__half *)out1, n, H, W, scale, tiles_x, tiles_y);
  return check_launch(what);
}

static int check_shape(int n, int H, int W) {
  DI_REQUIRE(n > 0 && H > 0 && W > 0, "empty feature map n=%d H=%d W=%d", n, H, W);
  DI_REQUIRE((long long)n * H * W * 256 < (1ll << 31), "map of %d x %d x %d texels exceeds the 2 GiB offset range", n, H, W);
  return DI_OK;
}

}  // namespace lt
}  // namespace di

extern "C" {

int di_local_attn_train_fwd(const void *q, const void *k, const void *v, void *out, float *lse, int n, int H, int W,
                            float scale, void *stream) {
  if (int rc = di::lt::check_shape(n, H, W)) return rc;
  using G = di::lt::Cfg<1, 4, di::lt::FWD>;
  return di::lt::launch<G>(k, v, q, nullptr, lse, nullptr, out, nullptr, n, H, W, scale, (hipStream_t)stream,
                           "local_attn_train_fwd");
}

int di_local_attn_train_bwd(const void *q, const void *k, const void *v, const void *out, const void *grad_out,
                            const float *lse, float *dsum, void *grad_q, void *grad_k, void *grad_v, int n, int H, int W,
                            float scale, void *stream) {
  if (int rc = di::lt::check_shape(n, H, W)) return rc;
  const long long npix = (long long)n * H * W;
  hipLaunchKernelGGL(di::lt::rowdot128_kernel, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const __half *)grad_out, (const __half *)out, dsum, npix);
  if (int rc = di::check_launch("local_attn_train_rowdot")) return rc;
  using GQ = di::lt::Cfg<1, 4, di::lt::BWD_Q>;
  if (int rc = di::lt::launch<GQ>(k, v, q, grad_out, const_cast<float *>(lse), dsum, nullptr, grad_q, n, H, W, scale,
                                  (hipStream_t)stream, "local_attn_train_bwd_q"))
    return rc;
  using GV = di::lt::Cfg<1, 4, di::lt::BWD_V>;
  if (int rc = di::lt::launch<GV>(q, grad_out, k, nullptr, const_cast<float *>(lse), dsum, grad_v, nullptr, n, H, W, scale,
                                  (hipStream_t)stream, "local_attn_train_bwd_v"))
    return rc;
  using GK = di::lt::Cfg<1, 4, di::lt::BWD_K>;
  return di::lt::launch<GK>(q, grad_out, k, v, const_cast<float *>(lse), dsum, nullptr, grad_k, n, H, W, scale,
                            (hipStream_t)stream, "local_attn_train_bwd_k");
}

}  // extern "C"
