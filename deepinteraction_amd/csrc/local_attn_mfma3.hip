// Fused 9x9 local-window attention on the gfx950 matrix cores, third generation: the row-pair MFMA
// formulation of local_attn_mfma2.hip with PRODUCER / CONSUMER wavefronts and direct-to-LDS loads.
//
// Measured on the second-generation kernel (per-wave shader-clock timestamps, tools/la_ts8.py):
// a wavefront that issues its own staging loads stalls ~100 cycles per global_load_dwordx4 while the
// texture path (64 B/clk per CU) drains the queue of all eight waves, and it cannot issue MFMAs
// meanwhile; the ds_write pass of the staged registers and the address arithmetic sit on the same
// waves.  Interleaving the loads with the MFMAs only moves the stall.  So the roles are split:
//
//   * PRODUCER wavefronts (4 of 12) do nothing but move data: `global_load_lds_dwordx4` (the LDS-DMA
//     form: 64 lanes x 16 B land as one contiguous KiB of LDS, no VGPRs, no ds_write pass) for one
//     channel unit of the K or V halo per step, two units ahead of the consumers, into a ring of
//     THREE LDS buffers; a counted `s_waitcnt vmcnt` retires the unit the consumers need next and the
//     step's single workgroup barrier publishes it.  The XOR swizzle of the LDS image is applied to
//     the per-lane SOURCE address (the destination of an LDS-DMA is lane-linear).  Out-of-image halo
//     texels (border tiles) load a clamped address and are overwritten with zeros before the barrier.
//   * CONSUMER wavefronts (8 of 12, one 8 x 2 query block each) only read LDS, run the MFMAs and the
//     softmax; their only vector-memory work - the next tile's Q fragments and the output stores of
//     the previous unit - is spread one instruction at a time through the MFMA passes.
//
// Everything else is the second generation's: 16 MFMA columns = 8 queries of row y + 8 of row y+1,
// S^T = K.Q^T / softmax per lane column / O^T = V^T.P^T with the softmax registers as B operand,
// ds_read_b64_tr_b16 for V^T, 64-channel units (K0 K1 V0 V1 per tile), persistent workgroups that
// walk one contiguous tile range per XCD.
#include <type_traits>

#include "di_common.h"

namespace di {
namespace m3 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int WX_, int WY_, int NPW_, int WPS_, int DBG_ = 0>
struct Cfg {
  static constexpr int WX = WX_, WY = WY_, NPW = NPW_, WPS = WPS_;
  static constexpr bool TS = DBG_ == 6;              // measurement only: per-wave phase timestamps
  static constexpr bool ONE_TILE = DBG_ == 4;        // measurement only: every load from one tile (L2 hits)
  static constexpr int CU = 64;                      // channels per staged unit
  static constexpr int NCW = WX * WY;                // consumer wavefronts
  static constexpr int NW = NCW + NPW, NT = NW * 64;
  static constexpr int TW = 8 * WX, TH = 2 * WY;     // tile of query pixels
  static constexpr int HC = TW + 8, HR = TH + 8;     // halo columns / rows
  static constexpr int S = CU * 2;                   // bytes of one texel slice (128)
  static constexpr int NSEG = S / 32, TPR = 256 / S;
  static constexpr int ROWB = HC * S;
  static constexpr int UNITB = HR * ROWB;            // one staged unit
  static constexpr int NGL = UNITB / 1024;           // LDS-DMA wave-instructions per unit
  static constexpr int GPW = (NGL + NPW - 1) / NPW;  // ... per producer wavefront
  static constexpr int NU = 2, KK = 2, NN = 4;
  static constexpr int NBUF = 3;
  static constexpr int LDS_BYTES = NBUF * UNITB;
  static_assert(UNITB % 1024 == 0, "a unit is a whole number of 1 KiB LDS-DMA rows");
};

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// byte offset of logical 16-B chunk `c16` inside the slice of halo column `hc` (an involution on c16)
template <class G>
__device__ __forceinline__ int swz(int hc, int c16) {
  const int f = (hc / G::TPR) % G::NSEG;
  return ((((c16 >> 1) ^ f)) << 5) | ((c16 & 1) << 4);
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
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
__device__ __forceinline__ int mod3(int x) { return x >= 6 ? x - 6 : (x >= 3 ? x - 3 : x); }

template <class G>
__global__ __launch_bounds__(G::NT, G::WPS) void local_attn_m3_kernel(
    const __half *__restrict__ q, const __half *__restrict__ k, const __half *__restrict__ v,
    __half *__restrict__ out, int n, int H, int W, float scale, int tiles_x, int tiles_y) {
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr int ROWB = G::ROWB, S = G::S, GPW = G::GPW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- tiles: XCD x (workgroups with blockIdx % 8 == x share an L2) owns the contiguous range
  // [T*x/8, T*(x+1)/8) and walks it `gxw` tiles per round
  const int per_img = tiles_x * tiles_y, ntiles = n * per_img;
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
  const int gxw = ((int)gridDim.x - xcd + 7) >> 3;
  const int t_end = (int)(((long long)ntiles * (xcd + 1)) >> 3);
  int tile = (int)(((long long)ntiles * xcd) >> 3) + wl;
  if (tile >= t_end) return;

  __shared__ unsigned long long ts_lds[G::TS ? 16 * 48 : 1];
  int tsi = 0;
#define DI_TS()                                                                                        \
  do {                                                                                                 \
    if (G::TS && blockIdx.x == 0 && lane == 0 && tsi < 48) ts_lds[wave * 48 + tsi++] = __builtin_readcyclecounter(); \
  } while (0)
  DI_TS();

  // Step s (global over this workgroup's tiles, 4 per tile: K0 K1 V0 V1) lives in LDS buffer s % 3.
  // b0 = buffer of the current tile's first unit.
  int b0 = 0;

  if (wave >= G::NCW) {
    // =============================================================== PRODUCER
    const int pw = wave - G::NCW;
    unsigned st_go[GPW];
    int st_pk[GPW];
#pragma unroll
    for (int s = 0; s < GPW; ++s) {
      const int row = min(pw * GPW + s, G::NGL - 1);         // 1 KiB LDS row of this instruction
      const int offb = row * 1024 + lane * 16;               // where the lane's 16 B land
      const int hr = offb / ROWB, rem = offb - hr * ROWB;
      const int hc = rem / S, cp = (rem - hc * S) >> 4;      // stored chunk position -> logical chunk
      const int c16 = swz<G>(hc, cp) >> 4;
      st_go[s] = (unsigned)(hr * W + hc) * 256u + c16 * 16;
      st_pk[s] = hr | (hc << 8);
    }
    unsigned ok_old = ~0u, ok_new = ~0u;                     // in-image masks of the two units in flight
    auto issue = [&](const __half *__restrict__ src, const TileCoord &t, int cu0, int buf) {
      long long tile_off = ((long long)(t.img * H + t.y0 - 4) * W + (t.x0 - 4)) * 256 + cu0 * 2;
      if (G::ONE_TILE) tile_off = ((long long)(16 - 4) * W + (16 - 4)) * 256 + cu0 * 2;
      const unsigned char *base = reinterpret_cast<const unsigned char *>(src) + tile_off;
      const bool interior = G::ONE_TILE || (t.y0 >= 4 && t.x0 >= 4 && t.y0 - 4 + G::HR <= H && t.x0 - 4 + G::HC <= W);
      ok_old = ok_new;
      ok_new = ~0u;
      unsigned char *dst0 = lds + buf * G::UNITB + pw * GPW * 1024;
      if (interior) {
#pragma unroll
        for (int s = 0; s < GPW; ++s)
          if (pw * GPW + s < G::NGL)
            __builtin_amdgcn_global_load_lds((gptr_t)(base + st_go[s]), (lptr_t)(dst0 + s * 1024), 16, 0, 0);
      } else {
        ok_new = 0;
#pragma unroll
        for (int s = 0; s < GPW; ++s) {
          const int gy = t.y0 - 4 + (st_pk[s] & 255), gx = t.x0 - 4 + (st_pk[s] >> 8);
          const int dy = min(max(gy, 0), H - 1) - gy, dx = min(max(gx, 0), W - 1) - gx;
          ok_new |= (unsigned)((dy | dx) == 0) << s;
          const unsigned off = st_go[s] + (unsigned)(__mul24(dy, W) + dx) * 256u;
          if (pw * GPW + s < G::NGL)
            __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(dst0 + s * 1024), 16, 0, 0);
        }
      }
    };
    // zero the out-of-image texels of a landed unit (border tiles only).  Inline asm: a plain LDS store
    // would make the compiler drain EVERY outstanding LDS-DMA first (vmcnt(0)), the newer unit included.
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    auto fix = [&](unsigned okbits, int buf) {
      if (okbits != ~0u) {
        const unsigned a0 = (unsigned)(size_t)(lptr_t)(lds + buf * G::UNITB + pw * GPW * 1024 + lane * 16);
        const u4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int s = 0; s < GPW; ++s)
          if (!((okbits >> s) & 1u) && pw * GPW + s < G::NGL)
            asm volatile("ds_write_b128 %0, %1" ::"v"(a0 + s * 1024), "v"(zero) : "memory");
      }
    };

    TileCoord cur = decode_tile(tile, tiles_x, per_img, G::TH, G::TW);
    issue(k, cur, 0, 0);
    issue(k, cur, G::CU, 1);
    wait_vm<GPW>();
    fix(ok_old, 0);
    lds_barrier();
    for (;;) {
      const bool has_next = tile + gxw < t_end;
      TileCoord nxt = cur;
      if (has_next) nxt = decode_tile(tile + gxw, tiles_x, per_img, G::TH, G::TW);
      static_for<0, 4>([&](auto uc) {
        constexpr int U = decltype(uc)::value;
        constexpr int U2 = U + 2;                         // the unit issued during this step
        const bool has2 = U2 < 4 || has_next;
        const bool has1 = U + 1 < 4 || has_next;          // the unit the consumers need next
        if (has2) {
          const __half *src = (U2 & 2) ? v : k;           // units 0,1 (and 4,5 = next tile) are K; 2,3 are V
          issue(src, U2 < 4 ? cur : nxt, (U2 & 1) * G::CU, mod3(b0 + U2));
        }
        if (has1) {
          if (has2) wait_vm<GPW>();
          else wait_vm<0>();
          fix(has2 ? ok_old : ok_new, mod3(b0 + U + 1));
          lds_barrier();
        }
      });
      if (!has_next) break;
      cur = nxt;
      tile += gxw;
      b0 = mod3(b0 + 4);
    }
    return;
  }

  // ================================================================= CONSUMER
  const int wx = wave % G::WX, wy = wave / G::WX;
  const int i = lane & 15, g = lane >> 4;
  const int j = i & 7, qrow = i >> 3;
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
  h4 pend[G::NN];
  __half *pend_dst = nullptr;
  bool pend_ok = false;
  auto st_pend = [&](int nl) {
    if (pend_ok) *reinterpret_cast<h4 *>(pend_dst + 16 * nl) = pend[nl];
  };

  TileCoord cur = decode_tile(tile, tiles_x, per_img, G::TH, G::TW);
  prep_q(cur);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) ld_q(kk);
  lds_barrier();                                    // unit 0 has landed
  DI_TS();

  for (;;) {
    const bool has_next = tile + gxw < t_end;
    TileCoord nxt = cur;
    if (has_next) nxt = decode_tile(tile + gxw, tiles_x, per_img, G::TH, G::TW);

    // ---------------- S^T = K . Q^T over the two K units
    f4 s[10];
#pragma unroll
    for (int rr = 0; rr < 10; ++rr) s[rr] = f4{0.f, 0.f, 0.f, 0.f};
    float sum = 0.f;
    h8 pf[5];
    static_for<0, 2>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const unsigned char *buf = lds + mod3(b0 + u) * G::UNITB;
      // explicit software pipeline: the K fragment of MFMA step j + PFK is requested before step j runs
      // (20 steps = 10 key rows x 2 channel halves); the order is pinned, the compiler would otherwise
      // sink the LDS reads next to their uses and expose their latency with only two waves per SIMD
      constexpr int PFK = 8;
      uint4 kf[20];
      static_for<0, PFK>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        kf[j] = *reinterpret_cast<const uint4 *>(buf + koff[j & 1] + (j >> 1) * ROWB);
      });
      static_for<0, 20>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int rr = j >> 1, kl = j & 1;
        if constexpr (j + PFK < 20)
          kf[j + PFK] = *reinterpret_cast<const uint4 *>(buf + koff[(j + PFK) & 1] + ((j + PFK) >> 1) * ROWB);
        if constexpr (u == 0 && kl == 0 && rr < G::NN) st_pend(rr);     // stores left over from the previous tile
        s[rr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, kf[j]), qf[u * G::KK + kl], s[rr], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
      DI_TS();
      if constexpr (u == 1) {
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
        DI_TS();
      }
      lds_barrier();
      DI_TS();
    });

    // ---------------- O^T = V^T . P^T over the two V units, each unit finishes 64 output channels
    const float inv = 1.f / sum;
    const int gy = cur.y0 + 2 * wy + qrow, gx = cur.x0 + 8 * wx + j;
    const bool pix_ok = gy < H && gx < W;
    __half *dst = out + ((long long)(cur.img * H + gy) * W + gx) * 128 + 4 * g;
    static_for<0, 2>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const bool more = u == 0 || has_next;
      if (u == 0 && has_next) prep_q(nxt);
      const unsigned char *buf = lds + mod3(b0 + 2 + u) * G::UNITB;
      f4 acc[G::NN];
#pragma unroll
      for (int nl = 0; nl < G::NN; ++nl) acc[nl] = f4{0.f, 0.f, 0.f, 0.f};
      // same explicit pipeline for the transposed V fragments (20 steps = 5 row pairs x 4 channel blocks,
      // two ds_read_b64_tr_b16 each)
      constexpr int PFV = 8;
      hv4 va[20], vb[20];
      auto rd_v = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int pr = j >> 2, nl = j & 3;
        const unsigned char *p0 = buf + vbase + ((nl ^ vsw) << 5) + 2 * pr * ROWB;
        va[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(p0));
        vb[j] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(p0 + ROWB));
      };
      static_for<0, PFV>(rd_v);
      static_for<0, 20>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int pr = j >> 2, nl = j & 3;
        if constexpr (j + PFV < 20) rd_v(std::integral_constant<int, j + PFV>{});
        if constexpr (nl == 0) {
          if constexpr (u == 0) {
            if (pr < 4 && has_next) ld_q(pr);          // qf is dead since the last K unit
          } else if constexpr (pr < G::NN) {
            st_pend(pr);                                // output channels of V unit 0
          }
        }
        h8 a;
        a[0] = va[j][0]; a[1] = va[j][1]; a[2] = va[j][2]; a[3] = va[j][3];
        a[4] = vb[j][0]; a[5] = vb[j][1]; a[6] = vb[j][2]; a[7] = vb[j][3];
        acc[nl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pf[pr], acc[nl], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
      DI_TS();
#pragma unroll
      for (int nl = 0; nl < G::NN; ++nl) {
        const f4 o = acc[nl] * inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) pend[nl][r] = (_Float16)o[r];
      }
      pend_dst = dst + u * G::CU;
      pend_ok = pix_ok;
      if (more) {
        lds_barrier();
        DI_TS();
      }
    });
    if (!has_next) break;
    cur = nxt;
    tile += gxw;
    b0 = mod3(b0 + 4);
  }
#pragma unroll
  for (int nl = 0; nl < G::NN; ++nl) st_pend(nl);   // the last V unit of the last tile

  if (G::TS && blockIdx.x == 0) {
    // only consumers reach this point; the producers have returned
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (lane == 0) {
      unsigned long long *dump = reinterpret_cast<unsigned long long *>(out) + 1 + wave * 48;
      for (int e = 0; e < 48; ++e) dump[e] = ts_lds[wave * 48 + e];
      if (wave == 0) reinterpret_cast<unsigned long long *>(out)[0] = G::NCW * 48;
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
  static int n_cu = 0;   // idempotent initialisation; a race only repeats the queries
  if (n_cu == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
      set_error("cannot query the CU count");
      return DI_ERR_LAUNCH;
    }
    hipError_t e = hipFuncSetAttribute((const void *)local_attn_m3_kernel<G>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return DI_ERR_LAUNCH;
    }
    n_cu = cus;
  }
  long long grid = (long long)n_cu * wg_per_cu;
  if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
  hipLaunchKernelGGL(local_attn_m3_kernel<G>, dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, stream,
                     (const __half *)q, (const __half *)k, (const __half *)v, (__half *)out, n, H, W, scale,
                     tiles_x, tiles_y);
  return check_launch("local_attn_m3");
}

}  // namespace m3

// cfg: 0 = 16x8 tile, 8 consumer + 4 producer wavefronts, 1 workgroup per CU; 1 = its timestamp build
int launch_local_attn_mfma3(const void *q, const void *k, const void *v, void *out, int n, int H, int W,
                            float scale, int cfg, hipStream_t stream) {
  switch (cfg) {
    case 0: return m3::launch<m3::Cfg<2, 4, 4, 3>>(q, k, v, out, n, H, W, scale, 1, stream);
    case 1: return m3::launch<m3::Cfg<2, 4, 4, 3, 6>>(q, k, v, out, n, H, W, scale, 1, stream);
    case 2: return m3::launch<m3::Cfg<2, 4, 4, 3, 4>>(q, k, v, out, n, H, W, scale, 1, stream);
  }
  set_error("unknown local_attn_mfma3 configuration %d", cfg);
  return DI_ERR_ARG;
}

}  // namespace di
