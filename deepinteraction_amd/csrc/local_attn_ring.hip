// Fused 9x9 local-window attention on the gfx950 matrix cores, fifth generation (fp16 maps, C = 128):
// ONE fat workgroup per CU, a 144 KB ring of halo ROWS filled by LDS-DMA, and NO workgroup barrier in the steady state.
//
// What the earlier generations measured (docs/DESIGN_history.md section 6): a launch takes (rounds of tiles) x (tile latency), and the
// tile latency (8.5-9 us) is a chain of dependent phases - stage a unit, barrier, 20 MFMAs per wave, barrier, ... - in which
// a wave is idle 70 % of the time; no unit of the CU is saturated.  The second generation (local_attn_mfma2.hip) holds two
// units in registers + two in LDS per workgroup and pays a commit pass and a barrier per unit; the DMA generations
// (round 3, deleted in round 4) moved 64-byte pieces, had at most 48 KB in flight and kept one barrier per 10-20 MFMAs.  Here:
//
//   * the halo of a 16 x 8 (or 8 x 16) query tile is staged as ROWS of 64 channels: one ring slot = (TW + 8) texels x 128 B
//     (128-B pieces: whole L2 lines; tools/micro/halo_read.hip: 64-B pieces do not scale with the bytes in flight, 128-B
//     pieces do).  A tile is four BLOCKS of HR rows - K channels 0-63, K 64-127, V 0-63, V 64-127 - and the ring holds three
//     blocks (144 KB): one block is being multiplied while up to two (96 KB per CU) are in flight;
//   * NPW PRODUCER wavefronts issue every LDS-DMA instruction (global_load_lds_dwordx4, 1 KB each; producer p owns the rows
//     r = p mod NPW of every block), texels outside the map read a zero line, the LDS swizzle of the second generation is
//     applied to the address each lane fetches.  A producer publishes `landed[p]` = the number of blocks whose rows it has
//     in LDS after a COUNTED s_waitcnt vmcnt - the oldest block, not the youngest - and refills a block's buffer as soon as
//     the eight consumers' `done[w]` counters say all of them are past it.  (A first version synchronised per ROW: two
//     LDS round trips of flag traffic per row made the producers the bottleneck - 122 us with 2 producers, 46 us with 8.)
//   * eight CONSUMER wavefronts (8 x 2 queries each, the row-pair MFMA tiles and the soft-max of the second generation,
//     bit-identical results) never touch the vector-memory path except for their 4 query loads and 8 output stores per
//     tile; they wait on `landed` with an LDS poll only when the producers are not ahead, and never on each other.
//     Flags are plain LDS words (one writer each, monotonic): no atomics, no s_barrier after the prologue.
//   * 10 waves at <= 168 VGPRs; every spin is bounded and a spin that gives up is LOUD: it bumps `di_ring_timeouts`, a producer
//     that gave up issues no further DMA (it never overwrites a block a consumer may still read), and a consumer that gave up
//     stores NaN for that tile and every later tile of its workgroup - a stuck pipeline can neither hang the device nor hand
//     back plausible numbers (tests: DI_RING_DBG=32 kills the producers after their first tile).
#include <stdlib.h>
#include <type_traits>

#include <hip/hip_ext.h>

#include "di_common.h"

namespace di {
namespace ring {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(128))) unsigned int zero_line[32];   // what a texel outside the map reads (128 B)
__device__ unsigned int timeouts;                                       // bounded spins that gave up (0 on a healthy run)

template <int WX_, int WY_, int SCHED_ = 0, int NPW_ = 2, int POL_ = 0, int POISON_ = 3, int SUB_ = 0>
struct Cfg {
  // SUB > 0: the ragged LAST round of tiles of an XCD's range is cut into sub-tiles of SUB WAVE-ROWS (2 SUB query rows x TW:
  // the halo rows 2 SUB s .. 2 SUB s + 2 SUB + 7 of the tile, multiplied by the consumer waves of those wave-rows) spread
  // over all of the XCD's workgroups, when that makes it shorter: 1 092 tiles on 256 workgroups are 4 full rounds + 68
  // tiles = 272 sub-tiles of one wave-row, one per workgroup, instead of a fifth round of full tiles on 68 of them.
  // MEASURED AND NOT KEPT (round 5, profiles/r05r_ring_subtiles.txt): bit-identical, and 39.3 us (SUB 1) / 36.5 us (SUB 2)
  // against 36.3 us - the launch is not quantised in rounds (the 68 workgroups of the fifth round have the L1 / L2 paths
  // of their XCDs to themselves and finish it in a fraction of a round), while sub-tiles re-read their halos: 2.5 x / 1.5 x
  // the bytes of the ragged round through the vector L1.  Default 0.
  static constexpr int SUB = NPW_ == 2 ? SUB_ : 0;
  static constexpr int NSUB = SUB > 0 ? WY_ / SUB : 1;   // sub-tiles of a tile
  static constexpr int SUBROWS = 2 * SUB + 8;            // halo rows of a sub-tile
  static constexpr int WX = WX_, WY = WY_, POL = POL_;    // POL: cache policy of the halo DMA (measurement)
  static constexpr bool POISON = (POISON_ & 1) != 0;   // consumers poison their stores; 0: measurement only (round 4: a timed-out wait computes on)
  static constexpr bool PDEAD = (POISON_ & 2) != 0;    // producers stop issuing after a timeout
  static constexpr int SCHED = SCHED_;                 // 1: LDS fragment reads interleaved with the MFMAs by sched_group_barrier
  static constexpr int NCW = WX * WY, NPW = NPW_;        // consumer / producer wavefronts
  static constexpr int NT = (NCW + NPW) * 64;
  static constexpr int TW = 8 * WX, TH = 2 * WY, HC = TW + 8, HR = TH + 8;
  static constexpr int S = 128;                        // bytes of a texel slice (64 channels)
  static constexpr int ROWB = HC * S, IPR = ROWB / 1024;   // one DMA instruction = 8 texels
  static constexpr int BLKB = HR * ROWB, NBLK = 3, NSLOT = NBLK * HR;
  static constexpr int BPT = 4;                        // blocks per tile: K lo, K hi, V lo, V hi
  static constexpr int RING = NBLK * BLKB;
  static constexpr int TSB = 768;                      // measurement (dbg & 16): 96 time stamps per wavefront of workgroup 0
  static constexpr int LDS_BYTES = RING + 64 + (NCW + NPW_) * TSB;   // + landed[NPW <= 8] at RING, done[8] at RING + 32, the time stamps
  static constexpr int RPB = HR / NPW_, IPB = RPB * IPR;  // rows / DMA instructions of one producer per block
  static constexpr int IPB2 = (SUBROWS / NPW_) * IPR;    // the same for a sub-tile block (NPW = 2: half of its rows per producer)
  static_assert(NCW == 8 && HR % NPW == 0 && (NPW == 2 || NPW == 4 || NPW == 8) && HC % 8 == 0 && LDS_BYTES <= 160 * 1024,
                "ring geometry");
};

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

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

// ---- LDS flag words and the DMA instruction.  All inline assembly on purpose: the compiler must neither reorder its own
// LDS accesses across them ("memory") nor learn about the DMA (it would put vmcnt(0) in front of every LDS read).
__device__ __forceinline__ unsigned long long lds_ld64(unsigned addr) {
  unsigned long long v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ uint4 lds_ld128(unsigned addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_st32(unsigned addr, unsigned val) {
  asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(val) : "memory");
}
__device__ __forceinline__ void dma16(const void *gp, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gp), "s"(lds_addr) : "memory", "m0");
}
// the same as a group-scope load (sc0): the halo has no reuse in the vector L1 (compile-time choice: a run-time switch per
// instruction costs the producers 20 us)
template <int POL>
__device__ __forceinline__ void dma16_policy(const void *gp, unsigned lds_addr) {
  if constexpr (POL == 2)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc0" : : "v"(gp), "s"(lds_addr) : "memory", "m0");
  else
    dma16(gp, lds_addr);
}
__device__ __forceinline__ unsigned lds_addr_of(const void *p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}
// at most n blocks' worth (IPB instructions each) of this wave's DMA still in flight
template <int IPB>
__device__ __forceinline__ void wait_blocks(int n) {
  static_assert(2 * IPB <= 63, "vmcnt is 6 bits");
  if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(IPB) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * IPB) : "memory");
}

// at most f full blocks' + s sub-tile blocks' worth of this wave's DMA still in flight (f + s <= 2)
template <int IPB, int IPB2>
__device__ __forceinline__ void wait_mix(int f, int s) {
  static_assert(2 * IPB <= 63, "vmcnt is 6 bits");
  switch (f * 3 + s) {
    case 1: asm volatile("s_waitcnt vmcnt(%0)" : : "n"(IPB2) : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * IPB2) : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(%0)" : : "n"(IPB) : "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(%0)" : : "n"(IPB + IPB2) : "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * IPB) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

constexpr int SPIN_LIMIT = 1 << 20;

template <class G>
__global__ __launch_bounds__(G::NT, 1) void local_attn_ring_kernel(
    const __half *__restrict__ q, const __half *__restrict__ k, const __half *__restrict__ v,
    __half *__restrict__ out, int n, int H, int W, float scale, int tiles_x, int tiles_y, int dbg,
    unsigned long long *__restrict__ ts_out) {
  // dbg (measurement only, DI_RING_DBG): 1 = no output stores, 2 = no query loads, 4 = no DMA (blocks announced at once),
  // 8 = consumers skip the LDS reads and MFMAs, 16 = phase time stamps of workgroup 0 (tools/ring_timeline.py),
  // 32 = fault injection: the producers give up after their first tile and every spin is short (tests of the NaN poisoning)
  extern __shared__ __align__(1024) unsigned char lds[];
  constexpr int ROWB = G::ROWB, S = G::S, HR = G::HR;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = lds_addr_of(lds);
  const unsigned f_landed = lds0 + G::RING, f_done = lds0 + G::RING + 32;
  const int spin_limit = (dbg & 32) ? 4096 : SPIN_LIMIT;
  // measurement: lane 0 of every wave of workgroup 0 stamps the shader clock into its own KB of LDS (no vector-memory traffic:
  // the producers count theirs); copied out at the end
  const bool ts_on = (dbg & 16) && blockIdx.x == 0;
  int tsi = 0;
  const unsigned ts_base = lds0 + G::RING + 64 + wave * G::TSB;
  auto stamp = [&](unsigned tag) {
    if (ts_on && tsi < G::TSB / 8 - 1) {
      const unsigned long long c = __builtin_readcyclecounter();
      asm volatile("ds_write_b64 %0, %1" : : "v"(ts_base + 8 * tsi), "v"((c & 0x00FFFFFFFFFFFFFFull) | ((unsigned long long)tag << 56)) : "memory");
      ++tsi;
    }
  };
  auto dump_ts = [&]() {
    if (ts_on && ts_out != nullptr) {
      asm volatile("ds_write_b64 %0, %1" : : "v"(ts_base + G::TSB - 8), "v"((unsigned long long)tsi) : "memory");
      if (lane < G::TSB / 16) {
        const uint4 v4 = lds_ld128(ts_base + lane * 16);
        reinterpret_cast<uint4 *>(ts_out)[wave * 64 + lane] = v4;
      }
    }
  };

  // ---- tiles: XCD x (workgroups with blockIdx % 8 == x share an L2) owns the contiguous range [T*x/8, T*(x+1)/8) and
  // walks it `gxw` tiles per round
  const int per_img = tiles_x * tiles_y, ntiles = n * per_img;
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
  const int gxw = ((int)gridDim.x - xcd + 7) >> 3;
  const int t_end = (int)(((long long)ntiles * (xcd + 1)) >> 3);
  const int t_beg = (int)(((long long)ntiles * xcd) >> 3);
  const int tile0 = t_beg + wl;
  // work items of this workgroup: `nfull` full tiles (tile0 + it * gxw), then either one more full tile (the ragged round as
  // it is) or its share of the ragged round's tiles cut into wave-row sub-tiles (item = tile, wave-row q)
  const int rounds = (t_end - t_beg) / gxw, left = (t_end - t_beg) - rounds * gxw;     // full rounds, tiles of the ragged one
  const bool cut = G::SUB > 0 && left > 0 && G::NSUB * left <= 2 * gxw;
  const int nsub = cut ? (G::NSUB * left - wl + gxw - 1) / gxw : 0;                      // sub-tiles wl, wl + gxw, ...
  const int nfull = rounds + ((!cut && wl < left) ? 1 : 0);
  const int ntl = nfull + (nsub > 0 ? nsub : 0);
  if (ntl <= 0) return;
  const int NF = nfull * G::BPT;                              // blocks of full tiles come first
  struct Item {
    TileCoord t;
    int q;                                                    // sub-tile (wave-rows q SUB .. q SUB + SUB - 1), -1: the whole tile
  };
  auto item = [&](int it) {
    Item w;
    if (it < nfull) {
      w.t = decode_tile(tile0 + it * gxw, tiles_x, per_img, G::TH, G::TW);
      w.q = -1;
    } else {
      const int sidx = wl + (it - nfull) * gxw;               // sub-tile index inside the ragged round
      w.t = decode_tile(t_beg + rounds * gxw + sidx / G::NSUB, tiles_x, per_img, G::TH, G::TW);
      w.q = sidx % G::NSUB;
    }
    return w;
  };

  if (tid < 16) lds_st32(f_landed + 4 * tid, 0);             // landed[0..1] (+ padding), done[0..7]
  __syncthreads();                                           // the only barrier of the kernel

  if (wave >= G::NCW) {
    // ------------------------------------------------------------------------------------------------ producer
    const int p = wave - G::NCW;
    // lane -> texel lane / 8 of the instruction's 8, LDS chunk position lane % 8 <- the global chunk that belongs there
    const int d_t = lane >> 3, d_pos = lane & 7;
    const int d_f = (d_t >> 1) & 3;                           // (texel / 2) % 4: 8 j + d_t has the same value for every j
    const int d_c16 = (((d_pos >> 1) ^ d_f) << 1) | (d_pos & 1);
    const int d_off = d_t * 256 + d_c16 * 16;
    const unsigned char *const zsrc = reinterpret_cast<const unsigned char *>(zero_line) + d_pos * 16;
    const int nblk = ntl * G::BPT;                            // blocks of this workgroup, in sequence
    int min_done = 0;                                         // cached min over done[w]: blocks every consumer is past
    int published = 0;                                        // blocks of mine announced in landed[p]
    int issued = 0;                                           // blocks whose DMA this wave has issued
    __builtin_amdgcn_s_setprio(2);                            // few instructions, all of them on the critical path
    auto publish = [&]() {
      lds_st32(f_landed + 4 * p, (unsigned)published);
      stamp(3);                                               // a block announced
    };
    auto reload_done = [&]() {
      const uint4 a = lds_ld128(f_done), b = lds_ld128(f_done + 16);
      const unsigned m = min(min(min(a.x, a.y), min(a.z, a.w)), min(min(b.x, b.y), min(b.z, b.w)));
      min_done = __builtin_amdgcn_readfirstlane((int)m);
    };
    // the oldest unannounced block Bo has landed once at most its n younger blocks' instructions are in flight
    auto wait_for = [&](int Bo, int nyoung) {
      const int f = min(max(NF - (Bo + 1), 0), nyoung);
      wait_mix<G::IPB, G::IPB2>(f, nyoung - f);
    };
    for (int it = 0; it < ntl; ++it) {
      const Item wi = item(it);
      const TileCoord t = wi.t;
      const int r_lo = wi.q < 0 ? 0 : 2 * G::SUB * wi.q, r_hi = wi.q < 0 ? HR : r_lo + G::SUBROWS;   // halo rows this item needs
      const bool interior = t.y0 >= 4 && t.x0 >= 4 && t.y0 - 4 + HR <= H && t.x0 - 4 + G::HC <= W;
      const long long tile_off = ((long long)(t.img * H + t.y0 - 4) * W + (t.x0 - 4)) * 256 + d_off;
      bool x_ok[G::IPR];
#pragma unroll
      for (int j = 0; j < G::IPR; ++j) {
        const int gx = t.x0 - 4 + 8 * j + d_t;
        x_ok[j] = gx >= 0 && gx < W;
      }
#pragma unroll
      for (int b = 0; b < G::BPT; ++b) {
        const int B = it * G::BPT + b;
        // the buffer of block B held block B - 3: free once every consumer is past it
        int spins = 0;
        if (B - G::NBLK >= min_done) {                        // (scalar: the buffer is usually free - skip the wait AND its give-up test)
        while (B - G::NBLK >= min_done) {
          if (published < B) {                                // meanwhile: announce the oldest block of mine that is in flight
            wait_for(published, B - published - 1);
            ++published;
            publish();
          } else {
            __builtin_amdgcn_s_sleep(1);
          }
          reload_done();
          if (++spins > spin_limit) {
            if (lane == 0) atomicAdd(&timeouts, 1u);
            break;
          }
        }
        // (decided AFTER the loop from a value forced uniform: a flag assigned inside the timeout branch is a divergent
        // live-out of the loop for the compiler, which then runs the whole producer under exec masks - 3 us per launch)
        // a producer that gave up issues nothing more: it never writes into a block a consumer may still be reading.
        // ((dbg & 32): fault injection - the consumers' waits then give up and count)
        if constexpr (G::PDEAD)
          if (__builtin_amdgcn_readfirstlane((int)(spins > spin_limit)) != 0) goto drain;
        }
        if constexpr (G::PDEAD)
          if ((dbg & 32) && it >= 1) goto drain;             // fault injection (tests)
        stamp(1);                                             // buffer free
        const unsigned dst = lds0 + (B % G::NBLK) * G::BLKB;
        {
        const int bb = b;
        const unsigned char *src = reinterpret_cast<const unsigned char *>(bb < 2 ? k : v) + tile_off + (bb & 1) * 128;
#pragma unroll
        for (int rr = 0; rr < G::RPB; ++rr) {
          const int r = p + rr * G::NPW;
          if (G::SUB > 0 && (r < r_lo || r >= r_hi)) continue;   // (wave-uniform) a sub-tile takes SUBROWS of the halo rows
          const unsigned char *row = src + (long long)r * W * 256;
          const int gy = t.y0 - 4 + r;
          const bool y_ok = gy >= 0 && gy < H;
#pragma unroll
          for (int j = 0; j < G::IPR; ++j) {
            const unsigned char *gp = row + j * 2048;
            if (!interior) gp = (y_ok && x_ok[j]) ? gp : zsrc;
            if (!(dbg & 4)) dma16_policy<G::POL>(gp, __builtin_amdgcn_readfirstlane(dst + r * ROWB + j * 1024));
          }
        }
        }
        stamp(2);                                             // issued
        issued = B + 1;
        if (B - published >= 2) {                             // never more than two blocks unannounced (vmcnt is 6 bits)
          wait_for(B - 2, 2);
          published = B - 1;
          publish();
        }
      }
    }
  drain:
    while (published < issued) {                              // drain: no DMA may outlive the workgroup's LDS
      wait_for(published, issued - published - 1);
      ++published;
      publish();
    }
    dump_ts();
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumer
  const int wx = wave % G::WX, wy = wave / G::WX;
  const int i = lane & 15, g = lane >> 4;
  const int j = i & 7, qrow = i >> 3;
  // ---- fragment constants (the LDS image of local_attn_mfma2.hip with 128-byte slices: 32-B segments XOR-swizzled by
  // (texel / 2) % 4)
  const int hcq = wx * 8 + i;                                // K fragment: key column i of the wave's 16
  const int fq = (hcq >> 1) & 3;
  int koff[2];
#pragma unroll
  for (int kl = 0; kl < 2; ++kl) koff[kl] = wy * 2 * ROWB + hcq * S + ((((kl * 4 + g) >> 1) ^ fq) << 5) + (((kl * 4 + g) & 1) << 4);
  const int kcv = wx * 8 + 4 * g + (i >> 2);                 // V^T fragment: key column addressed by this lane
  const int vsw = (kcv >> 1) & 3;
  const int vbase = wy * 2 * ROWB + kcv * S + (i & 3) * 8;
  // additive softmax masks: 0 where key c = 4g + r lies in the band of query column j (j <= c <= j + 8) and the key row
  // belongs to the window of the query's row, -inf elsewhere
  const float cs = scale * 1.44269504088896f;                // scores in log2 units
  f4 nm_mid, nm_first, nm_last;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool in_band = 4 * g + r >= j && 4 * g + r <= j + 8;
    nm_mid[r] = in_band ? 0.f : -INFINITY;
    nm_first[r] = (in_band && qrow == 0) ? 0.f : -INFINITY;  // key row 0: only the upper query row
    nm_last[r] = (in_band && qrow == 1) ? 0.f : -INFINITY;   // key row 9: only the lower query row
  }

  // block B is in LDS once every producer has announced more than B blocks
  int seen = 0;                                              // blocks [0, seen) are known to have landed
  // a wait that gave up: `seen` jumps past every block (no later wait spins again) and `poison` turns every value this wave
  // stores from then on into NaN (ORed into the bits of 1 / sum: no branch on the store path, the MFMA / store schedule is untouched)
  unsigned poison = 0;
  auto wait_landed = [&](int B) {
    int spins = 0;
    stamp(4);                                                 // starts waiting for a block
    // (round 6: the whole wait - and its give-up bookkeeping - behind ONE scalar test: with the flags read ahead the block has
    //  usually landed, and the three scalar selects of the fault model ran 20 times per 5 tiles for nothing: 37.0 -> us)
    if (seen <= B) {
    while (seen <= B) {
      unsigned m;
      if constexpr (G::NPW == 2) {
        const unsigned long long v2 = lds_ld64(f_landed);
        m = min((unsigned)v2, (unsigned)(v2 >> 32));
      } else {
        const uint4 a = lds_ld128(f_landed);
        m = min(min(a.x, a.y), min(a.z, a.w));
        if constexpr (G::NPW == 8) {
          const uint4 b = lds_ld128(f_landed + 16);
          m = min(m, min(min(b.x, b.y), min(b.z, b.w)));
        }
      }
      seen = __builtin_amdgcn_readfirstlane((int)m);
      if (seen > B) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > spin_limit) {
        if (lane == 0) atomicAdd(&timeouts, 1u);
        break;
      }
    }
    if constexpr (G::POISON) {                                // (after the loop, from a value forced uniform: see the producer)
      const bool gave_up = __builtin_amdgcn_readfirstlane((int)(spins > spin_limit)) != 0;
      poison = gave_up ? 0x7FFF7FFFu : poison;
      seen = gave_up ? 0x3FFFFFFF : seen;
    }
    }
    stamp(5);                                                 // has it
  };
  // the flags read AHEAD: issued before a block's MFMA loop, looked at after it - when the next block has landed meanwhile
  // (the usual case) wait_landed() returns without an LDS round trip (~300 clocks on a busy LDS, 20 times per 5 tiles)
  unsigned long long peek0 = 0, peek1 = 0;
  auto peek_begin = [&]() {
    if constexpr (G::NPW == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(peek0) : "v"(f_landed) : "memory");
  };
  auto peek_end = [&]() {
    if constexpr (G::NPW == 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(peek0) : : "memory");
      const unsigned m = min((unsigned)peek0, (unsigned)(peek0 >> 32));
      seen = max(seen, __builtin_amdgcn_readfirstlane((int)m));
    }
  };
  auto release = [&](int nb) {
    lds_st32(f_done + 4 * wave, (unsigned)nb);                // this wave is past blocks [0, nb)
    stamp(6);
  };

  // Q^T fragments (query i, channels kk*32 + 8g .. +7) straight from global; queries beyond the map edge (ragged tiles)
  // read a clamped texel, their results are never stored
  h8 qf[4];
  auto load_q = [&](const TileCoord &t) {
    const int gy = min(t.y0 + 2 * wy + qrow, H - 1), gx = min(t.x0 + 8 * wx + j, W - 1);
    const unsigned char *qb = reinterpret_cast<const unsigned char *>(q) + ((unsigned)((t.img * H + gy) * W + gx) << 8) + g * 16;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = __builtin_bit_cast(h8, *reinterpret_cast<const uint4 *>(qb + kk * 64));
  };

  Item wcur = item(0);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = h8{1, 1, 1, 1, 1, 1, 1, 1};
  int q_for = -1;                                            // the item qf holds the queries of
  for (int it = 0; it < ntl; ++it) {
    const bool has_next = it + 1 < ntl;
    Item wnxt = wcur;
    if (has_next) wnxt = item(it + 1);
    const TileCoord cur = wcur.t, nxt = wnxt.t;
    const int B0 = it * G::BPT;
    // a sub-tile belongs to the consumer waves of SUB wave-rows; the others are past its four blocks at once
    if (G::SUB > 0 && wcur.q >= 0 && wcur.q != wy / (G::SUB > 0 ? G::SUB : 1)) {
      release(B0 + G::BPT);
      wcur = wnxt;
      continue;
    }
    const bool next_mine = has_next && (wnxt.q < 0 || wnxt.q == wy / (G::SUB > 0 ? G::SUB : 1));
    if (q_for != it && !(dbg & 2)) load_q(cur);

    // ---------------- S^T = K . Q^T over the two K blocks
    f4 s[10];
#pragma unroll
    for (int rr = 0; rr < 10; ++rr) s[rr] = f4{0.f, 0.f, 0.f, 0.f};
    static_for<0, 2>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const int B = B0 + u;
      wait_landed(B);
      if constexpr (G::SCHED == 2) peek_begin();
      const unsigned char *buf = lds + (B % G::NBLK) * G::BLKB;
      if (!(dbg & 8))
#pragma unroll
      for (int kl = 0; kl < 2; ++kl) {                       // k-step outer: an accumulator's two MFMAs are 10 apart
#pragma unroll
        for (int rr = 0; rr < 10; ++rr) {
          const uint4 raw = *reinterpret_cast<const uint4 *>(buf + koff[kl] + rr * ROWB);
          s[rr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, raw), qf[u * 2 + kl], s[rr], 0, 0, 0);
        }
      }
      if constexpr (G::SCHED >= 1) {                         // AH reads ahead, then one read per MFMA (a wave's LDS reads return
        constexpr int AH = G::SCHED == 1 ? 6 : 12;            // one per ~50 clocks with 6 in flight: the block was LDS-latency bound)
        __builtin_amdgcn_sched_group_barrier(0x100, AH, 0);
#pragma unroll
        for (int e = 0; e < 20 - AH; ++e) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, AH, 0);
      }
      if constexpr (G::SCHED == 2) peek_end();
      release(B + 1);
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
    // (requested a whole tile ahead into a second register set the launch is 1.5 us SLOWER, 37.0 against 35.5 us: the requests
    // then share the L1 queue with the producers' DMA for longer)
    if (next_mine && !(dbg & 2)) {                                         // qf is dead until the next tile's first block
      load_q(nxt);
      q_for = it + 1;
    }

    // ---------------- O^T = V^T . P^T over the two V blocks, each block finishes 64 output channels
    const float inv = 1.f / sum;
    const int gy = cur.y0 + 2 * wy + qrow, gx = cur.x0 + 8 * wx + j;
    const bool pix_ok = gy < H && gx < W;
    __half *dst = out + ((long long)(cur.img * H + gy) * W + gx) * 128 + (g & 1) * 16 + (g >> 1) * 8;   // see the stores
    static_for<0, 2>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const int B = B0 + 2 + u;
      wait_landed(B);
      if constexpr (G::SCHED == 2) peek_begin();
      const unsigned char *buf = lds + (B % G::NBLK) * G::BLKB;
      f4 acc[4];
#pragma unroll
      for (int nl = 0; nl < 4; ++nl) acc[nl] = f4{0.f, 0.f, 0.f, 0.f};
      if (!(dbg & 8))
#pragma unroll
      for (int pr = 0; pr < 5; ++pr) {
#pragma unroll
        for (int nl = 0; nl < 4; ++nl) {
          const unsigned char *p0 = buf + vbase + ((nl ^ vsw) << 5) + 2 * pr * ROWB;
          const hv4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(p0));
          const hv4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(p0 + ROWB));
          h8 a;
          a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3];
          a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
          acc[nl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pf[pr], acc[nl], 0, 0, 0);
        }
      }
      if constexpr (G::SCHED >= 1) {                         // 2 AH transposed reads ahead, then two per MFMA
        constexpr int AH = G::SCHED == 1 ? 4 : 9;
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * AH, 0);
#pragma unroll
        for (int e = 0; e < 20 - AH; ++e) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, AH, 0);
      }
      if constexpr (G::SCHED == 2) peek_end();
      release(B + 1);
      // The accumulators hold, per 16-channel block nl, channels 16 nl + 4g .. + 3 of pixel i: stored as they are, an
      // instruction writes 32-byte pieces (4 instructions per block of 64 channels).  v_permlane16_swap exchanges the odd
      // 16-lane rows of one register with the even rows of another - a 2 x 2 transposition between (g & 1) and the block
      // pair - after which a lane owns 8 consecutive channels: two 16-byte stores, each 64 contiguous bytes per pixel
      // (38.1 -> 35.5 us on cold inputs; without any store 31.4).
      // (a block that never arrived: 1 / sum becomes NaN, and with it every value of the tile - poisoned, not guessed)
      const float inv_b = G::POISON ? __builtin_bit_cast(float, __builtin_bit_cast(unsigned, inv) | poison) : inv;
      unsigned wv[4][2];
#pragma unroll
      for (int nl = 0; nl < 4; ++nl) {
        const f4 o = acc[nl] * inv_b;
        h4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (_Float16)o[r];
        const uint2 raw = __builtin_bit_cast(uint2, ov);
        wv[nl][0] = raw.x;
        wv[nl][1] = raw.y;
      }
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const u2v sw = __builtin_amdgcn_permlane16_swap(wv[2 * pr][d], wv[2 * pr + 1][d], false, false);
          wv[2 * pr][d] = sw[0];
          wv[2 * pr + 1][d] = sw[1];
        }
      if (pix_ok && !(dbg & 1)) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
          *reinterpret_cast<uint4 *>(dst + u * 64 + 32 * pr) =
              make_uint4(wv[2 * pr][0], wv[2 * pr][1], wv[2 * pr + 1][0], wv[2 * pr + 1][1]);
      }
    });
    wcur = wnxt;
  }
  dump_ts();
}

// measurement (DI_RING_DBG & 16): 16 waves x 128 stamps, dumped by `di_local_attn_ring_stamps`
static unsigned long long *ts_buffer(int dbg) {
  static unsigned long long *buf = nullptr;
  if ((dbg & 16) && buf == nullptr) {
    if (hipMalloc((void **)&buf, 16 * 128 * 8) != hipSuccess) buf = nullptr;
    else (void)hipMemset(buf, 0, 16 * 128 * 8);
  }
  return buf;
}

template <class G>
static int launch(const void *q, const void *k, const void *v, void *out, int n, int H, int W, float scale,
                  hipStream_t stream) {
  const int tiles_x = (W + G::TW - 1) / G::TW, tiles_y = (H + G::TH - 1) / G::TH;
  const long long ntiles = (long long)n * tiles_x * tiles_y;
  DI_REQUIRE((long long)n * H * W * 256 < (1ll << 31), "map of %d x %d x %d texels exceeds the 2 GiB offset range", n, H, W);
  static LdsRaised lds_raised;
  if (int rc = ensure_lds(lds_raised, (const void *)local_attn_ring_kernel<G>, G::LDS_BYTES)) return rc;
  const int n_cu = device_cus();
  if (n_cu <= 0) return DI_ERR_LAUNCH;
  long long grid = n_cu;                                     // one workgroup per CU, a multiple of the 8 XCDs
  if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
  static const int grid_env = getenv("DI_LA_GRID") ? atoi(getenv("DI_LA_GRID")) : 0;
  if (grid_env > 0 && grid_env < grid) grid = grid_env / 8 * 8;
  static const int dbg_env = getenv("DI_RING_DBG") ? atoi(getenv("DI_RING_DBG")) : 0;
  hipEvent_t ev0, ev1;
  if (take_launch_events(ev0, ev1))                          // measurement: the dispatch's own begin / end time stamps
    hipExtLaunchKernelGGL(local_attn_ring_kernel<G>, dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, stream, ev0, ev1, 0,
                          (const __half *)q, (const __half *)k, (const __half *)v, (__half *)out, n, H, W, scale, tiles_x,
                          tiles_y, dbg_env, ts_buffer(dbg_env));
  else
    hipLaunchKernelGGL(local_attn_ring_kernel<G>, dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, stream,
                       (const __half *)q, (const __half *)k, (const __half *)v, (__half *)out, n, H, W, scale, tiles_x,
                       tiles_y, dbg_env, ts_buffer(dbg_env));
  return check_launch("local_attn_ring");
}

}  // namespace ring

// cfg 0: 16 x 8 query tiles (halo 24 x 16), 12-deep LDS read-ahead, flags read ahead (the AUTO choice for large maps);
// cfg 1: 8 x 16 tiles (halo 16 x 24: 112 x 200 maps leave no ragged tile); cfg 2-4: measurement variants
int launch_local_attn_ring(const void *q, const void *k, const void *v, void *out, int n, int H, int W, float scale,
                           int cfg, hipStream_t stream) {
  switch (cfg) {
    // (last parameter: cache policy of the halo DMA.  2 = sc0, 35.3 us on cold inputs against 36.4 for plain loads, 35.6 for sc1 and
    // 38.6-38.7 for nt / sc0 nt on one box)
    case 0: return ring::launch<ring::Cfg<2, 4, 2, 2, 2>>(q, k, v, out, n, H, W, scale, stream);
    case 1: return ring::launch<ring::Cfg<1, 8, 2, 2, 2>>(q, k, v, out, n, H, W, scale, stream);
    case 2: return ring::launch<ring::Cfg<2, 4, 1, 2, 2>>(q, k, v, out, n, H, W, scale, stream);   // shallower LDS read-ahead, no flag peek
    case 3: return ring::launch<ring::Cfg<2, 4, 0, 2, 2>>(q, k, v, out, n, H, W, scale, stream);   // the compiler's own schedule
    case 4: return ring::launch<ring::Cfg<2, 4, 2, 4, 2>>(q, k, v, out, n, H, W, scale, stream);   // four producer wavefronts
    case 5: return ring::launch<ring::Cfg<2, 4, 2, 2, 0>>(q, k, v, out, n, H, W, scale, stream);   // plain DMA loads
    case 6: return ring::launch<ring::Cfg<2, 4, 2, 2, 2, 3, 1>>(q, k, v, out, n, H, W, scale, stream);   // cfg 0 with the ragged round cut into sub-tiles of one wave-row (A/B: slower)
    case 7: return ring::launch<ring::Cfg<2, 4, 2, 2, 2, 3, 2>>(q, k, v, out, n, H, W, scale, stream);   // ... and as sub-tiles of two wave-rows
    case 8: return ring::launch<ring::Cfg<1, 8, 2, 2, 2, 3, 1>>(q, k, v, out, n, H, W, scale, stream);   // cfg 1 with one-wave-row sub-tiles
    // round 6 (36.1 us in r04zl, 37.9-38.0 us in r05 - the fault model?): cfg 0 without it, piece by piece
    case 9: return ring::launch<ring::Cfg<2, 4, 2, 2, 2, 0>>(q, k, v, out, n, H, W, scale, stream);    // no poisoning, producers keep issuing (round 4's behaviour)
    case 10: return ring::launch<ring::Cfg<2, 4, 2, 2, 2, 1>>(q, k, v, out, n, H, W, scale, stream);   // consumers poison, producers keep issuing
    case 11: return ring::launch<ring::Cfg<2, 4, 2, 2, 2, 2>>(q, k, v, out, n, H, W, scale, stream);   // producers stop, consumers do not poison
  }
  set_error("unknown local_attn_ring configuration %d", cfg);
  return DI_ERR_ARG;
}

int ring_stamps(unsigned long long *host_out, hipStream_t stream) {
  unsigned long long *buf = ring::ts_buffer(16);
  if (buf == nullptr) return DI_ERR_LAUNCH;
  if (hipStreamSynchronize(stream) != hipSuccess) return DI_ERR_LAUNCH;
  return hipMemcpy(host_out, buf, 16 * 128 * 8, hipMemcpyDeviceToHost) == hipSuccess ? DI_OK : DI_ERR_LAUNCH;
}

// the same without a synchronisation: the counter is copied into the caller's (pinned) host word behind everything queued on
// `stream`; the caller looks at it whenever it likes (the value is monotonic)
int ring_timeouts_async(unsigned *host_out, hipStream_t stream) {
  unsigned *dev = nullptr;
  if (hipGetSymbolAddress((void **)&dev, HIP_SYMBOL(ring::timeouts)) != hipSuccess) return DI_ERR_LAUNCH;
  return hipMemcpyAsync(host_out, dev, sizeof(unsigned), hipMemcpyDeviceToHost, stream) == hipSuccess ? DI_OK : DI_ERR_LAUNCH;
}

// bounded spins that gave up since the library was loaded (tests: must stay 0)
int ring_timeouts(unsigned *host_out, hipStream_t stream) {
  unsigned *dev = nullptr;
  if (hipGetSymbolAddress((void **)&dev, HIP_SYMBOL(ring::timeouts)) != hipSuccess) return DI_ERR_LAUNCH;
  if (hipMemcpyAsync(host_out, dev, sizeof(unsigned), hipMemcpyDeviceToHost, stream) != hipSuccess) return DI_ERR_LAUNCH;
  return hipStreamSynchronize(stream) == hipSuccess ? DI_OK : DI_ERR_LAUNCH;
}

}  // namespace di
