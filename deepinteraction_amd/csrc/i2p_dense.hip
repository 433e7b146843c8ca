// The pillar attention (image -> BEV, MMRI_I2P.forward + group_attn + nn.MultiheadAttention, reference
// encoder_utils.py:226-320) ON THE MATRIX CORES - round 6.  fp16 maps, C = 128, inference (no attention dropout); every other
// configuration keeps the wave-per-cell kernel of cross_modal.hip.
//
// Why: the wave-per-cell kernel is bound by VALU issue (PMC round 2-5: ~125 VALU instructions per round of four keys - 16
// v_dot2 for the corner scores and 32 v_fma_mix for the corner blend per lane - with the matrix pipe idle, 0.22 of the HBM
// roofline).  A cell has ONE query, so its products are matrix-vector products; but the matrix pipe does not care whether
// the 16 columns of a tile are 16 queries of one problem or the queries of 16 problems, as long as the rows line up:
//
//   * the keys of a GROUP of NC = 8 neighbouring cells (neighbours in the walk order of the sample: azimuth, then radius)
//     lie DENSELY packed in one stream (i2p_compact_kernel, once per sample on the side stream, shared by the layers): a
//     block of 4 keys = 16 RAW ROWS (4 bilinear corners each) may belong to any of the group's cells - no ragged last round
//     per cell (the wave-per-cell kernel ran 2.5 rounds of 4 for 6.8 keys per cell);
//   * S^T = F . Q^T : A = the 16 raw rows (16 x 128 channels, four k-steps of 32), B = the queries of the group's cells as
//     columns (a lane's 16 B of one query per k-step, kept in registers for the whole group) -> lane (column i, group g) gets
//     <corner r of key g, query i>, r = 0..3, in its four accumulator registers; the bilinear blend of the SCORE is four
//     FMAs with the key's weights, and only the column of the key's own cell is kept (the others are masked to -inf);
//   * O^T += F^T . P^T : the masked, exponentiated scores times the corner weights ARE the B operand of the 16x16x16 MFMA
//     (k = 4 g + r: exactly the accumulator layout, no cross-lane movement), F^T comes out of LDS through
//     ds_read_b64_tr_b16; the accumulators hold O[channel 16 blk + 4 g + r][cell i] for the 8 channel blocks;
//   * soft-max state per column with a LAZY reference maximum (rescale only when a later key beats the reference by more
//     than 2^8: the probabilities then live in [0, 256], fine for the fp16 B operand; a rescale pass is 32 multiplies and
//     almost never runs);
//   * the raw rows travel L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, a 16-lane group fetches one whole 256-B
//     row), NB blocks in flight per wavefront, each wavefront on its own 4-KB buffers: no barrier anywhere.  The 16-B slots
//     of a row are XOR-rotated by f(row) so that both the 16-B fragment reads of the score product and the transposed
//     8-B reads of the value product are bank-conflict free; the rotation is applied to the ADDRESS a lane fetches
//     (the DMA image is lane-linear);
//   * the keys of a group arrive the same way, 32 at a time (1 KB), in a two-slot ring.
//
// Per block of 4 keys: ~45 VALU instructions (was ~125), 4 + 8 MFMAs, 4 DMA instructions (as many as the 4 row loads of a
// round before).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include <hip/hip_ext.h>

#include "di_common.h"
#include "i2p_common.h"

namespace di {
namespace i2pd {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

constexpr int NC = 8;            // BEV cells per group at most (MFMA columns in use; 16 columns exist)
constexpr int kKeyCap = 48;      // a group closes before it would exceed this many keys (a single cell may: <= 120 keys); DI_I2PD_KEYCAP
constexpr int kResident = 8;      // resident wavefronts per CU (LDS: 20 KB each)
constexpr int kWavesPerCU = 16;  // shares per CU (DI_I2PD_WAVES): twice the resident wavefronts - guided shares, see share_t
constexpr int kBigSixteenths = 13;   // (DI_I2PD_BIG) measured: uniform 8 per CU 20.0 us; 16 per CU with 11 / 12 / 13 / 14 sixteenths: 20.5 / 19.4 / 18.8 / 18.9 us
constexpr int kMaxWaves = 4096;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLazy = 8.f;     // log2 units: the reference maximum is kept until a score beats it by more than this

struct DenseKey {                // 32 B
  unsigned off[4];               // byte offset of corner row r from the map base (a multiple of 256); off[0] bits 0..3: the column
  float w[4];                    // bilinear weights (0 where grid_sample's zero padding applies, and for padding keys)
};
struct GroupHdr {                // 64 B
  int cell[NC];                  // the group's BEV cells, column by column (-1: column not used)
  int sb_begin, sb_end;          // its superblocks (8 keys each) in the stream; every group has at least one
  int nk, pad[5];
};

// GROUPS.  The walk order of the cells is cut into CHUNKS of 8 positions; a chunk is cut greedily into groups of consecutive
// cells: a group closes before it would exceed kKeyCap keys (so the largest group is one crowded cell or ~6 superblocks - a
// group is the unit of load balance: the first version's fixed 8-cell groups had up to 30 superblocks, one wavefront ran
// 2.5 x the average).  Layout of the dense table (bytes):
//   group headers (<= one per cell) | per chunk: groups in front (nchunks + 1) | per chunk: superblocks in front (nchunks + 1) |
//   first group of every share (kMaxWaves + 1) | keys
__host__ __device__ inline int n_chunks(int ncell) { return (ncell + NC - 1) / NC; }
__host__ __device__ inline long long al256(long long x) { return (x + 255) / 256 * 256; }
__host__ __device__ inline long long off_csub(int ncell) { return al256((long long)ncell * (long long)sizeof(GroupHdr)); }
__host__ __device__ inline long long off_csb(int ncell) { return off_csub(ncell) + al256((long long)(n_chunks(ncell) + 1) * 4); }
__host__ __device__ inline long long off_wst(int ncell) { return off_csb(ncell) + al256((long long)(n_chunks(ncell) + 1) * 4); }
__host__ __device__ inline long long off_keys(int ncell) { return off_wst(ncell) + al256((long long)(kMaxWaves + 1) * 4); }

// SHARES.  Share w of W (W / 8 per XCD, in stream order inside the XCD's eighth of the stream) starts at the first group whose
// first superblock is >= t(w).  The first half of an XCD's shares (the workgroups that are dispatched first and fill every
// resident slot) take `big` sixteenths of its superblocks, the second half - dispatched as slots free up - the rest in small
// pieces: the small shares fill the tail that one share per resident wavefront leaves (the average wavefront lived 12.4 us of a
// 20.4 us launch: a share is cut at group boundaries, +- one group of up to 6 superblocks around a mean of 13).  big = 8:
// uniform shares.
__host__ __device__ inline unsigned share_t(unsigned w, unsigned nsb, unsigned W, unsigned big) {
  const unsigned wx = W >> 3, x = w / wx, j = w - x * wx, half = wx >> 1;
  if (x >= 8u) return nsb;
  const unsigned X0 = (unsigned)(((unsigned long long)x * nsb) >> 3), X1 = (unsigned)(((unsigned long long)(x + 1u) * nsb) >> 3);
  const unsigned len = X1 - X0, bigpart = (len * big) >> 4;
  if (half == 0u) return X0;
  return j < half ? X0 + (unsigned)(((unsigned long long)j * bigpart) / half)
                  : X0 + bigpart + (unsigned)(((unsigned long long)(j - half) * (len - bigpart)) / half);
}

// csub[c] / csb[c] = groups / superblocks of chunk c (one thread per chunk): the greedy cut, serially over the 8 cells
__global__ __launch_bounds__(256) void chunk_count_kernel(const int *__restrict__ cnt, const int *__restrict__ order,
                                                          int *__restrict__ csub, int *__restrict__ csb, int ncell, int nchunks,
                                                          int key_cap) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= nchunks) return;
  int k[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int pos = c * NC + j;
    k[j] = pos < ncell ? cnt[order != nullptr ? order[pos] : pos] : -1;
  }
  int keys = 0, cells = 0, nsub = 0, nsb = 0;
#pragma unroll
  for (int j = 0; j < NC; ++j)
    if (k[j] >= 0) {
      if (cells > 0 && keys + k[j] > key_cap) {
        nsb += max((keys + 7) >> 3, 1);
        ++nsub;
        keys = cells = 0;
      }
      keys += k[j];
      ++cells;
    }
  if (cells > 0) {
    nsb += max((keys + 7) >> 3, 1);
    ++nsub;
  }
  csub[c] = nsub;
  csb[c] = nsb;
}

// exclusive prefix sum IN PLACE of blockIdx.x-th array (a[n] = the total).  ONE workgroup per array, plain stores only (a
// captured graph replays this: no atomics, no memset).  A thread owns `per` consecutive entries (n <= 8 * 1024).
__global__ __launch_bounds__(1024) void scan_kernel(int *__restrict__ a0, int *__restrict__ a1, int n) {
  __shared__ int wsum[16];
  int *a = blockIdx.x == 0 ? a0 : a1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n + 1023) / 1024;
  int v[8];
  int mine = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int e = tid * per + k;
    v[k] = (k < per && e < n) ? a[e] : 0;
    mine += v[k];
  }
  int inc = mine;                                  // inclusive scan over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int run = base + inc - mine;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int e = tid * per + k;
    if (k < per && e < n) a[e] = run;
    run += v[k];
  }
  if (tid == 1023) a[n] = run;
}

__device__ __forceinline__ uint4 key_addr(const float4 &k0, unsigned stepy) {
  const int pix = __float_as_int(k0.x), info = __float_as_int(k0.y);
  const unsigned o00 = (unsigned)pix * 256u, dx = (info & 1) ? 256u : 0u, dy = (info & 2) ? stepy : 0u;
  return make_uint4(o00, o00 + dx, o00 + dy, o00 + dx + dy);
}

// One wavefront per chunk: the headers of its groups, their keys - cell after cell, ready to fetch (explicit corner offsets),
// tagged with the cell's column, padded to whole superblocks (padding: weight 0, column 15, the address of the group's last
// real key) - and the share table: share w of W starts at the first group whose first superblock is >= t(w) = w * all / W,
// i.e. behind the group that holds superblock t(w) - 1 (every group writes the shares that start right behind it).
// Lane j < 8 owns CELL j of the chunk, lane q < 8 owns GROUP q: the greedy cut is one serial pass over the 8 counts (wave
// uniform), whose results land in the owning lanes by compare-with-lane-id; everything after it is lane-parallel.  (The first
// version kept the cut in uniform arrays indexed by compile-time loops: 2 700 instructions, 13 us for the 4 050 chunks.)
__global__ __launch_bounds__(256) void compact_kernel(const int *__restrict__ cnt, const KeyEnt *__restrict__ keys,
                                                      const int *__restrict__ order, const int *__restrict__ csub,
                                                      const int *__restrict__ csb, GroupHdr *__restrict__ hdr,
                                                      int *__restrict__ wst, DenseKey *__restrict__ dense, int ncell, int nchunks,
                                                      int nslots, int Wi, int W, int key_cap, int big) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= nchunks) return;
  const int ncells = min(NC, ncell - c * NC);
  int mycell = -1, mycnt = 0;
  if (lane < ncells) {
    mycell = order != nullptr ? order[c * NC + lane] : c * NC + lane;
    mycnt = cnt[mycell];
  }
  const int G0 = csub[c], ngroups = csub[nchunks], nsb_all = csb[nchunks], sb_first = csb[c];
  // the cut
  int gkeys = 0, cells = 0, nsub = 0;
  int my_sub = 0, my_col = 0, my_in = 0;                 // lane j: its cell's group, column, keys in front of it in the group
  int g_nk = 0, g_nc = 0, g_lastcell = -1, g_lastcnt = 0;   // lane q: its group's keys, cells, last cell that has keys
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int cj = __builtin_amdgcn_readlane(mycnt, j), cellj = __builtin_amdgcn_readlane(mycell, j);
    if (j < ncells) {
      if (cells > 0 && gkeys + cj > key_cap) {
        ++nsub;
        gkeys = cells = 0;
      }
      if (lane == j) {
        my_sub = nsub;
        my_col = cells;
        my_in = gkeys;
      }
      if (lane == nsub) {
        g_nk += cj;
        g_nc += 1;
        if (cj > 0) {
          g_lastcell = cellj;
          g_lastcnt = cj;
        }
      }
      gkeys += cj;
      ++cells;
    }
  }
  nsub += ncells > 0 ? 1 : 0;
  // superblocks of the groups: lane q
  const int g_nsb = lane < nsub ? max((g_nk + 7) >> 3, 1) : 0;
  int inc = g_nsb;                                        // inclusive prefix over lanes 0 .. 7
#pragma unroll
  for (int d = 1; d < NC; d <<= 1) {
    const int t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  const int g_sb0 = sb_first + inc - g_nsb, g_sb1 = sb_first + inc;
  // headers: cells into their columns, -1 into the unused ones; the rest by the group's lane
  {
    const int sb_of_mine = __shfl(g_sb0, my_sub);
    if (lane < ncells) hdr[G0 + my_sub].cell[my_col] = mycell;
    const int q = lane >> 3, col = lane & 7;
    const int nc_q = __shfl(g_nc, q);
    if (q < nsub && col >= nc_q) hdr[G0 + q].cell[col] = -1;
    if (lane < nsub) {
      GroupHdr *H = hdr + G0 + lane;
      H->sb_begin = g_sb0;
      H->sb_end = g_sb1;
      H->nk = g_nk;
      // shares that start right behind this group: sb_begin < t(w) <= sb_end (and, for the very first group, t(w) = 0);
      // t is monotone: the first such share by bisection
      const unsigned ua = (unsigned)nsb_all, uw = (unsigned)W, ub = (unsigned)big;
      unsigned lo = 0u, hi = uw;                            // smallest w in [0, W] with t(w) > sb_begin (t(W) = all > sb_begin)
      while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (share_t(mid, ua, uw, ub) > (unsigned)g_sb0) hi = mid; else lo = mid + 1u;
      }
      unsigned w = lo;
      if (G0 + lane == 0)
        for (unsigned w0 = 0; w0 < uw && share_t(w0, ua, uw, ub) == 0u; ++w0) wst[w0] = 0;
      for (; w < uw && share_t(w, ua, uw, ub) <= (unsigned)g_sb1; ++w) wst[w] = G0 + lane + 1;
      if (G0 + lane == ngroups - 1) wst[W] = ngroups;
    }
    // keys: every lane copies ONE key per pass (a chunk has ~40) and one padding key: two independent load -> store
    // round trips
    const unsigned stepy = (unsigned)Wi * 256u;
    const int my_dst = sb_of_mine * 8 + my_in;
    int pre = mycnt;                                      // keys in front of cell j in the CHUNK: exclusive prefix over lanes 0 .. 7
#pragma unroll
    for (int d = 1; d < NC; d <<= 1) {
      const int t = __shfl_up(pre, d);
      if (lane >= d) pre += t;
    }
    const int total = __builtin_amdgcn_readlane(pre, NC - 1);
    pre -= mycnt;
    for (int L0 = 0; L0 < total; L0 += 64) {              // a UNIFORM loop: the lane shuffles below read lanes 0 .. 7, which must be active
      const int L = L0 + lane;
      int j = 0;
#pragma unroll
      for (int jj = 1; jj < NC; ++jj) j += (jj < ncells && L >= __builtin_amdgcn_readlane(pre, jj)) ? 1 : 0;
      // (cells without keys share their `pre` with the next cell: the LAST cell with pre <= L is the one that owns key L)
      const int e = L - __shfl(pre, j), cell = __shfl(mycell, j), colj = __shfl(my_col, j), d0 = __shfl(my_dst, j);
      if (L < total) {
        const float4 *kp = reinterpret_cast<const float4 *>(keys + (size_t)cell * nslots + e);
        const float4 k0 = kp[0];
        const float2 k1 = *reinterpret_cast<const float2 *>(kp + 1);
        uint4 a = key_addr(k0, stepy);
        a.x |= (unsigned)colj;
        uint4 *dst = reinterpret_cast<uint4 *>(dense + d0 + e);
        dst[0] = a;
        dst[1] = __builtin_bit_cast(uint4, make_float4(k0.z, k0.w, k1.x, k1.y));
      }
    }
    const int pi = lane & 7;
    const int n = __shfl(g_nsb * 8 - g_nk, q), pc = __shfl(g_lastcell, q), pn = __shfl(g_lastcnt, q);
    const int pd = __shfl(g_sb0 * 8 + g_nk, q);
    if (q < nsub && pi < n) {                             // < 8 padding keys, or 8 for a group without keys
      uint4 last = make_uint4(0u, 0u, 0u, 0u);            // the address part of the group's last real key (padding re-reads it)
      if (pc >= 0) last = key_addr(*reinterpret_cast<const float4 *>(keys + (size_t)pc * nslots + (pn - 1)), stepy);
      uint4 *dst = reinterpret_cast<uint4 *>(dense + pd + pi);
      dst[0] = make_uint4(last.x | 15u, last.y, last.z, last.w);
      dst[1] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
}

// ---- LDS-DMA (inline assembly on purpose: the compiler must not learn that these are vector-memory operations, or it puts
// vmcnt(0) in front of every LDS read; the "memory" clobber keeps its own LDS accesses on their side of each statement)
__device__ __forceinline__ unsigned lds_addr_of(const void *p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void dma16_flat(const void *gp, unsigned lds_addr) {            // 64-bit address per lane
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gp), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void dma16_base(unsigned voff, const void *sbase, unsigned lds_addr) {   // uniform base + 32-bit lane offset
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
// A group header (64 B) through the SCALAR cache: lgkmcnt, not vmcnt - a vector load here would make the compiler put a
// vmcnt(0) into the superblock loop at every group boundary, with the next superblocks' row DMAs in flight.  (Written as
// assembly because the compiler turns `lane-dependent choice among H.cell[..]` into per-lane vector loads.)
typedef int i16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ i16v load_hdr(const void *p) {
  i16v v;
  asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}
// cell of column j (0..7) out of a header's eight
__device__ __forceinline__ int pick_cell(const i16v &h, int j) {
  const int c01 = j & 1 ? h[1] : h[0], c23 = j & 1 ? h[3] : h[2], c45 = j & 1 ? h[5] : h[4], c67 = j & 1 ? h[7] : h[6];
  const int c03 = j & 2 ? c23 : c01, c47 = j & 2 ? c67 : c45;
  return j & 4 ? c47 : c03;
}
// Four instructions behind ONE M0: an instruction offset applies to BOTH addresses (global and LDS), so instruction t uses
// the base `sbase - 1024 t` and offset:1024 t - its data lands at M0 + 1024 t + 16 lane.  One statement: M0 must not change
// in between.
__device__ __forceinline__ void dma16_x4(unsigned v0, unsigned v1, unsigned v2, unsigned v3, const void *b0, const void *b1,
                                         const void *b2, const void *b3, unsigned lds_addr) {
  asm volatile(
      "s_mov_b32 m0, %8\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %4\n\t"
      "global_load_lds_dwordx4 %1, %5 offset:1024\n\t"
      "global_load_lds_dwordx4 %2, %6 offset:2048\n\t"
      "global_load_lds_dwordx4 %3, %7 offset:3072"
      :
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(lds_addr)
      : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
// v_permlane16_swap / v_permlane32_swap exchange the odd rows (halves) of one register with the even rows (halves) of
// another.  hipcc (ROCm 7.2) MISCOMPILES float arithmetic on the two results of the builtin: `float(r[0]) + float(r[1])`
// becomes `v_add_f32 v, r0, r0` (the second result is dropped; integer uses of r[0] / r[1] are fine - the ring kernel
// stores them - tools/hazard/permlane_swap_fold.hip).  Both results go through an empty asm statement, which keeps them apart.
__device__ __forceinline__ void xpair(float x, float &lo, float &hi, bool half32) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const u2v r = half32 ? __builtin_amdgcn_permlane32_swap(u, u, false, false) : __builtin_amdgcn_permlane16_swap(u, u, false, false);
  unsigned a = r[0], b = r[1];
  asm volatile("" : "+v"(a), "+v"(b));
  lo = __builtin_bit_cast(float, a);
  hi = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float max_raw(float a, float b) {      // v_max_f32 without fmaxf's canonicalising v_max x, x
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float xmax(float x) {         // over the four 16-lane rows (lanes with equal lane % 16)
  float a, b;
  xpair(x, a, b, false);
  x = max_raw(a, b);
  xpair(x, a, b, true);
  return max_raw(a, b);
}
__device__ __forceinline__ float xsum(float x) {
  float a, b;
  xpair(x, a, b, false);
  x = a + b;
  xpair(x, a, b, true);
  return a + b;
}
// f(row): the 16-B slots of raw row R are stored rotated (XOR) by f(R % 16).  ds_read_b128 serves a wavefront in the lane
// groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32): with A = rows {0-3, 12-15}, B = rows {4-11}, the score reads of
// a group are slot (4 s) ^ f(A-rows) and (4 s + 1) ^ f(B-rows) - conflict free when f(A) and f(B) are each closed under ^ 1;
// the transposed 8-B reads take 32 lanes = 8 rows x 32 B: conflict free when f(R) >> 1 is a bijection on rows 0-7 and on
// rows 8-15.  (The first version, f = 2 R + R / 8, assumed 16 CONSECUTIVE lanes per group: 35 % of the LDS cycles were
// bank-conflict cycles.)
__device__ __forceinline__ int slot_rot(int R) {
  R &= 15;
  return R < 8 ? 2 * R : R < 12 ? 2 * (R - 4) + 1 : 2 * (R - 12) + 1;
}


// One wavefront = one SHARE of the stream: the consecutive groups [wst[w], wst[w + 1]), processed as one pipelined sequence of
// superblocks (8 keys = 32 raw rows = 8 KB): the rows of superblock b + NB - 1, the keys 4 superblocks ahead and the next
// group's queries are in flight (LDS-DMA) while superblock b is multiplied; a group boundary costs the flush of 8 output rows
// and four LDS reads - no memory latency.  Headers come through the scalar cache (lgkmcnt, not vmcnt).
template <int NB>
__global__ __launch_bounds__(64) void attn_dense_kernel(
    const unsigned char *__restrict__ img, const unsigned char *__restrict__ qfold, const GroupHdr *__restrict__ hdr,
    const int *__restrict__ wst, const DenseKey *__restrict__ dense, __half *__restrict__ ctx, __half *__restrict__ valid_out,
    int W) {
  constexpr int SBYTES = 8192, FEAT = NB * SBYTES, KEYS = FEAT, QBUF = FEAT + 2048, WBYTES = QBUF + 2048;
  __shared__ __align__(1024) unsigned char smem[WBYTES];
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, g = lane >> 4;
  // XCD x (blocks with blockIdx % 8 == x) takes the x-th eighth of the shares = one sector of the scene (a band of columns
  // in one or two cameras)
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, wx = W >> 3;
  const int w = xcd * wx + lb;
  if (lb >= wx) return;
  const int G0 = __builtin_amdgcn_readfirstlane(wst[w]), G1 = __builtin_amdgcn_readfirstlane(wst[w + 1]);
  if (G0 >= G1) return;
  unsigned char *wbase = smem;
  const unsigned lds_w = __builtin_amdgcn_readfirstlane(lds_addr_of(wbase));

  // lane constants.  Raw row R = 4 key + corner (key 0..7 of the superblock), 256 B per row, 16-B slots rotated by f(R % 16).
  // Score product, tile h: lane (i, g) reads 16 B of row 16 h + i, k-step s: slot (4 s + g) ^ f(i).
  unsigned sc_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) sc_off[s] = i * 256 + (((4 * s + g) ^ slot_rot(i)) << 4);
  // Value product (transposed read), tile h: lane (i, g) addresses row 16 h + 4 g + (i >> 2), channels 16 blk + 4 (i & 3) .. + 3
  const int Rt = 4 * g + (i >> 2);
  unsigned tr_off[8];
#pragma unroll
  for (int blk = 0; blk < 8; ++blk) tr_off[blk] = Rt * 256 + (((2 * blk + ((i >> 1) & 1)) ^ slot_rot(Rt)) << 4) + ((i & 1) << 3);
  // DMA instruction t (= key t) of a superblock: lane (g, i) fetches slot i ^ f((4 t + g) % 16) of raw row 4 t + g (corner g)
  unsigned dma_slot[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) dma_slot[t] = (unsigned)((i ^ slot_rot(4 * t + g)) << 4);

  i16v hc = load_hdr(hdr + G0);            // the current group's header: cells 0..7, sb_begin 8, sb_end 9
  i16v hn = hc;                            // the next group's
  const int sb0 = hc[8], sb1 = load_hdr(hdr + (G1 - 1))[9];          // the share's superblocks
  const int nkeys = (sb1 - sb0) * 8;
  const DenseKey *dk = dense + (size_t)sb0 * 8;
  auto dma_keys = [&](int j) {             // 32 keys = 1 KB: lane -> key lane / 2, half lane % 2 (past the end: the last key again)
    const int k = min(32 * j + (lane >> 1), nkeys - 1);
    dma16_flat(reinterpret_cast<const unsigned char *>(dk + k) + (lane & 1) * 16, lds_w + KEYS + (j & 1) * 1024);
  };
  // the queries of group G (header H): 8 rows of 256 B -> the query buffer (two instructions: lane (g, i) fetches 16 B of the row
  // of cell 4 t + g)
  auto dma_q = [&](int G, const i16v &H) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // ONE query buffer: the reads of the previous group's rows are done
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int cell = max(pick_cell(H, 4 * t + g), 0);
      dma16_base((unsigned)cell * 256u + (unsigned)((i ^ (2 * ((4 * t + g) & 7))) & 15) * 16u, qfold, lds_w + QBUF + t * 1024);
    }
  };
  // everything superblock nb (relative to the share) needs from the key ring for its 8 x 4 row fetches.  `slot`: a compile-time
  // buffer index (the superblock loop is unrolled by NB: every LDS address is lane constant + immediate)
  auto issue = [&](int nb, int slot) {
    // the next 32 keys, while superblock nb - (NB - 1) = 4 c is multiplied: the slot they overwrite held chunk c - 1, whose last
    // superblock (4 c - 1) is done (the weights / columns of a superblock are read from the ring when it is multiplied)
    if ((nb & 3) == NB - 1 && 32 * ((nb >> 2) + 1) < nkeys) dma_keys((nb >> 2) + 1);
    const unsigned char *kr = wbase + KEYS + ((nb >> 2) & 1) * 1024 + (nb & 3) * 256;
    unsigned off[8];                       // all LDS reads first: every DMA statement is a fence for the compiler's LDS accesses
#pragma unroll
    for (int t = 0; t < 8; ++t) off[t] = *reinterpret_cast<const unsigned *>(kr + t * 32 + g * 4);
#pragma unroll
    for (int h = 0; h < 2; ++h)
      dma16_x4((off[4 * h] & ~255u) | dma_slot[0], (off[4 * h + 1] & ~255u) | dma_slot[1], (off[4 * h + 2] & ~255u) | dma_slot[2],
               (off[4 * h + 3] & ~255u) | dma_slot[3], img, img - 1024, img - 2048, img - 3072, lds_w + slot * SBYTES + h * 4096);
  };
  const int nsb = sb1 - sb0;

  dma_keys(0);
  dma_q(G0, hc);
  wait_vm<0>();
#pragma unroll
  for (int pb = 0; pb < NB - 1; ++pb)
    if (pb < nsb) issue(pb, pb);
  int G = G0;
  int g_end = hc[9] - sb0;                 // first superblock (relative) behind the current group
  h8 qt[4];
  auto read_q = [&](int Gq) {              // lane (i, g): 16 B of the query of column i % 8 per k-step
    // (the 16-B slots of row r are stored rotated by 2 r: the 16 lanes of a ds_read_b128 group - 8 rows x 2 k-groups - hit 16
    //  different slots; unrotated, the 8 rows of a group shared 4 banks: 8-way conflicts at every group boundary were 23 % of
    //  the kernel's LDS cycles)
    const int j = i & (NC - 1);
    const unsigned char *qb = wbase + QBUF + j * 256;
#pragma unroll
    for (int s = 0; s < 4; ++s) qt[s] = *reinterpret_cast<const h8 *>(qb + (((4 * s + g) ^ (2 * j)) << 4));
  };
  read_q(G0);
  if (G0 + 1 < G1) {                       // the ONE query buffer is free again
    hn = load_hdr(hdr + (G0 + 1));
    dma_q(G0 + 1, hn);
  }
  f4 acc[8];
#pragma unroll
  for (int blk = 0; blk < 8; ++blk) acc[blk] = f4{0.f, 0.f, 0.f, 0.f};
  float mref = -INFINITY, l = 0.f;

  // one superblock; SLOT = its row buffer (compile time)
  auto body = [&](auto SLOT, int b) {
    constexpr int slot = decltype(SLOT)::value, snext = (slot + NB - 1) % NB;
    const int nb = b + NB - 1;
    if (nb < nsb) {
      issue(nb, snext);
      wait_vm<8 * (NB - 1)>();             // superblock b has landed (younger: the NB - 1 superblocks behind it)
    } else {
      wait_vm<0>();
    }
    const unsigned char *fb = wbase + slot * SBYTES;
    // lane (i, g): the weights and columns of keys g and 4 + g, from the key ring
    const unsigned char *kc_ = wbase + KEYS + ((b >> 2) & 1) * 1024 + (b & 3) * 256 + g * 32;
    struct {
      f4 w[2];
      int col[2];
    } kc;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      kc.w[h] = *reinterpret_cast<const f4 *>(kc_ + h * 128 + 16);
      kc.col[h] = (int)(*reinterpret_cast<const unsigned *>(kc_ + h * 128) & 15u);
    }

    f4 D[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      D[h] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const h8 a = *reinterpret_cast<const h8 *>(fb + h * 4096 + sc_off[s]);
        D[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qt[s], D[h], 0, 0, 0);
      }
    }
    bool ok[2];
    float sc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      ok[h] = kc.col[h] == i;              // padding keys carry column 15 (never a cell: their weights are 0)
      const float v = (D[h][0] * kc.w[h][0] + D[h][1] * kc.w[h][1] + D[h][2] * kc.w[h][2] + D[h][3] * kc.w[h][3]) * kLog2e;
      sc[h] = ok[h] ? v : -INFINITY;
    }
    const float mx = xmax(max_raw(sc[0], sc[1]));
    const bool beat = mx > mref + kLazy;               // also the first key of a column (mref = -inf)
    const float mnew = beat ? mx : mref;
    if (__any(beat && mref > -INFINITY)) {             // rare: a later key beats the reference by more than 2^kLazy
      const float a = mnew == mref ? 1.f : __builtin_amdgcn_exp2f(mref - mnew);
#pragma unroll
      for (int blk = 0; blk < 8; ++blk) acc[blk] *= a;
      l *= a;
    }
    mref = mnew;
    h8 bc;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float p = ok[h] ? __builtin_amdgcn_exp2f(sc[h] - mref) : 0.f;
      l += p;
#pragma unroll
      for (int r = 0; r < 4; ++r) bc[4 * h + r] = (_Float16)(p * kc.w[h][r]);
    }
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
      const hv4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(fb + tr_off[blk]));
      const hv4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((hv4 __attribute__((address_space(3))) *)(fb + 4096 + tr_off[blk]));
      h8 at;
      at[0] = v0[0]; at[1] = v0[1]; at[2] = v0[2]; at[3] = v0[3];
      at[4] = v1[0]; at[5] = v1[1]; at[6] = v1[2]; at[7] = v1[3];
      acc[blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(at, bc, acc[blk], 0, 0, 0);
    }
    if (b + 1 == g_end) {
      // the group is complete: every one of its cells is written - cells without a key get their zero row (l = 0, acc = 0) -
      // and the padding column (15) is dropped
      const int ci = pick_cell(hc, i & (NC - 1));
      const float lt = xsum(l);
      const float inv = lt > 0.f ? 1.f / lt : 0.f;
      // a lane holds channels 16 blk + 4 g .. + 3 of its column; v_permlane16_swap between the blocks of a pair hands it 8
      // consecutive channels (g even: block 2 pr, channels 4 g .. 4 g + 7; g odd: block 2 pr + 1, channels 4 (g - 1) .. + 7):
      // four 16-byte stores per group instead of eight 8-byte ones (the stores share the texture path with the row fetches)
      unsigned wv[8][2];
#pragma unroll
      for (int blk = 0; blk < 8; ++blk) {
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (_Float16)(acc[blk][r] * inv);
        const uint2 raw = __builtin_bit_cast(uint2, o);
        wv[blk][0] = raw.x;
        wv[blk][1] = raw.y;
      }
#pragma unroll
      for (int pr = 0; pr < 4; ++pr)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const u2v sw = __builtin_amdgcn_permlane16_swap(wv[2 * pr][d], wv[2 * pr + 1][d], false, false);
          wv[2 * pr][d] = sw[0];
          wv[2 * pr + 1][d] = sw[1];
        }
      if (i < NC && ci >= 0) {
        __half *dst = ctx + (size_t)ci * 128 + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);
#pragma unroll
        for (int pr = 0; pr < 4; ++pr)
          *reinterpret_cast<uint4 *>(dst + 32 * pr) = make_uint4(wv[2 * pr][0], wv[2 * pr][1], wv[2 * pr + 1][0], wv[2 * pr + 1][1]);
        if (g == 0) valid_out[ci] = (__half)(lt > 0.f ? 1.f : 0.f);
      }
#pragma unroll
      for (int blk = 0; blk < 8; ++blk) acc[blk] = f4{0.f, 0.f, 0.f, 0.f};
      mref = -INFINITY;
      l = 0.f;
      ++G;
      if (G < G1) {
        hc = hn;
        g_end = hc[9] - sb0;
        // the queries were issued one boundary ago, behind superblock b + NB - 2's rows and in front of b + NB - 1's: the wait
        // of this iteration covered them for NB = 2; deeper rings leave NB - 2 younger superblocks in flight
        if (NB > 2) wait_vm<8 * (NB > 2 ? NB - 2 : 0)>();
        read_q(G);
        if (G + 1 < G1) {                  // the buffer is free: this group's queries are in registers
          hn = load_hdr(hdr + (G + 1));
          dma_q(G + 1, hn);
        }
      }
    }
  };
  for (int b0 = 0; b0 < nsb; b0 += NB) {
    body(std::integral_constant<int, 0>(), b0);
    if (b0 + 1 < nsb) body(std::integral_constant<int, 1>(), b0 + 1);
    if constexpr (NB > 2)
      if (b0 + 2 < nsb) body(std::integral_constant<int, 2>(), b0 + 2);
  }
}

template <int NB>
static int launch_attn(const void *img, const void *qfold, const GroupHdr *hdr, const int *wst, const DenseKey *dense, void *ctx,
                       void *valid, int W, hipStream_t s) {
  const int blocks = W;                                      // one wavefront per workgroup; W is a multiple of 8
  hipEvent_t ev0, ev1;
  if (take_launch_events(ev0, ev1))                          // measurement: the dispatch's own begin / end time stamps
    hipExtLaunchKernelGGL((attn_dense_kernel<NB>), dim3(blocks), dim3(64), 0, s, ev0, ev1, 0, (const unsigned char *)img,
                          (const unsigned char *)qfold, hdr, wst, dense, (__half *)ctx, (__half *)valid, W);
  else
    hipLaunchKernelGGL((attn_dense_kernel<NB>), dim3(blocks), dim3(64), 0, s, (const unsigned char *)img,
                       (const unsigned char *)qfold, hdr, wst, dense, (__half *)ctx, (__half *)valid, W);
  return check_launch("i2p_attn_dense_fwd");
}

// shares of the stream = resident wavefronts of the device (a multiple of 8: one eighth per XCD)
static int n_shares() {
  static const int per_cu = getenv("DI_I2PD_WAVES") ? atoi(getenv("DI_I2PD_WAVES")) : kWavesPerCU;
  const int cus = device_cus();
  if (cus <= 0) return 0;
  return std::min(kMaxWaves, std::max(8, cus * per_cu / 8 * 8));
}

}  // namespace i2pd
}  // namespace di

extern "C" {

long long di_i2p_dense_bytes(int Hb, int Wb, int T, int n_views, int P) {
  const long long ncell = (long long)Hb * Wb;
  // real keys + padding to whole superblocks (< 8 per group, 8 for a group without keys; at most one group per cell)
  const long long cap = std::min<long long>(ncell, std::max(P, 0)) * T * n_views + 8ll * ncell;
  return di::i2pd::off_keys((int)ncell) + cap * (long long)sizeof(di::i2pd::DenseKey);
}

int di_i2p_compact_keys(const void *key_table, const int32_t *cell_order, void *dense_table, int T, int n_views, int Wi, int Hb,
                        int Wb, void *stream) {
  DI_REQUIRE(key_table && dense_table, "null table");
  DI_REQUIRE(T > 0 && n_views > 0 && T * n_views <= di::kMaxSlots, "T*n_views=%d exceeds %d key slots", T * n_views, di::kMaxSlots);
  DI_REQUIRE(Hb > 0 && Wb > 0 && Wi > 0, "bad map shape");
  const int ncell = Hb * Wb, nchunks = di::i2pd::n_chunks(ncell);
  DI_REQUIRE(nchunks <= 8 * 1024, "%d cell chunks exceed the scan kernel's 8192", nchunks);
  DI_REQUIRE((long long)ncell * (T * n_views / 8 + 2) < (1ll << 20), "map too large for the share table's 32-bit arithmetic");
  const int W = di::i2pd::n_shares();
  if (W <= 0) return DI_ERR_LAUNCH;
  static const int key_cap = getenv("DI_I2PD_KEYCAP") ? atoi(getenv("DI_I2PD_KEYCAP")) : di::i2pd::kKeyCap;
  static const int big = getenv("DI_I2PD_BIG") ? std::min(15, std::max(1, atoi(getenv("DI_I2PD_BIG")))) : di::i2pd::kBigSixteenths;
  const int *cnt = reinterpret_cast<const int *>(key_table);
  const di::KeyEnt *keys = reinterpret_cast<const di::KeyEnt *>(cnt + 2 * (size_t)ncell);
  char *base = reinterpret_cast<char *>(dense_table);
  di::i2pd::GroupHdr *hdr = reinterpret_cast<di::i2pd::GroupHdr *>(base);
  int *csub = reinterpret_cast<int *>(base + di::i2pd::off_csub(ncell));
  int *csb = reinterpret_cast<int *>(base + di::i2pd::off_csb(ncell));
  int *wst = reinterpret_cast<int *>(base + di::i2pd::off_wst(ncell));
  di::i2pd::DenseKey *dense = reinterpret_cast<di::i2pd::DenseKey *>(base + di::i2pd::off_keys(ncell));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(di::i2pd::chunk_count_kernel, dim3((nchunks + 255) / 256), dim3(256), 0, s, cnt, cell_order, csub, csb, ncell,
                     nchunks, key_cap);
  hipLaunchKernelGGL(di::i2pd::scan_kernel, dim3(2), dim3(1024), 0, s, csub, csb, nchunks);
  hipLaunchKernelGGL(di::i2pd::compact_kernel, dim3((nchunks + 3) / 4), dim3(256), 0, s, cnt, keys, cell_order, csub, csb, hdr, wst,
                     dense, ncell, nchunks, T * n_views, Wi, W, key_cap, big);
  return di::check_launch("i2p_compact_keys");
}

int di_i2p_attn_dense_fwd(const void *img, const void *qfold, const void *key_table, const void *dense_table,
                          const int32_t *cell_order, void *ctx, void *valid, int n_views, int Hi, int Wi, int Hb, int Wb,
                          void *stream) {
  (void)key_table;
  (void)cell_order;                                           // the walk order is baked into the dense table's group headers
  DI_REQUIRE(img && qfold && dense_table && ctx && valid, "null argument");
  DI_REQUIRE(n_views > 0 && Hi > 0 && Wi > 0 && Hb > 0 && Wb > 0, "bad map shape");
  DI_REQUIRE((long long)n_views * Hi * Wi * 256 < (1ll << 32), "image map too large for 32-bit byte offsets");
  DI_REQUIRE((long long)Hb * Wb * 256 < (1ll << 32), "BEV map too large for 32-bit byte offsets");
  const int ncell = Hb * Wb;
  const int W = di::i2pd::n_shares();
  if (W <= 0) return DI_ERR_LAUNCH;
  const char *base = reinterpret_cast<const char *>(dense_table);
  const di::i2pd::GroupHdr *hdr = reinterpret_cast<const di::i2pd::GroupHdr *>(base);
  const int *wst = reinterpret_cast<const int *>(base + di::i2pd::off_wst(ncell));
  const di::i2pd::DenseKey *dense = reinterpret_cast<const di::i2pd::DenseKey *>(base + di::i2pd::off_keys(ncell));
  static const int nb_env = getenv("DI_I2PD_NB") ? atoi(getenv("DI_I2PD_NB")) : 2;
  hipStream_t s = (hipStream_t)stream;
  if (nb_env == 3) return di::i2pd::launch_attn<3>(img, qfold, hdr, wst, dense, ctx, valid, W, s);
  return di::i2pd::launch_attn<2>(img, qfold, hdr, wst, dense, ctx, valid, W, s);
}

}  // extern "C"
