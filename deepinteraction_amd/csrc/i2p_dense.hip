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

constexpr int NC = 8;            // BEV cells per group (MFMA columns in use; 16 columns exist)
constexpr int NW = 2;            // wavefronts per workgroup, one group per wavefront at a time
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLazy = 8.f;     // log2 units: the reference maximum is kept until a score beats it by more than this

struct DenseKey {                // 32 B
  unsigned off[4];               // byte offset of corner row r from the map base (a multiple of 256); off[0] bits 0..3: the column
  float w[4];                    // bilinear weights (0 where grid_sample's zero padding applies)
};

__host__ __device__ inline int n_groups(int ncell) { return (ncell + NC - 1) / NC; }
__host__ __device__ inline long long keys_offset(int ngroups) { return (((long long)(ngroups + 1) * 4 + 255) / 256) * 256; }

__device__ __forceinline__ int group_keys(const int *__restrict__ cnt, const int *__restrict__ order, int G, int ncell) {
  int s = 0;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int pos = G * NC + j;
    if (pos < ncell) s += cnt[order != nullptr ? order[pos] : pos];
  }
  return s;
}

// gcount[G] = keys of group G (one thread per group; the first version did this inside the one-workgroup scan: 32 dependent
// gathers per thread, 66 us)
__global__ __launch_bounds__(256) void group_count_kernel(const int *__restrict__ cnt, const int *__restrict__ order,
                                                          int *__restrict__ gcount, int ncell, int ngroups) {
  const int G = blockIdx.x * 256 + threadIdx.x;
  if (G < ngroups) gcount[G] = group_keys(cnt, order, G, ncell);
}

// gstart[G] = number of keys in front of group G in the dense stream (exclusive prefix sum of the groups' key counts, IN
// PLACE over gcount); gstart[ngroups] = all keys.  ONE workgroup, plain stores only (a captured graph replays this: no
// atomics, no memset).  A thread owns `per` consecutive groups (ngroups <= 8 * 1024).
__global__ __launch_bounds__(1024) void group_scan_kernel(int *__restrict__ gstart, int ngroups) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (ngroups + 1023) / 1024;
  int v[8];
  int mine = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int G = tid * per + k;
    v[k] = (k < per && G < ngroups) ? gstart[G] : 0;
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
    const int G = tid * per + k;
    if (k < per && G < ngroups) gstart[G] = run;
    run += v[k];
  }
  if (tid == 1023) gstart[ngroups] = run;
}

// One wavefront per group: the keys of its cells, cell after cell, into the dense stream - ready to fetch (explicit corner
// offsets) and tagged with the cell's column.
__global__ __launch_bounds__(256) void compact_kernel(const int *__restrict__ cnt, const KeyEnt *__restrict__ keys,
                                                      const int *__restrict__ order, const int *__restrict__ gstart,
                                                      DenseKey *__restrict__ dense, int ncell, int ngroups, int nslots, int Wi) {
  const int lane = threadIdx.x & 63;
  const int G = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (G >= ngroups) return;
  int mycell = 0, mycnt = 0;
  if (lane < NC && G * NC + lane < ncell) {
    mycell = order != nullptr ? order[G * NC + lane] : G * NC + lane;
    mycnt = cnt[mycell];
  }
  int run = gstart[G];
  const unsigned stepy = (unsigned)Wi * 256u;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = __builtin_amdgcn_readlane(mycnt, j), cell = __builtin_amdgcn_readlane(mycell, j);
    for (int e = lane; e < c; e += 64) {
      const float4 *kp = reinterpret_cast<const float4 *>(keys + (size_t)cell * nslots + e);
      const float4 k0 = kp[0];
      const float2 k1 = *reinterpret_cast<const float2 *>(kp + 1);
      const int pix = __float_as_int(k0.x), info = __float_as_int(k0.y);
      const unsigned o00 = (unsigned)pix * 256u, dx = (info & 1) ? 256u : 0u, dy = (info & 2) ? stepy : 0u;
      uint4 a = make_uint4(o00 | (unsigned)j, o00 + dx, o00 + dy, o00 + dx + dy);
      float4 b = make_float4(k0.z, k0.w, k1.x, k1.y);
      uint4 *dst = reinterpret_cast<uint4 *>(dense + run + e);
      dst[0] = a;
      dst[1] = __builtin_bit_cast(uint4, b);
    }
    run += c;
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
__device__ __forceinline__ float xmax(float x) {         // over the four 16-lane rows (lanes with equal lane % 16)
  float a, b;
  xpair(x, a, b, false);
  x = fmaxf(a, b);
  xpair(x, a, b, true);
  return fmaxf(a, b);
}
__device__ __forceinline__ float xsum(float x) {
  float a, b;
  xpair(x, a, b, false);
  x = a + b;
  xpair(x, a, b, true);
  return a + b;
}
__device__ __forceinline__ int slot_rot(int R) { return (2 * R + (R >> 3)) & 15; }   // f(row): distinct for rows 0..15

template <int NB>
__global__ __launch_bounds__(NW * 64) void attn_dense_kernel(
    const unsigned char *__restrict__ img, const __half *__restrict__ qfold, const int *__restrict__ cnt_tab,
    const int *__restrict__ order, const int *__restrict__ gstart, const DenseKey *__restrict__ dense,
    __half *__restrict__ ctx, __half *__restrict__ valid_out, int ncell, int ngroups) {
  // A SUPERBLOCK = 8 keys = 32 raw rows (8 KB): two score tiles of 16 rows, ONE value product with k = 32 rows.
  constexpr int SBYTES = 8192, FEAT = NB * SBYTES, WBYTES = FEAT + 2048;   // per wavefront: NB row buffers + two key slots
  __shared__ __align__(1024) unsigned char smem[NW * WBYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  // XCD x (blocks with blockIdx % 8 == x) takes the x-th eighth of the groups = one sector of the scene (a band of columns
  // in one or two cameras); its wavefronts sweep it together
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const int gp = (ngroups + 7) >> 3, glo = xcd * gp, ghi = min(glo + gp, ngroups);
  unsigned char *wbase = smem + wave * WBYTES;
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

  for (int G = glo + lb * NW + wave; G < ghi; G += nbx * NW) {
    int mycell = -1, mycnt = 0;
    if (lane < NC && G * NC + lane < ncell) {
      mycell = order != nullptr ? order[G * NC + lane] : G * NC + lane;
      mycnt = cnt_tab[mycell];
    }
    const int kbeg = __builtin_amdgcn_readfirstlane(gstart[G]);
    const int nk = __builtin_amdgcn_readfirstlane(gstart[G + 1]) - kbeg;
    const int nsb = (nk + 7) >> 3;
    if (lane < NC && mycell >= 0) valid_out[mycell] = (__half)(mycnt > 0 ? 1.f : 0.f);
    const int ci = __shfl(mycell, i & (NC - 1));
    h8 qt[4];
    {
      const __half *qp = qfold + (size_t)max(ci, 0) * 128 + 8 * g;
#pragma unroll
      for (int s = 0; s < 4; ++s) qt[s] = *reinterpret_cast<const h8 *>(qp + 32 * s);
      // the compiler's wait for these loads must sit HERE: placed at their first use it would be a vmcnt(0) inside the
      // block loop, in front of the first MFMA of every block - with the row DMAs of the next block in flight
#pragma unroll
      for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qt[s]));
    }
    f4 acc[8];
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) acc[blk] = f4{0.f, 0.f, 0.f, 0.f};
    float mref = -INFINITY, l = 0.f;

    if (nsb > 0) {
      const DenseKey *dk = dense + kbeg;
      auto dma_keys = [&](int j) {           // 32 keys = 1 KB: lane -> key lane / 2, half lane % 2 (past the end: the last key again)
        const int k = min(32 * j + (lane >> 1), nk - 1);
        dma16_flat(reinterpret_cast<const unsigned char *>(dk + k) + (lane & 1) * 16, lds_w + FEAT + (j & 1) * 1024);
      };
      struct KeyData {                       // lane (i, g): keys g and 4 + g of a superblock
        f4 w[2];
        int col[2];
      };
      KeyData kn;
      // everything superblock nb needs from the key ring: its 8 x 4 row fetches, and the weights and columns of a lane's keys
      auto issue = [&](int nb, int slot) {
        if ((nb & 3) == 0 && 32 * ((nb >> 2) + 1) < nk) dma_keys((nb >> 2) + 1);     // the next 32 keys, 4 superblocks ahead
        const unsigned char *kr = wbase + FEAT + ((nb >> 2) & 1) * 1024 + (nb & 3) * 256;
        unsigned off[8];                     // all LDS reads first: every DMA statement is a fence for the compiler's LDS accesses
#pragma unroll
        for (int t = 0; t < 8; ++t) off[t] = *reinterpret_cast<const unsigned *>(kr + t * 32 + g * 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          kn.w[h] = *reinterpret_cast<const f4 *>(kr + (4 * h + g) * 32 + 16);
          kn.col[h] = (int)(*reinterpret_cast<const unsigned *>(kr + (4 * h + g) * 32) & 15u);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
          dma16_base((off[t] & ~255u) | dma_slot[t & 3], img, lds_w + slot * SBYTES + t * 1024);
      };
      dma_keys(0);
      wait_vm<0>();
      KeyData kq[NB - 1];                    // key data of the superblocks in flight (their ring slot may be overwritten before they run)
#pragma unroll
      for (int pb = 0; pb < NB - 1; ++pb) {
        if (pb < nsb) issue(pb, pb);
        kq[pb] = kn;
      }
      int slot = 0;
      for (int b = 0; b < nsb; ++b) {
        const int nb = b + NB - 1;
        int snext = slot + NB - 1;
        if (snext >= NB) snext -= NB;
        if (nb < nsb) {
          issue(nb, snext);
          wait_vm<8 * (NB - 1)>();           // superblock b has landed (younger: the NB - 1 superblocks behind it)
        } else if (NB > 2 && nb - 1 < nsb) {
          wait_vm<8 * (NB > 2 ? NB - 2 : 0)>();
        } else {
          wait_vm<0>();
        }
        const unsigned char *fb = wbase + slot * SBYTES;
        const KeyData kc = kq[0];
#pragma unroll
        for (int q = 0; q + 1 < NB - 1; ++q) kq[q] = kq[q + 1];
        kq[NB - 2] = kn;

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
          ok[h] = (8 * b + 4 * h + g < nk) && (kc.col[h] == i);
          const float v = (D[h][0] * kc.w[h][0] + D[h][1] * kc.w[h][1] + D[h][2] * kc.w[h][2] + D[h][3] * kc.w[h][3]) * kLog2e;
          sc[h] = ok[h] ? v : -INFINITY;
        }
        const float mx = xmax(fmaxf(sc[0], sc[1]));
        const bool beat = mx > mref + kLazy;             // also the first key of a column (mref = -inf)
        const float mnew = beat ? mx : mref;
        if (__any(beat && mref > -INFINITY)) {           // rare: a later key beats the reference by more than 2^kLazy
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
        slot = slot + 1 == NB ? 0 : slot + 1;
      }
    }
    // every cell of the group is written: cells without a key get their zero row (l = 0, acc = 0)
    const float lt = xsum(l);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    if (i < NC && ci >= 0) {
      __half *dst = ctx + (size_t)ci * 128 + 4 * g;
#pragma unroll
      for (int blk = 0; blk < 8; ++blk) {
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (_Float16)(acc[blk][r] * inv);
        *reinterpret_cast<h4 *>(dst + 16 * blk) = o;
      }
    }
  }
}

template <int NB>
static int launch_attn(const void *img, const void *qfold, const int *cnt, const int32_t *order, const int *gstart,
                       const DenseKey *dense, void *ctx, void *valid, int ncell, int ngroups, int blocks, hipStream_t s) {
  hipEvent_t ev0, ev1;
  if (take_launch_events(ev0, ev1))                          // measurement: the dispatch's own begin / end time stamps
    hipExtLaunchKernelGGL((attn_dense_kernel<NB>), dim3(blocks), dim3(NW * 64), 0, s, ev0, ev1, 0, (const unsigned char *)img,
                          (const __half *)qfold, cnt, order, gstart, dense, (__half *)ctx, (__half *)valid, ncell, ngroups);
  else
    hipLaunchKernelGGL((attn_dense_kernel<NB>), dim3(blocks), dim3(NW * 64), 0, s, (const unsigned char *)img,
                       (const __half *)qfold, cnt, order, gstart, dense, (__half *)ctx, (__half *)valid, ncell, ngroups);
  return check_launch("i2p_attn_dense_fwd");
}

}  // namespace i2pd
}  // namespace di

extern "C" {

long long di_i2p_dense_bytes(int Hb, int Wb, int T, int n_views, int P) {
  const long long ncell = (long long)Hb * Wb;
  const long long cap = std::min<long long>(ncell, std::max(P, 0)) * T * n_views;
  return di::i2pd::keys_offset(di::i2pd::n_groups((int)ncell)) + std::max<long long>(cap, 1) * (long long)sizeof(di::i2pd::DenseKey);
}

int di_i2p_compact_keys(const void *key_table, const int32_t *cell_order, void *dense_table, int T, int n_views, int Wi, int Hb,
                        int Wb, void *stream) {
  DI_REQUIRE(key_table && dense_table, "null table");
  DI_REQUIRE(T > 0 && n_views > 0 && T * n_views <= di::kMaxSlots, "T*n_views=%d exceeds %d key slots", T * n_views, di::kMaxSlots);
  DI_REQUIRE(Hb > 0 && Wb > 0 && Wi > 0, "bad map shape");
  const int ncell = Hb * Wb, ngroups = di::i2pd::n_groups(ncell);
  DI_REQUIRE(ngroups <= 8 * 1024, "%d cell groups exceed the scan kernel's 8192", ngroups);
  const int *cnt = reinterpret_cast<const int *>(key_table);
  const di::KeyEnt *keys = reinterpret_cast<const di::KeyEnt *>(cnt + 2 * (size_t)ncell);
  int *gstart = reinterpret_cast<int *>(dense_table);
  di::i2pd::DenseKey *dense = reinterpret_cast<di::i2pd::DenseKey *>((char *)dense_table + di::i2pd::keys_offset(ngroups));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(di::i2pd::group_count_kernel, dim3((ngroups + 255) / 256), dim3(256), 0, s, cnt, cell_order, gstart, ncell, ngroups);
  hipLaunchKernelGGL(di::i2pd::group_scan_kernel, dim3(1), dim3(1024), 0, s, gstart, ngroups);
  hipLaunchKernelGGL(di::i2pd::compact_kernel, dim3((ngroups + 3) / 4), dim3(256), 0, s, cnt, keys, cell_order, gstart, dense, ncell,
                     ngroups, T * n_views, Wi);
  return di::check_launch("i2p_compact_keys");
}

int di_i2p_attn_dense_fwd(const void *img, const void *qfold, const void *key_table, const void *dense_table,
                          const int32_t *cell_order, void *ctx, void *valid, int n_views, int Hi, int Wi, int Hb, int Wb,
                          void *stream) {
  DI_REQUIRE(img && qfold && key_table && dense_table && ctx && valid, "null argument");
  DI_REQUIRE(n_views > 0 && Hi > 0 && Wi > 0 && Hb > 0 && Wb > 0, "bad map shape");
  DI_REQUIRE((long long)n_views * Hi * Wi * 256 < (1ll << 32), "image map too large for 32-bit byte offsets");
  const int ncell = Hb * Wb, ngroups = di::i2pd::n_groups(ncell);
  const int *cnt = reinterpret_cast<const int *>(key_table);
  const int *gstart = reinterpret_cast<const int *>(dense_table);
  const di::i2pd::DenseKey *dense =
      reinterpret_cast<const di::i2pd::DenseKey *>((const char *)dense_table + di::i2pd::keys_offset(ngroups));
  // one group per wavefront by default (a multiple of 8 blocks: one share of the groups per XCD); DI_I2PD_BLOCKS: fewer,
  // persistent workgroups (measurement)
  static const int want_blocks = getenv("DI_I2PD_BLOCKS") ? atoi(getenv("DI_I2PD_BLOCKS")) : 0;
  static const int nb_env = getenv("DI_I2PD_NB") ? atoi(getenv("DI_I2PD_NB")) : 2;
  const int gp = (ngroups + 7) / 8;
  int blocks = 8 * ((gp + di::i2pd::NW - 1) / di::i2pd::NW);
  if (want_blocks > 0) blocks = std::min(blocks, (want_blocks + 7) / 8 * 8);
  hipStream_t s = (hipStream_t)stream;
  if (nb_env == 3) return di::i2pd::launch_attn<3>(img, qfold, cnt, cell_order, gstart, dense, ctx, valid, ncell, ngroups, blocks, s);
  return di::i2pd::launch_attn<2>(img, qfold, cnt, cell_order, gstart, dense, ctx, valid, ncell, ngroups, blocks, s);
}

}  // extern "C"
