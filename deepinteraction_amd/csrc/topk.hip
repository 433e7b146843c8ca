// Top-Q proposals of the NMS-ed heat map (deepinteraction_decoder.py:242: `heatmap.view(B,-1).argsort(descending)
// [..., :Q]`; torch's sort-based top-k of 324 000 scores is 18 launches / ~118 us per forward).
//
// Radix SELECT instead of a sort.  Every element gets the 52-bit key
//        K = float_bits(score) << 20 | (0xFFFFF - index)          (scores >= 0: their bit patterns are ordered)
// so all keys are distinct and "value descending, lower index first on ties" is plain descending K - a
// deterministic rule where the reference's argsort leaves ties unspecified.
//
// Four launches (round 2, second version; the first one walked all five digit levels over the whole score map: 12
// launches, 95 us of kernel time):
//   1. hist     64 blocks: per-block LDS histogram of the TOP digit (11 value bits), stored as SUFFIX sums (plain stores)
//   2. pick     1 block:   sums the 64 suffix arrays: the digit T that holds the k-th largest key is where the total
//                          crosses k; the per-block suffix sums at T and T + 1 say how many keys of every slice lie
//                          above T and inside T -> exclusive scans = where every block of step 3 writes
//   3. collect  64 blocks: keys with digit > T (fewer than k) -> `above`, keys with digit == T -> `bin`, both FLAT
//   4. final    1 block:   above + bin fit 1024 slots -> sort them all; otherwise radix select with in-place compaction
//                          over `bin` (cached in LDS up to 16 384 keys): per level histogram, pick, keys above the
//                          picked digit -> selected, keys inside it -> compacted to the front; stops as soon as what
//                          is left fits the sort.  Then a bitonic sort of the <= 1024 survivors.
// No host synchronisation, no global atomics, no buffer zeroed by one kernel and accumulated into by another (see
// below), bit-reproducible.
#include "di_common.h"

namespace di {
namespace tk {

// NO global atomics and no buffer that is zeroed by one kernel and accumulated into by the next: every block writes
// its own partial histogram with plain stores and the pick kernel sums the partials.  (The first version accumulated
// one global histogram with atomicAdd after a plain-store clear by the previous kernel; replayed from a hipGraph that
// went wrong from the second replay on - atomics and plain stores to the same lines do not meet in the same cache
// level across the XCDs - while eager launches, with their full cache maintenance at every boundary, were fine.)
constexpr int kBins = 2048;
constexpr int kBlocks = 64;          // slices of the score row = partial histograms per sample (one wave scans them)
constexpr int kTopShift = 41;        // the top digit: key bits 41..51 = score bits 21..31
constexpr int kNT = 1024;            // threads of every kernel here
constexpr int kCache = 16384;        // keys of the selected top bin the final kernel caches in LDS (128 KB)
typedef unsigned long long u64;

struct State {
  int T;                     // the top digit that holds the k-th largest key
  int above;                 // keys with a larger top digit (< k): all selected
  int take;                  // k - above: how many keys of digit T are selected
  int M;                     // keys with top digit T
  int above_off[kBlocks];    // where slice p writes its keys inside `above` / `bin`
  int bin_off[kBlocks];
};
constexpr int kStateBytes = (sizeof(State) + 63) / 64 * 64;

__device__ __forceinline__ u64 make_key(float v, int idx) {
  return ((u64)__float_as_uint(v) << 20) | (u64)(0xFFFFFu - (unsigned)idx);
}
__device__ __forceinline__ int lanes_below(u64 m, int lane) { return __popcll(m & ((1ull << lane) - 1ull)); }

// LDS counter += 1 for every lane with `on`.  The lanes that share the bin of the first active lane are counted with
// ONE atomic, then the same once more for what is left (a row that is mostly zeros, or equal scores, would otherwise
// serialise up to 64 atomics on one address); the remaining lanes add one each.
__device__ __forceinline__ void count_digit(int *h, bool on, int d, int lane) {
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const u64 act = __ballot(on);
    if (act == 0) return;
    const int first = __ffsll((long long)act) - 1;
    const int d0 = __shfl(d, first);
    const bool mine = on && d == d0;
    const u64 same = __ballot(mine);
    if (lane == first) atomicAdd(&h[d0], __popcll(same));
    on = on && !mine;
  }
  if (on) atomicAdd(&h[d], 1);
}

// wave-aggregated append: every lane with `on` gets a distinct slot counted from *counter (LDS), one atomic per wave
__device__ __forceinline__ int append_slot(int *counter, bool on, int lane) {
  const u64 m = __ballot(on);
  if (m == 0) return 0;
  const int first = __ffsll((long long)m) - 1;
  int pos = 0;
  if (lane == first) pos = atomicAdd(counter, __popcll(m));
  return __shfl(pos, first) + lanes_below(m, lane);
}

// 1024 threads; h: LDS counts of nb <= 2048 bins.  Inclusive suffix sums (number of keys in bins >= b) of this
// thread's two bins b0 = nb-1-2t and b1 = b0-1 (reversed order: two per thread, wave scan, wave totals in wt[16]).
__device__ __forceinline__ void block_suffix(const int *h, int nb, int *wt, int &b0, int &b1, int &v0, int &v1, int &incl0,
                                             int &incl1) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  b0 = nb - 1 - 2 * t;
  b1 = b0 - 1;
  v0 = b0 >= 0 ? h[b0] : 0;
  v1 = b1 >= 0 ? h[b1] : 0;
  const int s = v0 + v1;
  int inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wt[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wt[w];
  incl1 = base + inc;
  incl0 = incl1 - v1;
}

// the bin T with  #(bins > T) < need <= #(bins >= T);  res[0] = T, res[1] = #(bins > T).  Ends with a barrier.
__device__ __forceinline__ void block_pick(const int *h, int nb, int need, int *wt, int *res) {
  int b0, b1, v0, v1, incl0, incl1;
  block_suffix(h, nb, wt, b0, b1, v0, v1, incl0, incl1);
  if (incl0 - v0 < need && need <= incl0) {
    res[0] = b0;
    res[1] = incl0 - v0;
  }
  if (incl0 < need && need <= incl1) {
    res[0] = b1;
    res[1] = incl0;
  }
  __syncthreads();
}

__device__ __forceinline__ void slice_of(int blk, int N, int &lo, int &hi) {
  const int per = (N + kBlocks - 1) / kBlocks;
  lo = min(blk * per, N);
  hi = min(lo + per, N);
}

// step 1: histogram of the top digit over this block's slice, stored as suffix sums
__global__ __launch_bounds__(kNT) void hist_kernel(const float *__restrict__ scores, int *__restrict__ part, int N) {
  __shared__ int lh[kBins];
  __shared__ int wt[16];
  const int b = blockIdx.y, t = threadIdx.x, lane = t & 63;
  for (int i = t; i < kBins; i += kNT) lh[i] = 0;
  __syncthreads();
  const float *s = scores + (size_t)b * N;
  int lo, hi;
  slice_of(blockIdx.x, N, lo, hi);
  constexpr int U = 4;
  for (int base = lo; base < hi; base += kNT * U) {                        // wave-uniform trip count; loads first
    unsigned bits[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * kNT + t;
      bits[u] = __float_as_uint(s[min(i, hi - 1)]);                         // clamped, not predicated: the loads stay batched
    }
#pragma unroll
    for (int u = 0; u < U; ++u) count_digit(lh, base + u * kNT + t < hi, (int)(bits[u] >> 21), lane);
  }
  __syncthreads();
  int b0, b1, v0, v1, incl0, incl1;
  block_suffix(lh, kBins, wt, b0, b1, v0, v1, incl0, incl1);
  int *dst = part + ((size_t)b * kBlocks + blockIdx.x) * kBins;
  dst[b0] = incl0;                                                         // kBins = 2 * kNT: both bins exist
  dst[b1] = incl1;
}

// step 2: one block per sample
__global__ __launch_bounds__(kNT) void pick_kernel(const int *__restrict__ part, unsigned char *__restrict__ states, int k) {
  __shared__ __align__(16) int h[kBins + 4];                               // total suffix sums; h[kBins] = 0
  __shared__ int res[2];
  const int b = blockIdx.x, t = threadIdx.x;
  const int *p = part + (size_t)b * kBlocks * kBins;
  State *st = reinterpret_cast<State *>(states + (size_t)b * kStateBytes);
  {
    // thread = (half of the slices, four adjacent bins): 32 independent 16-B loads in two batches, each issued before its adds
    const int half = t >> 9, b4 = (t & 511) * 4;
    const int *src = p + (size_t)half * 32 * kBins + b4;
    int4 c = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int q0 = 0; q0 < 32; q0 += 16) {
      int4 v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const int4 *>(src + (q0 + q) * kBins);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        c.x += v[q].x;
        c.y += v[q].y;
        c.z += v[q].z;
        c.w += v[q].w;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (half == 0) *reinterpret_cast<int4 *>(&h[b4]) = c;
    __syncthreads();
    if (half == 1) {
      int4 o = *reinterpret_cast<int4 *>(&h[b4]);
      o.x += c.x;
      o.y += c.y;
      o.z += c.z;
      o.w += c.w;
      *reinterpret_cast<int4 *>(&h[b4]) = o;
    }
  }
  if (t == 0) h[kBins] = 0;
  __syncthreads();
  // the suffix sums fall with the bin index: T is the last bin whose suffix sum still reaches k
#pragma unroll
  for (int j = 0; j < kBins / kNT; ++j) {
    const int i = t + j * kNT;
    if (h[i] >= k && h[i + 1] < k) {
      res[0] = i;
      res[1] = h[i + 1];
    }
  }
  __syncthreads();
  const int T = res[0], above = res[1];
  if (t < kBlocks) {                                                       // one wave: two exclusive scans over the slices
    const int a = T + 1 < kBins ? p[t * kBins + T + 1] : 0, m = p[t * kBins + T] - a;
    int ia = a, im = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int oa = __shfl_up(ia, d), om = __shfl_up(im, d);
      if (t >= d) {
        ia += oa;
        im += om;
      }
    }
    st->above_off[t] = ia - a;
    st->bin_off[t] = im - m;
    if (t == kBlocks - 1) st->M = im;
    if (t == 0) {
      st->T = T;
      st->above = above;
      st->take = k - above;
    }
  }
}

// step 3: every block moves the keys of its slice that lie above / inside digit T to their flat positions
__global__ __launch_bounds__(kNT) void collect_kernel(const float *__restrict__ scores, const unsigned char *__restrict__ states,
                                                      u64 *__restrict__ above, u64 *__restrict__ bin, int N, int k) {
  __shared__ int n_above, n_bin;
  const int b = blockIdx.y, t = threadIdx.x, lane = t & 63;
  const State *st = reinterpret_cast<const State *>(states + (size_t)b * kStateBytes);
  const int T = st->T;
  u64 *ab = above + (size_t)b * k + st->above_off[blockIdx.x];
  u64 *bn = bin + (size_t)b * N + st->bin_off[blockIdx.x];
  if (t == 0) n_above = n_bin = 0;
  __syncthreads();
  const float *s = scores + (size_t)b * N;
  int lo, hi;
  slice_of(blockIdx.x, N, lo, hi);
  constexpr int U = 4;
  for (int base = lo; base < hi; base += kNT * U) {                        // wave-uniform trip count; loads first
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * kNT + t;
      v[u] = s[min(i, hi - 1)];                                            // clamped, not predicated: the loads stay batched
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * kNT + t;
      const bool on = i < hi;
      const u64 K = make_key(v[u], i);
      const int d = (int)(K >> kTopShift);
      const bool in_bin = on && d == T, is_above = on && d > T;
      const int pb = append_slot(&n_bin, in_bin, lane);
      if (in_bin) bn[pb] = K;
      const int pa = append_slot(&n_above, is_above, lane);                // fewer than k in total
      if (is_above) ab[pa] = K;
    }
  }
}

// step 4: one block per sample
__global__ __launch_bounds__(kNT) void final_kernel(const unsigned char *__restrict__ states, const u64 *__restrict__ above,
                                                    u64 *__restrict__ bin, long long *__restrict__ out_idx,
                                                    float *__restrict__ out_val, int N, int k) {
  extern __shared__ __align__(16) unsigned char dyn_lds[];                 // kCache keys
  __shared__ u64 sel[1024];
  __shared__ int h[kBins];
  __shared__ int wt[16], res[2], n_sel, n_keep;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63;
  const State *st = reinterpret_cast<const State *>(states + (size_t)b * kStateBytes);
  const int a = st->above, M = st->M, take = st->take;
  const u64 *ab = above + (size_t)b * k;
  u64 *bn = bin + (size_t)b * N;
  int Pk = 64;                                                             // the sort network covers the next power of two >= k
  while (Pk < k) Pk <<= 1;
  sel[t] = 0ull;
  __syncthreads();
  if (t < a) sel[t] = ab[t];
  constexpr int U = 4;                                                     // keys per thread and round: loads first
  // Radix select with compaction over ck[0 .. cur): per level histogram of the digit, pick, then the keys above the
  // picked digit are selected and the keys inside it move to the front (a round reads its 4096 keys, then - after the
  // barrier - writes to positions below everything already read).  Stops when what is left fits the sort network.
  // Returns how many keys `sel` holds.  Instantiated for the LDS cache and for global memory.
  auto select = [&](u64 *ck) __attribute__((always_inline)) {
    if (t == 0) n_sel = a;
    int cur = M, need = take, n = k;
    const int shifts[4] = {30, 20, 10, 0}, nbits[4] = {11, 10, 10, 10};
    for (int l = 0; l < 4; ++l) {
      const int shift = shifts[l], bits = nbits[l];
      const unsigned mask = (1u << bits) - 1u;
      for (int i = t; i < kBins; i += kNT) h[i] = 0;
      if (t == 0) n_keep = 0;
      __syncthreads();                                                     // also: the cache / the last compaction is complete
      for (int base = 0; base < cur; base += kNT * U) {                    // wave-uniform trip count
        u64 K[U];
#pragma unroll
        for (int u = 0; u < U; ++u) K[u] = ck[min(base + u * kNT + t, cur - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) count_digit(h, base + u * kNT + t < cur, (int)((unsigned)(K[u] >> shift) & mask), lane);
      }
      __syncthreads();
      block_pick(h, 1 << bits, need, wt, res);
      const int T = res[0];
      need -= res[1];
      for (int base = 0; base < cur; base += kNT * U) {
        u64 K[U];
#pragma unroll
        for (int u = 0; u < U; ++u) K[u] = ck[min(base + u * kNT + t, cur - 1)];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool on = base + u * kNT + t < cur;
          const int d = (int)((unsigned)(K[u] >> shift) & mask);
          const bool up = on && d > T, keep = on && d == T;
          const int ps = append_slot(&n_sel, up, lane);
          if (up) sel[ps] = K[u];
          const int pk = append_slot(&n_keep, keep, lane);
          if (keep) ck[pk] = K[u];
        }
      }
      __syncthreads();
      cur = n_keep;
      const int have = n_sel;
      __syncthreads();                                                     // everybody has read n_keep before the next level resets it
      if (have + cur <= Pk) {                                              // what is left fits the sort (have + need = k <= have + cur)
        for (int e = t; e < cur; e += kNT) sel[have + e] = ck[e];
        n = have + cur;
        break;
      }
    }
    return n;
  };
  int n;
  if (a + M <= Pk) {
    if (t < M) sel[a + t] = bn[t];                                         // everything still in play fits the sort
    n = a + M;
  } else if (M <= kCache) {
    u64 *cache = reinterpret_cast<u64 *>(dyn_lds);
    for (int base = 0; base < M; base += kNT * 8) {                        // loads first: one round trip per 8192 keys
      u64 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = bn[min(base + u * kNT + t, M - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u * kNT + t < M) cache[base + u * kNT + t] = v[u];
    }
    n = select(cache);
  } else {
    n = select(bn);                                                        // a bin beyond the cache: same steps in global memory
  }
  __syncthreads();
  // bitonic sort (descending) of sel[0 .. P): a key per thread in a register; partners closer than a wave meet through
  // lane shuffles, the others through LDS
  int P = 64;
  while (P < n) P <<= 1;
  u64 x = sel[t];
  for (int len = 2; len <= P; len <<= 1) {
    for (int j = len >> 1; j > 0; j >>= 1) {
      u64 y;
      if (j >= 64) {
        __syncthreads();
        sel[t] = x;
        __syncthreads();
        y = sel[t ^ j];
      } else {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)x, j), hi = (unsigned)__shfl_xor((int)(unsigned)(x >> 32), j);
        y = ((u64)hi << 32) | lo;
      }
      const bool desc = (t & len) == 0, lower = (t & j) == 0;             // lower: this thread is the partner with the smaller index
      const bool want_max = lower == desc;
      x = want_max ? (x > y ? x : y) : (x < y ? x : y);
    }
  }
  if (t < k) {
    out_idx[(size_t)b * k + t] = (long long)(0xFFFFFu - (unsigned)(x & 0xFFFFFull));
    if (out_val != nullptr) out_val[(size_t)b * k + t] = __uint_as_float((unsigned)(x >> 20));
  }
}

struct Layout {
  size_t part, states, above, bin, total;
};
static Layout layout(int B, int N, int k) {
  Layout L;
  auto up = [](size_t v) { return (v + 63) / 64 * 64; };
  L.part = 0;
  L.states = up((size_t)B * kBlocks * kBins * sizeof(int));
  L.above = L.states + (size_t)B * kStateBytes;
  L.bin = L.above + up((size_t)B * k * sizeof(u64));
  L.total = L.bin + up((size_t)B * N * sizeof(u64));
  return L;
}

}  // namespace tk
}  // namespace di

extern "C" {

long long di_topk_workspace_bytes(int B, int N, int k) {
  if (B <= 0 || N <= 0 || k <= 0) return 0;
  return (long long)di::tk::layout(B, N, k).total + 64;
}

int di_topk_fwd(const float *scores, long long *out_idx, float *out_val, void *workspace, int B, int N, int k,
                void *stream) {
  using namespace di::tk;
  DI_REQUIRE(B > 0 && N > 0 && k > 0 && k <= 1024 && k <= N, "bad top-k shape (k <= min(N, 1024))");
  DI_REQUIRE(N <= (1 << 20), "N=%d exceeds the 2^20 indices of the composite key", N);
  DI_REQUIRE(scores && out_idx && workspace, "scores, out_idx and workspace are required");
  hipStream_t s = (hipStream_t)stream;
  unsigned char *w = reinterpret_cast<unsigned char *>(workspace);
  const Layout L = layout(B, N, k);
  int *part = reinterpret_cast<int *>(w + L.part);
  unsigned char *states = w + L.states;
  u64 *above = reinterpret_cast<u64 *>(w + L.above), *bin = reinterpret_cast<u64 *>(w + L.bin);
  constexpr int cache_bytes = kCache * (int)sizeof(u64);
  static di::LdsRaised lds_raised;
  if (int rc = di::ensure_lds(lds_raised, (const void *)final_kernel, cache_bytes)) return rc;
  hipLaunchKernelGGL(hist_kernel, dim3(kBlocks, B), dim3(kNT), 0, s, scores, part, N);
  hipLaunchKernelGGL(pick_kernel, dim3(B), dim3(kNT), 0, s, part, states, k);
  hipLaunchKernelGGL(collect_kernel, dim3(kBlocks, B), dim3(kNT), 0, s, scores, states, above, bin, N, k);
  hipLaunchKernelGGL(final_kernel, dim3(B), dim3(kNT), cache_bytes, s, states, above, bin, out_idx, out_val, N, k);
  return di::check_launch("topk_fwd");
}

}  // extern "C"
