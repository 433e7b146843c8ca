// Top-Q proposals of the NMS-ed heat map (deepinteraction_decoder.py:242: `heatmap.view(B,-1).argsort(descending)
// [..., :Q]`; torch's sort-based top-k of 324 000 scores is 18 launches / ~118 us per forward).
//
// Radix SELECT instead of a sort.  Every element gets the 52-bit key
//        K = float_bits(score) << 20 | (0xFFFFF - index)          (scores >= 0: their bit patterns are ordered)
// so all keys are distinct and "value descending, lower index first on ties" is plain descending K - a
// deterministic rule where the reference's argsort leaves ties unspecified.  Five digit levels (11+11+10 value bits,
// 10+10 index bits), each a multi-block histogram of the elements that still match the selected prefix and a
// one-block pick of the digit that holds the k-th largest key; then the blocks collect the exactly k keys >= K*
// into per-block segments and one block bitonic-sorts them.  12 small launches, no host synchronisation, no global atomics,
// bit-reproducible.
#include "di_common.h"

namespace di {
namespace tk {

// NO global atomics and no buffer that is zeroed by one kernel and accumulated into by the next: every block writes
// its own partial histogram with plain stores and the pick kernel sums the partials.  (The first version accumulated
// one global histogram with atomicAdd after a plain-store clear by the previous kernel; replayed from a hipGraph that
// went wrong from the second replay on - atomics and plain stores to the same lines do not meet in the same cache
// level across the XCDs - while eager launches, with their full cache maintenance at every boundary, were fine.)
constexpr int kBins = 2048;
constexpr int kHistBlocks = 32;      // partial histograms per sample
struct State {
  unsigned long long prefix;   // the digits selected so far, right aligned
  int need;                    // how many keys of the selected bin are still to be taken
  int pad;
};

__device__ __forceinline__ unsigned long long make_key(float v, int idx) {
  return ((unsigned long long)__float_as_uint(v) << 20) | (unsigned long long)(0xFFFFFu - (unsigned)idx);
}

// partial histogram of digit (K >> shift) & (2^bits - 1) over this block's slice of the elements whose higher bits
// equal state.prefix
__global__ __launch_bounds__(256) void hist_kernel(const float *__restrict__ scores, const State *__restrict__ st,
                                                   int *__restrict__ part, int N, int shift, int bits, int first) {
  __shared__ int lh[kBins];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < kBins; i += 256) lh[i] = 0;
  __syncthreads();
  const unsigned long long prefix = first ? 0ull : st[b].prefix;
  const unsigned mask = (1u << bits) - 1u;
  const float *s = scores + (size_t)b * N;
  const int per = (N + kHistBlocks - 1) / kHistBlocks;
  const int lo = blockIdx.x * per, hi = min(lo + per, N);
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const unsigned long long K = make_key(s[i], i);
    if ((K >> (shift + bits)) == prefix) atomicAdd(&lh[(unsigned)(K >> shift) & mask], 1);
  }
  __syncthreads();
  int *dst = part + ((size_t)b * kHistBlocks + blockIdx.x) * kBins;
  for (int i = threadIdx.x; i < kBins; i += 256) dst[i] = lh[i];
}

// one block per sample: the bin T with  #(bins > T) < need <= #(bins >= T); prefix <- prefix:T, need -= #(bins > T)
__global__ __launch_bounds__(1024) void pick_kernel(State *__restrict__ st, const int *__restrict__ part, int bits, int k,
                                                    int first) {
  __shared__ int suf[kBins];
  const int b = blockIdx.x, nb = 1 << bits, t = threadIdx.x;
  const int *h = part + (size_t)b * kHistBlocks * kBins;
  // the previous level's selection is read ahead of the scan: its barriers separate these loads from the single
  // thread that rewrites st[b] at the end
  const int need = first ? k : st[b].need;
  const unsigned long long prefix = first ? 0ull : st[b].prefix;
  // suffix sums over reversed bins (Hillis-Steele on <= 2048 entries, two per thread)
  for (int i = t; i < kBins; i += 1024) {
    int c = 0;
    if (i < nb)
      for (int p = 0; p < kHistBlocks; ++p) c += h[p * kBins + nb - 1 - i];
    suf[i] = c;                                                                  // suf[r]: bin nb-1-r
  }
  __syncthreads();
  for (int d = 1; d < nb; d <<= 1) {
    int v0 = 0, v1 = 0;
    const int i0 = t, i1 = t + 1024;
    if (i0 >= d) v0 = suf[i0 - d];
    if (i1 < kBins && i1 >= d) v1 = suf[i1 - d];
    __syncthreads();
    suf[i0] += v0;
    if (i1 < kBins) suf[i1] += v1;
    __syncthreads();
  }
  for (int r = t; r < nb; r += 1024) {
    const int incl = suf[r], excl = r ? suf[r - 1] : 0;
    if (excl < need && need <= incl) {
      st[b].prefix = (prefix << bits) | (unsigned long long)(nb - 1 - r);
      st[b].need = need - excl;
    }
  }
}

// every block gathers the keys >= K* of its slice (all keys are distinct; exactly k exist in total) into its own
// segment of `cand` (LDS counter, plain stores) and records how many it found
__global__ __launch_bounds__(256) void collect_kernel(const float *__restrict__ scores, const State *__restrict__ st,
                                                      unsigned long long *__restrict__ cand, int *__restrict__ cand_n,
                                                      int N, int k) {
  __shared__ int cnt;
  const int b = blockIdx.y;
  const unsigned long long Kstar = st[b].prefix;   // all 52 bits selected
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const float *s = scores + (size_t)b * N;
  const int per = (N + kHistBlocks - 1) / kHistBlocks;
  const int lo = blockIdx.x * per, hi = min(lo + per, N);
  unsigned long long *dst = cand + ((size_t)b * kHistBlocks + blockIdx.x) * k;
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const unsigned long long K = make_key(s[i], i);
    if (K >= Kstar) {
      const int pos = atomicAdd(&cnt, 1);          // LDS atomic: order irrelevant, the sort follows
      if (pos < k) dst[pos] = K;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) cand_n[b * kHistBlocks + blockIdx.x] = min(cnt, k);
}

// one block per sample: concatenate the segments (k keys in total), bitonic-sort them (descending), write the indices
__global__ __launch_bounds__(1024) void sort_kernel(const unsigned long long *__restrict__ cand, const int *__restrict__ cand_n,
                                                    long long *__restrict__ out_idx, float *__restrict__ out_val, int k) {
  __shared__ unsigned long long a[1024];
  __shared__ int off[kHistBlocks + 1];
  const int b = blockIdx.x, t = threadIdx.x;
  a[t] = 0ull;
  if (t == 0) {
    int o = 0;
    for (int p = 0; p < kHistBlocks; ++p) {
      off[p] = o;
      o += cand_n[b * kHistBlocks + p];
    }
    off[kHistBlocks] = o;
  }
  __syncthreads();
  for (int e = t; e < kHistBlocks * k; e += 1024) {
    const int p = e / k, j = e - p * k;
    if (j < off[p + 1] - off[p] && off[p] + j < 1024) a[off[p] + j] = cand[((size_t)b * kHistBlocks + p) * k + j];
  }
  __syncthreads();
  for (int len = 2; len <= 1024; len <<= 1) {
    for (int j = len >> 1; j > 0; j >>= 1) {
      const int p = t ^ j;
      if (p > t) {
        const bool desc = (t & len) == 0;
        const unsigned long long x = a[t], y = a[p];
        if ((x < y) == desc) {
          a[t] = y;
          a[p] = x;
        }
      }
      __syncthreads();
    }
  }
  if (t < k) {
    const unsigned long long K = a[t];
    out_idx[(size_t)b * k + t] = (long long)(0xFFFFFu - (unsigned)(K & 0xFFFFFull));
    if (out_val != nullptr) out_val[(size_t)b * k + t] = __uint_as_float((unsigned)(K >> 20));
  }
}

}  // namespace tk
}  // namespace di

extern "C" {

long long di_topk_workspace_bytes(int B, int k) {
  return (long long)B * ((long long)di::tk::kHistBlocks * (di::tk::kBins * (long long)sizeof(int) + (long long)k * 8 + 4) + 64) + 64;
}

int di_topk_fwd(const float *scores, long long *out_idx, float *out_val, void *workspace, int B, int N, int k,
                void *stream) {
  using namespace di::tk;
  DI_REQUIRE(B > 0 && N > 0 && k > 0 && k <= 1024 && k <= N, "bad top-k shape (k <= min(N, 1024))");
  DI_REQUIRE(N <= (1 << 20), "N=%d exceeds the 2^20 indices of the composite key", N);
  hipStream_t s = (hipStream_t)stream;
  unsigned char *w = reinterpret_cast<unsigned char *>(workspace);
  int *part = reinterpret_cast<int *>(w);
  State *st = reinterpret_cast<State *>(w + (size_t)B * kHistBlocks * kBins * sizeof(int));
  unsigned long long *cand = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(st) + (size_t)B * 64);
  int *cand_n = reinterpret_cast<int *>(cand + (size_t)B * kHistBlocks * k);
  const int shifts[5] = {41, 30, 20, 10, 0}, nbits[5] = {11, 11, 10, 10, 10};
  for (int l = 0; l < 5; ++l) {
    hipLaunchKernelGGL(hist_kernel, dim3(kHistBlocks, B), dim3(256), 0, s, scores, st, part, N, shifts[l], nbits[l], l == 0);
    hipLaunchKernelGGL(pick_kernel, dim3(B), dim3(1024), 0, s, st, part, nbits[l], k, l == 0);
  }
  hipLaunchKernelGGL(collect_kernel, dim3(kHistBlocks, B), dim3(256), 0, s, scores, st, cand, cand_n, N, k);
  hipLaunchKernelGGL(sort_kernel, dim3(B), dim3(1024), 0, s, cand, cand_n, out_idx, out_val, k);
  return di::check_launch("topk_fwd");
}

}  // extern "C"
