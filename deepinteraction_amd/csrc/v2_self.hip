// DeepInteraction++ head: the "self" branch of the V2 RoI blocks (reference decoder_utils.py:970-990 `ImageRCNNBlockV2`,
// :1086-1089 `PointRCNNBlockV2`) in ONE launch.
//
// As published, the mix `query * scale + self_feat * self_scale` broadcasts to (n, n, C) and row 0 is kept: EVERY query of a
// group receives the self-branch feature of the group's FIRST query.  The branch therefore works on one token per (sample,
// view) - at most 6 per sample - or one per sample (point block): the first query of the view attends to the view's queries
// (8 heads x 16), output projection + residual + norm1, FFN 128 -> hidden -> 128 (ReLU) + residual + LayerNorm, times
// `self_scale`.  Rounds 3-4 evaluated this with ~35 float32 torch launches per image block (arange / gather / compare /
// soft-max / four library GEMMs of 6 rows / layer norms): ~150 us of launch latency per block on the critical path of a
// forward and 70 of the 248 graph nodes.  Here: one workgroup per (sample, view) [image] or per sample [point], float32
// throughout, weights read once from L2 (0.6 MB per workgroup), and the result scattered to the rows of the queries that take
// it.  Latency-bound by construction (<= 6 workgroups): 1 024 threads per workgroup, every matrix-vector product split so that a
// thread issues <= 16 independent 16-byte loads and partial sums meet through lane shuffles (256 threads with 32-128 dependent
// loads each took 54 us per image block).
#include "di_common.h"

namespace di {
namespace v2s {

constexpr int NT = 1024, NWV = NT / 64, MAXQ = 512, MAXH = 1024;   // 16 wavefronts: every phase is a few loads per thread, all in flight

struct Args {
  const float *qk;              // (B*Q, 256) rows [q | k] of the block's packed self-attention projection   [image]
  const float *vt;              // (B, 128, Qp) the values, transposed                                        [image]
  const float *x;               // image: (B*Q, 128) the block's input tokens; point: (B*Q, 128) norm1(x + attention)
  const signed char *view;      // (B*Q) last view of a query, -1 = none                                       [image]
  const unsigned char *member;  // (B*Q) bit v = the query is seen by view v                                   [image]
  const float *wo, *bo, *n1w, *n1b;                       // attention output projection, norm1              [image]
  const float *sw1, *sb1, *sw2, *sb2, *snw, *snb;         // self FFN (hidden x 128, 128 x hidden), its LayerNorm
  const float *self_scale;      // (1)
  float *out;                   // (B*Q, 128)
  float scale, eps1, eps_s;
  int B, Q, Qp, V, hidden, image;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
  return v;
}

// LayerNorm statistics of 128 values held by threads 0..127 (two wavefronts): (mean, rstd) for every thread
__device__ __forceinline__ void ln_stats(float t, int tid, float eps, float *red, float &mean, float &rstd) {
  const float s = wave_sum(tid < 128 ? t : 0.f);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  mean = (red[0] + red[1]) * (1.f / 128.f);
  __syncthreads();
  const float d = tid < 128 ? t - mean : 0.f;
  const float s2 = wave_sum(d * d);
  if ((tid & 63) == 0) red[tid >> 6] = s2;
  __syncthreads();
  rstd = rsqrtf((red[0] + red[1]) * (1.f / 128.f) + eps);
  __syncthreads();
}

__global__ __launch_bounds__(NT) void v2_self_feature_kernel(Args A) {
  __shared__ float s_sc[8][MAXQ];
  __shared__ float s_vec[128], s_y[256], s_h[MAXH], s_red[NWV];
  __shared__ int s_first[NWV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Q = A.Q;
  const int b = A.image ? blockIdx.x / A.V : blockIdx.x, v = A.image ? blockIdx.x % A.V : 0;
  float yf = 0.f;                                            // threads 0..127: channel tid of the branch's token
  if (A.image) {
    // ---- the first query of the view (Q: the view is unused - then every query is a key, decoder_utils.py:976)
    int f = Q;
    for (int q = tid; q < Q; q += NT)
      if ((A.member[b * Q + q] >> v) & 1) f = min(f, q);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) f = min(f, __shfl_xor(f, s));
    if (lane == 0) s_first[wave] = f;
    __syncthreads();
    int first = s_first[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) first = min(first, s_first[w]);
    const int firstc = min(first, Q - 1);
    const bool all_keys = first >= Q;
    if (tid < 128) s_vec[tid] = A.qk[(size_t)(b * Q + firstc) * 256 + tid];
    __syncthreads();
    // ---- scores of the 8 heads over the view's queries
    for (int e = tid; e < 8 * Q; e += NT) {
      const int q = e >> 3, h = e & 7;
      const bool ok = all_keys || ((A.member[b * Q + q] >> v) & 1);
      float s = -INFINITY;
      if (ok) {
        const float4 *kp = reinterpret_cast<const float4 *>(A.qk + (size_t)(b * Q + q) * 256 + 128 + h * 16);
        const float *qp = s_vec + h * 16;
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 k4 = kp[j];
          d = fmaf(qp[4 * j], k4.x, d);
          d = fmaf(qp[4 * j + 1], k4.y, d);
          d = fmaf(qp[4 * j + 2], k4.z, d);
          d = fmaf(qp[4 * j + 3], k4.w, d);
        }
        s = d * A.scale;
      }
      s_sc[h][q] = s;
    }
    __syncthreads();
    // ---- soft-max per head (wave w < 8: head w)
    if (wave < 8) {
      const int h = wave;
      float m = -INFINITY;
      for (int q = lane; q < Q; q += 64) m = fmaxf(m, s_sc[h][q]);
      m = wave_max(m);
      float sum = 0.f;
      for (int q = lane; q < Q; q += 64) {
        const float p = __expf(s_sc[h][q] - m);
        s_sc[h][q] = p;
        sum += p;
      }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      for (int q = lane; q < Q; q += 64) s_sc[h][q] *= inv;
    }
    __syncthreads();
    // ---- attention output (channel tid), then the output projection + residual + norm1
    {                                                        // channel c = tid / 8, eight threads share its keys
      const int c = tid >> 3, part = tid & 7;
      const float *vr = A.vt + ((size_t)b * 128 + c) * A.Qp;
      const float *p = s_sc[c >> 4];
      float o = 0.f;
      for (int q = part; q < Q; q += 8) o = fmaf(p[q], vr[q], o);
      o += __shfl_xor(o, 1);
      o += __shfl_xor(o, 2);
      o += __shfl_xor(o, 4);
      if (part == 0) s_y[c] = o;
    }
    __syncthreads();
    {                                                        // output projection: row c = tid / 8, 16 columns per thread
      const int c = tid >> 3, part = tid & 7;
      const float4 *wr = reinterpret_cast<const float4 *>(A.wo + (size_t)c * 128 + part * 16);
      const float *yp = s_y + part * 16;
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 w4 = wr[j];
        a = fmaf(w4.x, yp[4 * j], a);
        a = fmaf(w4.y, yp[4 * j + 1], a);
        a = fmaf(w4.z, yp[4 * j + 2], a);
        a = fmaf(w4.w, yp[4 * j + 3], a);
      }
      a += __shfl_xor(a, 1);
      a += __shfl_xor(a, 2);
      a += __shfl_xor(a, 4);
      if (part == 0) s_h[c] = a;                             // (s_h is free until the FFN)
    }
    __syncthreads();
    float t = 0.f;
    if (tid < 128) t = A.x[(size_t)(b * Q + firstc) * 128 + tid] + (s_h[tid] + A.bo[tid]);
    float mean, rstd;
    ln_stats(t, tid, A.eps1, s_red, mean, rstd);
    if (tid < 128) yf = (t - mean) * rstd * A.n1w[tid] + A.n1b[tid];
  } else {
    if (tid < 128) yf = A.x[(size_t)(b * Q) * 128 + tid];    // query 0's norm1 output (decoder_utils.py:1086)
  }
  if (tid < 128) s_vec[tid] = yf;
  __syncthreads();
  // ---- self FFN: relu(W1 y + b1), W2 h + b2, residual, LayerNorm, times self_scale
  for (int o = tid >> 1; o < A.hidden; o += NT / 2) {         // two threads per hidden unit, 64 inputs each
    const int part = tid & 1;
    const float4 *wr = reinterpret_cast<const float4 *>(A.sw1 + (size_t)o * 128 + part * 64);
    const float *yp = s_vec + part * 64;
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float4 w4 = wr[j];
      a = fmaf(w4.x, yp[4 * j], a);
      a = fmaf(w4.y, yp[4 * j + 1], a);
      a = fmaf(w4.z, yp[4 * j + 2], a);
      a = fmaf(w4.w, yp[4 * j + 3], a);
    }
    a += __shfl_xor(a, 1);
    if (part == 0) s_h[o] = fmaxf(a + A.sb1[o], 0.f);
  }
  __syncthreads();
  {
    const int c = tid >> 3, part = tid & 7, hh = A.hidden >> 3;         // eight threads per output channel
    const float4 *wr = reinterpret_cast<const float4 *>(A.sw2 + (size_t)c * A.hidden + part * hh);
    const float *hp = s_h + part * hh;
    float a = 0.f;
#pragma unroll 16
    for (int j = 0; j < hh / 4; ++j) {
      const float4 w4 = wr[j];
      a = fmaf(w4.x, hp[4 * j], a);
      a = fmaf(w4.y, hp[4 * j + 1], a);
      a = fmaf(w4.z, hp[4 * j + 2], a);
      a = fmaf(w4.w, hp[4 * j + 3], a);
    }
    a += __shfl_xor(a, 1);
    a += __shfl_xor(a, 2);
    a += __shfl_xor(a, 4);
    if (part == 0) s_y[c] = a;
  }
  __syncthreads();
  float t2 = 0.f;
  if (tid < 128) t2 = yf + (s_y[tid] + A.sb2[tid]);
  float mean2, rstd2;
  ln_stats(t2, tid, A.eps_s, s_red, mean2, rstd2);
  if (tid < 128) s_vec[tid] = ((t2 - mean2) * rstd2 * A.snw[tid] + A.snb[tid]) * A.self_scale[0];
  __syncthreads();
  // ---- every query that takes this token's feature: its view is v (queries without a view read view 0's, as
  // `view.clamp(min=0)` does - they are zeroed by `keep` afterwards); point block: all of them
  for (int e = tid; e < Q * 128; e += NT) {
    const int q = e >> 7, c = e & 127;
    if (!A.image || max((int)A.view[b * Q + q], 0) == v) A.out[(size_t)(b * Q + q) * 128 + c] = s_vec[c];
  }
}

}  // namespace v2s
}  // namespace di

extern "C" int di_v2_self_feature(const float *qk, const float *vt, const float *x, const signed char *view,
                                  const unsigned char *member, const float *wo, const float *bo, const float *n1w,
                                  const float *n1b, float eps1, float scale, const float *sw1, const float *sb1,
                                  const float *sw2, const float *sb2, const float *snw, const float *snb, float eps_s,
                                  const float *self_scale, float *out, int B, int Q, int Qp, int V, int hidden, int image,
                                  void *stream) {
  using namespace di::v2s;
  DI_REQUIRE(B > 0 && Q > 0 && Q <= MAXQ, "1 .. %d queries, got %d", MAXQ, Q);
  DI_REQUIRE(hidden > 0 && hidden <= MAXH && hidden % 32 == 0, "hidden width %d unsupported (multiple of 32, <= %d)", hidden, MAXH);
  DI_REQUIRE(x && sw1 && sb1 && sw2 && sb2 && snw && snb && self_scale && out, "null argument");
  if (image) {
    DI_REQUIRE(qk && vt && view && member && wo && bo && n1w && n1b && V > 0 && V <= 8 && Qp >= Q, "image form: null argument / bad shape");
  }
  Args A{qk, vt, x, view, member, wo, bo, n1w, n1b, sw1, sb1, sw2, sb2, snw, snb, self_scale, out, scale, eps1, eps_s,
         B, Q, Qp, V, hidden, image};
  hipLaunchKernelGGL(v2_self_feature_kernel, dim3(image ? B * V : B), dim3(NT), 0, (hipStream_t)stream, A);
  return di::check_launch("v2_self_feature");
}
