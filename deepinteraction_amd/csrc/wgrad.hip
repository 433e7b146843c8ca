// Weight gradient of the 1x1 convolutions of the training step (reference encoder_utils.py:11-34 ConvBNReLU in train()
// mode; the backward of every `conv` there with kernel_size 1), float32:
//
//     dW[co][ci] = sum_p gy[p][co] * x[p][ci]          (+ db[co] = sum_p gy[p][co])          p < P = 134 400 / 32 400 pixels
//
// A reduction over 10^5 pixels into a 128 x 128 result.  As a library GEMM it is a handful of workgroups; as the slab-batched
// GEMM of round 3 (autograd.PixelLinear: torch.bmm over pixel slabs + sum) hipBLASLt runs it at 10 TFLOP/s - 418 us per
// call, 16 calls = 6.7 of the 40.7 ms of kernels of a float32 step (profiles/r05y_train_f32_kernel_tail.txt).  The bytes
// say 138 MB = ~30 us, the float32 matrix cores (v_mfma_f32_16x16x4_f32, 256 FLOP/clk/CU) 28 us.
//
// Here: one workgroup per (pixel slab, 128 x 128 block of dW), four wavefronts in 2 x 2, each owning 64 x 64 outputs as
// 4 x 4 MFMA tiles.  ONE workgroup per CU = one wavefront per SIMD: measured 50.5 us on 134 400 pixels x 128 x 128 (87
// TFLOP/s of the 155 the float32 matrix cores have) against 66.7 with two workgroups per CU and 66.9 with eight-wavefront
// workgroups that split the slab's pixels and add the halves through LDS (session r05z).  The contraction index of the MFMA is the PIXEL: lane (i = lane % 16, k = lane / 16) loads ONE float4 of
// gy - channels 4i .. 4i + 3 of pixel p0 + k - and ONE float4 of x; element t of the gy vector is the A operand of the tiles
// whose row i stands for channel 4i + t, element u of the x vector the B operand of the tiles whose column j stands for
// channel 4j + u: two 16-byte loads per lane feed 16 MFMAs (4 pixels x 64 x 64), every load instruction of a wave reads four
// runs of 256 contiguous bytes, nothing goes through LDS.  Eight k-steps (32 pixels) are in flight per wave while the
// previous eight are multiplied.  The slab partials are written out and summed in a fixed order by a second launch:
// bit-reproducible, no atomics.
#include "di_common.h"
#include <stdlib.h>

namespace di {
namespace wg {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int UN = 8;                    // k-steps (of 4 pixels) per stage
constexpr int STAGE = 4 * UN;            // pixels per stage

__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                           float *__restrict__ part, float *__restrict__ bpart, long long P,
                                                           int Cin, int Cout, int rows_per_slab) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, k = lane >> 4;
  const int slab = blockIdx.x;
  const int co0 = blockIdx.y * 128 + wm * 64, ci0 = blockIdx.z * 128 + wn * 64;
  const long long p_lo = (long long)slab * rows_per_slab;
  const long long p_hi = min(P, p_lo + rows_per_slab);       // > p_lo: the host sizes the grid so
  const int nfull = (int)((p_hi - p_lo) / STAGE);            // stages without a pixel past the slab

  f4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  f4 bsum = f4{0.f, 0.f, 0.f, 0.f};

  // addresses = a UNIFORM 64-bit base (SGPRs, advanced per k-step) + a 32-bit lane offset that never changes: no address
  // VGPRs are written inside the loop (the first version computed 64-bit lane addresses per load; the register allocator
  // reused destination registers of loads in flight for them and the compiler's wait insertion answered with vmcnt(0) in
  // the MIDDLE of a stage's loads - no stage was ever in flight while another was multiplied: 2.1 TB/s)
  const unsigned aoff = (unsigned)((k * Cout + co0 + 4 * i) * 4), boff = (unsigned)((k * Cin + ci0 + 4 * i) * 4);
  const char *gbase = reinterpret_cast<const char *>(gy + p_lo * Cout), *xbase = reinterpret_cast<const char *>(x + p_lo * Cin);
  // a stage's 2 x UN vectors of this lane.  No arithmetic touches a loaded value here: a select on it would make the
  // compiler wait for the load where it is issued, and the stage in flight would no longer overlap the one multiplied
  auto fetch = [&](f4 (&A)[UN], f4 (&B)[UN], int st) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long row = (long long)st * STAGE + 4 * u;
      A[u] = *reinterpret_cast<const f4 *>(gbase + row * Cout * 4 + aoff);
      B[u] = *reinterpret_cast<const f4 *>(xbase + row * Cin * 4 + boff);
    }
  };
  auto multiply = [&](const f4 (&A)[UN], const f4 (&B)[UN]) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      bsum += A[u];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[u][a], B[u][b], acc[a][b], 0, 0, 0);
    }
  };

  // Two stages per trip, no branch inside: [fetch s+1 | multiply s | fetch s+2 | multiply s+1].  (The scheduling barriers keep
  // a stage's loads in front of the previous stage's MFMAs: left alone the scheduler sinks them between the MFMAs to save
  // registers; with a branch in the body the compiler rotated the loop into [multiply | 32 loads | multiply] - one of the two
  // multiplications then ran with nothing in flight behind it.)
  f4 A0[UN], B0[UN], A1[UN], B1[UN];
  const int npair = nfull >> 1;
  if (nfull > 0) fetch(A0, B0, 0);
  __builtin_amdgcn_sched_barrier(0);
  for (int pr = 0; pr < npair; ++pr) {
    fetch(A1, B1, 2 * pr + 1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(A0, B0);
    __builtin_amdgcn_sched_barrier(0);
    fetch(A0, B0, min(2 * pr + 2, nfull - 1));               // (nfull even, last trip: a re-read that is never multiplied)
    __builtin_amdgcn_sched_barrier(0);
    multiply(A1, B1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (nfull & 1) multiply(A0, B0);                           // stage nfull - 1
  // the ragged last stage: pixels past the slab read the slab's last pixel and count as zero
  if (p_lo + (long long)nfull * STAGE < p_hi) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long p = p_lo + (long long)nfull * STAGE + 4 * u + k;
      const bool ok = p < p_hi;
      const long long pc = ok ? p : p_hi - 1;
      const f4 a = *reinterpret_cast<const f4 *>(gy + pc * Cout + co0 + 4 * i);
      const f4 b = *reinterpret_cast<const f4 *>(x + pc * Cin + ci0 + 4 * i);
      A0[u] = ok ? a : f4{0.f, 0.f, 0.f, 0.f};
      B0[u] = ok ? b : f4{0.f, 0.f, 0.f, 0.f};
    }
    multiply(A0, B0);
  }

  // tile (a, b), register r of lane (i, k): dW[co0 + 4 (4k + r) + a][ci0 + 4 i + b]; the four b are one 16-byte store
  const long long rec = (long long)Cout * Cin + Cout;        // a slab's record: dW block | db
  float *dst = part + slab * rec + (long long)co0 * Cin + ci0 + 4 * i;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f4 v = f4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
      *reinterpret_cast<f4 *>(dst + (long long)(4 * (4 * k + r) + a) * Cin) = v;
    }
  if (bpart != nullptr && wn == 0 && blockIdx.z == 0) {      // column sums of gy: the four k groups of the wave
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = bsum[t];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      bsum[t] = v;
    }
    if (k == 0) *reinterpret_cast<f4 *>(bpart + slab * rec + co0 + 4 * i) = bsum;
  }
}

// out[e] = sum over the slabs of part[slab][e], e < n4 float4s of a slab's record, slabs added in a fixed order: 64 vectors x
// 16 slab lanes per workgroup (slab lane s adds slabs s, s + 16, ... in order), then the 16 lanes' sums in order.  Vectors
// e < n4a go to out_a, the others (the bias gradient) to out_b.
__global__ __launch_bounds__(1024) void slab_sum_kernel(const float *__restrict__ part, float *__restrict__ out_a,
                                                        float *__restrict__ out_b, int nslab, long long n4, long long n4a) {
  __shared__ f4 red[16][64];
  const int c = threadIdx.x & 63, s = threadIdx.x >> 6;
  const long long e = (long long)blockIdx.x * 64 + c;
  f4 v = f4{0.f, 0.f, 0.f, 0.f};
  if (e < n4)
    for (int sl = s; sl < nslab; sl += 16) v += *reinterpret_cast<const f4 *>(part + ((long long)sl * n4 + e) * 4);
  red[s][c] = v;
  __syncthreads();
  if (s == 0 && e < n4) {
    f4 t = red[0][c];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][c];
    if (e < n4a) *reinterpret_cast<f4 *>(out_a + e * 4) = t;
    else if (out_b != nullptr) *reinterpret_cast<f4 *>(out_b + (e - n4a) * 4) = t;
  }
}

static int slab_rows(long long P, int Cin, int Cout) {
  const int n_cu = di::device_cus();
  if (n_cu <= 0) return -1;
  const long long blocks = (long long)(Cout / 128) * (Cin / 128);
  static const int per_cu = [] {                               // workgroups per CU (measurement switch)
    const char *e = getenv("DI_WGRAD_WG_PER_CU");
    return e != nullptr && atoi(e) > 0 ? atoi(e) : 1;
  }();
  const long long want = max(1ll, (long long)per_cu * n_cu / blocks);
  long long rows = (P + want - 1) / want;
  rows = (rows + STAGE - 1) / STAGE * STAGE;
  return (int)max((long long)STAGE, rows);
}

}  // namespace wg
}  // namespace di

extern "C" long long di_wgrad_workspace_floats(long long npix, int Cin, int Cout) {
  if (npix <= 0 || Cin <= 0 || Cout <= 0 || Cin % 128 != 0 || Cout % 128 != 0) {
    di::set_error("di_wgrad: npix %lld, Cin %d, Cout %d (channel counts must be multiples of 128)", npix, Cin, Cout);
    return -1;
  }
  const int rows = di::wg::slab_rows(npix, Cin, Cout);
  if (rows <= 0) return -1;
  const long long nslab = (npix + rows - 1) / rows;
  return nslab * ((long long)Cout * Cin + Cout);
}

extern "C" int di_wgrad_f32(const float *x, const float *grad_y, long long npix, int Cin, int Cout, float *grad_w,
                            float *grad_b, float *workspace, void *stream) {
  DI_REQUIRE(npix > 0 && Cin > 0 && Cout > 0 && Cin % 128 == 0 && Cout % 128 == 0,
             "di_wgrad_f32: npix %lld, Cin %d, Cout %d (channel counts must be multiples of 128)", npix, Cin, Cout);
  DI_REQUIRE(x != nullptr && grad_y != nullptr && grad_w != nullptr && workspace != nullptr, "di_wgrad_f32: null pointer");
  const int rows = di::wg::slab_rows(npix, Cin, Cout);
  if (rows <= 0) return DI_ERR_LAUNCH;
  const int nslab = (int)((npix + rows - 1) / rows);
  const long long rec = (long long)Cout * Cin + Cout;
  float *part = workspace, *bpart = workspace + (long long)Cout * Cin;       // inside every slab's record
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(di::wg::wgrad_f32_kernel, dim3(nslab, Cout / 128, Cin / 128), dim3(256), 0, s, x, grad_y, part,
                     grad_b != nullptr ? bpart : nullptr, npix, Cin, Cout, rows);
  if (int rc = di::check_launch("wgrad_f32_kernel")) return rc;
  const long long n4a = (long long)Cout * Cin / 4, n4 = grad_b != nullptr ? rec / 4 : n4a;
  hipLaunchKernelGGL(di::wg::slab_sum_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(1024), 0, s, part, grad_w, grad_b, nslab,
                     rec / 4, n4a);
  if (int rc = di::check_launch("slab_sum_kernel")) return rc;
  return DI_OK;
}
