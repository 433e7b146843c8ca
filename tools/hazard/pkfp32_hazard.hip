// Stand-alone probe (no code of this repository): do packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32) return
// wrong results while a matrix-core kernel shares the CUs?  Round 2 saw ONE of 25 000 pillars of this repo's key-table
// kernel come out different in 25-50 % of graph replays - lanes 48-63, the high half of a packed op - whenever an MFMA
// kernel ran on another stream, and cured it by compiling the library without packed-FP32 instructions.
//   build: hipcc -O3 --offload-arch=gfx950 pkfp32_hazard.hip -o pkfp32_hazard ;  run: ./pkfp32_hazard [iterations]
// Kernel `pk` runs the same affine + projection arithmetic twice per lane: once with inline-asm packed instructions, once
// as the compiler's scalar v_fma/v_mul; `mm` keeps every CU's matrix cores busy from a second stream.  The host compares
// every launch of `pk` (beside `mm`) bit for bit with a launch that ran alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { f2 d; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) { f2 d; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__global__ __launch_bounds__(256) void pk(const float *__restrict__ pts, const float *__restrict__ m, float *__restrict__ out, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const float x = pts[3 * t], y = pts[3 * t + 1], z = pts[3 * t + 2];
  f2 acc = {m[3], m[7]}, accs = acc;           // rows 0 and 1 of a 3x4 projection, packed / scalar
  acc = pk_fma(f2{m[0], m[4]}, f2{x, x}, acc); acc = pk_fma(f2{m[1], m[5]}, f2{y, y}, acc); acc = pk_fma(f2{m[2], m[6]}, f2{z, z}, acc);
  accs[0] = fmaf(m[0], x, accs[0]); accs[1] = fmaf(m[4], x, accs[1]); accs[0] = fmaf(m[1], y, accs[0]); accs[1] = fmaf(m[5], y, accs[1]);
  accs[0] = fmaf(m[2], z, accs[0]); accs[1] = fmaf(m[6], z, accs[1]);
  const float w = fmaf(m[8], x, fmaf(m[9], y, fmaf(m[10], z, m[11]))), iw = 1.f / w;
  for (int r = 0; r < 32; ++r) {                // a longer dependent chain of packed ops, as a loop of scale-and-shift
    acc = pk_fma(acc, f2{1.0009765625f, 0.9990234375f}, f2{0.5f, -0.25f});
    accs[0] = fmaf(accs[0], 1.0009765625f, 0.5f); accs[1] = fmaf(accs[1], 0.9990234375f, -0.25f);
  }
  const f2 uv = pk_mul(acc, f2{iw, iw});
  out[4 * t] = uv[0]; out[4 * t + 1] = uv[1]; out[4 * t + 2] = accs[0] * iw; out[4 * t + 3] = accs[1] * iw;
}
// variant 2: the forms hipcc's SLP vectoriser produced in the key-table kernel - IN PLACE (destination pair = source pair)
// with the halves SWAPPED by op_sel (the high result reads the low register the same instruction overwrites)
__global__ __launch_bounds__(256) void pk2(const float *__restrict__ pts, const float *__restrict__ m, float *__restrict__ out, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  f2 a = {pts[3 * t], pts[3 * t + 1]}, b = {m[0] * 0.001f, m[5] * 0.002f}, as = a;
  for (int r = 0; r < 48; ++r) {
    asm volatile("v_pk_fma_f32 %0, %0, %1, -1.0 op_sel:[1,0,0] op_sel_hi:[0,1,0]" : "+v"(a) : "v"(b));   // lo' = hi*b.lo - 1, hi' = lo*b.hi - 1
    asm volatile("v_pk_mul_f32 %0, %0, 0.5 op_sel_hi:[1,0]" : "+v"(a));
    asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(a) : "v"(b));               // swapped again
    const float lo = fmaf(as[1], b[0], -1.f) * 0.5f, hi = fmaf(as[0], b[1], -1.f) * 0.5f;
    as = f2{hi + b[0], lo + b[1]};
  }
  out[4 * t] = a[0]; out[4 * t + 1] = a[1]; out[4 * t + 2] = as[0]; out[4 * t + 3] = as[1];
}
__global__ __launch_bounds__(256) void mm(float *__restrict__ sink, int iters) {
  f4 c = {0.f, 0.f, 0.f, 0.f};
  h4 a = {(_Float16)threadIdx.x, (_Float16)1, (_Float16)2, (_Float16)3}, b = {(_Float16)0.5f, (_Float16)0.25f, (_Float16)1, (_Float16)2};
  for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  if (c[0] == 12345.678f) sink[0] = c[1];
}
__global__ void cmp(const float *__restrict__ a, const float *__restrict__ b, unsigned long long *cnt, int n, int run) {
  const int t = blockIdx.x * 256 + threadIdx.x;        // cnt[0..3]: lanes that differ, by lane quarter; cnt[4]: launches; cnt[5..]: first records
  if (t >= n) return;
  bool bad = false;
  for (int j = 0; j < 4; ++j) bad |= __float_as_uint(a[4 * t + j]) != __float_as_uint(b[4 * t + j]);
  if (bad) {
    atomicAdd(&cnt[(t & 63) >> 4], 1ull);
    if (atomicAdd(&cnt[8], 1ull) < 6) atomicExch(&cnt[9 + (atomicAdd(&cnt[6], 1ull) % 6)], ((unsigned long long)run << 32) | (unsigned)t);
    if (atomicExch(&cnt[7], (unsigned long long)run + 1) != (unsigned long long)run + 1) atomicAdd(&cnt[4], 1ull);
  }
}
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 10000, n = 25000 * 64;
  std::vector<float> hp(3 * n), hm = {1266.4f, 816.3f, 4.1f, -313.5f, -6.1f, 511.3f, -1260.9f, -720.1f, -0.01f, 0.99f, 0.008f, -0.43f};
  srand(1); for (auto &v : hp) v = (rand() / (float)RAND_MAX - 0.5f) * 100.f;
  float *pts, *m, *out, *ref, *sink; unsigned long long *cnt;
  (void)hipMalloc(&pts, 12 * n); (void)hipMalloc(&m, 48); (void)hipMalloc(&out, 16 * n); (void)hipMalloc(&ref, 16 * n); (void)hipMalloc(&sink, 4); (void)hipMalloc(&cnt, 128);
  (void)hipMemcpy(pts, hp.data(), 12 * n, hipMemcpyHostToDevice); (void)hipMemcpy(m, hm.data(), 48, hipMemcpyHostToDevice); (void)hipMemset(cnt, 0, 128);
  hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
  pk<<<n / 256, 256, 0, s1>>>(pts, m, ref, n); (void)hipDeviceSynchronize();
  std::vector<float> hr(4 * n); (void)hipMemcpy(hr.data(), ref, 16 * n, hipMemcpyDeviceToHost);
  long self_bad = 0; for (int t = 0; t < n; ++t) self_bad += memcmp(&hr[4 * t], &hr[4 * t + 2], 8) != 0;
  printf("alone: packed vs scalar differ in %ld of %d lanes (expected 0: an fma is an fma)\n", self_bad, n);
  for (int variant = 1; variant <= 2; ++variant) {
    if (variant == 2) {
      pk2<<<n / 256, 256, 0, s1>>>(pts, m, ref, n); (void)hipDeviceSynchronize();
      (void)hipMemcpy(hr.data(), ref, 16 * n, hipMemcpyDeviceToHost);
      self_bad = 0; for (int t = 0; t < n; ++t) self_bad += memcmp(&hr[4 * t], &hr[4 * t + 2], 8) != 0;
      printf("variant 2 (in place, halves swapped) alone: packed vs scalar differ in %ld of %d lanes\n", self_bad, n);
    }
    (void)hipMemset(cnt, 0, 128);
    for (int it = 0; it < iters; ++it) {
      mm<<<1024, 256, 0, s2>>>(sink, 4000);                          // ~50 us of matrix-core work on every CU
      if (variant == 1) pk<<<n / 256, 256, 0, s1>>>(pts, m, out, n);
      else pk2<<<n / 256, 256, 0, s1>>>(pts, m, out, n);
      cmp<<<n / 256, 256, 0, s1>>>(out, ref, cnt, n, it);
      if ((it & 255) == 255) (void)hipDeviceSynchronize();
    }
    (void)hipDeviceSynchronize();
    unsigned long long h[16]; (void)hipMemcpy(h, cnt, 128, hipMemcpyDeviceToHost);
    printf("variant %d beside the MFMA kernel: %llu of %d launches differ from the launch alone; lanes by quarter of the wave: %llu %llu %llu %llu\n",
           variant, h[4], iters, h[0], h[1], h[2], h[3]);
    for (int j = 0; j < 6 && j < (int)h[8]; ++j) printf("  record: run %llu lane %llu (lane in wave %llu)\n", h[9 + j] >> 32, h[9 + j] & 0xffffffffu, h[9 + j] & 63);
  }
  return 0;
}
