// Stand-alone probe 2: the ARITHMETIC of this repository's key-table kernel (affine + 4x4 projection + two IEEE divisions
// per lane, one wavefront per pillar, 120 (point, camera) slots) compiled by plain `hipcc -O3` (the SLP vectoriser packs it
// into v_pk_*_f32), run beside a matrix-core kernel on a second stream and compared bit for bit with a run alone.
//   build: hipcc -O3 --offload-arch=gfx950 keys_like.hip -o keys_like [-DNOPK: -fno-slp-vectorize equivalent, see below]
//   run:   ./keys_like [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
struct Affine { float a[9], t[3]; };
__device__ __forceinline__ void apply_affine(const Affine &f, float &x, float &y, float &z) {
  const float nx = x * f.a[0] + y * f.a[3] + z * f.a[6] + f.t[0];
  const float ny = x * f.a[1] + y * f.a[4] + z * f.a[7] + f.t[1];
  const float nz = x * f.a[2] + y * f.a[5] + z * f.a[8] + f.t[2];
  x = nx; y = ny; z = nz;
}
__device__ __forceinline__ bool project_point(const float *__restrict__ M, float x, float y, float z, float ori_H, float ori_W,
                                              float &nx, float &ny) {
  const float cx = M[0] * x + M[1] * y + M[2] * z + M[3];
  const float cy = M[4] * x + M[5] * y + M[6] * z + M[7];
  const float cz = M[8] * x + M[9] * y + M[10] * z + M[11];
  const float eps = 1e-5f, den = fmaxf(cz, eps);
  const float u = cx / den, v = cy / den;
  nx = (u / ori_W - 0.5f) * 2.f;
  ny = (v / ori_H - 0.5f) * 2.f;
  return cz > eps && nx > -1.f && nx < 1.f && ny > -1.f && ny < 1.f;
}
__global__ __launch_bounds__(256) void keys(const float *__restrict__ pillars, const int *__restrict__ num_points,
                                            const float *__restrict__ proj, const float *__restrict__ aug,
                                            float *__restrict__ out /*(P,128,2)*/, int *__restrict__ cnt, int P) {
  const int lane = threadIdx.x & 63, p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  Affine A;
  for (int i = 0; i < 9; ++i) A.a[i] = aug[i];
  for (int i = 0; i < 3; ++i) A.t[i] = aug[9 + i];
  const int np = num_points[p];
  int count = 0;
  for (int base = 0; base < 120; base += 64) {
    const int slot = base + lane, pt = slot / 6, cam = slot - pt * 6;
    bool ok = slot < 120 && pt < np;
    const bool live = ok;
    float ix = 0.f, iy = 0.f;
    if (ok) {
      const float *pp = pillars + ((size_t)p * 20 + pt) * 5;
      float x = pp[0], y = pp[1], z = pp[2], nx, ny;
      apply_affine(A, x, y, z);
      ok = project_point(proj + cam * 16, x, y, z, 448.f, 800.f, nx, ny);
      ix = ((nx + 1.f) * 200.f - 1.f) * 0.5f;
      iy = ((ny + 1.f) * 112.f - 1.f) * 0.5f;
    }
    const unsigned long long mask = __ballot(ok);
    if (slot < 128) { out[((size_t)p * 128 + slot) * 2] = live ? ix : -1.f; out[((size_t)p * 128 + slot) * 2 + 1] = live ? iy : (ok ? 1.f : -1.f); }
    count += __popcll(mask);
  }
  if (lane == 0) cnt[p] = count;
}
__global__ __launch_bounds__(256) void mm(float *__restrict__ sink, int iters) {
  f4 c = {0.f, 0.f, 0.f, 0.f};
  h4 a = {(_Float16)threadIdx.x, (_Float16)1, (_Float16)2, (_Float16)3}, b = {(_Float16)0.5f, (_Float16)0.25f, (_Float16)1, (_Float16)2};
  for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  if (c[0] == 12345.678f) sink[0] = c[1];
}
__global__ void cmp(const float *a, const float *b, unsigned long long *cnt, int n, int run) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n && __float_as_uint(a[t]) != __float_as_uint(b[t])) {
    atomicAdd(&cnt[((t >> 1) & 63) >> 4], 1ull);                 // by lane quarter of the wave (slot & 63)
    if (atomicExch(&cnt[7], (unsigned long long)run + 1) != (unsigned long long)run + 1) atomicAdd(&cnt[4], 1ull);
    if (atomicAdd(&cnt[8], 1ull) < 6) cnt[9 + atomicAdd(&cnt[6], 1ull) % 6] = ((unsigned long long)run << 32) | (unsigned)t;
  }
}
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000, P = 25000, n = P * 128 * 2;
  srand(2);
  auto rnd = [] { return rand() / (float)RAND_MAX; };
  std::vector<float> hp((size_t)P * 100), hproj(96), haug = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  std::vector<int> hn(P);
  for (int p = 0; p < P; ++p) {
    const float cx = (rnd() - 0.5f) * 100.f, cy = (rnd() - 0.5f) * 100.f;
    hn[p] = 1 + rand() % 20;
    for (int t = 0; t < 20; ++t) { float *q = &hp[((size_t)p * 20 + t) * 5]; q[0] = cx + rnd() * 0.6f; q[1] = cy + rnd() * 0.6f; q[2] = rnd() * 4.f - 3.f; q[3] = rnd(); q[4] = 0.f; }
  }
  for (int c = 0; c < 6; ++c) {                                    // six cameras looking outwards, nuScenes-like intrinsics
    const float th = c * 1.0471976f, f = 600.f, cs = cosf(th), sn = sinf(th);
    const float R[12] = {-sn, cs, 0, 0.1f * c, 0, 0, -1, 1.5f, cs, sn, 0, -0.5f};   // rows: right, down, forward
    float *M = &hproj[c * 16];
    for (int j = 0; j < 4; ++j) { M[j] = f * R[j] + 400.f * R[8 + j]; M[4 + j] = f * R[4 + j] + 224.f * R[8 + j]; M[8 + j] = R[8 + j]; M[12 + j] = j == 3; }
  }
  float *pil, *proj, *aug, *out, *ref, *sink; int *np, *cn; unsigned long long *cnt;
  (void)hipMalloc(&pil, hp.size() * 4); (void)hipMalloc(&proj, 384); (void)hipMalloc(&aug, 48); (void)hipMalloc(&out, (size_t)n * 4); (void)hipMalloc(&ref, (size_t)n * 4);
  (void)hipMalloc(&sink, 4); (void)hipMalloc(&np, P * 4); (void)hipMalloc(&cn, P * 4); (void)hipMalloc(&cnt, 128);
  (void)hipMemcpy(pil, hp.data(), hp.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(proj, hproj.data(), 384, hipMemcpyHostToDevice);
  (void)hipMemcpy(aug, haug.data(), 48, hipMemcpyHostToDevice); (void)hipMemcpy(np, hn.data(), P * 4, hipMemcpyHostToDevice); (void)hipMemset(cnt, 0, 128);
  hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
  keys<<<(P + 3) / 4, 256, 0, s1>>>(pil, np, proj, aug, ref, cn, P); (void)hipDeviceSynchronize();
  std::vector<int> hc(P); (void)hipMemcpy(hc.data(), cn, P * 4, hipMemcpyDeviceToHost);
  long valid = 0; for (int v : hc) valid += v;
  printf("alone: %ld valid (point, camera) slots in %d pillars\n", valid, P);
  for (int it = 0; it < iters; ++it) {
    mm<<<1024, 256, 0, s2>>>(sink, 4000);
    keys<<<(P + 3) / 4, 256, 0, s1>>>(pil, np, proj, aug, out, cn, P);
    cmp<<<(n + 255) / 256, 256, 0, s1>>>(out, ref, cnt, n, it);
    if ((it & 127) == 127) (void)hipDeviceSynchronize();
  }
  (void)hipDeviceSynchronize();
  unsigned long long h[16]; (void)hipMemcpy(h, cnt, 128, hipMemcpyDeviceToHost);
  printf("beside the MFMA kernel: %llu of %d launches differ from the launch alone; differing values by lane quarter: %llu %llu %llu %llu\n", h[4], iters, h[0], h[1], h[2], h[3]);
  for (int j = 0; j < 6 && j < (int)h[8]; ++j) { const unsigned t = h[9 + j] & 0xffffffffu; printf("  record: run %llu pillar %u slot %u (lane %u, camera %u) coordinate %u\n", h[9 + j] >> 32, t / 256, (t / 2) % 128, (t / 2) % 64, ((t / 2) % 128) % 6, t & 1); }
  return 0;
}
