// How fast can the CUs take in a 9x9-window halo?  Every workgroup walks 8 x 8-query tiles of a 6 x 112 x 200 x 128 fp16 map
// exactly as the window-attention kernel does (XCD-contiguous ranges, 16 x 16-texel halo of K and of V = 4 x the map
// through L2 -> L1, 1 x from HBM) and does nothing with the data: 16 B per lane loads, SEG contiguous bytes per texel and
// pass (64 / 128 / 256), results folded into one register.  Prints us per launch and the L2 -> L1 rate.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/halo_read.hip -o /tmp/halo_read && /tmp/halo_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int SEG, int TWQ, int THQ>
__global__ __launch_bounds__(256) void halo_read(const uint4 *__restrict__ k, const uint4 *__restrict__ v, unsigned *sink, int n,
                                                 int H, int W) {
  constexpr int HC = TWQ + 8, HR = THQ + 8, CPT = SEG / 16, NCHUNK = HC * HR * CPT, NLD = (NCHUNK + 255) / 256;
  const int tiles_x = (W + TWQ - 1) / TWQ, tiles_y = (H + THQ - 1) / THQ, per_img = tiles_x * tiles_y, ntiles = n * per_img;
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, gxw = ((int)gridDim.x - xcd + 7) >> 3;
  const int t_end = (int)(((long long)ntiles * (xcd + 1)) >> 3);
  unsigned acc = 0;
  for (int tile = (int)(((long long)ntiles * xcd) >> 3) + wl; tile < t_end; tile += gxw) {
    const int img = tile / per_img, r = tile % per_img, y0 = (r / tiles_x) * THQ, x0 = (r % tiles_x) * TWQ;
    for (int op = 0; op < 2; ++op) {
      const uint4 *src = op ? v : k;
      for (int seg = 0; seg < 256 / SEG; ++seg) {
        uint4 R[NLD];
#pragma unroll
        for (int s = 0; s < NLD; ++s) {
          const int e = min(s * 256 + (int)threadIdx.x, NCHUNK - 1);
          const int tex = e / CPT, c = e % CPT, hr = tex / HC, hc = tex % HC;
          const int gy = min(max(y0 - 4 + hr, 0), H - 1), gx = min(max(x0 - 4 + hc, 0), W - 1);
          R[s] = src[((size_t)(img * H + gy) * W + gx) * 16 + seg * CPT + c];
        }
#pragma unroll
        for (int s = 0; s < NLD; ++s) acc ^= R[s].x ^ R[s].y ^ R[s].z ^ R[s].w;
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int SEG, int TWQ, int THQ>
static void run(const char *name, int wg_per_cu, const uint4 *k[3], const uint4 *v[3], unsigned *sink) {
  const int n = 6, H = 112, W = 200;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int grid = 256 * wg_per_cu;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((halo_read<SEG, TWQ, THQ>), dim3(grid), dim3(256), 0, 0, k[i], v[i], sink, n, H, W);
  hipEventRecord(a);
  const int reps = 30;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((halo_read<SEG, TWQ, THQ>), dim3(grid), dim3(256), 0, 0, k[i % 3], v[i % 3], sink, n, H, W);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / reps;
  const int tiles = n * ((W + TWQ - 1) / TWQ) * ((H + THQ - 1) / THQ);
  const double l1 = 2.0 * tiles * (TWQ + 8) * (THQ + 8) * 256, hbm = 2.0 * n * H * W * 256;
  printf("%-44s %d WG/CU  %7.2f us   L2->L1 %6.1f MB = %5.2f TB/s   HBM %5.1f MB = %5.2f TB/s\n", name, wg_per_cu, us, l1 / 1e6,
         l1 / us / 1e6, hbm / 1e6, hbm / us / 1e6);
}

int main() {
  const size_t bytes = (size_t)6 * 112 * 200 * 256;
  const uint4 *k[3], *v[3];
  for (int i = 0; i < 3; ++i) {   // rotate over 3 x 69 MB x 2 > Infinity Cache
    void *p, *q;
    hipMalloc(&p, bytes);
    hipMalloc(&q, bytes);
    hipMemset(p, 1, bytes);
    hipMemset(q, 2, bytes);
    k[i] = (const uint4 *)p;
    v[i] = (const uint4 *)q;
  }
  unsigned *sink;
  hipMalloc((void **)&sink, 4);
  for (int wg = 2; wg <= 8; wg *= 2) {
    run<64, 8, 8>("8x8 tiles, 64 B per texel and pass", wg, k, v, sink);
    run<128, 8, 8>("8x8 tiles, 128 B per texel and pass", wg, k, v, sink);
    run<256, 8, 8>("8x8 tiles, 256 B per texel and pass", wg, k, v, sink);
    run<128, 16, 8>("16x8 tiles, 128 B per texel and pass", wg, k, v, sink);
    run<128, 16, 16>("16x16 tiles, 128 B per texel and pass", wg, k, v, sink);
    run<256, 32, 16>("32x16 tiles, 256 B per texel and pass", wg, k, v, sink);
  }
  return 0;
}
