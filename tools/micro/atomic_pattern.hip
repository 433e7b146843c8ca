// Float32 atomicAdd throughput of the scatter patterns of the gradient kernels: every 16-lane group adds 128 consecutive floats
// (one texel row of a C = 128 gradient map) at a pseudo-random texel, either
//   A: lane l adds floats 8 l .. 8 l + 7 (eight instructions, lanes 32 B apart: the layout of the forward's 16-B gathers), or
//   B: lane l adds floats l, l + 16, ..., l + 112 (eight instructions, each covering 64 consecutive bytes).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/atomic_pattern.hip -o tools/micro/atomic_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void scatter(float *map, int ntexel, int iters) {
  const int l16 = threadIdx.x & 15;
  unsigned grp = (blockIdx.x * 256 + threadIdx.x) >> 4;
  unsigned s = grp * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    float *row = map + (size_t)((s >> 8) % (unsigned)ntexel) * 128;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = MODE == 0 ? l16 * 8 + e : l16 + 16 * e;
      atomicAdd(row + c, 1.0f);
    }
  }
}

template <int MODE>
static void run(const char *name, float *map, int ntexel) {
  const int blocks = 4096, iters = 64;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  hipLaunchKernelGGL(scatter<MODE>, dim3(blocks), dim3(256), 0, 0, map, ntexel, iters);
  (void)hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(scatter<MODE>, dim3(blocks), dim3(256), 0, 0, map, ntexel, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  const double n = 5.0 * blocks * 256 * iters * 8;
  printf("%-44s %8.1f G atomic adds / s   (%d texels = %.1f MB)\n", name, n / ms / 1e6, ntexel, ntexel * 512.0 / 1e6);
}

int main() {
  for (int ntexel : {134400, 32400}) {   // the 6 x 112 x 200 image maps; the 180 x 180 BEV map
    float *map;
    (void)hipMalloc((void **)&map, (size_t)ntexel * 512);
    (void)hipMemset(map, 0, (size_t)ntexel * 512);
    run<0>("A: lane = 8 consecutive channels", map, ntexel);
    run<1>("B: lane = channels l, l + 16, ...", map, ntexel);
    (void)hipFree(map);
  }
  return 0;
}
