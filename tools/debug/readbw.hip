// Calibration: read-only / write-only / copy streaming bandwidth of this GPU for a buffer of the size the
// image-side attention reads (103 MB) and for a HBM-sized one.   hipcc --offload-arch=gfx950 -O3 tools/readbw.hip -o /tmp/readbw
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void rd(const uint4 *__restrict__ p, size_t n, unsigned *out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void wr(uint4 *__restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_uint4(i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void cp(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
int main() {
  for (size_t mb : {103, 138, 1000}) {
    const size_t n = mb * 1000 * 1000 / 16;
    uint4 *a, *b; unsigned *o;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&o, 4);
    hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256 * 4, 256 * 8, 256 * 16}) {
      float ms;
      for (int k = 0; k < 3; ++k) rd<<<grid, 256>>>(a, n, o);
      hipEventRecord(e0); for (int k = 0; k < 20; ++k) rd<<<grid, 256>>>(a, n, o); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); printf("%4zu MB grid %5d  read  %.2f TB/s", mb, grid, n * 16.0 * 20 / ms / 1e9);
      for (int k = 0; k < 3; ++k) wr<<<grid, 256>>>(b, n);
      hipEventRecord(e0); for (int k = 0; k < 20; ++k) wr<<<grid, 256>>>(b, n); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); printf("   write %.2f TB/s", n * 16.0 * 20 / ms / 1e9);
      for (int k = 0; k < 3; ++k) cp<<<grid, 256>>>(a, b, n);
      hipEventRecord(e0); for (int k = 0; k < 20; ++k) cp<<<grid, 256>>>(a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); printf("   copy %.2f TB/s (read+write)\n", 2 * n * 16.0 * 20 / ms / 1e9);
    }
    hipFree(a); hipFree(b); hipFree(o);
  }
  return 0;
}
