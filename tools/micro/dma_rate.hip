// How many LDS-DMA instructions (global_load_lds_dwordx4, 1 KB each) does ONE wavefront get through per microsecond, and
// how does the CU's intake scale with the number of wavefronts that issue them?  Every wavefront copies its own slice of an
// L2-resident buffer into its own 8 KB of LDS, 8 instructions in flight (counted vmcnt), nothing else.  Modes: M0 rewritten
// before every instruction (what a ring of arbitrary LDS rows needs) or M0 fixed + immediate offsets.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dma_rate.hip -o /tmp/dma_rate && /tmp/dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void dma_m0(const void *gp, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gp), "s"(lds_addr) : "memory", "m0");
}

template <int MODE>
__global__ __launch_bounds__(1024) void dma_rate(const unsigned char *__restrict__ src, unsigned *sink, int iters, int stride_b) {
  extern __shared__ __align__(1024) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const unsigned l0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds + wave * 8192;
  // the wave's 64 KB window of the source (L2 resident: grid * nw * 64 KB in total), walked 1 KB per instruction
  const unsigned char *base = src + ((size_t)(blockIdx.x * nw + wave) << 16) + lane * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned char *gp = base + (((it * 8 + j) * stride_b) & 0xFFFF);
      if (MODE == 0) {
        dma_m0(gp, __builtin_amdgcn_readfirstlane(l0 + j * 1024));
      } else {   // same M0 for the 8 instructions: the immediate offset moves BOTH addresses, so pre-subtract it from the pointer
        const unsigned char *g2 = gp - j * 1024;
        if (j == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(__builtin_amdgcn_readfirstlane(l0)) : "memory", "m0");
        switch (j) {
#define C(J) case J: asm volatile("global_load_lds_dwordx4 %0, off offset:%1" : : "v"(g2), "n"(J * 1024 > 4095 ? 0 : J * 1024) : "memory"); break;
          C(0) C(1) C(2) C(3)
#undef C
          default: dma_m0(gp, __builtin_amdgcn_readfirstlane(l0 + j * 1024)); break;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[threadIdx.x] == 0x7f && lds[threadIdx.x + 4096] == 0x7e) sink[0] = 1;
}

// the same bytes through registers (global_load_dwordx4 + ds_write_b128), 8 loads in flight per lane
__global__ __launch_bounds__(1024) void reg_rate(const unsigned char *__restrict__ src, unsigned *sink, int iters, int stride_b) {
  extern __shared__ __align__(1024) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  unsigned char *l0 = lds + wave * 8192 + lane * 16;
  const unsigned char *base = src + ((size_t)(blockIdx.x * nw + wave) << 16) + lane * 16;
  for (int it = 0; it < iters; ++it) {
    uint4 r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = *reinterpret_cast<const uint4 *>(base + (((it * 8 + j) * stride_b) & 0xFFFF));
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<uint4 *>(l0 + j * 1024) = r[j];
  }
  __syncthreads();
  if (lds[threadIdx.x] == 0x7f && lds[threadIdx.x + 4096] == 0x7e) sink[0] = 1;
}

int main() {
  const size_t bytes = (size_t)256 * 16 * 65536;   // 256 MB: every wave of the largest launch has its own 64 KB
  unsigned char *src;
  unsigned *sink;
  hipMalloc((void **)&src, bytes);
  hipMemset(src, 1, bytes);
  hipMalloc((void **)&sink, 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 256;   // 2048 instructions = 2 MB per wave (its 64 KB window 32 times: L2 hits after the first pass)
  for (int mode = 0; mode < 3; ++mode)
    for (int nw = 1; nw <= 16; nw *= 2) {
      const int lds_b = nw * 8192;
      auto run = [&]() {
        if (mode == 0) hipLaunchKernelGGL(dma_rate<0>, dim3(256), dim3(nw * 64), lds_b, 0, src, sink, iters, 1024);
        else if (mode == 1) hipLaunchKernelGGL(dma_rate<1>, dim3(256), dim3(nw * 64), lds_b, 0, src, sink, iters, 1024);
        else hipLaunchKernelGGL(reg_rate, dim3(256), dim3(nw * 64), lds_b, 0, src, sink, iters, 1024);
      };
      run();
      hipEventRecord(a);
      for (int r = 0; r < 5; ++r) run();
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      const double us = ms * 1e3 / 5, instr = (double)iters * 8;
      printf("%s  %2d waves/CU: %8.1f us  = %6.1f ns per 1 KB instruction and wave, %6.2f TB/s into LDS (chip), %5.1f B/clk/CU @2.4GHz\n",
             mode == 0 ? "LDS-DMA, M0 per instruction " : mode == 1 ? "LDS-DMA, M0 fixed + offsets  " : "registers + ds_write_b128   ", nw,
             us, us * 1e3 / instr, 256.0 * nw * instr * 1024 / us / 1e6, 256.0 * nw * instr * 1024 / us / 1e6 * 1e12 / 256 / 2.4e9);
    }
  return 0;
}
