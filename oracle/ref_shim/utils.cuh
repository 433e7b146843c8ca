// ORACLE BUILD SHIM - TEST INFRASTRUCTURE.
// Stands in for the reference's locatt_ops/utils.cuh (which needs CUDA + torch
// headers) so that the reference's kernels.cuh device code can be compiled for
// the HOST cpu by oracle/Makefile into oracle/_ref/liblocatt_ref.so.
// One "thread" of one "block": the grid-stride loops of the kernels then run
// the whole problem serially, in the reference's own statement order.
#pragma once
#include <cmath>
#include <cstddef>
struct ref_dim3 { int x, y, z; };
static ref_dim3 blockIdx = {0, 0, 0}, threadIdx = {0, 0, 0};
static ref_dim3 blockDim = {1, 1, 1}, gridDim = {1, 1, 1};
#define __global__ static
#define __ldg(p) (*(p))
// loop shapes of utils.cuh:12-16
#define KERNEL_LOOP(i, I) for (int i = threadIdx.x; i < (I); i += blockDim.x)
#define KERNEL_LOOP1d(i, I) \
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (I); i += gridDim.x * blockDim.x)
