// Hard voxelisation of a point cloud into pillars / voxels on the device - the producer of the `pts_metas`
// the hot path consumes.  Replaces spconv 2.1.21 `PointToVoxel` as wrapped by the
// reference's `SPConvVoxelization` (models/updated_modules/sparse_voxelize.py:9-70; called from
// detectors/deepinteraction.py:151-171): voxels (P, T, D) zero padded, coords (P, 3) int32 [z, y, x],
// num_points (P,) int32; points outside the range are dropped, points beyond T per voxel are dropped, at most
// `max_voxels` voxels.
//
// spconv's voxel ORDER is its hash-insertion order (implementation-defined, non-deterministic on its CUDA
// path) and WHICH T points of a crowded voxel survive is equally unspecified; here both are defined as the
// CPU ("first come") semantics: voxels in the order of their first point, the first T points of each voxel
// in point order.  Deterministic, bit-reproducible, and the consumers are permutation-invariant.
//
// Three small kernels around two library radix sorts (rocPRIM through torch):
//   voxel_keys     key = voxel id << 32 | point index   (invalid points: the maximum key)
//   voxel_heads    after sorting the keys: flag the first point of every voxel, emit its point index as the
//                  key of the second sort (which orders the voxels by first point)
//   voxel_scatter  every sorted point finds its voxel's output slot and its rank inside the voxel and copies
//                  its D features; the head writes the coordinates; the count is an atomicMax of rank + 1
#include "di_common.h"

namespace di {

constexpr long long kInvalidKey = 0x7fffffffffffffffll;

__global__ __launch_bounds__(256) void voxel_keys_kernel(const float *__restrict__ pts, int n, int stride,
                                                         const float *__restrict__ geo /* range(6), vsize(3) */,
                                                         int gx, int gy, int gz, long long *__restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = pts[(size_t)i * stride], y = pts[(size_t)i * stride + 1], z = pts[(size_t)i * stride + 2];
  // spconv: c = floor((p - range_min) / vsize), kept iff 0 <= c < grid on every axis
  const int cx = (int)floorf((x - geo[0]) / geo[6]);
  const int cy = (int)floorf((y - geo[1]) / geo[7]);
  const int cz = (int)floorf((z - geo[2]) / geo[8]);
  const bool ok = cx >= 0 && cx < gx && cy >= 0 && cy < gy && cz >= 0 && cz < gz && x == x && y == y && z == z;
  const long long vid = ((long long)cz * gy + cy) * gx + cx;
  keys[i] = ok ? ((vid << 32) | (long long)i) : kInvalidKey;
}

__global__ __launch_bounds__(256) void voxel_heads_kernel(const long long *__restrict__ sorted, int n,
                                                          int *__restrict__ head, long long *__restrict__ first_key) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long k = sorted[i];
  const bool valid = k != kInvalidKey;
  const bool h = valid && (i == 0 || (sorted[i - 1] >> 32) != (k >> 32));
  head[i] = h ? 1 : 0;
  // second sort: voxels by the index of their first point; position i rides in the low bits
  first_key[i] = h ? (((k & 0xffffffffll) << 32) | (long long)i) : kInvalidKey;
}

// slot_of_seg / head_of_seg from the second sort: rank r <-> head position hp = low 32 bits of sorted2[r]
__global__ __launch_bounds__(256) void voxel_slots_kernel(const long long *__restrict__ sorted2,
                                                          const int *__restrict__ seg_id, int n,
                                                          int *__restrict__ slot_of_seg, int *__restrict__ head_of_seg) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const long long k = sorted2[r];
  if (k == kInvalidKey) return;
  const int hp = (int)(k & 0xffffffffll);
  const int seg = seg_id[hp];
  slot_of_seg[seg] = r;
  head_of_seg[seg] = hp;
}

__global__ __launch_bounds__(256) void voxel_scatter_kernel(
    const float *__restrict__ pts, int stride, int D, const long long *__restrict__ sorted,
    const int *__restrict__ seg_id, const int *__restrict__ slot_of_seg, const int *__restrict__ head_of_seg, int n,
    int gx, int gy, int max_points, int max_voxels, float *__restrict__ voxels, int *__restrict__ coords,
    int *__restrict__ num_points) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long k = sorted[i];
  if (k == kInvalidKey) return;
  const int seg = seg_id[i];
  const int slot = slot_of_seg[seg];
  if (slot >= max_voxels) return;
  const int within = i - head_of_seg[seg];
  if (within == 0) {
    const long long vid = k >> 32;
    const int cx = (int)(vid % gx), cy = (int)((vid / gx) % gy), cz = (int)(vid / ((long long)gx * gy));
    coords[slot * 3 + 0] = cz;
    coords[slot * 3 + 1] = cy;
    coords[slot * 3 + 2] = cx;
  }
  if (within >= max_points) return;
  const int src = (int)(k & 0xffffffffll);
  for (int d = 0; d < D; ++d) voxels[((size_t)slot * max_points + within) * D + d] = pts[(size_t)src * stride + d];
  atomicMax(num_points + slot, within + 1);
}

}  // namespace di

extern "C" {

int di_voxel_keys(const float *pts, int n_pts, int pt_stride, const float *geo, int gx, int gy, int gz,
                  long long *keys, void *stream) {
  DI_REQUIRE(n_pts >= 0 && pt_stride >= 3 && gx > 0 && gy > 0 && gz > 0, "bad voxel grid");
  DI_REQUIRE((long long)gx * gy * gz < (1ll << 31), "voxel grid %d x %d x %d too large", gx, gy, gz);
  if (n_pts == 0) return DI_OK;
  hipLaunchKernelGGL(di::voxel_keys_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts, n_pts,
                     pt_stride, geo, gx, gy, gz, keys);
  return di::check_launch("voxel_keys");
}

int di_voxel_heads(const long long *sorted_keys, int n_pts, int32_t *head, long long *first_key, void *stream) {
  if (n_pts <= 0) return DI_OK;
  hipLaunchKernelGGL(di::voxel_heads_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     sorted_keys, n_pts, head, first_key);
  return di::check_launch("voxel_heads");
}

int di_voxel_scatter(const float *pts, int n_pts, int pt_stride, int n_feat, const long long *sorted_keys,
                     const long long *sorted_first, const int32_t *seg_id, int32_t *slot_of_seg, int32_t *head_of_seg,
                     int gx, int gy, int max_points, int max_voxels, float *voxels, int32_t *coords,
                     int32_t *num_points, void *stream) {
  DI_REQUIRE(n_feat >= 3 && n_feat <= pt_stride && max_points > 0 && max_voxels > 0, "bad voxel output shape");
  if (n_pts <= 0) return DI_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(di::voxel_slots_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, s, sorted_first, seg_id, n_pts,
                     slot_of_seg, head_of_seg);
  hipLaunchKernelGGL(di::voxel_scatter_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, s, pts, pt_stride, n_feat,
                     sorted_keys, seg_id, slot_of_seg, head_of_seg, n_pts, gx, gy, max_points, max_voxels, voxels,
                     coords, num_points);
  return di::check_launch("voxel_scatter");
}

}  // extern "C"
