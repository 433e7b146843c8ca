// Pairwise 3-D IoU of LiDAR boxes for the Hungarian cost matrix of the head loss (training):
// `BboxOverlaps3D(coordinate='lidar')` as the reference assigner calls it (core/bbox/assigners/hungarian_assigner.py:127,
// mmdet3d 0.17.1 iou3d: rotated BEV intersection x height overlap / union).  One thread per (box, ground truth) pair:
// Sutherland-Hodgman clipping of rectangle A by the four half-planes of rectangle B in float64 (<= 8 vertices), the
// same sequence of operations as the vectorised torch statement it replaces (det3d_compat.rotated_intersection_area:
// ~120 launches per call, 1.7 ms of host time in front of the Hungarian solve).
#include "di_common.h"

namespace di {

__device__ __forceinline__ void rect_corners(double x, double y, double dx, double dy, double yaw, double (&px)[4],
                                             double (&py)[4]) {
  const double hx = dx * 0.5, hy = dy * 0.5;
  const double c = cos(yaw), s = sin(yaw);
  const double lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};
#pragma unroll
  for (int v = 0; v < 4; ++v) {                      // x' = x cos + y sin, y' = -x sin + y cos: a positive yaw turns clockwise
    px[v] = x + lx[v] * c + ly[v] * s;
    py[v] = y - lx[v] * s + ly[v] * c;
  }
}

__global__ __launch_bounds__(256) void iou3d_lidar_kernel(const float *__restrict__ b1, int n, int s1,
                                                          const float *__restrict__ b2, int m, int s2,
                                                          float *__restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n * m) return;
  const float *a = b1 + (long long)(t / m) * s1, *b = b2 + (long long)(t % m) * s2;
  double px[4], py[4], qx[4], qy[4];
  rect_corners(a[0], a[1], a[3], a[4], a[6], px, py);
  rect_corners(b[0], b[1], b[3], b[4], b[6], qx, qy);
  constexpr int M = 8;
  double x[M], y[M], nx[M], ny[M];
  int cnt = 4;
#pragma unroll
  for (int v = 0; v < 4; ++v) x[v] = px[v], y[v] = py[v];
  for (int e = 0; e < 4; ++e) {
    const double ax = qx[e], ay = qy[e], ex = qx[(e + 1) & 3] - ax, ey = qy[(e + 1) & 3] - ay;   // clip edge, inside = left
    int k = 0;
    for (int v = 0; v < cnt; ++v) {
      const int w = v + 1 < cnt ? v + 1 : 0;
      const double sx = x[v], sy = y[v], tx = x[w], ty = y[w];
      const double ds = ex * (sy - ay) - ey * (sx - ax), de = ex * (ty - ay) - ey * (tx - ax);
      const bool s_in = ds >= 0, e_in = de >= 0;
      if (s_in != e_in && k < M) {                   // the edge crosses the clip line
        const double den = fabs(ds - de) > 1e-300 ? ds - de : 1.0, u = ds / den;
        nx[k] = sx + u * (tx - sx);
        ny[k] = sy + u * (ty - sy);
        ++k;
      }
      if (e_in && k < M) {
        nx[k] = tx;
        ny[k] = ty;
        ++k;
      }
    }
    cnt = k;
    for (int v = 0; v < cnt; ++v) x[v] = nx[v], y[v] = ny[v];
  }
  double area = 0;
  for (int v = 0; v < cnt; ++v) {
    const int w = v + 1 < cnt ? v + 1 : 0;
    area += x[v] * y[w] - x[w] * y[v];
  }
  const float bev = cnt >= 3 ? (float)(0.5 * fabs(area)) : 0.f;
  const float top = fminf(a[2] + a[5], b[2] + b[5]), bot = fmaxf(a[2], b[2]);
  const float ov = bev * fmaxf(top - bot, 0.f);
  const float va = a[3] * a[4] * a[5], vb = b[3] * b[4] * b[5];
  out[t] = ov / fmaxf(va + vb - ov, 1e-8f);
}

}  // namespace di

extern "C" int di_iou3d_lidar(const float *boxes1, int n, int stride1, const float *boxes2, int m, int stride2, float *out,
                              void *stream) {
  DI_REQUIRE(n >= 0 && m >= 0 && stride1 >= 7 && stride2 >= 7, "boxes are rows of >= 7 floats (x, y, z, dx, dy, dz, yaw)");
  if (n == 0 || m == 0) return DI_OK;
  DI_REQUIRE((long long)n * m < (1ll << 31), "%d x %d pairs", n, m);
  hipLaunchKernelGGL(di::iou3d_lidar_kernel, dim3((unsigned)(((long long)n * m + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, boxes1, n, stride1, boxes2, m, stride2, out);
  return di::check_launch("iou3d_lidar");
}
