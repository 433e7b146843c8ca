/* ORACLE - TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's only native code, the `locatt_ops`
 * local-window attention kernels:
 *   /root/reference/projects/mmdet3d_plugin/models/utils/ops/locatt_ops/
 *     kernels.cuh:4-42   cc2k      (window correlation)
 *     kernels.cuh:44-80  ck2c_ori  (window aggregation)
 *     kernels.cuh:82-119 ck2c_loc  (transposed-window aggregation)
 *   and the five host entry points (similar.cu:3-92, weighting.cu:3-122).
 *
 * Same conventions as the reference: NCHW contiguous float32 tensors, weights
 * (B,H,W,kH*kW), window slot k <-> (dy,dx) = (k / kW - rH, k % kW - rW),
 * out-of-image slots contribute 0 (and ARE written as 0 by cc2k),
 * accumulation in double (similar.cu:25 `f_cc2k<float,double>`), channel
 * accumulation order c = 0..C-1, window accumulation order k = 0..K-1.
 *
 * Pinned bit-for-bit against the reference kernels compiled for the host CPU
 * (oracle/_ref/liblocatt_ref.so, built by oracle/Makefile) in
 * tests/test_oracle_locatt.py.
 */
#include <stddef.h>

/* y[h,w,k] = sum_c x_ori[c,h,w] * x_loc[c,h+dy,w+dx]      (kernels.cuh:20-41) */
static void cc2k(const float *x_ori, const float *x_loc, int kH, int kW, int C,
                 int H, int W, float *y) {
  const int rH = kH >> 1, rW = kW >> 1, patch = kH * kW, per_channel = H * W;
  for (int indexO = 0; indexO < per_channel; ++indexO) {
    const int w_ori = indexO % W - rW;
    const int h_ori = indexO / W - rH;
    for (int indexK = 0; indexK < patch; ++indexK) {
      const int w = w_ori + indexK % kW;
      const int h = h_ori + indexK / kW;
      double val = 0.0;
      if (h > -1 && h < H && w > -1 && w < W) {
        const float *p_ori = x_ori + indexO;
        const float *p_loc = x_loc + h * W + w;
        for (int c = 0; c < C; ++c) {
          val += (double)(*p_ori * *p_loc); /* float product, double sum (:33) */
          p_ori += per_channel;
          p_loc += per_channel;
        }
      }
      y[(size_t)indexO * patch + indexK] = (float)val;
    }
  }
}

/* y[c,h,w] = sum_k x_loc[c,h+dy,w+dx] * weight[h,w,k]      (kernels.cuh:61-79) */
static void ck2c_ori(const float *x_loc, const float *x_weight, int kH, int kW,
                     int C, int H, int W, float *y) {
  const int rH = kH >> 1, rW = kW >> 1, patch = kH * kW, per_channel = H * W;
  const int per_inp = per_channel * C;
  for (int index = 0; index < per_inp; ++index) {
    const int index_ = index % per_channel;
    const int w_ori = index_ % W - rW;
    const int h_ori = index_ / W - rH;
    const float *p_weight = x_weight + (size_t)index_ * patch;
    const float *p_loc = x_loc + index - index_;
    double val = 0.0;
    for (int indexK = 0; indexK < patch; ++indexK) {
      const int w = w_ori + indexK % kW;
      const int h = h_ori + indexK / kW;
      if (h > -1 && h < H && w > -1 && w < W)
        val += (double)(p_loc[W * h + w] * p_weight[indexK]);
    }
    y[index] = (float)val;
  }
}

/* y[c,h,w] = sum_k x_ori[c,h-dy,w-dx] * weight[(h-dy,w-dx),k]  (kernels.cuh:99-118) */
static void ck2c_loc(const float *x_ori, const float *x_weight, int kH, int kW,
                     int C, int H, int W, float *y) {
  const int rH = kH >> 1, rW = kW >> 1, patch = kH * kW, per_channel = H * W;
  const int per_inp = per_channel * C;
  for (int index = 0; index < per_inp; ++index) {
    const int index_ = index % per_channel;
    const int w_ori = index_ % W + rW;
    const int h_ori = index_ / W + rH;
    const float *p_ori = x_ori + index - index_;
    double val = 0.0;
    for (int indexK = 0; indexK < patch; ++indexK) {
      const int w = w_ori - indexK % kW;
      const int h = h_ori - indexK / kW;
      const int indexW = W * h + w;
      if (h > -1 && h < H && w > -1 && w < W)
        val += (double)(p_ori[indexW] * x_weight[(size_t)indexW * patch + indexK]);
    }
    y[index] = (float)val;
  }
}

/* ---- the five entry points of localAttention.cpp:61-73, batch loop as in
 * similar.cu:23-37 / weighting.cu:24-38 ---- */

void oracle_similar_forward(const float *x_ori, const float *x_loc, int B, int C,
                            int H, int W, int kH, int kW, float *out) {
  const size_t per_input = (size_t)C * H * W, per_output = (size_t)H * W * kH * kW;
  for (int i = 0; i < B; ++i)
    cc2k(x_ori + i * per_input, x_loc + i * per_input, kH, kW, C, H, W,
         out + i * per_output);
}

/* similar.cu:43-92: is_ori -> ck2c_ori(x, grad), else ck2c_loc(x, grad) */
void oracle_similar_backward(const float *x, const float *grad_out, int B, int C,
                             int H, int W, int kH, int kW, int is_ori,
                             float *grad_inp) {
  const size_t per_input = (size_t)C * H * W, per_output = (size_t)H * W * kH * kW;
  for (int i = 0; i < B; ++i) {
    if (is_ori)
      ck2c_ori(x + i * per_input, grad_out + i * per_output, kH, kW, C, H, W,
               grad_inp + i * per_input);
    else
      ck2c_loc(x + i * per_input, grad_out + i * per_output, kH, kW, C, H, W,
               grad_inp + i * per_input);
  }
}

void oracle_weighting_forward(const float *x_ori, const float *x_weight, int B,
                              int C, int H, int W, int kH, int kW, float *out) {
  const size_t per_input = (size_t)C * H * W, per_output = (size_t)H * W * kH * kW;
  for (int i = 0; i < B; ++i)
    ck2c_ori(x_ori + i * per_input, x_weight + i * per_output, kH, kW, C, H, W,
             out + i * per_input);
}

/* weighting.cu:44-81: ck2c_loc(grad_out, weight) */
void oracle_weighting_backward_ori(const float *x_weight, const float *grad_out,
                                   int B, int C, int H, int W, int kH, int kW,
                                   float *grad_ori) {
  const size_t per_input = (size_t)C * H * W, per_output = (size_t)H * W * kH * kW;
  for (int i = 0; i < B; ++i)
    ck2c_loc(grad_out + i * per_input, x_weight + i * per_output, kH, kW, C, H, W,
             grad_ori + i * per_input);
}

/* weighting.cu:85-122: cc2k(grad_out, x_ori) */
void oracle_weighting_backward_weight(const float *x_ori, const float *grad_out,
                                      int B, int C, int H, int W, int kH, int kW,
                                      float *grad_weight) {
  const size_t per_input = (size_t)C * H * W, per_output = (size_t)H * W * kH * kW;
  for (int i = 0; i < B; ++i)
    cc2k(grad_out + i * per_input, x_ori + i * per_input, kH, kW, C, H, W,
         grad_weight + i * per_output);
}
