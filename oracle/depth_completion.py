"""Oracle restatement of ip_basic `fill_in_multiscale` (TEST INFRASTRUCTURE).

Follows /root/reference/projects/mmdet3d_plugin/models/utils/ip_basic/
depth_map_utils.py:134-287 as it is called from encoder_utils.py:175-182
(extrapolate=False, blur_type='bilateral', default kernels), with the OpenCV
calls restated in oracle/thirdparty.py (parity unpinned at the cv2 boundary).
"""
import numpy as np

from .thirdparty import (cv_bilateral_filter, cv_dilate, cv_median_blur5,
                         cv_morph_close)

FULL_KERNEL_5 = np.ones((5, 5), np.uint8)
FULL_KERNEL_9 = np.ones((9, 9), np.uint8)


def _cross(n):
    k = np.zeros((n, n), np.uint8)
    k[n // 2, :] = 1
    k[:, n // 2] = 1
    return k


CROSS_KERNEL_3, CROSS_KERNEL_5, CROSS_KERNEL_7 = _cross(3), _cross(5), _cross(7)


def fill_in_multiscale(depth_map, max_depth=100.0):
    """(H,W) float32 sparse depth -> dense depth (depth_map_utils.py:134-287)."""
    depths_in = np.float32(depth_map)                                      # :163
    valid_near = (depths_in > 0.1) & (depths_in <= 15.0)                   # :166
    valid_med = (depths_in > 15.0) & (depths_in <= 30.0)                   # :167
    valid_far = depths_in > 30.0                                           # :168

    s1 = np.copy(depths_in)                                                # :171-174
    valid = s1 > 0.1
    s1[valid] = max_depth - s1[valid]

    dil_far = cv_dilate(np.multiply(s1, valid_far), CROSS_KERNEL_3)        # :177-185
    dil_med = cv_dilate(np.multiply(s1, valid_med), CROSS_KERNEL_5)
    dil_near = cv_dilate(np.multiply(s1, valid_near), CROSS_KERNEL_7)
    valid_near, valid_med, valid_far = dil_near > 0.1, dil_med > 0.1, dil_far > 0.1

    s2 = np.copy(s1)                                                       # :193-196
    s2[valid_far] = dil_far[valid_far]
    s2[valid_med] = dil_med[valid_med]
    s2[valid_near] = dil_near[valid_near]

    s3 = cv_morph_close(s2, FULL_KERNEL_5)                                 # :199-200

    s4 = np.copy(s3)                                                       # :203-206
    blurred = cv_median_blur5(s3)
    valid = s3 > 0.1
    s4[valid] = blurred[valid]

    H, W = s4.shape                                                        # :209-213
    top_mask = np.ones((H, W), dtype=bool)
    first = np.argmax(s4 > 0.1, axis=0)        # 0 when the column has no valid pixel
    rows = np.arange(H)[:, None]
    top_mask[rows < first[None, :]] = False

    valid = s4 > 0.1                                                       # :216-222
    empty = ~valid & top_mask
    dilated = cv_dilate(s4, FULL_KERNEL_9)
    s5 = np.copy(s4)
    s5[empty] = dilated[empty]

    s6 = np.copy(s5)                                                       # :225-238 (extrapolate False)
    first = np.argmax(s5 > 0.1, axis=0)
    top_mask = np.ones((H, W), dtype=bool)
    top_mask[rows < first[None, :]] = False

    s7 = np.copy(s6)                                                       # :241-245
    for _ in range(6):
        empty = (s7 < 0.1) & top_mask
        dilated = cv_dilate(s7, FULL_KERNEL_5)
        s7[empty] = dilated[empty]

    blurred = cv_median_blur5(s7)                                          # :248-250
    valid = (s7 > 0.1) & top_mask
    s7[valid] = blurred[valid]

    blurred = cv_bilateral_filter(s7, 5, 0.5, 2.0)                         # :259-260 (same `valid`)
    s7[valid] = blurred[valid]

    s8 = np.copy(s7)                                                       # :263-266
    valid = s8 > 0.1
    s8[valid] = max_depth - s8[valid]
    return s8
