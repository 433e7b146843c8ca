"""Drop-in for the reference's JIT-built CUDA extension `locatt_ops`
(`models/utils/ops/locatt_ops/__init__.py:22-26`, pybind module `localattention`,
`localAttention.cpp:61-73`): the same five entry points with the same positional
signatures, backed by the gfx950 HIP kernels behind the C ABI.

    from ...ops import locatt_ops
    locatt_ops.localattention.similar_forward(x_ori, x_loc, kH, kW)

Inputs: (B,C,H,W) float32/float16 HIP tensors (any memory format; channels-last is the
native one), window tensors (B,H,W,kH*kW) float32.  Outputs are fresh tensors on the
current stream, feature maps in channels-last memory format.
"""
from deepinteraction_amd import ops as _ops


class _LocalAttention:
    similar_forward = staticmethod(_ops.similar_forward)
    similar_backward = staticmethod(_ops.similar_backward)
    weighting_forward = staticmethod(_ops.weighting_forward)
    weighting_backward_ori = staticmethod(_ops.weighting_backward_ori)
    weighting_backward_weight = staticmethod(_ops.weighting_backward_weight)


localattention = _LocalAttention()
