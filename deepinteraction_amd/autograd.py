"""torch.autograd bindings of the gather operators (training step).

Forward = the inference kernels; backward = the float32-atomic scatter kernels of csrc/backward.hip.  Only
feature maps receive gradients: sampling coordinates are functions of points, metas and detached box
predictions, exactly as in the reference (F.grid_sample's grid at encoder_utils.py:195,:297 is built from
data; the RoI boxes come from deep-copied, decoded predictions, decoder_utils.py:672-679)."""
import torch
from torch.amp import custom_bwd, custom_fwd

from . import ops

# Mixed precision (torch.autocast): the gather kernels take fp16 or float32 maps as they come; `custom_fwd` / `custom_bwd` make
# a Function's backward run under the autocast state of its forward (PixelLinear's GEMMs then run in the forward's dtype).
_fwd = custom_fwd(device_type='cuda')
_bwd = custom_bwd(device_type='cuda')

import os
OWN_WGRAD = os.environ.get('DI_OWN_WGRAD', '1') != '0'       # float32 1x1 weight gradients through csrc/wgrad.hip (A/B switch)


class BEVWarpGather(torch.autograd.Function):
    """warped = bilinear(bev, unproject(depth)) masked to pc_range  (encoder_utils.py:183-196)."""

    @staticmethod
    @_fwd
    def forward(ctx, bev, depth, img2lidar, aug_fwd, xs, ys, pc_range):
        ctx.save_for_backward(depth, img2lidar, aug_fwd, xs, ys, pc_range)
        ctx.bev_hw, ctx.bev_dtype = tuple(bev.shape[-2:]), bev.dtype
        return ops.bevwarp_gather(bev, depth, img2lidar, aug_fwd, xs, ys, pc_range)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        depth, img2lidar, aug_fwd, xs, ys, pc_range = ctx.saved_tensors
        g = ops.bevwarp_gather_bwd(grad_out, depth, img2lidar, aug_fwd, xs, ys, pc_range, ctx.bev_hw)
        return g.to(ctx.bev_dtype), None, None, None, None, None, None


class RoIAlign(torch.autograd.Function):
    """detectron2 ROIAlign(7x7, ratio 2, aligned) on channels-last maps, output (R,49,C)."""

    @staticmethod
    @_fwd
    def forward(ctx, feat, rois, spatial_scale):
        ctx.save_for_backward(rois)
        ctx.shape, ctx.dtype, ctx.scale = tuple(feat.shape), feat.dtype, float(spatial_scale)
        return ops.roi_align(feat, rois, spatial_scale)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        (rois,) = ctx.saved_tensors
        g = ops.roi_align_bwd(grad_out, rois, ctx.shape, ctx.scale)
        return g.to(ctx.dtype), None, None


class PixelLinear(torch.autograd.Function):
    """y = x @ W^T + b for x = the (pixels, C) view of a feature map (a 1x1 convolution), pixels >> C.

    The weight gradient  dW = dy^T @ x  reduces over 10^5 pixels into a 128 x 128 result: as ONE GEMM the library
    launches 16 workgroups (no split-K: 370 us on 134 400 pixels).  float32 with channel counts that are multiples of 128:
    the own kernel of csrc/wgrad.hip (float32 matrix cores over pixel slabs + a fixed-order sum, bias gradient included).
    Otherwise (fp16 under autocast, other widths) the pixels are cut into S slabs, the slabs are a BATCHED GEMM (S x 16
    workgroups) and the S partial results are summed."""

    @staticmethod
    @_fwd
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def _slabs(M):
        for S in (128, 100, 96, 81, 75, 64, 60, 50, 48, 40, 36, 32, 25, 24, 20, 16, 12, 10, 8, 6, 5, 4, 3, 2):
            if M % S == 0 and M // S >= 512:
                return S
        return 1

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        gy = gy.contiguous()
        if gy.dtype != weight.dtype:                 # autocast: the forward's GEMM ran in fp16
            weight = weight.to(gy.dtype)
        if x.dtype != gy.dtype:
            x = x.to(gy.dtype)
        if ctx.needs_input_grad[0]:
            gx = gy @ weight
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            M, S = x.shape[0], PixelLinear._slabs(x.shape[0])
            xc = x.contiguous()
            if OWN_WGRAD and M >= 4096 and ops.wgrad_supported(xc, gy):
                # float32: own kernel on the float32 matrix cores (csrc/wgrad.hip), 418 -> ~40 us on the image maps
                gw, gb = ops.wgrad(xc, gy, bias=want_b)
                want_b = False
            elif S > 1:
                gw = torch.bmm(gy.view(S, M // S, -1).transpose(1, 2), xc.view(S, M // S, -1)).sum(0)
            else:
                gw = gy.t() @ x
        if want_b:
            gb = gy.sum(0)
        return gx, gw, gb


class WideLinear(torch.autograd.Function):
    """y = x @ W^T + b for a FEW rows and VERY MANY outputs (DynamicConv's parameter generator: 200 tokens x 128 -> 32 768,
    reference decoder_utils.py:584-600).  The input gradient  dx = dy @ W  reduces over the 32 768 outputs into a 200 x 128
    result: as one GEMM the library launches a handful of workgroups without split-K (94 us per call for 8 MB of weights);
    here the reduction is cut into S slabs, the slabs are a batched GEMM and the partial results are summed."""

    @staticmethod
    @_fwd
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        gy = gy.contiguous()
        if gy.dtype != weight.dtype:
            weight = weight.to(gy.dtype)
        if x.dtype != gy.dtype:
            x = x.to(gy.dtype)
        N, K = gy.shape
        if ctx.needs_input_grad[0]:
            S = next((s_ for s_ in (64, 32, 16, 8) if K % s_ == 0 and K // s_ >= 256), 1)
            if S > 1:
                gx = torch.bmm(gy.view(N, S, K // S).transpose(0, 1), weight.view(S, K // S, -1)).sum(0)
            else:
                gx = gy @ weight
        if ctx.needs_input_grad[1]:
            gw = gy.t() @ x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        return gx, gw, gb


class I2PAttention(torch.autograd.Function):
    """ctx[cell] = sum_j d_j softmax_j(<qfold[cell], s_j>) s_j over the pillar's valid image keys
    (encoder_utils.py:257-320 with the single-head attention folded, see MMRI_I2P), d_j the attention dropout factor.
    Returns (ctx, valid, mass): mass = sum_j d_j p_j, the kept probability mass (= valid without dropout) - the folded
    value bias enters the module's output scaled by it, and its gradient flows back into the scores."""

    @staticmethod
    @_fwd
    def forward(ctx, img, qfold, pillars, coors, num_points, proj, aug_rev, ori_hw, dropout_p, seed, keys=None):
        ctx.save_for_backward(img, qfold, pillars, coors, num_points, proj, aug_rev)
        ctx.ori_hw, ctx.dropout_p, ctx.seed = ori_hw, float(dropout_p), int(seed)
        out, valid, mass = ops.i2p_attention(img, qfold, pillars, coors, num_points, proj, aug_rev, ori_hw, dropout_p,
                                             seed, keys, with_mass=True)
        ctx.mark_non_differentiable(valid)
        return out, valid, mass

    @staticmethod
    @_bwd
    def backward(ctx, grad_ctx, _grad_valid, grad_mass):
        img, qfold, pillars, coors, num_points, proj, aug_rev = ctx.saved_tensors
        g_img, g_q = ops.i2p_attention_bwd(img, qfold, grad_ctx, pillars, coors, num_points, proj, aug_rev,
                                           ctx.ori_hw, ctx.dropout_p, ctx.seed, grad_mass=grad_mass)
        return g_img.to(img.dtype), g_q.to(qfold.dtype), None, None, None, None, None, None, None, None, None


# ---------------------------------------------------------------------------------- DeepInteraction++ samplers
class MSDeformAttn(torch.autograd.Function):
    """mmcv MultiScaleDeformableAttention core on the packed [offsets | logits] projection (csrc/plusplus.hip /
    plusplus_bwd.hip).  Gradients: value, packed projection.  Reference points are constants."""

    @staticmethod
    @_fwd
    def forward(ctx, value, proj, ref, level_hw, n_points):
        L = len(level_hw)
        n_off = 8 * L * n_points * 2
        ctx.save_for_backward(value, proj, ref)
        ctx.level_hw, ctx.n_points, ctx.n_off = list(level_hw), n_points, n_off
        return ops.ms_deform_attn(value, proj[..., :n_off], proj[..., n_off:], ref, level_hw, n_points)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        value, proj, ref = ctx.saved_tensors
        n_off = ctx.n_off
        gv, gp = ops.ms_deform_attn_bwd(value, proj[..., :n_off], proj[..., n_off:], ref, ctx.level_hw, grad_out,
                                        ctx.n_points)
        return gv.to(value.dtype), gp, None, None, None


class GridGather(torch.autograd.Function):
    """out[g, n] = bilinear(feat[g // per_feat], grid[g, n]) + add[n]; gradient w.r.t. feat only (the grid is
    geometry, the additive term a constant encoding)."""

    @staticmethod
    @_fwd
    def forward(ctx, feat, grid, add, grids_per_feat):
        ctx.save_for_backward(grid)
        ctx.shape, ctx.dtype, ctx.per = tuple(feat.shape), feat.dtype, grids_per_feat
        return ops.grid_gather(feat, grid, add, grids_per_feat)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        (grid,) = ctx.saved_tensors
        return ops.grid_gather_bwd(grid, grad_out, ctx.shape, ctx.per).to(ctx.dtype), None, None, None


class PolarBEVSample(torch.autograd.Function):
    """out = mean over seeing cameras of bilinear(polar[cam], loc(cell, cam)) + bev  (fusion_transformerv4.py:581-640)."""

    @staticmethod
    @_fwd
    def forward(ctx, polar, bev, proj, aug_rev, cam_xy, params):
        ctx.save_for_backward(proj, aug_rev, cam_xy, params)
        ctx.shape, ctx.dtype = tuple(polar.shape), polar.dtype
        return ops.polar_bev_sample(polar, bev, proj, aug_rev, cam_xy, params)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        proj, aug_rev, cam_xy, params = ctx.saved_tensors
        gp = ops.polar_bev_sample_bwd(grad_out, proj, aug_rev, cam_xy, params, ctx.shape)
        return gp.to(ctx.dtype), grad_out, None, None, None, None
