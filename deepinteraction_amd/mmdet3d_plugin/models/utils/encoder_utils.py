"""MI355X-native cross-modal operators of the MMRI encoder.

Drop-in for the reference module of the same path
(`projects/mmdet3d_plugin/models/utils/encoder_utils.py`): same class names, constructor
arguments, `forward` signatures and `state_dict` keys -

    ConvBNReLU :11-34, similarFunction :36-57, weightingFunction :60-81,
    LocalContextAttentionBlock :84-135, BEVWarp :137-199, MMRI_P2I :202-213,
    MMRI_I2P :216-320

- but the execution is re-designed for CDNA4: channels-last feature maps end to end,
hand-written HIP kernels (through the C ABI in include/deepinteraction_hip.h) for the
window attention, the pillar attention and the BEV gather, depth completion on the device,
BatchNorm folded into the 1x1 projections at inference, no per-sample host syncs and no
GPU->CPU->GPU round trip.  Dense 1x1 projections are plain library GEMMs (hipBLASLt via
torch) on the (pixels, C) view of the channels-last maps.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from ....autograd import BEVWarpGather, I2PAttention, PixelLinear
from ....geometry import SampleGeometry
from ....utils import param_key

GEOM_KEY = '_di_geometry'      # per-forward cache placed in pts_metas by DeepInteractionEncoder


# measurement switch: 0 = the training BatchNorm + ReLU of ConvBNReLU through torch / MIOpen
FUSED_TRAINING_BN = os.environ.get('DI_TRAIN_FUSED_BN', '1') != '0'


class BatchNormReLU(torch.autograd.Function):
    """nn.BatchNorm2d in train() mode followed by nn.ReLU (reference ConvBNReLU, encoder_utils.py:11-34) on a channels-last
    map as one differentiable operator: batch statistics in float32, running statistics updated in place, the ReLU mask
    recomputed in the backward (csrc/batchnorm.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, relu, workspace):
        y, saved = ops.bn_train_fwd(x, weight, bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps,
                                    bn.momentum, relu, workspace)
        ctx.save_for_backward(x, saved)
        ctx.relu, ctx.ws, ctx.affine = bool(relu), workspace, weight is not None
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, saved = ctx.saved_tensors
        gx, gg, gb = ops.bn_train_bwd(x, grad_y, saved, ctx.relu, ctx.ws, ctx.affine)
        return gx, gg, gb, None, None, None


class ConvBNReLU(nn.Module):
    """Reference encoder_utils.py:11-34."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, groups=1,
                 norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU, bias='auto', inplace=True, affine=True):
        super().__init__()
        padding = dilation * (kernel_size - 1) // 2
        self.use_norm = norm_layer is not None
        self.use_activation = activation_layer is not None
        if bias == 'auto':
            bias = not self.use_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation=dilation,
                              groups=groups, bias=bias)
        if self.use_norm:
            self.bn = norm_layer(out_channels, affine=affine)
        if self.use_activation:
            self.activation = activation_layer(inplace=inplace)
        self._fold_cache = None
        self._bn_ws = None

    def forward(self, x):
        if not self.training and not torch.is_grad_enabled() and self.conv.kernel_size == (1, 1) and x.is_cuda:
            return pointwise(self, x)
        c = self.conv
        if (x.is_cuda and c.kernel_size == (1, 1) and c.stride == (1, 1) and c.groups == 1 and c.padding == (0, 0)):
            # training / fp32: a 1x1 convolution IS a GEMM on the (pixels, C) view of the channels-last map - and has to
            # be run as one: MIOpen's weight-gradient solver for this shape (fp32, NHWC, 134 400 pixels, 128 -> 128) is a
            # batched-GEMM kernel that takes 4.4 ms per call, 18 calls per training step (79 of 112 profiled ms);
            # through F.linear the same gradient is a batched (split over pixel slabs) hipBLASLt GEMM of tens of microseconds (autograd.PixelLinear)
            xl = ops.cl(x)
            n, C, H, W = xl.shape
            flat = xl.permute(0, 2, 3, 1).reshape(-1, C)
            if torch.is_grad_enabled() and c.weight.requires_grad:
                y = PixelLinear.apply(flat, c.weight.view(c.out_channels, C), c.bias)     # split-K weight gradient
            else:
                y = F.linear(flat, c.weight.view(c.out_channels, C), c.bias)
            x = y.view(n, H, W, c.out_channels).permute(0, 3, 1, 2)
        else:
            x = c(x)
        if self.use_norm:
            if self._fused_bn_ok(x):
                # training: batch statistics, normalisation and the ReLU in three launches (csrc/batchnorm.hip)
                if self._bn_ws is None or self._bn_ws.device != x.device:
                    self._bn_ws = ops.bn_workspace(self.bn.num_features, x.device)
                return BatchNormReLU.apply(x, self.bn.weight, self.bn.bias, self.bn, self.use_activation, self._bn_ws)
            x = self.bn(x)
        if self.use_activation:
            x = self.activation(x)
        return x

    def _fused_bn_ok(self, x):
        bn = self.bn
        return (FUSED_TRAINING_BN and bn.training and x.is_cuda and type(bn) is nn.BatchNorm2d and bn.track_running_stats
                and bn.momentum is not None and x.dim() == 4 and ops._is_cl(x) and x.dtype in (torch.float16, torch.float32)
                and bn.num_features % 8 == 0 and bn.num_features <= 256
                and (not self.use_activation or type(self.activation) is nn.ReLU)
                and (bn.weight is None or bn.weight.dtype == torch.float32))

    def folded(self, dtype):
        """Inference form `y = act(x W'^T + b')` with the BatchNorm folded in (fp32 math)."""
        w, b = self.conv.weight, self.conv.bias
        key = (dtype, param_key(self))
        if self._fold_cache is not None and self._fold_cache[0] == key:
            return self._fold_cache[1]
        with torch.no_grad():
            W = w.float().flatten(1)
            bias = torch.zeros(W.shape[0], device=W.device) if b is None else b.float()
            if self.use_norm:
                inv = torch.rsqrt(self.bn.running_var.float() + self.bn.eps)
                g = inv if self.bn.weight is None else inv * self.bn.weight.float()
                beta = 0.0 if self.bn.bias is None else self.bn.bias.float()
                W = W * g[:, None]
                bias = (bias - self.bn.running_mean.float()) * g + beta
            out = (W.to(dtype, copy=True).contiguous(), bias.to(dtype, copy=True).contiguous(),
                   bias.to(torch.float32, copy=True).contiguous())
        self._fold_cache = (key, out)
        return out


def conv3x3_module(owner, conv, x):
    """A 3x3 / pad 1 / 128-output `nn.Conv2d` of a neck: the HIP implicit-GEMM kernel for fp16 inference
    (csrc/conv3x3.hip; packed weights cached on `owner`, rebuilt when the parameters change), the library
    convolution otherwise (float32 parity path, training)."""
    x = ops.cl(x)
    if (x.is_cuda and x.dtype == torch.float16 and not torch.is_grad_enabled() and conv.in_channels % 32 == 0
            and conv.out_channels == 128 and conv.weight.dtype == torch.float16 and conv.kernel_size == (3, 3)
            and conv.padding == (1, 1) and conv.stride == (1, 1) and conv.groups == 1):
        cache = owner.__dict__.setdefault('_conv_cache', {})
        key = param_key(conv)
        hit = cache.get(id(conv))
        if hit is None or hit[0] != key:
            hit = (key, ops.pack_conv3x3(conv.weight, conv.bias))
            cache[id(conv)] = hit
        return ops.conv3x3(x, *hit[1])
    return conv(x)


def pointwise(m, x, x2=None):
    """Inference fast path of a 1x1 ConvBNReLU on channels-last maps: one GEMM on the
    (pixels, C) view with folded BN.  `x2` = second half of a channel concat
    (cat((x, x2), 1) is never materialised: split-K accumulate)."""
    x = ops.cl(x)
    n, C, H, W = x.shape
    Wf, bf, _ = m.folded(x.dtype)
    xf = x.permute(0, 2, 3, 1).reshape(-1, C)
    if x2 is None:
        y = F.linear(xf, Wf, bf)
    else:
        x2f = ops.cl(x2).permute(0, 2, 3, 1).reshape(-1, x2.shape[1])
        y = F.linear(xf, Wf[:, :C], bf)
        y = torch.addmm(y, x2f, Wf[:, C:].t())
    if m.use_activation:
        y = torch.relu_(y)
    return y.view(n, H, W, -1).permute(0, 3, 1, 2)


def _fusable(x, *mods):
    """fp16 HIP maps with 128 channels through 1x1 ConvBNReLU links of 128 outputs, inference form."""
    return (x.is_cuda and x.dtype == torch.float16 and x.shape[1] == 128 and not torch.is_grad_enabled()
            and all(isinstance(m, ConvBNReLU) and not m.training and m.conv.kernel_size == (1, 1)
                    and m.conv.out_channels == 128 and m.conv.groups == 1 for m in mods))


def project(seq, x):
    """A query/key/value projection (one ConvBNReLU or a Sequential of two) as ONE fused kernel when it
    is a 128-channel fp16 inference chain (ops.pointwise_chain), else link by link."""
    mods = list(seq) if isinstance(seq, nn.Sequential) else [seq]
    if len(mods) in (1, 2) and _fusable(x, *mods) and all(m.conv.in_channels == 128 for m in mods):
        w1, _, b1 = mods[0].folded(x.dtype)
        if len(mods) == 1:
            return ops.pointwise_chain(x, w1, b1, mods[0].use_activation)
        w2, _, b2 = mods[1].folded(x.dtype)
        return ops.pointwise_chain(x, w1, b1, mods[0].use_activation, w2=w2, b2=b2, relu2=mods[1].use_activation)
    return seq(x)


def _chain_consts(seq, dtype):
    """(LDS image, relu1, relu2, two links) of a one- or two-link projection (folded BatchNorm) for
    ops.pointwise_multi; the image lives ON the projection module and is rebuilt when one of its parameters changes."""
    mods = list(seq) if isinstance(seq, nn.Sequential) else [seq]
    key = (dtype,) + tuple(param_key(m) for m in mods)
    hit = seq.__dict__.get('_di_chain_image')
    if hit is None or hit[0] != key:
        w1, _, b1 = mods[0].folded(dtype)
        w2, b2 = (None, None)
        if len(mods) == 2:
            w2, _, b2 = mods[1].folded(dtype)
        hit = (key, ops.chain_image(w1, b1, w2, b2))
        seq.__dict__['_di_chain_image'] = hit
    return (hit[1], mods[0].use_activation, mods[-1].use_activation if len(mods) == 2 else False, len(mods) == 2)


def fusable_projections(x, *seqs):
    """All of `seqs` are one- or two-link 128-channel fp16 inference chains (the form ops.pointwise_multi takes)."""
    for seq in seqs:
        if isinstance(seq, tuple):          # an already packed chain (image, relu1, relu2, two links)
            continue
        mods = list(seq) if isinstance(seq, nn.Sequential) else [seq]
        if not (len(mods) in (1, 2) and _fusable(x, *mods) and all(m.conv.in_channels == 128 for m in mods)):
            return False
    return len(seqs) <= 4


def project_many(seqs, x, outs=None):
    """Several projections of the SAME map (e.g. query / key / value of a self-attention block): one launch that reads
    x once (ops.pointwise_multi) for 128-channel fp16 inference chains, else one after the other.  outs (fused form only):
    the maps to write into."""
    ok = 1 <= len(seqs) <= 4
    for seq in seqs:
        if isinstance(seq, tuple):
            continue
        mods = list(seq) if isinstance(seq, nn.Sequential) else [seq]
        ok = ok and len(mods) in (1, 2) and _fusable(x, *mods) and all(m.conv.in_channels == 128 for m in mods)
    if ok and len(seqs) > 1:
        return ops.pointwise_multi(x, [seq if isinstance(seq, tuple) else _chain_consts(seq, x.dtype) for seq in seqs], outs=outs)
    assert outs is None, 'caller-provided outputs need the fused path (check fusable_projections first)'
    assert not any(isinstance(seq, tuple) for seq in seqs), 'packed chains need the fused path'
    return [project(seq, x) for seq in seqs]


def mix2(proj1, a, b, proj2, c):
    """proj2(cat(proj1(cat(a, b)), c)) - the out_proj / integration pair of an encoder layer
    (deepinteraction_encoder.py:26-27, 31-32) - as ONE fused kernel when possible."""
    if _fusable(a, proj1, proj2) and proj1.conv.in_channels == 256 and proj2.conv.in_channels == 256 \
            and b.shape == a.shape and c.shape == a.shape and b.dtype == a.dtype and c.dtype == a.dtype:
        w1, _, b1 = proj1.folded(a.dtype)
        w2, _, b2 = proj2.folded(a.dtype)
        return ops.pointwise_chain(a, w1, b1, proj1.use_activation, x2=b, w2=w2, b2=b2,
                                   relu2=proj2.use_activation, x3=c)
    if not proj1.training and not torch.is_grad_enabled() and a.is_cuda:
        return pointwise(proj2, pointwise(proj1, a, b), c)
    return proj2(torch.cat((proj1(torch.cat((a, b), dim=1)), c), dim=1))


def mix2_folded(proj1, a, fold, mask, b, proj2, c, cache):
    """mix2 where the first input is `mask * (a @ w_f^T + b_f)` with fold = (w_f, b_f) - the output projection of the
    pillar attention on its valid cells (encoder_utils.py:314-319) - WITHOUT materialising it: w_f goes into the first
    half of proj1's weight and b_f becomes a bias on the marked pixels (ops.pointwise_chain mask / bm)."""
    # the fold tensors are derived (rebuilt whenever the attention's weights change: fresh tensors, version 0, perhaps at
    # a recycled address), so the entry is keyed on proj1 AND the identity of the fold tensors, which it keeps alive
    key = (param_key(proj1), tuple(id(t) for t in fold), tuple(t._version for t in fold))
    held = cache.get('fold')
    if cache.get('key') != key or held is None or any(x is not y for x, y in zip(held, fold)):
        w1, _, b1 = proj1.folded(torch.float32)
        w_f, b_f = fold[0].float(), fold[1].float()
        wa = w1[:, :128] @ w_f
        cache.update(key=key, fold=tuple(fold), w=torch.cat([wa, w1[:, 128:]], 1).to(a.dtype).contiguous(), b=b1.float().contiguous(),
                     bm=(w1[:, :128] @ b_f).contiguous())
    w2, _, b2 = proj2.folded(a.dtype)
    return ops.pointwise_chain(a, cache['w'], cache['b'], proj1.use_activation, x2=b, w2=w2, b2=b2,
                               relu2=proj2.use_activation, x3=c, mask=mask, bm=cache['bm'])


from torch.amp import custom_bwd as _cbwd, custom_fwd as _cfwd

# measurement switch: 0 = mixed-precision training keeps the unfused float32 window operators (round 4's first AMP figure)
FUSED_TRAINING_ATTENTION = os.environ.get('DI_TRAIN_FUSED_LA', '1') != '0'


class similarFunction(torch.autograd.Function):
    """Reference encoder_utils.py:36-57, bound to the HIP window kernels."""

    @staticmethod
    @_cfwd(device_type='cuda')
    def forward(ctx, x_ori, x_loc, kH, kW):
        ctx.save_for_backward(x_ori, x_loc)
        ctx.kHW = (kH, kW)
        return ops.similar_forward(x_ori, x_loc, kH, kW)

    @staticmethod
    @_cbwd(device_type='cuda')
    def backward(ctx, grad_outputs):
        x_ori, x_loc = ctx.saved_tensors
        kH, kW = ctx.kHW
        grad_ori = ops.similar_backward(x_loc, grad_outputs, kH, kW, True)
        grad_loc = ops.similar_backward(x_ori, grad_outputs, kH, kW, False)
        return grad_ori, grad_loc, None, None


class weightingFunction(torch.autograd.Function):
    """Reference encoder_utils.py:60-81, bound to the HIP window kernels."""

    @staticmethod
    @_cfwd(device_type='cuda')
    def forward(ctx, x_ori, x_weight, kH, kW):
        ctx.save_for_backward(x_ori, x_weight)
        ctx.kHW = (kH, kW)
        return ops.weighting_forward(x_ori, x_weight, kH, kW)

    @staticmethod
    @_cbwd(device_type='cuda')
    def backward(ctx, grad_outputs):
        x_ori, x_weight = ctx.saved_tensors
        kH, kW = ctx.kHW
        grad_ori = ops.weighting_backward_ori(x_weight, grad_outputs, kH, kW)
        grad_weight = ops.weighting_backward_weight(x_ori, grad_outputs, kH, kW)
        return grad_ori, grad_weight, None, None


class fusedWindowAttention(torch.autograd.Function):
    """similarFunction -> softmax(. * scale) -> weightingFunction (reference encoder_utils.py:132-134) as ONE differentiable
    operator for mixed-precision training: fp16 maps, forward keeps the log-sum-exp of every query, backward recomputes the
    probabilities on the matrix cores (csrc/local_attn_train.hip) - no (n, H, W, 81) tensor in either direction.  Under
    `torch.autocast` the inputs are cast to fp16; in float32 training the unfused float32 operators above stay in use."""

    @staticmethod
    @_cfwd(device_type='cuda', cast_inputs=torch.float16)
    def forward(ctx, query, key, value, scale):
        out, lse = ops.local_attention_train_fwd(query, key, value, scale)
        ctx.save_for_backward(query, key, value, out, lse)
        ctx.scale = float(scale)
        return out

    @staticmethod
    @_cbwd(device_type='cuda')
    def backward(ctx, grad_out):
        query, key, value, out, lse = ctx.saved_tensors
        gq, gk, gv = ops.local_attention_train_bwd(query, key, value, out, grad_out, lse, ctx.scale)
        return gq, gk, gv, None


class LocalContextAttentionBlock(nn.Module):
    """Reference encoder_utils.py:84-135."""

    def __init__(self, in_channels, out_channels, kernel_size, last_affine=True):
        super().__init__()
        self.f_similar = similarFunction.apply
        self.f_weighting = weightingFunction.apply
        self.kernel_size = kernel_size
        self.query_project = nn.Sequential(ConvBNReLU(in_channels, out_channels, kernel_size=1),
                                           ConvBNReLU(out_channels, out_channels, kernel_size=1))
        self.key_project = nn.Sequential(ConvBNReLU(in_channels, out_channels, kernel_size=1),
                                         ConvBNReLU(out_channels, out_channels, kernel_size=1))
        self.value_project = ConvBNReLU(in_channels, out_channels, kernel_size=1, affine=last_affine)
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, target_feats, source_feats, query=None, **kwargs):
        """`query`: the already projected query map, when the caller computed it together with other projections of
        `target_feats` (DeepInteractionEncoderLayer shares one read of the image map between both image-side blocks)."""
        if query is None and target_feats is source_feats:
            query, key, value = project_many([self.query_project, self.key_project, self.value_project], source_feats)
        else:
            if query is None:
                query = project(self.query_project, target_feats)
            key, value = project_many([self.key_project, self.value_project], source_feats)
        ks = self.kernel_size
        scale = 1.0 / math.sqrt(key.size(1))
        if not (torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad)):
            # inference: one fused kernel, the (n,H,W,81) weights stay in registers
            return ops.local_attention(query, key, value, ks, ks, scale)
        half = torch.is_autocast_enabled() or query.dtype == torch.float16
        if half and query.is_cuda and query.size(1) == 128 and ks == 9 and FUSED_TRAINING_ATTENTION:
            return fusedWindowAttention.apply(query, key, value, scale)
        weight = self.f_similar(query, key, ks, ks)
        weight = F.softmax(weight * scale, -1)
        return self.f_weighting(value, weight, ks, ks)


def sample_geometry(img_metas, pts_metas, b, img_hw, device):
    cache = pts_metas.get(GEOM_KEY) if isinstance(pts_metas, dict) else None
    if cache is not None and cache[b] is not None and cache[b].img_hw == tuple(img_hw):
        return cache[b]
    g = SampleGeometry(img_metas[b], tuple(img_hw), device)
    if cache is not None:
        cache[b] = g
    return g


class BEVWarp(nn.Module):
    """Reference encoder_utils.py:137-199: BEV features gathered onto every image pixel
    through a completed LiDAR depth map.  Depth scatter + completion depend only on the
    points and metas; they are computed once per sample and cached in the geometry object
    (the reference recomputes them in every layer, on the CPU)."""

    def __init__(self):
        super().__init__()

    @staticmethod
    def dense_depth(geom, pts, I_H, I_W):
        """Sparse depth scatter + ip_basic completion on the device (reference :155-182), cached per sample."""
        if getattr(geom, 'dense_depth', None) is None:
            p = pts if pts.dtype == torch.float32 else pts.float()
            geom.sparse_depth = ops.depth_scatter(p, geom.lidar2img, geom.aug_rev, I_H, I_W, geom.ori_hw)
            geom.dense_depth = ops.depth_complete(geom.sparse_depth)
        return geom.dense_depth

    def forward(self, lidar_feats, img_feats, img_metas, pts_metas, **kwargs):
        B, V, C, I_H, I_W = img_feats.shape
        out = []
        for b in range(B):
            geom = sample_geometry(img_metas, pts_metas, b, (I_H, I_W), lidar_feats.device)
            depth = self.dense_depth(geom, pts_metas['pts'][b], I_H, I_W)
            if torch.is_grad_enabled() and lidar_feats.requires_grad:       # training: d(warped)/d(bev)
                out.append(BEVWarpGather.apply(lidar_feats[b:b + 1], depth, geom.img2lidar, geom.aug_fwd,
                                               geom.xs, geom.ys, geom.pc_range))
            else:
                out.append(ops.bevwarp_gather(lidar_feats[b:b + 1], depth, geom.img2lidar, geom.aug_fwd,
                                              geom.xs, geom.ys, geom.pc_range))
        return out[0].unsqueeze(0) if B == 1 else torch.stack(out, 0)


class MMRI_P2I(nn.Module):
    """Reference encoder_utils.py:202-213."""

    def __init__(self, in_channels, out_channels, kernel_size, last_affine=True):
        super().__init__()
        self.Warp = BEVWarp()
        self.Local = LocalContextAttentionBlock(in_channels, out_channels, kernel_size, last_affine=True)

    def warp_kv_fusable(self, lidar_feats):
        """fp16 inference on 128-channel maps: the key / value projections can gather the warped map themselves."""
        L = self.Local
        return (lidar_feats.is_cuda and lidar_feats.dtype == torch.float16 and lidar_feats.shape[1] == 128
                and not torch.is_grad_enabled() and fusable_projections(lidar_feats, L.key_project, L.value_project))

    def warp_kv(self, lidar_feats, img_feats, img_metas, pts_metas, out=None):
        """key / value maps (B*N,C,H,W) of the local attention over the warped BEV map, WITHOUT the warped map in memory:
        `ops.warp_project` gathers the BEV samples inside the projection launch (one launch per sample).  out: (k, v) maps
        of that shape to write into (the second halves of the encoder layer's pair buffers)."""
        B, V, C, I_H, I_W = img_feats.shape
        L = self.Local
        chains = [_chain_consts(L.key_project, lidar_feats.dtype), _chain_consts(L.value_project, lidar_feats.dtype)]
        ks, vs = [], []
        for b in range(B):
            geom = sample_geometry(img_metas, pts_metas, b, (I_H, I_W), lidar_feats.device)
            depth = self.Warp.dense_depth(geom, pts_metas['pts'][b], I_H, I_W)
            outs = None if out is None else [out[0][b * V:(b + 1) * V], out[1][b * V:(b + 1) * V]]
            k, v = ops.warp_project(lidar_feats[b:b + 1], depth, geom.img2lidar, geom.aug_fwd, geom.xs, geom.ys,
                                    geom.pc_range, chains, outs=outs)
            ks.append(k)
            vs.append(v)
        if out is not None:
            return out
        return (ks[0], vs[0]) if B == 1 else (torch.cat(ks, 0), torch.cat(vs, 0))

    def forward(self, lidar_feats, img_feats, img_metas, pts_metas, query=None, warped=None, kv=None, **kwargs):
        """`query` / `warped` / `kv`: the projected query map / the warped BEV map / the key and value maps of the warped
        map (`warp_kv`) when the caller already has them."""
        B, N, C, H, W = img_feats.shape
        if kv is None and warped is None and query is not None and self.warp_kv_fusable(lidar_feats):
            kv = self.warp_kv(lidar_feats, img_feats, img_metas, pts_metas)
        if kv is not None:
            ks = self.Local.kernel_size
            out = ops.local_attention(query, kv[0], kv[1], ks, ks, 1.0 / math.sqrt(kv[0].size(1)))
            return out.view(B, N, -1, H, W)
        if warped is None:
            warped = self.Warp(lidar_feats, img_feats, img_metas, pts_metas)    # B, N, C, H, W
        B, N, C, H, W = warped.shape
        out = self.Local(img_feats.reshape(B * N, C, H, W), warped.reshape(B * N, C, H, W), query=query)
        return out.view(B, N, C, H, W)


class MMRI_I2P(nn.Module):
    """Reference encoder_utils.py:216-320: every non-empty pillar attends (single head) to the
    image features under its <= 20 points x 6 cameras.

    The reference materialises a (6,C,P*20) `grid_sample`, compacts it into 4 padding buckets
    (`group_attn`, :226-255) and runs nn.MultiheadAttention per bucket.  Here one wavefront
    per pillar projects, gathers and soft-maxes its own keys; the MHA projections are folded
    around the kernel (exact in eval mode):
        score_j = <Wk^T (Wq x + bq), s_j> / sqrt(C)      (the <q, bk> term is softmax-invariant)
        out     = (Wo Wv) sum_j a_j s_j + (Wo bv + bo)
    """

    def __init__(self, pts_channels, img_channels, dropout):
        super().__init__()
        self.pts_channels = pts_channels
        self.img_channels = img_channels
        self.dropout = dropout
        self.learnedAlign = nn.MultiheadAttention(pts_channels, 1, dropout=dropout, kdim=img_channels,
                                                  vdim=img_channels, batch_first=True)
        self._fold_cache = None

    def folded(self, dtype):
        la = self.learnedAlign
        if la._qkv_same_embed_dim:
            wq, wk, wv = la.in_proj_weight.chunk(3, 0)
        else:
            wq, wk, wv = la.q_proj_weight, la.k_proj_weight, la.v_proj_weight
        key = param_key(la)
        if self._fold_cache is None or self._fold_cache[0] != key:
            self._fold_cache = (key, {})                       # per dtype
        hit = self._fold_cache[1].get(dtype)
        if hit is not None:
            return hit
        with torch.no_grad():
            E = self.pts_channels
            bq, _, bv = la.in_proj_bias.float().chunk(3, 0)
            wq, wk, wv = wq.float(), wk.float(), wv.float()
            s = 1.0 / math.sqrt(E)
            w_qk = (wk.t() @ wq) * s                       # (img_channels, pts_channels)
            b_qk = (wk.t() @ bq) * s
            w_ov = la.out_proj.weight.float() @ wv          # (pts_channels, img_channels)
            b_ov = la.out_proj.weight.float() @ bv + la.out_proj.bias.float()
            out = tuple(t.to(dtype).contiguous() for t in (w_qk, b_qk, w_ov, b_ov))
        self._fold_cache[1][dtype] = out
        return out

    def folded_live(self, dtype):
        """The same folding as `folded`, inside autograd (training: gradients reach the MHA parameters).
        The dropped <q, bk> term shifts all scores of a pillar equally: its gradient is exactly zero in
        the reference too."""
        la = self.learnedAlign
        if la._qkv_same_embed_dim:
            wq, wk, wv = la.in_proj_weight.chunk(3, 0)
        else:
            wq, wk, wv = la.q_proj_weight, la.k_proj_weight, la.v_proj_weight
        bq, _, bv = la.in_proj_bias.chunk(3, 0)
        s = 1.0 / math.sqrt(self.pts_channels)
        w_qk = (wk.t() @ wq) * s
        b_qk = (wk.t() @ bq) * s
        w_ov = la.out_proj.weight @ wv
        b_ov = la.out_proj.weight @ bv + la.out_proj.bias
        return tuple(t.to(dtype) for t in (w_qk, b_qk, w_ov, b_ov))

    @staticmethod
    def pillar_keys(geom, pts_metas, s, e, img_hw, bev_hw):
        """The sample's key table (projection of every (point, camera) slot, masks, compaction): geometry only, so it
        is built once per forward and shared by all layers (cached in the per-sample geometry object)."""
        if getattr(geom, 'pillar_keys', None) is None:
            geom.pillar_keys = ops.i2p_key_table(pts_metas['pillars'][s:e], pts_metas['pillar_coors'][s:e],
                                                 pts_metas['pillars_num_points'][s:e], geom.lidar2img, geom.aug_rev,
                                                 geom.ori_hw, img_hw, bev_hw)
        return geom.pillar_keys

    def query_chain(self, dtype):
        """The folded query projection `qfold = x Wqk^T + bqk` as a packed chain for ops.pointwise_multi (one launch with
        the other projections of the BEV map)."""
        w_qk, b_qk, _, _ = self.folded(torch.float32)
        hit = self.__dict__.get('_qchain')
        if hit is None or hit[0] is not w_qk:
            hit = (w_qk, (ops.chain_image(w_qk, b_qk), False, False, False))
            self.__dict__['_qchain'] = hit
        return hit[1]

    def attend(self, qfold, img_feat, img_metas, pts_metas):
        """Inference: the attention pass alone on an already projected query map.  Returns ctx (B,Ci,Hb,Wb) - BEFORE the
        folded output projection - and valid (B,1,Hb,Wb); the caller applies `valid * (ctx Wov^T + bov)`."""
        B, Ci, Hb, Wb = qfold.shape
        _, V, _, Hi, Wi = img_feat.shape
        bounds = pillar_batch_bounds(pts_metas, B)
        ctxs, valids = [], []
        for b in range(B):
            s, e = bounds[b], bounds[b + 1]
            geom = sample_geometry(img_metas, pts_metas, b, (Hi, Wi), qfold.device)
            ctx, valid = ops.i2p_attention(img_feat[b], qfold[b:b + 1], pts_metas['pillars'][s:e],
                                           pts_metas['pillar_coors'][s:e], pts_metas['pillars_num_points'][s:e],
                                           geom.lidar2img, geom.aug_rev, geom.ori_hw,
                                           keys=self.pillar_keys(geom, pts_metas, s, e, (Hi, Wi), (Hb, Wb)))
            ctxs.append(ctx)
            valids.append(valid)
        return (ctxs[0], valids[0]) if B == 1 else (torch.cat(ctxs, 0), torch.cat(valids, 0))

    def forward(self, lidar_feat, img_feat, img_metas, pts_metas, **kwargs):
        B = len(img_metas)
        lidar_feat = ops.cl(lidar_feat)
        _, C, Hb, Wb = lidar_feat.shape
        _, V, Ci, Hi, Wi = img_feat.shape
        live = torch.is_grad_enabled()
        w_qk, b_qk, w_ov, b_ov = self.folded_live(lidar_feat.dtype) if live else self.folded(lidar_feat.dtype)
        # attention dropout (nn.MultiheadAttention(dropout) acts on the probabilities, training only):
        # decided per (pillar, key slot) by a counter-based hash of a fresh seed, regenerated in backward
        drop = float(self.dropout) if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if drop > 0 else 0
        flat = lidar_feat.permute(0, 2, 3, 1).reshape(-1, C)
        qfold = F.linear(flat, w_qk, b_qk).view(B, Hb, Wb, Ci).permute(0, 3, 1, 2)   # channels-last
        bounds = pillar_batch_bounds(pts_metas, B)
        outs = []
        for b in range(B):
            s, e = bounds[b], bounds[b + 1]
            geom = sample_geometry(img_metas, pts_metas, b, (Hi, Wi), lidar_feat.device)
            args = (img_feat[b], qfold[b:b + 1], pts_metas['pillars'][s:e], pts_metas['pillar_coors'][s:e],
                    pts_metas['pillars_num_points'][s:e], geom.lidar2img, geom.aug_rev, geom.ori_hw, drop,
                    (seed + b * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF,   # per-sample stream: no mask reuse across b
                    self.pillar_keys(geom, pts_metas, s, e, (Hi, Wi), (Hb, Wb)))
            if live:
                # every key's value carries the bias bv, so under attention dropout Wo.bv is weighted by the kept
                # probability mass of the cell (1 without dropout), as nn.MultiheadAttention does; bo by `valid`
                ctx, valid, mass = I2PAttention.apply(*args)
                la = self.learnedAlign
                b_v = la.out_proj.weight @ la.in_proj_bias.chunk(3, 0)[2]
                o = F.linear(ctx.permute(0, 2, 3, 1).reshape(-1, Ci), w_ov)
                o = o + mass.reshape(-1, 1) * b_v.to(o.dtype) + valid.reshape(-1, 1) * la.out_proj.bias.to(o.dtype)
            else:
                ctx, valid = ops.i2p_attention(*args)
                o = F.linear(ctx.permute(0, 2, 3, 1).reshape(-1, Ci), w_ov, b_ov)
                o = o * valid.reshape(-1, 1)                                     # empty pillars / cells stay 0
            outs.append(o.view(1, Hb, Wb, C).permute(0, 3, 1, 2))
        return outs[0] if B == 1 else torch.cat(outs, 0)


def pillar_batch_bounds(pts_metas, B):
    """Start offsets of each sample's pillars (they are batch-sorted, reference :265-273).
    The reference counts with B device->host syncs per layer; callers may pass the split
    they already know as `pts_metas['pillar_batch_bounds']`, otherwise it is computed once
    per forward and cached."""
    if 'pillar_batch_bounds' in pts_metas:
        return pts_metas['pillar_batch_bounds']
    coors = pts_metas['pillar_coors']
    if B == 1:
        bounds = [0, coors.shape[0]]
    else:
        cnt = torch.bincount(coors[:, 0].long(), minlength=B).cpu().tolist()
        bounds = [0]
        for c in cnt:
            bounds.append(bounds[-1] + c)
    if GEOM_KEY in pts_metas:
        pts_metas['pillar_batch_bounds'] = bounds
    return bounds
