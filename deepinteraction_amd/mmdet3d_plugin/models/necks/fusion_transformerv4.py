"""DeepInteraction++ neck (drop-in for reference `models/necks/fusion_transformerv4.py`): `FusionTransformerv4`
:26-139, `DeepInteractionLayer` :143-218, `MMRI_P2I` :220-240, `MMRI_I2P` :242-364, `MMRI_I2P_Polar` :470-640 - same
registry names, constructor arguments, forward signatures and `state_dict` keys.

MI355X execution: feature maps stay channels-last, so the (bs, H*W, C) token view the transformer layers work on is
free; deformable attention, the polar samplers and the short-sequence attention are HIP kernels
(csrc/plusplus.hip), the BEV->image warp and the pillar attention are the v1 kernels, dense projections / LayerNorm
are library calls.  The per-camera Python loop of the polar attention (:536) is one batched pass over all cameras.
mmcv / flash-attn are not dependencies: their bricks are in models/utils/transformer_bricks.py.
"""
import copy
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .... import ops, utils
from ....autograd import GridGather, PolarBEVSample
from ....geometry import PC_RANGE, aug_affine
from ....registry import ATTENTION, NECKS, TRANSFORMER_LAYER
from ..utils import encoder_utils as eu
from ..utils.transformer_bricks import (MultiScaleDeformableAttention, TransFFN, fused_tokens_ok, linear128, module_cache,
                                        post_norm)

if 'MultiScaleDeformableAttention' not in getattr(ATTENTION, 'module_dict', {}):
    ATTENTION.register_module(module=MultiScaleDeformableAttention)


def _tokens(x):
    """(n,C,H,W) -> (n, H*W, C): a view of the channels-last storage."""
    x = ops.cl(x)
    n, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(n, H * W, C)


def _map(t, H, W):
    """(n, H*W, C) contiguous tokens -> (n,C,H,W) channels-last view."""
    n, _, C = t.shape
    return t.reshape(n, H, W, C).permute(0, 3, 1, 2)


@ATTENTION.register_module()
class MMRI_P2I(nn.Module):
    """Reference :220-240: image tokens attend (1-level deformable attention) to the BEV map warped into their view."""

    def __init__(self, embed_dims, batch_first=True):
        super().__init__()
        self.embed_dims = embed_dims
        self.Warp = eu.BEVWarp()
        self.Local = MultiScaleDeformableAttention(embed_dims, num_levels=1, batch_first=batch_first)

    def forward(self, img_feats, lidar_feats, img_metas, pts_metas, reference_points=None, then_norm=None, **kwargs):
        B = lidar_feats.size(0)
        _, C, H, W = img_feats.shape
        q = _tokens(img_feats)
        lidar_feats = ops.cl(lidar_feats)
        if C == 128 and self.Local.can_fuse_tokens(q) and lidar_feats.dtype == torch.float16:      # the SAME predicate Local.forward tests
            # the value projection GATHERS its input: the warped map (34 MB per sample) is neither written nor read
            # (ops.warp_project, the launch the v1 P2I block uses for its key / value projections)
            vp = self.Local.value_proj
            chain = module_cache(self.Local, '_value_chain', [vp],
                                 lambda: (ops.chain_image(vp.weight.float(), vp.bias.float()), False, False, False))
            vs = []
            for b in range(B):
                geom = eu.sample_geometry(img_metas, pts_metas, b, (H, W), lidar_feats.device)
                depth = self.Warp.dense_depth(geom, pts_metas['pts'][b], H, W)
                vs.append(ops.warp_project(lidar_feats[b:b + 1], depth, geom.img2lidar, geom.aug_fwd, geom.xs, geom.ys,
                                           geom.pc_range, [chain], head_major=True)[0])
            pv = vs[0] if B == 1 else torch.cat(vs, 0)                               # (B*V, 8, H*W, 16) head-major
            out = self.Local(query=q, value=None, projected_value=pv, reference_points=reference_points,
                             spatial_shapes=[(H, W)], level_start_index=None, then_norm=then_norm)
            return _map(out, H, W)
        warped = self.Warp(lidar_feats, img_feats.reshape(B, -1, C, H, W), img_metas, pts_metas)
        v = _tokens(warped.reshape(-1, C, H, W))
        out = self.Local(query=q, value=v, reference_points=reference_points, spatial_shapes=[(H, W)],
                         level_start_index=None, then_norm=then_norm)
        return _map(out, H, W)


@ATTENTION.register_module()
class MMRI_I2P(eu.MMRI_I2P):
    """Reference :242-364: the v1 pillar attention kernel (bucketed `group_attn` == plain masked attention) plus the
    residual of :364.  `fp16_enabled` / `group_attn_enabled` only choose how the reference evaluates the same
    single-head attention; the flash-attention (8-head) variant is not built."""

    def __init__(self, embed_dims, dropout, batch_first=True, fp16_enabled=False, flash_attn=False,
                 group_attn_enabled=False):
        if flash_attn:
            raise NotImplementedError('MMRI_I2P(flash_attn=True) (8 heads) is not used by the reference configs')
        super().__init__(embed_dims, embed_dims, dropout)
        self.embed_dims = embed_dims
        self.group_attn_enabled = group_attn_enabled

    def forward(self, lidar_feat, img_feat, img_metas, pts_metas, **kwargs):
        B = lidar_feat.size(0)
        _, C, H, W = img_feat.shape
        img5 = ops.cl(img_feat).reshape(B, -1, C, H, W)
        lidar_feat = ops.cl(lidar_feat)
        if (C == 128 and lidar_feat.shape[1] == 128 and lidar_feat.is_cuda and lidar_feat.dtype == torch.float16
                and not torch.is_grad_enabled() and not self.training):
            # inference: the folded query / output projections through the fused chain kernels (no library GEMM); the output
            # projection applies its bias on the cells that have keys (mask) - the attention's rows of the others are 0
            def build():
                w_qk, b_qk, w_ov, b_ov = self.folded(torch.float32)
                return (w_qk.to(torch.float16).contiguous(), b_qk.contiguous(), w_ov.to(torch.float16).contiguous(),
                        torch.zeros_like(b_ov), b_ov.contiguous())
            w_qk, b_qk, w_ov, zero, b_ov = module_cache(self, '_pp_fold', [self.learnedAlign], build)
            _, _, Hb, Wb = lidar_feat.shape
            flat = lidar_feat.permute(0, 2, 3, 1).reshape(-1, 128)
            qfold = ops.token_linear(flat, w_qk, b_qk).view(B, Hb, Wb, 128).permute(0, 3, 1, 2)
            ctx, valid = self.attend(qfold, img5, img_metas, pts_metas)
            out = ops.pointwise_chain(ctx, w_ov, zero, False, mask=valid.to(torch.float16).contiguous(), bm=b_ov)
            return out + lidar_feat
        return super().forward(lidar_feat, img5, img_metas, pts_metas) + lidar_feat


def sine_position_tokens(n_x, n_y, num_feats, temperature=10000, x_major=False):
    """`TransSinePositionalEncoding` (:367-467), un-normalised 2-D branch, for an (n_y, n_x) grid as float32 tokens
    (n_y*n_x, 2*num_feats) = cat(pos_y, pos_x); x_major orders the tokens (x, y) instead of (y, x)."""
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode='floor') / num_feats)

    def enc(r):
        p = r[:, None] / dim_t
        return torch.stack((p[:, 0::2].sin(), p[:, 1::2].cos()), dim=2).view(r.shape[0], -1)
    px, py = enc(torch.arange(0., float(n_x), 1.)), enc(torch.arange(0., float(n_y), 1.))
    pos = torch.cat((py[:, None, :].expand(n_y, n_x, -1), px[None, :, :].expand(n_y, n_x, -1)), -1)
    if x_major:
        pos = pos.transpose(0, 1)
    return pos.reshape(n_y * n_x, -1).contiguous()


class PackedMHA(nn.Module):
    """Parameters of the reference's `FlashMultiheadAttention` (:715-760): `in_proj_weight`, `in_proj_bias`,
    `out_proj.*`.  The attention itself is `ops.mha_small`."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)

    def _attend(self, q, k, v):
        if not (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)):
            return ops.mha_small(q, k, v, self.num_heads)
        # training: scores materialised per (sequence, head) through library GEMMs and their autograd (what
        # flash-attn's backward recomputes); attention dropout as in FlashAttention(attention_dropout=dropout)
        N, T, E = q.shape
        h, d = self.num_heads, E // self.num_heads
        qh = q.reshape(N, T, h, d).transpose(1, 2) * (float(d) ** -0.5)
        kh = k.reshape(N, -1, h, d).transpose(1, 2)
        vh = v.reshape(N, -1, h, d).transpose(1, 2)
        a = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)).float(), -1).to(vh.dtype)
        if self.training and self.dropout > 0:
            a = F.dropout(a, self.dropout)
        return torch.matmul(a, vh).transpose(1, 2).reshape(N, T, E)

    def self_attention(self, x):
        E = self.embed_dim
        qkv = F.linear(x, self.in_proj_weight, self.in_proj_bias)               # (N,T,3E): q | k | v in place
        return self.out_proj(self._attend(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]))

    def cross_attention(self, x, memory):
        E = self.embed_dim
        q = F.linear(x, self.in_proj_weight[:E], self.in_proj_bias[:E])
        kv = F.linear(memory, self.in_proj_weight[E:], self.in_proj_bias[E:])   # (N,S,2E): k | v
        return self.out_proj(self._attend(q, kv[..., :E], kv[..., E:]))

    # ---- fp16 inference: every projection through the fused chain kernels, the output projection with the residual and the
    # layer's LayerNorm as its epilogue (ops.linear_ln)
    def _images(self):
        def build():
            E = self.embed_dim
            w, b = self.in_proj_weight.float(), self.in_proj_bias.float()
            return ([ops.chain_image(w[k * E:(k + 1) * E], b[k * E:(k + 1) * E]) for k in range(3)],
                    ops.chain_image(self.out_proj.weight.float(), self.out_proj.bias.float()))
        return module_cache(self, '_di_images', [self], build)

    def fused_self(self, x, norm):
        """LayerNorm(x + out_proj(attention(q, k, v of x))) for (N, T, 128) fp16 tokens."""
        N, T, E = x.shape
        x2 = x.reshape(-1, E).contiguous()
        qkv, out_image = self._images()
        q, k, v = (t.view(N, T, E) for t in ops.token_linear_multi(x2, qkv))
        a = ops.mha_small(q, k, v, self.num_heads)
        return ops.linear_ln(a.view(-1, E), out_image, x2, norm.weight, norm.bias, norm.eps).view(N, T, E)

    def fused_cross(self, x, memory, norm):
        N, T, E = x.shape
        S = memory.shape[1]
        x2 = x.reshape(-1, E).contiguous()
        qkv, out_image = self._images()
        q = ops.token_linear_multi(x2, qkv[:1])[0].view(N, T, E)
        k, v = (t.view(N, S, E) for t in ops.token_linear_multi(memory.reshape(-1, E).contiguous(), qkv[1:]))
        a = ops.mha_small(q, k, v, self.num_heads)
        return ops.linear_ln(a.view(-1, E), out_image, x2, norm.weight, norm.bias, norm.eps).view(N, T, E)


class _RayDecoderLayer(nn.Module):
    """`FlashTransformerDecoderLayer` (:763-769) = torch 1.9 post-norm nn.TransformerDecoderLayer (ReLU) around two
    packed attentions; same parameter names."""

    def __init__(self, d, heads, ff, dropout=0.1):
        super().__init__()
        self.self_attn, self.multihead_attn = PackedMHA(d, heads, dropout), PackedMHA(d, heads, dropout)
        self.linear1, self.linear2 = nn.Linear(d, ff), nn.Linear(ff, d)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d), nn.LayerNorm(d), nn.LayerNorm(d)
        self.drop = nn.Dropout(dropout)                        # torch's dropout / dropout1..3 (p = 0.1), training only

    def forward(self, tgt, memory):
        if (tgt.shape[-1] == 128 and self.self_attn.num_heads == 8 and fused_tokens_ok(tgt, self) and memory.dtype == tgt.dtype
                and self.norm1.weight.dtype == torch.float16 and self.linear1.out_features % 128 == 0
                and self.linear1.out_features <= 1024):
            tgt = self.self_attn.fused_self(tgt, self.norm1)
            tgt = self.multihead_attn.fused_cross(tgt, memory, self.norm2)
            images = module_cache(self, '_ffn_images', [self.linear1, self.linear2],
                                  lambda: ops.ffn_images(self.linear1.weight, self.linear1.bias, self.linear2.weight,
                                                         self.linear2.bias))
            shape = tgt.shape
            return ops.ffn_ln(tgt.reshape(-1, 128), images, self.norm3.weight, self.norm3.bias, self.norm3.eps).view(shape)
        dr = self.drop
        tgt = post_norm(self.norm1, tgt, dr(self.self_attn.self_attention(tgt)))
        tgt = post_norm(self.norm2, tgt, dr(self.multihead_attn.cross_attention(tgt, memory)))
        return post_norm(self.norm3, tgt, dr(self.linear2(dr(torch.relu(self.linear1(tgt))))))


class _Stack(nn.Module):
    def __init__(self, layers, d):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.norm = nn.LayerNorm(d)


class _RayTransformer(nn.Module):
    """`nn.Transformer(num_encoder_layers=0, custom_decoder=...)` (:486-492): zero encoder layers followed by the
    encoder's LayerNorm; keys `encoder.norm.*`, `decoder.layers.i.*`, `decoder.norm.*`."""

    def __init__(self, d, heads, ff, num_decoder_layers):
        super().__init__()
        self.encoder = _Stack([], d)
        self.decoder = _Stack([_RayDecoderLayer(d, heads, ff) for _ in range(num_decoder_layers)], d)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, src, tgt):
        memory = post_norm(self.encoder.norm, src)
        for layer in self.decoder.layers:
            tgt = layer(tgt, memory)
        return post_norm(self.decoder.norm, tgt)


@ATTENTION.register_module()
class MMRI_I2P_Polar(nn.Module):
    """Reference :470-640 - polar ray attention.  For every camera and image column w: R = 60 ray queries (BEV
    features sampled along the column's viewing ray + sine encoding) attend to the column's H image tokens through
    one transformer decoder layer; every BEV cell then reads the polar maps of the cameras that see it.

    All V cameras (and samples) run as ONE batch of B*V*W sequences; the two geometric samplers are HIP kernels; the
    ray geometry depends only on the metas and is prepared on the host in float32 exactly as the reference does."""

    def __init__(self, embed_dims, dropout, batch_first=True, radius_range=[1., 61., 1.0], num_decoder_layers=1,
                 pc_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]):
        super().__init__()
        self.embed_dims, self.dropout = embed_dims, dropout
        self.radius_range, self.pc_range = radius_range, pc_range
        self.radius = int((radius_range[1] - radius_range[0]) / radius_range[-1])
        self.im_scale = 4.
        self.transformer_layers = _RayTransformer(embed_dims, 8, embed_dims * 4, num_decoder_layers)
        self._const = {}
        self._params = None

    # -- host-side, feature independent (:538-572)
    def ray_grid(self, img_meta, H, W):
        """BEV sampling grid of the ray queries of one sample: (V, W*R, 2) float32 in [-1,1], tokens ordered (w, r);
        and the camera centres (V,2).  numpy float32 on the host (a few thousand points)."""
        R, f32 = self.radius, np.float32
        l2i = np.asarray(img_meta['lidar2img'], dtype=f32)
        c2l = np.asarray(img_meta['cam2lidar'], dtype=f32)
        V = l2i.shape[0]
        xr = np.arange(W, dtype=f32)
        cam = np.stack([(xr + f32(0.5)) * f32(self.im_scale), np.full(W, (H // 2) * self.im_scale, f32),
                        np.ones(W, f32), np.ones(W, f32)], 0)                    # (4,W): depth-1 points of row H//2
        on_ray = np.matmul(np.linalg.inv(l2i), cam[None])[:, :2]                 # (V,2,W)
        cam_xy = c2l[:, :2, 3]                                                   # (V,2)
        d = on_ray - cam_xy[:, :, None]
        d = d / np.sqrt((d * d).sum(1, keepdims=True))
        rr = self.radius_range
        depths = (np.arange(rr[0], rr[1], rr[2]) + rr[2] / 2).astype(f32)
        centers = depths[None, None, :, None] * d[:, :, None, :]                 # (V,2,R,W): NOT offset by the camera
        centers = centers.transpose(0, 3, 2, 1).reshape(-1, 2)                   # rows ordered (v, w, r)
        A = aug_affine(img_meta, False).astype(f32)
        p = centers @ A[:9].reshape(3, 3)[:2] + A[9:]                            # z = 0 before the augmentation
        r = self.pc_range
        g = np.stack([(p[:, 0] - f32(r[0])) / f32(r[3] - r[0]), (p[:, 1] - f32(r[1])) / f32(r[4] - r[1])], -1)
        g = g * f32(2) - f32(1)
        return torch.from_numpy(g.reshape(V, W * R, 2).astype(f32)), torch.from_numpy(np.ascontiguousarray(cam_xy))

    def sample_rays(self, img_metas, pts_metas, b, H, W, device):
        """Per-sample ray geometry on the device, cached with the rest of the sample's geometry for the forward."""
        geom = eu.sample_geometry(img_metas, pts_metas, b, (H, W), device)
        key = (H, W, self.radius, tuple(self.radius_range))
        if getattr(geom, 'polar_key', None) != key:
            grid, cam_xy = self.ray_grid(img_metas[b], H, W)
            geom.polar = (grid.to(device), cam_xy.to(device))
            geom.polar_key = key
        return geom

    def static_geometry_record(self, geom, img_meta):
        """Device copies of the ray geometry of a sample that will later run behind a captured hipGraph
        (`GraphedHotPath.prepare`), or None when this operator has not built rays for `geom` yet."""
        if getattr(geom, 'polar_key', None) is None:
            return None
        H, W = geom.polar_key[:2]
        grid, cam_xy = self.ray_grid(img_meta, H, W)
        return grid.to(geom.polar[0].device), cam_xy.to(geom.polar[1].device)

    def load_static_geometry(self, geom, rec):
        """New sample behind a captured hipGraph: rewrite the ray geometry IN PLACE (the graph has the addresses)."""
        if rec is not None:
            geom.polar[0].copy_(rec[0], non_blocking=True)
            geom.polar[1].copy_(rec[1], non_blocking=True)

    def constants(self, H, W, dtype, device):
        key = (H, W, dtype, device)
        if key not in self._const:
            R, nf = self.radius, self.embed_dims // 2
            self._const[key] = (sine_position_tokens(W, R, nf, x_major=True).to(device, dtype),   # rays, (w, r) order
                                sine_position_tokens(W, H, nf).to(device, dtype))                  # image, (h, w) order
        return self._const[key]

    def forward(self, lidar_feat, img_feat, img_metas, pts_metas, **kwargs):
        lidar_feat = ops.cl(lidar_feat)
        B, C, Hb, Wb = lidar_feat.shape
        _, _, H, W = img_feat.shape
        R, dev, dt = self.radius, lidar_feat.device, lidar_feat.dtype
        V = img_feat.shape[0] // B
        geoms = [self.sample_rays(img_metas, pts_metas, b, H, W, dev) for b in range(B)]
        grid = (geoms[0].polar[0] if B == 1 else torch.cat([g.polar[0] for g in geoms])).view(B * V, W * R, 2)
        cam_xy = (geoms[0].polar[1][None] if B == 1 else torch.stack([g.polar[1] for g in geoms])).contiguous()
        ray_pos, img_pos = self.constants(H, W, dt, dev)
        live = torch.is_grad_enabled() and (lidar_feat.requires_grad or img_feat.requires_grad
                                            or any(p.requires_grad for p in self.parameters()))
        if live:
            rays = GridGather.apply(lidar_feat, grid, ray_pos, V)
        else:
            rays = ops.grid_gather(lidar_feat, grid, add=ray_pos, grids_per_feat=V)         # (B*V, W*R, C)
        rays = rays.view(B * V * W, R, C)
        cols = (_tokens(img_feat) + img_pos).view(B * V, H, W, C).transpose(1, 2).reshape(B * V * W, H, C)
        polar = self.transformer_layers(cols, rays).view(B, V, W, R, C)                     # ray-major polar maps
        proj = geoms[0].lidar2img[None] if B == 1 else torch.stack([g.lidar2img for g in geoms])
        aug = geoms[0].aug_rev[None] if B == 1 else torch.stack([g.aug_rev for g in geoms])
        ishape = img_metas[0]['input_shape']                                                 # sample 0 for all (:606)
        pkey = (dev, float(ishape[0]), float(ishape[1]))
        if self._params is None or self._params[0] != pkey:
            self._params = (pkey, torch.tensor(list(self.pc_range) + [float(ishape[0]), float(ishape[1]),
                                                                      float(self.radius_range[0]), float(R)],
                                               dtype=torch.float32, device=dev))
        params = self._params[1]
        if live:
            return PolarBEVSample.apply(polar.contiguous(), lidar_feat, proj.contiguous(), aug.contiguous(), cam_xy, params)
        return ops.polar_bev_sample(polar, lidar_feat, proj.contiguous(), aug.contiguous(), cam_xy, params)


@TRANSFORMER_LAYER.register_module()
class DeepInteractionLayer(nn.Module):
    """Reference :143-218 (an mmcv `BaseTransformerLayer`: `attentions`, `ffns`, `norms` built from
    `operation_order`, post-norm) with the learnable residual `scale` of :160."""

    def __init__(self, attn_cfgs, ffn_cfgs, operation_order=None, norm_cfg=dict(type='LN'), batch_first=True,
                 init_cfg=None, **kwargs):
        super().__init__()
        self.batch_first, self.operation_order = batch_first, tuple(operation_order)
        self.pre_norm = operation_order[0] == 'norm'
        assert not self.pre_norm and norm_cfg.get('type', 'LN') == 'LN'
        n_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(n_attn)] if isinstance(attn_cfgs, dict) else list(attn_cfgs)
        assert len(attn_cfgs) == n_attn
        self.attentions = nn.ModuleList()
        for cfg in attn_cfgs:
            cfg = dict(cfg)
            cfg.setdefault('batch_first', batch_first)
            self.attentions.append(ATTENTION.build(cfg))
        self.embed_dims = self.attentions[0].embed_dims
        n_ffn = operation_order.count('ffn')
        ffn_cfgs = [copy.deepcopy(ffn_cfgs) for _ in range(n_ffn)] if isinstance(ffn_cfgs, dict) else list(ffn_cfgs)
        self.ffns = nn.ModuleList()
        for cfg in ffn_cfgs:
            cfg = dict(cfg)
            cfg.pop('type', None)
            cfg.setdefault('embed_dims', self.embed_dims)
            self.ffns.append(TransFFN(**cfg))
        self.norms = nn.ModuleList([nn.LayerNorm(self.embed_dims) for _ in range(operation_order.count('norm'))])
        self.scale = nn.Parameter(torch.ones(1))

    def forward(self, query, value, ms_query, reference_points, spatial_shapes, level_start_index, img_metas,
                pts_metas, **kwargs):
        qh, qw = query.shape[-2:]
        query = _tokens(query)
        ni = ai = fi = 0
        self_feat = None
        order = self.operation_order[:-2]
        fused = False                              # the previous step already applied this 'norm'
        for i, op in enumerate(order):
            nxt = self.norms[ni] if i + 1 < len(order) and order[i + 1] == 'norm' else None
            if op == 'self_attn':
                att = self.attentions[ai]
                if nxt is not None and isinstance(att, MultiScaleDeformableAttention) and att.can_fuse_norm(query, nxt):
                    # output projection + residual + the following norm in one launch that also keeps the un-normalised sum
                    query, self_feat = att(query=query, value=ms_query, identity=None, reference_points=reference_points,
                                           spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                                           then_norm=nxt, with_sum=True)
                    fused = True
                else:
                    query = att(query=query, value=ms_query, identity=None, reference_points=reference_points,
                                spatial_shapes=spatial_shapes, level_start_index=level_start_index)
                    self_feat = query
                ai += 1
            elif op == 'norm':
                if not fused:
                    query = post_norm(self.norms[ni], query)
                fused = False
                ni += 1
            elif op == 'cross_attn':
                att = self.attentions[ai]
                kw = {}
                if nxt is not None and isinstance(att, MMRI_P2I):
                    kw['then_norm'], fused = nxt, True
                out = att(_map(query, qh, qw), value, img_metas=img_metas, pts_metas=pts_metas,
                          reference_points=reference_points[:, :, 0:1, :], spatial_shapes=spatial_shapes,
                          level_start_index=level_start_index, **kw)
                query = _tokens(out)
                ai += 1
            elif op == 'ffn':
                if nxt is not None:
                    query, fused = self.ffns[fi].then_norm(nxt, query), True
                else:
                    query = self.ffns[fi](query, None)
                fi += 1
        tail = self.operation_order[-2:]
        if tail == ('ffn', 'norm'):
            self_feat = self.ffns[fi].then_norm(self.norms[ni], self_feat)
        else:
            for op in tail:
                if op == 'norm':
                    self_feat = post_norm(self.norms[ni], self_feat)
                    ni += 1
                elif op == 'ffn':
                    self_feat = self.ffns[fi](self_feat)
                    fi += 1
        return _map(torch.addcmul(self_feat, query, self.scale.to(query.dtype)), qh, qw)   # self_feat + scale * query


def reference_points(H, W, device):
    """Reference :129-138: token centres / size, (1, H*W, 2) as (x, y), float32."""
    ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing='ij')
    return torch.stack((rx.reshape(-1)[None] / W, ry.reshape(-1)[None] / H), -1).to(device)


@NECKS.register_module()
class FusionTransformerv4(nn.Module):
    """Reference :26-139."""

    def __init__(self, num_layers=2, num_lidar_maps=2, in_channels_img=64, in_channels_pts=128 * 3, hidden_channel=128,
                 bn_momentum=0.1, bias='auto', img_transformerlayers=None, pts_transformerlayers=None):
        super().__init__()
        b = bool(bias)                                         # 'auto' is a truthy string, as in the v1 neck
        self.shared_conv_pts = nn.Conv2d(in_channels_pts * num_lidar_maps, hidden_channel, 3, padding=1, bias=b)
        self.multi_scale_conv_img = nn.Conv2d(in_channels_img, hidden_channel, 3, padding=1, bias=b)
        self.multi_scale_conv_pts = nn.Conv2d(in_channels_pts, hidden_channel, 3, padding=1, bias=b)
        self.num_layers = num_layers
        self.img_fusion_blocks = nn.ModuleList(
            [TRANSFORMER_LAYER.build(copy.deepcopy(img_transformerlayers)) for _ in range(num_layers)])
        self.pts_fusion_blocks = nn.ModuleList(
            [TRANSFORMER_LAYER.build(copy.deepcopy(pts_transformerlayers)) for _ in range(num_layers)])
        self.bn_momentum = bn_momentum
        self._ref_cache = {}
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = self.bn_momentum

    def get_reference_points(self, feat):
        H, W = feat.shape[-2:]
        return reference_points(H, W, 'cpu')

    def _refs(self, H, W, L, device):
        key = (H, W, L, device)
        if key not in self._ref_cache:
            self._ref_cache[key] = reference_points(H, W, device).unsqueeze(-2).repeat(1, 1, L, 1).contiguous()
        return self._ref_cache[key]

    @staticmethod
    def _flatten(maps):
        return torch.cat([_tokens(m) for m in maps], 1), [tuple(m.shape[-2:]) for m in maps]

    def forward(self, img_feats, pts_feats, img_metas, pts_metas):
        pts_feats = list(pts_feats)
        cl = lambda t: t.contiguous(memory_format=torch.channels_last)
        pts_feat_conv = eu.conv3x3_module(self, self.shared_conv_pts, cl(pts_feats.pop(0)))              # :85
        ms_img = [eu.conv3x3_module(self, self.multi_scale_conv_img, cl(f)) for f in img_feats]
        ms_pts = [eu.conv3x3_module(self, self.multi_scale_conv_pts, cl(f)) for f in pts_feats]
        new_img, new_pts = ms_img[0], ms_pts[0]
        dev = new_img.device
        img_flat, shapes_img = self._flatten(ms_img)
        pts_flat, shapes_pts = self._flatten(ms_pts)
        ref_img = self._refs(*new_img.shape[-2:], len(ms_img), dev)
        ref_pts = self._refs(*new_pts.shape[-2:], len(ms_pts), dev)
        own_cache = eu.GEOM_KEY not in pts_metas
        own_bounds = 'pillar_batch_bounds' not in pts_metas
        if own_cache:
            pts_metas[eu.GEOM_KEY] = [None] * len(img_metas)                    # per-forward geometry / depth cache
        try:
            # The two sides of a layer read the previous layer's maps only: on the device they run as two branches (image
            # side on the caller's stream, BEV side - 32 400 tokens, launches that fill a fraction of the chip - on the side
            # stream; parallel paths under hipGraph capture), as the v1 neck does.  Every lazily cached per-sample product
            # belongs to ONE side (completed depth: image side; pillar key table, polar rays: BEV side), so each is created
            # and re-used on its own stream.
            fork = utils.OVERLAP & 1 and dev.type == 'cuda' and not torch.is_grad_enabled()
            for i in range(self.num_layers):
                img_side = lambda i=i, a=new_img, b=new_pts: self.img_fusion_blocks[i](
                    a, b, img_flat, ref_img, shapes_img, None, img_metas, pts_metas)
                pts_side = lambda i=i, a=new_img, b=new_pts: self.pts_fusion_blocks[i](
                    b, a, pts_flat, ref_pts, shapes_pts, None, img_metas, pts_metas)
                if fork:
                    t_img, t_pts = utils.fork_join(dev, img_side, pts_side)
                else:
                    t_img, t_pts = img_side(), pts_side()
                new_img, new_pts = t_img, t_pts
        finally:
            if own_cache:
                pts_metas.pop(eu.GEOM_KEY, None)
                if own_bounds:
                    pts_metas.pop('pillar_batch_bounds', None)
        return new_img, [pts_feat_conv, new_pts]
