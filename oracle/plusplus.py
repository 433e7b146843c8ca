"""ORACLE for DeepInteraction++ - TEST INFRASTRUCTURE, never imported by the product.

Pure-PyTorch fp32 CPU restatement with the reference's `state_dict` keys of

  FusionTransformerv4            necks/fusion_transformerv4.py:26-139
  DeepInteractionLayer           necks/fusion_transformerv4.py:143-218   (mmcv BaseTransformerLayer subclass)
  MMRI_P2I (++)                  necks/fusion_transformerv4.py:220-240   (BEVWarp + 1-level deformable attention)
  MMRI_I2P (++)                  necks/fusion_transformerv4.py:242-364   (v1 pillar attention + residual, :364)
  MMRI_I2P_Polar                 necks/fusion_transformerv4.py:470-640   (+ TransSinePositionalEncoding :367-467,
                                 FlashMultiheadAttention :715-760, FlashTransformerDecoderLayer :763-769)
  ImageRCNNBlockV2 / PointRCNNBlockV2   models/utils/decoder_utils.py:844-1089
  DeepInteractionPlusPlusDecoder        dense_heads/deepinteractionplusplus_decoder.py:20-320 (forward)

Pinned against the reference's own Python in tests/test_oracle_vs_reference.py (the mmcv / flash-attn /
nn.Transformer boundaries are restated: oracle/thirdparty.py, parity unpinned there).  The reference runs its
flash attention and the ++ pillar attention in fp16 (`auto_fp16`); the oracle keeps fp32 throughout.

Two behaviours of the published code are reproduced on purpose because results must be identical:
  * V2 RoI blocks mix `query (1,n,C) * scale + self_feat (n,1,C) * self_scale`, which broadcasts to (n,n,C)
    and then takes row 0: every query receives the self-branch feature of the FIRST query of its group
    (decoder_utils.py:986-990, :1086-1089).
  * `PointRCNNBlockV2.forward` ends without a `return` (decoder_utils.py:1089), so the published ++ decoder
    raises at `query_feat, on_the_image = ...` (deepinteractionplusplus_decoder.py:286).  The evident intent
    (v1: `return query_feat, None`, decoder_utils.py:841) is what the oracle and the product implement; the
    pin test recovers the local `query_feat` of the reference function at exit.
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import decoder as odec
from . import encoder as oenc
from .thirdparty import MultiScaleDeformableAttention, TransFFN, apply_3d_transformation

PC_RANGE = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]


# --------------------------------------------------------------------------- attention modules
class MMRI_P2I(nn.Module):
    """fusion_transformerv4.py:220-240."""

    def __init__(self, embed_dims, batch_first=True):
        super().__init__()
        self.embed_dims = embed_dims
        self.Warp = oenc.BEVWarp()
        self.Local = MultiScaleDeformableAttention(embed_dims, num_levels=1, batch_first=batch_first)

    def forward(self, img_feats, lidar_feats, img_metas, pts_metas, reference_points=None, **kwargs):
        B = lidar_feats.size(0)
        _, C, H, W = img_feats.shape
        warped = self.Warp(lidar_feats, img_feats.reshape(B, -1, C, H, W), img_metas, pts_metas)
        q = img_feats.flatten(-2).permute(0, 2, 1)
        v = warped.reshape(-1, C, H * W).permute(0, 2, 1)
        out = self.Local(query=q, value=v, reference_points=reference_points, spatial_shapes=[(H, W)],
                         level_start_index=None)
        return out.permute(0, 2, 1).reshape(-1, C, H, W)


class MMRI_I2P(oenc.MMRI_I2P):
    """fusion_transformerv4.py:242-364: the v1 pillar attention (group_attn == plain masked attention; the
    fp16 MultiheadAttentionFP16 wrapper only changes precision) plus the residual of :364."""

    def __init__(self, embed_dims, dropout, batch_first=True, fp16_enabled=False, flash_attn=False,
                 group_attn_enabled=False):
        assert not flash_attn
        super().__init__(embed_dims, embed_dims, dropout)
        self.embed_dims = embed_dims

    def forward(self, lidar_feat, img_feat, img_metas, pts_metas, **kwargs):
        B = lidar_feat.size(0)
        _, C, H, W = img_feat.shape
        return super().forward(lidar_feat, img_feat.reshape(B, -1, C, H, W), img_metas, pts_metas) + lidar_feat


def sine_position_encoding(x_range, y_range, num_feats, temperature=10000):
    """TransSinePositionalEncoding.forward, un-normalised 2-D branch (:413-425,:453-467): x_range (B,W),
    y_range (B,H) -> (B, 2*num_feats, H, W) = cat(pos_y, pos_x)."""
    B, W = x_range.shape
    H = y_range.shape[-1]
    x_embed = x_range.unsqueeze(-2).repeat(1, H, 1)
    y_embed = y_range.unsqueeze(-1).repeat(1, 1, W)
    dim_t = torch.arange(num_feats, dtype=torch.float32, device=x_range.device)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode='floor') / num_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


class PackedMHA(nn.Module):
    """FlashMultiheadAttention (:715-760): packed in-projection, softmax(QK^T/sqrt(d))V per head, out_proj.
    Keys `in_proj_weight`, `in_proj_bias`, `out_proj.*`.  No dropout in eval; fp32 here."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, q, k, v):
        E, h = self.embed_dim, self.num_heads
        wq, wk, wv = self.in_proj_weight.chunk(3)
        bq, bk, bv = self.in_proj_bias.chunk(3)
        q, k, v = F.linear(q, wq, bq), F.linear(k, wk, bk), F.linear(v, wv, bv)
        B, T, _ = q.shape
        S = k.shape[1]
        q = q.view(B, T, h, E // h).transpose(1, 2)
        k = k.view(B, S, h, E // h).transpose(1, 2)
        v = v.view(B, S, h, E // h).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(E // h), -1)
        return self.out_proj((a @ v).transpose(1, 2).reshape(B, T, E))


class _DecoderLayer(nn.Module):
    """torch 1.9 nn.TransformerDecoderLayer (post-norm, ReLU) with the two attentions swapped for
    FlashMultiheadAttention (:763-769)."""

    def __init__(self, d, heads, ff):
        super().__init__()
        self.self_attn, self.multihead_attn = PackedMHA(d, heads), PackedMHA(d, heads)
        self.linear1, self.linear2 = nn.Linear(d, ff), nn.Linear(ff, d)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d), nn.LayerNorm(d), nn.LayerNorm(d)

    def forward(self, tgt, memory):
        tgt = self.norm1(tgt + self.self_attn(tgt, tgt, tgt))
        tgt = self.norm2(tgt + self.multihead_attn(tgt, memory, memory))
        return self.norm3(tgt + self.linear2(F.relu(self.linear1(tgt))))


class _Stack(nn.Module):
    def __init__(self, layers, d):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.norm = nn.LayerNorm(d)


class _Transformer(nn.Module):
    """nn.Transformer(num_encoder_layers=0, custom_decoder=...) (:486-492): the encoder is zero layers followed
    by its final LayerNorm; keys `encoder.norm.*`, `decoder.layers.0.*`, `decoder.norm.*`."""

    def __init__(self, d, heads, ff, num_decoder_layers):
        super().__init__()
        self.encoder = _Stack([], d)
        self.decoder = _Stack([_DecoderLayer(d, heads, ff) for _ in range(num_decoder_layers)], d)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, src, tgt):
        memory = self.encoder.norm(src)
        for layer in self.decoder.layers:
            tgt = layer(tgt, memory)
        return self.decoder.norm(tgt)


class MMRI_I2P_Polar(nn.Module):
    """fusion_transformerv4.py:470-640."""

    def __init__(self, embed_dims, dropout, batch_first=True, radius_range=[1., 61., 1.0], num_decoder_layers=1,
                 pc_range=PC_RANGE):
        super().__init__()
        self.embed_dims, self.radius_range, self.pc_range = embed_dims, radius_range, pc_range
        self.radius = int((radius_range[1] - radius_range[0]) / radius_range[-1])
        self.im_scale = 4.
        self.transformer_layers = _Transformer(embed_dims, 8, embed_dims * 4, num_decoder_layers)

    def polar_geometry(self, cam_id, lidar2img, cam2lidar, img_metas, H, W):
        """:538-572: BEV sampling grid of the polar ray queries of one camera, (B,R,W,2) in [-1,1]."""
        B, R, dev = lidar2img.shape[0], self.radius, lidar2img.device
        xr = torch.arange(0., float(W), 1., device=dev).unsqueeze(0).repeat(B, 1)
        cam_coors = torch.stack([xr + 0.5, torch.zeros_like(xr) + H // 2, torch.ones_like(xr), torch.ones_like(xr)], 1)
        cam_coors[:, :2] *= self.im_scale
        img2lidar = torch.linalg.inv(lidar2img[:, cam_id])
        on_ray = torch.bmm(img2lidar, cam_coors)[:, :2]
        cam_xy = cam2lidar[:, cam_id, :2, -1:]
        dirs = on_ray - cam_xy
        dirs = dirs / dirs.norm(dim=1, p=2, keepdim=True)
        rr = self.radius_range
        depths = torch.arange(rr[0], rr[1], rr[2]) + rr[2] / 2
        centers = (depths[None, None, :, None].to(dirs) * dirs[:, :, None]).permute(0, 2, 3, 1)   # (B,R,W,2)
        grids = []
        for b in range(B):
            c = centers[b].reshape(R * W, 2)
            c = torch.cat([c, torch.zeros_like(c[:, :1])], -1)
            c = apply_3d_transformation(c, 'LIDAR', img_metas[b], reverse=False).view(R, W, 3)
            g = torch.zeros_like(c[..., :2])
            g[..., 0] = (c[..., 0] - self.pc_range[0]) / (self.pc_range[3] - self.pc_range[0])
            g[..., 1] = (c[..., 1] - self.pc_range[1]) / (self.pc_range[4] - self.pc_range[1])
            grids.append(g)
        return torch.stack(grids, 0) * 2 - 1, cam_xy

    def bev_sampling(self, cam_id, lidar2img, cam_xy, img_metas, Hb, Wb):
        """:581-628: where every BEV cell reads the camera's polar map: (B,Hb,Wb,2) grid + (B,1,Hb,Wb) mask."""
        B = lidar2img.shape[0]
        assert Hb == Wb, 'the reference mixes the two BEV axes (:581-588, :624-625); square maps only'
        zs = 10
        by, bx, bz = torch.meshgrid(torch.linspace(0, Hb - 1, Hb) + 0.5, torch.linspace(0, Wb - 1, Wb) + 0.5,
                                    torch.linspace(0, zs - 1, zs) + 0.5, indexing='ij')
        r = self.pc_range
        bx = bx / Hb * (r[3] - r[0]) + r[0]
        by = by / Wb * (r[4] - r[1]) + r[1]
        bz = bz / zs * (r[5] - r[2]) + r[2]
        pts = torch.stack([bx, by, bz], -1).reshape(-1, 3)
        re = torch.stack([apply_3d_transformation(pts, 'LIDAR', img_metas[b], reverse=True) for b in range(B)], 0)
        re = torch.cat((re, torch.ones_like(re[..., :1])), -1).transpose(1, 2).to(lidar2img.device)   # (B,4,N)
        xyz = torch.bmm(lidar2img[:, cam_id], re)[:, :3].transpose(1, 2)
        eps = 1e-5
        mask = xyz[..., 2:3] > eps
        xy = xyz[..., 0:2] / torch.maximum(xyz[..., 2:3], torch.ones_like(xyz[..., 2:3]) * eps)
        ishape = img_metas[0]['input_shape']                                                    # sample 0 (:606)
        xy = torch.stack([xy[..., 0] / ishape[1], xy[..., 1] / ishape[0]], -1)
        xy = 2 * xy - 1
        mask = (mask & (xy[..., 0:1] > -1.0) & (xy[..., 0:1] < 1.0) & (xy[..., 1:2] > -1.0) & (xy[..., 1:2] < 1.0))
        radius = torch.norm(re[:, :2, :] - cam_xy, dim=1)
        nr = (2 * (radius - self.radius_range[0]) / self.radius - 1).clamp(-1, 1)
        loc = torch.stack([xy[..., 0], nr], -1).view(B, Hb, Wb, zs, 2).mean(dim=3)
        mask = mask.view(B, Hb, Wb, zs, 1).sum(dim=3).permute(0, 3, 1, 2) > 0
        return loc, mask

    def forward(self, lidar_feat, img_feat, img_metas, pts_metas, **kwargs):
        B = lidar_feat.size(0)
        _, C, H, W = img_feat.shape
        R = self.radius
        img_feat = img_feat.reshape(B, -1, C, H, W)
        out = torch.zeros_like(lidar_feat)
        visibles = torch.zeros_like(out[:, 0:1])
        lidar2img = oenc.lidar2img_tensor(img_metas, lidar_feat)
        cam2lidar = lidar_feat.new_tensor([[list(map(list, m)) for m in meta['cam2lidar']] for meta in img_metas])
        dev = lidar_feat.device
        xr = torch.arange(0., float(W), 1., device=dev).unsqueeze(0).repeat(B, 1)
        img_pos = sine_position_encoding(xr, torch.arange(0., float(H), 1., device=dev).unsqueeze(0).repeat(B, 1),
                                         self.embed_dims // 2)
        ray_pos = sine_position_encoding(xr, torch.arange(0., float(R), 1., device=dev).unsqueeze(0).repeat(B, 1),
                                         self.embed_dims // 2)
        for cam in range(img_feat.shape[1]):
            grid, cam_xy = self.polar_geometry(cam, lidar2img, cam2lidar, img_metas, H, W)
            query = F.grid_sample(lidar_feat, grid, mode='bilinear', padding_mode='zeros', align_corners=False) + ray_pos
            rays = query.permute(2, 0, 3, 1).flatten(1, 2)                                      # (R, B*W, C)
            cols = (img_feat[:, cam] + img_pos).permute(2, 0, 3, 1).flatten(1, 2)               # (H, B*W, C)
            bev_out = self.transformer_layers(cols.transpose(0, 1), rays.transpose(0, 1))       # (B*W, R, C)
            bev_out = bev_out.view(B, W, R, C).permute(0, 3, 2, 1)                              # (B,C,R,W)
            loc, mask = self.bev_sampling(cam, lidar2img, cam_xy, img_metas, lidar_feat.shape[-2], lidar_feat.shape[-1])
            sampled = F.grid_sample(bev_out, loc, mode='bilinear', padding_mode='zeros', align_corners=False)
            out = out + sampled * mask
            visibles = visibles + mask
        visibles = torch.where(visibles == 0, torch.ones_like(visibles), visibles)
        return out / visibles + lidar_feat


# --------------------------------------------------------------------------- transformer layer + neck
ATTENTION = {'MultiScaleDeformableAttention': MultiScaleDeformableAttention, 'MMRI_P2I': MMRI_P2I,
             'MMRI_I2P': MMRI_I2P, 'MMRI_I2P_Polar': MMRI_I2P_Polar}


class DeepInteractionLayer(nn.Module):
    """fusion_transformerv4.py:143-218 on top of mmcv 1.3.18 `BaseTransformerLayer.__init__` (attentions / ffns /
    norms built from `operation_order`; post-norm since the order does not start with 'norm')."""

    def __init__(self, attn_cfgs, ffn_cfgs, operation_order=None, norm_cfg=dict(type='LN'), batch_first=True, **kwargs):
        super().__init__()
        self.batch_first, self.operation_order = batch_first, tuple(operation_order)
        self.pre_norm = operation_order[0] == 'norm'
        n_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(n_attn)] if isinstance(attn_cfgs, dict) else list(attn_cfgs)
        assert len(attn_cfgs) == n_attn and not self.pre_norm
        self.attentions = nn.ModuleList()
        for cfg in attn_cfgs:
            cfg = dict(cfg)
            cfg.setdefault('batch_first', batch_first)
            self.attentions.append(ATTENTION[cfg.pop('type')](**cfg))
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
        vh, vw = value.shape[-2:]
        E = self.embed_dims
        query = query.flatten(-2).permute(0, 2, 1)
        value = value.flatten(-2).reshape(-1, E, vh, vw)
        ni = ai = fi = 0
        for op in self.operation_order[:-2]:
            if op == 'self_attn':
                query = self.attentions[ai](query=query, value=ms_query, identity=None, reference_points=reference_points,
                                            spatial_shapes=spatial_shapes, level_start_index=level_start_index)
                ai += 1
                self_feat = query
            elif op == 'norm':
                query = self.norms[ni](query)
                ni += 1
            elif op == 'cross_attn':
                q_map = query.permute(0, 2, 1).reshape(-1, E, qh, qw)
                q_map = self.attentions[ai](q_map, value, img_metas=img_metas, pts_metas=pts_metas,
                                            reference_points=reference_points[:, :, 0:1, :],
                                            spatial_shapes=spatial_shapes, level_start_index=level_start_index)
                query = q_map.reshape(-1, E, qh * qw).permute(0, 2, 1)
                ai += 1
            elif op == 'ffn':
                query = self.ffns[fi](query, None)
                fi += 1
        for op in self.operation_order[-2:]:
            if op == 'norm':
                self_feat = self.norms[ni](self_feat)
                ni += 1
            elif op == 'ffn':
                self_feat = self.ffns[fi](self_feat)
                fi += 1
        query = self_feat + self.scale * query
        return query.permute(0, 2, 1).reshape(-1, E, qh, qw)


def reference_points(H, W):
    """FusionTransformerv4.get_reference_points (:129-138): cell centres / size, (1, H*W, 2) as (x, y)."""
    ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing='ij')
    return torch.stack((rx.reshape(-1)[None] / W, ry.reshape(-1)[None] / H), -1)


class FusionTransformerv4(nn.Module):
    """fusion_transformerv4.py:26-139."""

    def __init__(self, num_layers=2, num_lidar_maps=2, in_channels_img=64, in_channels_pts=128 * 3, hidden_channel=128,
                 bn_momentum=0.1, bias='auto', img_transformerlayers=None, pts_transformerlayers=None):
        super().__init__()
        b = bool(bias)
        self.shared_conv_pts = nn.Conv2d(in_channels_pts * num_lidar_maps, hidden_channel, 3, padding=1, bias=b)
        self.multi_scale_conv_img = nn.Conv2d(in_channels_img, hidden_channel, 3, padding=1, bias=b)
        self.multi_scale_conv_pts = nn.Conv2d(in_channels_pts, hidden_channel, 3, padding=1, bias=b)
        self.num_layers = num_layers

        def build(cfg):
            cfg = dict(copy.deepcopy(cfg))
            assert cfg.pop('type') == 'DeepInteractionLayer'
            return DeepInteractionLayer(**cfg)
        self.img_fusion_blocks = nn.ModuleList([build(img_transformerlayers) for _ in range(num_layers)])
        self.pts_fusion_blocks = nn.ModuleList([build(pts_transformerlayers) for _ in range(num_layers)])
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = bn_momentum

    @staticmethod
    def flatten_levels(maps):
        shapes = [tuple(m.shape[-2:]) for m in maps]
        return torch.cat([m.flatten(-2).permute(0, 2, 1) for m in maps], -2), shapes

    def forward(self, img_feats, pts_feats, img_metas, pts_metas):
        pts_feats = list(pts_feats)
        pts_feat_conv = self.shared_conv_pts(pts_feats.pop(0))                     # :85 (pops the stacked map)
        ms_img = [self.multi_scale_conv_img(f) for f in img_feats]
        ms_pts = [self.multi_scale_conv_pts(f) for f in pts_feats]
        new_img, new_pts = ms_img[0], ms_pts[0]
        ref_img = reference_points(*new_img.shape[-2:]).unsqueeze(-2).repeat(1, 1, len(ms_img), 1)
        img_flat, shapes_img = self.flatten_levels(ms_img)
        ref_pts = reference_points(*new_pts.shape[-2:]).unsqueeze(-2).repeat(1, 1, len(ms_pts), 1)
        pts_flat, shapes_pts = self.flatten_levels(ms_pts)
        for i in range(self.num_layers):
            t_img = self.img_fusion_blocks[i](new_img, new_pts, img_flat, ref_img, shapes_img, None, img_metas, pts_metas)
            t_pts = self.pts_fusion_blocks[i](new_pts, new_img, pts_flat, ref_pts, shapes_pts, None, img_metas, pts_metas)
            new_img, new_pts = t_img, t_pts
        return new_img, [pts_feat_conv, new_pts]


# --------------------------------------------------------------------------- decoder (++)
class _V2Refine:
    def _v2_init(self, hidden, dropout, sfx):
        self.ffn = TransFFN(embed_dims=hidden, feedforward_channels=hidden * 4, num_fcs=2, ffn_drop=dropout)
        self.self_ffn = TransFFN(embed_dims=hidden, feedforward_channels=hidden * 4, num_fcs=2, ffn_drop=dropout)
        setattr(self, 'self_norm' + sfx, nn.LayerNorm(hidden))
        self.scale = nn.Parameter(torch.ones(1) * 0.5)
        self.self_scale = nn.Parameter(torch.ones(1) * 0.5)
        for n in ('linear1', 'linear2'):
            delattr(self, n + sfx)

    def refine(self, q, roi, sfx=''):
        """decoder_utils.py:972-990 / :1069-1089.  q (n,1,C) sequence-first, roi (49,n,C) -> (1,n,C).  The mix
        broadcasts (1,n,C) with (n,1,C) and row 0 is what the caller keeps: see the module docstring."""
        g = lambda n: getattr(self, n + sfx)
        q = g('norm1')(q + g('dyconv_pre_self_attn')(q, q, q)[0])
        self_feat = q.clone()
        q = q.permute(1, 0, 2)
        q = g('norm2')(q + g('dyconv')(q, roi))
        q = g('norm3')(self.ffn(q))
        self_feat = g('self_norm')(self.self_ffn(self_feat))
        mixed = q * self.scale + self_feat * self.self_scale                       # (n,n,C)
        return mixed[0:1]


class ImageRCNNBlockV2(_V2Refine, odec.ImageRCNNBlock):
    """decoder_utils.py:844-993."""

    def __init__(self, num_views, num_proposals, out_size_factor_img, test_cfg, bbox_coder, hidden, heads, dropout):
        odec.ImageRCNNBlock.__init__(self, num_views, num_proposals, out_size_factor_img, test_cfg, bbox_coder, hidden,
                                     heads, dropout)
        self._v2_init(hidden, dropout, '')


class PointRCNNBlockV2(_V2Refine, odec.PointRCNNBlock):
    """decoder_utils.py:997-1089."""

    def __init__(self, hidden, heads, dropout, bbox_coder):
        odec.PointRCNNBlock.__init__(self, hidden, heads, dropout, bbox_coder)
        self._v2_init(hidden, dropout, '_pts')


class DeepInteractionPlusPlusDecoder(odec.DeepInteractionDecoder):
    """dense_heads/deepinteractionplusplus_decoder.py:20-320 (forward; loss / targets are the v1 code with the
    per-layer mask of :513-514)."""

    def __init__(self, num_views=6, out_size_factor_img=4, num_proposals=200, hidden_channel=128, num_classes=10,
                 num_mmpi=4, num_heads=8, dropout=0.1, common_heads=None, num_heatmap_convs=2, bbox_coder=None,
                 test_cfg=None, bn_momentum=0.1, **kw):
        super().__init__(num_views=num_views, out_size_factor_img=out_size_factor_img, num_proposals=num_proposals,
                         hidden_channel=hidden_channel, num_classes=num_classes, num_mmpi=num_mmpi, num_heads=num_heads,
                         dropout=dropout, common_heads=common_heads, num_heatmap_convs=num_heatmap_convs,
                         bbox_coder=bbox_coder, test_cfg=test_cfg, bn_momentum=bn_momentum, **kw)
        h = hidden_channel
        heads = dict(copy.deepcopy(common_heads))
        heads.update(heatmap=(num_classes, num_heatmap_convs))
        self.decode_head, self.pred_head = nn.ModuleList(), nn.ModuleList()
        for _ in range(num_mmpi // 2):
            self.decode_head.append(ImageRCNNBlockV2(num_views, num_proposals, out_size_factor_img, test_cfg,
                                                     self.bbox_coder, h, num_heads, dropout))
            self.pred_head.append(odec.FFN(h, heads))                               # :140 (128 in, not 256)
            self.decode_head.append(PointRCNNBlockV2(h, num_heads, dropout, self.bbox_coder))
            self.pred_head.append(odec.FFN(h, heads))
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = bn_momentum

    def mmpi(self, query_feat, res, first, new_lidar_feat, img_flat, img_metas, I_H, I_W):
        """:279-303: look-forward centre update and the cumulative image mask."""
        self.on_the_image_mask, rets = [], []
        look_forward = res['center'].detach().clone()                              # :281
        for l in range(self.num_mmpi):
            prev = query_feat.clone()
            query_pos = res['center'].detach().clone()
            query_feat, on = self.decode_head[l](prev, res, new_lidar_feat, img_flat, img_metas, I_H, I_W)
            res = self.pred_head[l](query_feat)
            delta = res['center'].clone()
            res['center'] = delta + look_forward                                   # :293
            look_forward = delta + query_pos                                       # :294
            if l % 2 == 0:
                m = on != -1
                self.on_the_image_mask.append(m if l == 0 else (m & self.on_the_image_mask[-1]))
            else:
                self.on_the_image_mask.append(self.on_the_image_mask[-1])
            m = self.on_the_image_mask[-1]
            for k in res:                                                          # :300-302, every layer
                res[k] = torch.where(m.unsqueeze(1), res[k], first[k])
            rets.append(res)
        return rets
