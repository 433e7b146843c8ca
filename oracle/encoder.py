"""Oracle restatement of the MMRI encoder (TEST INFRASTRUCTURE, PyTorch fp32 CPU).

Follows, function by function (all paths under
/root/reference/projects/mmdet3d_plugin/models/):
  utils/encoder_utils.py      ConvBNReLU :11-34, LocalContextAttentionBlock :84-135,
                              BEVWarp :137-199, MMRI_P2I :202-213, MMRI_I2P :216-320
  necks/deepinteraction_encoder.py  DeepInteractionEncoderLayer :8-33, DeepInteractionEncoder :35-85

Module/parameter names equal the reference's so a reference `state_dict` loads
(SURVEY.md 8(b)).  The geometry is split into small functions so tests can pin
each stage (projection -> scatter -> completion -> un-projection -> gather).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import depth_completion
from .locatt import TorchLocatt
from .thirdparty import apply_3d_transformation

PC_RANGE = (-54.0, -54.0, -5.0, 54.0, 54.0, 3.0)   # hard-coded at encoder_utils.py:190


class ConvBNReLU(nn.Module):
    """encoder_utils.py:11-34 (bias='auto' -> bias only when there is no norm)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, norm=True, act=True, affine=True):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, 1, (kernel_size - 1) // 2,
                              bias=not norm)
        self.use_norm, self.use_activation = norm, act
        if norm:
            self.bn = nn.BatchNorm2d(out_channels, affine=affine)

    def forward(self, x):
        x = self.conv(x)
        if self.use_norm:
            x = self.bn(x)
        return F.relu(x) if self.use_activation else x


class LocalContextAttentionBlock(nn.Module):
    """encoder_utils.py:84-135."""

    def __init__(self, in_channels, out_channels, kernel_size, locatt=TorchLocatt):
        super().__init__()
        self.kernel_size = kernel_size
        self.locatt = locatt
        self.query_project = nn.Sequential(ConvBNReLU(in_channels, out_channels, 1),
                                           ConvBNReLU(out_channels, out_channels, 1))
        self.key_project = nn.Sequential(ConvBNReLU(in_channels, out_channels, 1),
                                         ConvBNReLU(out_channels, out_channels, 1))
        self.value_project = ConvBNReLU(in_channels, out_channels, 1)
        for m in self.modules():                         # :120-125
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, target_feats, source_feats):
        q = self.query_project(target_feats)
        k = self.key_project(source_feats)
        v = self.value_project(source_feats)
        ks = self.kernel_size
        w = self.locatt.similar_forward(q, k, ks, ks)                # :132
        w = torch.softmax(w / math.sqrt(k.size(1)), -1)              # :133
        return self.locatt.weighting_forward(v, w, ks, ks)           # :134


# ------------------------------------------------------------------ geometry
def lidar2img_tensor(img_metas, like):
    """encoder_utils.py:144-148 / :260-264: float64 lists -> float32 tensor (B,6,4,4)."""
    return like.new_tensor(np.asarray([m['lidar2img'] for m in img_metas]))


def project_to_views(pts3, proj, ori_H, ori_W):
    """encoder_utils.py:157-171 and :281-296.

    pts3 (n,3), proj (6,4,4) -> depth (6,n) camera z BEFORE the divide, uv (6,n,2)
    in input pixels, norm (6,n,2) in [-1,1], mask (6,n) (z>1e-5 and strictly inside)."""
    pts4 = torch.cat((pts3, torch.ones_like(pts3[..., :1])), -1).unsqueeze(0).unsqueeze(-1)
    cam = torch.matmul(proj.unsqueeze(1), pts4).squeeze(-1)          # (6,n,4)
    eps = 1e-5
    depth = cam[..., 2]
    mask = cam[..., 2:3] > eps
    uv = cam[..., 0:2] / torch.maximum(cam[..., 2:3], torch.ones_like(cam[..., 2:3]) * eps)
    px = (uv[..., 0:1] / ori_W - 0.5) * 2
    py = (uv[..., 1:2] / ori_H - 0.5) * 2
    mask = mask & (px > -1.0) & (px < 1.0) & (py > -1.0) & (py < 1.0)
    return depth, uv, torch.cat((px, py), -1), mask[..., 0]


def scatter_depth(uv, depth, mask, ori_H, ori_W, I_H, I_W):
    """encoder_utils.py:172-174.  INT op.  Duplicate pixels: the reference's
    `index_put_` leaves it to write order (last writer wins; nondeterministic on a
    GPU).  Contract fixed here: the HIGHEST POINT INDEX wins (= sequential CPU order).
    Returns (6,I_H,I_W) float32."""
    V = uv.shape[0]
    out = np.zeros((V, I_H * I_W), dtype=np.float32)
    for i in range(V):
        m = mask[i]
        r = (uv[i, m, 1] / ori_H * I_H).long()
        c = (uv[i, m, 0] / ori_W * I_W).long()
        lin = (r * I_W + c).numpy()
        out[i][lin] = depth[i, m].numpy()        # numpy: the last of repeated indices is kept
    return torch.from_numpy(out).view(V, I_H, I_W)


def complete_depth(depth_map):
    """encoder_utils.py:175-182: per-view CPU `fill_in_multiscale`."""
    out = [torch.from_numpy(depth_completion.fill_in_multiscale(d.numpy())) for d in depth_map]
    return torch.stack(out)


def bev_sample_grid(depth_map, img2lidar, img_meta, ori_H, ori_W):
    """encoder_utils.py:183-194: un-project every feature pixel through its completed
    depth; returns grid (6,I_H,I_W,2) in [-1,1] (x->BEV width, y->BEV height) and lift_mask."""
    V, I_H, I_W = depth_map.shape
    xs = torch.linspace(0, ori_W - 1, I_W, dtype=torch.float32).view(1, 1, I_W).expand(V, I_H, I_W)
    ys = torch.linspace(0, ori_H - 1, I_H, dtype=torch.float32).view(1, I_H, 1).expand(V, I_H, I_W)
    xyd = torch.stack((xs * depth_map, ys * depth_map, depth_map, torch.ones_like(depth_map)), -1)
    xyz = img2lidar.view(V, 1, 1, 4, 4).matmul(xyd.unsqueeze(-1)).squeeze(-1)[..., :3]
    xyz = apply_3d_transformation(xyz.reshape(-1, 3), 'LIDAR', img_meta, reverse=False).view(V, I_H, I_W, 3)
    r = xyz.new_tensor(PC_RANGE)
    lift = ((xyz[..., 0] > r[0]) & (xyz[..., 1] > r[1]) & (xyz[..., 2] > r[2]) &
            (xyz[..., 0] < r[3]) & (xyz[..., 1] < r[4]) & (xyz[..., 2] < r[5]))
    grid = ((xyz[..., 0:2] - r[0:2]) / (r[3:5] - r[0:2]) - 0.5) * 2
    return grid, lift


class BEVWarp(nn.Module):
    """encoder_utils.py:137-199.  `pts_metas['dense_depth']` (B,6,I_H,I_W), when present,
    replaces scatter+completion so kernel parity does not hinge on OpenCV exactness."""

    def sparse_depth(self, pts, img_meta, proj, I_H, I_W):
        ori_H, ori_W = img_meta['input_shape']
        p3 = apply_3d_transformation(pts[..., :3], 'LIDAR', img_meta, reverse=True)
        depth, uv, _, mask = project_to_views(p3, proj, ori_H, ori_W)
        return scatter_depth(uv, depth, mask, ori_H, ori_W, I_H, I_W)

    def forward(self, lidar_feats, img_feats, img_metas, pts_metas):
        B, V, _, I_H, I_W = img_feats.shape
        lidar2img = lidar2img_tensor(img_metas, img_feats)
        img2lidar = torch.inverse(lidar2img)                                     # :149
        out = []
        for b in range(B):
            ori_H, ori_W = img_metas[b]['input_shape']
            if 'dense_depth' in pts_metas:
                dense = pts_metas['dense_depth'][b]
            else:
                sparse = self.sparse_depth(pts_metas['pts'][b], img_metas[b], lidar2img[b], I_H, I_W)
                dense = complete_depth(sparse)
            grid, lift = bev_sample_grid(dense, img2lidar[b], img_metas[b], ori_H, ori_W)
            bev = lidar_feats[b].unsqueeze(0).expand(V, -1, -1, -1)
            warped = F.grid_sample(bev, grid, mode='bilinear', padding_mode='zeros',
                                   align_corners=False)                           # :195
            warped = warped * lift.unsqueeze(1).to(warped.dtype)                 # :196
            out.append(warped)
        return torch.stack(out, 0)                                               # (B,6,C,H,W)


class MMRI_P2I(nn.Module):
    """encoder_utils.py:202-213."""

    def __init__(self, in_channels, out_channels, kernel_size, locatt=TorchLocatt):
        super().__init__()
        self.Warp = BEVWarp()
        self.Local = LocalContextAttentionBlock(in_channels, out_channels, kernel_size, locatt)

    def forward(self, lidar_feats, img_feats, img_metas, pts_metas):
        warped = self.Warp(lidar_feats, img_feats, img_metas, pts_metas)
        B, N, C, H, W = warped.shape
        return self.Local(img_feats.reshape(B * N, C, H, W), warped.reshape(B * N, C, H, W)).view(B, N, C, H, W)


class MMRI_I2P(nn.Module):
    """encoder_utils.py:216-320.  `group_attn` (:226-255) is restated as the plain
    masked single-head attention it equals (bucketing + compaction only changes
    the padding, never the math)."""

    def __init__(self, pts_channels, img_channels, dropout, chunk=2048):
        super().__init__()
        self.pts_channels, self.img_channels = pts_channels, img_channels
        self.learnedAlign = nn.MultiheadAttention(pts_channels, 1, dropout=dropout, kdim=img_channels,
                                                  vdim=img_channels, batch_first=True)
        self.chunk = chunk

    def keys_for(self, voxel, num_points, img_feat_b, img_meta, proj):
        """:279-310 for a slice of pillars: sampled (P, T*6, C) and mask (P, T*6);
        slot = point*6 + cam."""
        P, T, _ = voxel.shape
        V = proj.shape[0]
        pts = apply_3d_transformation(voxel.reshape(P * T, -1)[:, :3], 'LIDAR', img_meta, reverse=True)
        ori_H, ori_W = img_meta['input_shape']
        _, _, norm, mask = project_to_views(pts, proj, ori_H, ori_W)             # (6,n,2),(6,n)
        sampled = F.grid_sample(img_feat_b, norm.unsqueeze(-2), mode='bilinear', padding_mode='zeros',
                                align_corners=False).squeeze(-1).permute(2, 0, 1)  # (n,6,C)
        sampled = sampled.reshape(P, T * V, self.img_channels)
        mask = mask.permute(1, 0).reshape(P, T, V)
        slot_ok = torch.arange(T).view(1, T) < num_points.view(P, 1).long()        # :303-307
        mask = (mask & slot_ok.unsqueeze(-1)).reshape(P, T * V)
        return sampled, mask

    def forward(self, lidar_feat, img_feat, img_metas, pts_metas):
        B = len(img_metas)
        out = torch.zeros_like(lidar_feat)                                        # :259
        lidar2img = lidar2img_tensor(img_metas, lidar_feat)
        coors_all = pts_metas['pillar_coors']
        start = 0
        for b in range(B):
            end = start + int((coors_all[:, 0] == b).sum())                       # :266-271
            voxel = pts_metas['pillars'][start:end]
            coor = coors_all[start:end].long()
            npts = pts_metas['pillars_num_points'][start:end]
            P = voxel.shape[0]
            Q = lidar_feat[b][:, coor[:, 2], coor[:, 3]].t().unsqueeze(1)         # (P,1,C) :313
            attn = lidar_feat.new_zeros(P, 1, self.pts_channels)
            for s in range(0, P, self.chunk):
                e = min(P, s + self.chunk)
                K, mask = self.keys_for(voxel[s:e], npts[s:e], img_feat[b], img_metas[b], lidar2img[b])
                valid = mask.any(1)                                               # :314
                if valid.any():
                    o = self.learnedAlign(Q[s:e][valid], K[valid], K[valid],
                                          attn_mask=(~mask[valid]).unsqueeze(1))[0]
                    attn[s:e][valid] = o                                          # :316
            out[b][:, coor[:, 2], coor[:, 3]] = attn.squeeze(1).t()               # :318
            start = end
        return out


class DeepInteractionEncoderLayer(nn.Module):
    """necks/deepinteraction_encoder.py:8-33."""

    def __init__(self, hidden_channel, locatt=TorchLocatt):
        super().__init__()
        h = hidden_channel
        self.I2P_block = MMRI_I2P(h, h, 0.1)
        self.P_IML = LocalContextAttentionBlock(h, h, 9, locatt)
        self.P_out_proj = ConvBNReLU(2 * h, h, 1, act=False)
        self.P_integration = ConvBNReLU(2 * h, h, 1, act=False)
        self.P2I_block = MMRI_P2I(h, h, 9, locatt)
        self.I_IML = LocalContextAttentionBlock(h, h, 9, locatt)
        self.I_out_proj = ConvBNReLU(2 * h, h, 1, act=False)
        self.I_integration = ConvBNReLU(2 * h, h, 1, act=False)

    def forward(self, img_feat, lidar_feat, img_metas, pts_metas):
        B = lidar_feat.shape[0]
        BN, C, H, W = img_feat.shape
        img5 = img_feat.view(B, -1, C, H, W)
        i2p = self.I2P_block(lidar_feat, img5, img_metas, pts_metas)
        p2p = self.P_IML(lidar_feat, lidar_feat)
        p_aug = self.P_out_proj(torch.cat((i2p, p2p), 1))
        new_lidar = self.P_integration(torch.cat((p_aug, lidar_feat), 1))
        p2i = self.P2I_block(lidar_feat, img5, img_metas, pts_metas)
        i2i = self.I_IML(img_feat, img_feat)
        i_aug = self.I_out_proj(torch.cat((p2i.view(BN, -1, H, W), i2i), 1))
        new_img = self.I_integration(torch.cat((i_aug, img_feat), 1))
        return new_img, new_lidar


class DeepInteractionEncoder(nn.Module):
    """necks/deepinteraction_encoder.py:35-85 (bias='auto' is a truthy string -> bias=True, :53,:61)."""

    def __init__(self, num_layers=2, in_channels_img=64, in_channels_pts=128 * 3, hidden_channel=128,
                 bn_momentum=0.1, bias='auto', locatt=TorchLocatt):
        super().__init__()
        self.shared_conv_pts = nn.Conv2d(in_channels_pts, hidden_channel, 3, padding=1, bias=bool(bias))
        self.shared_conv_img = nn.Conv2d(in_channels_img, hidden_channel, 3, padding=1, bias=bool(bias))
        self.num_layers = num_layers
        self.fusion_blocks = nn.ModuleList(
            [DeepInteractionEncoderLayer(hidden_channel, locatt) for _ in range(num_layers)])
        for m in self.modules():                                                  # :74-77
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = bn_momentum

    def forward(self, img_feats, pts_feats, img_metas, pts_metas):
        new_img = self.shared_conv_img(img_feats)
        new_pts = self.shared_conv_pts(pts_feats)
        pts_conv = new_pts.clone()
        for blk in self.fusion_blocks:
            new_img, new_pts = blk(new_img, new_pts, img_metas, pts_metas)
        return new_img, [pts_conv, new_pts]
