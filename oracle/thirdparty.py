"""Oracle restatements of third-party semantics (TEST INFRASTRUCTURE).

Sources of these functions are NOT under /root/reference; they are restated
from the published algorithms of the versions the reference pins
(install.md): mmdet3d 0.17.1, detectron2 (ROIAlign aligned=True), OpenCV.
PARITY UNPINNED at these boundaries (the reference holds no tests/vectors).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from scipy import ndimage


# --------------------------------------------------------------------------
# mmdet3d 0.17.1  models/fusion_layers/coord_transform.py
# called at encoder_utils.py:156,189,280 and decoder_utils.py:692
# --------------------------------------------------------------------------
def apply_3d_transformation(pcd, coord_type, img_meta, reverse=False):
    """Replay (or invert) the recorded 3-D augmentation flow on (n,3) points.

    LiDAR conventions of mmdet3d 0.17.1: rotate = `p @ rot`, horizontal flip
    negates y, vertical flip negates x.  Identity when the keys are absent.
    """
    assert coord_type == 'LIDAR'
    dtype, device = pcd.dtype, pcd.device
    rot = (torch.as_tensor(np.asarray(img_meta['pcd_rotation']), dtype=dtype, device=device)
           if 'pcd_rotation' in img_meta else torch.eye(3, dtype=dtype, device=device))
    scale = img_meta['pcd_scale_factor'] if 'pcd_scale_factor' in img_meta else 1.0
    trans = (torch.as_tensor(np.asarray(img_meta['pcd_trans']), dtype=dtype, device=device)
             if 'pcd_trans' in img_meta else torch.zeros(3, dtype=dtype, device=device))
    hflip = img_meta.get('pcd_horizontal_flip', False)
    vflip = img_meta.get('pcd_vertical_flip', False)
    flow = list(img_meta.get('transformation_3d_flow', []))

    p = pcd.clone()
    if reverse:
        scale_f, trans_v, rot_m = 1.0 / scale, -trans, rot.inverse()
        flow = flow[::-1]
    else:
        scale_f, trans_v, rot_m = scale, trans, rot
    for op in flow:
        if op == 'T':
            p[:, :3] = p[:, :3] + trans_v
        elif op == 'S':
            p[:, :3] = p[:, :3] * scale_f
        elif op == 'R':
            p[:, :3] = p[:, :3] @ rot_m
        elif op == 'HF':
            if hflip:
                p[:, 1] = -p[:, 1]
        elif op == 'VF':
            if vflip:
                p[:, 0] = -p[:, 0]
        else:
            raise AssertionError(f'This 3D data transformation op ({op}) is not supported')
    return p


def aug_affine(img_meta, reverse):
    """The same flow composed into one affine `p' = p @ A + t` (float64).

    Used by tests to check the composed form the HIP kernels consume."""
    A = np.eye(3)
    t = np.zeros(3)
    rot = np.asarray(img_meta.get('pcd_rotation', np.eye(3)), dtype=np.float64)
    scale = float(img_meta.get('pcd_scale_factor', 1.0))
    trans = np.asarray(img_meta.get('pcd_trans', np.zeros(3)), dtype=np.float64)
    hflip = img_meta.get('pcd_horizontal_flip', False)
    vflip = img_meta.get('pcd_vertical_flip', False)
    flow = list(img_meta.get('transformation_3d_flow', []))
    if reverse:
        scale, trans, rot = 1.0 / scale, -trans, np.linalg.inv(rot)
        flow = flow[::-1]
    for op in flow:
        if op == 'T':
            t = t + trans
        elif op == 'S':
            A, t = A * scale, t * scale
        elif op == 'R':
            A, t = A @ rot, t @ rot
        elif op == 'HF' and hflip:
            F = np.diag([1.0, -1.0, 1.0])
            A, t = A @ F, t @ F
        elif op == 'VF' and vflip:
            F = np.diag([-1.0, 1.0, 1.0])
            A, t = A @ F, t @ F
    return A, t


# --------------------------------------------------------------------------
# mmdet3d 0.17.1  core/bbox/structures/lidar_box3d.py  LiDARInstance3DBoxes.corners
# called at decoder_utils.py:690-691,808
# --------------------------------------------------------------------------
def lidar_box_corners(boxes):
    """(Q,7) [x,y,z_bottom,dx,dy,dz,yaw] -> (Q,8,3), pre-v1.0 convention."""
    dims = boxes[:, 3:6]
    idx = np.stack(np.unravel_index(np.arange(8), [2, 2, 2]), axis=1)
    corners_norm = torch.from_numpy(idx).to(device=dims.device, dtype=dims.dtype)
    corners_norm = corners_norm[[0, 1, 3, 2, 4, 5, 7, 6]]
    corners_norm = corners_norm - dims.new_tensor([0.5, 0.5, 0])
    corners = dims.view(-1, 1, 3) * corners_norm.reshape(1, 8, 3)
    ang = boxes[:, 6]
    s, c = torch.sin(ang), torch.cos(ang)
    o, z = torch.ones_like(c), torch.zeros_like(c)
    rot_mat_T = torch.stack([torch.stack([c, -s, z]), torch.stack([s, c, z]), torch.stack([z, z, o])])
    corners = torch.einsum('aij,jka->aik', corners, rot_mat_T)
    corners = corners + boxes[:, :3].view(-1, 1, 3)
    return corners


# --------------------------------------------------------------------------
# detectron2 ROIPooler(output_size=7, scales=[s], sampling_ratio=2, 'ROIAlignV2')
# == ROIAlign(7, s, 2, aligned=True); decoder_utils.py:641-646,739-741,769-774,822-823
# --------------------------------------------------------------------------
def _roi_bilinear(feat, y, x):
    """feat (C,H,W); y,x (...,) float -> (C, ...) with the detectron2 edge rules."""
    C, H, W = feat.shape
    oob = (y < -1.0) | (y > H) | (x < -1.0) | (x > W)
    y = y.clamp(min=0)
    x = x.clamp(min=0)
    y_low = y.floor().long()
    x_low = x.floor().long()
    ycl = y_low >= H - 1
    xcl = x_low >= W - 1
    y_low = torch.where(ycl, torch.full_like(y_low, H - 1), y_low)
    x_low = torch.where(xcl, torch.full_like(x_low, W - 1), x_low)
    y_high = torch.where(ycl, y_low, y_low + 1)
    x_high = torch.where(xcl, x_low, x_low + 1)
    y = torch.where(ycl, y_low.to(y.dtype), y)
    x = torch.where(xcl, x_low.to(x.dtype), x)
    ly = y - y_low.to(y.dtype)
    lx = x - x_low.to(x.dtype)
    hy, hx = 1.0 - ly, 1.0 - lx
    v1 = feat[:, y_low, x_low]
    v2 = feat[:, y_low, x_high]
    v3 = feat[:, y_high, x_low]
    v4 = feat[:, y_high, x_high]
    val = (hy * hx) * v1 + (hy * lx) * v2 + (ly * hx) * v3 + (ly * lx) * v4
    return torch.where(oob, torch.zeros_like(val), val)


def roi_align_v2(feat, boxes, output_size=7, spatial_scale=1.0, sampling_ratio=2):
    """feat (1,C,H,W); boxes (q,4) xyxy in input pixels -> (q,C,7,7)."""
    assert feat.shape[0] == 1 and sampling_ratio > 0
    f = feat[0]
    q = boxes.shape[0]
    P, G = output_size, sampling_ratio
    start_w = boxes[:, 0] * spatial_scale - 0.5
    start_h = boxes[:, 1] * spatial_scale - 0.5
    end_w = boxes[:, 2] * spatial_scale - 0.5
    end_h = boxes[:, 3] * spatial_scale - 0.5
    bin_w = (end_w - start_w) / P
    bin_h = (end_h - start_h) / P
    p = torch.arange(P, dtype=feat.dtype, device=feat.device)
    g = torch.arange(G, dtype=feat.dtype, device=feat.device)
    # (q, P, G)
    ys = start_h[:, None, None] + p[None, :, None] * bin_h[:, None, None] + \
        (g[None, None, :] + 0.5) * bin_h[:, None, None] / G
    xs = start_w[:, None, None] + p[None, :, None] * bin_w[:, None, None] + \
        (g[None, None, :] + 0.5) * bin_w[:, None, None] / G
    # broadcast to (q, Ph, Gy, Pw, Gx)
    Y = ys[:, :, :, None, None].expand(q, P, G, P, G)
    X = xs[:, None, None, :, :].expand(q, P, G, P, G)
    val = _roi_bilinear(f, Y, X)                      # (C,q,P,G,P,G)
    out = val.sum(dim=(3, 5)) / float(G * G)          # (C,q,P,P)
    return out.permute(1, 0, 2, 3).contiguous()


# --------------------------------------------------------------------------
# OpenCV calls of ip_basic/depth_map_utils.py:177-260, on float32 single-channel
# --------------------------------------------------------------------------
def cv_dilate(img, kernel):
    """cv2.dilate, default anchor/border: max over the footprint, border ignored."""
    return ndimage.maximum_filter(img, footprint=kernel.astype(bool), mode='constant',
                                  cval=-np.inf).astype(np.float32)


def cv_erode(img, kernel):
    return ndimage.minimum_filter(img, footprint=kernel.astype(bool), mode='constant',
                                  cval=np.inf).astype(np.float32)


def cv_morph_close(img, kernel):
    """cv2.morphologyEx(MORPH_CLOSE): dilate then erode."""
    return cv_erode(cv_dilate(img, kernel), kernel)


def cv_median_blur5(img):
    """cv2.medianBlur(img, 5) for CV_32F: BORDER_REPLICATE."""
    return ndimage.median_filter(img, size=5, mode='nearest').astype(np.float32)


def cv_bilateral_filter(img, d, sigma_color, sigma_space):
    """cv2.bilateralFilter for CV_32FC1 (bilateralFilter_32f): BORDER_REFLECT_101,
    circular support r<=d/2, colour weight through the 4096-bin linearly
    interpolated exp LUT spanning [min,max] of the image."""
    src = np.asarray(img, dtype=np.float32)
    radius = d // 2
    gcc = np.float64(-0.5 / (sigma_color * sigma_color))
    gsc = np.float64(-0.5 / (sigma_space * sigma_space))
    mn, mx = float(src.min()), float(src.max())
    if abs(mn - mx) < np.finfo(np.float32).eps:
        return src.copy()
    nbins = 1 << 12
    scale_index = np.float32(nbins / (mx - mn))
    lut = np.zeros(nbins + 2, dtype=np.float32)
    last = 1.0
    for i in range(nbins + 2):
        if last > 0.0:
            val = np.float64(i) / np.float64(scale_index)
            lut[i] = np.float32(math.exp(val * val * gcc))
            last = lut[i]
        else:
            lut[i] = 0.0
    temp = np.pad(src, radius, mode='reflect')
    H, W = src.shape
    s = np.zeros((H, W), dtype=np.float32)
    ws = np.zeros((H, W), dtype=np.float32)
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            r = math.sqrt(i * i + j * j)
            if r > max(radius, 1):
                continue
            sw = np.float32(math.exp(r * r * gsc))
            val = temp[radius + i:radius + i + H, radius + j:radius + j + W]
            alpha = np.abs(val - src) * scale_index
            idx = np.floor(alpha).astype(np.int64)
            alpha = (alpha - idx.astype(np.float32)).astype(np.float32)
            idx = np.clip(idx, 0, nbins)
            w = sw * (lut[idx] + alpha * (lut[idx + 1] - lut[idx]))
            w = w.astype(np.float32)
            s += val * w
            ws += w
    return (s / ws).astype(np.float32)


# ---------------------------------------------------------------------------
# mmcv-full 1.3.18 transformer bricks used by DeepInteraction++ (mmcv/cnn/bricks/transformer.py,
# mmcv/ops/multi_scale_deform_attn.py).  Sources are not under /root/reference: restated from the
# published algorithm, PARITY UNPINNED at this boundary.
def ms_deform_attn_core(value, spatial_shapes, sampling_locations, attention_weights):
    """value (bs, sum HW, heads, d); spatial_shapes list of (H, W); sampling_locations
    (bs, nq, heads, L, P, 2) in [0,1] (x, y); attention_weights (bs, nq, heads, L, P) -> (bs, nq, heads*d).
    Bilinear, zero padding, align_corners=False (the `grid_sample` formulation mmcv ships as its CPU path)."""
    bs, _, nh, d = value.shape
    _, nq, _, L, P, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in spatial_shapes]
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (v, (h, w)) in enumerate(zip(value.split(sizes, dim=1), spatial_shapes)):
        v = v.flatten(2).transpose(1, 2).reshape(bs * nh, d, int(h), int(w))
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)                     # (bs*nh, nq, P, 2)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros', align_corners=False))
    w_ = attention_weights.transpose(1, 2).reshape(bs * nh, 1, nq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * w_).sum(-1).view(bs, nh * d, nq)
    return out.transpose(1, 2).contiguous()


class MultiScaleDeformableAttention(torch.nn.Module):
    """mmcv 1.3.18 `MultiScaleDeformableAttention` (defaults num_heads=8, num_points=4, dropout=0.1)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64, dropout=0.1,
                 batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__()
        nn = torch.nn
        self.embed_dims, self.num_heads, self.num_levels, self.num_points = embed_dims, num_heads, num_levels, num_points
        self.batch_first = batch_first
        self.dropout = nn.Dropout(dropout)
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        nn = torch.nn
        nn.init.constant_(self.sampling_offsets.weight, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 2).repeat(
            1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, i, :] *= i + 1
        self.sampling_offsets.bias.data = grid.view(-1)
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.)

    def sampling(self, query, reference_points, spatial_shapes):
        """Projected offsets + softmax weights -> (locations (bs,nq,h,L,P,2), weights (bs,nq,h,L,P))."""
        bs, nq, _ = query.shape
        off = self.sampling_offsets(query).view(bs, nq, self.num_heads, self.num_levels, self.num_points, 2)
        w = self.attention_weights(query).view(bs, nq, self.num_heads, self.num_levels * self.num_points)
        w = w.softmax(-1).view(bs, nq, self.num_heads, self.num_levels, self.num_points)
        assert reference_points.shape[-1] == 2
        shapes = torch.as_tensor([[int(h), int(w_)] for h, w_ in spatial_shapes], device=query.device)
        normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1).to(query.dtype)
        loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        return loc, w

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        bs, nv, _ = value.shape
        shapes = [(int(h), int(w)) for h, w in spatial_shapes]
        assert sum(h * w for h, w in shapes) == nv
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, nv, self.num_heads, -1)
        loc, w = self.sampling(query, reference_points, shapes)
        out = self.output_proj(ms_deform_attn_core(value, shapes, loc, w))
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return self.dropout(out) + identity


class TransFFN(torch.nn.Module):
    """mmcv 1.3.18 `FFN`: identity + Dropout(Linear(Dropout(act(Linear(x))))); keys `layers.0.0.*`, `layers.1.*`."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=dict(type='ReLU', inplace=True),
                 ffn_drop=0., dropout_layer=None, add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        nn = torch.nn
        assert num_fcs >= 2 and act_cfg['type'] in ('ReLU', 'GELU')
        self.embed_dims = embed_dims
        act = (lambda: nn.ReLU(inplace=True)) if act_cfg['type'] == 'ReLU' else nn.GELU
        layers, cin = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(cin, feedforward_channels), act(), nn.Dropout(ffn_drop)))
            cin = feedforward_channels
        layers += [nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop)]
        self.layers = nn.Sequential(*layers)
        # `dropout_layer` (DropPath in the Swin blocks) sits on the branch before the identity is added
        self.dropout_layer = build_dropout(dropout_layer) if dropout_layer else nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        return (x if identity is None else identity) + self.dropout_layer(out)


class DropPath(torch.nn.Module):
    """mmcv `DropPath` (stochastic depth): drops whole samples of the branch in training, identity in evaluation."""

    def __init__(self, drop_prob=0.1):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = (keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)).floor()
        return x.div(keep) * mask


def build_dropout(cfg):
    """mmcv `build_dropout`: {'type': 'DropPath' | 'Dropout', ...}."""
    cfg = dict(cfg)
    typ = cfg.pop('type')
    if typ == 'DropPath':
        return DropPath(**cfg)
    assert typ == 'Dropout'
    return torch.nn.Dropout(p=cfg.pop('drop_prob', 0.5), **cfg)
