"""Seeded synthetic nuScenes-shaped inputs for the interaction hot path.

There is no dataset in this environment; these generators produce exactly the
tensors and meta dictionaries the reference's detector hands to the neck/head
(`detectors/deepinteraction.py:103-106,132-140`, `nuscenes_dataset.py:58-67`,
`transform_3d.py:125-134`): FPN/SECOND-like feature maps, a 10-sweep-sized point
cloud, its 0.6 m pillars (<=20 points, first come first kept) and six pinhole
`lidar2img` matrices covering 360 degrees.  numpy/torch on the host, deterministic.
"""
import math

import numpy as np
import torch

PC_RANGE = (-54.0, -54.0, -5.0, 54.0, 54.0, 3.0)
CAM_YAWS_DEG = (0.0, 55.0, -55.0, 110.0, -110.0, 180.0)

SHAPE_R = dict(img_hw=(112, 200), input_shape=(448, 800), bev_hw=(180, 180), c_img=256, c_pts=512,
               n_points=262144, hidden=128)
SHAPE_A = dict(img_hw=(232, 400), input_shape=(928, 1600), bev_hw=(180, 180), c_img=256, c_pts=256,
               n_points=262144, hidden=128)
SHAPE_TINY = dict(img_hw=(16, 28), input_shape=(64, 112), bev_hw=(36, 36), c_img=24, c_pts=40,
                  n_points=6000, hidden=128)


def make_lidar2img(input_shape, raw_hw=(900, 1600), fx=1266.0, fy=1266.0, cx=800.0, cy=450.0,
                   cam_height=1.5):
    """Six 4x4 float64 `lidar2img` matrices: `viewpad @ lidar2cam` (nuscenes_dataset.py:58-67)
    then rescaled to `input_shape` as ScaleImageMultiViewImage does (transform_3d.py:125-134)."""
    return make_cameras(input_shape, raw_hw, fx, fy, cx, cy, cam_height)[0]


def make_cameras(input_shape, raw_hw=(900, 1600), fx=1266.0, fy=1266.0, cx=800.0, cy=450.0, cam_height=1.5):
    """(lidar2img, cam2lidar, cam_intrinsic): six 4x4 float64 each; the last two are what the ++ data pipeline adds
    (nuscenes_dataset.py:70-83): `cam2lidar` = inverse of `lidar2cam`, `cam_intrinsic` = the 4x4 viewpad."""
    H_in, W_in = input_shape
    out, c2l, intr = [], [], []
    for yaw in CAM_YAWS_DEG:
        a = math.radians(yaw)
        fwd = np.array([math.cos(a), math.sin(a), 0.0])
        right = np.array([math.sin(a), -math.cos(a), 0.0])
        down = np.array([0.0, 0.0, -1.0])
        R = np.stack([right, down, fwd])                 # cam <- lidar
        c = np.array([0.0, 0.0, cam_height])
        l2c = np.eye(4)
        l2c[:3, :3] = R
        l2c[:3, 3] = -R @ c
        viewpad = np.eye(4)
        viewpad[:3, :3] = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
        scale = np.eye(4)
        scale[0, 0] = W_in / raw_hw[1]
        scale[1, 1] = H_in / raw_hw[0]
        out.append(scale @ (viewpad @ l2c))
        c2l.append(np.linalg.inv(l2c))
        intr.append(viewpad.copy())
    return out, c2l, intr


def make_points(n, seed):
    """(n,5) float32: xyz, intensity, time lag.  Range 1+53u^2 (denser near ego)."""
    rng = np.random.default_rng(seed)
    u = rng.random(n)
    r = 1.0 + 53.0 * u * u
    az = rng.random(n) * 2.0 * math.pi
    z = rng.random(n) * 4.0 - 3.0
    inten = rng.random(n)
    dt = rng.integers(0, 10, n) * 0.05
    pts = np.stack([r * np.cos(az), r * np.sin(az), z, inten, dt], 1).astype(np.float32)
    return pts


def pillarize(points, bev_hw, max_points=20, max_pillars=60000, pc_range=PC_RANGE):
    """Hard voxelisation into BEV pillars (format of spconv PointToVoxel,
    sparse_voxelize.py:23-30): pillars (P,T,5) zero padded, coors (P,3) int32 [z,y,x],
    num_points (P,) int32.  Points beyond T per pillar are dropped; pillars in
    first-come order."""
    Hb, Wb = bev_hw
    sx = (pc_range[3] - pc_range[0]) / Wb
    sy = (pc_range[4] - pc_range[1]) / Hb
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    ix = np.floor((x - pc_range[0]) / sx).astype(np.int64)
    iy = np.floor((y - pc_range[1]) / sy).astype(np.int64)
    ok = (ix >= 0) & (ix < Wb) & (iy >= 0) & (iy < Hb) & (z >= pc_range[2]) & (z < pc_range[5])
    idx = np.nonzero(ok)[0]
    cell = iy[idx] * Wb + ix[idx]
    order = np.argsort(cell, kind='stable')
    cell_s, idx_s = cell[order], idx[order]
    uniq, first_pos, counts = np.unique(cell_s, return_index=True, return_counts=True)
    first_point = idx_s[first_pos]
    porder = np.argsort(first_point, kind='stable')[:max_pillars]
    P = len(porder)
    pillars = np.zeros((P, max_points, points.shape[1]), dtype=np.float32)
    num = np.minimum(counts[porder], max_points).astype(np.int32)
    for j, p in enumerate(porder):
        s = first_pos[p]
        pillars[j, :num[j]] = points[idx_s[s:s + num[j]]]
    coors = np.stack([np.zeros(P, np.int64), uniq[porder] // Wb, uniq[porder] % Wb], 1).astype(np.int32)
    return pillars, coors, num


def make_inputs(batch=1, shape=SHAPE_R, seed=0, device='cpu', dtype=torch.float32, aug=None):
    """Everything `DeepInteractionEncoder.forward(img_feats, pts_feats, img_metas, pts_metas)` and
    `DeepInteractionDecoder.forward` consume for `batch` samples (sample i uses seed+i)."""
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    img_feats, pts_feats, img_metas, pts_list = [], [], [], []
    pil, coo, num = [], [], []
    for b in range(batch):
        g = torch.Generator().manual_seed(seed + b)
        img_feats.append((torch.randn(6, shape['c_img'], Hi, Wi, generator=g) * 0.5).clamp_(min=0))
        pts_feats.append((torch.randn(1, shape['c_pts'], Hb, Wb, generator=g) * 0.5).clamp_(min=0))
        pts = make_points(shape['n_points'], seed + b)
        meta = dict(lidar2img=make_lidar2img(shape['input_shape']), input_shape=tuple(shape['input_shape']),
                    img_shape=[(shape['input_shape'][0], shape['input_shape'][1], 3)] * 6)
        if aug is not None:
            meta.update(aug)
            from_lidar = _apply_aug_np(pts[:, :3], aug)
            pts = np.concatenate([from_lidar.astype(np.float32), pts[:, 3:]], 1)
        p, c, n = pillarize(pts, (Hb, Wb))
        pil.append(torch.from_numpy(p))
        coo.append(torch.from_numpy(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1)))
        num.append(torch.from_numpy(n))
        pts_list.append(torch.from_numpy(pts).to(device))
        img_metas.append(meta)
    pts_metas = dict(pillars=torch.cat(pil).to(device), pillar_coors=torch.cat(coo).to(device),
                     pillars_num_points=torch.cat(num).to(device), pts=pts_list)
    return dict(img_feats=torch.cat(img_feats).to(device=device, dtype=dtype),
                pts_feats=torch.cat(pts_feats).to(device=device, dtype=dtype),
                img_metas=img_metas, pts_metas=pts_metas)


def _apply_aug_np(xyz, aug):
    """Forward 3-D augmentation flow (what the data pipeline would have applied to the points)."""
    p = xyz.astype(np.float64).copy()
    for op in aug.get('transformation_3d_flow', []):
        if op == 'R':
            p = p @ np.asarray(aug['pcd_rotation'], dtype=np.float64)
        elif op == 'S':
            p = p * aug['pcd_scale_factor']
        elif op == 'T':
            p = p + np.asarray(aug['pcd_trans'], dtype=np.float64)
        elif op == 'HF' and aug.get('pcd_horizontal_flip', False):
            p[:, 1] = -p[:, 1]
        elif op == 'VF' and aug.get('pcd_vertical_flip', False):
            p[:, 0] = -p[:, 0]
    return p


def example_aug(seed=0):
    """A GlobalRotScaleTrans + RandomFlip3D record like the training pipeline writes."""
    rng = np.random.default_rng(seed)
    ang = rng.uniform(-0.785, 0.785)
    c, s = math.cos(ang), math.sin(ang)
    return dict(pcd_rotation=np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]).T.astype(np.float32),
                pcd_scale_factor=float(rng.uniform(0.9, 1.1)),
                pcd_trans=rng.normal(0, 0.5, 3).astype(np.float32),
                pcd_horizontal_flip=True, pcd_vertical_flip=False,
                transformation_3d_flow=['R', 'S', 'T', 'HF'])


SHAPE_PP = dict(SHAPE_R, c_img=256, c_pts=256)
SHAPE_PP_TINY = dict(SHAPE_TINY, img_hw=(16, 28), c_img=24, c_pts=40)


def make_inputs_pp(batch=1, shape=SHAPE_PP, seed=0, device='cpu', dtype=torch.float32, aug=None):
    """Inputs of the DeepInteraction++ neck `FusionTransformerv4.forward(img_feats, pts_feats, img_metas, pts_metas)`
    (config 5): `img_feats` = 2 image levels (stride 4 and 8), `pts_feats` = [the two BEV maps stacked on the channel
    axis, BEV level 0, BEV level 1] (all at the BEV resolution); metas carry `cam2lidar` and `cam_intrinsic` too."""
    base = make_inputs(batch, shape, seed, device, dtype, aug)
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    img_feats, bev0, bev1 = [base['img_feats']], [], []
    lvl1 = []
    for b in range(batch):
        g = torch.Generator().manual_seed(10_000 + seed + b)
        lvl1.append((torch.randn(6, shape['c_img'], Hi // 2, Wi // 2, generator=g) * 0.5).clamp_(min=0))
        bev0.append((torch.randn(1, shape['c_pts'], Hb, Wb, generator=g) * 0.5).clamp_(min=0))
        bev1.append((torch.randn(1, shape['c_pts'], Hb, Wb, generator=g) * 0.5).clamp_(min=0))
    img_feats.append(torch.cat(lvl1).to(device=device, dtype=dtype))
    bev0, bev1 = torch.cat(bev0).to(device=device, dtype=dtype), torch.cat(bev1).to(device=device, dtype=dtype)
    _, c2l, intr = make_cameras(shape['input_shape'])
    for meta in base['img_metas']:
        meta['cam2lidar'], meta['cam_intrinsic'] = c2l, intr
    return dict(img_feats=img_feats, pts_feats=[torch.cat([bev0, bev1], 1), bev0, bev1],
                img_metas=base['img_metas'], pts_metas=base['pts_metas'])
