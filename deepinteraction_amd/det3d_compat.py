"""Restatements of the mmdet / mmdet3d (0.17.1) pieces the head's `loss`, `get_targets` and `get_bboxes`
call into (reference dense_heads/deepinteraction_decoder.py:315-638, core/bbox/assigners/hungarian_assigner.py).
None of these packages is under /root/reference or installed here, so they are restated from their published
algorithms - PARITY UNPINNED at this boundary and unit-tested from first principles
(tests/test_targets_loss.py).  Plain torch ops, any device.

    LiDARBoxes                 minimal `LiDARInstance3DBoxes` surface: .tensor, .gravity_center, .corners, len()
    boxes_iou3d_lidar          `BboxOverlaps3D(coordinate='lidar')`: rotated BEV intersection x height overlap
    FocalLossCost, ClassificationCost   mmdet match costs
    AssignResult, pseudo_sample  mmdet AssignResult + PseudoSampler.sample
    gaussian_radius, draw_heatmap_gaussian   mmdet3d.core.utils.gaussian (CenterPoint)
    clip_sigmoid               mmdet3d.models.utils
    FocalLoss, L1Loss, GaussianFocalLoss     mmdet losses with weight / avg_factor reduction
    circle_nms                 mmdet3d.core.post_processing (CenterPoint circle NMS)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- boxes
class LiDARBoxes:
    """(N, 7+) [x, y, z_bottom, dx, dy, dz, yaw, (vx, vy)] - mmdet3d 0.17.1 LiDARInstance3DBoxes, origin
    (0.5, 0.5, 0)."""

    def __init__(self, tensor, box_dim=None):
        t = torch.as_tensor(tensor, dtype=torch.float32)
        self.tensor = t.reshape(-1, box_dim or (t.shape[-1] if t.numel() else 7))
        self.box_dim = self.tensor.shape[-1]

    def __len__(self):
        return self.tensor.shape[0]

    def to(self, device):
        return LiDARBoxes(self.tensor.to(device), self.box_dim)

    @property
    def gravity_center(self):
        t = self.tensor
        return torch.cat([t[:, :2], t[:, 2:3] + t[:, 5:6] * 0.5], 1)

    @property
    def corners(self):
        return lidar_corners(self.tensor[:, :7])


def lidar_corners(boxes):
    """(N,7) -> (N,8,3): unit-cube vertices in the order [0,1,3,2,4,5,7,6] minus (0.5,0.5,0), scaled, rotated
    about z with `p @ [[c,-s,0],[s,c,0],[0,0,1]]`, translated."""
    dims = boxes[:, 3:6]
    idx = torch.tensor([[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0], [1, 0, 0], [1, 0, 1], [1, 1, 1], [1, 1, 0]],
                       dtype=dims.dtype, device=dims.device)
    norm = idx - dims.new_tensor([0.5, 0.5, 0.0])
    c = dims.view(-1, 1, 3) * norm.view(1, 8, 3)
    s, co = torch.sin(boxes[:, 6]), torch.cos(boxes[:, 6])
    x = c[..., 0] * co[:, None] + c[..., 1] * s[:, None]
    y = -c[..., 0] * s[:, None] + c[..., 1] * co[:, None]
    return torch.stack([x, y, c[..., 2]], -1) + boxes[:, None, :3]


def _rect_corners_bev(b):
    """(N,5) [x, y, dx, dy, yaw] -> (N,4,2), vertices in counter-clockwise order.  The yaw convention is the
    one of mmdet3d 0.17.1's iou3d kernel (`rotate_around_center`: x' = x cos + y sin, y' = -x sin + y cos), of
    `lidar_corners` above and of the `query_geometry` HIP kernel - a positive yaw turns the box CLOCKWISE in the
    BEV plane.  (A rotation keeps the orientation of the vertex loop, so the clipper's "inside = left" holds.)"""
    hx, hy = b[:, 2] * 0.5, b[:, 3] * 0.5
    lx = torch.stack([-hx, hx, hx, -hx], 1)
    ly = torch.stack([-hy, -hy, hy, hy], 1)
    c, s = torch.cos(b[:, 4])[:, None], torch.sin(b[:, 4])[:, None]
    return torch.stack([b[:, 0:1] + lx * c + ly * s, b[:, 1:2] - lx * s + ly * c], -1)


def rotated_intersection_area(a, b):
    """Area of the intersection of rotated rectangles, pairwise-aligned: a, b (N,5) -> (N,).
    Sutherland-Hodgman: polygon A (<= 8 vertices) clipped by the four half-planes of B, vectorised over N."""
    N = a.shape[0]
    if N == 0:
        return a.new_zeros(0)
    P = _rect_corners_bev(a.double())                      # subject polygon (N,4,2), ccw
    Q = _rect_corners_bev(b.double())
    M = 8
    poly = torch.cat([P, P[:, :1].expand(N, M - 4, 2)], 1)  # padded with copies of vertex 0
    cnt = torch.full((N,), 4, dtype=torch.long, device=a.device)
    ar = torch.arange(M, device=a.device)
    for e in range(4):
        A, B = Q[:, e], Q[:, (e + 1) % 4]                   # clip edge A->B, inside = left
        ed = B - A
        nxt = torch.where((ar[None] + 1) < cnt[:, None], ar[None] + 1, torch.zeros_like(ar[None]))
        S, E = poly, torch.gather(poly, 1, nxt[..., None].expand(N, M, 2))
        side = lambda X: ed[:, None, 0] * (X[..., 1] - A[:, None, 1]) - ed[:, None, 1] * (X[..., 0] - A[:, None, 0])
        ds, de = side(S), side(E)
        s_in, e_in = ds >= 0, de >= 0
        valid = ar[None] < cnt[:, None]
        t = ds / torch.where((ds - de).abs() > 1e-300, ds - de, torch.ones_like(ds))
        I = S + t[..., None] * (E - S)
        emit_i = valid & (s_in != e_in)                     # the edge crosses the clip line
        emit_e = valid & e_in
        pts = torch.stack([I, E], 2).reshape(N, 2 * M, 2)   # per edge: [intersection, end vertex]
        keep = torch.stack([emit_i, emit_e], 2).reshape(N, 2 * M)
        order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)[:, :M]
        cnt = keep.sum(1).clamp(max=M)
        poly = torch.gather(pts, 1, order[..., None].expand(N, M, 2))
        first = poly[:, :1].expand(N, M, 2)
        poly = torch.where((ar[None] < cnt[:, None])[..., None], poly, first)
    x, y = poly[..., 0], poly[..., 1]
    xn, yn = torch.roll(x, -1, 1), torch.roll(y, -1, 1)
    area = 0.5 * (x * yn - xn * y).sum(1).abs()
    return torch.where(cnt >= 3, area, torch.zeros_like(area)).to(a.dtype)


def boxes_iou3d_lidar(b1, b2):
    """`BboxOverlaps3D(coordinate='lidar')(b1, b2)`, mode 'iou': (N,7+), (M,7+) -> (N,M).
    overlap = rotated BEV intersection x height overlap; iou = overlap / max(v1 + v2 - overlap, 1e-8)."""
    N, M = b1.shape[0], b2.shape[0]
    if N == 0 or M == 0:
        return b1.new_zeros(N, M)
    i, j = torch.meshgrid(torch.arange(N, device=b1.device), torch.arange(M, device=b1.device), indexing='ij')
    a, b = b1[i.reshape(-1)], b2[j.reshape(-1)]
    bev = rotated_intersection_area(a[:, [0, 1, 3, 4, 6]], b[:, [0, 1, 3, 4, 6]])
    top = torch.minimum(a[:, 2] + a[:, 5], b[:, 2] + b[:, 5])
    bot = torch.maximum(a[:, 2], b[:, 2])
    ov = bev * (top - bot).clamp(min=0)
    va, vb = a[:, 3] * a[:, 4] * a[:, 5], b[:, 3] * b[:, 4] * b[:, 5]
    return (ov / (va + vb - ov).clamp(min=1e-8)).view(N, M)


# --------------------------------------------------------------------------- assignment plumbing
class FocalLossCost:
    """mmdet.core.bbox.match_costs.FocalLossCost: cls_pred (Q,C) logits, gt_labels (G,) -> (Q,G)."""

    def __init__(self, weight=1.0, alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


class ClassificationCost:
    """mmdet.core.bbox.match_costs.ClassificationCost: -softmax(cls_pred)[:, gt_labels] * weight."""

    def __init__(self, weight=1.0):
        self.weight = weight

    def __call__(self, cls_pred, gt_labels):
        return -cls_pred.softmax(-1)[:, gt_labels] * self.weight


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class SamplingResult:
    pass


def pseudo_sample(assign_result, bboxes, gt_bboxes):
    """mmdet PseudoSampler.sample: positives = gt_inds > 0, negatives = gt_inds == 0, no sub-sampling."""
    r = SamplingResult()
    rows = getattr(assign_result, 'host_rows', None)
    if rows is not None:
        # the assigner kept its matches on the host: the (sorted) index sets without a device round trip
        pos = np.sort(rows)
        mask = np.ones(assign_result.gt_inds.shape[0], dtype=bool)
        mask[pos] = False
        dev = assign_result.gt_inds.device
        r.pos_inds = torch.from_numpy(pos).to(dev)
        r.neg_inds = torch.from_numpy(np.nonzero(mask)[0]).to(dev)
    else:
        r.pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        r.neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
    r.pos_assigned_gt_inds = assign_result.gt_inds[r.pos_inds] - 1
    if gt_bboxes.numel() == 0:
        r.pos_gt_bboxes = gt_bboxes.view(-1, gt_bboxes.shape[-1] if gt_bboxes.dim() > 1 else 7)[:0]
    else:
        r.pos_gt_bboxes = gt_bboxes[r.pos_assigned_gt_inds.long(), :]
    return r


# --------------------------------------------------------------------------- heat-map targets
def gaussian_radius(det_size, min_overlap=0.5):
    """mmdet3d.core.utils.gaussian_radius (CornerNet / CenterPoint)."""
    height, width = det_size
    a1, b1 = 1, height + width
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + torch.sqrt(b1 ** 2 - 4 * a1 * c1)) / 2
    a2, b2 = 4, 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + torch.sqrt(b2 ** 2 - 4 * a2 * c2)) / 2
    a3, b3 = 4 * min_overlap, -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + torch.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def _gaussian_2d(shape, sigma=1.0):
    m, n = [(ss - 1.0) / 2.0 for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_heatmap_gaussian(heatmap, center, radius, k=1):
    """mmdet3d.core.utils.draw_heatmap_gaussian: element-wise max of `heatmap` (H,W) with a Gaussian of
    sigma = diameter / 6 around integer `center` (x, y), in place."""
    diameter = 2 * radius + 1
    g = _gaussian_2d((diameter, diameter), sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    masked_heatmap = heatmap[y - top:y + bottom, x - left:x + right]
    masked_gaussian = torch.from_numpy(g[radius - top:radius + bottom, radius - left:radius + right]).to(
        heatmap.device, torch.float32)
    if min(masked_gaussian.shape) > 0 and min(masked_heatmap.shape) > 0:
        torch.max(masked_heatmap, masked_gaussian * k, out=masked_heatmap)
    return heatmap


def draw_heatmap_gaussian_host(heatmap, center, radius, k=1):
    """`draw_heatmap_gaussian` on a float32 numpy map (the host-side target builder: single-threaded, no OpenMP team to
    wake per box); the same float64 Gaussian rounded to float32 and the same element-wise max."""
    diameter = 2 * radius + 1
    g = _gaussian_2d((diameter, diameter), sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    masked_heatmap = heatmap[y - top:y + bottom, x - left:x + right]
    masked_gaussian = g[radius - top:radius + bottom, radius - left:radius + right].astype(np.float32)
    if min(masked_gaussian.shape) > 0 and min(masked_heatmap.shape) > 0:
        np.maximum(masked_heatmap, masked_gaussian * np.float32(k), out=masked_heatmap)
    return heatmap


def clip_sigmoid(x, eps=1e-4):
    return torch.clamp(x.sigmoid(), min=eps, max=1 - eps)


# --------------------------------------------------------------------------- losses (mmdet semantics)
def _reduce(loss, weight, reduction, avg_factor):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return {'mean': loss.mean, 'sum': loss.sum, 'none': lambda: loss}[reduction]()
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


class FocalLoss(torch.nn.Module):
    """mmdet FocalLoss(use_sigmoid=True): pred (N,C) logits, target (N,) in [0,C] (C = background)."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        C = pred.shape[1]
        t = F.one_hot(target.clamp(max=C), C + 1)[:, :C].to(pred.dtype)
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        fw = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * fw
        if weight is not None:
            weight = weight.to(pred.dtype).view(-1, 1)
        return self.loss_weight * _reduce(loss, weight, reduction_override or self.reduction, avg_factor)


class L1Loss(torch.nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        return self.loss_weight * _reduce((pred - target).abs(), weight, reduction_override or self.reduction, avg_factor)


class GaussianFocalLoss(torch.nn.Module):
    """mmdet GaussianFocalLoss: pred in (0,1), Gaussian target; alpha 2, gamma 4."""

    def __init__(self, alpha=2.0, gamma=4.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.alpha, self.gamma, self.reduction, self.loss_weight = alpha, gamma, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        eps = 1e-12
        pos_w = target.eq(1)
        neg_w = (1 - target).pow(self.gamma)
        pos = -(pred + eps).log() * (1 - pred).pow(self.alpha) * pos_w
        neg = -(1 - pred + eps).log() * pred.pow(self.alpha) * neg_w
        return self.loss_weight * _reduce(pos + neg, weight, self.reduction, avg_factor)


LOSSES = dict(FocalLoss=FocalLoss, L1Loss=L1Loss, GaussianFocalLoss=GaussianFocalLoss)


def build_loss(cfg):
    cfg = dict(cfg)
    return LOSSES[cfg.pop('type')](**cfg)


# --------------------------------------------------------------------------- post-processing
def circle_nms(dets, thresh, post_max_size=83):
    """mmdet3d.core.post_processing.circle_nms: dets (n,3) [x, y, score] numpy -> kept indices (score order)."""
    x1, y1, scores = dets[:, 0], dets[:, 1], dets[:, 2]
    order = scores.argsort()[::-1].astype(np.int32)
    ndets = dets.shape[0]
    suppressed = np.zeros(ndets, dtype=np.int32)
    keep = []
    for _i in range(ndets):
        i = order[_i]
        if suppressed[i] == 1:
            continue
        keep.append(i)
        for _j in range(_i + 1, ndets):
            j = order[_j]
            if suppressed[j] == 1:
                continue
            if (x1[i] - x1[j]) ** 2 + (y1[i] - y1[j]) ** 2 <= thresh:
                suppressed[j] = 1
    return keep[:post_max_size]


def xywhr2xyxyr(boxes_xywhr):
    """mmdet3d.core.xywhr2xyxyr: (n,5) [x, y, w, h, r] -> [x1, y1, x2, y2, r]."""
    b = torch.zeros_like(boxes_xywhr)
    half_w, half_h = boxes_xywhr[:, 2] / 2, boxes_xywhr[:, 3] / 2
    b[:, 0], b[:, 1] = boxes_xywhr[:, 0] - half_w, boxes_xywhr[:, 1] - half_h
    b[:, 2], b[:, 3] = boxes_xywhr[:, 0] + half_w, boxes_xywhr[:, 1] + half_h
    b[:, 4] = boxes_xywhr[:, 4]
    return b


def boxes_iou_bev_xyxyr(a, b):
    """Pairwise rotated BEV IoU of [x1, y1, x2, y2, r] boxes (mmdet3d iou3d `boxes_iou_bev`): (N,5),(M,5) -> (N,M)."""
    N, M = a.shape[0], b.shape[0]
    if N == 0 or M == 0:
        return a.new_zeros(N, M)
    to_c = lambda t: torch.stack([(t[:, 0] + t[:, 2]) / 2, (t[:, 1] + t[:, 3]) / 2, t[:, 2] - t[:, 0], t[:, 3] - t[:, 1],
                                  t[:, 4]], 1)
    ca, cb = to_c(a), to_c(b)
    i, j = torch.meshgrid(torch.arange(N, device=a.device), torch.arange(M, device=a.device), indexing='ij')
    inter = rotated_intersection_area(ca[i.reshape(-1)], cb[j.reshape(-1)]).view(N, M)
    sa, sb = (ca[:, 2] * ca[:, 3])[:, None], (cb[:, 2] * cb[:, 3])[None]
    return inter / (sa + sb - inter).clamp(min=1e-8)


def nms_rotated_bev(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """mmdet3d 0.17.1 `nms_gpu(boxes [x1,y1,x2,y2,r], scores, thresh, pre_maxsize, post_max_size)`: greedy suppression
    in score order of boxes whose rotated BEV IoU with a kept box exceeds `thresh`; returns kept indices (int64)."""
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = boxes[order].contiguous()
    n = b.shape[0]
    if n == 0:
        return order
    iou = boxes_iou_bev_xyxyr(b, b)
    over = (iou > thresh).cpu()
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        suppressed |= over[i]
    keep = order[torch.tensor(keep, dtype=torch.long, device=order.device)]
    if post_max_size is not None:
        keep = keep[:post_max_size]
    return keep
