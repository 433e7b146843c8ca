"""Target assignment of the MMPI head - drop-in for reference `core/bbox/assigners/hungarian_assigner.py`
(`BBox3DL1Cost` :14-21, `BBoxBEVL1Cost` :24-36, `IoU3DCost` :39-46, `HeuristicAssigner3D` :49-92,
`HungarianAssigner3D` :95-153): same names, constructor kwargs and `assign` signatures.  The mmdet
pieces they build on (FocalLossCost, BboxOverlaps3D, AssignResult) are the restatements of
deepinteraction_amd/det3d_compat.py.  As in the reference the cost matrix is formed on the device and the
Hungarian matching itself runs on the host (scipy `linear_sum_assignment`, one D2H copy per layer)."""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from deepinteraction_amd.det3d_compat import AssignResult, ClassificationCost, FocalLossCost, boxes_iou3d_lidar


class BBox3DL1Cost:
    def __init__(self, weight):
        self.weight = weight

    def __call__(self, bboxes, gt_bboxes, train_cfg):
        return torch.cdist(bboxes, gt_bboxes, p=1) * self.weight


class BBoxBEVL1Cost:
    def __init__(self, weight):
        self.weight = weight

    def __call__(self, bboxes, gt_bboxes, train_cfg):
        pcr = train_cfg['point_cloud_range']
        pc_start = bboxes.new_tensor(pcr[0:2])
        pc_range = bboxes.new_tensor(pcr[3:5]) - pc_start
        a = (bboxes[:, :2] - pc_start) / pc_range            # box centres normalised to [0, 1]
        b = (gt_bboxes[:, :2] - pc_start) / pc_range
        return torch.cdist(a, b, p=1) * self.weight


class IoU3DCost:
    def __init__(self, weight):
        self.weight = weight

    def __call__(self, iou):
        return -iou * self.weight


class _BboxOverlaps3D:
    def __init__(self, coordinate='lidar', **_):
        assert coordinate == 'lidar'

    def __call__(self, b1, b2):
        if b1.is_cuda:                       # one launch (csrc/iou3d.hip); host tensors: the torch statement of the same clipper
            from deepinteraction_amd import ops
            return ops.iou3d_lidar(b1, b2)
        return boxes_iou3d_lidar(b1[:, :7], b2[:, :7])


_COSTS = dict(FocalLossCost=FocalLossCost, ClassificationCost=ClassificationCost, BBox3DL1Cost=BBox3DL1Cost, BBoxBEVL1Cost=BBoxBEVL1Cost,
              IoU3DCost=IoU3DCost)


def _build(cfg, table):
    cfg = dict(cfg)
    return table[cfg.pop('type')](**cfg)


class HeuristicAssigner3D:
    """Reference :49-92: every GT takes its nearest proposal (same class when query labels are given)."""

    def __init__(self, dist_thre=100, iou_calculator=dict(type='BboxOverlaps3D')):
        self.dist_thre = dist_thre
        self.iou_calculator = _build(iou_calculator, dict(BboxOverlaps3D=_BboxOverlaps3D))

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None, query_labels=None):
        dist_thre = self.dist_thre
        num_gts, num_bboxes = len(gt_bboxes), len(bboxes)
        bev_dist = torch.norm(bboxes[:, 0:2][None, :, :] - gt_bboxes[:, 0:2][:, None, :], dim=-1)
        if query_labels is not None:
            bev_dist = bev_dist + (query_labels[None] != gt_labels[:, None]) * dist_thre
        _, nearest_indices = bev_dist.min(1)
        assigned_gt_inds = torch.zeros(num_bboxes, device=bboxes.device, dtype=bboxes.dtype)
        assigned_gt_vals = torch.full((num_bboxes,), 10000.0, device=bboxes.device, dtype=bboxes.dtype)
        assigned_gt_labels = torch.full((num_bboxes,), -1.0, device=bboxes.device, dtype=bboxes.dtype)
        for idx_gts in range(num_gts):
            idx_pred = nearest_indices[idx_gts]
            if bev_dist[idx_gts, idx_pred] <= dist_thre and bev_dist[idx_gts, idx_pred] < assigned_gt_vals[idx_pred]:
                assigned_gt_vals[idx_pred] = bev_dist[idx_gts, idx_pred]
                assigned_gt_inds[idx_pred] = idx_gts + 1          # 0 negative, 1-based positives
                assigned_gt_labels[idx_pred] = gt_labels[idx_gts]
        max_overlaps = torch.zeros(num_bboxes, device=bboxes.device, dtype=bboxes.dtype)
        matched = torch.where(assigned_gt_inds > 0)
        iou = self.iou_calculator(gt_bboxes[assigned_gt_inds[matched].long() - 1], bboxes[matched]).diag()
        max_overlaps[matched] = iou
        return AssignResult(num_gts, assigned_gt_inds.long(), max_overlaps, labels=assigned_gt_labels)


class HungarianAssigner3D:
    """Reference :95-153."""

    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.), reg_cost=dict(type='BBoxBEVL1Cost', weight=1.0),
                 iou_cost=dict(type='IoU3DCost', weight=1.0), iou_calculator=dict(type='BboxOverlaps3D')):
        self.cls_cost = _build(cls_cost, _COSTS)
        self.reg_cost = _build(reg_cost, _COSTS)
        self.iou_cost = _build(iou_cost, _COSTS)
        self.iou_calculator = _build(iou_calculator, dict(BboxOverlaps3D=_BboxOverlaps3D))

    def assign(self, bboxes, gt_bboxes, gt_labels, cls_pred, train_cfg):
        num_gts, num_bboxes = gt_bboxes.size(0), bboxes.size(0)
        assigned_gt_inds = bboxes.new_full((num_bboxes,), -1, dtype=torch.long)     # 1. -1 by default
        assigned_labels = bboxes.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            if num_gts == 0:
                assigned_gt_inds[:] = 0                                              # all background
            return AssignResult(num_gts, assigned_gt_inds, None, labels=assigned_labels)
        # 2. weighted costs
        cls_cost = self.cls_cost(cls_pred[0].T, gt_labels)
        reg_cost = self.reg_cost(bboxes, gt_bboxes, train_cfg)
        iou = self.iou_calculator(bboxes, gt_bboxes)
        cost = cls_cost + reg_cost + self.iou_cost(iou)
        # 3. Hungarian matching on the host
        rows, cols = linear_sum_assignment(cost.detach().cpu())
        rows = torch.from_numpy(rows).to(bboxes.device)
        cols = torch.from_numpy(cols).to(bboxes.device)
        # 4. backgrounds, then foregrounds
        assigned_gt_inds[:] = 0
        assigned_gt_inds[rows] = cols + 1
        assigned_labels[rows] = gt_labels[cols]
        max_overlaps = torch.zeros_like(iou.max(1).values)
        max_overlaps[rows] = iou[rows, cols]
        return AssignResult(num_gts, assigned_gt_inds, max_overlaps, labels=assigned_labels)


    def assign_layers(self, bboxes, gt_bboxes, gt_labels, cls_pred, train_cfg, num_layers):
        """`assign` for the `num_layers` groups of Q consecutive proposals of one sample at once (every decoder layer is
        assigned separately, reference transfusion-style head :398-414): the three costs are row-wise / pair-wise, so ONE
        evaluation over all num_layers * Q rows gives bit-for-bit the per-layer matrices; one device -> host copy, the
        Hungarian solves on the host, and the ensemble result (the concatenation of the per-layer `AssignResult`s) is
        written with three scatters.  The matched rows / columns stay available on the host (`host_rows`, `host_cols`)
        so that the sampler needs no `nonzero` round trip."""
        num_gts, n = gt_bboxes.size(0), bboxes.size(0)
        Q = n // num_layers
        gt_inds = bboxes.new_full((n,), -1, dtype=torch.long)
        labels = bboxes.new_full((n,), -1, dtype=torch.long)
        if num_gts == 0 or n == 0:
            if num_gts == 0:
                gt_inds[:] = 0
            r = AssignResult(num_layers * num_gts, gt_inds, bboxes.new_zeros(n), labels=labels)
            r.host_rows = r.host_cols = np.zeros(0, dtype=np.int64)
            return r
        iou = self.iou_calculator(bboxes, gt_bboxes)
        cost = self.cls_cost(cls_pred[0].T, gt_labels) + self.reg_cost(bboxes, gt_bboxes, train_cfg) + self.iou_cost(iou)
        cost = cost.detach().cpu()
        rows, cols = [], []
        for l in range(num_layers):
            r_, c_ = linear_sum_assignment(cost[l * Q:(l + 1) * Q])
            rows.append(np.asarray(r_, dtype=np.int64) + l * Q)
            cols.append(np.asarray(c_, dtype=np.int64))
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        rows_d = torch.from_numpy(rows).to(bboxes.device)
        cols_d = torch.from_numpy(cols).to(bboxes.device)
        gt_inds[:] = 0
        gt_inds[rows_d] = cols_d + 1
        labels[rows_d] = gt_labels[cols_d]
        max_overlaps = bboxes.new_zeros(n)
        max_overlaps[rows_d] = iou[rows_d, cols_d]
        r = AssignResult(num_layers * num_gts, gt_inds, max_overlaps, labels=labels)
        r.host_rows, r.host_cols = rows, cols
        return r


def build_assigner(cfg):
    return _build(cfg, dict(HungarianAssigner3D=HungarianAssigner3D, HeuristicAssigner3D=HeuristicAssigner3D))
