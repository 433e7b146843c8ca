"""Target assignment of the MMPI head - drop-in for reference `core/bbox/assigners/hungarian_assigner.py`
(`BBox3DL1Cost` :14-21, `BBoxBEVL1Cost` :24-36, `IoU3DCost` :39-46, `HeuristicAssigner3D` :49-92,
`HungarianAssigner3D` :95-153): same names, constructor kwargs and `assign` signatures.  The mmdet
pieces they build on (FocalLossCost, BboxOverlaps3D, AssignResult) are the restatements of
deepinteraction_amd/det3d_compat.py.  As in the reference the cost matrix is formed on the device and the
Hungarian matching itself runs on the host (scipy `linear_sum_assignment`, one D2H copy per layer)."""
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


def build_assigner(cfg):
    return _build(cfg, dict(HungarianAssigner3D=HungarianAssigner3D, HeuristicAssigner3D=HeuristicAssigner3D))
