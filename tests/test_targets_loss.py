"""CPU: the head's training targets, losses and box post-processing.

  * first-principles checks of the restated mmdet / mmdet3d helpers (deepinteraction_amd/det3d_compat.py;
    parity unpinned at that boundary): rotated-box intersection against rasterisation and known answers,
    IoU identities, Gaussian radius / drawing, focal losses against their formulas, circle NMS;
  * the head's own logic - get_targets(_single) / loss / get_bboxes and the Hungarian / heuristic assigners -
    against the REFERENCE's own Python executed from /root/reference (skipped where it is absent), both
    sides calling the same restated helpers: labels, weights, targets, heat maps, every loss term and the
    decoded boxes must agree.
"""
import math

import numpy as np
import pytest
import torch

from deepinteraction_amd import det3d_compat as dc, synth
from deepinteraction_amd.configs import decoder_cfg
from deepinteraction_amd.mmdet3d_plugin import DeepInteractionDecoder
from deepinteraction_amd.mmdet3d_plugin.core.bbox.assigners import HeuristicAssigner3D, HungarianAssigner3D
from oracle.refpin import load_reference, reference_available

TRAIN_CFG = dict(
    dataset='nuScenes',
    assigner=dict(type='HungarianAssigner3D', iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar'),
                  cls_cost=dict(type='FocalLossCost', gamma=2, alpha=0.25, weight=0.15),
                  reg_cost=dict(type='BBoxBEVL1Cost', weight=0.25), iou_cost=dict(type='IoU3DCost', weight=0.25)),
    pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[288, 288, 40], voxel_size=[0.375, 0.375, 0.2],
    out_size_factor=8, code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2],
    point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0])


# ------------------------------------------------------------------ first principles
def _raster_area(a, b, n=1200):
    """Intersection area of two rotated rectangles [x,y,dx,dy,yaw] by point sampling on a fine grid."""
    lo = min(a[0] - a[2] - a[3], b[0] - b[2] - b[3]); hi = max(a[0] + a[2] + a[3], b[0] + b[2] + b[3])
    lo2 = min(a[1] - a[2] - a[3], b[1] - b[2] - b[3]); hi2 = max(a[1] + a[2] + a[3], b[1] + b[2] + b[3])
    xs = np.linspace(lo, hi, n); ys = np.linspace(lo2, hi2, n)
    X, Y = np.meshgrid(xs, ys)

    def inside(r):
        c, s = math.cos(r[4]), math.sin(r[4])
        # mmdet3d 0.17.1 yaw convention (clockwise in BEV): world = [[c, s], [-s, c]] . local
        u = (X - r[0]) * c - (Y - r[1]) * s
        w = (X - r[0]) * s + (Y - r[1]) * c
        return (np.abs(u) <= r[2] / 2) & (np.abs(w) <= r[3] / 2)
    return (inside(a) & inside(b)).mean() * (hi - lo) * (hi2 - lo2)


def test_rotated_intersection():
    g = torch.Generator().manual_seed(0)
    a = torch.cat([torch.rand(40, 2, generator=g) * 4, torch.rand(40, 2, generator=g) * 3 + 0.5,
                   (torch.rand(40, 1, generator=g) - 0.5) * 6.3], 1)
    b = torch.cat([a[:, :2] + (torch.rand(40, 2, generator=g) - 0.5) * 3, torch.rand(40, 2, generator=g) * 3 + 0.5,
                   (torch.rand(40, 1, generator=g) - 0.5) * 6.3], 1)
    got = dc.rotated_intersection_area(a, b)
    for i in range(40):
        ref = _raster_area(a[i].tolist(), b[i].tolist())
        assert abs(got[i].item() - ref) <= 0.02 * max(ref, 0.3), (i, got[i].item(), ref)
    # known answers: identical, axis-aligned overlap, disjoint, 45-degree square in a square, containment
    r = lambda *v: torch.tensor([v], dtype=torch.float32)
    assert abs(dc.rotated_intersection_area(r(1, 2, 3, 4, 0.7), r(1, 2, 3, 4, 0.7)).item() - 12) < 1e-5
    assert abs(dc.rotated_intersection_area(r(0, 0, 2, 2, 0), r(1, 1, 2, 2, 0)).item() - 1) < 1e-6
    assert dc.rotated_intersection_area(r(0, 0, 1, 1, 0.3), r(5, 5, 1, 1, 1.0)).item() == 0
    assert abs(dc.rotated_intersection_area(r(0, 0, 2, 2, 0), r(0, 0, 2, 2, math.pi / 4)).item()
               - (8 * math.sqrt(2) - 8)) < 1e-5                           # regular octagon
    assert abs(dc.rotated_intersection_area(r(0, 0, 10, 10, 0.2), r(0.5, 0.3, 1, 2, 1.1)).item() - 2) < 1e-6


def test_bev_polygon_follows_lidar_corners_convention():
    """The BEV rectangle the IoU / NMS code clips must be the footprint `lidar_corners` (and the query_geometry
    kernel, and mmdet3d 0.17.1's iou3d `rotate_around_center`) gives a box: off-axis centres, non-zero yaw."""
    g = torch.Generator().manual_seed(3)
    b = torch.cat([(torch.rand(9, 3, generator=g) - 0.5) * 20, torch.rand(9, 3, generator=g) * 3 + 0.5,
                   (torch.rand(9, 1, generator=g) - 0.5) * 6], 1)
    foot = dc.lidar_corners(b)[:, [0, 3, 7, 4], :2]                   # bottom face
    poly = dc._rect_corners_bev(b[:, [0, 1, 3, 4, 6]])
    for i in range(9):                                                # same vertex SET (any starting vertex)
        d = (foot[i][:, None, :] - poly[i][None, :, :]).norm(dim=-1)
        assert d.min(1).values.max() < 1e-4 and d.min(0).values.max() < 1e-4, i
    x, y = poly[..., 0], poly[..., 1]
    assert ((x * y.roll(-1, 1) - x.roll(-1, 1) * y).sum(1) > 0).all()      # counter-clockwise loop
    # the advisor's counter example: 4x1 boxes at (0,0) and (1,1), yaw pi/4 -> both lie along the ANTI-diagonal
    # (clockwise yaw), side by side: no overlap.  With the opposite sign they would overlap (IoU 0.478).
    a = torch.tensor([[0., 0, 0, 4, 1, 1, math.pi / 4]]); c = torch.tensor([[1., 1, 0, 4, 1, 1, math.pi / 4]])
    assert dc.boxes_iou3d_lidar(a, c).item() < 1e-6
    c2 = torch.tensor([[1., -1, 0, 4, 1, 1, math.pi / 4]])                # shifted ALONG the anti-diagonal
    inter = 4 - math.sqrt(2)
    assert abs(dc.boxes_iou3d_lidar(a, c2).item() - inter / (8 - inter)) < 1e-5


def test_iou3d_identities():
    g = torch.Generator().manual_seed(1)
    b = torch.cat([torch.rand(6, 3, generator=g) * 10, torch.rand(6, 3, generator=g) * 3 + 0.5,
                   torch.rand(6, 1, generator=g) * 6 - 3], 1)
    iou = dc.boxes_iou3d_lidar(b, b)
    assert torch.allclose(iou.diag(), torch.ones(6), atol=1e-5)
    assert torch.allclose(iou, iou.T, atol=1e-6) and (iou >= 0).all() and (iou <= 1 + 1e-6).all()
    # stacked boxes: same footprint, half height overlap -> iou = 0.5 h / (1.5 h) = 1/3
    a = torch.tensor([[0.0, 0, 0, 2, 3, 2, 0.4]]); c = torch.tensor([[0.0, 0, 1, 2, 3, 2, 0.4]])
    assert abs(dc.boxes_iou3d_lidar(a, c).item() - 1 / 3) < 1e-5
    assert dc.boxes_iou3d_lidar(a, torch.zeros(0, 7)).shape == (1, 0)


def test_gaussian_and_losses():
    r = dc.gaussian_radius((torch.tensor(4.0), torch.tensor(6.0)), min_overlap=0.1)
    assert 2.0 < float(r) < 6.0
    hm = torch.zeros(12, 12)
    dc.draw_heatmap_gaussian(hm, torch.tensor([10, 1]), 3)               # clipped at two borders
    assert hm[1, 10] == 1 and hm.max() == 1 and hm[1, 7] > 0 and hm[5, 10] == 0 and (hm >= 0).all()
    dc.draw_heatmap_gaussian(hm, torch.tensor([9, 1]), 2)                # element-wise max with what is there
    assert hm[1, 10] == 1 and hm[1, 9] == 1
    # focal loss against the formula
    g = torch.Generator().manual_seed(2)
    x = torch.randn(7, 4, generator=g); t = torch.tensor([0, 3, 4, 2, 4, 1, 4])      # 4 = background
    w = torch.tensor([1, 1, 0, 1, 1, 1, 1])
    got = dc.FocalLoss(gamma=2.0, alpha=0.25)(x, t, w, avg_factor=3.0)
    p = x.sigmoid(); ref = 0.0
    for i in range(7):
        for c in range(4):
            y = 1.0 if t[i] == c else 0.0
            pt = p[i, c] if y else 1 - p[i, c]
            ref += float(w[i]) * float(-(0.25 if y else 0.75) * (1 - pt) ** 2 * torch.log(pt))
    assert abs(got.item() - ref / 3.0) < 1e-5
    pr = torch.rand(2, 3, 5, 5, generator=g) * 0.98 + 0.01; tg = torch.rand(2, 3, 5, 5, generator=g); tg[0, 0, 2, 2] = 1
    gl = dc.GaussianFocalLoss()(pr, tg, avg_factor=1.0)
    ref = (-(pr + 1e-12).log() * (1 - pr) ** 2 * (tg == 1) - (1 - pr + 1e-12).log() * pr ** 2 * (1 - tg) ** 4).sum()
    assert torch.allclose(gl, ref, rtol=1e-5)
    assert torch.allclose(dc.L1Loss(loss_weight=0.25)(torch.ones(2, 3), torch.zeros(2, 3), torch.ones(2, 3), avg_factor=2.0),
                          torch.tensor(0.75))
    keep = dc.circle_nms(np.array([[0, 0, 0.9], [0.1, 0, 0.8], [1, 1, 0.7], [0.2, 0.1, 0.95]], np.float32), 0.175)
    assert keep == [3, 2]


def test_assigners_known_answers():
    gt = torch.tensor([[0.0, 0, 0, 2, 4, 1.5, 0.1, 0, 0], [10.0, 5, 0, 1, 1, 1, 0.0, 0, 0]])
    lab = torch.tensor([2, 7])
    boxes = torch.tensor([[9.8, 5.1, 0, 1, 1, 1, 0.0, 0, 0], [30.0, 30, 0, 1, 1, 1, 0, 0, 0],
                          [0.2, -0.1, 0, 2, 4, 1.5, 0.1, 0, 0], [-20.0, 3, 0, 2, 2, 2, 1.0, 0, 0]])
    cls = torch.full((1, 10, 4), -4.0); cls[0, 7, 0] = 3.0; cls[0, 2, 2] = 3.0
    r = HungarianAssigner3D(**{k: v for k, v in TRAIN_CFG['assigner'].items() if k != 'type'}).assign(
        boxes, gt, lab, cls, TRAIN_CFG)
    assert r.gt_inds.tolist() == [2, 0, 1, 0] and r.labels.tolist() == [7, -1, 2, -1]
    assert r.max_overlaps[0] > 0.5 and r.max_overlaps[2] > 0.7 and r.max_overlaps[1] == 0
    e = HungarianAssigner3D().assign(boxes, gt[:0], lab[:0], cls, TRAIN_CFG)
    assert e.gt_inds.tolist() == [0, 0, 0, 0] and e.max_overlaps is None
    h = HeuristicAssigner3D(dist_thre=5).assign(boxes, gt, None, lab, torch.tensor([7, 1, 2, 2]))
    assert h.gt_inds.tolist() == [2, 0, 1, 0]


def test_assign_layers_equals_per_layer_assignment():
    """`HungarianAssigner3D.assign_layers` (all decoder layers' costs in one evaluation, one device -> host copy) against
    `assign` called per layer: the concatenated result bit for bit, the matched rows on the host, the sync-free sampler."""
    g = torch.Generator().manual_seed(3)
    L, Q, G = 5, 40, 9
    gt = torch.cat([(torch.rand(G, 2, generator=g) - 0.5) * 80, torch.rand(G, 1, generator=g) * 2 - 2,
                    torch.rand(G, 3, generator=g) * 3 + 0.5, (torch.rand(G, 1, generator=g) - 0.5) * 6, torch.randn(G, 2, generator=g)], 1)
    lab = torch.randint(0, 10, (G,), generator=g)
    boxes = torch.cat([(torch.rand(L * Q, 2, generator=g) - 0.5) * 80, torch.rand(L * Q, 1, generator=g) * 2 - 2,
                       torch.rand(L * Q, 3, generator=g) * 3 + 0.5, (torch.rand(L * Q, 1, generator=g) - 0.5) * 6,
                       torch.randn(L * Q, 2, generator=g)], 1)
    boxes[::7, :7] = gt[torch.arange(boxes[::7].shape[0]) % G, :7]                   # some proposals sit on a ground truth
    cls = torch.randn(1, 10, L * Q, generator=g)
    asg = HungarianAssigner3D(**{k: v for k, v in TRAIN_CFG['assigner'].items() if k != 'type'})
    per = [asg.assign(boxes[l * Q:(l + 1) * Q], gt, lab, cls[..., l * Q:(l + 1) * Q], TRAIN_CFG) for l in range(L)]
    ens = asg.assign_layers(boxes, gt, lab, cls, TRAIN_CFG, L)
    assert ens.num_gts == sum(r.num_gts for r in per)
    assert torch.equal(ens.gt_inds, torch.cat([r.gt_inds for r in per]))
    assert torch.equal(ens.labels, torch.cat([r.labels for r in per]))
    assert torch.equal(ens.max_overlaps, torch.cat([r.max_overlaps for r in per]))
    assert sorted(ens.host_rows.tolist()) == torch.nonzero(ens.gt_inds > 0).flatten().tolist()
    fast = dc.pseudo_sample(ens, boxes, gt)
    slow = dc.pseudo_sample(dc.AssignResult(ens.num_gts, ens.gt_inds, ens.max_overlaps, ens.labels), boxes, gt)
    for k in ('pos_inds', 'neg_inds', 'pos_assigned_gt_inds', 'pos_gt_bboxes'):
        assert torch.equal(getattr(fast, k), getattr(slow, k)), k
    empty = asg.assign_layers(boxes, gt[:0], lab[:0], cls, TRAIN_CFG, L)
    assert empty.gt_inds.eq(0).all() and empty.host_rows.size == 0


def test_host_heatmap_drawing_equals_tensor_drawing():
    """The numpy Gaussian drawing of the host-side target builder against the tensor form (same float64 Gaussian, float32 max)."""
    hm_t, hm_n = torch.zeros(40, 50), np.zeros((40, 50), dtype=np.float32)
    for (x, y, r) in ((10, 12, 3), (11, 13, 5), (0, 0, 4), (49, 39, 6), (25, 2, 2)):
        dc.draw_heatmap_gaussian(hm_t, torch.tensor([x, y]), r)
        dc.draw_heatmap_gaussian_host(hm_n, (x, y), r)
    assert np.array_equal(hm_t.numpy(), hm_n) and (hm_n == 1).sum() == 5


# ------------------------------------------------------------------ against the reference's own Python
@pytest.fixture(scope='module')
def ref():
    if not reference_available():
        pytest.skip('/root/reference is not present on this machine')
    return load_reference('reference')


def _gt_from_preds(head, preds, n_gt, seed):
    """Ground truth near some predicted boxes (so that the matching is not degenerate) + far-away boxes."""
    g = torch.Generator().manual_seed(seed)
    pd = preds[0][0]
    Q = head.num_proposals
    dec = head.bbox_coder.decode(pd['heatmap'][..., -Q:].detach().clone(), pd['rot'][..., -Q:].detach().clone(),
                                 pd['dim'][..., -Q:].detach().clone(), pd['center'][..., -Q:].detach().clone(),
                                 pd['height'][..., -Q:].detach().clone(), pd['vel'][..., -Q:].detach().clone())
    out_boxes, out_labels = [], []
    for b in range(len(dec)):
        pick = torch.randperm(Q, generator=g)[:n_gt]
        bx = dec[b]['bboxes'][pick].clone()
        bx[:, :2] += torch.randn(n_gt, 2, generator=g) * 0.5
        bx[:, 3:6] = bx[:, 3:6].clamp(0.3, 8.0) * (1 + 0.1 * torch.randn(n_gt, 3, generator=g)).clamp(0.5, 1.5)
        bx[:, :2] = bx[:, :2].clamp(-50, 50)
        out_boxes.append(bx)
        out_labels.append(torch.randint(0, 10, (n_gt,), generator=g))
    return out_boxes, out_labels


@pytest.mark.parametrize('nms', [None, 'circle'])
def test_loss_targets_bboxes_match_reference(ref, nms):
    shape = synth.SHAPE_TINY
    cfg = decoder_cfg(bev=36, num_proposals=24)
    cfg['test_cfg']['nms_type'] = nms
    torch.manual_seed(7)
    R = ref.decoder.DeepInteractionDecoder(**dict(cfg, train_cfg=ref.stubs.ConfigDict(TRAIN_CFG)))
    M = DeepInteractionDecoder(**dict(cfg, train_cfg=TRAIN_CFG))
    M.load_state_dict(R.state_dict())
    R.eval()
    g = torch.Generator().manual_seed(0)
    Hi, Wi = shape['img_hw']
    B = 2
    p0, p1 = torch.randn(B, 128, 36, 36, generator=g), torch.randn(B, 128, 36, 36, generator=g)
    img = torch.randn(6 * B, 128, Hi, Wi, generator=g)
    metas = synth.make_inputs(B, shape, seed=0)['img_metas']
    with torch.no_grad():
        preds = R([p0, p1], img, metas)
    M.query_labels, M.on_the_image_mask = R.query_labels, R.on_the_image_mask
    gt_boxes, gt_labels = _gt_from_preds(M, preds, 5, 3)
    gt_ref = [ref.stubs.LiDARInstance3DBoxes(b) for b in gt_boxes]
    gt_our = [dc.LiDARBoxes(b) for b in gt_boxes]
    clone = lambda: [[{k: v.clone() for k, v in preds[0][0].items()}]]
    tr = R.get_targets(gt_ref, gt_labels, clone()[0])
    tm = M.get_targets(gt_our, gt_labels, clone()[0])
    names = ('labels', 'label_weights', 'bbox_targets', 'bbox_weights', 'ious', 'num_pos', 'matched_ious', 'heatmap')
    for n, a, b in zip(names, tr, tm):
        if torch.is_tensor(a):
            assert a.shape == b.shape and torch.allclose(a.float(), b.float(), atol=1e-6), n
        else:
            assert abs(a - b) < 1e-6, n
    assert tr[5] == 4 * 5 * B                                             # every GT matched in every layer
    lr = R.loss(gt_ref, gt_labels, clone())
    lm = M.loss(gt_our, gt_labels, clone())
    assert set(lr) == set(lm)
    for k in lr:
        assert torch.allclose(lr[k].float(), lm[k].float(), rtol=1e-5, atol=1e-6), (k, lr[k], lm[k])
    assert lr['loss_heatmap'] > 0 and lr['layer_3_loss_bbox'] > 0
    # get_bboxes asserts batch size 1
    one = lambda: [[{k: v[:1].clone() for k, v in preds[0][0].items()}]]   # (the reference decodes in place)
    R.query_labels, M.query_labels = R.query_labels[:1], M.query_labels[:1]
    m1 = [dict(metas[0], box_type_3d=ref.stubs.LiDARInstance3DBoxes)]
    br = R.get_bboxes(one(), m1)[0]
    bm = M.get_bboxes(one(), [metas[0]])[0]
    assert torch.allclose(br[0].tensor, bm[0].tensor, atol=1e-6) and torch.allclose(br[1], bm[1], atol=1e-7)
    assert torch.equal(br[2], bm[2]) and len(bm[0]) > 0


def test_loss_has_gradients_and_empty_gt():
    """Our head alone (no reference needed): losses are differentiable w.r.t. the predictions and a sample
    without ground truth yields pure background targets."""
    cfg = decoder_cfg(bev=36, num_proposals=16)
    torch.manual_seed(0)
    M = DeepInteractionDecoder(**dict(cfg, train_cfg=TRAIN_CFG))
    g = torch.Generator().manual_seed(5)
    Q4 = 64
    pd = dict(center=torch.rand(1, 2, Q4, generator=g) * 36, height=torch.randn(1, 1, Q4, generator=g),
              dim=torch.randn(1, 3, Q4, generator=g) * 0.3, rot=torch.randn(1, 2, Q4, generator=g),
              vel=torch.randn(1, 2, Q4, generator=g), heatmap=torch.randn(1, 10, Q4, generator=g),
              dense_heatmap=torch.randn(1, 10, 36, 36, generator=g),
              query_heatmap_score=torch.rand(1, 10, 16, generator=g))
    pd = {k: v.requires_grad_(True) for k, v in pd.items()}
    M.on_the_image_mask = [torch.rand(1, 16, generator=g) > 0.3, torch.rand(1, 16, generator=g) > 0.3]
    M.query_labels = torch.randint(0, 10, (1, 16), generator=g)
    gt = [dc.LiDARBoxes(torch.tensor([[3.0, -4, -1, 1.9, 4.5, 1.6, 0.3, 0, 0], [-20.0, 11, -1, 0.6, 0.7, 1.7, 1.0, 0.5, 0]]))]
    loss = M.loss(gt, [torch.tensor([0, 8])], [[pd]])
    total = sum(v for k, v in loss.items() if k != 'matched_ious')
    total.backward()
    for k in ('center', 'height', 'dim', 'rot', 'vel', 'heatmap', 'dense_heatmap'):
        assert pd[k].grad is not None and torch.isfinite(pd[k].grad).all() and pd[k].grad.abs().sum() > 0, k
    empty = M.get_targets([dc.LiDARBoxes(torch.zeros(0, 9))], [torch.zeros(0, dtype=torch.long)],
                          [{k: v.detach() for k, v in pd.items()}])
    assert (empty[0] == 10).all() and empty[5] == 0 and empty[7].abs().sum() == 0
    # `prepare_targets` (the ground-truth-only heat map drawn ahead of the loss, e.g. under the forward's graph replay)
    # changes nothing: same losses, and the prepared entries are consumed
    labels = [torch.tensor([0, 8])]
    M.prepare_targets(gt, labels, 'cpu')
    assert len(M._gt_targets) == 1
    again = M.loss(gt, labels, [[{k: v.detach() for k, v in pd.items()}]])
    assert len(M._gt_targets) == 0
    for k in loss:
        assert torch.equal(again[k].float(), loss[k].detach().float()), k


def test_pp_head_loss_matches_reference(ref):
    """DeepInteraction++ head: forward (with the three shims of oracle/refpin), targets and every loss term against
    the reference's own `DeepInteractionPlusPlusDecoder` - the ++ loss weights EVERY MMPI layer by the cumulative
    on-the-image mask (deepinteractionplusplus_decoder.py:513-514), the v1 loss only the image layers."""
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionPlusPlusDecoder
    shape = synth.SHAPE_TINY
    cfg = decoder_cfg(bev=36, num_proposals=24)
    torch.manual_seed(7)
    R = ref.decoder_pp.DeepInteractionPlusPlusDecoder(**dict(cfg, train_cfg=ref.stubs.ConfigDict(TRAIN_CFG)))
    M = DeepInteractionPlusPlusDecoder(**dict(cfg, train_cfg=TRAIN_CFG))
    M.load_state_dict(R.state_dict())
    R.eval()
    g = torch.Generator().manual_seed(0)
    Hi, Wi = shape['img_hw']
    B = 2
    p0, p1 = torch.randn(B, 128, 36, 36, generator=g), torch.randn(B, 128, 36, 36, generator=g)
    img = torch.randn(6 * B, 128, Hi, Wi, generator=g)
    metas = synth.make_inputs(B, shape, seed=0)['img_metas']
    with torch.no_grad():
        preds = R([p0, p1], img, metas)
    assert len(R.on_the_image_mask) == 4 and not all(m.all() for m in R.on_the_image_mask)
    M.query_labels, M.on_the_image_mask = R.query_labels, R.on_the_image_mask
    gt_boxes, gt_labels = _gt_from_preds(M, preds, 5, 3)
    gt_ref = [ref.stubs.LiDARInstance3DBoxes(b) for b in gt_boxes]
    gt_our = [dc.LiDARBoxes(b) for b in gt_boxes]
    clone = lambda: [[{k: v.clone() for k, v in preds[0][0].items()}]]
    lr = R.loss(gt_ref, gt_labels, clone())
    lm = M.loss(gt_our, gt_labels, clone())
    assert set(lr) == set(lm)
    for k in lr:
        assert torch.allclose(lr[k].float(), lm[k].float(), rtol=1e-5, atol=1e-6), (k, lr[k], lm[k])
    assert lr['layer_1_loss_bbox'] > 0


def test_rotated_nms_first_principles():
    """mmdet3d `nms_gpu` restated (det3d_compat.nms_rotated_bev): IoU matrix against rasterisation, known answers,
    greedy order, pre/post caps; and the head's get_bboxes with nms_type='rotate'."""
    g = torch.Generator().manual_seed(1)
    n = 12
    c = torch.cat([torch.rand(n, 2, generator=g) * 4, torch.rand(n, 2, generator=g) * 2 + 0.8,
                   (torch.rand(n, 1, generator=g) - 0.5) * 3], 1)
    iou = dc.boxes_iou_bev_xyxyr(dc.xywhr2xyxyr(c), dc.xywhr2xyxyr(c))
    assert torch.allclose(iou.diag(), torch.ones(n), atol=1e-6) and torch.allclose(iou, iou.t(), atol=1e-6)
    for i, j in ((0, 1), (2, 7), (3, 4), (5, 11)):
        inter = _raster_area(c[i].tolist(), c[j].tolist())
        ref = inter / (c[i, 2] * c[i, 3] + c[j, 2] * c[j, 3] - inter).item()
        assert abs(iou[i, j].item() - ref) <= 0.02, (i, j, iou[i, j].item(), ref)
    sq = torch.tensor([[0., 0., 1., 1., 0.], [0., 0., 1., 1., math.pi / 4], [3., 0., 1., 1., 0.2], [0., 0., 1., 1., 0.]])
    assert abs(dc.boxes_iou_bev_xyxyr(dc.xywhr2xyxyr(sq[:1]), dc.xywhr2xyxyr(sq[1:2])).item() - 2 ** -0.5) < 1e-5
    sc = torch.tensor([0.5, 0.9, 0.3, 0.8])
    assert dc.nms_rotated_bev(dc.xywhr2xyxyr(sq), sc, 0.6).tolist() == [1, 2]            # 3 and 0 overlap box 1 by 0.707
    assert dc.nms_rotated_bev(dc.xywhr2xyxyr(sq), sc, 0.75).tolist() == [1, 3, 2]        # box 0 == box 3: suppressed
    assert dc.nms_rotated_bev(dc.xywhr2xyxyr(sq), sc, 0.75, post_max_size=2).tolist() == [1, 3]
    assert dc.nms_rotated_bev(dc.xywhr2xyxyr(sq), sc, 0.75, pre_maxsize=2).tolist() == [1, 3]
    assert dc.nms_rotated_bev(sq[:0], sc[:0], 0.5).numel() == 0
    # through the head: pedestrians (class 8) closer than the threshold collapse to the best one
    cfg = decoder_cfg(bev=36, num_proposals=6)
    cfg['test_cfg'].update(nms_type='rotate', pre_maxsize=1000, post_maxsize=83)
    M = DeepInteractionDecoder(**cfg).eval()
    Q = 6
    pd = dict(center=torch.tensor([[[10., 10.05, 20., 10.1, 30., 5.], [10., 10., 20., 10.05, 30., 5.]]]),
              height=torch.zeros(1, 1, Q), dim=torch.zeros(1, 3, Q), rot=torch.tensor([[[0.] * Q, [1.] * Q]]),
              vel=torch.zeros(1, 2, Q), heatmap=torch.full((1, 10, Q), 3.0),
              query_heatmap_score=torch.tensor([0.9, 0.8, 0.7, 0.6, 0.5, 0.4]).view(1, 1, Q).expand(1, 10, Q).clone())
    M.query_labels = torch.tensor([[8, 8, 8, 8, 9, 0]])
    boxes, scores, labels = M.get_bboxes([[pd]], [dict()])[0]
    # queries 0, 1, 3 are pedestrians within a fraction of a box of each other (IoU > 0.175): only the best stays
    assert sorted(labels.tolist()) == [0, 8, 8, 9] and len(boxes) == 4
    assert abs(scores[labels == 8].max().item() - torch.sigmoid(torch.tensor(3.0)).item() * 0.9) < 1e-6
