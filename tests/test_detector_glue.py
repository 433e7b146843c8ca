"""Host-side checks of the inference glue (`DeepInteractionInference`, `bbox3d2result`): the call order and argument
routing of the reference's `extract_feat` / `simple_test_pts` / `simple_test` (detectors/deepinteraction.py:142-149,
:244-266), with recording stand-ins for the neck, the head and the LiDAR backbone."""
import torch

from deepinteraction_amd.det3d_compat import LiDARBoxes
from deepinteraction_amd.mmdet3d_plugin import DeepInteractionInference, bbox3d2result


class _Neck:
    def __init__(self):
        self.calls = []

    def __call__(self, img, pts, img_metas, pts_metas):
        self.calls.append((img, pts, img_metas, pts_metas))
        return 'new_img', 'new_pts'


class _Head:
    def __init__(self):
        self.calls = []

    def __call__(self, x, x_img, img_metas):
        self.calls.append(('forward', x, x_img))
        return 'outs'

    def get_bboxes(self, outs, img_metas, rescale=False):
        self.calls.append(('get_bboxes', outs, rescale))
        boxes = LiDARBoxes(torch.arange(18.).reshape(2, 9), box_dim=9)
        return [[boxes, torch.tensor([0.9, 0.2]), torch.tensor([3, 1], dtype=torch.int32)]]


def _glue(multi_scale):
    image_glue = lambda img, metas: ('lvl0', 'lvl1', 'lvl2')
    point_glue = lambda points: dict(pillars='pillars', pts=points)
    backbone = lambda points: ['bev0', 'bev1']
    neck, head = _Neck(), _Head()
    return DeepInteractionInference(image_glue, backbone, point_glue, neck, head, multi_scale=multi_scale), neck, head


def test_extract_feat_routes_first_levels_in_v1_and_lists_in_plusplus():
    det, neck, _ = _glue(False)
    metas = [dict()]
    assert det.extract_feat(['p'], 'img', metas) == ('new_img', 'new_pts')
    img, pts, m, pm = neck.calls[0]
    assert (img, pts) == ('lvl0', 'bev0') and m is metas and pm['pts'] == ['p']
    det, neck, _ = _glue(True)
    det.extract_feat(['p'], 'img', metas)
    img, pts, _, _ = neck.calls[0]
    assert img == ['lvl0', 'lvl1'] and pts == ['bev0', 'bev1']


def test_simple_test_returns_host_side_results_per_sample():
    det, _, head = _glue(False)
    out = det.simple_test(['p'], [dict()], img='img', rescale=True)
    assert head.calls == [('forward', 'new_pts', 'new_img'), ('get_bboxes', 'outs', True)]     # head(x=pts, x_img=img)
    assert len(out) == 1 and set(out[0]) == {'pts_bbox'}
    r = out[0]['pts_bbox']
    assert set(r) == {'boxes_3d', 'scores_3d', 'labels_3d'}
    assert isinstance(r['boxes_3d'], LiDARBoxes) and r['boxes_3d'].tensor.shape == (2, 9)
    assert r['scores_3d'].device.type == 'cpu' and r['labels_3d'].tolist() == [3, 1]


def test_bbox3d2result_attrs():
    b = LiDARBoxes(torch.zeros(1, 7))
    r = bbox3d2result(b, torch.ones(1), torch.zeros(1, dtype=torch.long), attrs=torch.tensor([2]))
    assert r['attrs_3d'].tolist() == [2] and 'attrs_3d' not in bbox3d2result(b, torch.ones(1), torch.zeros(1))
