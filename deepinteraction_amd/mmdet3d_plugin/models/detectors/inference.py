"""Inference glue of the reference detector around the hot path: `DeepInteraction.extract_feat` (reference
`models/detectors/deepinteraction.py:142-149`), `simple_test_pts` (:244-254), `simple_test` (:256-266) and mmdet3d 0.17.1's
`bbox3d2result` (`core/bbox/transforms.py`), composed from this repo's producers (`ImageGlue`, `PointGlue`) and modules
(`imgpts_neck`, `pts_bbox_head`).

The sparse LiDAR backbone is not built here (no spconv on ROCm): `pts_backbone` is
any callable `points -> sequence of BEV maps` (the reference's `pts_voxel_layer -> pts_voxel_encoder -> pts_middle_encoder ->
pts_backbone -> pts_neck` chain, :124-131), supplied by the caller."""
import torch
from torch import nn


def bbox3d2result(bboxes, scores, labels, attrs=None):
    """mmdet3d `bbox3d2result`: one sample's detections as host-side entries."""
    result = dict(boxes_3d=bboxes.to('cpu'), scores_3d=scores.cpu(), labels_3d=labels.cpu())
    if attrs is not None:
        result['attrs_3d'] = attrs.cpu()
    return result


class DeepInteractionInference(nn.Module):
    def __init__(self, image_glue, pts_backbone, point_glue, imgpts_neck, pts_bbox_head, multi_scale=False):
        """multi_scale: the DeepInteraction++ form (`Fusion_0075_plusplus.py`): the neck reads two image levels and the
        whole list of BEV maps (:148) instead of the first of each (:146)."""
        super().__init__()
        self.image_glue, self.point_glue = image_glue, point_glue
        self.pts_backbone = pts_backbone
        self.imgpts_neck, self.pts_bbox_head = imgpts_neck, pts_bbox_head
        self.multi_scale = multi_scale

    @torch.no_grad()
    def extract_feat(self, points, img, img_metas):
        img_feats = self.image_glue(img, img_metas)
        pts_feats = self.pts_backbone(points)
        pts_metas = self.point_glue(points)
        if not self.multi_scale:
            return self.imgpts_neck(img_feats[0], pts_feats[0], img_metas, pts_metas)
        return self.imgpts_neck(list(img_feats[:2]), pts_feats, img_metas, pts_metas)

    @torch.no_grad()
    def simple_test_pts(self, x, x_img, img_metas, rescale=False):
        outs = self.pts_bbox_head(x, x_img, img_metas)
        bbox_list = self.pts_bbox_head.get_bboxes(outs, img_metas, rescale=rescale)
        return [bbox3d2result(bboxes, scores, labels) for bboxes, scores, labels in bbox_list]

    @torch.no_grad()
    def simple_test(self, points, img_metas, img=None, rescale=False):
        img_feats, pts_feats = self.extract_feat(points, img=img, img_metas=img_metas)
        bbox_list = [dict() for _ in range(len(img_metas))]
        for result, pts_bbox in zip(bbox_list, self.simple_test_pts(pts_feats, img_feats, img_metas, rescale=rescale)):
            result['pts_bbox'] = pts_bbox
        return bbox_list
