"""TransFusionBBoxCoder - drop-in for reference `core/bbox/coders/transfusion_bbox_coder.py:7-126`
(same registry name, constructor kwargs, encode/decode semantics).  Inside the MMPI forward the
decode is fused into the `query_geometry` HIP kernel; this class is the host-visible API
(`get_bboxes`, target encoding) and works on any device with plain torch ops."""
import torch

from deepinteraction_amd.registry import BBOX_CODERS


@BBOX_CODERS.register_module()
class TransFusionBBoxCoder:
    def __init__(self, pc_range, out_size_factor, voxel_size, post_center_range=None, score_threshold=None,
                 code_size=8):
        self.pc_range = pc_range
        self.out_size_factor = out_size_factor
        self.voxel_size = voxel_size
        self.post_center_range = post_center_range
        self.score_threshold = score_threshold
        self.code_size = code_size

    def encode(self, dst_boxes):
        targets = torch.zeros([dst_boxes.shape[0], self.code_size]).to(dst_boxes.device)
        targets[:, 0] = (dst_boxes[:, 0] - self.pc_range[0]) / (self.out_size_factor * self.voxel_size[0])
        targets[:, 1] = (dst_boxes[:, 1] - self.pc_range[1]) / (self.out_size_factor * self.voxel_size[1])
        targets[:, 3] = dst_boxes[:, 3].log()
        targets[:, 4] = dst_boxes[:, 4].log()
        targets[:, 5] = dst_boxes[:, 5].log()
        targets[:, 2] = dst_boxes[:, 2] + dst_boxes[:, 5] * 0.5        # bottom centre -> gravity centre
        targets[:, 6] = torch.sin(dst_boxes[:, 6])
        targets[:, 7] = torch.cos(dst_boxes[:, 6])
        if self.code_size == 10:
            targets[:, 8:10] = dst_boxes[:, 7:]
        return targets

    def decode(self, heatmap, rot, dim, center, height, vel, filter=False):
        """(B,k,Q) head outputs -> list of dicts(bboxes (Q,7|9), scores, labels).  Unlike the
        reference (:57-70) the inputs are not modified in place."""
        final_preds = heatmap.max(1, keepdims=False).indices
        final_scores = heatmap.max(1, keepdims=False).values
        cx = center[:, 0:1] * self.out_size_factor * self.voxel_size[0] + self.pc_range[0]
        cy = center[:, 1:2] * self.out_size_factor * self.voxel_size[1] + self.pc_range[1]
        dim = dim.exp()
        height = height - dim[:, 2:3] * 0.5                               # gravity -> bottom centre
        rot = torch.atan2(rot[:, 0:1], rot[:, 1:2])
        parts = [cx, cy, height, dim, rot] + ([] if vel is None else [vel])
        final_box_preds = torch.cat(parts, dim=1).permute(0, 2, 1)
        if filter is False:
            return [dict(bboxes=final_box_preds[i], scores=final_scores[i], labels=final_preds[i])
                    for i in range(heatmap.shape[0])]
        if self.post_center_range is None:
            raise NotImplementedError('Need to reorganize output as a batch, only '
                                      'support post_center_range is not None for now!')
        pcr = torch.as_tensor(self.post_center_range, device=heatmap.device, dtype=final_box_preds.dtype)
        mask = (final_box_preds[..., :3] >= pcr[:3]).all(2) & (final_box_preds[..., :3] <= pcr[3:]).all(2)
        if self.score_threshold is not None:
            thresh = final_scores > self.score_threshold
        out = []
        for i in range(heatmap.shape[0]):
            cmask = mask[i]
            if self.score_threshold:
                cmask = cmask & thresh[i]
            out.append(dict(bboxes=final_box_preds[i, cmask], scores=final_scores[i, cmask],
                            labels=final_preds[i, cmask]))
        return out
