"""The point side of the reference detector that feeds the hot path: `DeepInteraction.voxelize` (reference
`models/detectors/deepinteraction.py:151-171`) and the `pts_metas` part of `extract_pts_feat` (:120-140), on the device
with this repo's voxeliser and WITHOUT host synchronisation.

The reference builds `pts_metas` from raw points with spconv's `PointToVoxel` (one call per sample, shapes that depend on
the data, a `torch.cat` over the batch).  A captured hipGraph wants fixed shapes: `padded=True` keeps `cap` rows per
sample (rows beyond a sample's pillar count are zero with `num_points = 0`, which every consumer skips), so the batch
split is known on the host (`pillar_batch_bounds`) and nothing depends on a device-side count.

Out of scope here: the frozen LiDAR backbone (`pts_voxel_layer` / `pts_middle_encoder` / `pts_backbone`) - its
output `pts_feats` is an input of the hot path.  The image side (`extract_img_feat`) is `image_glue.py`."""
import torch
from torch import nn

from ..updated_modules.sparse_voxelize import SPConvVoxelization


class PointGlue(nn.Module):
    def __init__(self, pts_pillar_layer, num_features=5):
        """pts_pillar_layer: the reference config dict (`Fusion_0075_refactor.py:151-155`: max_num_points, voxel_size,
        max_voxels, point_cloud_range) or an `SPConvVoxelization`; num_features: `HardSimpleVFE.num_features` (:156-159)."""
        super().__init__()
        self.pts_pillar_layer = pts_pillar_layer if isinstance(pts_pillar_layer, nn.Module) else \
            SPConvVoxelization(**pts_pillar_layer)
        self.num_features = num_features

    @torch.no_grad()
    def voxelize(self, points, padded=False):
        """Reference `voxelize(points, voxel_type='pillar')` (:151-171): per sample hard voxelisation, batch index padded
        in front of the coordinates; returns (voxels, num_points, coors_batch) in the reference's order."""
        layer = self.pts_pillar_layer
        voxels, coors, num_points = [], [], []
        for i, res in enumerate(points):
            if padded:
                v, c, n, _count = layer.forward_padded(res)          # capacity rows, no device->host sync
            else:
                v, c, n = layer(res)
            voxels.append(v)
            num_points.append(n)
            coors.append(torch.nn.functional.pad(c, (1, 0), mode='constant', value=i))
        return torch.cat(voxels, 0), torch.cat(num_points, 0), torch.cat(coors, 0)

    @torch.no_grad()
    def forward(self, points, padded=False):
        """points: list of (Np, >= num_features) float tensors on the device -> the `pts_metas` dict of
        `extract_pts_feat` (:132-140)."""
        pillars, num_points, coors = self.voxelize(points, padded)
        # HardSimpleVFE (mmdet3d voxel_encoders/voxel_encoder.py): mean of the points of a voxel
        center = pillars[:, :, :self.num_features].sum(1) / num_points.clamp(min=1).to(pillars.dtype).unsqueeze(-1)
        pts_metas = dict(pillar_center=center, pillars=pillars, pillars_num_points=num_points, pillar_coors=coors,
                         pts=points)
        if padded:
            cap = pillars.shape[0] // len(points)
            pts_metas['pillar_batch_bounds'] = [b * cap for b in range(len(points) + 1)]
        return pts_metas
