"""`SPConvVoxelization` - drop-in for reference `models/updated_modules/sparse_voxelize.py:9-70` (same constructor
arguments and `forward(points) -> (voxels, coordinates, num_points)`), without spconv: the hard voxelisation runs
on the MI355X (deepinteraction_amd/csrc/voxelize.hip).  Voxel order and the surviving points of crowded voxels
follow the CPU "first come" rule (spconv's GPU hash order is unspecified; the consumers are permutation
invariant).  `forward_padded` keeps the capacity-sized buffers and a device-side count (no host
synchronisation: what a captured graph wants, see deepinteraction_amd/graphed.py)."""
import numpy as np
import torch
from torch import nn
from torch.nn.modules.utils import _pair

from .... import ops


class SPConvVoxelization(nn.Module):
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels, num_point_features=5,
                 device=torch.device('cuda')):
        super().__init__()
        assert len(voxel_size) == 3 and len(point_cloud_range) == 6
        self.voxel_size = np.array(voxel_size)
        self.point_cloud_range = np.array(point_cloud_range)
        self.max_num_points = max_num_points
        self.num_point_features = num_point_features
        self.device = device
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else _pair(max_voxels)
        grid_size = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(voxel_size)
        self.grid_size = np.round(grid_size).astype(np.int64)

    def _cap(self):
        return self.max_voxels[0] if self.training else self.max_voxels[1]     # (train, test), reference :33-53

    def forward_padded(self, points):
        return ops.voxelize(points.contiguous().float(), self.voxel_size.tolist(), self.point_cloud_range.tolist(),
                            self.max_num_points, self._cap(), self.num_point_features)

    @torch.no_grad()
    def forward(self, points):
        voxels, coords, num, n = self.forward_padded(points)
        n = int(n)                                         # exact shapes, as spconv returns them: one sync
        return voxels[:n].clone(), coords[:n].clone(), num[:n].clone()

    def __repr__(self):
        return (f'{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range={self.point_cloud_range}, '
                f'max_num_points={self.max_num_points}, max_voxels={self.max_voxels}, '
                f'num_point_features={self.num_point_features})')
