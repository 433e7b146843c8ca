"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of hard voxelisation as the reference consumes it: `SPConvVoxelization.forward`
(/root/reference/projects/mmdet3d_plugin/models/updated_modules/sparse_voxelize.py:55-66) hands the points to
spconv's `PointToVoxel` (third-party, spconv 2.1.21 - `spconv.pytorch.utils.PointToVoxel`; NOT vendored in
/root/reference) and returns (voxels, coordinates [z,y,x], num_points).  spconv is absent from this image, so this
restates its published CPU algorithm (`points_to_voxel` "first come" loop: walk the points in order, look the
voxel up in a dense coordinate->slot table, open a new slot while fewer than max_voxels exist, append the point
while the voxel holds fewer than max_points).  PARITY UNPINNED against spconv itself: no spconv build and no
golden vectors for it exist in the reference tree; the pin is the reference's call site (argument order, output
triple, zyx coordinate order, float32 arithmetic) and the properties the consumers rely on (tests).
"""
import numpy as np


def points_to_voxel(points, voxel_size, pc_range, max_points, max_voxels, n_feat=None):
    points = np.asarray(points, dtype=np.float32)
    D = n_feat or points.shape[1]
    rng = np.asarray(pc_range, dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    grid = np.round((np.asarray(pc_range[3:], dtype=np.float64) - np.asarray(pc_range[:3], dtype=np.float64))
                    / np.asarray(voxel_size, dtype=np.float64)).astype(np.int64)          # x, y, z
    c = np.floor((points[:, :3] - rng[:3]) / vs)                                           # float32 arithmetic
    with np.errstate(invalid='ignore'):
        ok = np.all((c >= 0) & (c < grid[None].astype(np.float32)), axis=1) & ~np.isnan(points[:, :3]).any(1)
    c = np.where(ok[:, None], c, 0).astype(np.int64)
    slot_of = {}
    voxels = np.zeros((max_voxels, max_points, D), dtype=np.float32)
    coords = np.zeros((max_voxels, 3), dtype=np.int32)
    num = np.zeros((max_voxels,), dtype=np.int32)
    for i in np.nonzero(ok)[0]:
        key = (int(c[i, 2]), int(c[i, 1]), int(c[i, 0]))
        s = slot_of.get(key)
        if s is None:
            if len(slot_of) >= max_voxels:
                continue
            s = len(slot_of)
            slot_of[key] = s
            coords[s] = key
        if num[s] < max_points:
            voxels[s, num[s]] = points[i, :D]
            num[s] += 1
    P = len(slot_of)
    return voxels[:P], coords[:P], num[:P]
