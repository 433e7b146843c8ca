from .sparse_voxelize import SPConvVoxelization  # noqa: F401
