from .hungarian_assigner import (BBox3DL1Cost, BBoxBEVL1Cost, HeuristicAssigner3D, HungarianAssigner3D,  # noqa: F401
                                 IoU3DCost, build_assigner)
