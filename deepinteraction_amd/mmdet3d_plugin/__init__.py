"""Host-side mirror of the reference plugin surface for the interaction hot path
(`projects/mmdet3d_plugin/__init__.py:1-10`), limited to the hot path's modules."""
from .core.bbox.assigners import (BBox3DL1Cost, BBoxBEVL1Cost, HeuristicAssigner3D, HungarianAssigner3D,  # noqa: F401
                                  IoU3DCost)
from .core.bbox.coders.transfusion_bbox_coder import TransFusionBBoxCoder  # noqa: F401
from .models.dense_heads.deepinteraction_decoder import DeepInteractionDecoder  # noqa: F401
from .models.dense_heads.deepinteractionplusplus_decoder import DeepInteractionPlusPlusDecoder  # noqa: F401
from .models.necks.deepinteraction_encoder import DeepInteractionEncoder  # noqa: F401
from .models.necks.fusion_transformerv4 import (MMRI_I2P, MMRI_P2I, DeepInteractionLayer, FusionTransformerv4,  # noqa: F401
                                                MMRI_I2P_Polar)
from .models.detectors import (DeepInteractionInference, FrozenLidarBackbone, FrozenResNetFPN, FrozenSECOND,  # noqa: F401
                               FrozenSECONDFPN, FrozenSparseEncoder, FrozenSwinFPN, ImageGlue, PointGlue, bbox3d2result)
from .models.updated_modules import SPConvVoxelization  # noqa: F401

__all__ = ['DeepInteractionEncoder', 'DeepInteractionDecoder', 'FusionTransformerv4', 'DeepInteractionLayer',
           'MMRI_P2I', 'MMRI_I2P', 'MMRI_I2P_Polar', 'DeepInteractionPlusPlusDecoder', 'SPConvVoxelization', 'PointGlue', 'ImageGlue', 'FrozenResNetFPN', 'FrozenSwinFPN', 'FrozenLidarBackbone', 'FrozenSparseEncoder', 'FrozenSECOND', 'FrozenSECONDFPN', 'DeepInteractionInference', 'bbox3d2result', 'TransFusionBBoxCoder', 'HungarianAssigner3D',
           'HeuristicAssigner3D', 'BBox3DL1Cost', 'BBoxBEVL1Cost', 'IoU3DCost']
