from .deepinteraction import PointGlue  # noqa: F401
from .image_glue import FrozenResNetFPN, FrozenSwinFPN, ImageGlue  # noqa: F401
from .inference import DeepInteractionInference, bbox3d2result  # noqa: F401
from .lidar_glue import FrozenLidarBackbone, FrozenSECOND, FrozenSECONDFPN, FrozenSparseEncoder  # noqa: F401
