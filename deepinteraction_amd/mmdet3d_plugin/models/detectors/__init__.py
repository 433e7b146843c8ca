from .deepinteraction import PointGlue  # noqa: F401
from .image_glue import FrozenResNetFPN, ImageGlue  # noqa: F401
