from .deepinteraction import PointGlue  # noqa: F401
