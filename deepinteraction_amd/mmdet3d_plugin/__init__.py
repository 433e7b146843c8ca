"""Host-side mirror of the reference plugin surface for the interaction hot path
(`projects/mmdet3d_plugin/__init__.py:1-10`), limited to the hot path's modules."""
from .models.necks.deepinteraction_encoder import DeepInteractionEncoder  # noqa: F401

__all__ = ['DeepInteractionEncoder']
