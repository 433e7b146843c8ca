"""`plugin_dir='projects/mmdet3d_plugin/'` of the reference configs
(`projects/configs/nuscenes/Fusion_0075_refactor.py:1-2`, imported by `tools/train.py:105-118`)
resolves here: importing this package registers the MI355X-native interaction modules under
the reference's registry names.  The implementation lives in `deepinteraction_amd.mmdet3d_plugin`.
"""
from deepinteraction_amd.mmdet3d_plugin import *  # noqa: F401,F403
