"""Import the REFERENCE's own Python for the hot path, on CPU, in the build container
(TEST INFRASTRUCTURE).

/root/reference cannot be imported as is: mmcv / mmdet / mmdet3d / detectron2 / cv2
are not installed and `locatt_ops/__init__.py:11` asserts CUDA.  `load_reference()`
installs minimal stand-ins for exactly the third-party names the hot-path files
import (oracle/refpin/stubs.py - restated third-party semantics, parity unpinned
there) and binds `locatt_ops.localattention` to the reference's OWN kernels.cuh
compiled for the host (oracle/_ref/liblocatt_ref.so).  The reference modules are
then executed unmodified from where they lie; nothing is copied.

Used by tests/test_oracle_vs_reference.py (skipped when /root/reference is absent,
e.g. on the GPU box) and by oracle/refpin/make_golden.py, which writes the golden
vectors committed under tests/golden/.
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get('DI_REFERENCE_ROOT', '/root/reference')

_HOT_MODULES = {
    'encoder_utils': 'projects.mmdet3d_plugin.models.utils.encoder_utils',
    'depth_map_utils': 'projects.mmdet3d_plugin.models.utils.ip_basic.depth_map_utils',
    'encoder': 'projects.mmdet3d_plugin.models.necks.deepinteraction_encoder',
    'decoder_utils': 'projects.mmdet3d_plugin.models.utils.decoder_utils',
    'decoder': 'projects.mmdet3d_plugin.models.dense_heads.deepinteraction_decoder',
    'bbox_coder': 'projects.mmdet3d_plugin.core.bbox.coders.transfusion_bbox_coder',
    'assigner': 'projects.mmdet3d_plugin.core.bbox.assigners.hungarian_assigner',
}

_PKGS = [
    'projects', 'projects.mmdet3d_plugin', 'projects.mmdet3d_plugin.models',
    'projects.mmdet3d_plugin.models.utils', 'projects.mmdet3d_plugin.models.utils.ops',
    'projects.mmdet3d_plugin.models.utils.ip_basic', 'projects.mmdet3d_plugin.models.necks',
    'projects.mmdet3d_plugin.models.dense_heads', 'projects.mmdet3d_plugin.core',
    'projects.mmdet3d_plugin.core.bbox', 'projects.mmdet3d_plugin.core.bbox.coders',
    'projects.mmdet3d_plugin.core.bbox.assigners',
]


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, 'projects', 'mmdet3d_plugin'))


def load_reference(locatt_kind='reference'):
    """Returns a namespace of the reference's hot-path modules (see _HOT_MODULES)."""
    if not reference_available():
        raise FileNotFoundError(f'{REF_ROOT} not present')
    from . import stubs
    saved = {k: v for k, v in sys.modules.items()
             if k == 'projects' or k.startswith('projects.') or k.split('.')[0] in stubs.STUB_ROOTS}
    for k in saved:
        del sys.modules[k]
    try:
        stubs.install(locatt_kind)
        for name in _PKGS:                      # package shells: no __init__.py is executed
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF_ROOT, *name.split('.'))]
            m.__package__ = name
            sys.modules[name] = m
        sys.modules['projects.mmdet3d_plugin.models.utils.ops.locatt_ops'] = stubs.make_locatt_module(locatt_kind)
        ns = types.SimpleNamespace()
        for short, full in _HOT_MODULES.items():
            setattr(ns, short, importlib.import_module(full))
        ns.stubs = stubs
        return ns
    finally:
        for k in [k for k in sys.modules
                  if k == 'projects' or k.startswith('projects.') or k.split('.')[0] in stubs.STUB_ROOTS]:
            del sys.modules[k]
        sys.modules.update(saved)
