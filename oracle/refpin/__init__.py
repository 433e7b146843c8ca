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
    'encoder_pp': 'projects.mmdet3d_plugin.models.necks.fusion_transformerv4',
    'decoder_pp': 'projects.mmdet3d_plugin.models.dense_heads.deepinteractionplusplus_decoder',
}

_PKGS = [
    'projects', 'projects.mmdet3d_plugin', 'projects.mmdet3d_plugin.models',
    'projects.mmdet3d_plugin.models.utils', 'projects.mmdet3d_plugin.models.utils.ops',
    'projects.mmdet3d_plugin.models.utils.ip_basic', 'projects.mmdet3d_plugin.models.necks',
    'projects.mmdet3d_plugin.models.dense_heads', 'projects.mmdet3d_plugin.core',
    'projects.mmdet3d_plugin.core.bbox', 'projects.mmdet3d_plugin.core.bbox.coders',
    'projects.mmdet3d_plugin.core.bbox.assigners',
]


def fix_zero_layer_encoders(model):
    """torch 1.9.1 (the reference's pin) runs `nn.TransformerEncoder` with zero layers as `norm(src)`; torch >= 2.0
    indexes `self.layers[0]` and raises.  Restores the 1.9.1 behaviour on the instances of a reference model
    (MMRI_I2P_Polar builds nn.Transformer(num_encoder_layers=0), fusion_transformerv4.py:490-492)."""
    import torch
    for m in model.modules():
        if isinstance(m, torch.nn.TransformerEncoder) and len(m.layers) == 0:
            m.forward = (lambda norm: (lambda src, *a, **k: norm(src)))(m.norm)
        if isinstance(m, torch.nn.TransformerDecoder):
            # 1.9.1: `for mod in layers: out = mod(out, memory, ...)`, then norm; 2.x first probes
            # `layers[0].self_attn.batch_first`, which FlashMultiheadAttention does not have
            def dec(tgt, memory, *a, _m=m, **k):
                for layer in _m.layers:
                    tgt = layer(tgt, memory)
                return _m.norm(tgt) if _m.norm is not None else tgt
            m.forward = dec
    return model


def _patch_pp(ns):
    """Three shims around the reference's ++ classes so they run on CPU under torch 2.x (nothing else is changed):
      * FlashAttention.forward (fusion_transformerv4.py:672-712) asserts CUDA + half: replaced by the same
        (b s) flattening around the stubbed flash-attn function, in fp32;
      * FlashMultiheadAttention.forward (:748-760) gains **kwargs: torch >= 2.0's TransformerDecoderLayer passes
        `is_causal=`, torch 1.9.1 (the reference's pin) did not;
      * PointRCNNBlockV2.forward ends without `return` (decoder_utils.py:1089): the local `query_feat` at exit is
        captured and returned as `(query_feat, None)`, the v1 contract the ++ decoder unpacks (:286)."""
    import torch
    pp, du = ns.encoder_pp, ns.decoder_utils
    flash = ns.stubs.flash_attn_unpadded_kvpacked_func

    def fa_forward(self, q, kv, causal=False, key_padding_mask=None):
        assert key_padding_mask is None
        B, T, S = q.shape[0], q.shape[1], kv.shape[1]
        cq = torch.arange(0, (B + 1) * T, step=T, dtype=torch.int32)
        ck = torch.arange(0, (B + 1) * S, step=S, dtype=torch.int32)
        out = flash(q.flatten(0, 1), kv.flatten(0, 1), cq, ck, T, S, 0.0, softmax_scale=self.softmax_scale,
                    causal=causal)
        return out.view(B, T, *out.shape[1:]), None
    pp.FlashAttention.forward = fa_forward
    orig = pp.FlashMultiheadAttention.forward
    pp.FlashMultiheadAttention.forward = (
        lambda self, q, k, v, key_padding_mask=None, need_weights=None, attn_mask=None, **kw:
        orig(self, q, k, v, key_padding_mask=key_padding_mask, need_weights=need_weights, attn_mask=attn_mask))
    inner = du.PointRCNNBlockV2.forward
    code = inner.__code__

    def with_return(self, *a, **k):
        box = {}

        def prof(frame, event, arg):
            if event == 'return' and frame.f_code is code:
                box['q'] = frame.f_locals.get('query_feat')
        old = sys.getprofile()
        sys.setprofile(prof)
        try:
            r = inner(self, *a, **k)
        finally:
            sys.setprofile(old)
        return r if r is not None else (box['q'], None)
    du.PointRCNNBlockV2.forward = with_return


def load_reference_swin():
    """The reference's own image backbone of the ++ configuration (`models/backbones/swin.py`, `swin_utils.py`), imported
    unmodified on the stand-ins of `stubs.install()` + `stubs.install_swin_extras()`.  Returns the `swin` module."""
    if not reference_available():
        raise FileNotFoundError(f'{REF_ROOT} not present')
    from . import stubs
    roots = stubs.STUB_ROOTS
    saved = {k: v for k, v in sys.modules.items()
             if k == 'projects' or k.startswith('projects.') or k.split('.')[0] in roots}
    for k in saved:
        del sys.modules[k]
    try:
        stubs.install('reference')
        stubs.install_swin_extras()
        for name in ('projects', 'projects.mmdet3d_plugin', 'projects.mmdet3d_plugin.models',
                     'projects.mmdet3d_plugin.models.backbones'):
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF_ROOT, *name.split('.'))]
            m.__package__ = name
            sys.modules[name] = m
        return importlib.import_module('projects.mmdet3d_plugin.models.backbones.swin')
    finally:
        for k in [k for k in sys.modules if k == 'projects' or k.startswith('projects.') or k.split('.')[0] in roots]:
            del sys.modules[k]
        sys.modules.update(saved)


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
        _patch_pp(ns)
        ns.fix_zero_layer_encoders = fix_zero_layer_encoders
        return ns
    finally:
        for k in [k for k in sys.modules
                  if k == 'projects' or k.startswith('projects.') or k.split('.')[0] in stubs.STUB_ROOTS]:
            del sys.modules[k]
        sys.modules.update(saved)
