"""Golden vectors of the image backbone of the ++ configuration, made by running the REFERENCE'S OWN
`SwinTransformer` (models/backbones/swin.py, imported unmodified through oracle/refpin) in the build container:

    python -m oracle.refpin.make_golden_swin        # writes tests/golden/swin_t.npz

TEST INFRASTRUCTURE.  Weights and the input are regenerated from seeds by `case()` (shared with
tests/test_image_backbone.py), so only (sub-sampled) outputs are stored."""
import os

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden', 'swin_t.npz')
CFG = dict(embed_dims=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4, qkv_bias=True,
           qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, patch_norm=True, out_indices=(0, 1, 2, 3),
           with_cp=False, convert_weights=True)                 # Fusion_0075_plusplus.py:147-165
STRIDES = (4, 2, 1, 1)                                          # spatial sub-sampling of the stored stage maps


def case():
    """(backbone state in the checkpoint layout, image): 53 x 90 pixels - patch padding, window padding at every stage,
    an odd token map in front of a PatchMerging."""
    from deepinteraction_amd.mmdet3d_plugin import FrozenSwinFPN
    state, _ = FrozenSwinFPN(dtype=torch.float32).synthetic_state(seed=21)
    img = torch.randn(1, 3, 53, 90, generator=torch.Generator().manual_seed(22))
    return state, img


def subsample(feats):
    return {f'stage{i}': f[:, :, ::s, ::s].contiguous().numpy() for i, (f, s) in enumerate(zip(feats, STRIDES))}


def main():
    from oracle import refpin
    swin = refpin.load_reference_swin()
    state, img = case()
    model = swin.SwinTransformer(**CFG)
    model.eval()
    model.load_state_dict(state, strict=True)
    with torch.no_grad():
        feats = model(img)
    np.savez_compressed(OUT, **subsample([f.float() for f in feats]))
    print('wrote', OUT, {k: v.shape for k, v in subsample(feats).items()})


if __name__ == '__main__':
    main()
