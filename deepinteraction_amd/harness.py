"""Builders shared by bench.py, the smoke entry and the GPU tests: the hot path's two modules for a configuration
of BASELINE.json, with random-init weights of the reference architecture (there are no checkpoints here) and
non-trivial BatchNorm statistics, plus the host->device placement of a synthetic batch.  Product side only - nothing
here touches `oracle/`."""
import torch

from . import synth
from .configs import decoder_cfg

SHAPES = dict(R=synth.SHAPE_R, A=synth.SHAPE_A, TINY=synth.SHAPE_TINY)
SHAPES_PP = dict(R=synth.SHAPE_PP, A=synth.SHAPE_PP, TINY=synth.SHAPE_PP_TINY)


def randomize_bn(mods, seed=5):
    g = torch.Generator().manual_seed(seed)
    for m in mods:
        for sub in m.modules():
            if isinstance(sub, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                sub.running_mean.copy_(torch.randn(sub.running_mean.shape, generator=g) * 0.1)
                sub.running_var.copy_(torch.rand(sub.running_var.shape, generator=g) + 0.5)


def build_models(shape, num_proposals, dtype, device, seed=1234, train_cfg=None, num_layers=2):
    """(DeepInteractionEncoder, DeepInteractionDecoder) of Fusion_0075_refactor.py:185-224 at `shape`, eval mode; dtype
    float16 = the benched mixed mode of `precision.half_maps_` (fp16 maps, float32 token path)."""
    from .mmdet3d_plugin import DeepInteractionDecoder, DeepInteractionEncoder
    torch.manual_seed(seed)
    enc = DeepInteractionEncoder(num_layers=num_layers, in_channels_img=shape['c_img'], in_channels_pts=shape['c_pts'],
                                 hidden_channel=128)
    cfg = decoder_cfg(bev=shape['bev_hw'][0], num_proposals=num_proposals)
    if train_cfg is not None:
        cfg['train_cfg'] = train_cfg
    dec = DeepInteractionDecoder(**cfg)
    randomize_bn([enc, dec])
    from .precision import to_inference
    enc, dec = to_inference(enc.to(device), dec.to(device), dtype)       # fp16: maps fp16, token path float32
    return enc.eval(), dec.eval()


def condition_head(dec, branch=0.5, cross=None):
    """In place: make the random-init head a CONDITIONED one, for the parity tests that separate the kernels' error from
    the head's amplification of its input error.

    With random-init weights every RoI block of the MMPI chain (box -> RoI -> DynamicConv -> box, 4 blocks) multiplies a
    perturbation of its inputs by 2-3 (oracle, tests/tools/conditioned_head.py: an exact float32 encoder whose three maps
    are rounded ONCE to fp16 - the least any fp16-map implementation can do - already moves the last block's `dim` by
    3.7e-3 of its scale; per block the p99.9 goes 1e-4 -> 2e-4 -> 9e-4 -> 2.8e-3).  `branch` scales the three residual
    branches of every RoI block (self-attention out_proj, DynamicConv's final LayerNorm, the FFN's second Linear - for the
    ++ V2 blocks the second Linear of both mmcv FFNs): with
    0.5 the refinement is a perturbation of the residual stream, as in a trained head, and the same input error stays
    below 1e-3.  `cross` (optional) scales the q / k projections of the decoder layer's cross attention (logits |s| ~ 500 as
    initialised); it turned out NOT to be what sets the tail."""
    with torch.no_grad():
        for l, blk in enumerate(dec.decode_head):
            sfx = '' if l % 2 == 0 else '_pts'
            g = lambda n: getattr(blk, n + sfx)
            if hasattr(blk, 'linear2' + sfx):
                ffn_out = [g('linear2')]
            else:      # DeepInteraction++ V2 blocks: mmcv FFNs (`ffn`, and `self_ffn` of the parallel self branch)
                ffn_out = [blk.ffn.layers[1], blk.self_ffn.layers[1]]
            for t in [g('dyconv').norm3.weight, g('dyconv').norm3.bias, g('dyconv_pre_self_attn').out_proj.weight,
                      g('dyconv_pre_self_attn').out_proj.bias] + [x for lin in ffn_out for x in (lin.weight, lin.bias)]:
                t.mul_(branch)
        if cross is not None:
            for layer in dec.decoder:
                mha = layer.multihead_attn
                E = mha.embed_dim
                mha.in_proj_weight[:2 * E].mul_(cross)
                mha.in_proj_bias[:2 * E].mul_(cross)
    return dec


def to_device(inp, device, dtype):
    """A `synth.make_inputs` batch on `device`: channels-last feature maps in `dtype`, points / pillars as they are."""
    pm = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
    pm['pts'] = [p.to(device) for p in inp['pts_metas']['pts']]
    return dict(img_feats=inp['img_feats'].to(device, dtype).contiguous(memory_format=torch.channels_last),
                pts_feats=inp['pts_feats'].to(device, dtype).contiguous(memory_format=torch.channels_last),
                img_metas=inp['img_metas'], pts_metas=pm)


def forward(enc, dec, d):
    """One eager forward of the hot path; returns (encoder outputs, decoder output)."""
    img, pts = enc(d['img_feats'], d['pts_feats'], d['img_metas'], dict(d['pts_metas']))
    return (img, pts), dec(pts, img, d['img_metas'])


def build_models_pp(shape, num_proposals, dtype, device, seed=0, num_layers=2, train_cfg=None):
    """(FusionTransformerv4, DeepInteractionPlusPlusDecoder) of Fusion_0075_plusplus.py:210-303 (BASELINE.json configs[4])
    at `shape`, eval mode, random init - with the sampling-offset / attention-weight Linears moved off mmcv's zero init so
    that the deformable samples are spread.  dtype float16 = the mixed mode of `precision.half_maps_`: fp16 neck and
    heat-map heads, float32 token path in the head."""
    from .configs import encoder_pp_cfg
    from .mmdet3d_plugin import DeepInteractionPlusPlusDecoder, FusionTransformerv4
    torch.manual_seed(seed)
    enc = FusionTransformerv4(**encoder_pp_cfg(shape['c_img'], shape['c_pts'], num_layers))
    cfg = decoder_cfg(bev=shape['bev_hw'][0], num_proposals=num_proposals)
    if train_cfg is not None:
        cfg['train_cfg'] = train_cfg
    dec = DeepInteractionPlusPlusDecoder(**cfg)
    randomize_bn([enc, dec])
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in enc.modules():
            if hasattr(m, 'sampling_offsets'):
                for lin in (m.sampling_offsets, m.attention_weights):
                    lin.weight.add_(torch.randn(lin.weight.shape, generator=gen) * 0.05)
    from .precision import to_inference
    enc, dec = to_inference(enc.to(device), dec.to(device), dtype)
    return enc.eval(), dec.eval()


def to_device_pp(inp, device, dtype):
    """A `synth.make_inputs_pp` batch on `device`: channels-last lists of levels in `dtype`."""
    cl = lambda t: t.to(device, dtype).contiguous(memory_format=torch.channels_last)
    pm = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
    pm['pts'] = [p.to(device) for p in inp['pts_metas']['pts']]
    return dict(img_feats=[cl(f) for f in inp['img_feats']], pts_feats=[cl(f) for f in inp['pts_feats']],
                img_metas=inp['img_metas'], pts_metas=pm)
