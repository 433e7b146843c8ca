"""Builders shared by bench.py, the smoke entry and the GPU tests: the hot path's two modules for a configuration
of BASELINE.json, with random-init weights of the reference architecture (there are no checkpoints here) and
non-trivial BatchNorm statistics, plus the host->device placement of a synthetic batch.  Product side only - nothing
here touches `oracle/`."""
import torch

from . import synth
from .configs import decoder_cfg

SHAPES = dict(R=synth.SHAPE_R, A=synth.SHAPE_A, TINY=synth.SHAPE_TINY)


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
