"""GPU: the hot-path kernels at BASELINE.json's FULL sizes (Fusion_0075 shape R: image maps 6x128x112x200,
BEV 128x180x180; config-1 shape A: image maps 6x128x232x400), where the CPU oracle would take minutes,
through size-independent properties of the operators:

  window attention   linear in V; V == 1 gives exactly the in-image softmax mass (1 in the interior, < 1 on
                     the 4-pixel border ring where out-of-image slots score 0 and keep their mass);
                     translation equivariance in the interior; all kernel generations agree
  1x1-conv chains    linear without activations; equal to the library GEMM form
  pillar attention   a constant image gives exactly that constant in every valid cell, 0 elsewhere
  BEV gather         a constant BEV map gives that constant at every lifted pixel, 0 elsewhere
  cross attention    V == const gives const; invariant under a permutation of the keys
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from deepinteraction_amd import ops, synth
from deepinteraction_amd.geometry import SampleGeometry

DEV = 'cuda'


def _maps(n, H, W, seed, k=3):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return [torch.randn(n, 128, H, W, device=DEV, generator=g).relu().half()
            .contiguous(memory_format=torch.channels_last) for _ in range(k)]


@pytest.mark.parametrize('shape', [(6, 112, 200), (1, 180, 180), (6, 232, 400)])
def test_local_attention_full_size_properties(shape):
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    n, H, W = shape
    q, k, v = _maps(n, H, W, 0)
    sc = 1 / math.sqrt(128)
    la = lambda vv, var=ops.LA_AUTO: ops.local_attention(q, k, vv, 9, 9, sc, variant=var).float()
    out = la(v)
    # all generations of the kernel agree (fp16 P rounding in the MFMA paths: 1e-3 of the value scale)
    for var in (ops.LA_VALU, ops.LA_MFMA, ops.LA_MFMA + 2):
        d = (la(v, var) - out).abs().max().item()
        assert d <= 1e-3 * max(out.abs().max().item(), 1.0), (var, d)
    # V == 1: the softmax mass on in-image slots
    ones = torch.ones_like(v)
    mass = la(ones)
    inner = mass[:, :, 4:H - 4, 4:W - 4]
    assert (inner - 1).abs().max().item() <= 2e-3
    assert mass.max().item() <= 1 + 2e-3 and mass[:, :, 0, 0].max().item() < 0.9      # corner: 56 of 81 slots outside
    # linearity in V
    v2 = _maps(n, H, W, 1, 1)[0]
    lin = la((v * 0.5 + v2 * 0.25).half())
    ref = 0.5 * out + 0.25 * la(v2)
    assert (lin - ref).abs().max().item() <= 2e-3 * max(ref.abs().max().item(), 1.0)
    # translation equivariance (interior): shift every map by (3, 5) pixels
    sh = lambda t: torch.roll(t, shifts=(3, 5), dims=(2, 3)).contiguous(memory_format=torch.channels_last)
    out_s = ops.local_attention(sh(q), sh(k), sh(v), 9, 9, sc).float()
    a, b = out_s[:, :, 12:H - 12, 16:W - 16], torch.roll(out, shifts=(3, 5), dims=(2, 3))[:, :, 12:H - 12, 16:W - 16]
    assert (a - b).abs().max().item() <= 1e-3 * max(out.abs().max().item(), 1.0)


@pytest.mark.parametrize('shape', [(6, 112, 200), (6, 232, 400)])
def test_pointwise_chain_full_size(shape):
    n, H, W = shape
    x1, x2, x3 = _maps(n, H, W, 2)
    g = torch.Generator(device=DEV).manual_seed(3)
    w = lambda kk: (torch.randn(128, kk, device=DEV, generator=g) / math.sqrt(kk)).half()
    b1, b2 = torch.randn(128, device=DEV, generator=g), torch.randn(128, device=DEV, generator=g)
    w1, w2 = w(256), w(256)
    got = ops.pointwise_chain(x1, w1, b1, False, x2=x2, w2=w2, b2=b2, relu2=False, x3=x3).float()
    flat = lambda t: t.permute(0, 2, 3, 1).reshape(-1, 128)
    h = (F.linear(torch.cat([flat(x1), flat(x2)], 1).float(), w1.float(), b1)).half()
    ref = F.linear(torch.cat([h, flat(x3)], 1).float(), w2.float(), b2).view(n, H, W, 128).permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    # without activations the chain is affine: f(a) + f(b) - f(0) = f(a + b)
    z = torch.zeros_like(x1)
    f = lambda a: ops.pointwise_chain(a, w1[:, :128].contiguous(), b1, False, w2=w2[:, :128].contiguous(), b2=b2).float()
    s = f((x1 * 0.5 + x2 * 0.5).half())
    assert (s - (0.5 * f(x1) + 0.5 * f(x2))).abs().max().item() <= 4e-3 * s.abs().max().item()
    assert f(z).std(dim=(0, 2, 3)).max().item() == 0                              # a constant per channel


def test_gathers_full_size_constants():
    shape = synth.SHAPE_R
    inp = synth.make_inputs(1, shape, seed=0)
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    geom = SampleGeometry(inp['img_metas'][0], (Hi, Wi), DEV)
    pm = inp['pts_metas']
    # pillar attention on a constant image: softmax weights sum to one
    img = torch.full((6, 128, Hi, Wi), 0.75, device=DEV, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    qf = _maps(1, Hb, Wb, 4, 1)[0]
    ctx, valid = ops.i2p_attention(img, qf, pm['pillars'].to(DEV), pm['pillar_coors'].to(DEV),
                                   pm['pillars_num_points'].to(DEV), geom.lidar2img, geom.aug_rev, geom.ori_hw)
    v = valid[0, 0] > 0
    coor = pm['pillar_coors'].long()
    occupied = torch.zeros(Hb, Wb, dtype=torch.bool)
    occupied[coor[:, 2], coor[:, 3]] = True
    assert not (v.cpu() & ~occupied).any() and v.sum() > 0.5 * occupied.sum()
    c = ctx[0].float()
    # a key whose bilinear footprint hangs over the image edge samples zero padding: <= 0.75 there
    assert c[:, v].max().item() <= 0.75 + 1e-3 and c[:, ~v].abs().max().item() == 0
    assert (c[:, v] > 0.75 - 1e-3).float().mean().item() > 0.9
    # BEV gather of a constant map
    depth = ops.depth_complete(ops.depth_scatter(pm['pts'][0].to(DEV), geom.lidar2img, geom.aug_rev, Hi, Wi, geom.ori_hw))
    bev = torch.full((1, 128, Hb, Wb), 2.0, device=DEV, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    w = ops.bevwarp_gather(bev, depth, geom.img2lidar, geom.aug_fwd, geom.xs, geom.ys, geom.pc_range).float()
    assert w.max().item() <= 2.0 + 1e-3 and w.min().item() >= 0
    assert ((w - 2.0).abs() < 1e-3).float().mean().item() > 0.5                  # most pixels lift inside the range
    assert (w.amax(1) - w.amin(1)).max().item() <= 1e-3                          # every texel is one scalar x 128


def test_cross_attention_full_size_properties():
    g = torch.Generator(device=DEV).manual_seed(5)
    q = torch.randn(1, 200, 128, device=DEV, generator=g).half()
    kv = torch.randn(1, 32400, 256, device=DEV, generator=g).half()
    out = ops.mha_decode(q, kv, 8, 0.25).float()
    perm = torch.randperm(32400, device=DEV, generator=g)
    out_p = ops.mha_decode(q, kv[:, perm].contiguous(), 8, 0.25).float()
    assert (out - out_p).abs().max().item() <= 2e-3
    kvc = kv.clone()
    kvc[..., 128:] = 0.5
    assert (ops.mha_decode(q, kvc, 8, 0.25).float() - 0.5).abs().max().item() <= 1e-3
