"""The frozen image feature extractor in front of the hot path (`FrozenResNetFPN` / `ImageGlue`) against the oracle's
unfused restatement of mmdet's ResNet-50 + FPN (`oracle/image_backbone.py`; parity unpinned - mmdet is not importable
here, the anchor is the checkpoint key layout)."""
import pytest
import torch

from deepinteraction_amd.mmdet3d_plugin import FrozenResNetFPN, ImageGlue
from oracle import image_backbone as ob


@pytest.fixture(scope='module')
def nets():
    torch.manual_seed(0)
    return ob.randomize(ob.ResNet(50), 1).eval(), ob.randomize(ob.FPN(), 2).eval()


def test_checkpoint_key_layout(nets):
    backbone, neck = nets
    sd = backbone.state_dict()
    assert len(sd) == 318                       # torchvision resnet50: 320 entries, minus fc.weight / fc.bias
    for k in ('conv1.weight', 'bn1.running_var', 'layer1.0.downsample.0.weight', 'layer1.0.downsample.1.running_mean',
              'layer3.5.conv3.weight', 'layer4.2.bn3.num_batches_tracked'):
        assert k in sd, k
    assert 'layer1.1.downsample.0.weight' not in sd
    assert sd['layer2.0.conv2.weight'].shape == (128, 128, 3, 3)
    assert sd['layer4.0.downsample.0.weight'].shape == (2048, 1024, 1, 1)
    nd = neck.state_dict()
    assert sorted(nd) == sorted(f'{kind}.{i}.conv.{p}' for kind in ('lateral_convs', 'fpn_convs') for i in range(4)
                                for p in ('weight', 'bias'))


@pytest.mark.parametrize('levels', [None, (0,), (0, 1), (2, 4)])
def test_folded_network_equals_unfused_oracle(nets, levels):
    backbone, neck = nets
    net = FrozenResNetFPN(levels=levels, dtype=torch.float32).load_mmdet_state(backbone.state_dict(), neck.state_dict())
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 3, 96, 160, generator=g)
    with torch.no_grad():
        want = neck(backbone(img))
    got = net(img)
    idx = range(5) if levels is None else levels
    assert len(got) == len(idx)
    for o, l in zip(got, idx):
        w = want[l]
        assert o.shape == w.shape
        assert o.is_contiguous(memory_format=torch.channels_last) or o.shape[-1] == 1 or o.shape[-2] == 1
        scale = float(w.abs().max())
        assert float((o - w).abs().max()) <= 2e-5 * max(scale, 1.0), (l, float((o - w).abs().max()), scale)


def test_pruned_levels_skip_unused_convolutions(nets, monkeypatch):
    backbone, neck = nets
    net = FrozenResNetFPN(levels=(0,), dtype=torch.float32).load_mmdet_state(backbone.state_dict(), neck.state_dict())
    seen = []
    orig = net._conv
    monkeypatch.setattr(net, '_conv', lambda x, name, *a, **k: (seen.append(name), orig(x, name, *a, **k))[1])
    net(torch.zeros(1, 3, 64, 64))
    assert [n for n in seen if n.startswith('fpn_convs')] == ['fpn_convs.0']
    assert [n for n in seen if n.startswith('lateral')] == [f'lateral_convs.{i}' for i in range(4)]
    assert len([n for n in seen if n.startswith(('conv1', 'layer'))]) == 53


def test_fp16_channels_last_form(nets):
    backbone, neck = nets
    net = FrozenResNetFPN(levels=(0,)).load_mmdet_state(backbone.state_dict(), neck.state_dict())
    w, b = net._get('layer2.0.conv2')
    assert w.dtype == torch.float16 and w.is_contiguous(memory_format=torch.channels_last) and b.dtype == torch.float16
    assert not net.state_dict()                 # derived buffers: the checkpoint stays the mmdet one


def test_loading_fails_loudly(nets):
    backbone, neck = nets
    sd = dict(backbone.state_dict())
    del sd['layer3.2.bn2.running_var']
    with pytest.raises(KeyError, match='layer3.2.bn2.running_var'):
        FrozenResNetFPN().load_mmdet_state(sd, neck.state_dict())
    nd = dict(neck.state_dict())
    nd['fpn_convs.1.conv.weight'] = nd['fpn_convs.1.conv.weight'][:, :128]
    with pytest.raises(ValueError, match='fpn_convs.1.conv.weight'):
        FrozenResNetFPN().load_mmdet_state(backbone.state_dict(), nd)
    with pytest.raises(RuntimeError, match='no weights'):
        FrozenResNetFPN()(torch.zeros(1, 3, 32, 32))


def test_image_glue_mirrors_extract_img_feat(nets):
    backbone, neck = nets
    net = FrozenResNetFPN(levels=(0,), dtype=torch.float32).load_mmdet_state(backbone.state_dict(), neck.state_dict())
    glue = ImageGlue(net)
    metas = [dict(), dict()]
    img = torch.randn(2, 3, 3, 64, 96, generator=torch.Generator().manual_seed(5))
    (lvl0,) = glue(img, metas)
    assert lvl0.shape == (6, 256, 16, 24)
    assert all(m['input_shape'] == (64, 96) for m in metas)
    (flat,) = glue(img.view(6, 3, 64, 96), [dict()])
    assert torch.equal(flat, lvl0)
    assert glue(None, metas) is None


def test_synthetic_state_has_the_checkpoint_layout(nets):
    backbone, neck = nets
    net = FrozenResNetFPN(levels=(0,), dtype=torch.float32)
    bb, nk = net.synthetic_state(seed=4)
    ref = {k: v for k, v in backbone.state_dict().items() if not k.endswith('num_batches_tracked')}
    assert sorted(bb) == sorted(ref) and all(bb[k].shape == ref[k].shape for k in bb)
    assert sorted(nk) == sorted(neck.state_dict())
    (out,) = net.load_mmdet_state(bb, nk)(torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(6)))
    assert torch.isfinite(out).all() and 1e-3 < float(out.abs().max()) < 1e3       # fp16-safe magnitudes


@pytest.mark.gpu
def test_fp16_channels_last_on_the_device_against_the_oracle():
    """fp16 storage and MIOpen's fp16 convolutions against the fp32 oracle on the host; 53 folded layers deep, so the
    bound is a fraction of the map's magnitude rather than the 1e-3 of the hot path's own contract."""
    net = FrozenResNetFPN(levels=(0, 1))
    bb, nk = net.synthetic_state(seed=8)            # activations of O(1): the `nets` fixture's grow past fp16's range
    backbone, neck = ob.ResNet(50), ob.FPN()
    assert not backbone.load_state_dict(bb, strict=False).unexpected_keys and not neck.load_state_dict(nk).missing_keys
    net.load_mmdet_state(bb, nk).cuda()
    img = torch.randn(2, 3, 128, 160, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        want = neck(backbone(img))
    got = net(img.cuda())
    torch.cuda.synchronize()
    for o, l in zip(got, (0, 1)):
        assert o.dtype == torch.float16 and o.is_contiguous(memory_format=torch.channels_last)
        assert torch.isfinite(o).all()
        err = float((o.float().cpu() - want[l]).abs().max())
        assert err <= 3e-2 * float(want[l].abs().max()), (l, err)


# ---------------------------------------------------------------------------------------------------------------------
# DeepInteraction++ image side: the plugin's Swin-T (reference models/backbones/swin.py) + FPN
# ---------------------------------------------------------------------------------------------------------------------
import os

import numpy as np

from deepinteraction_amd.mmdet3d_plugin import FrozenSwinFPN
from oracle import refpin
from oracle.refpin import make_golden_swin as mgs

GOLDEN_SWIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'swin_t.npz')


def test_swin_stage_maps_against_the_golden_vectors_of_the_reference():
    """tests/golden/swin_t.npz was written by the REFERENCE'S OWN SwinTransformer (oracle/refpin/make_golden_swin.py): an
    image whose token map needs padding in the patch embedding, in every stage's windows and in a PatchMerging."""
    state, img = mgs.case()
    net = FrozenSwinFPN(dtype=torch.float32)
    net.load_mmdet_state(state, net.synthetic_state(0)[1])
    got = mgs.subsample([f.float() for f in net.stage_maps(img)])
    gold = np.load(GOLDEN_SWIN)
    assert sorted(gold.files) == sorted(got)
    for k in gold.files:
        assert got[k].shape == gold[k].shape, k
        err = float(np.abs(got[k] - gold[k]).max())
        assert err <= 2e-5 * max(1.0, float(np.abs(gold[k]).max())), (k, err)


@pytest.mark.skipif(not refpin.reference_available(), reason='/root/reference not present (e.g. on the GPU box)')
@pytest.mark.parametrize('hw', [(70, 101), (64, 112)])
def test_swin_fpn_against_the_reference_module(hw):
    """Direct comparison with the reference's SwinTransformer imported unmodified (all levels; the FPN behind it is the
    oracle's restatement of mmdet's - parity unpinned there)."""
    swin = refpin.load_reference_swin()
    net = FrozenSwinFPN(levels=None, dtype=torch.float32)
    bb, nk = net.synthetic_state(3)
    ref = swin.SwinTransformer(**mgs.CFG)
    ref.eval()
    ref.load_state_dict(bb, strict=True)                    # the product's synthetic state IS a reference checkpoint
    neck = ob.FPN(in_channels=(96, 192, 384, 768)).eval()
    neck.load_state_dict(nk)
    net.load_mmdet_state(bb, nk)
    img = torch.randn(2, 3, *hw, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        want = neck(ref(img))
    got = net(img)
    assert len(got) == 5
    for g, w in zip(got, want):
        assert g.shape == w.shape
        assert float((g - w).abs().max()) <= 2e-5 * max(1.0, float(w.abs().max()))


def test_swin_pruned_levels_and_loading_errors():
    net = FrozenSwinFPN(dtype=torch.float32)                # the ++ neck reads levels 0 and 1
    bb, nk = net.synthetic_state(5)
    net.load_mmdet_state(bb, nk)
    full = FrozenSwinFPN(levels=None, dtype=torch.float32).load_mmdet_state(bb, nk)
    img = torch.randn(1, 3, 56, 84, generator=torch.Generator().manual_seed(1))
    a, b = net(img), full(img)
    assert len(a) == 2 and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert a[0].is_contiguous(memory_format=torch.channels_last) and a[0].shape == (1, 256, 14, 21)
    assert not net.state_dict()
    bad = dict(bb)
    del bad['stages.2.blocks.4.attn.w_msa.qkv.bias']
    with pytest.raises(KeyError, match='stages.2.blocks.4.attn.w_msa.qkv.bias'):
        FrozenSwinFPN().load_mmdet_state(bad, nk)
    with pytest.raises(RuntimeError, match='no weights'):
        FrozenSwinFPN()(img)


def test_swin_fp16_form_on_the_host_stays_close():
    """fp16 weights and activations (the device form) against the float32 form of the same network: the error budget of
    12 blocks in half precision, checked on the host so that the device test only has to confirm the kernels."""
    bb, nk = FrozenSwinFPN().synthetic_state(7)
    f32 = FrozenSwinFPN(dtype=torch.float32).load_mmdet_state(bb, nk)
    f16 = FrozenSwinFPN(dtype=torch.float16).load_mmdet_state(bb, nk)
    img = torch.randn(1, 3, 56, 84, generator=torch.Generator().manual_seed(2))
    want = f32(img)
    try:
        got = f16(img)
    except RuntimeError as e:                               # a CPU build without half kernels for some op
        pytest.skip(f'half precision not runnable on this host: {e}')
    for g, w in zip(got, want):
        assert float((g.float() - w).abs().max()) <= 3e-2 * float(w.abs().max())


@pytest.mark.gpu
def test_swin_fp16_on_the_device():
    bb, nk = FrozenSwinFPN().synthetic_state(7)
    f32 = FrozenSwinFPN(dtype=torch.float32).load_mmdet_state(bb, nk)
    dev = FrozenSwinFPN().load_mmdet_state(bb, nk).cuda()
    img = torch.randn(2, 3, 70, 101, generator=torch.Generator().manual_seed(2))
    want = f32(img)
    got = dev(img.cuda())
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert g.dtype == torch.float16 and g.is_contiguous(memory_format=torch.channels_last)
        assert torch.isfinite(g).all()
        assert float((g.float().cpu() - w).abs().max()) <= 3e-2 * float(w.abs().max())


@pytest.mark.gpu
def test_one_pass_bias_residual_relu_epilogue_equals_the_torch_statements():
    """`ops.bias_act_` (csrc/epilogue.hip) against y + b (+ z) -> relu in float32 of the same fp16 inputs, and the frozen ResNet-50 +
    FPN with it against the same network on torch's conv-with-bias / add_ / relu_ passes (fp16 round-off of a 50-layer network)."""
    from deepinteraction_amd import ops
    from deepinteraction_amd.mmdet3d_plugin.models.detectors import image_glue as ig
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    for (n, C, H, W) in ((2, 64, 13, 9), (1, 256, 28, 50), (6, 8, 5, 7)):
        y = torch.randn(n, C, H, W, generator=g).half().to(dev).contiguous(memory_format=torch.channels_last)
        z = torch.randn(n, C, H, W, generator=g).half().to(dev).contiguous(memory_format=torch.channels_last)
        b = torch.randn(C, generator=g).to(dev)
        for res, relu in ((None, True), (z, True), (z, False), (None, False)):
            want = y.float() + b.view(1, -1, 1, 1) + (0 if res is None else res.float())
            want = want.relu() if relu else want
            got = ops.bias_act_(y.clone(memory_format=torch.channels_last), b, res, relu)
            assert got.is_contiguous(memory_format=torch.channels_last)
            assert (got.float() - want).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item())
    for (Hl, Wl, Hh, Wh) in ((56, 100, 28, 50), (7, 9, 4, 5), (5, 5, 5, 5), (9, 4, 3, 1)):
        lo = torch.randn(2, 16, Hl, Wl, generator=g).half().to(dev).contiguous(memory_format=torch.channels_last)
        hi = torch.randn(2, 16, Hh, Wh, generator=g).half().to(dev).contiguous(memory_format=torch.channels_last)
        want = lo + torch.nn.functional.interpolate(hi, size=(Hl, Wl), mode='nearest')
        assert torch.equal(ops.upsample_add_(lo.clone(memory_format=torch.channels_last), hi), want)        # one fp16 addition: exact
    net = ig.FrozenResNetFPN(levels=(0,), dtype=torch.float16)
    net.load_mmdet_state(*net.synthetic_state(1))
    net = net.to(dev).eval()
    img = torch.randn(2, 3, 96, 160, generator=g).to(dev)
    a = net(img)[0].float()
    net.FUSED = False
    bref = net(img)[0].float()
    net.FUSED = True
    assert (a - bref).abs().max().item() <= 2e-2 * max(1.0, bref.abs().max().item())
    # the stem: bias + ReLU behind the max-pool instead of in front of it - exact
    w, bias = net._get('conv1')
    y = torch.nn.functional.conv2d(img.half().contiguous(memory_format=torch.channels_last), w, None, stride=2, padding=3)
    front = torch.nn.functional.max_pool2d(torch.relu(y + bias.view(1, -1, 1, 1)), 3, 2, 1)
    behind = torch.relu(torch.nn.functional.max_pool2d(y, 3, 2, 1) + bias.view(1, -1, 1, 1))
    assert torch.equal(front, behind)
