"""Pin the oracle restatement against the REFERENCE'S OWN PYTHON, imported unmodified from
/root/reference with stand-ins for the absent third-party packages (oracle/refpin).

Runs only where /root/reference exists (the build container); on the GPU box the committed
golden vectors (tests/golden, tests/test_golden.py) carry the same pin."""
import numpy as np
import pytest
import torch

from oracle.refpin import load_reference, reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason='/root/reference not present')

from deepinteraction_amd import synth  # noqa: E402
from oracle import configs, decoder as odec, depth_completion, encoder as oenc  # noqa: E402


@pytest.fixture(scope='module')
def ref(oracle_libs):
    return load_reference('reference')


def _randomize_bn(m):
    for mod in m.modules():
        if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)


@pytest.mark.parametrize('aug', [False, True])
def test_encoder_matches_reference(ref, aug):
    shape = synth.SHAPE_TINY
    inp = synth.make_inputs(1, shape, seed=0, aug=synth.example_aug(0) if aug else None)
    torch.manual_seed(1234)
    R = ref.encoder.DeepInteractionEncoder(num_layers=2, in_channels_img=shape['c_img'],
                                           in_channels_pts=shape['c_pts'], hidden_channel=128)
    _randomize_bn(R)
    O = oenc.DeepInteractionEncoder(2, shape['c_img'], shape['c_pts'], 128)
    O.load_state_dict(R.state_dict())
    R.eval(), O.eval()
    with torch.no_grad():
        ri, (rp0, rp1) = R(inp['img_feats'], inp['pts_feats'], inp['img_metas'], inp['pts_metas'])
        oi, (op0, op1) = O(inp['img_feats'], inp['pts_feats'], inp['img_metas'], inp['pts_metas'])
    for a, b in ((ri, oi), (rp0, op0), (rp1, op1)):
        assert torch.allclose(a, b, rtol=0, atol=2e-6), (a - b).abs().max()
    assert ri.abs().max() > 0.1 and rp1.abs().max() > 0.1


def test_i2p_group_attn_equals_masked_attention(ref):
    """The reference's 4-bucket `group_attn` (encoder_utils.py:226-255) vs the plain masked
    attention the oracle/kernels implement, two-sample batch."""
    shape = synth.SHAPE_TINY
    inp = synth.make_inputs(2, shape, seed=3)
    torch.manual_seed(0)
    R = ref.encoder_utils.MMRI_I2P(128, 128, 0.1).eval()
    O = oenc.MMRI_I2P(128, 128, 0.1).eval()
    O.load_state_dict(R.state_dict())
    g = torch.Generator().manual_seed(1)
    bev = torch.randn(2, 128, *shape['bev_hw'], generator=g)
    img = torch.randn(2, 6, 128, *shape['img_hw'], generator=g)
    with torch.no_grad():
        a = R(bev, img, inp['img_metas'], inp['pts_metas'])
        b = O(bev, img, inp['img_metas'], inp['pts_metas'])
    assert torch.allclose(a, b, atol=2e-6), (a - b).abs().max()
    assert (a != 0).any()


def test_fill_in_multiscale_matches_reference_control_flow(ref):
    """The reference's depth_map_utils.fill_in_multiscale (its own control flow; cv2 calls go to
    the scipy stand-ins) vs oracle/depth_completion.py: bit-exact."""
    shape = synth.SHAPE_TINY
    inp = synth.make_inputs(1, shape, seed=2)
    sparse = oenc.BEVWarp().sparse_depth(inp['pts_metas']['pts'][0], inp['img_metas'][0],
                                         oenc.lidar2img_tensor(inp['img_metas'], torch.zeros(1))[0],
                                         *shape['img_hw'])
    rng = np.random.default_rng(0)
    for v in range(6):
        d = sparse[v].numpy().copy()
        d[rng.random(d.shape) < 0.6] = 0
        a, _ = ref.depth_map_utils.fill_in_multiscale(d.copy(), extrapolate=False, blur_type='bilateral')
        b = depth_completion.fill_in_multiscale(d.copy())
        assert np.array_equal(a, b)
        assert (b > 0).mean() > (d > 0).mean()


def test_bbox_coder_matches_reference(ref):
    kw = dict(pc_range=[-54.0, -54.0], out_size_factor=8, voxel_size=[0.075, 0.075], code_size=10)
    R = ref.bbox_coder.TransFusionBBoxCoder(**kw)
    O = odec.TransFusionBBoxCoder(**kw)
    g = torch.Generator().manual_seed(0)
    hm, rot, dim = torch.randn(2, 10, 7, generator=g), torch.randn(2, 2, 7, generator=g), torch.randn(2, 3, 7, generator=g)
    cen, hei, vel = torch.rand(2, 2, 7, generator=g) * 180, torch.randn(2, 1, 7, generator=g), torch.randn(2, 2, 7, generator=g)
    a = R.decode(hm.clone(), rot.clone(), dim.clone(), cen.clone(), hei.clone(), vel.clone())
    b = O.decode(hm, rot, dim, cen, hei, vel)
    for x, y in zip(a, b):
        for k in ('bboxes', 'scores', 'labels'):
            assert torch.equal(x[k], y[k])
    boxes = torch.rand(5, 9, generator=g) + 0.5
    assert torch.equal(R.encode(boxes), O.encode(boxes))
    # encode o decode round trip (known-answer)
    t = O.encode(boxes)
    d = O.decode(torch.zeros(1, 10, 5), t[:, 6:8].T[None], t[:, 3:6].T[None], t[:, 0:2].T[None], t[:, 2:3].T[None],
                 t[:, 8:10].T[None])[0]['bboxes']
    assert torch.allclose(d[:, :6], boxes[:, :6], atol=1e-4)


def _narrow_metas(metas):
    """All six cameras look forward: queries behind the ego are on NO image (fallback path),
    and with few queries some views see <= 1 query (skip path)."""
    import copy
    metas = copy.deepcopy(metas)
    for m in metas:
        m['lidar2img'] = [m['lidar2img'][0]] * 2 + [m['lidar2img'][1]] * 2 + [m['lidar2img'][2]] * 2
    return metas


@pytest.mark.parametrize('narrow', [False, True])
def test_decoder_matches_reference(ref, narrow):
    shape = synth.SHAPE_TINY
    cfg = configs.decoder_cfg(bev=36, num_proposals=24 if narrow else 40)
    torch.manual_seed(7)
    R = ref.decoder.DeepInteractionDecoder(**cfg)
    _randomize_bn(R)
    for n, p in R.named_parameters():
        if n.endswith('bias'):
            p.data.normal_(0, 0.05)
    O = odec.DeepInteractionDecoder(**cfg)
    O.load_state_dict(R.state_dict())
    R.eval(), O.eval()
    g = torch.Generator().manual_seed(0)
    Hi, Wi = shape['img_hw']
    p0, p1 = torch.randn(2, 128, 36, 36, generator=g), torch.randn(2, 128, 36, 36, generator=g)
    img = torch.randn(12, 128, Hi, Wi, generator=g)
    metas = synth.make_inputs(2, shape, seed=0)['img_metas']
    if narrow:
        metas = _narrow_metas(metas)
    with torch.no_grad():
        r = R([p0, p1], img, metas)[0][0]
        o = O([p0, p1], img, metas)[0][0]
    for k in r:
        assert torch.allclose(r[k], o[k], rtol=0, atol=1e-5), (k, (r[k] - o[k]).abs().max())
    assert torch.equal(R.query_labels, O.query_labels)          # INT: bit-exact
    for a, b in zip(R.on_the_image_mask, O.on_the_image_mask):
        assert torch.equal(a, b)
    if narrow:
        assert (~R.on_the_image_mask[0]).any()                   # the fallback path was exercised


def test_decoder_gradients_match_reference(ref):
    """Pins the oracle's GRADIENT semantics (where the reference detaches: heat-map proposals :225,
    next-level query positions :268,:281, RoI geometry decoder_utils.py:662-679,792-803) to the reference's
    own Python: same random linear functional of all outputs, autograd on both, eval-mode modules
    (the RoI blocks' stubs are differentiable restatements)."""
    shape = synth.SHAPE_TINY
    cfg = configs.decoder_cfg(bev=36, num_proposals=24)
    torch.manual_seed(7)
    R = ref.decoder.DeepInteractionDecoder(**cfg)
    _randomize_bn(R)
    O = odec.DeepInteractionDecoder(**cfg)
    O.load_state_dict(R.state_dict())
    R.eval(), O.eval()
    g = torch.Generator().manual_seed(0)
    Hi, Wi = shape['img_hw']
    base = [torch.randn(1, 128, 36, 36, generator=g), torch.randn(1, 128, 36, 36, generator=g),
            torch.randn(6, 128, Hi, Wi, generator=g)]
    metas = synth.make_inputs(1, shape, seed=0)['img_metas']
    grads = []
    for M in (R, O):
        p0, p1, img = (t.clone().requires_grad_(True) for t in base)
        out = M([p0, p1], img, metas)[0][0]
        gen = torch.Generator().manual_seed(3)
        loss = sum((out[k] * torch.randn(out[k].shape, generator=gen)).sum() for k in sorted(out))
        M.zero_grad()
        loss.backward()
        grads.append(([p0.grad, p1.grad, img.grad], {n: p.grad for n, p in M.named_parameters()}))
    for a, b in zip(*[gr[0] for gr in grads]):
        assert torch.allclose(a, b, rtol=0, atol=1e-4 * max(1.0, a.abs().max().item())), (a - b).abs().max()
    n = 0
    for name, gr in grads[0][1].items():
        go = grads[1][1][name]
        if gr is None:
            assert go is None or go.abs().max().item() == 0.0, name
            continue
        assert go is not None, name
        assert torch.allclose(gr, go, rtol=0, atol=1e-4 * max(1.0, gr.abs().max().item())), (name, (gr - go).abs().max())
        n += 1
    assert n > 100


def test_encoder_gradients_match_reference(ref):
    """Gradient pin of the oracle's MMRI encoder: the reference's own Python (its autograd Functions
    around the compiled reference locatt kernels, F.grid_sample, nn.MultiheadAttention through group_attn)
    vs the oracle's autograd, eval mode (dropout off), one layer."""
    shape = synth.SHAPE_TINY
    inp = synth.make_inputs(1, shape, seed=0)
    torch.manual_seed(1234)
    R = ref.encoder.DeepInteractionEncoder(num_layers=1, in_channels_img=shape['c_img'],
                                           in_channels_pts=shape['c_pts'], hidden_channel=128)
    _randomize_bn(R)
    O = oenc.DeepInteractionEncoder(1, shape['c_img'], shape['c_pts'], 128)
    O.load_state_dict(R.state_dict())
    R.eval(), O.eval()
    sparse = oenc.BEVWarp().sparse_depth(inp['pts_metas']['pts'][0], inp['img_metas'][0],
                                         oenc.lidar2img_tensor(inp['img_metas'], inp['img_feats'])[0], *shape['img_hw'])
    grads = []
    for M in (R, O):
        img = inp['img_feats'].clone().requires_grad_(True)
        pts = inp['pts_feats'].clone().requires_grad_(True)
        im, (p0, p1) = M(img, pts, inp['img_metas'], inp['pts_metas'])
        gen = torch.Generator().manual_seed(4)
        loss = sum((t * torch.randn(t.shape, generator=gen)).sum() for t in (im, p0, p1))
        M.zero_grad()
        loss.backward()
        grads.append(([img.grad, pts.grad], {n: p.grad for n, p in M.named_parameters()}))
    for a, b in zip(*[gr[0] for gr in grads]):
        assert torch.allclose(a, b, rtol=0, atol=2e-5 * max(1.0, a.abs().max().item())), (a - b).abs().max()
    n = 0
    for name, gr in grads[0][1].items():
        go = grads[1][1][name]
        assert (gr is None) == (go is None), name
        if gr is None:
            continue
        assert torch.allclose(gr, go, rtol=0, atol=5e-5 * max(1.0, gr.abs().max().item())), (name, (gr - go).abs().max())
        n += 1
    assert n > 60


# ---------------------------------------------------------------------------------- DeepInteraction++ (row a20)
from oracle import plusplus as opp  # noqa: E402


def _perturb(m, seed=7, scale=0.05):
    """The mmcv default inits zero the deformable-attention projections: move every parameter off its init so
    the pin sees all paths."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn(p.shape, generator=g) * scale)


@pytest.mark.parametrize('aug', [False, True])
def test_pp_encoder_matches_reference(ref, aug):
    shape = synth.SHAPE_PP_TINY
    inp = synth.make_inputs_pp(1, shape, seed=0, aug=synth.example_aug(1) if aug else None)
    torch.manual_seed(1234)
    cfg = configs.encoder_pp_cfg(shape['c_img'], shape['c_pts'])
    R = ref.fix_zero_layer_encoders(ref.encoder_pp.FusionTransformerv4(**cfg))
    _perturb(R)
    O = opp.FusionTransformerv4(**cfg)
    O.load_state_dict(R.state_dict())
    assert set(O.state_dict()) == set(R.state_dict())
    R.eval(), O.eval()
    with torch.no_grad():
        ri, (rp0, rp1) = R(list(inp['img_feats']), list(inp['pts_feats']), inp['img_metas'], inp['pts_metas'])
        oi, (op0, op1) = O(list(inp['img_feats']), list(inp['pts_feats']), inp['img_metas'], inp['pts_metas'])
    for a, b in ((ri, oi), (rp0, op0), (rp1, op1)):
        assert a.shape == b.shape
        assert torch.allclose(a, b, rtol=0, atol=5e-5), (a - b).abs().max()
    assert ri.abs().max() > 0.1 and rp1.abs().max() > 0.1


def test_pp_polar_attention_matches_reference(ref):
    """MMRI_I2P_Polar alone, two samples with different augmentation records."""
    shape = synth.SHAPE_PP_TINY
    inp = synth.make_inputs_pp(2, shape, seed=4, aug=synth.example_aug(2))
    torch.manual_seed(3)
    R = ref.fix_zero_layer_encoders(ref.encoder_pp.MMRI_I2P_Polar(128, 0.1)).eval()
    O = opp.MMRI_I2P_Polar(128, 0.1).eval()
    O.load_state_dict(R.state_dict())
    assert set(O.state_dict()) == set(R.state_dict())
    g = torch.Generator().manual_seed(1)
    bev = torch.randn(2, 128, *shape['bev_hw'], generator=g)
    img = torch.randn(12, 128, *shape['img_hw'], generator=g)
    with torch.no_grad():
        a = R(bev, img, inp['img_metas'], inp['pts_metas'])
        b = O(bev, img, inp['img_metas'], inp['pts_metas'])
    assert torch.allclose(a, b, atol=2e-5), (a - b).abs().max()
    assert (a - bev).abs().max() > 0.1


def _decoder_pp_pair(ref, bev, Q):
    cfg = configs.decoder_cfg(bev=bev, num_proposals=Q)
    torch.manual_seed(11)
    R = ref.decoder_pp.DeepInteractionPlusPlusDecoder(**cfg)
    _randomize_bn(R)
    O = opp.DeepInteractionPlusPlusDecoder(**cfg)
    O.load_state_dict(R.state_dict())
    assert set(O.state_dict()) == set(R.state_dict())
    return R.eval(), O.eval()


@pytest.mark.parametrize('aug', [False, True])
def test_pp_decoder_matches_reference(ref, aug):
    shape = synth.SHAPE_TINY
    Hb = shape['bev_hw'][0]
    R, O = _decoder_pp_pair(ref, Hb, 24)
    inp = synth.make_inputs(1, shape, seed=5, aug=synth.example_aug(3) if aug else None)
    g = torch.Generator().manual_seed(2)
    pts = [torch.randn(1, 128, Hb, Hb, generator=g), torch.randn(1, 128, Hb, Hb, generator=g)]
    img = torch.randn(6, 128, *shape['img_hw'], generator=g)
    with torch.no_grad():
        r = R(pts, img, inp['img_metas'])[0][0]
        o = O(pts, img, inp['img_metas'])[0][0]
    assert set(r) == set(o)
    for k in r:
        assert torch.allclose(r[k], o[k], atol=3e-5), (k, (r[k] - o[k]).abs().max())
    assert len(R.on_the_image_mask) == len(O.on_the_image_mask) == 4
    for a, b in zip(R.on_the_image_mask, O.on_the_image_mask):
        assert torch.equal(a.cpu(), b.cpu())
