"""GPU: the training step of the hot path - gradients of the HIP path against
torch autograd of the CPU oracle on the same seeded inputs, float32.

  * operator level: BEV gather, RoIAlign and pillar attention backward kernels (float32 atomics:
    summation order differs, 1e-4 of the gradient scale);
  * module level: encoder + decoder in train() mode (BatchNorm batch statistics, dropout 0) - gradients of
    a random linear functional of all head outputs w.r.t. both input feature maps and every parameter
    (2e-3 of each tensor's gradient scale: ~60 stacked fp32 stages with different reduction orders);
  * attention dropout of the pillar attention: forward and backward regenerate the same mask
    (directional derivative vs finite difference) and drop the expected fraction.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from deepinteraction_amd import ops, synth
from deepinteraction_amd.autograd import BEVWarpGather, I2PAttention, RoIAlign
from deepinteraction_amd.geometry import SampleGeometry
from oracle import configs, decoder as odec, encoder as oenc
from oracle.refpin import make_golden as mg
from oracle.thirdparty import roi_align_v2

DEV = 'cuda'


def _cl(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


def test_bevwarp_gather_backward():
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    shape = synth.SHAPE_TINY
    inp = synth.make_inputs(1, shape, seed=3)
    meta = inp['img_metas'][0]
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    g = torch.Generator().manual_seed(1)
    dense = torch.rand(6, Hi, Wi, generator=g) * 60.0
    dense[torch.rand(6, Hi, Wi, generator=g) < 0.2] = 0.0
    bev = torch.randn(1, 128, Hb, Wb, generator=g, requires_grad=True)
    w = torch.randn(6, 128, Hi, Wi, generator=g)
    # oracle: grid_sample autograd
    l2i = oenc.lidar2img_tensor([meta], bev)[0]
    grid, lift = oenc.bev_sample_grid(dense, torch.inverse(l2i), meta, *meta['input_shape'])
    warped = F.grid_sample(bev.expand(6, -1, -1, -1), grid, mode='bilinear', padding_mode='zeros',
                           align_corners=False) * lift.unsqueeze(1)
    (warped * w).sum().backward()
    # HIP
    geom = SampleGeometry(meta, (Hi, Wi), DEV)
    bev_d = _cl(bev.detach()).requires_grad_(True)
    out = BEVWarpGather.apply(bev_d, dense.to(DEV), geom.img2lidar, geom.aug_fwd, geom.xs, geom.ys, geom.pc_range)
    (out * w.to(DEV)).sum().backward()
    err = (bev_d.grad.cpu() - bev.grad).abs().max().item()
    assert err <= 1e-4 * bev.grad.abs().max().item(), err


def test_roi_align_backward():
    g = torch.Generator().manual_seed(2)
    N, C, H, W = 3, 128, 20, 31
    feat = torch.randn(N, C, H, W, generator=g)
    R = 17
    xy = torch.rand(R, 2, generator=g) * torch.tensor([W * 4.0, H * 4.0]) - 8.0        # some boxes hang over the edge
    wh = torch.rand(R, 2, generator=g) * 40.0 + 1.0
    idx = torch.randint(0, N, (R, 1), generator=g).float()
    rois = torch.cat([idx, xy, xy + wh], 1)
    wgt = torch.randn(R, 49, C, generator=g)
    ref = torch.zeros_like(feat)
    for n in range(N):
        sel = (idx[:, 0] == n).nonzero().flatten()
        if sel.numel() == 0:
            continue
        f = feat[n:n + 1].clone().requires_grad_(True)
        o = roi_align_v2(f, rois[sel, 1:], 7, 0.25, 2)                                  # (q,C,7,7)
        (o.flatten(2).transpose(1, 2) * wgt[sel]).sum().backward()
        ref[n] = f.grad[0]
    fd = _cl(feat).requires_grad_(True)
    out = RoIAlign.apply(fd, rois.to(DEV), 0.25)
    (out * wgt.to(DEV)).sum().backward()
    err = (fd.grad.cpu() - ref).abs().max().item()
    assert err <= 1e-4 * ref.abs().max().item(), err


def _i2p_pair(dropout):
    from deepinteraction_amd.mmdet3d_plugin.models.utils.encoder_utils import MMRI_I2P
    torch.manual_seed(5)
    O = oenc.MMRI_I2P(128, 128, dropout)
    mg.randomize(O, 4)
    M = MMRI_I2P(128, 128, dropout)
    M.load_state_dict(O.state_dict())
    return O, M


def test_i2p_attention_backward_vs_oracle():
    shape = synth.SHAPE_TINY
    inp = synth.make_inputs(1, shape, seed=6)
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    O, M = _i2p_pair(0.0)
    O.train(), M.train()
    g = torch.Generator().manual_seed(8)
    bev = torch.randn(1, 128, Hb, Wb, generator=g)
    img = torch.randn(1, 6, 128, Hi, Wi, generator=g)
    w = torch.randn(1, 128, Hb, Wb, generator=g)
    bo, io = bev.clone().requires_grad_(True), img.clone().requires_grad_(True)
    (O(bo, io, inp['img_metas'], inp['pts_metas']) * w).sum().backward()
    pm = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
    M = M.to(DEV)
    bd = _cl(bev).requires_grad_(True)
    idv = img.to(DEV).requires_grad_(True)
    (M(bd, idv, inp['img_metas'], pm) * w.to(DEV)).sum().backward()
    for name, got, ref in (('bev', bd.grad, bo.grad), ('img', idv.grad, io.grad)):
        err = (got.cpu() - ref).abs().max().item()
        assert err <= 2e-4 * ref.abs().max().item(), (name, err)
    ref_p = dict(O.named_parameters())
    for name, p in M.named_parameters():
        if ref_p[name].grad is None:          # e.g. nothing: all MHA parameters are used
            continue
        # the key bias only shifts all scores of a pillar: its exact gradient is 0; the oracle's is round-off
        ref = ref_p[name].grad
        err = (p.grad.cpu() - ref).abs().max().item()
        assert err <= 5e-4 * max(ref.abs().max().item(), 1e-3), (name, err)


def test_i2p_attention_dropout_is_consistent():
    """Same (seed, p): forward twice is identical, ~p of the keys are dropped, and the backward matches a
    finite difference of the forward along a random direction (i.e. it regenerates the same mask)."""
    shape = synth.SHAPE_TINY
    inp = synth.make_inputs(1, shape, seed=6)
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    geom = SampleGeometry(inp['img_metas'][0], (Hi, Wi), DEV)
    pm = inp['pts_metas']
    args = (pm['pillars'].to(DEV), pm['pillar_coors'].to(DEV), pm['pillars_num_points'].to(DEV), geom.lidar2img,
            geom.aug_rev, geom.ori_hw)
    g = torch.Generator().manual_seed(9)
    img = _cl(torch.randn(6, 128, Hi, Wi, generator=g).double().float())
    qf = _cl(torch.randn(1, 128, Hb, Wb, generator=g) * 0.05)
    a, _ = ops.i2p_attention(img, qf, *args, dropout_p=0.5, seed=77)
    b, _ = ops.i2p_attention(img, qf, *args, dropout_p=0.5, seed=77)
    c, _ = ops.i2p_attention(img, qf, *args, dropout_p=0.5, seed=78)
    z, valid = ops.i2p_attention(img, qf, *args)
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, z)
    # constant image => context = (kept mass) / (1 - p): the kept fraction of the softmax mass averages 1 - p
    ones = torch.ones_like(img)
    d, _ = ops.i2p_attention(ones, torch.zeros_like(qf), *args, dropout_p=0.5, seed=5)
    kept = (d[:, 0] * 0.5)[valid[:, 0] > 0]
    assert 0.45 <= kept.mean().item() <= 0.55, kept.mean().item()
    # the kept probability mass (what the folded value bias is scaled by): the same forward also returns it; without
    # dropout it is `valid`, and it does not depend on the image (zero scores: softmax of the key count only)
    d2, valid2, mass = ops.i2p_attention(img, torch.zeros_like(qf), *args, dropout_p=0.5, seed=5, with_mass=True)
    assert torch.equal(valid2, valid) and (mass[valid == 0] == 0).all()
    assert 0.9 <= (mass[valid > 0].mean() * 0.5 / kept.mean()).item() <= 1.1      # = kept mass / (1 - p), as measured above
    _, _, mass0 = ops.i2p_attention(img, qf, *args, with_mass=True)
    assert torch.allclose(mass0, valid, atol=1e-6)
    # directional derivative of <ctx, w> + <mass, wm>
    w = torch.randn(a.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    wm = torch.randn(mass.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    imr = img.clone().requires_grad_(True)
    qr = qf.clone().requires_grad_(True)
    out, _, ms = I2PAttention.apply(imr, qr, *args, 0.5, 77)
    ((out * w).sum() + (ms * wm).sum()).backward()
    di = torch.randn(img.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    dq = torch.randn(qf.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)) * 0.05
    eps = 1e-2

    def f(sign):
        o, _, m_ = ops.i2p_attention(_cl(img + sign * eps * di), _cl(qf + sign * eps * dq), *args, dropout_p=0.5, seed=77,
                                     with_mass=True)
        return (o * w).sum() + (m_ * wm).sum()
    fp, fm = f(1.0), f(-1.0)
    fd = ((fp - fm) / (2 * eps)).item()
    an = ((imr.grad * di).sum() + (qr.grad * dq).sum()).item()
    assert abs(fd - an) <= 2e-2 * max(abs(an), 1.0), (fd, an)


def test_training_step_gradients_match_oracle(inject_depth):
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionDecoder, DeepInteractionEncoder
    shape = synth.SHAPE_TINY
    torch.backends.cudnn.deterministic = True     # MIOpen's atomic solvers add run-to-run noise on the tiny maps
    cfg = configs.decoder_cfg(bev=shape['bev_hw'][0], num_proposals=24)
    cfg['dropout'] = 0.0
    torch.manual_seed(11)
    OE = oenc.DeepInteractionEncoder(1, shape['c_img'], shape['c_pts'], 128)
    OD = odec.DeepInteractionDecoder(**cfg)
    mg.randomize(OE, 21)
    mg.randomize(OD, 22)
    ME = DeepInteractionEncoder(1, shape['c_img'], shape['c_pts'], 128)
    MD = DeepInteractionDecoder(**cfg)
    ME.load_state_dict(OE.state_dict())
    MD.load_state_dict(OD.state_dict())
    for blk_o, blk_m in zip(OE.fusion_blocks, ME.fusion_blocks):       # attention dropout off on both sides
        blk_o.I2P_block.learnedAlign.dropout = 0.0
        blk_m.I2P_block.dropout = 0.0
    for m in (OE, OD, ME, MD):
        m.train()
    inp = synth.make_inputs(1, shape, seed=12)
    metas = inp['img_metas']
    # the completed depth is injected so that both sides unproject through identical depth maps
    sparse = oenc.BEVWarp().sparse_depth(inp['pts_metas']['pts'][0], metas[0],
                                         oenc.lidar2img_tensor(metas, inp['img_feats'])[0], *shape['img_hw'])
    pm = dict(inp['pts_metas'], dense_depth=oenc.complete_depth(sparse).unsqueeze(0))

    def run(E, D, img, pts, pmetas, dev, top=None):
        im, (p0, p1) = E(img, pts, metas, pmetas)
        out = D([p0, p1], im, metas, **({} if top is None else {'top_override': top}))[0][0]
        gen = torch.Generator().manual_seed(33)
        loss = 0.0
        for k in sorted(out):
            loss = loss + (out[k].float() * torch.randn(out[k].shape, generator=gen).to(dev)).sum()
        return loss

    ME, MD = ME.to(DEV), MD.to(DEV)
    pmd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in pm.items() if k != 'dense_depth'}
    pmd['pts'] = [p.to(DEV) for p in pm['pts']]
    inject_depth(pmd['pts'], pm['dense_depth'])        # test-only fixture: the product has no injection hook

    def check(name, got, ref, tol):
        scale = ref.abs().max().item()
        d = (got.cpu() - ref).abs()
        # + 2e-4 absolute: parameters whose exact gradient is 0 (a BatchNorm bias feeding a 1x1 conv that is
        # batch-normalised again; the key bias of a softmax) carry only round-off on both sides.
        # Train-mode BatchNorm statistics are sums whose order differs between the two sides (and, on the CPU, with
        # the thread count): pre-activations move by ~1e-6, and the handful of them that sit within that distance of
        # 0 flip their ReLU gate - a DISCRETE change of the gradient around that pixel (measured: the oracle against
        # itself at 1 vs 8 threads differs by 1 % of the scale on ~0.3 % of d img_feats).  So: the bulk within `tol`,
        # at most 1 % of the elements beyond it, none beyond 5 % of the scale.
        bad = (d > tol * scale + 2e-4).float().mean().item()
        assert bad <= 1e-2, (name, bad, d.max().item(), scale)
        assert d.max().item() <= 5e-2 * scale + 2e-4, (name, d.max().item(), scale)

    def check_params(mod_m, mod_o, tag, tol):
        ref, n = dict(mod_o.named_parameters()), 0
        for name, p in mod_m.named_parameters():
            r = ref[name].grad
            if r is None:
                assert p.grad is None or p.grad.abs().max().item() == 0.0, (tag, name)
                continue
            assert p.grad is not None and torch.isfinite(p.grad).all(), (tag, name)
            if r.abs().max().item() < 1e-7:
                continue                              # e.g. the key bias of a softmax: exact zero
            check(f'{tag}.{name}', p.grad, r, tol)
            n += 1
        return n

    def functional(outs, dev, seed):
        gen = torch.Generator().manual_seed(seed)
        return sum((t.float() * torch.randn(t.shape, generator=gen).to(dev)).sum() for t in outs)

    # ---- A. encoder: gradients of a random functional of its three outputs
    img_d = inp['img_feats'].to(DEV).requires_grad_(True)
    pts_d = inp['pts_feats'].to(DEV).requires_grad_(True)
    im_d, (p0_d, p1_d) = ME(img_d, pts_d, metas, pmd)
    functional((im_d, p0_d, p1_d), DEV, 31).backward()
    img_o = inp['img_feats'].clone().requires_grad_(True)
    pts_o = inp['pts_feats'].clone().requires_grad_(True)
    im_o, (p0_o, p1_o) = OE(img_o, pts_o, metas, pm)
    functional((im_o, p0_o, p1_o), 'cpu', 31).backward()
    check('encoder d img_feats', img_d.grad, img_o.grad, 2e-4)
    check('encoder d pts_feats', pts_d.grad, pts_o.grad, 2e-4)
    n_enc = check_params(ME, OE, 'encoder', 5e-4)
    assert n_enc > 60, n_enc

    # ---- B. decoder on IDENTICAL inputs (the oracle encoder's outputs): gradients w.r.t. the three
    # feature maps and every parameter.  (Fed with its own encoder's outputs, which differ by ~2e-6, the
    # decoder gradient moves by ~0.5 %: RoIs hanging over the map edge pool all-zero bins, and the
    # LayerNorm of a constant row (DynamicConv, decoder_utils.py:614-621) amplifies round-off of that
    # row by 1/sqrt(eps) in backward - in the reference just as here.  Stage-wise parity on identical
    # inputs plus the chain rule (part C) is the well-conditioned statement.)
    feats_o = [t.detach().clone().requires_grad_(True) for t in (im_o, p0_o, p1_o)]
    feats_d = [t.detach().to(DEV).requires_grad_(True) for t in (im_o, p0_o, p1_o)]
    out_d = MD([feats_d[1], feats_d[2]], feats_d[0], metas)[0][0]
    functional([out_d[k] for k in sorted(out_d)], DEV, 33).backward()
    out_o = OD([feats_o[1], feats_o[2]], feats_o[0], metas, top_override=MD.top_proposals.cpu())[0][0]
    functional([out_o[k] for k in sorted(out_o)], 'cpu', 33).backward()
    for name, a, b in zip(('img', 'pts_conv', 'pts'), feats_d, feats_o):
        check(f'decoder d {name}', a.grad, b.grad, 1e-3)
    n_dec = check_params(MD, OD, 'decoder', 2e-3)
    assert n_dec > 100, n_dec

    # ---- C. end to end: the chain rule through both modules agrees with feeding the decoder's input
    # gradients (B-style leaves of OUR encoder outputs) through OUR encoder backward, everything finite
    for m in (ME, MD):
        m.zero_grad()
    img_e = inp['img_feats'].to(DEV).requires_grad_(True)
    pts_e = inp['pts_feats'].to(DEV).requires_grad_(True)
    im_e, (p0_e, p1_e) = ME(img_e, pts_e, metas, pmd)
    out_e = MD([p0_e, p1_e], im_e, metas)[0][0]
    loss = functional([out_e[k] for k in sorted(out_e)], DEV, 33)
    g_feats = torch.autograd.grad(loss, [im_e, p0_e, p1_e], retain_graph=True)
    loss.backward(retain_graph=True)
    g_in = torch.autograd.grad([im_e, p0_e, p1_e], [img_e, pts_e], g_feats)
    # (two backward passes: the float32 atomics of the scatter kernels add in a different order each time)
    check('e2e d img_feats', img_e.grad, g_in[0].cpu(), 1e-3)
    check('e2e d pts_feats', pts_e.grad, g_in[1].cpu(), 1e-3)
    for mod in (ME, MD):
        for name, p in mod.named_parameters():
            assert p.grad is None or torch.isfinite(p.grad).all(), name


def test_encoder_gradients_at_shape_R_match_reference_golden():
    """Gradient parity at the BENCHED shape (BASELINE.json configs[2]: Fusion_0075_refactor shape R, 2 interaction
    layers): the product's backward - fused-window-attention entry points, pillar-attention / BEV-gather scatter kernels,
    every 1x1 and 3x3 weight gradient - against gradients the REFERENCE'S OWN Python produced for the same seeded inputs,
    weights and linear functional (tests/golden/grad_shapeR.npz, oracle/refpin/make_golden_grad_shapeR.py; eval-mode
    BatchNorm).  Compared on the stored strided samples: the bulk within 5e-4 of each tensor's gradient scale, <= 1 % of
    the entries beyond it (a projected point within float round-off of a texel boundary moves its bilinear corner),
    none beyond 5 %, absolute sums within 1e-3."""
    import os

    import numpy as np
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionEncoder
    from oracle.refpin import make_golden_grad_shapeR as gg
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    torch.backends.cudnn.deterministic = True
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'grad_shapeR.npz'))
    inp = gg.case()
    M = DeepInteractionEncoder(num_layers=2, in_channels_img=gg.SHAPE['c_img'], in_channels_pts=gg.SHAPE['c_pts'],
                               hidden_channel=128)
    mg.randomize(M, gg.SEED_WEIGHTS)
    M = M.to(DEV).eval()
    pm = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
    pm['pts'] = [p.to(DEV) for p in inp['pts_metas']['pts']]
    img = inp['img_feats'].to(DEV).requires_grad_(True)
    pts = inp['pts_feats'].to(DEV).requires_grad_(True)
    im, (p0, p1) = M(img, pts, inp['img_metas'], pm)
    # forward first: the three outputs against the reference's (samples)
    for name, t in (('out_img', im), ('out_pts_conv', p0), ('out_pts', p1)):
        s, a, m = gg.sample(t)
        d = np.abs(s - gold[name + '.sample']) / max(1.0, float(gold[name + '.absmax']))
        # bulk at float32 round-off; a pixel whose completed depth differs in the last bit may move a bilinear corner
        assert np.quantile(d, 0.99) <= 2e-5 and (d > 1e-3).mean() <= 5e-3, (name, np.quantile(d, 0.99), d.max())
    gg.functional((im, p0, p1), DEV).backward()
    torch.cuda.synchronize()
    got = dict([('d_img_feats', img.grad), ('d_pts_feats', pts.grad)] +
               [('p.' + n, p.grad) for n, p in M.named_parameters() if p.grad is not None])
    names = sorted(k[:-len('.sample')] for k in gold.files if k.endswith('.sample') and not k.startswith('out_'))
    assert len(names) > 60 and set(names) <= set(got), set(names) - set(got)
    worst, fails = {}, []
    for name in names:
        s, a, m = gg.sample(got[name])
        ref, scale = gold[name + '.sample'], float(gold[name + '.absmax'])
        if scale < 1e-7:
            continue
        d = np.abs(s - ref)
        # sums over 134 400 pixels (weight / bias gradients) or float32 atomics (scatter kernels) in another order than the
        # reference's: measured <= 1.1e-3 of the tensor's gradient scale on single entries; the bulk within 5e-4
        tol = 2e-3 * scale + 2e-4
        bad = float((d > tol).mean())
        worst[name] = dict(frac_beyond_2e3=bad, frac_beyond_5e4=float((d > 5e-4 * scale + 2e-4).mean()),
                           max_rel=float(d.max() / scale),
                           abssum_rel=abs(float(a) - float(gold[name + '.abssum'])) / max(float(gold[name + '.abssum']), 1e-12))
        if bad > 1e-2 or d.max() > 5e-2 * scale + 2e-4 or worst[name]['abssum_rel'] > 2e-3:
            fails.append((name, worst[name]))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    import json
    with open(os.path.join(out, 'grad_parity_shapeR.json'), 'w') as f:
        json.dump(worst, f, indent=1)
    assert not fails, fails[:5]
    assert np.median([w['frac_beyond_5e4'] for w in worst.values()]) == 0.0


def _golden(name):
    import os

    import numpy as np
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name)
    if not os.path.exists(path):
        pytest.skip(f'{name} not generated (oracle/refpin/make_golden_grad_more.py)')
    return np.load(path)


def _report(name, obj):
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), 'w') as f:
        json.dump(obj, f, indent=1)


def _grad_stats(got, gold, names, sampler):
    """Per gradient tensor, on the golden file's strided sample: relative L2 error, the fraction of entries beyond
    2e-3 / 5e-4 of the tensor's gradient scale, the largest error in units of the scale, and the relative difference of
    the absolute sums (over the WHOLE tensor: mass lost to an fp16 flush would show there)."""
    import numpy as np
    stats = {}
    for name in names:
        ref, scale = gold[name + '.sample'], float(gold[name + '.absmax'])
        if scale < 1e-7:
            continue
        s, a = sampler(got[name])
        d = np.abs(s - ref)
        stats[name] = dict(rel_l2=float(np.linalg.norm(s - ref) / max(np.linalg.norm(ref), 1e-30)),
                           frac_beyond_2e3=float((d > 2e-3 * scale + 2e-4).mean()),
                           frac_beyond_5e4=float((d > 5e-4 * scale + 2e-4).mean()), max_rel=float(d.max() / scale),
                           abssum_rel=abs(float(a) - float(gold[name + '.abssum'])) / max(float(gold[name + '.abssum']), 1e-12))
    return stats


@pytest.mark.parametrize('amp', [False, True])
def test_encoder_train_mode_gradients_at_shape_R_match_reference_golden(amp):
    """The TRAINING forward / backward of the MMRI encoder at the benched shape (BASELINE.json configs[2]) against gradients the
    REFERENCE'S OWN Python produced in train() mode for the same seeded inputs, weights and functional
    (tests/golden/grad_shapeR_train.npz, oracle/refpin/make_golden_grad_more.py enc_train; pillar-attention dropout off on both
    sides): BatchNorm batch statistics and their backward at full size, every window-attention / pillar-attention / BEV-gather
    gradient, every 1x1 / 3x3 weight gradient.
      * float32 (the reference's arithmetic): the same bounds as the eval-mode golden test above;
      * `amp` = what `bench.py --mode train --amp` runs - fp16 model weights with float32 masters (`train_step.half_weights_`),
        torch.autocast(fp16) with the fused window attention (csrc/local_attn_train.hip) and the own BatchNorm + ReLU kernels
        (csrc/batchnorm.hip): per-tensor bounds against the REFERENCE
        gradients (round-4 verdict: not cosine against the product's own float32), and no tensor loses more than 1e-2 of its
        absolute mass (an fp16 flush of small gradients would)."""
    import numpy as np
    from deepinteraction_amd import train_step
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionEncoder
    from oracle.refpin import make_golden_grad_more as gm, make_golden_grad_shapeR as gg
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    torch.backends.cudnn.deterministic = True
    gold = _golden('grad_shapeR_train.npz')
    inp = gg.case()
    M = DeepInteractionEncoder(num_layers=2, in_channels_img=gg.SHAPE['c_img'], in_channels_pts=gg.SHAPE['c_pts'],
                               hidden_channel=128)
    mg.randomize(M, gg.SEED_WEIGHTS)
    M = gm.enc_train_prepare(M).to(DEV)
    # (no loss scale HERE: the seeded functional weights every output element with an N(0, 1) number - gradients 1e3-1e5 times a
    # normalised loss's - so the stricter statement is made: even unscaled, no tensor loses mass to an fp16 flush.  The trainer's
    # dynamic scale has its own tests: test_amp_overflow_step_..., tests/test_host_logic.py::test_loss_scaler_...)
    scale = 1.0
    if amp:
        half, _ = train_step.half_weights_([M])
        assert len(half) > 40
    pm = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
    pm['pts'] = [p.to(DEV) for p in inp['pts_metas']['pts']]
    img = inp['img_feats'].to(DEV).requires_grad_(True)
    pts = inp['pts_feats'].to(DEV).requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.float16, enabled=amp):
        im, (p0, p1) = M(img, pts, inp['img_metas'], pm)
    if amp:
        assert im.dtype == torch.float16
    fwd = {}
    for name, t in (('out_img', im), ('out_pts_conv', p0), ('out_pts', p1)):
        s = gg.sample(t)[0]
        d = np.abs(s - gold[name + '.sample']) / max(1.0, float(gold[name + '.absmax']))
        fwd[name] = dict(max=float(d.max()), p99=float(np.quantile(d, 0.99)), median=float(np.median(d)),
                         frac_gt_1e3=float((d > 1e-3).mean()))
    (gg.functional((im, p0, p1), DEV) * scale).backward()
    torch.cuda.synchronize()
    got = dict([('d_img_feats', img.grad / scale), ('d_pts_feats', pts.grad / scale)] +
               [('p.' + n, p.grad.float() / scale) for n, p in M.named_parameters() if p.grad is not None])
    assert all(torch.isfinite(g).all() for g in got.values()), 'fp16 overflow'
    names = sorted(k[:-len('.sample')] for k in gold.files if k.endswith('.sample') and not k.startswith('out_'))
    assert len(names) > 60 and set(names) <= set(got), set(names) - set(got)
    # The BatchNorm bias of an `*_out_proj` feeds a 1x1 convolution that is batch-normalised again (`*_integration`,
    # deepinteraction_encoder.py:26-27,31-32): in train mode a constant per channel is removed by the next layer's mean, so its
    # exact gradient is 0 and both sides hold only the round-off of a 134 400-term sum (their ratio is O(1)).  Compared against
    # the scale of the same layer's weight gradient instead.
    zero_grad = [n for n in names if n.endswith('_out_proj.bn.bias')]
    assert len(zero_grad) == 4
    for n in zero_grad:
        w = n[:-len('bias')] + 'weight'
        assert float(got[n].abs().max()) <= 1e-3 * float(gold[w + '.absmax']), (n, float(got[n].abs().max()), float(gold[w + '.absmax']))
    names = [n for n in names if n not in zero_grad]
    stats = _grad_stats(got, gold, names, lambda t: gg.sample(t)[:2])
    _report(f'grad_parity_shapeR_train_{"amp" if amp else "f32"}.json', dict(forward=fwd, gradients=stats))
    if not amp:
        for name, f in fwd.items():
            # train-mode BatchNorm: batch statistics are 134 400-term sums in another order (1e-6), no fp16 anywhere
            assert f['p99'] <= 5e-5 and f['frac_gt_1e3'] <= 5e-3, (name, f)
        # measured (session r05b): relative L2 <= 3.6e-3 (median 6e-4), single entries <= 1.2 % of the tensor's scale, absolute
        # sums within 8e-4.  Batch statistics summed in another order move pre-activations by ~1e-6 and a handful of them flip
        # their ReLU gate - a discrete change of the gradient around that pixel, on the reference's CPU run against itself with
        # another thread count just the same (test_training_step_gradients_match_oracle): hence L2 and not a bulk bound.
        fails = [(n, w) for n, w in stats.items() if w['rel_l2'] > 1e-2 or w['max_rel'] > 5e-2 or w['abssum_rel'] > 5e-3]
        assert not fails, fails[:5]
        assert np.median([w['rel_l2'] for w in stats.values()]) <= 2e-3
    else:
        for name, f in fwd.items():
            assert f['median'] <= 2e-4 and f['p99'] <= 1e-3 and f['max'] <= 2e-3, (name, f)
        # measured (session r05h): outputs max 7.7e-4 / p99 3.9e-4 of their range; gradients relative L2 <= 5.1e-2 (median
        # 1.4e-2), absolute sums within 7.7e-3 - no tensor loses mass to an fp16 flush.  Asserted with head-room.
        fails = [(n, w) for n, w in stats.items() if w['rel_l2'] > 8e-2 or w['abssum_rel'] > 1.5e-2]
        assert not fails, fails[:5]
        assert np.median([w['rel_l2'] for w in stats.values()]) <= 2e-2


def test_head_gradients_at_shape_R_match_reference_golden():
    """Head gradients at the BENCHED shape (round-4 verdict, weak 4): `DeepInteractionDecoder` (Q = 200, one decoder layer +
    four RoI layers) in train() mode with dropout 0 on seeded shape-R feature maps, default initialisation with the RoI blocks
    conditioned as in a trained head (`make_golden_grad_more.head_prepare`), against the
    gradients the REFERENCE'S OWN Python produced (tests/golden/grad_head_shapeR.npz, make_golden_grad_more.py head): the
    proposals, labels and image masks are identical, the outputs agree to float32 round-off, and every gradient - the three
    feature maps and all parameters - lies within max(2e-3, 4 x the reference's own sensitivity to a 1e-6 input change) in
    relative L2 (the file's `.noise`, the largest of three perturbed reference runs: RoIs over the map edge + LayerNorm of
    constant rows make single tensors move by up to 1.2 % on the reference itself, the median tensor by 5e-5)."""
    import numpy as np
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionDecoder
    from oracle.refpin import make_golden_grad_more as gm
    torch.backends.cudnn.deterministic = True
    gold = _golden('grad_head_shapeR.npz')
    M = gm.head_prepare(DeepInteractionDecoder(**gm.head_cfg())).to(DEV)
    feats, metas = gm.head_case()
    f = [t.to(DEV).requires_grad_(True) for t in feats]
    res = M([f[1], f[2]], f[0], metas)[0][0]
    assert torch.equal(M.query_labels.cpu(), torch.from_numpy(gold['query_labels']))
    masks = torch.stack([m.cpu() for m in M.on_the_image_mask])
    assert torch.equal(masks, torch.from_numpy(gold['on_the_image_mask']).to(masks.dtype))
    fwd = {}
    for k, v in res.items():
        r = torch.from_numpy(gold['out.' + k])
        fwd[k] = float((v.detach().cpu() - r).abs().max() / max(1.0, float(r.abs().max())))
        # (`.noise`: what the reference's own outputs move by when its inputs move by 1e-6 - up to 2.5e-4 here)
        assert fwd[k] <= max(2e-4, 4.0 * float(gold['out.' + k + '.noise'])), (k, fwd[k], float(gold['out.' + k + '.noise']))
    gm.head_functional(res, DEV).backward()
    torch.cuda.synchronize()
    got = dict(zip(('d_img', 'd_pts_conv', 'd_pts'), (t.grad for t in f)))
    got.update({'p.' + n: p.grad for n, p in M.named_parameters() if p.grad is not None})
    names = sorted(k[:-len('.sample')] for k in gold.files if k.endswith('.sample'))
    assert len(names) > 250 and set(names) <= set(got), sorted(set(names) - set(got))[:5]
    stats, fails = {}, []
    for name in names:
        ref = gold[name + '.sample']
        if float(gold[name + '.absmax']) < 1e-5:
            continue                                          # exact zeros / pure round-off on both sides
        s = gm.sample(got[name], n=1024 if name.startswith('p.') else 4096)['sample']
        e = float(np.linalg.norm(s - ref) / max(np.linalg.norm(ref), 1e-30))
        noise = float(gold[name + '.noise'])
        stats[name] = dict(rel_l2=e, reference_noise=noise)
        if e > max(2e-3, 4.0 * noise):
            fails.append((name, e, noise))
    _report('grad_parity_head_shapeR.json', dict(forward=fwd, gradients=stats))
    assert len(stats) > 250
    assert not fails, fails[:8]


def test_full_training_step_with_loss():
    """Forward (train mode, dropout on) -> head.loss against ground truth -> backward -> one SGD step, all on
    the HIP path: every trainable parameter gets a finite gradient and the loss goes down."""
    from deepinteraction_amd import det3d_compat as dc
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionDecoder, DeepInteractionEncoder
    from test_targets_loss import TRAIN_CFG
    shape = synth.SHAPE_TINY
    torch.backends.cudnn.deterministic = True
    cfg = configs.decoder_cfg(bev=shape['bev_hw'][0], num_proposals=24)
    torch.manual_seed(2)
    enc = DeepInteractionEncoder(2, shape['c_img'], shape['c_pts'], 128).to(DEV).train()
    dec = DeepInteractionDecoder(**dict(cfg, train_cfg=TRAIN_CFG)).to(DEV).train()
    inp = synth.make_inputs(1, shape, seed=4)
    pm = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
    pm['pts'] = [p.to(DEV) for p in inp['pts_metas']['pts']]
    img, pts = inp['img_feats'].to(DEV), inp['pts_feats'].to(DEV)
    gt = [dc.LiDARBoxes(torch.tensor([[5.0, 3, -1.5, 1.9, 4.6, 1.7, 0.4, 1, 0], [-12.0, 8, -1.2, 0.7, 0.7, 1.8, 0.0, 0, 0],
                                      [20.0, -15, -1.0, 2.5, 8.0, 3.0, 1.3, 0, 2]]))]
    labels = [torch.tensor([0, 8, 3])]
    params = [p for m in (enc, dec) for p in m.parameters()]
    opt = torch.optim.SGD(params, lr=2e-4)

    def total_loss():
        torch.manual_seed(99)                        # same dropout masks on both evaluations
        im, p = enc(img, pts, inp['img_metas'], dict(pm))
        losses = dec.loss(gt, labels, dec(p, im, inp['img_metas']))
        assert set(losses) == {'loss_heatmap', 'matched_ious'} | {f'layer_{l}_loss_{k}' for l in range(4) for k in ('cls', 'bbox')}
        return sum(v for k, v in losses.items() if k != 'matched_ious')
    l0 = total_loss()
    opt.zero_grad()
    l0.backward()
    # `heatmap_head` only feeds the DETACHED proposal heat map (reference :223-225; the loss uses
    # heatmap_head_img's output, :311): it is the one module without a gradient - the reason the reference
    # config sets find_unused_parameters=True
    missing = [n for m in (enc, dec) for n, p in m.named_parameters() if p.grad is None]
    assert missing and all(n.startswith('heatmap_head.') for n in missing), missing[:5]
    assert all(torch.isfinite(p.grad).all() for p in params if p.grad is not None)
    opt.step()
    with torch.no_grad():
        l1 = total_loss()
    assert torch.isfinite(l0) and l1 < l0, (float(l0), float(l1))


@pytest.mark.parametrize('amp', [False, True])
def test_graphed_training_step(amp):
    """(`amp`: the same under torch.autocast(fp16) with the fused window attention of csrc/local_attn_train.hip inside the
    captured forward and backward.)
    `train_step.GraphedTrainer`: forward and backward of the hot path replayed as two captured hipGraphs around the eager
    head loss (host Hungarian step).  (i) it trains: finite losses, every trainable parameter except the detached heat-map
    head gets a finite gradient; (ii) replay is deterministic given the dropout seed word and the torch RNG state: the same
    step from the same weights gives the same loss twice; (iii) the device seed word - rewritten before every replay - really
    drives the pillar attention's dropout mask: another word, another loss."""
    from deepinteraction_amd import ops, train_step
    torch.backends.cudnn.deterministic = True
    tr = train_step.GraphedTrainer(synth.SHAPE_TINY, 24, torch.device(DEV), 1, pool=2, amp=amp)
    try:
        losses = [float(tr.step()) for _ in range(3)]
        assert all(math.isfinite(l) for l in losses), losses
        missing = [n for m in (tr.enc, tr.dec) for n, p in m.named_parameters() if p.grad is None]
        assert all(n.startswith('heatmap_head.') for n in missing), missing[:5]
        if amp:     # fp16 gradients are loss-scaled: a step may overflow (and is then skipped); the weights stay finite
            assert all(torch.isfinite(m).all() for m in tr._master) and float(tr.scaler.scale) > 0
            assert float(tr.scaler.skipped) < 3, 'every step overflowed at the initial loss scale'
        else:
            assert all(torch.isfinite(p.grad).all() for p in tr.params if p.grad is not None)

        def forward_loss(word):
            tr.h.load(tr.records[0])
            tr.seed_word.fill_(word)
            torch.manual_seed(5)
            torch.cuda.manual_seed(5)
            with torch.no_grad():
                outs = tr.graphed(tr.h.img_feats, tr.h.pts_feats)
                preds = [[dict(zip(tr.module.keys, outs))]]
                ls = tr.dec.loss([g[0] for g in tr.pool[0][1]], [g[1] for g in tr.pool[0][1]], preds)
            return float(sum(v for k, v in ls.items() if k != 'matched_ious'))
        a, b, c = forward_loss(1234), forward_loss(1234), forward_loss(99991)
        assert a == b, (a, b)
        assert a != c, (a, c)
    finally:
        ops.set_i2p_seed_tensor(None)


def _no_dropout(enc, dec):
    for m in list(enc.modules()) + list(dec.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        elif isinstance(getattr(m, 'dropout', None), float):
            m.dropout = 0.0


def test_graphed_training_step_two_samples_per_rank():
    """BASELINE configs[3] names the reference's `samples_per_gpu=2` (Fusion_0075_refactor.py:94): `GraphedTrainer(batch=2)`
    captures ONE forward / backward over both samples (BatchNorm statistics over the 12 images, as in the reference), every
    sample's pillars in its fixed slice of the static buffers.  (i) with dropout off, the replayed forward equals the eager
    train-mode forward of the same modules on the unpadded batch; (ii) it trains: finite losses, gradients everywhere but the
    detached heat-map head; (iii) a second pool batch with other pillar counts goes through the same capture."""
    from deepinteraction_amd import ops, train_step
    torch.backends.cudnn.deterministic = True
    tr = train_step.GraphedTrainer(synth.SHAPE_TINY, 24, torch.device(DEV), 1, batch=2, pool=2, prepare_model=_no_dropout)
    try:
        assert tr.h.batch == 2 and len(tr.h.bounds) == 3
        for i in range(2):
            d, _ = tr.pool[i]
            tr.h.load(tr.records[i])
            with torch.no_grad():
                outs = tr.graphed(tr.h.img_feats, tr.h.pts_feats)
                got = {k: v.clone() for k, v in zip(tr.module.keys, outs)}
                img, pts = tr.enc(d['img_feats'], d['pts_feats'], d['img_metas'], dict(d['pts_metas']))
                ref = tr.dec(pts, img, d['img_metas'])[0][0]
            torch.cuda.synchronize()
            for k in got:
                assert got[k].shape == ref[k].shape and got[k].shape[0] == 2, k
                err = float((got[k] - ref[k].float()).abs().max() / max(1.0, float(ref[k].abs().max())))
                assert err <= 2e-4, (i, k, err)
        losses = [float(tr.step()) for _ in range(3)]
        assert all(math.isfinite(l) for l in losses), losses
        missing = [n for m in (tr.enc, tr.dec) for n, p in m.named_parameters() if p.grad is None]
        assert all(n.startswith('heatmap_head.') for n in missing), missing[:5]
        assert all(torch.isfinite(p.grad).all() for p in tr.params if p.grad is not None)
    finally:
        ops.set_i2p_seed_tensor(None)


@pytest.mark.parametrize('graphed', [False, True])
def test_amp_overflow_step_is_skipped_and_the_loss_scale_backs_off(graphed):
    """Round-4 advice: the mixed-precision step backpropagates fp16 gradients, so it carries dynamic loss scaling and an
    overflow guard (`train_step.LossScaler`).  Injected overflow (loss scale 2^40): the step changes neither the float32
    masters, nor the model's fp16 weights, nor AdamW's step counter / moments, and the scale is halved; the next step at a sane
    scale updates the weights and everything stays finite.  Eager and graph-replayed trainer."""
    from deepinteraction_amd import ops, train_step
    torch.backends.cudnn.deterministic = True
    cls = train_step.GraphedTrainer if graphed else train_step.Trainer
    tr = cls(synth.SHAPE_TINY, 24, torch.device(DEV), 1, pool=1, amp=True)
    try:
        assert tr.scaler is not None and tr._master
        tr.scaler.scale.fill_(1.0)
        tr.step()                                                        # a clean step (round 4 ran at scale 1)
        torch.cuda.synchronize()
        assert float(tr.scaler.skipped) == 0.0
        masters = [m.detach().clone() for m in tr._master]
        halves = [p.detach().clone() for p in tr._half]
        st = tr.opt.state[tr._master[0]]
        n_steps, exp_avg = float(st['step']), st['exp_avg'].clone()
        tr.scaler.scale.fill_(2.0 ** 40)
        tr.step()
        torch.cuda.synchronize()
        assert float(tr.scaler.skipped) == 1.0 and float(tr.scaler.scale) == 2.0 ** 39
        assert all(torch.equal(a, b) for a, b in zip(masters, tr._master))
        assert all(torch.equal(a, b) for a, b in zip(halves, tr._half))
        assert float(st['step']) == n_steps and torch.equal(st['exp_avg'], exp_avg)
        tr.scaler.scale.fill_(2.0)
        loss = tr.step()
        torch.cuda.synchronize()
        assert float(tr.scaler.skipped) == 1.0 and math.isfinite(float(loss))
        assert any(not torch.equal(a, b) for a, b in zip(masters, tr._master))
        assert all(torch.isfinite(m).all() for m in tr._master)
        assert float(st['step']) == n_steps + 1
    finally:
        ops.set_i2p_seed_tensor(None)


def test_autocast_encoder_gradients_close_to_float32():
    """`DI_TRAIN_AMP=1` (opt-in; the reference trains this configuration in float32): the MMRI encoder's training forward under
    torch.autocast(fp16) - 1x1 / 3x3 convolutions and GEMMs in fp16; BatchNorm statistics, soft-max, the window tensors and the
    scatter kernels' accumulation in float32.  Same weights, same dropout masks, a seeded linear functional of the three outputs:
    the outputs agree to fp16 accuracy and every parameter gradient points the same way as the float32 one (cosine >= 0.985 - measured >= 0.991 -,
    norm within 3 %).  (The head is left out on purpose: as initialised it amplifies a 1e-3 input change by 2-3 per RoI block and
    flips Hungarian matches)"""
    from deepinteraction_amd import train_step
    torch.backends.cudnn.deterministic = True
    tr = train_step.Trainer(synth.SHAPE_TINY, 24, torch.device(DEV), 1, pool=1)
    d, _ = tr.pool[0]
    params = dict(tr.enc.named_parameters())

    def grads(amp):
        torch.manual_seed(3)
        torch.cuda.manual_seed(3)
        for p in params.values():
            p.grad = None
        with torch.autocast('cuda', dtype=torch.float16, enabled=amp):
            img, (p0, p1) = tr.enc(d['img_feats'], d['pts_feats'], d['img_metas'], dict(d['pts_metas']))
        gen = torch.Generator().manual_seed(31)
        outs = [t.float() for t in (img, p0, p1)]
        sum((t * torch.randn(t.shape, generator=gen).to(DEV)).sum() for t in outs).backward()
        return [t.detach().clone() for t in outs], {n: p.grad.detach().float().clone() for n, p in params.items() if p.grad is not None}
    o32, g32 = grads(False)
    o16, g16 = grads(True)
    for a, b in zip(o32, o16):
        assert (a - b).abs().max().item() <= 2e-2 * max(1.0, a.abs().max().item())
        assert (a - b).abs().median().item() <= 2e-3 * max(1.0, a.abs().max().item())
    assert set(g16) == set(g32) and len(g32) > 60
    # a bias in front of a BatchNorm (or the key bias of a soft-max) has an exactly-zero gradient: only round-off on both sides
    typical = sorted(a.norm().item() for a in g32.values())[len(g32) // 2]
    worst, bad = 1.0, []
    for n, a in g32.items():
        b = g16[n]
        if a.norm().item() < 1e-3 * typical:
            continue
        cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
        worst = min(worst, cos)
        if cos < 0.985 or abs(b.norm().item() / a.norm().item() - 1.0) > 3e-2:
            bad.append((n, round(cos, 4), a.norm().item(), b.norm().item()))
    assert not bad, bad[:8]
    assert worst < 1.0 - 1e-9            # fp16 really ran somewhere


def test_iou3d_kernel_matches_the_torch_clipper():
    """`di_iou3d_lidar` (the Hungarian cost's IoU, one launch) against the vectorised torch statement of the same
    Sutherland-Hodgman clipper it replaces (`det3d_compat.boxes_iou3d_lidar`, itself tested against known answers in
    tests/test_targets_loss.py): random boxes of 9 columns, identical boxes (IoU 1), touching and disjoint boxes, zero-size
    boxes, 90-degree rotations; empty sides."""
    from deepinteraction_amd import det3d_compat as dc
    g = torch.Generator().manual_seed(3)
    n, m = 1000, 30
    def boxes(k, spread):
        xy = (torch.rand(k, 2, generator=g) - 0.5) * spread
        z = torch.rand(k, 1, generator=g) * 2 - 2.5
        dims = torch.rand(k, 3, generator=g) * 4 + 0.3
        yaw = (torch.rand(k, 1, generator=g) - 0.5) * 12.0
        return torch.cat([xy, z, dims, yaw, torch.randn(k, 2, generator=g)], 1)
    b2 = boxes(m, 20.0)
    b1 = boxes(n, 20.0)
    b1[:m, :7] = b2[:, :7]                                     # identical boxes
    b1[m:2 * m, :7] = b2[:, :7]
    b1[m:2 * m, 6] += math.pi / 2                              # same centre, turned by 90 degrees
    b1[2 * m:3 * m, :7] = b2[:, :7]
    b1[2 * m:3 * m, 0] += b2[:, 3]                             # shifted by the box's own length along x (yaw-dependent overlap)
    b1[3 * m, 3:6] = 0.0                                       # a zero-size box
    b1[3 * m + 1, :2] = 1e4                                    # far away
    ref = dc.boxes_iou3d_lidar(b1[:, :7], b2[:, :7])
    got = ops.iou3d_lidar(b1.cuda(), b2.cuda()).cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 2e-6, float((got - ref).abs().max())
    assert float((got[:m].diag() - 1).abs().max()) <= 1e-5
    assert ops.iou3d_lidar(b1[:0].cuda(), b2.cuda()).shape == (0, m) and ops.iou3d_lidar(b1.cuda(), b2[:0].cuda()).shape == (n, 0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('C,affine,relu', [(128, True, True), (256, True, True), (128, False, False), (64, True, False)])
def test_fused_training_batchnorm_relu_matches_torch(dtype, C, affine, relu):
    """`BatchNormReLU` (csrc/batchnorm.hip: partial sums, fixed-order finalisation, apply) against nn.BatchNorm2d in train()
    mode + nn.ReLU with autograd, float64 statistics as the reference: outputs, input / affine gradients, running mean /
    variance (unbiased) and the batch counter; two steps (the running statistics accumulate); bit-reproducible."""
    from deepinteraction_amd.mmdet3d_plugin.models.utils.encoder_utils import BatchNormReLU
    g = torch.Generator().manual_seed(C + 7)
    n, H, W = 3, 13, 21
    bn = torch.nn.BatchNorm2d(C, affine=affine).to(DEV).train()
    ref = torch.nn.BatchNorm2d(C, affine=affine).to(DEV).double().train()
    if affine:
        with torch.no_grad():
            bn.weight.copy_(torch.rand(C, generator=g) + 0.5), bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
            ref.weight.copy_(bn.weight.double()), ref.bias.copy_(bn.bias.double())
    ws = ops.bn_workspace(C, DEV)
    tol = 2e-3 if dtype == torch.float16 else 2e-5
    for step in range(2):
        x = (torch.randn(n, C, H, W, generator=g) * 1.5 + 0.4).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(n, C, H, W, generator=g).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
        xa = x.clone().requires_grad_(True)
        y = BatchNormReLU.apply(xa, bn.weight, bn.bias, bn, relu, ws)
        y.backward(gy)
        xr = x.double().requires_grad_(True)
        yr = ref(xr)
        yr = torch.relu(yr) if relu else yr
        yr.backward(gy.double())
        rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-12))
        assert y.dtype == dtype and rel(y, yr) <= tol, rel(y, yr)
        assert rel(xa.grad, xr.grad) <= 2 * tol, rel(xa.grad, xr.grad)
        if affine:
            assert rel(bn.weight.grad, ref.weight.grad) <= 2 * tol and rel(bn.bias.grad, ref.bias.grad) <= 2 * tol
            bn.weight.grad = bn.bias.grad = ref.weight.grad = ref.bias.grad = None
        assert rel(bn.running_mean, ref.running_mean) <= 1e-5 and rel(bn.running_var, ref.running_var) <= 1e-5
        assert int(bn.num_batches_tracked) == step + 1
        if step == 1:
            rm = bn.running_mean.clone()
            bn2 = torch.nn.BatchNorm2d(C, affine=affine).to(DEV).train()
            if affine:
                with torch.no_grad():
                    bn2.weight.copy_(bn.weight), bn2.bias.copy_(bn.bias)
            y2 = BatchNormReLU.apply(x, bn2.weight, bn2.bias, bn2, relu, ws)
            y3 = BatchNormReLU.apply(x, bn2.weight, bn2.bias, bn2, relu, ws)
            assert torch.equal(y2, y3) and torch.equal(y2, y.detach())
