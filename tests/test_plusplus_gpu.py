"""DeepInteraction++ on the MI355X: the four ++ kernels against their CPU restatements,
the neck and the head against the oracle (shared state_dict) and against the golden vectors the REFERENCE's own
code produced (tests/golden/modules_pp.npz).  fp32 tolerances cover summation order only; fp16 stores every map
in half precision (2^-11 relative per store) through ~30 stages."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from deepinteraction_amd import configs, synth
from oracle import plusplus as opp
from oracle import thirdparty as tp
from oracle.refpin import make_golden as mg

pytestmark = pytest.mark.gpu
DEV = 'cuda'
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.float16, 4e-3)])
@pytest.mark.parametrize('levels', [[(9, 13)], [(12, 20), (6, 10)]])
def test_ms_deform_attn_kernel(dtype, tol, levels):
    from deepinteraction_amd import ops
    g = torch.Generator().manual_seed(3)
    bs, L, P, nq = 3, len(levels), 4, 217
    S = sum(h * w for h, w in levels)
    value = torch.randn(bs, S, 128, generator=g)
    off = torch.randn(bs, nq, 8 * L * P * 2, generator=g) * 3.0            # many samples leave the map
    logit = torch.randn(bs, nq, 8 * L * P, generator=g) * 2.0
    ref = torch.rand(1, nq, 1, 2, generator=g).repeat(1, 1, L, 1).contiguous()
    vq, oq, lq = value.to(dtype), off.to(dtype), logit.to(dtype)
    packed = torch.cat([oq, lq], -1).to(DEV)                                # one GEMM output, strided views into it
    n_off = off.shape[-1]
    out = ops.ms_deform_attn(vq.to(DEV), packed[..., :n_off], packed[..., n_off:], ref.to(DEV), levels)
    w = lq.float().view(bs, nq, 8, L * P).softmax(-1).view(bs, nq, 8, L, P)
    norm = torch.tensor([[w_, h_] for h_, w_ in levels], dtype=torch.float32)
    loc = ref[:, :, None, :, None, :] + oq.float().view(bs, nq, 8, L, P, 2) / norm[None, None, None, :, None, :]
    want = tp.ms_deform_attn_core(vq.float().view(bs, S, 8, 16), levels, loc, w)
    d = (out.float().cpu() - want).abs().max().item()
    assert d <= tol * max(1.0, want.abs().max().item()), d
    # per-batch reference points take the other addressing path
    ref_b = ref.repeat(bs, 1, 1, 1).contiguous()
    out_b = ops.ms_deform_attn(vq.to(DEV), packed[..., :n_off], packed[..., n_off:], ref_b.to(DEV), levels)
    assert torch.equal(out_b, out)


@pytest.mark.parametrize('levels', [[(9, 13)], [(12, 20), (6, 10)], [(112, 200), (56, 100)]])
def test_ms_deform_attn_head_major_kernel(levels):
    """Round 5: the inference form that gathers from a HEAD-MAJOR value map (bs, 8, S, 16) - lanes = (query, head, corner
    column, channel half), a quad's geometry shared by DPP broadcasts - against the CPU restatement of mmcv's core, the value
    map produced by the projection kernel that writes that layout (`ops.token_linear_hm`), many samples outside the map,
    shared and per-batch reference points; and the head-major projection itself against the channels-last one (bit for bit,
    same arithmetic, other addresses)."""
    from deepinteraction_amd import ops
    g = torch.Generator().manual_seed(5)
    bs, L, P = 3, len(levels), 4
    nq = levels[0][0] * levels[0][1] if levels[0][0] > 100 else 217
    S = sum(h * w for h, w in levels)
    x = torch.randn(bs * S, 128, generator=g).half().to(DEV)
    wv = (torch.randn(128, 128, generator=g) * 0.1).half().to(DEV)
    bv = (torch.randn(128, generator=g) * 0.1).to(DEV)
    v_cl = ops.token_linear(x, wv, bv).view(bs, S, 128)
    v_hm = ops.token_linear_hm(x, wv, bv, S)
    assert v_hm.shape == (bs, 8, S, 16)
    assert torch.equal(v_hm.permute(0, 2, 1, 3).reshape(bs, S, 128), v_cl)
    off = (torch.randn(bs, nq, 8 * L * P * 2, generator=g) * 3.0).half()    # many samples leave the map
    logit = (torch.randn(bs, nq, 8 * L * P, generator=g) * 2.0).half()
    ref = torch.rand(1, nq, 1, 2, generator=g).repeat(1, 1, L, 1).contiguous()
    n_off = off.shape[-1]
    packed = torch.zeros(bs, nq, 256, dtype=torch.float16)                   # strided views into a wider row
    packed[..., :n_off], packed[..., n_off:n_off + logit.shape[-1]] = off, logit
    packed = packed.to(DEV)
    o_v, l_v = packed[..., :n_off], packed[..., n_off:n_off + logit.shape[-1]]
    out = ops.ms_deform_attn(v_hm, o_v, l_v, ref.to(DEV), levels, head_major=True)
    w = logit.float().view(bs, nq, 8, L * P).softmax(-1).view(bs, nq, 8, L, P)
    norm = torch.tensor([[w_, h_] for h_, w_ in levels], dtype=torch.float32)
    loc = ref[:, :, None, :, None, :] + off.float().view(bs, nq, 8, L, P, 2) / norm[None, None, None, :, None, :]
    want = tp.ms_deform_attn_core(v_cl.float().cpu().view(bs, S, 8, 16), levels, loc, w)
    d = (out.float().cpu() - want).abs().max().item()
    assert d <= 4e-3 * max(1.0, want.abs().max().item()), d
    old = ops.ms_deform_attn(v_cl.contiguous(), o_v, l_v, ref.to(DEV), levels)      # the channels-last kernel: same maths
    assert (out.float() - old.float()).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item())
    ref_b = ref.repeat(bs, 1, 1, 1).contiguous()
    out_b = ops.ms_deform_attn(v_hm, o_v, l_v, ref_b.to(DEV), levels, head_major=True)
    assert torch.equal(out_b, out)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-5), (torch.float16, 2e-3)])
def test_grid_gather_kernel(dtype, tol):
    from deepinteraction_amd import ops
    g = torch.Generator().manual_seed(4)
    feat = torch.randn(2, 128, 11, 17, generator=g).to(dtype)
    grid = torch.rand(6, 301, 2, generator=g) * 2.6 - 1.3                   # partly outside
    add = torch.randn(301, 128, generator=g).to(dtype)
    out = ops.grid_gather(feat.to(DEV).contiguous(memory_format=torch.channels_last), grid.to(DEV), add.to(DEV), 3)
    want = F.grid_sample(feat.float().repeat_interleave(3, 0), grid[:, :, None, :], mode='bilinear',
                         padding_mode='zeros', align_corners=False)[..., 0].permute(0, 2, 1) + add.float()
    assert (out.float().cpu() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    out2 = ops.grid_gather(feat.to(DEV).contiguous(memory_format=torch.channels_last), grid.to(DEV), None, 3)
    assert (out2.float().cpu() - (want - add.float())).abs().max().item() <= tol * 4


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.float16, 3e-3)])
@pytest.mark.parametrize('T,S', [(60, 60), (60, 112), (7, 5), (70, 130), (33, 64), (16, 128), (65, 65)])
def test_mha_small_kernel(dtype, tol, T, S):
    from deepinteraction_amd import ops
    g = torch.Generator().manual_seed(5)
    N, E, H = 9, 128, 8
    qkv = torch.randn(N, T, 3 * E, generator=g).to(dtype)
    kv = torch.randn(N, S, 2 * E, generator=g).to(dtype)
    qd, kvd = qkv.to(DEV), kv.to(DEV)
    out = ops.mha_small(qd[..., :E], kvd[..., :E], kvd[..., E:], H)
    q = qkv[..., :E].float().view(N, T, H, 16).transpose(1, 2)
    k = kv[..., :E].float().view(N, S, H, 16).transpose(1, 2)
    v = kv[..., E:].float().view(N, S, H, 16).transpose(1, 2)
    want = (torch.softmax(q @ k.transpose(-1, -2) * 0.25, -1) @ v).transpose(1, 2).reshape(N, T, E)
    assert (out.float().cpu() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-6), (torch.float16, 2e-3)])
@pytest.mark.parametrize('C,res', [(128, True), (128, False), (64, True)])
def test_add_layernorm_kernel(dtype, tol, C, res):
    from deepinteraction_amd import ops
    g = torch.Generator().manual_seed(6)
    x = (torch.randn(3, 1001, C, generator=g) * 3 + 1).to(dtype)
    r = torch.randn(3, 1001, C, generator=g).to(dtype) if res else None
    w, b = (1 + 0.2 * torch.randn(C, generator=g)).to(dtype), (0.1 * torch.randn(C, generator=g)).to(dtype)
    out = ops.add_layernorm(x.to(DEV), None if r is None else r.to(DEV), w.to(DEV), b.to(DEV), 1e-5)
    want = F.layer_norm(x.float() + (r.float() if res else 0), (C,), w.float(), b.float(), 1e-5)
    assert (out.float().cpu() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())


def _pp_pair(cls_ref_builder, aug, dtype):
    from deepinteraction_amd.mmdet3d_plugin import FusionTransformerv4
    O, inp = mg.encoder_pp_case(opp.FusionTransformerv4, aug)
    shape = synth.SHAPE_PP_TINY
    M = FusionTransformerv4(**configs.encoder_pp_cfg(shape['c_img'], shape['c_pts']))
    M.load_state_dict(O.state_dict())
    return O, M.eval().to(DEV, dtype), _inject_depth(inp)


def _inject_depth(inp):
    """Dense depth from the oracle's completion (pinned bit-exactly to the reference's fill_in_multiscale) for
    both sides, so that the comparison does not hinge on float rounding inside the device completion (which has
    its own parity test in tests/test_encoder_gpu.py)."""
    from oracle import encoder as oenc
    Hi, Wi = inp['img_feats'][0].shape[-2:]
    l2i = oenc.lidar2img_tensor(inp['img_metas'], inp['img_feats'][0])
    dense = [oenc.complete_depth(oenc.BEVWarp().sparse_depth(inp['pts_metas']['pts'][b], inp['img_metas'][b], l2i[b],
                                                            Hi, Wi)) for b in range(len(inp['img_metas']))]
    inp['pts_metas'] = dict(inp['pts_metas'], dense_depth=torch.stack(dense))
    return inp


def _to_dev(inp, dtype, inject_depth=None):
    pm = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items() if k != 'dense_depth'}
    pm['pts'] = [p.to(DEV) for p in inp['pts_metas']['pts']]
    if 'dense_depth' in inp['pts_metas']:
        inject_depth(pm['pts'], inp['pts_metas']['dense_depth'])       # test-only fixture (tests/conftest.py)
    return ([f.to(DEV, dtype) for f in inp['img_feats']], [f.to(DEV, dtype) for f in inp['pts_feats']],
            inp['img_metas'], pm)


@pytest.mark.parametrize('aug', [False, True])
def test_polar_attention_matches_oracle(aug):
    from deepinteraction_amd.mmdet3d_plugin import MMRI_I2P_Polar
    shape = synth.SHAPE_PP_TINY
    inp = synth.make_inputs_pp(2, shape, seed=4, aug=synth.example_aug(2) if aug else None)
    torch.manual_seed(3)
    O = opp.MMRI_I2P_Polar(128, 0.1).eval()
    mg.randomize(O, 21)
    M = MMRI_I2P_Polar(128, 0.1).eval()
    M.load_state_dict(O.state_dict())
    g = torch.Generator().manual_seed(1)
    bev = torch.randn(2, 128, *shape['bev_hw'], generator=g)
    img = torch.randn(12, 128, *shape['img_hw'], generator=g)
    with torch.no_grad():
        want = O(bev, img, inp['img_metas'], inp['pts_metas'])
        got = M.to(DEV)(bev.to(DEV), img.to(DEV), inp['img_metas'], {})
    d = (got.float().cpu() - want).abs()
    assert d.max().item() <= 2e-4 * max(1.0, want.abs().max().item()), d.max().item()
    assert (want - bev).abs().max() > 0.1


@pytest.mark.parametrize('aug', [False, True])
@pytest.mark.parametrize('dtype,tol', [(torch.float32, 5e-4), (torch.float16, 3e-2)])
def test_pp_neck_matches_oracle(dtype, tol, aug, inject_depth):
    torch.backends.cudnn.deterministic = True
    O, M, inp = _pp_pair(None, aug, dtype)
    q = lambda ts: [t.to(dtype).float() for t in ts]
    with torch.no_grad():
        ri, (rp0, rp1) = O(q(inp['img_feats']), q(inp['pts_feats']), inp['img_metas'], inp['pts_metas'])
        gi, (gp0, gp1) = M(*_to_dev(inp, dtype, inject_depth))
    for name, got, ref in (('img', gi, ri), ('pts_conv', gp0, rp0), ('pts', gp1, rp1)):
        got = got.float().cpu()
        assert got.shape == ref.shape
        scale = max(ref.abs().max().item(), 1.0)
        d = (got - ref).abs()
        frac_bad = (d > tol * scale).float().mean().item()
        assert frac_bad <= 2e-3, (name, frac_bad, d.max().item(), scale)


@pytest.mark.parametrize('aug', [False, True])
def test_pp_neck_matches_reference_golden(aug, inject_depth):
    """The product neck (fp32) against vectors produced by the reference's own FusionTransformerv4."""
    torch.backends.cudnn.deterministic = True
    g = np.load(os.path.join(GOLD, 'modules_pp.npz'))
    _, M, inp = _pp_pair(None, aug, torch.float32)
    with torch.no_grad():
        img, (p0, p1) = M(*_to_dev(inp, torch.float32, inject_depth))
    for name, t in (('img', img), ('pts_conv', p0), ('pts', p1)):
        s = mg.summarize(t.float().cpu().contiguous())
        pre = f'enc{int(aug)}_{name}_'
        scale = max(1.0, float(np.abs(g[pre + 'sample']).max()))
        bad = (np.abs(s['sample'] - g[pre + 'sample']) > 5e-4 * scale).mean()
        assert bad <= 2e-3, (name, bad)
        assert abs(s['abssum'] - g[pre + 'abssum']) <= 1e-3 * abs(g[pre + 'abssum']), name


def test_pp_head_matches_oracle_and_golden():
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionPlusPlusDecoder
    O, (pts, img, metas) = mg.decoder_pp_case(opp.DeepInteractionPlusPlusDecoder)
    M = DeepInteractionPlusPlusDecoder(**configs.decoder_cfg(bev=36, num_proposals=40))
    M.load_state_dict(O.state_dict())
    M = M.eval().to(DEV)
    with torch.no_grad():
        want = O(pts, img, metas)[0][0]
        got = M([p.to(DEV) for p in pts], img.to(DEV), metas)[0][0]
    gold = np.load(os.path.join(GOLD, 'modules_pp.npz'))
    assert torch.equal(M.query_labels.cpu(), O.query_labels)                     # INT: bit-exact
    assert np.array_equal(M.query_labels.cpu().numpy(), gold['dec_query_labels'])
    assert len(M.on_the_image_mask) == 4
    for a, b in zip(M.on_the_image_mask, O.on_the_image_mask):
        assert torch.equal(a.cpu(), b)
    assert np.array_equal(torch.stack(M.on_the_image_mask).cpu().numpy(), gold['dec_on_the_image_mask'])
    for k in want:
        a = got[k].float().cpu()
        assert a.shape == want[k].shape, k
        scale = max(1.0, want[k].abs().max().item())
        assert (a - want[k]).abs().max().item() <= 2e-3 * scale, (k, (a - want[k]).abs().max().item())
        assert np.abs(a.numpy() - gold['dec_' + k]).max() <= 2e-3 * scale, k


def test_pp_head_fused_token_path_matches_oracle():
    """Round 4: the ++ head on the float32 token kernels (decoder_fused.FusedDecoder: V2 RoI blocks - main branch on the
    token programs with the mmcv FFN, the self branch of the first query per view / per sample, the look-forward centre
    update and the cumulative on-the-image mask in the heads step).  fp16 maps in, float32 token path: the oracle sees the
    same fp16-representable maps and float32 parameters (heat-map heads rounded through fp16 as `half_maps_` holds them)."""
    from deepinteraction_amd import decoder_fused, precision
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionPlusPlusDecoder
    O, (pts, img, metas) = mg.decoder_pp_case(opp.DeepInteractionPlusPlusDecoder)
    for m in (O.heatmap_head, O.heatmap_head_img):
        for t in list(m.parameters()) + list(m.buffers()):
            if t.is_floating_point():
                t.data = t.data.half().float()
    pts, img = [p.half().float() for p in pts], img.half().float()
    M = DeepInteractionPlusPlusDecoder(**configs.decoder_cfg(bev=36, num_proposals=40))
    M.load_state_dict(O.state_dict())
    M = M.eval().to(DEV)
    M.float()
    M.heatmap_head.half()
    M.heatmap_head_img.half()
    cl = lambda t: t.to(DEV, torch.float16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert decoder_fused.usable(M, cl(pts[0]), cl(img))
        got = M([cl(p) for p in pts], cl(img), metas)[0][0]
        free_top = O(pts, img, metas) and O.top_proposals.clone()
        # fp16 heat-map convolutions may reorder near-tied scores: the same SET of proposals, the oracle then runs on the
        # product's order (`top_override`, an oracle-side switch)
        assert sorted(M.top_proposals.cpu().flatten().tolist()) == sorted(free_top.flatten().tolist())
        want = O(pts, img, metas, top_override=M.top_proposals.cpu())[0][0]
    assert M._fused_path is not None                                                  # the fused path really ran
    assert torch.equal(M.query_labels.cpu(), O.query_labels)
    assert len(M.on_the_image_mask) == 4
    for a, b in zip(M.on_the_image_mask, O.on_the_image_mask):
        assert torch.equal(a.cpu(), b)
    for k in want:
        a = got[k].float().cpu()
        assert a.shape == want[k].shape, k
        scale = max(1.0, want[k].abs().max().item())
        d = (a - want[k]).abs() / scale
        if k in ('dense_heatmap', 'query_heatmap_score'):                          # fp16 heat-map convolutions
            assert d.max().item() <= 1e-3, (k, d.max().item())
            continue
        # float32 token path (2^-22 per product) under a random-init head whose five stages each multiply an input difference
        # by ~3 at this toy size (36 x 36 BEV cells, 16 x 28 image maps): measured median 1.3e-4, single queries of the last
        # stages up to 3e-2 when a RoI-align sample point crosses a bin border.  The full-size statement is
        # tests/test_shapePP_parity_gpu.py: the head on its own maps max <= 2e-3, `center` 1.2e-5.
        assert d.median().item() <= 1e-3 and (d > 1e-2).float().mean().item() <= 2e-2 and d.max().item() <= 1e-1, \
            (k, d.median().item(), d.max().item())


def test_v2_block_first_query_quirk():
    """Direct check of the published broadcasting behaviour the product reproduces: with the self branch switched
    on alone (scale 0, self_scale 1) every query of a sample gets the SAME vector out of the point block."""
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionPlusPlusDecoder
    M = DeepInteractionPlusPlusDecoder(**configs.decoder_cfg(bev=36, num_proposals=16)).eval().to(DEV)
    blk = M.decode_head[1]
    with torch.no_grad():
        blk.scale.fill_(0.0)
        blk.self_scale.fill_(1.0)
        x = torch.randn(2, 16, 128, device=DEV)
        roi = torch.randn(32, 49, 128, device=DEV)
        out = blk._refine_all(x, roi)
    assert torch.allclose(out, out[:, :1].expand_as(out)) and not torch.allclose(out[0], out[1])


@pytest.mark.parametrize('M,hidden', [(1000, 512), (37, 128), (134400, 512)])
def test_fused_ffn_layernorm(M, hidden):
    """ops.ffn_ln (FFN + residual + LayerNorm in one kernel, the hidden activation kept in registers chunk by chunk)
    against the float64 evaluation on the same fp16 operands (the kernel rounds the hidden activation to fp16, as the
    unfused fp16 path does): 1e-3 of the output scale (outputs are normalised: O(1))."""
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    from deepinteraction_amd import ops
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, 128, generator=g) * 0.7).half()
    w1 = (torch.randn(hidden, 128, generator=g) / 128 ** 0.5).half()
    b1 = (torch.randn(hidden, generator=g) * 0.1).half()
    w2 = (torch.randn(128, hidden, generator=g) / hidden ** 0.5).half()
    b2 = (torch.randn(128, generator=g) * 0.1).half()
    lw, lb = (1 + 0.1 * torch.randn(128, generator=g)).half(), (0.1 * torch.randn(128, generator=g)).half()
    dev = 'cuda'
    imgs = ops.ffn_images(w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev))
    got = ops.ffn_ln(x.to(dev), imgs, lw.to(dev), lb.to(dev), 1e-5).double().cpu()
    xd = x.to(dev).double()
    h = (xd @ w1.to(dev).double().t() + b1.to(dev).double()).relu().half().double()
    o = xd + h @ w2.to(dev).double().t() + b2.to(dev).double()
    ref = torch.nn.functional.layer_norm(o, (128,), lw.to(dev).double(), lb.to(dev).double(), 1e-5).cpu()
    err = (got - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err


def test_fused_linear_layernorm():
    """ops.linear_ln: LayerNorm(residual + x W^T + b) in one kernel (output projection of the deformable attention +
    its post-norm) against float64."""
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    from deepinteraction_amd import ops
    g = torch.Generator().manual_seed(3)
    M = 4099
    x, res = (torch.randn(M, 128, generator=g) * 0.7).half().cuda(), (torch.randn(M, 128, generator=g)).half().cuda()
    w, b = (torch.randn(128, 128, generator=g) / 11).half().cuda(), (torch.randn(128, generator=g) * 0.1).half().cuda()
    lw, lb = (1 + 0.1 * torch.randn(128, generator=g)).half().cuda(), (0.1 * torch.randn(128, generator=g)).half().cuda()
    got = ops.linear_ln(x, ops.chain_image(w.float(), b.float()), res, lw, lb, 1e-5).double()
    ref = torch.nn.functional.layer_norm(res.double() + x.double() @ w.double().t() + b.double(), (128,), lw.double(),
                                         lb.double(), 1e-5)
    err = (got - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize('num_levels,num_points', [(4, 4), (2, 8), (1, 2)])
def test_msda_module_outside_the_fused_form_takes_the_generic_kernels(num_levels, num_points):
    """the deformable-attention kernels - fused fp16 form AND the generic one, as this test found -
    cover 4 points on 1 or 2 levels (`kernels_cover` / `can_fuse_tokens`); mmcv's default num_levels=4, or other point
    counts, must not raise: they run mmcv's grid_sample formulation on torch ops (`_core_torch`, checked on the CPU against
    the oracle's restatement of mmcv's core), fp16 inference agreeing with the float32 run of the same module."""
    from deepinteraction_amd.mmdet3d_plugin.models.utils.transformer_bricks import MultiScaleDeformableAttention
    torch.manual_seed(3)
    m = MultiScaleDeformableAttention(128, num_heads=8, num_levels=num_levels, num_points=num_points, batch_first=True).to(DEV).eval()
    with torch.no_grad():
        m.sampling_offsets.weight.add_(torch.randn_like(m.sampling_offsets.weight) * 0.05)
        m.attention_weights.weight.add_(torch.randn_like(m.attention_weights.weight) * 0.05)
    shapes = [(12, 20), (6, 10), (3, 5), (2, 3)][:num_levels]
    S = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(1)
    q = torch.randn(2, 240, 128, generator=g).to(DEV)
    v = torch.randn(2, S, 128, generator=g).to(DEV)
    ref = torch.rand(2, 240, num_levels, 2, generator=g).to(DEV)
    with torch.no_grad():
        want = m(q, value=v, reference_points=ref, spatial_shapes=shapes)
        mh = MultiScaleDeformableAttention(128, num_heads=8, num_levels=num_levels, num_points=num_points, batch_first=True)
        mh.load_state_dict(m.state_dict())
        mh = mh.to(DEV).half().eval()
        assert not mh.can_fuse_tokens(q.half()) and not mh.kernels_cover()
        got = mh(q.half(), value=v.half(), reference_points=ref, spatial_shapes=shapes)
    assert got.dtype == torch.float16
    assert (got.float() - want).abs().max().item() <= 2e-2 * max(1.0, want.abs().max().item())
