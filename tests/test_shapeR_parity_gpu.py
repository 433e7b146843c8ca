"""GPU: parity of the WHOLE hot path at the benched configuration (BASELINE.json configs[1]: Fusion_0075_refactor
shape R - image features 6x256x112x200, BEV 512x180x180, 262 144 points - MMRI encoder 2 layers + MMPI decoder)
against the CPU oracle's full forward, shared state_dict, NO depth injection (the oracle runs its scipy
scatter/completion, the product its device chain).

Cases: fp32 and fp16; Q = 200 (config) and Q = 400 (reference tools/test.py:155 forces num_proposals=400 at test
time); B = 2 (Fusion_0075_refactor.py:94 samples_per_gpu=2); eager and hipGraph replay after `load()` of another
sample.  The measured statistics are written to gpurun_out/parity_shapeR.json; the asserted bounds are the
measured ones with head-room (see DESIGN.md "Numerics").

  fp32: INT outputs (proposals, labels, on-the-image masks) bit-exact; continuous outputs <= 2e-3 of the value
        scale (summation-order noise through ~40 stacked GEMM/LayerNorm stages; a pixel whose projected point
        sits within float round-off of a texel / range boundary may flip: bounded outlier fraction).
  fp16: feature maps stored in fp16 (2^-11 per store), fp32 accumulation.  The synthetic heat map of a random-init
        head is noise - thousands of scores lie within fp16 round-off of the Q-th - so the free-running proposal
        sets differ at the margin (reported, bounded); the continuous outputs are compared against the oracle
        decoder run on the product's proposals.
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from deepinteraction_amd import harness, synth
from deepinteraction_amd.graphed import GraphedHotPath
from oracle import parity

DEV = 'cuda'
SHAPE = synth.SHAPE_R
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}


def _report(name, stats):
    REPORT[name] = stats
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'parity_shapeR.json'), 'w') as f:
        json.dump(REPORT, f, indent=1)


@pytest.fixture(scope='module')
def ctx():
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    torch.backends.cudnn.deterministic = True
    threads = torch.get_num_threads()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    inp = synth.make_inputs(2, SHAPE, seed=100)
    inp['img_feats'] = inp['img_feats'].half().float()          # fp16-representable data: identical for every side
    inp['pts_feats'] = inp['pts_feats'].half().float()
    models = {}
    for q in (200, 400):
        enc, dec = harness.build_models(SHAPE, q, torch.float32, 'cpu')
        models[q] = (enc, dec)
    state = (models[200][0].state_dict(), models[200][1].state_dict())
    E, D200 = parity.build_oracle(SHAPE, 200, state=state)
    _, D400 = parity.build_oracle(SHAPE, 400, state=(state[0], models[400][1].state_dict()))
    ref_enc = parity.oracle_encoder(E, inp)
    c = dict(inp=inp, models=models, state=state, D={200: D200, 400: D400}, ref_enc=ref_enc, free={})
    c['free'][200] = parity.oracle_decoder(D200, ref_enc, inp['img_metas'])
    _report('oracle_seconds', dict(encoder_B2=ref_enc['seconds'], decoder_B2_Q200=c['free'][200]['seconds']))
    yield c
    torch.set_num_threads(threads)        # process-global: leave it as the other test modules found it


def _sample(inp, b):
    """Sample b of a 2-sample batch as a batch of one."""
    pm = inp['pts_metas']
    sel = pm['pillar_coors'][:, 0] == b
    coors = pm['pillar_coors'][sel].clone()
    coors[:, 0] = 0
    return dict(img_feats=inp['img_feats'][6 * b:6 * b + 6], pts_feats=inp['pts_feats'][b:b + 1],
                img_metas=[inp['img_metas'][b]],
                pts_metas=dict(pillars=pm['pillars'][sel], pillar_coors=coors,
                               pillars_num_points=pm['pillars_num_points'][sel], pts=[pm['pts'][b]]))


def _ref_slice(ref_enc, b):
    return dict(img=ref_enc['img'][6 * b:6 * b + 6], pts_conv=ref_enc['pts_conv'][b:b + 1],
                pts=ref_enc['pts'][b:b + 1])


def _product(c, q, dtype):
    enc, dec = c['models'][q]
    import copy
    from deepinteraction_amd import precision
    enc, dec = precision.to_inference(copy.deepcopy(enc).to(DEV), copy.deepcopy(dec).to(DEV), dtype)
    return enc.eval(), dec.eval()          # fp16 = the benched mixed mode: fp16 maps, float32 token path


def _run(enc, dec, d):
    with torch.no_grad():
        (img, pts), out = harness.forward(enc, dec, d)
    torch.cuda.synchronize()
    return (img, pts), out[0][0], dec.query_labels, list(dec.on_the_image_mask), dec.top_proposals


def _check_fp32(name, es, ds):
    for k, s in es.items():
        assert s['p999'] <= 2e-4 and s['frac_gt_1e3'] <= 2e-3, (name, k, s)
    assert ds['proposals_identical'] and ds['label_agreement'] == 1.0, (name, ds['proposal_set_overlap'])
    assert all(m == 1.0 for m in ds['mask_agreement']), (name, ds['mask_agreement'])
    for k, s in ds['keys'].items():
        assert s['max'] <= 2e-3, (name, k, s)


def _check_fp16(name, es, ds, same_params=False):
    """Round 3 (float32 token path): measured against the float32-parameter oracle - encoder maps max 3.8e-4 (median
    4e-5), decoder outputs median 0.7-1.3e-4, p99.9 1.3-3.4e-3, max 5e-3, 0.2-1.2 % beyond 1e-3, none beyond 1e-2 -
    which is the floor the fp16 ENCODER sets: the oracle itself, its encoder weights rounded to fp16 and its maps
    rounded once, puts p99.9 at 1.0-1.9e-3 and 0.1-1 % beyond 1e-3 with exact arithmetic everywhere
    (tests/tools/fp16_error_budget.py).  Bounds = measured with head-room."""
    for k, s in es.items():
        assert s['median'] <= 1e-4 and s['p999'] <= 5e-4 and s['frac_gt_1e3'] <= 1e-5, (name, k, s)
    assert ds['proposal_set_overlap'] >= 0.995, (name, ds['proposal_set_overlap'])
    assert ds['labels_equal_on_same_proposals']
    assert all(m >= 0.995 for m in ds['mask_agreement']), (name, ds['mask_agreement'])
    for k, s in ds['keys'].items():
        if k == 'query_heatmap_score':           # a score is 0 or the heat value: an NMS near-tie flips it whole -
            continue                             # checked entry by entry, tie-aware (_check_score_ties)
        assert s['median'] <= 2.5e-4 and s['p999'] <= 6e-3 and s['frac_gt_1e3'] <= 2.5e-2 and s['frac_gt_1e2'] <= 1e-3, \
            (name, k, s)
    assert ds['keys']['center']['abs_max'] <= 0.05, ds['keys']['center']       # BEV cells (0.6 m each)


def _check_score_ties(name, out, forced):
    """no exemption by name - every differing entry of `query_heatmap_score` must be the
    flip of a cell whose heat value lies within fp16 round-off of its 3x3 maximum (oracle/parity.py::heatmap_score_ties)."""
    t = parity.heatmap_score_ties(out['query_heatmap_score'], forced)
    assert t['flips'] <= 0.005 * t['entries'], (name, t)
    return t


def test_fp32_eager_B2_Q200(ctx):
    enc, dec = _product(ctx, 200, torch.float32)
    d = harness.to_device(ctx['inp'], DEV, torch.float32)
    got_enc, out, labels, masks, top = _run(enc, dec, d)
    es = parity.compare_encoder(got_enc, ctx['ref_enc'])
    ds = parity.compare_decoder(out, labels, masks, top, ctx['free'][200])
    _report('fp32_eager_B2_Q200', dict(encoder=es, decoder=ds))
    _check_fp32('fp32_B2_Q200', es, ds)


def test_fp16_eager_B2_Q200(ctx):
    enc, dec = _product(ctx, 200, torch.float16)
    d = harness.to_device(ctx['inp'], DEV, torch.float16)
    got_enc, out, labels, masks, top = _run(enc, dec, d)
    es = parity.compare_encoder(got_enc, ctx['ref_enc'])
    forced = parity.oracle_decoder(ctx['D'][200], ctx['ref_enc'], ctx['inp']['img_metas'], top_override=top.cpu())
    ds = parity.compare_decoder(out, labels, masks, top, ctx['free'][200], forced)
    ties = _check_score_ties('fp16_B2_Q200', out, forced)
    _report('fp16_eager_B2_Q200', dict(encoder=es, decoder=ds, heatmap_score_ties=ties))
    _check_fp16('fp16_B2_Q200', es, ds)


def test_fp16_identical_parameters_B1_Q200(ctx):
    """The arithmetic alone: the oracle holds the SAME parameters as the fp16 product (encoder and heat-map heads
    rounded through fp16, token path float32 on both sides), sample 0."""
    enc, dec = _product(ctx, 200, torch.float16)
    s0 = _sample(ctx['inp'], 0)
    d = harness.to_device(s0, DEV, torch.float16)
    got_enc, out, labels, masks, top = _run(enc, dec, d)
    E, D = parity.build_oracle(SHAPE, 200, state=ctx['state'], round_fp16='maps')
    ref = parity.oracle_encoder(E, s0)
    es = parity.compare_encoder(got_enc, ref)
    free = parity.oracle_decoder(D, ref, s0['img_metas'])
    forced = parity.oracle_decoder(D, ref, s0['img_metas'], top_override=top.cpu())
    ds = parity.compare_decoder(out, labels, masks, top, free, forced)
    ties = _check_score_ties('fp16_same_params', out, forced)
    _report('fp16_identical_parameters_B1_Q200', dict(encoder=es, decoder=ds, heatmap_score_ties=ties))
    _check_fp16('fp16_same_params', es, ds)


def test_fp16_conditioned_head_B1_Q200(ctx):
    """is the fp16 decoder's 1-3e-3 tail the kernels or the head?  The SAME encoder, the
    same fp16 mixed mode and kernels, but a head whose four RoI blocks are conditioned as in a trained network
    (`harness.condition_head`: residual branches x 0.5; with random-init weights every block multiplies its input error by
    2-3, tests/tools/conditioned_head.py).  Here the contract holds: p99.9 <= 1e-3 and max <= 1.5e-3 on every box output
    (measured in round 4: p99.9 <= 5.7e-4, max <= 9.2e-4).
    The oracle is the float32-parameter one (the comparison that showed the tail on the unconditioned head)."""
    import copy
    enc, dec = ctx['models'][200]
    dec_c = harness.condition_head(copy.deepcopy(dec))
    from deepinteraction_amd import precision
    pe, pd = precision.to_inference(copy.deepcopy(enc).to(DEV), copy.deepcopy(dec_c).to(DEV), torch.float16)
    s0 = _sample(ctx['inp'], 0)
    got_enc, out, labels, masks, top = _run(pe.eval(), pd.eval(), harness.to_device(s0, DEV, torch.float16))
    _, D = parity.build_oracle(SHAPE, 200, state=(ctx['state'][0], dec_c.state_dict()))
    ref0 = _ref_slice(ctx['ref_enc'], 0)
    free = parity.oracle_decoder(D, ref0, s0['img_metas'])
    forced = parity.oracle_decoder(D, ref0, s0['img_metas'], top_override=top.cpu())
    ds = parity.compare_decoder(out, labels, masks, top, free, forced)
    _report('fp16_conditioned_head_B1_Q200', dict(decoder=ds))
    assert ds['proposal_set_overlap'] >= 0.995 and ds['labels_equal_on_same_proposals']
    assert all(m >= 0.995 for m in ds['mask_agreement']), ds['mask_agreement']
    for k, s in ds['keys'].items():
        if k == 'query_heatmap_score':
            continue
        assert s['p999'] <= 1e-3 and s['max'] <= 1.5e-3, (k, s)           # measured: p99.9 <= 5.7e-4, max <= 9.2e-4 (round 4)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_Q400_B1(ctx, dtype):
    """num_proposals = 400 (reference tools/test.py:155), sample 0 alone."""
    enc, dec = _product(ctx, 400, dtype)
    s0 = _sample(ctx['inp'], 0)
    d = harness.to_device(s0, DEV, dtype)
    got_enc, out, labels, masks, top = _run(enc, dec, d)
    ref0 = _ref_slice(ctx['ref_enc'], 0)
    es = parity.compare_encoder(got_enc, ref0)
    free = parity.oracle_decoder(ctx['D'][400], ref0, s0['img_metas'])
    forced = None if dtype == torch.float32 else parity.oracle_decoder(ctx['D'][400], ref0, s0['img_metas'],
                                                                       top_override=top.cpu())
    ds = parity.compare_decoder(out, labels, masks, top, free, forced)
    tag = 'fp32' if dtype == torch.float32 else 'fp16'
    _report(f'{tag}_eager_B1_Q400', dict(encoder=es, decoder=ds))
    (_check_fp32 if dtype == torch.float32 else _check_fp16)(f'{tag}_B1_Q400', es, ds)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_graph_replay_after_load(ctx, dtype):
    """The benched launch mode: capture on sample 1, `load()` sample 0, replay - against the oracle on sample 0,
    and bit-identical to the eager forward on sample 0."""
    enc, dec = _product(ctx, 200, dtype)
    s0, s1 = _sample(ctx['inp'], 0), _sample(ctx['inp'], 1)
    d0, d1 = harness.to_device(s0, DEV, dtype), harness.to_device(s1, DEV, dtype)
    # capacity: the captured sample must hold the loaded one's pillars
    big, small = (d1, d0) if d1['pts_metas']['pillars'].shape[0] >= d0['pts_metas']['pillars'].shape[0] else (d0, d1)
    with torch.no_grad():
        g = GraphedHotPath(enc, dec, big)
        g()
        g.load(small)
        out = g()[0][0]
        torch.cuda.synchronize()
        labels, masks, top = dec.query_labels, list(dec.on_the_image_mask), dec.top_proposals
        got = {k: v.clone() for k, v in out.items()}
        labels, top = labels.clone(), top.clone()
        masks = [m.clone() for m in masks]
    b = 0 if small is d0 else 1
    refb = _ref_slice(ctx['ref_enc'], b)
    sb = s0 if b == 0 else s1
    free = parity.oracle_decoder(ctx['D'][200], refb, sb['img_metas'])
    forced = None if dtype == torch.float32 else parity.oracle_decoder(ctx['D'][200], refb, sb['img_metas'],
                                                                       top_override=top.cpu())
    ds = parity.compare_decoder(got, labels, masks, top, free, forced)
    tag = 'fp32' if dtype == torch.float32 else 'fp16'
    _report(f'{tag}_graph_B1_Q200', dict(decoder=ds))
    if dtype == torch.float32:
        assert ds['proposals_identical'] and all(m == 1.0 for m in ds['mask_agreement'])
        for k, s in ds['keys'].items():
            assert s['max'] <= 2e-3, (k, s)
    else:
        _check_score_ties('fp16_graph', got, forced)
        _check_fp16('fp16_graph', {}, ds)
    # replay == eager on the same sample, bit for bit
    (_, _), eager, elabels, _, etop = _run(enc, dec, small)
    assert torch.equal(etop, top) and torch.equal(elabels, labels)
    for k in got:
        assert torch.equal(got[k].float(), eager[k].float()), k
