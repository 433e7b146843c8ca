"""GPU: parity of the WHOLE DeepInteraction++ forward at the configuration of BASELINE.json configs[4]
(Fusion_0075_plusplus shapes: 2 image levels 6x256x112x200 / 6x256x56x100, BEV 180x180, 262 144 points; neck
`FusionTransformerv4` 2 layers + head `DeepInteractionPlusPlusDecoder`, Q = 200) against the CPU oracle's full forward
(`oracle/plusplus.py`, pinned to the reference's own Python by tests/test_oracle_vs_reference.py::test_pp_*), shared
state_dict, NO depth injection.  Reference: `necks/fusion_transformerv4.py:84-127,161-218`,
`dense_heads/deepinteractionplusplus_decoder.py:201-320`, `projects/configs/nuscenes/Fusion_0075_plusplus.py:210-303`.

Cases: fp32; fp16 (the mixed mode of `precision.half_maps_`: fp16 neck + heat-map heads, float32 token path) against
the oracle with the model's float32 parameters and against the oracle holding the product's parameter VALUES (the
arithmetic alone); hipGraph replay after `load()` bit-identical to eager.  Same statistics as shape R
(tests/test_shapeR_parity_gpu.py); the report goes to gpurun_out/parity_shapePP.json (committed as
profiles/r04_parity_shapePP.json).  Asserted bounds = measured floor with head-room, see `_check_fp16`.
"""
import copy
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from deepinteraction_amd import harness, precision, synth
from deepinteraction_amd.graphed import GraphedHotPath
from oracle import parity

DEV = 'cuda'
SHAPE = synth.SHAPE_PP
Q = 200
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}


def _report(name, stats):
    REPORT[name] = stats
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'parity_shapePP.json'), 'w') as f:
        json.dump(REPORT, f, indent=1)


@pytest.fixture(scope='module')
def ctx():
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    torch.backends.cudnn.deterministic = True
    threads = torch.get_num_threads()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    inp = synth.make_inputs_pp(1, SHAPE, seed=100)
    inp['img_feats'] = [f.half().float() for f in inp['img_feats']]      # fp16-representable data: identical for every side
    inp['pts_feats'] = [f.half().float() for f in inp['pts_feats']]
    other = synth.make_inputs_pp(1, SHAPE, seed=101)
    other['img_feats'] = [f.half().float() for f in other['img_feats']]
    other['pts_feats'] = [f.half().float() for f in other['pts_feats']]
    enc, dec = harness.build_models_pp(SHAPE, Q, torch.float32, 'cpu')
    state = (enc.state_dict(), dec.state_dict())
    E, D = parity.build_oracle_pp(SHAPE, Q, state)
    ref_enc = parity.oracle_encoder_pp(E, inp)
    free = parity.oracle_decoder(D, ref_enc, inp['img_metas'])
    _report('oracle_seconds', dict(neck_B1=ref_enc['seconds'], head_B1_Q200=free['seconds'],
                                   threads=torch.get_num_threads()))
    yield dict(inp=inp, other=other, models=(enc, dec), state=state, D=D, ref_enc=ref_enc, free=free)
    torch.set_num_threads(threads)


def _product(c, dtype):
    enc, dec = c['models']
    enc, dec = precision.to_inference(copy.deepcopy(enc).to(DEV), copy.deepcopy(dec).to(DEV), dtype)
    return enc.eval(), dec.eval()


def _run(enc, dec, d):
    with torch.no_grad():
        (img, pts), out = harness.forward(enc, dec, d)
    torch.cuda.synchronize()
    return (img, pts), out[0][0], dec.query_labels, list(dec.on_the_image_mask), dec.top_proposals


def _check_fp32(name, es, ds):
    for k, s in es.items():
        assert s['p999'] <= 5e-4 and s['frac_gt_1e3'] <= 2e-3, (name, k, s)
    assert ds['proposals_identical'] and ds['label_agreement'] == 1.0, (name, ds['proposal_set_overlap'])
    assert all(m == 1.0 for m in ds['mask_agreement']), (name, ds['mask_agreement'])
    for k, s in ds['keys'].items():
        assert s['max'] <= 2e-3, (name, k, s)


# fp16 bounds = the measured floor (profiles/r04_parity_shapePP.json) with 2x head-room.  Measured:
# neck maps median 7-9e-5, p99.9 5.9-6.3e-4, max 1.2-1.5e-3 of the map's range (fp16 storage of ~30 stacked maps; the
# LayerNorm-ed token maps are O(1) so 2^-11 per store is 5e-4 of the range), nothing beyond 1e-2; head (float32 token
# path on the fp16 maps) median 0.5-1.1e-4, p99.9 4-11e-4, max 1.2e-3.  Round 3 asserted 3e-2 with 0.2 % outliers at TINY.
FP16_ENC = dict(median=2e-4, p999=1.5e-3, frac_gt_1e2=0.0)
FP16_DEC = dict(median=2.5e-4, p999=2.5e-3, frac_gt_1e2=0.0)


def _check_fp16(name, es, ds):
    for k, s in es.items():
        assert s['median'] <= FP16_ENC['median'] and s['p999'] <= FP16_ENC['p999'] and s['max'] <= 5e-3 and \
            s['frac_gt_1e2'] <= FP16_ENC['frac_gt_1e2'], (name, k, s)
    assert ds['proposal_set_overlap'] >= 0.99, (name, ds['proposal_set_overlap'])
    assert ds['labels_equal_on_same_proposals']
    assert all(m >= 0.995 for m in ds['mask_agreement']), (name, ds['mask_agreement'])
    for k, s in ds['keys'].items():
        if k == 'query_heatmap_score':           # a score is 0 or the heat value: an NMS near-tie flips it whole
            assert s['frac_gt_1e3'] <= 2e-2, (name, k, s)
            continue
        assert s['median'] <= FP16_DEC['median'] and s['p999'] <= FP16_DEC['p999'] and s['max'] <= 5e-3 and \
            s['frac_gt_1e2'] <= FP16_DEC['frac_gt_1e2'], (name, k, s)


def test_fp32_eager(ctx):
    enc, dec = _product(ctx, torch.float32)
    got_enc, out, labels, masks, top = _run(enc, dec, harness.to_device_pp(ctx['inp'], DEV, torch.float32))
    es = parity.compare_encoder(got_enc, ctx['ref_enc'])
    ds = parity.compare_decoder(out, labels, masks, top, ctx['free'])
    _report('fp32_eager_B1_Q200', dict(encoder=es, decoder=ds))
    _check_fp32('fp32', es, ds)


def test_fp16_eager(ctx):
    enc, dec = _product(ctx, torch.float16)
    got_enc, out, labels, masks, top = _run(enc, dec, harness.to_device_pp(ctx['inp'], DEV, torch.float16))
    es = parity.compare_encoder(got_enc, ctx['ref_enc'])
    forced = parity.oracle_decoder(ctx['D'], ctx['ref_enc'], ctx['inp']['img_metas'], top_override=top.cpu())
    ds = parity.compare_decoder(out, labels, masks, top, ctx['free'], forced)
    # the head alone: the oracle head on the PRODUCT's fp16 neck outputs and proposals (what the fp16 neck costs the
    # head is the difference between this and the row above)
    nchw = lambda t: t.float().cpu().contiguous()            # (the product's maps are channels-last)
    own = dict(img=nchw(got_enc[0]), pts_conv=nchw(got_enc[1][0]), pts=nchw(got_enc[1][1]))
    forced_own = parity.oracle_decoder(ctx['D'], own, ctx['inp']['img_metas'], top_override=top.cpu())
    hs = parity.compare_decoder(out, labels, masks, top, forced_own, forced_own)
    # query_heatmap_score entry by entry, tie-aware (no exemption by name): against the oracle head on the product's OWN maps -
    # the heat-map heads' arithmetic alone decides which cell of a near-tie survives the NMS
    ties = parity.heatmap_score_ties(out['query_heatmap_score'], forced_own, tol=3e-3)
    assert ties['flips'] <= 0.01 * ties['entries'], ties
    _report('fp16_eager_B1_Q200', dict(encoder=es, decoder=ds, head_on_product_maps=hs, heatmap_score_ties=ties))
    _check_fp16('fp16', es, ds)
    for k, s in hs['keys'].items():              # the mixed-mode head itself is float32-accurate
        if k != 'query_heatmap_score':
            assert s['max'] <= 2e-3, ('head on its own maps', k, s)


def test_fp16_identical_parameters(ctx):
    """The arithmetic alone: the oracle holds the SAME parameter values as the fp16 product (neck and heat-map heads
    rounded through fp16, token path float32 on both sides)."""
    enc, dec = _product(ctx, torch.float16)
    got_enc, out, labels, masks, top = _run(enc, dec, harness.to_device_pp(ctx['inp'], DEV, torch.float16))
    E, D = parity.build_oracle_pp(SHAPE, Q, ctx['state'], round_fp16='maps')
    ref = parity.oracle_encoder_pp(E, ctx['inp'])
    es = parity.compare_encoder(got_enc, ref)
    free = parity.oracle_decoder(D, ref, ctx['inp']['img_metas'])
    forced = parity.oracle_decoder(D, ref, ctx['inp']['img_metas'], top_override=top.cpu())
    ds = parity.compare_decoder(out, labels, masks, top, free, forced)
    ties = parity.heatmap_score_ties(out['query_heatmap_score'], forced, tol=3e-3)
    assert ties['flips'] <= 0.01 * ties['entries'], ties
    _report('fp16_identical_parameters_B1_Q200', dict(encoder=es, decoder=ds, heatmap_score_ties=ties))
    _check_fp16('fp16_same_params', es, ds)


def test_fp16_conditioned_head(ctx):
    """Round-4 verdict, item 1(c): the ++ head's fp16 tail (`height` max 1.96e-3 / p99.9 1.57e-3 against the float32-parameter
    oracle, profiles/r04_parity_shapePP.json) - the kernels or the random-init head's gain?  The SAME neck, mixed mode and
    kernels with a head whose four V2 RoI blocks are conditioned as in a trained network (`harness.condition_head`: the
    residual branches x 0.5 on BOTH sides, as in tests/test_shapeR_parity_gpu.py::test_fp16_conditioned_head_B1_Q200).
    Measured (session r05a): every box output max <= 3.4e-4, p99.9 <= 3.1e-4.  Asserted: max <= 1e-3 - the `north_star` tolerance."""
    enc, dec = ctx['models']
    dec_c = harness.condition_head(copy.deepcopy(dec))
    pe, pd = precision.to_inference(copy.deepcopy(enc).to(DEV), copy.deepcopy(dec_c).to(DEV), torch.float16)
    got_enc, out, labels, masks, top = _run(pe.eval(), pd.eval(), harness.to_device_pp(ctx['inp'], DEV, torch.float16))
    _, D = parity.build_oracle_pp(SHAPE, Q, (ctx['state'][0], dec_c.state_dict()))
    free = parity.oracle_decoder(D, ctx['ref_enc'], ctx['inp']['img_metas'])
    forced = parity.oracle_decoder(D, ctx['ref_enc'], ctx['inp']['img_metas'], top_override=top.cpu())
    ds = parity.compare_decoder(out, labels, masks, top, free, forced)
    _report('fp16_conditioned_head_B1_Q200', dict(decoder=ds))
    assert ds['proposal_set_overlap'] >= 0.99 and ds['labels_equal_on_same_proposals']
    assert all(m >= 0.995 for m in ds['mask_agreement']), ds['mask_agreement']
    for k, s in ds['keys'].items():
        if k == 'query_heatmap_score':
            continue
        assert s['p999'] <= 6e-4 and s['max'] <= 1e-3, (k, s)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_graph_replay_after_load(ctx, dtype):
    """The benched launch mode: capture on one sample, `load()` the other, replay - bit-identical to the eager forward
    of the loaded sample; fp32 also against the oracle."""
    enc, dec = _product(ctx, dtype)
    d0 = harness.to_device_pp(ctx['inp'], DEV, dtype)
    d1 = harness.to_device_pp(ctx['other'], DEV, dtype)
    big, small = (d1, d0) if d1['pts_metas']['pillars'].shape[0] >= d0['pts_metas']['pillars'].shape[0] else (d0, d1)
    with torch.no_grad():
        g = GraphedHotPath(enc, dec, big)
        g()
        g.load(small)
        out = g()[0][0]
        torch.cuda.synchronize()
        got = {k: v.clone() for k, v in out.items()}
        labels, top = dec.query_labels.clone(), dec.top_proposals.clone()
        masks = [m.clone() for m in dec.on_the_image_mask]
    (_, _), eager, elabels, emasks, etop = _run(enc, dec, small)
    assert torch.equal(etop, top) and torch.equal(elabels, labels)
    for a, b in zip(masks, emasks):
        assert torch.equal(a, b)
    for k in got:
        assert torch.equal(got[k].float(), eager[k].float()), k
    if small is d0 and dtype == torch.float32:
        ds = parity.compare_decoder(got, labels, masks, top, ctx['free'])
        _report('fp32_graph_B1_Q200', dict(decoder=ds))
        assert ds['proposals_identical']
        for k, s in ds['keys'].items():
            assert s['max'] <= 2e-3, (k, s)
