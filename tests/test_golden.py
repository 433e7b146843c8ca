"""Oracle vs the committed golden vectors, which were produced by the REFERENCE'S OWN code
(oracle/refpin/make_golden.py).  CPU only; runs everywhere (also on the GPU box, where
/root/reference does not exist)."""
import os

import numpy as np
import pytest
import torch

from oracle import decoder as odec, depth_completion, encoder as oenc
from oracle.locatt import CLocatt, TorchLocatt
from oracle.refpin import make_golden as mg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_locatt_port_matches_reference_kernels_bitwise(oracle_libs):
    g = np.load(os.path.join(GOLD, 'locatt.npz'))
    a, b, w = mg.locatt_inputs()
    port = CLocatt('port')
    got = dict(similar_forward=port.similar_forward(a, b, 9, 9),
               weighting_forward=port.weighting_forward(a, w, 9, 9),
               similar_backward_ori=port.similar_backward(a, w, 9, 9, True),
               similar_backward_loc=port.similar_backward(a, w, 9, 9, False),
               weighting_backward_ori=port.weighting_backward_ori(w, a, 9, 9),
               weighting_backward_weight=port.weighting_backward_weight(a, b, 9, 9))
    for k, v in got.items():
        assert np.array_equal(v.numpy(), g[k]), k                       # bit-exact (double accumulation)
        t = getattr(TorchLocatt, k.replace('_ori', '').replace('_loc', '')) if False else None
    # the vectorised torch form, to float32 round-off
    assert np.allclose(TorchLocatt.similar_forward(a, b, 9, 9).numpy(), g['similar_forward'], atol=2e-5)
    assert np.allclose(TorchLocatt.weighting_forward(a, w, 9, 9).numpy(), g['weighting_forward'], atol=2e-5)


def test_depth_completion_matches_reference_bitwise():
    g = np.load(os.path.join(GOLD, 'depth_completion.npz'))['dense']
    sparse = mg.sparse_depth_input()
    for v in range(sparse.shape[0]):
        assert np.array_equal(depth_completion.fill_in_multiscale(sparse[v].copy()), g[v])


@pytest.mark.parametrize('aug', [False, True])
def test_encoder_matches_reference_golden(aug):
    g = np.load(os.path.join(GOLD, 'modules.npz'))
    m, inp = mg.encoder_case(oenc.DeepInteractionEncoder, aug)
    with torch.no_grad():
        img, (p0, p1) = m(inp['img_feats'], inp['pts_feats'], inp['img_metas'], inp['pts_metas'])
    for name, t in (('img', img), ('pts_conv', p0), ('pts', p1)):
        s = mg.summarize(t)
        pre = f'enc{int(aug)}_{name}_'
        assert np.allclose(s['sample'], g[pre + 'sample'], rtol=0, atol=5e-6), name
        for k in ('sum', 'abssum', 'possum'):
            assert abs(s[k] - g[pre + k]) <= 1e-5 * max(1.0, abs(g[pre + 'abssum'])), (name, k)


def test_decoder_matches_reference_golden():
    g = np.load(os.path.join(GOLD, 'modules.npz'))
    m, (pts, img, metas) = mg.decoder_case(odec.DeepInteractionDecoder)
    with torch.no_grad():
        r = m(pts, img, metas)[0][0]
    for k, v in r.items():
        assert np.allclose(v.numpy(), g['dec_' + k], rtol=0, atol=2e-5), k
    assert np.array_equal(m.query_labels.numpy(), g['dec_query_labels'])              # INT: bit-exact
    assert np.array_equal(torch.stack(m.on_the_image_mask).numpy(), g['dec_on_the_image_mask'])


# ---------------------------------------------------------------------------------- DeepInteraction++ (row a20)
from oracle import plusplus as opp  # noqa: E402


@pytest.mark.parametrize('aug', [False, True])
def test_pp_encoder_matches_reference_golden(aug):
    g = np.load(os.path.join(GOLD, 'modules_pp.npz'))
    m, inp = mg.encoder_pp_case(opp.FusionTransformerv4, aug)
    with torch.no_grad():
        img, (p0, p1) = m(list(inp['img_feats']), list(inp['pts_feats']), inp['img_metas'], inp['pts_metas'])
    for name, t in (('img', img), ('pts_conv', p0), ('pts', p1)):
        s = mg.summarize(t)
        pre = f'enc{int(aug)}_{name}_'
        assert np.allclose(s['sample'], g[pre + 'sample'], rtol=0, atol=2e-5), name
        for k in ('sum', 'abssum', 'possum'):
            assert abs(s[k] - g[pre + k]) <= 1e-5 * max(1.0, abs(g[pre + 'abssum'])), (name, k)


def test_pp_decoder_matches_reference_golden():
    g = np.load(os.path.join(GOLD, 'modules_pp.npz'))
    m, (pts, img, metas) = mg.decoder_pp_case(opp.DeepInteractionPlusPlusDecoder)
    with torch.no_grad():
        r = m(pts, img, metas)[0][0]
    for k, v in r.items():
        assert np.allclose(v.numpy(), g['dec_' + k], rtol=0, atol=3e-5), k
    assert np.array_equal(m.query_labels.numpy(), g['dec_query_labels'])              # INT: bit-exact
    assert np.array_equal(torch.stack(m.on_the_image_mask).numpy(), g['dec_on_the_image_mask'])


def test_oracle_encoder_matches_reference_golden_at_shape_R():
    """The oracle's MMRI encoder (2 layers, eval) at the BENCHED shape (Fusion_0075_refactor shape R) against the forward
    samples the reference's own Python produced there (tests/golden/grad_shapeR.npz, oracle/refpin/make_golden_grad_shapeR.py):
    the pin of the oracle at full size, not only at the small test shape.  The golden run is single-threaded on purpose: the
    reference's depth scatter is an index_put_ with duplicate indices whose winner depends on the thread schedule (with 8
    threads ~25 000 of 134 400 depth pixels change and the maps differ by up to 8 % of their range - from the oracle AND
    between two reference runs); sequential order is the contract the oracle and the product implement."""
    from oracle import encoder as oenc
    from oracle.refpin import make_golden as mg, make_golden_grad_shapeR as gg
    gold = np.load(os.path.join(GOLD, 'grad_shapeR.npz'))
    inp = gg.case()
    O = oenc.DeepInteractionEncoder(2, gg.SHAPE['c_img'], gg.SHAPE['c_pts'], 128)
    mg.randomize(O, gg.SEED_WEIGHTS)
    O.eval()
    with torch.no_grad():
        im, (p0, p1) = O(inp['img_feats'], inp['pts_feats'], inp['img_metas'], inp['pts_metas'])
    for name, t in (('out_img', im), ('out_pts_conv', p0), ('out_pts', p1)):
        s, _, _ = gg.sample(t)
        d = np.abs(s - gold[name + '.sample'])
        assert d.max() <= 2e-6 * max(1.0, float(gold[name + '.absmax'])), (name, d.max())


def test_oracle_head_gradients_match_reference_golden_at_shape_R():
    """The oracle's MMPI decoder at the BENCHED shape (Q = 200, train mode, dropout 0, default initialisation) against what
    the reference's own Python produced there (tests/golden/grad_head_shapeR.npz, oracle/refpin/make_golden_grad_more.py head):
    proposals / labels / masks bit-exact, outputs to float32 round-off, the gradients of the three feature maps and of all
    286 parameter tensors to 1e-4 in relative L2 (measured <= 3e-6) - the pin of the oracle's BACKWARD at full size.
    (The case's head is conditioned as in a trained network, `make_golden_grad_more.head_prepare`.)"""
    from oracle import decoder as odec
    from oracle.refpin import make_golden_grad_more as gm
    gold = np.load(os.path.join(GOLD, 'grad_head_shapeR.npz'))
    feats, metas = gm.head_case()
    O, res, grads = gm._head_run(odec.DeepInteractionDecoder, feats, metas)
    assert np.array_equal(O.query_labels.numpy(), gold['query_labels'])                # INT: bit-exact
    assert np.array_equal(torch.stack(O.on_the_image_mask).numpy(), gold['on_the_image_mask'])
    for k, v in res.items():
        assert np.allclose(v.detach().numpy(), gold['out.' + k], rtol=0, atol=3e-5), k
    names = sorted(k[:-len('.sample')] for k in gold.files if k.endswith('.sample'))
    assert len(names) > 250 and set(names) <= set(grads)
    for name in names:
        ref = gold[name + '.sample']
        if float(gold[name + '.absmax']) < 1e-5:
            continue
        s = gm.sample(grads[name], n=1024 if name.startswith('p.') else 4096)['sample']
        e = np.linalg.norm(s - ref) / max(np.linalg.norm(ref), 1e-30)
        assert e <= 1e-4, (name, e)
