"""GPU: the v1 product modules (HIP path) against the golden vectors produced by the REFERENCE'S OWN code
(tests/golden/modules.npz, generator oracle/refpin/make_golden.py) - float32, same seeded weights and inputs.
(DeepInteraction++ has the same test in tests/test_plusplus_gpu.py.)  The depth chain runs on the device, un-injected."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.refpin import make_golden as mg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DEV = 'cuda'


def _dev_inputs(inp):
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
    pm = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
    pm['pts'] = [p.to(DEV) for p in inp['pts_metas']['pts']]
    return cl(inp['img_feats']), cl(inp['pts_feats']), inp['img_metas'], pm


@pytest.mark.parametrize('aug', [False, True])
def test_encoder_matches_reference_golden(aug):
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionEncoder
    torch.backends.cudnn.deterministic = True
    g = np.load(os.path.join(GOLD, 'modules.npz'))
    m, inp = mg.encoder_case(DeepInteractionEncoder, aug)
    m = m.to(DEV)
    with torch.no_grad():
        img, (p0, p1) = m(*_dev_inputs(inp))
    for name, t in (('img', img), ('pts_conv', p0), ('pts', p1)):
        s = mg.summarize(t.float().cpu().contiguous())
        pre = f'enc{int(aug)}_{name}_'
        scale = max(1.0, float(np.abs(g[pre + 'sample']).max()))
        d = np.abs(s['sample'] - g[pre + 'sample'])
        # a projected point within float round-off of a texel boundary may flip a bilinear corner: bounded fraction
        assert (d > 2e-4 * scale).mean() <= 2e-3, (name, float(d.max()), float((d > 2e-4 * scale).mean()))
        assert abs(s['abssum'] - g[pre + 'abssum']) <= 1e-4 * abs(g[pre + 'abssum']), name


def test_decoder_matches_reference_golden():
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionDecoder
    torch.backends.cudnn.deterministic = True
    g = np.load(os.path.join(GOLD, 'modules.npz'))
    m, (pts, img, metas) = mg.decoder_case(DeepInteractionDecoder)
    m = m.to(DEV)
    with torch.no_grad():
        r = m([p.to(DEV) for p in pts], img.to(DEV), metas)[0][0]
    assert np.array_equal(m.query_labels.cpu().numpy(), g['dec_query_labels'])              # INT: bit-exact
    assert np.array_equal(torch.stack(m.on_the_image_mask).cpu().numpy(), g['dec_on_the_image_mask'])
    for k, v in r.items():
        ref = g['dec_' + k]
        assert np.allclose(v.float().cpu().numpy(), ref, rtol=0, atol=2e-4 * max(1.0, float(np.abs(ref).max()))), k
