"""GPU: BASELINE.json configs[0] - the single-layer BEV (180x180x256) <-> image (6x256x232x400) interaction block,
batch 1 ("shape A": 1600x928 input at stride 4) - the HIP encoder with ONE interaction layer against the CPU oracle
(shared state_dict, no depth injection), float32 and fp16.  The oracle's single-layer forward at this size is also what
`bench.py --shape A` times as its CPU baseline.  Statistics go to gpurun_out/parity_shapeA.json.

Bounds as at shape R (tests/test_shapeR_parity_gpu.py): fp32 p99.9 <= 2e-4 with a bounded fraction of texel-boundary
flips, fp16 median <= 1e-4 / p99.9 <= 5e-4 of the value scale."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from deepinteraction_amd import harness, precision, synth
from oracle import parity

DEV = 'cuda'
SHAPE = synth.SHAPE_A
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}


@pytest.fixture(scope='module')
def ctx():
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    torch.backends.cudnn.deterministic = True
    threads = torch.get_num_threads()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    inp = synth.make_inputs(1, SHAPE, seed=7)
    inp['img_feats'] = inp['img_feats'].half().float()
    inp['pts_feats'] = inp['pts_feats'].half().float()
    enc, _ = harness.build_models(SHAPE, 200, torch.float32, 'cpu', num_layers=1)
    E, _ = parity.build_oracle(SHAPE, 200, state=None, num_layers=1)
    E.load_state_dict(enc.state_dict())
    ref = parity.oracle_encoder(E.eval(), inp)
    REPORT['oracle_seconds_one_layer'] = ref['seconds']
    yield dict(inp=inp, enc=enc, ref=ref)
    torch.set_num_threads(threads)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_single_layer_block_shape_A(ctx, dtype):
    import copy
    enc = copy.deepcopy(ctx['enc']).to(DEV)
    enc = (enc.half() if dtype == torch.float16 else enc).eval()
    d = harness.to_device(ctx['inp'], DEV, dtype)
    with torch.no_grad():
        img, pts = enc(d['img_feats'], d['pts_feats'], d['img_metas'], dict(d['pts_metas']))
    torch.cuda.synchronize()
    es = parity.compare_encoder((img, pts), ctx['ref'])
    tag = 'fp32' if dtype == torch.float32 else 'fp16'
    REPORT[tag] = es
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'parity_shapeA.json'), 'w') as f:
        json.dump(REPORT, f, indent=1)
    for k, s in es.items():
        if dtype == torch.float32:
            assert s['p999'] <= 2e-4 and s['frac_gt_1e3'] <= 2e-3, (k, s)
        else:
            assert s['median'] <= 1e-4 and s['p999'] <= 5e-4 and s['frac_gt_1e3'] <= 1e-4, (k, s)
