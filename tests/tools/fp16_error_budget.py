"""TEST TOOL (CPU only, uses the oracle): where does the fp16 decoder error come from?

Runs the ORACLE decoder at shape R with selected parts rounded through fp16 and compares with the unrounded oracle,
proposals forced equal.  Usage: python tests/tools/fp16_error_budget.py [cache.pt]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepinteraction_amd import harness, synth  # noqa: E402
from oracle import parity  # noqa: E402

SHAPE = synth.SHAPE_R
cache = sys.argv[1] if len(sys.argv) > 1 else '/tmp/fp16_budget_cache.pt'
torch.set_num_threads(32)
inp = synth.make_inputs(1, SHAPE, seed=100)
inp['img_feats'] = inp['img_feats'].half().float()
inp['pts_feats'] = inp['pts_feats'].half().float()
enc, dec = harness.build_models(SHAPE, 200, torch.float32, 'cpu')
state = (enc.state_dict(), dec.state_dict())
E, D = parity.build_oracle(SHAPE, 200, state=state)
if os.path.exists(cache):
    ref_enc = torch.load(cache)
else:
    ref_enc = parity.oracle_encoder(E, inp)
    torch.save(ref_enc, cache)
free = parity.oracle_decoder(D, ref_enc, inp['img_metas'])
top = free['top']

h = lambda t: t.half().float()


def run(name, round_maps=False, round_weights=None):
    """round_weights: predicate on parameter name -> bool."""
    _, D2 = parity.build_oracle(SHAPE, 200, state=state)
    if round_weights is not None:
        for n, t in list(D2.named_parameters()) + list(D2.named_buffers()):
            if t.is_floating_point() and round_weights(n):
                t.data = h(t.data)
    e = dict(ref_enc)
    if round_maps:
        e = {k: (h(v) if torch.is_tensor(v) else v) for k, v in ref_enc.items()}
    got = parity.oracle_decoder(D2, e, inp['img_metas'], top_override=top)
    print(f'== {name}')
    for k in ('center', 'height', 'dim', 'rot', 'vel', 'heatmap'):
        s = parity.rel_stats(got['out'][k], free['out'][k])
        print(f"  {k:8s} max {s['max']:.2e} med {s['median']:.2e} p999 {s['p999']:.2e} >1e-3 {s['frac_gt_1e3']:.3f} scale {s['scale']:.1f}")
    print('  masks equal', [bool(torch.equal(a, b)) for a, b in zip(got['masks'], free['masks'])])


big = lambda n: n.startswith(('heatmap_head', 'shared_conv', 'class_encoding')) or 'multihead_attn.in_proj' in n
run('maps rounded to fp16 only', round_maps=True)
run('all decoder weights fp16', round_weights=lambda n: True)
run('big-map weights fp16 (heat map, cross-attn in_proj), token weights fp32', round_weights=big)
run('maps + all weights fp16', round_maps=True, round_weights=lambda n: True)
run('maps + big-map weights fp16', round_maps=True, round_weights=big)
print([n for n, _ in D.named_parameters()][:400:7])

# ---- the encoder's share: oracle encoder with fp16-rounded weights (exact arithmetic), outputs rounded to fp16
print('==== encoder with fp16 weights (exact arithmetic) -> fp32 decoder')
E2, _ = parity.build_oracle(SHAPE, 200, state=state)
for t in list(E2.parameters()) + list(E2.buffers()):
    if t.is_floating_point():
        t.data = h(t.data)
enc_w16 = parity.oracle_encoder(E2, inp)
for k in ('img', 'pts_conv', 'pts'):
    s = parity.rel_stats(enc_w16[k], ref_enc[k])
    print(f"  enc {k:8s} max {s['max']:.2e} med {s['median']:.2e} p999 {s['p999']:.2e}")
save = ref_enc
ref_enc = enc_w16
run('encoder weights fp16, fp32 maps, fp32 decoder')
run('encoder weights fp16, maps rounded, fp32 decoder', round_maps=True)
ref_enc = save
