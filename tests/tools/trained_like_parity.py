"""fp16 parity on TRAINED-LIKE weights.

The fp16 tail of the decoder outputs at the benched random-init weights (max 3.4e-3) was explained by the random-init head's
gain with a synthetic conditioning (`harness.condition_head`: residual branches x 0.5).  This tool replaces the synthetic
conditioning by what training does: the reference's own initialisation (the modules' `init_weights`, which the product
mirrors) + N steps of the product's own trainer (`train_step.Trainer`: forward + head loss with Hungarian targets +
backward + AdamW, lr 1e-4 - the reference's optimiser settings) on synthetic scenes with synthetic ground truth; then the
fp16 (mixed-mode) inference forward of those weights on a fresh sample against the CPU oracle holding the same float32
state_dict.  No `condition_head`.  Prints one JSON line; `tests/test_shapeR_parity_gpu.py::test_fp16_trained_like_weights`
runs it with a small step count.

    python tests/tools/trained_like_parity.py [--steps 300] [--shape R]
"""
import argparse
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(steps=300, shape_name='R', proposals=200, lr=None, seed=0, device='cuda'):
    from deepinteraction_amd import harness, precision, synth, train_step
    from oracle import parity
    shape = harness.SHAPES[shape_name]
    nprop = proposals if shape['bev_hw'][0] >= 100 else 24
    tr = train_step.Trainer(shape, nprop, torch.device(device), 1, batch=1, pool=4, rank=0, seed=seed, amp=False)
    if lr is not None:
        for g in tr.opt.param_groups:
            g['lr'] = lr
    t0 = time.time()
    losses = []
    for i in range(steps):
        losses.append(float(tr.step()))
    torch.cuda.synchronize()
    train_s = time.time() - t0
    enc, dec = tr.enc.eval(), tr.dec.eval()
    state = ({k: v.detach().float().cpu() for k, v in enc.state_dict().items()},
             {k: v.detach().float().cpu() for k, v in dec.state_dict().items()})
    # a sample the trainer has not seen
    inp = synth.make_inputs(1, shape, seed=4242)
    inp['img_feats'], inp['pts_feats'] = inp['img_feats'].half().float(), inp['pts_feats'].half().float()
    pe, pd = precision.to_inference(copy.deepcopy(enc), copy.deepcopy(dec), torch.float16)
    with torch.no_grad():
        (img, pts), out = harness.forward(pe.eval(), pd.eval(), harness.to_device(inp, device, torch.float16))
    torch.cuda.synchronize()
    res = out[0][0]
    top = pd.top_proposals.cpu()
    threads = torch.get_num_threads()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    E, D = parity.build_oracle(shape, nprop, state=state)
    ref = parity.oracle_encoder(E, inp)
    free = parity.oracle_decoder(D, ref, inp['img_metas'])
    forced = parity.oracle_decoder(D, ref, inp['img_metas'], top_override=top)
    torch.set_num_threads(threads)
    es = parity.compare_encoder((img.float().cpu(), [t.float().cpu() for t in pts]), ref)
    ds = parity.compare_decoder({k: v.float().cpu() for k, v in res.items()}, pd.query_labels.cpu(),
                                [m.cpu() for m in pd.on_the_image_mask], top, free, forced)
    ties = parity.heatmap_score_ties(res['query_heatmap_score'], forced, tol=2e-3)
    r3 = lambda x: float(f'{x:.3g}')
    keys = {k: dict(max=r3(v['max']), p999=r3(v['p999']), median=r3(v['median']), frac_gt_1e3=r3(v['frac_gt_1e3']))
            for k, v in ds['keys'].items() if k != 'query_heatmap_score'}
    box = [k for k in keys if k not in ('dense_heatmap', 'dense_heatmap_old')]
    return dict(steps=steps, shape=shape_name, train_seconds=round(train_s, 1), first_loss=r3(losses[0]) if losses else None,
                last_loss=r3(sum(losses[-10:]) / max(len(losses[-10:]), 1)) if losses else None,
                encoder={k: dict(max=r3(v['max']), p999=r3(v['p999'])) for k, v in es.items()},
                decoder=keys, worst_box_output_max=max(keys[k]['max'] for k in box),
                worst_box_output_p999=max(keys[k]['p999'] for k in box),
                proposal_set_overlap=ds['proposal_set_overlap'], labels_equal_on_same_proposals=ds['labels_equal_on_same_proposals'],
                mask_agreement=[r3(m) for m in ds['mask_agreement']], heatmap_score_ties=ties,
                note='fp16 mixed-mode product against the float32 oracle with the SAME trained-like state_dict; errors as '
                     '|got - ref| / max(1, max|ref|); no condition_head')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--shape', default='R')
    ap.add_argument('--lr', type=float, default=None)
    a = ap.parse_args()
    print(json.dumps(run(a.steps, a.shape, lr=a.lr)))
