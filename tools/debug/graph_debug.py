"""Debug: replayed hipGraph vs eager over a pool of samples (bench.py's per-step sequence)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepinteraction_amd import harness, parallel, synth
from deepinteraction_amd.graphed import GraphedHotPath
shape = synth.SHAPE_R
dev = torch.device('cuda')
enc, dec = harness.build_models(shape, 200, torch.float16, dev)
if len(sys.argv) > 1 and sys.argv[1] == 'unfused':
    dec.fused = False
pool = [harness.to_device(synth.make_inputs(1, shape, seed=parallel.sample_seed(i)), dev, torch.float16) for i in range(3)]
npil = [int(d['pts_metas']['pillars'].shape[0]) for d in pool]
cap = max(range(3), key=lambda i: npil[i])
print('pillars', npil, 'capture on', cap)
with torch.no_grad():
    eager = []
    for d in pool:
        (img, pts), out = harness.forward(enc, dec, d)
        torch.cuda.synchronize()
        eager.append(({k: v.float().clone() for k, v in out[0][0].items()}, dec.top_proposals.clone(), img.float().clone(), pts[1].float().clone()))
    g = GraphedHotPath(enc, dec, pool[cap])
    recs = [g.prepare(d) for d in pool]
    for it in range(9):
        i = it % 3
        g.load(recs[i])
        out = g()[0][0]
        torch.cuda.synchronize()
        ref, top, img, pts = eager[i]
        dmax = max(float((out[k].float() - ref[k]).abs().max()) for k in ref)
        print(f'iter {it} sample {i}: max|graph-eager| {dmax:.3e}  proposals identical {bool(torch.equal(dec.top_proposals, top))}  '
              f'enc img {float((g.enc_out[0].float() - img).abs().max()):.2e} pts {float((g.enc_out[1][1].float() - pts).abs().max()):.2e} '
              f'dense_heatmap {float((out["dense_heatmap"].float() - ref["dense_heatmap"]).abs().max()):.2e}')
