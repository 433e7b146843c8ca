"""++ forward: graph-replay time of the neck alone and of the head alone (config-5 shape, fp16 mixed mode)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import harness, synth
shape = synth.SHAPE_PP
dtype = torch.float16
enc, dec = harness.build_models_pp(shape, 200, dtype, 'cuda')
d = harness.to_device_pp(synth.make_inputs_pp(1, shape, seed=0), 'cuda', dtype)
def t(fn, n=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n, out
with torch.no_grad():
    pm = dict(d['pts_metas'])
    ms_enc, (img, pts) = t(lambda: enc(d['img_feats'], d['pts_feats'], d['img_metas'], dict(pm)))
    ms_dec, _ = t(lambda: dec(pts, img, d['img_metas']))
    print(f'neck {ms_enc:.3f} ms   head {ms_dec:.3f} ms')
    # v1 head on the same maps for comparison (fused token path)
    e1, d1 = harness.build_models(synth.SHAPE_R, 200, dtype, 'cuda')
    ms_v1, _ = t(lambda: d1(pts, img, d['img_metas']))
    print(f'v1 head (fused token path) on the same maps {ms_v1:.3f} ms')
