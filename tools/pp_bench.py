"""DeepInteraction++ forward (BASELINE.json configs[4]: Fusion_0075_plusplus neck + head, 2 image levels 112x200 and
56x100, BEV 180x180, Q=200) on one GPU: samples/s and a per-kernel breakdown of the ++ operators with the
algorithmic bytes DESIGN.md states for them."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import configs, ops, synth
from deepinteraction_amd.mmdet3d_plugin import DeepInteractionPlusPlusDecoder, FusionTransformerv4

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='f16', choices=['f32', 'f16'])
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--shape', default='PP', choices=['PP', 'TINY'])
a = ap.parse_args()
shape = synth.SHAPE_PP if a.shape == 'PP' else synth.SHAPE_PP_TINY
bev = shape['bev_hw'][0]
dt = dict(f32=torch.float32, f16=torch.float16)[a.dtype]
torch.manual_seed(0)
enc = FusionTransformerv4(**configs.encoder_pp_cfg(shape['c_img'], shape['c_pts'])).cuda().to(dt).eval()
dec = DeepInteractionPlusPlusDecoder(**configs.decoder_cfg(bev=bev, num_proposals=200 if bev >= 100 else 24)).cuda().to(dt).eval()
g = torch.Generator().manual_seed(1)
with torch.no_grad():                      # off the mmcv zero-init so that the sampling offsets are spread
    for m in enc.modules():
        if hasattr(m, 'sampling_offsets'):
            m.sampling_offsets.weight.add_(torch.randn(m.sampling_offsets.weight.shape, generator=g).cuda().to(dt) * 0.05)
            m.attention_weights.weight.add_(torch.randn(m.attention_weights.weight.shape, generator=g).cuda().to(dt) * 0.05)
inp = synth.make_inputs_pp(1, shape, seed=0)
pm = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
pm['pts'] = [p.cuda() for p in inp['pts_metas']['pts']]
cl = lambda t: t.cuda().to(dt).contiguous(memory_format=torch.channels_last)
img = [cl(f) for f in inp['img_feats']]
pts = [cl(f) for f in inp['pts_feats']]


def fwd():
    with torch.no_grad():
        im, p = enc(img, pts, inp['img_metas'], dict(pm))
        return dec(p, im, inp['img_metas'])[0][0]


for _ in range(3):
    out = fwd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    out = fwd()
torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / a.steps
print(f'DeepInteraction++ forward {a.dtype} (eager): {dtm*1e3:.2f} ms  ({1/dtm:.1f} samples/s), '
      f'finite: {all(torch.isfinite(v).all().item() for v in out.values())}, '
      f'peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB, pillars {pm["pillars"].shape[0]}')
from deepinteraction_amd.graphed import GraphedHotPath
gr = GraphedHotPath(enc, dec, dict(img_feats=img, pts_feats=pts, img_metas=inp['img_metas'], pts_metas=pm))
for _ in range(3):
    gr()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    gr()
torch.cuda.synchronize(); dtg = (time.perf_counter() - t0) / a.steps
print(f'DeepInteraction++ forward {a.dtype} (hipGraph replay): {dtg*1e3:.2f} ms  ({1/dtg:.1f} samples/s)')
ops.PROFILE = []
fwd(); fwd()
torch.cuda.synchronize()
agg = {}
for name, n, s, e in ops.PROFILE:
    k = (name, n)
    agg.setdefault(k, []).append(s.elapsed_time(e) * 1e3)
ops.PROFILE = None
s_ = 2 if dt == torch.float16 else 4
Hi, Wi = shape['img_hw']
alg = {  # algorithmic bytes per launch
    ('ms_deform_attn_fwd', 6 * Hi * Wi): lambda L: (6 * (Hi * Wi + (Hi // 2) * (Wi // 2) * (L - 1)) * 128 + 6 * Hi * Wi * (8 * L * 4 * 3 + 128)) * s_,
    ('ms_deform_attn_fwd', bev * bev): lambda L: (bev * bev * L * 128 + bev * bev * (8 * L * 4 * 3 + 128)) * s_,
}
for (name, n), ts in sorted(agg.items()):
    ts = sorted(ts)
    print(f'  {name:24s} n={n:7d} launches/forward={len(ts)//2:2d}  median {ts[len(ts)//2]:8.1f} us')
