"""Forward + backward of DeepInteraction++ (neck + head) at the config-5 shape, train() mode, random linear
functional of the head outputs as loss, no optimizer.  Reports ms per step."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import configs, synth
from deepinteraction_amd.mmdet3d_plugin import DeepInteractionPlusPlusDecoder, FusionTransformerv4

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='f32', choices=['f32', 'f16'])
ap.add_argument('--steps', type=int, default=5)
a = ap.parse_args()
shape = synth.SHAPE_PP
dt = dict(f32=torch.float32, f16=torch.float16)[a.dtype]
torch.manual_seed(0)
enc = FusionTransformerv4(**configs.encoder_pp_cfg(shape['c_img'], shape['c_pts'])).cuda().to(dt).train()
dec = DeepInteractionPlusPlusDecoder(**configs.decoder_cfg(bev=180, num_proposals=200)).cuda().to(dt).train()
inp = synth.make_inputs_pp(1, shape, seed=0)
pm = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
pm['pts'] = [p.cuda() for p in inp['pts_metas']['pts']]
cl = lambda t: t.cuda().to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
img = [cl(f) for f in inp['img_feats']]
pts = [cl(f) for f in inp['pts_feats']]
params = [p for m in (enc, dec) for p in m.parameters()]


def step():
    im, p = enc(img, pts, inp['img_metas'], dict(pm))
    out = dec(p, im, inp['img_metas'])[0][0]
    loss = sum(v.float().mean() for v in out.values())
    for q in params:
        q.grad = None
    loss.backward()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    l = step()
torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / a.steps
print(f'DeepInteraction++ forward+backward {a.dtype}: {dtm*1e3:.1f} ms/step  ({1/dtm:.1f} samples/s), loss {float(l):.4f}, '
      f'grad finite: {all(torch.isfinite(q.grad).all().item() for q in params if q.grad is not None)}, '
      f'peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB')
