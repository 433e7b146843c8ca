"""Wall time (host clock, device synchronised at every boundary) of the phases of the training step of `bench.py --mode train`."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import harness, train_step

dev = torch.device('cuda:0')
tr = train_step.Trainer(harness.SHAPES['R'], 200, dev, 1)
if os.environ.get('FUSED'):
    tr.opt = torch.optim.AdamW(tr.params, lr=1e-4, weight_decay=0.01, fused=True)
names = ['encoder fwd', 'decoder fwd', 'loss (targets + Hungarian)', 'zero_grad', 'backward', 'reduce + clip + AdamW']
acc = [0.0] * len(names)


def tick():
    torch.cuda.synchronize()
    return time.perf_counter()


def step(record):
    d, gts = tr.pool[tr.i % len(tr.pool)]
    tr.i += 1
    t = [tick()]
    img, pts = tr.enc(d['img_feats'], d['pts_feats'], d['img_metas'], dict(d['pts_metas']))
    t.append(tick())
    preds = tr.dec(pts, img, d['img_metas'])
    t.append(tick())
    losses = tr.dec.loss([g[0] for g in gts], [g[1] for g in gts], preds)
    loss = sum(v for k, v in losses.items() if k != 'matched_ious')
    t.append(tick())
    if os.environ.get('MANUAL_ZERO'):
        for p in tr.params:
            p.grad = None
    else:
        tr.opt.zero_grad(set_to_none=True)
    t.append(tick())
    loss.backward()
    t.append(tick())
    tr.reducer.finish()
    torch.nn.utils.clip_grad_norm_([p for p in tr.params if p.grad is not None], max_norm=0.1, norm_type=2)
    tr.opt.step()
    t.append(tick())
    if record:
        for i in range(len(names)):
            acc[i] += t[i + 1] - t[i]


for _ in range(3):
    step(False)
N = 5
for _ in range(N):
    step(True)
for n, a in zip(names, acc):
    print(f'{n:32s} {a / N * 1e3:8.2f} ms')
print(f'{"sum":32s} {sum(acc) / N * 1e3:8.2f} ms')
