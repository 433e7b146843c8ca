"""Wall time (host clock, device synchronised at every boundary) of the phases of the training step of `bench.py --mode train`:
median / mean / max over STEPS (default 20) steps - single steps on the GPU boxes show 2x outliers."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import harness, train_step

dev = torch.device('cuda:0')
tr = train_step.Trainer(harness.SHAPES['R'], 200, dev, 1)
if os.environ.get('FUSED'):
    tr.opt = torch.optim.AdamW(tr.params, lr=1e-4, weight_decay=0.01, fused=True)
names = ['encoder fwd', 'decoder fwd', 'loss (targets + Hungarian)', 'zero_grad', 'backward', 'reduce + clip + AdamW']
acc = [[] for _ in names]


def tick():
    if not os.environ.get('NOSYNC'):          # NOSYNC=1: the HOST's time per phase (where it blocks, how far it runs ahead)
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
            acc[i].append(t[i + 1] - t[i])


def step_graphed(record):
    i = tr.i % len(tr.pool)
    tr.i += 1
    _, gts = tr.pool[i]
    t = [tick()]
    tr.h.load(tr.records[i])
    tr.seed_word.random_(0, 2 ** 62)
    outs = tr.graphed(tr.h.img_feats, tr.h.pts_feats)
    tr.dec.prepare_targets([g[0] for g in gts], [g[1] for g in gts], outs[0].device)      # host work under the replay
    t.append(tick())
    t.append(t[-1])
    preds = [[dict(zip(tr.module.keys, outs))]]
    losses = tr.dec.loss([g[0] for g in gts], [g[1] for g in gts], preds)
    loss = sum(v for k, v in losses.items() if k != 'matched_ious')
    t.append(tick())
    tr._zero_grad()
    t.append(tick())
    tr._backward(loss)
    t.append(tick())
    tr.reducer.finish()
    tr._update()
    t.append(tick())
    if record:
        for i in range(len(names)):
            acc[i].append(t[i + 1] - t[i])


if os.environ.get('DI_TRAIN_GRAPH', '0') == '1':        # forward graph (load + replay) | - | loss | zero_grad | backward graph | ...
    tr = train_step.GraphedTrainer(harness.SHAPES['R'], 200, dev, 1, amp=bool(os.environ.get('AMP')) or None)
    names[0], names[1] = 'load + forward graph', '-'
    step = step_graphed
for _ in range(3):
    step(False)
N = int(os.environ.get('STEPS', '20'))
for _ in range(N):
    step(True)
med = lambda v: sorted(v)[len(v) // 2]
tot = [sum(a[k] for a in acc) for k in range(N)]
for n, a in zip(names, acc):
    print(f'{n:32s} median {med(a) * 1e3:8.2f} ms   mean {sum(a) / N * 1e3:8.2f} ms   max {max(a) * 1e3:8.2f} ms')
print(f'{"step":32s} median {med(tot) * 1e3:8.2f} ms   mean {sum(tot) / N * 1e3:8.2f} ms   max {max(tot) * 1e3:8.2f} ms   ({N} steps)')
