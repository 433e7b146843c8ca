"""Stand-alone timing of the BEV-warp gradient scatter (`di_bevwarp_gather_bwd`) on the geometry of a synthetic shape-R sample:
the arguments of its first call inside a training forward / backward are recorded and replayed.  DI_BW_DBG=1: without the atomics,
2: without the gradient loads.  Usage: python tools/bw_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import harness, ops, train_step

rec = []
orig = ops.bevwarp_gather_bwd
def spy(*a):
    rec.append(a)
    return orig(*a)
ops.bevwarp_gather_bwd = spy
import deepinteraction_amd.autograd as ag
tr = train_step.Trainer(harness.SHAPES['R'], 200, torch.device('cuda:0'), 1, amp=True)
tr.step()
torch.cuda.synchronize()
a = rec[0]
print('calls per step', len(rec), 'grad', tuple(a[0].shape), a[0].dtype)
for _ in range(3):
    orig(*a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    orig(*a)
torch.cuda.synchronize()
print(f'bevwarp_gather_bwd {(time.perf_counter() - t0) / 20 * 1e6:.1f} us (incl. the zero-fill of the float32 map)')
d = a[1]
print('depth: min %.2f max %.2f, share > 0: %.3f' % (float(d.min()), float(d.max()), float((d > 0).float().mean())))
