"""Timing of the fused training window attention (forward + lse, backward = row dot + three passes) against the unfused float32
operators, image-side and BEV-side shapes of shape R.  Usage: python tools/la_train_bench.py"""
import math, os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops
from deepinteraction_amd.mmdet3d_plugin.models.utils.encoder_utils import similarFunction, weightingFunction

def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6

for shape in [(6, 112, 200), (1, 180, 180)]:
    n, H, W = shape
    mk = lambda: torch.randn(n, 128, H, W, device='cuda').half().contiguous(memory_format=torch.channels_last)
    q, k, v, go = mk(), mk(), mk(), mk()
    scale = 1 / math.sqrt(128)
    out, lse = ops.local_attention_train_fwd(q, k, v, scale)
    ops.PROFILE = None
    f = t(lambda: ops.local_attention_train_fwd(q, k, v, scale))
    b = t(lambda: ops.local_attention_train_bwd(q, k, v, out, go, lse, scale))
    def unfused():
        qq, kk, vv = (x.detach().requires_grad_(True) for x in (q, k, v))
        w = similarFunction.apply(qq, kk, 9, 9)
        o = weightingFunction.apply(vv, F.softmax(w * scale, -1), 9, 9)
        o.backward(go)
    u = t(unfused, 5)
    rel = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max())
    print(f'{shape}: fused fwd {f:.1f} us, bwd {b:.1f} us; unfused float32-window fwd+bwd {u:.1f} us')
