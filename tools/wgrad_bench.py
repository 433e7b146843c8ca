"""Graph-replay timing of csrc/wgrad.hip against the slab-batched hipBLASLt GEMM it replaces (autograd.PixelLinear)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops
from deepinteraction_amd.autograd import PixelLinear


def timed(f, n=20):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            f()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for P, Cin, Cout in [(134400, 128, 128), (134400, 256, 128), (32400, 128, 128), (32400, 256, 128), (32400, 512, 128)]:
    x = torch.randn(P, Cin, device='cuda')
    gy = torch.randn(P, Cout, device='cuda')
    S = PixelLinear._slabs(P)
    lib = lambda: (torch.bmm(gy.view(S, P // S, -1).transpose(1, 2), x.view(S, P // S, -1)).sum(0), gy.sum(0))
    own = lambda: ops.wgrad(x, gy, bias=True)
    byt = (P * (Cin + Cout)) * 4
    t_own, t_lib = timed(own), timed(lib)
    print(f'P {P:6d} {Cin:3d} -> {Cout:3d}: own {t_own:7.1f} us ({byt / t_own / 1e6:5.2f} TB/s, {2 * P * Cin * Cout / t_own / 1e6:6.1f} TFLOP/s)   '
          f'slab-batched library GEMM + sums {t_lib:7.1f} us', flush=True)
