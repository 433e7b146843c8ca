import math, os, sys, torch
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/deepinteraction_amd') else os.getcwd())
from deepinteraction_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
xs = [torch.randn(6, 128, 112, 200, device='cuda', generator=g).relu().half().contiguous(memory_format=torch.channels_last) for _ in range(3)]
mk = lambda: ((torch.randn(128, 128, device='cuda', generator=g) / math.sqrt(128)).half(), torch.randn(128, device='cuda', generator=g) * 0.1)
chains = []
for two in (True, True, False, True):
    w1, b1 = mk(); w2, b2 = mk() if two else (None, None)
    chains.append((ops.chain_image(w1, b1, w2, b2), True, True, two))
def timed(name, f):
    f(0); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for r in range(9): f(r % 3)
    gr.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): gr.replay()
    e.record(); torch.cuda.synchronize()
    print(f'{name:40s} {s.elapsed_time(e)/90*1e3:7.2f} us', flush=True)
timed('multi, 4 chains', lambda i: ops.pointwise_multi(xs[i], chains))
timed('multi, 2 chains', lambda i: ops.pointwise_multi(xs[i], chains[1:3]))
