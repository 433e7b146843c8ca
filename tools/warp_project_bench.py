import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops, synth, harness
from deepinteraction_amd.geometry import SampleGeometry
DEV='cuda'
shape = harness.SHAPES['R']
inp = synth.make_inputs(1, shape, seed=4)
Hi, Wi = shape['img_hw']; Hb, Wb = shape['bev_hw']
geom = SampleGeometry(inp['img_metas'][0], (Hi, Wi), DEV)
g = torch.Generator().manual_seed(5)
bev = torch.randn(1, 128, Hb, Wb, generator=g).half().to(DEV).contiguous(memory_format=torch.channels_last)
depth = (torch.rand(6, Hi, Wi, generator=g) * 60 + 0.5).to(DEV)
mk = lambda: ((torch.randn(128, 128, generator=g) / math.sqrt(128)).half().to(DEV), (torch.randn(128, generator=g) * 0.1).to(DEV))
(w1, b1), (w2, b2), (wv, bv) = mk(), mk(), mk()
packed = [(ops.chain_image(w1, b1, w2, b2), True, True, True), (ops.chain_image(wv, bv), False, False, False)]
args = (depth, geom.img2lidar, geom.aug_fwd, geom.xs, geom.ys, geom.pc_range)
def timed(name, f):
    f(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for r in range(10): f()
    gr.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): gr.replay()
    e.record(); torch.cuda.synchronize()
    print(f'{name:40s} {s.elapsed_time(e)/100*1e3:7.2f} us')
timed('gather', lambda: ops.bevwarp_gather(bev, *args))
w = ops.bevwarp_gather(bev, *args)
timed('project (2 chains)', lambda: ops.pointwise_multi(w, packed))
timed('gather + project', lambda: ops.pointwise_multi(ops.bevwarp_gather(bev, *args), packed))
timed('fused warp_project', lambda: ops.warp_project(bev, *args, packed))
