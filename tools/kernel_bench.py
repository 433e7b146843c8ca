"""Per-kernel timing of the encoder hot path at a named shape (HIP events, current stream)."""
import argparse
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops, synth  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='R')
    ap.add_argument('--dtype', default='f16')
    a = ap.parse_args()
    shape = dict(R=synth.SHAPE_R, A=synth.SHAPE_A, TINY=synth.SHAPE_TINY)[a.shape]
    dt = dict(f16=torch.float16, f32=torch.float32)[a.dtype]
    s = 2 if dt == torch.float16 else 4
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    C = 128
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)

    def rnd(*sh):
        return torch.randn(*sh, device=dev, generator=g).relu().to(dt).contiguous(memory_format=torch.channels_last)
    for name, (n, H, W) in dict(image=(6, Hi, Wi), bev=(1, Hb, Wb)).items():
        q, k, v = rnd(n, C, H, W), rnd(n, C, H, W), rnd(n, C, H, W)
        byt = 4 * n * C * H * W * s
        for vname, var in (('valu', ops.LA_VALU), ('auto', ops.LA_AUTO), ('m2c0', ops.LA_MFMA), ('m2c2', ops.LA_MFMA + 2)):
            if var != ops.LA_VALU and dt != torch.float16:
                continue
            us = timeit(lambda: ops.local_attention(q, k, v, 9, 9, 1 / math.sqrt(C), variant=var))
            print(f'local_attn_fwd[{vname}] {name:5s} {a.dtype}: {us:8.1f} us  algorithmic {byt/1e6:7.1f} MB  -> {byt/us/1e6:6.3f} TB/s'
                  f'  ({byt/us/1e6/8.0*100:4.1f}% of 8 TB/s)')
        us = timeit(lambda: ops.similar_forward(q, k, 9, 9))
        print(f'  similar_fwd   {name:5s}: {us:8.1f} us')
        w = torch.softmax(ops.similar_forward(q, k, 9, 9), -1)
        us = timeit(lambda: ops.weighting_forward(v, w, 9, 9))
        print(f'  weighting_fwd {name:5s}: {us:8.1f} us')
    inp = synth.make_inputs(1, shape, seed=0)
    from deepinteraction_amd.geometry import SampleGeometry
    geom = SampleGeometry(inp['img_metas'][0], (Hi, Wi), dev)
    pts = inp['pts_metas']['pts'][0].to(dev)
    us = timeit(lambda: ops.depth_scatter(pts, geom.lidar2img, geom.aug_rev, Hi, Wi, geom.ori_hw))
    print(f'depth_scatter ({pts.shape[0]} pts): {us:8.1f} us')
    sparse = ops.depth_scatter(pts, geom.lidar2img, geom.aug_rev, Hi, Wi, geom.ori_hw)
    us = timeit(lambda: ops.depth_complete(sparse))
    print(f'depth_complete: {us:8.1f} us')
    dense = ops.depth_complete(sparse)
    bev = rnd(1, C, Hb, Wb)
    us = timeit(lambda: ops.bevwarp_gather(bev, dense, geom.img2lidar, geom.aug_fwd, geom.xs, geom.ys, geom.pc_range))
    byt = C * Hb * Wb * s + 6 * Hi * Wi * 8 + 6 * C * Hi * Wi * s
    print(f'bevwarp_gather: {us:8.1f} us  algorithmic {byt/1e6:.1f} MB -> {byt/us/1e6:.3f} TB/s')
    img = rnd(6, C, Hi, Wi)
    pm = inp['pts_metas']
    pil, coo, num = pm['pillars'].to(dev), pm['pillar_coors'].to(dev), pm['pillars_num_points'].to(dev)
    P = pil.shape[0]
    us = timeit(lambda: ops.i2p_attention(img, bev, pil, coo, num, geom.lidar2img, geom.aug_rev, geom.ori_hw))
    byt = 6 * C * Hi * Wi * s + P * 20 * 12 + P * 20 + P * C * s + C * Hb * Wb * s
    print(f'i2p_attention (P={P}): {us:8.1f} us  algorithmic {byt/1e6:.1f} MB -> {byt/us/1e6:.3f} TB/s')
    Hb_, Wb_ = shape['bev_hw']
    rng = list(synth.PC_RANGE)
    vs = [(rng[3] - rng[0]) / Wb_, (rng[4] - rng[1]) / Hb_, rng[5] - rng[2]]
    p32 = pts.float().contiguous()
    us = timeit(lambda: ops.voxelize(p32, vs, rng, 20, 60000))
    byt = pts.shape[0] * (20 + 16 + 8 + 16) + 60000 * 20 * 5 * 4 * 2
    print(f'voxelize ({pts.shape[0]} pts -> pillars, cap 60000): {us:8.1f} us  (~{byt/1e6:.1f} MB incl. zero fill)')


if __name__ == '__main__':
    main()
