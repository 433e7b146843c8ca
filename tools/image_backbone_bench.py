"""Frozen image feature extractor (`FrozenResNetFPN`, torch / MIOpen) in front of the hot path: device time of one sample's
six 448x800 camera images (`Fusion_0075_refactor.py` `img_scale`), captured into a hipGraph and replayed, for the levels
the neck reads.  Prints one JSON line.  Random-init weights in the checkpoint layout (`synthetic_state`).

    python tools/image_backbone_bench.py [--levels 0] [--cams 6] [--hw 448 800] [--steps 20] [--eager]
"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd.mmdet3d_plugin import FrozenResNetFPN


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--levels', type=int, nargs='+', default=[0])
    ap.add_argument('--all-levels', action='store_true', help='compute all five FPN levels as the reference does')
    ap.add_argument('--cams', type=int, default=6)
    ap.add_argument('--hw', type=int, nargs=2, default=[448, 800])
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--eager', action='store_true')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    net = FrozenResNetFPN(levels=None if args.all_levels else tuple(args.levels))
    net.load_mmdet_state(*net.synthetic_state(0)).to(dev)
    img = torch.randn(args.cams, 3, *args.hw, device=dev).to(torch.float16).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        out = net(img)
    torch.cuda.synchronize()
    if args.eager:
        run = lambda: net(img)
    else:
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net(img)
            with torch.cuda.graph(graph, stream=side):
                out = net(img)
        torch.cuda.current_stream().wait_stream(side)
        run = graph.replay
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    # flops of the convolutions actually run (2 x multiply-adds, stride-aware output sizes)
    half = lambda hw: (math.ceil(hw[0] / 2), math.ceil(hw[1] / 2))
    cur = half(tuple(args.hw))
    flops = 2 * 64 * 3 * 49 * cur[0] * cur[1]
    cur = half(cur)                                 # max-pool
    size = {}
    for name, cin, cout, k, st in net._plan[1:]:
        if name.endswith('.conv2') and st > 1:
            cur = half(cur)                         # the stride sits on the 3x3; the downsample conv follows it in the plan
        flops += 2 * cout * cin * k * k * cur[0] * cur[1]
        size[int(name[5]) - 1] = cur
    low, outs = net._needed()
    for i in range(low, 4):
        flops += 2 * 256 * net.stage_channels[i] * size[i][0] * size[i][1]
    for i in outs:
        flops += 2 * 256 * 256 * 9 * size[i][0] * size[i][1]
    flops *= args.cams
    print(json.dumps(dict(tool='image_backbone_bench', cams=args.cams, hw=args.hw, levels=list(net.levels),
                          launch='eager' if args.eager else 'graph', ms_per_sample=round(ms, 3),
                          tflops=round(flops / ms / 1e9, 1), gflop_per_sample=round(flops / 1e9, 1),
                          out_shapes=[list(o.shape) for o in out], dtype='f16', layout='channels_last')))


if __name__ == '__main__':
    main()
