"""Dump the key row of a cell whose count glitches under [key build || conv] graph replays."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepinteraction_amd import ops, synth
from deepinteraction_amd.geometry import SampleGeometry
shape = synth.SHAPE_R
Hi, Wi = shape['img_hw']; Hb, Wb = shape['bev_hw']
inp = synth.make_inputs(1, shape, seed=0)
geom = SampleGeometry(inp['img_metas'][0], (Hi, Wi), 'cuda')
pm = inp['pts_metas']
pts_dev = pm['pts'][0].cuda()
a = (pm['pillars'].cuda(), pm['pillar_coors'].cuda(), pm['pillars_num_points'].cuda(), geom.lidar2img, geom.aug_rev, geom.ori_hw)
n = Hb * Wb
g = torch.Generator(device='cuda').manual_seed(0)
x = (torch.randn(6, 256, Hi, Wi, device='cuda', generator=g) * 0.5).clamp_(min=0).half().contiguous(memory_format=torch.channels_last)
conv = torch.nn.Conv2d(256, 128, 3, padding=1).cuda().half()
packed = ops.pack_conv3x3(conv.weight, conv.bias)
def rows(tab, cell):
    keys = tab[n * 8:].view(torch.int32).view(n, 120, 8)
    fl = tab[n * 8:].view(torch.float32).view(n, 120, 8)
    c = int(tab[:n * 4].view(torch.int32)[cell])
    return c, [(int(keys[cell, j, 0]), int(keys[cell, j, 1]) >> 8, int(keys[cell, j, 1]) & 3, [round(float(v), 4) for v in fl[cell, j, 2:6]]) for j in range(c)]
with torch.no_grad():
    ops.depth_complete(ops.depth_scatter(pts_dev, geom.lidar2img, geom.aug_rev, Hi, Wi, geom.ori_hw))
    ops.conv3x3(x, *packed)
    reft = ops.i2p_key_table(*a, (Hi, Wi), (Hb, Wb)).table.clone()
    ref = reft[:n * 4].view(torch.int32)
    torch.cuda.synchronize()
    side_stream = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        main = torch.cuda.current_stream()
        side_stream.wait_stream(main)
        with torch.cuda.stream(side_stream):
            sp = ops.depth_scatter(pts_dev, geom.lidar2img, geom.aug_rev, Hi, Wi, geom.ori_hw)
            dn = ops.depth_complete(sp)
            tab = ops.i2p_key_table(*a, (Hi, Wi), (Hb, Wb)).table
        y = ops.conv3x3(x, *packed)
        main.wait_stream(side_stream)
    shown = 0
    for it in range(400):
        gr.replay()
        torch.cuda.synchronize()
        c = tab[:n * 4].view(torch.int32)
        nbad = globals().get("nbad", 0) + int(not torch.equal(c, ref)); globals()["nbad"] = nbad
        if not torch.equal(c, ref) and shown < 1:
            shown += 1
            for cell in (c != ref).nonzero().flatten().tolist()[:2]:
                print('replay', it, 'cell', cell)
                print('   glitched:', rows(tab, cell))
                print('   reference:', rows(reft, cell))
    print('glitched replays', globals().get('nbad', 0), 'of 400')
