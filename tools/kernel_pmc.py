"""A few launches of the hot kernels at the benched shapes (for rocprofv3 --pmc passes; see tools/pmc_session.sh)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops
which = sys.argv[1].split(',') if len(sys.argv) > 1 else ['conv', 'la', 'pw']
g = torch.Generator(device='cuda').manual_seed(0)
def cl(*s): return (torch.randn(*s, device='cuda', generator=g) * 0.5).clamp_(min=0).half().contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    if 'conv' in which:
        for (n, Cin, H, W) in [(6, 256, 112, 200), (1, 512, 180, 180)]:
            x = cl(n, Cin, H, W)
            conv = torch.nn.Conv2d(Cin, 128, 3, padding=1).cuda().half()
            packed = ops.pack_conv3x3(conv.weight, conv.bias)
            for _ in range(3):
                ops.conv3x3(x, *packed)
                ops.conv3x3(x, *packed, use_staged=False)
    if 'la' in which:
        q, k, v = cl(6, 128, 112, 200), cl(6, 128, 112, 200), cl(6, 128, 112, 200)
        for _ in range(3):
            ops.local_attention(q, k, v, 9, 9, 1 / math.sqrt(128))
        q, k, v = cl(1, 128, 180, 180), cl(1, 128, 180, 180), cl(1, 128, 180, 180)
        for _ in range(3):
            ops.local_attention(q, k, v, 9, 9, 1 / math.sqrt(128))
    if 'pw' in which:
        x = cl(6, 128, 112, 200)
        w = (torch.randn(128, 128, device='cuda', generator=g) / 11).half()
        b = torch.randn(128, device='cuda', generator=g) * 0.1
        for _ in range(3):
            ops.pointwise_chain(x, w, b, True, w2=w, b2=b, relu2=True)
    if 'mix' in which:                       # out_proj + integration pair of an encoder layer, image side (chain<256,256>)
        x1, x2, x3 = cl(6, 128, 112, 200), cl(6, 128, 112, 200), cl(6, 128, 112, 200)
        w1 = (torch.randn(128, 256, device='cuda', generator=g) / 16).half()
        w2 = (torch.randn(128, 256, device='cuda', generator=g) / 16).half()
        b = torch.randn(128, device='cuda', generator=g) * 0.1
        for _ in range(3):
            ops.pointwise_chain(x1, w1, b, False, x2=x2, w2=w2, b2=b, relu2=False, x3=x3)
    if 'multi' in which:                     # the four projections of the image map (q, k two links; v one; q of P2I two)
        x = cl(6, 128, 112, 200)
        mk = lambda: ((torch.randn(128, 128, device='cuda', generator=g) / 11).half(), torch.randn(128, device='cuda', generator=g) * 0.1)
        chains = []
        for two in (True, True, False, True):
            w1, b1 = mk()
            w2, b2 = mk() if two else (None, None)
            chains.append((ops.chain_image(w1, b1, w2, b2), True, True, two))
        for _ in range(3):
            ops.pointwise_multi(x, chains)
    if 'i2p' in which or 'bw' in which or 'xa' in which:
        from deepinteraction_amd import synth
        from deepinteraction_amd.geometry import SampleGeometry
        shape = synth.SHAPE_R
        Hi, Wi = shape['img_hw']; Hb, Wb = shape['bev_hw']
        inp = synth.make_inputs(1, shape, seed=0)
        geom = SampleGeometry(inp['img_metas'][0], (Hi, Wi), 'cuda')
        pm = inp['pts_metas']
        img, bev = cl(6, 128, Hi, Wi), cl(1, 128, Hb, Wb) * 0.2
    if 'i2p' in which:
        a = (pm['pillars'].cuda(), pm['pillar_coors'].cuda(), pm['pillars_num_points'].cuda(), geom.lidar2img, geom.aug_rev, geom.ori_hw)
        for _ in range(3):
            keys = ops.i2p_key_table(*a, (Hi, Wi), (Hb, Wb))
            ops.i2p_attention(img, bev, *a, keys=keys)
            ops.i2p_attention(img, bev, *a, keys=keys._replace(dense=None))       # the wave-per-cell kernel beside it
    if 'bw' in which:
        pts = pm['pts'][0].cuda()
        dense = ops.depth_complete(ops.depth_scatter(pts, geom.lidar2img, geom.aug_rev, Hi, Wi, geom.ori_hw))
        for _ in range(3):
            ops.bevwarp_gather(bev, dense, geom.img2lidar, geom.aug_fwd, geom.xs, geom.ys, geom.pc_range)
    if 'xa' in which:
        q = (torch.randn(1, 200, 128, device='cuda', generator=g) * 0.5).half()
        kv = (torch.randn(1, 180 * 180, 256, device='cuda', generator=g) * 0.5).half()
        for _ in range(3):
            ops.mha_decode(q, kv, 8, 0.25)
torch.cuda.synchronize()
