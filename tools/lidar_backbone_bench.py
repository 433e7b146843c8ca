"""The frozen LiDAR branch alone at shape R (262 144 points, 0.075 m voxels on 1440 x 1440 x 41): wall time per stage
(synchronised) and the active-voxel counts per resolution.  python tools/lidar_backbone_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import synth
from deepinteraction_amd.mmdet3d_plugin.models.detectors import lidar_glue as lg

dev = 'cuda'
if os.environ.get('LIDAR_ONLY') != '1':          # (under rocprofv3 the solver timing's candidate kernels would fill the table)
    torch.backends.cudnn.benchmark = True        # MIOpen times its solvers per shape, as bench.py's sensor lines do
rng = list(synth.PC_RANGE)
grid = 1440
layer = dict(max_num_points=10, max_voxels=(120000, 160000), point_cloud_range=rng,
             voxel_size=[(rng[3] - rng[0]) / grid, (rng[4] - rng[1]) / grid, (rng[5] - rng[2]) / 41.0])
net = lg.FrozenLidarBackbone.synthetic(layer, (41, grid, grid), dev).eval()
pts = [synth.make_inputs(1, synth.SHAPE_R, seed=s)['pts_metas']['pts'][0].to(dev) for s in range(3)]


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


ms_all, _ = timed(lambda: net([pts[0]]), n=20)
if os.environ.get('LIDAR_ONLY') == '1':          # under rocprofv3: the branch alone, 21 forwards
    print(f'whole branch {ms_all:.2f} ms'); sys.exit(0)
ms_vox, (v, c, n) = timed(lambda: net.pts_voxel_layer(pts[0]))
feats = v[:, :, :5].sum(1) / n.clamp(min=1).to(v.dtype).unsqueeze(-1)
coors = torch.nn.functional.pad(c.long(), (1, 0))
ms_mid, x = timed(lambda: net.middle(feats, coors, 1))
ms_bb, y = timed(lambda: net.neck(net.backbone(x)))
print(f'voxels {v.shape[0]}; whole branch {ms_all:.2f} ms = voxelise {ms_vox:.2f} + sparse encoder {ms_mid:.2f} + SECOND / FPN {ms_bb:.2f}')
if net.middle._h:
    net.middle.HIP = False
    ms_t, xt = timed(lambda: net.middle(feats, coors, 1))
    net.middle.HIP = True
    print(f'sparse encoder: csrc/sparse_conv.hip {ms_mid:.2f} ms, torch formulation (gather matrix + GEMM) {ms_t:.2f} ms; '
          f'max |diff| {(x.float() - xt.float()).abs().max().item():.3e} of {xt.float().abs().max().item():.3f}')
    # the device path level by level
    from deepinteraction_amd import ops
    import torch.nn.functional as F
    shape = (41, grid, grid)
    c = coors.long()
    keys = torch.sort((((c[:, 0] * 41 + c[:, 1]) * grid + c[:, 2]) * grid + c[:, 3]).to(torch.int32)).values
    for st in net.middle.plan():
        if st[0] != 'down':
            continue
        ms_n, nbr = timed(lambda: ops.sparse_neighbours(keys, keys, 1, shape, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1)))
        fill = (nbr >= 0).float().mean().item() * 27
        cin = st[2]
        f = ops.sparse_rows(torch.randn(keys.numel(), cin, device=dev))
        h = net.middle._h[st[1].rsplit('.', 1)[0] + '.0.1'] if st[1] != 'conv_out' else net.middle._h['encoder_layers.encoder_layer4.0.1']
        ms_c, _ = timed(lambda: ops.sparse_conv(f, nbr, *h, relu=True), n=10)
        ms_o, (okeys, oshape) = timed(lambda: ops.sparse_output_keys(keys, 1, shape, st[4], st[5], st[6]))
        ms_on, onbr = timed(lambda: ops.sparse_neighbours(keys, okeys, 1, shape, oshape, st[4], st[5], st[6]))
        flop = 2.0 * (nbr >= 0).sum().item() * cin * h[3]
        print(f'level {shape}: {keys.numel()} voxels, {fill:.1f} neighbours each; submanifold table {ms_n:.3f} ms, one {cin} -> {h[3]} '
              f'product {ms_c:.3f} ms ({flop / ms_c / 1e9:.1f} TFLOP/s useful); output set {ms_o:.3f} ms -> {okeys.numel()} voxels, its table {ms_on:.3f} ms')
        keys, shape = okeys, oshape
# rulebooks vs products of the sparse encoder
xs = lg.SparseTensor(coors, feats.half(), (41, grid, grid), 1)
o = torch.argsort(xs.keys()); xs.coords, xs.feats = xs.coords[o], xs.feats[o]
ms_rb, nbr = timed(lambda: lg.subm_rulebook(xs))
w, b = net.middle._p['conv_input']
ms_cv, _ = timed(lambda: lg.sparse_conv(xs.feats, nbr, w, b, True))
ms_dn, (oc, onbr, osh) = timed(lambda: lg.strided_rulebook(xs, (3, 3, 3), (2, 2, 2), (1, 1, 1)))
print(f'finest level: submanifold rulebook {ms_rb:.2f} ms, one 5 -> 16 product {ms_cv:.2f} ms, strided rulebook {ms_dn:.2f} ms -> {oc.shape[0]} voxels')
