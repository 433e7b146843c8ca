"""The frozen LiDAR branch in front of the hot path (`lidar_glue.py`: mmdet3d SparseEncoder without spconv - rulebooks by sort +
searchsorted, one gather + one GEMM per layer - SECOND, SECONDFPN) against the oracle's DENSE restatement of the published
spconv / mmdet3d semantics (`oracle/sparse_encoder.py`; parity unpinned: neither package is in the image - the anchors are the
module / state-dict layout of reference config `Fusion_0075_refactor.py:160-184` and the sparse-convolution definitions)."""
import pytest
import torch

from deepinteraction_amd.mmdet3d_plugin.models.detectors import lidar_glue as lg
from oracle import sparse_encoder as osp


def _voxels(shape, batch, n, seed):
    g = torch.Generator().manual_seed(seed)
    D, H, W = shape
    coors = []
    for b in range(batch):
        # clustered like a scan: a few dense blobs + scattered singles (isolated voxels, voxels at the borders)
        c = torch.stack([torch.randint(0, D, (n,), generator=g), torch.randint(0, H, (n,), generator=g),
                         torch.randint(0, W, (n,), generator=g)], 1)
        blob = torch.stack([torch.randint(0, 6, (n,), generator=g), torch.randint(H // 3, H // 3 + 6, (n,), generator=g),
                            torch.randint(W // 2, W // 2 + 7, (n,), generator=g)], 1)
        blob = torch.minimum(blob, torch.tensor([D - 1, H - 1, W - 1]))
        c = torch.cat([c, blob, torch.tensor([[0, 0, 0], [D - 1, H - 1, W - 1], [D - 1, 0, W - 1]])], 0).unique(dim=0)
        coors.append(torch.nn.functional.pad(c, (1, 0), value=b))
    coors = torch.cat(coors, 0)
    coors = coors[torch.randperm(coors.shape[0], generator=g)]         # voxel order is arbitrary (spconv's hash order)
    feats = torch.randn(coors.shape[0], 5, generator=g)
    return feats, coors.int()


def test_state_dict_layout_is_mmdet3ds():
    m = lg.FrozenSparseEncoder()
    keys = m.state_keys()
    assert len(keys) == (1 + 3 + 1 + 8 * 2) * 5          # conv_input, 3 strided, conv_out: 1 conv each; 8 basic blocks: 2 each
    for k in ('conv_input.0.weight', 'conv_input.1.running_var', 'encoder_layers.encoder_layer1.0.conv1.weight',
              'encoder_layers.encoder_layer1.1.bn2.bias', 'encoder_layers.encoder_layer1.2.0.weight',
              'encoder_layers.encoder_layer3.2.1.running_mean', 'encoder_layers.encoder_layer4.1.conv2.weight', 'conv_out.0.weight'):
        assert k in keys, k
    assert 'encoder_layers.encoder_layer4.2.0.weight' not in keys      # the last stage has no strided layer
    sd = m.synthetic_state(0)
    assert sd['conv_input.0.weight'].shape == (16, 3, 3, 3, 5)          # spconv 2.x: (C_out, kD, kH, kW, C_in)
    assert sd['encoder_layers.encoder_layer3.2.0.weight'].shape == (128, 3, 3, 3, 64)
    assert sd['conv_out.0.weight'].shape == (128, 3, 1, 1, 128)
    b = lg.FrozenSECOND().synthetic_state(0)
    assert b['blocks.0.0.weight'].shape == (128, 256, 3, 3) and b['blocks.1.0.weight'].shape == (256, 128, 3, 3)
    assert 'blocks.1.15.weight' in b and 'blocks.1.16.running_var' in b and 'blocks.1.18.weight' not in b
    n = lg.FrozenSECONDFPN().synthetic_state(0)
    assert n['deblocks.0.0.weight'].shape == (256, 128, 1, 1)           # Conv2d (use_conv_for_no_stride)
    assert n['deblocks.1.0.weight'].shape == (256, 256, 2, 2)           # ConvTranspose2d: (C_in, C_out, k, k)


@pytest.mark.parametrize('shape,batch,n', [((41, 24, 32), 2, 500), ((41, 16, 16), 1, 40), ((41, 8, 8), 1, 0)])
def test_sparse_encoder_equals_dense_restatement(shape, batch, n):
    """41 -> 21 -> 11 -> 5 -> 2 levels (padding 1, 1, [0, 1, 1], then the (3, 1, 1) / (2, 1, 1) output layer) as in the reference
    configuration; submanifold layers never grow the active set, strided layers activate every output whose window holds a voxel."""
    feats, coors = _voxels(shape, batch, n, seed=3)
    m = lg.FrozenSparseEncoder(sparse_shape=shape, dtype=torch.float32)
    sd = m.synthetic_state(1)
    m.load_mmdet_state(sd)
    got = m(feats, coors, batch)
    want = osp.sparse_encoder(feats, coors, batch, sd, shape)
    assert got.shape == want.shape == (batch, 256, shape[1] // 8, shape[2] // 8)
    assert (got - want).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item())
    assert ((got != 0) == (want != 0)).float().mean().item() > 0.999    # the same active output sites


def test_rulebooks_against_brute_force():
    shape = (9, 7, 8)
    feats, coors = _voxels(shape, 1, 30, seed=5)
    x = lg.SparseTensor(coors.long(), feats, shape, 1)
    order = torch.argsort(x.keys())
    x.coords, x.feats = x.coords[order], x.feats[order]
    nbr = lg.subm_rulebook(x)
    where = {tuple(c.tolist()): i for i, c in enumerate(x.coords)}
    for m_, c in enumerate(x.coords.tolist()):
        o = 0
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    assert int(nbr[m_, o]) == where.get((c[0], c[1] + dz, c[2] + dy, c[3] + dx), -1)
                    o += 1
    oc, onbr, oshape = lg.strided_rulebook(x, (3, 3, 3), (2, 2, 2), (0, 1, 1))
    assert oshape == (4, 4, 4)
    want = set()
    for c in x.coords.tolist():
        for kz in range(3):
            for ky in range(3):
                for kx in range(3):
                    z, y, xx = c[1] + 0 - kz, c[2] + 1 - ky, c[3] + 1 - kx
                    if z % 2 == 0 and y % 2 == 0 and xx % 2 == 0 and 0 <= z // 2 < 4 and 0 <= y // 2 < 4 and 0 <= xx // 2 < 4:
                        want.add((c[0], z // 2, y // 2, xx // 2))
    assert set(map(tuple, oc.tolist())) == want


def test_second_and_fpn_equal_unfused_oracle():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 256, 24, 16, generator=g)
    bb, nk = lg.FrozenSECOND(dtype=torch.float32), lg.FrozenSECONDFPN(dtype=torch.float32)
    sb, sn = bb.synthetic_state(3), nk.synthetic_state(4)
    bb.load_mmdet_state(sb), nk.load_mmdet_state(sn)
    got = nk(bb(x))[0]
    want = osp.second_fpn(osp.second(x, sb), sn)[0]
    assert got.shape == want.shape == (2, 512, 24, 16)
    assert (got - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.gpu
def test_lidar_backbone_from_raw_points_on_the_device():
    """points -> HIP voxeliser (0.075 m voxels at the reference range scaled to a 352^2 x 41 grid) -> HardSimpleVFE -> sparse
    encoder -> SECOND -> SECONDFPN, fp16 on the device, against the float32 oracle fed with the SAME voxels."""
    from deepinteraction_amd import synth
    dev = 'cuda'
    rng = list(synth.PC_RANGE)
    grid = 352                                   # 44 BEV cells per side: even, as SECOND (stride 2) + SECONDFPN (deconv x 2) need
    vs = [(rng[3] - rng[0]) / grid, (rng[4] - rng[1]) / grid, (rng[5] - rng[2]) / 41.0]
    layer = dict(max_num_points=10, voxel_size=vs, max_voxels=(120000, 160000), point_cloud_range=rng)
    net = lg.FrozenLidarBackbone.synthetic(layer, (41, grid, grid), dev, dtype=torch.float16, seed=0).eval()
    inp = synth.make_inputs(1, synth.SHAPE_TINY, seed=2)
    pts = [p.to(dev) for p in inp['pts_metas']['pts']]
    out = net(pts)[0]
    assert out.shape == (1, 512, grid // 8, grid // 8) and out.dtype == torch.float16 and bool(torch.isfinite(out).all())
    v, c, n = net.pts_voxel_layer(pts[0])
    feats = osp.hard_simple_vfe(v.cpu(), n.cpu())
    mid = lg.FrozenSparseEncoder(sparse_shape=(41, grid, grid), dtype=torch.float32)
    sd = mid.synthetic_state(0)
    want = osp.second_fpn(osp.second(osp.sparse_encoder(feats, torch.nn.functional.pad(c.cpu().long(), (1, 0)), 1, sd, (41, grid, grid)),
                                     lg.FrozenSECOND().synthetic_state(1)), lg.FrozenSECONDFPN().synthetic_state(2))[0]
    err = (out.float().cpu() - want).abs()
    assert err.max().item() <= 3e-2 * max(1.0, want.abs().max().item()) and err.mean().item() <= 2e-3 * max(1.0, want.abs().max().item())
