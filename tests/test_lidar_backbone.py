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
    assert net.middle._h and net.backbone._h               # the device kernels ran: sparse_conv.hip and the own 3x3 kernel
    net.middle.HIP = net.backbone.HIP = False
    ref16 = net(pts)[0]                                     # the torch formulation of both, same fp16 weights
    net.middle.HIP = net.backbone.HIP = True
    assert (out.float() - ref16.float()).abs().max().item() <= 2e-2 * max(1.0, ref16.float().abs().max().item())
    v, c, n = net.pts_voxel_layer(pts[0])
    feats = osp.hard_simple_vfe(v.cpu(), n.cpu())
    mid = lg.FrozenSparseEncoder(sparse_shape=(41, grid, grid), dtype=torch.float32)
    sd = mid.synthetic_state(0)
    want = osp.second_fpn(osp.second(osp.sparse_encoder(feats, torch.nn.functional.pad(c.cpu().long(), (1, 0)), 1, sd, (41, grid, grid)),
                                     lg.FrozenSECOND().synthetic_state(1)), lg.FrozenSECONDFPN().synthetic_state(2))[0]
    err = (out.float().cpu() - want).abs()
    assert err.max().item() <= 3e-2 * max(1.0, want.abs().max().item()) and err.mean().item() <= 2e-3 * max(1.0, want.abs().max().item())


# ---------------------------------------------------------------- csrc/sparse_conv.hip (the device path of FrozenSparseEncoder)
def _sorted_level(shape, batch, n, seed, dev):
    _, coors = _voxels(shape, batch, n, seed)
    x = lg.SparseTensor(coors.long().to(dev), torch.zeros(coors.shape[0], 1, device=dev), shape, batch)
    order = torch.argsort(x.keys())
    x.coords = x.coords[order]
    return x


@pytest.mark.gpu
@pytest.mark.parametrize('shape,batch', [((9, 40, 36), 2), ((41, 64, 72), 1)])
def test_device_rulebooks_equal_the_searchsorted_rulebooks(shape, batch):
    """`di_sparse_nbr` / `di_sparse_mark` against the torch rulebooks (which tests above pin to brute force): integer work, equal."""
    from deepinteraction_amd import ops
    dev = 'cuda'
    x = _sorted_level(shape, batch, 600, 5, dev)
    keys = x.keys().to(torch.int32)
    nbr = ops.sparse_neighbours(keys, keys, batch, shape, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    assert torch.equal(nbr.t().long(), lg.subm_rulebook(x))
    for ks, st, pd in (((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))):
        ocoords, onbr, oshape = lg.strided_rulebook(x, ks, st, pd)
        okeys, oshape2 = ops.sparse_output_keys(keys, batch, shape, ks, st, pd)
        assert oshape2 == oshape
        want_keys = ((ocoords[:, 0] * oshape[0] + ocoords[:, 1]) * oshape[1] + ocoords[:, 2]) * oshape[2] + ocoords[:, 3]
        assert torch.equal(okeys.long(), want_keys)
        got = ops.sparse_neighbours(keys, okeys, batch, shape, oshape, ks, st, pd)
        assert torch.equal(got.t().long(), onbr)
    # an empty level
    e = torch.empty(0, dtype=torch.int32, device=dev)
    ok, osh = ops.sparse_output_keys(e, 1, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    assert ok.numel() == 0 and ops.sparse_neighbours(e, ok, 1, shape, osh, (3, 3, 3), (2, 2, 2), (1, 1, 1)).shape == (27, 0)


@pytest.mark.gpu
@pytest.mark.parametrize('cin,cout,K', [(8, 16, 27), (16, 16, 27), (16, 32, 27), (32, 32, 27), (32, 64, 27), (64, 64, 27),
                                        (64, 128, 27), (128, 128, 27), (128, 128, 3)])
def test_sparse_conv_kernel_equals_gather_and_product(cin, cout, K):
    """One launch (gather, MFMA product, bias / residual / ReLU) against the float64 gather + product of the same fp16 numbers:
    ragged row counts, rows without any neighbour, offsets nobody has, with and without residual / bias / ReLU."""
    from deepinteraction_amd import ops
    dev = 'cuda'
    g = torch.Generator().manual_seed(cin * 1000 + cout + K)
    for M_in, M_out, fill in ((1000, 777, 0.3), (5, 128, 0.05), (300, 1, 1.0), (4000, 4099, 0.6)):
        feats = torch.randn(M_in, cin, generator=g).half()
        real = 5 if cin == 8 else cin                       # conv_input: 5 real channels, rows padded to 8
        feats[:, real:] = 0
        nbr = torch.randint(0, M_in, (K, M_out), generator=g, dtype=torch.int32)
        nbr[torch.rand(K, M_out, generator=g) > fill] = -1
        nbr[K // 2] = -1                                     # an offset no voxel has
        nbr[:, M_out // 2] = -1                              # a voxel without neighbours
        w = torch.randn(K * real, cout, generator=g) * (2.0 / (K * real * fill + 1)) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        res = torch.randn(M_out, cout, generator=g).half()
        frag, cin_pad = ops.sparse_weight_fragments(w, K, real)
        w16 = w.half().double().view(K, real, cout)
        X = torch.cat([feats[:, :real].double(), torch.zeros(1, real, dtype=torch.float64)], 0)[nbr.long()]      # (K, M_out, real)
        base = torch.einsum('kmc,kcd->md', X, w16)
        for use_bias, use_res, relu in ((True, False, True), (True, True, True), (False, False, False)):
            want = base + (bias.double() if use_bias else 0) + (res.double() if use_res else 0)
            want = want.relu() if relu else want
            got = ops.sparse_conv(ops.sparse_rows(feats.to(dev), cin), nbr.to(dev), frag.to(dev), bias.to(dev) if use_bias else None, cin_pad,
                                  cout, relu=relu, residual=ops.sparse_rows(res.to(dev), cout) if use_res else None)
            assert got.shape == (M_out + 1, cout) and not bool(got[-1].any())                       # the zero row travels with the rows
            err = (got[:-1].double().cpu() - want).abs()
            tol = 2e-3 * max(1.0, want.abs().max().item())                       # fp16 output rounding
            assert err.max().item() <= tol, (cin, cout, K, M_out, use_bias, use_res, relu, err.max().item(), tol)
    out = ops.sparse_conv(ops.sparse_rows(feats.to(dev), cin), torch.empty((K, 0), dtype=torch.int32, device=dev), frag.to(dev), None, cin_pad, cout)
    assert out.shape == (1, cout) and not bool(out.any())


@pytest.mark.gpu
def test_sparse_encoder_device_kernels_against_the_torch_formulation_and_the_oracle():
    """The whole `FrozenSparseEncoder` on `csrc/sparse_conv.hip` (fp16) == its torch formulation on the device (fp16, same
    weights) within fp16 round-off, and the float32 dense oracle within the fp16 budget; two samples, one of them nearly empty."""
    dev = 'cuda'
    shape = (41, 64, 72)
    feats, coors = _voxels(shape, 1, 900, 11)
    f2, c2 = _voxels(shape, 1, 3, 12)
    c2[:, 0] = 1
    feats, coors = torch.cat([feats, f2], 0), torch.cat([coors, c2], 0)
    mid = lg.FrozenSparseEncoder(sparse_shape=shape, dtype=torch.float16)
    sd = mid.synthetic_state(3)
    mid.load_mmdet_state(sd).to(dev)
    assert mid._h and mid.HIP
    got = mid(feats.to(dev), coors.to(dev), 2)
    mid.HIP = False
    ref16 = mid(feats.to(dev), coors.to(dev), 2)
    mid.HIP = True
    want = osp.sparse_encoder(feats, coors, 2, sd, shape)
    assert got.shape == want.shape == ref16.shape
    scale = max(1.0, want.abs().max().item())
    assert (got.float() - ref16.float()).abs().max().item() <= 1e-2 * scale
    err = (got.float().cpu() - want).abs()
    assert err.max().item() <= 3e-2 * scale and err.mean().item() <= 2e-3 * scale
    assert bool(((got != 0).any(1) == (ref16 != 0).any(1)).all())                    # the same active cells


@pytest.mark.gpu
def test_sparse_encoder_device_kernels_on_an_empty_scan():
    """No voxel at all: every level is empty, the BEV map is zero (the kernels are launched on empty levels, nothing is read)."""
    dev = 'cuda'
    shape = (41, 64, 72)
    mid = lg.FrozenSparseEncoder(sparse_shape=shape, dtype=torch.float16)
    mid.load_mmdet_state(mid.synthetic_state(3)).to(dev)
    out = mid(torch.zeros(0, 5, device=dev), torch.zeros(0, 4, dtype=torch.int32, device=dev), 1)
    assert out.shape == (1, 256, 8, 9) and not bool(out.any())


def test_sparse_weights_in_the_spconv1_layout_of_mmdet3d_0_17_load_the_same():
    """mmdet3d 0.17.1 bundles spconv 1.x: `SparseConvolution.weight` is (kD, kH, kW, C_in, C_out); spconv 2.x stores (C_out, kD, kH,
    kW, C_in).  A state dict in either layout gives the same module."""
    shape = (41, 32, 40)
    feats, coors = _voxels(shape, 1, 300, 21)
    a = lg.FrozenSparseEncoder(sparse_shape=shape, dtype=torch.float32)
    sd = a.synthetic_state(5)
    sd1 = {k: (v.permute(1, 2, 3, 4, 0).contiguous() if v.dim() == 5 else v) for k, v in sd.items()}
    assert sd1['conv_out.0.weight'].shape == (3, 1, 1, 128, 128) and sd1['conv_input.0.weight'].shape == (3, 3, 3, 5, 16)
    b = lg.FrozenSparseEncoder(sparse_shape=shape, dtype=torch.float32)
    ya = a.load_mmdet_state(sd).to('cpu')(feats, coors, 1)
    yb = b.load_mmdet_state(sd1).to('cpu')(feats, coors, 1)
    assert torch.equal(ya, yb)
    with pytest.raises(ValueError):
        lg.spconv2_layout(torch.zeros(3, 3, 16, 16), (3, 3, 3), 16, 16)


@pytest.mark.gpu
def test_lidar_branch_at_the_benched_shape_equals_its_torch_formulation():
    """Shape R as `bench.py --from-lidar` runs it (262 144 points, 1440 x 1440 x 41 voxels of 0.075 m, 160 000 -> 402 000 -> 337 000
    -> 109 000 -> 51 000 active voxels): the branch on `csrc/sparse_conv.hip` + the own 3x3 kernel against the torch formulation of
    the same fp16 weights - same active BEV cells, values within fp16 round-off of 21 chained convolutions."""
    from deepinteraction_amd import synth
    dev = 'cuda'
    rng = list(synth.PC_RANGE)
    grid = 1440
    layer = dict(max_num_points=10, max_voxels=(120000, 160000), point_cloud_range=rng,
                 voxel_size=[(rng[3] - rng[0]) / grid, (rng[4] - rng[1]) / grid, (rng[5] - rng[2]) / 41.0])
    net = lg.FrozenLidarBackbone.synthetic(layer, (41, grid, grid), dev).eval()
    pts = [synth.make_inputs(1, synth.SHAPE_R, seed=0)['pts_metas']['pts'][0].to(dev)]
    v, c, n = net.pts_voxel_layer(pts[0])
    feats = v[:, :, :5].sum(1) / n.clamp(min=1).to(v.dtype).unsqueeze(-1)
    coors = torch.nn.functional.pad(c.long(), (1, 0))
    got = net.middle(feats, coors, 1)
    net.middle.HIP = False
    want = net.middle(feats, coors, 1)
    net.middle.HIP = True
    assert got.shape == want.shape == (1, 256, 180, 180)
    assert bool(((got != 0).any(1) == (want != 0).any(1)).all())
    scale = want.float().abs().max().item()
    err = (got.float() - want.float()).abs()
    assert err.max().item() <= 5e-3 * scale and err.mean().item() <= 2e-4 * scale, (err.max().item(), err.mean().item(), scale)
    out = net(pts)[0]
    assert out.shape == (1, 512, 180, 180) and bool(torch.isfinite(out).all())


@pytest.mark.gpu
@pytest.mark.parametrize('ks,st,pd', [((1, 1, 1), (1, 1, 1), (0, 0, 0)), ((1, 3, 3), (1, 1, 1), (0, 1, 1)), ((3, 3, 1), (1, 2, 1), (1, 0, 0)),
                                      ((2, 2, 2), (2, 2, 2), (0, 0, 0)), ((3, 3, 3), (1, 2, 2), (1, 1, 0)), ((3, 2, 3), (2, 1, 2), (0, 1, 1)),
                                      ((3, 3, 3), (3, 3, 3), (1, 1, 1))])
def test_device_rulebooks_for_other_geometries(ks, st, pd):
    """Kernel extents 1 - 3, strides 1 - 3, paddings 0 - 1 per axis (not only the encoder's three layer types): output set and
    neighbour table of `di_sparse_mark` / `di_sparse_nbr` equal the torch rulebook's, two samples, voxels on every border."""
    from deepinteraction_amd import ops
    dev = 'cuda'
    shape, batch = (7, 23, 18), 2
    x = _sorted_level(shape, batch, 400, 9, dev)
    keys = x.keys().to(torch.int32)
    ocoords, onbr, oshape = lg.strided_rulebook(x, ks, st, pd)
    okeys, oshape2 = ops.sparse_output_keys(keys, batch, shape, ks, st, pd)
    assert oshape2 == oshape
    want_keys = ((ocoords[:, 0] * oshape[0] + ocoords[:, 1]) * oshape[1] + ocoords[:, 2]) * oshape[2] + ocoords[:, 3]
    assert torch.equal(okeys.long(), want_keys)
    assert torch.equal(ops.sparse_neighbours(keys, okeys, batch, shape, oshape, ks, st, pd).t().long(), onbr)


def test_operand_fragment_order_and_row_form_of_the_device_kernels():
    """Host-side constants of `csrc/sparse_conv.hip` (no GPU): weight fragments [K][cin_pad / 32][cout / 16][lane = 16 g + i][8] with
    element e = W[o][32 kk + 8 g + e][16 mt + i] (zero beyond cin), feature rows padded to a multiple of 8 channels + the zero row."""
    from deepinteraction_amd import ops
    g = torch.Generator().manual_seed(0)
    for K, cin, cout in ((27, 5, 16), (27, 16, 32), (3, 128, 128), (27, 64, 128)):
        w = torch.randn(K * cin, cout, generator=g)
        frag, cin_pad = ops.sparse_weight_fragments(w, K, cin)
        assert cin_pad == (cin + 31) // 32 * 32 and frag.shape == (K, cin_pad // 32, cout // 16, 4, 16, 8) and frag.dtype == torch.float16
        w3 = w.view(K, cin, cout).half()
        for (o, kk, mt, gq, i, e) in ((0, 0, 0, 0, 0, 0), (K - 1, cin_pad // 32 - 1, cout // 16 - 1, 3, 15, 7), (K // 2, 0, 0, 1, 7, 3)):
            c = 32 * kk + 8 * gq + e
            want = w3[o, c, 16 * mt + i] if c < cin else torch.tensor(0.0).half()
            assert frag[o, kk, mt, gq, i, e] == want
    f = torch.randn(7, 5, generator=g)
    rows = ops.sparse_rows(f)
    assert rows.shape == (8, 8) and rows.dtype == torch.float16 and not bool(rows[-1].any()) and not bool(rows[:, 5:].any())
    assert torch.equal(rows[:7, :5], f.half())
