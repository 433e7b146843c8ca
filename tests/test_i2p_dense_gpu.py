"""The matrix-core pillar attention (csrc/i2p_dense.hip, round 6) against the wave-per-cell kernel it replaces in the fp16
inference forward (csrc/cross_modal.hip; that one is checked against the oracle's MMRI_I2P - reference
encoder_utils.py:226-320 - in tests/test_encoder_gpu.py::test_i2p_module, which now runs the dense kernel as well)."""
import pytest
import torch

from deepinteraction_amd import ops, synth
from deepinteraction_amd.geometry import SampleGeometry

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _sample(shape, seed):
    inp = synth.make_inputs(1, shape, seed=seed)
    Hi, Wi = shape['img_hw']
    geom = SampleGeometry(inp['img_metas'][0], (Hi, Wi), DEV)
    pm = inp['pts_metas']
    args = (pm['pillars'].to(DEV), pm['pillar_coors'].to(DEV), pm['pillars_num_points'].to(DEV), geom.lidar2img, geom.aug_rev,
            geom.ori_hw)
    return args


def _maps(shape, seed, qscale):
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    g = torch.Generator(device=DEV).manual_seed(seed)
    img = torch.randn(6, 128, Hi, Wi, device=DEV, generator=g).half().contiguous(memory_format=torch.channels_last)
    qf = (torch.randn(1, 128, Hb, Wb, device=DEV, generator=g) * qscale).half().contiguous(memory_format=torch.channels_last)
    return img, qf


def _both(shape, args, img, qf, sector_order):
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    keys_d = ops.i2p_key_table(*args, (Hi, Wi), (Hb, Wb), dense=True, sector_order=sector_order)
    keys_w = ops.i2p_key_table(*args, (Hi, Wi), (Hb, Wb), dense=False)
    assert keys_d.dense is not None and keys_w.dense is None
    got, gv = ops.i2p_attention(img, qf, *args, keys=keys_d, sector_order=sector_order)
    ref, rv = ops.i2p_attention(img, qf, *args, keys=keys_w, sector_order=sector_order)
    torch.cuda.synchronize()
    return got.float(), gv.float(), ref.float(), rv.float(), keys_d


@pytest.mark.parametrize('sector_order', [True, False])
@pytest.mark.parametrize('shape_name,qscale', [('TINY', 0.1), ('R', 0.1), ('R', 1.0)])
def test_dense_kernel_matches_the_wave_per_cell_kernel(shape_name, qscale, sector_order):
    """Same inputs, same key table: every cell of ctx / valid.  qscale 1.0: logits of +-30 and more - one-hot soft-maxes and
    the lazy reference maximum's rescale path (a later key beating the first by more than 2^8)."""
    shape = synth.SHAPE_R if shape_name == 'R' else synth.SHAPE_TINY
    args = _sample(shape, seed=3)
    img, qf = _maps(shape, 5, qscale)
    got, gv, ref, rv, keys = _both(shape, args, img, qf, sector_order)
    assert torch.equal(gv, rv)
    assert not torch.isnan(got).any()
    empty = rv[0, 0] == 0
    assert got[0][:, empty].abs().max().item() == 0
    # fp16 outputs of a convex combination of fp16 rows: 1 ulp of the output + the fp16 rounding of the probabilities
    scale = ref.abs().max().item()
    err = (got - ref).abs()
    assert err.max().item() <= 4e-3 * scale, (err.max().item(), scale)
    assert err.mean().item() <= 2e-4 * scale
    # the dense stream: every group owns whole superblocks (8 keys), at least one, enough for its keys; the groups cover
    # every cell once; no group exceeds the key cap unless it is a single cell
    Hb, Wb = shape['bev_hw']
    ncell = Hb * Wb
    cnt = keys.table[:ncell * 4].view(torch.int32).cpu()
    nchunks = (ncell + 7) // 8
    al = lambda x: (x + 255) // 256 * 256
    off_csub = al(ncell * 64)
    ngroups = int(keys.dense[off_csub + nchunks * 4:off_csub + nchunks * 4 + 4].view(torch.int32)[0])
    assert nchunks <= ngroups <= ncell
    hdr = keys.dense[:ngroups * 64].view(torch.int32).view(ngroups, 16).cpu()
    cells, sb_begin, sb_end, nk = hdr[:, :8], hdr[:, 8], hdr[:, 9], hdr[:, 10]
    assert int(sb_begin[0]) == 0 and bool((sb_begin[1:] == sb_end[:-1]).all())
    span = (sb_end - sb_begin) * 8
    assert bool((span >= 8).all()) and bool((span >= nk).all())
    assert bool(((span < nk + 8) | ((nk == 0) & (span == 8))).all())
    assert int(nk.sum()) == int(cnt.sum())
    seen = cells[cells >= 0]
    assert seen.numel() == ncell and seen.unique().numel() == ncell
    single = (cells >= 0).sum(1) == 1
    assert bool(((nk <= 48) | single).all())
    per_group = torch.where(cells >= 0, cnt[cells.clamp(min=0).long()], torch.zeros_like(cells)).sum(1)
    assert torch.equal(per_group.to(torch.int32), nk)


def test_dense_kernel_is_permutation_invariant_and_convex():
    """Size-independent properties at the benched shape: a constant image gives that constant wherever a key exists (the
    soft-max weights sum to one, bilinear footprints over the edge sample zero padding: <= the constant), and the walk
    order of the cells does not change any value (the grouping does: other neighbours share an MFMA tile)."""
    shape = synth.SHAPE_R
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    args = _sample(shape, seed=0)
    img, qf = _maps(shape, 9, 0.3)
    k1 = ops.i2p_key_table(*args, (Hi, Wi), (Hb, Wb), sector_order=True)
    k2 = ops.i2p_key_table(*args, (Hi, Wi), (Hb, Wb), sector_order=False)
    a, va = ops.i2p_attention(img, qf, *args, keys=k1, sector_order=True)
    b, vb = ops.i2p_attention(img, qf, *args, keys=k2, sector_order=False)
    assert torch.equal(va, vb)
    assert (a.float() - b.float()).abs().max().item() <= 2e-3 * a.float().abs().max().item()
    const = torch.full_like(img, 0.75)
    c, vc = ops.i2p_attention(const, qf, *args, keys=k1, sector_order=True)
    v = vc[0, 0] > 0
    c = c[0].float()
    assert c[:, v].max().item() <= 0.75 + 1e-3 and c[:, ~v].abs().max().item() == 0
    assert (c[:, v] > 0.75 - 1e-3).float().mean().item() > 0.9


def test_dense_kernel_replays_bit_identically():
    """Two launches on the same inputs are bit-identical (no atomics, fixed summation order) - what graph replay relies on."""
    shape = synth.SHAPE_R
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    args = _sample(shape, seed=1)
    img, qf = _maps(shape, 2, 0.3)
    outs = []
    for _ in range(3):
        k = ops.i2p_key_table(*args, (Hi, Wi), (Hb, Wb))
        outs.append(ops.i2p_attention(img, qf, *args, keys=k)[0].clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_dense_kernel_without_any_pillar():
    """Edge case: a sample without pillars (P = 0) - every group of the stream is one padding superblock; every cell of ctx and
    valid is written with zeros (the caller allocates them uninitialised)."""
    shape = synth.SHAPE_TINY
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    args = list(_sample(shape, seed=3))
    args[0], args[1], args[2] = args[0][:0].contiguous(), args[1][:0].contiguous(), args[2][:0].contiguous()
    img, qf = _maps(shape, 5, 0.3)
    k = ops.i2p_key_table(*args, (Hi, Wi), (Hb, Wb))
    assert k.dense is not None
    ctx, valid = ops.i2p_attention(img, qf, *args, keys=k)
    torch.cuda.synchronize()
    assert float(ctx.float().abs().max()) == 0.0 and float(valid.float().abs().max()) == 0.0


_TUNED_SCRIPT = r'''
import torch
from deepinteraction_amd import ops, synth
from deepinteraction_amd.geometry import SampleGeometry
shape = synth.SHAPE_R
Hi, Wi = shape['img_hw']; Hb, Wb = shape['bev_hw']
inp = synth.make_inputs(1, shape, seed=7)
geom = SampleGeometry(inp['img_metas'][0], (Hi, Wi), 'cuda')
pm = inp['pts_metas']
args = (pm['pillars'].cuda(), pm['pillar_coors'].cuda(), pm['pillars_num_points'].cuda(), geom.lidar2img, geom.aug_rev, geom.ori_hw)
g = torch.Generator(device='cuda').manual_seed(1)
img = torch.randn(6, 128, Hi, Wi, device='cuda', generator=g).half().contiguous(memory_format=torch.channels_last)
qf = (torch.randn(1, 128, Hb, Wb, device='cuda', generator=g) * 0.5).half().contiguous(memory_format=torch.channels_last)
kd = ops.i2p_key_table(*args, (Hi, Wi), (Hb, Wb))
kw = ops.i2p_key_table(*args, (Hi, Wi), (Hb, Wb), dense=False)
a, va = ops.i2p_attention(img, qf, *args, keys=kd)
b, vb = ops.i2p_attention(img, qf, *args, keys=kw)
torch.cuda.synchronize()
assert torch.equal(va, vb) and not torch.isnan(a.float()).any()
sc = b.float().abs().max().item()
assert (a.float() - b.float()).abs().max().item() <= 4e-3 * sc
ncell = Hb * Wb
al = lambda x: (x + 255) // 256 * 256
nchunks = (ncell + 7) // 8
off = al(ncell * 64) + nchunks * 4
ngroups = int(kd.dense[off:off + 4].view(torch.int32)[0])
print('TUNED_OK', ngroups)
'''


@pytest.mark.parametrize('env', [dict(DI_I2PD_KEYCAP='6', DI_I2PD_WAVES='3'), dict(DI_I2PD_KEYCAP='400', DI_I2PD_NB='3', DI_I2PD_WAVES='5')])
def test_dense_kernel_with_other_group_and_share_sizes(env):
    """The stream's cut is a tuning choice, not part of the result: a key cap of 6 (almost every cell its own group, single
    cells far over the cap, ~4x the groups) with 3 shares per CU, and no cap at all (always 8 cells per group: groups of up to
    ~25 superblocks) with the three-buffer pipeline - same outputs as the wave-per-cell kernel.  (Own process: the switches are
    read at the first launch.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''), **env)
    r = subprocess.run([sys.executable, '-c', _TUNED_SCRIPT], env=e, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'TUNED_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    ngroups = int(r.stdout.split('TUNED_OK')[1].split()[0])
    assert (ngroups > 12000) if env['DI_I2PD_KEYCAP'] == '6' else (ngroups == (180 * 180 + 7) // 8)
