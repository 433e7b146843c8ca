"""GPU parity of the MMPI decoder path (HIP kernels through the C ABI + batched host logic)
against the CPU oracle on the same seeded inputs."""
import copy
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from deepinteraction_amd import ops, synth
from oracle import configs, decoder as odec
from oracle.refpin import make_golden as mg
from oracle.thirdparty import apply_3d_transformation, lidar_box_corners, roi_align_v2

DEV = 'cuda'


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_heatmap_nms(dtype):
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(2, 10, 20, 24, generator=g), torch.randn(2, 10, 20, 24, generator=g)
    a, b = a.to(dtype), b.to(dtype)
    got = ops.heatmap_nms(a.to(DEV), b.to(DEV), 3, (8, 9)).cpu()
    heat = (a.float().sigmoid() + b.float().sigmoid()) / 2
    lm = torch.zeros_like(heat)
    lm[:, :, 1:-1, 1:-1] = torch.nn.functional.max_pool2d(heat, 3, 1, 0)
    lm[:, 8], lm[:, 9] = heat[:, 8], heat[:, 9]
    ref = heat * (heat == lm)
    # INT part (which cells survive) bit-exact; surviving values to float round-off of the sigmoid
    assert torch.equal(got != 0, ref != 0)
    assert torch.allclose(got, ref, atol=2e-6)


def _res(B, Q, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(center=torch.rand(B, 2, Q, generator=g) * 34 + 1, height=torch.randn(B, 1, Q, generator=g) * 0.8 - 1,
                dim=torch.randn(B, 3, Q, generator=g) * 0.4 + 0.5, rot=torch.randn(B, 2, Q, generator=g),
                heatmap=torch.randn(B, 10, Q, generator=g), vel=torch.randn(B, 2, Q, generator=g))


@pytest.mark.parametrize('aug', [False, True])
def test_query_geometry(aug):
    """decode + corners + projection + rectangles vs the oracle's restatement of the same chain
    (decoder_utils.py:666-738, :804-819).  on_img is an INT output: exact except where the
    projected centre is within float round-off of the image border."""
    from deepinteraction_amd.mmdet3d_plugin.models.utils.decoder_utils import QueryGeometry
    B, Q = 2, 50
    a = synth.example_aug(2) if aug else None
    metas = synth.make_inputs(B, synth.SHAPE_TINY, seed=0, aug=a)['img_metas']
    metas[1]['flip'] = True
    metas[1]['img_crop_offset'] = (3.0, -2.0)
    res = _res(B, Q, 1)
    cfg = configs.decoder_cfg(bev=36)
    coder = odec.TransFusionBBoxCoder(**{k: v for k, v in cfg['bbox_coder'].items() if k != 'type'})
    geom = QueryGeometry(metas, DEV)
    cell = cfg['test_cfg']['out_size_factor'] * cfg['test_cfg']['voxel_size'][0]
    rd = {k: v.to(DEV) for k, v in res.items()}
    on, ri, _ = ops.query_geometry(rd, geom.proj, geom.aug_rev, geom.per_sample, cell, (-54.0, -54.0), 1.0, 1.0,
                                   True, False)
    _, _, rb = ops.query_geometry(rd, None, None, None, cell, (-54.0, -54.0), cell, 2.0, False, True)
    on, ri, rb = on.cpu(), ri.cpu(), rb.cpu()
    boxes = coder.decode(res['heatmap'], res['rot'], res['dim'], res['center'], res['height'], res['vel'])
    pos = res['center'] * cell - 54.0
    pos3d = torch.cat([pos, res['height']], 1)
    for s in range(B):
        m = metas[s]
        h, w = m['input_shape']
        cor = lidar_box_corners(boxes[s]['bboxes'][:, :7])
        pts = torch.cat([pos3d[s], cor.permute(2, 0, 1).reshape(3, -1)], -1).T
        pts = apply_3d_transformation(pts, 'LIDAR', m, reverse=True)
        for v in range(6):
            p2 = torch.cat([pts, torch.ones(len(pts), 1)], -1) @ torch.tensor(np.asarray(m['lidar2img'][v]), dtype=torch.float32).t()
            z = p2[:, 2].clamp(min=1e-5)
            cx, cy = p2[:, 0] / z, p2[:, 1] / z
            if 'img_crop_offset' in m:
                cx, cy = cx - m['img_crop_offset'][0], cy - m['img_crop_offset'][1]
            if m.get('flip', False):
                cx = m['img_shape'][0][1] - cx
            ref_on = (cx[:Q] > 0) & (cx[:Q] < w) & (cy[:Q] > 0) & (cy[:Q] < h)
            margin = torch.minimum(torch.minimum(cx[:Q], w - cx[:Q]).abs(), torch.minimum(cy[:Q], h - cy[:Q]).abs())
            sure = margin > 1e-2
            assert torch.equal(on[s, v].bool()[sure], ref_on[sure])
            X, Y = cx[Q:].reshape(Q, 8), cy[Q:].reshape(Q, 8)
            ref_rect = torch.stack([X.min(1).values, Y.min(1).values, X.max(1).values, Y.max(1).values], 1)
            vis = ref_on & sure                              # rects matter only for on-image queries
            assert torch.allclose(ri[s, v][vis], ref_rect[vis], rtol=2e-4, atol=2e-2)
        box = boxes[s]['bboxes'][:, :7].clone()
        box[:, 3:6] *= 2
        cc = (lidar_box_corners(box)[..., :2] + 54.0) / cell
        ref_b = torch.stack([cc[..., 0].min(-1).values, cc[..., 1].min(-1).values,
                             cc[..., 0].max(-1).values, cc[..., 1].max(-1).values], -1)
        assert torch.allclose(rb[s], ref_b, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-5), (torch.float16, 1e-3)])
def test_roi_align(dtype, tol):
    g = torch.Generator().manual_seed(0)
    N, C, H, W, R = 3, 128, 14, 22, 40
    feat = torch.randn(N, C, H, W, generator=g).to(dtype)
    x0, y0 = torch.rand(R, generator=g) * W * 4 - 8, torch.rand(R, generator=g) * H * 4 - 8
    bw, bh = torch.rand(R, generator=g) * 40 + 0.5, torch.rand(R, generator=g) * 30 + 0.5
    n = torch.randint(0, N, (R,), generator=g).float()
    rois = torch.stack([n, x0, y0, x0 + bw, y0 + bh], 1)
    rois[0] = torch.tensor([0, -500.0, -500.0, -400.0, -400.0])          # entirely outside -> zeros
    rois[1] = torch.tensor([1, 10.0, 10.0, 10.0, 10.0])                  # zero-size box
    got = ops.roi_align(feat.to(DEV), rois.to(DEV), 0.25).float().cpu()  # (R,49,C)
    for r in range(R):
        ref = roi_align_v2(feat[int(n[r] if r > 1 else rois[r, 0])][None].float(), rois[r:r + 1, 1:], 7, 0.25, 2)
        ref = ref[0].flatten(1).t()                                       # (49,C)
        assert torch.allclose(got[r], ref, atol=tol * max(1.0, ref.abs().max().item())), r
    assert torch.all(got[0] == 0)
    # known answer: RoIAlign of a linear ramp is exact (bilinear reproduces linear functions)
    ramp = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W).expand(1, 8, H, W).contiguous()
    roi = torch.tensor([[0, 8.0, 8.0, 64.0, 40.0]])
    out = ops.roi_align(ramp.to(DEV), roi.to(DEV), 0.25).cpu()[0, :, 0].view(7, 7)
    centres = (8 * 0.25 - 0.5) + (torch.arange(7) + 0.5) * (56 * 0.25 / 7)
    assert torch.allclose(out, centres.view(1, 7).expand(7, 7), atol=1e-5)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.float16, 1e-3)])
@pytest.mark.parametrize('shape', [(2, 37, 1300, 8), (1, 200, 32400, 8), (1, 300, 700, 4)])
def test_mha_decode(dtype, tol, shape):
    B, Q, S, H = shape
    E = H * 16
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, Q, E, generator=g).to(dtype)
    kv = torch.randn(B, S, 2 * E, generator=g).to(dtype)
    kv[0, 5, :E] *= 6.0                                                   # a dominant key forces max updates
    got = ops.mha_decode(q.to(DEV), kv.to(DEV), H, 0.25).float().cpu()
    qh = q.float().view(B, Q, H, 16).transpose(1, 2) * 0.25
    kh = kv.float()[..., :E].reshape(B, S, H, 16).transpose(1, 2)
    vh = kv.float()[..., E:].reshape(B, S, H, 16).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2), -1) @ vh).transpose(1, 2).reshape(B, Q, E)
    assert torch.allclose(got, ref, atol=tol * max(1.0, ref.abs().max().item())), (got - ref).abs().max()


def _narrow(metas):
    metas = copy.deepcopy(metas)
    for m in metas:
        m['lidar2img'] = [m['lidar2img'][0]] * 2 + [m['lidar2img'][1]] * 2 + [m['lidar2img'][2]] * 2
    return metas


@pytest.mark.parametrize('narrow', [False, True])
@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-3), (torch.float16, 5e-2)])
def test_decoder_forward(dtype, tol, narrow):
    """Whole DeepInteractionDecoder forward, shared state_dict.  INT outputs (query labels,
    on-image masks) bit-exact.  fp32: summation-order noise through 1 decoder layer + 4 RoI
    layers (LayerNorms re-normalise, so absolute 2e-3 on O(1..40) outputs).  fp16 features:
    the query path runs through ~40 fp16 GEMM/LN stages; prediction heads are fp32."""
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionDecoder
    shape = synth.SHAPE_TINY
    cfg = configs.decoder_cfg(bev=36, num_proposals=24 if narrow else 40)
    torch.manual_seed(7)
    O = odec.DeepInteractionDecoder(**cfg)
    mg.randomize(O, 9)
    M = DeepInteractionDecoder(**cfg)
    M.load_state_dict(O.state_dict())
    O.eval(), M.eval()
    g = torch.Generator().manual_seed(0)
    Hi, Wi = shape['img_hw']
    p0, p1 = torch.randn(2, 128, 36, 36, generator=g), torch.randn(2, 128, 36, 36, generator=g)
    img = torch.randn(12, 128, Hi, Wi, generator=g)
    metas = synth.make_inputs(2, shape, seed=0)['img_metas']
    if narrow:
        metas = _narrow(metas)
    p0q, p1q, imq = p0.to(dtype), p1.to(dtype), img.to(dtype)
    with torch.no_grad():
        M = M.to(DEV, dtype)
        o = M([p0q.to(DEV), p1q.to(DEV)], imq.to(DEV), metas)[0][0]
        # fp16 heat-map logits may reorder near-tied proposals: the continuous comparison below
        # then uses the product's picks; the picks themselves are checked separately
        r_free = O([p0q.float(), p1q.float()], imq.float(), metas)[0][0]
        labels_free, masks_free = O.query_labels, O.on_the_image_mask
        r = r_free if dtype == torch.float32 else O([p0q.float(), p1q.float()], imq.float(), metas,
                                                    top_override=M.top_proposals.cpu())[0][0]
    if dtype == torch.float32:
        assert torch.equal(M.query_labels.cpu(), labels_free)                 # INT: bit-exact
        for a, b in zip(M.on_the_image_mask, masks_free):
            assert torch.equal(a.cpu(), b)
    else:
        same = (M.query_labels.cpu() == labels_free).float().mean().item()
        assert same >= 0.9, same
    if narrow:
        assert (~masks_free[0]).any()                                         # fallback path exercised
    Q = cfg['num_proposals']
    for k in r:
        got = o[k].float().cpu()
        assert got.shape == r[k].shape, k
        d = (got - r[k]).abs()
        scale = max(1.0, r[k].abs().max().item())
        if dtype == torch.float32:
            assert d.max().item() <= tol * scale, (k, d.max().item())
        else:
            # a query whose projected centre sits within fp16 error of an image border can flip
            # its on-image mask from layer 2 on (a discrete change): bound the bulk, not the max
            assert d.median().item() <= 1e-2 * scale, (k, d.median().item())
            assert (d > tol * scale).float().mean().item() <= 0.15, (k, (d > tol * scale).float().mean().item())
            if k not in ('dense_heatmap', 'query_heatmap_score'):
                first = d[..., :2 * Q]                                       # layers before any flip can matter
                assert first.max().item() <= 0.3 * scale * 0.05 + 0.15, (k, first.max().item())


@pytest.mark.parametrize('B,N,k', [(1, 324000, 200), (2, 324000, 400), (3, 5000, 1024), (1, 300, 300), (2, 70000, 1)])
def test_topk_radix_select(B, N, k):
    """di_topk_fwd: tie-free data -> exactly torch.topk's indices; heavy ties (quantised scores, many zeros) -> the
    same multiset of values, descending, and within equal values ascending indices starting from the lowest ones."""
    g = torch.Generator().manual_seed(N + k)
    x = torch.rand(B, N, generator=g)
    x[torch.rand(B, N, generator=g) < 0.6] = 0.0                              # the NMS zeroes most cells
    xd = x.to(DEV)
    idx, val = ops.topk(xd, k, with_values=True)
    ref = torch.topk(x, k, dim=-1, largest=True, sorted=True)
    assert torch.equal(val.cpu(), ref.values)
    pos = ref.values > 0                                                     # positive scores are distinct here
    assert torch.equal(idx.cpu()[pos], ref.indices[pos])
    # quantised: 16 distinct values -> massive ties
    q = (torch.rand(B, N, generator=g) * 16).floor() / 16
    idx, val = ops.topk(q.to(DEV), k, with_values=True)
    idx, val = idx.cpu(), val.cpu()
    assert torch.equal(val, torch.topk(q, k, dim=-1).values)
    assert torch.equal(q.gather(1, idx), val)
    for b in range(B):
        for v in val[b].unique():
            sel = idx[b][val[b] == v]
            assert torch.all(sel[1:] > sel[:-1])                             # ascending indices inside a tie group
            allv = (q[b] == v).nonzero().flatten()
            if (val[b] == v).sum() < allv.numel():                           # partially taken group: the lowest indices
                assert torch.equal(sel, allv[:sel.numel()])
    assert torch.equal(ops.topk(q.to(DEV), k), ops.topk(q.to(DEV), k))       # bit-reproducible
