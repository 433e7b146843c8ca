"""GPU: the token-level kernels of the MMPI decoder (csrc/token.hip, C ABI di_token_* / di_dynconv_fwd / di_roi_select /
di_query_init / di_pred_heads) against the same arithmetic in float64 on the same fp16 operands, and the fused decoder
forward (deepinteraction_amd/decoder_fused.py) against the module-by-module product path it replaces.

Tolerance: fp16 storage of every stage output (2^-11 relative), float32 accumulation: 2e-3 of the value scale per
kernel (a LayerNorm output is O(1), so this is absolute there)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from deepinteraction_amd import decoder_fused, ops, synth
from deepinteraction_amd.configs import decoder_cfg

DEV = 'cuda'


def _close(got, ref, tol=2e-3):
    got, ref = got.double().cpu(), ref.double().cpu()
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), (err, ref.abs().max().item())


def _ln(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w.double(), b.double(), eps)


@pytest.mark.parametrize('M', [1, 37, 200, 400])
@pytest.mark.parametrize('case', ['plain384', 'pos', 'relu256', 'gelu512', 'res_ln', 'concat', 'two_ln', 'keep'])
def test_token_linear(M, case):
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    cases = ['plain384', 'pos', 'relu256', 'gelu512', 'res_ln', 'concat', 'two_ln', 'keep']
    g = torch.Generator().manual_seed(cases.index(case) * 1000 + M)
    r = lambda *s: torch.randn(*s, generator=g)
    K, N = dict(plain384=(128, 384), pos=(128, 128), relu256=(128, 256), gelu512=(128, 512), res_ln=(512, 128),
                concat=(256, 384), two_ln=(6272, 128), keep=(256, 128))[case]
    x, w, b = r(M, K).half(), (r(N, K) / math.sqrt(K)).half(), r(N) * 0.1
    d = lambda t: None if t is None else t.to(DEV)
    kw, ref = {}, None
    xd = x.double()
    if case == 'pos':
        p = r(M, K).half()
        kw['pos'] = d(p)
        xd = (x + p).double()                               # the add is an fp16 add, as in the module path
    if case == 'concat':
        kw['x2'] = d(x[:, 128:].contiguous())
        x_dev = d(x[:, :128].contiguous())
    else:
        x_dev = d(x)
    y = xd @ w.double().t() + b.double()
    if case == 'relu256':
        kw['act1'], y = 1, y.relu()
    if case == 'gelu512':
        kw['act1'], y = 2, F.gelu(y)
    if case in ('res_ln', 'keep'):
        res, lw, lb = r(M, 128).half(), (1 + 0.2 * r(128)).half(), (0.1 * r(128)).half()
        kw.update(res1=d(res), ln1=(d(lw), d(lb)))
        y = _ln(y + res.double(), lw, lb)
    if case == 'keep':
        keep = (torch.rand(M, generator=g) > 0.3).to(torch.uint8)
        kw['keep'] = d(keep)
        y = y * keep.double()[:, None]
    if case == 'two_ln':                                      # DynamicConv out_layer: LN3 -> relu -> +res -> LN2
        res, lw, lb = r(M, 128).half(), (1 + 0.2 * r(128)).half(), (0.1 * r(128)).half()
        lw2, lb2 = (1 + 0.2 * r(128)).half(), (0.1 * r(128)).half()
        kw.update(ln1=(d(lw), d(lb)), act2=True, res2=d(res), ln2=(d(lw2), d(lb2)))
        y = _ln(_ln(y, lw, lb).relu() + res.double(), lw2, lb2)
    got = ops.token_linear(x_dev, d(w), d(b), **kw)
    assert got.shape == (M, N)
    _close(got, y)


def test_token_linear_wide_generator():
    """K = 128 -> N = 32768 (DynamicConv.dynamic_layer), weight-stationary kernel."""
    g = torch.Generator().manual_seed(5)
    x, w, b = torch.randn(200, 128, generator=g).half(), (torch.randn(32768, 128, generator=g) / 11).half(), \
        torch.randn(32768, generator=g) * 0.1
    got = ops.token_linear(x.to(DEV), w.to(DEV), b.to(DEV))
    _close(got, x.double() @ w.double().t() + b.double())


@pytest.mark.parametrize('B,Q,masked', [(1, 200, False), (2, 200, True), (1, 400, True), (2, 37, True), (1, 512, False)])
def test_token_mha(B, Q, masked):
    g = torch.Generator().manual_seed(Q + B)
    E, H = 128, 8
    qkv = torch.randn(B * Q, 3 * E, generator=g).half()
    member = view = None
    allowed = torch.ones(B, Q, Q, dtype=torch.bool)
    if masked:
        member = torch.randint(0, 64, (B, Q), generator=g).to(torch.uint8)
        view = torch.randint(-1, 6, (B, Q), generator=g).to(torch.int8)
        for b in range(B):
            for q in range(Q):
                v = int(view[b, q])
                if v >= 0:
                    member[b, q] |= (1 << v)                                  # a query is a member of its own view
        bits = (member.long()[:, None, :] >> view.long().clamp(min=0)[:, :, None]) & 1
        allowed = (bits == 1) | (view.long()[:, :, None] < 0)
    got = ops.token_mha(qkv.to(DEV), B, Q, H, 0.25, None if member is None else member.view(-1).to(DEV),
                        None if view is None else view.view(-1).to(DEV))
    q, k, v = (t.double().view(B, Q, H, 16).transpose(1, 2) for t in qkv.split(E, dim=1))
    sc = (q @ k.transpose(-1, -2)) * 0.25
    sc = sc.masked_fill(~allowed[:, None], float('-inf'))
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * Q, E)
    _close(got, ref)


def test_dynconv_core():
    """relu(LN2(relu(LN1(roi @ p1)) @ p2)) with the parameters in the kernel's permuted layout (decoder_fused._dyn_perm)."""
    g = torch.Generator().manual_seed(2)
    R = 23
    roi = torch.randn(R, 49, 128, generator=g).half()
    params = (torch.randn(R, 32768, generator=g) / 11).half()                       # reference layout
    n = lambda: ((1 + 0.2 * torch.randn(128, generator=g)).half(), (0.1 * torch.randn(128, generator=g)).half())
    n1, n2 = n(), n()
    perm = decoder_fused._dyn_perm('cpu')
    got = ops.dynconv(roi.to(DEV), params[:, perm].contiguous().to(DEV), tuple(t.to(DEV) for t in n1),
                      tuple(t.to(DEV) for t in n2))
    p1 = params[:, :16384].double().view(R, 128, 128)
    p2 = params[:, 16384:].double().view(R, 128, 128)
    f1 = _ln(torch.bmm(roi.double(), p1), *n1).relu().half().double()               # the kernel keeps F1 in fp16
    ref = _ln(torch.bmm(f1, p2), *n2).relu()
    _close(got, ref, 4e-3)


@pytest.mark.parametrize('B,V,Q', [(1, 6, 200), (2, 6, 37), (1, 3, 400)])
def test_roi_select(B, V, Q):
    """INT bookkeeping of the image RoI block, bit-exact against the torch formulation it replaces."""
    g = torch.Generator().manual_seed(Q)
    on = (torch.rand(B, V, Q, generator=g) < 0.25).int()
    on[0, V - 1] = 0
    on[0, V - 1, 3] = 1                                                             # a view with ONE centre: skipped
    rect = torch.randn(B, V, Q, 4, generator=g)
    rois, view, member, keep, on_img = ops.roi_select(rect.to(DEV), on.to(DEV))
    sel = on.bool() & (on.sum(-1, keepdim=True) > 1)
    vid = torch.arange(V).view(1, V, 1)
    last = torch.where(sel, vid, torch.full_like(vid, -1)).max(1).values
    lastc = last.clamp(min=0)
    assert torch.equal(view.cpu().view(B, Q).long(), last)
    assert torch.equal(keep.cpu().view(B, Q).bool(), last >= 0)
    assert torch.equal(on_img.cpu(), last.float())
    bits = (sel.long() << vid).sum(1)
    assert torch.equal(member.cpu().view(B, Q).long(), bits)
    want = torch.cat([(torch.arange(B).view(B, 1) * V + lastc).float().unsqueeze(-1),
                      rect.gather(1, lastc.view(B, 1, Q, 1).expand(B, 1, Q, 4)).squeeze(1)], -1).view(-1, 5)
    assert torch.equal(rois.cpu(), want)
    rb = torch.randn(B, Q, 4, generator=g)
    want = torch.cat([torch.arange(B).float().view(B, 1, 1).expand(B, Q, 1), rb], -1).view(-1, 5)
    assert torch.equal(ops.roi_select(rb.to(DEV)).cpu(), want)


def _decoder(Q, seed=7):
    from deepinteraction_amd import harness
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionDecoder
    torch.manual_seed(seed)
    dec = DeepInteractionDecoder(**decoder_cfg(bev=36, num_proposals=Q))
    harness.randomize_bn([dec])
    g = torch.Generator().manual_seed(3)
    for n, p in dec.named_parameters():                         # biases / LayerNorm affine off their trivial init
        if p.dim() == 1:
            p.data.add_(torch.randn(p.shape, generator=g) * 0.05)
    return dec.to(DEV).half().eval()


@pytest.mark.parametrize('B,Q', [(1, 40), (2, 24), (1, 200)])
def test_fused_decoder_matches_module_path(B, Q):
    """The fused forward against the module-by-module path of the same head (same weights, fp16): identical INT
    outputs (proposals, labels, on-the-image masks), continuous outputs within fp16 noise of ~40 stacked stages."""
    torch.backends.cudnn.deterministic = True
    dec = _decoder(Q)
    shape = synth.SHAPE_TINY
    Hi, Wi = shape['img_hw']
    g = torch.Generator().manual_seed(0)
    cl = lambda t: t.half().to(DEV).contiguous(memory_format=torch.channels_last)
    p0, p1 = cl(torch.randn(B, 128, 36, 36, generator=g)), cl(torch.randn(B, 128, 36, 36, generator=g))
    img = cl(torch.randn(6 * B, 128, Hi, Wi, generator=g))
    metas = synth.make_inputs(B, shape, seed=0)['img_metas']
    with torch.no_grad():
        assert decoder_fused.usable(dec, p0, img)
        fused = dec([p0, p1], img, metas)[0][0]
        f_labels, f_masks, f_top = dec.query_labels.clone(), [m.clone() for m in dec.on_the_image_mask], dec.top_proposals.clone()
        dec.fused = False
        ref = dec([p0, p1], img, metas)[0][0]
    assert torch.equal(f_top, dec.top_proposals) and torch.equal(f_labels, dec.query_labels)
    agree = [float((a == b).float().mean()) for a, b in zip(f_masks, dec.on_the_image_mask)]
    assert min(agree) >= 0.97, agree                              # a centre within fp16 noise of an image border may flip
    assert set(fused) == set(ref)
    for k in ref:
        a, b = fused[k].float(), ref[k].float()
        assert a.shape == b.shape, k
        d = (a - b).abs()
        scale = max(1.0, b.abs().max().item())
        assert d.median().item() <= 3e-3 * scale, (k, d.median().item())
        assert (d > 3e-2 * scale).float().mean().item() <= 0.1, (k, (d > 3e-2 * scale).float().mean().item())
