"""GPU: the float32 token-level kernels of the MMPI decoder (csrc/token32.hip: di_token_program / di_token_wide /
di_token_splitk / di_dynconv_fwd / di_roi_select / di_query_init; csrc/cross_attn.hip: di_kv_project_fwd /
di_mha_decode_x_fwd) against the same arithmetic in float64, and the fused decoder forward
(deepinteraction_amd/decoder_fused.py) against the module-by-module float32 path of the same head.

Tolerance: float32 products and accumulation: 2e-5 of the value scale per kernel (the cross attention: 1e-4 - its
probabilities and values pass the matrix cores as fp16)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from deepinteraction_amd import decoder_fused, ops, precision, synth
from deepinteraction_amd.configs import decoder_cfg

DEV = 'cuda'


def _close(got, ref, tol=2e-5):
    got, ref = got.double().cpu(), ref.double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), (err, ref.abs().max().item())


def _ln(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w.double(), b.double(), eps)


def _gen(seed):
    g = torch.Generator().manual_seed(seed)
    return g, (lambda *s: torch.randn(*s, generator=g))


D = lambda t: None if t is None else t.to(DEV)
PK = lambda w: ops.pack_linear(w.to(DEV))              # weight of a program's linear step
SP = lambda w: ops.split_hi_lo(w.to(DEV))            # float32 weight -> (hi, lo) fp16 pair on the device
JOIN = lambda p: p[0].double().cpu() + p[1].double().cpu() / 2048.0


@pytest.mark.parametrize('B,Q', [(1, 1), (1, 37), (1, 200), (2, 200), (1, 400)])
@pytest.mark.parametrize('case', ['plain384', 'pos', 'relu256', 'gelu512', 'res_ln', 'concat', 'chain', 'keep'])
def test_program_linear_steps(B, Q, case):
    """LOAD / LINEAR / ROWOP / STORE steps in the combinations the decoder uses."""
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    cases = ['plain384', 'pos', 'relu256', 'gelu512', 'res_ln', 'concat', 'chain', 'keep']
    g, r = _gen(cases.index(case) * 1000 + Q + B)
    M = B * Q
    K, N = dict(plain384=(128, 384), pos=(128, 128), relu256=(128, 256), gelu512=(128, 512), res_ln=(512, 128),
                concat=(256, 384), chain=(128, 128), keep=(256, 128))[case]
    x, w, b = r(M, K), r(N, K) / math.sqrt(K), r(N) * 0.1
    xd = x.double()
    p = ops.TokenProgram()
    y_dev = torch.empty((M, N), dtype=torch.float32, device=DEV)
    if case == 'pos':
        pe = r(M, K)
        p.load(0, D(x), pos=D(pe))
        xd = xd + pe.double()
    elif case == 'concat':
        p.load(0, D(x[:, :128].contiguous())).load(0, D(x[:, 128:].contiguous()), col=128)
    else:
        p.load(0, D(x))
    y = xd @ w.double().t() + b.double()
    act = dict(relu256=1, gelu512=2).get(case, 0)
    if act == 1:
        y = y.relu()
    if act == 2:
        y = F.gelu(y)
    p.linear(0, 1, PK(w), D(b), act=act)
    if case in ('res_ln', 'keep'):
        res, lw, lb = r(M, 128), 1 + 0.2 * r(128), 0.1 * r(128)
        keep = (torch.rand(M, generator=g) > 0.3).to(torch.uint8) if case == 'keep' else None
        p.load(2, D(res)).rowop(1, 1, aux=2, ln=(D(lw), D(lb)), keep=D(keep))
        y = _ln(y + res.double(), lw, lb)
        if keep is not None:
            y = y * keep.double()[:, None]
    if case == 'chain':              # DynamicConv tail: LN -> relu -> + res -> LN -> FFN (512) -> + -> LN
        res, lw, lb, lw2, lb2 = r(M, 128), 1 + 0.2 * r(128), 0.1 * r(128), 1 + 0.2 * r(128), 0.1 * r(128)
        w1, b1, w2, b2 = r(512, 128) / 11, r(512) * 0.1, r(128, 512) / 22, r(128) * 0.1
        p.rowop(1, 1, ln=(D(lw), D(lb)), relu=True).load(2, D(res)).rowop(1, 1, aux=2, ln=(D(lw2), D(lb2)))
        p.linear(1, 0, PK(w1), D(b1), act=2).linear(0, 2, PK(w2), D(b2)).rowop(2, 1, aux=1, ln=(D(lw), D(lb)))
        z = _ln(_ln(y, lw, lb).relu() + res.double(), lw2, lb2)
        y = _ln(z + F.gelu(z @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double(), lw, lb)
    p.store(1, y_dev)
    p.run(B, Q)
    _close(y_dev, y)


@pytest.mark.parametrize('B,Q', [(1, 37), (2, 200)])
def test_program_roles(B, Q):
    """Roles: an N-split linear (each role a 128-column third, stored side by side) and an FFN whose hidden dimension is
    split over the roles (partial sums in a workspace, added by LOAD_PARTS of a second program)."""
    g, r = _gen(Q)
    M = B * Q
    x, wq, bq = r(M, 128), r(384, 128) / 11, r(384) * 0.1
    w1, b1, w2, b2 = r(512, 128) / 11, r(512) * 0.1, r(128, 512) / 22, r(128) * 0.1
    qkv = torch.empty((M, 384), dtype=torch.float32, device=DEV)
    ws = torch.empty((4, M, 128), dtype=torch.float32, device=DEV)
    out = torch.empty((M, 128), dtype=torch.float32, device=DEV)
    xd = D(x)
    p = ops.TokenProgram(roles=4)
    p.load(0, xd)
    p.linear(0, 1, PK(wq), D(bq), roles=(1, 3), n_per_role=128).store(1, qkv, roles=(1, 3), n=128, role_offset=128)
    p.linear(0, 1, PK(w1), D(b1), act=2, n_per_role=128).linear(1, 2, PK(w2), None, k_per_role=128)
    p.store(2, ws, n=128, role_offset=M * 128)
    p.run(B, Q)
    ops.TokenProgram().load_parts(0, ws, 4, M, D(b2)).store(0, out).run(B, Q)
    _close(qkv, x.double() @ wq.double().t() + bq.double())
    _close(out, F.gelu(x.double() @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double())


@pytest.mark.parametrize('B,Q,masked', [(1, 200, False), (2, 200, True), (1, 400, True), (2, 37, True), (1, 512, False)])
def test_program_attention(B, Q, masked):
    g, r = _gen(Q + B)
    E, H = 128, 8
    qkv = r(B * Q, 3 * E)
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
    out = torch.empty((B * Q, E), dtype=torch.float32, device=DEV)
    Qp = (Q + 15) // 16 * 16
    qk, vt = torch.empty((B * Q, 256), dtype=torch.float32, device=DEV), torch.zeros((B, 128, Qp), dtype=torch.float32, device=DEV)
    ops.TokenProgram().load(0, D(qkv)).store(0, qk).store_t(0, vt, col=256).run(B, Q)       # the producer's two stores
    assert torch.equal(qk.cpu(), qkv[:, :256])
    assert torch.equal(vt.cpu()[:, :, :Q], qkv[:, 256:].view(B, Q, 128).transpose(1, 2))
    p = ops.TokenProgram()
    p.attn(0, qk, vt, 0.25, None if member is None else D(member.view(-1)), None if view is None else D(view.view(-1)))
    p.store(0, out)
    p.run(B, Q)
    q, k, v = (t.double().view(B, Q, H, 16).transpose(1, 2) for t in qkv.split(E, dim=1))
    sc = (q @ k.transpose(-1, -2)) * 0.25
    sc = sc.masked_fill(~allowed[:, None], float('-inf'))
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * Q, E)
    _close(out, ref)


def _ref_params(w, b, x):
    """The generator's output in the reference layout: (M, 32768) float64."""
    return x.double() @ w.double().t() + b.double()


def _join_params(params):
    """(M, 65536) hi / lo fragments -> (M, 32768) float64 in the reference layout."""
    M = params.shape[0]
    f = params.double().cpu().view(M, 64, 2, 512)
    v = (f[:, :, 0] + f[:, :, 1] / 2048.0).reshape(M, 32768)                       # value order
    out = torch.empty_like(v)
    out[:, decoder_fused._dyn_value_order('cpu')] = v
    return out


def test_wide_generator():
    """K = 128 -> N = 32768 (DynamicConv.dynamic_layer), weight-stationary kernel, hi / lo fragment output."""
    g, r = _gen(5)
    x, w, b = r(200, 128), r(32768, 128) / 11, r(32768) * 0.1
    wp, bv = decoder_fused._dyn_layout(D(w), D(b))
    xs = ops.split_rows(D(x))
    _close(_join_params(ops.token_wide(xs, wp, bv)), _ref_params(w, b, x), 1e-6)
    _close(_join_params(ops.token_wide(xs[:37], wp, bv)), _ref_params(w, b, x[:37]), 1e-6)
    # the split store of a program writes the same rows
    y_hl = torch.empty((200, 256), dtype=torch.float16, device=DEV)
    ops.TokenProgram().load(0, D(x)).store_hl(0, y_hl).run(1, 200)
    assert torch.equal(y_hl, xs)


@pytest.mark.parametrize('M', [1, 37, 200, 400])
def test_splitk_and_parts(M):
    """DynamicConv out_layer: split-K partial sums + the LOAD_PARTS step that sums them."""
    g, r = _gen(M)
    x, w, b = r(M, 6272), r(128, 6272) / 80, r(128) * 0.1
    ws, ns = ops.token_splitk(ops.dense_to_f2p(D(x)), ops.pack_ksteps(D(w)))
    assert ns == 14
    out = torch.empty((M, 128), dtype=torch.float32, device=DEV)
    ops.TokenProgram().load_parts(0, ws, ns, M, D(b)).store(0, out).run(1, M)
    _close(out, x.double() @ w.double().t() + b.double(), 2e-6)


def test_dynconv_core():
    """relu(LN2(relu(LN1(roi @ p1)) @ p2)) from the generator's fragments."""
    g, r = _gen(2)
    R = 23
    roi, y = r(R, 49, 128), r(R, 128)
    w, b = r(32768, 128) / 120, r(32768) * 0.02
    n = lambda: (1 + 0.2 * r(128), 0.1 * r(128))
    n1, n2 = n(), n()
    wp, bv = decoder_fused._dyn_layout(D(w), D(b))
    params_dev = ops.token_wide(ops.split_rows(D(y)), wp, bv)
    got = ops.dynconv(ops.split_rows(D(roi).view(-1, 128)).view(R, 49, 256), params_dev, tuple(D(t) for t in n1), tuple(D(t) for t in n2))
    params = _join_params(params_dev)                                               # what the kernel multiplied by
    p1 = params[:, :16384].view(R, 128, 128)
    p2 = params[:, 16384:].view(R, 128, 128)
    f1 = _ln(torch.bmm(roi.double(), p1), *n1).relu()
    ref = _ln(torch.bmm(f1, p2), *n2).relu()
    _close(ops.f2p_to_dense(got).view(R, 49, 128), ref, 2e-5)


def test_query_init():
    g, r = _gen(9)
    B, Q, H, W, ncls = 2, 37, 12, 12, 10
    bev = r(B, 128, H, W).half().to(DEV).contiguous(memory_format=torch.channels_last)
    top = torch.randint(0, ncls * H * W, (B, Q), generator=g)
    ce_w, ce_b = r(128, ncls), r(128)
    pe = (r(128, 2), r(128), r(128, 128) / 11, r(128))
    feat, pemb, pos, labels = ops.query_init(bev, D(top), D(ce_w), D(ce_b), tuple(D(t) for t in pe))
    cls, cell = top // (H * W), top % (H * W)
    tok = bev.float().cpu().permute(0, 2, 3, 1).reshape(B, H * W, 128).double()
    want = tok.gather(1, cell[:, :, None].expand(-1, -1, 128)) + ce_w.double().t()[cls] + ce_b.double()
    _close(feat.view(B, Q, 128), want)
    xy = torch.stack([(cell % W).double() + 0.5, (cell // W).double() + 0.5], -1)
    wpe = (xy @ pe[0].double().t() + pe[1].double()).relu() @ pe[2].double().t() + pe[3].double()
    _close(pemb.view(B, Q, 128), wpe)
    assert torch.equal(pos.cpu().double(), xy) and torch.equal(labels.cpu(), cls)


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


@pytest.mark.parametrize('B,Q,S', [(1, 200, 32400), (2, 37, 1000), (1, 400, 5000)])
def test_cross_attention_x(B, Q, S):
    """kv_project (float32 weight split in two fp16 halves, positional bias) + mha_decode_x (float32-accurate logits of
    magnitude ~300) + the COMBINE step, against float64 on the same fp16 map."""
    g, r = _gen(S + Q)
    E, H = 128, 8
    x = (r(B, S, E) * 2).half()
    w = r(2 * E, E) * 0.4
    kb, vb = r(S, E) * 8, r(S, E)
    q = r(B, Q, E) * 6
    kx = ops.kv_project(D(x), ops.pack_kv_weight(D(w)), D(kb), D(vb))
    K = x.double() @ w[:E].double().t() + kb.double()
    Vv = x.double() @ w[E:].double().t() + vb.double()
    got_k = kx[..., :E].double().cpu() + kx[..., E:2 * E].double().cpu() / 2048.0
    _close(got_k, K, 2e-6)
    _close(kx[..., 2 * E:], Vv, 1e-3)
    scratch, nr = ops.mha_decode_x(D(q), kx, 0.25, merge=False)
    out = torch.empty((B * Q, E), dtype=torch.float32, device=DEV)
    ops.TokenProgram().combine(0, scratch, nr).store(0, out).run(B, Q)
    assert torch.allclose(ops.mha_decode_x(D(q), kx, 0.25), out, rtol=1e-5, atol=1e-6)      # the merge as its own launch
    qh = q.double().view(B, Q, H, 16).transpose(1, 2)
    kh = K.view(B, S, H, 16).transpose(1, 2)
    vh = kx[..., 2 * E:].double().cpu().view(B, S, H, 16).transpose(1, 2)          # the fp16 V the kernel read
    sc = (qh @ kh.transpose(-1, -2)) * 0.25
    assert sc.abs().max().item() > 100                                               # the regime that needs float32 logits
    ref = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(B * Q, E)
    _close(out, ref, 1e-3)


def test_roi_align_float32_out():
    g, r = _gen(4)
    feat = r(3, 128, 20, 30).half().to(DEV).contiguous(memory_format=torch.channels_last)
    rois = torch.tensor([[0, 1.5, 2.0, 9.0, 12.0], [2, -3.0, -1.0, 40.0, 25.0], [1, 5.0, 5.0, 5.5, 5.2]])
    a = ops.roi_align(feat, D(rois), 0.5)
    b = ops.roi_align(feat, D(rois), 0.5, out_f32=True)
    assert b.dtype == torch.float32 and torch.equal(b.half(), a)
    c = ops.roi_align(feat, D(rois), 0.5, out_hl=True)                     # the split form carries the float32 values
    assert c.shape == (3, 49, 256) and torch.equal(c, ops.split_rows(b.view(-1, 128)).view(3, 49, 256))


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
    return dec.to(DEV).eval()


@pytest.mark.parametrize('B,Q', [(1, 40), (2, 24), (1, 200)])
def test_fused_decoder_matches_module_path(B, Q):
    """The fused forward (fp16 maps, float32 token path) against the module-by-module path of the same head in float32 on
    the same fp16-representable maps: identical INT outputs (proposals, labels, on-the-image masks), continuous outputs
    within 1e-3 (the module path's heat-map heads are fp16 too, so the proposals are the same set)."""
    import copy
    torch.backends.cudnn.deterministic = True
    dec32 = _decoder(Q)
    dec = copy.deepcopy(dec32)
    dec.heatmap_head.half(), dec.heatmap_head_img.half()            # precision.half_maps_ for the head
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
    assert min(agree) == 1.0, agree
    assert set(fused) == set(ref)
    for k in ref:
        a, b = fused[k].float(), ref[k].float()
        assert a.shape == b.shape, k
        d = (a - b).abs()
        scale = max(1.0, b.abs().max().item())
        print(k, 'max', d.max().item() / scale, 'median', d.median().item() / scale)
        assert d.median().item() <= 1e-4 * scale and d.max().item() <= 1e-2 * scale, (k, d.max().item(), scale)
