"""Oracle pinning for the local-window attention ops (CPU).

* the C restatement (oracle/locatt_c.c) must equal, BIT FOR BIT, the reference's own
  kernels.cuh compiled for the host (oracle/_ref/liblocatt_ref.so) - when that build exists;
* the vectorised torch form must agree with the C form to float32 round-off;
* known-answer tests from first principles (the reference ships none).
"""
import math

import pytest
import torch

from oracle.locatt import CLocatt, TorchLocatt, local_attention

CASES = [
    ('similar_forward', lambda a, b, w, kh, kw: (a, b, kh, kw)),
    ('weighting_forward', lambda a, b, w, kh, kw: (a, w, kh, kw)),
    ('similar_backward', lambda a, b, w, kh, kw: (a, w, kh, kw, True)),
    ('similar_backward', lambda a, b, w, kh, kw: (a, w, kh, kw, False)),
    ('weighting_backward_ori', lambda a, b, w, kh, kw: (w, a, kh, kw)),
    ('weighting_backward_weight', lambda a, b, w, kh, kw: (a, b, kh, kw)),
]


def _inputs(B, C, H, W, kh, kw, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g),
            torch.randn(B, H, W, kh * kw, generator=g))


@pytest.mark.parametrize('shape', [(2, 16, 11, 13, 9, 9), (1, 8, 5, 4, 3, 5), (1, 24, 9, 20, 7, 7)])
def test_port_equals_compiled_reference_bitwise(oracle_libs, shape):
    if not CLocatt.available('reference'):
        pytest.skip('oracle/_ref not built (no /root/reference on this box)')
    B, C, H, W, kh, kw = shape
    a, b, w = _inputs(B, C, H, W, kh, kw)
    port, ref = CLocatt('port'), CLocatt('reference')
    for name, mk in CASES:
        args = mk(a, b, w, kh, kw)
        assert torch.equal(getattr(port, name)(*args), getattr(ref, name)(*args)), name


@pytest.mark.parametrize('shape', [(2, 16, 11, 13, 9, 9), (1, 8, 5, 4, 3, 5)])
def test_torch_form_matches_port(oracle_libs, shape):
    B, C, H, W, kh, kw = shape
    a, b, w = _inputs(B, C, H, W, kh, kw, seed=1)
    port = CLocatt('port')
    for name, mk in CASES:
        args = mk(a, b, w, kh, kw)
        x, y = getattr(port, name)(*args), getattr(TorchLocatt, name)(*args)
        assert torch.allclose(x, y, rtol=1e-5, atol=2e-5), name


def test_known_answer_constant_input_border_leak():
    """q = k = const: every in-image slot scores C*c^2, out-of-image slots score 0 and still
    take softmax mass (kernels.cuh:28-39).  Interior pixel -> exactly uniform 1/81."""
    C, H, W, c = 4, 12, 12, 0.5
    x = torch.full((1, C, H, W), c)
    w = TorchLocatt.similar_forward(x, x, 9, 9)
    p = torch.softmax(w / math.sqrt(C), -1)
    assert torch.allclose(p[0, 6, 6], torch.full((81,), 1 / 81.0), atol=1e-7)
    # corner pixel: 25 in-image slots with score s, 56 OOB slots with score 0
    s = C * c * c / math.sqrt(C)
    z = 25 * math.exp(s) + 56
    inside = p[0, 0, 0].view(9, 9)[4:, 4:]
    assert torch.allclose(inside, torch.full((5, 5), math.exp(s) / z), atol=1e-6)
    assert torch.allclose(p[0, 0, 0].sum(), torch.tensor(1.0), atol=1e-6)
    # weighting: OOB slots contribute value 0 -> output = (25 e^s / z) * c at the corner
    out = TorchLocatt.weighting_forward(x, p, 9, 9)
    assert abs(out[0, 0, 0, 0].item() - 25 * math.exp(s) / z * c) < 1e-6


def test_known_answer_slot_order():
    """slot k <-> (dy,dx) = (k//kW - rH, k%kW - rW): an impulse in k at (y+1, x-2) lights slot
    (1+1)*5 + (-2+2) = 10 of a 3x5 window."""
    q = torch.ones(1, 1, 7, 9)
    k = torch.zeros(1, 1, 7, 9)
    k[0, 0, 4, 3] = 2.0
    w = TorchLocatt.similar_forward(q, k, 3, 5)
    assert w[0, 3, 5].nonzero().flatten().tolist() == [10]
    assert w[0, 3, 5, 10].item() == 2.0


def test_gradients_are_the_adjoint():
    """The backward entry points are the exact adjoints of the forward ones (autograd check
    on the torch form, float64)."""
    g = torch.Generator().manual_seed(3)
    a = torch.randn(1, 3, 6, 7, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(1, 3, 6, 7, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(1, 6, 7, 15, generator=g, dtype=torch.float64, requires_grad=True)
    go_w = torch.randn(1, 6, 7, 15, generator=g, dtype=torch.float64)
    go_c = torch.randn(1, 3, 6, 7, generator=g, dtype=torch.float64)
    s = TorchLocatt.similar_forward(a, b, 3, 5)
    ga, gb = torch.autograd.grad(s, (a, b), go_w)
    assert torch.allclose(ga, TorchLocatt.similar_backward(b.detach(), go_w, 3, 5, True))
    assert torch.allclose(gb, TorchLocatt.similar_backward(a.detach(), go_w, 3, 5, False))
    o = TorchLocatt.weighting_forward(a, w, 3, 5)
    ga, gw = torch.autograd.grad(o, (a, w), go_c)
    assert torch.allclose(ga, TorchLocatt.weighting_backward_ori(w.detach(), go_c, 3, 5))
    assert torch.allclose(gw, TorchLocatt.weighting_backward_weight(a.detach(), go_c, 3, 5))


def test_fused_definition():
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(1, 8, 6, 6, generator=g).relu() for _ in range(3))
    out = local_attention(q, k, v, 9, 9)
    # brute force for one pixel
    y, x = 1, 4
    sc, vals = [], []
    for kk in range(81):
        yy, xx = y + kk // 9 - 4, x + kk % 9 - 4
        inb = 0 <= yy < 6 and 0 <= xx < 6
        sc.append((q[0, :, y, x] * k[0, :, yy, xx]).sum() if inb else torch.tensor(0.0))
        vals.append(v[0, :, yy, xx] if inb else torch.zeros(8))
    p = torch.softmax(torch.stack(sc) / math.sqrt(8), 0)
    ref = (p[:, None] * torch.stack(vals)).sum(0)
    assert torch.allclose(out[0, :, y, x], ref, atol=1e-6)
