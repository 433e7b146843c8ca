"""csrc/wgrad.hip: the float32 weight / bias gradient of the training step's 1x1 convolutions (dW = gy^T x over 10^5 pixels)
on the float32 matrix cores, against a float64 formulation; through the C ABI (ctypes)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from deepinteraction_amd import ops
from deepinteraction_amd.autograd import PixelLinear


@pytest.mark.parametrize('P,Cin,Cout', [(134400, 128, 128), (32400, 128, 128), (32400, 256, 128), (4100, 128, 256), (33, 128, 128),
                                        (64801, 512, 128)])
def test_wgrad_matches_float64(P, Cin, Cout):
    g = torch.Generator(device='cuda').manual_seed(P + Cin)
    x = torch.randn(P, Cin, device='cuda', generator=g)
    gy = torch.randn(P, Cout, device='cuda', generator=g) * (torch.rand(P, 1, device='cuda', generator=g) < 0.7)   # ReLU-like zero rows
    gw, gb = ops.wgrad(x, gy, bias=True)
    ref_w = gy.double().t() @ x.double()
    ref_b = gy.double().sum(0)
    lib_w = gy.t() @ x                                    # the library's float32 GEMM: the error a float32 reduction has
    scale = ref_w.abs().max().item()
    err, err_lib = (gw.double() - ref_w).abs().max().item() / scale, (lib_w.double() - ref_w).abs().max().item() / scale
    assert err <= max(2e-6, 2 * err_lib), (err, err_lib)
    assert (gb.double() - ref_b).abs().max().item() <= 2e-6 * max(ref_b.abs().max().item(), 1.0) * max(1.0, P ** 0.5 / 50)
    gw2, gb2 = ops.wgrad(x, gy, bias=True)                # fixed summation order: bit-reproducible
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    gw3, none = ops.wgrad(x, gy)
    assert none is None and torch.equal(gw3, gw)


def test_wgrad_rejects_other_shapes():
    x, gy = torch.zeros(100, 96, device='cuda'), torch.zeros(100, 128, device='cuda')
    assert not ops.wgrad_supported(x, gy)
    assert not ops.wgrad_supported(x.half(), gy.half())
    with pytest.raises(AssertionError):
        ops.wgrad(x, gy)


def test_pixel_linear_backward_uses_it_and_matches_autograd():
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(32400, 128, device='cuda', generator=g, requires_grad=True)
    w = (torch.randn(128, 128, device='cuda', generator=g) * 0.1).requires_grad_()
    b = torch.randn(128, device='cuda', generator=g).requires_grad_()
    gy = torch.randn(32400, 128, device='cuda', generator=g)
    PixelLinear.apply(x, w, b).backward(gy)
    got = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    torch.nn.functional.linear(x.double(), w.double(), b.double()).backward(gy.double())
    for a, r in zip(got, (x, w, b)):
        assert (a.double() - r.grad.double()).abs().max().item() <= 3e-6 * r.grad.abs().max().item()
