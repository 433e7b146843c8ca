"""CPU tests of host-side logic added in round 2 (no GPU, no HIP library calls)."""
import math

import torch

from deepinteraction_amd import ops
from deepinteraction_amd.autograd import PixelLinear


def test_pixel_linear_matches_autograd_of_linear():
    """`PixelLinear` (1x1 convolution as a GEMM with a slab-batched weight gradient) has the gradients of F.linear."""
    torch.manual_seed(0)
    for M in (2 * 3 * 7 * 512, 1000, 64 * 2100):        # slabs 128..2, no slab split, the image-map pixel count
        x = torch.randn(M, 16, dtype=torch.float64, requires_grad=True)
        w = torch.randn(8, 16, dtype=torch.float64, requires_grad=True)
        b = torch.randn(8, dtype=torch.float64, requires_grad=True)
        g = torch.randn(M, 8, dtype=torch.float64)
        ref = torch.autograd.grad(torch.nn.functional.linear(x, w, b), (x, w, b), g)
        got = torch.autograd.grad(PixelLinear.apply(x, w, b), (x, w, b), g)
        for r, o in zip(ref, got):
            assert torch.allclose(r, o, rtol=1e-12, atol=1e-9)
        S = PixelLinear._slabs(M)
        assert M % S == 0 and (S == 1 or M // S >= 512)


def test_bev_sector_order_is_an_azimuth_sorted_permutation():
    Hb, Wb = 180, 180
    order = ops.bev_sector_order(Hb, Wb, 'cpu').long()
    assert order.numel() == Hb * Wb and torch.equal(torch.sort(order).values, torch.arange(Hb * Wb))
    y = (order // Wb).double() - (Hb - 1) / 2
    x = (order % Wb).double() - (Wb - 1) / 2
    wedge = torch.floor((torch.atan2(y, x) + math.pi) / (2 * math.pi) * 512).clamp(max=511)
    assert bool((wedge[1:] >= wedge[:-1]).all())                      # one sweep around the ego vehicle
    # an eighth of the list (what one XCD walks) spans an eighth of the circle
    q = order.numel() // 8
    assert float(wedge[q - 1] - wedge[0]) <= 512 / 8 + 2


def test_chain_and_ffn_images_have_the_kernel_layout_size():
    w1, b1 = torch.randn(512, 128), torch.randn(512)
    w2, b2 = torch.randn(128, 512), torch.randn(128)
    imgs = ops.ffn_images(w1, b1, w2, b2)
    assert len(imgs) == 4 and all(t.dtype == torch.uint8 and t.numel() == 2 * 128 * 128 * 2 + 1024 for t in imgs)
    # b2 rides in chunk 0 only
    tail = lambda t: t[2 * 128 * 128 * 2 + 512:].view(torch.float32)
    assert float(tail(imgs[0]).abs().sum()) > 0 and all(float(tail(t).abs().sum()) == 0 for t in imgs[1:])
