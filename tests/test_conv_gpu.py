"""GPU: the implicit-GEMM 3x3 convolution (csrc/conv3x3.hip, C ABI di_conv3x3_fwd) against torch's float64 convolution
on the same fp16 operands: the full shapes of the shared convolutions (6x256x112x200 -> 128, 512x180x180 -> 128), the
heat-map head pair (BatchNorm folded + ReLU, NCHW class logits), ragged tiles and borders.  fp32 accumulation, fp16
output: 1e-3 of the value scale."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from deepinteraction_amd import ops

DEV = 'cuda'


def _ref(x, w, b, bn=None, relu=False):
    y = F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    if bn is not None:
        y = F.batch_norm(y, bn.running_mean.double(), bn.running_var.double(), bn.weight.double(), bn.bias.double(), False, 0.0,
                         bn.eps)
    return y.relu() if relu else y


@pytest.mark.parametrize('shape', [(6, 256, 112, 200, 128), (1, 512, 180, 180, 128), (2, 128, 36, 36, 128),
                                   (1, 32, 5, 7, 128), (1, 64, 9, 17, 128), (3, 96, 16, 33, 128), (1, 128, 180, 180, 10),
                                   (13, 32, 16, 256, 128), (1, 64, 41, 50, 128),      # 16-row tiles / ragged 12-row tiles
                                   (2, 32, 6, 19, 4),
                                   (2, 256, 21, 50, 10), (1, 128, 8, 16, 16), (1, 384, 9, 33, 3)])   # the class-head kernel: chunk groups, ragged
def test_conv3x3_matches_torch(shape):
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    n, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(Cin + H)
    x = (torch.randn(n, Cin, H, W, generator=g) * 0.5).clamp_(min=0).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).half()
    b = (torch.randn(Cout, generator=g) * 0.1).half()
    dev = torch.device(DEV)
    if H * W * n > 200000:                       # the float64 reference of the big maps runs on the GPU (tolerance unchanged)
        ref = _ref(x.to(dev), w.to(dev), b.to(dev)).cpu()
    else:
        ref = _ref(x, w, b)
    packed = ops.pack_conv3x3(w.to(dev), b.to(dev))
    # both kernels for 128 output channels (weights through LDS / weights from L2), NHWC and NCHW for the class logits
    for nchw, staged in ([(False, True), (False, False)] if Cout == 128 else [(False, False), (True, False)]):
        got = ops.conv3x3(x.to(dev).contiguous(memory_format=torch.channels_last), *packed, out_nchw=nchw, use_staged=staged)
        assert got.shape == ref.shape
        if nchw:
            assert got.is_contiguous()
        err = (got.double().cpu() - ref).abs().max().item()
        assert err <= 1e-3 * max(1.0, ref.abs().max().item()), (nchw, staged, err, ref.abs().max().item())


def test_conv3x3_folds_batchnorm_and_relu():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 128, 20, 37, generator=g).half()
    conv = torch.nn.Conv2d(128, 128, 3, padding=1, bias=False)
    bn = torch.nn.BatchNorm2d(128).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.2, generator=g)
        bn.running_var.uniform_(0.5, 1.5, generator=g)
        bn.weight.normal_(1, 0.2, generator=g)
        bn.bias.normal_(0, 0.2, generator=g)
    conv, bn = conv.half(), bn.half()
    ref = _ref(x, conv.weight.detach(), None, bn, relu=True)
    packed = ops.pack_conv3x3(conv.weight.to(DEV), None, bn.to(DEV))
    got = ops.conv3x3(x.to(DEV).contiguous(memory_format=torch.channels_last), *packed, relu=True)
    err = (got.double().cpu() - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err          # the folded weights are re-rounded to fp16
    assert (got == 0).float().mean().item() > 0.2                        # the ReLU is active


def test_modules_use_the_hip_convolution():
    """The encoder's shared convolutions and the decoder's heat-map heads go through di_conv3x3_fwd in the fp16 inference
    form and agree with the library convolution of the same modules."""
    from deepinteraction_amd import harness, synth
    enc, dec = harness.build_models(synth.SHAPE_TINY, 40, torch.float16, DEV)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(6, 32, 16, 28, generator=g).half().to(DEV)
    conv = torch.nn.Conv2d(32, 128, 3, padding=1).half().to(DEV)
    with torch.no_grad():
        ops.PROFILE = []
        try:
            a = enc._shared_conv(conv, x)
            feat = torch.randn(2, 128, 36, 36, generator=g).half().to(DEV)
            h = dec._heatmap(dec.heatmap_head, feat)
            used = [p[0] for p in ops.PROFILE]
        finally:
            ops.PROFILE = None
        assert used.count('conv3x3_fwd') == 3, used
        b = conv(x.contiguous(memory_format=torch.channels_last))
        href = dec.heatmap_head(feat.contiguous(memory_format=torch.channels_last))
    assert (a.float() - b.float()).abs().max().item() <= 2e-3 * max(1.0, b.float().abs().max().item())
    assert h.shape == href.shape and h.is_contiguous()
    assert (h.float() - href.float()).abs().max().item() <= 4e-3 * max(1.0, href.float().abs().max().item())
