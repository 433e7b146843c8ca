"""GPU: the fused mixed-precision window attention of the training step (csrc/local_attn_train.hip: forward with the
log-sum-exp kept, backward that recomputes the soft-max on the matrix cores) against
  * a plain float32 torch formulation of the reference chain  similar -> softmax(. * scale) -> weighting  with autograd
    gradients (zero-padded keys take part in the soft-max, reference encoder_utils.py:132-134, kernels.cuh:4-80), on small maps
    with ragged tiles and borders everywhere, and
  * the unfused float32 HIP operators (`similarFunction` / `weightingFunction`, the path of float32 training) at the benched
    image-side and BEV-side shapes.
Tolerance: fp16 operands, fp16 probabilities / dS in the second MFMA, fp16 results - 3e-3 of the tensor's largest magnitude
(measured 0.6-1.2e-3); the inference kernel's own bound is 2e-3."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from deepinteraction_amd import ops
from deepinteraction_amd.mmdet3d_plugin.models.utils.encoder_utils import (fusedWindowAttention, similarFunction,
                                                                            weightingFunction)


def _maps(n, H, W, seed, gscale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    mk = lambda s: (torch.randn(n, 128, H, W, generator=g) * s).half().cuda().contiguous(memory_format=torch.channels_last)
    return mk(0.5), mk(0.5), mk(1.0), mk(gscale)


def _torch_reference(q, k, v, go, scale):
    q, k, v = (t.float().detach().requires_grad_(True) for t in (q, k, v))
    n, C, H, W = q.shape
    kw = F.unfold(k, 9, padding=4).view(n, C, 81, H * W)
    vw = F.unfold(v, 9, padding=4).view(n, C, 81, H * W)
    logits = (q.reshape(n, C, 1, H * W) * kw).sum(1) * scale              # zero-padded keys: logit 0, in the soft-max
    p = logits.softmax(1)
    out = (p.unsqueeze(1) * vw).sum(2).view(n, C, H, W)
    out.backward(go.float())
    lse = torch.logsumexp(logits, 1).view(n, H, W) / math.log(2.0)
    return out.detach(), lse.detach(), q.grad, k.grad, v.grad


def _rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-20))


@pytest.mark.parametrize('shape', [(2, 8, 8), (1, 5, 7), (2, 21, 30), (1, 16, 40), (3, 9, 17)])
@pytest.mark.parametrize('gscale', [1.0, 1e-3])
def test_fused_training_attention_matches_autograd_of_the_plain_formulation(shape, gscale):
    n, H, W = shape
    q, k, v, go = _maps(n, H, W, seed=H * 100 + W, gscale=gscale)
    scale = 1.0 / math.sqrt(128)
    ro, rl, rq, rk, rv = _torch_reference(q, k, v, go, scale)
    out, lse = ops.local_attention_train_fwd(q, k, v, scale)
    assert _rel(out, ro) <= 2e-3, _rel(out, ro)
    assert float((lse - rl).abs().max()) <= 2e-3
    inf = ops.local_attention(q, k, v, 9, 9, scale)                       # the inference kernel: same arithmetic
    assert torch.equal(out, inf)
    gq, gk, gv = ops.local_attention_train_bwd(q, k, v, out, go, lse, scale)
    for name, got, ref in (('q', gq, rq), ('k', gk, rk), ('v', gv, rv)):
        assert torch.isfinite(got.float()).all(), name
        assert _rel(got, ref) <= 3e-3, (name, _rel(got, ref))


@pytest.mark.parametrize('shape', [(6, 112, 200), (1, 180, 180)])
def test_fused_training_attention_against_the_unfused_float32_operators_at_the_benched_shapes(shape):
    n, H, W = shape
    q, k, v, go = _maps(n, H, W, seed=7, gscale=1e-2)
    scale = 1.0 / math.sqrt(128)
    qq, kk, vv = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    w = similarFunction.apply(qq, kk, 9, 9)
    o = weightingFunction.apply(vv, F.softmax(w * scale, -1), 9, 9)
    o.backward(go)
    qf, kf, vf = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    with torch.autocast('cuda', dtype=torch.float16):
        of = fusedWindowAttention.apply(qf, kf, vf, scale)
    of.backward(go)
    assert _rel(of, o) <= 2e-3
    for name, got, ref in (('q', qf.grad, qq.grad), ('k', kf.grad, kk.grad), ('v', vf.grad, vv.grad)):
        assert got.dtype == torch.float16 and _rel(got, ref) <= 3e-3, (name, _rel(got, ref))
    # deterministic: no atomics anywhere
    of2, l2 = ops.local_attention_train_fwd(q, k, v, scale)
    g2 = ops.local_attention_train_bwd(q, k, v, of2, go, l2, scale)
    assert torch.equal(g2[0], qf.grad) and torch.equal(g2[1], kf.grad) and torch.equal(g2[2], vf.grad)
