"""Tensor-level entry points of the HIP hot path.

Thin plumbing: torch tensors in (device memory owned by PyTorch's allocator, the
current HIP stream), raw pointers into `libdeepinteraction_hip.so` through its C ABI.
Feature maps are logical (n,C,H,W) tensors in `torch.channels_last` memory format, i.e.
physically (n,H,W,C) - the layout the kernels are written for.

No CPU implementation lives here; CPU tensors are rejected.
"""
import ctypes

import collections
import math
import os

import torch

from . import _lib

_DT = {torch.float32: _lib.DI_F32, torch.float16: _lib.DI_F16}

# When a list, every profiled launch appends (kernel name, n, start_event, end_event): HIP events
# recorded on the launch stream right around the kernel (bench.py's live roofline measurement).
PROFILE = None


class _DispatchTimer:
    """Start 'event' of a profiled launch: `elapsed_time(end)` (ms) is the kernel's OWN duration when the launch bound a pair of
    events to its dispatch (`di_timed_begin`: the window-attention and deformable-attention kernels - what rocprofv3 reports),
    else the interval between two events recorded on the stream around it; `stream_ms(end)` is always the latter."""

    def __init__(self):
        self.start = torch.cuda.Event(enable_timing=True)
        self.handles = None
        self.us = None

    def stream_ms(self, end):
        return self.start.elapsed_time(end)

    def dispatch_bound(self):
        return self.handles is not None

    def elapsed_time(self, end):
        if self.handles is None:
            return self.start.elapsed_time(end)
        if self.us is None:
            out = ctypes.c_float(0.0)
            _lib.call('di_timed_elapsed_us', self.handles[0], self.handles[1], 1, ctypes.byref(out))
            self.us = float(out.value)
        return self.us * 1e-3


def _profiled(name, n, launch):
    if PROFILE is None:
        return launch()
    t, e = _DispatchTimer(), torch.cuda.Event(enable_timing=True)
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.call('di_timed_begin', ctypes.byref(a), ctypes.byref(b))
    t.start.record()
    launch()
    e.record()
    if _lib.lib().di_timed_consumed():
        t.handles = (a, b)
    else:
        out = ctypes.c_float(0.0)
        _lib.call('di_timed_elapsed_us', a, b, 0, ctypes.byref(out))          # never recorded: just destroy them
    PROFILE.append((name, n, t, e))


def _code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f'HIP hot path supports float32/float16 feature maps, got {t.dtype}')


def _dev(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _lib.HipLibraryError('HIP hot path called with a CPU tensor - there is no CPU fallback')


def _stream():
    return torch.cuda.current_stream().cuda_stream


def cl(t):
    """(n,C,H,W) tensor physically channels-last (no copy when it already is)."""
    return t.contiguous(memory_format=torch.channels_last)


def _is_cl(t):
    return t.is_contiguous(memory_format=torch.channels_last)


def empty_cl(n, C, H, W, like):
    return torch.empty((n, C, H, W), dtype=like.dtype, device=like.device, memory_format=torch.channels_last)


def zeros_cl(n, C, H, W, like):
    return empty_cl(n, C, H, W, like).zero_()


def _f32(t):
    return t.contiguous().float() if t.dtype != torch.float32 or not t.is_contiguous() else t


# ------------------------------------------------------------------ local-window attention
LA_AUTO, LA_VALU = 0, 1
LA_MFMA = 3           # + configuration 0..4 of the persistent pipelined row-pair kernel (1 = 8x8 tiles)
LA_RING = 24          # + 0 / 1: the ring generation (one workgroup per CU, LDS-DMA rows, flag-synchronised), 16x8 / 8x16 tiles; + 2..8: measurement variants (6 / 7 / 8: the ragged last round cut into sub-tiles - measured, slower)


def local_attention(q, k, v, kH, kW, scale, variant=LA_AUTO):
    """softmax_k(<q[p], k[p+off_k]> * scale) . v[p+off_k], fused (no (n,H,W,81) tensor)."""
    _dev(q, k, v)
    q, k, v = cl(q), cl(k), cl(v)
    n, C, H, W = q.shape
    out = empty_cl(n, C, H, W, q)
    _profiled('local_attn_fwd', n, lambda: _lib.call(
        'di_local_attn_fwd_ex', q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), n, H, W, C, kH, kW,
        float(scale), _code(q), variant, _stream()))
    return out


def ring_timeouts():
    """Bounded flag spins of the ring window-attention kernel that gave up since the library was loaded (synchronises the
    current stream).  0 on a healthy device; the affected query tiles hold NaN (csrc/local_attn_ring.hip)."""
    n = int(_lib.lib().di_local_attn_ring_timeouts(_stream()))
    if n < 0:
        raise _lib.HipLibraryError('di_local_attn_ring_timeouts failed: ' + _lib.lib().di_last_error().decode())
    return n


_RING_SEEN = [0]        # the counter value the last health check (or GraphedHotPath replay check) has already reported


def check_ring_health(seen=None):
    """Raise when a window-attention launch gave up on a flag spin SINCE THE LAST CHECK (its output tiles are NaN-poisoned).
    The device counter is process-global and monotonic: a fault is reported once - the baseline then moves up, so that later
    launches / replays whose outputs are fine do not keep raising for an old transient.  `seen`: a caller's own baseline
    (a one-element list, updated in place); default = the process-wide one."""
    seen = _RING_SEEN if seen is None else seen
    n = ring_timeouts()
    if n > seen[0]:
        new, seen[0] = n - seen[0], n
        _RING_SEEN[0] = max(_RING_SEEN[0], n)
        raise _lib.HipLibraryError(f'local_attn_ring: {new} bounded spins gave up since the last check ({n} since the library '
                                   'was loaded) - the affected output tiles hold NaN; the device (or another process sharing '
                                   'it) stalled a workgroup for > 0.1 s')


def local_attention_train_usable(q, k, v, kH, kW):
    """The fused training kernels take fp16 maps of 128 channels and 9 x 9 windows (the configuration of every
    `LocalContextAttentionBlock` of Fusion_0075_refactor under autocast)."""
    return (q.is_cuda and q.dtype == torch.float16 and k.dtype == torch.float16 and v.dtype == torch.float16 and
            q.shape[1] == 128 and kH == 9 and kW == 9 and q.shape == k.shape == v.shape)


def local_attention_train_fwd(q, k, v, scale):
    """(out, lse): the fused window attention with the per-query log2-sum-exp2 kept for the backward."""
    _dev(q, k, v)
    q, k, v = cl(q), cl(k), cl(v)
    n, C, H, W = q.shape
    out = empty_cl(n, C, H, W, q)
    lse = torch.empty((n, H, W), dtype=torch.float32, device=q.device)
    _profiled('local_attn_train_fwd', n, lambda: _lib.call(
        'di_local_attn_train_fwd', q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), n, H, W,
        float(scale), _stream()))
    return out, lse


def local_attention_train_bwd(q, k, v, out, grad_out, lse, scale):
    """(grad_q, grad_k, grad_v) of the fused window attention; the soft-max is recomputed from `lse`."""
    _dev(q, k, v, out, grad_out, lse)
    q, k, v, out, grad_out = cl(q), cl(k), cl(v), cl(out), cl(grad_out.to(q.dtype))
    n, C, H, W = q.shape
    gq, gk, gv = (empty_cl(n, C, H, W, q) for _ in range(3))
    dsum = torch.empty((n, H, W), dtype=torch.float32, device=q.device)
    _profiled('local_attn_train_bwd', n, lambda: _lib.call(
        'di_local_attn_train_bwd', q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), grad_out.data_ptr(),
        lse.data_ptr(), dsum.data_ptr(), gq.data_ptr(), gk.data_ptr(), gv.data_ptr(), n, H, W, float(scale), _stream()))
    return gq, gk, gv


def local_attention_kernel_name():
    """What LA_AUTO launches for fp16 / C=128 / 9x9 maps (bench.py's roofline line names it)."""
    return ('local_attn_ring_kernel: one workgroup per CU, 16x8 tiles, 144 KB ring of 128-byte halo rows by LDS-DMA, '
            'flag-synchronised producer / consumer wavefronts (small maps: local_attn_m2_kernel, 8x8 tiles)')


def similar_forward(x_ori, x_loc, kH, kW):
    _dev(x_ori, x_loc)
    x_ori, x_loc = cl(x_ori), cl(x_loc)
    n, C, H, W = x_ori.shape
    out = torch.empty((n, H, W, kH * kW), dtype=torch.float32, device=x_ori.device)
    _profiled('locatt_similar_fwd', n, lambda: _lib.call('di_locatt_similar_fwd', x_ori.data_ptr(), x_loc.data_ptr(), out.data_ptr(), n, H, W, C, kH,
              kW, _code(x_ori), _stream()))
    return out


def similar_backward(x, grad_out, kH, kW, is_ori):
    _dev(x, grad_out)
    x, grad_out = cl(x), _f32(grad_out)
    n, C, H, W = x.shape
    out = empty_cl(n, C, H, W, x)
    _profiled('locatt_similar_bwd', n, lambda: _lib.call('di_locatt_similar_bwd', x.data_ptr(), grad_out.data_ptr(), out.data_ptr(), n, H, W, C, kH,
              kW, int(bool(is_ori)), _code(x), _stream()))
    return out


def weighting_forward(x_ori, x_weight, kH, kW):
    _dev(x_ori, x_weight)
    x_ori, x_weight = cl(x_ori), _f32(x_weight)
    n, C, H, W = x_ori.shape
    out = empty_cl(n, C, H, W, x_ori)
    _profiled('locatt_weighting_fwd', n, lambda: _lib.call('di_locatt_weighting_fwd', x_ori.data_ptr(), x_weight.data_ptr(), out.data_ptr(), n, H, W, C,
              kH, kW, _code(x_ori), _stream()))
    return out


def weighting_backward_ori(x_weight, grad_out, kH, kW):
    _dev(x_weight, grad_out)
    x_weight, grad_out = _f32(x_weight), cl(grad_out)
    n, C, H, W = grad_out.shape
    out = empty_cl(n, C, H, W, grad_out)
    _profiled('locatt_weighting_bwd_ori', n, lambda: _lib.call('di_locatt_weighting_bwd_ori', x_weight.data_ptr(), grad_out.data_ptr(), out.data_ptr(), n, H,
              W, C, kH, kW, _code(grad_out), _stream()))
    return out


def weighting_backward_weight(x_ori, grad_out, kH, kW):
    _dev(x_ori, grad_out)
    x_ori = cl(x_ori)
    grad_out = cl(grad_out.to(x_ori.dtype))
    n, C, H, W = x_ori.shape
    out = torch.empty((n, H, W, kH * kW), dtype=torch.float32, device=x_ori.device)
    _profiled('locatt_weighting_bwd_weight', n, lambda: _lib.call('di_locatt_weighting_bwd_weight', x_ori.data_ptr(), grad_out.data_ptr(), out.data_ptr(), n, H,
              W, C, kH, kW, _code(x_ori), _stream()))
    return out


# ------------------------------------------------------------------ fused 1x1-convolution chains
def pointwise_chain(x1, w1, b1, relu1, x2=None, w2=None, b2=None, relu2=False, x3=None, mask=None, bm=None):
    """y = act2(W2 . [act1(W1 . [x1 ; x2] + b1 + mask * bm) ; x3] + b2) over the pixels of fp16 channels-last maps
    (C = 128); the second link and the masked bias (mask (n,1,H,W) fp16, bm (128) float32) are optional.
    w* fp16 (128, 128 or 256), b* float32 (128)."""
    _dev(x1, w1, b1)
    x1 = cl(x1)
    n, C, H, W = x1.shape
    assert C == 128 and x1.dtype == torch.float16, 'fused chains are fp16, 128 channels'
    k1 = 128 if x2 is None else 256
    k2 = 0 if w2 is None else (128 if x3 is None else 256)
    assert w1.shape == (128, k1) and w1.dtype == torch.float16 and w1.is_contiguous()
    assert b1.dtype == torch.float32 and b1.numel() == 128
    ptr = lambda t: 0 if t is None else t.data_ptr()
    if x2 is not None:
        x2 = cl(x2)
        assert x2.shape == x1.shape and x2.dtype == x1.dtype
    if x3 is not None:
        x3 = cl(x3)
        assert x3.shape == x1.shape and x3.dtype == x1.dtype
    if w2 is not None:
        assert w2.shape == (128, k2) and w2.dtype == torch.float16 and w2.is_contiguous()
        assert b2.dtype == torch.float32 and b2.numel() == 128
    if mask is not None:
        assert mask.dtype == torch.float16 and mask.numel() == n * H * W and mask.is_contiguous()
        assert bm.dtype == torch.float32 and bm.numel() == 128
    y = empty_cl(n, 128, H, W, x1)
    _profiled('pointwise_chain', n * (k1 + k2), lambda: _lib.call(
        'di_pointwise_chain_masked_fwd', x1.data_ptr(), ptr(x2), ptr(x3), w1.data_ptr(), b1.data_ptr(), ptr(w2),
        ptr(b2), y.data_ptr(), n * H * W, k1, k2, int(bool(relu1)), int(bool(relu2)), ptr(mask), ptr(bm),
        _stream()))
    return y


def chain_image(w1, b1, w2=None, b2=None):
    """The 66 560-byte LDS image of a chain that `pointwise_multi` moves by LDS-DMA: W1 | W2 as 128 rows x 256 B with the
    16-B chunks XOR-swizzled by the row (conflict-free ds_read_b128 fragments), W2's columns re-ordered for the
    register-resident hidden operand (k = 8g + 4t + r <-> channel 32kk + 16t + 4g + r), the rows of the last link
    re-ordered for 16-B output stores, then b1 | b2 (float32)."""
    dev = w1.device
    # output rows of the LAST link: image row 16nb + 4g + r holds channel 32(nb/2) + 8g + 4(nb%2) + r, so that a lane's
    # fragment pair is 8 consecutive channels (one 16-B store)
    rr = torch.arange(128, device=dev)
    nb, gq, rq = rr // 16, (rr % 16) // 4, rr % 4
    last = 32 * (nb // 2) + 8 * gq + 4 * (nb % 2) + rq
    if w2 is None:
        w1, b1 = w1[last], b1[last]
    else:
        w2, b2 = w2[last], b2[last]

    def img(w, perm):
        w = w.detach().to(torch.float16).contiguous()
        if perm:
            w = w.view(128, 4, 2, 4, 4).permute(0, 1, 3, 2, 4).contiguous()          # [r][kk][g][t][4]
        W = w.view(128, 16, 8)
        r = torch.arange(128, device=dev).view(128, 1)
        c = torch.arange(16, device=dev).view(1, 16)
        out = torch.empty_like(W)
        out[r.expand(128, 16), c ^ (r & 15)] = W
        return out.view(-1).view(torch.uint8)
    zero_w = torch.zeros(128 * 128 * 2, dtype=torch.uint8, device=dev)
    zero_b = torch.zeros(128, dtype=torch.float32, device=dev)
    parts = [img(w1, False), img(w2, True) if w2 is not None else zero_w,
             b1.detach().float().contiguous().view(torch.uint8),
             (b2.detach().float().contiguous() if b2 is not None else zero_b).view(torch.uint8)]
    return torch.cat(parts).contiguous()


def ffn_images(w1, b1, w2, b2):
    """Chunk images of `ffn_ln` from the two Linear layers of an FFN (w1 (hidden,128), w2 (128,hidden), hidden = 128 k):
    chunk c = chain_image(W1[128c:128c+128], b1[128c:..], W2[:, 128c:128c+128], b2 for c == 0 else 0)."""
    hidden = w1.shape[0]
    assert w1.shape[1] == 128 and w2.shape == (128, hidden) and hidden % 128 == 0 and hidden // 128 <= 8
    zero = torch.zeros(128, dtype=torch.float32, device=w1.device)
    return [chain_image(w1[128 * c:128 * c + 128].float(), b1[128 * c:128 * c + 128].float(),
                        w2[:, 128 * c:128 * c + 128].float(), b2.float() if c == 0 else zero)
            for c in range(hidden // 128)]


def ffn_ln(x, images, ln_w, ln_b, eps=1e-5):
    """LayerNorm(x + W2 . relu(W1 . x + b1) + b2) over (M,128) fp16 tokens in one launch (`images` = ffn_images(...))."""
    _dev(x)
    assert x.dtype == torch.float16 and x.dim() == 2 and x.shape[1] == 128 and x.is_contiguous()
    assert ln_w.dtype == torch.float16 and ln_b.dtype == torch.float16 and ln_w.numel() == 128
    y = torch.empty_like(x)
    arr = (ctypes.c_void_p * len(images))(*[t.data_ptr() for t in images])
    _profiled('ffn_ln_fwd', x.shape[0], lambda: _lib.call(
        'di_ffn_ln_fwd', x.data_ptr(), len(images), ctypes.addressof(arr), None, ln_w.data_ptr(), ln_b.data_ptr(),
        float(eps), y.data_ptr(), x.shape[0], _stream()))
    return y


def linear_ln(x, image, residual, ln_w, ln_b, eps=1e-5, with_sum=False):
    """LayerNorm(residual + x @ W^T + b) over (M,128) fp16 tokens in one launch; image = chain_image(W, b) (128 x 128).
    with_sum: also return the un-normalised sum (fp16) -> (y, sum)."""
    _dev(x, residual)
    assert x.dtype == torch.float16 and x.dim() == 2 and x.shape[1] == 128 and x.is_contiguous()
    assert residual.shape == x.shape and residual.dtype == x.dtype and residual.is_contiguous()
    assert ln_w.dtype == torch.float16 and ln_b.dtype == torch.float16 and ln_w.numel() == 128
    y = torch.empty_like(x)
    pre = torch.empty_like(x) if with_sum else None
    arr = (ctypes.c_void_p * 1)(image.data_ptr())
    _lib.call('di_ffn_ln_fwd_ex', x.data_ptr(), 1, ctypes.addressof(arr), residual.data_ptr(), ln_w.data_ptr(),
              ln_b.data_ptr(), float(eps), y.data_ptr(), 0 if pre is None else pre.data_ptr(), x.shape[0], _stream())
    return (y, pre) if with_sum else y


def token_linear(x, w, b, relu=False):
    """y = act(x @ w^T + b) over (M, 128) fp16 tokens with the fused chain kernel (csrc/pointwise.hip) instead of a library
    GEMM: w (128, 128) fp16 contiguous, b (128) float32.  A channels-last map IS its token matrix, so this is
    `pointwise_chain` without the 4-D view."""
    _dev(x, w, b)
    assert x.dtype == torch.float16 and x.dim() == 2 and x.shape[1] == 128 and x.is_contiguous()
    assert w.shape == (128, 128) and w.dtype == torch.float16 and w.is_contiguous() and b.dtype == torch.float32 and b.numel() == 128
    y = torch.empty_like(x)
    _lib.call('di_pointwise_chain_masked_fwd', x.data_ptr(), 0, 0, w.data_ptr(), b.data_ptr(), 0, 0, y.data_ptr(), x.shape[0],
              128, 0, int(bool(relu)), 0, 0, 0, _stream())
    return y


def token_linear_hm(x, w, b, tokens_per_map):
    """`token_linear` whose output is written HEAD-MAJOR: x (bs * T, 128) fp16 tokens -> (bs, 8, T, 16), the value layout
    `ms_deform_attn(..., head_major=True)` gathers from (two corners of a footprint row = 64 contiguous bytes)."""
    _dev(x, w, b)
    assert x.dtype == torch.float16 and x.dim() == 2 and x.shape[1] == 128 and x.is_contiguous()
    assert w.shape == (128, 128) and w.dtype == torch.float16 and w.is_contiguous() and b.dtype == torch.float32 and b.numel() == 128
    T = int(tokens_per_map)
    assert T > 0 and x.shape[0] % T == 0
    y = torch.empty((x.shape[0] // T, 8, T, 16), dtype=x.dtype, device=x.device)
    _lib.call('di_pointwise_chain_hm_fwd', x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), x.shape[0], T, 0, _stream())
    return y


def token_linear_multi(x, images):
    """Several 128-output projections of the SAME (M, 128) fp16 tokens in one launch that reads x once (`pointwise_multi`):
    images = [chain_image(w_k, b_k)] with w_k (128, 128) - pad narrower projections with zero rows.  Returns one (M, 128)
    tensor per image."""
    _dev(x)
    assert x.dtype == torch.float16 and x.dim() == 2 and x.shape[1] == 128 and x.is_contiguous() and 1 <= len(images) <= 4
    nc = len(images)
    ys = [torch.empty_like(x) for _ in images]
    P, I = ctypes.c_void_p * nc, ctypes.c_int * nc
    for im in images:
        assert im.dtype == torch.uint8 and im.numel() == 2 * 128 * 128 * 2 + 1024 and im.is_cuda
    a_im, a_y, zero = P(*[im.data_ptr() for im in images]), P(*[y.data_ptr() for y in ys]), I(*([0] * nc))
    _lib.call('di_pointwise_multi_fwd', x.data_ptr(), nc, ctypes.addressof(a_im), ctypes.addressof(a_y), ctypes.addressof(zero),
              ctypes.addressof(zero), ctypes.addressof(zero), x.shape[0], _stream())
    return ys


def _check_outs(outs, nc, n, H, W, like):
    assert len(outs) == nc
    for y in outs:
        assert y.shape == (n, 128, H, W) and y.dtype == like.dtype and y.device == like.device and _is_cl(y), \
            'caller-provided outputs must be channels-last maps of the result shape'
    return list(outs)


def pointwise_multi(x, chains, outs=None):
    """Several chains over one fp16 channels-last map (C = 128) in ONE launch, x read once.  chains: list of
    (image, relu1, relu2, two_links) with image = chain_image(...).  Returns one map per chain.  outs: write into these
    (e.g. halves of a pair buffer that ONE window-attention launch then reads) instead of allocating."""
    _dev(x)
    x = cl(x)
    n, C, H, W = x.shape
    assert C == 128 and x.dtype == torch.float16 and 1 <= len(chains) <= 4
    nc = len(chains)
    ys = [empty_cl(n, 128, H, W, x) for _ in chains] if outs is None else _check_outs(outs, nc, n, H, W, x)
    P, I = ctypes.c_void_p * nc, ctypes.c_int * nc
    for (im, r1, r2, two) in chains:
        assert im.dtype == torch.uint8 and im.numel() == 2 * 128 * 128 * 2 + 1024 and im.is_cuda
    a_im, a_y = P(*[c[0].data_ptr() for c in chains]), P(*[y.data_ptr() for y in ys])
    a_r1, a_r2 = I(*[int(bool(c[1])) for c in chains]), I(*[int(bool(c[2])) for c in chains])
    a_two = I(*[int(bool(c[3])) for c in chains])
    _profiled('pointwise_multi', (n, nc), lambda: _lib.call(
        'di_pointwise_multi_fwd', x.data_ptr(), nc, ctypes.addressof(a_im), ctypes.addressof(a_y), ctypes.addressof(a_r1),
        ctypes.addressof(a_r2), ctypes.addressof(a_two), n * H * W, _stream()))
    return ys


# ------------------------------------------------------------------ image -> BEV
I2PKeys = collections.namedtuple('I2PKeys', 'table T V bev_hw dense dense_order', defaults=(None, None))


I2P_DENSE = int(os.environ.get('DI_I2P_DENSE', '1'))      # 0: never build / use the dense key stream (A/B measurements)


def i2p_key_table(pillars, coors, num_points, proj, aug_rev, ori_hw, img_hw, bev_hw, dense=True, sector_order=True):
    """Geometry pass of the pillar attention for ONE sample: the per-cell key table (valid-key count, pillar id and the
    compacted sampling coordinates of every (point, camera) slot that lands in an image).  Depends on the points and the
    metas only - build it once per sample and hand it to every `i2p_attention` call (all encoder layers).
    dense: also pack the keys into the dense stream the matrix-core attention pass reads (`di_i2p_compact_keys`, two small
    launches; groups of 8 cells of the walk order - `sector_order` must then match the `i2p_attention` calls)."""
    _dev(pillars, coors, num_points, proj, aug_rev)
    P, T, D = pillars.shape
    V = proj.shape[0]
    assert pillars.dtype == torch.float32 and pillars.is_contiguous()
    assert coors.dtype == torch.int32 and coors.is_contiguous() and coors.shape[1] == 4
    assert num_points.dtype == torch.int32 and num_points.is_contiguous()
    assert proj.dtype == torch.float32 and proj.is_contiguous() and aug_rev.dtype == torch.float32
    (Hi, Wi), (Hb, Wb) = img_hw, bev_hw
    table = torch.empty(int(_lib.lib().di_i2p_key_table_bytes(Hb, Wb, T, V)), dtype=torch.uint8, device=pillars.device)
    _lib.call('di_i2p_build_keys', pillars.data_ptr(), coors.data_ptr(), num_points.data_ptr(), proj.data_ptr(),
              aug_rev.data_ptr(), table.data_ptr(), P, T, D, V, Hi, Wi, Hb, Wb, float(ori_hw[0]), float(ori_hw[1]),
              _stream())
    if not (dense and I2P_DENSE):
        return I2PKeys(table, T, V, (Hb, Wb))
    order = bev_sector_order(Hb, Wb, pillars.device) if sector_order else None
    dtab = torch.empty(int(_lib.lib().di_i2p_dense_bytes(Hb, Wb, T, V, P)), dtype=torch.uint8, device=pillars.device)
    _lib.call('di_i2p_compact_keys', table.data_ptr(), None if order is None else order.data_ptr(), dtab.data_ptr(), T, V, Wi, Hb,
              Wb, _stream())
    return I2PKeys(table, T, V, (Hb, Wb), dtab, bool(sector_order))


_CELL_ORDER = {}


def bev_sector_order(Hb, Wb, device):
    """The BEV cells sorted by azimuth around the map centre (the ego vehicle), then by radius: the walk order of the
    pillar attention (one eighth of it per XCD).  A constant of the map shape, cached on the device."""
    key = (Hb, Wb, str(device))
    hit = _CELL_ORDER.get(key)
    if hit is None:
        y, x = torch.meshgrid(torch.arange(Hb, dtype=torch.float64) - (Hb - 1) / 2,
                              torch.arange(Wb, dtype=torch.float64) - (Wb - 1) / 2, indexing='ij')
        ang = torch.atan2(y, x).reshape(-1)
        wedge = torch.floor((ang + math.pi) / (2 * math.pi) * 512).clamp_(max=511)          # 512 wedges of 0.7 degrees
        rad = torch.hypot(y, x).reshape(-1)
        hit = torch.argsort(wedge * 1e6 + rad, stable=True).to(torch.int32).to(device)
        _CELL_ORDER[key] = hit
    return hit


def copy_bytes(dst, src):
    """dst <- src for two dense uint8 device buffers of the same size, in ONE streaming launch (`GraphedHotPath.load`)."""
    assert dst.is_cuda and src.is_cuda and dst.dtype == torch.uint8 and src.dtype == torch.uint8
    assert dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel()
    _lib.call('di_copy_d2d', dst.data_ptr(), src.data_ptr(), dst.numel(), _stream())


def iou3d_lidar(b1, b2):
    """(N, >=7), (M, >=7) float32 LiDAR boxes on the device -> (N, M) 3-D IoU, ONE launch (`BboxOverlaps3D` of the
    Hungarian assigner; csrc/iou3d.hip)."""
    _dev(b1, b2)
    b1, b2 = b1.detach().float().contiguous(), b2.detach().float().contiguous()
    out = torch.empty((b1.shape[0], b2.shape[0]), dtype=torch.float32, device=b1.device)
    _lib.call('di_iou3d_lidar', b1.data_ptr(), b1.shape[0], b1.shape[1], b2.data_ptr(), b2.shape[0], b2.shape[1],
              out.data_ptr(), _stream())
    return out


def bn_workspace(C, device):
    """Scratch of one training BatchNorm (partial sums + coefficients); one per module, not shared between launches in flight."""
    return torch.empty(int(_lib.lib().di_bn_workspace_floats(int(C))), dtype=torch.float32, device=device)


def bn_train_fwd(x, weight, bias, running_mean, running_var, num_batches, eps, momentum, relu, workspace):
    """Training BatchNorm2d (+ ReLU) of a channels-last map (csrc/batchnorm.hip): (y, saved[4C] = mean | rstd | gamma | beta);
    the running statistics are updated in place."""
    _dev(x)
    assert _is_cl(x), 'the map must be channels-last'
    n, C, H, W = x.shape
    y = torch.empty_like(x)
    saved = torch.empty(4 * C, dtype=torch.float32, device=x.device)
    ptr = lambda t: 0 if t is None else t.data_ptr()
    for t in (weight, bias, running_mean, running_var):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    assert num_batches is None or num_batches.dtype == torch.int64
    _lib.call('di_bn_train_fwd', x.data_ptr(), n * H * W, C, _code(x), ptr(weight), ptr(bias), float(eps), float(momentum),
              ptr(running_mean), ptr(running_var), ptr(num_batches), int(bool(relu)), y.data_ptr(), saved.data_ptr(),
              workspace.data_ptr(), _stream())
    return y, saved


def wgrad_supported(x, grad_y):
    """True when `wgrad` takes this pair: dense float32 (pixels, C) matrices on the device, channel counts multiples of 128."""
    return (x.is_cuda and x.dtype == torch.float32 and grad_y.dtype == torch.float32 and x.dim() == 2 and grad_y.dim() == 2
            and x.shape[0] == grad_y.shape[0] and x.shape[0] > 0 and x.shape[1] % 128 == 0 and grad_y.shape[1] % 128 == 0
            and x.is_contiguous() and grad_y.is_contiguous())


def wgrad(x, grad_y, bias=False):
    """(grad_w (Cout, Cin), grad_b (Cout,) or None) of y = x @ W^T + b for x (pixels, Cin), grad_y (pixels, Cout), float32
    (csrc/wgrad.hip: float32 matrix cores, pixel slabs summed in a fixed order)."""
    _dev(x, grad_y)
    assert wgrad_supported(x, grad_y), 'wgrad: dense float32 (pixels, C) operands with C % 128 == 0'
    P, Cin = x.shape
    Cout = grad_y.shape[1]
    nws = int(_lib.lib().di_wgrad_workspace_floats(P, Cin, Cout))
    if nws < 0:
        raise _lib.HipLibraryError(_lib.lib().di_last_error().decode())
    ws = torch.empty(nws, dtype=torch.float32, device=x.device)
    gw = torch.empty(Cout, Cin, dtype=torch.float32, device=x.device)
    gb = torch.empty(Cout, dtype=torch.float32, device=x.device) if bias else None
    _lib.call('di_wgrad_f32', x.data_ptr(), grad_y.data_ptr(), P, Cin, Cout, gw.data_ptr(), 0 if gb is None else gb.data_ptr(),
              ws.data_ptr(), _stream())
    return gw, gb


def bn_train_bwd(x, grad_y, saved, relu, workspace, affine):
    """(grad_x, grad_gamma, grad_beta) of `bn_train_fwd`; the ReLU mask is recomputed from x."""
    _dev(x, grad_y, saved)
    grad_y = cl(grad_y.to(x.dtype))
    n, C, H, W = x.shape
    gx = torch.empty_like(x)
    gg = torch.empty(C, dtype=torch.float32, device=x.device) if affine else None
    gb = torch.empty(C, dtype=torch.float32, device=x.device) if affine else None
    _lib.call('di_bn_train_bwd', x.data_ptr(), grad_y.data_ptr(), n * H * W, C, _code(x), saved.data_ptr(), int(bool(relu)),
              gx.data_ptr(), 0 if gg is None else gg.data_ptr(), 0 if gb is None else gb.data_ptr(), workspace.data_ptr(),
              _stream())
    return gx, gg, gb


_I2P_SEED_REF = [None]


def set_i2p_seed_tensor(t):
    """Register (or clear with None) the device int64 word that every pillar-attention launch adds to its dropout seed when
    it runs - what a graph-replayed training step rewrites before each replay (`train_step.GraphedTrainer`)."""
    if t is not None:
        assert t.is_cuda and t.dtype == torch.int64 and t.numel() == 1
    _I2P_SEED_REF[0] = t
    _lib.call('di_i2p_set_seed_ptr', 0 if t is None else t.data_ptr())


def i2p_attention(img, qfold, pillars, coors, num_points, proj, aug_rev, ori_hw, dropout_p=0.0, seed=0, keys=None,
                  sector_order=True, with_mass=False):
    """One sample.  img (V,C,Hi,Wi), qfold (1,C,Hb,Wb) channels-last; pillars (P,T,D) f32,
    coors (P,4) i32, num_points (P,) i32, proj (V,4,4) f32, aug_rev (12,) f32; `keys`: the sample's `i2p_key_table`
    (built here when not given).  Returns ctx (1,C,Hb,Wb) and valid (1,1,Hb,Wb) (same dtype as img), every cell written;
    with_mass: also the kept probability mass (1,1,Hb,Wb) of every cell under attention dropout."""
    _dev(img, qfold)
    img, qfold = cl(img), cl(qfold)
    V, C, Hi, Wi = img.shape
    _, _, Hb, Wb = qfold.shape
    if keys is None:
        keys = i2p_key_table(pillars, coors, num_points, proj, aug_rev, ori_hw, (Hi, Wi), (Hb, Wb))
    T = keys.T
    assert keys.V == V and keys.bev_hw == (Hb, Wb), 'key table of another geometry'
    ctx = empty_cl(1, C, Hb, Wb, img)          # (allocating NCHW and converting was a 16 us transposing copy of garbage)
    valid = torch.empty((1, 1, Hb, Wb), dtype=img.dtype, device=img.device)
    order = bev_sector_order(Hb, Wb, img.device).data_ptr() if sector_order else None
    if (keys.dense is not None and keys.dense_order == bool(sector_order) and not with_mass and dropout_p == 0.0
            and img.dtype == torch.float16 and C == 128):
        # the matrix-core pass (csrc/i2p_dense.hip): fp16 maps, 128 channels, no attention dropout
        _profiled('i2p_attn_fwd', V, lambda: _lib.call(
            'di_i2p_attn_dense_fwd', img.data_ptr(), qfold.data_ptr(), keys.table.data_ptr(), keys.dense.data_ptr(), order,
            ctx.data_ptr(), valid.data_ptr(), V, Hi, Wi, Hb, Wb, _stream()))
        return ctx, valid
    if with_mass:
        mass = torch.empty_like(valid)
        _lib.call('di_i2p_attn_fwd_mass', img.data_ptr(), qfold.data_ptr(), keys.table.data_ptr(), order, ctx.data_ptr(),
                  valid.data_ptr(), mass.data_ptr(), T, V, Hi, Wi, Hb, Wb, C, float(dropout_p), int(seed), _code(img),
                  _stream())
        return ctx, valid, mass
    _lib.call('di_i2p_attn_fwd', img.data_ptr(), qfold.data_ptr(), keys.table.data_ptr(), order, ctx.data_ptr(),
              valid.data_ptr(), T, V, Hi, Wi, Hb, Wb, C, float(dropout_p), int(seed), _code(img), _stream())
    return ctx, valid


def i2p_attention_bwd(img, qfold, grad_ctx, pillars, coors, num_points, proj, aug_rev, ori_hw, dropout_p=0.0,
                      seed=0, grad_mass=None):
    """Gradients of i2p_attention w.r.t. img (V,C,Hi,Wi) and qfold (1,C,Hb,Wb): float32, channels-last.  grad_mass
    (1,1,Hb,Wb): the gradient of the kept mass (`with_mass` of the forward)."""
    _dev(img, qfold, grad_ctx)
    img, qfold = cl(img), cl(qfold)
    grad_ctx = cl(grad_ctx.to(img.dtype))
    V, C, Hi, Wi = img.shape
    _, _, Hb, Wb = qfold.shape
    P, T, D = pillars.shape
    g_img = torch.empty((V, C, Hi, Wi), dtype=torch.float32, device=img.device, memory_format=torch.channels_last).zero_()
    g_q = torch.empty((1, C, Hb, Wb), dtype=torch.float32, device=img.device, memory_format=torch.channels_last).zero_()
    gm = None if grad_mass is None else grad_mass.to(img.dtype).contiguous()
    _lib.call('di_i2p_attn_bwd_mass', img.data_ptr(), qfold.data_ptr(), grad_ctx.data_ptr(),
              None if gm is None else gm.data_ptr(), pillars.data_ptr(),
              coors.data_ptr(), num_points.data_ptr(), proj.data_ptr(), aug_rev.data_ptr(), g_img.data_ptr(),
              g_q.data_ptr(), P, T, D, V, Hi, Wi, Hb, Wb, C, float(ori_hw[0]), float(ori_hw[1]), float(dropout_p),
              int(seed), _code(img), _stream())
    return g_img, g_q


# ------------------------------------------------------------------ BEV -> image
def depth_scatter(pts, proj, aug_rev, Hi, Wi, ori_hw):
    """pts (Np,>=3) f32 -> sparse depth (V,Hi,Wi) f32; duplicates: highest point index wins."""
    _dev(pts, proj, aug_rev)
    assert pts.dtype == torch.float32 and pts.dim() == 2 and pts.stride(1) == 1
    V = proj.shape[0]
    packed = torch.zeros((V, Hi, Wi), dtype=torch.int64, device=pts.device)
    depth = torch.empty((V, Hi, Wi), dtype=torch.float32, device=pts.device)
    _lib.call('di_depth_scatter', pts.data_ptr(), pts.shape[0], pts.stride(0), proj.data_ptr(),
              aug_rev.data_ptr(), packed.data_ptr(), depth.data_ptr(), V, Hi, Wi, float(ori_hw[0]),
              float(ori_hw[1]), _stream())
    return depth


def depth_complete(sparse):
    """ip_basic fill_in_multiscale per view on the GPU: (V,H,W) f32 -> (V,H,W) f32."""
    _dev(sparse)
    assert sparse.dtype == torch.float32 and sparse.is_contiguous() and sparse.dim() == 3
    V, H, W = sparse.shape
    dense = torch.empty_like(sparse)
    scratch = torch.empty(3 * V * H * W + 2 * V * ((H * W + 255) // 256), dtype=torch.float32, device=sparse.device)
    iscratch = torch.empty(2 * V * W, dtype=torch.int32, device=sparse.device)
    _lib.call('di_depth_complete', sparse.data_ptr(), dense.data_ptr(), scratch.data_ptr(),
              iscratch.data_ptr(), V, H, W, _stream())
    return dense


def bevwarp_gather(bev, depth, img2lidar, aug_fwd, xs, ys, pc_range):
    """bev (1,C,Hb,Wb) channels-last, depth (V,Hi,Wi) f32 -> warped (V,C,Hi,Wi) channels-last."""
    _dev(bev, depth, img2lidar, aug_fwd, xs, ys, pc_range)
    bev = cl(bev)
    _, C, Hb, Wb = bev.shape
    V, Hi, Wi = depth.shape
    assert depth.dtype == torch.float32 and depth.is_contiguous()
    out = empty_cl(V, C, Hi, Wi, bev)
    _lib.call('di_bevwarp_gather_fwd', bev.data_ptr(), depth.data_ptr(), img2lidar.data_ptr(),
              aug_fwd.data_ptr(), xs.data_ptr(), ys.data_ptr(), pc_range.data_ptr(), out.data_ptr(), V, Hi,
              Wi, Hb, Wb, C, _code(bev), _stream())
    return out


def warp_project(bev, depth, img2lidar, aug_fwd, xs, ys, pc_range, chains, head_major=False, outs=None):
    """`pointwise_multi(bevwarp_gather(bev, depth, ...), chains)` in one launch: the warped map is gathered into the
    projection kernel's registers and never written (bit-identical outputs).  At most two chains (the P2I block's key / value
    projections): both weight images stay resident in LDS.  bev (1,128,Hb,Wb) fp16 channels-last.
    head_major: the (single-link) chains write (V, 8, Hi*Wi, 16) - see `token_linear_hm`."""
    _dev(bev, depth, img2lidar, aug_fwd, xs, ys, pc_range)
    bev = cl(bev)
    _, C, Hb, Wb = bev.shape
    V, Hi, Wi = depth.shape
    assert C == 128 and bev.dtype == torch.float16 and depth.dtype == torch.float32 and depth.is_contiguous()
    nc = len(chains)
    assert 1 <= nc <= 2, 'both weight images stay resident in LDS: at most two chains'
    if head_major:
        assert not any(c[3] for c in chains)
        ys_ = [torch.empty((V, 8, Hi * Wi, 16), dtype=bev.dtype, device=bev.device) for _ in chains]
        P, I = ctypes.c_void_p * nc, ctypes.c_int * nc
        a_im, a_y = P(*[c[0].data_ptr() for c in chains]), P(*[y.data_ptr() for y in ys_])
        a_r1, zero, a_hm = I(*[int(bool(c[1])) for c in chains]), I(*([0] * nc)), I(*([Hi * Wi] * nc))
        _lib.call('di_pointwise_multi_warp_hm_fwd', bev.data_ptr(), depth.data_ptr(), img2lidar.data_ptr(), aug_fwd.data_ptr(),
                  xs.data_ptr(), ys.data_ptr(), pc_range.data_ptr(), V, Hi, Wi, Hb, Wb, nc, ctypes.addressof(a_im),
                  ctypes.addressof(a_y), ctypes.addressof(a_r1), ctypes.addressof(zero), ctypes.addressof(zero),
                  ctypes.addressof(a_hm), _stream())
        return ys_
    ys_ = [empty_cl(V, 128, Hi, Wi, bev) for _ in chains] if outs is None else _check_outs(outs, nc, V, Hi, Wi, bev)
    P, I = ctypes.c_void_p * nc, ctypes.c_int * nc
    for (im, r1, r2, two) in chains:
        assert im.dtype == torch.uint8 and im.numel() == 2 * 128 * 128 * 2 + 1024 and im.is_cuda
    a_im, a_y = P(*[c[0].data_ptr() for c in chains]), P(*[y.data_ptr() for y in ys_])
    a_r1, a_r2 = I(*[int(bool(c[1])) for c in chains]), I(*[int(bool(c[2])) for c in chains])
    a_two = I(*[int(bool(c[3])) for c in chains])
    _profiled('pointwise_multi_warp', V, lambda: _lib.call(
        'di_pointwise_multi_warp_fwd', bev.data_ptr(), depth.data_ptr(), img2lidar.data_ptr(), aug_fwd.data_ptr(),
        xs.data_ptr(), ys.data_ptr(), pc_range.data_ptr(), V, Hi, Wi, Hb, Wb, nc, ctypes.addressof(a_im),
        ctypes.addressof(a_y), ctypes.addressof(a_r1), ctypes.addressof(a_r2), ctypes.addressof(a_two), _stream()))
    return ys_


def bevwarp_gather_bwd(grad_out, depth, img2lidar, aug_fwd, xs, ys, pc_range, bev_hw):
    """Gradient of bevwarp_gather w.r.t. the BEV map: (1,C,Hb,Wb) float32, channels-last."""
    _dev(grad_out, depth)
    grad_out = cl(grad_out)
    V, C, Hi, Wi = grad_out.shape
    Hb, Wb = bev_hw
    g = torch.empty((1, C, Hb, Wb), dtype=torch.float32, device=grad_out.device, memory_format=torch.channels_last).zero_()
    _lib.call('di_bevwarp_gather_bwd', grad_out.data_ptr(), depth.data_ptr(), img2lidar.data_ptr(),
              aug_fwd.data_ptr(), xs.data_ptr(), ys.data_ptr(), pc_range.data_ptr(), g.data_ptr(), V, Hi, Wi, Hb,
              Wb, C, _code(grad_out), _stream())
    return g


# ------------------------------------------------------------------ MMPI decoder
def heatmap_nms(dense_a, dense_b, nms_kernel, k1_classes):
    """(sigmoid(a)+sigmoid(b))/2 masked to its local maxima; float32 (B,Cc,H,W)."""
    _dev(dense_a, dense_b)
    a, b = dense_a.contiguous(), dense_b.contiguous()
    B, Cc, H, W = a.shape
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=a.device)
    mask = 0
    for c in k1_classes:
        mask |= 1 << c
    _lib.call('di_heatmap_nms', a.data_ptr(), b.data_ptr(), out.data_ptr(), B, Cc, H, W, nms_kernel, mask,
              _code(a), _stream())
    return out


def topk(scores, k, with_values=False):
    """Indices (B,k) int64 of the k largest entries of each row of non-negative float32 `scores` (B,N), ordered by
    value descending and lower index first on ties (radix select + k-element sort, csrc/topk.hip)."""
    _dev(scores)
    assert scores.dtype == torch.float32 and scores.dim() == 2 and scores.is_contiguous()
    B, N = scores.shape
    idx = torch.empty((B, k), dtype=torch.int64, device=scores.device)
    val = torch.empty((B, k), dtype=torch.float32, device=scores.device) if with_values else None
    ws = torch.empty((int(_lib.lib().di_topk_workspace_bytes(B, N, k)),), dtype=torch.uint8, device=scores.device)
    _lib.call('di_topk_fwd', scores.data_ptr(), idx.data_ptr(), 0 if val is None else val.data_ptr(), ws.data_ptr(),
              B, N, k, _stream())
    return (idx, val) if with_values else idx


def query_geometry(res, proj, aug_rev, per_sample, cell, pc_xy, bev_cell, dim_scale, want_img, want_bev, ld=None):
    """res: dict of float32 (B,k,Q) tensors (center, height, dim, rot) - or, with `ld`, column windows of (B,k,ld)
    tensors (views sliced on the last axis).  Returns (on_img (B,V,Q) int32, rect_img (B,V,Q,4), rect_bev (B,Q,4));
    absent groups are None."""
    if ld is None:
        c, h, d, r = (res[k].contiguous() for k in ('center', 'height', 'dim', 'rot'))
    else:
        c, h, d, r = (res[k] for k in ('center', 'height', 'dim', 'rot'))
        for t in (c, h, d, r):
            assert t.stride(2) == 1 and t.stride(1) == ld and t.stride(0) == ld * t.shape[1]
    _dev(c, h, d, r)
    assert c.dtype == torch.float32 and h.dtype == torch.float32
    B, _, Q = c.shape
    V = proj.shape[1] if proj is not None else 0
    dev = c.device
    on = torch.empty((B, V, Q), dtype=torch.int32, device=dev) if want_img else None
    ri = torch.empty((B, V, Q, 4), dtype=torch.float32, device=dev) if want_img else None
    rb = torch.empty((B, Q, 4), dtype=torch.float32, device=dev) if want_bev else None
    p = lambda t: 0 if t is None else t.data_ptr()
    _lib.call('di_query_geometry_ld', c.data_ptr(), h.data_ptr(), d.data_ptr(), r.data_ptr(), p(proj), p(aug_rev),
              p(per_sample), p(on), p(ri), p(rb), B, Q, Q if ld is None else ld, V, float(cell), float(pc_xy[0]),
              float(pc_xy[1]), float(bev_cell), float(dim_scale), _stream())
    return on, ri, rb


def roi_align(feat, rois, spatial_scale, out_f32=False, out_hl=False):
    """feat (N,C,H,W) channels-last; rois (R,5) f32 [n,x0,y0,x1,y1] -> (R,49,C) (float32 with out_f32; with out_hl - fp16
    map, C = 128 - (R,49,2C) fp16 rows [hi | lo] of float32 accuracy, the matrix operand of `dynconv`)."""
    _dev(feat, rois)
    feat = cl(feat)
    N, C, H, W = feat.shape
    rois = rois.contiguous()
    assert rois.dtype == torch.float32 and rois.shape[1] == 5
    R = rois.shape[0]
    if out_hl:
        assert feat.dtype == torch.float16 and C == 128
        out = torch.empty((R, 49, 2 * C), dtype=torch.float16, device=feat.device)
        code = _lib.DI_F16_HL
    else:
        out = torch.empty((R, 49, C), dtype=torch.float32 if out_f32 else feat.dtype, device=feat.device)
        code = _lib.DI_F32 if out_f32 else _code(feat)
    _lib.call('di_roi_align_x_fwd', feat.data_ptr(), rois.data_ptr(), out.data_ptr(), R, N, H, W, C,
              float(spatial_scale), _code(feat), code, _stream())
    return out


def roi_align_bwd(grad_out, rois, feat_shape, spatial_scale):
    """Gradient of roi_align w.r.t. the feature maps: grad_out (R,49,C) -> (N,C,H,W) float32, channels-last."""
    _dev(grad_out, rois)
    grad_out = grad_out.contiguous()
    N, C, H, W = feat_shape
    R = rois.shape[0]
    g = torch.empty((N, C, H, W), dtype=torch.float32, device=grad_out.device, memory_format=torch.channels_last).zero_()
    _lib.call('di_roi_align_bwd', grad_out.data_ptr(), rois.contiguous().data_ptr(), g.data_ptr(), R, N, H, W, C,
              float(spatial_scale), _code(grad_out), _stream())
    return g


def mha_decode(q, kv, num_heads, scale):
    """q (B,Q,E), kv (B,S,2E) = [K|V] -> softmax(q k^T scale) v, (B,Q,E)."""
    _dev(q, kv)
    q, kv = q.contiguous(), kv.contiguous()
    B, Q, E = q.shape
    S = kv.shape[1]
    assert kv.shape[2] == 2 * E and kv.dtype == q.dtype
    n = _lib.lib().di_mha_decode_scratch_floats(B, Q, S, num_heads)
    scratch = torch.empty(n, dtype=torch.float32, device=q.device)
    out = torch.empty_like(q)
    _lib.call('di_mha_decode_fwd', q.data_ptr(), kv.data_ptr(), out.data_ptr(), scratch.data_ptr(), B, Q, S,
              num_heads, E // num_heads, float(scale), _code(q), _stream())
    return out


# ------------------------------------------------------------------ pillar / voxel producer
_VOXEL_GEO = {}


def voxelize(points, voxel_size, pc_range, max_points, max_voxels, n_feat=None):
    """Hard voxelisation (spconv PointToVoxel semantics, first-come order).  points (N,D') float32 on the device
    -> voxels (max_voxels, max_points, n_feat), coords (max_voxels, 3) int32 [z,y,x], num_points (max_voxels,)
    int32, n_voxels (0-dim int64 tensor on the device: no host synchronisation here).  Rows >= n_voxels are zero."""
    _dev(points)
    assert points.dtype == torch.float32 and points.dim() == 2 and (points.shape[0] == 0 or points.stride(1) == 1)
    N, stride = points.shape[0], points.stride(0)
    D = n_feat or points.shape[1]
    dev = points.device
    grid = [int(round((pc_range[3 + a] - pc_range[a]) / voxel_size[a])) for a in range(3)]
    gkey = (tuple(float(x) for x in pc_range), tuple(float(x) for x in voxel_size), str(dev))
    geo = _VOXEL_GEO.get(gkey)                    # a device constant: built once (no host->device copy per call, so
    if geo is None:                               # the op can be captured in a hipGraph)
        geo = _VOXEL_GEO[gkey] = torch.tensor(list(pc_range) + list(voxel_size), dtype=torch.float32, device=dev)
    voxels = torch.zeros((max_voxels, max_points, D), dtype=torch.float32, device=dev)
    coords = torch.zeros((max_voxels, 3), dtype=torch.int32, device=dev)
    num = torch.zeros((max_voxels,), dtype=torch.int32, device=dev)
    if N == 0:
        return voxels, coords, num, torch.zeros((), dtype=torch.int64, device=dev)
    keys = torch.empty(N, dtype=torch.int64, device=dev)
    _lib.call('di_voxel_keys', points.data_ptr(), N, stride, geo.data_ptr(), grid[0], grid[1], grid[2],
              keys.data_ptr(), _stream())
    skeys = torch.sort(keys).values
    head = torch.empty(N, dtype=torch.int32, device=dev)
    first = torch.empty(N, dtype=torch.int64, device=dev)
    _lib.call('di_voxel_heads', skeys.data_ptr(), N, head.data_ptr(), first.data_ptr(), _stream())
    seg = (torch.cumsum(head, 0, dtype=torch.int32) - 1).contiguous()
    sfirst = torch.sort(first).values
    slot_of_seg = torch.empty(N, dtype=torch.int32, device=dev)
    head_of_seg = torch.empty(N, dtype=torch.int32, device=dev)
    _lib.call('di_voxel_scatter', points.data_ptr(), N, stride, D, skeys.data_ptr(), sfirst.data_ptr(),
              seg.data_ptr(), slot_of_seg.data_ptr(), head_of_seg.data_ptr(), grid[0], grid[1], max_points,
              max_voxels, voxels.data_ptr(), coords.data_ptr(), num.data_ptr(), _stream())
    n_vox = torch.clamp(head.sum(dtype=torch.int64), max=max_voxels)
    return voxels, coords, num, n_vox


def bias_act_(y, bias, residual=None, relu=True):
    """In place: y = act(y + bias[c] (+ residual)) on a channels-last fp16 map (n, C, H, W) - the one-pass epilogue of a library
    convolution (torch's conv2d-with-bias, add_ and relu_ are one pass EACH).  bias float32 (C)."""
    _dev(y, bias)
    n, C, H, W = y.shape
    assert y.dtype == torch.float16 and y.is_contiguous(memory_format=torch.channels_last), 'channels-last fp16 map'
    assert bias.dtype == torch.float32 and bias.numel() == C and bias.is_contiguous()
    if residual is not None:
        assert residual.shape == y.shape and residual.dtype == y.dtype and residual.is_contiguous(memory_format=torch.channels_last)
    _lib.call('di_bias_act_inplace', y.data_ptr(), bias.data_ptr(), residual.data_ptr() if residual is not None else None,
              n * H * W, C, int(bool(relu)), _stream())
    return y


def upsample_add_(lo, hi):
    """In place: lo += F.interpolate(hi, size=lo.shape[2:], mode='nearest') on channels-last fp16 maps - the FPN top-down step
    as one pass."""
    _dev(lo, hi)
    n, C, Hl, Wl = lo.shape
    assert hi.shape[:2] == (n, C) and lo.dtype == hi.dtype == torch.float16
    assert lo.is_contiguous(memory_format=torch.channels_last) and hi.is_contiguous(memory_format=torch.channels_last)
    _lib.call('di_upsample_add_inplace', lo.data_ptr(), hi.data_ptr(), n, Hl, Wl, hi.shape[2], hi.shape[3], C, _stream())
    return lo


# ------------------------------------------------------------------ sparse 3-D convolutions of the frozen LiDAR encoder
def _geo16(batch, in_shape, out_shape, ksize, stride, padding):
    return (ctypes.c_int32 * 16)(int(batch), *map(int, in_shape), *map(int, out_shape), *map(int, ksize), *map(int, stride),
                                 *map(int, padding))


def sparse_out_shape(in_shape, ksize, stride, padding):
    return tuple((n + 2 * p - k) // s + 1 for n, k, s, p in zip(in_shape, ksize, stride, padding))


def sparse_output_keys(in_keys, batch, in_shape, ksize, stride, padding):
    """spconv `SparseConv3d`'s output set: the SORTED int32 keys ((b * D + z) * H + y) * W + x of the output cells whose window
    holds an active input voxel (a byte map of the output grid, marked by one launch; its non-zero scan is the key list - this
    synchronises, the count decides the shapes that follow) and the output shape."""
    _dev(in_keys)
    assert in_keys.dtype == torch.int32 and in_keys.is_contiguous()
    out_shape = sparse_out_shape(in_shape, ksize, stride, padding)
    occ = torch.zeros(int(batch) * out_shape[0] * out_shape[1] * out_shape[2], dtype=torch.uint8, device=in_keys.device)
    geo = _geo16(batch, in_shape, out_shape, ksize, stride, padding)
    _lib.call('di_sparse_mark', in_keys.data_ptr(), in_keys.numel(), ctypes.addressof(geo), occ.data_ptr(), _stream())
    return torch.nonzero(occ).squeeze(1).to(torch.int32), out_shape


def sparse_rowstart(in_keys, batch, in_shape):
    """First position of the sorted key list for every row (b, z, y) of the level's grid (+ a sentinel): the search ranges of
    `sparse_neighbours` - once per level."""
    _dev(in_keys)
    assert in_keys.dtype == torch.int32 and in_keys.is_contiguous()
    rowstart = torch.empty(int(batch) * int(in_shape[0]) * int(in_shape[1]) + 1, dtype=torch.int32, device=in_keys.device)
    geo = _geo16(batch, in_shape, in_shape, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    _lib.call('di_sparse_rowstart', in_keys.data_ptr(), in_keys.numel(), ctypes.addressof(geo), rowstart.data_ptr(), _stream())
    return rowstart


def sparse_neighbours(in_keys, out_keys, batch, in_shape, out_shape, ksize, stride, padding, rowstart=None):
    """The rulebook of a sparse convolution as a neighbour table (K, M_out) int32: row of `in_keys` (sorted) that kernel offset o
    (index order kd, kh, kw) of output voxel m reads, or -1.  rowstart: `sparse_rowstart` of the input level (shared by the tables
    of a level); None = computed here."""
    _dev(in_keys, out_keys)
    assert in_keys.dtype == out_keys.dtype == torch.int32 and in_keys.is_contiguous() and out_keys.is_contiguous()
    K = int(ksize[0]) * int(ksize[1]) * int(ksize[2])
    nbr = torch.empty((K, out_keys.numel()), dtype=torch.int32, device=in_keys.device)
    if out_keys.numel() == 0:
        return nbr
    if rowstart is None:
        rowstart = sparse_rowstart(in_keys, batch, in_shape)
    assert rowstart.dtype == torch.int32 and rowstart.numel() == int(batch) * int(in_shape[0]) * int(in_shape[1]) + 1
    geo = _geo16(batch, in_shape, out_shape, ksize, stride, padding)
    _lib.call('di_sparse_nbr', in_keys.data_ptr(), out_keys.data_ptr(), in_keys.numel(), out_keys.numel(), ctypes.addressof(geo),
              rowstart.data_ptr(), nbr.data_ptr(), _stream())
    return nbr


def sparse_weight_fragments(weight, K, cin):
    """(K * cin, cout) float weights (offsets in kernel index order, then input channels) -> (fp16 fragments in the MFMA operand
    order of di_sparse_conv_fwd, cin_pad)."""
    cout = weight.shape[1]
    cin_pad = (cin + 31) // 32 * 32
    w = weight.new_zeros((K, cin_pad, cout))
    w[:, :cin] = weight.reshape(K, cin, cout)
    w = w.view(K, cin_pad // 32, 4, 8, cout // 16, 16).permute(0, 1, 4, 2, 5, 3)        # K, kk, mt, g, i, e
    return w.contiguous().to(torch.float16), cin_pad


def sparse_conv(feats, nbr, wfrag, bias, cin_pad, cout, relu=True, residual=None):
    """out[m] = act(sum_o feats[nbr[o, m]] W[o] + bias (+ residual[m])): gather, product and epilogue in one launch.  feats
    (M_in + 1, cin) with a ZERO LAST ROW (`sparse_rows`), the result has the same form: (M_out + 1, cout), last row zero."""
    _dev(feats, nbr, wfrag)
    assert feats.dtype == torch.float16 and feats.is_contiguous() and nbr.dtype == torch.int32 and nbr.is_contiguous()
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == cout)
    K, M_out = nbr.shape
    M_in, cin = feats.shape[0] - 1, feats.shape[1]
    assert M_in >= 0 and wfrag.dtype == torch.float16 and wfrag.numel() == K * cin_pad * cout
    out = torch.empty((M_out + 1, cout), dtype=torch.float16, device=feats.device)
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == torch.float16 and residual.is_contiguous()
    _profiled('sparse_conv_fwd', M_out, lambda: _lib.call(
        'di_sparse_conv_fwd', feats.data_ptr(), nbr.data_ptr(), wfrag.data_ptr(), bias.data_ptr() if bias is not None else None,
        residual.data_ptr() if residual is not None else None, out.data_ptr(), M_in, M_out, K, cin, cin_pad, cout,
        int(bool(relu)), _stream()))
    return out


def sparse_rows(feats, cin=None):
    """(M, C) features -> the kernels' form: fp16 (M + 1, cin) rows, channels zero-padded to `cin` (default: the next multiple of
    8), one zero row behind the last voxel."""
    M, C = feats.shape
    cin = (C + 7) // 8 * 8 if cin is None else cin
    return torch.nn.functional.pad(feats.to(torch.float16), (0, cin - C, 0, 1)).contiguous()


# ------------------------------------------------------------------ DeepInteraction++ operators
def _rows(t):
    """(rows.., cols) view with unit column stride and uniform row stride -> (data_ptr, row stride in elements)."""
    assert t.stride(-1) == 1
    rs = t.stride(-2)
    for d in range(t.dim() - 2):
        assert t.stride(d) == t.stride(d + 1) * t.shape[d + 1], 'rows must be uniformly strided'
    return t.data_ptr(), rs


def ms_deform_attn(value, offsets, logits, ref, level_hw, n_points=4, head_major=False):
    """mmcv MultiScaleDeformableAttention core.  value (bs,S,128) contiguous; offsets (bs,nq,8*L*P*2) and logits
    (bs,nq,8*L*P): views (possibly into one packed projection) with unit column stride; ref (1|bs,nq,L,2) float32;
    level_hw [(H,W),...] -> (bs,nq,128).  head_major: value is (bs, 8, S, 16) fp16 (`token_linear_hm`)."""
    _dev(value, offsets, logits, ref)
    if head_major:
        bs, _, S, _ = value.shape
        nq, L = offsets.shape[1], len(level_hw)
        assert value.shape == (bs, 8, S, 16) and value.is_contiguous() and value.dtype == torch.float16
        assert offsets.dtype == logits.dtype == torch.float16
        assert offsets.shape[-1] == 8 * L * n_points * 2 and logits.shape[-1] == 8 * L * n_points
        assert sum(h * w for h, w in level_hw) == S and ref.dtype == torch.float32 and ref.is_contiguous()
        assert ref.shape[1:] == (nq, L, 2) and ref.shape[0] in (1, bs)
        op, ors = _rows(offsets)
        lp, lrs = _rows(logits)
        out = torch.empty((bs, nq, 128), dtype=value.dtype, device=value.device)
        hw = _level_array(level_hw)
        _profiled('ms_deform_attn_fwd', bs * nq, lambda: _lib.call(
            'di_ms_deform_attn_hm_fwd', value.data_ptr(), op, ors, lp, lrs, ref.data_ptr(), int(ref.shape[0] == 1),
            out.data_ptr(), bs, nq, L, n_points, ctypes.addressof(hw), _stream()))
        return out
    bs, S, E = value.shape
    nq, L = offsets.shape[1], len(level_hw)
    assert E == 128 and value.is_contiguous() and offsets.dtype == value.dtype == logits.dtype
    assert offsets.shape[-1] == 8 * L * n_points * 2 and logits.shape[-1] == 8 * L * n_points
    assert sum(h * w for h, w in level_hw) == S and ref.dtype == torch.float32 and ref.is_contiguous()
    assert ref.shape[1:] == (nq, L, 2) and ref.shape[0] in (1, bs)
    op, ors = _rows(offsets)
    lp, lrs = _rows(logits)
    out = torch.empty((bs, nq, E), dtype=value.dtype, device=value.device)
    hw = _level_array(level_hw)
    _profiled('ms_deform_attn_fwd', bs * nq, lambda: _lib.call(
        'di_ms_deform_attn_fwd', value.data_ptr(), op, ors, lp, lrs, ref.data_ptr(), int(ref.shape[0] == 1),
        out.data_ptr(), bs, nq, L, n_points, ctypes.addressof(hw), _code(value), _stream()))
    return out


def grid_gather(feat, grid, add=None, grids_per_feat=1):
    """Bilinear grid_sample (zeros, align_corners=False) of a channels-last map at explicit points.  feat (Bf,C,H,W)
    channels-last, grid (Bg,N,2) float32 in [-1,1] (x,y), add (N,C) or None -> (Bg,N,C); grid g reads
    feat[g // grids_per_feat]."""
    _dev(feat, grid)
    assert _is_cl(feat) and grid.dtype == torch.float32 and grid.is_contiguous() and grid.shape[-1] == 2
    Bf, C, H, W = feat.shape
    Bg, N = grid.shape[:2]
    assert Bg == Bf * grids_per_feat
    if add is not None:
        assert add.shape == (N, C) and add.dtype == feat.dtype and add.is_contiguous()
    out = torch.empty((Bg, N, C), dtype=feat.dtype, device=feat.device)
    _lib.call('di_grid_gather_fwd', feat.data_ptr(), grid.data_ptr(), 0 if add is None else add.data_ptr(),
              out.data_ptr(), Bg, N, grids_per_feat, H, W, C, _code(feat), _stream())
    return out


def polar_bev_sample(polar, bev, proj, aug_rev, cam_xy, params):
    """fusion_transformerv4.py:581-640 for all cameras at once.  polar (B,V,Wp,R,C) contiguous (ray-major), bev (B,C,Hb,Wb)
    channels-last, proj (B,V,4,4) / aug_rev (B,12) / cam_xy (B,V,2) / params (10,) float32 -> (B,C,Hb,Wb)."""
    _dev(polar, bev, proj, aug_rev, cam_xy, params)
    B, V, Wp, R, C = polar.shape
    assert polar.is_contiguous() and _is_cl(bev) and bev.dtype == polar.dtype and bev.shape[:2] == (B, C)
    for t, shp in ((proj, (B, V, 4, 4)), (aug_rev, (B, 12)), (cam_xy, (B, V, 2)), (params, (10,))):
        assert t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shp, (t.shape, shp)
    Hb, Wb = bev.shape[-2:]
    out = empty_cl(B, C, Hb, Wb, bev)
    _profiled('polar_bev_sample_fwd', B * Hb * Wb, lambda: _lib.call(
        'di_polar_bev_sample_fwd', polar.data_ptr(), bev.data_ptr(), proj.data_ptr(), aug_rev.data_ptr(),
        cam_xy.data_ptr(), params.data_ptr(), out.data_ptr(), B, V, R, Wp, Hb, Wb, C, _code(bev), _stream()))
    return out


def mha_small(q, k, v, num_heads):
    """softmax(q k^T / sqrt(16)) v per head for many short sequences.  q (N,T,E), k/v (N,S,E) views with unit column
    stride (k and v share a row stride, e.g. halves of one packed projection) -> (N,T,E) contiguous."""
    _dev(q, k, v)
    N, T, E = q.shape
    S = k.shape[1]
    assert E == num_heads * 16 and k.shape == v.shape == (N, S, E) and q.dtype == k.dtype == v.dtype
    qp, qrs = _rows(q)
    kp, krs = _rows(k)
    vp, vrs = _rows(v)
    assert krs == vrs
    out = torch.empty((N, T, E), dtype=q.dtype, device=q.device)
    _profiled('mha_small_fwd', N, lambda: _lib.call(
        'di_mha_small_fwd', qp, qrs, kp, vp, krs, out.data_ptr(), E, N, T, S, num_heads, 16, 0.25, _code(q), _stream()))
    return out


def add_layernorm(x, res, weight, bias, eps=1e-5):
    """LayerNorm(x + res) over the last dim (res may be None); x (..., C) contiguous, C <= 128."""
    _dev(x, weight, bias)
    C = x.shape[-1]
    assert x.is_contiguous() and weight.dtype == x.dtype == bias.dtype and weight.shape == (C,)
    if res is not None:
        assert res.shape == x.shape and res.dtype == x.dtype and res.is_contiguous()
    out = torch.empty_like(x)
    n = x.numel() // C
    _profiled('add_layernorm_fwd', n, lambda: _lib.call(
        'di_add_layernorm_fwd', x.data_ptr(), 0 if res is None else res.data_ptr(), weight.data_ptr(), bias.data_ptr(),
        out.data_ptr(), n, C, float(eps), _code(x), _stream()))
    return out


def _level_array(level_hw):
    L = len(level_hw)
    return (ctypes.c_int32 * (2 * L))(*[int(x) for pair in level_hw for x in pair])


def ms_deform_attn_bwd(value, offsets, logits, ref, level_hw, grad_out, n_points=4):
    """-> (grad_value (bs,S,128) float32, grad_proj (bs,nq,8*L*P*3) in the input dtype: [d offsets | d logits])."""
    _dev(value, offsets, logits, ref, grad_out)
    bs, S, E = value.shape
    nq, L = offsets.shape[1], len(level_hw)
    op, ors = _rows(offsets)
    lp, lrs = _rows(logits)
    grad_out = grad_out.contiguous()
    gv = torch.zeros((bs, S, E), dtype=torch.float32, device=value.device)
    ncol = 8 * L * n_points * 3
    gp = torch.empty((bs, nq, ncol), dtype=value.dtype, device=value.device)
    hw = _level_array(level_hw)
    _lib.call('di_ms_deform_attn_bwd', value.data_ptr(), op, ors, lp, lrs, ref.data_ptr(), int(ref.shape[0] == 1),
              grad_out.data_ptr(), gv.data_ptr(), gp.data_ptr(), ncol, bs, nq, L, n_points, ctypes.addressof(hw),
              _code(value), _stream())
    return gv, gp


def grid_gather_bwd(grid, grad_out, feat_shape, grids_per_feat=1):
    """-> grad_feat (Bf,C,H,W) float32, channels-last."""
    _dev(grid, grad_out)
    Bf, C, H, W = feat_shape
    Bg, N = grid.shape[:2]
    grad_out = grad_out.contiguous()
    gf = torch.empty((Bf, C, H, W), dtype=torch.float32, device=grid.device, memory_format=torch.channels_last).zero_()
    _lib.call('di_grid_gather_bwd', grid.data_ptr(), grad_out.data_ptr(), gf.data_ptr(), Bg, N, grids_per_feat, H, W, C,
              _code(grad_out), _stream())
    return gf


def polar_bev_sample_bwd(grad_out, proj, aug_rev, cam_xy, params, polar_shape):
    """grad_out (B,C,Hb,Wb) channels-last -> grad_polar (B,V,Wp,R,C) float32."""
    _dev(grad_out, proj)
    B, V, Wp, R, C = polar_shape
    grad_out = cl(grad_out)
    Hb, Wb = grad_out.shape[-2:]
    gp = torch.zeros(polar_shape, dtype=torch.float32, device=grad_out.device)
    _lib.call('di_polar_bev_sample_bwd', grad_out.data_ptr(), proj.data_ptr(), aug_rev.data_ptr(), cam_xy.data_ptr(),
              params.data_ptr(), gp.data_ptr(), B, V, R, Wp, Hb, Wb, C, _code(grad_out), _stream())
    return gp


# ------------------------------------------------------------------ 3x3 convolutions (implicit GEMM on the matrix cores)
def pack_conv3x3(weight, bias=None, bn=None):
    """Constants of `conv3x3` from a torch Conv2d weight (Cout,Cin,3,3) [+ bias] [+ a following eval-mode BatchNorm2d,
    folded in]: (w_packed fp16 (Cout_pad, 9, Cin), w_staged fp16 (Cin/32, 3, 3, 128, 32) in LDS slot order or None, bias float32
    (Cout))."""
    with torch.no_grad():
        w = weight.detach().float()
        Cout = w.shape[0]
        b = torch.zeros(Cout, device=w.device) if bias is None else bias.detach().float()
        if bn is not None:
            g = torch.rsqrt(bn.running_var.float() + bn.eps)
            if bn.weight is not None:
                g = g * bn.weight.float()
            w = w * g[:, None, None, None]
            b = (b - bn.running_mean.float()) * g + (0.0 if bn.bias is None else bn.bias.float())
        wp = w.permute(0, 2, 3, 1).reshape(Cout, 9, -1)
        if Cout < 16:
            wp = torch.cat([wp, wp.new_zeros(16 - Cout, 9, wp.shape[-1])])
        ws = None
        Cin = w.shape[1]
        if Cout == 128 and Cin % 32 == 0:          # (chunk, ky, kx, n, c): one contiguous 24 KB tile per (chunk, ky)
            # rows permuted for 16-B output stores: staged row 16nb + 4g + r = channel 32(nb/2) + 8g + 4(nb%2) + r
            rr = torch.arange(128, device=w.device)
            nb, gq, rq = rr // 16, (rr % 16) // 4, rr % 4
            wr = w[32 * (nb // 2) + 8 * gq + 4 * (nb % 2) + rq]
            ws = wr.view(128, Cin // 32, 32, 3, 3).permute(1, 3, 4, 0, 2).to(torch.float16).contiguous()
            # ... stored in LDS order: the 16-B slot s of row R = kx * 128 + n of a (chunk, ky) tile sits at slot
            # (s + 2 * (R >> 2)) & 3 (the conflict-free rotation of csrc/conv3x3.hip lds_off), so that the tile can be
            # moved by linear copies / LDS-DMA
            t = ws.view(Cin // 32, 3, 384, 4, 8)
            R = torch.arange(384, device=w.device).view(384, 1)
            sl = torch.arange(4, device=w.device).view(1, 4)
            rot = torch.empty_like(t)
            rot[:, :, R.expand(384, 4), (sl + 2 * (R >> 2)) & 3] = t
            ws = rot.view(Cin // 32, 3, 3, 128, 32).contiguous()
        return wp.to(torch.float16).contiguous(), ws, b.contiguous()


def conv3x3(x, w_packed, w_staged, bias, relu=False, out_nchw=False, use_staged=True, out_f32=False):
    """x (n,Cin,H,W) channels-last fp16 -> (n,Cout,H,W), channels-last (or contiguous NCHW when out_nchw; float32 with
    out_f32 - NCHW only)."""
    _dev(x, w_packed, bias)
    x = cl(x)
    n, Cin, H, W = x.shape
    Cout = bias.numel()
    assert x.dtype == torch.float16 and w_packed.dtype == torch.float16 and w_packed.shape[1:] == (9, Cin)
    assert bias.dtype == torch.float32
    assert out_nchw or not out_f32
    if out_nchw:
        y = torch.empty((n, Cout, H, W), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    else:
        y = empty_cl(n, Cout, H, W, x)
    ws = 0 if (w_staged is None or not use_staged) else w_staged.data_ptr()
    _profiled('conv3x3_fwd', n * Cin, lambda: _lib.call('di_conv3x3_fwd', x.data_ptr(), w_packed.data_ptr(), ws,
                                                       bias.data_ptr(), y.data_ptr(), n, H, W, Cin, Cout, int(bool(relu)),
                                                       (2 if out_f32 else 1) if out_nchw else 0, _stream()))
    return y


# ------------------------------------------------------------------ token-level kernels of the MMPI decoder (float32)
def _f32c(t):
    assert t.dtype == torch.float32 and t.stride(-1) == 1, 'token kernels take row-major float32'
    return t


class TokenProgram:
    """One launch of `di_token_program` (csrc/token32.hip): a short program of steps that every group of 16 consecutive
    tokens of a sample runs on rows held in LDS buffers 0..2 (512 floats wide).  `roles` > 1 launches that many
    workgroups per token group; a step runs in the roles of its `roles=(lo, hi)` range (default: all) - see
    include/deepinteraction_hip.h.  The builder keeps the tensors it was given alive until `run` has launched."""

    def __init__(self, roles=1):
        self.steps, self.refs, self.head_desc, self.roles = [], [], None, roles

    def _add(self, kind, src=0, dst=0, aux=-1, K=0, N=0, a=0, b=0, f=0.0, p0=None, p1=None, p2=None, ld0=0, ld1=0,
             roles=None, rt=0, rc=0, nch=0, roff=0):
        st = _lib.TokStep()
        st.kind, st.src, st.dst, st.aux, st.K, st.N, st.a, st.b, st.f = kind, src, dst, aux, K, N, a, b, float(f)
        st.role_lo, st.role_hi = (0, self.roles - 1) if roles is None else roles
        assert 0 <= st.role_lo <= st.role_hi < self.roles
        st.rt, st.rc, st.nch, st.roff = rt, rc, nch, roff
        for name, t in (('p0', p0), ('p1', p1), ('p2', p2)):
            if t is not None:
                _dev(t)
                setattr(st, name, t.data_ptr())
                self.refs.append(t)
        st.ld0, st.ld1 = ld0, ld1
        self.steps.append(st)
        return self

    def load(self, dst, x, pos=None, col=0, roles=None):
        """buf[dst][:, col:col+K] = x (+ pos); x, pos (M,K) float32."""
        _f32c(x)
        if pos is not None:
            assert _f32c(pos).shape == x.shape
        return self._add(_lib.TOK_LOAD, dst=dst, K=x.shape[1], a=col, p0=x, p1=pos, ld0=x.stride(0),
                         ld1=0 if pos is None else pos.stride(0), roles=roles)

    def load_parts(self, dst, workspace, nslices, total_rows, bias=None, roles=None):
        """buf[dst][:, :128] = sum of the partial sums workspace (nslices, total_rows, 128) (+ bias)."""
        return self._add(_lib.TOK_LOAD_PARTS, dst=dst, a=nslices, b=total_rows, p0=workspace, p1=bias, roles=roles)

    def attn(self, dst, qk, vt, scale, member=None, view=None, roles=None):
        """buf[dst][:, :128] = self attention (8 heads x 16) among the sample's tokens from qk (M, 256) = [q|k] rows and
        vt (B, 128, Qp) = the values transposed (`store_t`), Qp = Q rounded up to 16."""
        assert _f32c(qk).shape[1] == 256 and (member is None) == (view is None)
        assert _f32c(vt).dim() == 3 and vt.shape[1] == 128 and vt.shape[2] % 16 == 0 and vt.is_contiguous()
        st = self._add(_lib.TOK_ATTN, dst=dst, f=scale * 1.4426950408889634, b=vt.shape[2], p0=qk, p1=member, p2=view,
                       ld0=qk.stride(0), roles=roles)
        self.steps[-1].p3 = vt.data_ptr()
        self.refs.append(vt)
        return st

    def combine(self, dst, scratch, nrange, roles=None):
        """buf[dst][:, :128] = merged key-range states of `mha_decode_x`."""
        return self._add(_lib.TOK_COMBINE, dst=dst, a=nrange, p0=scratch, roles=roles)

    def linear(self, src, dst, w, bias=None, act=0, roles=None, n_per_role=None, k_per_role=None):
        """buf[dst][:, :N] = act(buf[src][:, :K] @ W.T + bias); w = `pack_linear(W)` for W (N,K) float32; act 0 none / 1 ReLU
        / 2 GELU.  n_per_role: role r of `roles` computes output channels [r*n, (r+1)*n) (into buf[dst][:, :n]);
        k_per_role: role r multiplies input channels [r*k, (r+1)*k) (from buf[src][:, :k]) - a partial sum, no bias."""
        assert w.dtype == torch.float16 and w.dim() == 6 and w.is_contiguous()
        N, K = w.shape[0] * 16, w.shape[1] * 128
        assert bias is None or (_f32c(bias).numel() == N)
        rt = rc = 0
        if n_per_role is not None:
            assert n_per_role % 16 == 0
            rt, N = n_per_role // 16, n_per_role
        if k_per_role is not None:
            assert k_per_role % 128 == 0 and bias is None
            rc, K = k_per_role // 128, k_per_role
        return self._add(_lib.TOK_LINEAR, src=src, dst=dst, K=K, N=N, a=act, p0=w, p1=bias, roles=roles, rt=rt, rc=rc,
                         nch=w.shape[1])

    def rowop(self, src, dst, aux=-1, ln=None, eps=1e-5, relu=False, keep=None, roles=None):
        """buf[dst][:, :128] = keep * relu?(LayerNorm?(buf[src] + buf[aux]?))."""
        lw, lb = ln if ln is not None else (None, None)
        return self._add(_lib.TOK_ROWOP, src=src, dst=dst, aux=aux, b=int(bool(relu)), f=eps, p0=lw, p1=lb, p2=keep,
                         roles=roles)

    def store(self, src, y, col=0, roles=None, n=None, role_offset=0):
        """y (M,N) = buf[src][:, col:col+N]; with n / role_offset: role r stores n columns at y + r * role_offset elements
        (columns r*n.. of a wider row, or slice r of a (roles, M, n) workspace)."""
        _f32c(y)
        return self._add(_lib.TOK_STORE, src=src, N=y.shape[-1] if n is None else n, a=col, p0=y, ld0=y.stride(-2),
                         roles=roles, roff=role_offset)

    def store_t(self, src, vt, col=0, roles=None):
        """vt (B, N, Qp)[sample, c, q] = buf[src][q - q0, col + c]: the transposed store (`attn` reads its values so)."""
        assert _f32c(vt).dim() == 3 and vt.is_contiguous() and vt.shape[2] % 16 == 0
        return self._add(_lib.TOK_STORE, src=src, N=vt.shape[1], a=col, b=1, p0=vt, ld0=vt.shape[2], roles=roles)

    def store_hl(self, src, y_hl, col=0, roles=None):
        """y_hl (M, 2N) fp16 = [hi N | lo N] of buf[src][:, col:col+N]: the split store (`token_wide` reads tokens so)."""
        assert y_hl.dtype == torch.float16 and y_hl.stride(-1) == 1 and y_hl.shape[-1] % 16 == 0
        return self._add(_lib.TOK_STORE, src=src, N=y_hl.shape[-1] // 2, a=col, b=2, p0=y_hl, ld0=y_hl.stride(-2), roles=roles)

    def heads(self, src, w2, b2, qpos, outs, cls, center_head, ldo, col0, keep=None, first=None, pos_out=None, roles=None,
              per_role=False, qpos2=None, pos2_out=None):
        """Second layers of the prediction heads on the hidden rows in buf[src] (see include/deepinteraction_hip.h);
        per_role: role r evaluates head r only, its 64 hidden channels at columns 0..63."""
        h = _lib.TokHeads()
        n = len(cls)
        assert n <= _lib.TOK_MAX_HEADS and (keep is None or first is not None)
        h.w2, h.b2, h.qpos = w2.data_ptr(), b2.data_ptr(), qpos.data_ptr()
        h.keep = None if keep is None else keep.data_ptr()
        h.pos_out = None if pos_out is None else pos_out.data_ptr()
        assert (qpos2 is None) == (pos2_out is None)
        h.qpos2 = None if qpos2 is None else _f32c(qpos2).data_ptr()          # ++: pos2_out = raw centre offset + qpos2
        h.pos2_out = None if pos2_out is None else _f32c(pos2_out).data_ptr()
        for i in range(n):
            h.out[i] = outs[i].data_ptr()
            h.first[i] = first[i].data_ptr() if first is not None else None
            h.cls[i] = int(cls[i])
        h.nheads, h.center_head, h.ldo, h.col0 = n, center_head, ldo, col0
        self.refs += [w2, b2, qpos, keep, pos_out, qpos2, pos2_out] + list(outs) + (list(first) if first is not None else [])
        self.head_desc = h
        return self._add(_lib.TOK_HEADS, src=src, a=int(bool(per_role)), roles=roles)

    def run(self, B, Q, stamps=None):
        """Launch.  `stamps` (int64 (nsteps + 1,), device): profiling aid - shader-clock stamps of workgroup (0,0,0)."""
        n = len(self.steps)
        arr = (_lib.TokStep * n)(*self.steps)
        hp = ctypes.addressof(self.head_desc) if self.head_desc is not None else 0
        if stamps is None:
            _lib.call('di_token_program', ctypes.addressof(arr), n, hp, B, Q, _stream())
        else:
            assert stamps.dtype == torch.int64 and stamps.numel() >= n + 1
            _lib.call('di_token_program_timed', ctypes.addressof(arr), n, hp, B, Q, stamps.data_ptr(), _stream())


def token_wide(x_hl, w_packed, bias):
    """DynamicConv's parameter generator (weight stationary): tokens x_hl (M,256) fp16 = [hi | lo] (`TokenProgram.store_hl`
    / `split_rows`) -> params (M, 65536) fp16, the hi / lo fragments `dynconv` reads; w_packed, bias from
    `decoder_fused._dyn_layout`."""
    _dev(x_hl, w_packed)
    M = x_hl.shape[0]
    assert x_hl.shape[1] == 256 and x_hl.dtype == torch.float16 and x_hl.stride(1) == 1
    assert w_packed.dtype == torch.float16 and w_packed.numel() == 32768 * 128 * 2 and w_packed.is_contiguous()
    assert bias.dtype == torch.float32 and bias.numel() == 32768
    y = torch.empty((M, 65536), dtype=torch.float16, device=x_hl.device)
    _lib.call('di_token_wide', x_hl.data_ptr(), x_hl.stride(0), w_packed.data_ptr(), bias.data_ptr(), y.data_ptr(), M, _stream())
    return y


def split_rows(x):
    """(M,N) float32 -> (M,2N) fp16 rows [hi | lo] (tests; the product's programs write this form with `store_hl`)."""
    hi, lo = split_hi_lo(x)
    return torch.cat([hi, lo], -1).contiguous()


def token_splitk(f2p, w_packed):
    """Partial sums of DynamicConv's out_layer over K slices: f2p (K/32, M, 64) fp16 from `dynconv`, w_packed =
    `pack_ksteps(W)` for W (128, K) -> (workspace float32 (slices, M, 128), slices)."""
    _dev(f2p, w_packed)
    nks, M, _ = f2p.shape
    K = nks * 32
    assert f2p.dtype == torch.float16 and f2p.is_contiguous() and f2p.shape[2] == 64
    assert w_packed.dtype == torch.float16 and w_packed.numel() == 128 * K * 2 and w_packed.is_contiguous()
    ws = torch.empty(int(_lib.lib().di_token_splitk_workspace_bytes(M, K)) // 4, dtype=torch.float32, device=f2p.device)
    ns = ctypes.c_int(0)
    _lib.call('di_token_splitk', f2p.data_ptr(), w_packed.data_ptr(), ws.data_ptr(), M, K, ctypes.addressof(ns), _stream())
    return ws, ns.value


def dynconv(roi, params, n1, n2, eps=1e-5):
    """roi (R,49,256) fp16 = [hi | lo] (`roi_align(out_hl=True)`), params (R, 65536) fp16 from `token_wide`, n1/n2 =
    (weight, bias) float32 of DynamicConv.norm1/2 -> relu(LN2(relu(LN1(roi @ p1)) @ p2)) as f2p (196, R, 64) fp16 (k-step
    major, hi | lo)."""
    _dev(roi, params)
    R = roi.shape[0]
    assert roi.shape == (R, 49, 256) and roi.is_contiguous() and roi.dtype == torch.float16
    assert params.shape == (R, 65536) and params.is_contiguous() and params.dtype == torch.float16
    assert n1[0].dtype == torch.float32
    f2p = torch.empty((196, R, 64), dtype=torch.float16, device=roi.device)
    _lib.call('di_dynconv_fwd', roi.data_ptr(), params.data_ptr(), n1[0].data_ptr(), n1[1].data_ptr(), n2[0].data_ptr(),
              n2[1].data_ptr(), f2p.data_ptr(), R, float(eps), _stream())
    return f2p


def f2p_to_dense(f2p):
    """(K/32, M, 64) hi | lo fp16 -> (M, K) float64 (tests)."""
    nks, M, _ = f2p.shape
    v = f2p[..., :32].double() + f2p[..., 32:].double() / 2048.0
    return v.permute(1, 0, 2).reshape(M, nks * 32)


def dense_to_f2p(x):
    """(M, K) float32 -> (K/32, M, 64) hi | lo fp16 (tests)."""
    M, K = x.shape
    hi, lo = split_hi_lo(x)
    return torch.cat([hi.view(M, K // 32, 32), lo.view(M, K // 32, 32)], -1).permute(1, 0, 2).contiguous()


def roi_select(rect, on=None):
    """Image block: on (B,V,Q) int32, rect (B,V,Q,4) -> rois (B*Q,5), view int8 (B*Q), member uint8, keep uint8,
    on_img float32 (B,Q).  Point block (on None): rect (B,Q,4) -> rois (B*Q,5)."""
    _dev(rect)
    dev = rect.device
    if on is None:
        B, Q = rect.shape[:2]
        rois = torch.empty((B * Q, 5), dtype=torch.float32, device=dev)
        _lib.call('di_roi_select', 0, rect.data_ptr(), rois.data_ptr(), 0, 0, 0, 0, B, 0, Q, _stream())
        return rois
    B, V, Q = on.shape
    assert on.dtype == torch.int32 and on.is_contiguous() and rect.is_contiguous()
    rois = torch.empty((B * Q, 5), dtype=torch.float32, device=dev)
    view = torch.empty((B * Q,), dtype=torch.int8, device=dev)
    member = torch.empty((B * Q,), dtype=torch.uint8, device=dev)
    keep = torch.empty((B * Q,), dtype=torch.uint8, device=dev)
    on_img = torch.empty((B, Q), dtype=torch.float32, device=dev)
    _lib.call('di_roi_select', on.data_ptr(), rect.data_ptr(), rois.data_ptr(), view.data_ptr(), member.data_ptr(),
              keep.data_ptr(), on_img.data_ptr(), B, V, Q, _stream())
    return rois, view, member, keep, on_img


def v2_self_feature(c, x, B, Q, qkv=None, view=None, member=None, V=0):
    """The self branch of a ++ V2 RoI block in one launch (csrc/v2_self.hip): c = the block's constants
    (decoder_fused._block_consts_v2); image block: x = the block's input tokens, qkv = ((B*Q,256) [q | k] rows, (B,128,Qp)
    V^T), view int8 / member uint8 of ops.roi_select; point block (qkv None): x = norm1(x + attention).  -> (B*Q,128) float32,
    already times `self_scale`."""
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape == (B * Q, 128)
    out = torch.empty((B * Q, 128), dtype=torch.float32, device=x.device)
    image = qkv is not None
    ptr = lambda t: 0 if t is None else t.data_ptr()
    qk, vt = qkv if image else (None, None)
    if image:
        assert qk.dtype == vt.dtype == torch.float32 and qk.is_contiguous() and vt.is_contiguous() and qk.shape == (B * Q, 256)
        assert vt.shape[:2] == (B, 128) and view.dtype == torch.int8 and member.dtype == torch.uint8
    hidden = c['sw1'].shape[0]
    for k in ('sw1', 'sb1', 'sw2', 'sb2', 'wo', 'bo'):
        assert c[k].dtype == torch.float32 and c[k].is_contiguous()
    _lib.call('di_v2_self_feature', ptr(qk), ptr(vt), x.data_ptr(), ptr(view), ptr(member), c['wo'].data_ptr(), c['bo'].data_ptr(),
              c['n1'][0].data_ptr(), c['n1'][1].data_ptr(), float(c['eps'][0]), float(c['scale']), c['sw1'].data_ptr(),
              c['sb1'].data_ptr(), c['sw2'].data_ptr(), c['sb2'].data_ptr(), c['sn'][0].data_ptr(), c['sn'][1].data_ptr(),
              float(c['sn_eps']), c['self_scale'].data_ptr(), out.data_ptr(), B, Q, int(vt.shape[2]) if image else Q, int(V),
              int(hidden), int(image), _stream())
    return out


def query_init(bev, top, ce_w, ce_b, pe):
    """bev (B,128,H,W) channels-last fp16, top (B,Q) int64 flattened (class, cell) picks, ce_w (128,ncls) / ce_b (128)
    float32, pe = float32 (w1 (128,2), b1, w2 (128,128), b2) -> feat (B*Q,128), pos_embed (B*Q,128) float32, pos
    (B,Q,2) float32, labels (B,Q) int64."""
    _dev(bev, top)
    B, C, H, W = bev.shape
    Q = top.shape[1]
    assert C == 128 and _is_cl(bev) and bev.dtype == torch.float16 and top.dtype == torch.int64 and top.is_contiguous()
    assert ce_w.dtype == torch.float32 and ce_w.is_contiguous() and ce_b.dtype == torch.float32
    dev = bev.device
    feat = torch.empty((B * Q, 128), dtype=torch.float32, device=dev)
    pemb = torch.empty((B * Q, 128), dtype=torch.float32, device=dev)
    pos = torch.empty((B, Q, 2), dtype=torch.float32, device=dev)
    labels = torch.empty((B, Q), dtype=torch.int64, device=dev)
    _lib.call('di_query_init', bev.data_ptr(), top.data_ptr(), ce_w.data_ptr(), ce_b.data_ptr(), pe[0].data_ptr(),
              pe[1].data_ptr(), pe[2].data_ptr(), pe[3].data_ptr(), feat.data_ptr(), pemb.data_ptr(), pos.data_ptr(),
              labels.data_ptr(), B, Q, H, W, ce_w.shape[1], _stream())
    return feat, pemb, pos, labels


def pack_linear(w):
    """(N,K) float32 weight (N multiple of 16, K of 128) -> fp16 (N/16, K/128, 4, 2, 64, 8): the hi / lo split of
    `split_hi_lo` in MFMA fragment order - block (t, c) holds, per k-step kk and half h, the 64 lanes' 8 halfs
    W_h[16t + i][128c + 32kk + 8g + j] at lane 16g + i (csrc/token32.hip linear_issue: every load instruction of a wave
    reads one contiguous KiB)."""
    N, K = w.shape
    assert N % 16 == 0 and K % 128 == 0
    hi, lo = split_hi_lo(w)
    x = torch.stack([hi, lo])                                             # (h, N, K)
    x = x.view(2, N // 16, 16, K // 128, 4, 4, 8)                         # (h, t, i, c, kk, g, j)
    return x.permute(1, 3, 4, 0, 5, 2, 6).contiguous().view(N // 16, K // 128, 4, 2, 64, 8)


def pack_ksteps(w):
    """(N,K) float32 weight (N multiple of 16, K of 32) -> fp16 (N/16, K/32, 2, 64, 8): hi / lo split in k-step order -
    [tile t][k-step ks][half h][lane 16g + i][j] = W_h[16t + i][32ks + 8g + j] (csrc/token32.hip splitk_kernel)."""
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0
    hi, lo = split_hi_lo(w)
    x = torch.stack([hi, lo]).view(2, N // 16, 16, K // 32, 4, 8)          # (h, t, i, ks, g, j)
    return x.permute(1, 3, 0, 4, 2, 5).contiguous().view(N // 16, K // 32, 2, 64, 8)


def split_hi_lo(w):
    """float32 tensor -> (hi, lo) fp16 with w = hi + lo / 2048 to ~2^-22 (the low half pre-scaled out of the fp16
    subnormals): the operand form of the float32-accurate fp16 MFMA passes (csrc/cross_attn.hip)."""
    w = w.detach().float()
    hi = w.half()
    lo = ((w - hi.float()) * 2048.0).half()
    return hi.contiguous(), lo.contiguous()


def pack_kv_weight(w):
    """(256,128) float32 K/V projection weight (rows [K ; V]) -> the packed operand of `kv_project`: rows permuted so
    that MFMA row i' of tile (pt, nb) is channel 32pt + 8(i' >> 2) + 4nb + (i' & 3), then `pack_linear`."""
    r = torch.arange(256, device=w.device)
    pt, nb, ip = r // 32, (r % 32) // 16, r % 16
    return pack_linear(w.detach().float()[32 * pt + 8 * (ip // 4) + 4 * nb + (ip % 4)])


def kv_project(x_tokens, w_packed, kbias, vbias):
    """x_tokens (B,S,128) fp16 (a view of the channels-last BEV map) -> (B,S,384) fp16 = [Khi | Klo | V]
    (K = Khi + Klo / 2048 = Wk x + kbias, V = Wv x + vbias; kbias / vbias (S,128) float32; w_packed = `pack_kv_weight`)."""
    _dev(x_tokens, w_packed)
    B, S, C = x_tokens.shape
    assert C == 128 and x_tokens.dtype == torch.float16 and x_tokens.is_contiguous()
    assert w_packed.dtype == torch.float16 and w_packed.numel() == 256 * 128 * 2 and w_packed.is_contiguous()
    assert kbias.shape == (S, 128) and vbias.shape == (S, 128) and kbias.dtype == torch.float32 and kbias.is_contiguous()
    out = torch.empty((B, S, 384), dtype=torch.float16, device=x_tokens.device)
    _lib.call('di_kv_project_fwd', x_tokens.data_ptr(), w_packed.data_ptr(), kbias.data_ptr(), vbias.contiguous().data_ptr(),
              out.data_ptr(), B, S, _stream())
    return out


def mha_decode_x(q, kx, scale, merge=True):
    """q (B,Q,128) float32 (unscaled), kx (B,S,384) from `kv_project` -> softmax(q k^T scale) v, (B*Q,128) float32; with
    merge=False -> (scratch, nrange): the partial soft-max states a `TokenProgram.combine` step merges."""
    _dev(q, kx)
    B, Q, E = q.shape
    S = kx.shape[1]
    assert E == 128 and q.dtype == torch.float32 and q.is_contiguous() and kx.shape == (B, S, 384) and kx.is_contiguous()
    nrange = int(_lib.lib().di_mha_decode_x_ranges(B, Q, S))
    scratch = torch.empty(B * 8 * Q * nrange * 18, dtype=torch.float32, device=q.device)
    out = torch.empty((B * Q, 128), dtype=torch.float32, device=q.device) if merge else None
    _lib.call('di_mha_decode_x_fwd', q.data_ptr(), kx.data_ptr(), scratch.data_ptr(), 0 if out is None else out.data_ptr(),
              B, Q, S, float(scale), _stream())
    return out if merge else (scratch, nrange)
