"""CPU tests of host-side logic added in round 2 (no GPU, no HIP library calls)."""
import math

import torch

from deepinteraction_amd import ops
from deepinteraction_amd.autograd import PixelLinear, WideLinear


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


def test_wide_linear_matches_autograd_of_linear():
    """`WideLinear` (DynamicConv's 128 -> 32 768 parameter generator in the training path): the slab-batched input
    gradient, the weight and bias gradients against autograd of F.linear, float64; also an output width the slabs do not
    divide (single-GEMM path)."""
    g = torch.Generator().manual_seed(4)
    for n, k in ((20, 32768), (7, 1000)):
        x = torch.randn(n, 128, generator=g, dtype=torch.float64, requires_grad=True)
        W = torch.randn(k, 128, generator=g, dtype=torch.float64, requires_grad=True)
        b = torch.randn(k, generator=g, dtype=torch.float64, requires_grad=True)
        gy = torch.randn(n, k, generator=g, dtype=torch.float64)
        y = WideLinear.apply(x, W, b)
        y.backward(gy)
        got = (x.grad.clone(), W.grad.clone(), b.grad.clone())
        x.grad = W.grad = b.grad = None
        y2 = torch.nn.functional.linear(x, W, b)
        y2.backward(gy)
        assert torch.equal(y, y2)
        for a, c in zip(got, (x.grad, W.grad, b.grad)):
            assert float((a - c).abs().max()) <= 1e-10 * float(c.abs().max())


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


def test_graph_input_arena_layout():
    """GraphedHotPath keeps every per-sample input in one allocation: the views keep shape, strides (channels-last maps
    stay channels-last), dtype and content; a second arena with the same layout (a prepared record) switches all of them
    with ONE copy."""
    import torch
    from deepinteraction_amd.graphed import GraphedHotPath
    g = GraphedHotPath.__new__(GraphedHotPath)
    g.glue = None
    gen = torch.Generator().manual_seed(0)
    g.img_feats = torch.randn(2, 8, 5, 7, generator=gen).half().contiguous(memory_format=torch.channels_last)
    g.pts_feats = [torch.randn(1, 4, 6, 6, generator=gen), torch.randn(1, 4, 3, 3, generator=gen).contiguous(memory_format=torch.channels_last)]
    g.pts = [torch.randn(11, 5, generator=gen)]
    g.pillars = torch.randn(9, 20, 5, generator=gen)
    g.pillar_coors = torch.randint(0, 50, (9, 4), generator=gen, dtype=torch.int32)
    g.pillars_num_points = torch.randint(0, 20, (9,), generator=gen, dtype=torch.int32)
    before = [t.clone() for t in g._input_list()]
    strides = [t.stride() for t in g._input_list()]
    g._build_arena()
    after = g._input_list()
    assert isinstance(g.pts_feats, list) and len(g.pts_feats) == 2 and g.pillars_num_points.dtype == torch.int32
    base = g._arena.untyped_storage().data_ptr()
    for b, a, st in zip(before, after, strides):
        assert torch.equal(a, b) and a.stride() == st and a.dtype == b.dtype
        assert a.untyped_storage().data_ptr() == base and a.data_ptr() % 256 == base % 256
    assert g.img_feats.is_contiguous(memory_format=torch.channels_last)
    # a record arena: same layout, other content; one copy switches every view
    other = torch.empty_like(g._arena)
    views = g._arena_views(other)
    for v, b in zip(views, before):
        v.copy_(b * 2 if b.is_floating_point() else b + 1)
    g._arena.copy_(other)
    for a, b in zip(g._input_list(), before):
        assert torch.equal(a, b * 2 if b.is_floating_point() else b + 1)


def test_loss_scaler_skips_an_overflow_step_and_backs_off():
    """`train_step.LossScaler` (mixed-precision training; round-4 advice): GradScaler's rules with the overflow flag consumed by
    the fused AdamW - a step whose gradients hold inf / NaN changes neither the parameters nor the optimizer state and halves
    the scale; clean steps unscale the gradients exactly and grow the scale after `growth_interval` of them."""
    from deepinteraction_amd.train_step import LossScaler
    p = torch.nn.Parameter(torch.ones(4))
    opt = torch.optim.AdamW([p], lr=0.1, fused=True)
    sc = LossScaler('cpu', init_scale=8.0, growth_interval=2)

    def step(grad):
        p.grad = grad.clone()
        sc.unscale_([p.grad])
        opt.grad_scale, opt.found_inf = None, sc.found_inf
        opt.step()
        sc.update()
    step(torch.tensor([8.0, 16.0, float('inf'), 1.0]))
    assert torch.equal(p.detach(), torch.ones(4)) and float(sc.scale) == 4.0 and float(sc.skipped) == 1.0
    assert float(opt.state[p]['step']) == 0.0
    step(torch.tensor([4.0, 8.0, -4.0, 0.0]))
    assert torch.equal(p.grad, torch.tensor([1.0, 2.0, -1.0, 0.0]))          # unscaled exactly (powers of two)
    assert float(opt.state[p]['step']) == 1.0 and bool((p.detach()[:3] != 1.0).all()) and float(sc.scale) == 4.0
    step(torch.tensor([4.0, 4.0, 4.0, 4.0]))
    assert float(sc.scale) == 8.0 and float(sc.skipped) == 1.0                # two clean steps: the scale grows
    step(torch.tensor([float('nan'), 0.0, 0.0, 0.0]))
    assert float(sc.scale) == 4.0 and float(sc.skipped) == 2.0 and torch.isfinite(p).all()


def test_lane_launchers_run_one_callable_per_lane_thread_and_hand_exceptions_over():
    """graphed.LaneLaunchers (one launching host thread per in-flight lane): every callable runs on ITS lane's thread, `run`
    returns after all of them, an exception raised on a lane is re-raised by `run`, and the pool keeps working after it."""
    import threading
    import pytest
    from deepinteraction_amd.graphed import LaneLaunchers
    pool = LaneLaunchers([None, None, None])
    seen = [[] for _ in range(3)]
    main = threading.get_ident()

    def job(l):
        return lambda: seen[l].append(threading.get_ident())
    for _ in range(5):
        pool.run([job(0), job(1), job(2)])
    assert all(len(s) == 5 and len(set(s)) == 1 and s[0] != main for s in seen)
    assert len({s[0] for s in seen}) == 3                      # three different threads, the same one per lane every time

    def boom():
        raise ValueError('lane 1 failed')
    with pytest.raises(ValueError, match='lane 1 failed'):
        pool.run([job(0), boom, job(2)])
    assert len(seen[0]) == 6 and len(seen[2]) == 6             # the other lanes still ran their callable
    pool.run([job(0), job(1), job(2)])
    assert [len(s) for s in seen] == [7, 6, 7]
    grads = []
    pool.run([lambda: grads.append(torch.is_grad_enabled())] * 3)
    assert grads == [False, False, False]                      # the lanes issue inference work
    pool.close()
    with pytest.raises(AssertionError):
        pool.run([job(0)])


def test_msda_torch_formulation_matches_mmcv_core():
    """`MultiScaleDeformableAttention._core_torch` (the path of configurations the kernels do not cover: mmcv's default
    num_levels=4, other point counts) against the oracle's restatement of mmcv's `multi_scale_deformable_attn_pytorch`."""
    import torch
    from deepinteraction_amd.mmdet3d_plugin.models.utils.transformer_bricks import MultiScaleDeformableAttention
    from oracle import thirdparty as tp
    torch.manual_seed(0)
    for L, P in ((4, 4), (2, 8), (1, 2)):
        m = MultiScaleDeformableAttention(128, num_heads=8, num_levels=L, num_points=P, batch_first=True).eval()
        assert not m.kernels_cover()
        shapes = [(12, 20), (6, 10), (3, 5), (2, 3)][:L]
        S = sum(h * w for h, w in shapes)
        bs, nq = 2, 50
        v, off, lg = torch.randn(bs, S, 128), torch.randn(bs, nq, 8 * L * P * 2) * 2, torch.randn(bs, nq, 8 * L * P)
        ref = torch.rand(bs, nq, L, 2)
        got = m._core_torch(v, off, lg, ref, shapes)
        w = lg.view(bs, nq, 8, L * P).softmax(-1).view(bs, nq, 8, L, P)
        norm = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float32)
        loc = ref[:, :, None, :, None, :] + off.view(bs, nq, 8, L, P, 2) / norm[None, None, None, :, None, :]
        want = tp.ms_deform_attn_core(v.view(bs, S, 8, 16), shapes, loc, w)
        assert (got - want).abs().max().item() <= 1e-5
    assert MultiScaleDeformableAttention(128, num_heads=8, num_levels=2, num_points=4).kernels_cover()
