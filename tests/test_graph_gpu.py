"""GPU: the captured hipGraph of the whole forward (deepinteraction_amd/graphed.py) replays to the same
results as the eager forward, and `load()` of another sample (other points, pillars, metas; fewer
pillars than the captured capacity) is equivalent to an eager forward on that sample."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from deepinteraction_amd import synth
from deepinteraction_amd.configs import decoder_cfg
from deepinteraction_amd.graphed import GraphedHotPath
from deepinteraction_amd.mmdet3d_plugin import DeepInteractionDecoder, DeepInteractionEncoder


def _to_device(inp, dtype):
    pm = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
    pm['pts'] = [p.cuda() for p in inp['pts_metas']['pts']]
    return dict(img_feats=inp['img_feats'].cuda().to(dtype).contiguous(memory_format=torch.channels_last),
                pts_feats=inp['pts_feats'].cuda().to(dtype).contiguous(memory_format=torch.channels_last),
                img_metas=inp['img_metas'], pts_metas=pm)


def _eager(enc, dec, d):
    with torch.no_grad():
        img, pts = enc(d['img_feats'], d['pts_feats'], d['img_metas'], dict(d['pts_metas']))
        return dec(pts, img, d['img_metas'])[0][0]


def _same(a, b):
    for key in a:
        x, y = a[key].float(), b[key].float()
        assert x.shape == y.shape, key
        # identical kernels on identical inputs; only atomics-free code paths -> exact
        assert torch.equal(x, y), (key, (x - y).abs().max().item())


def test_graph_replay_matches_eager_and_load_switches_sample():
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    shape = synth.SHAPE_TINY
    # MIOpen's default solver for the 36x36 heat-map convolution accumulates with atomics (run-to-run
    # ulp noise that flips near-tie proposals of a random-init head); ask for its deterministic solvers
    torch.backends.cudnn.deterministic = True
    torch.manual_seed(3)
    enc = DeepInteractionEncoder(2, shape['c_img'], shape['c_pts'], 128).cuda().half().eval()
    dec = DeepInteractionDecoder(**decoder_cfg(bev=shape['bev_hw'][0], num_proposals=50)).cuda().half().eval()
    # sample A has the larger point cloud -> it defines the captured capacity
    a = _to_device(synth.make_inputs(1, shape, seed=1), torch.float16)
    small = dict(shape, n_points=shape['n_points'] // 2)
    b = _to_device(synth.make_inputs(1, small, seed=2), torch.float16)
    assert b['pts_metas']['pillars'].shape[0] < a['pts_metas']['pillars'].shape[0]
    for _ in range(2):                 # MIOpen / hipBLASLt pick their kernels on the first calls
        _eager(enc, dec, a), _eager(enc, dec, b)
    ref_a = {k: v.clone() for k, v in _eager(enc, dec, a).items()}
    ref_b = {k: v.clone() for k, v in _eager(enc, dec, b).items()}
    g = GraphedHotPath(enc, dec, a)
    _same(g()[0][0], ref_a)
    _same(g()[0][0], ref_a)            # replay twice: no state leaks between replays
    g.load(b)
    _same(g()[0][0], ref_b)
    g.load(a)
    _same(g()[0][0], ref_a)
    # load_raw: straight from a raw batch with NCHW-contiguous maps (what a backbone hands over) into the captured buffers
    raw_b = dict(b, img_feats=b['img_feats'].contiguous(), pts_feats=b['pts_feats'].contiguous())
    assert not raw_b['img_feats'].is_contiguous(memory_format=torch.channels_last)
    g.load_raw(raw_b)
    _same(g()[0][0], ref_b)
    g.load_raw(a)
    _same(g()[0][0], ref_a)


def _to_device_pp(inp, dtype):
    pm = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
    pm['pts'] = [p.cuda() for p in inp['pts_metas']['pts']]
    cl = lambda t: t.cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    return dict(img_feats=[cl(f) for f in inp['img_feats']], pts_feats=[cl(f) for f in inp['pts_feats']],
                img_metas=inp['img_metas'], pts_metas=pm)


def test_graph_replay_of_the_plusplus_forward():
    """DeepInteraction++ neck + head behind the same capture: lists of levels as static inputs, the polar ray
    geometry refreshed in place by `load()` (other cameras' augmentation record, fewer pillars)."""
    from deepinteraction_amd import configs
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionPlusPlusDecoder, FusionTransformerv4
    shape = synth.SHAPE_PP_TINY
    torch.backends.cudnn.deterministic = True
    torch.manual_seed(5)
    enc = FusionTransformerv4(**configs.encoder_pp_cfg(shape['c_img'], shape['c_pts'])).cuda().half().eval()
    dec = DeepInteractionPlusPlusDecoder(**decoder_cfg(bev=shape['bev_hw'][0], num_proposals=50)).cuda().half().eval()
    g_ = torch.Generator().manual_seed(1)
    with torch.no_grad():               # off the zero init of the deformable-attention projections
        for m in enc.modules():
            if hasattr(m, 'sampling_offsets'):
                m.sampling_offsets.weight.add_((torch.randn(m.sampling_offsets.weight.shape, generator=g_) * 0.05).cuda().half())
                m.attention_weights.weight.add_((torch.randn(m.attention_weights.weight.shape, generator=g_) * 0.05).cuda().half())
    a = _to_device_pp(synth.make_inputs_pp(1, shape, seed=1), torch.float16)
    small = dict(shape, n_points=shape['n_points'] // 2)
    b = _to_device_pp(synth.make_inputs_pp(1, small, seed=2, aug=synth.example_aug(4)), torch.float16)
    assert b['pts_metas']['pillars'].shape[0] < a['pts_metas']['pillars'].shape[0]
    for _ in range(2):
        _eager(enc, dec, a), _eager(enc, dec, b)
    ref_a = {k: v.clone() for k, v in _eager(enc, dec, a).items()}
    ref_b = {k: v.clone() for k, v in _eager(enc, dec, b).items()}
    assert not all(torch.equal(ref_a[k], ref_b[k]) for k in ref_a)
    g = GraphedHotPath(enc, dec, a)
    _same(g()[0][0], ref_a)
    g.load(b)
    _same(g()[0][0], ref_b)
    g.load(a)
    _same(g()[0][0], ref_a)


def test_topk_replays_from_a_graph():
    """The radix-select top-k replayed from a hipGraph on changing inputs (its first version kept one global histogram
    - atomicAdd after a memset node / plain-store clear - and went wrong from the SECOND replay on; eager was fine)."""
    from deepinteraction_amd import ops
    B, N, k = 2, 324000, 200
    g0 = torch.Generator(device='cuda').manual_seed(0)
    xs = []
    for _ in range(3):
        x = torch.rand(B, N, device='cuda', generator=g0)
        x[torch.rand(B, N, device='cuda', generator=g0) < 0.8] = 0
        xs.append(x)
    ref = [ops.topk(x, k).clone() for x in xs]
    static = xs[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.topk(static, k)
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        idx = ops.topk(static, k)
    for it in range(7):
        static.copy_(xs[it % 3])
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(idx, ref[it % 3]), it


def test_many_replays_over_a_pool_match_eager_at_the_benched_shape():
    """bench.py's per-step sequence - load(record of the next pool sample), replay - at shape R, fp16, Q = 200: every
    replay bit-identical to the eager forward of that sample (proposals, labels, all outputs)."""
    from deepinteraction_amd import harness, parallel
    shape = synth.SHAPE_R
    enc, dec = harness.build_models(shape, 200, torch.float16, 'cuda')
    pool = [harness.to_device(synth.make_inputs(1, shape, seed=parallel.sample_seed(i)), 'cuda', torch.float16)
            for i in range(3)]
    with torch.no_grad():
        eager = []
        for d in pool:
            _, out = harness.forward(enc, dec, d)
            torch.cuda.synchronize()
            eager.append(({k: v.clone() for k, v in out[0][0].items()}, dec.top_proposals.clone(), dec.query_labels.clone()))
        cap = max(range(3), key=lambda i: int(pool[i]['pts_metas']['pillars'].shape[0]))
        g = GraphedHotPath(enc, dec, pool[cap])
        recs = [g.prepare(d) for d in pool]
        for it in range(7):
            i = it % 3
            g.load(recs[i])
            out = g()[0][0]
            torch.cuda.synchronize()
            ref, top, labels = eager[i]
            assert torch.equal(dec.top_proposals, top) and torch.equal(dec.query_labels, labels), (it, i)
            _same(out, ref)
    assert g.num_nodes() is None or g.num_nodes() < 200


def test_two_forwards_in_flight_match_eager_at_the_benched_shape():
    """bench.py's default step: TWO independent captured forwards (own static buffers, shared modules and weight
    caches), each load()ed with its own sample and replayed on its own stream at the same time - every replay
    bit-identical to the eager forward of its sample."""
    from deepinteraction_amd import harness, parallel
    shape = synth.SHAPE_R
    enc, dec = harness.build_models(shape, 200, torch.float16, 'cuda')
    pool = [harness.to_device(synth.make_inputs(1, shape, seed=parallel.sample_seed(i)), 'cuda', torch.float16)
            for i in range(3)]
    with torch.no_grad():
        eager = []
        for d in pool:
            _, out = harness.forward(enc, dec, d)
            torch.cuda.synchronize()
            eager.append({k: v.clone() for k, v in out[0][0].items()})
        cap = max(range(3), key=lambda i: int(pool[i]['pts_metas']['pillars'].shape[0]))
        graphs = [GraphedHotPath(enc, dec, pool[cap]) for _ in range(2)]
        recs = [graphs[0].prepare(d) for d in pool]
        lanes = [torch.cuda.Stream() for _ in graphs]
        for lane in lanes:
            lane.wait_stream(torch.cuda.current_stream())
        it = 0
        for rnd in range(5):
            used, outs = [], []
            for g, lane in zip(graphs, lanes):
                with torch.cuda.stream(lane):
                    g.load(recs[it % 3])
                    outs.append(g()[0][0])
                used.append(it % 3)
                it += 1
            torch.cuda.synchronize()
            for i, out in zip(used, outs):
                _same(out, eager[i])


def test_point_glue_rebuilds_pts_metas_and_the_graph_can_start_from_points():
    """Detector glue (reference detectors/deepinteraction.py:120-171): `PointGlue` on the raw points reproduces the
    `pts_metas` the inputs carry (pillars in first-come order, coordinates, counts: bit-exact), and a captured forward
    that voxelises inside every replay (`GraphedHotPath(glue=...)`, capacity-sized buffers, no host sync) equals the
    forward on given pillars."""
    from deepinteraction_amd import harness, parallel
    from deepinteraction_amd.mmdet3d_plugin import PointGlue
    shape = synth.SHAPE_R
    Hb, Wb = shape['bev_hw']
    rng = list(synth.PC_RANGE)
    glue = PointGlue(dict(max_num_points=20, max_voxels=(30000, 60000), point_cloud_range=rng,
                          voxel_size=[(rng[3] - rng[0]) / Wb, (rng[4] - rng[1]) / Hb, rng[5] - rng[2]])).eval()
    pool = [harness.to_device(synth.make_inputs(1, shape, seed=parallel.sample_seed(i)), 'cuda', torch.float16)
            for i in range(2)]
    for d in pool:
        pm, ref = glue(d['pts_metas']['pts']), d['pts_metas']
        assert torch.equal(pm['pillars'], ref['pillars'])
        assert torch.equal(pm['pillars_num_points'], ref['pillars_num_points'])
        assert torch.equal(pm['pillar_coors'], ref['pillar_coors'])
        n = ref['pillars_num_points'].clamp(min=1).float().unsqueeze(-1)
        assert torch.allclose(pm['pillar_center'], ref['pillars'].sum(1) / n)
    enc, dec = harness.build_models(shape, 200, torch.float16, 'cuda')
    with torch.no_grad():
        eager = []
        for d in pool:
            _, out = harness.forward(enc, dec, d)
            torch.cuda.synchronize()
            eager.append({k: v.clone() for k, v in out[0][0].items()})
        cap = max(range(2), key=lambda i: int(pool[i]['pts_metas']['pts'][0].shape[0]))
        g = GraphedHotPath(enc, dec, pool[cap], glue=glue)
        recs = [g.prepare(d) for d in pool]
        for it in range(4):
            g.load(recs[it % 2])
            out = g()[0][0]
            torch.cuda.synchronize()
            _same(out, eager[it % 2])


def test_the_graph_can_start_from_the_camera_images():
    """`GraphedHotPath(image_net=...)`: the frozen ResNet-50 + FPN stand-in (`FrozenResNetFPN`, torch / MIOpen) runs inside
    every replay on the static image buffer; the result equals the forward on the maps the same network produces
    eagerly, and `load()` switches the images with the rest of the sample."""
    from deepinteraction_amd.mmdet3d_plugin import FrozenResNetFPN
    shape = synth.SHAPE_TINY
    torch.backends.cudnn.deterministic = True
    torch.manual_seed(4)
    enc = DeepInteractionEncoder(2, shape['c_img'], shape['c_pts'], 128).cuda().half().eval()
    dec = DeepInteractionDecoder(**decoder_cfg(bev=shape['bev_hw'][0], num_proposals=50)).cuda().half().eval()
    net = FrozenResNetFPN(out_channels=shape['c_img'], levels=(0,))
    net.load_mmdet_state(*net.synthetic_state(1)).cuda()
    H, W = shape['input_shape']
    pool, want = [], []
    for i in range(2):
        d = _to_device(synth.make_inputs(1, shape, seed=5 + i), torch.float16)
        cams = torch.randn(6, 3, H, W, generator=torch.Generator().manual_seed(20 + i))
        d['images'] = cams.cuda().half().contiguous(memory_format=torch.channels_last)
        for _ in range(2):
            d['img_feats'] = net(d['images'])[0]
        assert d['img_feats'].shape == (6, shape['c_img'], *shape['img_hw'])
        pool.append(d)
    with torch.no_grad():
        for d in pool:
            for _ in range(2):
                img, pts = enc(d['img_feats'], d['pts_feats'], d['img_metas'], dict(d['pts_metas']))
            want.append((img.clone(), [p.clone() for p in pts], {k: v.clone() for k, v in dec(pts, img, d['img_metas'])[0][0].items()}))
    cap = max(range(2), key=lambda i: int(pool[i]['pts_metas']['pillars'].shape[0]))
    g = GraphedHotPath(enc, dec, pool[cap], image_net=net)
    recs = [g.prepare(d) for d in pool]
    for it in range(4):
        g.load(recs[it % 2])
        out = g()[0][0]
        torch.cuda.synchronize()
        img, pts = g.enc_out
        wimg, wpts, wout = want[it % 2]
        scale = float(wimg.float().abs().max())
        assert float((img.float() - wimg.float()).abs().max()) <= 2e-3 * scale
        for p, wp in zip(pts, wpts):
            assert float((p.float() - wp.float()).abs().max()) <= 2e-3 * float(wp.float().abs().max())
        if torch.equal(img, wimg) and all(torch.equal(p, wp) for p, wp in zip(pts, wpts)):
            _same(out, wout)            # same kernels on the same maps: the head's outputs are then identical too


@pytest.mark.parametrize('n_lanes', [2, 4])
def test_resident_handover_captures_match_eager_at_the_benched_shape(n_lanes):
    """bench.py's default step since round 5 (`--handover resident`, `bench.graphed_steps`): one captured forward per pool
    sample, captured from the prepared (padded) record, lane l replaying the captures of samples l, l + L, ... in turn on its
    stream - every replay bit-identical to the eager forward of ITS sample; then the copying hand-over on the same captures
    (`step_copy`, the `copy_handover` figure) and `load_geometry` after a producer-style in-place write of the inputs."""
    import bench
    from deepinteraction_amd import harness, parallel
    shape = synth.SHAPE_R
    enc, dec = harness.build_models(shape, 200, torch.float16, 'cuda')
    pool = [harness.to_device(synth.make_inputs(1, shape, seed=parallel.sample_seed(i)), 'cuda', torch.float16)
            for i in range(4)]
    with torch.no_grad():
        eager = []
        for d in pool:
            _, out = harness.forward(enc, dec, d)
            torch.cuda.synchronize()
            eager.append({k: v.clone() for k, v in out[0][0].items()})
        assert not all(torch.equal(eager[0][k], eager[1][k]) for k in eager[0])
        cap = max(range(4), key=lambda i: int(pool[i]['pts_metas']['pillars'].shape[0]))
        step, step_copy, step1, graphs, records, g = bench.graphed_steps(
            lambda inp, ov: GraphedHotPath(enc, dec, inp, overlap=ov), pool, cap, n_lanes, True)
        assert len(graphs) == 4 and g is graphs[-1]
        if n_lanes == 4:
            # the bench's default: four captures replayed side by side, 8 rounds of 3 steps; the outputs are wiped between the
            # rounds, so every round's replays must have produced them again - under the others' traffic
            for rnd in range(8):
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                for i, gi in enumerate(graphs):
                    _same(gi.out[0][0], eager[i])
                    for v in gi.out[0][0].values():
                        v.zero_()
            return
        for rnd in range(4):
            step()
        torch.cuda.synchronize()
        for i, gi in enumerate(graphs):                      # every capture still holds (and has just recomputed) its sample
            _same(gi.out[0][0], eager[i])
        for i in range(5):
            step1()
        torch.cuda.synchronize()
        _same(graphs[0].out[0][0], eager[0])
        # zero-copy hand-over proper: a producer writes sample 1 INTO capture 3's static buffers, geometry refreshed in place
        r, gi = records[1], graphs[3]
        gi.img_feats.copy_(r.img_feats); gi.pts_feats.copy_(r.pts_feats)
        for dst, src in zip(gi.pts, r.pts):
            dst.copy_(src)
        gi.pillars.copy_(r.pillars); gi.pillar_coors.copy_(r.pillar_coors); gi.pillars_num_points.copy_(r.pillars_num_points)
        gi.load_geometry(pool[1]['img_metas'])
        _same(gi()[0][0], eager[1])
        torch.cuda.synchronize()
        # the copying hand-over of rounds 2-4 on the lanes' first captures
        for rnd in range(3):
            step_copy()
        torch.cuda.synchronize()
        _same(graphs[0].out[0][0], eager[(2 * 2) % 4])       # lane 0 loaded records 0, 2, 4
        _same(graphs[1].out[0][0], eager[(2 * 2 + 1) % 4])


def test_inflight_lanes_serving_form():
    """graphed.InflightLanes: four lanes (single-stream captures, own streams, own launching threads), records loaded per lane,
    replays issued side by side - every lane's outputs bit-identical to the eager forward of the sample it was given, over
    several rounds with the samples rotating through the lanes."""
    from deepinteraction_amd import harness, parallel
    from deepinteraction_amd.graphed import InflightLanes
    shape = synth.SHAPE_R
    enc, dec = harness.build_models(shape, 200, torch.float16, 'cuda')
    pool = [harness.to_device(synth.make_inputs(1, shape, seed=parallel.sample_seed(i)), 'cuda', torch.float16)
            for i in range(5)]
    with torch.no_grad():
        eager = []
        for d in pool:
            _, out = harness.forward(enc, dec, d)
            torch.cuda.synchronize()
            eager.append({k: v.clone() for k, v in out[0][0].items()})
        cap = max(range(5), key=lambda i: int(pool[i]['pts_metas']['pillars'].shape[0]))
        lanes = InflightLanes(enc, dec, pool[cap], n_lanes=4)
        recs = [lanes.prepare(d) for d in pool]
        for rnd in range(5):
            given = [(rnd + l) % 5 for l in range(4)]
            for l, i in enumerate(given):
                lanes.load(l, recs[i])
            outs = lanes.replay()
            lanes.synchronize()
            for l, i in enumerate(given):
                _same(outs[l][0][0], eager[i])
        outs = lanes.replay(which=[2])                       # a single lane, from the calling thread
        lanes.synchronize()
        _same(outs[0][0][0], eager[given[2]])
        lanes.close()
