"""The data-parallel training step of the interaction hot path on synthetic nuScenes-shaped batches (BASELINE.json
configs[2] at 1 GPU, configs[3] at N GPUs), as `bench.py --mode train` runs it.

One process per GPU; every rank draws its own samples (`parallel.sample_ids`), runs encoder + decoder forward in
train() mode, the head loss against synthetic ground truth (Hungarian assignment on the host, as the reference),
backward through the HIP kernels, the bucketed gradient all-reduce over RCCL/xGMI LAUNCHED FROM BACKWARD HOOKS
(`parallel.GradientReducer`: overlapped with the rest of backward; parameters no rank touched keep `grad = None`, the
`find_unused_parameters=True` of the reference config) and AdamW with the reference's lr / weight decay / grad clip
(projects/configs/nuscenes/Fusion_0075_refactor.py:252-253).
"""
import torch

from . import det3d_compat as dc, harness, parallel, synth

TRAIN_CFG = dict(
    dataset='nuScenes',
    assigner=dict(type='HungarianAssigner3D', iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar'),
                  cls_cost=dict(type='FocalLossCost', gamma=2, alpha=0.25, weight=0.15),
                  reg_cost=dict(type='BBoxBEVL1Cost', weight=0.25), iou_cost=dict(type='IoU3DCost', weight=0.25)),
    pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[1440, 1440, 40], voxel_size=[0.075, 0.075, 0.2],
    out_size_factor=8, code_weights=[1.0] * 8 + [0.2, 0.2], point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0])


def synth_gt(seed, n=30):
    """Synthetic ground truth of one sample: `n` boxes (x, y, z, dx, dy, dz, yaw, vx, vy) and labels."""
    g = torch.Generator().manual_seed(seed)
    xy = (torch.rand(n, 2, generator=g) - 0.5) * 100
    z = torch.rand(n, 1, generator=g) * 2 - 2.5
    dims = torch.stack([torch.rand(n, generator=g) * 2 + 0.5, torch.rand(n, generator=g) * 5 + 0.5,
                        torch.rand(n, generator=g) * 2 + 0.8], 1)
    yaw = (torch.rand(n, 1, generator=g) - 0.5) * 6.28
    vel = torch.randn(n, 2, generator=g)
    return dc.LiDARBoxes(torch.cat([xy, z, dims, yaw, vel], 1)), torch.randint(0, 10, (n,), generator=g)


def half_weights_(models):
    """Mixed precision: turn the convolution / linear / attention-projection parameters of `models` into fp16 IN PLACE and
    return (those parameters, their float32 master copies).  Normalisation layers stay float32."""
    nn = torch.nn
    half, master = [], []
    for m in models:
        for mod in m.modules():
            if isinstance(mod, (nn.Conv1d, nn.Conv2d, nn.Linear, nn.MultiheadAttention)):
                for p in mod.parameters(recurse=False):
                    if p.dtype == torch.float32:
                        master.append(torch.nn.Parameter(p.detach().clone()))
                        p.data = p.data.half()
                        half.append(p)
    return half, master


class LossScaler:
    """Dynamic loss scaling for the mixed-precision step, with `torch.amp.GradScaler`'s rules (scale * backoff on a step whose
    gradients hold inf / NaN - that step is SKIPPED - and * growth after `growth_interval` clean steps) but no host round trip:
    the overflow flag stays on the device and the fused AdamW consumes it (`optimizer.found_inf`), so the step can sit between
    two hipGraph replays.  fp16 activation gradients flush below 6e-8 and overflow above 65504; the reference trains this
    configuration in float32 (no scaler needed), or in fp16 through mmcv's Fp16OptimizerHook with a loss scale."""

    def __init__(self, device, init_scale=2.0 ** 10, growth_factor=2.0, backoff_factor=0.5, growth_interval=200):
        dev = torch.device(device)
        self.scale = torch.full((), float(init_scale), dtype=torch.float32, device=dev)
        self.inv_scale = torch.empty_like(self.scale)
        self.growth_tracker = torch.zeros((), dtype=torch.int32, device=dev)
        self.found_inf = torch.zeros((), dtype=torch.float32, device=dev)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.skipped = torch.zeros((), dtype=torch.float32, device=dev)     # steps skipped so far (read by tests / bench)

    def unscale_(self, grads):
        """grads /= scale in place (float32 tensors); `found_inf` = 1 when any of them holds inf / NaN."""
        self.found_inf.zero_()
        torch.reciprocal(self.scale, out=self.inv_scale)
        if grads:
            torch._amp_foreach_non_finite_check_and_unscale_(list(grads), self.found_inf, self.inv_scale)

    def update(self):
        self.skipped += self.found_inf
        torch._amp_update_scale_(self.scale, self.growth_tracker, self.found_inf, self.growth_factor, self.backoff_factor,
                                 self.growth_interval)


def trainable_parameters(enc, dec):
    """The parameter list the step optimises and the gradient reducer buckets: every parameter of neck and head, in
    registration order (the reducer buckets them in reverse: the order backward produces their gradients)."""
    return [p for m in (enc, dec) for p in m.parameters()]


def synthetic_backward(named_params, rank, step, unused=()):
    """Host-only stand-in for a backward pass (bench.py --dry-run --mode train, tests/test_parallel.py): every parameter
    whose name does not start with a prefix in `unused` receives the gradient w * ones through autograd - so the reducer's
    post-accumulate hooks fire as in a real step - with w = a number that depends on (rank, step, parameter index).
    Returns {name: w} of the parameters this rank touched."""
    loss, ws = 0.0, {}
    for i, (n, p) in enumerate(named_params):
        if not p.requires_grad or any(n.startswith(u) for u in unused):
            continue
        w = float((rank + 1) * 0.5 + (step + 1) * 0.125 + (i % 7) * 0.03125)      # exact in float32
        ws[n] = w
        loss = loss + (p * w).sum()
    loss.backward()
    return ws


class Trainer:
    def __init__(self, shape, num_proposals, device, world, batch=1, pool=2, rank=0, seed=0, amp=None, model='v1'):
        """model: 'v1' = Fusion_0075_refactor (BASELINE configs[2] / [3]); 'pp' = DeepInteraction++ (configs[4]:
        FusionTransformerv4 neck + DeepInteractionPlusPlusDecoder, `shape` a SHAPE_PP dict) - the same step, eager launches."""
        import os
        self.model = model
        # mixed precision (opt-in; the reference trains this configuration in float32): the hot path under torch.autocast(fp16)
        self.amp = os.environ.get('DI_TRAIN_AMP', '0') == '1' if amp is None else bool(amp)
        bev = shape['bev_hw'][0]
        tc = dict(TRAIN_CFG, grid_size=[bev * 8, bev * 8, 40], voxel_size=[108.0 / (bev * 8)] * 2 + [0.2])
        if model == 'pp':
            self.enc, self.dec = harness.build_models_pp(shape, num_proposals, torch.float32, device, seed=seed, train_cfg=tc)
        else:
            self.enc, self.dec = harness.build_models(shape, num_proposals, torch.float32, device, seed=seed, train_cfg=tc)
        self.enc.train(), self.dec.train()                            # identical initial weights on every rank
        self.params = trainable_parameters(self.enc, self.dec)
        # Mixed precision keeps the convolution / linear / attention-projection parameters of the MODEL in fp16 and their
        # float32 MASTER copies in the optimizer (the values autocast would produce by casting the float32 weight in every
        # step - 216 cast launches forward and as many backward, 2 ms of a captured step - are the fp16 rounding of the
        # master, which one multi-tensor copy per step now writes).  Normalisation layers stay float32.
        self._half, self._master = [], []
        if self.amp and os.environ.get('DI_TRAIN_HALF_WEIGHTS', '1') != '0':
            self._half, self._master = half_weights_((self.enc, self.dec))
        self._master_grad = [torch.empty_like(m) for m in self._master]
        master_of = {id(p): m for p, m in zip(self._half, self._master)}
        self.opt_params = [master_of.get(id(p), p) for p in self.params]
        # (fused: one multi-tensor launch for the whole update instead of ~15 foreach passes - the update sits on the device's
        # critical path between two steps)
        self._fused_opt = torch.device(device).type == 'cuda'
        self.opt = torch.optim.AdamW(self.opt_params, lr=1e-4, weight_decay=0.01, fused=self._fused_opt)
        # mixed precision backpropagates fp16 gradients: dynamic loss scaling + an overflow guard in front of the update
        self.scaler = LossScaler(device) if self.amp else None
        self.world = world
        self.reducer = parallel.GradientReducer(self.params, world)
        # a small pool of device-resident batches per rank, built before the timed region (the data loader is out
        # of scope; generating 262 144 points + pillars on the host takes longer than the step)
        self.pool = []
        for i in range(pool):
            ids = parallel.sample_ids(i, batch, rank, world)
            if model == 'pp':
                d = harness.to_device_pp(synth.make_inputs_pp(batch, shape, seed=parallel.sample_seed(ids[0])), device, torch.float32)
            else:
                d = harness.to_device(synth.make_inputs(batch, shape, seed=parallel.sample_seed(ids[0])), device, torch.float32)
            self.pool.append((d, [synth_gt(parallel.sample_seed(s)) for s in ids]))
        self.i = 0

    def step(self):
        d, gts = self.pool[self.i % len(self.pool)]
        self.i += 1
        with torch.autocast('cuda', dtype=torch.float16, enabled=self.amp):
            img, pts = self.enc(d['img_feats'], d['pts_feats'], d['img_metas'], dict(d['pts_metas']))
            preds = self.dec(pts, img, d['img_metas'])
        preds = [[{k: v.float() for k, v in preds[0][0].items()}]]
        losses = self.dec.loss([g[0] for g in gts], [g[1] for g in gts], preds)
        loss = sum(v for k, v in losses.items() if k != 'matched_ious')
        self._zero_grad()
        self._backward(loss)                                          # bucket all-reduces start inside
        self.reducer.finish()
        self._update()
        return loss

    def _backward(self, loss):
        if self.scaler is None:
            loss.backward()
        else:
            loss.backward(self.scaler.scale.to(loss.dtype))           # d(scale * loss): the scale is a device scalar, no sync

    def _zero_grad(self):
        self.opt.zero_grad(set_to_none=True)
        for p in self._half:
            p.grad = None

    def _update(self):
        """Gradient clipping + AdamW (reference Fusion_0075_refactor.py:252-253) on the float32 parameters / masters."""
        if self._half:
            src, dst = [], []
            for p, m, buf in zip(self._half, self._master, self._master_grad):
                m.grad = None if p.grad is None else buf
                if p.grad is not None:
                    src.append(p.grad)
                    dst.append(buf)
            if dst:
                torch._foreach_copy_(dst, src)                       # fp16 gradients -> float32 master gradients
        live = [p for p in self.opt_params if p.grad is not None]
        skip = False
        if self.scaler is not None:
            # (after the all-reduce: every rank sees the same summed gradients, so every rank takes the same decision)
            self.scaler.unscale_([p.grad for p in live])
            if self._fused_opt:
                self.opt.grad_scale, self.opt.found_inf = None, self.scaler.found_inf   # the fused update skips itself on the device
            else:
                skip = bool(self.scaler.found_inf.item())
        if not skip:
            # (on an overflow step the clip factor is 0 or NaN: those gradients are never applied)
            torch.nn.utils.clip_grad_norm_(live, max_norm=0.1, norm_type=2)
            self.opt.step()
        if self.scaler is not None:
            self.scaler.update()
        if self._half:
            with torch.no_grad():
                torch._foreach_copy_(self._half, self._master)       # the model's fp16 weights = the rounded masters


class _HotPathModule(torch.nn.Module):
    """encoder + decoder as ONE callable of the two static feature-map tensors (what `make_graphed_callables` captures):
    points / pillars / geometry are the owner's static buffers, refreshed in place between replays."""

    def __init__(self, enc, dec, owner, amp=False):
        super().__init__()
        self.enc, self.dec, self.amp = enc, dec, amp
        self._owner = [owner]                 # (a list: not a submodule)
        self.keys = None

    def forward(self, img_feats, pts_feats):
        o = self._owner[0]
        for g in o.sample_geom:
            g.forget()
        self.dec.static_geometry = o.query_geom
        try:
            with torch.autocast('cuda', dtype=torch.float16, enabled=self.amp, cache_enabled=False):
                img, pts = self.enc(img_feats, pts_feats, o.img_metas, o._pts_metas())
                out = self.dec(pts, img, o.img_metas)[0][0]
        finally:
            self.dec.static_geometry = None
        self.keys = sorted(out)
        return tuple(out[k].float() for k in self.keys)


class GraphedTrainer(Trainer):
    """The training step with forward and backward of the hot path as TWO captured hipGraphs (`torch.cuda.
    make_graphed_callables`) around the eager head loss, whose Hungarian assignment is a host round trip as in the
    reference.  The eager step issues ~3 000 launches from Python (52 ms of wall time for 41 ms of kernels, host-bound);
    replayed, the step is bound by its kernels.  Static input buffers, padded points / pillars and in-place geometry
    refresh are the inference graph's (`graphed.GraphedHotPath`)."""

    def __init__(self, shape, num_proposals, device, world, batch=1, pool=2, rank=0, seed=0, amp=None, prepare_model=None):
        """batch: samples per rank inside ONE capture (the reference config trains with `samples_per_gpu=2`,
        Fusion_0075_refactor.py:94: BatchNorm statistics then run over both samples, as they do there).
        prepare_model: optional callable(encoder, decoder) applied before the capture (tests switch dropout off with it)."""
        super().__init__(shape, num_proposals, device, world, batch=batch, pool=pool, rank=rank, seed=seed, amp=amp)
        if prepare_model is not None:
            prepare_model(self.enc, self.dec)
        from .graphed import GraphedHotPath
        cap = max(range(len(self.pool)), key=lambda i: int(self.pool[i][0]['pts_metas']['pillars'].shape[0]))
        h = GraphedHotPath.__new__(GraphedHotPath)           # the static-buffer half of the inference graph, no capture
        h.enc, h.dec, h.glue, h.image_net, h._img_key = self.enc, self.dec, None, None, 'img_feats'
        inputs = self.pool[cap][0]
        h.img_feats, h.pts_feats = h._clone(inputs['img_feats']), h._clone(inputs['pts_feats'])
        pm = inputs['pts_metas']
        h.batch = len(inputs['img_metas'])
        h.img_metas = [dict(m) for m in inputs['img_metas']]
        h.pts = [p.clone() for p in pm['pts']]
        if h.batch == 1:
            h.pillars, h.pillar_coors, h.pillars_num_points = pm['pillars'].clone(), pm['pillar_coors'].clone(), pm['pillars_num_points'].clone()
            h.bounds = [0, h.pillars.shape[0]]
        else:
            # one fixed slice of the static pillar buffers per sample slot, as large as the largest count any pool batch has
            # in that slot (padding pillars carry num_points = 0: every kernel skips them before touching memory)
            B = h.batch
            counts = [torch.bincount(d['pts_metas']['pillar_coors'][:, 0].long(), minlength=B).cpu().tolist() for d, _ in self.pool]
            h.bounds = [0]
            for s_ in range(B):
                h.bounds.append(h.bounds[-1] + max(c[s_] for c in counts))
            total = h.bounds[-1]
            h.pillars = pm['pillars'].new_zeros((total,) + tuple(pm['pillars'].shape[1:]))
            h.pillar_coors = pm['pillar_coors'].new_zeros((total, pm['pillar_coors'].shape[1]))
            h.pillars_num_points = pm['pillars_num_points'].new_zeros((total,))
            slot = pm['pillar_coors'][:, 0].long()
            for s_ in range(B):
                lo = h.bounds[s_]
                sel = (slot == s_).nonzero().flatten()
                h.pillars[lo:lo + sel.numel()] = pm['pillars'][sel]
                h.pillar_coors[lo:lo + sel.numel()] = pm['pillar_coors'][sel]
                h.pillar_coors[lo:h.bounds[s_ + 1], 0] = s_
                h.pillars_num_points[lo:lo + sel.numel()] = pm['pillars_num_points'][sel]
        from .geometry import SampleGeometry
        from .mmdet3d_plugin.models.utils.decoder_utils import QueryGeometry
        Hi, Wi = h.img_feats.shape[-2:]
        h.sample_geom = [SampleGeometry(m, (Hi, Wi), h.img_feats.device) for m in h.img_metas]
        h.query_geom = QueryGeometry(h.img_metas, h.img_feats.device)
        h._build_arena()
        self.h = h
        self.records = [h.prepare(d) for d, _ in self.pool]
        # attention dropout of the pillar attention: the captured launches add this device word to their (baked) seed
        from . import ops
        # (registered for the duration of the captures only: the pointer is baked into the captured launches, whose lifetime is
        # this object's, and no launch outside them - another trainer, inference - ever sees it)
        self.seed_word = torch.zeros(1, dtype=torch.int64, device=h.img_feats.device)
        ops.set_i2p_seed_tensor(self.seed_word)
        self.module = _HotPathModule(self.enc, self.dec, h, amp=self.amp)
        # no garbage collection INSIDE the captures: a collected autograd graph of a warm-up iteration (Python Function nodes are
        # freed by the collector, not by reference counting) releases device memory through calls a capturing process may not make
        import gc
        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            self.graphed = torch.cuda.make_graphed_callables(self.module, (h.img_feats, h.pts_feats), allow_unused_input=True)
        finally:
            ops.set_i2p_seed_tensor(None)
            if was_enabled:
                gc.enable()

    def step(self):
        i = self.i % len(self.pool)
        self.i += 1
        _, gts = self.pool[i]
        self.h.load(self.records[i])
        self.seed_word.random_(0, 2 ** 62)                                            # a fresh dropout mask for this step
        outs = self.graphed(self.h.img_feats, self.h.pts_feats)                       # replay: forward graph
        self.dec.prepare_targets([g[0] for g in gts], [g[1] for g in gts], outs[0].device)   # host work under the replay
        preds = [[dict(zip(self.module.keys, outs))]]
        losses = self.dec.loss([g[0] for g in gts], [g[1] for g in gts], preds)
        loss = sum(v for k, v in losses.items() if k != 'matched_ious')
        self._zero_grad()
        self._backward(loss)                                                         # eager loss backward + backward graph
        self.reducer.finish()
        self._update()
        return loss


def bench(args, rank, world, device):
    """`bench.py --mode train`: returns rank 0's JSON line (a dict)."""
    pp = getattr(args, 'model', 'v1') == 'pp'
    shape = (harness.SHAPES_PP if pp else harness.SHAPES)[args.shape]
    import os
    # --amp / --train-eager of bench.py; the environment switches of round 4's first measurements still work
    amp = bool(getattr(args, 'amp', False)) or os.environ.get('DI_TRAIN_AMP', '0') == '1'
    eager = bool(getattr(args, 'train_eager', False)) or os.environ.get('DI_TRAIN_GRAPH', '1') == '0' or pp
    cls = Trainer if eager else GraphedTrainer
    extra = dict(model='pp') if pp else {}
    if rank == 0:
        import sys
        print(f'[train] {cls.__name__}, {args.batch} sample(s) per rank'
              f'{" (the reference configuration: samples_per_gpu=2)" if args.batch == 2 else ""}, '
              f'{"mixed precision" if amp else "float32"}', file=sys.stderr)
    tr = cls(shape, args.proposals, device, world, batch=args.batch, pool=max(2, min(args.pool, 2)), rank=rank, amp=amp, **extra)
    losses = []
    calibration = 0
    if tr.scaler is not None:
        # settle the dynamic loss scale BEFORE warm-up and the timed region (untimed, not counted as warm-up): a step that
        # overflows at the initial scale is skipped and halves the scale - the timed steps should be steps that update
        while calibration < 24:
            before = float(tr.scaler.skipped)
            tr.step()
            calibration += 1
            if float(tr.scaler.skipped) == before:
                break
    for _ in range(args.warmup):
        tr.step()
    elapsed = parallel.timed_region(lambda: losses.append(tr.step()), args.steps, device)
    losses = [float(l) for l in losses]
    roofline = None
    if rank == 0 and args.gpus == 1 and getattr(args, 'roofline_steps', 0) > 0:
        roofline = train_roofline(tr if eager else None, shape, args, device, world, amp, pp)
    out = dict(metric='samples/sec training step (forward + loss + backward + gradient all-reduce + AdamW)',
                value=round(parallel.throughput(args.batch, args.steps, elapsed, world), 3), unit='samples/s',
                n_gpus=args.gpus, steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 2),
                higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f16' if amp else 'f32', data='synthetic',
                config=dict(workload=f'{"Fusion_0075_plusplus (DeepInteraction++)" if pp else "Fusion_0075_refactor"} training step (shape {args.shape}): {"FusionTransformerv4 neck + ++" if pp else "MMRI encoder + MMPI"} '
                                     'decoder forward, head loss (Hungarian assignment on the host), backward, '
                                     'bucketed gradient all-reduce launched from backward hooks, AdamW + grad clip',
                            batch_per_gpu=args.batch, global_batch=args.batch * args.gpus, trainer=cls.__name__,
                            num_proposals=args.proposals, pool=len(tr.pool),
                            precision=('mixed: fp16 activations under torch.autocast incl. the fused window attention forward / '
                                       'backward; float32 master weights, BatchNorm statistics, soft-max, scatter accumulation, '
                                       'loss; dynamic loss scaling with the overflow flag kept on the device (a step with inf / NaN '
                                       'gradients is skipped by the fused AdamW)') if amp else 'float32',
                            launch='host launches' if eager else 'forward and backward of the hot path as two replayed hipGraphs '
                                                                 'around the eager loss (Hungarian assignment on the host)',
                            parallelism=f'dp{args.gpus} by sample, RCCL all-reduce of gradients only'),
                first_loss=round(losses[0], 4), last_loss=round(losses[-1], 4),
                **({} if tr.scaler is None else {'loss_scale': float(tr.scaler.scale), 'skipped_steps': int(tr.scaler.skipped),
                                                  'loss_scale_calibration_steps': calibration}))
    if roofline is not None:
        out['roofline'] = roofline
    return out


# (kernel name of ops.PROFILE, maps of n*C*H*W elements read + written per call, what it is)
_TRAIN_KERNELS = {
    'local_attn_train_bwd': (19, 'fused window-attention backward (csrc/local_attn_train.hip: row-dot + BWD_Q | BWD_V | BWD_K programs)'),
    'local_attn_train_fwd': (4, 'fused window-attention forward with the saved log-sum-exp'),
    'locatt_similar_fwd': (2, 'locatt_ops similar forward (float32, unfused: + the (n,H,W,81) weight tensor)'),
    'locatt_similar_bwd': (2, 'locatt_ops similar backward'),
    'locatt_weighting_fwd': (2, 'locatt_ops weighting forward'),
    'locatt_weighting_bwd_ori': (2, 'locatt_ops weighting backward w.r.t. the values'),
    'locatt_weighting_bwd_weight': (2, 'locatt_ops weighting backward w.r.t. the weights'),
    'ms_deform_attn_bwd': (None, 'multi-scale deformable attention backward (DeepInteraction++)'),
}


def train_roofline(eager_trainer, shape, args, device, world, amp, pp):
    """The training line's `roofline`: the window-attention kernels of the step (its largest own kernels by bytes), timed
    live with HIP events in EAGER steps of the same model and data after the timed region (events cannot bracket a kernel
    inside a graph replay): every profiled launch of the image-side maps, the dominant one (largest total time) on top."""
    from . import measure, ops
    tr = eager_trainer
    if tr is None:
        tr = Trainer(shape, args.proposals, device, world, batch=args.batch, pool=2, rank=0, amp=amp, **(dict(model='pp') if pp else {}))
    tr.step()
    torch.cuda.synchronize()
    ops.PROFILE = []
    for _ in range(max(1, min(args.roofline_steps, 2))):
        tr.step()
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    Hi, Wi = shape['img_hw']
    n_img = 6 * args.batch
    rows = []
    steps_p = max(1, min(args.roofline_steps, 2))
    if pp:      # the ++ neck has no window attention: its byte-bound kernel is the deformable attention (forward launches, 2 levels)
        es = 2 if amp else 4
        nq = n_img * Hi * Wi
        S = n_img * (Hi * Wi + (Hi // 2) * (Wi // 2))
        alg = (S * 128 + nq * (8 * 2 * 4 * 3 + 128)) * es
        ev = [(n, s_, e) for (nm, n, s_, e) in prof if nm == 'ms_deform_attn_fwd' and n == nq]
        if ev:
            r = measure.kernel_row('ms_deform_attn_fwd', 'multi-scale deformable attention forward of the image tokens (csrc/plusplus.hip), '
                                   'self attention (2 levels) and P2I (1 level) launches together', prof, 'ms_deform_attn_fwd', nq, alg)
            r['total_us_per_step'] = round(sum(s_.stream_ms(e) for _, s_, e in ev) * 1e3 / steps_p, 1)
            rows.append(r)
    for name, (maps, what) in _TRAIN_KERNELS.items():
        ev = [(n, s, e) for (nm, n, s, e) in prof if nm == name and n == n_img]
        if not ev or maps is None:
            continue
        es = 2 if name.startswith('local_attn_train') else 4
        alg = maps * n_img * 128 * Hi * Wi * es + (n_img * Hi * Wi * 81 * 4 if name.startswith('locatt') else 0)
        r = measure.kernel_row(name, what, prof, name, n_img, alg)
        r['total_us_per_step'] = round(sum(s.stream_ms(e) for _, s, e in ev) * 1e3 / steps_p, 1)
        rows.append(r)
    if not rows:
        return None
    rows.sort(key=lambda r: -r['total_us_per_step'])
    top = dict(rows[0])
    top.update(bound='hbm', traffic=None, kernels=rows,
               timed_in='eager training steps right after the timed region, HIP events on the launch stream around each call '
                        '(a call of the fused backward is four launches); image-side maps only')
    return top
