"""Whole-forward hipGraph capture of the interaction hot path.

The MMRI encoder + MMPI decoder forward issues ~600 kernel launches and no device->host
synchronisation, so on MI355X its eager wall time is launch-bound (8.1 ms per sample for 5.5 ms of
kernels, profiles/r01c).  `GraphedHotPath` captures one forward into a hipGraph
(`torch.cuda.CUDAGraph` on ROCm) over STATIC device buffers and replays it:

    g = GraphedHotPath(encoder, decoder, inputs)      # inputs: dict as produced by to_device()
    out = g()                                         # replay on the current inputs
    g.load(new_inputs)                                # copy a new sample into the static buffers
    out = g()
    g.img_feats, g.pts_feats, g.pts, g.pillars ...    # or: the producer writes the next sample INTO the static buffers
                                                      # (zero-copy hand-over; geometry through load_geometry())

Everything that is shape-static is captured as is.  What varies per sample is handled outside the
graph, in place:
  * feature maps, points, pillars: the static tensors are views into ONE arena; `prepare()` packs a batch into a
    record with the same layout and `load(record)` is a single device-to-device copy (same shapes required);
  * raw points and pillars: the static buffers have a fixed CAPACITY; a smaller sample is padded -
    pillars with `num_points = 0` (the pillar kernel skips them before touching memory), points with
    NaN coordinates (every projection test compares false, so they are never scattered);
  * `img_metas`: the small per-sample geometry buffers (lidar2img, their inverses, augmentation
    affines, pixel grids) are recomputed on the host and copied into the SAME device buffers
    (`SampleGeometry.update`, `QueryGeometry.update`), whose addresses the graph has baked in.
No work is skipped on replay: depth scatter/completion, all gathers, attention and GEMM kernels
are part of the graph.
"""
import queue
import threading

import torch

from .geometry import SampleGeometry
from .mmdet3d_plugin.models.utils.decoder_utils import QueryGeometry
from .mmdet3d_plugin.models.utils.encoder_utils import GEOM_KEY


class GraphedHotPath:
    def __init__(self, encoder, decoder, inputs, warmup=3, glue=None, image_net=None, overlap=None):
        """overlap: the fork / join sites of the captured forward (`utils.OVERLAP` bits; None = the process default).  0 = a
        single-stream capture: what several captures replayed side by side on their own streams should be - their
        parallelism comes from the other samples in flight, and every extra branch stream competes for the process's four
        hardware queues (four lanes: 1 108-1 125 samples/s with the forked captures, 1 188-1 213 with single-stream ones;
        one sample at a time the forked capture is the faster one, 1.385 against 1.414 ms).
        image_net: a `FrozenResNetFPN` (mmdet3d_plugin/models/detectors/image_glue.py) - the captured forward then STARTS
        FROM THE CAMERA IMAGES: `inputs['images']` ((B*N, 3, H, W), the network's dtype) takes the place of
        `inputs['img_feats']` as the static input of the image slot (`self.img_feats` holds the images), and the feature
        levels the neck reads are produced inside every replay.
        glue: a `PointGlue` (mmdet3d_plugin/models/detectors) - the captured forward then STARTS FROM THE POINTS:
        the pillars / coordinates / counts of `pts_metas` are rebuilt by the voxeliser inside every replay (capacity-sized
        buffers, no host synchronisation) instead of being copied in by `load()`."""
        assert not encoder.training and not decoder.training, 'graph capture is for the inference form'
        self.enc, self.dec = encoder, decoder
        self.glue = glue
        self.image_net = image_net
        self._img_key = 'images' if image_net is not None else 'img_feats'
        # one map per modality (v1 neck) or a list of levels (DeepInteraction++ neck); the camera images with `image_net`
        self.img_feats = self._clone(inputs[self._img_key])
        self.pts_feats = self._clone(inputs['pts_feats'])
        if image_net is not None:
            with torch.no_grad():
                first_img = image_net(self.img_feats)[0]
        else:
            first_img = self.img_feats[0] if isinstance(self.img_feats, list) else self.img_feats
        dev = first_img.device
        pm = inputs['pts_metas']
        self.batch = len(inputs['img_metas'])
        self.img_metas = [dict(m) for m in inputs['img_metas']]
        self.pts = [p.clone() for p in pm['pts']]
        if glue is not None:                # rebuilt from the points inside the forward: no static pillar buffers
            self.pillars = self.pillar_coors = self.pillars_num_points = None
            self.bounds = None
        else:
            self.pillars = pm['pillars'].clone()
            self.pillar_coors = pm['pillar_coors'].clone()
            self.pillars_num_points = pm['pillars_num_points'].clone()
        if glue is not None:
            pass
        elif self.batch == 1:
            self.bounds = [0, self.pillars.shape[0]]
        else:   # fixed per-sample pillar capacity: the split of the capture-time batch
            cnt = torch.bincount(self.pillar_coors[:, 0].long(), minlength=self.batch).cpu().tolist()
            self.bounds = [0]
            for c in cnt:
                self.bounds.append(self.bounds[-1] + c)
        Hi, Wi = first_img.shape[-2:]
        self.sample_geom = [SampleGeometry(m, (Hi, Wi), dev) for m in self.img_metas]
        self.query_geom = QueryGeometry(self.img_metas, dev)
        self.graph = None
        self.out = None
        self._health = torch.zeros(1, dtype=torch.int32).pin_memory() if dev.type == 'cuda' else None
        # the process-global, monotonic device counter as this capture has already reported it (construction: what earlier
        # launches of the process left behind is not this capture's fault)
        self._health_seen = [0]
        self._build_arena()
        from . import utils
        with utils.overlap(overlap):          # this thread's forwards only: other threads keep their fork / join sites
            self._capture(warmup)

    @staticmethod
    def _clone(x):
        if isinstance(x, (list, tuple)):
            return [t.clone(memory_format=torch.preserve_format) for t in x]
        return x.clone(memory_format=torch.preserve_format)

    # ------------------------------------------------------------------ input arena
    # All per-sample input tensors (feature maps, points, pillars) are views into ONE device allocation, in a layout that
    # depends on their shapes only: `load()` of a prepared record is then one device-to-device copy instead of one per
    # tensor (8-12 blit launches of 3-25 us each in front of every replay).
    def _input_list(self):
        seq = []
        for x in (self.img_feats, self.pts_feats):
            seq += list(x) if isinstance(x, list) else [x]
        seq += list(self.pts)
        if self.glue is None:
            seq += [self.pillars, self.pillar_coors, self.pillars_num_points]
        return seq

    def _adopt(self, views):
        it = iter(views)
        take = lambda x: [next(it) for _ in x] if isinstance(x, list) else next(it)
        self.img_feats, self.pts_feats, self.pts = take(self.img_feats), take(self.pts_feats), take(self.pts)
        if self.glue is None:
            self.pillars, self.pillar_coors, self.pillars_num_points = next(it), next(it), next(it)

    def _arena_views(self, arena):
        return [arena[o:o + n].view(dt).as_strided(shape, stride) for o, n, dt, shape, stride in self._layout]

    def _build_arena(self):
        tensors = self._input_list()
        self._layout, total = [], 0
        for t in tensors:
            n = t.numel() * t.element_size()
            extent = sum((d - 1) * st for d, st in zip(t.shape, t.stride())) + 1 if t.numel() else 0
            assert extent == t.numel(), 'static inputs must be dense (contiguous in some dimension order)'
            self._layout.append((total, n, t.dtype, tuple(t.shape), tuple(t.stride())))
            total += (n + 255) // 256 * 256
        self._arena = torch.empty(max(total, 256), dtype=torch.uint8, device=tensors[0].device)
        views = self._arena_views(self._arena)
        for v, t in zip(views, tensors):
            v.copy_(t)
        self._adopt(views)

    def _pts_metas(self):
        if self.glue is not None:
            pm = self.glue(self.pts, padded=True)           # voxelisation inside the (captured) forward
            pm[GEOM_KEY] = self.sample_geom
            return pm
        return {'pillars': self.pillars, 'pillar_coors': self.pillar_coors,
                'pillars_num_points': self.pillars_num_points, 'pts': self.pts,
                'pillar_batch_bounds': self.bounds, GEOM_KEY: self.sample_geom}

    def _forward(self):
        for g in self.sample_geom:          # depth scatter + completion belong to every forward
            g.forget()
        self.dec.static_geometry = self.query_geom
        try:
            img_in = self.img_feats
            if self.image_net is not None:      # frozen image network inside the (captured) forward
                levels = self.image_net(img_in)
                img_in = levels[0] if len(levels) == 1 else list(levels)
            img, pts = self.enc(img_in, self.pts_feats, self.img_metas, self._pts_metas())
            self.enc_out = (img, pts)           # static buffers too: valid after every replay
            return self.dec(pts, img, self.img_metas)
        finally:
            self.dec.static_geometry = None

    def _capture(self, warmup):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):          # library plans, kernel attributes, folded-weight caches
                self._forward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # node count of the captured forward (measurement only): a throw-away capture that keeps the hipGraph_t, then
        # the real one (a graph object created with keep_graph=True re-instantiates on replay with this torch build)
        self._nodes = None
        try:
            import os
            if os.environ.get('DI_GRAPH_NODES', '1') == '0':
                raise RuntimeError('node count disabled')
            from . import _lib
            probe = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.no_grad(), torch.cuda.graph(probe):
                self._forward()
            n = int(_lib.lib().di_graph_node_count(int(probe.raw_cuda_graph())))
            self._nodes = n if n >= 0 else None
            probe.reset()
            del probe
        except Exception:      # a torch build without raw graph access just reports None
            self._nodes = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = self._forward()

    def check_health(self, stream=None):
        """Synchronises `stream` (default: the current one; InflightLanes passes the lane's) and raises when a captured
        kernel reported a fault it could only report through device memory since the last check (the ring window attention's
        bounded spins: affected tiles are NaN in the outputs).  Call it where the outputs are read.  A fault is reported
        ONCE: the baseline moves up, later replays with sound outputs do not raise again."""
        from . import ops
        with torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext():
            ops.check_ring_health(self._health_seen)

    def num_nodes(self):
        """Nodes (kernel launches, copies, memsets) of the captured forward, or None when torch does not expose it."""
        return self._nodes

    def __call__(self):
        # faults a captured kernel can only report through device memory (the ring window attention's bounded spins: the
        # affected tiles are NaN): a pinned host word receives the device counter behind every replay, without a
        # synchronisation; what an EARLIER replay reported is checked here
        if self._health is not None and int(self._health[0]) > self._health_seen[0]:
            new, self._health_seen[0] = int(self._health[0]) - self._health_seen[0], int(self._health[0])
            raise RuntimeError(f'local_attn_ring: {new} bounded spins gave up in an earlier replay - the affected output tiles of '
                               'THAT replay hold NaN (ops.check_ring_health); reported once, later replays are checked afresh')
        self.graph.replay()
        if self._health is not None:
            from . import _lib, ops
            _lib.call('di_local_attn_ring_timeouts_async', self._health.data_ptr(), ops._stream())
        return self.out

    # ------------------------------------------------------------------ per-sample inputs
    class Record:
        """One batch, device resident and already in the captured layout (points / pillars padded to the captured
        capacity, geometry constants packed): `load(record)` is a handful of device-to-device copies and no host
        work, so it can sit inside a timed per-sample loop."""
        __slots__ = ('arena', 'img_feats', 'pts_feats', 'pts', 'pillars', 'pillar_coors', 'pillars_num_points', 'img_metas',
                     'sample_geom', 'query_geom', 'extra')

    @staticmethod
    def _padded(src, like, fill):
        n = src.shape[0]
        if n > like.shape[0]:
            raise ValueError(f'sample of {n} rows exceeds the captured capacity {like.shape[0]}')
        out = torch.full_like(like, fill)
        out[:n].copy_(src.to(like.dtype))
        return out

    def prepare(self, inputs):
        """Pack a batch (dict as produced by `harness.to_device`) into a `Record` (may synchronise; do it ahead of
        the per-sample loop).  Same shapes as the captured batch; points / pillars up to the captured capacity."""
        if self.batch != len(inputs['img_metas']):
            raise ValueError('batch size differs from the captured one')
        r = GraphedHotPath.Record()
        r.img_feats, r.pts_feats = inputs[self._img_key], inputs['pts_feats']
        pm = inputs['pts_metas']
        r.pts = [self._padded(src, dst, float('nan')) for dst, src in zip(self.pts, pm['pts'])]
        if self.glue is not None:
            r.pillars = r.pillar_coors = r.pillars_num_points = None
        elif self.batch == 1:
            r.pillars = self._padded(pm['pillars'], self.pillars, 0.0)
            r.pillar_coors = self._padded(pm['pillar_coors'], self.pillar_coors, 0)
            r.pillars_num_points = self._padded(pm['pillars_num_points'], self.pillars_num_points, 0)
        else:      # every sample's pillars go to that sample's fixed slice of the captured buffers
            r.pillars = torch.zeros_like(self.pillars)
            r.pillar_coors = torch.zeros_like(self.pillar_coors)
            r.pillars_num_points = torch.zeros_like(self.pillars_num_points)
            b = pm['pillar_coors'][:, 0].long()
            for s in range(self.batch):
                lo, hi = self.bounds[s], self.bounds[s + 1]
                sel = (b == s).nonzero().flatten()
                if sel.numel() > hi - lo:
                    raise ValueError(f'sample {s}: {sel.numel()} pillars exceed the captured capacity {hi - lo}')
                n = sel.numel()
                r.pillars[lo:lo + n] = pm['pillars'][sel]
                r.pillar_coors[lo:lo + n] = pm['pillar_coors'][sel]
                r.pillar_coors[lo:hi, 0] = s
                r.pillars_num_points[lo:lo + n] = pm['pillars_num_points'][sel]
        # the record's own arena, laid out like the captured one: its fields become views into it
        srcs = []
        for x in (r.img_feats, r.pts_feats):
            srcs += list(x) if isinstance(x, (list, tuple)) else [x]
        srcs += list(r.pts)
        if self.glue is None:
            srcs += [r.pillars, r.pillar_coors, r.pillars_num_points]
        r.arena = torch.empty_like(self._arena)
        views = self._arena_views(r.arena)
        assert len(views) == len(srcs), 'the batch has a different structure from the captured one'
        for v, src in zip(views, srcs):
            if tuple(v.shape) != tuple(src.shape):
                raise ValueError(f'input of shape {tuple(src.shape)} where the captured forward has {tuple(v.shape)}')
            v.copy_(src)
        it = iter(views)
        take = lambda x: [next(it) for _ in x] if isinstance(x, (list, tuple)) else next(it)
        r.img_feats, r.pts_feats, r.pts = take(r.img_feats), take(r.pts_feats), take(r.pts)
        if self.glue is None:
            r.pillars, r.pillar_coors, r.pillars_num_points = next(it), next(it), next(it)
        r.img_metas = [dict(m) for m in inputs['img_metas']]
        dev = self.pts[0].device
        r.sample_geom = [SampleGeometry._pack(m, g.img_hw).to(dev) for g, m in zip(self.sample_geom, r.img_metas)]
        r.query_geom = QueryGeometry._pack(r.img_metas)[0].to(dev)
        r.extra = []                      # per-sample constants some operators keep next to the geometry (++ rays)
        for mod in self.enc.modules():
            if hasattr(mod, 'static_geometry_record'):
                r.extra.append((mod, [mod.static_geometry_record(g, m) for g, m in zip(self.sample_geom, r.img_metas)]))
        return r

    def record_inputs(self, r):
        """The input dict of a prepared record (points / pillars padded to the captured capacity).  A forward captured from
        it - `GraphedHotPath(enc, dec, g.record_inputs(r))` - has this capture's layout and capacity and holds the record's
        sample in its static buffers: the ZERO-COPY hand-over (one captured forward per in-flight slot whose producer writes
        its feature maps, points and pillars straight into the slot's `img_feats` / `pts_feats` / `pts` / `pillars*` views;
        `bench.py --handover resident` emulates that producer with one capture per pool sample)."""
        return {self._img_key: r.img_feats, 'pts_feats': r.pts_feats, 'img_metas': [dict(m) for m in r.img_metas],
                'pts_metas': {'pts': r.pts, 'pillars': r.pillars, 'pillar_coors': r.pillar_coors,
                              'pillars_num_points': r.pillars_num_points}}

    def load_raw(self, inputs):
        """`load(prepare(inputs))` without the intermediate record: every input of a raw batch (dict as produced by
        `harness.to_device`; the maps in ANY dense memory format, e.g. the NCHW a backbone hands over) goes straight into
        its captured buffer - one strided copy per tensor, which also does the channels-last conversion - and the
        geometry constants are packed on the host and uploaded.  Batch size 1 per capture (the benched form); other
        batches go through `prepare()`."""
        if self.batch != 1 or self.batch != len(inputs['img_metas']):
            return self.load(self.prepare(inputs))
        pm = inputs['pts_metas']

        def put(dst, src, fill=None):
            if isinstance(dst, (list, tuple)):
                for d, s_ in zip(dst, src):
                    put(d, s_, fill)
                return
            if fill is None:
                if tuple(dst.shape) != tuple(src.shape):
                    raise ValueError(f'input of shape {tuple(src.shape)} where the captured forward has {tuple(dst.shape)}')
                dst.copy_(src, non_blocking=True)
                return
            n = src.shape[0]
            if n > dst.shape[0]:
                raise ValueError(f'sample of {n} rows exceeds the captured capacity {dst.shape[0]}')
            dst[:n].copy_(src, non_blocking=True)
            dst[n:].fill_(fill)

        put(self.img_feats, inputs[self._img_key])
        put(self.pts_feats, inputs['pts_feats'])
        put(self.pts, pm['pts'], float('nan'))
        if self.glue is None:
            put(self.pillars, pm['pillars'], 0.0)
            put(self.pillar_coors, pm['pillar_coors'], 0)
            put(self.pillars_num_points, pm['pillars_num_points'], 0)
        self.load_geometry(inputs['img_metas'])                  # pageable host temporaries: blocking copies

    def load_geometry(self, img_metas):
        """The zero-copy hand-over's other half: the producer has written maps / points / pillars into the static buffers
        (`img_feats`, `pts_feats`, `pts`, `pillars*`: padded as `prepare()` pads), this refreshes the per-sample geometry
        constants from `img_metas` in place (host 4x4 inverses + a few small H2D copies)."""
        self.img_metas = [dict(m) for m in img_metas]
        for g, m in zip(self.sample_geom, self.img_metas):
            g._buf.copy_(SampleGeometry._pack(m, g.img_hw))
            g.forget()
        self.query_geom._buf.copy_(QueryGeometry._pack(self.img_metas)[0])
        for mod in self.enc.modules():
            if hasattr(mod, 'static_geometry_record'):
                for g, m in zip(self.sample_geom, self.img_metas):
                    mod.load_static_geometry(g, mod.static_geometry_record(g, m))

    def load(self, inputs):
        """Make a new batch the current one: copy it into the static buffers and refresh the geometry constants in
        place.  `inputs`: a `Record` from `prepare()` (device-to-device copies only) or a raw input dict."""
        r = inputs if isinstance(inputs, GraphedHotPath.Record) else self.prepare(inputs)
        if r.arena.shape != self._arena.shape:
            raise ValueError('record prepared for a different captured layout')
        if self._arena.data_ptr() % 16 == 0 and r.arena.data_ptr() % 16 == 0:
            from . import ops
            ops.copy_bytes(self._arena, r.arena)                  # every feature map, the points and the pillars: ONE launch
        else:
            self._arena.copy_(r.arena, non_blocking=True)
        self.img_metas = r.img_metas
        for g, buf in zip(self.sample_geom, r.sample_geom):
            g._buf.copy_(buf, non_blocking=True)
            g.forget()
        self.query_geom._buf.copy_(r.query_geom, non_blocking=True)
        for mod, recs in r.extra:
            for g, rec in zip(self.sample_geom, recs):
                mod.load_static_geometry(g, rec)


class LaneLaunchers:
    """One HOST THREAD per in-flight lane (a lane = a HIP stream on which captured forwards are replayed one behind the other).

    Why: a replay of the 93-node forward costs the launching thread 0.3 ms of hipGraphLaunch into an idle queue and 0.7-1.0 ms
    while the GPU is busy (tools/launch_cost.py, round 5) - of the same order as the 0.9-1.0 ms the GPU needs per sample when
    several are in flight.  With ONE launching thread four lanes are host-bound (1 006 samples/s, the launches of a step take
    2.98 of its 3.97 ms); with a thread per lane the launches of a step run side by side (CUDAGraph.replay and the ctypes calls
    release the GIL) and the same four lanes reach 1 090.

        pool = LaneLaunchers(streams)
        pool.run([fn0, fn1, ...])      # fn_l runs on lane l's thread with lane l's stream current; returns when all have returned
        pool.close()

    `run` returns when every lane has ISSUED its work (nothing is synchronised with the GPU); an exception raised on a lane is
    re-raised by `run`."""

    def __init__(self, streams):
        self.streams = list(streams)                # (None entries: no stream switch - the host-logic tests run without a device)
        self._device = torch.cuda.current_device() if torch.cuda.is_available() else None
        self._jobs = [queue.SimpleQueue() for _ in self.streams]
        self._done = queue.SimpleQueue()
        self._threads = [threading.Thread(target=self._work, args=(l,), daemon=True, name=f'lane{l}')
                         for l in range(len(self.streams))]
        for t in self._threads:
            t.start()

    def _work(self, l):
        import contextlib
        if self._device is not None:
            torch.cuda.set_device(self._device)
        on = torch.cuda.stream(self.streams[l]) if self.streams[l] is not None else contextlib.nullcontext()
        with torch.no_grad(), on:
            while True:
                fn = self._jobs[l].get()
                if fn is None:
                    return
                try:
                    fn()
                    self._done.put(None)
                except BaseException as e:         # handed to the caller of run()
                    self._done.put(e)

    def run(self, fns):
        assert len(fns) == len(self.streams)
        for q, fn in zip(self._jobs, fns):
            q.put(fn)
        err = None
        for _ in fns:
            e = self._done.get()
            err = err or e
        if err is not None:
            raise err

    def close(self):
        for q in self._jobs:
            q.put(None)
        for t in self._threads:
            t.join()
        self._threads = []


class InflightLanes:
    """Several samples in flight on one GPU, the serving form of what `bench.py` times: `n_lanes` lanes, each a HIP stream with
    its own SINGLE-STREAM capture of the forward (`GraphedHotPath(overlap=0)`: the lanes are each other's parallelism) and its own
    launching host thread (`LaneLaunchers`).  Use four lanes, or a multiple: a process has four hardware queues, and a lane count
    that is not a multiple of four puts two lanes on one queue (5 lanes: 1 053 samples/s against 1 202 with 4).

        lanes = InflightLanes(encoder, decoder, example_inputs, n_lanes=4)
        lanes[l].load(record)              # or: the producer writes into lanes[l].img_feats / .pts_feats / .pts / .pillars* and
        lanes[l].load_geometry(img_metas)  #     the geometry constants are refreshed in place (zero-copy hand-over)
        outs = lanes.replay()              # one replay per lane, issued side by side from the lanes' threads (no synchronisation)
        lanes.synchronize()                # outs[l] are lane l's static output tensors: valid until its next replay

    `example_inputs` sets the shapes and the point / pillar capacity of every lane (`prepare()` pads smaller samples)."""

    def __init__(self, encoder, decoder, example_inputs, n_lanes=4, launch_threads=True, **capture_kwargs):
        self.slots = [GraphedHotPath(encoder, decoder, example_inputs, overlap=0 if n_lanes > 1 else None, **capture_kwargs)
                      for _ in range(n_lanes)]
        self.streams = [torch.cuda.Stream() for _ in self.slots]
        for s_ in self.streams:
            s_.wait_stream(torch.cuda.current_stream())
        self._launchers = LaneLaunchers(self.streams) if launch_threads and n_lanes > 1 else None

    def __len__(self):
        return len(self.slots)

    def __getitem__(self, l):
        return self.slots[l]

    def prepare(self, inputs):
        return self.slots[0].prepare(inputs)            # every lane has the same layout: a record loads into any of them

    def load(self, l, record):
        """`slots[l].load(record)` on lane l's stream (ordered behind the lane's previous replay)."""
        with torch.cuda.stream(self.streams[l]):
            self.slots[l].load(record)

    def replay(self, which=None, check_health=False):
        """One replay of every lane (or of the lanes listed in `which`), issued side by side; returns their static outputs.
        `check_health=True` (serving paths that must never hand out a poisoned tile): synchronises the replayed lanes and
        raises BEFORE returning when a kernel of these replays reported a fault through device memory (the ring window
        attention's bounded spins: NaN tiles).  Default False: no synchronisation, a fault is raised by the NEXT replay."""
        which = list(range(len(self.slots))) if which is None else list(which)
        if self._launchers is not None and len(which) == len(self.slots):
            self._launchers.run(self.slots)
        else:
            for l in which:
                with torch.cuda.stream(self.streams[l]):
                    self.slots[l]()
        if check_health:
            self.check_health(which)
        return [self.slots[l].out for l in which]

    def check_health(self, which=None):
        """Synchronous health check of the listed lanes (all by default), each under ITS stream."""
        for l in (range(len(self.slots)) if which is None else which):
            self.slots[l].check_health(self.streams[l])

    def synchronize(self):
        for s_ in self.streams:
            s_.synchronize()

    def close(self):
        if self._launchers is not None:
            self._launchers.close()
            self._launchers = None
