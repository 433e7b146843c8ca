"""Whole-forward hipGraph capture of the interaction hot path.

The MMRI encoder + MMPI decoder forward issues ~600 kernel launches and no device->host
synchronisation, so on MI355X its eager wall time is launch-bound (8.1 ms per sample for 5.5 ms of
kernels, profiles/r01c).  `GraphedHotPath` captures one forward into a hipGraph
(`torch.cuda.CUDAGraph` on ROCm) over STATIC device buffers and replays it:

    g = GraphedHotPath(encoder, decoder, inputs)      # inputs: dict as produced by to_device()
    out = g()                                         # replay on the current inputs
    g.load(new_inputs)                                # copy a new sample into the static buffers
    out = g()

Everything that is shape-static is captured as is.  What varies per sample is handled outside the
graph, in place:
  * feature maps: `copy_` into the static tensors (same shape required);
  * raw points and pillars: the static buffers have a fixed CAPACITY; a smaller sample is padded -
    pillars with `num_points = 0` (the pillar kernel skips them before touching memory), points with
    NaN coordinates (every projection test compares false, so they are never scattered);
  * `img_metas`: the small per-sample geometry buffers (lidar2img, their inverses, augmentation
    affines, pixel grids) are recomputed on the host and copied into the SAME device buffers
    (`SampleGeometry.update`, `QueryGeometry.update`), whose addresses the graph has baked in.
No work is skipped on replay: depth scatter/completion, all gathers, attention and GEMM kernels
are part of the graph.
"""
import torch

from .geometry import SampleGeometry
from .mmdet3d_plugin.models.utils.decoder_utils import QueryGeometry
from .mmdet3d_plugin.models.utils.encoder_utils import GEOM_KEY


class GraphedHotPath:
    def __init__(self, encoder, decoder, inputs, warmup=3):
        assert not encoder.training and not decoder.training, 'graph capture is for the inference form'
        self.enc, self.dec = encoder, decoder
        # one map per modality (v1 neck) or a list of levels (DeepInteraction++ neck)
        self.img_feats = self._clone(inputs['img_feats'])
        self.pts_feats = self._clone(inputs['pts_feats'])
        first_img = self.img_feats[0] if isinstance(self.img_feats, list) else self.img_feats
        dev = first_img.device
        pm = inputs['pts_metas']
        self.batch = len(inputs['img_metas'])
        self.img_metas = [dict(m) for m in inputs['img_metas']]
        self.pts = [p.clone() for p in pm['pts']]
        self.pillars = pm['pillars'].clone()
        self.pillar_coors = pm['pillar_coors'].clone()
        self.pillars_num_points = pm['pillars_num_points'].clone()
        if self.batch == 1:
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
        self._capture(warmup)

    @staticmethod
    def _clone(x):
        if isinstance(x, (list, tuple)):
            return [t.clone(memory_format=torch.preserve_format) for t in x]
        return x.clone(memory_format=torch.preserve_format)

    @staticmethod
    def _copy(dst, src):
        if isinstance(dst, list):
            for d, s_ in zip(dst, src):
                d.copy_(s_, non_blocking=True)
        else:
            dst.copy_(src, non_blocking=True)

    def _pts_metas(self):
        return {'pillars': self.pillars, 'pillar_coors': self.pillar_coors,
                'pillars_num_points': self.pillars_num_points, 'pts': self.pts,
                'pillar_batch_bounds': self.bounds, GEOM_KEY: self.sample_geom}

    def _forward(self):
        for g in self.sample_geom:          # depth scatter + completion belong to every forward
            g.sparse_depth = g.dense_depth = None
        self.dec.static_geometry = self.query_geom
        try:
            img, pts = self.enc(self.img_feats, self.pts_feats, self.img_metas, self._pts_metas())
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
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = self._forward()

    def __call__(self):
        self.graph.replay()
        return self.out

    @staticmethod
    def _fit(dst, src, fill):
        n = src.shape[0]
        if n > dst.shape[0]:
            raise ValueError(f'sample of {n} rows exceeds the captured capacity {dst.shape[0]}')
        dst[:n].copy_(src, non_blocking=True)
        if n < dst.shape[0]:
            dst[n:].fill_(fill)

    def load(self, inputs):
        """Copy a new batch into the static buffers (same shapes; points / pillars up to the captured
        capacity) and refresh the geometry constants in place."""
        self._copy(self.img_feats, inputs['img_feats'])
        self._copy(self.pts_feats, inputs['pts_feats'])
        pm = inputs['pts_metas']
        if self.batch != len(inputs['img_metas']):
            raise ValueError('batch size differs from the captured one')
        for dst, src in zip(self.pts, pm['pts']):
            self._fit(dst, src.to(dst.dtype), float('nan'))
        if self.batch == 1:
            self._fit(self.pillars, pm['pillars'], 0.0)
            self._fit(self.pillar_coors, pm['pillar_coors'], 0)
            self._fit(self.pillars_num_points, pm['pillars_num_points'], 0)
        else:
            b = pm['pillar_coors'][:, 0].long()
            for s in range(self.batch):
                lo, hi = self.bounds[s], self.bounds[s + 1]
                sel = (b == s).nonzero().flatten()
                self._fit(self.pillars[lo:hi], pm['pillars'][sel], 0.0)
                self._fit(self.pillar_coors[lo:hi], pm['pillar_coors'][sel], 0)
                self._fit(self.pillars_num_points[lo:hi], pm['pillars_num_points'][sel], 0)
        self.img_metas = [dict(m) for m in inputs['img_metas']]
        for b, (g, m) in enumerate(zip(self.sample_geom, self.img_metas)):
            g.update(m)
            for mod in self.enc.modules():       # per-sample constants some operators keep next to the geometry
                if hasattr(mod, 'refresh_static_geometry'):
                    mod.refresh_static_geometry(g, m)
        self.query_geom.update(self.img_metas)
