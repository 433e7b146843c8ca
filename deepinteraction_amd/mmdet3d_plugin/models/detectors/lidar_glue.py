"""The frozen LiDAR feature extractor in front of the hot path (round 6): the reference detector's
`pts_voxel_layer -> pts_voxel_encoder -> pts_middle_encoder -> pts_backbone -> pts_neck` chain (`extract_pts_feat`,
reference `models/detectors/deepinteraction.py:120-131`; config `Fusion_0075_refactor.py:146-184`: hard voxelisation at
0.075 x 0.075 x 0.2 m, `HardSimpleVFE`, mmdet3d `SparseEncoder` (spconv), `SECOND`, `SECONDFPN`) on PyTorch-ROCm WITHOUT spconv:

* `FrozenSparseEncoder` - mmdet3d 0.17.1 `SparseEncoder(block_type='basicblock')`.  A sparse tensor is (coordinates (M, 4) int
  [b, z, y, x], features (M, C)); the RULEBOOK of a layer is a neighbour table (row of the input feature that kernel offset o of
  output voxel m reads, or none); submanifold convolutions (`SubMConv3d`: outputs exactly on the input's voxels) share the table
  of their resolution (spconv's `indice_key`), strided `SparseConv3d` layers make the next resolution: an output voxel exists
  where any input voxel lies in its window; BatchNorm folded (eval mode: frozen).  Two forms of the same arithmetic:
  - on the device in fp16 (the product): `csrc/sparse_conv.hip` through `ops.sparse_output_keys / sparse_neighbours / sparse_conv` -
    sorted int32 voxel keys per level, tables by a mark kernel + one binary search per kernel row, every convolution ONE launch
    (gather, MFMA product, bias / residual / ReLU);
  - torch operators (CPU tensors, float32, `DI_SPARSE_HIP=0`): tables by sorting the keys and `torch.searchsorted` for all k^3
    offsets at once, `X = feats[nbr]` as the im2col matrix (M_out, k^3 C_in) and `Y = X W + b` - what the CPU tests pin against
    brute force and the dense oracle, and what the device form is compared with.
* `FrozenSECOND`, `FrozenSECONDFPN` - dense 2-D convolutions, BatchNorm folded, channels-last fp16: SECOND's stride-1 layers of 128
  output channels on the hot path's own 3x3 kernel (`ops.conv3x3`), the rest through torch / MIOpen (frozen backbones on
  PyTorch-ROCm: BASELINE configs[2]).
* `FrozenLidarBackbone(points) -> [BEV map (B, 512, 180, 180)]` = `pts_feats` of the hot path; any callable of that signature is
  what `DeepInteractionInference(pts_backbone=...)` takes.

State-dict keys mirror mmdet3d's modules (`conv_input.0.weight`, `encoder_layers.encoder_layer1.0.conv1.weight`, `...bn1.*`,
`conv_out.1.running_var`, `blocks.0.3.weight`, `deblocks.1.0.weight` ...), sparse weights in spconv 2.x layout
(C_out, kD, kH, kW, C_in) or the spconv 1.x layout of mmdet3d 0.17.1's bundled ops (kD, kH, kW, C_in, C_out) - told apart by
shape -, so a reference checkpoint's `pts_middle_encoder.* / pts_backbone.* / pts_neck.*` entries load by name.
Parity: against `oracle/sparse_encoder.py` (dense restatement of the published spconv semantics) - UNPINNED: spconv / mmdet3d
are not in this image.  Shapes depend on the data (the number of active voxels): this part runs eagerly, in front of the
captured forward."""
import os

import torch
import torch.nn.functional as F
from torch import nn

from ..updated_modules.sparse_voxelize import SPConvVoxelization


def _out_dim(n, k, s, p):
    return (n + 2 * p - k) // s + 1


class SparseTensor:
    """coords (M, 4) int64 [b, z, y, x] sorted by key, feats (M, C), spatial shape (D, H, W), batch size."""

    def __init__(self, coords, feats, shape, batch):
        self.coords, self.feats, self.shape, self.batch = coords, feats, tuple(int(x) for x in shape), int(batch)

    def keys(self):
        D, H, W = self.shape
        c = self.coords
        return ((c[:, 0] * D + c[:, 1]) * H + c[:, 2]) * W + c[:, 3]

    def dense(self):
        """(B, C, D, H, W), zeros where no voxel is active (spconv `SparseConvTensor.dense()`)."""
        D, H, W = self.shape
        out = self.feats.new_zeros((self.batch * D * H * W, self.feats.shape[1]))
        out[self.keys()] = self.feats
        return out.view(self.batch, D, H, W, -1).permute(0, 4, 1, 2, 3)


def _offsets(ksize, device):
    kd, kh, kw = ksize
    z, y, x = torch.meshgrid(torch.arange(kd, device=device), torch.arange(kh, device=device), torch.arange(kw, device=device),
                             indexing='ij')
    return torch.stack([z.reshape(-1), y.reshape(-1), x.reshape(-1)], 1)          # (K, 3), kernel index order (kd, kh, kw)


def _lookup(sorted_keys, query_keys, valid):
    """Row of `sorted_keys` equal to each query key, or -1 (also where `valid` is False)."""
    M = sorted_keys.numel()
    if M == 0:
        return torch.full_like(query_keys, -1)
    pos = torch.searchsorted(sorted_keys, query_keys.clamp(min=0)).clamp(max=M - 1)
    hit = valid & (sorted_keys[pos] == query_keys)
    return torch.where(hit, pos, torch.full_like(pos, -1))


def subm_rulebook(x, ksize=(3, 3, 3)):
    """Neighbour table of a submanifold convolution: nbr[m, o] = input row at coords[m] + offset_o - (k - 1) / 2, or -1."""
    D, H, W = x.shape
    off = _offsets(ksize, x.coords.device) - torch.tensor([(k - 1) // 2 for k in ksize], device=x.coords.device)
    c = x.coords[:, None, 1:] + off[None]                                            # (M, K, 3)
    ok = ((c >= 0) & (c < torch.tensor([D, H, W], device=c.device))).all(-1)
    key = ((x.coords[:, None, 0] * D + c[..., 0]) * H + c[..., 1]) * W + c[..., 2]
    return _lookup(x.keys(), key, ok)


def strided_rulebook(x, ksize, stride, padding):
    """`SparseConv3d`: (output coords sorted by key, neighbour table nbr[m_out, o], output shape).  An output voxel exists where
    at least one input voxel lies in its window: out = (in + pad - off) / stride for every offset that divides."""
    dev = x.coords.device
    D, H, W = x.shape
    ks, st, pd = (torch.tensor(v, device=dev) for v in (ksize, stride, padding))
    oshape = tuple(_out_dim(n, k, s, p) for n, k, s, p in zip((D, H, W), ksize, stride, padding))
    osz = torch.tensor(oshape, device=dev)
    off = _offsets(ksize, dev)                                                       # (K, 3)
    num = x.coords[:, None, 1:] + pd[None, None] - off[None]                         # (M, K, 3)
    o = torch.div(num, st[None, None], rounding_mode='floor')
    ok = ((num % st[None, None]) == 0).all(-1) & ((o >= 0) & (o < osz)).all(-1)
    okey = ((x.coords[:, None, 0] * oshape[0] + o[..., 0]) * oshape[1] + o[..., 1]) * oshape[2] + o[..., 2]
    out_keys = torch.unique(okey[ok])                                                # sorted
    b = torch.div(out_keys, oshape[0] * oshape[1] * oshape[2], rounding_mode='floor')
    r = out_keys % (oshape[0] * oshape[1] * oshape[2])
    oz = torch.div(r, oshape[1] * oshape[2], rounding_mode='floor')
    oy = torch.div(r % (oshape[1] * oshape[2]), oshape[2], rounding_mode='floor')
    ox = r % oshape[2]
    ocoords = torch.stack([b, oz, oy, ox], 1)
    # for every output voxel and offset: the input voxel at out * stride - pad + off
    ic = ocoords[:, None, 1:] * st[None, None] - pd[None, None] + off[None]
    iok = ((ic >= 0) & (ic < torch.tensor([D, H, W], device=dev))).all(-1)
    ikey = ((ocoords[:, None, 0] * D + ic[..., 0]) * H + ic[..., 1]) * W + ic[..., 2]
    return ocoords, _lookup(x.keys(), ikey, iok), oshape


def sparse_conv(feats, nbr, weight, bias, relu, residual=None):
    """Y[m] = sum_o feats[nbr[m, o]] W[o] (+ bias, + residual, ReLU): one gather + one GEMM.  weight (K * C_in, C_out)."""
    M, K = nbr.shape
    pad = torch.cat([feats, feats.new_zeros((1, feats.shape[1]))], 0)               # row M = the zero row (index -1)
    X = pad[nbr].reshape(M, K * feats.shape[1])
    Y = torch.addmm(bias, X, weight) if bias is not None else X @ weight
    if residual is not None:
        Y = Y + residual
    return torch.relu_(Y) if relu else Y


def spconv2_layout(w, ksize, cin, cout):
    """A sparse-convolution weight in either published layout -> spconv 2.x (C_out, kD, kH, kW, C_in).  mmdet3d 0.17.1 (the
    reference's version) bundles spconv 1.x, whose `SparseConvolution.weight` is (kD, kH, kW, C_in, C_out): a checkpoint trained
    there - `Fusion_0075_refactor.pth` - carries that form; spconv 2.x (what `sparse_voxelize.py:3` imports) stores (C_out, kD,
    kH, kW, C_in).  The two are told apart by where the kernel extents sit (ambiguous only for a cubic kernel with C_in = C_out =
    its extent, which no layer of the encoder has)."""
    k = tuple(ksize)
    if tuple(w.shape) == (cout,) + k + (cin,) and not (tuple(w.shape) == k + (cin, cout)):
        return w
    if tuple(w.shape) == k + (cin, cout):
        return w.permute(4, 0, 1, 2, 3).contiguous()
    raise ValueError(f'sparse convolution weight of shape {tuple(w.shape)}: neither spconv 2.x {(cout,) + k + (cin,)} nor '
                     f'spconv 1.x {k + (cin, cout)}')


def _fold(conv_w, bn, prefix, sd, ksize=None, cin=None, cout=None):
    """(K * C_in, C_out) weight and (C_out) bias of `conv (bias=False) -> BatchNorm1d (eval)`; the weight in spconv 2.x layout or
    (when the geometry is given) in either layout."""
    w = sd[conv_w].float()                                   # (C_out, kD, kH, kW, C_in)
    if ksize is not None:
        w = spconv2_layout(w, ksize, cin, cout)
    g, b = sd[f'{bn}.weight'].float(), sd[f'{bn}.bias'].float()
    mu, var = sd[f'{bn}.running_mean'].float(), sd[f'{bn}.running_var'].float()
    scale = g / torch.sqrt(var + 1e-3)                       # norm_cfg: BN1d eps = 1e-3 (mmdet3d SparseEncoder default)
    co = w.shape[0]
    wk = w.reshape(co, -1).t().contiguous() * scale[None]    # (K * C_in, C_out): offsets in (kd, kh, kw) order, then C_in
    return wk, b - mu * scale


class FrozenSparseEncoder(nn.Module):
    """mmdet3d `SparseEncoder` (`middle_encoders/sparse_encoder.py`, block_type='basicblock', order conv-norm-act), frozen."""

    def __init__(self, in_channels=5, sparse_shape=(41, 1440, 1440), output_channels=128, base_channels=16,
                 encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                 encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, (0, 1, 1)), (0, 0)), dtype=torch.float16):
        super().__init__()
        self.in_channels, self.sparse_shape, self.output_channels = in_channels, tuple(sparse_shape), output_channels
        self.base_channels, self.encoder_channels, self.encoder_paddings = base_channels, encoder_channels, encoder_paddings
        self.dtype = dtype
        self._p, self._h = {}, {}

    HIP = os.environ.get('DI_SPARSE_HIP', '1') != '0'       # 0: the torch formulation (gather matrix + GEMM) on the device too

    # ---- the module tree as (kind, key prefix, geometry) in execution order
    def plan(self):
        steps = [('subm', 'conv_input', self.in_channels, self.base_channels)]
        cin = self.base_channels
        for i, blocks in enumerate(self.encoder_channels):
            for j, cout in enumerate(blocks):
                pre = f'encoder_layers.encoder_layer{i + 1}.{j}'
                if j == len(blocks) - 1 and i != len(self.encoder_channels) - 1:
                    pad = self.encoder_paddings[i][j]
                    pad = tuple(pad) if isinstance(pad, (tuple, list)) else (pad,) * 3
                    steps.append(('down', pre, cin, cout, (3, 3, 3), (2, 2, 2), pad))
                else:
                    steps.append(('block', pre, cout, cout))
                cin = cout
        steps.append(('down', 'conv_out', cin, self.output_channels, (3, 1, 1), (2, 1, 1), (0, 0, 0)))
        return steps

    def state_keys(self):
        keys = []
        for st in self.plan():
            if st[0] == 'block':
                for n in (1, 2):
                    keys += [f'{st[1]}.conv{n}.weight'] + [f'{st[1]}.bn{n}.{p}' for p in ('weight', 'bias', 'running_mean', 'running_var')]
            else:
                keys += [f'{st[1]}.0.weight'] + [f'{st[1]}.1.{p}' for p in ('weight', 'bias', 'running_mean', 'running_var')]
        return keys

    def synthetic_state(self, seed=0):
        """Random weights of the checkpoint's shapes (no checkpoint here): He-scaled, BatchNorm statistics off the identity."""
        g = torch.Generator().manual_seed(seed)
        sd = {}

        def conv(name, cout, ks, cin):
            fan = ks[0] * ks[1] * ks[2] * cin
            sd[name] = torch.randn(cout, *ks, cin, generator=g) * (2.0 / fan) ** 0.5

        def bn(name, c):
            sd[f'{name}.weight'] = torch.rand(c, generator=g) * 0.5 + 0.75
            sd[f'{name}.bias'] = torch.randn(c, generator=g) * 0.1
            sd[f'{name}.running_mean'] = torch.randn(c, generator=g) * 0.1
            sd[f'{name}.running_var'] = torch.rand(c, generator=g) * 0.5 + 0.75
        for st in self.plan():
            if st[0] == 'block':
                for n in (1, 2):
                    conv(f'{st[1]}.conv{n}.weight', st[3], (3, 3, 3), st[2])
                    bn(f'{st[1]}.bn{n}', st[3])
            else:
                ks = (3, 3, 3) if st[0] == 'subm' else st[4]
                conv(f'{st[1]}.0.weight', st[3], ks, st[2])
                bn(f'{st[1]}.1', st[3])
        return sd

    def load_mmdet_state(self, sd):
        missing = [k for k in self.state_keys() if k not in sd]
        assert not missing, f'missing SparseEncoder entries: {missing[:4]}'
        self._p, self._geom = {}, {}
        for st in self.plan():
            if st[0] == 'block':
                for n in (1, 2):
                    self._p[f'{st[1]}.{n}'] = _fold(f'{st[1]}.conv{n}.weight', f'{st[1]}.bn{n}', st[1], sd, (3, 3, 3), st[2], st[3])
                    self._geom[f'{st[1]}.{n}'] = (27, st[2])
            else:
                ks = (3, 3, 3) if st[0] == 'subm' else st[4]
                self._p[st[1]] = _fold(f'{st[1]}.0.weight', f'{st[1]}.1', st[1], sd, ks, st[2], st[3])
                self._geom[st[1]] = (ks[0] * ks[1] * ks[2], st[2])
        self._master = dict(self._p)                          # float32, where load_mmdet_state ran
        return self

    def to(self, *a, **k):
        dev = a[0] if a else k.get('device')
        self._p = {n: (w.to(dev, self.dtype), b.to(dev, self.dtype)) for n, (w, b) in self._master.items()}
        self._h = {}
        if torch.device(dev).type == 'cuda' and self.dtype == torch.float16:
            # the HIP kernels' form: weights as MFMA operand fragments (csrc/sparse_conv.hip), float32 bias
            from .... import ops
            for n, (w, b) in self._master.items():
                K, cin = self._geom[n]
                frag, cin_pad = ops.sparse_weight_fragments(w, K, cin)
                self._h[n] = (frag.to(dev), b.float().to(dev), cin_pad, w.shape[1])
        return self

    @torch.no_grad()
    def _forward_hip(self, voxel_features, coors, batch_size):
        """The same layers on `csrc/sparse_conv.hip`: a level = its sorted int32 keys + one fp16 feature row per key; rulebooks by
        `di_sparse_mark` / `di_sparse_nbr`, every convolution ONE launch (gather, MFMA product, bias / residual / ReLU)."""
        from .... import ops
        shape = self.sparse_shape
        D, H, W = shape
        assert batch_size * D * H * W < 2 ** 31, 'linear voxel keys are 32-bit'
        c = coors.long()
        keys, order = torch.sort((((c[:, 0] * D + c[:, 1]) * H + c[:, 2]) * W + c[:, 3]).to(torch.int32))
        feats = ops.sparse_rows(voxel_features[order])       # (M + 1, 8): channels padded, the zero row missing neighbours read
        nbr = rows = None                                     # per level: the submanifold table, the row table of the sorted keys
        for st in self.plan():
            if rows is None:
                rows = ops.sparse_rowstart(keys, batch_size, shape)
            if st[0] in ('subm', 'block') and nbr is None:
                nbr = ops.sparse_neighbours(keys, keys, batch_size, shape, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), rows)
            if st[0] == 'subm':
                feats = ops.sparse_conv(feats, nbr, *self._h[st[1]], relu=True)
            elif st[0] == 'block':
                h = ops.sparse_conv(feats, nbr, *self._h[f'{st[1]}.1'], relu=True)
                feats = ops.sparse_conv(h, nbr, *self._h[f'{st[1]}.2'], relu=True, residual=feats)
            else:
                okeys, oshape = ops.sparse_output_keys(keys, batch_size, shape, st[4], st[5], st[6])
                onbr = ops.sparse_neighbours(keys, okeys, batch_size, shape, oshape, st[4], st[5], st[6], rows)
                feats = ops.sparse_conv(feats, onbr, *self._h[st[1]], relu=True)
                keys, shape, nbr, rows = okeys, oshape, None, None
        D, H, W = shape
        d = feats.new_zeros((batch_size * D * H * W, feats.shape[1]))
        d[keys.long()] = feats[:-1]
        return d.view(batch_size, D, H, W, -1).permute(0, 4, 1, 2, 3).reshape(batch_size, -1, H, W)

    @torch.no_grad()
    def forward(self, voxel_features, coors, batch_size):
        """voxel_features (M, in_channels), coors (M, 4) int [b, z, y, x] -> dense BEV map (B, C * D_out, H / 8, W / 8)."""
        if voxel_features.is_cuda and getattr(self, '_h', None) and self.HIP:
            return self._forward_hip(voxel_features, coors, batch_size)
        coords = coors.long()
        x = SparseTensor(coords, voxel_features.to(self.dtype), self.sparse_shape, batch_size)
        order = torch.argsort(x.keys())
        x.coords, x.feats = x.coords[order], x.feats[order]
        nbr = None                                            # the submanifold rulebook of the current resolution
        for st in self.plan():
            if st[0] in ('subm', 'block') and nbr is None:
                nbr = subm_rulebook(x)
            if st[0] == 'subm':
                w, b = self._p[st[1]]
                x.feats = sparse_conv(x.feats, nbr, w, b, relu=True)
            elif st[0] == 'block':                            # SparseBasicBlock: conv-bn-relu, conv-bn, + identity, relu
                (w1, b1), (w2, b2) = self._p[f'{st[1]}.1'], self._p[f'{st[1]}.2']
                h = sparse_conv(x.feats, nbr, w1, b1, relu=True)
                x.feats = sparse_conv(h, nbr, w2, b2, relu=True, residual=x.feats)
            else:
                w, b = self._p[st[1]]
                ocoords, onbr, oshape = strided_rulebook(x, st[4], st[5], st[6])
                x = SparseTensor(ocoords, sparse_conv(x.feats, onbr, w, b, relu=True), oshape, batch_size)
                nbr = None
        d = x.dense()                                         # (B, C, D, H, W)
        B, C, D, H, W = d.shape
        return d.reshape(B, C * D, H, W)


class _FrozenConvStack(nn.Module):
    """Dense conv / deconv + BatchNorm (folded) + ReLU stacks through torch (MIOpen), channels-last."""

    def __init__(self, dtype):
        super().__init__()
        self.dtype = dtype
        self._p = {}

    @staticmethod
    def _fold2d(w, sd, bn, transposed, eps):
        g, b = sd[f'{bn}.weight'].float(), sd[f'{bn}.bias'].float()
        scale = g / torch.sqrt(sd[f'{bn}.running_var'].float() + eps)
        w = w.float() * (scale[None, :, None, None] if transposed else scale[:, None, None, None])
        return w, b - sd[f'{bn}.running_mean'].float() * scale

    def to(self, *a, **k):
        dev = a[0] if a else k.get('device')
        self._p = {n: tuple(t.to(dev, self.dtype) if torch.is_tensor(t) else t for t in v) for n, v in self._p.items()}
        return self


class FrozenSECOND(_FrozenConvStack):
    """mmdet3d `SECOND` (`backbones/second.py`): per stage `Conv2d(3x3, stride) + BN + ReLU` then `layer_num` x
    (`Conv2d(3x3) + BN + ReLU`); returns every stage's map.  Keys `blocks.{i}.{3 k}.weight`, `blocks.{i}.{3 k + 1}.*`."""

    def __init__(self, in_channels=256, out_channels=(128, 256), layer_nums=(5, 5), layer_strides=(1, 2), eps=1e-3,
                 dtype=torch.float16):
        super().__init__(dtype)
        self.in_channels, self.out_channels, self.layer_nums, self.layer_strides, self.eps = \
            in_channels, tuple(out_channels), tuple(layer_nums), tuple(layer_strides), eps

    def plan(self):
        cin = self.in_channels
        for i, (co, n, s) in enumerate(zip(self.out_channels, self.layer_nums, self.layer_strides)):
            for k in range(n + 1):
                yield i, k, (cin if k == 0 else co), co, (s if k == 0 else 1)
            cin = co

    def synthetic_state(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for i, k, ci, co, s in self.plan():
            sd[f'blocks.{i}.{3 * k}.weight'] = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5
            for p, v in (('weight', torch.rand(co, generator=g) * 0.5 + 0.75), ('bias', torch.randn(co, generator=g) * 0.1),
                         ('running_mean', torch.randn(co, generator=g) * 0.1), ('running_var', torch.rand(co, generator=g) * 0.5 + 0.75)):
                sd[f'blocks.{i}.{3 * k + 1}.{p}'] = v
        return sd

    def load_mmdet_state(self, sd):
        self._p, self._master = {}, {}
        for i, k, ci, co, s in self.plan():
            w, b = self._fold2d(sd[f'blocks.{i}.{3 * k}.weight'], sd, f'blocks.{i}.{3 * k + 1}', False, self.eps)
            self._p[(i, k)] = (w.contiguous(memory_format=torch.channels_last), b, s)
            self._master[(i, k)] = (w, b, s, ci, co)
        return self

    HIP = os.environ.get('DI_SPARSE_HIP', '1') != '0'

    def to(self, *a, **k):
        super().to(*a, **k)
        dev = a[0] if a else k.get('device')
        self._h = {}
        if torch.device(dev).type == 'cuda' and self.dtype == torch.float16:
            # stride-1 layers of 128 output channels (the whole first stage of the reference configuration) run on the hot
            # path's own 3x3 kernel (csrc/conv3x3.hip, BatchNorm folded, ReLU in the epilogue); the rest through MIOpen
            from .... import ops
            self._b32 = {key: v[1].float().to(dev) for key, v in self._master.items()}
            for key, (w, b, s, ci, co) in self._master.items():
                if s == 1 and co == 128 and ci % 32 == 0:
                    wp, ws, bias = ops.pack_conv3x3(w, b)
                    self._h[key] = (wp.to(dev), ws.to(dev), bias.to(dev))
        return self

    @torch.no_grad()
    def forward(self, x):
        x = x.to(self.dtype).contiguous(memory_format=torch.channels_last)
        outs = []
        for i in range(len(self.out_channels)):
            for k in range(self.layer_nums[i] + 1):
                h = getattr(self, '_h', {}).get((i, k)) if (x.is_cuda and self.HIP) else None
                if h is not None and x.shape[2] >= 12:
                    from .... import ops
                    x = ops.conv3x3(x, h[0], h[1], h[2], relu=True)
                else:
                    w, b, s = self._p[(i, k)]
                    b32 = getattr(self, '_b32', {}).get((i, k)) if (x.is_cuda and self.HIP and x.dtype == torch.float16) else None
                    if b32 is not None and w.shape[0] % 8 == 0:        # library convolution + the one-pass bias / ReLU epilogue
                        from .... import ops
                        y = F.conv2d(x, w, None, stride=s, padding=1)
                        x = ops.bias_act_(y if y.is_contiguous(memory_format=torch.channels_last)
                                          else y.contiguous(memory_format=torch.channels_last), b32, None, True)
                    else:
                        x = torch.relu_(F.conv2d(x, w, b, stride=s, padding=1))
            outs.append(x)
        return tuple(outs)


class FrozenSECONDFPN(_FrozenConvStack):
    """mmdet3d `SECONDFPN` (`necks/second_fpn.py`): per input a `ConvTranspose2d(k = stride)` (stride > 1) or - with
    `use_conv_for_no_stride` - a `Conv2d(k = stride)` (stride 1), + BN + ReLU; the outputs concatenated.  Keys `deblocks.{i}.0 / .1`."""

    def __init__(self, in_channels=(128, 256), out_channels=(256, 256), upsample_strides=(1, 2), use_conv_for_no_stride=True,
                 eps=1e-3, dtype=torch.float16):
        super().__init__(dtype)
        self.in_channels, self.out_channels, self.upsample_strides = tuple(in_channels), tuple(out_channels), tuple(upsample_strides)
        self.use_conv_for_no_stride, self.eps = use_conv_for_no_stride, eps
        self._b32 = {}

    def _transposed(self, s):
        return s > 1 or (s == 1 and not self.use_conv_for_no_stride)

    def synthetic_state(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for i, (ci, co, s) in enumerate(zip(self.in_channels, self.out_channels, self.upsample_strides)):
            shape = (ci, co, s, s) if self._transposed(s) else (co, ci, s, s)
            sd[f'deblocks.{i}.0.weight'] = torch.randn(*shape, generator=g) * (2.0 / (s * s * ci)) ** 0.5
            for p, v in (('weight', torch.rand(co, generator=g) * 0.5 + 0.75), ('bias', torch.randn(co, generator=g) * 0.1),
                         ('running_mean', torch.randn(co, generator=g) * 0.1), ('running_var', torch.rand(co, generator=g) * 0.5 + 0.75)):
                sd[f'deblocks.{i}.1.{p}'] = v
        return sd

    def load_mmdet_state(self, sd):
        self._p = {}
        for i, s in enumerate(self.upsample_strides):
            t = self._transposed(s)
            w, b = self._fold2d(sd[f'deblocks.{i}.0.weight'], sd, f'deblocks.{i}.1', t, self.eps)
            self._p[i] = (w, b, s, t)
        return self

    @torch.no_grad()
    def forward(self, xs):
        ups = []
        for i, x in enumerate(xs):
            w, b, s, t = self._p[i]
            if x.is_cuda and x.dtype == torch.float16 and b.numel() % 8 == 0 and FrozenSECOND.HIP:
                from .... import ops                         # library (de)convolution + the one-pass bias / ReLU epilogue
                y = F.conv_transpose2d(x, w, None, stride=s) if t else F.conv2d(x, w, None, stride=s)
                if not y.is_contiguous(memory_format=torch.channels_last):
                    y = y.contiguous(memory_format=torch.channels_last)
                if i not in self._b32:
                    self._b32[i] = b.float().contiguous()
                ups.append(ops.bias_act_(y, self._b32[i], None, True))
            else:
                y = F.conv_transpose2d(x, w, b, stride=s) if t else F.conv2d(x, w, b, stride=s)
                ups.append(torch.relu_(y))
        out = torch.cat(ups, 1) if len(ups) > 1 else ups[0]
        return [out.contiguous(memory_format=torch.channels_last)]


class FrozenLidarBackbone(nn.Module):
    """points (list of (Np, >= 5) float tensors on the device) -> [BEV map (B, 512, H / 8, W / 8)]: reference
    `extract_pts_feat` lines :124-131 (voxelise 'voxel', HardSimpleVFE, SparseEncoder, SECOND, SECONDFPN)."""

    def __init__(self, pts_voxel_layer, middle, backbone, neck, num_features=5):
        super().__init__()
        self.pts_voxel_layer = pts_voxel_layer if isinstance(pts_voxel_layer, nn.Module) else SPConvVoxelization(**pts_voxel_layer)
        self.middle, self.backbone, self.neck, self.num_features = middle, backbone, neck, num_features

    @torch.no_grad()
    def forward(self, points):
        feats, coors = [], []
        for i, res in enumerate(points):
            v, c, n = self.pts_voxel_layer(res)                         # (M, T, 5), (M, 3) [z, y, x], (M,)
            # HardSimpleVFE: mean of a voxel's points
            feats.append(v[:, :, :self.num_features].sum(1) / n.clamp(min=1).to(v.dtype).unsqueeze(-1))
            coors.append(F.pad(c.long(), (1, 0), value=i))
        x = self.middle(torch.cat(feats, 0), torch.cat(coors, 0), len(points))
        return self.neck(self.backbone(x))

    @classmethod
    def synthetic(cls, voxel_layer_cfg, sparse_shape, device, dtype=torch.float16, seed=0):
        """The reference configuration with random weights of the checkpoint's shapes."""
        mid = FrozenSparseEncoder(sparse_shape=sparse_shape, dtype=dtype)
        mid.load_mmdet_state(mid.synthetic_state(seed)).to(device)
        bb = FrozenSECOND(dtype=dtype)
        bb.load_mmdet_state(bb.synthetic_state(seed + 1)).to(device)
        nk = FrozenSECONDFPN(dtype=dtype)
        nk.load_mmdet_state(nk.synthetic_state(seed + 2)).to(device)
        return cls(voxel_layer_cfg, mid, bb, nk)
