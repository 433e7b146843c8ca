"""MI355X-native building blocks of the MMPI decoder.

Drop-in for the reference module of the same path
(`projects/mmdet3d_plugin/models/utils/decoder_utils.py`): same class names, constructor
arguments and `state_dict` keys -

    PositionEmbeddingLearned :16-32, TransformerDecoderLayer :35-113, MultiheadAttention
    :116-243, FFN :498-581, DynamicConv :584-629, ImageRCNNBlock :632-761, PointRCNNBlock :765-841

Execution differs where the reference is host-bound: the per-sample / per-view Python loops
with data-dependent `continue`s and boolean indexing (:681-759) become one batched, shape-static
pass over all (sample, view, query) triples - per-view query subsets are attention masks, the
"<= 1 query in view => skip" rule and the "later view overwrites" rule are evaluated on the
device - so a forward issues no device->host synchronisation.  Box decode, corner projection,
rectangles, RoIAlign and the 200 x 32400 cross attention are HIP kernels (C ABI in
include/deepinteraction_hip.h); the dense projections are library GEMMs.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from ....autograd import PixelLinear, RoIAlign, WideLinear
from ....geometry import aug_affine
from ....utils import param_key


# measurement switch: 1 = the library's single-GEMM gradients in the decoder's training path
SLABBED_GEMMS = os.environ.get('DI_TRAIN_PLAIN_GEMMS', '0') != '1'


class PointwiseConv1d(nn.Conv1d):
    """nn.Conv1d; with kernel size 1 on the device it is evaluated as the matrix product it is.  The library convolution path for
    these shapes ((B, 256 | 64, 200 tokens), 60 calls per training step) is layout transposes + a generic convolution kernel forward
    and a naive one backward - 2.7 ms of a 32 ms step; as (O, C) @ (B, C, Q) it is a GEMM and its autograd.  Same parameters, same
    `state_dict` keys."""

    def forward(self, x):
        if (x.is_cuda and x.dim() == 3 and self.kernel_size == (1,) and self.stride == (1,) and self.padding == (0,)
                and self.dilation == (1,) and self.groups == 1):
            y = torch.matmul(self.weight.squeeze(-1), x)
            return y if self.bias is None else y + self.bias.view(1, -1, 1)
        return super().forward(x)


class ConvModule(nn.Module):
    """mmcv `ConvModule(conv -> BN -> ReLU)` surface used by the head (sub-modules `.conv`,
    `.bn`, `.activate`; bias='auto' => no conv bias when a norm follows)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias='auto',
                 conv_cfg=None, norm_cfg=None, **kwargs):
        super().__init__()
        typ = 'Conv2d' if conv_cfg is None else conv_cfg['type']
        conv = {'Conv1d': PointwiseConv1d, 'Conv2d': nn.Conv2d}[typ]
        self.with_norm = norm_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = conv(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)
        if self.with_norm:
            bn = {'BN1d': nn.BatchNorm1d, 'BN2d': nn.BatchNorm2d, 'BN': nn.BatchNorm2d}[norm_cfg['type']]
            self.bn = bn(out_channels)
        self.activate = nn.ReLU(inplace=True)
        nn.init.kaiming_normal_(self.conv.weight, mode='fan_out', nonlinearity='relu')

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = self.bn(x)
        return self.activate(x)


def build_conv_layer(cfg, *args, **kwargs):
    typ = 'Conv2d' if cfg is None else cfg['type']
    return {'Conv1d': PointwiseConv1d, 'Conv2d': nn.Conv2d}[typ](*args, **kwargs)


class PositionEmbeddingLearned(nn.Module):
    """Reference :16-32.  `tokens()` is the inference form: float32 (coordinates reach 180 BEV
    cells - too coarse in fp16), BatchNorm folded, token-major output."""

    def __init__(self, input_channel, num_pos_feats=288):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            PointwiseConv1d(input_channel, num_pos_feats, kernel_size=1), nn.BatchNorm1d(num_pos_feats),
            nn.ReLU(inplace=True), PointwiseConv1d(num_pos_feats, num_pos_feats, kernel_size=1))
        self._fold_cache = None

    def forward(self, xyz):
        xyz = xyz.transpose(1, 2).contiguous()
        return self.position_embedding_head(xyz)

    def tokens(self, xyz, out_dtype):
        """xyz (B,P,k) -> (B,P,C) in `out_dtype`."""
        if self.training or torch.is_grad_enabled():
            return self.forward(xyz.to(self.position_embedding_head[0].weight.dtype)).transpose(1, 2).to(out_dtype)
        c1, bn, _, c2 = self.position_embedding_head
        key = param_key(self)
        if self._fold_cache is None or self._fold_cache[0] != key:
            with torch.no_grad():
                g = torch.rsqrt(bn.running_var.float() + bn.eps) * bn.weight.float()
                w1 = c1.weight.float()[:, :, 0] * g[:, None]
                b1 = (c1.bias.float() - bn.running_mean.float()) * g + bn.bias.float()
                w2 = c2.weight.detach().to(torch.float32, copy=True)[:, :, 0].contiguous()
                self._fold_cache = (key, (w1, b1, w2, c2.bias.detach().to(torch.float32, copy=True)))
        w1, b1, w2, b2 = self._fold_cache[1]
        return F.linear(torch.relu_(F.linear(xyz.float(), w1, b1)), w2, b2).to(out_dtype)


class MultiheadAttention(nn.Module):
    """Reference :116-243 (parameters `in_proj_weight`, `in_proj_bias`, `out_proj.*`).
    `forward` keeps the (L,N,E) sequence-first API; the decoder layer uses `heads()` pieces."""

    def __init__(self, embed_dim, num_heads, dropout=0., bias=True, **unused):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, query, key, value, need_weights=False, **unused):
        o = mha_tokens(query.transpose(0, 1), key.transpose(0, 1), value.transpose(0, 1), self.in_proj_weight,
                       self.in_proj_bias, self.out_proj, self.num_heads, self.dropout if self.training else 0.0)
        return o.transpose(0, 1), None


def _post_norm(norm, x, branch=None):
    """norm(x + branch): one fused kernel at inference (see transformer_bricks.post_norm)."""
    from .transformer_bricks import post_norm
    return post_norm(norm, x, branch)


def mha_tokens(q_in, k_in, v_in, w, b, out_proj, H, dropout_p=0.0, key_allowed=None):
    """Batch-first multi-head attention on small token sets with library GEMMs.
    q_in (B,L,E), k_in/v_in (B,S,E); key_allowed: optional bool (B,L,S) - query l attends only to the
    keys s with key_allowed[b,l,s] (a row without any allowed key yields NaN; callers drop it).
    Returns (B,L,E)."""
    B, L, E = q_in.shape
    S = k_in.shape[1]
    D = E // H
    if k_in is q_in and v_in is q_in:
        qkv = F.linear(q_in, w, b)
        q, k, v = qkv.split(E, dim=-1)
    else:
        q = F.linear(q_in, w[:E], b[:E])
        k = F.linear(k_in, w[E:2 * E], b[E:2 * E])
        v = F.linear(v_in, w[2 * E:], b[2 * E:])
    q = q.view(B, L, H, D).transpose(1, 2) * (float(D) ** -0.5)
    k = k.view(B, S, H, D).transpose(1, 2)
    v = v.view(B, S, H, D).transpose(1, 2)
    sc = torch.matmul(q, k.transpose(-1, -2)).float()                      # (B,H,L,S)
    if key_allowed is not None:
        sc = sc.masked_fill(~key_allowed.view(B, 1, L, S), float('-inf'))
    a = torch.softmax(sc, -1).to(v.dtype)
    if dropout_p > 0:
        a = F.dropout(a, dropout_p)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, L, E)
    return out_proj(o)


class TransformerDecoderLayer(nn.Module):
    """Reference :35-113 (post-norm; positional embeddings added to q, k and v)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation='relu',
                 self_posembed=None, cross_posembed=None, cross_only=False):
        super().__init__()
        self.cross_only = cross_only
        if not cross_only:
            self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1, self.dropout2, self.dropout3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)
        self.activation = {'relu': F.relu, 'gelu': F.gelu, 'glu': F.glu}[activation]
        self.self_posembed, self.cross_posembed = self_posembed, cross_posembed
        self._kpe_cache = None

    def key_pos_embed(self, key_pos, dtype):
        """cross_posembed(key_pos) as tokens (B,Pk,C).  The BEV grid is constant, so at
        inference the embedding is computed once and reused (the reference recomputes the
        32 400-token Conv1d stack on every call)."""
        if self.training or torch.is_grad_enabled():
            return self.cross_posembed.tokens(key_pos, dtype)
        key = (key_pos.data_ptr(), key_pos.shape, dtype, param_key(self.cross_posembed))
        if self._kpe_cache is None or self._kpe_cache[0] != key:
            with torch.no_grad():
                pe = self.cross_posembed.tokens(key_pos[:1], dtype).contiguous()
            self._kpe_cache = (key, pe, key_pos)          # keep key_pos alive: its data_ptr is in the key
        return self._kpe_cache[1]

    def forward(self, query, key, query_pos, key_pos, attn_mask=None):
        """query (B,C,Pq), key (B,C,Pk), query_pos (B,Pq,2), key_pos (B,Pk,2) -> (B,C,Pq)."""
        dt = query.dtype
        qpe = self.self_posembed.tokens(query_pos, dt) if self.self_posembed is not None else 0
        kpe = self.key_pos_embed(key_pos, dt) if self.cross_posembed is not None else 0
        x = query.transpose(1, 2)                                              # (B,Pq,C)
        ktok = key.transpose(1, 2)                                             # (B,Pk,C): a view of channels-last
        if not self.cross_only:
            sa = self.self_attn
            qq = x + qpe
            x = _post_norm(self.norm1, x, self.dropout1(mha_tokens(qq, qq, qq, sa.in_proj_weight, sa.in_proj_bias,
                                                                   sa.out_proj, sa.num_heads,
                                                                   sa.dropout if self.training else 0.0)))
        ca = self.multihead_attn
        E = ca.embed_dim
        q = F.linear(x + qpe, ca.in_proj_weight[:E], ca.in_proj_bias[:E])
        kin = ktok + kpe
        if kin.is_cuda and torch.is_grad_enabled() and ca.in_proj_weight.requires_grad and SLABBED_GEMMS:
            # training: the weight gradient reduces over the 32 400 keys - through the slab-batched GEMM of PixelLinear
            kv = PixelLinear.apply(kin.reshape(-1, E), ca.in_proj_weight[E:], ca.in_proj_bias[E:]).view(*kin.shape[:-1], 2 * E)
        else:
            kv = F.linear(kin, ca.in_proj_weight[E:], ca.in_proj_bias[E:])     # (B,Pk,2C) = [K | V]
        if x.is_cuda and not (torch.is_grad_enabled() and (q.requires_grad or kv.requires_grad)):
            o = ops.mha_decode(q, kv, ca.num_heads, float(ca.head_dim) ** -0.5)
        else:
            # training: the (B,H,Q,32400) score tensor is materialised, as in the reference (:471-485),
            # through library GEMMs and their autograd
            Bq, Lq, _ = q.shape
            H, D = ca.num_heads, ca.head_dim
            qh = q.view(Bq, Lq, H, D).transpose(1, 2) * (float(D) ** -0.5)
            kh = kv[..., :E].reshape(Bq, -1, H, D).transpose(1, 2)
            vh = kv[..., E:].reshape(Bq, -1, H, D).transpose(1, 2)
            sc = torch.matmul(qh, kh.transpose(-1, -2))
            if sc.dtype == torch.float16:
                # mixed precision: the soft-max kernel accumulates in float32 whatever its operand type; fp16 in / fp16 out is the
                # value of float() -> softmax -> half() without two cast passes and a float32 copy of the (H, Q, 32 400) scores
                # in each direction (autocast would up-cast the operand: switched off for this one call)
                with torch.autocast('cuda', enabled=False):
                    a = torch.softmax(sc, -1)
            else:
                a = torch.softmax(sc.float(), -1).to(vh.dtype)
            if self.training and ca.dropout > 0:
                a = F.dropout(a, ca.dropout)
            o = torch.matmul(a, vh).transpose(1, 2).reshape(Bq, Lq, E)
        x = _post_norm(self.norm2, x, self.dropout2(ca.out_proj(o)))
        x = _post_norm(self.norm3, x, self.dropout3(self.linear2(self.dropout(self.activation(self.linear1(x))))))
        return x.transpose(1, 2)


class FFN(nn.Module):
    """Reference :498-581: per head ConvModule(in->64, BN1d, ReLU) + Conv1d(64->classes).
    At inference the heads are evaluated together in float32 (BN folded, first layers stacked
    into one GEMM, second layers block-diagonal): box regression feeds geometry, so it is kept
    out of fp16."""

    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, init_bias=-2.19,
                 conv_cfg=dict(type='Conv1d'), norm_cfg=dict(type='BN1d'), bias='auto', **kwargs):
        super().__init__()
        self.heads = heads
        self.init_bias = init_bias
        for head in self.heads:
            classes, num_conv = self.heads[head]
            layers, c_in = [], in_channels
            for _ in range(num_conv - 1):
                layers.append(ConvModule(c_in, head_conv, kernel_size=final_kernel, stride=1,
                                         padding=final_kernel // 2, bias=bias, conv_cfg=conv_cfg,
                                         norm_cfg=norm_cfg))
                c_in = head_conv
            layers.append(build_conv_layer(conv_cfg, head_conv, classes, kernel_size=final_kernel, stride=1,
                                           padding=final_kernel // 2, bias=True))
            self.__setattr__(head, nn.Sequential(*layers))
        self._fold_cache = None

    def init_weights(self):
        for head in self.heads:
            if head == 'heatmap':
                self.__getattr__(head)[-1].bias.data.fill_(self.init_bias)

    def _fusable(self):
        return all(len(getattr(self, h)) == 2 and getattr(self, h)[0].conv.kernel_size == (1,)
                   for h in self.heads)

    def folded(self):
        key = param_key(self)
        if self._fold_cache is not None and self._fold_cache[0] == key:
            return self._fold_cache[1]
        with torch.no_grad():
            w1, b1, w2rows, b2 = [], [], [], []
            n = len(self.heads)
            for i, h in enumerate(self.heads):
                cm, last = getattr(self, h)[0], getattr(self, h)[1]
                W = cm.conv.weight.float().flatten(1)
                bias = torch.zeros(W.shape[0], device=W.device) if cm.conv.bias is None else cm.conv.bias.float()
                if cm.with_norm:
                    g = torch.rsqrt(cm.bn.running_var.float() + cm.bn.eps) * cm.bn.weight.float()
                    W, bias = W * g[:, None], (bias - cm.bn.running_mean.float()) * g + cm.bn.bias.float()
                w1.append(W)
                b1.append(bias)
                L = last.weight.float().flatten(1)                        # (classes, 64)
                row = torch.zeros(L.shape[0], n * L.shape[1], device=L.device)
                row[:, i * L.shape[1]:(i + 1) * L.shape[1]] = L
                w2rows.append(row)
                b2.append(last.bias.float())
            out = (torch.cat(w1), torch.cat(b1), torch.cat(w2rows), torch.cat(b2),
                   [self.heads[h][0] for h in self.heads])
        self._fold_cache = (key, out)
        return out

    def forward(self, x):
        if self.training or torch.is_grad_enabled() or not self._fusable():
            return {h: self.__getattr__(h)(x) for h in self.heads}
        W1, b1, W2, b2, sizes = self.folded()
        t = x.float().transpose(1, 2)                                      # (B,Q,Cin)
        y = F.linear(torch.relu_(F.linear(t, W1, b1)), W2, b2).transpose(1, 2)   # (B,sum classes,Q)
        return dict(zip(self.heads, y.split(sizes, dim=1)))


class DynamicConv(nn.Module):
    """Reference :584-629."""

    def __init__(self, cfg=None):
        super().__init__()
        self.hidden_dim = 128
        self.dim_dynamic = 128
        self.num_dynamic = 2
        self.num_params = self.hidden_dim * self.dim_dynamic
        self.dynamic_layer = nn.Linear(self.hidden_dim, self.num_dynamic * self.num_params)
        self.norm1 = nn.LayerNorm(self.dim_dynamic)
        self.norm2 = nn.LayerNorm(self.hidden_dim)
        self.activation = nn.ReLU()
        pooler_resolution = 7
        self.out_layer = nn.Linear(self.hidden_dim * pooler_resolution ** 2, self.hidden_dim)
        self.norm3 = nn.LayerNorm(self.hidden_dim)

    def forward(self, pro_features, roi_features):
        """pro (1,n,128), roi (49,n,128) -> (n,128) (reference layout)."""
        return self.forward_nk(pro_features[0], roi_features.permute(1, 0, 2))

    def forward_nk(self, pro, roi):
        """pro (n,128), roi (n,49,128) (the RoIAlign kernel's native output) -> (n,128)."""
        dl = self.dynamic_layer
        if pro.is_cuda and torch.is_grad_enabled() and dl.weight.requires_grad and pro.dim() == 2 and SLABBED_GEMMS:
            params = WideLinear.apply(pro, dl.weight, dl.bias)              # training: slabbed input gradient
        else:
            params = dl(pro)                                                # (n, 2*128*128)
        p1 = params[:, :self.num_params].view(-1, self.hidden_dim, self.dim_dynamic)
        p2 = params[:, self.num_params:].view(-1, self.dim_dynamic, self.hidden_dim)
        f = self.activation(_post_norm(self.norm1, torch.bmm(roi, p1)))
        f = self.activation(_post_norm(self.norm2, torch.bmm(f, p2)))
        f = self.out_layer(f.flatten(1))                                    # 49-major flatten (:624)
        return self.activation(_post_norm(self.norm3, f))


class QueryGeometry:
    """Per-forward device constants of the RoI blocks: lidar2img (B,V,4,4), reverse augmentation
    affine (B,12) and [w, h, flip, orig_w, crop_x, crop_y] (B,6) - one host->device copy."""

    def __init__(self, img_metas, device):
        host, B, V = self._pack(img_metas)
        buf = host.to(device)
        self._buf = buf
        self.B, self.V = B, V
        self.proj = buf[:B * V * 16].view(B, V, 4, 4)
        self.aug_rev = buf[B * V * 16:B * V * 16 + B * 12].view(B, 12)
        self.per_sample = buf[B * V * 16 + B * 12:].view(B, 6)

    @staticmethod
    def _pack(img_metas):
        l2i = np.stack([np.asarray(m['lidar2img'], dtype=np.float32) for m in img_metas])
        aug = np.stack([aug_affine(m, True) for m in img_metas]).astype(np.float32)
        ps = []
        for m in img_metas:
            h, w = m['input_shape'][:2]
            flip = 1.0 if m.get('flip', False) else 0.0
            orig_w = float(m['img_shape'][0][1]) if 'img_shape' in m else float(w)
            crop = m.get('img_crop_offset', (0.0, 0.0))
            ps.append([float(w), float(h), flip, orig_w, float(crop[0]), float(crop[1])])
        B, V = l2i.shape[:2]
        host = torch.from_numpy(np.concatenate([l2i.reshape(-1), aug.reshape(-1),
                                                np.asarray(ps, np.float32).reshape(-1)]))
        return host, B, V

    def update(self, img_metas):
        """Refresh IN PLACE for new samples (same batch / view count): addresses stay valid for a
        captured hipGraph."""
        host, B, V = self._pack(img_metas)
        assert (B, V) == (self.B, self.V), 'batch or view count changed: rebuild the geometry'
        self._buf.copy_(host)


def _roi_align(feat, rois, scale):
    """RoIAlign; with autograd w.r.t. the feature maps when they need a gradient (training)."""
    if torch.is_grad_enabled() and feat.requires_grad:
        return RoIAlign.apply(feat, rois.detach(), scale)
    return ops.roi_align(feat, rois, scale)


class _RCNNBase(nn.Module):
    def _stack(self, x, roi, sfx, key_allowed=None):
        """decoder_utils.py:743-756 / :824-837 on tokens x (B,Q,C): self-attn + LN, DynamicConv + LN,
        FFN(GELU) + LN.  key_allowed (B,Q,Q): the keys each query may attend to."""
        g = lambda n: getattr(self, n + sfx)
        sa = g('dyconv_pre_self_attn')
        p = sa.dropout if self.training else 0.0
        a = mha_tokens(x, x, x, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj, sa.num_heads, p, key_allowed)
        x = _post_norm(g('norm1'), x, g('dropout1')(a))
        shp = x.shape
        dy = g('dyconv').forward_nk(x.reshape(-1, shp[-1]), roi)
        x = _post_norm(g('norm2'), x, g('dropout2')(dy.view(shp)))
        x = _post_norm(g('norm3'), x, g('dropout3')(g('linear2')(g('dropout')(g('activation')(g('linear1')(x))))))
        return x


class ImageRCNNBlock(_RCNNBase):
    """Reference :632-761 - per-query RoI feature sampling on the image feature maps."""

    def __init__(self, num_views, num_proposals, out_size_factor_img, test_cfg, bbox_coder, hidden_channel,
                 num_heads, dropout):
        super().__init__()
        self.num_views, self.num_proposals = num_views, num_proposals
        self.out_size_factor_img, self.test_cfg, self.bbox_coder = out_size_factor_img, test_cfg, bbox_coder
        self.dyconv = DynamicConv(None)
        self.dyconv_pre_self_attn = nn.MultiheadAttention(hidden_channel, num_heads, dropout=dropout)
        self.dropout1, self.norm1 = nn.Dropout(dropout), nn.LayerNorm(hidden_channel)
        self.dropout2, self.norm2 = nn.Dropout(dropout), nn.LayerNorm(hidden_channel)
        self.dropout3, self.norm3 = nn.Dropout(dropout), nn.LayerNorm(hidden_channel)
        self.linear1 = nn.Linear(hidden_channel, hidden_channel * 4)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(hidden_channel * 4, hidden_channel)
        self.activation = nn.GELU()

    def forward(self, query_feat, res_layer, new_lidar_feat, img_feat_flatten, img_metas, img_h, img_w,
                geom=None, **kwargs):
        B, C, Q = query_feat.shape
        V = self.num_views
        geom = geom or QueryGeometry(img_metas, query_feat.device)
        cfg = self.test_cfg
        res32 = {k: res_layer[k].detach().float() for k in ('center', 'height', 'dim', 'rot')}
        on, rect, _ = ops.query_geometry(res32, geom.proj, geom.aug_rev, geom.per_sample,
                                         cfg['out_size_factor'] * cfg['voxel_size'][0], cfg['pc_range'][:2],
                                         1.0, 1.0, True, False)
        # Which (view, query) results survive in the reference?  A view with <= 1 centre on it is
        # skipped (:726) and a later view overwrites an earlier one (:728,:759): query q keeps the output
        # of its LAST valid view v*(q) only.  That output depends on the other queries only through the
        # self-attention among the queries of view v*(q) (:745), so the whole block is evaluated for the
        # <= Q surviving (v*(q), q) pairs instead of all V*Q - same results, 1/V of the RoI work - with
        # the per-view query subsets expressed as an attention mask.  No host synchronisation.
        onb = on.bool()
        view_ok = on.sum(-1, keepdim=True) > 1                                   # (B,V,1)
        sel = onb & view_ok                                                      # (B,V,Q)
        vid = torch.arange(V, device=on.device).view(1, V, 1)
        last = torch.where(sel, vid, torch.full_like(vid, -1)).max(1).values     # (B,Q): v*(q) or -1
        lastc = last.clamp(min=0)
        rect_q = rect.gather(1, lastc.view(B, 1, Q, 1).expand(B, 1, Q, 4)).squeeze(1)        # (B,Q,4)
        maps = img_feat_flatten.reshape(B * V, C, img_h, img_w)
        idx = (torch.arange(B, device=rect.device).view(B, 1) * V + lastc).to(torch.float32).unsqueeze(-1)
        rois = torch.cat([idx, rect_q], -1).view(-1, 5)
        roi = _roi_align(maps, rois, 1.0 / self.out_size_factor_img)             # (B*Q, 49, C)
        x = self._refine_views(query_feat.transpose(1, 2), roi, sel, last, lastc)            # (B,Q,C)
        out = torch.where((last >= 0).unsqueeze(-1), x, torch.zeros_like(x))     # unseen queries: 0 (:665)
        return out.transpose(1, 2), last.to(torch.float32)


    def _refine_views(self, x, roi, sel, last, lastc):
        """The attention / DynamicConv / FFN stack on tokens x (B,Q,C), every query inside the query set of its
        last valid view: sel (B,V,Q) membership, last (B,Q) = v*(q) or -1, lastc = last clamped to >= 0."""
        B, Q, _ = x.shape
        key_allowed = sel.gather(1, lastc.view(B, Q, 1).expand(B, Q, Q))         # [b,q,k] = sel[b, v*(q), k]
        # queries no camera sees are discarded by the caller; give them every key so that their softmax rows
        # (and the gradients flowing through the shared GEMMs in training) stay finite
        key_allowed = key_allowed | (last < 0).unsqueeze(-1)
        return self._stack(x, roi, '', key_allowed=key_allowed)


class PointRCNNBlock(_RCNNBase):
    """Reference :765-841 - per-query RoI feature sampling on the BEV map."""

    def __init__(self, hidden_channel, num_heads, dropout, bbox_coder):
        super().__init__()
        self.bbox_coder = bbox_coder
        self.dyconv_pts = DynamicConv(None)
        self.dyconv_pre_self_attn_pts = nn.MultiheadAttention(hidden_channel, num_heads, dropout=dropout)
        self.dropout1_pts, self.norm1_pts = nn.Dropout(dropout), nn.LayerNorm(hidden_channel)
        self.dropout2_pts, self.norm2_pts = nn.Dropout(dropout), nn.LayerNorm(hidden_channel)
        self.dropout3_pts, self.norm3_pts = nn.Dropout(dropout), nn.LayerNorm(hidden_channel)
        self.linear1_pts = nn.Linear(hidden_channel, hidden_channel * 4)
        self.dropout_pts = nn.Dropout(dropout)
        self.linear2_pts = nn.Linear(hidden_channel * 4, hidden_channel)
        self.activation_pts = nn.GELU()

    def forward(self, query_feat, res_layer, new_lidar_feat, img_feat_flatten, img_metas, img_h, img_w,
                geom=None, **kwargs):
        B, C, Q = query_feat.shape
        bc = self.bbox_coder
        res32 = {k: res_layer[k].detach().float() for k in ('center', 'height', 'dim', 'rot')}
        cell = bc.out_size_factor * bc.voxel_size[0]
        _, _, rect = ops.query_geometry(res32, None, None, None, cell, bc.pc_range[:2], cell, 2.0, False, True)
        idx = torch.arange(B, device=rect.device, dtype=torch.float32).view(B, 1, 1).expand(B, Q, 1)
        rois = torch.cat([idx, rect], -1).view(-1, 5)
        roi = _roi_align(new_lidar_feat, rois, 1.0)                             # (B*Q, 49, C)
        x = self._refine_all(query_feat.transpose(1, 2), roi)
        return x.transpose(1, 2), None

    def _refine_all(self, x, roi):
        return self._stack(x, roi, '_pts')


# ---------------------------------------------------------------------------------- DeepInteraction++ (V2 blocks)
class _V2Mix:
    """What `ImageRCNNBlockV2` / `PointRCNNBlockV2` (reference :844-1089) change in the refinement stack: the FFN
    becomes an mmcv `FFN` (ReLU, identity inside), and a parallel "self" branch (FFN + LayerNorm on the
    post-self-attention feature) is mixed in with the learnable `scale` / `self_scale` (both 0.5).

    Reproduced as published: the mix `query (1,n,C) * scale + self_feat (n,1,C) * self_scale` broadcasts to
    (n,n,C) and row 0 is kept (:986-990, :1086-1089), i.e. EVERY query of a group receives the self-branch feature
    of the group's FIRST query (lowest query index)."""

    def _v2_init(self, hidden, dropout, sfx):
        from .transformer_bricks import TransFFN
        for n in ('linear1', 'linear2', 'dropout', 'activation'):
            delattr(self, n + sfx)
        setattr(self, 'self_norm' + sfx, nn.LayerNorm(hidden))
        self.ffn = TransFFN(embed_dims=hidden, feedforward_channels=hidden * 4, num_fcs=2, ffn_drop=dropout,
                            act_cfg=dict(type='ReLU', inplace=True))
        self.self_ffn = TransFFN(embed_dims=hidden, feedforward_channels=hidden * 4, num_fcs=2, ffn_drop=dropout,
                                 act_cfg=dict(type='ReLU', inplace=True))
        self.scale = nn.Parameter(torch.ones(1) * 0.5)
        self.self_scale = nn.Parameter(torch.ones(1) * 0.5)

    def _attend(self, q_tok, x, sfx, key_allowed):
        g = lambda n: getattr(self, n + sfx)
        sa = g('dyconv_pre_self_attn')
        p = sa.dropout if self.training else 0.0
        a = mha_tokens(q_tok, x, x, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj, sa.num_heads, p, key_allowed)
        return _post_norm(g('norm1'), q_tok, g('dropout1')(a))

    def _main_branch(self, y, roi, sfx):
        g = lambda n: getattr(self, n + sfx)
        shp = y.shape
        dy = g('dyconv').forward_nk(y.reshape(-1, shp[-1]), roi)
        z = _post_norm(g('norm2'), y, g('dropout2')(dy.view(shp)))
        return self.ffn.then_norm(g('norm3'), z)


class ImageRCNNBlockV2(_V2Mix, ImageRCNNBlock):
    """Reference decoder_utils.py:844-993."""

    def __init__(self, num_views, num_proposals, out_size_factor_img, test_cfg, bbox_coder, hidden_channel,
                 num_heads, dropout):
        ImageRCNNBlock.__init__(self, num_views, num_proposals, out_size_factor_img, test_cfg, bbox_coder,
                                hidden_channel, num_heads, dropout)
        self._v2_init(hidden_channel, dropout, '')

    def _refine_views(self, x, roi, sel, last, lastc):
        B, Q, C = x.shape
        V = sel.shape[1]
        key_allowed = sel.gather(1, lastc.view(B, Q, 1).expand(B, Q, Q)) | (last < 0).unsqueeze(-1)
        y = self._attend(x, x, '', key_allowed)                                  # x is also the key / value set
        z = self._main_branch(y, roi, '')
        # self branch of the first query of every view, evaluated among that view's queries
        ar = torch.arange(Q, device=x.device).view(1, 1, Q)
        first = torch.where(sel, ar, torch.full_like(ar, Q)).min(-1).values      # (B,V); Q when the view is unused
        firstc = first.clamp(max=Q - 1)
        xf = x.gather(1, firstc.unsqueeze(-1).expand(B, V, C))
        yf = self._attend(xf, x, '', sel | (first >= Q).unsqueeze(-1))           # (B,V,C)
        sf = self.self_ffn.then_norm(self.self_norm, yf)
        s = sf.gather(1, lastc.unsqueeze(-1).expand(B, Q, C))                    # self feature of view v*(q)
        return z * self.scale + s * self.self_scale


class PointRCNNBlockV2(_V2Mix, PointRCNNBlock):
    """Reference decoder_utils.py:997-1089.  The published `forward` has no `return` (:1089); `(query_feat, None)`
    - the v1 contract the ++ head unpacks (deepinteractionplusplus_decoder.py:286) - is returned here."""

    def __init__(self, hidden_channel, num_heads, dropout, bbox_coder):
        PointRCNNBlock.__init__(self, hidden_channel, num_heads, dropout, bbox_coder)
        self._v2_init(hidden_channel, dropout, '_pts')

    def _refine_all(self, x, roi):
        y = self._attend(x, x, '_pts', None)
        z = self._main_branch(y, roi, '_pts')
        s0 = self.self_ffn.then_norm(self.self_norm_pts, y[:, 0:1].contiguous())   # query 0's self feature, for all
        return z * self.scale + s0 * self.self_scale
