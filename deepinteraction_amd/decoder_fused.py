"""Fused fp16 inference form of `DeepInteractionDecoder.forward` (reference
`dense_heads/deepinteraction_decoder.py:201-313`, `models/utils/decoder_utils.py:35-841`).

Same arithmetic as the module-by-module path in `mmdet3d_plugin/models/dense_heads/deepinteraction_decoder.py`, but
every stage on the B*Q query tokens is ONE HIP kernel (csrc/token.hip): a linear layer with everything around it
(positional-embedding add, bias, ReLU / GELU, residual add, LayerNorms), the self attention among the queries, the
DynamicConv core, the RoI bookkeeping, the six prediction heads with centre offset / on-the-image merge written
straight into the concatenated output tensors.  ~100 launches instead of ~370, no float32<->float16 casts, no
`torch.cat` / `where` / `gather` glue; LayerNorm and residual adds are evaluated on the float32 accumulators.

Weights are read from the live module (a reference checkpoint loads into it unchanged); derived forms (float32 biases,
folded BatchNorm, the permuted DynamicConv generator) are cached per module and rebuilt when a parameter changes.
"""
import torch
import torch.nn.functional as F

from . import ops
from . import utils
from .utils import fork_join, param_key


def usable(dec, lidar_feat, img_feat):
    """The fused path covers the fp16 inference form of the reference configs (v1 RoI blocks, 128 channels, 8 heads)."""
    from .mmdet3d_plugin.models.utils.decoder_utils import ImageRCNNBlock, PointRCNNBlock
    if torch.is_grad_enabled() or dec.training or not lidar_feat.is_cuda or lidar_feat.dtype != torch.float16:
        return False
    if lidar_feat.shape[1] != 128 or dec.num_heads != 8 or dec.num_decoder_layers != 1 or not dec.auxiliary:
        return False
    if dec.num_proposals > 512 or dec.class_encoding.weight.dtype != torch.float16:
        return False
    for l, blk in enumerate(dec.decode_head):
        if type(blk) is not (ImageRCNNBlock if l % 2 == 0 else PointRCNNBlock):
            return False
    layer = dec.decoder[0]
    if layer.cross_only or layer.self_posembed is None or layer.cross_posembed is None:
        return False
    if layer.activation is not F.relu or layer.linear1.out_features % 128:
        return False
    return all(ffn._fusable() for ffn in list(dec.prediction_heads) + list(dec.pred_head))


class _Cache:
    """Derived constant tensors of one module, rebuilt when any of its parameters / buffers changes."""

    def __init__(self):
        self.key, self.val = None, None

    def get(self, module, build):
        key = param_key(module)
        if self.key != key:
            with torch.no_grad():
                self.val = build()
            self.key = key
        return self.val


def _f32(t):
    return t.detach().to(torch.float32, copy=True).contiguous()


def _dyn_perm(device):
    """Row permutation of DynamicConv.dynamic_layer that makes the generated parameters arrive as
    [p1^T (d, c) | p2^T (e, k-permuted d)] (csrc/token.hip dynconv_kernel)."""
    d = torch.arange(128).view(128, 1)
    c = torch.arange(128).view(1, 128)
    first = (c * 128 + d).reshape(-1)                                  # mine[d*128 + c] = ref[c*128 + d]
    j = torch.arange(128)
    kk, g, t, r = j // 32, (j % 32) // 8, (j % 8) // 4, j % 4
    dj = 32 * kk + 16 * t + 4 * g + r                                  # k slot j of the MFMA <-> channel d
    e = torch.arange(128).view(128, 1)
    second = (16384 + dj.view(1, 128) * 128 + e).reshape(-1)           # mine[16384 + e*128 + j] = ref[16384 + d(j)*128 + e]
    return torch.cat([first, second]).to(device)


def _mha_consts(mha):
    """(in_proj_weight fp16 (3E,E), in_proj_bias f32, out_proj.weight, out_proj.bias f32) of either attention class."""
    return (mha.in_proj_weight.detach(), _f32(mha.in_proj_bias), mha.out_proj.weight.detach(), _f32(mha.out_proj.bias))


def _ln(norm):
    return (norm.weight.detach(), norm.bias.detach())


def _heads_consts(ffn, dtype):
    W1, b1, W2, b2, sizes = ffn.folded()                               # float32, BatchNorm folded, heads stacked
    rows, col = [], 0
    for h, n in enumerate(sizes):
        rows.append(W2[col:col + n, h * 64:(h + 1) * 64])
        col += n
    return (W1.to(dtype).contiguous(), b1.contiguous(), torch.cat(rows).contiguous(), b2.contiguous(), list(sizes))


class FusedDecoder:
    """Holds only derived constants; the module is passed to `forward` (so that copies of the head stay independent)."""

    def __init__(self):
        self.caches = {}

    def _c(self, name, module, build):
        return self.caches.setdefault(name, _Cache()).get(module, build)

    # ------------------------------------------------------------------ pieces
    def _block(self, blk, sfx, x, roi, B, Q, member=None, view=None, keep=None):
        """decoder_utils.py:743-756 / :824-837 on tokens x (B*Q,128)."""
        g = lambda n: getattr(blk, n + sfx)
        sa, dy = g('dyconv_pre_self_attn'), g('dyconv')
        wq, bq, wo, bo = self._c(('sa', id(blk)), sa, lambda: _mha_consts(sa))
        qkv = ops.token_linear(x, wq, bq)
        att = ops.token_mha(qkv, B, Q, sa.num_heads, float(sa.head_dim) ** -0.5, member, view)
        y = ops.token_linear(att, wo, bo, res1=x, ln1=_ln(g('norm1')), eps=g('norm1').eps)

        def dyn():
            perm = _dyn_perm(dy.dynamic_layer.weight.device)
            return (dy.dynamic_layer.weight.detach()[perm].contiguous(), _f32(dy.dynamic_layer.bias)[perm].contiguous(),
                    _f32(dy.out_layer.bias))
        wd, bd, bout = self._c(('dy', id(blk)), dy, dyn)
        params = ops.token_linear(y, wd, bd)                                        # (B*Q, 2*128*128), fused layout
        f2 = ops.dynconv(roi, params, _ln(dy.norm1), _ln(dy.norm2), dy.norm1.eps)
        z = ops.token_linear(f2.view(B * Q, -1), dy.out_layer.weight.detach(), bout, ln1=_ln(dy.norm3), act2=True,
                             res2=y, ln2=_ln(g('norm2')), eps=g('norm2').eps)
        l1, l2 = g('linear1'), g('linear2')
        b1, b2 = self._c(('ffn', id(blk)), [l1.bias, l2.bias], lambda: (_f32(l1.bias), _f32(l2.bias)))
        h = ops.token_linear(z, l1.weight.detach(), b1, act1=2)
        return ops.token_linear(h, l2.weight.detach(), b2, res1=z, ln1=_ln(g('norm3')), keep=keep, eps=g('norm3').eps)

    def _cross_kv(self, layer, lidar_flat_tokens, bev_pos):
        """K and V of the cross attention (decoder_utils.py:98-105): they depend on the BEV map only, not on the queries."""
        ca = layer.multihead_attn
        cw = self._c('dl_ca', ca, lambda: _mha_consts(ca))[0]
        E = ca.embed_dim
        kpe = layer.key_pos_embed(bev_pos, lidar_flat_tokens.dtype)
        return F.linear(lidar_flat_tokens + kpe, cw[E:], ca.in_proj_bias[E:])       # (B,HW,2C) = [K | V]

    def _decoder_layer(self, layer, x, qpe, kv, B, Q):
        """decoder_utils.py:83-113 (post-norm; positional embeddings added to q, k and v)."""
        sa, ca = layer.self_attn, layer.multihead_attn
        wq, bq, wo, bo = self._c('dl_sa', sa, lambda: _mha_consts(sa))
        qkv = ops.token_linear(x, wq, bq, pos=qpe)
        att = ops.token_mha(qkv, B, Q, sa.num_heads, float(sa.head_dim) ** -0.5)
        x = ops.token_linear(att, wo, bo, res1=x, ln1=_ln(layer.norm1), eps=layer.norm1.eps)
        cw, cb, cwo, cbo = self._c('dl_ca', ca, lambda: _mha_consts(ca))
        E = ca.embed_dim
        q = ops.token_linear(x, cw[:E], cb[:E], pos=qpe)
        o = ops.mha_decode(q.view(B, Q, E), kv, ca.num_heads, float(ca.head_dim) ** -0.5).view(B * Q, E)
        x = ops.token_linear(o, cwo, cbo, res1=x, ln1=_ln(layer.norm2), eps=layer.norm2.eps)
        b1, b2 = self._c('dl_ffn', [layer.linear1.bias, layer.linear2.bias],
                         lambda: (_f32(layer.linear1.bias), _f32(layer.linear2.bias)))
        h = ops.token_linear(x, layer.linear1.weight.detach(), b1, act1=1)
        return ops.token_linear(h, layer.linear2.weight.detach(), b2, res1=x, ln1=_ln(layer.norm3), eps=layer.norm3.eps)

    # ------------------------------------------------------------------ forward
    def forward(self, dec, pts_inputs, img_inputs, img_metas):
        from .mmdet3d_plugin.models.utils.decoder_utils import QueryGeometry
        lidar_feat, new_lidar_feat = ops.cl(pts_inputs[0]), ops.cl(pts_inputs[1])
        B, C, H, W = lidar_feat.shape
        HW, Q, ncls = H * W, dec.num_proposals, dec.num_classes
        dev = lidar_feat.device
        img_feat = ops.cl(img_inputs)
        BN, I_C, I_H, I_W = img_feat.shape
        V = dec.num_views

        tokens = lidar_feat.permute(0, 2, 3, 1).reshape(B, HW, C)                    # view of the channels-last map

        def first_map():        # everything that needs the shared-conv BEV map only
            return dec._heatmap(dec.heatmap_head, lidar_feat), self._cross_kv(dec.decoder[0], tokens, dec._bev_pos(dev))

        if utils.OVERLAP & 8:
            dense_heatmap_img, (dense_heatmap, kv) = fork_join(
                dev, lambda: dec._heatmap(dec.heatmap_head_img, new_lidar_feat), first_map)
        else:
            dense_heatmap, kv = first_map()
            dense_heatmap_img = dec._heatmap(dec.heatmap_head_img, new_lidar_feat)
        k1 = {'nuScenes': (8, 9), 'Waymo': (1, 2)}.get(dec.test_cfg['dataset'], ())
        heatmap = ops.heatmap_nms(dense_heatmap, dense_heatmap_img, dec.nms_kernel_size,
                                  [c for c in k1 if c < ncls]).view(B, ncls, HW)
        top = ops.topk(heatmap.view(B, -1), Q)                                       # ties: lower index first
        pe_consts = self._c('self_pe', dec.decoder[0].self_posembed, lambda: self._pe_consts(dec.decoder[0].self_posembed))
        ce = dec.class_encoding
        ce_w, ce_b = self._c('ce', ce, lambda: (ce.weight.detach()[:, :, 0].contiguous(), ce.bias.detach().contiguous()))
        x, qpe, pos, labels = ops.query_init(lidar_feat, top, ce_w, ce_b, pe_consts)
        dec.query_labels, dec.top_proposals = labels, top

        x = self._decoder_layer(dec.decoder[0], x, qpe, kv, B, Q)

        heads0 = self._c('ph0', dec.prediction_heads[0], lambda: _heads_consts(dec.prediction_heads[0], x.dtype))
        names, cls = list(dec.prediction_heads[0].heads), heads0[4]
        ic = names.index('center')
        L = dec.num_mmpi
        first = [torch.empty((B, n, Q), dtype=torch.float32, device=dev) for n in cls]
        final = [torch.empty((B, n, L * Q), dtype=torch.float32, device=dev) for n in cls]
        pos1 = torch.empty((B, Q, 2), dtype=torch.float32, device=dev)
        ops.pred_heads(x, None, heads0, pos, first, B, Q, Q, 0, ic, pos_out=pos1)
        pos = pos1

        geom = dec.static_geometry if dec.static_geometry is not None else QueryGeometry(img_metas, dev)
        maps = img_feat.view(B * V, C, I_H, I_W)
        cfg, bc = dec.test_cfg, dec.bbox_coder
        cell_bev = bc.out_size_factor * bc.voxel_size[0]
        dec.on_the_image_mask = []
        res, ld, col = dict(zip(names, first)), Q, 0                                 # where the previous stage's boxes are
        for l in range(L):
            blk = dec.decode_head[l]
            r32 = {k: res[k][:, :, col:col + Q] for k in ('center', 'height', 'dim', 'rot')}
            prev = x
            if l % 2 == 0:
                on, rect, _ = ops.query_geometry(r32, geom.proj, geom.aug_rev, geom.per_sample,
                                                 cfg['out_size_factor'] * cfg['voxel_size'][0], cfg['pc_range'][:2],
                                                 1.0, 1.0, True, False, ld=ld)
                rois, view, member, keep, on_img = ops.roi_select(rect, on)
                roi = ops.roi_align(maps, rois, 1.0 / blk.out_size_factor_img)       # (B*Q,49,C)
                x = self._block(blk, '', prev, roi, B, Q, member, view, keep)
                dec.on_the_image_mask.append(keep.view(B, Q).bool())
            else:
                _, _, rect = ops.query_geometry(r32, None, None, None, cell_bev, bc.pc_range[:2], cell_bev, 2.0,
                                                False, True, ld=ld)
                roi = ops.roi_align(new_lidar_feat, ops.roi_select(rect), 1.0)
                x = self._block(blk, '_pts', prev, roi, B, Q)
                keep = None
            hc = self._c(('ph', l), dec.pred_head[l], lambda: _heads_consts(dec.pred_head[l], x.dtype))
            pos_next = torch.empty((B, Q, 2), dtype=torch.float32, device=dev)
            ops.pred_heads(x, prev, hc, pos, final, B, Q, L * Q, l * Q, ic, keep=keep,
                           first=first if keep is not None else None, pos_out=pos_next)
            pos = pos_next
            res, ld, col = dict(zip(names, final)), L * Q, l * Q

        out = dict(zip(names, final))
        top_index = top % HW
        out['query_heatmap_score'] = heatmap.gather(index=top_index[:, None, :].expand(-1, ncls, -1), dim=-1)
        out['dense_heatmap'] = dense_heatmap_img
        return [[out]]

    @staticmethod
    def _pe_consts(pe):
        c1, bn, _, c2 = pe.position_embedding_head
        g = torch.rsqrt(bn.running_var.float() + bn.eps) * bn.weight.float()
        w1 = (c1.weight.float()[:, :, 0] * g[:, None]).contiguous()
        b1 = ((c1.bias.float() - bn.running_mean.float()) * g + bn.bias.float()).contiguous()
        return (w1, b1, _f32(c2.weight[:, :, 0]), _f32(c2.bias))
