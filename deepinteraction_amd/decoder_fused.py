"""Fused inference form of `DeepInteractionDecoder.forward` on fp16 feature maps (reference
`dense_heads/deepinteraction_decoder.py:201-313`, `models/utils/decoder_utils.py:35-841`).

Same arithmetic as the module-by-module path in `mmdet3d_plugin/models/dense_heads/deepinteraction_decoder.py`, with
the B*Q query tokens handled by the float32 token kernels of csrc/token32.hip: everything that is parallel over
tokens runs as ONE `TokenProgram` launch per dependency level (self attention among the queries + output projection +
residual LayerNorm + the next projection; merge of the cross attention + the FFN + the prediction heads + the next
block's packed projection; ...), the rest is the DynamicConv parameter generator / core / split-K out_layer, the RoI
bookkeeping and the gathers from the maps.  ~50 launches per forward.

Precision (round 3, tests/tools/fp16_error_budget.py): the feature MAPS stay fp16; the token state, the RoI features,
the generated DynamicConv parameters, the cross attention's K / q and EVERY weight of this path are float32 - rounding
any of them to fp16 alone moves 3-30 % of the box outputs by more than 1e-3 (the cross attention's logits reach
|s| ~ 500).  Weights are read from the live module (a reference checkpoint loads into it unchanged): float32 parameters
are used as they are (`deepinteraction_amd.precision.half_maps_` keeps them so), fp16 parameters are widened - their
rounding is then part of the result; derived forms (folded BatchNorm, the permuted DynamicConv generator, the split
K/V weight, the positional K/V bias) are cached per module and rebuilt when a parameter changes.
"""
import torch
import torch.nn.functional as F

from . import ops
from . import utils
from .utils import fork_join, param_key


def usable(dec, lidar_feat, img_feat):
    """The fused path covers the fp16-map inference form of the reference configs (v1 RoI blocks, 128 channels, 8 heads)."""
    from .mmdet3d_plugin.models.utils.decoder_utils import ImageRCNNBlock, ImageRCNNBlockV2, PointRCNNBlock, PointRCNNBlockV2
    if torch.is_grad_enabled() or dec.training or not lidar_feat.is_cuda or lidar_feat.dtype != torch.float16:
        return False
    pp = is_plusplus(dec)
    if lidar_feat.shape[1] != 128 or dec.num_heads != 8 or dec.num_decoder_layers != 1 or not dec.auxiliary:
        return False
    if dec.num_proposals > 512 or dec.num_classes * lidar_feat.shape[2] * lidar_feat.shape[3] > (1 << 20):
        return False
    for l, blk in enumerate(dec.decode_head):
        want = ((ImageRCNNBlockV2, PointRCNNBlockV2) if pp else (ImageRCNNBlock, PointRCNNBlock))[l % 2]
        if type(blk) is not want:
            return False
        sfx = '' if l % 2 == 0 else '_pts'
        hidden = blk.ffn.feedforward_channels if pp else getattr(blk, 'linear1' + sfx).out_features
        if hidden not in (128, 256, 384, 512):
            return False
        if pp and not (blk.ffn.num_fcs == 2 and blk.self_ffn.num_fcs == 2 and blk.ffn.add_identity and blk.self_ffn.add_identity):
            return False
        # the one-launch self branch (csrc/v2_self.hip: hidden <= 1024 in multiples of 32, <= 8 views, <= 512 queries - the
        # proposal bound above): anything wider takes the module path (torch) instead of failing in the kernel's argument check
        if pp and not (blk.self_ffn.feedforward_channels <= 1024 and blk.self_ffn.feedforward_channels % 32 == 0):
            return False
    if pp and img_feat is not None and lidar_feat.shape[0] > 0 and img_feat.shape[0] // lidar_feat.shape[0] > 8:
        return False
    layer = dec.decoder[0]
    if layer.cross_only or layer.self_posembed is None or layer.cross_posembed is None:
        return False
    if layer.activation is not F.relu or layer.linear1.out_features not in (128, 256, 384, 512):
        return False
    for head in (dec.heatmap_head, dec.heatmap_head_img):
        cm, last = head[0], head[1]
        if (cm.conv.kernel_size != (3, 3) or cm.conv.padding != (1, 1) or cm.conv.stride != (1, 1)
                or last.kernel_size != (3, 3) or last.padding != (1, 1) or last.stride != (1, 1)
                or not isinstance(cm.activate, torch.nn.ReLU) or cm.conv.out_channels != 128 or last.out_channels > 16):
            return False
    return all(ffn._fusable() and len(ffn.heads) * 64 <= 512 for ffn in list(dec.prediction_heads) + list(dec.pred_head))


def is_plusplus(dec):
    """DeepInteraction++ head (V2 RoI blocks, 128-channel prediction heads, look-forward centres, cumulative image mask)."""
    return type(dec).__name__ == 'DeepInteractionPlusPlusDecoder'


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
    return t.detach().float().contiguous()


def _dyn_value_order(device):
    """ref[V]: the index into DynamicConv.dynamic_layer's 32 768 outputs of value V of the fragment order
    (csrc/token32.hip): V = (((p*8 + nb)*4 + kk)*64 + 16g + i)*8 + j;  p = 0: p1[c = 32kk + 8g + j][d = 16nb + i] (index
    c*128 + d), p = 1: p2[d = 32kk + 8g + j][e = 16nb + i] (index 16384 + d*128 + e)."""
    V = torch.arange(32768)
    j, lane, kk, nb, pr = V % 8, (V // 8) % 64, (V // 512) % 4, (V // 2048) % 8, V // 16384
    i, g = lane % 16, lane // 16
    k = 32 * kk + 8 * g + j
    n = 16 * nb + i
    return (pr * 16384 + k * 128 + n).to(device)


def _dyn_layout(weight, bias):
    """(w_packed, bias_v) of `ops.token_wide` from DynamicConv.dynamic_layer: rows in value order, then the MFMA row
    mapping of the generator kernel (a wave's 64 values V0..: row i' of tile nb <-> V0 + 16 (i' >> 2) + 4 nb + (i' & 3)),
    then `ops.pack_linear`."""
    dev = weight.device
    ref = _dyn_value_order(dev)
    wv, bv = weight.detach().float()[ref], bias.detach().float()[ref].contiguous()
    r = torch.arange(32768, device=dev)
    wq, nb, ip = r // 64, (r % 64) // 16, r % 16
    rows = 64 * wq + 16 * (ip // 4) + 4 * nb + (ip % 4)
    return ops.pack_linear(wv[rows]), bv


def _split(w):
    """float32 weight -> the (hi, lo) fp16 pair of the three-pass MFMA product."""
    return ops.split_hi_lo(w)


def _pack(w):
    """float32 weight of a program's linear step, packed in MFMA fragment order (hi / lo split)."""
    return ops.pack_linear(w.detach().float())


def _mha_consts(mha):
    """(in_proj_weight (3E,E) packed, in_proj_bias, out_proj.weight packed, out_proj.bias)."""
    return (_pack(mha.in_proj_weight), _f32(mha.in_proj_bias), _pack(mha.out_proj.weight), _f32(mha.out_proj.bias))


def _ln(norm):
    return (_f32(norm.weight), _f32(norm.bias))


def _heads_consts(ffn):
    W1, b1, W2, b2, sizes = ffn.folded()                               # float32, BatchNorm folded, heads stacked
    rows, col = [], 0
    for h, n in enumerate(sizes):
        rows.append(W2[col:col + n, h * 64:(h + 1) * 64])
        col += n
    return (_pack(W1), b1.contiguous(), torch.cat(rows).contiguous(), b2.contiguous(), list(sizes))


class FusedDecoder:
    """Holds only derived constants; the module is passed to `forward` (so that copies of the head stay independent)."""

    def __init__(self):
        self.caches = {}

    def _c(self, name, module, build):
        return self.caches.setdefault(name, _Cache()).get(module, build)

    # ------------------------------------------------------------------ constants
    def _block_consts(self, blk, sfx):
        g = lambda n: getattr(blk, n + sfx)

        def build():
            sa, dy = g('dyconv_pre_self_attn'), g('dyconv')
            wd, bd = _dyn_layout(dy.dynamic_layer.weight, dy.dynamic_layer.bias)
            return dict(sa=_mha_consts(sa), scale=float(sa.head_dim) ** -0.5,
                        n1=_ln(g('norm1')), n2=_ln(g('norm2')), n3=_ln(g('norm3')),
                        eps=(g('norm1').eps, g('norm2').eps, g('norm3').eps),
                        wd=wd, bd=bd,
                        dn1=_ln(dy.norm1), dn2=_ln(dy.norm2), dn3=_ln(dy.norm3), deps=(dy.norm1.eps, dy.norm3.eps),
                        wout=ops.pack_ksteps(dy.out_layer.weight.detach().float()), bout=_f32(dy.out_layer.bias),
                        w1=_pack(g('linear1').weight), b1=_f32(g('linear1').bias),
                        w2=_pack(g('linear2').weight), b2=_f32(g('linear2').bias), act=2)
        return self._c(('blk', id(blk)), blk, build)

    def _block_consts_v2(self, blk, sfx):
        """`ImageRCNNBlockV2` / `PointRCNNBlockV2` (decoder_utils.py:844-1089): the v1 stack with an mmcv FFN (ReLU) in the
        main branch, whose closing LayerNorm carries the learnable `scale` (the mix is `main * scale + self * self_scale`),
        and the constants of the self branch (FFN + LayerNorm on ONE token per query group; evaluated with torch in float32)."""
        g = lambda n: getattr(blk, n + sfx)

        def build():
            sa, dy = g('dyconv_pre_self_attn'), g('dyconv')
            wd, bd = _dyn_layout(dy.dynamic_layer.weight, dy.dynamic_layer.bias)
            l1, l2 = blk.ffn.layers[0][0], blk.ffn.layers[1]
            s1, s2 = blk.self_ffn.layers[0][0], blk.self_ffn.layers[1]
            sc, ssc = blk.scale.detach().float(), blk.self_scale.detach().float()
            E = sa.embed_dim
            return dict(sa=_mha_consts(sa), scale=float(sa.head_dim) ** -0.5,
                        n1=_ln(g('norm1')), n2=_ln(g('norm2')),
                        n3=((_f32(g('norm3').weight) * sc).contiguous(), (_f32(g('norm3').bias) * sc).contiguous()),
                        eps=(g('norm1').eps, g('norm2').eps, g('norm3').eps),
                        wd=wd, bd=bd,
                        dn1=_ln(dy.norm1), dn2=_ln(dy.norm2), dn3=_ln(dy.norm3), deps=(dy.norm1.eps, dy.norm3.eps),
                        wout=ops.pack_ksteps(dy.out_layer.weight.detach().float()), bout=_f32(dy.out_layer.bias),
                        w1=_pack(l1.weight), b1=_f32(l1.bias), w2=_pack(l2.weight), b2=_f32(l2.bias), act=1,
                        # self branch (float32 torch): attention output projection + norm1, self FFN + its LayerNorm
                        wo=_f32(sa.out_proj.weight), bo=_f32(sa.out_proj.bias), heads=sa.num_heads,
                        sw1=_f32(s1.weight), sb1=_f32(s1.bias), sw2=_f32(s2.weight), sb2=_f32(s2.bias),
                        sn=_ln(g('self_norm')), sn_eps=g('self_norm').eps, self_scale=ssc)
        return self._c(('blk2', id(blk)), blk, build)

    @staticmethod
    def _self_feature_img(c, x, qkv, view, member, B, Q, V):
        """decoder_utils.py:970-990 as the product restates it (`ImageRCNNBlockV2._refine_views`): per valid view the FIRST
        query of the view attends to the view's queries (q / k / v rows are the ones the main attention uses), goes
        through norm1 and the self FFN + LayerNorm; query q then receives the feature of ITS view v*(q) (the published
        broadcast keeps row 0 of every group).  (B*Q, 128) float32, already times `self_scale`.  One launch
        (csrc/v2_self.hip; rounds 3-4: ~35 float32 torch launches on <= 6 tokens)."""
        return ops.v2_self_feature(c, x, B, Q, qkv=qkv, view=view, member=member, V=V)

    @staticmethod
    def _self_feature_pts(c, y, B, Q):
        """decoder_utils.py:1086-1089 (`PointRCNNBlockV2._refine_all`): the self feature of query 0, for every query."""
        return ops.v2_self_feature(c, y, B, Q)

    def _layer_consts(self, layer):
        def build():
            sa, ca = layer.self_attn, layer.multihead_attn
            return dict(sa=_mha_consts(sa), sa_scale=float(sa.head_dim) ** -0.5, ca=_mha_consts(ca),
                        ca_scale=float(ca.head_dim) ** -0.5,
                        n1=_ln(layer.norm1), n2=_ln(layer.norm2), n3=_ln(layer.norm3),
                        eps=(layer.norm1.eps, layer.norm2.eps, layer.norm3.eps),
                        w1=_pack(layer.linear1.weight), b1=_f32(layer.linear1.bias),
                        w2=_pack(layer.linear2.weight), b2=_f32(layer.linear2.bias),
                        wq=_pack(_f32(ca.in_proj_weight)[:ca.embed_dim]), bq=_f32(ca.in_proj_bias)[:ca.embed_dim].contiguous())
        return self._c('layer', layer, build)

    def _kv_consts(self, layer, bev_pos):
        """Split K/V weight and the positional K/V bias of the cross attention (decoder_utils.py:98-100:
        `key + key_pos` goes through the K and V projections; the BEV grid is constant)."""
        def build():
            ca = layer.multihead_attn
            E = ca.embed_dim
            w, b = _f32(ca.in_proj_weight)[E:], _f32(ca.in_proj_bias)[E:]
            kpe = layer.cross_posembed.tokens(bev_pos[:1], torch.float32)[0]            # (S,E) float32
            bias = torch.addmm(b, kpe, w.t())                                            # (S,2E)
            return ops.pack_kv_weight(w), bias[:, :E].contiguous(), bias[:, E:].contiguous()
        return self._c('kv', [p for p in layer.multihead_attn.parameters()] +
                       [t for t in list(layer.cross_posembed.parameters()) + list(layer.cross_posembed.buffers())] + [bev_pos],
                       build)

    @staticmethod
    def _pe_consts(pe):
        c1, bn, _, c2 = pe.position_embedding_head
        g = torch.rsqrt(bn.running_var.float() + bn.eps) * bn.weight.float()
        w1 = (c1.weight.float()[:, :, 0] * g[:, None]).contiguous()
        b1 = ((c1.bias.float() - bn.running_mean.float()) * g + bn.bias.float()).contiguous()
        return (w1, b1, _f32(c2.weight[:, :, 0]), _f32(c2.bias))

    # ------------------------------------------------------------------ pieces
    @staticmethod
    def _heads_and_next(p, x_buf, hid_buf, qkv_buf, heads, nxt, qkv_out):
        """Tail of a stage on the rows in buf[x_buf] (`K` = 128, or 256 with the previous tokens at columns 128..255):
        roles 0..nheads-1 evaluate one prediction head each (hidden 64 channels), roles nheads..nheads+2 one 128-column
        third of the NEXT block's packed self-attention projection."""
        nh = len(heads['cls'])
        hw1, hb1 = heads.pop('w1'), heads.pop('b1')
        p.linear(x_buf, hid_buf, hw1, hb1, act=1, roles=(0, nh - 1), n_per_role=64)
        p.heads(hid_buf, roles=(0, nh - 1), per_role=True, **heads)
        if nxt is not None:                                  # roles nh, nh + 1: q and k rows; role nh + 2: the values, transposed
            p.linear(x_buf, qkv_buf, nxt[0], nxt[1], roles=(nh, nh + 2), n_per_role=128)
            p.store(qkv_buf, qkv_out[0], roles=(nh, nh + 1), n=128, role_offset=128)
            p.store_t(qkv_buf, qkv_out[1], roles=(nh + 2, nh + 2))

    def _consts_of(self, blk, sfx):
        return self._block_consts_v2(blk, sfx) if hasattr(blk, 'self_ffn') else self._block_consts(blk, sfx)

    def _attend(self, blk, sfx, x, qkv, B, Q, member=None, view=None):
        """decoder_utils.py:743-746 / :824-826: y = norm1(x + self_attention(x)) from the packed projection `qkv` = (rows
        [q | k], V^T) the previous program wrote."""
        c = self._consts_of(blk, sfx)
        y = torch.empty((B * Q, 128), dtype=torch.float32, device=x.device)
        y_hl = torch.empty((B * Q, 256), dtype=torch.float16, device=x.device)      # the same rows split: the generator's operand
        p = ops.TokenProgram()
        p.attn(0, qkv[0], qkv[1], c['scale'], member, view)
        p.linear(0, 1, c['sa'][2], c['sa'][3])
        p.load(2, x)
        p.rowop(1, 1, aux=2, ln=c['n1'], eps=c['eps'][0])
        p.store(1, y).store_hl(1, y_hl)
        p.run(B, Q)
        return y, y_hl

    def _refine(self, blk, sfx, x, y, roi, B, Q, heads, next_qkv_w, keep=None, self_feat=None):
        """decoder_utils.py:747-756 / :827-837 on y (B*Q,128) and the RoI features, then this stage's prediction heads
        (`heads` = the arguments of TokenProgram.heads, on [x' ; x]) and the next block's packed projection.  Returns
        (x', qkv').  Two programs around the DynamicConv kernels; the weights of a token group are spread over the roles
        (<= 128 KB per workgroup)."""
        c = self._consts_of(blk, sfx)                                               # self_feat: the V2 blocks' self branch
        M, dev = B * Q, x.device
        f32e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        y, y_hl = y
        params = ops.token_wide(y_hl, c['wd'], c['bd'])                              # (M, 65536) hi / lo fragments
        f2p = ops.dynconv(roi, params, c['dn1'], c['dn2'], c['deps'][0])
        ws, ns = ops.token_splitk(f2p, c['wout'])
        # FFN with the hidden dimension split over the roles: role r holds hidden channels [128r, 128r + 128)
        nh = c['w1'].shape[0] * 16 // 128
        z, ws2 = f32e(M, 128), f32e(nh, M, 128)
        p = ops.TokenProgram(roles=nh)
        p.load_parts(0, ws, ns, M, c['bout'])
        p.rowop(0, 0, ln=c['dn3'], eps=c['deps'][1], relu=True)
        p.load(1, y)
        p.rowop(0, 0, aux=1, ln=c['n2'], eps=c['eps'][1])                            # z
        p.store(0, z, roles=(0, 0))
        p.linear(0, 1, c['w1'], c['b1'], act=c['act'], n_per_role=128)          # GELU (v1) / ReLU (the mmcv FFN of V2)
        p.linear(1, 2, c['w2'], None, k_per_role=128)
        p.store(2, ws2, n=128, role_offset=M * 128)
        p.run(B, Q)
        nhd = len(heads['cls'])
        xn = f32e(M, 128)
        qkv_n = (f32e(M, 256), f32e(B, 128, (Q + 15) // 16 * 16)) if next_qkv_w is not None else None
        p = ops.TokenProgram(roles=nhd + (3 if next_qkv_w is not None else 0))
        p.load_parts(2, ws2, nh, M, c['b2'])
        p.load(0, z)
        if self_feat is None:
            p.rowop(2, 2, aux=0, ln=c['n3'], eps=c['eps'][2], keep=keep)             # x'
            p.store(2, xn, roles=(0, 0))
            p.load(2, x, col=128)                                                    # [x' ; x]
        else:       # V2: x' = keep * (norm3(.) * scale + self * self_scale); the 128-channel heads see x' alone
            p.rowop(2, 2, aux=0, ln=c['n3'], eps=c['eps'][2])                        # (norm3's affine carries `scale`)
            p.load(0, self_feat)
            p.rowop(2, 2, aux=0, keep=keep)
            p.store(2, xn, roles=(0, 0))
        self._heads_and_next(p, 2, 0, 1, heads, next_qkv_w, qkv_n)
        p.run(B, Q)
        return xn, qkv_n

    # ------------------------------------------------------------------ forward
    def forward(self, dec, pts_inputs, img_inputs, img_metas):
        from .mmdet3d_plugin.models.utils.decoder_utils import QueryGeometry
        lidar_feat, new_lidar_feat = ops.cl(pts_inputs[0]), ops.cl(pts_inputs[1])
        B, C, H, W = lidar_feat.shape
        HW, Q, ncls = H * W, dec.num_proposals, dec.num_classes
        M = B * Q
        dev = lidar_feat.device
        img_feat = ops.cl(img_inputs)
        BN, I_C, I_H, I_W = img_feat.shape
        V = dec.num_views
        layer = dec.decoder[0]

        tokens = lidar_feat.permute(0, 2, 3, 1).reshape(B, HW, C)                    # view of the channels-last map
        kvc = self._kv_consts(layer, dec._bev_pos(dev))

        def first_map():        # everything that needs the shared-conv BEV map only
            return dec._heatmap(dec.heatmap_head, lidar_feat, out_f32=True), ops.kv_project(tokens, *kvc)

        if utils.OVERLAP & 8:
            dense_heatmap_img, (dense_heatmap, kx) = fork_join(
                dev, lambda: dec._heatmap(dec.heatmap_head_img, new_lidar_feat, out_f32=True), first_map)
        else:
            dense_heatmap, kx = first_map()
            dense_heatmap_img = dec._heatmap(dec.heatmap_head_img, new_lidar_feat, out_f32=True)
        k1 = {'nuScenes': (8, 9), 'Waymo': (1, 2)}.get(dec.test_cfg['dataset'], ())
        heatmap = ops.heatmap_nms(dense_heatmap, dense_heatmap_img, dec.nms_kernel_size,
                                  [c for c in k1 if c < ncls]).view(B, ncls, HW)
        top = ops.topk(heatmap.view(B, -1), Q)                                       # ties: lower index first
        pe_consts = self._c('self_pe', layer.self_posembed, lambda: self._pe_consts(layer.self_posembed))
        ce = dec.class_encoding
        ce_w, ce_b = self._c('ce', ce, lambda: (_f32(ce.weight[:, :, 0]), _f32(ce.bias)))
        x, qpe, pos, labels = ops.query_init(lidar_feat, top, ce_w, ce_b, pe_consts)
        dec.query_labels, dec.top_proposals = labels, top

        heads0 = self._c('ph0', dec.prediction_heads[0], lambda: _heads_consts(dec.prediction_heads[0]))
        names, cls = list(dec.prediction_heads[0].heads), heads0[4]
        ic = names.index('center')
        L = dec.num_mmpi
        first = [torch.empty((B, n, Q), dtype=torch.float32, device=dev) for n in cls]
        final = [torch.empty((B, n, L * Q), dtype=torch.float32, device=dev) for n in cls]
        pp = is_plusplus(dec)
        blk_consts = [self._consts_of(dec.decode_head[l], '' if l % 2 == 0 else '_pts') for l in range(L)]
        f32e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)

        # ---- decoder layer (decoder_utils.py:83-113: post-norm; positional embeddings added to q, k and v)
        c = self._layer_consts(layer)
        Qp = (Q + 15) // 16 * 16
        qk, vt = f32e(M, 256), f32e(B, 128, Qp)
        p = ops.TokenProgram(roles=3)                                                # a third of the packed projection each
        p.load(0, x, pos=qpe).linear(0, 1, c['sa'][0], c['sa'][1], n_per_role=128)
        p.store(1, qk, roles=(0, 1), n=128, role_offset=128).store_t(1, vt, roles=(2, 2))
        p.run(B, Q)
        x1, qc = f32e(M, 128), f32e(M, 128)
        p = ops.TokenProgram()
        p.attn(0, qk, vt, c['sa_scale'])
        p.linear(0, 1, c['sa'][2], c['sa'][3])
        p.load(2, x).load(0, qpe)
        p.rowop(1, 1, aux=2, ln=c['n1'], eps=c['eps'][0])                            # x1
        p.store(1, x1)
        p.rowop(0, 0, aux=1)                                                         # x1 + qpe
        p.linear(0, 2, c['wq'], c['bq'])
        p.store(2, qc)
        p.run(B, Q)
        o = ops.mha_decode_x(qc.view(B, Q, 128), kx, c['ca_scale'])
        x2 = f32e(M, 128)
        p = ops.TokenProgram()
        p.load(0, o).load(2, x1)
        p.linear(0, 1, c['ca'][2], c['ca'][3])
        p.rowop(1, 1, aux=2, ln=c['n2'], eps=c['eps'][1])                            # x2
        p.store(1, x2)
        p.run(B, Q)
        nh = c['w1'].shape[0] * 16 // 128                                            # FFN: hidden dimension split over the roles
        wsf = f32e(nh, M, 128)
        p = ops.TokenProgram(roles=nh)
        p.load(0, x2)
        p.linear(0, 1, c['w1'], c['b1'], act=1, n_per_role=128)
        p.linear(1, 2, c['w2'], None, k_per_role=128)
        p.store(2, wsf, n=128, role_offset=M * 128)
        p.run(B, Q)
        x3, qkv_b = f32e(M, 128), (f32e(M, 256), f32e(B, 128, Qp))
        pos1 = f32e(B, Q, 2)
        p = ops.TokenProgram(roles=len(cls) + 3)
        p.load_parts(2, wsf, nh, M, c['b2'])
        p.load(1, x2)
        p.rowop(2, 2, aux=1, ln=c['n3'], eps=c['eps'][2])                            # x3
        p.store(2, x3, roles=(0, 0))
        self._heads_and_next(p, 2, 0, 1, dict(w1=heads0[0], b1=heads0[1], w2=heads0[2], b2=heads0[3], qpos=pos, outs=first,
                                              cls=cls, center_head=ic, ldo=Q, col0=0, pos_out=pos1),
                             (blk_consts[0]['sa'][0], blk_consts[0]['sa'][1]), qkv_b)
        p.run(B, Q)
        x, pos = x3, pos1

        geom = dec.static_geometry if dec.static_geometry is not None else QueryGeometry(img_metas, dev)
        maps = img_feat.view(B * V, C, I_H, I_W)
        cfg, bc = dec.test_cfg, dec.bbox_coder
        cell_bev = bc.out_size_factor * bc.voxel_size[0]
        dec.on_the_image_mask = []
        look, keep_cum = pos, None                   # ++: look-forward centres (:281), cumulative on-the-image mask (:295-302)
        res, ld, col = dict(zip(names, first)), Q, 0                                 # where the previous stage's boxes are
        for l in range(L):
            blk = dec.decode_head[l]
            r32 = {k: res[k][:, :, col:col + Q] for k in ('center', 'height', 'dim', 'rot')}
            hc = self._c(('ph', l), dec.pred_head[l], lambda: _heads_consts(dec.pred_head[l]))
            pos_next = f32e(B, Q, 2)
            nxt = (blk_consts[l + 1]['sa'][0], blk_consts[l + 1]['sa'][1]) if l + 1 < L else None
            heads = dict(w1=hc[0], b1=hc[1], w2=hc[2], b2=hc[3], qpos=pos, outs=final, cls=cls, center_head=ic, ldo=L * Q,
                         col0=l * Q, pos_out=pos_next)
            if pp:      # centre = offset + look-forward (:293); the next look-forward = offset + this stage's query position (:294)
                look_next = f32e(B, Q, 2)
                heads.update(qpos=look, qpos2=pos, pos2_out=look_next)
            # The RoI side (box geometry -> RoIs -> RoIAlign) and the token side (self attention among the queries) of a
            # block are independent up to the DynamicConv core and CAN run as two branches (DI_OVERLAP bit 32; the image
            # block's attention is masked by the per-view membership, so only its RoIAlign runs beside it).  Measured on
            # one box, one sample at a time: 1.519-1.527 ms with the fork against 1.503-1.510 without (two in flight: 896
            # against 902 samples/s) - the fork / join events cost more than the 6-18 us they hide.  Off by default.
            par = (lambda a, b: fork_join(dev, a, b)) if utils.OVERLAP & 32 else (lambda a, b: [a(), b()])
            if l % 2 == 0:
                on, rect, _ = ops.query_geometry(r32, geom.proj, geom.aug_rev, geom.per_sample,
                                                 cfg['out_size_factor'] * cfg['voxel_size'][0], cfg['pc_range'][:2],
                                                 1.0, 1.0, True, False, ld=ld)
                rois, view, member, keep, on_img = ops.roi_select(rect, on)
                sfeat = self._self_feature_img(blk_consts[l], x, qkv_b, view, member, B, Q, V) if pp else None
                y, roi = par(lambda: self._attend(blk, '', x, qkv_b, B, Q, member, view),
                             lambda: ops.roi_align(maps, rois, 1.0 / blk.out_size_factor_img, out_hl=True))   # (B*Q,49,2C) hi | lo
                if pp:
                    keep_cum = keep if keep_cum is None else keep & keep_cum
                heads.update(keep=keep_cum if pp else keep, first=first)
                x, qkv_b = self._refine(blk, '', x, y, roi, B, Q, heads, nxt, keep, sfeat)
                dec.on_the_image_mask.append((keep_cum if pp else keep).view(B, Q).bool())
            else:
                def rois_bev():
                    _, _, rect = ops.query_geometry(r32, None, None, None, cell_bev, bc.pc_range[:2], cell_bev, 2.0,
                                                    False, True, ld=ld)
                    return ops.roi_align(new_lidar_feat, ops.roi_select(rect), 1.0, out_hl=True)
                y, roi = par(lambda: self._attend(blk, '_pts', x, qkv_b, B, Q), rois_bev)
                sfeat = None
                if pp:                                        # every stage is merged with the first one's result (:299-302)
                    sfeat = self._self_feature_pts(blk_consts[l], y[0], B, Q)
                    heads.update(keep=keep_cum, first=first)
                    dec.on_the_image_mask.append(dec.on_the_image_mask[-1])
                x, qkv_b = self._refine(blk, '_pts', x, y, roi, B, Q, heads, nxt, None, sfeat)
            pos = pos_next
            if pp:
                look = look_next
            res, ld, col = dict(zip(names, final)), L * Q, l * Q

        out = dict(zip(names, final))
        top_index = top % HW
        out['query_heatmap_score'] = heatmap.gather(index=top_index[:, None, :].expand(-1, ncls, -1), dim=-1)
        out['dense_heatmap'] = dense_heatmap_img
        return [[out]]
