"""The two mmcv-full 1.3.18 transformer bricks DeepInteraction++ builds on (reference imports them at
necks/fusion_transformerv4.py:21-22 and models/utils/decoder_utils.py:14), MI355X-native: same constructor
arguments, parameter names and maths, token-major (bs, N, C) execution with the deformable sampling in one HIP
kernel (csrc/plusplus.hip) instead of mmcv's CUDA op.  mmcv is not a dependency."""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .... import ops
from ....autograd import MSDeformAttn
from ....utils import param_key


def post_norm(norm, x, branch=None):
    """`norm(x + branch)` (branch may be None) - the post-norm step.  One fused kernel at inference for C <= 128
    fp16/fp32 tokens on the GPU (ops.add_layernorm), the plain modules otherwise."""
    if (x.is_cuda and not torch.is_grad_enabled() and isinstance(norm, nn.LayerNorm) and norm.elementwise_affine
            and x.shape[-1] <= 128 and x.shape[-1] % 8 == 0 and x.dtype in (torch.float16, torch.float32)
            and norm.weight.dtype == x.dtype):
        return ops.add_layernorm(x.contiguous(), None if branch is None else branch.contiguous(), norm.weight,
                                 norm.bias, norm.eps)
    return norm(x if branch is None else x + branch)


def fused_tokens_ok(x, *mods):
    """fp16 inference on the GPU over 128-channel tokens: the form the fused chain kernels of csrc/pointwise.hip take (they
    replace the library GEMM + bias + residual + LayerNorm launches of a transformer block)."""
    return (x.is_cuda and x.dtype == torch.float16 and x.shape[-1] == 128 and not torch.is_grad_enabled()
            and all(not m.training for m in mods))


def module_cache(owner, name, key_mods, build):
    """A derived constant of `owner` (packed / padded weights for a kernel), rebuilt when a parameter of `key_mods` changes."""
    key = tuple(param_key(m) for m in key_mods)
    hit = owner.__dict__.get(name)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            hit = (key, build())
        owner.__dict__[name] = hit
    return hit[1]


def padded128(w, b):
    """(rows <= 128, 128) weight and its bias padded with zero rows to the 128 outputs a chain computes (float32)."""
    W = torch.zeros(128, 128, dtype=torch.float32, device=w.device)
    B = torch.zeros(128, dtype=torch.float32, device=w.device)
    W[:w.shape[0]] = w.float()
    B[:b.shape[0]] = b.float()
    return W, B


def linear128(lin, x, relu=False):
    """nn.Linear(128, 128) on fp16 inference tokens through the fused chain kernel (ops.token_linear)."""
    w, b = module_cache(lin, '_di_w16', [lin], lambda: (lin.weight.detach().to(torch.float16).contiguous(),
                                                        lin.bias.detach().float().contiguous()))
    shape = x.shape
    return ops.token_linear(x.reshape(-1, 128).contiguous(), w, b, relu).view(shape)


class TransFFN(nn.Module):
    """mmcv `FFN`: identity + Dropout(Linear(Dropout(act(Linear(x))))); keys `layers.0.0.*`, `layers.1.*`."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=dict(type='ReLU', inplace=True),
                 ffn_drop=0., dropout_layer=None, add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        assert num_fcs >= 2 and act_cfg.get('type', 'ReLU') == 'ReLU' and dropout_layer is None
        self.embed_dims, self.feedforward_channels, self.num_fcs = embed_dims, feedforward_channels, num_fcs
        layers, cin = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(cin, feedforward_channels), nn.ReLU(inplace=True),
                                        nn.Dropout(ffn_drop)))
            cin = feedforward_channels
        layers += [nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop)]
        self.layers = nn.Sequential(*layers)
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return out
        return (x if identity is None else identity) + out

    def _fusable(self, norm, x):
        return (x.is_cuda and x.dtype == torch.float16 and not torch.is_grad_enabled() and not self.training
                and self.add_identity and self.num_fcs == 2 and self.embed_dims == 128
                and self.feedforward_channels % 128 == 0 and self.feedforward_channels <= 1024
                and isinstance(norm, nn.LayerNorm) and norm.elementwise_affine and norm.normalized_shape == (128,)
                and norm.weight.dtype == torch.float16 and self.layers[0][0].weight.dtype == torch.float16)

    def then_norm(self, norm, x):
        """norm(self(x)) with the identity add folded into the normalisation kernel; fp16 inference with 128-channel
        tokens: the whole FFN + norm is ONE kernel (ops.ffn_ln: the hidden activation never reaches memory)."""
        if self._fusable(norm, x):
            l1, l2 = self.layers[0][0], self.layers[1]
            key = param_key(self)
            hit = self.__dict__.get('_ffn_images')
            if hit is None or hit[0] != key:
                with torch.no_grad():
                    hit = (key, ops.ffn_images(l1.weight, l1.bias, l2.weight, l2.bias))
                self.__dict__['_ffn_images'] = hit
            shape = x.shape
            y = ops.ffn_ln(x.reshape(-1, 128).contiguous(), hit[1], norm.weight, norm.bias, norm.eps)
            return y.view(shape)
        if not self.add_identity:
            return post_norm(norm, self.layers(x))
        return post_norm(norm, x, self.layers(x))


class MultiScaleDeformableAttention(nn.Module):
    """mmcv `MultiScaleDeformableAttention` (defaults: 8 heads, 4 points, dropout 0.1); parameters
    `sampling_offsets`, `attention_weights`, `value_proj`, `output_proj`.

    Inference on the GPU: ONE packed GEMM produces offsets and logits, `ops.ms_deform_attn` does softmax, location
    arithmetic, bilinear gathers and the weighted sum."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64, dropout=0.1,
                 batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__()
        assert embed_dims % num_heads == 0
        self.embed_dims, self.num_heads, self.num_levels, self.num_points = embed_dims, num_heads, num_levels, num_points
        self.batch_first, self.im2col_step = batch_first, im2col_step
        self.dropout = nn.Dropout(dropout)
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self._pack_cache = None
        self.init_weights()

    def init_weights(self):
        nn.init.constant_(self.sampling_offsets.weight, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 2).repeat(
            1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.)

    def packed(self):
        """[sampling_offsets ; attention_weights] as one (heads*L*P*3, C) projection (inference cache)."""
        if torch.is_grad_enabled():
            return (torch.cat([self.sampling_offsets.weight, self.attention_weights.weight]),
                    torch.cat([self.sampling_offsets.bias, self.attention_weights.bias]))
        key = param_key(self)
        if self._pack_cache is None or self._pack_cache[0] != key:
            with torch.no_grad():
                self._pack_cache = (key, (torch.cat([self.sampling_offsets.weight, self.attention_weights.weight]).contiguous(),
                                          torch.cat([self.sampling_offsets.bias, self.attention_weights.bias]).contiguous()))
        return self._pack_cache[1]

    def projection_images(self):
        """The offset / logit projections as chain images for the fused kernels (inference cache): one 128-output chain when
        both fit (one level: 64 + 32 columns), else one chain per projection (two levels: 128 offsets | 64 logits + padding)."""
        def build():
            so, aw = self.sampling_offsets, self.attention_weights
            if so.out_features + aw.out_features <= 128:
                return [ops.chain_image(*padded128(torch.cat([so.weight, aw.weight]), torch.cat([so.bias, aw.bias])))]
            assert so.out_features <= 128 and aw.out_features <= 128
            return [ops.chain_image(*padded128(so.weight, so.bias)), ops.chain_image(*padded128(aw.weight, aw.bias))]
        return module_cache(self, '_proj_images', [self.sampling_offsets, self.attention_weights], build)

    def can_fuse_tokens(self, query, key_padding_mask=None):
        """The fused fp16 inference form (chain-kernel projections, head-major value map, `ms_deform_attn_hm_kernel`) covers
        what its kernels cover - 128 channels in 8 heads of 16, 4 points on 1 or 2 levels, no padding mask; anything else
        (mmcv's default num_levels=4, other num_points) takes the generic kernel path."""
        return self.kernels_cover() and key_padding_mask is None and fused_tokens_ok(query, self)

    def kernels_cover(self):
        """The deformable-attention kernels (csrc/plusplus.hip, forward and backward) cover the reference configurations: 128
        channels in 8 heads, 4 points on 1 or 2 levels.  Anything else (mmcv's default num_levels=4, other point counts)
        runs `_core_torch` - mmcv's own `multi_scale_deformable_attn_pytorch` formulation on torch ops."""
        return self.embed_dims == 128 and self.num_heads == 8 and self.num_points == 4 and self.num_levels in (1, 2)

    def _core_torch(self, v, offsets, logits, ref, shapes):
        """mmcv `multi_scale_deformable_attn_pytorch` (mmcv/ops/multi_scale_deform_attn.py): per level F.grid_sample
        (bilinear, zero padding, align_corners=False) of the head-split value map at reference + offset / (W, H), weighted
        by the soft-max over levels x points.  v (bs,S,C), offsets (bs,nq,heads*L*P*2), logits (bs,nq,heads*L*P)."""
        bs, S, C = v.shape
        nq, Hh, L, P = offsets.shape[1], self.num_heads, self.num_levels, self.num_points
        d = C // Hh
        w = logits.float().view(bs, nq, Hh, L * P).softmax(-1).view(bs, nq, Hh, L, P)
        norm = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float32, device=v.device)
        refb = ref.float().expand(bs, -1, -1, -1) if ref.shape[0] == 1 else ref.float()
        loc = refb[:, :, None, :, None, :] + offsets.float().view(bs, nq, Hh, L, P, 2) / norm[None, None, None, :, None, :]
        grids = 2 * loc - 1
        vals = v.float().view(bs, S, Hh, d).split([h_ * w_ for h_, w_ in shapes], dim=1)
        acc = []
        for l, (h_, w_) in enumerate(shapes):
            vl = vals[l].flatten(2).transpose(1, 2).reshape(bs * Hh, d, h_, w_)
            gl = grids[:, :, :, l].transpose(1, 2).flatten(0, 1)                      # (bs*heads, nq, P, 2)
            acc.append(F.grid_sample(vl, gl, mode='bilinear', padding_mode='zeros', align_corners=False))
        att = w.transpose(1, 2).reshape(bs * Hh, 1, nq, L * P)
        out = (torch.stack(acc, dim=-2).flatten(-2) * att).sum(-1).view(bs, Hh * d, nq)
        return out.transpose(1, 2).contiguous().to(v.dtype)

    def can_fuse_norm(self, query, norm):
        """LayerNorm(identity + output_proj(.)) as the epilogue of the output projection (ops.linear_ln)."""
        return (self.batch_first and self.embed_dims == 128 and fused_tokens_ok(query, self) and isinstance(norm, nn.LayerNorm)
                and norm.elementwise_affine and norm.weight.dtype == torch.float16)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, then_norm=None, with_sum=False,
                projected_value=None, **kwargs):
        """`then_norm`: an nn.LayerNorm applied to the result (the layer's next 'norm' step) - the residual add is
        then folded into the normalisation kernel; `with_sum` (with `then_norm`, fused form only): also return the
        un-normalised result -> (normed, sum).  `projected_value`: value_proj(value) when the caller already has it - HEAD-MAJOR
        (bs, 8, S, 16) for the fused fp16 inference form (`can_fuse_tokens`), (bs, S, C) otherwise."""
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        shapes = [(int(h), int(w)) for h, w in spatial_shapes]
        bs, nq, _ = query.shape
        fused = self.can_fuse_tokens(query, key_padding_mask)
        n_off = self.num_heads * self.num_levels * self.num_points * 2
        n_log = n_off // 2
        ref = reference_points.to(torch.float32).contiguous()
        if fused:
            # every projection through the fused chain kernels: no library GEMM, the query is read once for offsets + logits;
            # the value map HEAD-MAJOR (bs, 8, S, 16): the gathers of ops.ms_deform_attn fetch 64-byte pieces
            if projected_value is not None:
                v = projected_value
            else:
                wv, bv = module_cache(self.value_proj, '_di_w16', [self.value_proj],
                                      lambda: (self.value_proj.weight.detach().to(torch.float16).contiguous(),
                                               self.value_proj.bias.detach().float().contiguous()))
                v = ops.token_linear_hm(value.reshape(-1, 128).contiguous(), wv, bv, value.shape[1])
            images = self.projection_images()
            ys = ops.token_linear_multi(query.reshape(-1, 128).contiguous(), images)
            if len(images) == 1:
                proj = ys[0].view(bs, nq, 128)
                offsets, logits = proj[..., :n_off], proj[..., n_off:n_off + n_log]
            else:
                offsets, logits = ys[0].view(bs, nq, 128)[..., :n_off], ys[1].view(bs, nq, 128)[..., :n_log]
            out = ops.ms_deform_attn(v, offsets, logits, ref, shapes, self.num_points, head_major=True)
        else:
            v = self.value_proj(value) if projected_value is None else projected_value
            if key_padding_mask is not None:
                v = v.masked_fill(key_padding_mask[..., None], 0.0)
            w, b = self.packed()
            proj = F.linear(query, w, b)                                           # (bs, nq, heads*L*P*3)
            if not (self.kernels_cover() and v.is_cuda):
                out = self._core_torch(v, proj[..., :n_off], proj[..., n_off:], ref, shapes)
            elif torch.is_grad_enabled() and (v.requires_grad or proj.requires_grad):
                out = MSDeformAttn.apply(v.contiguous(), proj, ref, shapes, self.num_points)
            else:
                out = ops.ms_deform_attn(v.contiguous(), proj[..., :n_off], proj[..., n_off:], ref, shapes, self.num_points)
        if then_norm is not None and self.can_fuse_norm(out, then_norm) and identity.shape == out.shape:
            # output projection + residual + post-norm in one kernel (ops.linear_ln)
            image = module_cache(self, '_out_image', [self.output_proj],
                                 lambda: ops.chain_image(self.output_proj.weight.float(), self.output_proj.bias.float()))
            y = ops.linear_ln(out.reshape(-1, 128).contiguous(), image, identity.reshape(-1, 128).contiguous(),
                              then_norm.weight, then_norm.bias, then_norm.eps, with_sum=with_sum)
            return (y[0].view(out.shape), y[1].view(out.shape)) if with_sum else y.view(out.shape)
        assert not with_sum, 'with_sum belongs to the fused form (check can_fuse_norm first)'
        out = linear128(self.output_proj, out) if fused else self.output_proj(out)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        if then_norm is not None:
            return post_norm(then_norm, identity, self.dropout(out))
        return self.dropout(out) + identity
