"""The image side of the reference detector in front of the hot path: `DeepInteraction.extract_img_feat` (reference
`models/detectors/deepinteraction.py:100-118`) with the configuration `Fusion_0075_refactor.py:132-145` - a ResNet-50
(`norm_eval=True`, four stage outputs) and an FPN(256, num_outs=5), both mmdet 2.14.0 classes - as a FROZEN inference
stand-in on ROCm.  The convolutions run through torch / MIOpen: this is plumbing in front of the path, not a kernel of it
(VERDICT round 2, item 10).

What is MI355X-specific here is the form the frozen network is kept in:
  * every BatchNorm is folded into its convolution when the checkpoint is loaded (53 normalisation launches and their
    fp32 statistics reads disappear; the folded weights are computed in fp32 and rounded once);
  * weights and activations are fp16, channels-last - the layout the hot path's kernels read (`DESIGN.md` section 2), so
    the encoder's input needs no layout pass (the NCHW hand-over costs ~60 us per sample, `bench.py --from-raw`);
  * only what the requested FPN levels depend on is computed: the v1 neck reads level 0 only
    (`deepinteraction.py:146`), DeepInteraction++ levels 0-1 (:148) - the 3x3 output convolutions of the other levels and
    the max-pool level are skipped.  `levels=None` computes all `num_outs` levels as the reference does;
  * no host synchronisation and only static shapes: the module can be captured into the same hipGraph as the hot path.

`FrozenSwinFPN` is the same for the DeepInteraction++ configuration (`Fusion_0075_plusplus.py:147-170`: the plugin's own
Swin-T, reference `models/backbones/swin.py`, in front of the same FPN) - see its docstring.

`load_mmdet_state` takes the two state dicts in mmdet's own key layout (the reference's checkpoints: `img_backbone.*`,
`img_neck.*` with the prefix stripped), checks every expected key and shape, and fails loudly on a mismatch.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

_STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}
_BN_EPS = 1e-5


def _fold(conv_w, bn, prefix):
    """conv (no bias) followed by evaluation-mode BatchNorm == conv with scaled weights and a bias (fp32 arithmetic)."""
    gamma, beta = bn[prefix + '.weight'].float(), bn[prefix + '.bias'].float()
    mean, var = bn[prefix + '.running_mean'].float(), bn[prefix + '.running_var'].float()
    scale = gamma / torch.sqrt(var + _BN_EPS)
    return conv_w.float() * scale[:, None, None, None], beta - mean * scale


def _take(sd, key, shape, what):
    if key not in sd:
        raise KeyError(f'{what}: missing `{key}`')
    t = sd[key]
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f'{what}: `{key}` has shape {tuple(t.shape)}, expected {tuple(shape)}')
    return t


class _FrozenImageNet(nn.Module):
    """What the two frozen image networks share: derived (non-persistent) weight buffers and the FPN with level pruning."""

    def __init__(self, stage_channels, out_channels, num_outs, levels, dtype):
        super().__init__()
        self.stage_channels = list(stage_channels)
        self.out_channels, self.num_outs, self.dtype = out_channels, num_outs, dtype
        self.levels = tuple(range(num_outs)) if levels is None else tuple(sorted(set(levels)))
        assert self.levels and 0 <= self.levels[0] and self.levels[-1] < num_outs
        self._loaded = False

    def _needed(self):
        """(lowest lateral level to build, set of 3x3 output convolutions to run) for the requested levels: level l < 4
        needs the laterals l..3 (top-down sums) and its own output convolution; the pooled levels >= 4 need output 3."""
        outs = {min(l, 3) for l in self.levels}
        return min(outs), outs

    def _put_t(self, key, t):
        if hasattr(self, key):
            delattr(self, key)
        self.register_buffer(key, t, persistent=False)

    def _put(self, name, w, b):
        key = name.replace('.', '_')
        self._put_t(key + '_w', w.to(self.dtype).contiguous(memory_format=torch.channels_last) if w.dim() == 4
                    else w.to(self.dtype).contiguous())
        self._put_t(key + '_b', b.to(self.dtype).contiguous())
        self._put_t(key + '_b32', b.float().contiguous())       # the one-pass epilogue's bias (ops.bias_act_)

    def _get(self, name):
        key = name.replace('.', '_')
        return getattr(self, key + '_w'), getattr(self, key + '_b')

    # One-pass epilogue (csrc/epilogue.hip) behind the library convolution on the device: torch's conv2d-with-bias is MIOpen's
    # convolution + a bias kernel, `relu_` and `add_` one more pass each - in the ResNet-50 these passes took as long as the
    # convolutions.  DI_FUSED_EPILOGUE=0: the torch statements (A/B, and what CPU tensors / float32 run).
    FUSED = os.environ.get('DI_FUSED_EPILOGUE', '1') != '0'

    def _conv(self, x, name, stride=1, pad=0, relu=False, residual=None):
        w, b = self._get(name)
        if self.FUSED and x.is_cuda and x.dtype == torch.float16 and w.shape[0] % 8 == 0:
            from .... import ops
            y = F.conv2d(x, w, None, stride=stride, padding=pad)
            if not y.is_contiguous(memory_format=torch.channels_last):
                y = y.contiguous(memory_format=torch.channels_last)
            if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
                residual = residual.contiguous(memory_format=torch.channels_last)
            return ops.bias_act_(y, getattr(self, name.replace('.', '_') + '_b32'), residual, relu)
        y = F.conv2d(x, w, b, stride=stride, padding=pad)
        if residual is not None:
            y = y.add_(residual)
        return y.relu_() if relu else y

    def _load_neck(self, neck_state):
        C = self.out_channels
        for i, cin in enumerate(self.stage_channels):
            for kind, ci, k in (('lateral_convs', cin, 1), ('fpn_convs', C, 3)):
                w = _take(neck_state, f'{kind}.{i}.conv.weight', (C, ci, k, k), 'img_neck')
                b = _take(neck_state, f'{kind}.{i}.conv.bias', (C,), 'img_neck')
                self._put(f'{kind}.{i}', w.float(), b.float())

    def _synthetic_neck(self, g):
        nk, C = {}, self.out_channels
        for i, cin in enumerate(self.stage_channels):
            for kind, ci, k in (('lateral_convs', cin, 1), ('fpn_convs', C, 3)):
                # the output convolutions bring the maps to the magnitude of `synth.make_inputs`' feature maps (std ~0.3)
                gain = 0.05 if kind == 'fpn_convs' else 1.0
                nk[f'{kind}.{i}.conv.weight'] = gain * torch.randn(C, ci, k, k, generator=g) * (1.0 / (ci * k * k)) ** 0.5
                nk[f'{kind}.{i}.conv.bias'] = gain * 0.05 * torch.randn(C, generator=g)
        return nk

    def _fpn(self, feats):
        """mmdet FPN (start_level 0, nearest up-sampling to the finer map's size, extra levels = stride-2 subsamplings of
        the last output), restricted to what `self.levels` depends on.  feats: the four stage maps, channels-last."""
        low, out_convs = self._needed()
        lat = {i: self._conv(feats[i], f'lateral_convs.{i}') for i in range(low, 4)}
        for i in range(3, low, -1):
            lo, hi = lat[i - 1], lat[i]
            if (self.FUSED and lo.is_cuda and lo.dtype == torch.float16 and lo.shape[1] % 8 == 0
                    and lo.is_contiguous(memory_format=torch.channels_last) and hi.is_contiguous(memory_format=torch.channels_last)):
                from .... import ops
                ops.upsample_add_(lo, hi)
            else:
                lo.add_(F.interpolate(hi, size=lo.shape[2:], mode='nearest'))
        outs = {i: self._conv(lat[i], f'fpn_convs.{i}', 1, 1) for i in out_convs}
        for l in range(4, self.levels[-1] + 1):
            outs[l] = F.max_pool2d(outs[l - 1], 1, stride=2)
        return tuple(outs[l] for l in self.levels)


class FrozenResNetFPN(_FrozenImageNet):
    def __init__(self, depth=50, out_channels=256, num_outs=5, levels=(0,), dtype=torch.float16):
        assert depth in _STAGE_BLOCKS, 'Bottleneck ResNets only (the reference configures depth 50)'
        super().__init__([256 << s for s in range(4)], out_channels, num_outs, levels, dtype)
        self.depth = depth
        self._plan = self._make_plan()

    # ------------------------------------------------------------------ structure
    def _make_plan(self):
        """The list of convolutions, in execution order: (name, cin, cout, k, stride)."""
        plan = [('conv1', 3, 64, 7, 2)]
        inplanes = 64
        for s, blocks in enumerate(_STAGE_BLOCKS[self.depth]):
            planes, stride = 64 << s, 1 if s == 0 else 2
            for j in range(blocks):
                p = f'layer{s + 1}.{j}'
                st = stride if j == 0 else 1
                plan += [(p + '.conv1', inplanes, planes, 1, 1), (p + '.conv2', planes, planes, 3, st),
                         (p + '.conv3', planes, planes * 4, 1, 1)]
                if j == 0 and (st != 1 or inplanes != planes * 4):
                    plan.append((p + '.downsample', inplanes, planes * 4, 1, st))
                inplanes = planes * 4
        return plan

    @torch.no_grad()
    def load_mmdet_state(self, backbone_state, neck_state):
        """backbone_state / neck_state: the `img_backbone.` / `img_neck.` parts of a reference checkpoint (mmdet key
        names, prefix stripped).  Every convolution of the plan must be present with the right shape."""
        for name, cin, cout, k, _ in self._plan:
            if name.endswith('.downsample'):
                w = _take(backbone_state, name + '.0.weight', (cout, cin, k, k), 'img_backbone')
                bn = {'bn.' + s: _take(backbone_state, f'{name}.1.{s}', (cout,), 'img_backbone')
                      for s in ('weight', 'bias', 'running_mean', 'running_var')}
            else:
                w = _take(backbone_state, name + '.weight', (cout, cin, k, k), 'img_backbone')
                bn_name = 'bn1' if name == 'conv1' else name.replace('.conv', '.bn')
                bn = {'bn.' + s: _take(backbone_state, f'{bn_name}.{s}', (cout,), 'img_backbone')
                      for s in ('weight', 'bias', 'running_mean', 'running_var')}
            self._put(name, *_fold(w, bn, 'bn'))
        self._load_neck(neck_state)
        self._loaded = True
        return self

    def synthetic_state(self, seed=0):
        """Random-init state dicts in the checkpoint layout (there is no network for the reference's weights): He-scaled
        convolutions, BatchNorm statistics away from the identity.  -> (backbone_state, neck_state)."""
        g = torch.Generator().manual_seed(seed)
        bb = {}
        for name, cin, cout, k, _ in self._plan:
            w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
            if name.endswith('.downsample'):
                wkey, bn = name + '.0.weight', name + '.1'
            else:
                wkey, bn = name + '.weight', 'bn1' if name == 'conv1' else name.replace('.conv', '.bn')
            bb[wkey] = w
            # the last normalisation of a block starts small, as a trained residual branch is: activations stay O(1)
            gain = 0.25 if name.endswith('.conv3') else 1.0
            bb[bn + '.weight'] = gain * (0.75 + 0.5 * torch.rand(cout, generator=g))
            bb[bn + '.bias'] = 0.05 * torch.randn(cout, generator=g)
            bb[bn + '.running_mean'] = 0.05 * torch.randn(cout, generator=g)
            bb[bn + '.running_var'] = 0.75 + 0.5 * torch.rand(cout, generator=g)
        nk = self._synthetic_neck(g)
        return bb, nk

    # ------------------------------------------------------------------ forward
    def _bottleneck(self, x, p, stride, has_down):
        identity = self._conv(x, p + '.downsample', stride) if has_down else x
        y = self._conv(x, p + '.conv1', relu=True)
        y = self._conv(y, p + '.conv2', stride, 1, relu=True)
        return self._conv(y, p + '.conv3', relu=True, residual=identity)         # conv3 + bias, + identity, ReLU

    @torch.no_grad()
    def forward(self, img):
        """img: (B*N, 3, H, W), any float dtype / layout -> tuple of the requested FPN levels, channels-last `dtype`,
        in the order of `self.levels`."""
        if not self._loaded:
            raise RuntimeError('FrozenResNetFPN: no weights loaded (load_mmdet_state)')
        x = img.to(dtype=self.dtype, memory_format=torch.channels_last)
        if self.FUSED and x.is_cuda and x.dtype == torch.float16:
            # max-pool(relu(y + b)) == relu(max-pool(y) + b) exactly (per-channel constant, monotonic, rounding is monotonic too):
            # the epilogue pass runs on the pooled map, a quarter of the stem's output
            from .... import ops
            w, _ = self._get('conv1')
            x = F.max_pool2d(F.conv2d(x, w, None, stride=2, padding=3), kernel_size=3, stride=2, padding=1)
            x = ops.bias_act_(x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last),
                              self.conv1_b32, None, True)
        else:
            x = self._conv(x, 'conv1', 2, 3, relu=True)
            x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        feats = []
        for s, blocks in enumerate(_STAGE_BLOCKS[self.depth]):
            for j in range(blocks):
                p = f'layer{s + 1}.{j}'
                x = self._bottleneck(x, p, 2 if (j == 0 and s > 0) else 1, j == 0)
            feats.append(x)
        return self._fpn(feats)


class FrozenSwinFPN(_FrozenImageNet):
    """The image side of the DeepInteraction++ configuration (`Fusion_0075_plusplus.py:147-170`): the plugin's own
    `SwinTransformer` (reference `models/backbones/swin.py:461-754`, Swin-T: embed 96, depths 2-2-6-2, heads 3-6-12-24, 7x7
    windows, `patch_norm`, all four stage outputs normalised) + mmdet `FPN([96,192,384,768] -> 256, num_outs=5)`, as a frozen
    inference stand-in through torch ops, reading the reference checkpoint's own keys (`img_backbone.` / `img_neck.`).

    Kept for this machine as: tokens in (B, H, W, C) - a stage's normalised output IS the channels-last map the FPN's
    convolutions read (a permuted view, no copy); the 1/sqrt(d) query scale folded into the qkv weights; every block's
    relative-position bias gathered once at load time to (heads, 49, 49), and for the shifted blocks summed with the
    region mask of the padded map size on first use (static shapes: one additive term per block); evaluation-mode only
    (DropPath / Dropout are the identity).  Semantics followed exactly, including the reference's zero-padding of the
    token map to multiples of the window AFTER norm1 (the padded tokens take part in the attention of their windows as
    keys, swin.py:181-185) and the channel-major 2x2 unfold order of `PatchMerging` (swin_utils.py:329-374)."""

    def __init__(self, embed_dims=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4,
                 patch_size=4, out_channels=256, num_outs=5, levels=(0, 1), dtype=torch.float16):
        super().__init__([embed_dims << s for s in range(4)], out_channels, num_outs, levels, dtype)
        assert len(depths) == 4 and len(num_heads) == 4
        self.embed_dims, self.depths, self.num_heads = embed_dims, tuple(depths), tuple(num_heads)
        self.window, self.mlp_ratio, self.patch = window_size, mlp_ratio, patch_size
        self._bias_cache = {}

    # ------------------------------------------------------------------ weights
    def _entries(self):
        """(checkpoint key, shape) of every backbone entry the forward uses."""
        E, P, N = self.embed_dims, self.patch, self.window * self.window
        out = [('patch_embed.projection.weight', (E, 3, P, P)), ('patch_embed.projection.bias', (E,)),
               ('patch_embed.norm.weight', (E,)), ('patch_embed.norm.bias', (E,))]
        for s, (depth, heads) in enumerate(zip(self.depths, self.num_heads)):
            C = E << s
            for j in range(depth):
                b = f'stages.{s}.blocks.{j}.'
                out += [(b + 'norm1.weight', (C,)), (b + 'norm1.bias', (C,)),
                        (b + 'attn.w_msa.relative_position_bias_table', ((2 * self.window - 1) ** 2, heads)),
                        (b + 'attn.w_msa.relative_position_index', (N, N)),
                        (b + 'attn.w_msa.qkv.weight', (3 * C, C)), (b + 'attn.w_msa.qkv.bias', (3 * C,)),
                        (b + 'attn.w_msa.proj.weight', (C, C)), (b + 'attn.w_msa.proj.bias', (C,)),
                        (b + 'norm2.weight', (C,)), (b + 'norm2.bias', (C,)),
                        (b + 'ffn.layers.0.0.weight', (self.mlp_ratio * C, C)), (b + 'ffn.layers.0.0.bias', (self.mlp_ratio * C,)),
                        (b + 'ffn.layers.1.weight', (C, self.mlp_ratio * C)), (b + 'ffn.layers.1.bias', (C,))]
            if s < 3:
                d = f'stages.{s}.downsample.'
                out += [(d + 'norm.weight', (4 * C,)), (d + 'norm.bias', (4 * C,)), (d + 'reduction.weight', (2 * C, 4 * C))]
            out += [(f'norm{s}.weight', (C,)), (f'norm{s}.bias', (C,))]
        return out

    @staticmethod
    def _key(name):
        return 'sw_' + name.replace('.', '_')

    def _p(self, name):
        return getattr(self, self._key(name))

    @torch.no_grad()
    def load_mmdet_state(self, backbone_state, neck_state):
        for key, shape in self._entries():
            t = _take(backbone_state, key, shape, 'img_backbone')
            if key.endswith('relative_position_index'):
                continue
            if key.endswith('relative_position_bias_table'):      # -> (heads, 49, 49), gathered once
                index = backbone_state[key.replace('bias_table', 'index')].reshape(-1).long()
                N = self.window * self.window
                t = t.float()[index].view(N, N, -1).permute(2, 0, 1)
            elif key.endswith('qkv.weight') or key.endswith('qkv.bias'):   # q <- q / sqrt(d), folded (fp32, one rounding)
                s_idx = int(key.split('.')[1])
                C = self.embed_dims << s_idx
                t = t.float().clone()
                t[:C] *= (C // self.num_heads[s_idx]) ** -0.5
            elif key == 'patch_embed.projection.weight':
                t = t.float().contiguous(memory_format=torch.channels_last)
            self._put_t(self._key(key), t.to(self.dtype).contiguous(memory_format=torch.channels_last) if t.dim() == 4
                        else t.to(self.dtype).contiguous())
        self._load_neck(neck_state)
        self._bias_cache = {}
        self._loaded = True
        return self

    def synthetic_state(self, seed=0):
        """Random-init state dicts in the checkpoint layout -> (backbone_state, neck_state)."""
        g = torch.Generator().manual_seed(seed)
        W = self.window
        coords = torch.stack(torch.meshgrid(torch.arange(W), torch.arange(W), indexing='ij')).flatten(1)
        rel = coords[:, :, None] - coords[:, None, :] + (W - 1)
        index = rel[0] * (2 * W - 1) + rel[1]                       # the standard Swin pair index (== the reference's buffer)
        bb = {}
        for key, shape in self._entries():
            if key.endswith('relative_position_index'):
                bb[key] = index.clone()
            elif key.endswith('norm.weight') or key.endswith('norm1.weight') or key.endswith('norm2.weight') or \
                    (key.startswith('norm') and key.endswith('.weight')):
                bb[key] = 0.75 + 0.5 * torch.rand(shape, generator=g)
            elif key.endswith('.bias'):
                bb[key] = 0.05 * torch.randn(shape, generator=g)
            elif key.endswith('bias_table'):
                bb[key] = 0.5 * torch.randn(shape, generator=g)
            else:
                fan = 1
                for d in shape[1:]:
                    fan *= d
                bb[key] = torch.randn(shape, generator=g) * fan ** -0.5
        return bb, self._synthetic_neck(g)

    # ------------------------------------------------------------------ forward
    def _ln(self, x, name):
        return F.layer_norm(x, x.shape[-1:], self._p(name + '.weight'), self._p(name + '.bias'), 1e-5)

    def _region_mask(self, Hp, Wp, device):
        """(nW, 49, 49): 0 inside a region of the cyclically shifted map, -100 across regions (swin.py:195-217)."""
        W, sh = self.window, self.window // 2
        ids = torch.zeros(Hp, Wp, device=device)
        cnt = 0
        for h in (slice(0, -W), slice(-W, -sh), slice(-sh, None)):
            for w in (slice(0, -W), slice(-W, -sh), slice(-sh, None)):
                ids[h, w] = cnt
                cnt += 1
        win = ids.view(Hp // W, W, Wp // W, W).permute(0, 2, 1, 3).reshape(-1, W * W)
        diff = win[:, None, :] - win[:, :, None]
        return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))

    def _attn_bias(self, b, shifted, Hp, Wp, device):
        bias = self._p(b + 'attn.w_msa.relative_position_bias_table')           # (heads, 49, 49)
        if not shifted:
            return bias
        key = (b, Hp, Wp)
        if key not in self._bias_cache:       # (nW, heads, 49, 49): relative-position bias + region mask, summed once
            mask = self._region_mask(Hp, Wp, device)
            self._bias_cache[key] = (bias.float()[None] + mask[:, None]).to(self.dtype)
        return self._bias_cache[key]

    def _block(self, x, s, j):
        b = f'stages.{s}.blocks.{j}.'
        B, H, Wd, C = x.shape
        W, heads = self.window, self.num_heads[s]
        shift = W // 2 if j % 2 else 0
        y = self._ln(x, b + 'norm1')
        pad_r, pad_b = (W - Wd % W) % W, (W - H % W) % W
        if pad_r or pad_b:
            y = F.pad(y, (0, 0, 0, pad_r, 0, pad_b))
        Hp, Wp = H + pad_b, Wd + pad_r
        if shift:
            y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
        nW = (Hp // W) * (Wp // W)
        y = y.view(B, Hp // W, W, Wp // W, W, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, W * W, C)
        qkv = F.linear(y, self._p(b + 'attn.w_msa.qkv.weight'), self._p(b + 'attn.w_msa.qkv.bias'))
        q, k, v = qkv.view(B * nW, W * W, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
        attn = q @ k.transpose(-2, -1)                                          # the scale sits in the q weights
        bias = self._attn_bias(b, bool(shift), Hp, Wp, x.device)
        if shift:
            attn = (attn.view(B, nW, heads, W * W, W * W) + bias[None]).view(B * nW, heads, W * W, W * W)
        else:
            attn = attn + bias[None]
        y = (torch.softmax(attn, -1) @ v).transpose(1, 2).reshape(B * nW, W * W, C)
        y = F.linear(y, self._p(b + 'attn.w_msa.proj.weight'), self._p(b + 'attn.w_msa.proj.bias'))
        y = y.view(B, Hp // W, Wp // W, W, W, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        if shift:
            y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
        if pad_r or pad_b:
            y = y[:, :H, :Wd]
        x = x + y
        y = self._ln(x, b + 'norm2')
        y = F.gelu(F.linear(y, self._p(b + 'ffn.layers.0.0.weight'), self._p(b + 'ffn.layers.0.0.bias')))
        return x + F.linear(y, self._p(b + 'ffn.layers.1.weight'), self._p(b + 'ffn.layers.1.bias'))

    def _merge(self, x, s):
        d = f'stages.{s}.downsample.'
        B, H, Wd, C = x.shape
        if H % 2 or Wd % 2:
            x = F.pad(x, (0, 0, 0, Wd % 2, 0, H % 2))
            H, Wd = H + H % 2, Wd + Wd % 2
        # nn.Unfold order: channel-major, then the 2 x 2 positions (c*4 + kh*2 + kw)
        x = x.view(B, H // 2, 2, Wd // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(B, H // 2, Wd // 2, 4 * C)
        return F.linear(self._ln(x, d + 'norm'), self._p(d + 'reduction.weight'))

    @torch.no_grad()
    def stage_maps(self, img):
        """The backbone alone: the four normalised stage outputs (`SwinTransformer.forward`, swin.py:737-754) as
        channels-last (B, C, H, W) maps."""
        if not self._loaded:
            raise RuntimeError('FrozenSwinFPN: no weights loaded (load_mmdet_state)')
        x = img.to(dtype=self.dtype, memory_format=torch.channels_last)
        P = self.patch
        ph, pw = (P - x.shape[2] % P) % P, (P - x.shape[3] % P) % P
        if ph or pw:
            x = F.pad(x, (0, pw, 0, ph))
        x = F.conv2d(x, self._p('patch_embed.projection.weight'), self._p('patch_embed.projection.bias'), stride=P)
        x = self._ln(x.permute(0, 2, 3, 1), 'patch_embed.norm')                # (B, H, W, C): a view of the channels-last map
        feats = []
        for s, depth in enumerate(self.depths):
            for j in range(depth):
                x = self._block(x, s, j)
            feats.append(self._ln(x, f'norm{s}').permute(0, 3, 1, 2))           # channels-last (B, C, H, W) view
            if s < 3:
                x = self._merge(x, s)
        return feats

    @torch.no_grad()
    def forward(self, img):
        """img: (B*N, 3, H, W) -> tuple of the requested FPN levels, channels-last `dtype`, in the order of `self.levels`."""
        return self._fpn(self.stage_maps(img))


class ImageGlue(nn.Module):
    """`extract_img_feat` (:100-118): flattens the camera dimension, records `input_shape` in the metas and returns the
    FPN levels the neck reads."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    @torch.no_grad()
    def forward(self, img, img_metas):
        if img is None:
            return None
        input_shape = tuple(img.shape[-2:])
        for meta in img_metas:
            meta.update(input_shape=input_shape)
        if img.dim() == 5:              # (B, N, C, H, W) -> (B*N, C, H, W); the reference squeezes B == 1, same result
            img = img.reshape(-1, *img.shape[2:])
        return self.net(img)
