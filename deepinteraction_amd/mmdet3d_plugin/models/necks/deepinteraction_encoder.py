"""MMRI encoder (drop-in for reference `models/necks/deepinteraction_encoder.py`:
DeepInteractionEncoderLayer :8-33, DeepInteractionEncoder :35-85).

Same registry name, constructor kwargs, forward signature/returns and `state_dict` keys;
channels-last + HIP execution (see models/utils/encoder_utils.py).  Per forward the
feature-independent geometry (projection constants, sparse depth, depth completion) is built
once per sample and shared by all layers through a private key in `pts_metas`.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .... import ops
from ....registry import NECKS
from .... import utils
from ....utils import fork_join, param_key
from ..utils.encoder_utils import (GEOM_KEY, BEVWarp, conv3x3_module, ConvBNReLU, LocalContextAttentionBlock, MMRI_I2P, MMRI_P2I,
                                   fusable_projections, mix2, mix2_folded, pillar_batch_bounds,
                                   project_many, _fusable as _fusable_mods,
                                   sample_geometry)


def _fusable_mix(x, proj1, proj2):
    return (_fusable_mods(x, proj1, proj2) and proj1.conv.in_channels == 256 and proj2.conv.in_channels == 256)


class DeepInteractionEncoderLayer(nn.Module):
    def __init__(self, hidden_channel):
        super().__init__()
        self.I2P_block = MMRI_I2P(hidden_channel, hidden_channel, 0.1)
        self.P_IML = LocalContextAttentionBlock(hidden_channel, hidden_channel, 9)
        self.P_out_proj = ConvBNReLU(2 * hidden_channel, hidden_channel, kernel_size=1,
                                     norm_layer=nn.BatchNorm2d, activation_layer=None)
        self.P_integration = ConvBNReLU(2 * hidden_channel, hidden_channel, kernel_size=1,
                                        norm_layer=nn.BatchNorm2d, activation_layer=None)
        self.P2I_block = MMRI_P2I(hidden_channel, hidden_channel, 9)
        self.I_IML = LocalContextAttentionBlock(hidden_channel, hidden_channel, 9)
        self.I_out_proj = ConvBNReLU(2 * hidden_channel, hidden_channel, kernel_size=1,
                                     norm_layer=nn.BatchNorm2d, activation_layer=None)
        self.I_integration = ConvBNReLU(2 * hidden_channel, hidden_channel, kernel_size=1,
                                        norm_layer=nn.BatchNorm2d, activation_layer=None)

    # The BEV side and the image side of a layer read the same inputs and are independent (reference :21-33).  The BEV
    # side is a chain of SMALL launches (127-276 workgroups on a 180 x 180 map: half of the 256 CUs idle), so at
    # inference it runs on a second HIP stream, forked from and joined with the caller's stream - under hipGraph capture
    # the two sides become parallel branches of the graph.

    def _bev_side(self, lidar_feat, img5, img_metas, pts_metas):
        P, I2P = self.P_IML, self.I2P_block
        if (not torch.is_grad_enabled() and not I2P.training and I2P.pts_channels == 128 and I2P.img_channels == 128
                and fusable_projections(lidar_feat, P.query_project, P.key_project, P.value_project)):
            # fp16 inference: the three projections of P_IML and the folded query of the pillar attention are ONE launch
            # over the BEV map; the attention's output projection is folded into P_out_proj (no GEMM, no mask pass)
            q, k, v, qfold = project_many([P.query_project, P.key_project, P.value_project,
                                           I2P.query_chain(lidar_feat.dtype)], lidar_feat)
            ctx, valid = I2P.attend(qfold, img5, img_metas, pts_metas)
            P2P_feat = ops.local_attention(q, k, v, P.kernel_size, P.kernel_size, 1.0 / math.sqrt(k.size(1)))
            if _fusable_mix(ctx, self.P_out_proj, self.P_integration):
                return mix2_folded(self.P_out_proj, ctx, I2P.folded(torch.float32)[2:], valid, P2P_feat, self.P_integration,
                                   lidar_feat, self.__dict__.setdefault('_mix_cache', {}))
            _, _, w_ov, b_ov = I2P.folded(ctx.dtype)
            o = F.linear(ctx.permute(0, 2, 3, 1), w_ov, b_ov).permute(0, 3, 1, 2) * valid
            return mix2(self.P_out_proj, o, P2P_feat, self.P_integration, lidar_feat)
        I2P_feat = self.I2P_block(lidar_feat, img5, img_metas, pts_metas)
        P2P_feat = self.P_IML(lidar_feat, lidar_feat)
        # P_integration(cat(P_out_proj(cat(I2P, P2P)), lidar)) (:26-27): one fused kernel at inference
        return mix2(self.P_out_proj, I2P_feat, P2P_feat, self.P_integration, lidar_feat)

    # ONE window-attention launch for BOTH image-side attentions of the layer (round 6): I_IML (keys / values = the image map)
    # and P2I (keys / values = the warped BEV map) have the same query pixels, window and scale, so their maps are laid out as
    # the two halves of three PAIR BUFFERS (2 BN images each: [I_IML | P2I] queries, keys, values) - written in place by the
    # image map's projection launch and by the warp-project launch - and the attention kernel, which works image by image,
    # walks 2 BN images: 2 x 1 092 tiles = 8.5 rounds of the 256 workgroups instead of 2 x 4.27 (two ragged last rounds, two
    # prologues), no kernel change.  DI_PAIR_ATTN=0 restores the two launches.
    PAIR = __import__('os').environ.get('DI_PAIR_ATTN', '1') != '0'

    def _pair_buffers(self, img_feat, lidar_feat):
        I, PL = self.I_IML, self.P2I_block.Local
        if not (self.PAIR and not torch.is_grad_enabled() and self.P2I_block.warp_kv_fusable(lidar_feat)
                and I.kernel_size == PL.kernel_size
                and fusable_projections(img_feat, I.query_project, I.key_project, I.value_project, PL.query_project)):
            return None
        BN, _, H, W = img_feat.shape
        return tuple(ops.empty_cl(2 * BN, 128, H, W, img_feat) for _ in range(3))

    def _image_side(self, img_feat, img5, lidar_feat, img_metas, pts_metas, warped=None, warped_ready=None, kv=None, pair=None):
        BN, I_C, I_H, I_W = img_feat.shape
        # fp16 inference: the four projections of the image map (query / key / value of I_IML and the query of P2I)
        # are ONE launch that reads it once.
        I, PL = self.I_IML, self.P2I_block.Local
        if pair is None and kv is None and warped is None:
            pair = self._pair_buffers(img_feat, lidar_feat)
        if pair is not None:
            q2, k2, v2 = pair
            project_many([I.query_project, I.key_project, I.value_project, PL.query_project], img_feat,
                         outs=[q2[:BN], k2[:BN], v2[:BN], q2[BN:]])
            if kv is None:
                self.P2I_block.warp_kv(lidar_feat, img5, img_metas, pts_metas, out=(k2[BN:], v2[BN:]))
            if warped_ready is not None:
                torch.cuda.current_stream().wait_event(warped_ready)
            both = ops.local_attention(q2, k2, v2, I.kernel_size, I.kernel_size, 1.0 / math.sqrt(128))
            return mix2(self.I_out_proj, both[BN:], both[:BN], self.I_integration, img_feat)
        if fusable_projections(img_feat, I.query_project, I.key_project, I.value_project, PL.query_project):
            q_i, k_i, v_i, q_p = project_many([I.query_project, I.key_project, I.value_project, PL.query_project], img_feat)
            if warped_ready is not None:
                torch.cuda.current_stream().wait_event(warped_ready)
            P2I_feat = self.P2I_block(lidar_feat, img5, img_metas, pts_metas, query=q_p, warped=warped, kv=kv)
            I2I_feat = ops.local_attention(q_i, k_i, v_i, I.kernel_size, I.kernel_size, 1.0 / math.sqrt(k_i.size(1)))
        else:
            if warped_ready is not None:
                torch.cuda.current_stream().wait_event(warped_ready)
            P2I_feat = self.P2I_block(lidar_feat, img5, img_metas, pts_metas, warped=warped)
            I2I_feat = self.I_IML(img_feat, img_feat)
        return mix2(self.I_out_proj, P2I_feat.view(BN, -1, I_H, I_W), I2I_feat, self.I_integration, img_feat)

    def forward(self, img_feat, lidar_feat, img_metas, pts_metas):
        batch_size = lidar_feat.shape[0]
        BN, I_C, I_H, I_W = img_feat.shape
        img5 = img_feat.view(batch_size, -1, I_C, I_H, I_W)
        if utils.OVERLAP & 1 and lidar_feat.is_cuda and not torch.is_grad_enabled():
            # fork: everything already queued on the caller's stream (the layer inputs, the per-sample geometry)
            # precedes the BEV side; join: the caller's stream continues after both sides.  Tensors crossing the fork
            # or the join stay referenced until after the join, so neither stream's allocator pool can recycle them
            # under a kernel of the other stream.
            if not utils.OVERLAP & 16:
                return tuple(fork_join(lidar_feat.device,
                                       lambda: self._image_side(img_feat, img5, lidar_feat, img_metas, pts_metas),
                                       lambda: self._bev_side(lidar_feat, img5, img_metas, pts_metas)))
            # the BEV -> image warp needs the BEV map and the depth only: it opens the side stream's work and is awaited
            # by the image side after its own projections.  `warped` stays referenced until after the join (allocator
            # contract of fork_join).
            main, side = torch.cuda.current_stream(lidar_feat.device), utils.side_stream(lidar_feat.device, 0)
            pair = self._pair_buffers(img_feat, lidar_feat)       # allocated in front of the fork: both streams write into them
            side.wait_stream(main)
            with torch.cuda.stream(side):
                # fp16 inference: the warp is gathered inside the key / value projection launch (no warped map in memory)
                warped = kv = None
                if pair is not None:
                    kv = self.P2I_block.warp_kv(lidar_feat, img5, img_metas, pts_metas, out=(pair[1][BN:], pair[2][BN:]))
                elif self.P2I_block.warp_kv_fusable(lidar_feat) and fusable_projections(
                        img_feat, self.I_IML.query_project, self.I_IML.key_project, self.I_IML.value_project,
                        self.P2I_block.Local.query_project):
                    kv = self.P2I_block.warp_kv(lidar_feat, img5, img_metas, pts_metas)
                else:
                    warped = self.P2I_block.Warp(lidar_feat, img5, img_metas, pts_metas)
                ready = torch.cuda.Event()
                ready.record(side)
                new_lidar_feat = self._bev_side(lidar_feat, img5, img_metas, pts_metas)
            new_img_feat = self._image_side(img_feat, img5, lidar_feat, img_metas, pts_metas, warped, ready, kv, pair)
            main.wait_stream(side)
            del warped, kv, pair
            return new_img_feat, new_lidar_feat
        new_lidar_feat = self._bev_side(lidar_feat, img5, img_metas, pts_metas)
        new_img_feat = self._image_side(img_feat, img5, lidar_feat, img_metas, pts_metas)
        return new_img_feat, new_lidar_feat


@NECKS.register_module()
class DeepInteractionEncoder(nn.Module):
    def __init__(self, num_layers=2, in_channels_img=64, in_channels_pts=128 * 3, hidden_channel=128,
                 bn_momentum=0.1, bias='auto'):
        super().__init__()
        # mmcv build_conv_layer(dict(type='Conv2d'), ..., bias=bias): 'auto' is a truthy string
        # -> both shared convs carry a bias (reference :45-62)
        self.shared_conv_pts = nn.Conv2d(in_channels_pts, hidden_channel, kernel_size=3, padding=1,
                                         bias=bool(bias))
        self.shared_conv_img = nn.Conv2d(in_channels_img, hidden_channel, kernel_size=3, padding=1,
                                         bias=bool(bias))
        self.num_layers = num_layers
        self.fusion_blocks = nn.ModuleList(
            [DeepInteractionEncoderLayer(hidden_channel) for _ in range(num_layers)])
        self.bn_momentum = bn_momentum
        self.init_weights()

    def init_weights(self):
        self.init_bn_momentum()

    def init_bn_momentum(self):
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = self.bn_momentum

    def _shared_conv(self, conv, x):
        return conv3x3_module(self, conv, x)

    def forward(self, img_feats, pts_feats, img_metas, pts_metas):
        own_geom = GEOM_KEY not in pts_metas
        own_bounds = 'pillar_batch_bounds' not in pts_metas
        dev = img_feats.device
        I_H, I_W = img_feats.shape[-2:]
        if own_geom:
            pts_metas[GEOM_KEY] = [None] * len(img_metas)
        geoms = [sample_geometry(img_metas, pts_metas, b, (I_H, I_W), dev) for b in range(len(img_metas))]

        def depth_only():
            for b, g in enumerate(geoms):
                BEVWarp.dense_depth(g, pts_metas['pts'][b], I_H, I_W)

        def keys_only():
            bounds = pillar_batch_bounds(pts_metas, len(img_metas))
            for b, g in enumerate(geoms):
                MMRI_I2P.pillar_keys(g, pts_metas, bounds[b], bounds[b + 1], (I_H, I_W), tuple(pts_feats.shape[-2:]))

        def depth_maps():
            # what depends on the points and the metas only: sparse depth + completion (~15 tiny launches per sample)
            # and the key table of the pillar attention
            depth_only()
            keys_only()

        def pts_conv():
            y = self._shared_conv(self.shared_conv_pts, pts_feats)
            return y, y.clone()

        if img_feats.is_cuda and not torch.is_grad_enabled() and utils.OVERLAP & 6:
            # the image conv fills the chip for 4.1 waves of tiles, the BEV conv for 1.1 and the depth chain never does:
            # concurrent branches (fork/join, parallel paths under capture)
            if utils.OVERLAP & 2:
                outs = fork_join(dev, lambda: self._shared_conv(self.shared_conv_img, img_feats), pts_conv,
                                 depth_maps if utils.OVERLAP & 4 else None)
                new_img_feat, (new_pts_feat, pts_feat_conv) = outs[0], outs[1]
            else:
                def convs():
                    return self._shared_conv(self.shared_conv_img, img_feats), pts_conv()
                if utils.OVERLAP & 64:
                    # round 6: the key table (projection + the dense stream of the matrix-core pillar attention: 5 launches,
                    # ~36 us) on a branch of its own beside the depth chain (~90 us) - together they had become longer than
                    # the two convolutions they hide under
                    (new_img_feat, (new_pts_feat, pts_feat_conv)) = fork_join(dev, convs, depth_only, keys_only)[0]
                else:
                    (new_img_feat, (new_pts_feat, pts_feat_conv)), _ = fork_join(dev, convs, depth_maps)
        else:
            new_img_feat = self._shared_conv(self.shared_conv_img, img_feats)
            new_pts_feat, pts_feat_conv = pts_conv()
        try:
            for i in range(self.num_layers):
                new_img_feat, new_pts_feat = self.fusion_blocks[i](new_img_feat, new_pts_feat, img_metas,
                                                                   pts_metas)
        finally:
            if own_geom:
                pts_metas.pop(GEOM_KEY, None)
                if own_bounds:
                    pts_metas.pop('pillar_batch_bounds', None)
        return new_img_feat, [pts_feat_conv, new_pts_feat]
