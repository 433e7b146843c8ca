"""Oracle restatement of the MMPI decoder forward (TEST INFRASTRUCTURE, PyTorch fp32 CPU).

Follows (paths under /root/reference/projects/mmdet3d_plugin/):
  models/utils/decoder_utils.py   PositionEmbeddingLearned :16-32, TransformerDecoderLayer :35-113,
                                  MultiheadAttention / multi_head_attention_forward :116-495,
                                  FFN :498-581, DynamicConv :584-629, ImageRCNNBlock :632-761,
                                  PointRCNNBlock :765-841
  models/dense_heads/deepinteraction_decoder.py  DeepInteractionDecoder.__init__ :21-159,
                                  create_2D_grid :162-169, forward :201-313
  core/bbox/coders/transfusion_bbox_coder.py     decode :39-91
Third-party pieces (mmcv ConvModule, detectron2 ROIAlignV2, mmdet3d corners /
apply_3d_transformation) come from oracle/thirdparty.py (parity unpinned there).
Module / parameter names equal the reference's (`state_dict` compatible).
The per-sample / per-view Python loops of the reference are kept on purpose: this is also
the CPU baseline.
"""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F

from .thirdparty import apply_3d_transformation, lidar_box_corners, roi_align_v2


class ConvModule(nn.Module):
    """mmcv ConvModule(conv -> BN -> ReLU), bias='auto' => no conv bias with a norm."""

    def __init__(self, cin, cout, k, dim):
        super().__init__()
        conv, bn = (nn.Conv1d, nn.BatchNorm1d) if dim == 1 else (nn.Conv2d, nn.BatchNorm2d)
        self.conv = conv(cin, cout, k, padding=k // 2, bias=False)
        self.bn = bn(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class PositionEmbeddingLearned(nn.Module):
    """decoder_utils.py:16-32."""

    def __init__(self, input_channel, num_pos_feats):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            nn.Conv1d(input_channel, num_pos_feats, 1), nn.BatchNorm1d(num_pos_feats), nn.ReLU(inplace=True),
            nn.Conv1d(num_pos_feats, num_pos_feats, 1))

    def forward(self, xyz):
        return self.position_embedding_head(xyz.transpose(1, 2).contiguous())


class MultiheadAttention(nn.Module):
    """decoder_utils.py:116-495 restricted to what the decoder uses: packed in-proj, no masks,
    eval-mode dropout.  Inputs (L,N,E)/(S,N,E); returns (L,N,E)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, query, key, value):
        L, N, E = query.shape
        S = key.shape[0]
        H, D = self.num_heads, E // self.num_heads
        w, b = self.in_proj_weight, self.in_proj_bias
        q = F.linear(query, w[:E], b[:E]) * (float(D) ** -0.5)                    # :421
        k = F.linear(key, w[E:2 * E], b[E:2 * E])
        v = F.linear(value, w[2 * E:], b[2 * E:])
        q = q.contiguous().view(L, N * H, D).transpose(0, 1)
        k = k.contiguous().view(S, N * H, D).transpose(0, 1)
        v = v.contiguous().view(S, N * H, D).transpose(0, 1)
        a = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)                # :471-483
        o = torch.bmm(a, v).transpose(0, 1).contiguous().view(L, N, E)
        return F.linear(o, self.out_proj.weight, self.out_proj.bias)


class TransformerDecoderLayer(nn.Module):
    """decoder_utils.py:35-113 (post-norm; pos-embeds added to q, k AND v)."""

    def __init__(self, d_model, nhead, dim_feedforward, dropout, activation, self_posembed, cross_posembed):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.activation = {'relu': F.relu, 'gelu': F.gelu}[activation]
        self.self_posembed, self.cross_posembed = self_posembed, cross_posembed

    def forward(self, query, key, query_pos, key_pos):
        qpe = self.self_posembed(query_pos).permute(2, 0, 1)
        kpe = self.cross_posembed(key_pos).permute(2, 0, 1)
        query, key = query.permute(2, 0, 1), key.permute(2, 0, 1)
        q = query + qpe
        query = self.norm1(query + self.self_attn(q, q, q))                        # :96-99
        kk = key + kpe
        query = self.norm2(query + self.multihead_attn(query + qpe, kk, kk))      # :101-105
        query = self.norm3(query + self.linear2(self.activation(self.linear1(query))))
        return query.permute(1, 2, 0)


class FFN(nn.Module):
    """decoder_utils.py:498-581: per head ConvModule(in->64)+Conv1d(64->classes, bias)."""

    def __init__(self, in_channels, heads, head_conv=64):
        super().__init__()
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            layers, c = [], in_channels
            for _ in range(num_conv - 1):
                layers.append(ConvModule(c, head_conv, 1, dim=1))
                c = head_conv
            layers.append(nn.Conv1d(head_conv, classes, 1, bias=True))
            setattr(self, head, nn.Sequential(*layers))

    def forward(self, x):
        return {h: getattr(self, h)(x) for h in self.heads}


class DynamicConv(nn.Module):
    """decoder_utils.py:584-629."""

    def __init__(self):
        super().__init__()
        self.hidden_dim = self.dim_dynamic = 128
        self.num_params = 128 * 128
        self.dynamic_layer = nn.Linear(128, 2 * self.num_params)
        self.norm1, self.norm2 = nn.LayerNorm(128), nn.LayerNorm(128)
        self.out_layer = nn.Linear(128 * 49, 128)
        self.norm3 = nn.LayerNorm(128)

    def forward(self, pro_features, roi_features):
        f = roi_features.permute(1, 0, 2)                                  # (q,49,128)
        params = self.dynamic_layer(pro_features).permute(1, 0, 2)         # (q,1,32768)
        p1 = params[:, :, :self.num_params].view(-1, 128, 128)
        p2 = params[:, :, self.num_params:].view(-1, 128, 128)
        f = F.relu(self.norm1(torch.bmm(f, p1)))
        f = F.relu(self.norm2(torch.bmm(f, p2)))
        return F.relu(self.norm3(self.out_layer(f.flatten(1))))


class TransFusionBBoxCoder:
    """core/bbox/coders/transfusion_bbox_coder.py:9-91 (decode without filtering)."""

    def __init__(self, pc_range, out_size_factor, voxel_size, post_center_range=None, score_threshold=None,
                 code_size=8):
        self.pc_range, self.out_size_factor, self.voxel_size = pc_range, out_size_factor, voxel_size
        self.post_center_range, self.score_threshold, self.code_size = post_center_range, score_threshold, code_size

    def encode(self, dst_boxes):
        t = torch.zeros(dst_boxes.shape[0], self.code_size)
        t[:, 0] = (dst_boxes[:, 0] - self.pc_range[0]) / (self.out_size_factor * self.voxel_size[0])
        t[:, 1] = (dst_boxes[:, 1] - self.pc_range[1]) / (self.out_size_factor * self.voxel_size[1])
        t[:, 3:6] = dst_boxes[:, 3:6].log()
        t[:, 2] = dst_boxes[:, 2] + dst_boxes[:, 5] * 0.5
        t[:, 6], t[:, 7] = torch.sin(dst_boxes[:, 6]), torch.cos(dst_boxes[:, 6])
        if self.code_size == 10:
            t[:, 8:10] = dst_boxes[:, 7:]
        return t

    def decode(self, heatmap, rot, dim, center, height, vel):
        """Works on copies (the reference mutates deep copies, decoder_utils.py:672-679)."""
        labels = heatmap.max(1).indices
        scores = heatmap.max(1).values
        center = center.clone()
        center[:, 0] = center[:, 0] * self.out_size_factor * self.voxel_size[0] + self.pc_range[0]
        center[:, 1] = center[:, 1] * self.out_size_factor * self.voxel_size[1] + self.pc_range[1]
        dim = dim.exp()
        height = height - dim[:, 2:3] * 0.5
        rot = torch.atan2(rot[:, 0:1], rot[:, 1:2])
        parts = [center, height, dim, rot] + ([] if vel is None else [vel])
        boxes = torch.cat(parts, 1).permute(0, 2, 1)
        return [dict(bboxes=boxes[i], scores=scores[i], labels=labels[i]) for i in range(heatmap.shape[0])]


class _RCNNBase(nn.Module):
    def refine(self, q, roi, sfx=''):
        """decoder_utils.py:743-756 / :824-837: self-attn + LN, DynamicConv + LN, FFN(GELU) + LN.
        q (n,1,128) sequence-first, roi (49,n,128)."""
        g = lambda n: getattr(self, n + sfx)
        q = g('norm1')(q + g('dyconv_pre_self_attn')(q, q, q)[0])
        q = q.permute(1, 0, 2)
        q = g('norm2')(q + g('dyconv')(q, roi))
        q = g('norm3')(q + g('linear2')(F.gelu(g('linear1')(q))))
        return q


class ImageRCNNBlock(_RCNNBase):
    """decoder_utils.py:632-761."""

    def __init__(self, num_views, num_proposals, out_size_factor_img, test_cfg, bbox_coder, hidden, heads, dropout):
        super().__init__()
        self.num_views, self.num_proposals = num_views, num_proposals
        self.out_size_factor_img, self.test_cfg, self.bbox_coder = out_size_factor_img, test_cfg, bbox_coder
        self.dyconv = DynamicConv()
        self.dyconv_pre_self_attn = nn.MultiheadAttention(hidden, heads, dropout=dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(hidden), nn.LayerNorm(hidden), nn.LayerNorm(hidden)
        self.linear1, self.linear2 = nn.Linear(hidden, hidden * 4), nn.Linear(hidden * 4, hidden)

    def forward(self, query_feat, res_layer, new_lidar_feat, img_feat_flatten, img_metas, img_h, img_w):
        B, Q = query_feat.shape[0], self.num_proposals
        prev = query_feat
        out = torch.zeros_like(query_feat)                                                   # :665
        cfg = self.test_cfg
        # the geometry is built from DETACHED copies of the predictions (:662, :667-679): no gradient
        # reaches the heads through RoI coordinates
        res_layer = {k: v.detach() for k, v in res_layer.items()}
        pos = res_layer['center'] * cfg['out_size_factor'] * cfg['voxel_size'][0] + cfg['pc_range'][0]
        pos3d = torch.cat([pos, res_layer['height']], 1)                                     # (B,3,Q) :666-667
        boxes = self.bbox_coder.decode(res_layer['heatmap'], res_layer['rot'], res_layer['dim'],
                                       res_layer['center'], res_layer['height'], res_layer.get('vel'))
        on_mask = torch.ones(B, Q) * -1
        for s in range(B):
            meta = img_metas[s]
            l2i = pos3d.new_tensor(meta['lidar2img'])
            flip = meta.get('flip', False)
            crop = pos3d.new_tensor(meta['img_crop_offset']) if 'img_crop_offset' in meta else 0
            h, w = meta['input_shape'][:2]
            corners = lidar_box_corners(boxes[s]['bboxes'][:, :7])                           # (Q,8,3)
            pts = torch.cat([pos3d[s], corners.permute(2, 0, 1).reshape(3, -1)], -1).T       # (Q+8Q,3) :691
            pts = apply_3d_transformation(pts, 'LIDAR', meta, reverse=True)
            for v in range(self.num_views):
                p4 = torch.cat([pts, pts.new_ones(pts.shape[0], 1)], -1)
                p2 = p4 @ l2i[v].t()
                z = torch.clamp(p2[:, 2], min=1e-5)                                          # :699
                cx, cy = p2[:, 0] / z - (crop[0] if torch.is_tensor(crop) else 0), \
                    p2[:, 1] / z - (crop[1] if torch.is_tensor(crop) else 0)
                if flip:
                    cx = meta['img_shape'][0][1] - cx                                        # :710-714
                ctr_x, ctr_y = cx[:Q], cy[:Q]
                cor_x, cor_y = cx[Q:].reshape(Q, 8), cy[Q:].reshape(Q, 8)
                on = (ctr_x > 0) & (ctr_x < w) & (ctr_y > 0) & (ctr_y < h)                    # :722-723
                if on.sum() <= 1:                                                            # :726
                    continue
                on_mask[s, on] = v
                rect = torch.stack([cor_x[on].min(1).values, cor_y[on].min(1).values,
                                    cor_x[on].max(1).values, cor_y[on].max(1).values], 1)   # :730-738
                feat = img_feat_flatten[s:s + 1, v].reshape(1, -1, img_h, img_w)
                roi = roi_align_v2(feat, rect, 7, 1.0 / self.out_size_factor_img, 2)         # :739-741
                qv = prev[s, :, on][None].permute(2, 0, 1)                                   # (n,1,128)
                roi = roi.flatten(2).permute(2, 0, 1)                                        # (49,n,128)
                qv = self.refine(qv, roi)
                out[s, :, on] = qv[0].permute(1, 0)                                          # later view overwrites :759
        return out, on_mask


class PointRCNNBlock(_RCNNBase):
    """decoder_utils.py:765-841."""

    def __init__(self, hidden, heads, dropout, bbox_coder):
        super().__init__()
        self.bbox_coder = bbox_coder
        self.dyconv_pts = DynamicConv()
        self.dyconv_pre_self_attn_pts = nn.MultiheadAttention(hidden, heads, dropout=dropout)
        self.norm1_pts, self.norm2_pts, self.norm3_pts = nn.LayerNorm(hidden), nn.LayerNorm(hidden), nn.LayerNorm(hidden)
        self.linear1_pts, self.linear2_pts = nn.Linear(hidden, hidden * 4), nn.Linear(hidden * 4, hidden)

    def forward(self, query_feat, res_layer, new_lidar_feat, img_feat_flatten, img_metas, img_h, img_w):
        B = query_feat.shape[0]
        out = torch.zeros_like(query_feat)
        res_layer = {k: v.detach() for k, v in res_layer.items()}                            # :792-803 deep copies of .detach()
        boxes = self.bbox_coder.decode(res_layer['heatmap'], res_layer['rot'], res_layer['dim'],
                                       res_layer['center'], res_layer['height'], res_layer.get('vel'))
        bc = self.bbox_coder
        for s in range(B):
            box = boxes[s]['bboxes'][:, :7].clone()
            box[:, 3:6] *= 2                                                                 # :807
            cor = lidar_box_corners(box)
            cc = (cor[..., :2] - bc.pc_range[0]) / (bc.voxel_size[0] * bc.out_size_factor)   # :810
            rect = torch.stack([cc[..., 0].min(-1).values, cc[..., 1].min(-1).values,
                                cc[..., 0].max(-1).values, cc[..., 1].max(-1).values], -1)
            roi = roi_align_v2(new_lidar_feat[s:s + 1], rect, 7, 1.0, 2).flatten(2).permute(2, 0, 1)
            qv = query_feat[s:s + 1].permute(2, 0, 1)
            qv = self.refine(qv, roi, '_pts')
            out[s] = qv.permute(0, 2, 1)[0]
        return out, None


class DeepInteractionDecoder(nn.Module):
    """dense_heads/deepinteraction_decoder.py:19-313 (forward only; initialize_by_heatmap=True)."""

    def __init__(self, num_views=6, out_size_factor_img=4, num_proposals=200, auxiliary=True, hidden_channel=128,
                 num_classes=10, num_mmpi=4, num_decoder_layers=1, num_heads=8, nms_kernel_size=3,
                 ffn_channel=256, dropout=0.1, bn_momentum=0.1, activation='relu', common_heads=None,
                 num_heatmap_convs=2, bbox_coder=None, test_cfg=None, **unused):
        super().__init__()
        self.num_classes, self.num_proposals, self.auxiliary = num_classes, num_proposals, auxiliary
        self.nms_kernel_size, self.test_cfg, self.num_views = nms_kernel_size, test_cfg, num_views
        self.num_decoder_layers, self.num_mmpi = num_decoder_layers, num_mmpi
        bc = dict(bbox_coder)
        bc.pop('type', None)
        self.bbox_coder = TransFusionBBoxCoder(**bc)
        h = hidden_channel
        self.heatmap_head = nn.Sequential(ConvModule(h, h, 3, dim=2), nn.Conv2d(h, num_classes, 3, padding=1))
        self.heatmap_head_img = copy.deepcopy(self.heatmap_head)
        self.class_encoding = nn.Conv1d(num_classes, h, 1)
        self.decoder = nn.ModuleList([TransformerDecoderLayer(
            h, num_heads, ffn_channel, dropout, activation, PositionEmbeddingLearned(2, h),
            PositionEmbeddingLearned(2, h)) for _ in range(num_decoder_layers)])
        heads = dict(copy.deepcopy(common_heads))
        heads.update(heatmap=(num_classes, num_heatmap_convs))
        self.prediction_heads = nn.ModuleList([FFN(h, heads) for _ in range(num_decoder_layers)])
        self.decode_head, self.pred_head = nn.ModuleList(), nn.ModuleList()
        for _ in range(num_mmpi // 2):
            self.decode_head.append(ImageRCNNBlock(num_views, num_proposals, out_size_factor_img, test_cfg,
                                                   self.bbox_coder, h, num_heads, dropout))
            self.pred_head.append(FFN(h * 2, heads))
            self.decode_head.append(PointRCNNBlock(h, num_heads, dropout, self.bbox_coder))
            self.pred_head.append(FFN(h * 2, heads))
        xs = test_cfg['grid_size'][0] // test_cfg['out_size_factor']
        ys = test_cfg['grid_size'][1] // test_cfg['out_size_factor']
        self.bev_pos = self.create_2D_grid(xs, ys)
        for p in self.decoder.parameters():                                                  # :173-175
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = bn_momentum

    @staticmethod
    def create_2D_grid(x_size, y_size):
        """:162-169: cell centres (col+0.5, row+0.5), row-major over (x_size, y_size)."""
        by, bx = torch.meshgrid(torch.linspace(0, x_size - 1, x_size), torch.linspace(0, y_size - 1, y_size),
                                indexing='ij')
        base = torch.cat([(bx + 0.5)[None], (by + 0.5)[None]], 0)[None]
        return base.view(1, 2, -1).permute(0, 2, 1)

    def mmpi(self, query_feat, res, first, new_lidar_feat, img_flat, img_metas, I_H, I_W):
        """:279-297: the alternating image / point RoI refinement."""
        self.on_the_image_mask, rets = [], []
        for l in range(self.num_mmpi):
            prev = query_feat.clone()
            query_pos = res['center'].detach().permute(0, 2, 1)                              # :281 .detach().clone()
            query_feat, on = self.decode_head[l](prev, res, new_lidar_feat, img_flat, img_metas, I_H, I_W)
            res = self.pred_head[l](torch.cat([query_feat, prev], 1))                          # :289
            res['center'] = res['center'] + query_pos.permute(0, 2, 1)
            if l % 2 == 0:
                m = on != -1
                self.on_the_image_mask.append(m)
                for k in res:                                                                  # :293-295
                    res[k] = torch.where(m.unsqueeze(1), res[k], first[k])
            rets.append(res)
        return rets

    def forward(self, pts_inputs, img_inputs, img_metas, top_override=None):
        """`top_override` (B,Q) flattened (class*HW + cell) picks replaces the arg-sort (tests of
        the continuous part with an fp16 product whose near-tied proposals may reorder)."""
        lidar_feat, new_lidar_feat = pts_inputs[0], pts_inputs[1]
        B, C = lidar_feat.shape[:2]
        flat = lidar_feat.view(B, C, -1)
        bev_pos = self.bev_pos.repeat(B, 1, 1)
        BN, I_C, I_H, I_W = img_inputs.shape
        dense = self.heatmap_head(lidar_feat)
        dense_img = self.heatmap_head_img(new_lidar_feat)
        heat = (dense.detach().sigmoid() + dense_img.detach().sigmoid()) / 2                  # :225 (detached there too)
        pad = self.nms_kernel_size // 2
        local_max = torch.zeros_like(heat)
        local_max[:, :, pad:-pad, pad:-pad] = F.max_pool2d(heat, self.nms_kernel_size, 1, 0)   # :229-230
        ones = {'nuScenes': (8, 9), 'Waymo': (1, 2)}[self.test_cfg['dataset']]
        for c in ones:                                                                        # :232-237
            local_max[:, c] = heat[:, c]
        # (test infrastructure: how far every cell is from being the maximum of its 3x3 neighbourhood - 0 for the survivors of
        #  the NMS; the tie-aware comparison of `query_heatmap_score` reads it, oracle/parity.py::heatmap_score_ties)
        self.nms_margin = (local_max - heat).view(B, heat.shape[1], -1)
        self.nms_heat = heat.view(B, heat.shape[1], -1).clone()
        heat = heat * (heat == local_max)
        heat = heat.view(B, heat.shape[1], -1)
        top = heat.view(B, -1).argsort(dim=-1, descending=True)[..., :self.num_proposals]     # :242
        if top_override is not None:
            top = top_override
        self.top_proposals = top                 # flattened (class, cell) picks, for the parity reports
        top_class = top // heat.shape[-1]
        top_index = top % heat.shape[-1]
        query_feat = flat.gather(-1, top_index[:, None, :].expand(-1, C, -1))
        self.query_labels = top_class
        one_hot = F.one_hot(top_class, self.num_classes).permute(0, 2, 1)
        query_feat = query_feat + self.class_encoding(one_hot.float())                         # :249-251
        query_pos = bev_pos.gather(1, top_index[:, :, None].expand(-1, -1, 2))
        for i in range(self.num_decoder_layers):
            query_feat = self.decoder[i](query_feat, flat, query_pos, bev_pos)
            res = self.prediction_heads[i](query_feat)
            res['center'] = res['center'] + query_pos.permute(0, 2, 1)
            first = res
            query_pos = res['center'].detach().permute(0, 2, 1)                              # :268 .detach().clone()
        img_flat = img_inputs.view(B, self.num_views, I_C, -1)
        rets = self.mmpi(query_feat, res, first, new_lidar_feat, img_flat, img_metas, I_H, I_W)
        rets[0]['query_heatmap_score'] = heat.gather(-1, top_index[:, None, :].expand(-1, self.num_classes, -1))
        rets[0]['dense_heatmap'] = dense_img
        if not self.auxiliary:
            return [rets[-1]]
        out = {}
        for k in rets[0]:
            out[k] = rets[0][k] if k in ('dense_heatmap', 'dense_heatmap_old', 'query_heatmap_score') \
                else torch.cat([r[k] for r in rets], -1)
        return [[out]]
