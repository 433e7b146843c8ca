"""MMPI decoder head (drop-in for reference `models/dense_heads/deepinteraction_decoder.py`,
`DeepInteractionDecoder` :19-313: same registry name, constructor kwargs, forward signature /
return structure, side attributes `query_labels` / `on_the_image_mask`, `state_dict` keys).

Forward only in this round (loss / get_targets / get_bboxes are SURVEY 8(f) "next" rows).
Shape-static, sync-free execution: fused heat-map NMS kernel, library top-k, batched RoI blocks
(see models/utils/decoder_utils.py), prediction heads in float32.
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn

from .... import ops
from ....registry import HEADS, build_bbox_coder
from ..utils.decoder_utils import (FFN, ConvModule, ImageRCNNBlock, PointRCNNBlock, PositionEmbeddingLearned,
                                   QueryGeometry, TransformerDecoderLayer, build_conv_layer)


@HEADS.register_module()
class DeepInteractionDecoder(nn.Module):
    def __init__(self, num_views=0, out_size_factor_img=4, num_proposals=128, auxiliary=True, hidden_channel=128,
                 num_classes=4, num_mmpi=4, num_decoder_layers=1, num_heads=8, learnable_query_pos=False,
                 initialize_by_heatmap=False, nms_kernel_size=1, ffn_channel=256, dropout=0.1, bn_momentum=0.1,
                 activation='relu', common_heads=dict(), num_heatmap_convs=2, conv_cfg=dict(type='Conv1d'),
                 norm_cfg=dict(type='BN1d'), bias='auto', loss_cls=None, loss_bbox=None, loss_heatmap=None,
                 train_cfg=None, test_cfg=None, bbox_coder=None, ret_idx=None):
        super().__init__()
        self.num_classes = num_classes
        self.num_proposals = num_proposals
        self.auxiliary = auxiliary
        self.num_heads = num_heads
        self.num_decoder_layers = num_decoder_layers
        self.bn_momentum = bn_momentum
        self.learnable_query_pos = learnable_query_pos
        self.initialize_by_heatmap = initialize_by_heatmap
        self.nms_kernel_size = nms_kernel_size
        if not initialize_by_heatmap:
            raise NotImplementedError('only the heat-map initialised head of the reference configs is built')
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.use_sigmoid_cls = (loss_cls or {}).get('use_sigmoid', False)
        if loss_cls is not None and not self.use_sigmoid_cls:
            self.num_classes += 1
        self.loss_cfg = dict(loss_cls=loss_cls, loss_bbox=loss_bbox, loss_heatmap=loss_heatmap)
        self.bbox_coder = build_bbox_coder(bbox_coder)

        self.heatmap_head = nn.Sequential(
            ConvModule(hidden_channel, hidden_channel, kernel_size=3, padding=1, bias=bias,
                       conv_cfg=dict(type='Conv2d'), norm_cfg=dict(type='BN2d')),
            build_conv_layer(dict(type='Conv2d'), hidden_channel, num_classes, kernel_size=3, padding=1,
                             bias=bool(bias)))
        self.heatmap_head_img = copy.deepcopy(self.heatmap_head)
        self.class_encoding = nn.Conv1d(num_classes, hidden_channel, 1)

        self.decoder = nn.ModuleList([
            TransformerDecoderLayer(hidden_channel, num_heads, ffn_channel, dropout, activation,
                                    self_posembed=PositionEmbeddingLearned(2, hidden_channel),
                                    cross_posembed=PositionEmbeddingLearned(2, hidden_channel))
            for _ in range(num_decoder_layers)])
        self.prediction_heads = nn.ModuleList()
        for _ in range(num_decoder_layers):
            heads = copy.deepcopy(common_heads)
            heads.update(dict(heatmap=(self.num_classes, num_heatmap_convs)))
            self.prediction_heads.append(FFN(hidden_channel, heads, conv_cfg=conv_cfg, norm_cfg=norm_cfg, bias=bias))

        self.decode_head, self.pred_head = nn.ModuleList(), nn.ModuleList()
        self.num_mmpi = num_mmpi
        assert num_mmpi % 2 == 0
        self.num_views, self.out_size_factor_img = num_views, out_size_factor_img
        for _ in range(num_mmpi // 2):
            heads = copy.deepcopy(common_heads)
            heads.update(dict(heatmap=(self.num_classes, num_heatmap_convs)))
            self.decode_head.append(ImageRCNNBlock(num_views, num_proposals, out_size_factor_img, test_cfg,
                                                   self.bbox_coder, hidden_channel, num_heads, dropout))
            self.pred_head.append(FFN(hidden_channel * 2, heads, conv_cfg=conv_cfg, norm_cfg=norm_cfg, bias=bias))
            self.decode_head.append(PointRCNNBlock(hidden_channel, num_heads, dropout, self.bbox_coder))
            self.pred_head.append(FFN(hidden_channel * 2, heads, conv_cfg=conv_cfg, norm_cfg=norm_cfg, bias=bias))

        x_size = test_cfg['grid_size'][0] // test_cfg['out_size_factor']
        y_size = test_cfg['grid_size'][1] // test_cfg['out_size_factor']
        self.bev_pos = self.create_2D_grid(x_size, y_size)
        self._bev_pos_dev = None
        self.img_feat_pos = None
        self.query_labels = None
        self.on_the_image_mask = []
        self.ret_idx = ret_idx
        self.static_geometry = None       # optional persistent QueryGeometry (deepinteraction_amd.graphed)
        self.init_weights()

    def create_2D_grid(self, x_size, y_size):
        """Reference :162-169: BEV cell centres (col + 0.5, row + 0.5), row-major."""
        by, bx = torch.meshgrid(torch.linspace(0, x_size - 1, x_size), torch.linspace(0, y_size - 1, y_size),
                                indexing='ij')
        coord_base = torch.cat([(bx + 0.5)[None], (by + 0.5)[None]], dim=0)[None]
        return coord_base.view(1, 2, -1).permute(0, 2, 1)

    def init_weights(self):
        for m in self.decoder.parameters():
            if m.dim() > 1:
                nn.init.xavier_uniform_(m)
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = self.bn_momentum

    def _bev_pos(self, device):
        if self._bev_pos_dev is None or self._bev_pos_dev.device != device:
            self._bev_pos_dev = self.bev_pos.to(device)
        return self._bev_pos_dev

    def _heatmap(self, head, feat):
        return head(ops.cl(feat)).contiguous()                                  # (B,num_classes,H,W) NCHW

    def forward(self, pts_inputs, img_inputs, img_metas):
        lidar_feat, new_lidar_feat = ops.cl(pts_inputs[0]), ops.cl(pts_inputs[1])
        B, C, H, W = lidar_feat.shape
        HW = H * W
        dev = lidar_feat.device
        img_feat = ops.cl(img_inputs)
        BN, I_C, I_H, I_W = img_feat.shape
        lidar_flat = lidar_feat.view(B, C, HW)                                   # strided view of channels-last
        bev_pos = self._bev_pos(dev)                                             # (1,HW,2)

        dense_heatmap = self._heatmap(self.heatmap_head, lidar_feat)
        dense_heatmap_img = self._heatmap(self.heatmap_head_img, new_lidar_feat)
        k1 = {'nuScenes': (8, 9), 'Waymo': (1, 2)}.get(self.test_cfg['dataset'], ())
        heatmap = ops.heatmap_nms(dense_heatmap.detach(), dense_heatmap_img.detach(), self.nms_kernel_size,
                                  [c for c in k1 if c < self.num_classes]).view(B, self.num_classes, HW)

        # top proposals over all (class, cell) pairs: reference argsort(descending)[:Q] (:242)
        top = heatmap.view(B, -1).topk(self.num_proposals, dim=-1, largest=True, sorted=True).indices
        top_class = top // HW
        top_index = top % HW
        self.query_labels = top_class
        self.top_proposals = top                                                 # flattened (class, cell) picks
        tokens = lidar_flat.transpose(1, 2)                                      # (B,HW,C) contiguous view
        query_tok = tokens.gather(1, top_index[:, :, None].expand(-1, -1, C))    # (B,Q,C)
        # class_encoding(one_hot) == the class column of the 1x1 conv + bias (:249-251)
        w_ce = self.class_encoding.weight[:, :, 0].t()                           # (num_classes, C)
        query_tok = query_tok + w_ce[top_class] + self.class_encoding.bias
        query_feat = query_tok.transpose(1, 2)                                   # (B,C,Q)
        query_pos = torch.stack([(top_index % W).float() + 0.5, (top_index // W).float() + 0.5], -1)  # (:253)

        for i in range(self.num_decoder_layers):
            query_feat = self.decoder[i](query_feat, lidar_flat, query_pos, bev_pos)
            res_layer = self.prediction_heads[i](query_feat)
            res_layer['center'] = res_layer['center'] + query_pos.permute(0, 2, 1)
            first_res_layer = res_layer
            query_pos = res_layer['center'].detach().permute(0, 2, 1)

        img_feat_flatten = img_feat.view(B, self.num_views, I_C, -1)
        # a caller that replays a captured graph keeps ONE geometry object alive and refreshes it in place
        geom = self.static_geometry if self.static_geometry is not None else QueryGeometry(img_metas, dev)
        self.on_the_image_mask = []
        ret_dicts = []
        for layer_idx in range(self.num_mmpi):
            prev_query_feat = query_feat
            query_pos = res_layer['center'].detach().permute(0, 2, 1)
            query_feat, on_the_image = self.decode_head[layer_idx](
                query_feat=prev_query_feat, res_layer=res_layer, new_lidar_feat=new_lidar_feat,
                img_feat_flatten=img_feat_flatten, img_metas=img_metas, img_h=I_H, img_w=I_W, geom=geom)
            res_layer = self.pred_head[layer_idx](torch.cat([query_feat, prev_query_feat], dim=1))
            res_layer['center'] = res_layer['center'] + query_pos.permute(0, 2, 1)
            if layer_idx % 2 == 0:
                mask = on_the_image != -1
                self.on_the_image_mask.append(mask)
                res_layer = {k: torch.where(mask.unsqueeze(1), v, first_res_layer[k]) for k, v in res_layer.items()}
            ret_dicts.append(res_layer)

        ret_dicts[0]['query_heatmap_score'] = heatmap.gather(
            index=top_index[:, None, :].expand(-1, self.num_classes, -1), dim=-1)
        ret_dicts[0]['dense_heatmap'] = dense_heatmap_img
        if self.auxiliary is False:
            return [ret_dicts[-1]]
        new_res = {}
        for key in ret_dicts[0].keys():
            if key not in ['dense_heatmap', 'dense_heatmap_old', 'query_heatmap_score']:
                new_res[key] = torch.cat([ret_dict[key] for ret_dict in ret_dicts], dim=-1)
            else:
                new_res[key] = ret_dicts[0][key]
        return [[new_res]]
