"""DeepInteraction++ head (drop-in for reference `models/dense_heads/deepinteractionplusplus_decoder.py`,
`DeepInteractionPlusPlusDecoder` :20-320): the v1 head with V2 RoI blocks, 128-channel prediction heads on the
refined query alone (:140,:147,:291), the look-forward centre update (:281-294) and a cumulative on-the-image mask
applied after every MMPI layer (:295-302; targets weighted per layer, :513-514).  Everything else - heat-map
initialisation, the BEV transformer layer, targets, losses, box decoding - is inherited unchanged, as in the
reference (the two files differ only in the lines cited)."""
import torch

from ....registry import HEADS
from ..utils.decoder_utils import FFN, ImageRCNNBlockV2, PointRCNNBlockV2
from .deepinteraction_decoder import DeepInteractionDecoder


@HEADS.register_module()
class DeepInteractionPlusPlusDecoder(DeepInteractionDecoder):
    def _add_mmpi_pair(self, heads, num_proposals, hidden_channel, num_heads, dropout, conv_cfg, norm_cfg, bias):
        self.decode_head.append(ImageRCNNBlockV2(self.num_views, num_proposals, self.out_size_factor_img, self.test_cfg,
                                                 self.bbox_coder, hidden_channel, num_heads, dropout))
        self.pred_head.append(FFN(hidden_channel, heads, conv_cfg=conv_cfg, norm_cfg=norm_cfg, bias=bias))
        self.decode_head.append(PointRCNNBlockV2(hidden_channel, num_heads, dropout, self.bbox_coder))
        self.pred_head.append(FFN(hidden_channel, heads, conv_cfg=conv_cfg, norm_cfg=norm_cfg, bias=bias))

    def _mmpi(self, query_feat, res_layer, first_res_layer, new_lidar_feat, img_feat_flatten, img_metas, I_H, I_W,
              geom):
        self.on_the_image_mask = []
        ret_dicts = []
        look_forward = res_layer['center'].detach()                              # :281
        for layer_idx in range(self.num_mmpi):
            query_pos = res_layer['center'].detach()
            query_feat, on_the_image = self.decode_head[layer_idx](
                query_feat=query_feat, res_layer=res_layer, new_lidar_feat=new_lidar_feat,
                img_feat_flatten=img_feat_flatten, img_metas=img_metas, img_h=I_H, img_w=I_W, geom=geom)
            res_layer = self.pred_head[layer_idx](query_feat)
            delta = res_layer['center']
            res_layer['center'] = delta + look_forward                           # :293
            look_forward = delta + query_pos                                     # :294
            if layer_idx % 2 == 0:
                mask = on_the_image != -1
                if layer_idx > 0:
                    mask = mask & self.on_the_image_mask[-1]
                self.on_the_image_mask.append(mask)
            else:
                self.on_the_image_mask.append(self.on_the_image_mask[-1])
            mask = self.on_the_image_mask[-1]
            res_layer = {k: torch.where(mask.unsqueeze(1), v, first_res_layer[k]) for k, v in res_layer.items()}
            ret_dicts.append(res_layer)
        return ret_dicts

    def _layer_mask(self, l):
        return self.on_the_image_mask[l]                                         # :513-514, every layer
