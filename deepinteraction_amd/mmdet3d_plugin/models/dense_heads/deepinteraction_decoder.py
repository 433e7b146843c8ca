"""MMPI decoder head (drop-in for reference `models/dense_heads/deepinteraction_decoder.py`,
`DeepInteractionDecoder` :19-313: same registry name, constructor kwargs, forward signature /
return structure, side attributes `query_labels` / `on_the_image_mask`, `state_dict` keys).

`forward` is the hot path: shape-static, sync-free execution (fused heat-map NMS kernel, library top-k,
batched RoI blocks - see models/utils/decoder_utils.py - prediction heads in float32).
`get_targets` / `get_targets_single` / `loss` / `get_bboxes` (reference :315-638) are host logic on torch
tensors: they mirror the reference line by line, with the mmdet / mmdet3d helpers they call restated in
deepinteraction_amd/det3d_compat.py (parity unpinned at that third-party boundary).
"""
import copy

import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .... import decoder_fused, ops
from ....det3d_compat import (AssignResult, LiDARBoxes, build_loss, circle_nms, clip_sigmoid, draw_heatmap_gaussian,
                              draw_heatmap_gaussian_host,
                              nms_rotated_bev, xywhr2xyxyr,
                              gaussian_radius, pseudo_sample)
from ....registry import HEADS, build_bbox_coder
from ....utils import param_key
from ...core.bbox.assigners import build_assigner
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
        if loss_cls is not None:                       # reference :75-77 (mmdet build_loss)
            self.loss_cls, self.loss_bbox = build_loss(loss_cls), build_loss(loss_bbox)
            self.loss_heatmap = build_loss(loss_heatmap)
        self.sampling = False
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
            self._add_mmpi_pair(heads, num_proposals, hidden_channel, num_heads, dropout, conv_cfg, norm_cfg, bias)

        x_size = test_cfg['grid_size'][0] // test_cfg['out_size_factor']
        y_size = test_cfg['grid_size'][1] // test_cfg['out_size_factor']
        self.bev_pos = self.create_2D_grid(x_size, y_size)
        self._bev_pos_dev = None
        self.img_feat_pos = None
        self.query_labels = None
        self.on_the_image_mask = []
        self.ret_idx = ret_idx
        self.static_geometry = None       # optional persistent QueryGeometry (deepinteraction_amd.graphed)
        self.fused = True                 # fp16 inference: the token-kernel path (deepinteraction_amd/decoder_fused.py)
        self._fused_path = None
        self.init_weights()
        self._init_assigner_sampler()

    def _add_mmpi_pair(self, heads, num_proposals, hidden_channel, num_heads, dropout, conv_cfg, norm_cfg, bias):
        """One image RoI block + one point RoI block with their prediction heads (reference :131-147)."""
        self.decode_head.append(ImageRCNNBlock(self.num_views, num_proposals, self.out_size_factor_img, self.test_cfg,
                                               self.bbox_coder, hidden_channel, num_heads, dropout))
        self.pred_head.append(FFN(hidden_channel * 2, heads, conv_cfg=conv_cfg, norm_cfg=norm_cfg, bias=bias))
        self.decode_head.append(PointRCNNBlock(hidden_channel, num_heads, dropout, self.bbox_coder))
        self.pred_head.append(FFN(hidden_channel * 2, heads, conv_cfg=conv_cfg, norm_cfg=norm_cfg, bias=bias))

    def _init_assigner_sampler(self):
        """Reference :185-200: PseudoSampler + the assigner(s) named by train_cfg."""
        if self.train_cfg is None:
            return
        assigner = self.train_cfg['assigner']
        self.bbox_assigner = ([build_assigner(a) for a in assigner] if isinstance(assigner, list)
                              else build_assigner(assigner))

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

    def _heatmap(self, head, feat, out_f32=False):
        """(B,num_classes,H,W) NCHW heat-map logits: ConvModule(3x3 + BN + ReLU) + Conv 3x3 (reference :96-119).  fp16
        maps at inference: two launches of the HIP implicit-GEMM kernel (BatchNorm folded; weights of either float type
        are packed to fp16 once); `out_f32`: float32 logits (they feed the NMS comparison and the top-Q pick)."""
        feat = ops.cl(feat)
        cm, last = head[0], head[1]
        if (feat.is_cuda and torch.float16 in (feat.dtype, cm.conv.weight.dtype) and not torch.is_grad_enabled()
                and not self.training and feat.shape[1] % 32 == 0 and cm.conv.out_channels == 128
                and last.out_channels <= 16):
            feat = feat.half()                      # mixed mode off the fused path: fp16 heads on widened maps
            cache = self.__dict__.setdefault('_heat_cache', {})
            key = param_key(head)
            hit = cache.get(id(head))
            if hit is None or hit[0] != key:
                hit = (key, ops.pack_conv3x3(cm.conv.weight, cm.conv.bias, cm.bn if cm.with_norm else None),
                       ops.pack_conv3x3(last.weight, last.bias))
                cache[id(head)] = hit
            mid = ops.conv3x3(feat, *hit[1], relu=True)
            return ops.conv3x3(mid, *hit[2], out_nchw=True, out_f32=out_f32)
        out = head(feat.to(cm.conv.weight.dtype)).contiguous()
        return out.float() if out_f32 else out

    def forward(self, pts_inputs, img_inputs, img_metas):
        if self.fused and (type(self)._mmpi is DeepInteractionDecoder._mmpi or decoder_fused.is_plusplus(self)) \
                and decoder_fused.usable(self, pts_inputs[0], img_inputs):
            if self._fused_path is None:
                self._fused_path = decoder_fused.FusedDecoder()
            return self._fused_path.forward(self, pts_inputs, img_inputs, img_metas)
        wd = self.class_encoding.weight.dtype
        if pts_inputs[0].dtype != wd:                 # mixed mode (precision.half_maps_) off the fused path: widen the maps
            pts_inputs, img_inputs = [p.to(wd) for p in pts_inputs], img_inputs.to(wd)
        lidar_feat, new_lidar_feat = ops.cl(pts_inputs[0]), ops.cl(pts_inputs[1])
        B, C, H, W = lidar_feat.shape
        HW = H * W
        dev = lidar_feat.device
        img_feat = ops.cl(img_inputs)
        BN, I_C, I_H, I_W = img_feat.shape
        lidar_flat = lidar_feat.view(B, C, HW)                                   # strided view of channels-last
        bev_pos = self._bev_pos(dev)                                             # (1,HW,2)

        dense_heatmap = self._heatmap(self.heatmap_head, lidar_feat, out_f32=True)
        dense_heatmap_img = self._heatmap(self.heatmap_head_img, new_lidar_feat, out_f32=True)
        k1 = {'nuScenes': (8, 9), 'Waymo': (1, 2)}.get(self.test_cfg['dataset'], ())
        heatmap = ops.heatmap_nms(dense_heatmap.detach(), dense_heatmap_img.detach(), self.nms_kernel_size,
                                  [c for c in k1 if c < self.num_classes]).view(B, self.num_classes, HW)

        # top proposals over all (class, cell) pairs: reference argsort(descending)[:Q] (:242)
        flat_scores = heatmap.view(B, -1)
        if flat_scores.is_cuda and self.num_proposals <= 1024 and flat_scores.shape[1] <= (1 << 20):
            top = ops.topk(flat_scores, self.num_proposals)           # radix select; ties: lower index first
        else:
            top = flat_scores.topk(self.num_proposals, dim=-1, largest=True, sorted=True).indices
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
        ret_dicts = self._mmpi(query_feat, res_layer, first_res_layer, new_lidar_feat, img_feat_flatten, img_metas,
                               I_H, I_W, geom)

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

    def _mmpi(self, query_feat, res_layer, first_res_layer, new_lidar_feat, img_feat_flatten, img_metas, I_H, I_W,
              geom):
        """Reference :279-297: the alternating image / point RoI refinement; returns the per-layer dicts."""
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
        return ret_dicts

    def _layer_mask(self, l):
        """The on-the-image mask that weights the targets of MMPI layer `l` (reference :506-508), or None."""
        return self.on_the_image_mask[l // 2] if l % 2 == 0 else None

    # ------------------------------------------------------------------ ground-truth-only targets
    def dense_heatmap_target(self, gt_bboxes_3d, gt_labels_3d, device):
        """The dense heat-map target of one sample (reference :450-475): one Gaussian per GT box on the BEV grid.  The boxes
        are host data: radius, centre and the element-wise max are evaluated on the host (the same float32 arithmetic, no
        device round trip per box) and the finished map is uploaded once.  Returns (map on `device`, number of peaks)."""
        tc = self.train_cfg
        gt_labels_host = gt_labels_3d.cpu()
        gt = torch.cat([gt_bboxes_3d.gravity_center, gt_bboxes_3d.tensor[:, 3:]], dim=1).float().cpu()
        grid_size = torch.tensor(tc['grid_size'])
        pc_range = torch.tensor(tc['point_cloud_range'])
        voxel_size = torch.tensor(tc['voxel_size'])
        fmap = grid_size[:2] // tc['out_size_factor']           # [x_len, y_len]
        heatmap = np.zeros((self.num_classes, int(fmap[1]), int(fmap[0])), dtype=np.float32)
        for idx in range(len(gt)):
            width = gt[idx][3] / voxel_size[0] / tc['out_size_factor']
            length = gt[idx][4] / voxel_size[1] / tc['out_size_factor']
            if width > 0 and length > 0:
                radius = gaussian_radius((length, width), min_overlap=tc['gaussian_overlap'])
                radius = max(tc['min_radius'], int(radius))
                coor_x = (gt[idx][0] - pc_range[0]) / voxel_size[0] / tc['out_size_factor']
                coor_y = (gt[idx][1] - pc_range[1]) / voxel_size[1] / tc['out_size_factor']
                center_int = torch.tensor([coor_x, coor_y], dtype=torch.float32).to(torch.int32)
                draw_heatmap_gaussian_host(heatmap[int(gt_labels_host[idx])], center_int, radius)
        peaks = int((heatmap == 1).sum())
        return torch.from_numpy(heatmap).to(device), peaks

    def prepare_targets(self, gt_bboxes_3d, gt_labels_3d, device):
        """Optional, ahead of `loss`: evaluate the targets that depend on the ground truth ALONE (the dense heat map) - e.g.
        while the device runs the forward the loss will wait for.  `loss` / `get_targets` pick them up (same objects) and
        compute whatever was not prepared; the values are the same either way."""
        # (replaces whatever an earlier call left behind; an entry keeps its box object so that a recycled id() cannot match)
        self._gt_targets = {id(b): (b, self.dense_heatmap_target(b, l, device)) for b, l in zip(gt_bboxes_3d, gt_labels_3d)}

    # ------------------------------------------------------------------ targets (reference :315-482)
    def get_targets(self, gt_bboxes_3d, gt_labels_3d, preds_dict):
        """gt_bboxes_3d: list of LiDARInstance3DBoxes-like (`.tensor`, `.gravity_center`); preds_dict: the
        `[dict]` of one forward.  Returns (labels, label_weights, bbox_targets, bbox_weights, ious, num_pos,
        matched_ious, heatmap) as the reference."""
        self._heatmap_peaks = {}
        res = [self.get_targets_single(gt_bboxes_3d[b], gt_labels_3d[b],
                                       {k: v[b:b + 1] for k, v in preds_dict[0].items()}, b)
               for b in range(len(gt_bboxes_3d))]
        self._gt_targets = {}                                    # prepared targets do not outlive the call they were made for
        cat = lambda i: torch.cat([r[i] for r in res], dim=0)
        num_pos = int(np.sum([r[5] for r in res]))
        matched_ious = float(np.mean([r[6] for r in res]))
        return cat(0), cat(1), cat(2), cat(3), cat(4), num_pos, matched_ious, cat(7)

    def get_targets_single(self, gt_bboxes_3d, gt_labels_3d, preds_dict, batch_idx):
        num_proposals = preds_dict['center'].shape[-1]
        dev = preds_dict['center'].device
        det = lambda k: preds_dict[k].detach().clone()          # "carefully! do not change the network outputs"
        vel = det('vel') if 'vel' in preds_dict else None
        boxes_dict = self.bbox_coder.decode(det('heatmap'), det('rot'), det('dim'), det('center'), det('height'), vel)
        bboxes_tensor = boxes_dict[0]['bboxes']
        score = preds_dict['heatmap'].detach()
        gt_bboxes_tensor = gt_bboxes_3d.tensor.to(dev)
        gt_labels_host = gt_labels_3d.cpu()                      # (the labels arrive as host data)
        gt_labels_3d = gt_labels_3d.to(dev)
        num_layer = self.num_mmpi if self.auxiliary else 1
        Q = self.num_proposals
        tc = self.train_cfg
        zeros = lambda: bboxes_tensor.new_zeros(Q)
        if (tc['assigner']['type'] == 'HungarianAssigner3D' and hasattr(self.bbox_assigner, 'assign_layers')
                and not os.environ.get('DI_ASSIGN_PER_LAYER')):          # (the switch: A/B timing of the two forms)
            # every layer is assigned separately; their costs are evaluated together (one device -> host copy per sample)
            ens = self.bbox_assigner.assign_layers(bboxes_tensor[:Q * num_layer], gt_bboxes_tensor, gt_labels_3d,
                                                   score[..., :Q * num_layer], tc, num_layer)
        else:
            assign_result_list = []
            for l in range(num_layer):                          # every layer is assigned separately
                boxes_l = bboxes_tensor[Q * l:Q * (l + 1), :]
                score_l = score[..., Q * l:Q * (l + 1)]
                if tc['assigner']['type'] == 'HungarianAssigner3D':
                    r = self.bbox_assigner.assign(boxes_l, gt_bboxes_tensor, gt_labels_3d, score_l, tc)
                elif tc['assigner']['type'] == 'HeuristicAssigner':
                    r = self.bbox_assigner.assign(boxes_l, gt_bboxes_tensor, None, gt_labels_3d,
                                                  self.query_labels[batch_idx])
                else:
                    raise NotImplementedError
                assign_result_list.append(r)
            ens = AssignResult(
                num_gts=sum(r.num_gts for r in assign_result_list),
                gt_inds=torch.cat([r.gt_inds for r in assign_result_list]),
                max_overlaps=torch.cat([zeros() if r.max_overlaps is None else r.max_overlaps
                                        for r in assign_result_list]),
                labels=torch.cat([r.labels for r in assign_result_list]))
        sampling = pseudo_sample(ens, bboxes_tensor, gt_bboxes_tensor)
        pos_inds, neg_inds = sampling.pos_inds, sampling.neg_inds
        assert len(pos_inds) + len(neg_inds) == num_proposals

        code = self.bbox_coder.code_size
        bbox_targets = torch.zeros([num_proposals, code], device=dev)
        bbox_weights = torch.zeros([num_proposals, code], device=dev)
        ious = torch.clamp(ens.max_overlaps, min=0.0, max=1.0)
        labels = bboxes_tensor.new_zeros(num_proposals, dtype=torch.long)
        label_weights = bboxes_tensor.new_zeros(num_proposals, dtype=torch.long)
        if gt_labels_3d is not None:                            # default label = background
            labels += self.num_classes
        if len(pos_inds) > 0:                                   # only positives regress
            bbox_targets[pos_inds, :] = self.bbox_coder.encode(sampling.pos_gt_bboxes)
            bbox_weights[pos_inds, :] = 1.0
            labels[pos_inds] = 1 if gt_labels_3d is None else gt_labels_3d[sampling.pos_assigned_gt_inds]
            label_weights[pos_inds] = 1.0 if tc['pos_weight'] <= 0 else tc['pos_weight']
        if len(neg_inds) > 0:
            label_weights[neg_inds] = 1.0

        # dense heat-map target: a function of the ground truth alone (see `dense_heatmap_target`)
        if getattr(self, '_heatmap_peaks', None) is None:                # (get_targets_single called on its own)
            self._heatmap_peaks = {}
        entry = getattr(self, '_gt_targets', {}).pop(id(gt_bboxes_3d), None)
        ready = entry[1] if entry is not None and entry[0] is gt_bboxes_3d else None
        heatmap, peaks = ready if ready is not None else self.dense_heatmap_target(gt_bboxes_3d, gt_labels_host, dev)
        self._heatmap_peaks[batch_idx] = peaks                           # `loss` normalises by it: known on the host
        mean_iou = ious[pos_inds].sum() / max(len(pos_inds), 1)
        return (labels[None], label_weights[None], bbox_targets[None], bbox_weights[None], ious[None],
                int(pos_inds.shape[0]), float(mean_iou), heatmap[None])

    # ------------------------------------------------------------------ loss (reference :484-547)
    def loss(self, gt_bboxes_3d, gt_labels_3d, preds_dicts, **kwargs):
        (labels, label_weights, bbox_targets, bbox_weights, ious, _num_pos, matched_ious,
         heatmap) = self.get_targets(gt_bboxes_3d, gt_labels_3d, preds_dicts[0])
        Q, L = self.num_proposals, self.num_mmpi
        B = labels.shape[0]
        # The per-layer terms of the reference (one classification and one box loss per MMPI layer, each normalised by that
        # layer's number of positives, :506-540) evaluated for ALL layers at once: the element-wise losses over (B, L*Q, .)
        # in one pass, then a sum per layer - 5 x fewer launches in the phase the device spends waiting for Python.
        masks = [self._layer_mask(l) for l in range(L)]
        if any(m is not None for m in masks):                   # image layers: only queries some camera sees
            full = torch.cat([torch.ones_like(label_weights[..., :Q], dtype=bbox_weights.dtype) if m is None
                              else m.to(bbox_weights.dtype) for m in masks], dim=-1)            # (B, L*Q)
            label_weights = label_weights * full.to(label_weights.dtype)        # (0 / 1 masks: exact in any dtype)
            bbox_weights = bbox_weights * full[:, :, None]
        num_pos = bbox_weights.max(-1).values.view(B, L, Q).sum((0, 2)).clamp(min=1)            # (L,) = max(num_pos, 1)
        preds_dict = preds_dicts[0][0]
        loss_dict = dict()
        loss_dict['loss_heatmap'] = self.loss_heatmap(clip_sigmoid(preds_dict['dense_heatmap'].float()), heatmap,
                                                      avg_factor=max(float(sum(self._heatmap_peaks.values())), 1))
        code_weights = self.train_cfg.get('code_weights', None)
        per_layer = lambda t: t.view(B, L, Q, -1).sum((0, 2, 3))                                 # (B, L*Q, .) -> (L,)
        cls_score = preds_dict['heatmap'][..., :L * Q].permute(0, 2, 1).reshape(-1, self.num_classes).float()
        cls_el = self.loss_cls(cls_score, labels[..., :L * Q].reshape(-1), label_weights[..., :L * Q].reshape(-1),
                               reduction_override='none')
        loss_cls = per_layer(cls_el.view(B, L * Q, -1)) / num_pos
        parts = [preds_dict[k][..., :L * Q] for k in ('center', 'height', 'dim', 'rot')]
        if 'vel' in preds_dict:
            parts.append(preds_dict['vel'][..., :L * Q])
        preds = torch.cat(parts, dim=1).permute(0, 2, 1).float()                                 # (B, L*Q, code_size)
        reg_w = bbox_weights[:, :L * Q, :] * bbox_weights.new_tensor(code_weights)
        box_el = self.loss_bbox(preds, bbox_targets[:, :L * Q, :], reg_w, reduction_override='none')
        loss_bbox = per_layer(box_el) / num_pos
        for l in range(L):
            loss_dict[f'layer_{l}_loss_cls'] = loss_cls[l]
            loss_dict[f'layer_{l}_loss_bbox'] = loss_bbox[l]
        layer_loss_cls = loss_cls[-1]
        loss_dict['matched_ious'] = layer_loss_cls.new_tensor(matched_ious)
        return loss_dict

    # ------------------------------------------------------------------ boxes (reference :549-638)
    def get_bboxes(self, preds_dicts, img_metas, img=None, rescale=False, for_roi=False):
        """Last layer's Q queries -> [[boxes, scores, labels]] (batch size 1, as the reference asserts)."""
        rets = []
        Q = self.num_proposals
        for preds_dict in preds_dicts:
            pd = preds_dict[0]
            batch_size = pd['heatmap'].shape[0]
            one_hot = F.one_hot(self.query_labels, num_classes=self.num_classes).permute(0, 2, 1)
            batch_score = pd['heatmap'][..., -Q:].sigmoid() * pd['query_heatmap_score'] * one_hot
            vel = pd['vel'][..., -Q:] if 'vel' in pd else None
            temp = self.bbox_coder.decode(batch_score, pd['rot'][..., -Q:], pd['dim'][..., -Q:],
                                          pd['center'][..., -Q:], pd['height'][..., -Q:], vel, filter=True)
            if self.test_cfg['dataset'] == 'nuScenes':
                tasks = [dict(indices=[0, 1, 2, 3, 4, 5, 6, 7], radius=-1), dict(indices=[8], radius=0.175),
                         dict(indices=[9], radius=0.175)]
            elif self.test_cfg['dataset'] == 'Waymo':
                tasks = [dict(indices=[0], radius=0.7), dict(indices=[1], radius=0.7), dict(indices=[2], radius=0.7)]
            ret_layer = []
            for i in range(batch_size):
                boxes3d, scores, labels = temp[i]['bboxes'], temp[i]['scores'], temp[i]['labels']
                if self.test_cfg['nms_type'] is not None:
                    keep_mask = torch.zeros_like(scores)
                    for task in tasks:
                        task_mask = torch.zeros_like(scores)
                        for cls_idx in task['indices']:
                            task_mask += labels == cls_idx
                        task_mask = task_mask.bool()
                        if task['radius'] > 0 and self.test_cfg['nms_type'] == 'circle':
                            dets = torch.cat([boxes3d[task_mask][:, :2], scores[:, None][task_mask]], dim=1)
                            keep = torch.tensor(circle_nms(dets.detach().cpu().numpy(), task['radius']),
                                                dtype=torch.long, device=scores.device)
                        elif task['radius'] > 0:        # rotated BEV NMS, the task's `radius` is the IoU threshold
                            bev = boxes3d[task_mask][:, [0, 1, 3, 4, 6]]
                            keep = nms_rotated_bev(xywhr2xyxyr(bev), scores[task_mask], task['radius'],
                                                   self.test_cfg.get('pre_maxsize'), self.test_cfg.get('post_maxsize'))
                        else:
                            keep = torch.arange(int(task_mask.sum()), device=scores.device)
                        if keep.shape[0] != 0:
                            keep_mask[torch.where(task_mask != 0)[0][keep]] = 1
                    keep_mask = keep_mask.bool()
                    ret = dict(bboxes=boxes3d[keep_mask], scores=scores[keep_mask], labels=labels[keep_mask])
                else:
                    ret = dict(bboxes=boxes3d, scores=scores, labels=labels)
                ret_layer.append(ret)
            rets.append(ret_layer)
        assert len(rets) == 1 and len(rets[0]) == 1
        box_type = img_metas[0].get('box_type_3d', LiDARBoxes)
        r = rets[0][0]
        return [[box_type(r['bboxes'], box_dim=r['bboxes'].shape[-1]), r['scores'], r['labels'].int()]]
