"""Head/neck constructor kwargs of the reference configs, for the bench, smoke and tests (
values of reference projects/configs/nuscenes/Fusion_0075_refactor.py:185-224,243-251)."""
import copy

POINT_CLOUD_RANGE = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]


def decoder_cfg(bev=180, num_proposals=200, voxel=None, num_views=6):
    """`bev` = BEV cells per side; the reference has 180 (= 1440 / 8, voxel 0.075)."""
    osf = 8
    voxel = 108.0 / (bev * osf) if voxel is None else voxel
    return copy.deepcopy(dict(
        num_views=num_views, out_size_factor_img=4, num_proposals=num_proposals, auxiliary=True,
        hidden_channel=128, num_classes=10, num_mmpi=4, num_heads=8, learnable_query_pos=False,
        initialize_by_heatmap=True, nms_kernel_size=3, ffn_channel=256, dropout=0.1, bn_momentum=0.1,
        activation='relu',
        common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
        bbox_coder=dict(type='TransFusionBBoxCoder', pc_range=POINT_CLOUD_RANGE[:2], voxel_size=[voxel, voxel],
                        out_size_factor=osf, post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                        score_threshold=0.0, code_size=10),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2, alpha=0.25, reduction='mean', loss_weight=1.0),
        loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25),
        loss_heatmap=dict(type='GaussianFocalLoss', reduction='mean', loss_weight=1.0),
        train_cfg=None,
        test_cfg=dict(dataset='nuScenes', grid_size=[bev * osf, bev * osf, 40], out_size_factor=osf,
                      pc_range=POINT_CLOUD_RANGE[0:2], voxel_size=[voxel, voxel], nms_type=None),
    ))


def encoder_pp_cfg(in_channels_img=256, in_channels_pts=256, num_layers=2):
    """`imgpts_neck` of reference projects/configs/nuscenes/Fusion_0075_plusplus.py:210-267."""
    ffn = dict(type='FFN', embed_dims=128, feedforward_channels=512, num_fcs=2, ffn_drop=0.1,
               act_cfg=dict(type='ReLU', inplace=True))
    msda = dict(type='MultiScaleDeformableAttention', embed_dims=128, num_levels=2, batch_first=True)
    return copy.deepcopy(dict(
        num_layers=num_layers, in_channels_img=in_channels_img, in_channels_pts=in_channels_pts, hidden_channel=128,
        bn_momentum=0.1, bias='auto',
        img_transformerlayers=dict(
            type='DeepInteractionLayer',
            attn_cfgs=[msda, dict(type='MMRI_P2I', embed_dims=128, batch_first=True)],
            ffn_cfgs=ffn,
            operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm', 'ffn', 'norm')),
        pts_transformerlayers=dict(
            type='DeepInteractionLayer',
            attn_cfgs=[msda, dict(type='MMRI_I2P_Polar', embed_dims=128, dropout=0.1, batch_first=True),
                       dict(type='MMRI_I2P', embed_dims=128, dropout=0.1, batch_first=True, fp16_enabled=True,
                            group_attn_enabled=True)],
            ffn_cfgs=ffn,
            operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')),
    ))
