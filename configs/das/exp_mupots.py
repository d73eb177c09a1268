# DAS on MuCo-3DHP / MuPoTS-3D: 3-stage MSPN-50, J=21, root joint 14, plain BN, 2 recursive-update layers.
_base_ = [
    '../_base_/datasets/muco.py', '../_base_/models/das.py',
    '../_base_/schedules/mmdet_schedule_1x.py', '../_base_/default_runtime.py'
]
fpn_channels = 256
num_joints = 21
model = dict(
    pretrained='weights/3xmspn50_coco_256x192-e348f18e_20201123.pth',
    backbone=dict(
        _delete_=True,
        type='MSPN2',
        unit_channels=256,
        num_stages=3,
        num_units=4,
        num_blocks=[3, 4, 6, 3],
        norm_cfg=dict(type='BN'),
        frozen_stages=1,
        norm_eval=False),
    neck=dict(
        type='FPN',
        in_channels=[256, 256, 256, 256],
        out_channels=fpn_channels,
        norm_cfg=dict(type='BN'),
        num_outs=4),
    bbox_head=dict(
        type='DASHead',
        in_channels=fpn_channels,
        stacked_convs=2,
        feat_channels=fpn_channels,
        regress_ranges=((-1, 80), (80, 160), (160, 320), (320, 1e8)),
        strides=[8, 16, 32, 64],
        center_sample_radius=1.5,
        num_joints=num_joints,
        depth_factor=1,
        z_norm=50,
        root_idx=14,
        recursive_update=dict(num_joints=num_joints, num_layers=2)),
    train_cfg=dict(code_weight=[1.0, 1.0, 1] + [2] * num_joints * 6),
    test_cfg=dict(nms_across_levels=False, nms_pre=1000, nms_post=100, nms_thr=0.9, score_thr=0.07))

optimizer = dict(lr=2e-3, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
optimizer_config = dict(_delete_=True, grad_clip=dict(max_norm=35, norm_type=2))
runner = dict(type='EpochBasedRunner', max_iters=None, max_epochs=22)
lr_config = dict(policy='step', warmup='linear', warmup_iters=250, warmup_ratio=1.0 / 3, step=[16, 20])
log_config = dict(interval=50)
checkpoint_config = dict(interval=1, max_keep_ckpts=20)
evaluation = dict(interval=1)
find_unused_parameters = True
fp16 = dict(loss_scale=dict(init_scale=512))
