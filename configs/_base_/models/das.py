# Base DAS detector. The experiment files replace the backbone wholesale (`_delete_=True`) and
# override parts of neck / bbox_head; every key they do not mention stays active, notably
# neck.start_level=1 + add_extra_convs='on_output' (the stride-4 backbone map is not used by the head)
# and the head's branch widths / DCN / recursive-update settings.
model = dict(
    type='DAS',
    pretrained='open-mmlab://detectron2/resnet50_caffe',
    backbone=dict(  # placeholder, always replaced by MSPN2 in configs/das/*
        type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
        norm_cfg=dict(type='BN', requires_grad=True), norm_eval=False, style='caffe'),
    neck=dict(
        type='FPN',
        in_channels=[256, 512, 1024, 2048],
        out_channels=256,
        start_level=1,
        add_extra_convs='on_output',
        num_outs=5,
        relu_before_extra_convs=True),
    bbox_head=dict(
        type='DASHead',
        num_classes=1,
        in_channels=256,
        feat_channels=256,
        stacked_convs=2,
        strides=[8, 16, 32, 64, 128],
        center_sample_radius=1.5,
        num_joints=15,
        cls_branch=(256,),
        reg_branch=((256,), (256,), (256,), (256,)),  # root offset, root depth, joint uvd, joint sigma
        centerness_on_reg=True,
        conv_bias=True,
        dcn_on_last_conv=True,
        recursive_update=dict(prev_loss=True, num_heads=4, in_channels=256, feat_channels=256, num_layers=1, dim=3)))
