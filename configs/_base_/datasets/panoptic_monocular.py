# CMU Panoptic, monocular. The real-data classes / CPU augmentation pipeline of the reference are
# outside this repo's scope; the benchmark and tests use `SyntheticPoseDataset`,
# which emits frames and GT rows [cx,cy,depth, J x (u,v,dz), J x vis] with the statistics in BASELINE.md.
dataset_type = 'SyntheticPoseDataset'
num_joints = 15
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
data = dict(
    samples_per_gpu=4,
    workers_per_gpu=4,
    train=dict(type=dataset_type, num_joints=num_joints, img_shape=(512, 832), length=4096, seed=0),
    val=dict(type=dataset_type, num_joints=num_joints, img_shape=(512, 832), length=64, seed=1, test_mode=True),
    test=dict(type=dataset_type, num_joints=num_joints, img_shape=(512, 832), length=64, seed=1, test_mode=True))
evaluation = dict(interval=1)
