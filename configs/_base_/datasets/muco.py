# MuCo-3DHP (train) / MuPoTS-3D (test) topology: 21 joints. Synthetic stand-in, see panoptic_monocular.py.
dataset_type = 'SyntheticPoseDataset'
num_joints = 21
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
data = dict(
    samples_per_gpu=4,
    workers_per_gpu=4,
    train=dict(type=dataset_type, num_joints=num_joints, img_shape=(512, 832), length=4096, seed=0),
    val=dict(type=dataset_type, num_joints=num_joints, img_shape=(768, 1024), length=64, seed=1, test_mode=True),
    test=dict(type=dataset_type, num_joints=num_joints, img_shape=(768, 1024), length=64, seed=1, test_mode=True))
evaluation = dict(interval=1)
