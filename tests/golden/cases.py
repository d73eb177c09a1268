"""Seeded input builders shared by make_golden.py (reference side) and the tests (oracle /
HIP side). Inputs and weights are regenerated from numpy RandomState seeds (stable across
numpy versions), so the committed .npz fixtures only hold the reference's OUTPUTS plus the
state-dict key/shape manifest."""
import zlib

import numpy as np
import torch

J = 3
HEAD_CFG = dict(num_joints=J, root_idx=1, depth_factor=20, z_norm=50, strides=[8, 16], stacked_convs=2,
                num_heads=4, num_layers=2, regress_ranges=((-1, 80), (80, 1e8)),
                code_weight=[1.0, 1.0, 1] + [2] * J * 6, prev_loss=True, feat_channels=32)
TEST_CFG = dict(nms_pre=50, nms_post=100, nms_thr=0.9, score_thr=0.07)
HEAD_SIZES = [(16, 24), (8, 12)]

# exp_mupots.py head topology (configs/das/exp_mupots.py:7,33-46): J = 21, root 14, depth_factor 1, two
# recursive-update layers, four levels (tiny widths / maps)
MUPOTS_J = 21
MUPOTS_CFG = dict(num_joints=MUPOTS_J, root_idx=14, depth_factor=1, z_norm=50, strides=[8, 16, 32, 64], stacked_convs=2,
                  num_heads=4, num_layers=2, regress_ranges=((-1, 80), (80, 160), (160, 320), (320, 1e8)),
                  code_weight=[1.0, 1.0, 1] + [2] * MUPOTS_J * 6, prev_loss=True, feat_channels=32)
MUPOTS_SIZES = [(16, 24), (8, 12), (4, 6), (2, 3)]

FULL_J = 15
FULL_SIZES = [(64, 104), (32, 52), (16, 26), (8, 13)]
FULL_STRIDES = [8, 16, 32, 64]
FULL_TEST_CFG = dict(nms_pre=1000, nms_post=100, nms_thr=0.9, score_thr=0.07)


def det_fill(sd, seed=0, scale=None):
    """Deterministic fill keyed by crc32(key name): identical on every platform."""
    for k, v in sd.items():
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(k.encode())) % (2 ** 31))
        shp = tuple(v.shape)
        if k.endswith('num_batches_tracked'):
            v.zero_()
        elif k.endswith('.mask'):
            continue
        elif k.endswith('running_var'):
            v.copy_(torch.from_numpy(rs.uniform(0.5, 1.5, shp).astype(np.float32)))
        elif k.endswith('running_mean'):
            v.copy_(torch.from_numpy((0.1 * rs.standard_normal(shp)).astype(np.float32)))
        elif k.endswith('.scale'):
            v.copy_(torch.tensor(float(rs.uniform(0.8, 1.2))))
        elif v.dim() <= 1:
            if ('bn' in k or 'gn' in k) and k.endswith('weight'):
                v.copy_(torch.from_numpy(rs.uniform(0.5, 1.5, shp).astype(np.float32)))
            else:
                v.copy_(torch.from_numpy((0.1 * rs.standard_normal(shp)).astype(np.float32)))
        else:
            fan_in = int(np.prod(shp[1:]))
            s = (1.0 / fan_in) ** 0.5 if scale is None else scale
            v.copy_(torch.from_numpy((s * rs.standard_normal(shp)).astype(np.float32)))
    return sd


def sd_from_manifest(keys, shapes, dtypes, seed, masks=None):
    """Rebuild a state dict from the manifest stored in a fixture."""
    sd = {}
    for k, s, d in zip(keys, shapes, dtypes):
        sd[str(k)] = torch.zeros(tuple(int(i) for i in s), dtype=getattr(torch, str(d)))
    det_fill(sd, seed)
    for k in sd:
        if k.endswith('.mask'):
            d = sd[k].shape[1]
            pat = [[0, 0, 1], [1, 1, 0]] if d == 3 else [[0, 1], [1, 0]]
            sd[k] = torch.tensor(pat * 3, dtype=torch.float32)
    return sd


def manifest(sd):
    keys = np.array(list(sd.keys()))
    shapes = np.array([np.array(list(v.shape) + [-1] * (4 - v.dim())) for v in sd.values()])
    dtypes = np.array([str(v.dtype).replace('torch.', '') for v in sd.values()])
    return dict(sd_keys=keys, sd_shapes=shapes, sd_dtypes=dtypes)


def manifest_shapes(z):
    return [[int(i) for i in row if i >= 0] for row in z['sd_shapes']]


def randn(seed, *shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def make_gt(rs, G, J, W, H, two_d_first=False, spread=20.0):
    """GT rows [cx,cy,depth, J x (u,v,dz), J x vis] (cmupanoptic_mono_dataset.py:218-222)."""
    c = np.stack([rs.uniform(10, W - 10, G), rs.uniform(10, H - 10, G)], 1)
    depth = rs.uniform(0.2, 0.7, G)
    uv = c[:, None] + rs.normal(0, spread, (G, J, 2))
    dz = rs.normal(0, 20, (G, J, 1))
    dz[:, 1] = 0
    if two_d_first and G > 0:
        dz[0] = 0
    uvd = np.concatenate([uv, dz], -1).reshape(G, 3 * J)
    vis = (rs.uniform(0, 1, (G, J)) > 0.2).astype(np.float32)
    g = np.concatenate([c, depth[:, None], uvd, vis], 1).astype(np.float32)
    return (torch.from_numpy(g), torch.from_numpy(c.astype(np.float32)),
            torch.from_numpy(depth.astype(np.float32)))


def head_gts(counts=(3, 0), seed=5):
    rs = np.random.RandomState(seed)
    W, H = HEAD_SIZES[0][1] * 8, HEAD_SIZES[0][0] * 8
    g = [make_gt(rs, n, J, W, H, two_d_first=(i == 0)) for i, n in enumerate(counts)]
    poses, c2d, dep = [x[0] for x in g], [x[1] for x in g], [x[2] for x in g]
    labels = [torch.zeros(len(x), dtype=torch.long) for x in poses]
    return dict(gt_labels_3d=labels, gt_poses_3d=poses, centers2d=c2d, depths=dep)


def head_feats(seed=11, B=2, C=32, sizes=None):
    return [randn(seed + i, B, C, h, w) for i, (h, w) in enumerate(sizes or HEAD_SIZES)]


def mupots_gts(counts=(2, 3), seed=6):
    rs = np.random.RandomState(seed)
    W, H = MUPOTS_SIZES[0][1] * 8, MUPOTS_SIZES[0][0] * 8
    g = [make_gt(rs, n, MUPOTS_J, W, H, two_d_first=(i == 1)) for i, n in enumerate(counts)]
    for x in g:      # (make_gt pins dz of joint 1; the mupots root is joint 14)
        x[0][:, 3 + 3 * 14 + 2] = 0
    poses, c2d, dep = [x[0] for x in g], [x[1] for x in g], [x[2] for x in g]
    labels = [torch.zeros(len(x), dtype=torch.long) for x in poses]
    return dict(gt_labels_3d=labels, gt_poses_3d=poses, centers2d=c2d, depths=dep)


# soft OKS-NMS cases (nms_type != 'hard'): the default threshold, and a low one (strong rescoring: the order of the
# remaining candidates changes from round to round) with a small cap
SOFT_DECODE_CFGS = {
    'a': dict(FULL_TEST_CFG, nms_type='soft'),
    'b': dict(FULL_TEST_CFG, nms_type='soft', nms_thr=0.05, nms_post=20),
}
SOFT_DECODE_INPUTS = {'a': dict(), 'b': dict(seed=91, bias=-3.5)}


def full_decode_inputs(seed=21, B=2, Jn=FULL_J, sizes=FULL_SIZES, bias=-4.0):
    """Tie-free random eval-mode head outputs at the full 512x832 level sizes."""
    cls, pose, ctr = [], [], []
    for i, (h, w) in enumerate(sizes):
        cls.append(randn(seed + 10 * i, B, 1, h, w) + bias)
        ctr.append(randn(seed + 10 * i + 1, B, 1, h, w) + 0.5)
        p = randn(seed + 10 * i + 2, B, 3 + 6 * Jn, h, w)
        p[:, 3:3 + 3 * Jn] *= 30.0
        p[:, 2] = p[:, 2].abs() * 0.2 + 0.2
        pose.append(p)
    return cls, pose, ctr
