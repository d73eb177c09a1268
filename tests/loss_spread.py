"""Measurement helper of the test suite (it runs the oracle, so it lives under tests/): how sharp can a train-mode loss
comparison at full width be? For the 1-stage and the 4-stage net (B = 2,
512 x 832, the weights and frames of tests/test_full_width_gpu.py): the CPU oracle in f64 and f32, then the HIP path n
times in f32 and n times in bf16 on the same weights and frames (running statistics restored before every run).
The statistics of a train-mode BatchNorm are summed with float atomics (the order varies run to run) and ~50 / ~200
such layers over two frames amplify last-bit differences: the run-to-run spread printed here is the yardstick
behind that test's bands."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import bench  # noqa: E402
from das_amd.datasets import SyntheticPoseDataset, collate  # noqa: E402
from oracle import backbone as ob, head as oh, loss as ol  # noqa: E402  (the oracle as the checker)
from test_full_width_gpu import build, oracle_hcfg, split_sd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
KEYS = ('loss_cls', 'loss_depth', 'loss_pose', 'loss_centerness')
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=2, seed=0)
ss = [ds[i] for i in range(2)]
img = torch.stack([s['img'] for s in ss])
gts = {k: [s[k] for s in ss] for k in ('gt_labels_3d', 'gt_poses_3d', 'centers2d', 'depths')}
for stages in (1, 4):
    cfg = bench.model_cfg(stages, 'f32')
    hcfg = oracle_hcfg(cfg)
    model = build(cfg, 0)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    ref = {}
    for dt in (torch.float64, torch.float32):
        bsd, nsd, hsd = [{k: (v.to(dt) if v.is_floating_point() else v) for k, v in d.items()} for d in split_sd(model)]
        g = {k: [t.to(dt) if t.is_floating_point() else t for t in v] for k, v in gts.items()}
        with torch.no_grad():
            feats = ob.fpn_forward(nsd, ob.mspn2_forward(bsd, img.to(dt), stages, (3, 4, 6, 3), train=True), train=True)
            outs = oh.head_forward(hsd, feats, hcfg, '', True)
            ref[dt] = {k: float(v) for k, v in ol.head_loss(hsd, '', *outs, g, hcfg).items()}
    for k in KEYS:
        t, f = ref[torch.float64][k], ref[torch.float32][k]
        print(f'{stages} stage(s) oracle {k:16s} f64 {t:12.5f}  f32 {f:12.5f}  ({abs(f - t) / abs(t) * 100:.4f} % apart)', flush=True)
    data = collate(ss, device='cuda')
    for dt in ('f32', 'bf16'):
        m = build(bench.model_cfg(stages, dt), 0)
        m.load_state_dict(sd0)
        m.to('cuda').train()
        rows = []
        for _ in range(n):
            m.load_state_dict(sd0)
            with torch.no_grad():
                rows.append({k: float(v) for k, v in m.train_step(data)['log_vars'].items()})
        for k in KEYS:
            v = torch.tensor([r[k] for r in rows], dtype=torch.float64)
            t = ref[torch.float64][k]
            print(f'{stages} stage(s) hip {dt:4s} {k:16s} min {v.min():11.5f}  max {v.max():11.5f}  spread '
                  f'{(v.max() - v.min()) / abs(t) * 100:7.4f} %  farthest from f64 {(v - t).abs().max() / abs(t) * 100:7.4f} %', flush=True)
