"""Dev probe: what does a RARE host stall cost the step, with and without the ground truth's upload events (das_amd.datasets.
mark_uploaded)? Every fifth step the host sleeps S ms right before it queues the head; mean GPU step time over 30 steps,
interleaved repetitions. Without the events the detector's side stream (target assignment, three counts read back) waits for the
whole previous step: the host runs in lockstep with the GPU and has ~20 ms of slack; with them it runs ahead until the launch
queue is full (1-2 steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration

S = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=4, train=True)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=16, seed=0)
data = collate([ds[i] for i in range(16)], device=dev)
opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0, max_grad_norm=35.0)
ev0 = data['gt_poses_3d'][0]._das_uploaded
cnt, sleep_ms = [0], [0.0]
head_ft = model.bbox_head.forward_train


def slow_head(*a, **k):
    cnt[0] += 1
    if sleep_ms[0] > 0 and cnt[0] % 5 == 0:
        time.sleep(sleep_ms[0] * 1e-3)
    return head_ft(*a, **k)


model.bbox_head.forward_train = slow_head


def run(tagged, stall, n=30):
    for key in ('gt_poses_3d', 'centers2d', 'depths'):
        for t in data[key]:
            if tagged:
                t._das_uploaded = ev0
            elif hasattr(t, '_das_uploaded'):
                del t._das_uploaded
    sleep_ms[0] = stall
    for _ in range(3):
        train_iteration(model, opt, data, 2e-3)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        train_iteration(model, opt, data, 2e-3)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for _ in range(4):
    train_iteration(model, opt, data, 2e-3)
res = {}
for rep in range(3):
    for tagged in (True, False):
        for stall in (0.0, S):
            res.setdefault((tagged, stall), []).append(run(tagged, stall))
for (tagged, stall), v in sorted(res.items(), reverse=True):
    print('upload events %-3s  a %3.0f ms host stall every 5th step: mean step %.2f ms  (%s)' % ('on' if tagged else 'off', stall, sum(v) / len(v), ' '.join('%.2f' % x for x in v)))
