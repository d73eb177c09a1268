"""Dev tool: the B = 8 one-stage INFERENCE step (forward + decode, BASELINE configs[1]) under dispatch variations, ONE
process: for each `key=value[,key=value...]` argument the tuning is set, 3 warm-up + N timed steps run eagerly, the tuning
is reset; `-r R` interleaves R rounds of all arms (defaults first in every round) and prints median and range of the
per-round medians. Special keys: DCNF=<0|1> (das_amd.autograd.DCN_FUSED), GRAPH=1 (replay the forward as one hipGraph).
usage: tune_infer.py [-n steps] [-r rounds] cfg1 cfg2 ..."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from das_amd import _lib, autograd as ag, graphs  # noqa: E402
from das_amd.datasets import SyntheticPoseDataset, collate  # noqa: E402

args = sys.argv[1:]
N, R = 20, 3
while args and args[0] in ('-n', '-r'):
    if args[0] == '-n':
        N = int(args[1])
    else:
        R = int(args[1])
    args = args[2:]
lib = _lib.load()
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=1, train=False)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=8, seed=0)
data = collate([ds[i] for i in range(8)], device=dev)
bench.calibrate_scores(model, data['img'], data['img_metas'])


def step():
    return model(data['img'], data['img_metas'], return_loss=False, rescale=True)


def run(cfg):
    _lib.check(lib.das_tuning_reset(), 'reset')
    graph = False
    for kv in [c for c in cfg.split(',') if c]:
        k, v = kv.split('=')
        if k == 'DCNF':
            ag.DCN_FUSED = bool(int(v))
        elif k == 'GRAPH':
            graph = bool(int(v))
        else:
            _lib.check(lib.das_tuning_set(k.encode(), int(v)), k)
    if graph:
        graphs.enable_inference_graph(model, data['img'])
    with torch.no_grad():
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
        evs[0].record()
        for i in range(N):
            step()
            evs[i + 1].record()
        torch.cuda.synchronize()
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(N))
    if graph:
        model._graphed_infer = None
    ag.DCN_FUSED = True
    _lib.check(lib.das_tuning_reset(), 'reset')
    return ts[N // 2]


arms = [''] + args
res = {a: [] for a in arms}
for _ in range(R):
    for a in arms:
        res[a].append(run(a))
base = statistics.median(res[''])
for a in arms:
    m = statistics.median(res[a])
    print(f'{(a or "defaults"):60s} median {m:7.3f} ms  ({m - base:+.3f})  range {min(res[a]):.3f} .. {max(res[a]):.3f}', flush=True)
