"""Dev tool: where does the host time of one training sample go? cProfile over dataset[i] of the loader bench's
CMUPanopticDataset + train pipeline (GPU launches are asynchronous: what shows up is Python / numpy / PIL / copies)."""
import cProfile, io, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools', 'dev'))
import numpy as np
import torch
import loader_bench as LB
from das_amd.datasets import build_dataset

root = tempfile.mkdtemp(prefix='das_pp_')
LB.make_tree(root, 32)
ds = build_dataset(LB.dataset_cfg(root))
np.random.seed(0)
for i in range(8):
    ds[i]
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
n = 0
for i in range(64):
    if ds[i % len(ds)] is not None:
        n += 1
pr.disable()
torch.cuda.synchronize()
print(f'{(time.perf_counter() - t0) / 64 * 1e3:.2f} ms per sample (64 samples, {n} kept)')
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
