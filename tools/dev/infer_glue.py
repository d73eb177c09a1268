"""Dev tool: torch.profiler over one infer bench step: the aten ops (copies, fills, small elementwise launches) by
count and by the python source line that issued them."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
from das_amd.datasets import SyntheticPoseDataset, collate

B = 8
dev = torch.device('cuda', 0)
model = bench.build_model(dev, num_stages=1, train=False)
ds = SyntheticPoseDataset(num_joints=bench.J, img_shape=(bench.H, bench.W), length=B, seed=0)
data = collate([ds[i] for i in range(B)], device=dev)
metas = data['img_metas']
bench.calibrate_scores(model, data['img'], metas)
for _ in range(3):
    model(data['img'], metas, return_loss=False, rescale=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    model(data['img'], metas, return_loss=False, rescale=True)
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::cat', 'aten::_to_copy', 'aten::contiguous', 'aten::clone', 'aten::mul', 'aten::add', 'aten::sigmoid', 'aten::rsqrt'):
        by[(ev.name, str(ev.input_shapes)[:100])] += 1
for (name, shp), n in by.most_common(70):
    print(f'{n:4d}  {name:18s} {shp}')
