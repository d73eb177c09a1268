"""Dev tool: is the one-rank RCCL step the plain step? Plain twice (run-to-run floor), forced collectives with and
without overlap."""
import os, sys
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import das_amd
from das_amd.datasets import SyntheticPoseDataset, collate
from das_amd.optim import FlatSGD, train_iteration
from test_model_gpu import tiny_detector_cfg
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='34567')
torch.cuda.set_device(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def run(force, overlap=True):
    torch.manual_seed(0)
    cfg = tiny_detector_cfg()
    cfg['backbone']['compute_dtype'] = 'f32'
    model = das_amd.build_model(cfg)
    model.init_weights()
    model.to('cuda').train()
    opt = FlatSGD(model, lr=2e-3, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0,
                  max_grad_norm=35.0, bucket_mb=1, overlap=overlap, force_collectives=force)
    ds = SyntheticPoseDataset(num_joints=15, img_shape=(128, 192), length=4, seed=3, max_persons=3)
    data = collate([ds[i] for i in range(2)], device='cuda')
    gs = []
    for _ in range(steps):
        train_iteration(model, opt, data, 2e-3)
        gs.append(opt.flat_g.detach().clone())
    torch.cuda.synchronize()
    return opt.flat_p.detach().clone(), gs


def cmp(name, a, b):
    (pa, ga), (pb, gb) = a, b
    print(f'{name}: params max abs diff {float((pa - pb).abs().max()):.3e}; per-step grad max abs diff',
          [f'{float((x - y).abs().max()):.3e}' for x, y in zip(ga, gb)], ' grad max', f'{float(ga[0].abs().max()):.3e}', flush=True)


r1, r2 = run(False), run(False)
cmp('plain vs plain', r1, r2)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
f1 = run(True)
cmp('plain vs rccl overlapped', r1, f1)
f2 = run(True, overlap=False)
cmp('plain vs rccl at the end', r1, f2)
f3 = run(True)
cmp('rccl overlapped vs rccl overlapped', f1, f3)
dist.destroy_process_group()
