"""What the GPU box's host offers the CPU baseline: logical CPUs, cgroup quota, and how a conv scales with threads."""
import os
import subprocess
import time

import torch
import torch.nn.functional as F

print('affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, 'n/a')
print(subprocess.run('lscpu | head -20; free -g | head -2', shell=True, capture_output=True, text=True).stdout)
x = torch.randn(2, 256, 128, 208)
w = torch.randn(256, 256, 3, 3)
for n in (8, 16, 32, 64, 128, 256):
    if n > len(os.sched_getaffinity(0)):
        break
    torch.set_num_threads(n)
    F.conv2d(x, w, padding=1)
    t0 = time.perf_counter()
    for _ in range(3):
        F.conv2d(x, w, padding=1)
    dt = (time.perf_counter() - t0) / 3
    print(f'threads {n}: conv 3x3 256->256 @2x128x208 {dt * 1e3:.1f} ms = {2 * 2 * 128 * 208 * 256 * 2304 / dt / 1e9:.0f} GFLOP/s', flush=True)
