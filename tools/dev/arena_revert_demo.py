"""Dev only: shows that tests/test_full_width_gpu.py's BatchNorm-statistics tests FAIL when the two-buffer fix of the zero
arena (das_amd.nn._ZeroArena, commit 6237e9b) is reverted to the one-buffer form of rounds 3-4 (refill in place).
    python tools/dev/arena_revert_demo.py > profiles/r06_bn_stats_test_catches_arena_bug.txt
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from das_amd import nn as dnn  # noqa: E402


def take_one_buffer(self, n, device):
    """rounds 3-4: ONE buffer, zeroed in place when used up — slices handed out earlier and not consumed yet lose their sums"""
    n = (n + 63) // 64 * 64
    if self.bufs is None or self.bufs[0].device != device:
        self.bufs = [torch.zeros(self.cap, dtype=torch.float32, device=device)]
        self.cur, self.off = 0, 0
    if self.off + n > self.cap:
        self.bufs[0].zero_()
        self.off = 0
    s = self.bufs[0][self.off:self.off + n]
    self.off += n
    return s


if __name__ == '__main__':
    tests = ['tests/test_full_width_gpu.py::test_batchnorm_statistics_survive_arena_wraps_inside_one_step']
    print('=== with the shipped two-buffer arena')
    rc0 = pytest.main(['-q', '-x', '-s'] + tests)
    print('=== with the one-buffer arena of rounds 3-4 (fix of commit 6237e9b reverted by monkeypatch)')
    dnn._ZeroArena.take = take_one_buffer
    rc1 = pytest.main(['-q', '-x', '-s'] + tests)
    print(f'exit codes: shipped {int(rc0)}, reverted {int(rc1)} (expected 0 and 1)')
    sys.exit(0 if (rc0 == 0 and rc1 != 0) else 1)
