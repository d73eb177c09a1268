"""das_amd.loader.PrefetchLoader on the CPU (plain-Python samples): batch order, re-draw of dropped samples, exception
hand-over, bounded look-ahead — the host logic tools/train.py relies on (the reference's dataloader workers:
configs/das/exp_panoptic.py:159-160)."""
import threading
import time

import pytest

from das_amd.loader import PrefetchLoader


class Toy:
    def __init__(self, n, drop=(), fail=(), delay=None):
        self.n, self.drop, self.fail, self.delay = n, set(drop), set(fail), delay or {}
        self.calls, self.lock = [], threading.Lock()

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        with self.lock:
            self.calls.append(i)
        time.sleep(self.delay.get(i, 0.0))
        if i in self.fail:
            raise ValueError(f'sample {i}')
        return None if i in self.drop else dict(idx=i)


def keep(samples, device=None):
    return [s['idx'] for s in samples]


@pytest.mark.parametrize('workers', [0, 1, 3])
def test_batches_arrive_in_order_whatever_the_worker_timing(workers):
    ds = Toy(12, delay={0: 0.05, 5: 0.03})
    batches = [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [10, 11]]
    assert list(PrefetchLoader(ds, batches, keep, device='cpu', workers=workers)) == batches
    assert sorted(ds.calls) == list(range(12))


@pytest.mark.parametrize('workers', [0, 2])
def test_dropped_samples_are_redrawn_from_the_following_indices(workers):
    ds = Toy(6, drop={1, 2, 5})
    out = list(PrefetchLoader(ds, [[0, 1], [4, 5]], keep, device='cpu', workers=workers))
    assert out == [[0, 3], [4, 0]]          # 1 -> 2 -> 3; 5 wraps around to 0


@pytest.mark.parametrize('workers', [0, 2])
def test_a_worker_exception_surfaces_at_its_batch(workers):
    ds = Toy(8, fail={5})
    it = iter(PrefetchLoader(ds, [[0, 1], [2, 3], [4, 5], [6, 7]], keep, device='cpu', workers=workers))
    assert next(it) == [0, 1] and next(it) == [2, 3]
    with pytest.raises(ValueError, match='sample 5'):
        next(it)


def test_look_ahead_is_bounded():
    ds = Toy(40)
    it = iter(PrefetchLoader(ds, [[i] for i in range(40)], keep, device='cpu', workers=2, depth=2))
    assert next(it) == [0]
    time.sleep(0.2)
    assert len(ds.calls) <= 1 + 2 + 2 + 1     # consumed + workers + depth (+ the permit the consumer just returned)
    assert [b for b in it] == [[i] for i in range(1, 40)]


def test_pack_to_device_keeps_shapes_dtypes_and_values():
    """One buffer, one copy: the packed views must be the arrays (mixed dtypes, odd sizes, an empty one in the middle)."""
    import numpy as np
    from das_amd.datasets import pack_to_device
    rs = np.random.RandomState(0)
    arrays = [rs.randn(3, 15, 4).astype(np.float32), np.arange(3, dtype=np.int64), np.zeros((0, 4), np.float32),
              rs.randn(5).astype(np.float64), rs.randint(0, 9, (2, 7)).astype(np.int32), rs.randn(1, 1).astype(np.float32)]
    out = pack_to_device(arrays, 'cpu')
    assert len(out) == len(arrays)
    for a, t in zip(arrays, out):
        assert tuple(t.shape) == a.shape and t.numpy().dtype == a.dtype
        np.testing.assert_array_equal(t.numpy(), a)
