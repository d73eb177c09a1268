"""SyncBN arithmetic: two emulated ranks (the two halves of a batch, statistics summed by hand where the
process group would all-reduce them) reproduce full-batch BatchNorm forward and backward. GPU only; the
real multi-process path uses torch.distributed.all_reduce at exactly the points emulated here
(`das_amd/autograd.py::ConvBNTrainFn`)."""
import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def nhwc(t, dtype=torch.float32):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def stats_of(raw):
    f = raw.float()
    return torch.cat([f.sum((0, 1, 2)), f.square().sum((0, 1, 2))]).contiguous()


@pytest.mark.parametrize('relu,res', [(True, False), (True, True), (False, False)])
def test_two_emulated_ranks_equal_full_batch(relu, res):
    from das_amd import ops as o
    B, H, W, C = 4, 9, 7, 64
    raw = nhwc(cases.randn(1, B, C, H, W) * 1.3 + 0.2)
    dy = nhwc(cases.randn(2, B, C, H, W))
    r = nhwc(cases.randn(3, B, C, H, W)) if res else None
    gamma, beta = (cases.randn(4, C).abs() + 0.5).to(DEV), cases.randn(5, C).to(DEV)
    rows = B * H * W

    # full batch, one rank
    y, mean, invstd = o.bn_train_apply(raw, stats_of(raw), gamma, beta, None, None, 0.1, 1e-5, residual=r, relu=relu)
    d_full, dres_full, dg_full, db_full = o.bn_train_backward(dy, y if (relu and res) else None, raw, mean, invstd, gamma,
                                                              relu, res, beta=beta)

    halves = [slice(0, 2), slice(2, 4)]
    st = [stats_of(raw[h]) for h in halves]
    total = st[0] + st[1]                                 # = all_reduce(stats)
    ys, outs = [], []
    for h in halves:
        yh, mh, ih = o.bn_train_apply(raw[h].contiguous(), total.clone(), gamma, beta, None, None, 0.1, 1e-5,
                                      residual=r[h].contiguous() if res else None, relu=relu, stat_count=rows)
        torch.testing.assert_close(mh, mean, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(ih, invstd, rtol=1e-5, atol=1e-6)
        ys.append(yh)
    torch.testing.assert_close(torch.cat(ys), y, rtol=1e-6, atol=1e-6)

    # backward: local sums of the other rank, obtained from a local (unsynchronised) call
    local = []
    for h, yh in zip(halves, ys):
        _, _, dg, db = o.bn_train_backward(dy[h].contiguous(), yh if (relu and res) else None, raw[h].contiguous(), mean,
                                           invstd, gamma, relu, res, beta=beta)
        local.append(torch.cat([db, dg]))                 # layout of `sums`: [sum dz, sum dz*xhat]
    for i, (h, yh) in enumerate(zip(halves, ys)):
        other = local[1 - i]

        def all_reduce(t, other=other):
            t += other

        d, dres, dg, db = o.bn_train_backward_sync(dy[h].contiguous(), yh if (relu and res) else None,
                                                   raw[h].contiguous(), mean, invstd, gamma, relu, res, beta,
                                                   all_reduce, 2)
        outs.append((d, dres, dg, db))
        if relu and res:
            # the same two-phase backward with the ReLU mask handed over as bits (one byte per 16-byte vector of y) instead of y
            per = 16 // yh.element_size()
            wts = (2 ** torch.arange(per, device=yh.device)).to(torch.int32)
            bits = ((yh.reshape(-1, per) > 0).to(torch.int32) * wts).sum(1).to(torch.uint8)
            d2, dres2, dg2, db2 = o.bn_train_backward_sync(dy[h].contiguous(), None, raw[h].contiguous(), mean, invstd, gamma,
                                                           relu, res, beta, all_reduce, 2, bits=bits)
            assert torch.equal(dres2, dres)
            torch.testing.assert_close(d2, d, rtol=1e-5, atol=2e-6)
            torch.testing.assert_close(dg2, dg, rtol=1e-5, atol=1e-4)
            torch.testing.assert_close(db2, db, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(torch.cat([a[0] for a in outs]), d_full, rtol=1e-5, atol=2e-6)
    if res:
        torch.testing.assert_close(torch.cat([a[1] for a in outs]), dres_full, rtol=0, atol=0)
    # the ranks' local parameter gradients add up to the full-batch ones (the gradient all-reduce does that)
    torch.testing.assert_close(outs[0][2] + outs[1][2], dg_full, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(outs[0][3] + outs[1][3], db_full, rtol=1e-5, atol=1e-4)


def test_row_count_check_is_looked_at_lazily_and_raises_when_its_answer_says_so():
    """autograd.verify_rows: the answer of the per-step row-count all-reduce reaches the host through page-locked memory
    and an event; it is looked at once it HAS arrived (never waited for inside the forward pass) and at the latest
    ROWS_CHECK_LAG steps late."""
    from das_amd import autograd as ag
    ag._rows_pending.clear()

    def push(hi, lo, dev=None):
        if dev is None:
            dev = torch.tensor([hi, lo], dtype=torch.float64, device='cuda')
        host = torch.empty(2, dtype=torch.float64, pin_memory=True)
        host.copy_(dev, non_blocking=True)
        ev = torch.cuda.Event(blocking=True)
        ev.record()
        ag._rows_pending.append((host, ev, int(hi)))
    push(128.0, -128.0)
    torch.cuda.synchronize()
    ag.verify_rows()
    assert not ag._rows_pending
    push(144.0, -128.0)
    with pytest.raises(RuntimeError, match='between 128 and 144'):
        ag.verify_rows(wait=True)
    assert not ag._rows_pending
    # an answer that has not arrived is left alone ... (a long kernel in front of the copy)
    from das_amd import _lib
    import ctypes
    # (the device-side values exist before the long kernel: creating them behind it would block the host until it has finished)
    devs = [torch.tensor([64.0, -64.0], dtype=torch.float64, device='cuda') for _ in range(ag.ROWS_CHECK_LAG + 1)]
    torch.cuda.synchronize()
    _lib.check(_lib.load().das_dev_occupy_cus(1, 64, 1024, 300000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'occupy')
    push(64.0, -64.0, devs[0])
    ag.verify_rows()
    assert len(ag._rows_pending) == 1
    # ... unless more than ROWS_CHECK_LAG are queued
    for i in range(ag.ROWS_CHECK_LAG):
        push(64.0, -64.0, devs[1 + i])
    ag.verify_rows()
    assert len(ag._rows_pending) <= ag.ROWS_CHECK_LAG
    ag.verify_rows(wait=True)
    assert not ag._rows_pending


def test_a_row_count_mismatch_in_the_last_step_is_reported_before_weights_are_saved():
    """the row-count answer of a step is read up to ROWS_CHECK_LAG steps later; a mismatch in the LAST steps before a
    checkpoint / evaluation / the end of the run must still raise — optim.finish_checks() (called by tools/train.py at every
    epoch end, before the checkpoint is written, and at the end of the run) waits for whatever is still in flight."""
    from das_amd import autograd as ag
    from das_amd.optim import finish_checks
    ag._rows_pending.clear()
    dev = torch.tensor([26624.0, -13312.0], dtype=torch.float64, device='cuda')     # (a short last batch on one rank)
    host = torch.empty(2, dtype=torch.float64, pin_memory=True)
    host.copy_(dev, non_blocking=True)
    ev = torch.cuda.Event(blocking=True)
    ev.record()
    ag._rows_pending.append((host, ev, 26624))
    with pytest.raises(RuntimeError, match='different numbers of pixel rows'):
        finish_checks()
    assert not ag._rows_pending
    finish_checks()          # nothing pending: a no-op
