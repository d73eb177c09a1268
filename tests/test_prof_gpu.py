"""das_prof_*: the library's own launch timing (include/das_hip.h) — what bench.py's priced_step is built on.
One record per C entry point, recorded around its launches on the stream they go to; nested entry points do not record
again; off, nothing is recorded."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_one_record_per_entry_point_with_kernel_names_and_positive_times():
    from das_amd import _lib, ops
    lib = _lib.load()
    torch.manual_seed(0)
    x = torch.randn(2, 32, 48, 64, device=DEV).to(torch.bfloat16)
    w = ops.pack_weight(torch.randn(64, 64, 3, 3, device=DEV) * 0.05, torch.bfloat16)
    ops.conv2d(x, w, 3, 3, 1, 1)          # (first call outside the pass: kernel attributes)
    torch.cuda.synchronize()
    n_before = lib.das_prof_count()
    ops.conv2d(x, w, 3, 3, 1, 1)
    assert lib.das_prof_count() == n_before, 'nothing is recorded while profiling is off'

    ops.profile_begin()
    assert lib.das_prof_count() == 0
    y = ops.conv2d(x, w, 3, 3, 1, 1)
    k_conv = ops.last_kernel()
    raw = torch.randn(2 * 32 * 48, 64, device=DEV).to(torch.bfloat16).view(2, 32, 48, 64)
    dy = torch.randn_like(raw)
    mean, invstd, gamma, beta = (torch.zeros(64, device=DEV), torch.ones(64, device=DEV), torch.ones(64, device=DEV),
                                 torch.zeros(64, device=DEV))
    # das_bn_train_backward calls the phase entry points internally: still ONE record
    ops.bn_train_backward(dy, None, raw, mean, invstd, gamma, True, False, beta=beta)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):          # the events go to the stream the launch goes to
        ops.add3(raw, dy)
    torch.cuda.current_stream().wait_stream(side)
    ents = ops.profile_end()
    recs = ops.profile_records()
    assert len(recs) == 3, recs
    assert recs[0][0] == k_conv and recs[1][0].startswith('das_bn_train_backward') and recs[2][0] == 'das_add3', recs
    assert all(0.0 < ms < 50.0 for _, ms in recs), recs
    # the wrappers' entries resolve to the same records
    assert len(ents) == 2 and ents[0][0] == k_conv and abs(ents[0][2].elapsed_time() - recs[0][1]) < 1e-6
    assert abs(ents[1][2].elapsed_time() - recs[1][1]) < 1e-6
    assert float(y.float().abs().sum()) > 0
    # a second pass reuses the event pairs
    ops.profile_begin()
    ops.conv2d(x, w, 3, 3, 1, 1)
    ops.profile_end()
    assert len(ops.profile_records()) == 1


def test_span_of_a_batched_weight_gradient_covers_its_launch_and_reduce_pass():
    from das_amd import ops
    torch.manual_seed(1)
    x = torch.randn(2, 32, 48, 64, device=DEV).to(torch.bfloat16)
    dy = torch.randn(2, 32, 48, 64, device=DEV).to(torch.bfloat16)
    outs = [torch.zeros(64, 3, 3, 64, device=DEV) for _ in range(3)]
    items = [(x, dy, 3, 3, 1, 1, o) for o in outs]
    ops.conv2d_wgrad_batch(items)
    torch.cuda.synchronize()
    ops.profile_begin()
    ops.conv2d_wgrad_batch(items)
    ents = ops.profile_end()
    assert len(ents) == 1 and ents[0][5] == 3 and ents[0][2].i1 - ents[0][2].i0 == 1
    assert ents[0][2].elapsed_time() > 0
