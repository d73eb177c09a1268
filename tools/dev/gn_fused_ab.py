"""Dev only: conv 3x3 256 -> 256 over the five FPN levels (+ bias) followed by GroupNorm(32) + ReLU, with the GroupNorm sums taken in
the conv's epilogue (DasConvDesc.gn_sums) or by das_groupnorm_nhwc's own statistics pass. Per-op times (median of interleaved
rounds, cold operands: a 600 MB fill between the timed launches).
    python tools/dev/gn_fused_ab.py > profiles/r06_gn_fused_ab.md"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from das_amd import ops  # noqa: E402

DEV, BF = 'cuda', torch.bfloat16
SIZES = [(128, 208), (64, 104), (32, 52), (16, 26), (8, 13)]


def timed(fn, cold):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        cold.fill_(1.0)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


def main():
    torch.manual_seed(0)
    cold = torch.empty(300 << 20, dtype=BF, device=DEV)
    print('| B | conv us | conv + sums us | groupnorm (stats + apply) us | groupnorm (apply only) us | pair: separate | pair: fused |')
    print('|---|---|---|---|---|---|---|')
    for B in (16, 8):
        x = ops.Ragged.from_levels([torch.randn(B, h, w, 256, device=DEV).to(BF) for h, w in SIZES[1:]] if False else
                                   [torch.randn(B, h // 2, w // 2, 256, device=DEV).to(BF) for h, w in SIZES])
        w = ops.pack_weight(torch.randn(256, 256, 3, 3, device=DEV) / 48, BF)
        bias = torch.randn(256, device=DEV)
        gamma, beta = torch.rand(256, device=DEV) + 0.5, torch.randn(256, device=DEV)
        y, out = x.new(256), x.new(256)
        n = ops.groupnorm_stats_size(x, 32)
        ws = torch.zeros(n, device=DEV)
        res = {k: [] for k in 'abcd'}
        for _ in range(5):
            res['a'].append(timed(lambda: ops.conv2d(x, w, 3, 3, 1, 1, shift=bias, out=y), cold))
            res['b'].append(timed(lambda: ops.conv2d(x, w, 3, 3, 1, 1, shift=bias, out=y, gn_sums=(ws, 32)), cold))
            ws.zero_()
            res['c'].append(timed(lambda: ops.groupnorm(y, gamma, beta, 32, out=out), cold))
            ops.conv2d(x, w, 3, 3, 1, 1, shift=bias, out=y, gn_sums=(ws, 32))
            res['d'].append(timed(lambda: ops.groupnorm(y, gamma, beta, 32, out=out, ws=ws, have_sums=True), cold))
            ws.zero_()
        m = {k: statistics.median(v) for k, v in res.items()}
        print(f"| {B} | {m['a']:.1f} | {m['b']:.1f} | {m['c']:.1f} | {m['d']:.1f} | {m['a'] + m['c']:.1f} | {m['b'] + m['d']:.1f} |", flush=True)
    print(f'\n(kernel of the conv: `{ops.last_kernel()}` is the last launch; rows per launch: B x 8840 + ...)')


if __name__ == '__main__':
    main()
