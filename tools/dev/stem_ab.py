"""Dev only: the 7 x 7 stride-2 stem conv (8 stored channels -> 64, bf16) on conv_stem7x7_kernel and on the generic conv_reg_kernel,
B x 512 x 832 frames, training forward (statistics) and eval forward (folded BatchNorm + ReLU). Interleaved rounds, median (min) us;
cold = a 600 MB fill between the timed launches (operands out of the 256 MiB cache, as inside the step).
    python tools/dev/stem_ab.py > profiles/r06_stem_ab.md"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from das_amd import _lib as _l, ops  # noqa: E402

if os.environ.get('DASLIB'):   # dev: a variant build of the library (conv_stem.hip's DAS_STEM_VAR)
    _l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), os.environ['DASLIB'])

DEV, BF = 'cuda', torch.bfloat16


def timed(fn, cold):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        if cold is not None:
            cold.fill_(1.0)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


def main():
    torch.manual_seed(0)
    print('| frames | mode | operands | conv_reg_kernel us (min) | conv_stem7x7_kernel us (min) | ratio | HBM floor us (6.2 TB/s) |')
    print('|---|---|---|---|---|---|---|')
    cold = torch.empty(300 << 20, dtype=BF, device=DEV)
    for B in (16, 8):
        x = torch.zeros(B, 512, 832, 8, dtype=BF, device=DEV)
        x[..., :3] = torch.randn(B, 512, 832, 3, device=DEV).to(BF)
        w8 = torch.zeros(64, 8, 7, 7, device=DEV)
        w8[:, :3] = torch.randn(64, 3, 7, 7, device=DEV) / 12
        w = ops.pack_weight(w8, BF)
        y = torch.empty(B, 256, 416, 64, dtype=BF, device=DEV)
        stats = torch.zeros(16 * 128, device=DEV)
        sc, sh = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)
        floor = (x.numel() + y.numel()) * 2 / 6.2e6
        for mode, fn in (('train: output + statistics', lambda: ops.conv2d(x, w, 7, 7, 2, 3, stats=stats, out=y)),
                         ('eval: scale / shift / ReLU', lambda: ops.conv2d(x, w, 7, 7, 2, 3, scale=sc, shift=sh, relu=True, out=y))):
            for cname, c in (('warm', None), ('cold', cold)):
                res = {0: [], 1: []}
                for _ in range(5):
                    for arm in (0, 1):
                        with ops.tuning(**{'conv.stem7x7': arm}):
                            fn()
                            res[arm].append(timed(fn, c))
                m0, m1 = statistics.median(res[0]), statistics.median(res[1])
                print(f'| {B} x 512 x 832 | {mode} | {cname} | {m0:.1f} ({min(res[0]):.1f}) | {m1:.1f} ({min(res[1]):.1f}) | {m0 / m1:.2f} | {floor:.1f} |',
                      flush=True)


if __name__ == '__main__':
    main()
