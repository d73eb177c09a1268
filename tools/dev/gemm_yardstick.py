"""Dev only (never imported by the package): what does the VENDOR library reach on this MI355X for dense bf16 GEMMs of the
train step's heaviest (M, N, K), beside this repo's own kernels on the same problems? (VERDICT r5 #1c: a yardstick for the
tile kernels other than the 2.5 PF spec.)

  vendor   torch.matmul on bf16 operands that are ALREADY plain row-major matrices (hipBLASLt / rocBLAS behind ATen;
           `torch.backends.cuda.preferred_blas_library` is tried for both) — for a 3x3 conv this is the GEMM on a
           pre-materialised im2col matrix, i.e. WITHOUT the 9x operand traffic of building it: an upper bound for an
           implicit-GEMM conv, not a drop-in.
  das      ops.conv2d (forward / data-gradient tile kernels, im2col on the fly) or ops.conv2d_wgrad on the real geometry.

Random normal operands (zero-filled ones clock 15-20 % higher: cdna guide 5.4 rule 25). Interleaved rounds, median and min.

    python tools/dev/gemm_yardstick.py > profiles/r06_gemm_yardstick.md
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from das_amd import _lib, ops  # noqa: E402

DEV = 'cuda'
B = 16
# (tag, kind, B, H, W, Cin, Cout, k): kind 'fwd' = forward / data-gradient GEMM (M = pixels, N = Cout, K = k*k*Cin),
# 'wgrad' = weight gradient (M = k*k*Cin, N = Cout, K = pixels). Shapes: mspn_mmpose.py:81-157, das_head.py:112-161.
CASES = [
    ('head 3x3 256->256, levels 0-3 (M = 141 440)', 'fwd', B, 85, 104, 256, 256, 3),      # 8840 px / image as one 85 x 104 plane
    ('stage-3 3x3 256->256 @32x52', 'fwd', B, 32, 52, 256, 256, 3),
    ('stage-2 3x3 128->128 @64x104', 'fwd', B, 64, 104, 128, 128, 3),
    ('stage-4 3x3 512->512 @16x26', 'fwd', B, 16, 26, 512, 512, 3),
    ('1x1 1024->256 @32x52', 'fwd', B, 32, 52, 1024, 256, 1),
    ('1x1 512->2048 @16x26', 'fwd', B, 16, 26, 512, 2048, 1),
    ('1x1 2048->2048 @16x26 (skip conv)', 'fwd', B, 16, 26, 2048, 2048, 1),
    ('1x1 512->256 @64x104', 'fwd', B, 64, 104, 512, 256, 1),
    ('wgrad head 3x3 256->256 (K = 141 440)', 'wgrad', B, 85, 104, 256, 256, 3),
    ('wgrad 3x3 256->256 @32x52', 'wgrad', B, 32, 52, 256, 256, 3),
    ('wgrad 3x3 128->128 @64x104', 'wgrad', B, 64, 104, 128, 128, 3),
    ('wgrad 1x1 256->1024 @32x52', 'wgrad', B, 32, 52, 256, 1024, 1),
    ('wgrad 1x1 2048->2048 @16x26', 'wgrad', B, 16, 26, 2048, 2048, 1),
]
ROUNDS, INNER = 7, 5


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(INNER):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / INNER * 1e3     # us


def main():
    lib = _lib.load()
    torch.manual_seed(0)
    print('# Vendor-library yardstick for the tile / weight-gradient kernels (round 6, MI355X)\n')
    print(f'`python tools/dev/gemm_yardstick.py`: torch {torch.__version__}, device {torch.cuda.get_device_name(0)}; bf16, random normal '
          f'operands, {ROUNDS} interleaved rounds x {INNER} launches, median (min) per arm. `vendor` = torch.matmul on plain '
          'row-major matrices (for a 3x3 conv: on a PRE-BUILT im2col matrix — no operand expansion inside the timed call, '
          'an upper bound for an implicit-GEMM conv); `das` = this repo on the real conv geometry. TF = 2 M N K / time.\n')
    print('| problem | M | N | K | vendor us (min) | vendor TF | das us (min) | das TF | das / vendor | das kernel |')
    print('|---|---|---|---|---|---|---|---|---|---|')
    libs = []
    for name in ('hipblaslt', 'cublaslt', 'cublas'):
        try:
            torch.backends.cuda.preferred_blas_library(name)
            libs.append(name)
        except Exception:
            pass
    for tag, kind, b, H, W, Cin, Cout, k in CASES:
        rows = b * H * W
        x = torch.randn(b, H, W, Cin, device=DEV, dtype=torch.bfloat16)
        if kind == 'fwd':
            M, N, K = rows, Cout, k * k * Cin
            a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
            wt = (torch.randn(N, K, device=DEV) / K ** 0.5).to(torch.bfloat16)
            w = wt.reshape(Cout, k, k, Cin).contiguous()
            y = ops.conv2d(x, w, k, k, 1, k // 2)
            das = lambda: ops.conv2d(x, w, k, k, 1, k // 2, out=y)                       # noqa: E731
            vend = lambda: torch.matmul(a, wt.t())                                      # noqa: E731
        else:
            M, N, K = k * k * Cin, Cout, rows
            dy = torch.randn(b, H, W, Cout, device=DEV, dtype=torch.bfloat16)
            a = torch.randn(K, M, device=DEV, dtype=torch.bfloat16)      # im2col matrix (pixels x taps*Cin), pre-built
            d2 = dy.reshape(K, N)
            out = ops.conv2d_wgrad(x, dy, k, k, 1, k // 2)
            das = lambda: ops.conv2d_wgrad(x, dy, k, k, 1, k // 2, out=out)              # noqa: E731
            vend = lambda: torch.matmul(d2.t(), a)                                      # noqa: E731
        best = {}
        for name in libs or [None]:
            if name:
                torch.backends.cuda.preferred_blas_library(name)
            for _ in range(3):
                vend()
            torch.cuda.synchronize()
            best[name] = []
        for _ in range(3):
            das()
        kern = lib.das_last_kernel().decode()
        td = []
        for _ in range(ROUNDS):
            for name in libs or [None]:
                if name:
                    torch.backends.cuda.preferred_blas_library(name)
                best[name].append(timed(vend))
            td.append(timed(das))
        vname, tv = min(best.items(), key=lambda kv: statistics.median(kv[1]))
        fl = 2.0 * M * N * K
        mv, md = statistics.median(tv), statistics.median(td)
        print(f'| {tag} | {M} | {N} | {K} | {mv:.1f} ({min(tv):.1f}) [{vname}] | {fl / mv / 1e6:.0f} | {md:.1f} ({min(td):.1f}) | '
              f'{fl / md / 1e6:.0f} | {mv / md:.2f} | `{kern}` |', flush=True)
        del x, a
    print('\nReading: `das / vendor` > 1 means this repo\'s kernel is faster than the vendor GEMM on the pre-built matrix.')


if __name__ == '__main__':
    main()
