#!/bin/bash
# balanced pixel tiles: parity, per-shape tables with / without, interleaved step A/B
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_conv_tiles_gpu.py -x -q -k "balanced or forced or real" 2>&1 | tail -5 > gpurun_out/r5/bal_tests.txt
cat gpurun_out/r5/bal_tests.txt
TUNE=conv.balance_rows=0 timeout 600 python tools/dev/train_shapes.py > gpurun_out/r5/ts_bal0.txt 2>&1
TUNE=conv.balance_rows=1 timeout 600 python tools/dev/train_shapes.py > gpurun_out/r5/ts_bal1.txt 2>&1
TUNE=conv.balance_rows=2 timeout 600 python tools/dev/train_shapes.py > gpurun_out/r5/ts_bal2.txt 2>&1
for f in 0 1 2; do sed -n 2,14p gpurun_out/r5/ts_bal$f.txt; done
timeout 900 python tools/dev/tune_step.py -n 8 -r 5 conv.balance_rows=0 conv.balance_rows=2 2>&1 | tail -4 | tee gpurun_out/r5/tune_bal.txt
