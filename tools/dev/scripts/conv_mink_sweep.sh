for t in 0 100000; do
  echo "== big-tile kernel only for K >= $t"
  DAS_DEV_BIG_MINK=$t python3 tools/dev/conv_bench.py 16 2>&1 | grep -v amdgpu.ids
done
