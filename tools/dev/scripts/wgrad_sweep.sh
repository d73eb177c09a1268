for t in 256 384 512 768 1024 2048; do
  echo "== target blocks $t"
  DAS_DEV_WGRAD_BLOCKS=$t python3 tools/dev/wgrad_bench.py 2>&1 | grep -v amdgpu.ids
done
