#!/bin/bash
# A/B of two builds of libdas_hip.so on ONE box (boxes of the pool differ by a few %): tools/dev/ab/libdas_hip_old.so
# against the in-tree library. usage: lib_ab.sh "<bench flags for old>" "<bench flags for new>" [rounds]
cd "$GRAFT_REPO_ROOT"
L=das_amd/csrc/libdas_hip.so
cp $L /tmp/new.so
for r in $(seq 1 ${3:-2}); do
  cp tools/dev/ab/libdas_hip_old.so $L
  echo "old: $(python bench.py --no-cpu-baseline $1 2>&1 | tail -1 | cut -c1-120)"
  cp /tmp/new.so $L
  echo "new: $(python bench.py --no-cpu-baseline $2 2>&1 | tail -1 | cut -c1-120)"
done
