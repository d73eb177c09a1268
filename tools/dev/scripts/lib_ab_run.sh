#!/bin/bash
# run one python command with tools/dev/ab/libdas_hip_old.so swapped in, then with the in-tree library (one box)
cd "$GRAFT_REPO_ROOT"
L=das_amd/csrc/libdas_hip.so
cp $L /tmp/new.so
for tag in variant intree; do
  if [ $tag = variant ]; then cp tools/dev/ab/libdas_hip_old.so $L; else cp /tmp/new.so $L; fi
  echo "== $tag"
  python3 "$@" 2>&1 | grep -v amdgpu
done
cp /tmp/new.so $L
