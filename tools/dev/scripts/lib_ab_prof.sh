#!/bin/bash
# rocprofv3 kernel sums of the train bench with tools/dev/ab/libdas_hip_old.so and with the in-tree library (one box)
# usage: lib_ab_prof.sh "<bench flags old>" "<bench flags new>" <grep pattern>
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
L=das_amd/csrc/libdas_hip.so
cp $L /tmp/new.so
mkdir -p gpurun_out/prof
for tag in old new; do
  if [ $tag = old ]; then cp tools/dev/ab/libdas_hip_old.so $L; FL="$1"; else cp /tmp/new.so $L; FL="$2"; fi
  rm -rf /tmp/abp_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/abp_$tag -o tr -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $FL > /tmp/abp_$tag.log 2>&1
  db=$(find /tmp/abp_$tag -name "*.db" | head -1)
  python3 tools/dev/rocprof_summary.py "$db" gpurun_out/prof/ab_$tag.md "bench.py --steps 5 --warmup 2 $FL" > /dev/null
  echo "== $tag: $(tail -1 /tmp/abp_$tag.log | cut -c1-110)"
  grep -E "total kernel|$3" gpurun_out/prof/ab_$tag.md | cut -c1-150
done
cp /tmp/new.so $L
