#!/bin/bash
# rocprofv3 kernel stats of one python command with tools/dev/ab/libdas_hip_old.so and with the in-tree library
# usage: lib_ab_cmd.sh <grep pattern> <python script and args...>
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
L=das_amd/csrc/libdas_hip.so
cp $L /tmp/new.so
pat=$1; shift
for tag in old new; do
  if [ $tag = old ]; then cp tools/dev/ab/libdas_hip_old.so $L; else cp /tmp/new.so $L; fi
  rm -rf /tmp/abc_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/abc_$tag -o tr -- python3 "$@" > /tmp/abc_$tag.log 2>&1
  db=$(find /tmp/abc_$tag -name "*.db" | head -1)
  python3 tools/dev/rocprof_summary.py "$db" /tmp/abc_$tag.md "$*" > /dev/null
  echo "== $tag"
  grep -E "$pat" /tmp/abc_$tag.md | cut -c1-140
done
cp /tmp/new.so $L
