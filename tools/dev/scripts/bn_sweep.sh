R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "1024,256" "256,1024" "512,1024" "256,256" "2048,256"; do
  export DAS_DEV_BN_REDUCE=$cfg
  for i in 0 1 3 4 7; do
    rm -rf /tmp/p_$i
    rocprofv3 --kernel-trace --stats -d /tmp/p_$i -o r -- python3 $R/tools/dev/bn_bench.py $i recompute > /tmp/o_$i.log 2>&1
    grep -E "shape|backward" /tmp/o_$i.log
    python3 - <<PY
import sqlite3,glob
db=glob.glob('/tmp/p_$i/*.db')[0]
c=sqlite3.connect(db)
for n,cnt,a in c.execute("select name,count(*),avg(end-start) from kernels where name like '%bn_%' group by name order by 3 desc"):
    print('   cfg $cfg', n[:60].replace('(anonymous namespace)::',''), cnt, round(a/1e3,1),'us')
PY
  done
done
