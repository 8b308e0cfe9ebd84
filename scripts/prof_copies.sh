#!/bin/bash
# which memory copies does a training step issue?  (rocprofv3 memory-copy trace, no counters)   bash scripts/prof_copies.sh
ROOT=$PWD
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d $OUT/prof_copies -- python3 $ROOT/bench.py --mode train --steps 4 --warmup 2 > $OUT/prof_copies.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/prof_copies/*/*memory_copy_trace.csv")
print(f)
rows = list(csv.DictReader(open(f[0])))
print(rows[0].keys())
c = collections.Counter((r.get("Direction"), r.get("Bytes") or str(int(r.get("End_Timestamp",0))-int(r.get("Start_Timestamp",0)))) for r in rows)
for k, v in c.most_common(40):
    print(v, k)
PY
rm -rf $OUT/prof_copies
