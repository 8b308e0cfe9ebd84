#!/bin/bash
# rocprofv3 kernel stats of the config-3 training step (via gpurun): bash scripts/prof_train.sh <tag> -> gpurun_out/<tag>_train_step_kernel_stats.csv
TAG=${1:-r03}
ROOT=$PWD
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$TAG -- python3 $ROOT/bench.py --mode train --steps 13 --warmup 3 > $OUT/prof_train_$TAG.log 2>&1
cp $(ls $OUT/prof_train_$TAG/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_train_step_kernel_stats.csv
rm -rf $OUT/prof_train_$TAG
cd $ROOT
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/${TAG}_train_step_kernel_stats.csv")))
steps = 16.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("kernel time per step %.3f ms, launches per step %.0f" % (tot / steps / 1e6, calls / steps))
for r in rows[:28]:
    print("%7.1f/step %8.1f us avg %6.2f%%  %s" % (int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:150]))
PY
