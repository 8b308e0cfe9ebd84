#!/bin/bash
# kernel sequence of the training step around a given kernel name (rocprofv3 kernel trace): bash scripts/prof_sequence.sh <name substring>
PAT=${1:-copyBuffer}
ROOT=$PWD
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_seq -- python3 $ROOT/bench.py --mode train --steps 3 --warmup 2 > $OUT/prof_seq.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/prof_seq/*/*kernel_trace.csv")
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60] for r in rows]
pairs = collections.Counter()
for i, n in enumerate(names):
    if "$PAT" in n:
        pairs[(names[i - 1] if i else "", names[i + 1] if i + 1 < len(names) else "")] += 1
for k, v in pairs.most_common(25):
    print(v, " after:", k[0], " before:", k[1])
# also dump one step's sequence
last = len(names) - 1 - names[::-1].index("cfm_adam_clip_kernel")
prev = last - 1 - names[:last][::-1].index("cfm_adam_clip_kernel")
open("$OUT/train_step_sequence.txt", "w").write("\n".join("%8.1f us  %s" % ((int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3, names[i]) for i in range(prev + 1, last + 1)))
print("step kernels:", last - prev)
t0 = int(rows[prev + 1]["Start_Timestamp"]); t1 = int(rows[last]["End_Timestamp"])
busy = sum(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(prev + 1, last + 1))
print("step span %.3f ms, sum of kernel durations %.3f ms, idle %.3f ms" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
agg = collections.defaultdict(lambda: [0, 0])
for i in range(prev + 1, last + 1):
    a = agg[names[i]]; a[0] += 1; a[1] += int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    print("%4d x %7.1f us = %7.1f us  %s" % (n, t / n / 1e3, t / 1e3, k))
gaps = sorted(((int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3, names[i], names[i + 1]) for i in range(prev + 1, last))
print("largest gaps (us):")
for g in gaps[-12:]:
    print("  %7.1f  after %s  before %s" % g)
PY
rm -rf $OUT/prof_seq
