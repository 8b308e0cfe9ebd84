#!/bin/bash
# Refresh the judged profile artifacts on the GPU box (everything lands in gpurun_out/refresh/, copy into profiles/ afterwards).
# Usage (via gpurun): bash scripts/refresh_profiles.sh r03
set -e
TAG=${1:-r03}
OUT=$PWD/gpurun_out/refresh
mkdir -p $OUT
ROOT=$PWD
timeout -k 10 300 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fwd -- python3 $ROOT/bench.py --steps 100 --warmup 20 --train-steps 0 --no-cpu-baseline --no-parity --no-live-traffic > $OUT/prof_fwd.log 2>&1
cp $(ls $OUT/prof_fwd/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_rocprofv3_kernel_stats_bench.csv
echo "forward stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train -- python3 $ROOT/bench.py --mode train --steps 13 --warmup 3 > $OUT/prof_train.log 2>&1
cp $(ls $OUT/prof_train/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_train_step_kernel_stats_final.csv
echo "train stats done"
cd $ROOT
timeout -k 10 400 python scripts/collect_traffic.py $TAG > $OUT/collect_traffic.log 2>&1
cp profiles/${TAG}_hbm_traffic.json profiles/${TAG}_mfma_util.json $OUT/ 2>/dev/null || true
timeout -k 10 200 python scripts/bench_streaming.py 2>/dev/null > $OUT/${TAG}_streaming_config5.txt
echo "streaming done"
timeout -k 10 300 python scripts/bench_config4.py 2>/dev/null > $OUT/${TAG}_config4_bench.txt
echo "config 4 done"
rm -rf $OUT/prof_fwd $OUT/prof_train gpurun_out/traffic
echo "traffic done"
tail -c 600 $OUT/${TAG}_bench_default.json
