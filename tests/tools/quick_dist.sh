#!/bin/bash
# kernel distribution of the bench command only (step 1 of final_profiles.sh).  usage: quick_dist.sh [tag]
set -e -o pipefail
TAG=${1:-quick}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="--no-cpu-baseline --batched-extra 0 --split-extra 0 --inference-extra 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $B --no-kernel-timer --steps 20 --warmup 3 > $O/${TAG}_bench_under_profiler.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
python3 $R/tests/tools/kernel_dist.py $(ls $O/stats/*/*kernel_trace.csv | head -1) $O/${TAG}_kernel_dist.txt 25
rm -rf $O/stats
