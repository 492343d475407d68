#!/bin/bash
# Round-end evidence run on the GPU box (one gpurun call): PMC traffic passes, kernel stats of the bench command, MFMA-pipe
# utilisation of the trunk kernels (f32 and split-bf16 arithmetic), then the default bench.  Summaries land in
# gpurun_out/final/ (copied into profiles/ by hand); the raw traces are deleted.
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="--no-cpu-baseline --batched-extra 0 --split-extra 0"
echo "[1/6] PMC FETCH_SIZE"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/bench.py --steps 2 --warmup 1 --no-kernel-timer --eager $B > /dev/null 2> $O/pmc_f.err
echo "[2/6] PMC WRITE_SIZE"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/bench.py --steps 2 --warmup 1 --no-kernel-timer --eager $B > /dev/null 2> $O/pmc_w.err
python3 $R/tests/tools/pmc_summary.py "$O/pmc_f/*/*counter_collection.csv" "$O/pmc_w/*/*counter_collection.csv" $O/pmc_traffic.json > /dev/null
cp $O/pmc_traffic.json $R/profiles/r01_pmc_traffic.json
rm -rf $O/pmc_f $O/pmc_w
echo "[3/6] kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $B > $O/bench_under_profiler.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 $R/tests/tools/kernel_dist.py $(ls $O/stats/*/*kernel_trace.csv | head -1) $O/kernel_dist.txt
python3 $R/tests/tools/overlap.py $(ls $O/stats/*/*kernel_trace.csv | head -1) 0.2 > $O/overlap.txt
rm -rf $O/stats
echo "[4/6] PMC MFMA busy, trunk, f32"; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -- python3 $R/tests/tools/bench_conv.py --quick --trunk > $O/bench_conv_f32.txt 2> /dev/null
python3 $R/tests/tools/pmc_mfma.py "$O/pmc_m/*/*counter_collection.csv" $O/pmc_mfma_trunk.json > /dev/null
rm -rf $O/pmc_m
echo "[5/6] PMC MFMA busy, trunk, split-bf16"; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -- python3 $R/tests/tools/bench_conv.py --quick --trunk --mma > $O/bench_conv_split.txt 2> /dev/null
python3 $R/tests/tools/pmc_mfma.py "$O/pmc_m/*/*counter_collection.csv" $O/pmc_mfma_trunk_split_bf16.json > /dev/null
rm -rf $O/pmc_m
cp $O/pmc_mfma_trunk.json $R/profiles/r01_pmc_mfma_trunk.json
echo "[6/6] default bench"; cd $R; python3 bench.py > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
