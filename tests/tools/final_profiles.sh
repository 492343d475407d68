#!/bin/bash
# Round-2 evidence run on the GPU box (one gpurun call): kernel stats + (kernel, grid) distribution of the bench command,
# PMC HBM-traffic passes, MFMA-pipe utilisation and stall counters of the trunk GEMMs in the default (bf16x3) and the
# f32-input arithmetic.  Summaries land in gpurun_out/final/ (copied into profiles/ by hand); raw traces are deleted.
# usage: final_profiles.sh [tag]      (tag: file-name prefix, default r02)
set -e -o pipefail
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="--no-cpu-baseline --batched-extra 0 --split-extra 0 --inference-extra 0"
echo "[1/6] kernel stats of the bench command (20 timed + 3 warm-up + 2 capture warm-up steps = 25 step-equivalents)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $B --no-kernel-timer --steps 20 --warmup 3 > $O/${TAG}_bench_under_profiler.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
python3 $R/tests/tools/kernel_dist.py $(ls $O/stats/*/*kernel_trace.csv | head -1) $O/${TAG}_kernel_dist.txt 25
rm -rf $O/stats
echo "[2/6] PMC FETCH_SIZE"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/bench.py --steps 2 --warmup 1 --no-kernel-timer --eager $B > /dev/null 2> $O/pmc_f.err
echo "[3/6] PMC WRITE_SIZE"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/bench.py --steps 2 --warmup 1 --no-kernel-timer --eager $B > /dev/null 2> $O/pmc_w.err
python3 $R/tests/tools/pmc_summary.py "$O/pmc_f/*/*counter_collection.csv" "$O/pmc_w/*/*counter_collection.csv" $O/${TAG}_pmc_traffic.json > /dev/null
rm -rf $O/pmc_f $O/pmc_w
echo "[4/6] PMC MFMA busy, trunk GEMMs, default arithmetic (bf16x3)"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -- python3 $R/tests/tools/bench_conv.py --quick --trunk --mode 3 > $O/${TAG}_bench_conv_bf16x3.txt 2> /dev/null
python3 $R/tests/tools/pmc_mfma.py "$O/pmc_m/*/*counter_collection.csv" $O/${TAG}_pmc_mfma_trunk.json > /dev/null
rm -rf $O/pmc_m
echo "[5/6] PMC MFMA busy, trunk GEMMs, f32-input MFMA"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -- python3 $R/tests/tools/bench_conv.py --quick --trunk --mode 0 > $O/${TAG}_bench_conv_f32.txt 2> /dev/null
python3 $R/tests/tools/pmc_mfma.py "$O/pmc_m/*/*counter_collection.csv" $O/${TAG}_pmc_mfma_trunk_f32.json > /dev/null
rm -rf $O/pmc_m
python3 - <<PY
import json
t = json.load(open("$O/${TAG}_pmc_traffic.json"))
m = json.load(open("$O/${TAG}_pmc_mfma_trunk.json"))
vals = [v["mfma_util_pct"] for k, v in m.items()]
json.dump({"traffic": t,
           "backbone_conv_mfma_util": {"min": min(vals), "max": max(vals), "arithmetic": "bf16x3",
                                       "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on tests/tools/bench_conv.py --quick --trunk --mode 3 "
                                                 "(ResNet-34 stage convs of a 5x320x800 clip: forward, dgrad, wgrad), profiles/${TAG}_pmc_mfma_trunk.json; "
                                                 "busy cycles of the bf16 matrix pipe / (kernel cycles x 1024 SIMDs)"}},
          open("$O/${TAG}_pmc_summary.json", "w"), indent=1)
PY
echo "[6/6] stall counters of the trunk GEMMs (default arithmetic)"
bash $R/tests/tools/pmc_stall.sh ${TAG}_bf16x3 --mode 3 > /dev/null 2>&1 || echo "stall pass failed"
cp $R/gpurun_out/stall_${TAG}_bf16x3.json $O/${TAG}_pmc_stall_trunk.json || true
ls -la $O
