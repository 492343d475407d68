#!/bin/bash
# Round-3 evidence run on the GPU box (one gpurun call): kernel stats + (kernel, grid) distribution of the bench command,
# PMC HBM-traffic passes of the bench, MFMA-pipe utilisation of the trunk GEMMs AS THE TRUNK SCHEDULE RUNS THEM (forward / dgrad
# on packed weights: conv3p_kernel; weight gradient: wgrad3s_kernel), the 8(f) rows.  Summaries land in gpurun_out/final/
# (copied into profiles/ by hand); raw traces are deleted.  The summary records the sha256 of the kernel sources it was taken on.
# usage: r03_profiles.sh [tag]
set -e -o pipefail
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="--no-cpu-baseline --batched-extra 0 --split-extra 0 --inference-extra 0"
echo "[1/5] kernel stats of the bench command (20 timed + 3 warm-up + 2 capture warm-up steps = 25 step-equivalents)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $B --no-kernel-timer --steps 20 --warmup 3 > $O/${TAG}_bench_under_profiler.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
python3 $R/tests/tools/kernel_dist.py $(ls $O/stats/*/*kernel_trace.csv | head -1) $O/${TAG}_kernel_dist.txt 25
rm -rf $O/stats
echo "[2/5] PMC FETCH_SIZE"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/bench.py --steps 2 --warmup 1 --no-kernel-timer --eager $B > /dev/null 2> $O/pmc_f.err
echo "[3/5] PMC WRITE_SIZE"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/bench.py --steps 2 --warmup 1 --no-kernel-timer --eager $B > /dev/null 2> $O/pmc_w.err
python3 $R/tests/tools/pmc_summary.py "$O/pmc_f/*/*counter_collection.csv" "$O/pmc_w/*/*counter_collection.csv" $O/${TAG}_pmc_traffic.json > /dev/null
rm -rf $O/pmc_f $O/pmc_w
echo "[4/5] PMC MFMA busy, trunk GEMMs as the schedule runs them (bf16x3; forward / dgrad on packed weights)"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -- python3 $R/tests/tools/bench_conv.py --quick --trunk --mode 3 --packed > $O/${TAG}_bench_conv_trunk.txt 2> /dev/null
python3 $R/tests/tools/pmc_mfma.py "$O/pmc_m/*/*counter_collection.csv" $O/${TAG}_pmc_mfma_trunk.json > /dev/null
rm -rf $O/pmc_m
python3 - <<PY
import json, sys
sys.path.insert(0, "$R/tests/tools")
from kernel_sha import kernel_sources_sha
t = json.load(open("$O/${TAG}_pmc_traffic.json"))
m = json.load(open("$O/${TAG}_pmc_mfma_trunk.json"))
rows = {k: v["mfma_util_pct"] for k, v in m.items() if k.startswith("conv3p_kernel") or k.startswith("conv_wgrad3x3") or k.startswith("wgrad3s_kernel")}
vals = list(rows.values())
json.dump({"kernel_sources_sha256": kernel_sources_sha("$R"), "traffic": t,
           "backbone_conv_mfma_util": {"min": min(vals), "max": max(vals), "rows": rows, "arithmetic": "bf16x3",
                                       "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on tests/tools/bench_conv.py --quick --trunk --mode 3 --packed "
                                                 "(ResNet-34 stage convs of a 5x320x800 clip as the trunk schedule runs them: forward + dgrad = conv3p_kernel on packed "
                                                 "weights, weight gradient = wgrad3s_kernel), profiles/${TAG}_pmc_mfma_trunk.json; busy cycles of the bf16 matrix "
                                                 "pipe / (kernel cycles x 1024 SIMDs)"}},
          open("$O/${TAG}_pmc_summary.json", "w"), indent=1)
PY
echo "[5/5] the 8(f) rows"
python3 $R/tests/tools/bench_next_rows.py > $O/${TAG}_next_rows.json 2> $O/next_rows.err || echo "next rows failed"
ls -la $O
