#!/bin/bash
# Where the waves of conv3p_kernel spend their cycles: rocprofv3 --pmc passes on bench_conv3p.py --pmc, summarised by pmc_any.py.
# usage: pmc_conv3p.sh <tag> [bench_conv3p args, e.g. --clips 8]
set -e -o pipefail
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc3p_$tag
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INST_LEVEL_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
    i=$((i+1))
    echo "[pmc_conv3p $tag] pass $i: $set"
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tests/tools/bench_conv3p.py --pmc "$@" > /dev/null 2> $O/p$i.err || echo "pass $i failed"
done
python3 $R/tests/tools/pmc_any.py "$O/p*/*/*counter_collection.csv" $R/gpurun_out/pmc3p_$tag.json conv3p_kernel > /dev/null
rm -rf $O
