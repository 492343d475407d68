#!/bin/bash
# Where the waves of the trunk GEMM kernels spend their cycles, per arithmetic mode (f32 | split_bf16): four rocprofv3 --pmc
# passes on bench_conv.py --quick --trunk, summarised by pmc_any.py into gpurun_out/stall_<tag>.json.
# usage: pmc_stall.sh <tag> [bench_conv args, e.g. --mma]
set -e -o pipefail
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stall_$tag
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INST_LEVEL_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    echo "[pmc_stall $tag] pass $i: $set"
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tests/tools/bench_conv.py --quick --trunk "$@" > /dev/null 2> $O/p$i.err || echo "pass $i failed"
done
python3 $R/tests/tools/pmc_any.py "$O/p*/*/*counter_collection.csv" $R/gpurun_out/stall_$tag.json > /dev/null
rm -rf $O
