#!/bin/bash
# Diagnostic PMC passes beyond scripts/collect_profiles.sh: LDS conflicts, wait classes, instruction mix per kernel.
#   gpurun --timeout 900 -- 'bash scripts/pmc_extra.sh [tag]'   ->  gpurun_out/pmcx_<tag>/ + summary.txt
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
tag=${1:-x}
O=$R/gpurun_out/pmcx_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
            "act:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
            "wait:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
            "inst:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
            "inst2:SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_BUSY_CYCLES" \
            "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
            "ta:SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM" \
            "lvl:SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR"; do
    t=${pass%%:*}; ctr=${pass#*:}
    mkdir -p $O/$t
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$t -o p -- python3 $R/bench.py --steps 3 --warmup 1 --preheat-ms 0 \
        --no-cpu-baseline --no-graph --profile-steps 1 > $O/$t.json 2> $O/$t.err || { echo "pass $t failed"; tail -3 $O/$t.err; }
    echo "pmc $t done"
done
cd $R
python scripts/pmc_extra.py $O > $O/summary.txt
cat $O/summary.txt
