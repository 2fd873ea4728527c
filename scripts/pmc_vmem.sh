#!/bin/bash
# Vector-memory path counters of the step's launches (texture addresser / L1 / L2 request side), one pass per group:
#   gpurun --timeout 900 -- 'bash scripts/pmc_vmem.sh'   ->  gpurun_out/pmcv/summary.txt
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmcv
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in "ta:TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
            "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
            "tcc:TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
            "sq:SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    t=${pass%%:*}; ctr=${pass#*:}
    mkdir -p $O/$t
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$t -o p -- python3 $R/bench.py --steps 3 --warmup 1 --preheat-ms 0 \
        --no-cpu-baseline --no-graph --profile-steps 1 > $O/$t.json 2> $O/$t.err || { echo "pass $t failed"; tail -3 $O/$t.err; }
    echo "pmc $t done"
done
cd $R
python scripts/pmc_extra.py $O > $O/summary.txt
cat $O/summary.txt
