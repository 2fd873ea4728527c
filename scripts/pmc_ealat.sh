#!/bin/bash
# Where are a kernel's L2 misses served from?  rocprofv3 lists no Infinity-Cache (MALL) counters on gfx950, but the L2's
# memory-side queue depth does tell: mean read latency = TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ (cycles a fabric read stays
# outstanding).  A launch whose reads hit the 256 MiB Infinity Cache shows a lower mean than one streaming from HBM (Adam).
#   gpurun --timeout 600 -- 'bash scripts/pmc_ealat.sh [tag]'  ->  gpurun_out/ealat_<tag>/summary.txt
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
tag=${1:-x}
O=$R/gpurun_out/ealat_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in "rd:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum" "wr:TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum" "hit:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
    t=${pass%%:*}; ctr=${pass#*:}
    mkdir -p $O/$t
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$t -o p -- python3 $R/bench.py --steps 3 --warmup 1 --preheat-ms 0 \
        --no-cpu-baseline --no-graph --profile-steps 1 > $O/$t.json 2> $O/$t.err || { echo "pass $t failed"; tail -3 $O/$t.err; }
    echo "pmc $t done"
done
cd $R
python scripts/pmc_extra.py $O > $O/summary.txt
cat $O/summary.txt
