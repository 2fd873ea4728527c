#!/bin/bash
# LDS counters of the training step for several library variants (M2M_LIB_PATH): bash scripts/pmc_lds_variants.sh name=path ...
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%=*}; path=${spec#*=}
  O=$R/gpurun_out/pmcl_$name; rm -rf $O; mkdir -p $O/lds $O/wait
  export M2M_LIB_PATH=$path
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/lds -o p -- python3 $R/bench.py --steps 3 --warmup 1 --preheat-ms 0 --no-cpu-baseline --no-graph --profile-steps 1 > $O/lds.json 2> $O/lds.err
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/wait -o p -- python3 $R/bench.py --steps 3 --warmup 1 --preheat-ms 0 --no-cpu-baseline --no-graph --profile-steps 1 > $O/wait.json 2> $O/wait.err
  echo "=== $name"; python3 $R/scripts/pmc_extra.py $O | grep -A12 "tower_bwd_group\|tower_fwd_group" 
done
