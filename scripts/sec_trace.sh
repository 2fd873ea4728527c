#!/bin/bash
# Launch timelines of the secondary configurations at their cfg batches (rocprofv3 --kernel-trace over scripts/sec_prof.py).
#   gpurun -- 'bash scripts/sec_trace.sh'  ->  gpurun_out/sec/{mimic_b128,mmimdb_b32}_step_timeline.txt
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/sec
cd /tmp && export TMPDIR=/tmp
for cfg in "mimic 128 embed_fwd_kernel" "mmimdb 32 embed_fwd_group"; do
  set -- $cfg
  export TASK=$1 B=$2
  rm -rf $R/gpurun_out/sec/tr_$1
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sec/tr_$1 -o t -- python3 $R/scripts/sec_prof.py > $R/gpurun_out/sec/$1.out 2> $R/gpurun_out/sec/$1.err
  python3 $R/scripts/timeline2.py $R/gpurun_out/sec/tr_$1 $3 > $R/gpurun_out/sec/$1_b$2_step_timeline.txt
  rm -rf $R/gpurun_out/sec/tr_$1
  echo "== $1 B=$2"; cat $R/gpurun_out/sec/$1_b$2_step_timeline.txt
done
