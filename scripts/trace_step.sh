#!/bin/bash
# Kernel timeline of one replayed training step (rocprofv3 --kernel-trace): run on the GPU box from the repo root.
#   gpurun -- 'bash scripts/trace_step.sh [bench.py args]'   ->  gpurun_out/timeline.txt
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
rm -rf $R/gpurun_out/tl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o t -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 1 "$@" > $R/gpurun_out/tl_bench.json 2> $R/gpurun_out/tl_bench.err
cd $R
python scripts/timeline.py gpurun_out/tl/t_kernel_trace.csv 25 > gpurun_out/timeline.txt
cat gpurun_out/timeline.txt
