#!/bin/bash
# One GPU call: the -m gpu suite, then the bench (short, no CPU baseline) with the per-launch breakdown.
# usage: bash scripts/gpu_check.sh <tag> [extra bench args]
tag=${1:-chk}; shift
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/${tag}_test.log 2>&1; rc=$?
tail -3 gpurun_out/${tag}_test.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/${tag}_test.log | head -20; exit $rc; }
python bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d.get("kernels_us"))
PY
