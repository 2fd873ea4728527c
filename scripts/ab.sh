#!/bin/bash
# A/B in ONE gpurun call (boxes differ by several percent): runs the short bench for each "NAME=ENVSTRING" argument, twice, interleaved.
# usage: bash scripts/ab.sh tag "A=M2M_STATIC_STEPS=1" "B=M2M_STATIC_STEPS=0"
tag=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  env $envs python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/${tag}_${name}_${rep}.json 2> gpurun_out/${tag}_${name}_${rep}.err || { tail -5 gpurun_out/${tag}_${name}_${rep}.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_${name}_${rep}.json").read().strip().split("\n")[-1])
print("${name} rep${rep}:", round(d["value"]), d["ms_per_step"], d.get("kernels_us"))
PY
done; done
