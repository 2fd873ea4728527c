#!/bin/bash
# A/B of library builds in ONE gpurun call: "NAME=path/to/lib.so[,ENV=v...]" ..., interleaved, REPS (default 2) repetitions of 200 timed steps.
tag=$1; shift
mkdir -p gpurun_out/$tag
for rep in $(seq 1 ${REPS:-2}); do
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}; lib=${rest%%,*}; envs=""
  [ "$rest" != "$lib" ] && envs=$(echo "${rest#*,}" | tr ',' ' ')
  env M2M_LIB_PATH=$PWD/$lib $envs python bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-steps ${PSTEPS:-10} > gpurun_out/$tag/${name}_${rep}.json 2> gpurun_out/$tag/${name}_${rep}.err || { tail -5 gpurun_out/$tag/${name}_${rep}.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/$tag/${name}_${rep}.json").read().strip().split("\n")[-1])
k=d["kernels_us"]
print("${name} rep${rep}: %d samples/s %.4f ms | wgrad %.1f adam+pack %.1f bwd %.1f+%.1f fwd %.1f+%.1f heads %.1f embeds %.1f" % (d["value"], d["ms_per_step"], k["towers_wgrad[all+embeds]"], k["adam+pack"], k.get("tower_bwd[fusion]", 0.0) + k.get("tower_bwd[fusion]+heads", 0.0), k["towers_bwd[image+audio]"], k["towers_fwd[image+audio]"], k["tower_fwd[fusion]"], k.get("heads_ce", 0.0), k.get("embeds_fwd[image+audio]", 0.0)))
PY
done; done
