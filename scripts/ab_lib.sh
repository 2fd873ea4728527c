#!/bin/bash
# A/B of library builds / environments in ONE gpurun call: "NAME=ENV1=v1 ENV2=v2" (space-separated assignments), interleaved, twice.
tag=$1; shift
mkdir -p gpurun_out
for rep in 1 2 3; do
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  env $envs python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/${tag}_${name}_${rep}.json 2> gpurun_out/${tag}_${name}_${rep}.err || { tail -5 gpurun_out/${tag}_${name}_${rep}.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_${name}_${rep}.json").read().strip().split("\n")[-1])
k=d["kernels_us"]
print("${name} rep${rep}: %d samples/s %.4f ms | wgrad %.1f adam+pack %.1f bwd %.1f+%.1f fwd %.1f+%.1f heads %.1f embeds %.1f" % (d["value"], d["ms_per_step"], k["towers_wgrad[all+embeds]"], k["adam+pack"], k.get("tower_bwd[fusion]", 0.0) + k.get("tower_bwd[fusion]+heads", 0.0), k["towers_bwd[image+audio]"], k["towers_fwd[image+audio]"], k["tower_fwd[fusion]"], k.get("heads_ce", 0.0), k.get("embeds_fwd[image+audio]", 0.0)))
PY
done; done
