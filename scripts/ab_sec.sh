#!/bin/bash
# A/B of library builds / environments on the secondary configurations at their cfg batches, ONE gpurun call:
#   bash scripts/ab_sec.sh TAG "NAME=ENV1=v1 ENV2=v2" ...   (interleaved, three repetitions; 200 steps each)
tag=$1; shift
mkdir -p gpurun_out
for rep in 1 2 3; do
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  env $envs python scripts/bench_configs.py --cfg-batch-only --steps 200 --warmup 20 > gpurun_out/${tag}_${name}_${rep}.jsonl 2> gpurun_out/${tag}_${name}_${rep}.err || { tail -5 gpurun_out/${tag}_${name}_${rep}.err; exit 1; }
  python - <<PY
import json
r=[json.loads(l) for l in open("gpurun_out/${tag}_${name}_${rep}.jsonl") if l.startswith("{")]
print("${name} rep${rep}: " + " | ".join("%s B=%d %.4f ms %d samples/s" % (d["metric"].split()[2], d["batch"], d["ms_per_step"], d["value"]) for d in r))
PY
done; done
