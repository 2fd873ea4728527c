#!/bin/bash
# Like ab.sh, but 3 interleaved repetitions of 400 timed steps each and a summary of the means (boxes and runs differ by ~1 %).
tag=$1; shift
mkdir -p gpurun_out
for rep in 1 2 3; do
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  env $envs python bench.py --steps 400 --warmup 20 --no-cpu-baseline --profile-steps 20 > gpurun_out/${tag}_${name}_${rep}.json 2> gpurun_out/${tag}_${name}_${rep}.err || { tail -5 gpurun_out/${tag}_${name}_${rep}.err; exit 1; }
done; done
python - "$tag" "$@" <<'PY'
import json,sys
tag=sys.argv[1]
for spec in sys.argv[2:]:
    name=spec.split('=')[0]; vals=[]; ks={}
    for rep in (1,2,3):
        d=json.loads(open(f"gpurun_out/{tag}_{name}_{rep}.json").read().strip().split("\n")[-1])
        vals.append(d["value"])
        for k,v in d["kernels_us"].items(): ks.setdefault(k,[]).append(v)
    print(f"{name:8s} samples/s {[round(v) for v in vals]} mean {sum(vals)/3:.0f}   " + " ".join(f"{k.split('[')[0]}={sum(v)/3:.1f}" for k,v in ks.items()))
PY
