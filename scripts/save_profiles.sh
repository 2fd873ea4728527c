#!/bin/bash
# Copy the judged subset of gpurun_out/final (scripts/collect_profiles.sh) into profiles/ under the round's names.
set -e
R=${1:-r04}; F=gpurun_out/final
cp $F/bench_default.json profiles/${R}_bench_default.json
cp $F/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp $F/stats/s_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp $F/step_timeline.txt profiles/${R}_step_timeline.txt
cp $F/pmc.json profiles/${R}_pmc.json
cp $F/pmc_summary_table.md profiles/${R}_pmc_summary.md
cp $F/secondary_configs.jsonl profiles/${R}_secondary_configs.jsonl
for f in ddp2_parity_fp32.json ddp2_parity_bf16.json ddp2_parity_fp32_pipelined.json ddp2_parity_bf16_pipelined.json bench_ddp2_gloo_rehearsal.json; do [ -s $F/$f ] && cp $F/$f profiles/${R}_$f; done
for f in mimic_b128_step_timeline.txt mmimdb_b32_step_timeline.txt; do [ -s $F/$f ] && cp $F/$f profiles/${R}_$f; done
[ -s $F/gputest.log ] && tail -12 $F/gputest.log > profiles/${R}_gputest_tail.log || true
