#!/bin/bash
# Everything profiles/ holds for a round, in one go.  Run on the GPU box from the repo root:
#   gpurun --timeout 1200 -- 'bash scripts/collect_profiles.sh'      ->  gpurun_out/final/
# (PMC counters are collected in their own passes, with --kernel-trace only.)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
rm -rf $O
mkdir -p $O/pmc
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline \
    > $O/bench_under_rocprof.json 2> $O/rocprof.err
echo "stats done"
for pass in "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE" \
            "valu:SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
    tag=${pass%%:*}
    ctr=${pass#*:}
    mkdir -p $O/pmc/$tag
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc/$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --preheat-ms 0 \
        --no-cpu-baseline --no-graph --profile-steps 1 > $O/pmc/$tag.json 2> $O/pmc/$tag.err
    echo "pmc $tag done"
done
cd $R
python scripts/timeline.py $O/stats/s_kernel_trace.csv 25 > $O/step_timeline.txt
python scripts/pmc_summary.py $O/pmc $O/pmc.json > $O/pmc_summary_table.md
python scripts/bench_configs.py > $O/secondary_configs.jsonl 2> $O/secondary.err
echo "secondary done"
# two ranks on this one GPU over gloo: N ranks x batch B + exchange == one rank on N x B == the oracle (scripts/ddp_parity.py)
timeout -k 10 300 python scripts/ddp_parity.py fp32 S 2> $O/ddp2_fp32.err | grep '^{' > $O/ddp2_parity_fp32.json || echo "ddp parity fp32 FAILED"
timeout -k 10 300 python scripts/ddp_parity.py bf16 B 2> $O/ddp2_bf16.err | grep '^{' > $O/ddp2_parity_bf16.json || echo "ddp parity bf16 FAILED"
# the same with the exchange pipelined with the optimizer (parallel.PipelinedGradSync: one all-reduce per parameter segment)
timeout -k 10 300 python scripts/ddp_parity.py fp32 S pipelined 2> $O/ddp2_fp32_pipe.err | grep '^{' > $O/ddp2_parity_fp32_pipelined.json || echo "ddp parity fp32 pipelined FAILED"
timeout -k 10 300 python scripts/ddp_parity.py bf16 B pipelined 2> $O/ddp2_bf16_pipe.err | grep '^{' > $O/ddp2_parity_bf16_pipelined.json || echo "ddp parity bf16 pipelined FAILED"
M2M_DIST_BACKEND=gloo M2M_FORCE_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_ddp2_gloo_rehearsal.json 2> $O/bench_ddp2.err || echo "ddp bench rehearsal FAILED"
# launch timelines of the secondary configurations at their cfg batches, and the GPU suite at this commit
bash scripts/sec_trace.sh > $O/sec_trace.log 2>&1 && cp gpurun_out/sec/*_step_timeline.txt $O/ || echo "sec trace FAILED"
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1 || echo "GPU TESTS FAILED"
tail -1 $O/gputest.log
echo "all done"
