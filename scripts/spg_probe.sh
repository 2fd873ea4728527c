#!/bin/bash
# Does the bench line depend on K (timed steps), on the steps per graph, on the input slots?  
#   one step per graph            : one 27 MB batch, re-read every step (stays in the Infinity Cache)
#   ten steps, ten slots (default): 270 MB of inputs cycling, as an epoch over a resident dataset
#   ten steps, shared slot        : M2M_CAPTURE_SHARE_SLOTS=1 (diagnostic): the ten-step graph on one batch
for rep in 1 2 3; do
for K in 20 200; do
for v in "1:0" "10:0" "10:1"; do
spg=${v%%:*}; sh=${v##*:}
M2M_CAPTURE_SHARE_SLOTS=$sh python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-module-path --steps-per-graph $spg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K', d['steps'], 'steps/graph', $spg, 'shared slot', $sh, d['ms_per_step'])"
done; done; done
