#!/bin/bash
# Does the bench line depend on K (timed steps) and on the steps per graph?  (it should not: DESIGN.md section 7)
for rep in 1 2 3; do
for K in 20 50 200; do
for spg in 1 10; do
python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-module-path --steps-per-graph $spg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K', d['steps'], 'spg', $spg, d['ms_per_step'], round(d['ms_per_step']*d['steps'],3), d['shader_clock_mhz'])"
done; done; done
