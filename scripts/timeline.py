#!/usr/bin/env python3
"""Print the kernel timeline of one training step from a rocprofv3 --kernel-trace csv.

usage: timeline.py <kernel_trace.csv> [step index]  -- a step starts at its first launch: the grouped patch-embedding forward
(which carries the step prologue) or, on paths without it, step_prologue_kernel."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
heads = [i for i, n in enumerate(names) if 'embed_fwd_group_kernel' in n]
if not heads:
    heads = [i for i, n in enumerate(names) if n.startswith('step_prologue_kernel')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
k = min(k, len(heads) - 2)
seg = rows[heads[k]:heads[k + 1] + 1]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    s = (int(r['Start_Timestamp']) - t0) / 1000
    e = (int(r['End_Timestamp']) - t0) / 1000
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r['Queue_Id']} {r['Grid_Size_X']:>6}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}  {r['Kernel_Name'][:70]}")
