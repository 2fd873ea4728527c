#!/usr/bin/env python3
"""Print the kernel timeline of one training step from a rocprofv3 --kernel-trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
names = [r['Kernel_Name'] for r in rows]
adam = [i for i, n in enumerate(names) if n.startswith('step_prologue_kernel')]   # one per training step
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
seg = sorted(rows[adam[k]:adam[k + 1] + 1], key=lambda r: int(r['Start_Timestamp']))
t0 = int(seg[0]['End_Timestamp'])
for r in seg[1:]:
    s = (int(r['Start_Timestamp']) - t0) / 1000
    e = (int(r['End_Timestamp']) - t0) / 1000
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r['Queue_Id']} {r['Grid_Size_X']:>6}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}  {r['Kernel_Name'][:70]}")
