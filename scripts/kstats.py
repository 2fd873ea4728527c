"""Per-kernel summary of a rocprofv3 kernel trace: python scripts/kstats.py <dir-or-csv> [skip_first_n_per_kernel]"""
import collections
import csv
import glob
import os
import sys

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    d[(name, r.get("Grid_Size", ""))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("VGPR_Count", "") or r.get("Arch_VGPR_Count", ""), r.get("LDS_Block_Size", "")))
rows = []
for (name, grid), v in d.items():
    v.sort()
    w = v[skip:] if len(v) > skip + 2 else v
    durs = sorted(e - s for s, e, _, _ in w)
    rows.append((sum(durs) / len(durs) * (len(v)), name, grid, len(v), durs[len(durs) // 2] / 1e3, sum(durs) / len(durs) / 1e3, v[0][2], v[0][3]))
rows.sort(reverse=True)
print(f"{'kernel':70s} {'grid':>8s} {'calls':>6s} {'median us':>10s} {'mean us':>9s} {'vgpr':>5s} {'lds':>7s}")
for _, name, grid, n, med, mean, vg, lds in rows[:40]:
    print(f"{name[:70]:70s} {grid:>8s} {n:6d} {med:10.1f} {mean:9.1f} {vg:>5s} {lds:>7s}")
