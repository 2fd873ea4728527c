"""One step's kernels from a rocprofv3 kernel trace: python scripts/timeline2.py <dir> <first-kernel-substring> [which]"""
import csv, glob, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
first = sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
i0, i1 = idx[which], idx[which + 1]
t0 = int(rows[i0]["Start_Timestamp"])
prev = None
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:56]
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap {gap:5.1f}  q{r['Queue_Id']} {n:56s} grid={r['Grid_Size_X']} wg={r['Workgroup_Size_X']}")
    prev = e
print(f"step span {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us, {i1 - i0} kernels")
