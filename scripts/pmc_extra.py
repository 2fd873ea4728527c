"""Per-kernel medians of every counter found under <dir>/*/p_counter_collection.csv (scripts/pmc_extra.sh)."""
import collections, csv, glob, os, sys

KEEP = ("tower_", "embed_", "adam_kernel", "heads_kernel", "pack_all")


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(root, "*", "p_counter_collection.csv"))):
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if any(s in k for s in KEEP):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    med = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    print(f"## {k}")
    wc = med.get("SQ_WAVE_CYCLES", 0)
    for n in sorted(med):
        extra = f"   ({med[n] / wc:.3f} of wave cycles)" if wc and n.startswith(("SQ_WAIT", "SQ_ACTIVE")) else ""
        print(f"  {n:34s} {med[n]:16,.0f}{extra}")
    if med.get("SQ_LDS_IDX_ACTIVE"):
        print(f"  -> LDS bank-conflict share of LDS-array cycles: {med.get('SQ_LDS_BANK_CONFLICT', 0) / med['SQ_LDS_IDX_ACTIVE']:.3f}")
    if med.get("GRBM_GUI_ACTIVE") and med.get("SQ_LDS_IDX_ACTIVE"):
        cyc = med["GRBM_GUI_ACTIVE"] / 8
        print(f"  -> LDS array busy: {med['SQ_LDS_IDX_ACTIVE'] / (cyc * 256):.3f} of CU cycles (if the counter sums over CUs)")
