#!/usr/bin/env python3
"""Loop statistics of a kernel in an ISA listing (scripts/isa_probe.sh): for every loop with >= MIN MFMAs, the number of MFMA /
LDS-read / global-load / VALU instructions and of full waits (s_waitcnt lgkmcnt(0) / vmcnt(0)) -- a loop whose every ds_read is
followed by lgkmcnt(0) has its LDS round trips exposed."""
import re
import sys

path, kname = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 16
s = open(path).read().split('\n')
start = [i for i, l in enumerate(s) if l.startswith(kname) and ':' in l][0]
end = [i for i, l in enumerate(s) if i > start and l.startswith('.Lfunc_end')][0]
body = s[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        seg = body[a:i + 1]
        n = lambda pat: sum(bool(re.search(pat, x)) for x in seg)
        if n(r'v_mfma') >= min_mfma:
            pats = [("mfma", r'v_mfma'), ("ds_read", r'ds_read'), ("ds_write", r'ds_write'), ("global/buffer load", r'global_load|buffer_load'),
                    ("store", r'global_store|buffer_store'), ("valu", r'^\s+v_(?!mfma)'), ("salu", r'^\s+s_(?!waitcnt|nop)'),
                    ("lgkmcnt(0)", r'lgkmcnt\(0\)'), ("lgkmcnt(any)", r'lgkmcnt'), ("vmcnt(0)", r'vmcnt\(0\)'), ("vmcnt(any)", r'vmcnt'),
                    ("s_nop", r's_nop')]
            print(f"loop lines {a}-{i} ({i - a} lines): " + "  ".join(f"{k} {n(p)}" for k, p in pats))
