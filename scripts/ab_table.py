#!/usr/bin/env python3
"""profiles/r04_ab_results.txt from the A/B outputs under gpurun_out/ (scripts/ab_lib2.sh writes one bench JSON line per arm and
repetition).  The labels say what each block compared."""
import collections
import glob
import json
import os
import re

ALLPLAIN = ("(these three blocks ran a build whose `if (plain) *p = v; else __builtin_nontemporal_store(v, p);` had been merged by the "
            "compiler into ONE plain store: every stored operand of the backward went through L2 / the memory-side cache -- two-tower "
            "backward 123 -> 131-150 us.  The hand-off switches below therefore did nothing; what the blocks do show is the price of "
            "plain stores and that Adam's non-temporal moment streams win part of it back.  Valid comparison: r4z, r5a)")
LABELS = collections.OrderedDict([
    ('r4a', ("round-3 end state, re-measured on this round's first box", {'bench': 'round-3 library'})),
    ('r4b', ("backward column loop written as issue/consume stages (sched_barrier) vs the plain loop (DESIGN 4g.2)", {'NEW': 'staged loop', 'OLD': 'plain loop (round 3)'})),
    ('r4c', ("token-mixing backward: all 16 table cells of a pass staged at once (32 spills) -- NEGATIVE", {'NEW': '16 cells staged', 'S1': 'staged column loop only', 'OLD': 'plain loop (round 3)'})),
    ('r4d', ("token-mixing backward: 8 table cells per column tile staged (1 spill) -- kept", {'NEW': '8 cells staged', 'S1': 'staged column loop only'})),
    ('r4e', ("heads kernel rewritten (padded rows, loads before LDS writes, 32-lane softmax): heads_ce 13.3 -> 9.2 us", {'NEW': 'new heads kernel', 'S2': 'previous build'})),
    ('r4g', ("next step's weight prefetch interleaved with the products (M2M_BWD_PF_INTERLEAVE=1) -- no gain, default 0", {'NEW': 'interleaved prefetch', 'S3': 'previous build'})),
    ('r4j', ("weight-gradient kernel with 4 waves x 3 column tiles (M2M_WG48) -- NEGATIVE (512 VGPRs, spills), default 0", {'NEW': 'WG48 depth 2', 'D1': 'WG48 depth 1', 'EL': 'WG48, embedding workgroups last', 'S4': 'previous build (5 waves x 2)'})),
    ('r4l', ("forward token-mixing residual: reads batched before the guarded stores -- no change (kept, simpler ISA)", {'NEW': 'batched reads', 'S5': 'previous build'})),
    ('r4m', ("rewritten fused Adam + re-pack kernel (M2M_FUSED_UPDATE=1) vs separate adam + pack_all at 8.3 M parameters -- NEGATIVE at this size (75 vs 63 us); default only for <= 4 M parameters", {'FU': 'fused Adam + pack', 'SEP': 'adam_kernel + pack_all_kernel'})),
    ('r4n', ("embeddings as the prologue of the towers forward launch (M2M_EMBED_FOLD=1) -- NEGATIVE (103.8 vs 84.9 + 21.5 us), opt-in", {'FOLD': 'embeds folded', 'SEP': 'separate embeds launch'})),
    ('r4o', ("recompute form of the weight gradient (M2M_WGRAD_RECOMP=1, no Hact hand-off) vs stored operands -- NEGATIVE for the step", {'RC': 'recompute, 4 waves', 'ST': 'stored operands (default)'})),
    ('r4p', ("recompute form with 8 waves (RC_WAVES=8) -- NEGATIVE", {'RC8': 'recompute, 8 waves', 'ST': 'stored operands (default)'})),
    ('r4q', ("token-mixing backward with two sample pairs per pass (M2M_TOK_PU=2, 58 spills) -- NEGATIVE; NEW = w1tc image no longer re-packed", {'PU2': 'PU=2', 'NEW': 'w1tc skip (kept)'})),
    ('r4r', ("weight-gradient kernel: own h/dh fragments requested two steps ahead (M2M_WG_HD2) -- kept (-1.5 us in the 30-step event timing)", {'HD2': 'two-step fragment prefetch', 'S8': 'previous build'})),
    ('r4v', ("stored operands of the last-written backward blocks kept cache-resident + non-temporal optimizer streams " + ALLPLAIN,
             {'BASE': 'all off', 'R2': '2 resident blocks', 'R2A1': '+ Adam m/v non-temporal', 'R2A7P': '+ Adam p non-temporal, re-pack loads non-temporal',
              'R3A7P': '3 resident blocks, all non-temporal', 'A7P': 'optimizer non-temporal only', 'R2A3': '2 blocks, Adam m/v + p loads', 'R1A1': '1 block, Adam m/v'})),
    ('r4w', ("a share K/M of every block's 32-row pairs cache-resident " + ALLPLAIN,
             {'BASE': 'all off', 'A1': 'Adam m/v non-temporal', 'A1F13': '+ 1/3 resident', 'A1F12': '+ 1/2 resident', 'A1F14': '+ 1/4 resident', 'A1R1': '+ 1 block resident', 'A1F23': '+ 2/3 resident'})),
    ('r4x', ("the same on a second box " + ALLPLAIN,
             {'A1': 'Adam m/v non-temporal', 'A1F12': '+ 1/2 resident', 'A1F11': '+ all plain', 'A1F23': '+ 2/3 resident', 'A3F12': 'Adam m/v + p loads, 1/2', 'BASE': 'all off'})),
    ('r4y', ("forward column loop: the eight table look-ups as one batch (M2M_FWD_STAGED) -- neutral (83.9 vs 84.1 us), kept; embedding forward with a two-stage ring "
             "and two workgroups per CU: 21.3 -> 16.1 us (this block still on the all-plain-stores build: compare inside the block only)",
             {'NEW': 'staged forward look-ups', 'FPLAIN': 'M2M_FWD_STAGED=0', 'EMBD2': 'staged + embedding ring 2 / 2 workgroups per CU'})),
    ('r4z', ("the commit before the store-policy experiment (OLD) against the experiment's build with every switch off (NEW0): the merged plain store costs 20-30 us in the two-tower backward",
             {'OLD': 'committed library (non-temporal operand stores)', 'NEW0': 'experiment build, switches off (= all plain stores)', 'NEW': 'experiment build, 1/2 + Adam nt', 'NEWF0': 'experiment build, Adam nt only'})),
    ('r5a', ("VALID comparison after the revert (operand stores non-temporal again): Adam's moment streams non-temporal (M2M_ADAM_NT=1) and the embedding launch at two workgroups per CU -- both kept",
             {'OLD': 'committed library', 'NEW': 'reverted stores + staged forward', 'A1': '+ Adam m/v non-temporal', 'EMBD2': '+ embedding ring 2 / 2 workgroups per CU', 'A1E': 'both'})),
    ('r5b', ("k-splits of the audio embedding with two workgroups per CU (M2M_EMBED_SPLITS) -- 2 stays", {'S2': '2 splits (default)', 'S3': '3 splits', 'S4': '4 splits', 'S1': 'no split'})),
    ('r5f', ("re-pack: four embedding slots per thread so that the grid fits one round of four workgroups per CU -- no change, not kept", {'SPT4': '4 slots per thread', 'SPT1': '1 slot per thread'})),
    ('r5h', ("re-pack: the embeddings' packed slots from two 16-byte row loads instead of eight 4-byte gathers -- kept (re-pack -1.3 us)", {'NEW': '16-byte row loads', 'OLD': 'previous build'})),
    ('r5k', ("embedding forward: patch slots four stages ahead, weight fragments two (two rings) -- embedding launch +0.7 us on a cache-resident batch, step -0.25 %: not kept", {'NEW': 'two rings', 'OLD': 'one ring of two stages'})),
    ('r5l', ("one-launch Adam + re-pack with W2 in 8-row x 512-column tiles (M2M_AP_ROWTILES=1), moments still plain: the update itself 70 -> 57 us, but the plain moment streams slow the other launches", {'SEP': 'flat Adam, then re-pack', 'FUR': 'one launch, row tiles', 'FUC': 'one launch, 32-column-group tiles'})),
    ('r5m', ("the same with the moment streams non-temporal (compile-time switch: large models) -- kept, now the default", {'SEP': 'flat Adam, then re-pack', 'FUR': 'one launch, row tiles, nt moments', 'FURP': 'one launch, row tiles, plain moments'})),
    ('r5n', ("one-launch form: tile width and non-temporal masters -- 512 columns, plain masters stay", {'W512': '512 columns (default)', 'NTP': '+ masters non-temporal', 'W1024': '1024 columns', 'W256': '256 columns'})),
    ('r5w', ("heads kernel with 8 samples per workgroup (-DHEAD_S=8) -- heads 9.0 -> 10.9 us: 16 stays", {'S16': '16 samples per workgroup (default)', 'S8': '8 samples per workgroup'})),
    ('r5y', ("backward column loop: s_setprio 3 around the weight-prefetch issue -- no change", {'BASE': 'default', 'PRIO': 's_setprio around the prefetch'})),
    ('r5p', ("one-launch form instantiated per hidden_dim (128 instead of 254 registers: four workgroups per CU) -- kept", {'DK': 'per-hidden_dim instantiation', 'OLD': 'run-time switch over hidden_dim'})),
])


def main():
    print("Round 4 A/B measurements (MI355X, one gpurun box per block, variants interleaved in ONE process sequence by scripts/ab_lib2.sh:")
    print("200 timed steps of `bench.py --steps 200 --warmup 20`, REPS repetitions; kernels_us = HIP-event per-launch times of the last repetition).")
    print("Boxes differ by up to 5 % (shader clock under load, see DESIGN 4g): compare rows INSIDE a block only.\n")
    for d, (title, names) in LABELS.items():
        groups = collections.OrderedDict()
        for f in sorted(glob.glob(f'gpurun_out/{d}/*.json')):
            name = os.path.basename(f)[:-5]
            m = re.match(r'(.+)_(\d+)$', name)
            try:
                j = json.loads(open(f).read().strip().splitlines()[-1])
            except Exception:
                continue
            if 'ms_per_step' not in j:
                continue
            groups.setdefault(m.group(1) if m else name, []).append((j['ms_per_step'], j.get('kernels_us', {})))
        if not groups:
            continue
        print(f"== {d}: {title}")
        for k in names:
            if k not in groups:
                continue
            v = groups[k]
            ms = ", ".join(f"{x[0]:.4f}" for x in v)
            ku = v[-1][1]
            print(f"  {names[k]:52s} ms/step {ms}")
            print("      " + "  ".join(f"{kk.split('[')[0]}{'[fusion]' if 'fusion' in kk else ''} {vv}" for kk, vv in ku.items()))
        print()


SEC = collections.OrderedDict([
    ('r5c', ("classification heads pool the wide towers' outputs themselves (m2m_head.tokens, M2M_HEADS_POOL) vs the token-mean launches; MLPOLD = the MLP product chains before their operand reads were batched",
             {'POOL': 'heads pool (default)', 'SEP': 'token-mean launches', 'MLPOLD': 'heads pool, old MLP chains'})),
    ('r5d', ("MIMIC-H: the static MLP as extra workgroups of the time tower's token-mixing launches (M2M_MLP_RIDE)", {'RIDE': 'MLP rides (default)', 'NORIDE': "MLP's own two launches"})),
    ('r5e', ("MIMIC-H: the input projection's weight gradient in the towers' weight-gradient launch (M2M_MIMIC_EMBED_WGRAD_MERGED)", {'MERGED': 'merged (default)', 'SEP': 'launch of its own'})),
    ('r5g', ("one-launch update: the embeddings' slots with 16-byte row accesses instead of per-lane 4-byte gathers", {'VEC': '16-byte accesses', 'OLD': 'previous build'})),
    ('r5o', ("one-launch update: W2 in 8-row x 512-column tiles (M2M_AP_ROWTILES; MIMIC-H's towers are narrower than a tile: unchanged)", {'RT': 'row tiles (default)', 'CT': '32-column-group tiles'})),
])


def secondary():
    print("Secondary configurations at their cfg batches (scripts/ab_sec.sh: scripts/bench_configs.py --cfg-batch-only, 200 steps, three interleaved repetitions; ms per step):\n")
    for tag, (title, names) in SEC.items():
        print(f"== {tag}: {title}")
        for k, label in names.items():
            rows = collections.OrderedDict()
            for f in sorted(glob.glob(f'gpurun_out/{tag}_{k}_*.jsonl')):
                for l in open(f):
                    if l.startswith('{'):
                        d = json.loads(l)
                        rows.setdefault(f"{d['metric'].split()[2]} B={d['batch']}", []).append(d['ms_per_step'])
            if rows:
                print(f"  {label:32s} " + "   ".join(f"{m}: " + ", ".join(f"{v:.4f}" for v in vs) for m, vs in rows.items()))
        print()


if __name__ == "__main__":
    main()
    secondary()
