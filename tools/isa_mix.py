#!/usr/bin/env python3
"""Static instruction mix of one kernel of a hipcc -S output, by loop depth (the rollout's hot code is depth >= 3 in the split-role kernels):
VALU / MFMA / LDS / VMEM / SALU counts, the SGPR-spill traffic (v_readlane / v_writelane) and the most frequent opcodes.
   python tools/isa_mix.py file.s kernel_name_substring"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    cur = 0
    by = collections.defaultdict(collections.Counter)
    for l in lines[start + 1:end]:
        t = l.strip()
        if t.startswith(".LBB") or (t.startswith(";") and "Depth=" in t):
            m = re.search(r"Depth=(\d+)", l)
            if t.startswith(".LBB"):
                cur = int(m.group(1)) if m else 0
            elif m:
                cur = int(m.group(1))
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        by[cur][t.split()[0]] += 1
    print(lines[start].split(":")[0])
    for d in sorted(by):
        c = by[d]
        tot = sum(c.values())
        mf = sum(n for o, n in c.items() if o.startswith("v_mfma"))
        va = sum(n for o, n in c.items() if o.startswith("v_") and not o.startswith("v_mfma"))
        sp = c["v_readlane_b32"] + c["v_writelane_b32"]
        ds = sum(n for o, n in c.items() if o.startswith("ds_"))
        vm = sum(n for o, n in c.items() if o.startswith("buffer_") or o.startswith("global_") or o.startswith("scratch_"))
        sa = sum(n for o, n in c.items() if o.startswith("s_"))
        print(f"depth {d}: {tot} instructions: MFMA {mf}, other VALU {va} (of them v_readlane/v_writelane {sp}), LDS {ds}, VMEM {vm}, scalar {sa}")
        print("   ", ", ".join(f"{o} {n}" for o, n in c.most_common(14)))


if __name__ == "__main__":
    main()
