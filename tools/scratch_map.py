#!/usr/bin/env python3
"""Where do a kernel's scratch (spill) accesses sit?  Walks the ISA of one function in a hipcc -S output and prints, per basic-block
label, the scratch loads / stores and the MFMA count of that block (a block with MFMAs is a hot GEMM loop body).
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Iinclude -Ineuraloc_amd/csrc -o /tmp/duo.s neuraloc_amd/csrc/nocf_duo.hip
   python tools/scratch_map.py /tmp/duo.s rollout_duo_bwd_kernelILi3E"""
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and key in l and l.rstrip().endswith(tuple(":")) or (l.startswith("_Z") and key in l and ":" in l))
end = next(i for i in range(start, len(txt)) if ".end_amdhsa_kernel" in txt[i] or txt[i].startswith("\t.section\t.rodata"))
lab, blocks, order = "entry", {}, []
for l in txt[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        lab = m.group(1)
    b = blocks.setdefault(lab, {"ld": 0, "st": 0, "mfma": 0, "n": 0})
    if lab not in order:
        order.append(lab)
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    b["n"] += 1
    if t.startswith("scratch_load"):
        b["ld"] += 1
    elif t.startswith("scratch_store"):
        b["st"] += 1
    elif t.startswith("v_mfma"):
        b["mfma"] += 1
tot_l = sum(b["ld"] for b in blocks.values()); tot_s = sum(b["st"] for b in blocks.values())
print(f"{key}: {end - start} lines, scratch loads {tot_l}, stores {tot_s}")
for lab in order:
    b = blocks[lab]
    if b["ld"] or b["st"]:
        print(f"  {lab:14s} instr {b['n']:5d}  mfma {b['mfma']:4d}  scratch_load {b['ld']:3d}  scratch_store {b['st']:3d}")
