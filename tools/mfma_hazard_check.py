#!/usr/bin/env python3
"""Static check of the wait states around hand-written (inline-asm) MFMAs in a gfx950 .s file.

hipcc pads hazards only for instructions it models; an `asm volatile("v_mfma ...")` is opaque to its hazard
recogniser (cdna_hip_programming.md 5.7 item 2), and an `asm volatile("s_nop ...")` statement is NOT ordered
against register-only instructions (item 3), so the scheduler may move a VALU read of an accumulator above
the nops that were meant to cover it.  This script walks every kernel whose name matches the given pattern
and reports, in program order within each basic block,

  (R) a non-MFMA instruction that reads or overwrites a VGPR written by a v_mfma_f32_16x16x4_f32 fewer than
      NEED wait states earlier (8 passes: 12 states; instructions count 1 state each, `s_nop N` counts N+1);
  (W) a v_mfma whose SrcA / SrcB / SrcC VGPR was written by a VALU instruction fewer than 2 states earlier.

Usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -I include -I neuraloc_amd/csrc \
           -o /tmp/k.s neuraloc_amd/csrc/nocf_duo.hip && python tools/mfma_hazard_check.py /tmp/k.s rollout_duo
Exit code 1 when a hazard is found.
"""
import re
import sys

NEED_READ = 12
NEED_WRITE = 2


def regs_of(tok):
    """VGPRs named by one operand token: v12, v[12:15]; AGPRs ignored"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def parse(line):
    line = line.split(";")[0].split("//")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
    ops = [t.split()[0] if t else t for t in ops]            # drop modifiers like "offen", "sc1"
    return op, ops


def check_kernel(name, lines):
    bad = []
    last_mfma_write = {}        # vgpr -> state counter at the mfma
    last_valu_write = {}        # vgpr -> state counter
    state = 0
    for ln, raw in lines:
        p = parse(raw)
        if p is None:
            if raw.strip().endswith(":") and not raw.strip().startswith(";"):   # label: conservatively keep tracking (fallthrough)
                pass
            continue
        op, ops = p
        if op == "s_nop":
            state += int(ops[0], 0) + 1
            continue
        if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm" or op == "s_barrier":
            state += 1
            continue
        is_mfma = op.startswith("v_mfma")
        dst = regs_of(ops[0]) if ops else set()
        srcs = set()
        for t in ops[1:]:
            srcs |= regs_of(t)
        is_store = op.startswith(("buffer_store", "global_store", "ds_write", "flat_store", "scratch_store"))
        if is_store:
            srcs |= dst
            dst = set()
        if is_mfma:
            for r in srcs:
                if r in last_valu_write and state - last_valu_write[r] < NEED_WRITE:
                    bad.append((ln, "W", raw.strip(), "v%d written %d states earlier" % (r, state - last_valu_write[r])))
            # an accumulate chain (SrcC == vDst of an earlier MFMA) is interlocked by the hardware: no check
            for r in dst:
                last_mfma_write[r] = state
                last_valu_write.pop(r, None)
        else:
            for r in srcs | dst:
                if r in last_mfma_write and state - last_mfma_write[r] < NEED_READ:
                    bad.append((ln, "R", raw.strip(), "v%d written by an MFMA %d states earlier" % (r, state - last_mfma_write[r])))
            for r in dst:
                last_mfma_write.pop(r, None)
                if op.startswith("v_"):
                    last_valu_write[r] = state
                else:
                    last_valu_write.pop(r, None)
        state += 1
    return bad


def main():
    path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    kernels, cur, name = [], None, None
    for i, raw in enumerate(open(path, errors="replace"), 1):
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            name, cur = m.group(1), []
            kernels.append((name, cur))
            continue
        if cur is not None:
            if raw.strip().startswith(".Lfunc_end"):
                cur = None
                continue
            cur.append((i, raw))
    total = 0
    for name, lines in kernels:
        if pat not in name:
            continue
        nm = sum(1 for _, r in lines if r.strip().startswith("v_mfma"))
        if nm == 0:
            continue
        bad = check_kernel(name, lines)
        print("%s: %d MFMAs, %d hazards" % (name, nm, len(bad)))
        for ln, kind, text, why in bad[:12]:
            print("   line %d [%s] %s   <- %s" % (ln, kind, text, why))
        total += len(bad)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
