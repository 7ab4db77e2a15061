#!/usr/bin/env python3
"""Walks a hipcc -S output for the VMEM store-data hazard: a buffer / global / scratch store of MORE THAN 64 BITS whose data registers are
written by a VALU instruction in one of the next two issue slots (the store reads its data over several cycles; CDNA3 ISA guide, "VMEM store
more than 8 bytes followed by a write of the store data: 1 wait state, 2 with an SGPR offset").  hipcc's hazard recognizer does not insert the
wait states for raw buffer stores with an SGPR soffset on gfx950 (round 4: the adjoint's obar stream came out with the y stream's fourth
component in the last quarter of the lanes).  usage: python tools/store_hazard_check.py file.s [more.s ...]; exit code 1 if anything is found"""
import re
import sys

STORE = re.compile(r"^(buffer_store_dwordx[34]|global_store_dwordx[34]|scratch_store_dwordx[34]|flat_store_dwordx[34])\s+(.*)$")
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = REG.search(tok)
    if not m:
        return set()
    if m.group(1):
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def main():
    bad = 0
    for path in sys.argv[1:]:
        fn = None
        body = []
        for ln in open(path):
            t = ln.strip()
            if ln.startswith("_Z") and t.endswith(":") or (ln.startswith("_Z") and ":" in t.split(";")[0]):
                fn = t.split(":")[0]
                body = []
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            body.append(t)
            if len(body) < 2:
                continue
            # look back: is one of the previous two instructions a wide store whose data this VALU writes?
            if not t.startswith("v_") or t.startswith("v_mfma") and False:
                continue
            dst = regs(t.split(",")[0])
            if not dst:
                continue
            for back in (1, 2):
                if len(body) <= back:
                    break
                prev = body[-1 - back]
                m = STORE.match(prev)
                if not m:
                    if not prev.startswith("s_nop"):
                        continue
                    break                                   # an s_nop in between: wait states were inserted
                ops = m.group(2).split(",")
                data = regs(ops[0]) if prev.startswith("buffer_store") else regs(ops[1] if len(ops) > 1 else "")
                if prev.startswith("scratch_store"):
                    data = regs(ops[1] if len(ops) > 1 else "")
                if dst & data:
                    bad += 1
                    print(f"{path}: {fn}: `{prev[:70]}` then ({back} slot(s) later) `{t[:70]}` overwrites store data v{sorted(dst & data)}")
    print(f"{bad} store-data hazard(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
