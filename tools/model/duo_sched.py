#!/usr/bin/env python3
"""Back-of-the-envelope schedule model of the split-role kernel: two sequential programs (role A, role B) share one SIMD whose vector ALU
executes EITHER an fp32 MFMA stream OR ordinary VALU work (tools/micro/mfma_valu_overlap.hip: a co-resident wave makes no VALU progress while
the other one streams v_mfma_f32_16x16x4_f32), exchange hops are pure delays.  All 8 members of a group are taken as symmetric.
Prints the steady-state cycles per evaluation (all tiles) for a few static orders of the per-tile phases.   python tools/model/duo_sched.py"""
import itertools

HOP = 2500          # store -> visible to the consumer's poll (L2), incl. the poll's round trip
# per phase and tile: (cycles the vector ALU is held: MFMAs at 32 + VALU instructions, further latency of the role itself: LDS, store
# acknowledgements, barriers -- the other role may use the ALU meanwhile).  From the timeline of one evaluation (tools/duo_timeline.py)
# and the instruction counters (tools/duo_pmc.sh), swarm50.
W = dict(O=(700, 800), P1=(1850, 650), C=(1200, 700), P2=(4600, 1700), X=(2200, 800), P3=(4600, 1000), P4=(1500, 1500))
# dependencies: phase -> (producer phase, epoch offset, hop?)
DEP = dict(O=("P4", -1, True), P1=("O", 0, True), C=("P1", 0, False), P2=("P1", 0, True), X=("O", 0, True), P3=("P2", 0, True), P4=("P3", 0, False))


def simulate(orderA, orderB, epochs=40, NT=2):
    """orderA / orderB: list of (phase, tile) per epoch.  Non-preemptive: a phase, once started, holds the ALU until done."""
    done = {}                                # (phase, tile, epoch) -> finish time
    for t in range(NT):
        done[("P4", t, -1)] = 0.0
    progs = {"A": orderA, "B": orderB}
    pc = {"A": 0, "B": 0}                    # index into the unrolled program
    alu_free = 0.0
    tA = {"A": 0.0, "B": 0.0}                # when the role's previous phase finished
    total = {r: len(progs[r]) * epochs for r in progs}
    epoch_end = []
    while any(pc[r] < total[r] for r in progs):
        # candidate next phase of every role and the time its inputs are there
        cand = []
        for r in progs:
            if pc[r] >= total[r]:
                continue
            e, i = divmod(pc[r], len(progs[r]))
            ph, t = progs[r][i]
            prod, de, hop = DEP[ph]
            key = (prod, t, e + de)
            if key not in done:
                continue
            ready = max(done[key] + (HOP if hop else 0), tA[r])
            cand.append((max(ready, alu_free), ready, r, ph, t, e))
        if not cand:
            raise RuntimeError("deadlock: %r" % pc)
        cand.sort()
        start, ready, r, ph, t, e = cand[0]
        alu_free = start + W[ph][0]
        fin = alu_free + W[ph][1]
        done[(ph, t, e)] = fin
        tA[r] = fin
        pc[r] += 1
        if ph == "P4" and t == NT - 1:
            epoch_end.append(fin)
    # steady state: mean of the last epochs
    k = len(epoch_end)
    return (epoch_end[-1] - epoch_end[k // 2]) / (k - 1 - k // 2)


def main():
    NT = 2
    cur_A = [(p, t) for t in range(NT) for p in ("O", "P1", "C", "P2")]
    cur_B = [(p, t) for t in range(NT) for p in ("X", "P3", "P4")]
    print("shipped order          A:", cur_A, "\n                       B:", cur_B, "\n   -> %.0f cycles per evaluation" % simulate(cur_A, cur_B))
    work = sum(a for a, _ in W.values()) * NT
    print("ALU work per evaluation (both roles, %d tiles): %d; one tile's dependency cycle: %d" % (NT, work, sum(sum(W[p]) for p in ("O", "P1", "P2", "P3", "P4")) + 4 * HOP))
    best = []
    phasesA = [(p, t) for t in range(NT) for p in ("O", "P1", "C", "P2")]
    phasesB = [(p, t) for t in range(NT) for p in ("X", "P3", "P4")]

    def valid(order, chains):
        pos = {x: i for i, x in enumerate(order)}
        return all(pos[(a, t)] < pos[(b, t)] for t in range(NT) for a, b in chains)
    candA = [o for o in itertools.permutations(phasesA) if valid(o, [("O", "P1"), ("P1", "C"), ("P1", "P2")]) and o[0] == ("O", 0)]
    candB = [o for o in itertools.permutations(phasesB) if valid(o, [("P3", "P4")])]
    for oa in candA:
        for ob in candB:
            try:
                best.append((simulate(list(oa), list(ob), epochs=24), oa, ob))
            except RuntimeError:
                pass
    best.sort(key=lambda x: x[0])
    for v, oa, ob in best[:8]:
        print("%.0f  A: %s  B: %s" % (v, " ".join(p + str(t) for p, t in oa), " ".join(p + str(t) for p, t in ob)))


if __name__ == "__main__":
    main()
