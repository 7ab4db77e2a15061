#!/usr/bin/env python3
"""Diagnostic (-DNOCF_STAMPS build, tools/build_stamps.sh): per-wave timeline of ONE evaluation of group 0 / member 0 of the
split-role ADJOINT kernel (nocf_duo_bwd.inc), both role workgroups, in shader cycles relative to the earliest stamp.
   python tools/duo_bwd_timeline.py [n]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("NOCF_LIB_PATH", os.path.join(REPO, "neuraloc_amd", "csrc", "libnocf_stamps.so"))
os.environ["NOCF_ENV_WATCH"] = "1"
import torch                                   # noqa: E402
import bench                                   # noqa: E402
import neuraloc_amd as na                      # noqa: E402
from neuraloc_amd import _lib                  # noqa: E402

A_PT = {0: "own': entry", 1: "own': partial sbar valid", 2: "own': recurrences done, GB stored",
        4: "P1': at the GB gather", 5: "P1': GB staged (barrier passed)", 6: "P1': gemm done", 7: "P1': epilogue + stores",
        8: "at the AB gather", 10: "P2': AB staged (barrier passed)", 11: "P2': gemm done", 12: "P2': QB stored", 9: "streams + A'(A gbar) done"}
B_PT = {20: "tile entry", 21: "own states scattered + barrier", 22: "forces + barrier", 23: "F stored",
        24: "P3': at the QB gather", 25: "P3': QB staged (barrier passed)", 26: "P3': gemm done", 27: "P3': ybar valid, obar written",
        28: "P3': barrier passed", 29: "P4': done (SP stored)"}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    meta, sd, xtarget, xInit = bench.load_workload("swarm50")
    dev = torch.device("cuda:0")
    net, prob = bench.build_objects(meta, sd, xtarget, dev)
    net.train(); prob.train()
    x = bench.make_states(meta, xInit, n, 200).to(dev)
    buf = torch.zeros(64 * 128 + 4096, dtype=torch.int64, device=dev)
    assert _lib.lib().nocf_debug_set_stamp_buffer(buf.data_ptr()) == 0, "this is not the NOCF_STAMPS build"
    for _ in range(2):
        net.zero_grad()
        Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
        torch.cuda.synchronize()
        buf.zero_()
        Jc.backward()
        torch.cuda.synchronize()
    print("kernel:", _lib.lib().nocf_last_rollout_kernel().decode())
    tall = buf[:64 * 128].view(8, 2, 4, 128).cpu()
    tl = tall[0]
    t0 = int(tl[tl > 0].min())
    if "--brief" in sys.argv:
        def at(role, w, pid):
            v = int(tl[role, w, pid])
            return (v - t0) * 24 if v > 0 else -1
        ow = 0 if at(0, 0, 1) >= 0 else 2
        print(f"n={n} NOCF_DUO_DBG={os.environ.get('NOCF_DUO_DBG', '0')}: own' valid->GB stored {at(0, ow, 2) - at(0, ow, 1)}, GB->P1' staged {at(0, ow, 5) - at(0, ow, 2)}, "
              f"P1' gemm {at(0, 0, 6) - at(0, 0, 5)}, epi {at(0, 0, 7) - at(0, 0, 6)}, AB hop {at(0, 0, 10) - at(0, 0, 8)}, P2' gemm {at(0, 0, 11) - at(0, 0, 10)}, "
              f"QB store {at(0, 0, 12) - at(0, 0, 11)}, QB hop {at(1, 0, 25) - at(0, 0, 12)}, P3' gemm {at(1, 0, 26) - at(1, 0, 25)}, epi {at(1, 0, 27) - at(1, 0, 26)}, "
              f"barrier {at(1, 0, 28) - at(1, 0, 27)}, P4' {at(1, 0, 29) - at(1, 0, 28)}, forces {at(1, 0, 22) - at(1, 0, 20)}; own' valid -> P4' done {at(1, 0, 29) - at(0, ow, 1)}")
        return
    nt_tiles = max(1, min(4, (((n + 15) // 16) + 31) // 32))
    rows = []
    for t in range(nt_tiles):
        for i, name in A_PT.items():
            rows.append(("A", 40 * t + i, f"A' tile {t}: {name}"))
    for t in range(nt_tiles):
        for i, name in B_PT.items():
            rows.append(("B", 40 * t + i, f"B' tile {t}: {name}"))
    out = []
    for role, pid, name in rows:
        r = 0 if role == "A" else 1
        vals = [(int(tl[r, w, pid]) - t0) * 24 if int(tl[r, w, pid]) > 0 else -1 for w in range(4)]
        if max(vals) < 0:
            continue
        out.append((min(v for v in vals if v >= 0), name, vals))
    out.sort()
    print(f"n={n}: one evaluation of group 0 / member 0 of the adjoint (2.4 GHz cycles since the earliest stamp; 24-cycle resolution); -1 = this wave has no such point")
    print(f"{'point':50s}" + "".join(f"   wave{w}" for w in range(4)))
    for _, name, vals in out:
        print(f"{name:50s}" + "".join(f"{v:8d}" for v in vals))
    print("\nspread over the 8 members of group 0 (earliest / latest wave of any member), same clock:")
    for role, pid, name in rows:
        r = 0 if role == "A" else 1
        v = tall[:, r, :, pid]
        v = v[v > 0]
        if v.numel() == 0:
            continue
        print(f"{name:50s}{(int(v.min()) - t0) * 24:9d}{(int(v.max()) - t0) * 24:9d}   spread {(int(v.max()) - int(v.min())) * 24:6d}")


if __name__ == "__main__":
    main()
