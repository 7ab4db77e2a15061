#!/usr/bin/env python3
"""Diagnostic (-DNOCF_STAMPS build, tools/build_stamps.sh): per-wave timeline of ONE evaluation of group 0 / member 0 of the
split-role kernel (nocf_duo.hip), both role workgroups, in shader cycles relative to the earliest stamp.
   python tools/duo_timeline.py [n]"""
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

A_PT = {0: "own: entry", 1: "own: partial gradients valid", 2: "own: RK done, S stored", 3: "own: costs done",
        4: "P1: at the S gather", 5: "P1: S staged (barrier passed)", 6: "P1: gemm done", 7: "P1: epilogue + stores", 18: "P1: stores acknowledged (NOCF_DUO_DBG=4)",
        8: "at the U gather", 9: "costs, z / A^T z done (owner waves)", 10: "P2: U staged (barrier passed)", 11: "P2: gemm done", 16: "P2: bias / w read", 17: "P2: tanh done", 12: "P2: V stored"}
B_PT = {20: "tile entry", 21: "own states loaded + barrier", 22: "pair sums + barrier", 23: "QW stored",
        24: "P3: at the V gather", 25: "P3: V staged (barrier passed)", 26: "P3: gemm done", 27: "P3: th valid, y written",
        28: "P3: barrier passed", 29: "P4: done (G stored)"}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    meta, sd, xtarget, xInit = bench.load_workload("swarm50")
    dev = torch.device("cuda:0")
    net, prob = bench.build_objects(meta, sd, xtarget, dev)
    x = bench.make_states(meta, xInit, n, 200).to(dev)
    buf = torch.zeros(64 * 128 + 4096, dtype=torch.int64, device=dev)
    assert _lib.lib().nocf_debug_set_stamp_buffer(buf.data_ptr()) == 0, "this is not the NOCF_STAMPS build"
    with torch.no_grad():
        for _ in range(2):
            buf.zero_()
            na.OCflow(x, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
        torch.cuda.synchronize()
    print("kernel:", _lib.lib().nocf_last_rollout_kernel().decode())
    tall = buf[:64 * 128].view(8, 2, 4, 128).cpu()          # [member][role][wave][point]
    tl = tall[0]
    t0 = int(tl[tl > 0].min())
    ngmax = 16 if os.environ.get("NOCF_DUO_G", "16") == "16" else 32           # groups per launch: 16 of 32 workgroups (fine form) or 32 of 16
    nt_tiles = max(1, min(4, (((n + 15) // 16) + ngmax - 1) // ngmax))
    rows = []
    for t in range(nt_tiles):
        for i, name in A_PT.items():
            rows.append(("A", 40 * t + i, f"A tile {t}: {name}"))
    for t in range(nt_tiles):
        for i, name in B_PT.items():
            rows.append(("B", 40 * t + i, f"B tile {t}: {name}"))
    out = []
    for role, pid, name in rows:
        r = 0 if role == "A" else 1
        vals = [(int(tl[r, w, pid]) - t0) * 24 if int(tl[r, w, pid]) > 0 else -1 for w in range(4)]      # 100 MHz ticks -> 2.4 GHz cycles
        if max(vals) < 0:
            continue
        out.append((min(v for v in vals if v >= 0), name, vals))
    out.sort()
    print(f"n={n}: one evaluation of group 0 / member 0 (2.4 GHz cycles since the earliest stamp, from the 100 MHz chip-wide clock: 24-cycle resolution); -1 = this wave has no such point")
    print(f"{'point':46s}" + "".join(f"   wave{w}" for w in range(4)))
    for _, name, vals in out:
        print(f"{name:46s}" + "".join(f"{v:8d}" for v in vals))
    print("\nowner polls (member, tile, wave): failed polls before the partial gradients were valid; cycles from entry to the first poll's answer")
    for mem in range(8):
        for tt in range(nt_tiles):
            for w in range(4):
                if int(tall[mem, 0, w, 40 * tt + 0]) > 0:
                    print(f"  member {mem} tile {tt} wave {w}: spins {int(tall[mem, 0, w, 40 * tt + 14])}, first answer after {(int(tall[mem, 0, w, 40 * tt + 15]) - int(tall[mem, 0, w, 40 * tt + 0])) * 24} cycles, "
                          f"valid after {(int(tall[mem, 0, w, 40 * tt + 1]) - int(tall[mem, 0, w, 40 * tt + 0])) * 24}")
    print("\nspread over the 8 members of group 0 (earliest / latest wave of any member), same clock:")
    for role, pid, name in rows:
        r = 0 if role == "A" else 1
        v = tall[:, r, :, pid]
        v = v[v > 0]
        if v.numel() == 0:
            continue
        print(f"{name:46s}{(int(v.min()) - t0) * 24:9d}{(int(v.max()) - t0) * 24:9d}   spread {(int(v.max()) - int(v.min())) * 24:6d}")


if __name__ == "__main__":
    main()
