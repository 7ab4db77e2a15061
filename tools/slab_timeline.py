#!/usr/bin/env python3
"""Diagnostic (-DNOCF_STAMPS build): per-wave timeline of ONE evaluation (step 40, stage 1) of one workgroup of the slab kernel,
in shader cycles relative to the evaluation's start.
  hipcc ... -DNOCF_JIT_ONLY -DNOCF_STAMPS=1 -o neuraloc_amd/csrc/libnocf_stamps.so ...; NOCF_SLAB=2 python tools/slab_timeline.py [n]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("NOCF_LIB_PATH", os.path.join(REPO, "neuraloc_amd", "csrc", "libnocf_stamps.so"))

import torch                                   # noqa: E402
import bench                                   # noqa: E402
import neuraloc_amd as na                      # noqa: E402
from neuraloc_amd import _lib                  # noqa: E402

PT = {1: "S gathered (barrier passed)", 2: "P1 gemm done", 3: "P1 z + epilogue + stores", 5: "U: at the gather", 6: "U gathered (barrier passed)",
      7: "P2 gemm done", 8: "P2 epilogue + stores", 10: "V: at the gather", 11: "V gathered (barrier passed)", 12: "zf + P3 gemm done",
      13: "y + barrier", 14: "P4 done (stores issued)", 16: "reduce: entry", 17: "reduce: partials valid", 18: "reduce: done (S stores issued)"}
GLOBAL = {0: "evaluation entry", 50: "x-only costs done", 51: "z sums done", 52: "cost partials + barrier", 53: "cost: wave sums",
          54: "cost: integrals", 55: "copy done (end)"}


MONO = {0: "evaluation entry", 1: "s fragments + barrier (skipped after an RK stage)", 2: "P1 + z + epilogue + barrier", 3: "P2 gemm", 4: "P2 epilogue + barrier",
        5: "z rows + P3 gemm", 6: "y + barrier", 7: "P4 partials + barrier", 8: "g rows (final / control evaluations only)", 9: "physics: x, p read",
        10: "physics: sin/cos shared", 11: "physics scalars + RK update", 12: "end barrier"}


def mono():
    meta, sd, xtarget, xInit = bench.load_workload("singlequad")
    dev = torch.device("cuda:0")
    net, prob = bench.build_objects(meta, sd, xtarget, dev)
    x = bench.make_states(meta, xInit, meta["n_full"], 200).to(dev)
    buf = torch.zeros(264 * 12 + 8 * 64, dtype=torch.int64, device=dev)
    assert _lib.lib().nocf_debug_set_stamp_buffer(buf.data_ptr()) == 0, "this is not the NOCF_STAMPS build"
    with torch.no_grad():
        for _ in range(2):
            na.OCflow(x, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
        torch.cuda.synchronize()
    tl = buf[264 * 12:].view(8, 64).cpu()
    t0 = int(tl[:4, 0][tl[:4, 0] > 0].min())
    print("singlequad, mono kernel: one evaluation of workgroup 9 (cycles since its entry)")
    print(f"{'point':44s}" + "".join(f"   wave{w}" for w in range(4)) + "   delta(w0)")
    prev = 0
    for pid in range(13):
        row = [int(tl[w, pid]) - t0 if int(tl[w, pid]) > 0 else -1 for w in range(4)]
        print(f"{MONO[pid]:44s}" + "".join(f"{v:8d}" for v in row) + f"{row[0] - prev:10d}")
        prev = row[0]


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "mono":
        return mono()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    meta, sd, xtarget, xInit = bench.load_workload("swarm50")
    dev = torch.device("cuda:0")
    net, prob = bench.build_objects(meta, sd, xtarget, dev)
    x = bench.make_states(meta, xInit, n, 200).to(dev)
    buf = torch.zeros(264 * 12 + 8 * 64, dtype=torch.int64, device=dev)
    assert _lib.lib().nocf_debug_set_stamp_buffer(buf.data_ptr()) == 0, "this is not the NOCF_STAMPS build"
    with torch.no_grad():
        for _ in range(2):
            na.OCflow(x, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
        torch.cuda.synchronize()
    tl = buf[264 * 12:].view(8, 64).cpu()
    t0 = int(tl[:4, 0][tl[:4, 0] > 0].min())
    nt_tiles = 2 if n > 512 else 1
    order = [(0, GLOBAL[0])]
    for t in range(nt_tiles):
        order += [(i + 25 * t, f"tile {t}: {PT[i]}") for i in (1, 2, 3)]
    order += [(50, GLOBAL[50])]
    for t in range(nt_tiles):
        order += [(i + 25 * t, f"tile {t}: {PT[i]}") for i in (5, 6, 7, 8)]
    order += [(51, GLOBAL[51])]
    for t in range(nt_tiles):
        order += [(i + 25 * t, f"tile {t}: {PT[i]}") for i in (10, 11, 12, 13, 14)]
    for t in range(nt_tiles):
        order += [(i + 25 * t, f"tile {t}: {PT[i]}") for i in (16, 17, 18)]
    order += [(i, GLOBAL[i]) for i in (52, 53, 54, 55)]
    print(f"n={n}: one evaluation of workgroup 9 (cycles since its entry)")
    print(f"{'point':44s}" + "".join(f"   wave{w}" for w in range(4)) + "   delta(w0)")
    prev = 0
    for pid, name in order:
        row = [int(tl[w, pid]) - t0 if int(tl[w, pid]) > 0 else -1 for w in range(4)]
        d0 = row[0] - prev if row[0] >= 0 else 0
        if row[0] >= 0:
            prev = row[0]
        print(f"{name:44s}" + "".join(f"{v:8d}" for v in row) + f"{d0:10d}")


if __name__ == "__main__":
    main()
