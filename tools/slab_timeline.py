#!/usr/bin/env python3
"""Diagnostic (-DNOCF_STAMPS build, tools/build_stamps.sh): per-wave timeline of ONE evaluation (step 40, stage 1) of one workgroup of the
one-CU kernel (singlequad), in shader cycles relative to the evaluation's start.  python tools/slab_timeline.py [mono]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("NOCF_LIB_PATH", os.path.join(REPO, "neuraloc_amd", "csrc", "libnocf_stamps.so"))

import torch                                   # noqa: E402
import bench                                   # noqa: E402
import neuraloc_amd as na                      # noqa: E402
from neuraloc_amd import _lib                  # noqa: E402

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
    return mono()                               # (the slab kernel this tool was written for is gone: section 3.1c of DESIGN.md)


if __name__ == "__main__":
    main()
