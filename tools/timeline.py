#!/usr/bin/env python3
"""Diagnostic (-DNOCF_STAMPS build): per-wave timeline of ONE evaluation of one workgroup of the rollout kernel
(step 40, stage 1, workgroup 7), in shader cycles relative to the evaluation's start.
  tools/build_stamps.sh && NOCF_LIB_PATH=neuraloc_amd/csrc/libnocf_stamps.so python tools/timeline.py [workload]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("NOCF_LIB_PATH", os.path.join(REPO, "neuraloc_amd", "csrc", "libnocf_stamps.so"))

import torch                                   # noqa: E402
import bench                                   # noqa: E402
import neuraloc_amd as na                      # noqa: E402
from neuraloc_amd import _lib                  # noqa: E402

POINTS = {0: "eval entry", 1: "z=As done", 8: "open: entry", 9: "open: acts in regs", 10: "open: stream end", 11: "open: epilogue end",
          2: "open: barrier passed", 16: "fwd: entry", 17: "fwd: acts in regs", 18: "fwd: stream end", 19: "fwd: epilogue end",
          3: "fwd: barrier passed", 24: "bwd: entry", 25: "bwd: acts in regs", 26: "bwd: stream end", 27: "bwd: epilogue end",
          4: "bwd: barrier passed", 32: "close: entry", 33: "close: acts in regs", 34: "close: stream end",
          35: "close: partials written", 36: "close: split barrier", 37: "close: epilogue end", 5: "close: barrier passed",
          6: "physics: entry (sweep done)", 7: "physics: p^2 done", 14: "physics: obstacle done", 40: "physics: sums done", 41: "physics: reduce barrier", 42: "RK: sweep done", 43: "RK: finish done", 44: "RK: barrier passed"}
ORDER = [0, 1, 8, 9, 10, 11, 2, 16, 17, 18, 19, 3, 24, 25, 26, 27, 4, 32, 33, 34, 35, 36, 37, 5, 6, 7, 14, 40, 41, 42, 43, 44]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "swarm50"
    meta, sd, xtarget, xInit = bench.load_workload(wl)
    dev = torch.device("cuda:0")
    net = na.Phi(nTh=meta["nTh"], m=meta["m"], d=meta["d"], alph=meta["alph"])
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    cls = {"Cross2D": na.Cross2D, "SwarmTraj": na.SwarmTraj, "Quadcopter": na.Quadcopter}[meta["prob_class"]]
    kw = {} if meta["prob_class"] == "Quadcopter" else {"r": meta["r"]}
    prob = cls(xtarget.to(dev), obstacle=meta["obstacle"], alph_Q=meta["alph_Q"], alph_W=meta["alph_W"], **kw)
    prob.eval()
    n, nt = meta["n_full"], meta["nt"]
    x = bench.make_states(meta, xInit, n, 200).to(dev)
    nwg = (n + 3) // 4
    buf = torch.zeros(nwg * 12 + 8 * 64, dtype=torch.int64, device=dev)
    rc = _lib.lib().nocf_debug_set_stamp_buffer(buf.data_ptr())
    assert rc == 0, "this is not the NOCF_STAMPS build"
    with torch.no_grad():
        for _ in range(2):
            na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", meta["alph"])
        torch.cuda.synchronize()
    tl = buf[nwg * 12:].view(8, 64).cpu()
    t0 = int(tl[:, 0][tl[:, 0] > 0].min())
    print(f"{'point':28s}" + "".join(f"   wave{w}" for w in range(8)))
    for p in ORDER:
        row = [int(tl[w, p]) - t0 if int(tl[w, p]) > 0 else -1 for w in range(8)]
        print(f"{POINTS[p]:28s}" + "".join(f"{v:8d}" for v in row))


if __name__ == "__main__":
    main()
