#!/usr/bin/env python3
"""Diagnostic (-DNOCF_STAMPS build): per-wave timeline of ONE evaluation of the adjoint kernel (workgroup 7)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("NOCF_LIB_PATH", os.path.join(REPO, "neuraloc_amd", "csrc", "libnocf_stamps.so"))

import torch                                   # noqa: E402
import bench                                   # noqa: E402
import neuraloc_amd as na                      # noqa: E402
from neuraloc_amd import _lib                  # noqa: E402

POINTS = {38: "eval: s loaded", 0: "phi_eval entry", 2: "open barrier", 3: "fwd barrier", 4: "bwd barrier", 5: "close barrier",
          39: "phi_eval done", 41: "physics reduce barrier", 45: "physics done", 46: "cotangents formed", 47: "xgrad done",
          54: "vjp: zb done", 55: "vjp: ybar phase done", 56: "vjp: vbar phase done", 57: "vjp: ubar phase done",
          58: "vjp: closing done", 59: "vjp: rows streamed"}
ORDER = [38, 0, 2, 3, 4, 5, 39, 41, 45, 46, 47, 54, 55, 56, 57, 58, 59]


MONO_POINTS = {0: "eval start", 1: "s rows + fragments", 2: "P1 + epilogue", 3: "barrier", 4: "P2", 5: "P2 epilogue", 6: "barrier", 7: "P3 + epilogue",
               8: "barrier", 9: "P4 partials", 10: "grad Phi rows (2 barriers)", 11: "physics sums", 12: "cotangents (physics adjoint)",
               13: "gbar fragments + barrier", 14: "P1' + epilogue", 15: "barrier", 16: "P2'", 17: "P2' epilogue", 18: "barrier", 19: "P3'",
               20: "P3' epilogue", 21: "barrier", 22: "P4' partials", 23: "barrier + sbar rows", 24: "outer products (dK1, dK0, dM)", 25: "barrier",
               26: "RK recurrences + barrier"}


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "swarm50"
    meta, sd, xtarget, xInit = bench.load_workload(wl)
    dev = torch.device("cuda:0")
    net = na.Phi(nTh=meta["nTh"], m=meta["m"], d=meta["d"], alph=meta["alph"])
    net.load_state_dict(sd)
    net = net.to(dev).train()
    cls = {"Cross2D": na.Cross2D, "SwarmTraj": na.SwarmTraj, "Quadcopter": na.Quadcopter}[meta["prob_class"]]
    kw = {} if meta["prob_class"] == "Quadcopter" else {"r": meta["r"]}
    prob = cls(xtarget.to(dev), obstacle=meta["obstacle"], alph_Q=meta["alph_Q"], alph_W=meta["alph_W"], **kw)
    prob.train()
    n, nt = meta["n_full"], meta["nt"]
    x = bench.make_states(meta, xInit, n, 200).to(dev)
    buf = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
    L = _lib.lib()
    L.nocf_debug_set_timeline_buffer.argtypes = [__import__("ctypes").c_void_p]
    assert L.nocf_debug_set_timeline_buffer(buf.data_ptr()) == 0, "this is not the NOCF_STAMPS build"
    Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", meta["alph"])
    Jc.backward()
    torch.cuda.synchronize()
    tl = buf.view(8, 64).cpu()
    if _lib.lib().nocf_last_rollout_kernel().decode() == "rollout_mono_bwd_kernel":      # the one-CU adjoint (nocf_mono_bwd.inc): 4 waves
        t0 = int(tl[:4, 0].min())
        print(f"{'point (cycles since the start; d = since the previous point, wave 0)':70s}" + "".join(f"   wave{w}" for w in range(4)))
        prev = 0
        for p in range(27):
            row = [int(tl[w, p]) - t0 for w in range(4)]
            print(f"{MONO_POINTS[p]:58s} d {row[0] - prev:6d}  " + "".join(f"{v:8d}" for v in row))
            prev = row[0]
        if int(tl[0, 27]) > 0:
            a, b, c_ = (int(tl[0, q]) - t0 for q in (27, 28, 29))
            print(f"inside the cotangents (wave 0): rows read at {a}, sin / cos back at {b}, components formed at {c_}")
        return
    t0 = int(tl[:, 38][tl[:, 38] > 0].min())
    print(f"{'point':28s}" + "".join(f"   wave{w}" for w in range(8)))
    for p in ORDER:
        row = [int(tl[w, p]) - t0 if int(tl[w, p]) > 0 else -1 for w in range(8)]
        print(f"{POINTS[p]:28s}" + "".join(f"{v:8d}" for v in row))


if __name__ == "__main__":
    main()
