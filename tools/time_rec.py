#!/usr/bin/env python3
"""GPU box: the recording forward and the adjoint of one swarm50 training iteration, kernel times from the library's HIP events.
   python tools/time_rec.py [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NOCF_JIT", "0")
import neuraloc_amd as na                                   # noqa: E402
from neuraloc_amd import _lib                               # noqa: E402
from bench import load_workload, make_states, build_objects  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    meta, sd, xtarget, xInit = load_workload("swarm50")
    net, prob = build_objects(meta, sd, xtarget, dev)
    net.train(); prob.train()
    x = make_states(meta, xInit, meta["n_full"], seed=200).to(dev)
    L = _lib.lib()

    def window(fn):
        torch.cuda.synchronize()
        L.nocf_profile_begin()
        r = fn()
        torch.cuda.synchronize()
        kms, nl = C.c_double(0.0), C.c_int32(0)
        L.nocf_profile_end(C.byref(kms), C.byref(nl))
        return r, kms.value, L.nocf_last_rollout_kernel().decode()

    f, b = [], []
    for i in range(reps + 2):
        net.zero_grad()
        (Jc, _), ms, fk = window(lambda: na.OCflow(x, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"]))
        _, ms2, bk = window(lambda: Jc.backward())
        if i >= 2:
            f.append(ms); b.append(ms2)
    print(f"recording forward {fk}: {sum(f) / len(f):.3f} ms (min {min(f):.3f}); adjoint {bk}: {sum(b) / len(b):.3f} ms (min {min(b):.3f})")


if __name__ == "__main__":
    main()
