#!/usr/bin/env python3
"""time one training iteration (trainOC.py:170-174: zero_grad, Jc = OCflow(...), Jc.backward(), step) through the HIP
forward + hand-written adjoint.  usage: python tools/time_train.py [workload] [reps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuraloc_amd as na                                   # noqa: E402
from bench import load_workload, make_states                # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "swarm50"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    meta, sd, xtarget, xInit = load_workload(wl)
    nt, alph = meta["nt"], meta["alph"]
    net = na.Phi(nTh=meta["nTh"], m=meta["m"], d=meta["d"], alph=alph)
    net.load_state_dict(sd)
    net = net.to(dev)
    if meta["prob_class"] == "Quadcopter":
        prob = na.Quadcopter(xtarget.to(dev), obstacle=None, alph_Q=meta["alph_Q"], alph_W=meta["alph_W"])
    else:
        cls = {"Cross2D": na.Cross2D, "SwarmTraj": na.SwarmTraj}[meta["prob_class"]]
        prob = cls(xtarget.to(dev), obstacle=meta["obstacle"], alph_Q=meta["alph_Q"], alph_W=meta["alph_W"], r=meta["r"])
    x = make_states(meta, xInit, meta["n_full"], seed=200).to(dev)
    net.train(); prob.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    def it():
        opt.zero_grad()
        Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph)
        Jc.backward()
        opt.step()
        return Jc
    it(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        J = it()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t1 = time.perf_counter()
    with torch.no_grad():
        for _ in range(reps):
            na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph)
    torch.cuda.synchronize()
    df = (time.perf_counter() - t1) / reps
    print(json.dumps({"workload": wl, "n": x.shape[0], "nt": nt, "train_iter_ms": dt * 1e3, "forward_only_ms": df * 1e3,
                      "train_traj_per_s": x.shape[0] / dt, "Jc": float(J)}))


if __name__ == "__main__":
    main()
