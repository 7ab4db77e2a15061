#!/usr/bin/env python3
"""quick parity + speed check of the group kernel (NOCF_GROUP=1) against the per-tile kernel on swarm50"""
import os, sys, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch, numpy as np
import bench
import neuraloc_amd as na

meta, sd, xtarget, xInit = bench.load_workload("swarm50")
dev = torch.device("cuda:0")
net = na.Phi(nTh=meta["nTh"], m=meta["m"], d=meta["d"], alph=meta["alph"]); net.load_state_dict(sd); net = net.to(dev).eval()
prob = na.SwarmTraj(xtarget.to(dev), obstacle=meta["obstacle"], alph_Q=meta["alph_Q"], alph_W=meta["alph_W"], r=meta["r"]); prob.eval()
for n, nt in ((16, 2), (40, 4), (1024, 80)):
    x = bench.make_states(meta, xInit, n, 200).to(dev)
    out = {}
    for mode in ("0", "1"):
        os.environ["NOCF_GROUP"] = mode
        with torch.no_grad():
            Jc, cs = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", meta["alph"])
            _, tab = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", meta["alph"], noMean=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", meta["alph"])
            e1.record(); torch.cuda.synchronize()
        out[mode] = (float(Jc), torch.cat(tab, 1).cpu(), e0.elapsed_time(e1) / 3)
    d = (out["0"][1] - out["1"][1]).abs().max().item()
    print(f"n={n} nt={nt}: Jc tile {out['0'][0]:.7e} group {out['1'][0]:.7e}  max|persample diff| {d:.3e}  ms tile {out['0'][2]:.3f} group {out['1'][2]:.3f}")
