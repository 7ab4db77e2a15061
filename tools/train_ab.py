#!/usr/bin/env python3
"""first training iteration of swarm50 (n = 1024, nt = 80) with the slab kernel and with the tile kernel as the recording forward:
Jc and every parameter gradient side by side (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neuraloc_amd as na
dev = torch.device("cuda:0")
meta, sd, xtarget, xInit = bench.load_workload("swarm50")
x = bench.make_states(meta, xInit, meta["n_full"], 200).to(dev)
res = {}
for slab in ("1", "0"):
    os.environ["NOCF_SLAB"] = slab
    net, prob = bench.build_objects(meta, sd, xtarget, dev)
    net.train(); prob.train()
    Jc, cs = na.OCflow(x, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
    Jc.backward()
    res[slab] = (float(Jc), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
print("Jc slab %.9e tile %.9e rel %.2e" % (res["1"][0], res["0"][0], abs(res["1"][0] - res["0"][0]) / abs(res["0"][0])))
for k in res["1"][1]:
    a, b = res["1"][1][k], res["0"][1][k]
    print("%-22s max|grad| %.4e  max abs diff %.3e  rel to max %.2e" % (k, float(b.abs().max()), float((a - b).abs().max()), float((a - b).abs().max() / b.abs().max())))
