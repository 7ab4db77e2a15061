#!/usr/bin/env python3
"""time the double-precision rollout on the bench workloads (usage: python tools/f64_time.py [workload ...])"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neuraloc_amd as na
dev = torch.device("cuda:0")
for wl in (sys.argv[1:] or ["swarm50", "singlequad", "softcorridor"]):
    meta, sd, xtarget, xInit = bench.load_workload(wl)
    net, prob = bench.build_objects(meta, sd, xtarget, dev)
    x = bench.make_states(meta, xInit, meta["n_full"], 200).to(dev)
    with torch.no_grad():
        J32, _ = na.OCflow(x, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
    net = net.to(torch.float64); prob.xtarget = prob.xtarget.to(torch.float64); x64 = x.to(torch.float64)
    with torch.no_grad():
        J64, _ = na.OCflow(x64, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): na.OCflow(x64, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"workload": wl, "n": x.shape[0], "nt": meta["nt"], "f64_ms": dt * 1e3, "traj_per_s": x.shape[0] / dt,
                      "Jc_f64": float(J64), "Jc_f32": float(J32), "rel_diff": abs(float(J64) - float(J32)) / abs(float(J64))}))
