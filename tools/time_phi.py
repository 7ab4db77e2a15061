#!/usr/bin/env python3
"""time one grad-Phi evaluation (nocf_phi_grad_f32) at the bench batch size: isolates the GEMM phases"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
import neuraloc_amd as na

wl = sys.argv[1] if len(sys.argv) > 1 else "swarm50"
meta, sd, xtarget, xInit = bench.load_workload(wl)
dev = torch.device("cuda:0")
net = na.Phi(nTh=meta["nTh"], m=meta["m"], d=meta["d"], alph=meta["alph"])
net.load_state_dict(sd); net = net.to(dev).eval()
n = meta["n_full"]
x = bench.make_states(meta, xInit, n, 200).to(dev)
s = torch.cat((x, torch.full((n, 1), 0.3, device=dev)), 1).contiguous()
with torch.no_grad():
    for _ in range(5): g = net.getGrad(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record()
    for _ in range(reps): g = net.getGrad(s)
    e1.record(); torch.cuda.synchronize()
print(f"{wl}: getGrad n={n}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call (includes the pack kernel + launch gaps)")
