#!/usr/bin/env python3
"""Diagnostic: slab vs tile kernel on a Cross2D problem (PD = 2 instantiation) with a closed-form m = 512 network."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import neuraloc_amd as na
from test_slab_gpu import _synth_state_dict, ALPH
DEV = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "swap12"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 5
stepper = sys.argv[4] if len(sys.argv) > 4 else "rk4"
torch.manual_seed(11)
prob, x0, _, _ = na.initProb(name, n, 8, 0.5, ALPH, lambda t: t.float().to(DEV))
prob.eval()
d = x0.shape[1]
net = na.Phi(nTh=2, m=512, d=d, alph=ALPH); net.load_state_dict(_synth_state_dict(2, 512, d, seed=d % 5)); net = net.to(DEV).eval()
tabs = {}
for slab in ("2", "0"):
    os.environ["NOCF_SLAB"] = slab
    with torch.no_grad():
        _, csn = na.OCflow(x0, net, prob, [0.0, 1.0], nt, stepper, ALPH, noMean=True)
    tabs[slab] = torch.cat(csn, 1).cpu().double()
a, b = tabs["2"], tabs["0"]
off = (a - b).abs() > 1e-3 + 1e-3 * b.abs()
rows = off.any(1).nonzero().flatten().tolist()
print(f"{name} d={d} n={n} nt={nt} {stepper}: rows off {len(rows)}: {rows[:24]}; per column {off.sum(0).tolist()}")
for r in rows[:3]:
    print("   slab", a[r].tolist(), "\n   tile", b[r].tolist())
    dist = ((b - a[r]).abs() / (b.abs() + 1e-3)).max(1).values
    j = int(dist.argmin())
    print("   (closest tile row:", j, "max rel", float(dist[j]), ")")
