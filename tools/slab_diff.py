#!/usr/bin/env python3
"""Diagnostic: slab kernel vs tile kernel per-sample tables on swarm50 (which rows / columns differ)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import neuraloc_amd as na
from conftest import load_golden
from util_hip import closed_form_normal, make_net, make_prob
DEV = torch.device("cuda:0")
g = load_golden("swarm50"); m = g.meta
for training in (False, True):
    net, prob = make_net(g, DEV), make_prob(g, DEV, training=training)
    for n in (512, 1024):
        x = (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], 3)).contiguous().to(DEV)
        tabs = {}
        for slab in ("1", "0"):
            os.environ["NOCF_SLAB"] = slab
            with torch.no_grad():
                _, csn = na.OCflow(x, net, prob, [0.0, 1.0], 6, "rk4", m["alph"], noMean=True)
            tabs[slab] = torch.cat(csn, 1).cpu().double()
        a, b = tabs["1"], tabs["0"]
        off = (a - b).abs() > 1e-3 + 1e-3 * b.abs()
        rows = off.any(1).nonzero().flatten().tolist()
        print(f"training={training} n={n}: rows off {len(rows)}: {rows[:20]}; per column off {off.sum(0).tolist()}; max rel per col {[float(((a[:, j]-b[:, j]).abs()/(b[:, j].abs()+1e-3)).max()) for j in range(7)]}")
        for r in rows[:4]:
            print("   row", r, "slab", [f"{v:.6e}" for v in a[r].tolist()], "\n          tile", [f"{v:.6e}" for v in b[r].tolist()])
