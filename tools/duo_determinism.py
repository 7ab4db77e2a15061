#!/usr/bin/env python3
"""Diagnostic: run-to-run determinism of the split-role kernel in both exchange forms (NOCF_DUO_FAST=1 / 0) at one to four tiles per group:
four launches per case, per-sample cost tables compared bit for bit.   python tools/duo_determinism.py [nt]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["NOCF_ENV_WATCH"] = "1"
import torch
import neuraloc_amd as na
from conftest import load_golden
from util_hip import closed_form_normal, make_net, make_prob
DEV = torch.device("cuda:0")
g = load_golden("swarm50"); m = g.meta
net, prob = make_net(g, DEV), make_prob(g, DEV, training=False)
for fast in ("1", "0"):
    os.environ["NOCF_DUO_FAST"] = fast
    for n in (512, 1024, 2048):
        x = (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], 3)).contiguous().to(DEV)
        tabs = []
        for r in range(4):
            with torch.no_grad():
                _, cs = na.OCflow(x, net, prob, [0.0, 1.0], int(sys.argv[1]) if len(sys.argv) > 1 else 10, "rk4", m["alph"], noMean=True)
            torch.cuda.synchronize(); na.check_errors(sync=True)
            tabs.append(torch.cat(cs, 1).cpu())
        line = f"FAST={fast} n={n}:"
        for r in range(1, 4):
            d = (tabs[r] != tabs[0]).any(dim=1)
            cols = (tabs[r] != tabs[0]).any(dim=0).tolist()
            line += f" run{r} rows differing {int(d.sum())} cols {cols} first rows {torch.nonzero(d).flatten()[:6].tolist()};"
        print(line, flush=True)
