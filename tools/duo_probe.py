#!/usr/bin/env python3
"""Quick parity + timing probe of the split-role kernel (nocf_duo.hip) against the per-tile kernel on the pretrained swarm50
network.  Diagnostics only.   python tools/duo_probe.py [nt] [n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ["NOCF_ENV_WATCH"] = "1"
import torch

import neuraloc_amd as na
from neuraloc_amd import _lib
from conftest import load_golden
from util_hip import closed_form_normal, make_net, make_prob

DEV = torch.device("cuda:0")


def table(x, net, prob, nt, alph):
    with torch.no_grad():
        _, csn = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph, noMean=True)
    torch.cuda.synchronize()
    return torch.cat(csn, 1).cpu(), _lib.lib().nocf_last_rollout_kernel().decode()


def timeit(x, net, prob, nt, alph, reps=5):
    with torch.no_grad():
        for _ in range(2):
            na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def setenv(**kw):
    for k in ("NOCF_DUO", "NOCF_DUO_FAST", "NOCF_DUO_MAP"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)


def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ns = [int(a) for a in sys.argv[2:]] or [16, 100, 512, 1024, 2048]
    g = load_golden("swarm50")
    m = g.meta
    for training in (False, True):
        net, prob = make_net(g, DEV), make_prob(g, DEV, training=training)
        for n in ns:
            x = (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], 3)).contiguous().to(DEV)
            setenv()
            duo, kd = table(x, net, prob, nt, m["alph"])
            duo2, _ = table(x, net, prob, nt, m["alph"])
            setenv(NOCF_DUO=0)
            tile, kt = table(x, net, prob, nt, m["alph"])
            off = ((duo.double() - tile.double()).abs() > 1e-3 + 1e-3 * tile.double().abs()).any(dim=1)
            keep = ~off
            md = max(abs(duo[keep, j].double().mean().item() - tile[keep, j].double().mean().item()) / (abs(tile[keep, j].double().mean().item()) + 1e-9)
                     for j in range(7)) if keep.any() else float("nan")
            cols = [abs(duo[keep, j].double().mean().item() - tile[keep, j].double().mean().item()) / (abs(tile[keep, j].double().mean().item()) + 1e-9) for j in range(7)] if keep.any() else []
            line = f"train={int(training)} n={n:5d} [{kd}] vs [{kt}]: rows off {int(off.sum())}, worst mean rel diff {md:.2e} (cols {' '.join('%.1e' % c for c in cols)}), deterministic {torch.equal(duo, duo2)}, nan {int(torch.isnan(duo).sum())}"
            if not training:
                setenv()
                t_duo = timeit(x, net, prob, nt, m["alph"])
                setenv(NOCF_DUO_MAP=1)
                t_map = timeit(x, net, prob, nt, m["alph"])
                setenv(NOCF_DUO_FAST=0)
                t_wt = timeit(x, net, prob, nt, m["alph"])
                setenv(NOCF_DUO=0)
                t_tile = timeit(x, net, prob, nt, m["alph"])
                line += f" | ms/call duo {t_duo:.3f} (map1 {t_map:.3f}, write-through {t_wt:.3f}) tile {t_tile:.3f}"
            print(line, flush=True)
            if off.any():
                idx = torch.nonzero(off).flatten()[:4].tolist()
                for i in idx:
                    print("   row", i, "duo", [f"{v:.5e}" for v in duo[i].tolist()], "tile", [f"{v:.5e}" for v in tile[i].tolist()])


if __name__ == "__main__":
    main()
