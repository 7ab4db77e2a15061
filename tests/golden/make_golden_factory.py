#!/usr/bin/env python3
"""
Pin the problem/sample factory: for every problem name the REFERENCE's initProb (src/initProb.py:25-249) knows, dump its
xtarget, xInit, r, obstacle, alph_Q/alph_W, nAgents, problem class and a seeded x0 / x0v into tests/golden/factory.npz.
Runs only in the build container (imports /root/reference read-only); the fixture is data.

usage:  python tests/golden/make_golden_factory.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REF)
from src.initProb import initProb as ref_initProb      # noqa: E402  (reference)

NAMES = ["hardcorridor", "midcross2", "midcross20", "midcross30", "midcross4", "singlequad", "softcorridor", "swap12", "swap12_1pair",
         "swap12_2pair", "swap12_3pair", "swap12_4pair", "swap12_5pair", "swap2", "swarm", "swarm50"]
ALPH = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
SEED, NTRAIN, NVAL, VAR0 = 1234, 6, 5, 0.7


def main():
    out, meta = {}, {}
    for name in NAMES:
        torch.manual_seed(SEED)
        try:
            prob, x0, x0v, xInit = ref_initProb(name, NTRAIN, NVAL, var0=VAR0, alph=ALPH, cvt=lambda t: t.float())
        except SystemExit:
            print("reference does not know", name)
            continue
        out[f"{name}/xtarget"] = prob.xtarget.detach().float().numpy()
        out[f"{name}/xInit"] = xInit.detach().float().numpy()
        out[f"{name}/x0"] = x0.detach().float().numpy()
        out[f"{name}/x0v"] = x0v.detach().float().numpy()
        meta[name] = dict(cls=type(prob).__name__, obstacle=prob.obstacle, r=float(getattr(prob, "r", 0.0)), alph_Q=float(prob.alph_Q),
                          alph_W=float(prob.alph_W), nAgents=int(prob.nAgents), d=int(x0.shape[1]),
                          xtarget_shape=list(prob.xtarget.shape))
        print(name, meta[name])
    out["meta"] = np.array(json.dumps(dict(problems=meta, seed=SEED, n_train=NTRAIN, n_val=NVAL, var0=VAR0, alph=ALPH,
                                           torch=torch.__version__)))
    np.savez_compressed(os.path.join(HERE, "factory.npz"), **out)
    print("wrote factory.npz with", len(meta), "problems")


if __name__ == "__main__":
    main()
