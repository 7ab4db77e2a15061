#!/usr/bin/env python3
"""Golden PARAMETER GRADIENTS of the rollout, from the reference's own autograd (trainOC.py:172-173:
Jc = OCflow(...); Jc.backward()), for the next scope row (the backward of the rollout).

Build container only (imports /root/reference).  For each pretrained checkpoint: prob.train(), the fixture
states of tests/golden/<name>.npz, a short rollout, dJc/dtheta for every parameter.  The oracle's autograd is
cross-checked against the reference's (same forward bits, backward equal to fp32 rounding) before anything is written.  Output: tests/golden/grads.npz
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REF)           # ONLY the reference on the path while its `src` namespace package is imported:
                                  # this repository's own `src/` shims (a regular package) would win otherwise

from src.Phi import Phi as RefPhi                      # noqa: E402
from src.OCflow import OCflow as RefOCflow             # noqa: E402
from src.initProb import initProb as ref_initProb      # noqa: E402

sys.path.insert(1, REPO)          # now the repository (oracle/), behind the already-imported reference modules
from oracle import ocflow_oracle as orc                # noqa: E402

NT = {"swap2": 8, "softcorridor": 8, "swap12": 12, "swarm50": 20, "singlequad": 8}    # 20 = trainOC.py:29 default
NS = 16


def main():
    out = {}
    for name, nt in NT.items():
        z = np.load(os.path.join(HERE, name + ".npz"))
        meta = json.loads(str(z["meta"]))
        alph = meta["alph"]
        prob, _, _, _ = ref_initProb(name, 4, 4, var0=meta["var0"], alph=alph, cvt=lambda t: t.float())
        prob.train()
        sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
        net = RefPhi(nTh=meta["nTh"], m=meta["m"], d=meta["d"], alph=alph)
        net.load_state_dict(sd)
        net = net.float()
        x = torch.from_numpy(z["x"])[:NS]
        Jc, cs = RefOCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph)
        Jc.backward()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        # oracle autograd must agree exactly (same ops, same order)
        P = orc.PhiParams.from_state_dict({k: v.clone() for k, v in sd.items()})
        leaves = [*P.K, *P.b, P.w, P.A, P.cw, P.cb]
        for t in leaves:
            t.requires_grad_(True)
        S = orc.ProbSpec.from_object(prob)
        oJ, _ = orc.rollout(x, P, S, [0.0, 1.0], nt, "rk4", alph)
        oJ.backward()
        omap = {"A": P.A.grad, "c.weight": P.cw.grad, "c.bias": P.cb.grad, "w.weight": P.w.grad}
        for i in range(meta["nTh"]):
            omap[f"N.layers.{i}.weight"] = P.K[i].grad
            omap[f"N.layers.{i}.bias"] = P.b[i].grad
        assert torch.equal(Jc.detach(), oJ.detach()), name
        worst = 0.0
        for k, gk in grads.items():
            ok = omap[k] if omap[k] is not None else torch.zeros_like(gk)
            # forward values are bit-identical; gradient accumulation order differs between the two autograd graphs
            # (module vs functional), so the backward agrees to fp32 rounding only
            rel = (gk - ok).abs().max().item() / max(gk.abs().max().item(), 1e-30)
            if rel > 2e-5:
                raise SystemExit(f"oracle autograd != reference autograd for {name} {k}: rel {rel:g}")
            worst = max(worst, rel)
            out[f"{name}/grad/{k}"] = gk.numpy()
        # fp64 truth (oracle in double): lets the GPU tests bound the HIP error by the reference's own fp32-vs-fp64 gap
        P64 = orc.PhiParams.from_state_dict({k: v.clone() for k, v in sd.items()}, dtype=torch.float64)
        for t in [*P64.K, *P64.b, P64.w, P64.A, P64.cw, P64.cb]:
            t.requires_grad_(True)
        J64, _ = orc.rollout(x.double(), P64, S.to(torch.float64), [0.0, 1.0], nt, "rk4", alph)
        J64.backward()
        m64 = {"A": P64.A.grad, "c.weight": P64.cw.grad, "c.bias": P64.cb.grad, "w.weight": P64.w.grad}
        for i in range(meta["nTh"]):
            m64[f"N.layers.{i}.weight"] = P64.K[i].grad
            m64[f"N.layers.{i}.bias"] = P64.b[i].grad
        gap = 0.0
        for k, gk in grads.items():
            g64 = m64[k] if m64[k] is not None else torch.zeros_like(gk, dtype=torch.float64)
            out[f"{name}/grad64/{k}"] = g64.numpy()
            gap = max(gap, (gk.double() - g64).abs().max().item() / max(g64.abs().max().item(), 1e-30))
        out[f"{name}/Jc64"] = J64.detach().numpy()
        print(name, "reference fp32 vs fp64: Jc rel %.2e, worst grad rel %.2e" % (abs(float(Jc) - float(J64)) / abs(float(J64)), gap))
        out[f"{name}/Jc"] = Jc.detach().numpy()
        out[f"{name}/nt"] = np.array(nt)
        out[f"{name}/ns"] = np.array(NS)
        print(name, float(Jc), "max rel oracle-vs-reference grad diff %.2e" % worst, {k: float(v.abs().max()) for k, v in grads.items()}, flush=True)
    np.savez_compressed(os.path.join(HERE, "grads.npz"), **out)
    print("oracle autograd == reference autograd to fp32 rounding (<= 2e-5 of max|grad|); tests/golden/grads.npz written")


if __name__ == "__main__":
    main()
