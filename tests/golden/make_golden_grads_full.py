#!/usr/bin/env python3
"""Golden PARAMETER GRADIENTS at TRAINING SIZE (trainOC.py:172-174: Jc = OCflow(x0, net, prob, ...); Jc.backward() at
n_train rows): swarm50 n = 1024, nt = 80 and singlequad n = 4096, nt = 50, prob.train(), the BASELINE-size batch of
tests/util_hip.full_states (the one make_golden.py's full/* entries use).

Build container only (imports /root/reference).  Jc is a batch MEAN of per-sample costs (src/OCflow.py:80-90), so the
gradient of the full batch is the row-weighted sum of the gradients of row chunks: the reference's own fp32 autograd
(`grad`) and the oracle's fp64 autograd (`grad64`, the truth the GPU tests bound the HIP error with) are both taken
chunk by chunk (the full batch's autograd graph does not fit this container's memory: SwarmTraj keeps N x N x 3 pair
tensors per evaluation) and summed in float64 in chunk order.  The chunked reference value is cross-checked against
one unchunked reference forward (Jc equal to fp32 rounding) before anything is written.
Output: tests/golden/grads_full.npz
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REF)           # ONLY the reference on the path while its `src` namespace package is imported

from src.Phi import Phi as RefPhi                      # noqa: E402
from src.OCflow import OCflow as RefOCflow             # noqa: E402
from src.initProb import initProb as ref_initProb      # noqa: E402

sys.path.insert(1, REPO)
from oracle import ocflow_oracle as orc                # noqa: E402


def closed_form_normal(n, d, seed):
    """the table of tests/util_hip.closed_form_normal / make_golden.py"""
    i = np.arange(n * d, dtype=np.float64) + 1.0 + 1000.0 * seed
    u1 = np.clip(np.mod(i * 0.6180339887498949, 1.0), 1e-9, 1.0)
    u2 = np.mod(i * 0.7548776662466927 + 0.31, 1.0)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(z.reshape(n, d).astype(np.float32))


def full_states(z, meta, seed):
    xInit = torch.from_numpy(z["xInit"])
    xi = closed_form_normal(meta["n_full"], meta["d"], seed)
    if meta["name"] == "singlequad":
        xi[:, 3:] = 0.0
    x = xInit + meta["var0"] * xi
    x[0] = xInit[0]
    return x.contiguous()


CASES = {"swarm50": dict(chunk32=128, chunk64=64), "singlequad": dict(chunk32=1024, chunk64=512)}


def main():
    only = sys.argv[1:] or list(CASES)
    path = os.path.join(HERE, "grads_full.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name in only:
        cfg = CASES[name]
        z = np.load(os.path.join(HERE, name + ".npz"))
        meta = json.loads(str(z["meta"]))
        alph, nt, n = meta["alph"], meta["nt"], meta["n_full"]
        seed = int(z["full/seed"])
        x = full_states(z, meta, seed)
        assert x.shape[0] == n
        prob, _, _, _ = ref_initProb(name, 4, 4, var0=meta["var0"], alph=alph, cvt=lambda t: t.float())
        prob.train()
        sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
        net = RefPhi(nTh=meta["nTh"], m=meta["m"], d=meta["d"], alph=alph)
        net.load_state_dict(sd)
        net = net.float()
        t0 = time.time()
        # (1) the reference's own fp32 autograd, chunk by chunk; accumulated in float64 in chunk order
        g32 = {k: torch.zeros_like(p, dtype=torch.float64) for k, p in net.named_parameters()}
        J32 = 0.0
        cs32 = np.zeros(7)
        for r0 in range(0, n, cfg["chunk32"]):
            xc = x[r0:r0 + cfg["chunk32"]]
            net.zero_grad()
            Jc, cs = RefOCflow(xc, net, prob, [0.0, 1.0], nt, "rk4", alph)
            wgt = xc.shape[0] / n
            Jc.backward()
            for k, p in net.named_parameters():
                if p.grad is not None:
                    g32[k] += wgt * p.grad.double()
            J32 += wgt * float(Jc)
            cs32 += wgt * np.array([float(c) for c in cs])
            print(f"  {name} fp32 rows {r0}..{r0 + xc.shape[0]}  Jc {float(Jc):.7e}  ({time.time() - t0:.0f} s)", flush=True)
        with torch.no_grad():
            Jall, _ = RefOCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph)
        rel = abs(J32 - float(Jall)) / abs(float(Jall))
        print(f"  {name}: chunked Jc {J32:.9e} vs one reference call {float(Jall):.9e} (rel {rel:.2e})", flush=True)
        assert rel <= 2e-6, "the chunked mean is not the reference's mean"
        # (2) fp64 truth: the oracle in double, same chunking
        S = orc.ProbSpec.from_object(prob).to(torch.float64)
        P64 = orc.PhiParams.from_state_dict({k: v.clone() for k, v in sd.items()}, dtype=torch.float64)
        leaves = [*P64.K, *P64.b, P64.w, P64.A, P64.cw, P64.cb]
        for t in leaves:
            t.requires_grad_(True)
        J64 = 0.0
        cs64 = np.zeros(7)
        acc = [torch.zeros_like(t) for t in leaves]
        for r0 in range(0, n, cfg["chunk64"]):
            xc = x[r0:r0 + cfg["chunk64"]].double()
            for t in leaves:
                t.grad = None
            Jc, cs = orc.rollout(xc, P64, S, [0.0, 1.0], nt, "rk4", alph)
            wgt = xc.shape[0] / n
            Jc.backward()
            for a, t in zip(acc, leaves):
                if t.grad is not None:
                    a += wgt * t.grad
            J64 += wgt * float(Jc)
            cs64 += wgt * np.array([float(c) for c in cs])
            print(f"  {name} fp64 rows {r0}..{r0 + xc.shape[0]}  Jc {float(Jc):.12e}  ({time.time() - t0:.0f} s)", flush=True)
        nTh = meta["nTh"]
        m64 = {"A": acc[2 * nTh + 1], "c.weight": acc[2 * nTh + 2], "c.bias": acc[2 * nTh + 3], "w.weight": acc[2 * nTh]}
        for i in range(nTh):
            m64[f"N.layers.{i}.weight"] = acc[i]
            m64[f"N.layers.{i}.bias"] = acc[nTh + i]
        gap = 0.0
        for k in list(out):
            if k.startswith(name + "/"):
                del out[k]
        for k, gk in g32.items():
            g64 = m64[k].reshape(gk.shape)
            out[f"{name}/grad/{k}"] = gk.float().numpy()
            out[f"{name}/grad64/{k}"] = g64.numpy()
            sc = max(g64.abs().max().item(), 1e-30)
            gp = (gk - g64).abs().max().item() / sc
            print(f"    {k:22s} max|g64| {sc:.4e}  reference fp32-vs-fp64 gap {gp:.2e}")
            gap = max(gap, gp)
        out[f"{name}/Jc"] = np.array(J32, dtype=np.float32)
        out[f"{name}/Jc64"] = np.array(J64)
        out[f"{name}/cs"] = cs32.astype(np.float32)
        out[f"{name}/cs64"] = cs64
        out[f"{name}/nt"] = np.array(nt)
        out[f"{name}/n"] = np.array(n)
        out[f"{name}/seed"] = np.array(seed)
        print(f"{name}: n {n} nt {nt}  Jc32 {J32:.7e}  Jc64 {J64:.10e}  rel {abs(J32 - J64) / abs(J64):.2e}  worst grad gap {gap:.2e}  ({time.time() - t0:.0f} s)", flush=True)
        np.savez_compressed(path, **out)
    print("tests/golden/grads_full.npz written:", sorted({k.split('/')[0] for k in out}))


if __name__ == "__main__":
    main()
