#!/usr/bin/env python3
"""
Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, read-only).  It imports
donken/NeuralOC's own modules (src.Phi, src.OCflow, src.initProb), loads the five
pretrained checkpoints, runs the reference on explicit stored inputs and writes
inputs + reference outputs as compressed .npz files.  Nothing of the reference's
source travels: the fixtures are data (weights exported from the checkpoints'
state_dict, input states, expected outputs).

While doing so it pins the oracle: every stored output is also recomputed with
oracle/ocflow_oracle.py and the script asserts bit equality (max|diff| == 0) on
this CPU/torch build, at fixture size and at the full BASELINE.json sizes.

usage:  python tests/golden/make_golden.py            (rewrites tests/golden/*.npz)
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

from src.Phi import Phi as RefPhi                      # noqa: E402  (reference)
from src.OCflow import OCflow as RefOCflow             # noqa: E402
from src.initProb import initProb as ref_initProb      # noqa: E402

sys.path.insert(1, REPO)          # now the repository (oracle/), behind the already-imported reference modules

from oracle import ocflow_oracle as orc                # noqa: E402

torch.set_num_threads(8)

# name -> (nt, n_full, var0)  : BASELINE.json configs with the reference's true d
CONFIGS = {
    "swap2":        dict(nt=20, n_full=1024, var0=1.0),
    "softcorridor": dict(nt=50, n_full=1024, var0=1.0),
    "swap12":       dict(nt=20, n_full=2048, var0=1.0),
    "swarm50":      dict(nt=80, n_full=1024, var0=0.1),
    "singlequad":   dict(nt=50, n_full=4096, var0=0.1),
}
N_FIX = 40          # samples per fixture
N_TRAJ = 6          # samples whose full trajectories are stored


def closed_form_normal(n, d, seed):
    """RNG-free pseudo-normal table (Box-Muller over Weyl sequences) so inputs can be
    regenerated anywhere without trusting a generator's stream."""
    i = np.arange(n * d, dtype=np.float64) + 1.0 + 1000.0 * seed
    u1 = np.mod(i * 0.6180339887498949, 1.0)
    u2 = np.mod(i * 0.7548776662466927 + 0.31, 1.0)
    u1 = np.clip(u1, 1e-9, 1.0)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.reshape(n, d).astype(np.float32)


def make_states(name, xInit, n, var0, seed):
    d = xInit.shape[1]
    xi = torch.from_numpy(closed_form_normal(n, d, seed))
    if name == "singlequad":        # initProb perturbs only x,y,z (src/initProb.py:132-136)
        xi[:, 3:] = 0.0
    x = xInit + var0 * xi
    x[0] = xInit[0]                 # sample 0 is the RNG-free centre (SURVEY 8c known answers)
    return x.contiguous()


def t2n(t):
    return t.detach().cpu().numpy()


def same(a, b, what):
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    if a.dtype != b.dtype:
        b = b.to(a.dtype)
    if not torch.equal(a, b):
        diff = (a.double() - b.double()).abs().max().item()
        raise SystemExit(f"ORACLE != REFERENCE for {what}: max|diff|={diff:g}")


def ref_table(x, net, prob, tspan, nt, stepper, alph):
    _, cs = RefOCflow(x, net, prob, tspan, nt, stepper, alph, noMean=True)
    return torch.cat([c.reshape(-1, 1).to(x.dtype) for c in cs], dim=1)


def run_case(out, tag, x, net, prob, P, tspan, nt, stepper, alph, traj=True):
    """Run the reference, cross-check the oracle bit-for-bit, store under prefix `tag/`."""
    S = orc.ProbSpec.from_object(prob)
    Jc, cs = RefOCflow(x, net, prob, tspan, nt, stepper, alph)
    oJ, ocs = orc.rollout(x, P, S, tspan, nt, stepper, alph)
    same(Jc, oJ, f"{tag} Jc")
    for a, b in zip(cs, ocs):
        same(a, b, f"{tag} cs")
    tab = ref_table(x, net, prob, tspan, nt, stepper, alph)
    same(tab, orc.persample_table(x, P, S, tspan, nt, stepper, alph), f"{tag} persample")
    out[f"{tag}/Jc"] = t2n(Jc)
    out[f"{tag}/cs"] = np.array([c.item() for c in cs], dtype=np.float64).astype(t2n(Jc).dtype)
    out[f"{tag}/persample"] = t2n(tab)
    out[f"{tag}/tspan"] = np.array(tspan, dtype=np.float64)
    out[f"{tag}/nt"] = np.array(nt)
    if traj:
        xs = x[:N_TRAJ]
        zF, cF = RefOCflow(xs, net, prob, tspan, nt, stepper, alph, intermediates=True)
        ozF, ocF = orc.rollout(xs, P, S, tspan, nt, stepper, alph, intermediates=True)
        same(zF, ozF, f"{tag} zFull")
        same(cF, ocF, f"{tag} ctrlFull")
        out[f"{tag}/zFull"] = t2n(zF)
        out[f"{tag}/ctrlFull"] = t2n(cF)


def unit_vectors(out, x, net, prob, P):
    """per-function goldens: Phi value/gradient and the problem physics."""
    S = orc.ProbSpec.from_object(prob)
    d = x.shape[1]
    s = torch.cat((x, torch.full((x.shape[0], 1), 0.37, dtype=x.dtype)), 1)
    v, g = net(s), net.getGrad(s)
    same(v, orc.phi_value(P, s), "phi value")
    same(g, orc.phi_grad(P, s), "phi grad")
    out["unit/s"], out["unit/phi"], out["unit/gradphi"] = t2n(s), t2n(v), t2n(g)
    p = g[:, :d].contiguous()
    for mode in ("eval", "train"):
        prob.eval() if mode == "eval" else prob.train()
        S.training = prob.training
        L, H, Q, W = prob.calcLHQW(x, p)
        gp, ct = prob.calcGradpH(x, p), prob.calcCtrls(x, p)
        oL, oH, oQ, oW = orc.prob_LHQW(S, x, p)
        for a, b, nm in ((L, oL, "L"), (H, oH, "H"), (Q, oQ, "Q"), (W, oW, "W"),
                         (gp, orc.prob_gradpH(S, x, p), "gradpH"), (ct, orc.prob_ctrls(S, x, p), "ctrls")):
            same(torch.as_tensor(a).to(x.dtype).reshape(-1), torch.as_tensor(b).to(x.dtype).reshape(-1),
                 f"unit {mode} {nm}")
        lhqw = torch.cat([torch.as_tensor(t).to(x.dtype).reshape(-1, 1) for t in (L, H, Q, W)], 1)
        out[f"unit/{mode}/LHQW"] = t2n(lhqw)
        out[f"unit/{mode}/gradpH"] = t2n(gp)
        out[f"unit/{mode}/ctrls"] = t2n(ct)
    out["unit/p"] = t2n(p)


def synth_fill(shape, phase, scale):
    """closed-form weight fill, no RNG."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    v = scale * np.sin(0.37 * i + 0.11 * (i % 7) + phase)
    return torch.from_numpy(v.reshape(shape).astype(np.float32))


def export_weights(out, net):
    for k, v in net.state_dict().items():
        out[f"sd/{k}"] = t2n(v)


def main():
    summary = {}
    for idx, (name, cfg) in enumerate(CONFIGS.items()):
        ck = torch.load(f"{REF}/experiments/oc/pretrained/{name}_nn_checkpt.pth",
                        map_location="cpu", weights_only=False)
        a = ck["args"]
        alph = [float(v) for v in a.alph]
        torch.manual_seed(0)
        prob, _, _, xInit = ref_initProb(name, 4, 4, var0=cfg["var0"], alph=alph, cvt=lambda t: t.float())
        d = xInit.shape[1]
        net = RefPhi(nTh=a.nTh, m=a.m, d=d, alph=alph)
        net.load_state_dict(ck["state_dict"])
        net = net.float().eval()
        P = orc.PhiParams.from_module(net)
        out = {}
        export_weights(out, net)
        meta = dict(name=name, d=d, m=a.m, nTh=a.nTh, alph=alph, nt=cfg["nt"], n_full=cfg["n_full"],
                    var0=cfg["var0"], prob_class=type(prob).__name__, obstacle=prob.obstacle,
                    r=prob.r, alph_Q=prob.alph_Q, alph_W=prob.alph_W, n_agents=prob.nAgents,
                    torch=torch.__version__)
        out["xtarget"], out["xInit"] = t2n(prob.xtarget), t2n(xInit)
        x = make_states(name, xInit, N_FIX, cfg["var0"], seed=idx)
        out["x"] = t2n(x)
        with torch.no_grad():
            prob.eval()
            run_case(out, "eval_rk4", x, net, prob, P, [0.0, 1.0], cfg["nt"], "rk4", alph)
            run_case(out, "eval_rk1", x, net, prob, P, [0.0, 1.0], cfg["nt"], "rk1", alph, traj=False)
            run_case(out, "eval_seg", x, net, prob, P, [0.25, 0.9], 7, "rk4", alph)   # shock-style segment
            prob.train()
            run_case(out, "train_rk4", x, net, prob, P, [0.0, 1.0], cfg["nt"], "rk4", alph)
            unit_vectors(out, x, net, prob, P)
            # SURVEY 8(c) known answers: xInit alone at the README eval nt
            prob.eval()
            nt_eval = 80 if name == "swarm50" else 50
            Jc, cs = RefOCflow(xInit, net, prob, [0.0, 1.0], nt_eval, "rk4", alph)
            out["xinit_eval/Jc"] = t2n(Jc)
            out["xinit_eval/cs"] = np.array([c.item() for c in cs], dtype=np.float32)
            out["xinit_eval/nt"] = np.array(nt_eval)
            # fp64 yardstick of the same reference on the same inputs
            net64 = RefPhi(nTh=a.nTh, m=a.m, d=d, alph=alph)
            net64.load_state_dict(ck["state_dict"])
            net64 = net64.double().eval()
            prob64, _, _, _ = ref_initProb(name, 4, 4, var0=cfg["var0"], alph=alph, cvt=lambda t: t.double())
            prob64.eval()
            J64, cs64 = RefOCflow(x.double(), net64, prob64, [0.0, 1.0], cfg["nt"], "rk4", alph)
            out["eval_rk4_f64/Jc"] = t2n(J64)
            out["eval_rk4_f64/cs"] = np.array([c.item() for c in cs64], dtype=np.float64)
            out["eval_rk4_f64/persample"] = t2n(ref_table(x.double(), net64, prob64, [0.0, 1.0], cfg["nt"], "rk4", alph))
            # full BASELINE size: oracle must equal the reference there too (not stored, only checked)
            xf = make_states(name, xInit, cfg["n_full"], cfg["var0"], seed=100 + idx)
            S = orc.ProbSpec.from_object(prob)
            Jf, csf = RefOCflow(xf, net, prob, [0.0, 1.0], cfg["nt"], "rk4", alph)
            oJf, ocsf = orc.rollout(xf, P, S, [0.0, 1.0], cfg["nt"], "rk4", alph)
            same(Jf, oJf, f"{name} full Jc")
            for u, v in zip(csf, ocsf):
                same(u, v, f"{name} full cs")
            out["full/Jc"] = t2n(Jf)
            out["full/cs"] = np.array([c.item() for c in csf], dtype=np.float32)
            out["full/seed"] = np.array(100 + idx)
        out["meta"] = np.array(json.dumps(meta))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        summary[name] = dict(Jc_eval=float(out["eval_rk4/Jc"]), Jc_xinit=float(out["xinit_eval/Jc"]),
                             Jc_full=float(out["full/Jc"]))
        print(name, summary[name], flush=True)

    # synthetic deeper networks (nTh=3,4) so the multi-layer recursion of Phi.getGrad is pinned too
    for name, data, nTh, m, nt in (("synth_nTh3_midcross4", "midcross4", 3, 24, 12),
                                   ("synth_nTh4_softcorridor", "softcorridor", 4, 20, 10),
                                   ("synth_nTh3_swarm", "swarm", 3, 40, 8)):
        alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
        prob, _, _, xInit = ref_initProb(data, 4, 4, var0=1.0, alph=alph, cvt=lambda t: t.float())
        xInit = xInit.float()
        d = xInit.shape[1]
        net = RefPhi(nTh=nTh, m=m, d=d, alph=alph).float().eval()
        sd = net.state_dict()
        for j, (k, v) in enumerate(sd.items()):
            fan = v.shape[-1] if v.dim() > 1 else 4
            sd[k] = synth_fill(tuple(v.shape), 0.3 * j, 0.8 / np.sqrt(fan))
        sd["w.weight"] = sd["w.weight"] + 1.0
        net.load_state_dict(sd)
        P = orc.PhiParams.from_module(net)
        out = {}
        export_weights(out, net)
        var0 = 0.4 if data == "swarm" else 1.0
        x = make_states(data, xInit, N_FIX, var0, seed=50 + nTh)
        out["x"], out["xtarget"], out["xInit"] = t2n(x), t2n(prob.xtarget), t2n(xInit)
        meta = dict(name=name, data=data, d=d, m=m, nTh=nTh, alph=alph, nt=nt, var0=var0,
                    prob_class=type(prob).__name__, obstacle=prob.obstacle, r=prob.r,
                    alph_Q=prob.alph_Q, alph_W=prob.alph_W, n_agents=prob.nAgents, torch=torch.__version__)
        with torch.no_grad():
            prob.eval()
            run_case(out, "eval_rk4", x, net, prob, P, [0.0, 1.0], nt, "rk4", alph)
            prob.train()
            run_case(out, "train_rk4", x, net, prob, P, [0.0, 1.0], nt, "rk4", alph)
            unit_vectors(out, x, net, prob, P)
        out["meta"] = np.array(json.dumps(meta))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, float(out["eval_rk4/Jc"]), float(out["train_rk4/Jc"]), flush=True)

    with open(os.path.join(HERE, "SUMMARY.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("oracle == reference bit-for-bit on every stored output; fixtures written.")


if __name__ == "__main__":
    main()
