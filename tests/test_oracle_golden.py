"""CPU: the oracle against the reference's own outputs (tests/golden/*.npz, produced by
tests/golden/make_golden.py from an import of donken/NeuralOC).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import ocflow_oracle as orc

# same torch build => bit equality was asserted at generation time; across CPU models / BLAS
# kernels the fp32 GEMM summation order may differ, so the portable bar is a tight tolerance.
RTOL, ATOL = 2e-5, 2e-5


def spec_of(g, training):
    m = g.meta
    kind = {"Cross2D": orc.KIND_CROSS2D, "SwarmTraj": orc.KIND_SWARM, "Quadcopter": orc.KIND_QUAD}[m["prob_class"]]
    return orc.ProbSpec(kind=kind, xtarget=g.t("xtarget"), obstacle=m["obstacle"], alph_Q=m["alph_Q"],
                        alph_W=m["alph_W"], r=m["r"], training=training)


def close(a, b, rtol=RTOL, atol=ATOL):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    scale = b.abs().max().clamp_min(1.0)
    assert a.shape == b.shape
    err = (a - b).abs().max().item()
    assert err <= atol * scale + rtol * scale, f"max err {err:g} vs scale {scale.item():g}"


@pytest.mark.parametrize("tag", ["eval_rk4", "eval_rk1", "eval_seg", "train_rk4"])
def test_rollout_costs(golden, tag):
    g = golden
    if not g.has(tag + "/Jc"):
        pytest.skip("case not in this fixture")
    P = orc.PhiParams.from_state_dict(g.state_dict())
    S = spec_of(g, training=tag.startswith("train"))
    stepper = "rk1" if tag.endswith("rk1") else "rk4"
    tspan = [float(v) for v in g[tag + "/tspan"]]
    nt = int(g[tag + "/nt"])
    with torch.no_grad():
        Jc, cs = orc.rollout(g.t("x"), P, S, tspan, nt, stepper, g.meta["alph"])
        tab = orc.persample_table(g.t("x"), P, S, tspan, nt, stepper, g.meta["alph"])
    close(Jc, g[tag + "/Jc"], rtol=1e-4)
    for got, want in zip(cs, g[tag + "/cs"]):
        assert abs(got.item() - want) <= 1e-4 * abs(want) + 1e-6
    # per-sample table: allow isolated mask flips in Q/W columns (discontinuous costs)
    want = torch.from_numpy(g[tag + "/persample"]).double()
    bad = ((tab.double() - want).abs() > 1e-3 * want.abs() + 1e-3).sum().item()
    assert bad <= 2, f"{bad} per-sample entries off"


@pytest.mark.parametrize("tag", ["eval_rk4", "eval_seg", "train_rk4"])
def test_rollout_trajectories(golden, tag):
    g = golden
    if not g.has(tag + "/zFull"):
        pytest.skip("case not in this fixture")
    P = orc.PhiParams.from_state_dict(g.state_dict())
    S = spec_of(g, training=tag.startswith("train"))
    tspan = [float(v) for v in g[tag + "/tspan"]]
    nt = int(g[tag + "/nt"])
    n = g[tag + "/zFull"].shape[0]
    with torch.no_grad():
        zF, cF = orc.rollout(g.t("x")[:n], P, S, tspan, nt, "rk4", g.meta["alph"], intermediates=True)
    d = g.meta["d"]
    close(zF[:, :d], g[tag + "/zFull"][:, :d], rtol=1e-5, atol=1e-5)   # states
    close(zF[:, d:], g[tag + "/zFull"][:, d:], rtol=1e-4, atol=1e-4)   # running costs
    close(cF, g[tag + "/ctrlFull"], rtol=1e-4, atol=1e-4)
    assert float(cF[:, :, 0].abs().max()) == 0.0          # ctrlFull[...,0] is never written upstream


def test_phi_and_physics_units(golden):
    g = golden
    P = orc.PhiParams.from_state_dict(g.state_dict())
    s = g.t("unit/s")
    with torch.no_grad():
        close(orc.phi_value(P, s), g["unit/phi"])
        close(orc.phi_grad(P, s), g["unit/gradphi"])
        x, p = s[:, :-1].contiguous(), g.t("unit/p")
        for mode in ("eval", "train"):
            S = spec_of(g, training=(mode == "train"))
            L, H, Q, W = orc.prob_LHQW(S, x, p)
            got = torch.cat([torch.as_tensor(t).float().reshape(-1, 1) for t in (L, H, Q, W)], 1)
            close(got, g[f"unit/{mode}/LHQW"])
            close(orc.prob_gradpH(S, x, p), g[f"unit/{mode}/gradpH"])
            close(orc.prob_ctrls(S, x, p), g[f"unit/{mode}/ctrls"])


def test_known_answers_on_xinit(golden_pretrained):
    """Jc and the 7 costs of the pretrained models on the RNG-free xInit."""
    g = golden_pretrained
    P = orc.PhiParams.from_state_dict(g.state_dict())
    S = spec_of(g, training=False)
    with torch.no_grad():
        Jc, cs = orc.rollout(g.t("xInit"), P, S, [0.0, 1.0], int(g["xinit_eval/nt"]), "rk4", g.meta["alph"])
    assert abs(Jc.item() - float(g["xinit_eval/Jc"])) <= 1e-4 * abs(float(g["xinit_eval/Jc"]))
    survey = {"softcorridor": 6.4451027e+01, "swap2": 7.5607623e+02, "swap12": 5.4430332e+03,
              "swarm50": 1.5968813e+03, "singlequad": 2.2499763e+03}[g.name]
    assert abs(Jc.item() - survey) <= 1e-4 * survey


def test_fp64_yardstick(golden_pretrained):
    """the fp32 oracle sits at the fp32 noise floor of the reference's own fp64 run"""
    g = golden_pretrained
    want = float(g["eval_rk4_f64/Jc"])
    got = float(g["eval_rk4/Jc"])
    assert abs(got - want) <= 2e-5 * abs(want)


def test_gradient_identity(golden):
    """Phi.getGrad == autograd of Phi, in fp64 for a sharp check."""
    g = golden
    P = orc.PhiParams.from_state_dict(g.state_dict()).to(torch.float64)
    s = g.t("unit/s").double().requires_grad_(True)
    v = orc.phi_value(P, s).sum()
    (auto,) = torch.autograd.grad(v, s)
    with torch.no_grad():
        ana = orc.phi_grad(P, s)
    assert (auto - ana).abs().max().item() <= 1e-9 * max(1.0, ana.abs().max().item())
