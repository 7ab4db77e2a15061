"""GPU parity tests of the double-precision rollout (nocf_rollout_f64, neuraloc_amd/csrc/nocf_f64.inc): the reference's
`--prec double` path (trainOC.py:76-79, evalOC.py:19,28-31).  Against the reference's own stored double-precision outputs
(`eval_rk4_f64/*` of every workload fixture, written by tests/golden/make_golden.py from the imported reference) and against the
oracle run in float64 on every problem initProb knows, deeper networks, rk1, a time segment, ragged batches, training-mode masks
and with intermediates.

Tolerance: rel 1e-9 (+ abs 1e-9 of the row's largest entry) -- both sides compute in double and differ by summation order only."""
import pytest
import torch

import neuraloc_amd as na
from oracle import ocflow_oracle as orc
from conftest import load_golden
from util_hip import make_net, make_prob

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
F64 = torch.float64
ALPH = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]


def _rows_off(tab, want, rtol=1e-9):
    t, w = tab.to(F64).cpu(), torch.as_tensor(want).to(F64).cpu()
    tol = rtol * w.abs() + rtol * w.abs().max(dim=-1, keepdim=True).values + 1e-12
    return int(((t - w).abs() > tol).any(dim=-1).sum())


@pytest.mark.parametrize("name", ["swap2", "swap12", "softcorridor", "singlequad", "swarm50"])
def test_f64_rollout_against_the_references_double_run(name):
    g = load_golden(name)
    if not g.has("eval_rk4_f64/Jc"):
        pytest.skip("no double-precision entry in this fixture")
    m = g.meta
    net = make_net(g, DEV).to(F64)
    prob = make_prob(g, DEV, training=False)
    prob.xtarget = prob.xtarget.to(F64)
    x = g.t("x").to(F64).to(DEV)
    with torch.no_grad():
        Jc, cs = na.OCflow(x, net, prob, [0.0, 1.0], m["nt"], "rk4", m["alph"])
        _, csn = na.OCflow(x, net, prob, [0.0, 1.0], m["nt"], "rk4", m["alph"], noMean=True)
    assert Jc.dtype == F64 and csn[0].dtype == F64
    assert _rows_off(torch.cat(csn, 1), g["eval_rk4_f64/persample"]) == 0
    want_cs = torch.as_tensor(g["eval_rk4_f64/cs"]).to(F64)
    got_cs = torch.stack([c for c in cs]).cpu()
    assert bool(((got_cs - want_cs).abs() <= 1e-9 * want_cs.abs() + 1e-12).all()), (got_cs, want_cs)
    want_J = float(g["eval_rk4_f64/Jc"])
    assert abs(float(Jc) - want_J) <= 1e-9 * abs(want_J)


def _synth_state_dict64(nTh, m, d, seed):
    net = na.Phi(nTh=nTh, m=m, d=d)
    sd = net.state_dict()
    for j, (k, v) in enumerate(sd.items()):
        i = torch.arange(v.numel(), dtype=F64)
        fan = v.shape[-1] if v.dim() > 1 else 4
        sd[k] = (0.8 / fan ** 0.5 * torch.sin(0.37 * i + 0.11 * (i % 7) + 0.3 * j + seed)).reshape(v.shape)
    sd["w.weight"] = sd["w.weight"] + 1.0
    return sd


@pytest.mark.parametrize("name", sorted(na.initProb.__globals__["PROBLEM_NAMES"]))
@pytest.mark.parametrize("training", [False, True])
def test_f64_every_initprob_problem_against_the_oracle_in_double(name, training):
    torch.manual_seed(7)
    prob, x0, _, _ = na.initProb(name, 13, 13, 0.5, ALPH, lambda t: t.to(F64).to(DEV))
    prob.train() if training else prob.eval()
    d = x0.shape[1]
    nTh, m = (3, 40) if d > 30 else (2, 24)
    sd = _synth_state_dict64(nTh, m, d, seed=len(name))
    net = na.Phi(nTh=nTh, m=m, d=d, alph=ALPH).to(F64)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    P = orc.PhiParams.from_state_dict(sd, dtype=F64)
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu().to(F64)
    nt = 6
    with torch.no_grad():
        _, csn = na.OCflow(x0, net, prob, [0.0, 1.0], nt, "rk4", ALPH, noMean=True)
        zF, cF = na.OCflow(x0[:5], net, prob, [0.0, 1.0], nt, "rk4", ALPH, intermediates=True)
        want = orc.persample_table(x0.cpu(), P, S, [0.0, 1.0], nt, "rk4", ALPH)
        zW, cW = orc.rollout(x0[:5].cpu(), P, S, [0.0, 1.0], nt, "rk4", ALPH, intermediates=True)
    assert _rows_off(torch.cat(csn, 1), want) == 0, name
    assert zF.shape == zW.shape and cF.shape == cW.shape and zF.dtype == F64
    assert float((zF.cpu() - zW).abs().max()) <= 1e-9 * max(1.0, float(zW.abs().max()))
    assert float((cF.cpu() - cW).abs().max()) <= 1e-9 * max(1.0, float(cW.abs().max()))
    assert float(cF[:, :, 0].abs().max()) == 0.0                       # src/OCflow.py:41-43: slot 0 stays zero


@pytest.mark.parametrize("nTh,m,n,stepper,tspan", [(4, 64, 1, "rk4", [0.0, 1.0]), (3, 100, 5, "rk1", [0.0, 1.0]),
                                                    (2, 512, 7, "rk4", [0.25, 0.9]), (2, 130, 1030, "rk4", [0.0, 1.0]),
                                                    # wide layers (m > 256: the register-tiled products) with ragged row blocks and k tails
                                                    (2, 300, 9, "rk4", [0.0, 1.0]), (3, 260, 1025, "rk4", [0.0, 1.0]), (2, 257, 3, "rk1", [0.0, 0.5])])
def test_f64_depths_widths_steppers_and_ragged_batches(nTh, m, n, stepper, tspan):
    torch.manual_seed(3)
    prob, x0, _, _ = na.initProb("midcross4", n, 4, 0.5, ALPH, lambda t: t.to(F64).to(DEV))
    prob.eval()
    d = x0.shape[1]
    sd = _synth_state_dict64(nTh, m, d, seed=nTh)
    net = na.Phi(nTh=nTh, m=m, d=d, alph=ALPH).to(F64)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    P = orc.PhiParams.from_state_dict(sd, dtype=F64)
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu().to(F64)
    nt = 4
    with torch.no_grad():
        Jc, cs = na.OCflow(x0, net, prob, tspan, nt, stepper, ALPH)
        _, csn = na.OCflow(x0, net, prob, tspan, nt, stepper, ALPH, noMean=True)
        want = orc.persample_table(x0.cpu(), P, S, tspan, nt, stepper, ALPH)
    tab = torch.cat(csn, 1)
    assert _rows_off(tab, want) == 0
    for j in range(7):                                               # the kernel's fixed-order sums against the per-sample table
        assert abs(float(cs[j]) - float(tab[:, j].mean())) <= 1e-12 * max(1.0, abs(float(cs[j])))


def test_f64_refuses_mixed_precision():
    g = load_golden("swap2")
    net32 = make_net(g, DEV)
    prob = make_prob(g, DEV, training=False)
    x64 = g.t("x").to(F64).to(DEV)
    with torch.no_grad(), pytest.raises(RuntimeError, match="double-precision call"):
        na.OCflow(x64, net32, prob, [0.0, 1.0], 4, "rk4", g.meta["alph"])            # fp32 network, fp64 states


@pytest.mark.parametrize("nTh,m,n", [(2, 24, 1), (3, 40, 7), (4, 64, 33), (2, 512, 1030), (2, 300, 17), (3, 263, 5)])
def test_f64_phi_value_and_gradient_against_the_oracle_in_double(nTh, m, n):
    """Phi.forward / Phi.getGrad in double (src/Phi.py:91-138) on random points s = [x, t]"""
    d = 8
    sd = _synth_state_dict64(nTh, m, d, seed=nTh + m)
    net = na.Phi(nTh=nTh, m=m, d=d, alph=ALPH).to(F64)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    P = orc.PhiParams.from_state_dict(sd, dtype=F64)
    g = torch.Generator().manual_seed(5)
    s = torch.randn(n, d + 1, generator=g, dtype=F64)
    with torch.no_grad():
        val, grad = net(s.to(DEV)), net.getGrad(s.to(DEV))
        wv, wg = orc.phi_value(P, s), orc.phi_grad(P, s)
    assert val.shape == (n, 1) and grad.shape == (n, d + 1) and val.dtype == F64
    assert float((val.cpu() - wv).abs().max()) <= 1e-11 * max(1.0, float(wv.abs().max()))
    assert float((grad.cpu() - wg).abs().max()) <= 1e-11 * max(1.0, float(wg.abs().max()))


@pytest.mark.parametrize("name", ["softcorridor", "hardcorridor", "swarm", "midcross4", "singlequad"])
@pytest.mark.parametrize("training", [False, True])
def test_f64_problem_calls_against_the_oracle_in_double(name, training):
    """prob.calcLHQW / calcGradpH / calcCtrls on float64 tensors (the reference's problem objects after --prec double)"""
    if name not in na.initProb.__globals__["PROBLEM_NAMES"]:
        pytest.skip("not an initProb problem")
    torch.manual_seed(2)
    prob, x0, _, _ = na.initProb(name, 19, 4, 0.5, ALPH, lambda t: t.to(F64).to(DEV))
    prob.train() if training else prob.eval()
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu().to(F64)
    g = torch.Generator().manual_seed(9)
    p = 0.7 * torch.randn(x0.shape, generator=g, dtype=F64)
    L, H, Q, W = prob.calcLHQW(x0, p.to(DEV))
    gp, ct = prob.calcGradpH(x0, p.to(DEV)), prob.calcCtrls(x0, p.to(DEV))
    wL, wH, wQ, wW = orc.prob_LHQW(S, x0.cpu(), p)
    for got, want in ((L, wL), (H, wH), (Q, wQ), (W, wW), (gp, orc.prob_gradpH(S, x0.cpu(), p)), (ct, orc.prob_ctrls(S, x0.cpu(), p))):
        want = torch.as_tensor(want, dtype=F64).reshape(got.shape)
        assert got.dtype == F64
        assert float((got.cpu() - want).abs().max()) <= 1e-11 * max(1.0, float(want.abs().max()))
