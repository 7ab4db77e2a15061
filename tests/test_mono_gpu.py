"""GPU parity tests of the one-CU weight-stationary kernel (neuraloc_amd/csrc/nocf_mono.inc): two-layer networks of up to 128
hidden units with d+1 <= 16 (singlequad).  Against the reference's stored outputs, the oracle and the per-tile kernel
(NOCF_MONO=0), with and without intermediates.  Tolerances as in test_hip_parity.py."""
import pytest
import torch

import neuraloc_amd as na
from oracle import ocflow_oracle as orc
from conftest import load_golden
from util_hip import closed_form_normal, count_off, make_net, make_oracle, make_prob

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _table(x, net, prob, tspan, nt, stepper, alph):
    with torch.no_grad():
        _, csn = na.OCflow(x, net, prob, tspan, nt, stepper, alph, noMean=True)
    return torch.cat(csn, 1).cpu()


def _off(tab, want):
    """per-sample costs rel 1e-3 + abs 1e-3, plus 1e-6 of the row's largest entry: cHJfin = |Phi(z(T),T) - G(z(T))| is the
    difference of two numbers ~7e3 here, so two correct fp32 evaluations differ in it by a few ulp(7e3) ~ 1e-3..5e-3
    (the oracle itself moves by that much when x is scaled by 1 + 1e-6; tools/mono_diff.py prints the rows)"""
    t, w = tab.double(), want.double()
    return (t - w).abs() > 1e-3 + 1e-3 * w.abs() + 1e-6 * w.abs().max(dim=1, keepdim=True).values


def _flips(tab, want):
    return int(_off(tab, want).any(dim=1).sum())


def test_mono_kernel_is_the_default_for_singlequad(monkeypatch, capfd):
    g = load_golden("singlequad")
    net, prob = make_net(g, DEV), make_prob(g, DEV, training=False)
    monkeypatch.setenv("NOCF_DEBUG", "1")
    _table(g.t("x").to(DEV), net, prob, [0.0, 1.0], 4, "rk4", g.meta["alph"])
    torch.cuda.synchronize()
    assert "mono kernel" in capfd.readouterr().err


@pytest.mark.parametrize("n", [1, 5, 16, 17, 100, 1000, 4096])
@pytest.mark.parametrize("stepper,tspan", [("rk4", [0.0, 1.0]), ("rk1", [0.25, 0.9])])
def test_mono_matches_tile_kernel_and_oracle_on_singlequad(n, stepper, tspan, monkeypatch):
    g = load_golden("singlequad")
    net, prob = make_net(g, DEV), make_prob(g, DEV, training=False)
    m = g.meta
    xi = closed_form_normal(n, m["d"], 5)
    xi[:, 3:] = 0.0
    x = (g.t("xInit") + m["var0"] * xi).contiguous()
    nt = 12
    monkeypatch.setenv("NOCF_MONO", "1")
    mono = _table(x.to(DEV), net, prob, tspan, nt, stepper, m["alph"])
    assert torch.equal(mono, _table(x.to(DEV), net, prob, tspan, nt, stepper, m["alph"])), "not run-to-run deterministic"
    monkeypatch.setenv("NOCF_MONO", "0")
    tile = _table(x.to(DEV), net, prob, tspan, nt, stepper, m["alph"])
    assert _flips(mono, tile) <= n // 1024, f"mono vs tile kernel: {_flips(mono, tile)} samples differ"
    keep = ~_off(mono, tile).any(dim=1)
    for j in range(7):
        a, b = mono[keep, j].double().mean().item(), tile[keep, j].double().mean().item()
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, f"column {j}: mean {a} vs {b}"
    if n <= 100:
        P, S = make_oracle(g, False)
        want = orc.persample_table(x, P, S, tspan, nt, stepper, m["alph"])
        assert _flips(mono, want) == 0, f"mono vs oracle: {_flips(mono, want)} samples off"


def test_mono_intermediates_against_oracle_and_tile_kernel(monkeypatch):
    """zFull / ctrlFull incl. the control evaluation at (z_{k+1}, t_k) and ctrlFull[...,0] == 0 (src/OCflow.py:41-55)"""
    g = load_golden("singlequad")
    net, prob = make_net(g, DEV), make_prob(g, DEV, training=False)
    m = g.meta
    x = g.t("x")[:21].contiguous()
    nt, d = 9, m["d"]
    monkeypatch.setenv("NOCF_MONO", "1")
    with torch.no_grad():
        zF, cF = na.OCflow(x.to(DEV), net, prob, [0.0, 1.0], nt, "rk4", m["alph"], intermediates=True)
    monkeypatch.setenv("NOCF_MONO", "0")
    with torch.no_grad():
        zT, cT = na.OCflow(x.to(DEV), net, prob, [0.0, 1.0], nt, "rk4", m["alph"], intermediates=True)
        P, S = make_oracle(g, False)
        zW, cW = orc.rollout(x, P, S, [0.0, 1.0], nt, "rk4", m["alph"], intermediates=True)
    assert zF.shape == (21, d + 4, nt + 1) and cF.shape == (21, 4, nt + 1)
    assert float(cF[:, :, 0].abs().max()) == 0.0
    bad, worst = count_off(zF.cpu()[:, :d], zW[:, :d], 1e-5, 1e-4)
    assert bad == 0, f"{bad} state entries off (worst {worst:g})"
    bad, worst = count_off(cF.cpu(), cW, 1e-4, 1e-4 * float(cW.abs().max()) + 1e-5)
    assert bad == 0, f"{bad} control entries off (worst {worst:g})"
    bad, worst = count_off(zF.cpu(), zT.cpu(), 1e-4, 1e-3)
    assert bad == 0, f"mono vs tile trajectories: {bad} entries off (worst {worst:g})"


@pytest.mark.parametrize("m_", [24, 64, 100, 128])
@pytest.mark.parametrize("training", [False, True])
def test_mono_other_widths_against_oracle(m_, training):
    """quadcopter networks of other widths (hidden units zero-padded to 32 / 64 / 128 by the images)"""
    from util_hip import synth_state_dict as _synth_state_dict
    alph = [5000.0, 0.0, 0.0, 0.1, 0.05, 0.02]
    torch.manual_seed(2)
    prob, x0, _, _ = na.initProb("singlequad", 37, 8, 0.3, alph, lambda t: t.float().to(DEV))
    prob.train() if training else prob.eval()
    d = x0.shape[1]
    sd = _synth_state_dict(2, m_, d, seed=m_ % 5)
    net = na.Phi(nTh=2, m=m_, d=d, alph=alph)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    P = orc.PhiParams.from_state_dict(sd)
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu()
    got = _table(x0, net, prob, [0.0, 1.0], 6, "rk4", alph)
    want = orc.persample_table(x0.cpu(), P, S, [0.0, 1.0], 6, "rk4", alph)
    assert _flips(got, want) == 0, f"m={m_}: {_flips(got, want)} samples off"


@pytest.mark.parametrize("n,nt,stepper", [(37, 5, "rk4"), (4096, 3, "rk4"), (100, 6, "rk1"), (1, 4, "rk4"), (16, 2, "rk1"), (20000, 2, "rk4")])
def test_mono_adjoint_against_both_per_tile_adjoints(n, nt, stepper, monkeypatch):
    """training of singlequad: the one-CU kernel is the recording forward (stage inputs) and the one-CU adjoint (nocf_mono_bwd.inc: grad Phi
    from the activation record -- or re-run from registers, NOCF_ACT_REC=0 --, weight gradients accumulated in the kernel, one partial
    vector per workgroup) is the backward.  Against the
    per-tile adjoint with the activation record (NOCF_MONO_BWD=0), recomputing (NOCF_ACT_REC=0 too) and behind the tile kernel's recording
    forward (NOCF_MONO_REC=0): same Jc from the same forward, gradients and dJc/dx0 equal up to the rounding of the forward sweeps."""
    from neuraloc_amd import _lib
    g = load_golden("singlequad")
    m = g.meta
    x = (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], 7)).contiguous().to(DEV)
    out = {}
    knobs = ("NOCF_ACT_REC", "NOCF_MONO_REC", "NOCF_MONO_BWD")
    for tag, env in (("mid", {}), ("midnorec", {"NOCF_ACT_REC": "0"}), ("rec", {"NOCF_MONO_BWD": "0"}), ("norec", {"NOCF_MONO_BWD": "0", "NOCF_ACT_REC": "0"}),
                     ("tile", {"NOCF_MONO_BWD": "0", "NOCF_ACT_REC": "0", "NOCF_MONO_REC": "0"})):
        for k_ in knobs:
            monkeypatch.delenv(k_, raising=False)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        net = make_net(g, DEV).train()
        prob = make_prob(g, DEV, training=True)
        xx = x.clone().requires_grad_(True)
        Jc, _ = na.OCflow(xx, net, prob, [0.0, 1.0], nt, stepper, m["alph"])
        kern = _lib.lib().nocf_last_rollout_kernel().decode()
        assert kern == ("rollout_mono_kernel" if tag != "tile" else "rollout_kernel<shape-specialised>"), kern
        Jc.backward()
        torch.cuda.synchronize()
        kern = _lib.lib().nocf_last_rollout_kernel().decode()
        assert kern == ("rollout_mono_bwd_kernel" if tag.startswith("mid") else "rollout_bwd_kernel"), kern
        out[tag] = (float(Jc.detach()), [p.grad.detach().clone() for p in net.parameters()], xx.grad.detach().clone())
    assert out["mid"][0] == out["midnorec"][0] == out["rec"][0] == out["norec"][0]
    assert abs(out["mid"][0] - out["tile"][0]) <= 2e-5 * abs(out["tile"][0])
    for other, tol in (("midnorec", 2e-4), ("rec", 5e-4), ("norec", 5e-4), ("tile", 2e-3)):
        for ga, gb in zip(out["mid"][1], out[other][1]):
            scale = float(gb.abs().max())
            assert torch.isfinite(ga).all() and float((ga - gb).abs().max()) <= tol * scale + 1e-12, (other, float((ga - gb).abs().max()), scale)
        assert float((out["mid"][2] - out[other][2]).abs().max()) <= tol * float(out[other][2].abs().max()) + 1e-12


@pytest.mark.parametrize("name,n,stepper,training,m_", [
    ("singlequad", 11, "rk4", True, 128), ("singlequad", 21, "rk1", False, 64), ("singlequad", 33, "rk4", True, 120), ("singlequad", 17, "rk4", True, 96),
    ("midcross4", 25, "rk4", True, 80), ("singlequad", 9, "rk4", False, 104),
    ("swap12", 21, "rk4", True, 64), ("swap12", 9, "rk1", False, 128), ("swap12_5pair", 33, "rk4", True, 96), ("swap12_4pair", 5, "rk4", False, 48),
    ("midcross4", 13, "rk4", True, 128), ("midcross4", 16, "rk1", False, 64), ("softcorridor", 7, "rk4", True, 64),
    ("midcross4", 40, "rk4", False, 128), ("swap2", 1, "rk4", True, 48), ("midcross2", 9, "rk4", False, 128)])
def test_mono_adjoint_against_oracle_fp64_autograd(name, n, stepper, training, m_):
    """the one-CU adjoint on medium networks of every problem class with d + 1 <= 32 (quadcopter physics; point agents with obstacle and
    interaction terms, both mask modes; one and two input k-blocks), ragged batches, hidden widths that are padded: dJc/dtheta against the
    oracle differentiated by torch autograd in fp64"""
    from neuraloc_amd import _lib
    from util_hip import synth_state_dict as _synth_state_dict
    from test_hip_parity import _oracle_grads64
    alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
    torch.manual_seed(11)
    prob, x0, _, _ = na.initProb(name, 40, 40, 0.5, alph, lambda t: t.float().to(DEV))
    prob.train() if training else prob.eval()
    x0 = x0[:n].contiguous()
    d = x0.shape[1]
    sd = _synth_state_dict(2, m_, d, seed=len(name))
    net = na.Phi(nTh=2, m=m_, d=d, alph=alph)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    nt = 6
    Jc, _ = na.OCflow(x0, net, prob, [0.0, 1.0], nt, stepper, alph)
    Jc.backward()
    torch.cuda.synchronize()
    assert _lib.lib().nocf_last_rollout_kernel().decode() == "rollout_mono_bwd_kernel"
    J64, want = _oracle_grads64(x0, sd, prob, nt, stepper, alph, 2)
    assert abs(Jc.item() - J64) <= 2e-5 * abs(J64)
    for k, p in net.named_parameters():
        w = want[k] if want[k] is not None else torch.zeros_like(p, dtype=torch.float64).cpu()
        scale = w.abs().max().item()
        err = (p.grad.cpu().double() - w).abs().max().item()
        assert err <= 2e-4 * scale + 1e-6, f"{name} {k}: err {err:g} at scale {scale:g}"


@pytest.mark.parametrize("name,m_,training", [("swap12", 64, True), ("swap12", 128, False), ("swap12", 96, True), ("swap12_5pair", 48, True), ("swap12_3pair", 72, False)])
def test_mono_wider_inputs_against_oracle(name, m_, training, capfd, monkeypatch):
    """medium networks on problems with 17 <= d + 1 <= 32 (two input k-blocks: 12 agents) run the one-CU kernel too; against the oracle"""
    from util_hip import synth_state_dict as _synth_state_dict
    monkeypatch.setenv("NOCF_DEBUG", "1")
    alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
    torch.manual_seed(3)
    prob, x0, _, _ = na.initProb(name, 37, 8, 0.3, alph, lambda t: t.float().to(DEV))
    prob.train() if training else prob.eval()
    d = x0.shape[1]
    sd = _synth_state_dict(2, m_, d, seed=m_ % 7)
    net = na.Phi(nTh=2, m=m_, d=d, alph=alph)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    P = orc.PhiParams.from_state_dict(sd)
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu()
    got = _table(x0, net, prob, [0.0, 1.0], 6, "rk4", alph)
    assert "mono kernel" in capfd.readouterr().err
    want = orc.persample_table(x0.cpu(), P, S, [0.0, 1.0], 6, "rk4", alph)
    assert _flips(got, want) == 0, f"{name} m={m_}: {_flips(got, want)} samples off"
