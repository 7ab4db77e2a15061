"""GPU parity tests: the HIP path, called through the package (ctypes -> C ABI), against
 (a) the reference's own outputs stored in tests/golden/*.npz and
 (b) the oracle on the same inputs.

Stated fp32 tolerances (the reference's own fp32-vs-fp64 noise floor is
1e-6 relative on states and up to 1.6e-3 on near-zero per-sample costs):
   states / trajectories : |diff| <= 1e-4 + 1e-5 |ref|
   batch means (Jc, cs)  : rel 1e-4 (abs 1e-6 for tiny terms)
   per-sample costs      : rel 1e-3 + abs 1e-3, isolated mask-flip outliers counted and bounded
"""
import os

import numpy as np
import pytest
import torch

import neuraloc_amd as na
from neuraloc_amd import _lib
from oracle import ocflow_oracle as orc
from util_hip import count_off, full_states, make_net, make_oracle, make_prob, poison_allocator

pytestmark = pytest.mark.gpu
def load_golden_by_name(name):
    from conftest import load_golden
    return load_golden(name)


DEV = torch.device("cuda:0")


def mean_close(got, want, name=""):
    got, want = float(got), float(want)
    assert abs(got - want) <= 1e-4 * abs(want) + 1e-6, f"{name}: hip {got:.8e} vs ref {want:.8e}"


def test_mfma_tile_layout():
    """the 4x4x1 MFMA tile code computes out[4,64] = a[4,K] @ b[K,64] (asymmetric data)"""
    K = 37
    a = torch.arange(4 * K, dtype=torch.float32).reshape(4, K) * 0.25 - 3.0
    b = (torch.arange(K * 64, dtype=torch.float32).reshape(K, 64) % 13) - 6.0 + 0.5 * (torch.arange(64) % 3)
    ad, bd = a.to(DEV), b.to(DEV)
    out = torch.empty(4, 64, device=DEV)
    rc = _lib.lib().nocf_selftest_mfma(_lib.ptr(ad), _lib.ptr(bd), K, _lib.ptr(out), _lib.stream_ptr(DEV))
    _lib.check(rc, "selftest")
    torch.cuda.synchronize()
    want = a.double() @ b.double()
    assert (out.cpu().double() - want).abs().max().item() <= 1e-3 * want.abs().max().item() * 1e-3 + 1e-2


def test_phi_value_and_gradient(golden):
    g = golden
    net = make_net(g, DEV)
    s = g.t("unit/s")
    with torch.no_grad():
        val = net(s.to(DEV)).cpu()
        grad = net.getGrad(s.to(DEV)).cpu()
    assert val.shape == (s.shape[0], 1) and grad.shape == s.shape
    for got, key in ((val, "unit/phi"), (grad, "unit/gradphi")):
        want = torch.from_numpy(g[key])
        scale = want.abs().max().item()
        err = (got - want).abs().max().item()
        assert err <= 2e-5 * scale + 1e-5, f"{g.name} {key}: err {err:g} at scale {scale:g}"


def test_phi_gradient_ragged_batch(golden_pretrained):
    """batch sizes that are not a multiple of the 4-sample tile, including 1"""
    g = golden_pretrained
    net = make_net(g, DEV)
    s = g.t("unit/s")
    want = torch.from_numpy(g["unit/gradphi"])
    with torch.no_grad():
        for n in (1, 3, 5, 39):
            got = net.getGrad(s[:n].to(DEV)).cpu()
            assert (got - want[:n]).abs().max().item() <= 2e-5 * want.abs().max().item() + 1e-5


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_problem_physics(golden, mode):
    g = golden
    prob = make_prob(g, DEV, training=(mode == "train"))
    x = g.t("unit/s")[:, :-1].contiguous().to(DEV)
    p = g.t("unit/p").to(DEV)
    L, H, Q, W = prob.calcLHQW(x, p)
    got = torch.cat((L, H, Q, W), 1).cpu()
    want = torch.from_numpy(g[f"unit/{mode}/LHQW"])
    for j, nm in enumerate("LHQW"):
        bad, worst = count_off(got[:, j], want[:, j], 1e-5, 1e-5 + 2e-4 * (nm in "LHW") * float(g.meta["alph_W"] > 0) *
                               float(g.meta["n_agents"] > 2) * max(1.0, g.meta["alph_W"]))
        assert bad == 0, f"{g.name} {mode} {nm}: {bad} off, worst {worst:g}"
    gp = prob.calcGradpH(x, p).cpu()
    ct = prob.calcCtrls(x, p).cpu()
    for got2, key in ((gp, "gradpH"), (ct, "ctrls")):
        want2 = torch.from_numpy(g[f"unit/{mode}/{key}"])
        assert got2.shape == want2.shape
        assert (got2 - want2).abs().max().item() <= 1e-5 * want2.abs().max().item() + 1e-5


@pytest.mark.parametrize("tag", ["eval_rk4", "eval_rk1", "eval_seg", "train_rk4"])
def test_rollout_against_reference_golden(golden, tag):
    g = golden
    if not g.has(tag + "/Jc"):
        pytest.skip("case not in this fixture")
    net = make_net(g, DEV)
    prob = make_prob(g, DEV, training=tag.startswith("train"))
    stepper = "rk1" if tag.endswith("rk1") else "rk4"
    tspan = [float(v) for v in g[tag + "/tspan"]]
    nt = int(g[tag + "/nt"])
    alph = g.meta["alph"]
    x = g.t("x").to(DEV)
    with torch.no_grad():
        Jc, cs = na.OCflow(x, net, prob, tspan, nt, stepper, alph)
        Jn, csn = na.OCflow(x, net, prob, tspan, nt, stepper, alph, noMean=True)
    tab = torch.cat(csn, 1).cpu()
    assert tab.shape == (x.shape[0], 7) and Jn.shape == (x.shape[0], 1)
    want = torch.from_numpy(g[tag + "/persample"])
    # Per-sample costs: rel 1e-3 + abs 1e-3.  Q/W/L are discontinuous in x (masks), so a sample that
    # sits within fp32 noise of a threshold may flip -- the reference's own fp32 run flips one swarm50
    # train-mode sample against its fp64 run (rel 7e-3 on that sample, 1.9e-4 on the batch mean).
    # Such rows are counted, bounded and reported, and the batch means are compared without them.
    off = (tab.double() - want.double()).abs() > 1e-3 + 1e-3 * want.double().abs()
    flipped = off.any(dim=1)
    nflip = int(flipped.sum())
    assert nflip <= 2, f"{g.name} {tag}: {nflip} samples beyond rel 1e-3 + abs 1e-3: rows {flipped.nonzero().flatten().tolist()}"
    names = ["L", "G", "HJt", "HJfin", "HJgrad", "Q", "W"]
    if nflip == 0:
        mean_close(Jc, g[tag + "/Jc"], f"{g.name} {tag} Jc")
        for j, nm in enumerate(names):
            mean_close(cs[j], g[tag + "/cs"][j], f"{g.name} {tag} {nm}")
    else:
        print(f"[mask-flip] {g.name} {tag}: rows {flipped.nonzero().flatten().tolist()} excluded from the mean check")
        keep = ~flipped
        for j, nm in enumerate(names):
            mean_close(tab[keep, j].double().mean(), want[keep, j].double().mean(), f"{g.name} {tag} {nm} (unflipped rows)")
    # the reported means / Jc are exactly the means of the per-sample table (deterministic fp64 reduction)
    for j, nm in enumerate(names):
        assert abs(float(cs[j]) - tab[:, j].double().mean().item()) <= 2e-6 * abs(float(cs[j])) + 1e-9
    Jtab = (tab[:, 0].double().mean() + alph[0] * tab[:, 1].double().mean() + alph[3] * tab[:, 2].double().mean()
            + alph[4] * tab[:, 3].double().mean() + alph[5] * tab[:, 4].double().mean()).item()
    assert abs(float(Jc) - Jtab) <= 2e-6 * abs(Jtab)
    assert (Jn.cpu() - (tab[:, 0:1] + alph[0] * tab[:, 1:2] + alph[3] * tab[:, 2:3] + alph[4] * tab[:, 3:4]
                        + alph[5] * tab[:, 4:5])).abs().max().item() <= 1e-5 * Jn.abs().max().item()


@pytest.mark.parametrize("tag", ["eval_rk4", "eval_seg", "train_rk4"])
def test_intermediates_against_reference_golden(golden, tag):
    g = golden
    if not g.has(tag + "/zFull"):
        pytest.skip("case not in this fixture")
    net = make_net(g, DEV)
    prob = make_prob(g, DEV, training=tag.startswith("train"))
    tspan = [float(v) for v in g[tag + "/tspan"]]
    nt = int(g[tag + "/nt"])
    zw = torch.from_numpy(g[tag + "/zFull"])
    cw = torch.from_numpy(g[tag + "/ctrlFull"])
    n, d = zw.shape[0], g.meta["d"]
    with torch.no_grad():
        zF, cF = na.OCflow(g.t("x")[:n].to(DEV), net, prob, tspan, nt, "rk4", g.meta["alph"], intermediates=True)
    zF, cF = zF.cpu(), cF.cpu()
    assert zF.shape == zw.shape and cF.shape == cw.shape
    bad, worst = count_off(zF[:, :d], zw[:, :d], 1e-5, 1e-4)
    assert bad == 0, f"{g.name} {tag}: {bad} state entries off (worst {worst:g})"
    bad, worst = count_off(zF[:, d:], zw[:, d:], 1e-3, 1e-3)
    assert bad <= 4, f"{g.name} {tag}: {bad} running-cost entries off (worst {worst:g})"
    # controls are -grad Phi (or thrust/torques): sums of m terms far larger than the result, so the
    # fp32 noise scales with max|ctrl| (the reference's own fp32-vs-fp64 gap reaches 7e-4 at max|ctrl|=65
    # on swarm50 eval_seg): abs 1e-4*max|ref| + rel 1e-4
    bad, worst = count_off(cF, cw, 1e-4, 1e-4 * float(cw.abs().max()) + 1e-5)
    assert bad == 0, f"{g.name} {tag}: {bad} control entries off (worst {worst:g})"
    assert float(cF[:, :, 0].abs().max()) == 0.0


def test_full_size_against_reference_and_oracle(golden_pretrained):
    """BASELINE.json sizes: Jc/cs against the reference's stored result and the oracle's per-sample table"""
    g = golden_pretrained
    m = g.meta
    net = make_net(g, DEV)
    prob = make_prob(g, DEV, training=False)
    x = full_states(g, int(g["full/seed"]))
    with torch.no_grad():
        Jc, cs = na.OCflow(x.to(DEV), net, prob, [0.0, 1.0], m["nt"], "rk4", m["alph"])
        _, csn = na.OCflow(x.to(DEV), net, prob, [0.0, 1.0], m["nt"], "rk4", m["alph"], noMean=True)
    mean_close(Jc, g["full/Jc"], f"{g.name} full Jc")
    for j in range(7):
        mean_close(cs[j], g["full/cs"][j], f"{g.name} full cs[{j}]")
    P, S = make_oracle(g, training=False)
    with torch.no_grad():
        want = orc.persample_table(x, P, S, [0.0, 1.0], m["nt"], "rk4", m["alph"])
    bad, worst = count_off(torch.cat(csn, 1), want, 1e-3, 1e-3)
    assert bad <= max(2, x.shape[0] // 500), f"{g.name}: {bad} per-sample entries off (worst {worst:g})"


def test_rk4_is_fourth_order(golden_pretrained):
    """size-independent property: halving h shrinks the final-state error ~16x (asserted: >= 8x)"""
    g = golden_pretrained
    if g.name in ("swap2",):
        pytest.skip("hard-corridor masks make the flow non-smooth")
    net = make_net(g, DEV)
    prob = make_prob(g, DEV, training=False)
    x = g.t("x")[:8].to(DEV)
    d = g.meta["d"]

    def final_state(nt):
        with torch.no_grad():
            zF, _ = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", g.meta["alph"], intermediates=True)
        return zF[:, :d, -1].double().cpu()

    # fp64 oracle errors vs nt=320 (measured): nt=16 -> 1.6e-3..0.43, nt=32 -> 8e-5..6e-3 (ratio >= 20)
    ref = final_state(128)
    e1 = (final_state(16) - ref).abs().max().item()
    e2 = (final_state(32) - ref).abs().max().item()
    assert e2 < e1 / 8.0 or e1 < 1e-3, f"{g.name}: errors {e1:g} -> {e2:g}"


def test_shard_invariance_and_determinism(golden_pretrained):
    """splitting the batch changes nothing per sample; repeated calls are bitwise identical"""
    g = golden_pretrained
    net = make_net(g, DEV)
    prob = make_prob(g, DEV, training=False)
    x = g.t("x").to(DEV)
    nt, alph = min(10, g.meta["nt"]), g.meta["alph"]
    with torch.no_grad():
        _, a = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph, noMean=True)
        _, b = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph, noMean=True)
        _, lo = na.OCflow(x[:17], net, prob, [0.0, 1.0], nt, "rk4", alph, noMean=True)
        _, hi = na.OCflow(x[17:], net, prob, [0.0, 1.0], nt, "rk4", alph, noMean=True)
    a, b = torch.cat(a, 1), torch.cat(b, 1)
    assert torch.equal(a, b)
    assert torch.equal(a, torch.cat((torch.cat(lo, 1), torch.cat(hi, 1)), 0))


def test_error_behaviour():
    g_net = na.Phi(2, 8, 4).to(DEV)
    prob = na.Cross2D(torch.zeros(4, device=DEV))
    x = torch.zeros(3, 4, device=DEV)
    with torch.no_grad():
        with pytest.raises(ValueError):
            na.OCflow(x, g_net, prob, [0.0, 1.0], 0)
        with pytest.raises(ValueError):
            na.OCflow(x, g_net, prob, [0.0, 1.0], 4, stepper="rk2")
    with pytest.raises(ValueError):
        g_net.getGrad(x)                                   # (x without the time column; under autograd or not)
    x0 = x.clone()
    with torch.no_grad():
        na.OCflow(x, g_net, prob, [0.0, 1.0], 2)
    assert torch.equal(x, x0)                              # inputs are never mutated
def test_shock_rollout_matches_two_reference_style_segments(golden_pretrained):
    """the two-segment shocked rollout (src/plotter.py:815-824) against the oracle"""
    from neuraloc_amd.shock import shock_rollout
    g = golden_pretrained
    net = make_net(g, DEV)
    prob = make_prob(g, DEV, training=False)
    P, S = make_oracle(g, training=False)
    x = g.t("x")[:6]
    d, nt, t_s = g.meta["d"], 20, 0.1
    shock = 0.05 * torch.arange(1, d + 1, dtype=torch.float32).reshape(1, -1) / d
    with torch.no_grad():
        res = shock_rollout(x.to(DEV), net, prob, nt, t_s, shock.to(DEV))
        nS = int(t_s * nt)
        z1, _ = orc.rollout(x, P, S, [0.0, t_s], nS, "rk4", g.meta["alph"], intermediates=True)
        xs = z1[:, :d, -1] + shock
        z2, _ = orc.rollout(xs, P, S, [t_s, 1.0], 1 + nt - nS, "rk4", g.meta["alph"], intermediates=True)
    want = torch.cat((z1[:, :d, :], z2[:, :d, :]), dim=2)
    assert res["nShock"] == nS and res["traj"].shape == want.shape
    bad, worst = count_off(res["traj"].cpu(), want, 1e-5, 1e-4)
    assert bad == 0, f"{g.name}: {bad} shocked-trajectory entries off (worst {worst:g})"
    with pytest.raises(ValueError):
        shock_rollout(x.to(DEV), net, prob, nt, 0.01, shock.to(DEV))      # nShock = 0: the reference divides by zero


@pytest.mark.parametrize("name,n", [("singlequad", 48), ("singlequad", 4096), ("softcorridor", 32), ("swarm50", 16)])
def test_shock_sweep_with_the_shared_prefix_matches_per_shock_time_rollouts_and_the_oracle(name, n, monkeypatch):
    """BASELINE config 5 (shock-eval sweep).  shock_sweep integrates the unshocked trajectory ONCE to the largest shock time and runs all
    second segments in one launch (nocf_rollout_segments_f32; singlequad) or one by one (networks without a segment kernel); every
    (t_s, shock) pair must equal the reference-style pair of rollouts of shock_rollout (src/plotter.py:815-824: segment 1 re-integrated
    from t0 with its own h = t_s / int(t_s nt)) to the stated STATE tolerance 1e-4 + 1e-5 |ref| -- the shared prefix steps with
    h = T / int(T nt), equal to every segment's own h to the last bit or two -- and the small case also equals the oracle.  A shock time
    off the shared grid (0.33 at nt = 50: 16 steps of 0.020625) takes the per-time path inside the same sweep."""
    from neuraloc_amd.shock import shock_rollout, shock_sweep, _on_shared_grid
    g = load_golden_by_name(name)
    net, prob = make_net(g, DEV), make_prob(g, DEV, training=False)
    d, nt = g.meta["d"], 50 if name != "swarm50" else 20
    x = (full_states(g, 7)[:n] if n > g.t("x").shape[0] else g.t("x")[:n]).to(DEV)
    times = [0.1 * k for k in range(1, 10)] + ([0.33] if n < 100 else [])
    T, N, h, on = _on_shared_grid((0.0, 1.0), nt, times)
    assert N == int(0.9 * nt) and all(k is not None for k in on[:9]) and (len(on) == 9 or on[9] is None)
    shocks = torch.zeros(2 if n < 100 else 1, d, device=DEV)
    shocks[0, 0:3] = torch.tensor([0.5, -0.5, 0.25])[: min(3, d)]
    if shocks.shape[0] > 1:
        shocks[1, :] = 0.02
    launches = []
    import importlib
    ocm = importlib.import_module("neuraloc_amd.OCflow")       # (the module: the package re-exports the function under the same name)
    real = ocm._launch_segments
    monkeypatch.setattr(ocm, "_launch_segments", lambda *a, **k: (launches.append(len(a[3])), real(*a, **k))[1])
    got = shock_sweep(x, net, prob, nt, times, shocks, alph=g.meta["alph"])
    assert len(got) == len(times) * shocks.shape[0]
    kern = _lib.lib().nocf_last_rollout_kernel().decode()
    if name == "singlequad":
        assert launches and sum(launches) == 9 * shocks.shape[0], launches              # all on-grid pairs went through the segment launches
        assert max(launches) <= 16
    P, S = make_oracle(g, training=False)
    for r in got:
        ref = shock_rollout(x, net, prob, nt, r["t_s"], shocks[r["shock_index"]:r["shock_index"] + 1], alph=g.meta["alph"])
        assert r["nShock"] == ref["nShock"] and r["traj"].shape == ref["traj"].shape == (n, d, nt + 3)
        bad, worst = count_off(r["traj"].cpu(), ref["traj"].cpu(), 1e-5, 1e-4)
        assert bad == 0, f"{name} t_s={r['t_s']}: {bad} trajectory entries off (worst {worst:g})"
        cs = ref["ctrl"].abs().max().item()
        assert (r["ctrl"] - ref["ctrl"]).abs().max().item() <= 1e-4 * cs + 1e-4, f"{name} t_s={r['t_s']}: controls"
        assert torch.allclose(r["x_shocked"], ref["x_shocked"], rtol=1e-5, atol=1e-4)
        for key in ("costs1", "costs2"):
            assert abs(float(r[key][0]) - float(ref[key][0])) <= 2e-4 * abs(float(ref[key][0])) + 1e-6, (name, r["t_s"], key)
            for j in range(7):
                a, b = float(r[key][1][j]), float(ref[key][1][j])
                assert abs(a - b) <= 2e-4 * abs(b) + 1e-5, (name, r["t_s"], key, j, a, b)
        if n <= 48 and r["shock_index"] == 0 and abs(r["t_s"] - 0.5) < 1e-9:          # ... and the oracle, on one pair
            nS = int(r["t_s"] * nt)
            with torch.no_grad():
                z1, _ = orc.rollout(x.cpu(), P, S, [0.0, r["t_s"]], nS, "rk4", g.meta["alph"], intermediates=True)
                xs = z1[:, :d, -1] + shocks[0:1].cpu()
                z2, _ = orc.rollout(xs, P, S, [r["t_s"], 1.0], 1 + nt - nS, "rk4", g.meta["alph"], intermediates=True)
            want = torch.cat((z1[:, :d, :], z2[:, :d, :]), dim=2)
            bad, worst = count_off(r["traj"].cpu(), want, 1e-5, 1e-4)
            assert bad == 0, f"{name}: {bad} entries off the oracle (worst {worst:g})"
    assert kern.startswith("rollout_")


# ---- every problem initProb knows (15 names), small deterministic nets, both modes: HIP vs the oracle
def _synth_state_dict(nTh, m, d, seed):
    import neuraloc_amd as na_
    net = na_.Phi(nTh=nTh, m=m, d=d)
    sd = net.state_dict()
    for j, (k, v) in enumerate(sd.items()):
        n = v.numel()
        i = torch.arange(n, dtype=torch.float64)
        fan = v.shape[-1] if v.dim() > 1 else 4
        sd[k] = (0.8 / fan ** 0.5 * torch.sin(0.37 * i + 0.11 * (i % 7) + 0.3 * j + seed)).reshape(v.shape).float()
    sd["w.weight"] = sd["w.weight"] + 1.0
    return sd


@pytest.mark.parametrize("name", sorted(na.initProb.__globals__["PROBLEM_NAMES"]))
@pytest.mark.parametrize("training", [False, True])
def test_every_initprob_problem_against_oracle(name, training):
    alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
    torch.manual_seed(7)
    prob, x0, _, xInit = na.initProb(name, 24, 24, 0.5, alph, lambda t: t.float().to(DEV))
    prob.train() if training else prob.eval()
    d = x0.shape[1]
    nTh, m = (3, 40) if d > 30 else (2, 24)
    sd = _synth_state_dict(nTh, m, d, seed=len(name))
    net = na.Phi(nTh=nTh, m=m, d=d, alph=alph)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    P = orc.PhiParams.from_state_dict(sd)
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu()
    nt = 6
    with torch.no_grad():
        _, csn = na.OCflow(x0, net, prob, [0.0, 1.0], nt, "rk4", alph, noMean=True)
        zF, cF = na.OCflow(x0[:5], net, prob, [0.0, 1.0], nt, "rk4", alph, intermediates=True)
        want = orc.persample_table(x0.cpu(), P, S, [0.0, 1.0], nt, "rk4", alph)
        zW, cW = orc.rollout(x0[:5].cpu(), P, S, [0.0, 1.0], nt, "rk4", alph, intermediates=True)
    tab = torch.cat(csn, 1).cpu()
    off = (tab.double() - want.double()).abs() > 1e-3 + 1e-3 * want.double().abs()
    assert int(off.any(dim=1).sum()) <= 2, f"{name}: {int(off.any(dim=1).sum())} samples off"
    bad, worst = count_off(zF.cpu()[:, :d], zW[:, :d], 1e-5, 1e-4)
    assert bad == 0, f"{name}: {bad} state entries off (worst {worst:g})"
    bad, worst = count_off(cF.cpu(), cW, 1e-4, 1e-4 * float(cW.abs().max()) + 1e-5)
    assert bad == 0, f"{name}: {bad} control entries off (worst {worst:g})"


# ---- training: Jc.backward() through the HIP rollout vs the reference's autograd gradients
@pytest.mark.parametrize("name", ["swap2", "softcorridor", "swap12", "swarm50", "singlequad"])
def test_backward_matches_reference_parameter_gradients(name):
    """dJc/dtheta for every parameter vs (a) the reference's fp32 autograd (tests/golden/grads.npz, made by
    make_golden_grads.py) and (b) the fp64 truth: the HIP error against fp64 may not exceed 4x the reference's own
    fp32-vs-fp64 gap (plus 2e-5 of the gradient scale)."""
    import os
    from conftest import GOLDEN_DIR, load_golden
    G = np.load(os.path.join(GOLDEN_DIR, "grads.npz"))
    g = load_golden(name)
    nt, ns = int(G[f"{name}/nt"]), int(G[f"{name}/ns"])
    net = make_net(g, DEV)
    net.train()
    prob = make_prob(g, DEV, training=True)
    x = g.t("x")[:ns].to(DEV)
    poison_allocator(DEV, big=2)                                    # (records and row streams start as NaN, not as an earlier test's values)
    Jc, cs = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", g.meta["alph"])
    J64 = float(G[f"{name}/Jc64"])
    assert abs(Jc.item() - J64) <= 1e-5 * abs(J64)
    Jc.backward()
    for k, p in net.named_parameters():
        ref32 = torch.from_numpy(G[f"{name}/grad/{k}"]).double()
        ref64 = torch.from_numpy(G[f"{name}/grad64/{k}"])
        got = p.grad.detach().cpu().double()
        assert got.shape == ref64.shape, k
        scale = ref64.abs().max().item()
        gap = (ref32 - ref64).abs().max().item()
        err = (got - ref64).abs().max().item()
        assert err <= 4 * gap + 2e-5 * scale + 1e-7, f"{name} {k}: err {err:g}, reference gap {gap:g}, scale {scale:g}"


@pytest.mark.parametrize("name,bwd_kernel", [("swarm50", "rollout_duo_bwd_kernel"), ("singlequad", "rollout_mono_bwd_kernel")])
def test_backward_at_training_size_matches_reference_parameter_gradients(name, bwd_kernel):
    """trainOC.py:172-174 at its own size: n_train rows (swarm50 1024 x nt 80, singlequad 4096 x nt 50), prob.train().  Every parameter
    gradient and Jc against tests/golden/grads_full.npz (make_golden_grads_full.py: the reference's fp32 autograd and the fp64 truth,
    chunk-summed): the HIP error against fp64 may not exceed 4x the reference's own fp32-vs-fp64 gap + 2e-5 of the gradient scale --
    the bound of the 16-row test above, now on the batch the bench times (two tiles per group on the split-role adjoint)."""
    import os
    from conftest import GOLDEN_DIR, load_golden
    from neuraloc_amd import _lib
    G = np.load(os.path.join(GOLDEN_DIR, "grads_full.npz"))
    g = load_golden(name)
    nt, n = int(G[f"{name}/nt"]), int(G[f"{name}/n"])
    assert (n, nt) == (g.meta["n_full"], g.meta["nt"])
    net = make_net(g, DEV)
    net.train()
    prob = make_prob(g, DEV, training=True)
    x = full_states(g, int(G[f"{name}/seed"])).to(DEV)
    assert x.shape[0] == n
    big = torch.full((1 << 30,), float("nan"), device=DEV)          # (4 GiB of NaN for the tape and the row streams to be carved from)
    del big
    poison_allocator(DEV, big=2)
    Jc, cs = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", g.meta["alph"])
    J64, J32 = float(G[f"{name}/Jc64"]), float(G[f"{name}/Jc"])
    assert abs(Jc.item() - J64) <= 4 * abs(J32 - J64) + 1e-5 * abs(J64), (Jc.item(), J32, J64)
    for j in range(7):
        c64, c32 = float(G[f"{name}/cs64"][j]), float(G[f"{name}/cs"][j])
        assert abs(float(cs[j]) - c64) <= 4 * abs(c32 - c64) + 1e-4 * abs(c64) + 1e-6, (j, float(cs[j]), c32, c64)
    Jc.backward()
    assert _lib.lib().nocf_last_rollout_kernel().decode() == bwd_kernel      # the adjoint the bench's training line times
    for k, p in net.named_parameters():
        ref32 = torch.from_numpy(G[f"{name}/grad/{k}"]).double()
        ref64 = torch.from_numpy(G[f"{name}/grad64/{k}"])
        got = p.grad.detach().cpu().double()
        assert got.shape == ref64.shape, k
        scale = ref64.abs().max().item()
        gap = (ref32 - ref64).abs().max().item()
        err = (got - ref64).abs().max().item()
        assert err <= 4 * gap + 2e-5 * scale + 1e-7, f"{name} {k}: err {err:g}, reference gap {gap:g}, scale {scale:g}"


def _oracle_grads64(x, sd, prob, nt, stepper, alph, nTh):
    P = orc.PhiParams.from_state_dict({k: v.clone() for k, v in sd.items()}, dtype=torch.float64)
    for t in [*P.K, *P.b, P.w, P.A, P.cw, P.cb]:
        t.requires_grad_(True)
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu()
    J, _ = orc.rollout(x.double().cpu(), P, S.to(torch.float64), [0.0, 1.0], nt, stepper, alph)
    J.backward()
    out = {"A": P.A.grad, "c.weight": P.cw.grad, "c.bias": P.cb.grad, "w.weight": P.w.grad}
    for i in range(nTh):
        out[f"N.layers.{i}.weight"], out[f"N.layers.{i}.bias"] = P.K[i].grad, P.b[i].grad
    return float(J), out


@pytest.mark.parametrize("name,n,stepper,training,nTh", [(a, b, c_, e, 2) for a, b, c_, e in [
    ("midcross4", 13, "rk4", True), ("midcross4", 16, "rk1", False), ("softcorridor", 7, "rk4", True),
    ("midcross2", 9, "rk4", False), ("swap12", 10, "rk1", True), ("swarm", 5, "rk4", True),
    ("swap2", 1, "rk4", True), ("midcross20", 12, "rk4", True), ("singlequad", 11, "rk4", True),
    ("singlequad", 8, "rk1", False)]] + [("midcross4", 9, "rk4", True, 3), ("swap12", 6, "rk4", True, 4),
                                         ("swarm", 5, "rk4", False, 3), ("singlequad", 7, "rk4", True, 3)])
def test_backward_against_oracle_fp64_autograd(name, n, stepper, training, nTh):
    """ragged batches, both steppers, both mask modes, obstacle / interaction problems: dJc/dtheta vs the oracle
    differentiated by torch autograd in fp64"""
    alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
    torch.manual_seed(11)
    prob, x0, _, _ = na.initProb(name, 24, 24, 0.5, alph, lambda t: t.float().to(DEV))
    prob.train() if training else prob.eval()
    x0 = x0[:n].contiguous()
    d = x0.shape[1]
    m = 40 if d > 30 else 24
    sd = _synth_state_dict(nTh, m, d, seed=len(name))
    net = na.Phi(nTh=nTh, m=m, d=d, alph=alph)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    nt = 6
    Jc, _ = na.OCflow(x0, net, prob, [0.0, 1.0], nt, stepper, alph)
    Jc.backward()
    J64, want = _oracle_grads64(x0, sd, prob, nt, stepper, alph, nTh)
    assert abs(Jc.item() - J64) <= 2e-5 * abs(J64)
    for k, p in net.named_parameters():
        w = want[k] if want[k] is not None else torch.zeros_like(p, dtype=torch.float64).cpu()
        scale = w.abs().max().item()
        err = (p.grad.cpu().double() - w).abs().max().item()
        assert err <= 2e-4 * scale + 1e-6, f"{name} {k}: err {err:g} at scale {scale:g}"


def test_backward_shards_add_up(golden_pretrained):
    """n_total: gradients of two shards (each normalised by the global batch) sum to the full-batch gradient"""
    from neuraloc_amd.train import ocflow_train
    g = golden_pretrained
    prob = make_prob(g, DEV, training=True)
    x = g.t("x")[:24].to(DEV)
    alph, nt = g.meta["alph"], 8
    net_full = make_net(g, DEV).train()
    J, _ = ocflow_train(x, net_full, prob, [0.0, 1.0], nt, "rk4", alph)
    J.backward()
    net_sh = make_net(g, DEV).train()
    for xs in (x[:10], x[10:]):
        Js, _ = ocflow_train(xs.contiguous(), net_sh, prob, [0.0, 1.0], nt, "rk4", alph, n_total=24)
        Js.backward()
    for (k, a), (_, b) in zip(net_full.named_parameters(), net_sh.named_parameters()):
        scale = a.grad.abs().max().item()
        assert (a.grad - b.grad).abs().max().item() <= 2e-5 * scale + 1e-7, k


def test_training_step_reduces_objective(golden_pretrained):
    """trainOC.py:160-176 in miniature: a few Adam steps through the HIP forward+backward lower Jc from a fresh net"""
    g = golden_pretrained
    torch.manual_seed(3)
    meta = g.meta
    net = na.Phi(nTh=2, m=meta["m"], d=meta["d"], alph=meta["alph"]).to(DEV)
    prob = make_prob(g, DEV, training=True)
    x = g.t("x")[:32].to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=0.01)
    hist = []
    for _ in range(12):
        opt.zero_grad()
        Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], 8, "rk4", meta["alph"])
        Jc.backward()
        opt.step()
        hist.append(Jc.item())
    assert hist[-1] < hist[0], hist


def test_backward_two_quadcopters_with_interaction():
    """Quadcopter.py:98-110 accumulates L while it subtracts it from H (the running-L loop) and adds the pair cost
    for two craft: the adjoint carries both; checked against the fp64 oracle autograd."""
    alph = [50.0, 0.0, 40.0, 0.5, 0.25, 0.125]
    d, m, n, nt = 24, 24, 6, 5
    xt = torch.zeros(d)
    xt[0:3] = torch.tensor([2.0, 2.0, 2.0]); xt[12:15] = torch.tensor([-2.0, 2.0, 2.0])
    prob = na.Quadcopter(xt.to(DEV), obstacle=None, alph_Q=0.0, alph_W=alph[2], r=1.5)
    prob.train()
    i = torch.arange(n * d, dtype=torch.float64).reshape(n, d)
    x0 = (0.3 * torch.sin(0.7 * i + 0.2)).float()
    x0[:, 12:15] += torch.tensor([0.8, 0.3, -0.2])             # the two craft start within 2r of each other
    x0 = x0.to(DEV)
    sd = _synth_state_dict(2, m, d, seed=5)
    net = na.Phi(nTh=2, m=m, d=d, alph=alph)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    Jc, cs = na.OCflow(x0, net, prob, [0.0, 1.0], nt, "rk4", alph)
    assert cs[6].item() > 0.0                                   # the interaction term is live
    Jc.backward()
    J64, want = _oracle_grads64(x0, sd, prob, nt, "rk4", alph, 2)
    assert abs(Jc.item() - J64) <= 2e-5 * abs(J64)
    for k, p in net.named_parameters():
        w = want[k] if want[k] is not None else torch.zeros_like(p, dtype=torch.float64).cpu()
        scale = w.abs().max().item()
        err = (p.grad.cpu().double() - w).abs().max().item()
        assert err <= 2e-4 * scale + 1e-6, f"{k}: err {err:g} at scale {scale:g}"


@pytest.mark.parametrize("name", ["swarm50", "singlequad"])
def test_specialised_and_generic_instantiations_agree(name, monkeypatch):
    """the shape-specialised kernel (compile-time plan) and the generic one (plan read from the workspace) are the
    same arithmetic in the same order: per-sample tables, trajectories, controls and parameter gradients agree to ulps"""
    from conftest import load_golden
    g = load_golden(name)
    alph, nt = g.meta["alph"], 10
    x = g.t("x").to(DEV)
    out = {}
    for fixed in ("1", "0"):
        monkeypatch.setenv("NOCF_FIXED", fixed)
        net = make_net(g, DEV)
        prob = make_prob(g, DEV, training=False)
        with torch.no_grad():
            _, csn = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph, noMean=True)
            zF, cF = na.OCflow(x[:7], net, prob, [0.0, 1.0], nt, "rk4", alph, intermediates=True)
        net.train(); prob.train()
        Jc, _ = na.OCflow(x[:12].contiguous(), net, prob, [0.0, 1.0], 6, "rk4", alph)
        Jc.backward()
        out[fixed] = (torch.cat(csn, 1).cpu(), zF.cpu(), cF.cpu(), Jc.item(),
                      torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu())
    a, b = out["1"], out["0"]
    # same source, two instantiations: the compiler may contract a*b+c differently once strides are literals, so the
    # comparison allows a few ulps (swarm50 comes out bit-identical, singlequad differs in the last bits)
    for u, v in zip(a[:3], b[:3]):
        assert (u - v).abs().max().item() <= 2e-6 * v.abs().max().item()
    assert abs(a[3] - b[3]) <= 2e-6 * abs(b[3])
    scale = b[4].abs().max().item()
    assert (a[4] - b[4]).abs().max().item() <= 2e-6 * scale


@pytest.mark.parametrize("name", ["midcross4", "midcross20", "midcross30", "swarm", "swap12_3pair", "swap12_4pair", "swap12_5pair"])
def test_specialised_default_width_shapes_agree_with_generic(name, monkeypatch):
    """the initProb problems at the reference's default width (m = 32, nTh = 2) have specialised tile-kernel
    instantiations (evaluation when d+1 > 32, record + adjoint for all): same results as the generic instantiation"""
    alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
    torch.manual_seed(3)
    prob, x0, _, _ = na.initProb(name, 20, 20, 0.5, alph, lambda t: t.float().to(DEV))
    d = x0.shape[1]
    sd = _synth_state_dict(2, 32, d, seed=len(name))
    out = {}
    for fixed in ("1", "0"):
        monkeypatch.setenv("NOCF_FIXED", fixed)
        monkeypatch.setenv("NOCF_LANE", "0")                 # the tile kernels, also where the lane kernel would qualify
        net = na.Phi(nTh=2, m=32, d=d, alph=alph)
        net.load_state_dict(sd)
        net = net.to(DEV)
        prob.eval()
        with torch.no_grad():
            _, csn = na.OCflow(x0, net, prob, [0.0, 1.0], 6, "rk4", alph, noMean=True)
        net.train(); prob.train()
        Jc, _ = na.OCflow(x0, net, prob, [0.0, 1.0], 6, "rk4", alph)
        Jc.backward()
        out[fixed] = (torch.cat(csn, 1).cpu(), Jc.item(), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu())
    a, b = out["1"], out["0"]
    assert (a[0] - b[0]).abs().max().item() <= 2e-6 * b[0].abs().max().item()
    assert abs(a[1] - b[1]) <= 2e-6 * abs(b[1])
    assert (a[2] - b[2]).abs().max().item() <= 2e-6 * b[2].abs().max().item()


@pytest.mark.parametrize("name", ["softcorridor", "swap12", "singlequad"])
def test_backward_gradient_wrt_initial_states(name):
    """x0.requires_grad: Jc.backward() also fills x0.grad (the adjoint at t0), like autograd through the reference"""
    from conftest import load_golden
    g = load_golden(name)
    alph, nt = g.meta["alph"], 6
    net = make_net(g, DEV).train()
    prob = make_prob(g, DEV, training=True)
    x = g.t("x")[:9].to(DEV).requires_grad_(True)
    Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph)
    Jc.backward()
    P = orc.PhiParams.from_state_dict(g.state_dict(), dtype=torch.float64)
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu()
    x64 = g.t("x")[:9].double().requires_grad_(True)
    J64, _ = orc.rollout(x64, P, S.to(torch.float64), [0.0, 1.0], nt, "rk4", alph)
    J64.backward()
    # the fp32 oracle (= the reference's arithmetic) differentiated by autograd gives the fp32 noise floor per sample
    P32 = orc.PhiParams.from_state_dict(g.state_dict())
    x32 = g.t("x")[:9].clone().requires_grad_(True)
    J32, _ = orc.rollout(x32, P32, S, [0.0, 1.0], nt, "rk4", alph)
    J32.backward()
    got, want = x.grad.cpu().double(), x64.grad
    norm = want.abs().max(dim=1).values
    row_err = (got - want).abs().max(dim=1).values / norm
    ref_err = (x32.grad.double() - want).abs().max(dim=1).values / norm
    # a pair sitting on the interaction threshold can flip its mask between two fp32 evaluation orders (the lane kernel
    # that records the stages, the tile kernel, the oracle): such a sample differs at the percent level -- counted and
    # bounded, like the mask-flip rows of the forward tests -- every other row stays within the fp32 noise floor
    bad = row_err > 4 * ref_err + 2e-4
    assert int(bad.sum()) <= 1 and float(row_err.max()) <= 5e-2, (row_err, ref_err)


@pytest.mark.parametrize("m,nTh,d_name", [(1024, 2, "midcross20"), (128, 6, "singlequad"), (320, 3, "swarm")])
def test_wide_and_deep_networks_against_oracle(m, nTh, d_name):
    """shapes far from the shipped checkpoints (16 column blocks on 8 waves; six residual layers; an odd width):
    the generic instantiation against the oracle"""
    alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
    torch.manual_seed(5)
    prob, x0, _, _ = na.initProb(d_name, 10, 10, 0.5, alph, lambda t: t.float().to(DEV))
    prob.eval()
    d = x0.shape[1]
    sd = _synth_state_dict(nTh, m, d, seed=m % 7)
    net = na.Phi(nTh=nTh, m=m, d=d, alph=alph)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    P = orc.PhiParams.from_state_dict(sd)
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu()
    with torch.no_grad():
        _, csn = na.OCflow(x0, net, prob, [0.0, 1.0], 5, "rk4", alph, noMean=True)
        want = orc.persample_table(x0.cpu(), P, S, [0.0, 1.0], 5, "rk4", alph)
    tab = torch.cat(csn, 1).cpu()
    off = (tab.double() - want.double()).abs() > 1e-3 + 1e-3 * want.double().abs()
    assert int(off.any(dim=1).sum()) == 0, f"{int(off.any(dim=1).sum())} samples off"


def test_backward_on_a_time_segment_with_many_steps(golden_pretrained):
    """tspan = [0.2, 0.7], nt = 40: the adjoint uses the same double-precision time bookkeeping as the forward sweep"""
    g = golden_pretrained
    alph, nt = g.meta["alph"], 40
    net = make_net(g, DEV).train()
    prob = make_prob(g, DEV, training=True)
    x = g.t("x")[:6].to(DEV)
    Jc, _ = na.OCflow(x, net, prob, [0.2, 0.7], nt, "rk4", alph)
    Jc.backward()
    P = orc.PhiParams.from_state_dict(g.state_dict(), dtype=torch.float64)
    leaves = [*P.K, *P.b, P.w, P.A, P.cw, P.cb]
    for t in leaves:
        t.requires_grad_(True)
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu()
    J64, _ = orc.rollout(g.t("x")[:6].double(), P, S.to(torch.float64), [0.2, 0.7], nt, "rk4", alph)
    J64.backward()
    assert abs(Jc.item() - float(J64)) <= 2e-5 * abs(float(J64))
    want = {"A": P.A.grad, "c.weight": P.cw.grad, "c.bias": P.cb.grad, "w.weight": P.w.grad,
            "N.layers.0.weight": P.K[0].grad, "N.layers.0.bias": P.b[0].grad, "N.layers.1.weight": P.K[1].grad, "N.layers.1.bias": P.b[1].grad}
    for k, p in net.named_parameters():
        w = want[k] if want[k] is not None else torch.zeros_like(p, dtype=torch.float64).cpu()
        scale = w.abs().max().item()
        assert (p.grad.cpu().double() - w).abs().max().item() <= 1e-3 * scale + 1e-6, k


@pytest.mark.parametrize("K,m,n", [(1, 1, 1), (257, 32, 25), (5000, 16, 5), (40001, 64, 64), (1000, 1, 33), (333, 61, 3),
                                   (20011, 128, 128), (3000, 128, 13), (9000, 1, 128), (700, 100, 97)])
def test_small_output_contraction_kernel(K, m, n):
    """nocf_contract_f32 (the weight-gradient contraction of the small networks) against a float64 matmul; accumulate mode"""
    from neuraloc_amd.train import _contract
    g = torch.Generator().manual_seed(K + m + n)
    X = torch.randn(K, m, generator=g).to(DEV)
    Y = torch.randn(K, n, generator=g).to(DEV)
    want = (X.double().t() @ Y.double()).cpu()
    got = _contract(X, Y).cpu().double()
    tol = 1e-5 * (K ** 0.5) + 1e-5
    assert got.shape == want.shape and (got - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    acc = _contract(X, Y, _contract(X, Y)).cpu().double()
    assert (acc - 2 * want).abs().max().item() <= 2 * tol * max(1.0, want.abs().max().item())
    again = _contract(X, Y).cpu().double()
    assert torch.equal(again, got)                       # fixed-order two-stage sum: run-to-run identical


def test_jit_specialisation_of_an_unlisted_shape(monkeypatch, capfd):
    """NOCF_JIT=1: a shape without a built-in specialised instantiation gets its own library (hipcc, once, cached);
    it takes the specialised kernels and agrees with the generic instantiation"""
    import shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc on this box")
    alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
    torch.manual_seed(9)
    prob, x0, _, _ = na.initProb("midcross4", 16, 16, 0.5, alph, lambda t: t.float().to(DEV))
    d, m = x0.shape[1], 48
    sd = _synth_state_dict(2, m, d, seed=4)
    out = {}
    for jit in ("0", "1"):
        monkeypatch.setenv("NOCF_JIT", jit)
        monkeypatch.setenv("NOCF_DEBUG", "1")
        monkeypatch.setenv("NOCF_MONO", "0")             # this shape would take the one-CU kernel, which is not shape-specialised
        net = na.Phi(nTh=2, m=m, d=d, alph=alph)
        net.load_state_dict(sd)
        net = net.to(DEV)
        prob.eval()
        capfd.readouterr()
        with torch.no_grad():
            _, csn = na.OCflow(x0, net, prob, [0.0, 1.0], 6, "rk4", alph, noMean=True)
        err = capfd.readouterr().err
        assert ("shape-specialised" in err) == (jit == "1"), err[-400:]
        net.train(); prob.train()
        Jc, _ = na.OCflow(x0, net, prob, [0.0, 1.0], 6, "rk4", alph)
        Jc.backward()
        out[jit] = (torch.cat(csn, 1).cpu(), Jc.item(), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu())
    a, b = out["1"], out["0"]
    assert (a[0] - b[0]).abs().max().item() <= 2e-6 * b[0].abs().max().item()
    assert abs(a[1] - b[1]) <= 2e-6 * abs(b[1])
    assert (a[2] - b[2]).abs().max().item() <= 2e-6 * b[2].abs().max().item()


def test_jit_auto_mode_compiles_in_the_background_and_the_next_process_uses_the_cache(tmp_path):
    """NOCF_JIT=auto (the default outside this test-suite): the first process that meets an unlisted shape keeps the generic
    instantiation and starts hipcc in a child process; a later process loads the cached per-shape library and takes the
    specialised kernels; both agree"""
    import shutil
    import subprocess
    import sys
    import time
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc on this box")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    key = (8, 40, 2, 9, 4)
    so = os.path.join(repo, "neuraloc_amd", "csrc", "jit", "libnocf_d%d_m%d_t%d_r%d_a%d.so" % key)
    for f in (so, so + ".log"):
        if os.path.exists(f):
            os.unlink(f)
    child = r"""
import os, sys
sys.path.insert(0, os.environ["NOCF_REPO"]); sys.path.insert(0, os.path.join(os.environ["NOCF_REPO"], "tests"))
import torch
import neuraloc_amd as na
from util_hip import synth_state_dict
dev = torch.device("cuda:0")
alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
torch.manual_seed(9)
prob, x0, _, _ = na.initProb("midcross4", 16, 16, 0.5, alph, lambda t: t.float().to(dev))
net = na.Phi(nTh=2, m=40, d=x0.shape[1], alph=alph)
net.load_state_dict(synth_state_dict(2, 40, x0.shape[1], seed=4))
net = net.to(dev); prob.eval()
with torch.no_grad():
    a = na.OCflow(x0, net, prob, [0.0, 1.0], 6, "rk4", alph, noMean=True)[1]
    b = na.OCflow(x0, net, prob, [0.0, 1.0], 6, "rk4", alph, noMean=True)[1]
torch.cuda.synchronize()
assert all(torch.equal(u, v) for u, v in zip(a, b))          # one process, one kernel: run-to-run identical
print("TABLE", " ".join("%.9e" % float(c.double().sum()) for c in a))
"""
    env = dict(os.environ)
    env.update({"NOCF_JIT": "auto", "NOCF_DEBUG": "1", "NOCF_MONO": "0", "NOCF_REPO": repo})
    env.pop("NOCF_LIB_PATH", None)
    r1 = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-3000:]
    assert "compiling them in the background" in r1.stderr and "shape-specialised" not in r1.stderr
    t0 = time.time()
    while not os.path.exists(so) and time.time() - t0 < 420:
        time.sleep(2.0)
    assert os.path.exists(so), "the background compilation did not deliver: " + (open(so + ".log").read()[-2000:] if os.path.exists(so + ".log") else "no log")
    r2 = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert "shape-specialised" in r2.stderr and "compiling them in the background" not in r2.stderr
    t1 = [float(v) for v in [ln for ln in r1.stdout.splitlines() if ln.startswith("TABLE")][-1].split()[1:]]
    t2 = [float(v) for v in [ln for ln in r2.stdout.splitlines() if ln.startswith("TABLE")][-1].split()[1:]]
    assert all(abs(a - b) <= 2e-6 * max(1.0, abs(b)) for a, b in zip(t1, t2))


@pytest.mark.parametrize("name", ["swap2", "softcorridor", "swap12"])
def test_lane_adjoint_agrees_with_tile_adjoint(name, monkeypatch, capfd):
    """small networks: the one-wave-per-sample adjoint (gradient rows in registers) and the 4-samples-per-wave tile
    adjoint (row streams + contractions) are two implementations of the same gradient"""
    from conftest import load_golden
    g = load_golden(name)
    alph, nt = g.meta["alph"], 7
    x = g.t("x")[:21].to(DEV)
    out = {}
    for lane in ("1", "0"):
        monkeypatch.setenv("NOCF_LANE", lane)
        monkeypatch.setenv("NOCF_DEBUG", "1")
        net = make_net(g, DEV).train()
        prob = make_prob(g, DEV, training=True)
        xx = x.clone().requires_grad_(True)
        capfd.readouterr()
        Jc, _ = na.OCflow(xx, net, prob, [0.0, 1.0], nt, "rk4", alph)
        Jc.backward()
        err = capfd.readouterr().err
        assert ("lane adjoint kernel" in err) == (lane == "1")
        out[lane] = (Jc.item(), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu(), xx.grad.cpu())
    a, b = out["1"], out["0"]
    assert abs(a[0] - b[0]) <= 1e-5 * abs(b[0])
    # per-sample adjoints tell whether a pair sat on the interaction threshold and flipped its mask between the two
    # forward sweeps (lane vs tile rounding): at most one such sample, and only then may the summed gradients differ
    # beyond fp32 accumulation noise
    row = (a[2] - b[2]).abs().max(dim=1).values / b[2].abs().max(dim=1).values
    flips = int((row > 1e-3).sum())
    assert flips <= 1, row
    tol = 2e-4 if flips == 0 else 2e-2
    assert (a[1] - b[1]).abs().max().item() <= tol * b[1].abs().max().item()
