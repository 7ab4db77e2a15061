"""GPU parity tests of the split-role weight-stationary kernel (neuraloc_amd/csrc/nocf_duo.hip): wide two-layer networks
(m = 512) on point-agent problems.  Checked against the oracle, against the reference's stored outputs (swarm50 fixture) and
against the per-tile kernel (NOCF_DUO=0) on the same inputs.

Tolerances as in test_hip_parity.py: per-sample costs rel 1e-3 + abs 1e-3 (mask flips counted), means rel 1e-4."""
import os

import numpy as np
import pytest
import torch

import neuraloc_amd as na
from neuraloc_amd import _lib
from oracle import ocflow_oracle as orc
from conftest import load_golden
from util_hip import closed_form_normal, count_off, make_net, make_oracle, make_prob, poison_allocator, synth_state_dict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
ALPH = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]


@pytest.fixture(autouse=True, params=["g16", "g8"])
def form(request, monkeypatch):
    """every test of this file runs on both group geometries of the forward (nocf_duo.hip, DuoCfg): 16 members of 32 hidden units with the
    contraction split over wave pairs (the default since round 5) and 8 members of 64 (rounds 3-4; the adjoint's geometry)"""
    monkeypatch.setenv("NOCF_DUO_G", request.param[1:])
    return request.param


def _table(x, net, prob, tspan, nt, stepper, alph):
    with torch.no_grad():
        _, csn = na.OCflow(x, net, prob, tspan, nt, stepper, alph, noMean=True)
    return torch.cat(csn, 1).cpu()


def _flips(tab, want):
    off = (tab.double() - want.double()).abs() > 1e-3 + 1e-3 * want.double().abs()
    return int(off.any(dim=1).sum())


SENS_EPS = 3e-6


def _sensitive_in_fp64(xr, P64, S64, nt, alph, eps=SENS_EPS, table=False):
    """Is a row one on which two correct fp32 evaluations may disagree?  The oracle in float64 is integrated from x and from x (1 +- eps),
    eps = 3e-6 -- the size of the state difference between two fp32 evaluations of one rollout (the reference's own fp32 run differs from
    its fp64 run by up to 1.1e-6 relative in the states) -- and the row is SENSITIVE when either moves one of its seven costs past
    the per-sample tolerance (rel 1e-3 + abs 1e-3): chaotic rows, and rows whose eval-mode obstacle / interaction mask flips (one-sided,
    hence both signs).  Returns (mask, change / tolerance per row[, the fp64 table])."""
    xs = xr.double()
    a = orc.persample_table(xs, P64, S64, [0.0, 1.0], nt, "rk4", alph)
    amp = torch.zeros(xs.shape[0], dtype=torch.float64)
    for sgn in (1.0, -1.0):
        b = orc.persample_table(xs * (1.0 + sgn * eps), P64, S64, [0.0, 1.0], nt, "rk4", alph)
        amp = torch.maximum(amp, ((a - b).abs() / (1e-3 + 1e-3 * a.abs())).amax(dim=1))
    return (amp > 1.0, amp, a) if table else (amp > 1.0, amp)


def _explained(xr, ta, tb, P64, S64, nt, alph):
    """Rows on which two kernels differ by more than the per-sample tolerance are accepted for exactly two reasons, both CHECKED in the float64
    oracle: the row is sensitive (_sensitive_in_fp64: no fp32 evaluation can be expected to reproduce it), or it is MILDLY amplifying (it moves
    by >= 10 % of the tolerance under the 3e-6 change; ordinary rows: 2-3 %) and both kernels are within 1.25 tolerances of the fp64 truth,
    on opposite sides of it.  The one such row these sweeps meet, row 2624 of the 4096-row batch, in units of the tolerance: fp32 ORACLE (the
    reference's own arithmetic) +0.51 ... +0.61 depending on the host's BLAS, per-tile kernel +1.02, default geometry +0.29, fine geometry
    -0.57; amplification 0.2.  Anything else is a kernel error.  Returns (mask of explained rows, sensitivity mask, amplification)."""
    sens, amp, o64 = _sensitive_in_fp64(xr, P64, S64, nt, alph, table=True)
    tol = 1.25 * (1e-3 + 1e-3 * o64.abs())
    near = ((ta.double() - o64).abs() <= tol).all(dim=1) & ((tb.double() - o64).abs() <= tol).all(dim=1) & (amp >= 0.1)
    return sens | near, sens, amp


def _kernel():
    return _lib.lib().nocf_last_rollout_kernel().decode()


def test_duo_kernel_is_the_default_for_swarm50(monkeypatch):
    g = load_golden("swarm50")
    net, prob = make_net(g, DEV), make_prob(g, DEV, training=False)
    x = g.t("x").to(DEV)
    for k in ("NOCF_DUO", "NOCF_DUO_MAP", "NOCF_DUO_FAST"):
        monkeypatch.delenv(k, raising=False)
    _table(x, net, prob, [0.0, 1.0], 4, "rk4", g.meta["alph"])
    assert _kernel() == "rollout_duo_kernel"
    monkeypatch.setenv("NOCF_DUO", "0")
    _table(x, net, prob, [0.0, 1.0], 4, "rk4", g.meta["alph"])
    assert _kernel().startswith("rollout_kernel")


@pytest.mark.parametrize("n", [1, 5, 16, 17, 39, 100, 128, 256, 512, 513, 1000, 1024, 1025, 2048, 2049, 4096])
@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("variant", ["default", "write-through", "static-map"])
def test_duo_matches_tile_kernel_and_oracle_on_swarm50(n, training, variant, monkeypatch, form):
    """pretrained swarm50 network, batches that fill 1..32 groups with one to four sample tiles and more than one launch (ragged
    tails included); both exchange forms (default: same-XCD groups keep their payload in L2; write-through everywhere) and both
    role maps (default: the CU census pairs the two roles of a member on one CU; static).
    A few of these states are chaotic at nt = 10 (a 1e-6 relative change of x moves their terminal cost by 2 %, in the oracle
    too), so per-sample rows may differ between two correct fp32 evaluations: such rows are counted and bounded, the batch means
    must agree."""
    if variant != "default" and n not in ((5, 100, 1024, 2049) if form == "g8" else (5, 100, 256)):
        pytest.skip("the exchange / map variants run on a subset of the sizes")
    if form == "g16" and n in (513, 1024, 1025, 2048, 4096):
        pytest.skip("the fine geometry is the library's choice up to 256 rows; forced beyond, it runs 512 / 1000 / 2049 rows (two to four tiles per group, a second launch)")
    g = load_golden("swarm50")
    net, prob = make_net(g, DEV), make_prob(g, DEV, training=training)
    m = g.meta
    x = (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], 3)).contiguous()
    nt = 10
    monkeypatch.setenv("NOCF_DUO", "1")
    if variant == "write-through":
        monkeypatch.setenv("NOCF_DUO_FAST", "0")
    if variant == "static-map":
        monkeypatch.setenv("NOCF_DUO_MAP", "1")
    duo = _table(x.to(DEV), net, prob, [0.0, 1.0], nt, "rk4", m["alph"])
    assert _kernel() == "rollout_duo_kernel"
    again = _table(x.to(DEV), net, prob, [0.0, 1.0], nt, "rk4", m["alph"])
    assert torch.equal(duo, again), "the split-role kernel is not run-to-run deterministic"
    assert not torch.isnan(duo).any()
    monkeypatch.setenv("NOCF_DUO", "0")
    tile = _table(x.to(DEV), net, prob, [0.0, 1.0], nt, "rk4", m["alph"])
    # Rows on which two correct fp32 evaluations may differ are CHECKED, not assumed (_explained): every differing row must amplify a 3e-6
    # relative change of its input past the tolerance in the FLOAT64 oracle, or have both kernels within its tolerance (as the test at the end of this
    # file shows on 4096 rows); a differing row that is not sensitive there is a kernel error.  Their number stays bounded (n / 128; the
    # default geometry and the per-tile kernel differ on 8 of 4096, the fine geometry -- other summation order -- on up to 4 of 1000).
    off = ((duo.double() - tile.double()).abs() > 1e-3 + 1e-3 * tile.double().abs()).any(dim=1)
    rows = torch.nonzero(off).flatten()
    assert len(rows) <= max(2, n // 128), f"split-role vs tile kernel: {len(rows)} samples differ"
    if len(rows):
        P64 = orc.PhiParams.from_state_dict(g.state_dict(), dtype=torch.float64)
        S64 = make_oracle(g, training)[1].to(torch.float64)
        ok, _, _ = _explained(x[rows], duo[rows], tile[rows], P64, S64, nt, m["alph"])
        assert bool(ok.all()), f"rows {rows[~ok].tolist()} differ between the kernels, are not sensitive in the fp64 oracle and not both within its tolerance"
    keep = ~off
    for j in range(7):                                   # batch means over the rows that are not chaotic / mask-flipped
        a, b = duo[keep, j].double().mean().item(), tile[keep, j].double().mean().item()
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, f"column {j}: mean {a} vs {b}"
    if n <= 100:
        P, S = make_oracle(g, training)
        want = orc.persample_table(x, P, S, [0.0, 1.0], nt, "rk4", m["alph"])
        assert _flips(duo, want) <= 2, f"split-role kernel vs oracle: {_flips(duo, want)} samples off"
    na.check_errors(sync=True)


@pytest.mark.parametrize("width", [512, 256, 384, 192, 130])
@pytest.mark.parametrize("name,stepper,tspan,training", [
    ("swarm", "rk4", [0.0, 1.0], False), ("swarm", "rk1", [0.0, 1.0], True), ("midcross20", "rk4", [0.25, 0.9], True),
    ("swap12", "rk4", [0.0, 1.0], False), ("softcorridor", "rk4", [0.0, 1.0], True), ("swap2", "rk1", [0.1, 0.7], False),
    ("midcross30", "rk4", [0.0, 1.0], False), ("hardcorridor", "rk4", [0.0, 1.0], False)])
def test_duo_on_other_point_agent_problems(name, stepper, tspan, training, width, form):
    """m = 512 and m = 256 networks (closed-form weights; src/Phi.py:16-52 is uniform in m) on Cross2D / SwarmTraj problems of other dimensions:
    d+1 from 5 to 97.  The 256-wide network runs with four members of 64 hidden units per group (DuoCfg<4, 16>), four own samples per member;
    every other width between 129 and 511 runs zero-padded to 256 or 512 (a padded unit adds exact zeros)."""
    if name not in na.initProb.__globals__["PROBLEM_NAMES"]:
        pytest.skip("not an initProb problem")
    if width <= 256 and form == "g8":
        pytest.skip("networks of up to 256 hidden units have one geometry")
    if width not in (512, 256) and (stepper != "rk4" or name in ("midcross30", "hardcorridor")):
        pytest.skip("the padded widths run on a subset of the problems")
    torch.manual_seed(11)
    prob, x0, _, _ = na.initProb(name, 37, 8, 0.5, ALPH, lambda t: t.float().to(DEV))
    prob.train() if training else prob.eval()
    d = x0.shape[1]
    sd = synth_state_dict(2, width, d, seed=d % 5)
    net = na.Phi(nTh=2, m=width, d=d, alph=ALPH)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    P = orc.PhiParams.from_state_dict(sd)
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu()
    nt = 5
    got = _table(x0, net, prob, tspan, nt, stepper, ALPH)
    assert _kernel() == "rollout_duo_kernel"
    want = orc.persample_table(x0.cpu(), P, S, tspan, nt, stepper, ALPH)
    assert _flips(got, want) <= (2 if training else 1), f"{name}: {_flips(got, want)} samples off"
    with torch.no_grad():
        Jc, cs = na.OCflow(x0, net, prob, tspan, nt, stepper, ALPH)
    for j in range(7):
        assert abs(float(cs[j]) - got[:, j].double().mean().item()) <= 2e-6 * abs(float(cs[j])) + 1e-9


@pytest.mark.parametrize("n", [70, 1030, 4100])
def test_duo_256_wide_swarm_network_matches_tile_kernel_and_oracle(n, monkeypatch):
    """a swarm50-shaped network of 256 hidden units (closed-form weights) on the swarm50 problem: one to four tiles per group, ragged tails,
    a second launch (64 groups x 4 tiles = 4096 rows per launch) -- against the per-tile kernel on every row and the oracle on the small batch"""
    g = load_golden("swarm50")
    m = g.meta
    prob = make_prob(g, DEV, training=False)
    sd = synth_state_dict(2, 256, m["d"], seed=2)
    net = na.Phi(nTh=2, m=256, d=m["d"], alph=m["alph"])
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    x = (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], 5)).contiguous()
    nt = 6
    monkeypatch.setenv("NOCF_DUO", "1")
    duo = _table(x.to(DEV), net, prob, [0.0, 1.0], nt, "rk4", m["alph"])
    assert _kernel() == "rollout_duo_kernel"
    again = _table(x.to(DEV), net, prob, [0.0, 1.0], nt, "rk4", m["alph"])
    assert torch.equal(duo, again), "not run-to-run deterministic"
    monkeypatch.setenv("NOCF_DUO", "0")
    tile = _table(x.to(DEV), net, prob, [0.0, 1.0], nt, "rk4", m["alph"])
    assert _kernel().startswith("rollout_kernel")
    off = ((duo.double() - tile.double()).abs() > 1e-3 + 1e-3 * tile.double().abs()).any(dim=1)
    rows = torch.nonzero(off).flatten()
    assert len(rows) <= max(2, n // 128), f"{len(rows)} samples differ from the per-tile kernel"
    P64 = orc.PhiParams.from_state_dict(sd, dtype=torch.float64)
    S64 = make_oracle(g, False)[1].to(torch.float64)
    if len(rows):
        ok, _, _ = _explained(x[rows], duo[rows], tile[rows], P64, S64, nt, m["alph"])
        assert bool(ok.all()), f"rows {rows[~ok].tolist()} differ from the per-tile kernel without a reason in the fp64 oracle"
    for j in range(7):
        a, b = duo[~off, j].double().mean().item(), tile[~off, j].double().mean().item()
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, f"column {j}: mean {a} vs {b}"
    if n <= 100:
        want = orc.persample_table(x, orc.PhiParams.from_state_dict(sd), make_oracle(g, False)[1], [0.0, 1.0], nt, "rk4", m["alph"])
        assert _flips(duo, want) <= 2
    # ... intermediates (one more evaluation per step) and the recording forward (stage inputs) of the same geometry against the per-tile kernel
    xs = x[:min(n, 1030)].to(DEV)
    out, rec = {}, {}
    for flag in ("1", "0"):
        monkeypatch.setenv("NOCF_DUO", flag)
        with torch.no_grad():
            zF, cF = na.OCflow(xs, net, prob, [0.0, 1.0], 4, "rk4", m["alph"], intermediates=True)
        assert (_kernel() == "rollout_duo_kernel") == (flag == "1")
        out[flag] = (zF.cpu(), cF.cpu())
        rec[flag] = _record(xs, net, prob, 3, m["alph"])
        assert (_kernel() == "rollout_duo_kernel") == (flag == "1")
    d = m["d"]
    assert torch.equal(out["1"][0][:, :, 0], out["0"][0][:, :, 0]) and float(out["1"][1][:, :, 0].abs().max()) == 0.0
    assert (out["1"][0][:, :d, :3] - out["0"][0][:, :d, :3]).abs().max().item() <= 2e-3
    assert (out["1"][1][:, :, :3] - out["0"][1][:, :, :3]).abs().max().item() <= 2e-2 * out["0"][1].abs().max().item()
    (_, zd, sd_, _), (_, zt, st_, _) = rec["1"], rec["0"]
    assert torch.equal(sd_[0], st_[0]) and torch.equal(sd_[:, :, -1], st_[:, :, -1])
    assert (sd_[:8] - st_[:8]).abs().max().item() <= 2e-3 and (sd_ != 0).any(dim=2).all()
    na.check_errors(sync=True)


def test_duo_full_size_against_reference():
    """BASELINE size (n = 1024, nt = 80) on the pretrained network: the reference's stored means"""
    from util_hip import full_states
    g = load_golden("swarm50")
    if not g.has("full/Jc"):
        pytest.skip("no full-size entry")
    net, prob = make_net(g, DEV), make_prob(g, DEV, training=False)
    x = full_states(g, int(g["full/seed"])).to(DEV)
    with torch.no_grad():
        Jc, cs = na.OCflow(x, net, prob, [0.0, 1.0], g.meta["nt"], "rk4", g.meta["alph"])
    assert _kernel() == "rollout_duo_kernel"
    assert abs(float(Jc) - float(g["full/Jc"])) <= 1e-4 * abs(float(g["full/Jc"]))
    for j in range(7):
        want = float(g["full/cs"][j])
        assert abs(float(cs[j]) - want) <= 1e-4 * abs(want) + 1e-6


@pytest.mark.parametrize("name,training", [("swap12", False), ("swap12", True), ("midcross20", False), ("swarm", True), ("hardcorridor", False)])
@pytest.mark.parametrize("n", [777, 1500])
def test_duo_several_tiles_on_other_problems_against_the_tile_kernel(name, training, n, monkeypatch):
    """two and three sample tiles per group on Cross2D / SwarmTraj problems of other sizes (other agent counts in the cost pass)"""
    if name not in na.initProb.__globals__["PROBLEM_NAMES"]:
        pytest.skip("not an initProb problem")
    torch.manual_seed(13)
    prob, x0, _, _ = na.initProb(name, n, 8, 0.5, ALPH, lambda t: t.float().to(DEV))
    prob.train() if training else prob.eval()
    d = x0.shape[1]
    net = na.Phi(nTh=2, m=512, d=d, alph=ALPH)
    net.load_state_dict(synth_state_dict(2, 512, d, seed=d % 5))
    net = net.to(DEV).eval()
    monkeypatch.setenv("NOCF_DUO", "1")
    duo = _table(x0, net, prob, [0.0, 1.0], 5, "rk4", ALPH)
    assert _kernel() == "rollout_duo_kernel"
    assert torch.equal(duo, _table(x0, net, prob, [0.0, 1.0], 5, "rk4", ALPH)), "not run-to-run deterministic"
    monkeypatch.setenv("NOCF_DUO", "0")
    tile = _table(x0, net, prob, [0.0, 1.0], 5, "rk4", ALPH)
    assert _flips(duo, tile) <= max(2, n // 128), f"{name}: {_flips(duo, tile)} samples differ"
    keep = ~((duo.double() - tile.double()).abs() > 1e-3 + 1e-3 * tile.double().abs()).any(dim=1)
    for j in range(7):
        a, b = duo[keep, j].double().mean().item(), tile[keep, j].double().mean().item()
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, f"{name} column {j}: mean {a} vs {b}"


def test_a_timed_out_exchange_raises_and_poisons_the_outputs(monkeypatch):
    """NOCF_DUO_SPIN_MAX=1 (diagnostic knob) lets every bounded poll give up at once: the kernel must finish (no hang), the means must be
    NaN and the Python layer must raise -- a call whose results are consumed on the host (noMean, intermediates) raises ITSELF, a means
    call (asynchronous by design: no host synchronisation on the path) at the next call into the package or in check_errors(sync=True)."""
    g = load_golden("swarm50")
    net, prob = make_net(g, DEV), make_prob(g, DEV, training=False)
    x = g.t("x").to(DEV)
    monkeypatch.setenv("NOCF_DUO", "1")
    monkeypatch.setenv("NOCF_DUO_SPIN_MAX", "1")
    with torch.no_grad():
        Jc, cs = na.OCflow(x, net, prob, [0.0, 1.0], 6, "rk4", g.meta["alph"])
    torch.cuda.synchronize()
    assert torch.isnan(Jc) and all(torch.isnan(c) for c in cs)
    with pytest.raises(RuntimeError, match="timed out"):
        na.check_errors(sync=True)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="timed out"):
            na.OCflow(x, net, prob, [0.0, 1.0], 6, "rk4", g.meta["alph"], noMean=True)
        with pytest.raises(RuntimeError, match="timed out"):
            na.OCflow(x, net, prob, [0.0, 1.0], 6, "rk4", g.meta["alph"], intermediates=True)
        Jc, cs = na.OCflow(x, net, prob, [0.0, 1.0], 6, "rk4", g.meta["alph"])
    torch.cuda.synchronize()
    assert torch.isnan(Jc)
    monkeypatch.delenv("NOCF_DUO_SPIN_MAX")
    with pytest.raises(RuntimeError, match="timed out"):             # the failed call's status has arrived: the next call raises
        na.OCflow(x, net, prob, [0.0, 1.0], 6, "rk4", g.meta["alph"])
    with torch.no_grad():                                            # ... once; the package then works again
        Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], 6, "rk4", g.meta["alph"])
    na.check_errors(sync=True)
    assert torch.isfinite(Jc)


def test_a_timed_out_training_step_poisons_the_gradients(monkeypatch):
    """the same knob on the training path: forward tape + split-role adjoint; the parameter gradients must come out NaN (never
    garbage that an optimizer would apply) and the check must raise"""
    g = load_golden("swarm50")
    net, prob = make_net(g, DEV).train(), make_prob(g, DEV, training=True)
    x = g.t("x")[:24].to(DEV)
    monkeypatch.setenv("NOCF_DUO", "1")
    monkeypatch.setenv("NOCF_DUO_SPIN_MAX", "1")
    Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], 4, "rk4", g.meta["alph"])
    try:
        Jc.backward()
    except RuntimeError as ex:                                       # (the forward's status may arrive before the backward starts)
        assert "timed out" in str(ex)
    else:
        torch.cuda.synchronize()
        assert all(torch.isnan(p.grad).all() for p in net.parameters())
        with pytest.raises(RuntimeError, match="timed out"):
            na.check_errors(sync=True)
    monkeypatch.delenv("NOCF_DUO_SPIN_MAX")
    try:
        na.check_errors(sync=True)
    except RuntimeError:
        pass


def test_probation_falls_back_to_the_tile_kernel_in_process():
    """a process whose FIRST split-role launches time out (a GPU shared with another rank or job) switches itself to the per-tile kernels
    and repeats the call -- a fresh launch, no re-exec -- instead of handing out NaN: run in a child (the switch is per process)"""
    import subprocess
    import sys
    code = (
        "import os, sys, torch\n"
        "sys.path.insert(0, os.environ['NOCF_REPO']); sys.path.insert(0, os.path.join(os.environ['NOCF_REPO'], 'tests'))\n"
        "import neuraloc_amd as na\n"
        "from neuraloc_amd import _lib\n"
        "from conftest import load_golden\n"
        "from util_hip import make_net, make_prob\n"
        "dev = torch.device('cuda:0'); g = load_golden('swarm50')\n"
        "net, prob = make_net(g, dev), make_prob(g, dev, training=False); x = g.t('x').to(dev)\n"
        "with torch.no_grad():\n"
        "    Jc, cs = na.OCflow(x, net, prob, [0.0, 1.0], 6, 'rk4', g.meta['alph'])\n"
        "    k = _lib.lib().nocf_last_rollout_kernel().decode()\n"
        "    Jt, _ = na.OCflow(x, net.train(), prob, [0.0, 1.0], 6, 'rk4', g.meta['alph'])\n"
        "na.check_errors(sync=True)\n"
        "assert os.environ.get('NOCF_DUO') == '1'      # the switch is a library override: the environment (what children inherit) is untouched\n"
        "print('FALLBACK-OK', k, float(Jc))\n"
        "assert torch.isfinite(Jc) and k.startswith('rollout_kernel')\n")
    env = dict(os.environ)
    env.update({"NOCF_REPO": REPO, "NOCF_DUO_PROBATION": "3", "NOCF_DUO_SPIN_MAX": "1", "NOCF_DUO": "1", "NOCF_JIT": "0"})
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FALLBACK-OK" in r.stdout, f"rc {r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}"
    assert "switching this process to the per-tile kernels" in r.stderr


def _record(x, net, prob, nt, alph):
    """nocf_rollout_record_f32 through the C ABI: (Jc sums, z_out, s_all)"""
    import ctypes as C
    n, d = x.shape
    phi_st, _k1, ws = net._c_struct(n)
    prob_st, _k2 = prob._c_struct(x.device)
    persample = torch.empty(n, 7, device=DEV)
    sums = torch.empty(8, device=DEV)
    z_out = torch.empty(n, d + 4, device=DEV)
    s_all = torch.zeros(nt * 4, n, d + 1, device=DEV)
    alph_c = (C.c_float * 6)(*[float(a) for a in alph[:6]])
    rc = _lib.lib().nocf_rollout_record_f32(C.byref(phi_st), C.byref(prob_st), _lib.ptr(x), n, 0.0, 1.0, nt, _lib.NOCF_RK4, alph_c,
                                            _lib.ptr(z_out), _lib.ptr(persample), _lib.ptr(sums), _lib.ptr(s_all),
                                            _lib.ptr(ws), ws.numel(), _lib.stream_ptr(DEV))
    assert rc == 0
    torch.cuda.synchronize()
    return sums.cpu(), z_out.cpu(), s_all.cpu(), persample.cpu()


@pytest.mark.parametrize("n", [70, 600, 2100])
def test_duo_recording_forward_matches_the_tile_kernel(n, monkeypatch):
    """training: the stage inputs the recording forward stores (what the adjoint re-evaluates at), the final states and the
    cost rows, against the per-tile kernel's; one, two and several tiles per group, more than one launch"""
    g = load_golden("swarm50")
    m = g.meta
    x = (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], 5)).contiguous().to(DEV)
    net = make_net(g, DEV)
    prob = make_prob(g, DEV, training=True)
    nt = 6
    res = {}
    for duo in ("1", "0"):
        monkeypatch.setenv("NOCF_DUO", duo)
        res[duo] = _record(x, net, prob, nt, m["alph"])
        assert (_kernel() == "rollout_duo_kernel") == (duo == "1")
    (_, zd, sd_, pd_), (_, zt, st_, pt_) = res["1"], res["0"]
    # In training mode at 6 steps the obstacle terms (1e7 per agent inside a block) amplify a rounding difference by ~3x per
    # stage (measured: 2e-6 at evaluation 1, 3e-4 at 7, 1.5e-2 at 23, between two correct fp32 kernels), so the first two steps are
    # compared tightly and the rest against that growth; the time entries are exact.
    assert torch.equal(sd_[0], st_[0])                                 # evaluation 0 is x itself (and t0)
    assert torch.equal(sd_[:, :, -1], st_[:, :, -1])                   # stage times
    assert (sd_[:8] - st_[:8]).abs().max().item() <= 2e-3
    drift = (sd_ - st_).abs().amax(dim=(0, 2))                         # per row, over all evaluations
    assert int((drift > 0.1).sum()) <= 2 + n // 128, f"{int((drift > 0.1).sum())} rows drift apart"
    assert float(drift.median()) <= 5e-3
    assert (sd_ != 0).any(dim=2).all(), "an evaluation's stage inputs were not recorded"


def test_duo_training_step_matches_the_tile_kernel(monkeypatch):
    """Jc.backward() with the split-role kernel as the recording forward: Jc and every parameter gradient against the tile kernel's
    (the adjoint kernel is the same; the two forwards differ by fp32 rounding, which the obstacle terms of the training mode -- 1e7 per
    agent inside a block -- amplify: 1e-2 of the largest gradient entry)"""
    g = load_golden("swarm50")
    m = g.meta
    x = (g.t("xInit") + m["var0"] * closed_form_normal(70, m["d"], 5)).contiguous().to(DEV)
    res = {}
    for duo in ("1", "0"):
        monkeypatch.setenv("NOCF_DUO", duo)
        net = make_net(g, DEV).train()
        prob = make_prob(g, DEV, training=True)
        Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], 6, "rk4", m["alph"])
        Jc.backward()
        res[duo] = (float(Jc.detach()), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu())
    assert abs(res["1"][0] - res["0"][0]) <= 2e-5 * abs(res["0"][0])
    assert (res["1"][1] - res["0"][1]).abs().max().item() <= 1e-2 * res["0"][1].abs().max().item()


@pytest.mark.parametrize("tag", ["eval_rk4", "eval_seg", "train_rk4", "eval_rk1"])
def test_duo_intermediates_against_reference_golden(tag, monkeypatch):
    """intermediates=True on the split-role kernel (one more evaluation per step: the controls at (z_{k+1}, t_k)): the reference's stored
    trajectories and controls, incl. slot 0 of the controls (zero) and the off-by-one control time (src/OCflow.py:51-55)"""
    g = load_golden("swarm50")
    if not g.has(tag + "/zFull"):
        pytest.skip("no such entry")
    monkeypatch.setenv("NOCF_DUO", "1")
    net = make_net(g, DEV)
    prob = make_prob(g, DEV, training=tag.startswith("train"))
    tspan = [float(v) for v in g[tag + "/tspan"]]
    nt = int(g[tag + "/nt"])
    stepper = "rk1" if tag.endswith("rk1") else "rk4"
    zw = torch.from_numpy(g[tag + "/zFull"])
    cw = torch.from_numpy(g[tag + "/ctrlFull"])
    x = g.t("x")[:zw.shape[0]].to(DEV)
    with torch.no_grad():
        zF, cF = na.OCflow(x, net, prob, tspan, nt, stepper, g.meta["alph"], intermediates=True)
    assert _kernel() == "rollout_duo_kernel"
    assert zF.shape == zw.shape and cF.shape == cw.shape
    bad, worst = count_off(zF.cpu(), zw, 1e-5, 1e-4)
    assert bad <= zw.numel() // 2000, f"{tag}: {bad} trajectory entries off (worst {worst:g})"
    bad, worst = count_off(cF.cpu(), cw, 1e-4, 1e-4 * float(cw.abs().max()) + 1e-5)
    assert bad <= cw.numel() // 2000, f"{tag}: {bad} control entries off (worst {worst:g})"
    assert float(cF[:, :, 0].abs().max()) == 0.0


@pytest.mark.parametrize("n", [5, 600, 2100])
def test_duo_intermediates_match_the_tile_kernel(n, monkeypatch):
    g = load_golden("swarm50")
    m = g.meta
    net, prob = make_net(g, DEV), make_prob(g, DEV, training=False)
    x = (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], 7)).contiguous().to(DEV)
    out = {}
    for duo in ("1", "0"):
        monkeypatch.setenv("NOCF_DUO", duo)
        with torch.no_grad():
            zF, cF = na.OCflow(x, net, prob, [0.0, 1.0], 8, "rk4", m["alph"], intermediates=True)
        assert (_kernel() == "rollout_duo_kernel") == (duo == "1")
        out[duo] = (zF.cpu(), cF.cpu())
    na.check_errors(sync=True)
    # at 8 steps some of these trajectories are unstable (a rounding difference grows 5x per step: 1e-5 after one step, 0.25 at the end, between
    # two correct fp32 kernels), so the first half of the time slices is compared tightly and the rest by count
    d = m["d"]
    (zd, cd), (zt, ct) = out["1"], out["0"]
    assert torch.equal(zd[:, :, 0], zt[:, :, 0]) and float(cd[:, :, 0].abs().max()) == 0.0
    assert (zd[:, :d, :5] - zt[:, :d, :5]).abs().max().item() <= 2e-3
    assert (cd[:, :, :5] - ct[:, :, :5]).abs().max().item() <= 2e-2 * ct.abs().max().item()
    drift = (zd[:, :d, :] - zt[:, :d, :]).abs().amax(dim=(1, 2))
    assert int((drift > 1e-2).sum()) <= 2 + n // 20, f"{int((drift > 1e-2).sum())} trajectories drift apart"
    calm = drift <= 1e-3                                               # on the calm trajectories the running cost L (column d) agrees too
    rel = (zd[calm, d, :] - zt[calm, d, :]).abs() / (1.0 + zt[calm, d, :].abs())
    assert float(rel.max()) <= 2e-3


@pytest.mark.parametrize("n,nt,stepper", [(39, 6, "rk4"), (1024, 4, "rk4"), (2049, 3, "rk4"), (100, 7, "rk1")])
def test_activation_record_gives_the_gradients_of_the_recomputing_adjoint(n, nt, stepper, monkeypatch):
    """training of the 512-wide network: the recording forward keeps u0, tanh(o), tanh(q), a and grad Phi of every evaluation (the
    activation record, nocf_rollout_record_act_f32) and the adjoint loads them instead of re-running grad Phi's forward sweep.  Same
    forward launch either way (Jc identical), gradients equal up to the rounding of the two forward sweeps; ragged tiles, two tiles
    per group and a second launch (row offsets of the record) included."""
    g = load_golden("swarm50")
    m = g.meta
    x = (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], 5)).contiguous().to(DEV)
    out = {}
    for rec in ("1", "0"):
        monkeypatch.setenv("NOCF_ACT_REC", rec)                        # "1": the tape + the split-role adjoint (nocf_duo_bwd.inc); "0": recompute, per tile
        net = make_net(g, DEV).train()
        prob = make_prob(g, DEV, training=True)
        xx = x.clone().requires_grad_(True)
        poison_allocator(DEV, big=2)
        Jc, _ = na.OCflow(xx, net, prob, [0.0, 1.0], nt, stepper, m["alph"])
        assert _kernel() == "rollout_duo_kernel"
        Jc.backward()
        torch.cuda.synchronize()
        na.check_errors(sync=True)
        assert _kernel() == ("rollout_duo_bwd_kernel" if rec == "1" else "rollout_bwd_kernel")
        out[rec] = (float(Jc.detach()), [p.grad.detach().clone() for p in net.parameters()], xx.grad.detach().clone())
    a, b = out["1"], out["0"]
    assert a[0] == b[0]
    for ga, gb in zip(a[1], b[1]):
        scale = float(gb.abs().max())
        assert torch.isfinite(ga).all() and float((ga - gb).abs().max()) <= 2e-4 * scale + 1e-12, (float((ga - gb).abs().max()), scale)
    assert float((a[2] - b[2]).abs().max()) <= 2e-4 * float(b[2].abs().max()) + 1e-12


@pytest.mark.parametrize("n,nt,stepper,training", [(16, 3, "rk4", True), (37, 2, "rk4", False), (530, 2, "rk4", True), (20, 5, "rk1", True), (300, 2, "rk4", True)])
def test_tape_adjoint_three_ways_with_all_cost_terms(n, nt, stepper, training, monkeypatch):
    """the split-role adjoint -- with the weight gradients contracted from the row streams (default) and accumulated in the kernel by the
    weight-gradient roles (NOCF_DUO_DW=1); with the column sums (dw, db1, db0) formed in the kernel's epilogues (default) and summed from the
    streamed rows afterwards (NOCF_DUO_CSUM=0) -- against BOTH per-tile adjoints (with the activation record and recomputing) on the pretrained swarm50 network
    with every multiplier switched on (the checkpoint trains with alph[3:6] = 0: HJt / HJfin / HJgrad exercise the sign masks, the
    terminal block and the value's rows) and the swarm squeezed so that agents interact and sit inside the obstacles' supports
    (the physics pass of role B' runs; unsqueezed most rows skip it).  n = 300 with the weight-gradient roles: 16 groups of which six have no rows
    (their clamped addresses point into another group's rows); the allocator's free blocks are filled with NaN in front of every variant."""
    g = load_golden("swarm50")
    m = g.meta
    alph = list(m["alph"])
    alph[3], alph[4], alph[5] = 2.0, 3.0, 1.5
    x = (0.3 * (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], 9))).contiguous().to(DEV)
    out = {}
    for tag, env in (("tape", {}), ("tape+dw", {"NOCF_DUO_DW": "1"}), ("tape, sums afterwards", {"NOCF_DUO_CSUM": "0"}), ("tape+dw, sums afterwards", {"NOCF_DUO_DW": "1", "NOCF_DUO_CSUM": "0"}),
                     ("tile+record", {"NOCF_DUO_BWD": "0"}), ("recompute", {"NOCF_ACT_REC": "0"})):
        monkeypatch.delenv("NOCF_DUO_BWD", raising=False)
        monkeypatch.delenv("NOCF_ACT_REC", raising=False)
        monkeypatch.delenv("NOCF_DUO_DW", raising=False)
        monkeypatch.delenv("NOCF_DUO_CSUM", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = make_net(g, DEV).train()
        prob = make_prob(g, DEV, training=training)
        xx = x.clone().requires_grad_(True)
        poison_allocator(DEV)                                      # (a row read before it is written must not find the previous variant's values)
        Jc, _ = na.OCflow(xx, net, prob, [0.0, 1.0], nt, stepper, alph)
        Jc.backward()
        torch.cuda.synchronize()
        na.check_errors(sync=True)
        assert _kernel() == ("rollout_duo_bwd_kernel" if tag.startswith("tape") else "rollout_bwd_kernel")
        out[tag] = (float(Jc.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters()}, xx.grad.detach().clone())
    ref = out["recompute"]
    for tag in ("tape", "tape+dw", "tape, sums afterwards", "tape+dw, sums afterwards", "tile+record"):
        got = out[tag]
        assert got[0] == ref[0]
        for k in ref[1]:
            scale = float(ref[1][k].abs().max())
            err = float((got[1][k] - ref[1][k]).abs().max())
            assert torch.isfinite(got[1][k]).all() and err <= 2e-4 * scale + 1e-12, f"{tag} {k}: {err:g} at scale {scale:g}"
        assert float((got[2] - ref[2]).abs().max()) <= 2e-4 * float(ref[2].abs().max()) + 1e-12, tag


@pytest.mark.parametrize("name,n,stepper,training", [("midcross20", 9, "rk4", True), ("swarm", 6, "rk4", True), ("swap12", 7, "rk1", True),
                                                      ("midcross30", 5, "rk4", False), ("softcorridor", 11, "rk4", True), ("swap2", 4, "rk4", True)])
@pytest.mark.parametrize("width", [512, 256])
def test_tape_adjoint_on_other_problems_against_oracle_fp64_autograd(name, n, stepper, training, width, form):
    """m = 512 networks on other Cross2D / SwarmTraj problems (2 ... 30 agents, obstacles of both kinds, pair and many-agent interaction
    forms, d + 1 from 5 to 61): forward tape + split-role adjoint vs the oracle differentiated by torch autograd in float64.  m = 256: the
    recording forward is the split-role kernel with four members per group (it writes the activation record), the adjoint the per-tile one
    loading that record (`trainOC.py --m 256`)."""
    if width == 256 and form == "g8":
        pytest.skip("the 256-wide network has one geometry")
    alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
    torch.manual_seed(11)
    prob, x0, _, _ = na.initProb(name, 24, 24, 0.5, alph, lambda t: t.float().to(DEV))
    prob.train() if training else prob.eval()
    x0 = x0[:n].contiguous()
    d = x0.shape[1]
    sd = synth_state_dict(2, width, d, seed=len(name))
    net = na.Phi(nTh=2, m=width, d=d, alph=alph)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    nt = 4
    Jc, _ = na.OCflow(x0, net, prob, [0.0, 1.0], nt, stepper, alph)
    assert _kernel() == "rollout_duo_kernel"
    Jc.backward()
    torch.cuda.synchronize()
    na.check_errors(sync=True)
    assert _kernel() == ("rollout_duo_bwd_kernel" if width == 512 else "rollout_bwd_kernel")
    P = orc.PhiParams.from_state_dict({k: v.clone() for k, v in sd.items()}, dtype=torch.float64)
    for t in [*P.K, *P.b, P.w, P.A, P.cw, P.cb]:
        t.requires_grad_(True)
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu()
    J, _ = orc.rollout(x0.double().cpu(), P, S.to(torch.float64), [0.0, 1.0], nt, stepper, alph)
    J.backward()
    want = {"A": P.A.grad, "c.weight": P.cw.grad, "c.bias": P.cb.grad, "w.weight": P.w.grad,
            "N.layers.0.weight": P.K[0].grad, "N.layers.0.bias": P.b[0].grad, "N.layers.1.weight": P.K[1].grad, "N.layers.1.bias": P.b[1].grad}
    assert abs(Jc.detach().item() - float(J)) <= 2e-5 * abs(float(J))
    for k, p in net.named_parameters():
        w = want[k] if want[k] is not None else torch.zeros_like(p, dtype=torch.float64).cpu()
        scale = w.abs().max().item()
        err = (p.grad.cpu().double() - w.reshape(p.shape)).abs().max().item()
        assert err <= 2e-4 * scale + 1e-6, f"{name} {k}: err {err:g} at scale {scale:g}"


@pytest.mark.parametrize("training", [False, True])
def test_rows_that_differ_between_two_kernels_are_sensitive_in_the_fp64_oracle(training, monkeypatch, capsys):
    """The allowance of the sweep above ("a few swarm50 states are chaotic at nt = 10") demonstrated instead of asserted: on 4096 rows the
    rows on which the split-role kernel and the per-tile kernel disagree (beyond rel 1e-3 + abs 1e-3) are taken to the ORACLE IN FLOAT64
    and integrated from x and from x (1 +- 3e-6): it must move there by more than the same threshold (_sensitive_in_fp64) -- i.e. the row
    amplifies the distance between two fp32 evaluations past the tolerance in exact arithmetic too, so two correct fp32 evaluations (different
    summation orders, 6e-8 per operation) cannot be expected to agree on it -- or, failing that, both kernels must be within the tolerance of
    the fp64 truth (_explained); a control group of rows on which the kernels agree is mostly NOT sensitive.  The number of such rows is
    printed and bounded like the sweep's (n / 128); what is ASSERTED about each of them is why it may differ."""
    n, nt = 4096, 10
    g = load_golden("swarm50")
    net, prob = make_net(g, DEV), make_prob(g, DEV, training=training)
    m = g.meta
    x = (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], 3)).contiguous()
    monkeypatch.setenv("NOCF_DUO", "1")
    duo = _table(x.to(DEV), net, prob, [0.0, 1.0], nt, "rk4", m["alph"])
    assert _kernel() == "rollout_duo_kernel"
    monkeypatch.setenv("NOCF_DUO", "0")
    tile = _table(x.to(DEV), net, prob, [0.0, 1.0], nt, "rk4", m["alph"])
    off = ((duo.double() - tile.double()).abs() > 1e-3 + 1e-3 * tile.double().abs()).any(dim=1).cpu()
    rows = torch.nonzero(off).flatten()
    P, S = make_oracle(g, training)
    P64 = orc.PhiParams.from_state_dict(g.state_dict(), dtype=torch.float64)
    S64 = S.to(torch.float64)

    with capsys.disabled():
        print(f"\n[chaotic rows] training={training}: {len(rows)} of {n} rows differ between the split-role and the per-tile kernel: {rows.tolist()}")
    assert len(rows) <= n // 128, f"{len(rows)} rows differ"        # (the sweep's bound; measured: 8 with the default geometry, 13 with the fine one)
    frac_sens = 1.0
    if len(rows):
        ok, sens, amp = _explained(x[rows], duo[rows], tile[rows], P64, S64, nt, m["alph"])
        frac_sens = float(sens.double().mean())
        with capsys.disabled():
            print("[chaotic rows]   fp64 oracle, x vs x(1+-3e-6): change / tolerance per differing row:", [f"{float(v):.1f}" for v in amp],
                  "; not sensitive but both kernels within the fp64 tolerance:", rows[ok & ~sens].tolist())
        assert bool(ok.all()), f"rows {rows[~ok].tolist()} differ between the kernels and are neither sensitive in the fp64 oracle nor both within its tolerance"
    ctrl = torch.nonzero(~off).flatten()[:: max(1, (n - len(rows)) // 64)][:64]
    mv, amp = _sensitive_in_fp64(x[ctrl], P64, S64, nt, m["alph"])
    with capsys.disabled():
        print(f"[chaotic rows]   control ({len(ctrl)} agreeing rows): {int(mv.sum())} sensitive, median change / tolerance {float(amp.median()):.3f}")
    # the differing rows ARE the sensitive ones: most of them, against a small minority of the rows the kernels agree on (eval-mode masks
    # flip under 3e-6 on a few percent of all rows)
    assert frac_sens >= 0.75 and int(mv.sum()) <= len(ctrl) // 4
