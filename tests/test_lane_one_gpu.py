"""Small configurations, one launch per call (round 6): the lane kernel's LAST workgroup forms the column sums, the means and Jc itself
(nocf_lane.inc: agent-scope fences + a self-resetting ticket word per stream) instead of a second launch of cost_sum_kernel.  Same arithmetic
in the same order, so the results must be BITWISE those of the two-launch path (NOCF_LANE_ONE=0) -- whatever workgroup finishes last, for any
grid size, call after call (the ticket wraps), and on two streams at once (each stream has its own ticket word)."""
import os

import pytest
import torch

import neuraloc_amd as na
from neuraloc_amd import _lib
from conftest import load_golden
from util_hip import closed_form_normal, make_net, make_oracle, make_prob
from oracle import ocflow_oracle as orc

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _states(g, n, seed):
    m = g.meta
    return (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], seed)).contiguous().to(DEV)


def _call(x, net, prob, m, one, **kw):
    os.environ["NOCF_LANE_ONE"] = "1" if one else "0"
    try:
        with torch.no_grad():
            out = na.OCflow(x, net, prob, [0.0, 1.0], m["nt"], "rk4", m["alph"], **kw)
        torch.cuda.synchronize()
        assert _lib.lib().nocf_last_rollout_kernel().decode() == "rollout_lane_kernel"
        return out
    finally:
        os.environ.pop("NOCF_LANE_ONE", None)


@pytest.mark.parametrize("name", ["swap2", "softcorridor", "swap12"])
@pytest.mark.parametrize("training", [False, True])
def test_one_launch_equals_two_launches_bitwise(name, training):
    g = load_golden(name)
    m = g.meta
    net, prob = make_net(g, DEV), make_prob(g, DEV, training)
    for n in (1, 3, 4, 5, 39, 256, 1021, 2048):                 # one workgroup, ragged tails, many workgroups
        x = _states(g, n, 7 + n)
        J2, c2 = _call(x, net, prob, m, one=False)
        for rep in range(3):                                    # the ticket word wraps to zero behind every launch
            J1, c1 = _call(x, net, prob, m, one=True)
            assert torch.equal(J1, J2), (name, n, rep, float(J1), float(J2))
            for a, b in zip(c1, c2):
                assert torch.equal(a, b)
    # ... and against the oracle at one size (the means are what a driver prints)
    x = _states(g, 64, 3)
    J1, c1 = _call(x, net, prob, m, one=True)
    P, S = make_oracle(g, training)
    Jo, co = orc.rollout(x.cpu(), P, S, [0.0, 1.0], m["nt"], "rk4", m["alph"])
    assert abs(float(J1) - float(Jo)) <= 1e-4 * abs(float(Jo))


def test_one_launch_nomean_and_intermediates_unchanged():
    g = load_golden("softcorridor")
    m = g.meta
    net, prob = make_net(g, DEV), make_prob(g, DEV, False)
    x = _states(g, 130, 11)
    Ja, ca = _call(x, net, prob, m, one=True, noMean=True)
    Jb, cb = _call(x, net, prob, m, one=False, noMean=True)
    assert torch.equal(Ja, Jb) and all(torch.equal(a, b) for a, b in zip(ca, cb))
    za, ua = _call(x, net, prob, m, one=True, intermediates=True)
    zb, ub = _call(x, net, prob, m, one=False, intermediates=True)
    assert torch.equal(za, zb) and torch.equal(ua, ub)


def test_two_streams_at_once_have_their_own_ticket_words():
    g = load_golden("swap12")
    m = g.meta
    net, prob = make_net(g, DEV), make_prob(g, DEV, False)
    xs = [_states(g, n, 20 + n) for n in (777, 2048)]
    want = [_call(x, net, prob, m, one=False)[0] for x in xs]
    streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    os.environ["NOCF_LANE_ONE"] = "1"
    try:
        got = [[], []]
        with torch.no_grad():
            for rep in range(40):
                for i, st in enumerate(streams):
                    with torch.cuda.stream(st):
                        got[i].append(na.OCflow(xs[i], net, prob, [0.0, 1.0], m["nt"], "rk4", m["alph"])[0])
        torch.cuda.synchronize()
    finally:
        os.environ.pop("NOCF_LANE_ONE", None)
    for i in range(2):
        for J in got[i]:
            assert torch.equal(J, want[i])
