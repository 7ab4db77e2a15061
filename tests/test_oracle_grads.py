"""CPU: the oracle's autograd reproduces the reference's parameter gradients (tests/golden/grads.npz, produced by
tests/golden/make_golden_grads.py from `Jc.backward()` through the reference's OCflow, trainOC.py:172-173).
This pins the CHECKER for the next scope row (the hand-written backward of the rollout)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, PRETRAINED, load_golden
from oracle import ocflow_oracle as orc
from util_oracle import spec_of

G = np.load(os.path.join(GOLDEN_DIR, "grads.npz"))


@pytest.mark.parametrize("name", PRETRAINED)
def test_oracle_autograd_matches_reference_gradients(name):
    g = load_golden(name)
    nt, ns = int(G[f"{name}/nt"]), int(G[f"{name}/ns"])
    P = orc.PhiParams.from_state_dict({k: v.clone() for k, v in g.state_dict().items()})
    leaves = [*P.K, *P.b, P.w, P.A, P.cw, P.cb]
    for t in leaves:
        t.requires_grad_(True)
    S = spec_of(g, training=True)
    Jc, _ = orc.rollout(g.t("x")[:ns], P, S, [0.0, 1.0], nt, "rk4", g.meta["alph"])
    Jc.backward()
    assert abs(float(Jc) - float(G[f"{name}/Jc"])) <= 1e-5 * abs(float(G[f"{name}/Jc"]))
    got = {"A": P.A.grad, "c.weight": P.cw.grad, "c.bias": P.cb.grad, "w.weight": P.w.grad}
    for i in range(g.meta["nTh"]):
        got[f"N.layers.{i}.weight"] = P.K[i].grad
        got[f"N.layers.{i}.bias"] = P.b[i].grad
    for k, v in got.items():
        want = torch.from_numpy(G[f"{name}/grad/{k}"])
        v = v if v is not None else torch.zeros_like(want)
        rel = (v - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
        assert rel <= 1e-4, f"{name} {k}: rel {rel:g}"
