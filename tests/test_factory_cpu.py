"""CPU: the problem/sample factory against the reference's own initProb (src/initProb.py:25-249), pinned by
tests/golden/factory.npz (made by tests/golden/make_golden_factory.py from the reference import): every problem name,
xtarget, xInit, r, obstacle, alph_Q / alph_W, nAgents, the problem class and a seeded x0 / x0v (same RNG draws in the same order)."""
import json
import os

import numpy as np
import pytest
import torch

import neuraloc_amd as na
from neuraloc_amd.initProb import PROBLEM_NAMES, initProb, resample

HERE = os.path.dirname(os.path.abspath(__file__))
Z = np.load(os.path.join(HERE, "golden", "factory.npz"))
META = json.loads(str(Z["meta"]))


def test_same_problem_names_as_the_reference():
    assert sorted(PROBLEM_NAMES) == sorted(META["problems"])


@pytest.mark.parametrize("name", sorted(META["problems"]))
def test_factory_matches_reference(name):
    m = META["problems"][name]
    torch.manual_seed(META["seed"])
    prob, x0, x0v, xInit = initProb(name, META["n_train"], META["n_val"], var0=META["var0"], alph=META["alph"], cvt=lambda t: t.float())
    assert type(prob).__name__ == m["cls"]
    assert prob.obstacle == m["obstacle"]
    assert float(getattr(prob, "r", 0.0)) == m["r"]
    assert float(prob.alph_Q) == m["alph_Q"] and float(prob.alph_W) == m["alph_W"]
    assert int(prob.nAgents) == m["nAgents"] and x0.shape[1] == m["d"]
    assert np.array_equal(prob.xtarget.reshape(-1).numpy(), Z[f"{name}/xtarget"].reshape(-1))
    assert np.array_equal(xInit.numpy(), Z[f"{name}/xInit"])
    assert np.array_equal(x0.numpy(), Z[f"{name}/x0"]), "seeded x0 differs: RNG draws are not in the reference's order"
    assert np.array_equal(x0v.numpy(), Z[f"{name}/x0v"])


def test_unknown_name_raises():
    with pytest.raises(ValueError):
        initProb("nosuchproblem", 4, 4, 1.0, [1.0] * 6, lambda t: t.float())


def test_resample_draws_around_xinit():
    torch.manual_seed(3)
    _, x0, _, xInit = initProb("swap12", 8, 8, 0.5, META["alph"], lambda t: t.float())
    torch.manual_seed(4)
    x1 = resample(x0, xInit, 0.5, lambda t: t.float())
    torch.manual_seed(4)
    want = xInit + 0.5 * torch.randn(*x0.shape)
    assert torch.equal(x1, want)


class Other:                                              # an arbitrary pickled class: not allow-listed
    pass


def test_checkpoint_files_load_without_running_code(tmp_path):
    """the {'args': Namespace, 'state_dict'} layout goes through the weights-only unpickler (argparse.Namespace allow-listed);
    a file holding any other object is refused unless the caller says it is trusted"""
    import argparse
    from neuraloc_amd.checkpoint import load_file, save_checkpoint
    net = na.Phi(2, 8, 4)
    good = tmp_path / "good_checkpt.pth"
    save_checkpoint(str(good), net, argparse.Namespace(data="swap2", m=8, nTh=2, alph=[1.0] * 6))
    ck = load_file(str(good))
    assert ck["args"].data == "swap2" and list(ck["state_dict"]) == list(net.state_dict())

    bad = tmp_path / "bad_checkpt.pth"
    torch.save({"args": Other(), "state_dict": {}}, str(bad))
    with pytest.raises(RuntimeError, match="weights-only"):
        load_file(str(bad))
