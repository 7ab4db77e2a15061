"""GPU: the trainOC-style driver end to end -- a short training run through the HIP
forward + adjoint lowers the validation loss, writes a reference-layout checkpoint, and that file loads back."""
import glob
import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pytestmark = pytest.mark.gpu


def test_short_training_run_and_checkpoint_roundtrip(tmp_path, capsys):
    import trainOC
    from neuraloc_amd.checkpoint import load_checkpoint
    import neuraloc_amd as na
    best = trainOC.main(["--data", "softcorridor", "--niters", "60", "--val_freq", "10", "--n_train", "256", "--nt", "12",
                         "--m", "16", "--save", str(tmp_path), "--seed", "1", "--lr", "0.02", "--sample_freq", "25"])
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln[:5].isdigit()]
    assert len(lines) == 60
    first, last = float(lines[0].split()[3]), float(lines[-1].split()[3])
    assert last < 0.5 * first, (first, last)
    files = glob.glob(os.path.join(str(tmp_path), "*_checkpt.pth"))
    assert len(files) == 1
    net, prob, x0, x0v, xInit, a = load_checkpoint(files[0], device="cuda:0", n_train=64, n_val=64)
    assert a.data == "softcorridor" and a.m == 16 and net.m == 16
    prob.eval()
    with torch.no_grad():
        Jc, _ = na.OCflow(xInit, net.eval(), prob, [0.0, 1.0], 12, "rk4", net.alph)
    assert torch.isfinite(Jc) and best < float("inf")


def test_short_training_run_in_double_precision(tmp_path, capsys):
    """trainOC.py --prec double (trainOC.py:44,76-79): rollout, adjoint and Adam in float64"""
    import trainOC
    trainOC.main(["--data", "softcorridor", "--niters", "30", "--val_freq", "10", "--n_train", "128", "--nt", "8",
                  "--m", "16", "--save", str(tmp_path), "--seed", "1", "--lr", "0.02", "--sample_freq", "100", "--prec", "double"])
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln[:5].isdigit()]
    assert len(lines) == 30
    first, last = float(lines[0].split()[3]), float(lines[-1].split()[3])
    assert last < 0.7 * first, (first, last)


def test_evalOC_on_a_reference_checkpoint(tmp_path, capsys):
    """evalOC.py with the reference's flag set on a checkpoint in the reference's layout (weights = the pretrained softcorridor
    network of tests/golden/softcorridor.npz): the printed costs on xInit are the reference's known answers (SURVEY 8(c),
    fixture xinit_eval/*), the deployment line has timeOC.py's format, the trajectory file and the shocked rollouts exist"""
    import argparse
    import re
    import numpy as np
    import evalOC
    import neuraloc_amd as na
    from neuraloc_amd.checkpoint import save_checkpoint
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from conftest import load_golden
    g = load_golden("softcorridor")
    m = g.meta
    net = na.Phi(nTh=m["nTh"], m=m["m"], d=m["d"], alph=m["alph"])
    net.load_state_dict(g.state_dict())
    ck = os.path.join(str(tmp_path), "softcorridor_nn_checkpt.pth")
    save_checkpoint(ck, net, argparse.Namespace(data="softcorridor", m=m["m"], nTh=m["nTh"], alph=m["alph"], n_train=64, var0=1.0))
    save = os.path.join(str(tmp_path), "eval")
    out = evalOC.main(["--resume", ck, "--nt", str(int(g["xinit_eval/nt"])), "--save", save, "--do_shock", "--batch", "64",
                       "--alph", "1.0, 1.0, 1.0, 1.0, 1.0, 1.0", "--approach", "ocflow", "--prec", "single"])
    text = capsys.readouterr().out
    want_J, want_cs = float(g["xinit_eval/Jc"]), g["xinit_eval/cs"].astype(float)
    assert abs(out["Jc"] - want_J) <= 1e-4 * abs(want_J)
    for got, want in zip(out["cs"], want_cs):
        assert abs(got - want) <= 1e-4 * abs(want) + 1e-6
    # the reference's two log lines (evalOC.py:76-85): header, then L+G, L, a0 G, a3 HJt, a4 HJfin, a5 HJgrad, Q, W
    assert "just xInit" in text and "HJgrad" in text
    a = m["alph"]
    row = "         {:12.4e} {:11.3e} {:11.3e} {:11.3e} {:11.3e} {:11.3e} {:11.3e} {:11.3e}".format(
        want_cs[0] + a[0] * want_cs[1], want_cs[0], a[0] * want_cs[1], a[3] * want_cs[2], a[4] * want_cs[3], a[5] * want_cs[4], want_cs[5], want_cs[6])
    got_row = [ln for ln in text.splitlines() if re.match(r"^\s+[-0-9.]+e[-+]\d+\s", ln)][0]
    for gv, wv in zip(got_row.split(), row.split()):
        assert abs(float(gv) - float(wv)) <= 2e-3 * abs(float(wv)) + 1e-9, (got_row, row)
    assert re.search(r"^time: +[0-9.]+ +avg time / RK4 timestep: +[0-9.]+$", text, re.M), "timeOC.py:80 line format"
    assert os.path.exists(os.path.join(save, "deploy_times"))
    z = np.load(os.path.join(save, "figs", "eval_softcorridor_nn.npz"))        # the reference names its outputs basename[:-12]
    assert z["zFull"].shape == (1, m["d"] + 4, int(g["xinit_eval/nt"]) + 1) and z["ctrlFull"].shape[1] == m["d"]
    assert os.path.exists(os.path.join(save, "figs", "eval_softcorridor_nn_shock.npz"))
    assert "majorshock at t=0.1" in text
    # --prec double (evalOC.py:19,28-31): the same evaluation through the double-precision rollout, incl. the shocked rollouts
    save64 = os.path.join(str(tmp_path), "eval64")
    out64 = evalOC.main(["--resume", ck, "--nt", str(int(g["xinit_eval/nt"])), "--save", save64, "--do_shock", "--batch", "16",
                         "--prec", "double"])
    assert abs(out64["Jc"] - want_J) <= 1e-4 * abs(want_J)               # the fp32 known answer, to fp32 accuracy
    for got, want in zip(out64["cs"], want_cs):
        assert abs(got - want) <= 1e-4 * abs(want) + 1e-6
    z64 = np.load(os.path.join(save64, "figs", "eval_softcorridor_nn.npz"))
    assert z64["zFull"].dtype == np.float64 and z64["zFull"].shape == z["zFull"].shape
    assert float(np.abs(z64["zFull"][:, :m["d"]] - z["zFull"][:, :m["d"]]).max()) <= 1e-3
    assert os.path.exists(os.path.join(save64, "figs", "eval_softcorridor_nn_majorshock.npz"))


def test_training_path_validates_its_inputs_and_parameter_versions():
    """the training entry checks x against the network like the evaluation entry does, and a backward after
    the parameters changed raises instead of silently using the new weights"""
    import neuraloc_amd as na
    dev = torch.device("cuda:0")
    alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
    torch.manual_seed(0)
    prob, x0, _, _ = na.initProb("softcorridor", 16, 16, 0.5, alph, lambda t: t.float().to(dev))
    net = na.Phi(2, 16, 4, alph=alph).to(dev)
    with pytest.raises(ValueError, match="d="):
        na.OCflow(torch.zeros(16, 6, device=dev), net, prob, [0.0, 1.0], 4, "rk4", alph)
    with pytest.raises(ValueError, match="alph"):
        na.OCflow(x0, net, prob, [0.0, 1.0], 4, "rk4", alph[:3])
    Jc, _ = na.OCflow(x0, net, prob, [0.0, 1.0], 4, "rk4", alph)
    with torch.no_grad():
        net.w.weight.add_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        Jc.backward()
