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
