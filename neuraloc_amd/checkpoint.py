"""Checkpoint I/O compatible with the reference.

The reference saves `{'args': argparse.Namespace, 'state_dict': net.state_dict()}` whenever validation improves
(trainOC.py:199-207) and rebuilds Phi from `args.m`, `args.nTh`, `args.alph`, `args.data` (evalOC.py:51-64).
State-dict keys are identical here, so its .pth files load unchanged and files written here load there."""
import argparse

import torch

from .Phi import Phi
from .initProb import initProb


def load_file(path, trusted=False):
    """the {'args', 'state_dict'} dict of a checkpoint file.  The reference's files pickle an argparse.Namespace next to the
    tensors; that one class is allow-listed for the weights-only unpickler, so loading a checkpoint cannot run code.
    trusted=True (or NOCF_TRUST_CHECKPOINTS=1) falls back to the full unpickler for files that hold other objects."""
    import os
    import pickle
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as exc:
        # only the "unsupported global" refusal of the weights-only unpickler lands here: a missing or unreadable file (OSError),
        # a truncated archive (RuntimeError / EOFError) ... propagate as they are and never reach the full unpickler
        if trusted or os.environ.get("NOCF_TRUST_CHECKPOINTS", "0") not in ("", "0"):
            return torch.load(path, map_location="cpu", weights_only=False)
        raise RuntimeError(f"{path}: the weights-only unpickler refused an object in this file ({exc}); if you trust the file -- the full "
                           "unpickler can run code -- pass trusted=True or set NOCF_TRUST_CHECKPOINTS=1") from exc


def load_checkpoint(path, device="cuda:0", n_train=None, n_val=None, var0=None, dtype=torch.float32):
    """-> (net, prob, x0, x0v, xInit, args); `args` is the pickled Namespace (see load_file).  dtype=torch.float64 is the
    reference's --prec double (evalOC.py:28-31: `cvt` and net.to(prec))"""
    ck = load_file(path)
    a = ck["args"]
    alph = [float(v) for v in a.alph]
    dev = torch.device(device)
    cvt = lambda t: t.to(dtype).to(dev)                   # noqa: E731
    prob, x0, x0v, xInit = initProb(a.data, n_train or getattr(a, "n_train", 1024), n_val or getattr(a, "n_train", 1024),
                                    var0=var0 if var0 is not None else getattr(a, "var0", 1.0), alph=alph, cvt=cvt)
    net = Phi(nTh=a.nTh, m=a.m, d=x0.shape[1], alph=alph)
    net.load_state_dict(ck["state_dict"])
    return net.to(dtype).to(dev), prob, x0, x0v, xInit, a


def save_checkpoint(path, net, args):
    """same dict layout as trainOC.py:199-207; `args` may be a Namespace or a dict with data/m/nTh/alph"""
    if isinstance(args, dict):
        args = argparse.Namespace(**args)
    torch.save({"args": args, "state_dict": {k: v.detach().cpu() for k, v in net.state_dict().items()}}, path)
