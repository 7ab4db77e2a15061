"""Problem / sample factory with the reference's `initProb(sData, nTrain, nVal, var0, alph, cvt)`
signature and return value (prob, x0, x0v, xInit)  (src/initProb.py:9-249).

The 18 named problems are restated as DATA (targets, start centres, radii, obstacles); the
random draws are made in the same order and shapes as upstream so a seeded caller sees the
same batches."""
import torch
from torch.nn.functional import pad

from .problem import Cross2D, Quadcopter, SwarmTraj

# ---- Cross2D family: name -> (xtarget, xInit, obstacle, r or None for the class default, which batch size x0v uses)
_SWAP12_T = [2, 2, 0, 0, 10, 0, -10, 0, 5, 5, -5, -5, -4, 2, -6, -1, 5, -5, -5, 5, 2, -2, -2, -2]
_SWAP12_I = [0, 0, 2, 2, -10, 0, 10, 0, -5, -5, 5, 5, -6, -1, -4, 2, -5, 5, 5, -5, -2, -2, 2, -2]

_CROSS = {
    "softcorridor": dict(tgt=[2, 2, -2, 2], ini=[-2, -2, 2, -2], obstacle="softcorridor", r=0.5, val="train"),
    "midcross2":    dict(tgt=[2, 2, -2, 2], ini=[-2, -2, 2, -2], obstacle=None, r=None, val="train"),
    "swap2":        dict(tgt=[10., 0., -10., 0.], ini=[-10., 0., 10., 0.], obstacle="hardcorridor", r=1.0, val="val"),
    "swap12":       dict(tgt=_SWAP12_T, ini=_SWAP12_I, obstacle=None, r=0.5, val="val"),
}
for _k, _pairs in (("swap12_5pair", 5), ("swap12_4pair", 4), ("swap12_3pair", 3), ("swap12_2pair", 2), ("swap12_1pair", 1)):
    _CROSS[_k] = dict(tgt=_SWAP12_T[:4 * _pairs], ini=_SWAP12_I[:4 * _pairs], obstacle=None, r=0.5, val="val")

# midcross4/20/30: agents on a line, targets mirrored (src/initProb.py:155-189)
_MIDLINE = {"midcross4": (4, 2.0, 0.4), "midcross20": (20, 6.0, 0.15), "midcross30": (30, 6.0, 0.2)}

# ---- SwarmTraj family: first half of the formation; the second half is shifted by (0,-0.5,-3)
_SWARM32 = [-2., 2., 8., -1., 2., 8., 0., 2., 8., 1., 2., 8., 2., 2., 8.,
            -2.5, 3., 8., -1.5, 3., 8., -0.5, 3., 8., 0.5, 3., 8., 1.5, 3., 8., 2.5, 3., 8.,
            -2., 4., 8., -1., 4., 8., 0., 4., 8., 1., 4., 8., 2., 4., 8.]
_SWARM50 = [-2., 2., 6., -1., 2., 6., 0., 2., 6., 1., 2., 6., 2., 2., 6., 3., 2., 6., 4., 2., 6.,
            -2.5, 3., 7., -1.5, 3., 7., -0.5, 3., 7., 0.5, 3., 7., 1.5, 3., 7., 2.5, 3., 7., 3.5, 3., 7.,
            -2., 4., 8., -1., 4., 8., 0., 4., 8., 1., 4., 8., 2., 4., 8., 3., 4., 8., 4., 4., 8.,
            -2., 3., 5., -1., 3., 5., 1., 3., 5., 2., 3., 5.]
_SWARM = {"swarm": (_SWARM32, 0.2), "swarm50": (_SWARM50, 0.1)}

PROBLEM_NAMES = sorted(list(_CROSS) + list(_MIDLINE) + list(_SWARM) + ["singlequad"])


def initProb(sData, nTrain, nVal, var0, alph, cvt):
    """
    :param sData:  name of the problem (one of PROBLEM_NAMES)
    :param nTrain: batch size drawn from rho_0
    :param nVal:   validation batch size
    :param var0:   scale of rho_0
    :param alph:   6 multipliers; alph[1], alph[2] go into the problem object
    :param cvt:    dtype/device conversion, e.g. lambda t: t.float().to('cuda')
    :return: prob, x0 [nTrain,d], x0v, xInit [1,d]
    """
    if sData in _CROSS:
        c = _CROSS[sData]
        d = len(c["tgt"])
        xtarget = cvt(torch.tensor(c["tgt"]))
        xInit = cvt(torch.tensor(c["ini"])).reshape(1, -1)
        x0 = xInit + cvt(var0 * torch.randn(nTrain, d))
        x0v = xInit + cvt(var0 * torch.randn(nTrain if c["val"] == "train" else nVal, d))
        kw = {} if c["r"] is None else {"r": c["r"]}
        if sData == "softcorridor":
            xtarget = xtarget.reshape(1, -1)
        prob = Cross2D(xtarget, obstacle=c["obstacle"], alph_Q=alph[1], alph_W=alph[2], **kw)
    elif sData in _MIDLINE:
        nAgents, span, r = _MIDLINE[sData]
        d = 2 * nAgents
        xx = torch.linspace(-span, span, nAgents)
        if sData == "midcross30":
            rows_t = torch.tensor([6, 4, 2]).view(-1, 1).repeat(nAgents // 3, 1).view(-1)
            rows_i = torch.tensor([-6, -4, -2]).view(-1, 1).repeat(nAgents // 3, 1).view(-1)
        else:
            rows_t = span * torch.ones(nAgents)
            rows_i = -span * torch.ones(nAgents)
        xtarget = cvt(torch.stack((xx.flip(dims=[0]), rows_t), dim=1).reshape(1, -1))
        xInit = cvt(torch.stack((xx, rows_i), dim=1).reshape(1, -1))
        x0 = xInit + cvt(var0 * torch.randn(nTrain, d))
        x0v = xInit + cvt(var0 * torch.randn(nVal, d))
        prob = Cross2D(xtarget, obstacle=None, alph_Q=alph[1], alph_W=alph[2], r=r)
    elif sData in _SWARM:
        half, r = _SWARM[sData]
        first = cvt(torch.tensor(half)).view(-1, 3)
        xtarget = torch.cat((first, cvt(torch.tensor([0, -0.5, -3])) + first), dim=0).view(-1)
        d = xtarget.numel()
        halfTrain = nTrain // 2
        xInit = (cvt(torch.tensor([1, -1, -1])) * xtarget.view(-1, 3) + cvt(torch.tensor([0, 0, 10]))).view(1, -1)
        # half the batch starts around xInit, half around the target (src/initProb.py:113-117);
        # the reference's "zero the velocities" loop slices an empty range and changes nothing
        x0 = torch.cat((xInit + cvt(var0 * torch.randn(halfTrain, d)),
                        xtarget + cvt(var0 * torch.randn(halfTrain, d))), dim=0)
        x0v = xInit + cvt(var0 * torch.randn(halfTrain, d))
        prob = SwarmTraj(xtarget, obstacle="blocks", alph_Q=alph[1], alph_W=alph[2], r=r)
    elif sData == "singlequad":
        d = 12
        xtarget = cvt(torch.tensor([2., 2., 2., 0., 0., 0., 0., 0., 0., 0., 0., 0.]))
        centre = cvt(torch.tensor([-1.5, -1.5, -1.5]))
        x0 = pad(centre + cvt(var0 * torch.randn(nTrain, 3)), [0, d - 3, 0, 0], value=0)
        xInit = pad(centre.view(1, -1), [0, d - 3, 0, 0], value=0)
        x0v = pad(cvt(torch.tensor([-1.5, -1.5, -1.5]) + var0 * torch.randn(nVal, 3)), [0, d - 3, 0, 0], value=0)
        prob = Quadcopter(xtarget, obstacle=None, alph_Q=0.0, alph_W=0.0)
    else:
        # the reference prints and exit(1)s (src/initProb.py:244-246)
        raise ValueError(f"incorrect value passed to --data: {sData!r}; known: {PROBLEM_NAMES}")
    return prob, x0, x0v, xInit


def resample(x0, xInit, var0, cvt):
    """fresh training batch around xInit (src/initProb.py:252-262)"""
    return xInit + cvt(var0 * torch.randn(*x0.shape))
