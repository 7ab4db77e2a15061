"""helpers for the GPU parity tests: build package objects / oracle records from a golden fixture"""
import torch

import neuraloc_amd as na
from oracle import ocflow_oracle as orc

CLS = {"Cross2D": na.Cross2D, "SwarmTraj": na.SwarmTraj, "Quadcopter": na.Quadcopter}
KIND = {"Cross2D": orc.KIND_CROSS2D, "SwarmTraj": orc.KIND_SWARM, "Quadcopter": orc.KIND_QUAD}


def make_net(g, dev):
    m = g.meta
    net = na.Phi(nTh=m["nTh"], m=m["m"], d=m["d"], alph=m["alph"])
    net.load_state_dict(g.state_dict())
    return net.to(dev).eval()


def make_prob(g, dev, training):
    m = g.meta
    xt = g.t("xtarget").to(dev)
    if m["prob_class"] == "Quadcopter":
        prob = na.Quadcopter(xt, obstacle=None, alph_Q=m["alph_Q"], alph_W=m["alph_W"])
    else:
        prob = CLS[m["prob_class"]](xt, obstacle=m["obstacle"], alph_Q=m["alph_Q"], alph_W=m["alph_W"], r=m["r"])
    prob.train() if training else prob.eval()
    return prob


def make_oracle(g, training):
    m = g.meta
    P = orc.PhiParams.from_state_dict(g.state_dict())
    S = orc.ProbSpec(kind=KIND[m["prob_class"]], xtarget=g.t("xtarget"), obstacle=m["obstacle"],
                     alph_Q=m["alph_Q"], alph_W=m["alph_W"], r=m["r"], training=training)
    return P, S


def closed_form_normal(n, d, seed):
    """same table as tests/golden/make_golden.py (Box-Muller over Weyl sequences)"""
    import numpy as np
    i = np.arange(n * d, dtype=np.float64) + 1.0 + 1000.0 * seed
    u1 = np.clip(np.mod(i * 0.6180339887498949, 1.0), 1e-9, 1.0)
    u2 = np.mod(i * 0.7548776662466927 + 0.31, 1.0)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(z.reshape(n, d).astype(np.float32))


def full_states(g, seed):
    """the BASELINE-size batch make_golden.py used for its full/* entries"""
    m = g.meta
    xInit = g.t("xInit")
    xi = closed_form_normal(m["n_full"], m["d"], seed)
    if m["name"] == "singlequad":
        xi[:, 3:] = 0.0
    x = xInit + m["var0"] * xi
    x[0] = xInit[0]
    return x.contiguous()


def count_off(got, want, rtol, atol):
    got, want = got.double().cpu(), torch.as_tensor(want).double()
    return int(((got - want).abs() > atol + rtol * want.abs()).sum()), float((got - want).abs().max())


def synth_state_dict(nTh, m, d, seed):
    """closed-form weights (no RNG): entries s*sin(a i + b j + phase), s ~ 1/sqrt(fan_in)"""
    def fill(rows, cols, a, b, ph, s):
        i = torch.arange(rows, dtype=torch.float64).unsqueeze(1)
        j = torch.arange(cols, dtype=torch.float64).unsqueeze(0)
        return (s * torch.sin(a * i + b * j + ph)).float()
    r = min(10, d + 1)
    sd = {"A": fill(r, d + 1, 0.37, 0.11, 0.1 + seed, 1.0 / (d + 1) ** 0.5),
          "c.weight": fill(1, d + 1, 0.0, 0.23, 0.4 + seed, 0.3), "c.bias": torch.tensor([0.05]),
          "w.weight": 1.0 + fill(1, m, 0.0, 0.31, 0.7 + seed, 0.2),
          "N.layers.0.weight": fill(m, d + 1, 0.41, 0.13, 0.2 + seed, 1.0 / (d + 1) ** 0.5),
          "N.layers.0.bias": fill(1, m, 0.0, 0.19, 0.3 + seed, 0.1).reshape(m)}
    for l in range(1, nTh):
        sd[f"N.layers.{l}.weight"] = fill(m, m, 0.29 + 0.01 * l, 0.17, 0.5 + seed + l, 1.0 / m ** 0.5)
        sd[f"N.layers.{l}.bias"] = fill(1, m, 0.0, 0.27, 0.6 + seed + l, 0.1).reshape(m)
    return sd


def poison_allocator(dev, big=4):
    """Fill what torch's caching allocator will hand out next with NaN: blocks of every pool size are allocated, filled and freed.  A kernel
    that reads a row before its producer wrote it (or a row nobody writes) then computes with NaN instead of with the previous, identical
    run's values -- the repeated runs of one test would otherwise hide exactly that (found this way: the weight-gradient roles' clamped rows)."""
    ts = [torch.full((1 << 28,), float("nan"), device=dev) for _ in range(big)]
    for words, cnt in ((1 << 12, 100), (1 << 14, 100), (1 << 16, 100), (77312, 50), (200000, 50), (1 << 18, 40), (1 << 20, 30), (1 << 21, 30), (1 << 23, 12), (1 << 25, 6)):
        ts += [torch.full((words,), float("nan"), device=dev) for _ in range(cnt)]
    torch.cuda.synchronize(dev)
    del ts
