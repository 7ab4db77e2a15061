"""Does the adjoint with the weight-gradient roles (NOCF_DUO_DW=1) read a row nobody wrote?  The allocator's free blocks are filled with NaN in
front of every call, so such a read shows as NaN gradients instead of as the previous run's (identical) values; prints, per batch size, the
parameters with non-finite gradients and which (group, role) partial sums of the scratch hold them.
usage: python tools/repro_dw.py <G: 8|16[,..]> <n[,n..]>      (round 5: found the clamped rows of groups without rows, nocf_duo_bwd.inc duo_dw_role)"""
import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import neuraloc_amd as na
from neuraloc_amd import _lib
from conftest import load_golden
from util_hip import closed_form_normal, make_net, make_prob
DEV = torch.device("cuda:0")
g = load_golden("swarm50"); m = g.meta
alph = list(m["alph"]); alph[3], alph[4], alph[5] = 2.0, 3.0, 1.5
def poison():
    ts = [torch.full((1 << 28,), float("nan"), device=DEV) for _ in range(8)]
    for words, cnt in ((1 << 12, 200), (1 << 14, 200), (1 << 16, 200), (77312, 100), (200000, 100), (1 << 18, 50), (1 << 20, 40), (1 << 21, 40), (1 << 23, 20), (1 << 25, 10)):
        ts += [torch.full((words,), float("nan"), device=DEV) for _ in range(cnt)]
    torch.cuda.synchronize(); del ts
for G in sys.argv[1].split(","):
  for n in [int(v) for v in sys.argv[2].split(',')]:
    os.environ["NOCF_DUO_G"] = G
    x = (0.3 * (g.t("xInit") + m["var0"] * closed_form_normal(n, m["d"], 9))).contiguous().to(DEV)
    for tag, env in (("tape+dw", {"NOCF_DUO_DW": "1"}),):
        os.environ.pop("NOCF_DUO_DW", None)
        os.environ.update(env)
        net = make_net(g, DEV).train(); prob = make_prob(g, DEV, training=True)
        xx = x.clone().requires_grad_(True)
        poison()
        Jc, _ = na.OCflow(xx, net, prob, [0.0, 1.0], 2, "rk4", alph)
        Jc.backward(); torch.cuda.synchronize(); na.check_errors(sync=True)
        bad = [k for k, p in net.named_parameters() if not torch.isfinite(p.grad).all()]
        if bad:
            from neuraloc_amd import train as _tr
            sc = _tr._SCRATCH[("dw", DEV)]
            p1 = sc[:32 * 512 * 512].view(32, 512, 512); p0 = sc[32 * 512 * 512:32 * 512 * 672].view(32, 512, 160)
            print("    dK1p nonfinite per (group, which):", [(i // 2, i % 2, int((~torch.isfinite(p1[i])).sum())) for i in range(32) if not torch.isfinite(p1[i]).all()])
            print("    dK0p nonfinite per (group, which):", [(i // 2, i % 2, int((~torch.isfinite(p0[i, :, :151])).sum())) for i in range(32) if not torch.isfinite(p0[i, :, :151]).all()])
        for k, p in net.named_parameters():
            if k in bad:
                nz = (~torch.isfinite(p.grad)).nonzero()
                print("   ", k, tuple(p.grad.shape), "nonfinite", nz.shape[0], "rows", nz[:, 0].min().item(), nz[:, 0].max().item(), "cols", nz[:, 1].min().item(), nz[:, 1].max().item())
        print(_lib.lib().nocf_last_rollout_kernel().decode(), G, n, tag, float(Jc.detach()), "nonfinite:", bad, "x:", bool(torch.isfinite(xx.grad).all()), flush=True)
