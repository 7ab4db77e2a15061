#!/usr/bin/env python3
"""evalOC-style driver on the MI355X path: load a NeuralOC checkpoint, report the
costs on xInit in the reference's log layout (evalOC.py:76-85), time the deployment like timeOC.py:76-81
(nex=1, nt steps) and a batch, optionally run the shocked rollouts of evalOC.py:113-122.  No plotting."""
import argparse
import time

import torch

import neuraloc_amd as na
from neuraloc_amd.checkpoint import load_checkpoint
from neuraloc_amd.shock import shock_rollout

p = argparse.ArgumentParser("Optimal Control (MI355X)")
p.add_argument("--nt", type=int, default=50, help="number of time steps")
p.add_argument("--resume", type=str, required=True, help="checkpoint written by trainOC (reference or this repo)")
p.add_argument("--batch", type=int, default=1024, help="batch size of the throughput line")
p.add_argument("--gpu", type=int, default=0)
p.add_argument("--do_shock", action="store_true")


def main():
    args = p.parse_args()
    dev = f"cuda:{args.gpu}"
    net, prob, x0, _, xInit, a = load_checkpoint(args.resume, device=dev, n_train=args.batch, n_val=args.batch)
    prob.eval()
    net.eval()
    alph = net.alph
    with torch.no_grad():
        Jc, cs = na.OCflow(xInit, net, prob, [0.0, 1.0], args.nt, "rk4", alph)
        print("{:8s} {:12s} {:11s} {:11s} {:11s} {:11s} {:11s} {:11s} {:11s} ".format(
            "just xInit", "L+G", "L", "G w/ a0", "HJt", "HJfin", "HJgrad", "Q", "W"))
        print("         {:12.4e} {:11.3e} {:11.3e} {:11.3e} {:11.3e} {:11.3e} {:11.3e} {:11.3e}".format(
            cs[0] + alph[0] * cs[1], cs[0], alph[0] * cs[1], alph[3] * cs[2], alph[4] * cs[3], alph[5] * cs[4], cs[5], cs[6]))
        for name, x in (("deployment (nex=1)", xInit), (f"batch (nex={x0.shape[0]})", x0)):
            for _ in range(3):
                na.OCflow(x, net, prob, [0.0, 1.0], args.nt, "rk4", alph)
            torch.cuda.synchronize()
            t0 = time.time()
            reps = 20
            for _ in range(reps):
                na.OCflow(x, net, prob, [0.0, 1.0], args.nt, "rk4", alph)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / reps
            print("%s time: %5f   avg time / RK4 timestep: %5f   trajectories/s: %.1f" % (name, dt, dt / args.nt, x.shape[0] / dt))
        if args.do_shock:
            d = xInit.shape[1]
            shock = torch.zeros(1, d, device=xInit.device)
            shock[0, : min(4, d)] = torch.tensor([-0.2, -0.7, -0.0, -0.6])[: min(4, d)]      # the reference's "minor shock"
            res = shock_rollout(xInit, net, prob, args.nt, 0.1, shock)
            print("shock at t=0.1: nShock=%d, final state error %.4e" %
                  (res["nShock"], float((res["traj"][0, :, -1] - prob.xtarget).norm())))


if __name__ == "__main__":
    main()
