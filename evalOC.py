#!/usr/bin/env python3
"""evalOC-style driver on the MI355X path.

Same flags as the reference's evalOC.py:14-24 (--nt --alph --resume --save --prec --approach --make_vid --do_shock): loads a
NeuralOC checkpoint (the reference's .pth files or ones written here), evaluates the costs on xInit and prints them in the
reference's log layout (evalOC.py:76-85), runs the intermediates rollout the reference's plots start from and saves it
(`<save>/figs/eval_<name>.npz`: zFull, ctrlFull -- figures and videos themselves are out of scope),
times the deployment like timeDeployment/timeOC.py:76-81 (nex = 1, nt steps; same 'time: ... avg time / RK4 timestep: ...'
line, appended to `<save>/deploy_times`) plus a batch, and with --do_shock runs the reference's two softcorridor shocks
(evalOC.py:113-122) through neuraloc_amd.shock for any problem.

--prec double runs the double-precision rollout (nocf_rollout_f64), like the reference's evalOC.py:28-31.
Differences, on purpose: --make_vid only states that videos are out of scope; --gpu/--batch are additions."""
import argparse
import os
import time

import numpy as np
import torch

import neuraloc_amd as na
from neuraloc_amd.checkpoint import load_checkpoint
from neuraloc_amd.shock import shock_rollout

p = argparse.ArgumentParser("Optimal Control")
p.add_argument("--nt", type=int, default=50, help="number of time steps")
p.add_argument("--alph", type=str, default="1.0, 1.0, 1.0, 1.0, 1.0, 1.0", help="(ignored like in the reference: the checkpoint's alph is used, evalOC.py:53-54)")
p.add_argument("--resume", type=str, default="experiments/oc/pretrained/softcorridor_nn_checkpt.pth")
p.add_argument("--save", type=str, default="experiments/oc/eval")
p.add_argument("--prec", type=str, default="single", choices=["single", "double"], help="single or double precision")
p.add_argument("--approach", type=str, default="ocflow", choices=["ocflow"])
p.add_argument("--make_vid", default=False, action="store_true", help="including this flag will produce video")
p.add_argument("--do_shock", default=False, action="store_true", help="including this flag will incorporate shocks")
p.add_argument("--batch", type=int, default=1024, help="(addition) batch size of the throughput line")
p.add_argument("--gpu", type=int, default=0, help="(addition) device index")


def main(argv=None):
    args = p.parse_args(argv)
    args.alph = [float(item) for item in args.alph.split(",")]
    prec = torch.float64 if args.prec == "double" else torch.float32          # evalOC.py:28-31
    os.makedirs(os.path.join(args.save, "figs"), exist_ok=True)
    print(args)
    dev = f"cuda:{args.gpu}"
    print(" ")
    print("loading model: {:}".format(args.resume))
    print(" ")
    net, prob, x0, _, xInit, a = load_checkpoint(args.resume, device=dev, n_train=args.batch, n_val=args.batch, dtype=prec)
    prob.eval()
    net.eval()
    alph = net.alph
    nt = args.nt
    strTitle = "eval_" + os.path.basename(args.resume)[:-12]
    out = {}
    with torch.no_grad():
        print("{:8s} {:12s} {:11s} {:11s} {:11s} {:11s} {:11s} {:11s} {:11s} ".format(
            "just xInit", "L+G", "L", "G w/ a0", "HJt", "HJfin", "HJgrad", "Q", "W"))
        Jc, cs = na.OCflow(xInit, net, prob, tspan=[0.0, 1.0], nt=nt, stepper="rk4", alph=alph)
        zFull, ctrlFull = na.OCflow(xInit, net, prob, tspan=[0.0, 1.0], nt=nt, stepper="rk4", alph=alph, intermediates=True)
        print("         {:12.4e} {:11.3e} {:11.3e} {:11.3e} {:11.3e} {:11.3e} {:11.3e} {:11.3e}".format(
            cs[0] + alph[0] * cs[1], cs[0], alph[0] * cs[1], alph[3] * cs[2], alph[4] * cs[3], alph[5] * cs[4], cs[5], cs[6]))
        sPath = os.path.join(args.save, "figs", strTitle + ".npz")
        np.savez(sPath, zFull=zFull.cpu().numpy(), ctrlFull=ctrlFull.cpu().numpy(), Jc=float(Jc), cs=np.array([float(c) for c in cs]))
        print("trajectory saved to " + sPath + " (plots are out of scope here)")
        out["Jc"], out["cs"] = float(Jc), [float(c) for c in cs]

        # -------TIME THE DEPLOYED MODEL (timeDeployment/timeOC.py:76-81: one call, nex = 1), then a warm call and a batch
        with open(os.path.join(args.save, "deploy_times"), "a") as timeFile:
            print("problem: ", a.data, file=timeFile)
            print("device: ", torch.cuda.get_device_name(dev), file=timeFile)
            torch.cuda.synchronize()
            start = time.time()
            na.OCflow(xInit, net, prob, tspan=[0.0, 1.0], nt=nt, stepper="rk4", alph=alph)
            torch.cuda.synchronize()
            end = time.time()
            line = "time: %5f   avg time / RK4 timestep: %5f" % (end - start, (end - start) / nt)
            print(line, file=timeFile)
            print(line)
            out["deploy_time"] = end - start
        for name, x in (("warm deployment (nex=1)", xInit), (f"batch (nex={x0.shape[0]})", x0)):
            for _ in range(3):
                na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph)
            torch.cuda.synchronize()
            t0 = time.time()
            reps = 20
            for _ in range(reps):
                na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph)
            torch.cuda.synchronize()
            na.check_errors()
            dt = (time.time() - t0) / reps
            print("%s time: %5f   avg time / RK4 timestep: %5f   trajectories/s: %.1f" % (name, dt, dt / nt, x.shape[0] / dt))
        if args.make_vid:
            print("video not implemented on this path (plotting is out of scope); the trajectory file above holds the frames' data")
        if args.do_shock:
            d = xInit.shape[1]
            for tag, vals in (("shock", [-0.2, -0.7, -0.0, -0.6]), ("majorshock", [-1.4, -1.0, -5.2, -2.8])):       # evalOC.py:115-120
                shock = torch.zeros(1, d, device=xInit.device, dtype=xInit.dtype)
                shock[0, : min(4, d)] = torch.tensor(vals, dtype=xInit.dtype)[: min(4, d)]
                res = shock_rollout(xInit, net, prob, nt, 0.1, shock)
                np.savez(os.path.join(args.save, "figs", f"{strTitle}_{tag}.npz"), traj=res["traj"].cpu().numpy(), ctrl=res["ctrl"].cpu().numpy())
                print("%s at t=0.1: nShock=%d, final state error %.4e" %
                      (tag, res["nShock"], float((res["traj"][0, :, -1] - prob.xtarget.reshape(-1)).norm())))
    return out


if __name__ == "__main__":
    main()
