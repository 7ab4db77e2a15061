#!/usr/bin/env python3
"""trainOC-style driver on the MI355X path.

Same flags, log columns and checkpoint layout as the reference driver (trainOC.py:22-63 flags, :155-160 header,
:176-196 iteration line, :199-207 checkpoint, :249-265 lr decay / resampling / alph switch, including the reference's behaviour at an lr
decay: its "reload of the best parameters" is a no-op because bestParams aliases the live tensors, and so it is here by default
(--lr_reload clone really rolls back, and then also resets Adam's moments)); every OCflow call --
forward, Jc.backward(), validation -- runs in the HIP kernels.  No plotting (viz_freq is accepted and ignored).

One GPU:   python trainOC.py --data softcorridor --niters 200
N GPUs:    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 trainOC.py ...
           (one rank per GPU; each rank draws n_train/N samples; the 8 cost sums and one flat gradient buffer are
           all-reduced over RCCL per iteration, so every rank takes the same Adam step)."""
import argparse
import datetime
import os
import time

import torch

import neuraloc_amd as na
from neuraloc_amd.checkpoint import save_checkpoint
from neuraloc_amd.initProb import PROBLEM_NAMES, initProb, resample


# (flag, type, default, note) -- names and defaults are the reference driver's (trainOC.py:22-63); notes are ours
_FLAGS = [
    ("data", str, "softcorridor", "problem name"),
    ("nt", int, 20, "RK4 steps while training"),
    ("nt_val", int, 32, "RK4 steps in validation plots (kept for compatibility)"),
    ("alph", str, "100.0, 10000.0, 300.0, 0.2, 0.2, 0.2", "weights of G, Q, W, HJt, HJfin, HJgrad"),
    ("m", int, 32, "width of Phi"),
    ("nTh", int, 2, "depth of Phi"),
    ("niters", int, 1800, "Adam iterations"),
    ("lr", float, 0.01, "learning rate"),
    ("optim", str, "adam", "only adam"),
    ("weight_decay", float, 0.0, ""),
    ("resume", str, None, "checkpoint to continue from"),
    ("save", str, "experiments/oc/run", "output directory"),
    ("gpu", int, 0, "device index of a single-process run"),
    ("prec", str, "single", "single or double (trainOC.py:44,76-79): double runs the whole training -- rollout, adjoint, Adam -- in float64"),
    ("approach", str, "ocflow", ""),
    ("lr_reload", str, "alias", "(addition) alias: the reference's behaviour (its reload at lr_freq is a no-op); clone: really roll back to the best validated parameters and reset Adam's moments"),
    ("viz_freq", int, 100, "ignored: nothing is plotted"),
    ("val_freq", int, 25, "validate every this many iterations"),
    ("log_freq", int, 1, "print every this many iterations"),
    ("lr_freq", int, 600, "decay the learning rate every this many iterations"),
    ("lr_decay", float, 0.1, "decay factor"),
    ("n_train", int, 1024, "GLOBAL batch size"),
    ("var0", float, 1.0, "scale of rho_0"),
    ("sample_freq", int, 100, "draw a new batch every this many iterations"),
    ("new_alph", str, None, "'iter, a0, ..., a5': switch the weights at that iteration"),
    ("seed", int, None, "torch seed (rank is added); the reference is unseeded"),
]
_CHOICES = {"data": PROBLEM_NAMES, "optim": ["adam"], "prec": ["single", "double"], "approach": ["ocflow"]}


def parse_args(argv=None):
    p = argparse.ArgumentParser("Optimal Control (MI355X)")
    for name, typ, dflt, note in _FLAGS:
        p.add_argument("--" + name, type=typ, default=dflt, help=note or None, choices=_CHOICES.get(name))
    a = p.parse_args(argv)
    a.alph = [float(v) for v in a.alph.split(",")]
    if a.new_alph is not None:
        a.new_alph = [float(v) for v in a.new_alph.split(",")]
    return a


def main(argv=None):
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("NOCF_TRAIN_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    else:
        dist = None
        dev = torch.device("cuda", args.gpu)
    assert torch.cuda.is_available(), "trainOC.py needs an MI355X: there is no CPU path"
    if args.seed is not None:
        torch.manual_seed(args.seed + rank)
    say = print if rank == 0 else (lambda *a, **k: None)
    n_change = int(args.new_alph[0]) if args.new_alph is not None else -1
    prec = torch.float64 if args.prec == "double" else torch.float32
    cvt = lambda t: t.to(prec).to(dev)                                 # noqa: E731
    lo, hi = na.shard_rows(args.n_train, rank, world)
    n_local = hi - lo
    alph = args.alph
    prob, x0, x0v, xInit = initProb(args.data, n_local, n_local, var0=args.var0, alph=alph, cvt=cvt)
    d, m, nTh, tspan = x0.size(1), args.m, args.nTh, [0.0, 1.0]
    net = na.Phi(nTh=nTh, m=m, d=d, alph=alph)
    if args.resume is not None:
        from neuraloc_amd.checkpoint import load_file
        ck = load_file(args.resume)
        m, nTh = ck["args"].m, ck["args"].nTh
        net = na.Phi(nTh=nTh, m=m, d=d, alph=alph)                      # alph from the command line wins (trainOC.py:129)
        net.load_state_dict(ck["state_dict"])
    net = net.to(prec).to(dev)
    if dist:                                                           # identical initial parameters on every rank
        for prm in net.parameters():
            dist.broadcast(prm.data, src=0)
    optim = torch.optim.Adam(net.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    stamp = datetime.datetime.now().strftime("%Y_%m_%d_%H_%M_%S")
    title = args.data + "_" + stamp + "_alph{:}_{:}_{:}_{:}_{:}_{:}_m{:}".format(*[int(v) for v in alph], m)
    say("DIMENSION={:}  m={:}  nTh={:}   alpha={:}".format(d, m, nTh, alph))
    say("nt={:}   nt_val={:}".format(args.nt, args.nt_val))
    say("Number of trainable parameters: {}".format(sum(p.numel() for p in net.parameters() if p.requires_grad)))
    say("data={:} device={:} ranks={:}".format(args.data, dev, world))
    say("n_train={:}".format(args.n_train))
    say("{:5s} {:7s} {:6s}   {:9s}  {:8s}  {:8s}  {:8s}  {:8s}  {:8s}  {:8s}  {:8s}     {:9s}  {:8s}  {:8s}  {:8s}  {:8s}  {:8s}  {:8s}  {:8s}".format(
        "iter", "lr", "  time", "loss", "L", "G", "HJt", "HJfin", "HJgrad", "Q", "W",
        "valLoss", "valL", "valG", "valHJt", "valHJf", "valHJg", "valQ", "valW"))

    rollout = na.OCflow_sharded if dist else na.OCflow
    best_loss, best_params, total = float("inf"), None, 0.0
    net.train()
    prob.train()
    end = time.time()
    for itr in range(1, args.niters + 1):
        optim.zero_grad()
        Jc, cs = rollout(x0, net, prob, tspan, args.nt, "rk4", net.alph)
        Jc.backward()
        torch.cuda.synchronize()
        na.check_errors()                                 # (after the synchronisation and BEFORE the step: a timed-out rollout raises, its NaN gradients are not applied)
        optim.step()
        dt = time.time() - end
        total += dt
        line = "{:05d} {:7.1e} {:6.2f}   {:9.3e}  {:8.2e}  {:8.2e}  {:8.2e}  {:8.2e}  {:8.2e}  {:8.2e}  {:8.2e}".format(
            itr, optim.param_groups[0]["lr"], dt, Jc.item(), *[c.item() for c in cs])
        if itr % args.val_freq == 0 or itr == args.niters:
            with torch.no_grad():
                net.eval()
                prob.eval()
                vl, vcs = rollout(x0v, net, prob, tspan, args.nt, "rk4", net.alph)   # nt, not nt_val: trainOC.py:191
                line += "    {:9.2e}  {:8.2e}  {:8.2e}  {:8.2e}  {:8.2e}  {:8.2e}  {:8.2e}  {:8.2e} ".format(
                    vl.item(), *[c.item() for c in vcs])
                if vl.item() < best_loss:
                    best_loss = vl.item()
                    best_params = {k: v.detach().clone() for k, v in net.state_dict().items()}
                    if rank == 0:
                        os.makedirs(args.save, exist_ok=True)
                        save_checkpoint(os.path.join(args.save, title + "_checkpt.pth"), net, args)
                net.train()
                prob.train()
        if itr % args.log_freq == 0:
            say(line)
        if itr % args.lr_freq == 0 and best_params is not None:
            # In the reference, bestParams = net.state_dict() ALIASES the live tensors (trainOC.py:199), so its
            # load_state_dict(bestParams) at lr_freq (:250-253) is a no-op: the default here too (--lr_reload alias).  --lr_reload clone
            # rolls back to the best validated parameters for real; Adam's moments belong to the abandoned iterates and are reset.
            if args.lr_reload == "clone":
                net.load_state_dict(best_params)
                optim.state.clear()
            for g in optim.param_groups:
                g["lr"] *= args.lr_decay
        if itr % args.sample_freq == 0:
            x0 = resample(x0, xInit, args.var0, cvt)
        if itr == n_change:
            # like trainOC.py:258-263: the problem is rebuilt with the new Q/W weights; net.alph (what OCflow gets) stays
            alph = args.new_alph[1:]
            prob, _, _, _ = initProb(args.data, n_local, n_local, var0=args.var0, alph=alph, cvt=cvt)
            prob.train()
            say("alph values changed")
        end = time.time()
    say("Training Time: {:} seconds".format(total))
    say("Training has finished.  " + os.path.join(args.save, title))
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    return best_loss


if __name__ == "__main__":
    main()
