#!/usr/bin/env python3
"""Diagnostic: per-phase shader cycles of the slab kernel (thread 0 of every workgroup), from the -DNOCF_STAMPS=2 build.

  hipcc ... -DNOCF_JIT_ONLY -DNOCF_STAMPS=2 -o neuraloc_amd/csrc/libnocf_stamps.so neuraloc_amd/csrc/nocf_kernels.hip
  NOCF_LIB_PATH=neuraloc_amd/csrc/libnocf_stamps.so python tools/slab_stamps.py [n]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("NOCF_LIB_PATH", os.path.join(REPO, "neuraloc_amd", "csrc", "libnocf_stamps.so"))

import torch                                   # noqa: E402
import bench                                   # noqa: E402
import neuraloc_amd as na                      # noqa: E402
from neuraloc_amd import _lib                  # noqa: E402

NAMES = ["wait S + gather S + barrier", "P1: gemm, z, epilogue, store, publish", "wait U + gather + barrier", "P2: gemm .. publish",
         "wait V + gather + barrier", "P3 + P4: gemm .. publish", "wait G", "reduce + RK + store S | x-costs (waves 2,3)", "publish S",
         "barrier after publish S", "sum p^2 + costs + copy", "(loop top)"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    meta, sd, xtarget, xInit = bench.load_workload("swarm50")
    dev = torch.device("cuda:0")
    net = na.Phi(nTh=meta["nTh"], m=meta["m"], d=meta["d"], alph=meta["alph"])
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    prob = na.SwarmTraj(xtarget.to(dev), obstacle=meta["obstacle"], alph_Q=meta["alph_Q"], alph_W=meta["alph_W"], r=meta["r"])
    prob.eval()
    nt = meta["nt"]
    x = bench.make_states(meta, xInit, n, 200).to(dev)
    nwg = 256 + 8
    buf = torch.zeros(nwg * 12, dtype=torch.int64, device=dev)
    rc = _lib.lib().nocf_debug_set_stamp_buffer(buf.data_ptr())
    assert rc == 0, "this is not the NOCF_STAMPS build"
    with torch.no_grad():
        for _ in range(2):
            na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", meta["alph"])
        torch.cuda.synchronize()
    st = buf.view(nwg, 12).cpu()
    live = st.sum(1) > 0
    acc = st[live].double()
    mean = acc.mean(0)
    tot = mean.sum().item()
    evals = 4 * nt + 1
    print(f"n={n}: {acc.shape[0]} workgroups, {tot:.0f} cycles per workgroup, {tot / evals:.0f} per evaluation (NOCF_SLAB_FAST={os.environ.get('NOCF_SLAB_FAST', '1 (default: same-XCD form where the placement allows)')})")
    for i, nm in enumerate(NAMES):
        print(f"  {nm:32s} {mean[i].item() / evals:9.0f} cyc/eval  {100 * mean[i].item() / tot:5.1f} %   (min {acc[:, i].min().item() / evals:7.0f}, max {acc[:, i].max().item() / evals:7.0f})")


if __name__ == "__main__":
    main()
