#!/usr/bin/env python3
"""Diagnostic: per-phase shader-cycle shares of the rollout kernel, from the -DNOCF_STAMPS build.

  NOCF_STAMPS_LEVEL=2 tools/build_stamps.sh   (hipcc ... -DNOCF_STAMPS=2 -o neuraloc_amd/csrc/libnocf_stamps.so)
  NOCF_LIB_PATH=neuraloc_amd/csrc/libnocf_stamps.so python tools/phase_stamps.py [workload]

Read the SHARES, not the absolute time: the stamps add fences the production kernel does not have.
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

NAMES = ["opening: epilogue+barrier", "opening: z=As + weight stream (wave 0)", "forward: epilogue+barrier", "forward: weight stream",
         "backward: epilogue+barrier", "backward: weight stream", "closing: split-K sum+epilogue+barrier", "closing: weight stream",
         "physics sums", "RK update/finish", "entry barrier", "-"]




def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "swarm50"
    meta, sd, xtarget, xInit = bench.load_workload(wl)
    dev = torch.device("cuda:0")
    net = na.Phi(nTh=meta["nTh"], m=meta["m"], d=meta["d"], alph=meta["alph"])
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    cls = {"Cross2D": na.Cross2D, "SwarmTraj": na.SwarmTraj, "Quadcopter": na.Quadcopter}[meta["prob_class"]]
    kw = {} if meta["prob_class"] == "Quadcopter" else {"r": meta["r"]}
    prob = cls(xtarget.to(dev), obstacle=meta["obstacle"], alph_Q=meta["alph_Q"], alph_W=meta["alph_W"], **kw)
    prob.eval()
    n, nt = meta["n_full"], meta["nt"]
    x = bench.make_states(meta, xInit, n, 200).to(dev)
    nwg = (n + 3) // 4 + 8
    buf = torch.zeros(nwg * 12, dtype=torch.int64, device=dev)
    rc = _lib.lib().nocf_debug_set_stamp_buffer(buf.data_ptr())
    assert rc == 0, "this is not the NOCF_STAMPS build"
    with torch.no_grad():
        for _ in range(2):
            na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", meta["alph"])
        torch.cuda.synchronize()
    st = buf.view(nwg, 12).cpu().double()
    st = st[st.sum(1) > 0]
    mean = st.mean(0)
    tot = mean.sum().item()
    evals = 4 * nt + 1
    print(f"workload {wl}: {st.shape[0]} workgroups, {tot:.0f} cycles per workgroup, {tot / evals:.0f} per evaluation")
    for i, nm in enumerate(NAMES):
        if mean[i] > 0:
            print(f"  {nm:20s} {mean[i].item() / evals:9.0f} cyc/eval  {100 * mean[i].item() / tot:5.1f} %")


if __name__ == "__main__":
    main()
