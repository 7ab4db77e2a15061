"""one Adam iteration in double precision (trainOC.py --prec double) on a BASELINE workload: python tools/f64_train_time.py [swarm50|singlequad|softcorridor ...]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
import neuraloc_amd as na
from neuraloc_amd import _lib

for wl in (sys.argv[1:] or ["swarm50", "singlequad", "softcorridor"]):
    meta, sd, xtarget, xInit = bench.load_workload(wl)
    dev = torch.device("cuda:0")
    net, prob = bench.build_objects(meta, sd, xtarget, dev)
    net = net.double().train()
    kw = {} if meta["prob_class"] == "Quadcopter" else {"r": meta["r"]}
    prob = type(prob)(xtarget.double().to(dev), obstacle=meta["obstacle"], alph_Q=meta["alph_Q"], alph_W=meta["alph_W"], **kw)
    prob.train()
    x = bench.make_states(meta, xInit, meta["n_full"], 200).double().to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-5)

    def step():
        opt.zero_grad()
        Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
        Jc.backward()
        opt.step()
        return Jc

    step(); torch.cuda.synchronize(); t = time.time()
    for _ in range(3):
        J = step()
    torch.cuda.synchronize()
    print(wl, "n", meta["n_full"], "nt", meta["nt"], "fp64 Adam iteration", round((time.time() - t) / 3 * 1e3, 1), "ms  Jc", float(J),
          _lib.lib().nocf_last_rollout_kernel().decode(), flush=True)
