import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench, neuraloc_amd as na
dev = torch.device("cuda:0")
for name in ("swarm50", "singlequad"):
    meta, sd, xtarget, xInit = bench.load_workload(name)
    net, prob = bench.build_objects(meta, sd, xtarget, dev)
    net.train(); prob.train()
    x = bench.make_states(meta, xInit, meta["n_full"], seed=200).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    ts = []
    for it in range(25):
        torch.cuda.synchronize(); t = time.perf_counter()
        opt.zero_grad()
        Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
        t1 = time.perf_counter()
        Jc.backward()
        t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        ts.append((round((t3 - t) * 1e3, 2), round((t1 - t) * 1e3, 2), round((t2 - t1) * 1e3, 2)))
    print(name, "per iteration (total ms, host time in forward call, host time in backward call):", ts[3:], flush=True)
    print(name, "reserved MB", torch.cuda.memory_reserved() >> 20, "allocated MB", torch.cuda.memory_allocated() >> 20)
