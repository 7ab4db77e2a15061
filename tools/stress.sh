#!/bin/bash
# stress of the exchange protocols: long runs of the split-role kernel (both exchange forms, both role maps, 1 / 2 / 4 tiles per group) and
# repeated parity suites; every Jc of a configuration must be identical from call to call (bench.py raises if a rollout timed out)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for n in 2048 1024 512 256 128 100; do
  for knobs in "NOCF_DUO_FAST=1" "NOCF_DUO_FAST=0" "NOCF_DUO_MAP=1" "NOCF_DUO_G=16" "NOCF_DUO_G=8" "NOCF_DUO_DBG=8" "NOCF_DUO_DBG=16"; do     # (round 5: both geometries at every size, predictive waiting off / forced on)
    env $knobs timeout 900 python bench.py --n $n --steps 1500 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('n=$n $knobs steps=1500 kernel_ms=%.3f Jc=%.9e' % (j['roofline']['kernel_ms'], j['config']['Jc']))"
  done
done
timeout 600 python bench.py --workload singlequad --steps 3000 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('singlequad steps=3000 kernel_ms=%.3f Jc=%.9e' % (j['roofline']['kernel_ms'], j['config']['Jc']))"
for i in 1 2; do timeout 1500 python -m pytest tests/test_duo_gpu.py tests/test_mono_gpu.py -q 2>&1 | tail -1; done
# training: 200 Adam iterations of the two weight-stationary training paths at a learning rate of 0 (every Jc identical, every gradient finite)
timeout 900 python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench, neuraloc_amd as na
dev = torch.device("cuda:0")
for name in ("swarm50", "singlequad"):
    meta, sd, xtarget, xInit = bench.load_workload(name)
    net, prob = bench.build_objects(meta, sd, xtarget, dev)
    net.train(); prob.train()
    x = bench.make_states(meta, xInit, meta["n_full"], seed=200).to(dev)
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    vals, g0 = set(), None
    for it in range(200):
        opt.zero_grad()
        Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
        Jc.backward()
        opt.step()
        if it % 20 == 0:
            vals.add(float(Jc))
            g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
            assert torch.isfinite(g).all()
            g0 = g.clone() if g0 is None else g0
            assert torch.equal(g, g0), "gradients differ between identical iterations"
    na.check_errors(sync=True)
    print(f"train {name}: 200 iterations, distinct Jc values {len(vals)}, gradients bitwise identical")
PY
