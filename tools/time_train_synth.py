"""one Adam iteration on a synthetic network for an initProb problem: python tools/time_train_synth.py swap12 128 2048 20"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import neuraloc_amd as na
from neuraloc_amd import _lib
from util_hip import synth_state_dict
name, m_, n, nt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
DEV = torch.device("cuda:0")
alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
prob, x0, _, _ = na.initProb(name, n, 8, 0.3, alph, lambda t: t.float().to(DEV))
prob.train()
d = x0.shape[1]
net = na.Phi(nTh=2, m=m_, d=d, alph=alph); net.load_state_dict(synth_state_dict(2, m_, d, seed=1)); net = net.to(DEV).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
def step():
    opt.zero_grad()
    Jc, _ = na.OCflow(x0, net, prob, [0.0, 1.0], nt, "rk4", alph)
    Jc.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t = time.time()
for _ in range(10): step()
torch.cuda.synchronize()
print(name, "m", m_, "n", n, "nt", nt, "NOCF_MONO_BWD", os.environ.get("NOCF_MONO_BWD", "1"), _lib.lib().nocf_last_rollout_kernel().decode(), round((time.time() - t) / 10 * 1e3, 3), "ms per Adam iteration")
