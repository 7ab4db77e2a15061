"""diagnostic: the one-CU adjoint vs the per-tile adjoint vs the fp64 oracle's autograd on one synthetic case
   python tools/dbg_mono_bwd.py swap2 19 rk4 0 128"""
import os, sys
os.environ["NOCF_ENV_WATCH"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import neuraloc_amd as na
from util_hip import synth_state_dict
from test_hip_parity import _oracle_grads64
name, n, stepper, training, m_ = sys.argv[1], int(sys.argv[2]), sys.argv[3], bool(int(sys.argv[4])), int(sys.argv[5])
DEV = torch.device("cuda:0")
alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
res = {}
for tag, env in (("mid", "1"), ("tile", "0")):
    os.environ["NOCF_MONO_BWD"] = env
    torch.manual_seed(11)
    prob, x0, _, _ = na.initProb(name, 40, 40, 0.5, alph, lambda t: t.float().to(DEV))
    prob.train() if training else prob.eval()
    x0 = x0[:n].contiguous()
    d = x0.shape[1]
    sd = synth_state_dict(2, m_, d, seed=len(name))
    net = na.Phi(nTh=2, m=m_, d=d, alph=alph); net.load_state_dict(sd); net = net.to(DEV).train()
    xx = x0.clone().requires_grad_(True)
    Jc, _ = na.OCflow(xx, net, prob, [0.0, 1.0], 6, stepper, alph)
    Jc.backward(); torch.cuda.synchronize()
    res[tag] = ({k: p.grad.cpu().double() for k, p in net.named_parameters()}, xx.grad.cpu().double())
J64, want = _oracle_grads64(x0, sd, prob, 6, stepper, alph, 2)
for k in res["mid"][0]:
    w = want[k] if want[k] is not None else torch.zeros_like(res["mid"][0][k])
    sc = w.abs().max().item() + 1e-30
    print(f"{k:20s} scale {sc:10.3e}  mid-oracle {(res['mid'][0][k]-w).abs().max().item()/sc:9.2e}  tile-oracle {(res['tile'][0][k]-w).abs().max().item()/sc:9.2e}  mid-tile {(res['mid'][0][k]-res['tile'][0][k]).abs().max().item()/sc:9.2e}")
dx = (res["mid"][1] - res["tile"][1]).abs().max(1).values
print("dJ/dx0 rows differing mid vs tile:", [(i, f"{v:.2e}") for i, v in enumerate(dx.tolist()) if v > 1e-3 * res["tile"][1].abs().max().item()])
