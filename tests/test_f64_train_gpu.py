"""Training in double precision (`trainOC.py --prec double`, trainOC.py:44,76-79,172-174): Jc.backward() through the double-precision rollout
-- nocf_rollout_record_f64 + nocf_rollout_bwd_f64 (csrc/nocf_f64_bwd.inc) -- against the oracle differentiated by torch autograd in fp64.
Both sides are double, so the tolerance is rounding of a different summation order: 1e-9 relative to the largest entry."""
import pytest
import torch

import neuraloc_amd as na
from neuraloc_amd import _lib
from util_hip import synth_state_dict
from test_hip_parity import _oracle_grads64

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
F64 = torch.float64


@pytest.mark.parametrize("name,n,stepper,training,nTh,m", [
    ("midcross4", 13, "rk4", True, 2, 24), ("midcross4", 16, "rk1", False, 2, 24), ("softcorridor", 7, "rk4", True, 2, 32),
    ("swap2", 5, "rk4", True, 2, 16), ("swap12", 10, "rk1", True, 2, 32), ("swarm", 5, "rk4", True, 2, 40), ("singlequad", 11, "rk4", True, 2, 128),
    ("singlequad", 8, "rk1", False, 3, 24), ("midcross4", 9, "rk4", True, 3, 24), ("swap12", 6, "rk4", True, 4, 24), ("midcross20", 6, "rk4", True, 2, 300),
    ("swarm50", 3, "rk4", True, 2, 512)])
def test_double_precision_backward_against_oracle_autograd(name, n, stepper, training, nTh, m):
    alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
    torch.manual_seed(11)
    prob, x0, _, _ = na.initProb(name, 24, 24, 0.5, alph, lambda t: t.to(F64).to(DEV))
    prob.train() if training else prob.eval()
    x0 = x0[:n].contiguous()
    d = x0.shape[1]
    sd = synth_state_dict(nTh, m, d, seed=len(name))
    net = na.Phi(nTh=nTh, m=m, d=d, alph=alph)
    net.load_state_dict(sd)
    net = net.to(F64).to(DEV).train()
    nt = 4
    xx = x0.clone().requires_grad_(True)
    Jc, cs = na.OCflow(xx, net, prob, [0.0, 1.0], nt, stepper, alph)
    assert Jc.dtype == F64 and Jc.requires_grad
    Jc.backward()
    torch.cuda.synchronize()
    assert _lib.lib().nocf_last_rollout_kernel().decode() == "rollout_bwd_f64_kernel"
    J64, want = _oracle_grads64(x0.cpu(), sd, prob, nt, stepper, alph, nTh)
    assert abs(Jc.item() - J64) <= 1e-10 * abs(J64)
    for k, p in net.named_parameters():
        w = want[k] if want[k] is not None else torch.zeros_like(p).cpu()
        scale = w.abs().max().item()
        err = (p.grad.cpu() - w.reshape(p.shape)).abs().max().item()
        assert p.grad.dtype == F64 and err <= 1e-9 * scale + 1e-12, f"{name} {k}: err {err:g} at scale {scale:g}"
    # dJc/dx0 against the oracle differentiated with respect to the initial states
    from oracle import ocflow_oracle as orc
    P = orc.PhiParams.from_state_dict({k: v.clone() for k, v in sd.items()}, dtype=F64)
    S = orc.ProbSpec.from_object(prob)
    S.xtarget = S.xtarget.cpu()
    xo = x0.detach().cpu().clone().requires_grad_(True)
    Jo, _ = orc.rollout(xo, P, S.to(F64), [0.0, 1.0], nt, stepper, alph)
    Jo.backward()
    assert float((xx.grad.cpu() - xo.grad).abs().max()) <= 1e-9 * float(xo.grad.abs().max()) + 1e-12


def test_double_precision_training_step_reduces_the_objective():
    alph = [100.0, 1.0e3, 50.0, 0.5, 0.25, 0.125]
    torch.manual_seed(3)
    prob, x0, _, _ = na.initProb("softcorridor", 64, 8, 0.3, alph, lambda t: t.to(F64).to(DEV))
    prob.train()
    d = x0.shape[1]
    net = na.Phi(nTh=2, m=32, d=d, alph=alph).to(F64).to(DEV).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    vals = []
    for _ in range(8):
        opt.zero_grad()
        Jc, _ = na.OCflow(x0, net, prob, [0.0, 1.0], 6, "rk4", alph)
        Jc.backward()
        opt.step()
        vals.append(float(Jc))
    assert vals[-1] < vals[0]
