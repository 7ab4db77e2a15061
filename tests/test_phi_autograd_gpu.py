"""net(x) and net.getGrad(x) under torch autograd (src/Phi.py:91-138 is plain differentiable torch in the reference): first-order gradients of
both calls -- dPhi/dx through the grad Phi kernel, dPhi/dtheta through the value rows of nocf_phi_value_bwd_f32, the vector-Jacobian product of
grad Phi through nocf_phi_grad_bwd_f32 -- against the oracle differentiated by torch autograd in double precision."""
import pytest
import torch

import neuraloc_amd as na
from oracle import ocflow_oracle as orc
from util_hip import synth_state_dict

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("nTh,m,d,n", [(2, 16, 4, 37), (2, 32, 24, 100), (2, 128, 12, 50), (3, 24, 8, 33), (4, 40, 4, 9), (2, 512, 150, 21), (2, 64, 24, 1)])
def test_value_call_gradients_against_oracle_fp64_autograd(nTh, m, d, n):
    alph = [1.0] * 6
    sd = synth_state_dict(nTh, m, d, seed=nTh + m)
    net = na.Phi(nTh=nTh, m=m, d=d, alph=alph)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, d + 1, generator=g) * 0.7
    gout = torch.randn(n, 1, generator=g)
    xx = x.to(DEV).requires_grad_(True)
    out = net(xx)
    assert out.shape == (n, 1) and out.requires_grad
    out.backward(gout.to(DEV))
    P = orc.PhiParams.from_state_dict({k: v.clone() for k, v in sd.items()}, dtype=torch.float64)
    for t in [*P.K, *P.b, P.w, P.A, P.cw, P.cb]:
        t.requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    want = orc.phi_value(P, x64)
    assert float((out.detach().cpu().double() - want.detach()).abs().max()) <= 2e-5 * float(want.detach().abs().max()) + 1e-5
    want.backward(gout.double())
    ref = {"A": P.A.grad, "c.weight": P.cw.grad, "c.bias": P.cb.grad, "w.weight": P.w.grad}
    for i in range(nTh):
        ref[f"N.layers.{i}.weight"], ref[f"N.layers.{i}.bias"] = P.K[i].grad, P.b[i].grad
    for k, p in net.named_parameters():
        w = ref[k].reshape(p.shape)
        scale = w.abs().max().item()
        err = (p.grad.cpu().double() - w).abs().max().item()
        assert err <= 2e-4 * scale + 1e-6, f"{k}: err {err:g} at scale {scale:g}"
    gx = xx.grad.cpu().double()
    assert float((gx - x64.grad).abs().max()) <= 2e-4 * float(x64.grad.abs().max()) + 1e-6


@pytest.mark.parametrize("nTh,m,d,n", [(2, 16, 4, 37), (2, 32, 24, 100), (2, 128, 12, 50), (3, 24, 8, 33), (4, 40, 4, 9), (2, 512, 150, 21), (2, 64, 24, 1)])
def test_gradient_call_vjp_against_oracle_fp64_autograd(nTh, m, d, n):
    """net.getGrad(x).backward(gbar): (d grad Phi / d x)' gbar and d(gbar . grad Phi)/dtheta against double backward through the oracle"""
    alph = [1.0] * 6
    sd = synth_state_dict(nTh, m, d, seed=nTh + m)
    net = na.Phi(nTh=nTh, m=m, d=d, alph=alph)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n, d + 1, generator=g) * 0.7
    gbar = torch.randn(n, d + 1, generator=g)
    xx = x.to(DEV).requires_grad_(True)
    out = net.getGrad(xx)
    assert out.shape == (n, d + 1) and out.requires_grad
    out.backward(gbar.to(DEV))
    P = orc.PhiParams.from_state_dict({k: v.clone() for k, v in sd.items()}, dtype=torch.float64)
    for t in [*P.K, *P.b, P.w, P.A, P.cw, P.cb]:
        t.requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    want = orc.phi_grad(P, x64)
    assert float((out.detach().cpu().double() - want.detach()).abs().max()) <= 2e-5 * float(want.detach().abs().max()) + 1e-5
    want.backward(gbar.double())
    ref = {"A": P.A.grad, "c.weight": P.cw.grad, "c.bias": P.cb.grad, "w.weight": P.w.grad}
    for i in range(nTh):
        ref[f"N.layers.{i}.weight"], ref[f"N.layers.{i}.bias"] = P.K[i].grad, P.b[i].grad
    for k, p in net.named_parameters():
        w = ref[k].reshape(p.shape) if ref[k] is not None else torch.zeros(p.shape, dtype=torch.float64)
        scale = w.abs().max().item()
        err = (p.grad.cpu().double() - w).abs().max().item()
        assert err <= 2e-4 * scale + 1e-6, f"{k}: err {err:g} at scale {scale:g}"
    gx = xx.grad.cpu().double()
    assert float((gx - x64.grad).abs().max()) <= 2e-4 * float(x64.grad.abs().max()) + 1e-6


def test_calls_without_grad_are_the_plain_kernels_and_double_precision_still_refuses_autograd():
    sd = synth_state_dict(2, 32, 4, seed=1)
    net = na.Phi(nTh=2, m=32, d=4, alph=[1.0] * 6)
    net.load_state_dict(sd)
    net = net.to(DEV)
    x = torch.randn(8, 5, device=DEV)
    with torch.no_grad():
        a, b = net(x), net.getGrad(x)
    assert not a.requires_grad and not b.requires_grad
    for p in net.parameters():
        p.requires_grad_(False)
    assert not net(x).requires_grad and not net.getGrad(x).requires_grad     # nothing to differentiate: the plain calls
    for p in net.parameters():
        p.requires_grad_(True)
    net64 = na.Phi(nTh=2, m=32, d=4, alph=[1.0] * 6).double().to(DEV)
    with pytest.raises(NotImplementedError):
        net64(x.double())                               # double precision: evaluation only
