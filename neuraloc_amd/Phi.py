"""Host-side mirror of the reference's value network interface (src/Phi.py).

`Phi(nTh, m, d, r=10, alph=[1.0]*6)` keeps the reference constructor, attribute names and
state_dict keys (A, c.weight, c.bias, w.weight, N.layers.{i}.weight/.bias), so checkpoints
and NeuralOC-style drivers work unchanged.  The arithmetic does not live here: `forward`
and `getGrad` hand the parameter tensors to the HIP library (nocf_phi_forward_f32 /
nocf_phi_grad_f32).  No torch fallback exists -- CPU tensors raise.
"""
import copy
import ctypes as C

import torch
from torch.autograd.function import once_differentiable
import torch.nn as nn

from . import _lib


def antiderivTanh(x):
    """activation of the ResNet: |x| + log(1+exp(-2|x|)) (src/Phi.py:8-9).  Plain elementwise
    helper kept for API compatibility; the rollout evaluates it inside the HIP kernels."""
    ax = x.abs()
    return ax + torch.log(1 + torch.exp(-2.0 * ax))


def derivTanh(x):
    """1 - tanh^2 (src/Phi.py:12-13; unused upstream, kept for the import surface)."""
    return 1 - torch.tanh(x) ** 2


class ResNN(nn.Module):
    """Parameter container of the ResNet body N (src/Phi.py:16-38): one (d+1)->m opening
    layer and nTh-1 m->m residual layers, step hN = 1/(nTh-1)."""

    def __init__(self, d, m, nTh=2):
        super().__init__()
        if nTh < 2:
            # the reference prints and exit(1)s (src/Phi.py:25-27); a library raises instead
            raise ValueError("nTh must be an integer >= 2")
        self.d, self.m, self.nTh = d, m, nTh
        first = nn.Linear(d + 1, m, bias=True)
        second = nn.Linear(m, m, bias=True)
        self.layers = nn.ModuleList([first, second] + [copy.deepcopy(second) for _ in range(nTh - 2)])
        self.act = antiderivTanh
        self.h = 1.0 / (self.nTh - 1)

    def forward(self, x):
        raise NotImplementedError("ResNN is evaluated inside the fused HIP kernels; call Phi(x) / Phi.getGrad(x)")



_WS_POOL = 4


def _pool_get(pool, key, nbytes, dev):
    """workspace of at least nbytes for (device, stream): most recently used last, at most _WS_POOL entries"""
    ws = pool.pop(key, None)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pool[key] = ws
    while len(pool) > _WS_POOL:
        pool.pop(next(iter(pool)))
    return ws


class Phi(nn.Module):
    """Phi(x,t) = w' N([x;t]) + 1/2 [x;t]' A'A [x;t] + c'[x;t] + c_b   (src/Phi.py:56-138)."""

    def __init__(self, nTh, m, d, r=10, alph=[1.0] * 6):
        super().__init__()
        self.m, self.nTh, self.d, self.alph = m, nTh, d, alph
        r = min(r, d + 1)
        # same construction order as the reference so a seeded init draws the same numbers
        self.A = nn.Parameter(torch.zeros(r, d + 1), requires_grad=True)
        self.A = nn.init.xavier_uniform_(self.A)
        self.c = nn.Linear(d + 1, 1, bias=True)
        self.w = nn.Linear(m, 1, bias=False)
        self.N = ResNN(d, m, nTh=nTh)
        self.w.weight.data = torch.ones(self.w.weight.data.shape)
        self.c.weight.data = torch.zeros(self.c.weight.data.shape)
        self.c.bias.data = torch.zeros(self.c.bias.data.shape)
        self._ws = None

    # ---- boundary helpers -------------------------------------------------------------
    def _c_struct(self, n=0):
        """(NocfPhi, keep-alive list, workspace tensor) for the current parameters; n = rollout batch size
        (0: a Phi-only call) sizes the optional activation-exchange area."""
        dev = self.A.device
        lay = self.N.layers
        keep = []

        def dv(t, name):
            t = _lib.require_device_f32(t.detach(), name)
            keep.append(t)
            return t.data_ptr()

        if self.nTh == 2:
            Kst, bst = lay[1].weight, lay[1].bias
        else:
            Kst = torch.stack([lay[i].weight.detach() for i in range(1, self.nTh)])
            bst = torch.stack([lay[i].bias.detach() for i in range(1, self.nTh)])
        st = _lib.NocfPhi()
        st.d, st.m, st.nTh, st.r = self.d, self.m, self.nTh, self.A.shape[0]
        st.K0 = dv(lay[0].weight, "N.layers.0.weight")
        st.b0 = dv(lay[0].bias, "N.layers.0.bias")
        st.K = dv(Kst, "N.layers[1:].weight")
        st.b = dv(bst, "N.layers[1:].bias")
        st.w = dv(self.w.weight, "w.weight")
        st.A = dv(self.A, "A")
        st.cw = dv(self.c.weight, "c.weight")
        st.cb = 0.0
        st.cb_dev = dv(self.c.bias, "c.bias")           # read on the device: no device-to-host copy (a sync) per call
        # One workspace per module AND STREAM (every call repacks it and the kernels scribble in it while they run): calls on the same
        # Phi from two streams do not share scratch memory.  That makes them memory-safe, not concurrent: the weight-stationary kernels
        # (m = 512: split-role; m <= 128: one-CU) assume the device to themselves -- the split-role kernel needs every CU, two of them at
        # once push each other into the exchange timeout (the call then raises, include/nocf.h).  The pool keeps the _WS_POOL most
        # recently used workspaces (about 50 MB each for a 2048-row swarm50 batch) and is not copied by deepcopy / pickle.
        nbytes = _lib.lib().nocf_rollout_workspace_bytes(self.d, self.m, self.nTh, int(n))
        if nbytes == 0:
            raise RuntimeError("nocf_workspace_bytes: unsupported (d, m, nTh)")
        if self._ws is None:
            self._ws = {}
        ws = _pool_get(self._ws, (dev.index, torch.cuda.current_stream(dev).cuda_stream), nbytes, dev)
        return st, keep, ws

    def _c_struct64(self):
        """(NocfPhi64, keep-alive list, workspace) for a double-precision rollout: the module after .to(torch.float64)"""
        dev = self.A.device
        lay = self.N.layers
        keep = []

        def dv(t, name):
            t = _lib.require_device_f64(t.detach(), name)
            keep.append(t)
            return t.data_ptr()

        if self.nTh == 2:
            Kst, bst = lay[1].weight, lay[1].bias
        else:
            Kst = torch.stack([lay[i].weight.detach() for i in range(1, self.nTh)])
            bst = torch.stack([lay[i].bias.detach() for i in range(1, self.nTh)])
        st = _lib.NocfPhi64()
        st.d, st.m, st.nTh, st.r = self.d, self.m, self.nTh, self.A.shape[0]
        st.K0 = dv(lay[0].weight, "N.layers.0.weight")
        st.b0 = dv(lay[0].bias, "N.layers.0.bias")
        st.K = dv(Kst, "N.layers[1:].weight")
        st.b = dv(bst, "N.layers[1:].bias")
        st.w = dv(self.w.weight, "w.weight")
        st.A = dv(self.A, "A")
        st.cw = dv(self.c.weight, "c.weight")
        st.cb_dev = dv(self.c.bias, "c.bias")
        nbytes = _lib.lib().nocf_workspace_bytes_f64(self.d, self.m, self.nTh)
        if nbytes == 0:
            raise RuntimeError("nocf_workspace_bytes_f64: unsupported (d, m, nTh)")
        pool = getattr(self, "_ws64", None)
        if pool is None:
            pool = self._ws64 = {}
        ws = _pool_get(pool, (dev.index, torch.cuda.current_stream(dev).cuda_stream), nbytes, dev)        # (per stream: see _c_struct)
        return st, keep, ws

    def __getstate__(self):
        st = self.__dict__.copy()                        # (scratch memory is not state: deepcopy / pickle leave it behind)
        st["_ws"] = None
        st.pop("_ws64", None)
        return st

    def _phi64(self, x, value):
        """Phi(s) (n-by-1) or grad Phi (n-by-(d+1)) in double precision (nocf_phi_f64)"""
        x = _lib.require_device_f64(x, "x")
        self._guard_no_autograd(x, "Phi.forward" if value else "Phi.getGrad")
        if x.dim() != 2 or x.shape[1] != self.d + 1:
            raise ValueError(f"x must be n-by-{self.d + 1}")
        st, keep, ws = self._c_struct64()
        n = x.shape[0]
        out = torch.empty((n, 1) if value else (n, self.d + 1), dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            rc = _lib.lib().nocf_phi_f64(C.byref(st), _lib.ptr(x), n, _lib.ptr(out) if value else None, None if value else _lib.ptr(out),
                                         _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device))
        _lib.check(rc, "nocf_phi_f64")
        return out

    def _guard_no_autograd(self, x, what):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError(
                f"{what}: autograd through this stand-alone call is not built (no reference driver differentiates it); "
                "in single precision Jc.backward() through OCflow(...), net(x).backward() and net.getGrad(x).backward() are -- "
                "neuraloc_amd/train.py, Phi.forward, Phi.getGrad.  Call under torch.no_grad()")

    def forward(self, x):
        """Phi(s), n-by-1 (src/Phi.py:91-96)."""
        if isinstance(x, torch.Tensor) and x.dtype == torch.float64:
            return self._phi64(x, value=True)
        x = _lib.require_device_f32(x, "x")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _PhiValueFn.apply(x, self, *[p for _, p in self.named_parameters()])      # first-order autograd (src/Phi.py:91-96)
        return self._value_f32(x)

    def _value_f32(self, x):
        if x.dim() != 2 or x.shape[1] != self.d + 1:
            raise ValueError(f"x must be n-by-{self.d + 1}")
        st, keep, ws = self._c_struct()
        out = torch.empty(x.shape[0], 1, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _lib.lib().nocf_phi_forward_f32(C.byref(st), _lib.ptr(x), x.shape[0], _lib.ptr(out),
                                                 _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device))
        _lib.check(rc, "nocf_phi_forward_f32")
        return out

    def getGrad(self, x):
        """analytic gradient of Phi wrt (x,t), n-by-(d+1) (src/Phi.py:99-138)."""
        if isinstance(x, torch.Tensor) and x.dtype == torch.float64:
            return self._phi64(x, value=False)
        x = _lib.require_device_f32(x, "x")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _PhiGradFn.apply(x, self, *[p for _, p in self.named_parameters()])       # first-order autograd through grad Phi (src/Phi.py:99-138)
        return self._grad_f32(x)

    def _grad_f32(self, x):
        if x.dim() != 2 or x.shape[1] != self.d + 1:
            raise ValueError(f"x must be n-by-{self.d + 1}")
        st, keep, ws = self._c_struct()
        out = torch.empty(x.shape[0], self.d + 1, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _lib.lib().nocf_phi_grad_f32(C.byref(st), _lib.ptr(x), x.shape[0], _lib.ptr(out),
                                              _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device))
        _lib.check(rc, "nocf_phi_grad_f32")
        return out


class _PhiValueFn(torch.autograd.Function):
    """net(x) under autograd (single precision): the value from nocf_phi_forward_f32; backward: dPhi/dx = gout . grad Phi (nocf_phi_grad_f32),
    the parameter gradients from the value rows nocf_phi_value_bwd_f32 streams (any depth), contracted here.  First order only."""

    @staticmethod
    def forward(ctx, x, net, *params):
        xd = x.detach()
        if xd.dim() != 2 or xd.shape[1] != net.d + 1:
            raise ValueError(f"x must be n-by-{net.d + 1}")
        ctx.net, ctx.x_req = net, bool(x.requires_grad)
        ctx.save_for_backward(xd)
        ctx.versions = [p._version for p in net.parameters()]
        return net._value_f32(xd)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        net = ctx.net
        if [p._version for p in net.parameters()] != ctx.versions:
            raise RuntimeError("Phi backward: a parameter was modified in place between this forward and its backward")
        dev = x.device
        n, D1, m, L = x.shape[0], net.d + 1, net.m, net.nTh - 1
        g = gout.detach().reshape(-1).to(torch.float32).contiguous()
        with torch.no_grad():
            gx = net._grad_f32(x) * g[:, None] if ctx.x_req else None
            st, keep, ws = net._c_struct(n)
            rows = 2 * n
            Y, Ob, Wb = (torch.empty(rows, m, device=dev) for _ in range(3))
            V, Ab, Qb, U0 = (torch.empty(L, rows, m, device=dev) for _ in range(4))
            Gb, Sx = torch.empty(rows, D1, device=dev), torch.empty(rows, D1, device=dev)
            with torch.cuda.device(dev):
                rc = _lib.lib().nocf_phi_value_bwd_f32(
                    C.byref(st), _lib.ptr(x), n, _lib.ptr(g), _lib.ptr(Y), _lib.ptr(Ob), _lib.ptr(V), _lib.ptr(Ab), _lib.ptr(Qb),
                    _lib.ptr(U0), _lib.ptr(Wb), _lib.ptr(Gb), _lib.ptr(Sx), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
            _lib.check(rc, "nocf_phi_value_bwd_f32")
            # the value's rows are the second block; the first block carries zero cotangents
            ObV, WbV, sT = Ob[n:], Wb[n:], Sx[n:]
            grads = {"N.layers.0.weight": ObV.t() @ sT, "N.layers.0.bias": ObV.sum(0)}
            for i in range(1, L + 1):
                grads[f"N.layers.{i}.weight"] = Qb[i - 1, n:].t() @ U0[i - 1, n:]
                grads[f"N.layers.{i}.bias"] = Qb[i - 1, n:].sum(0)
            grads["w.weight"] = WbV.sum(0).reshape(1, -1)
            grads["c.weight"] = (g @ sT).reshape(1, -1)
            grads["c.bias"] = g.sum().reshape(1)
            dM = 0.5 * (sT * g[:, None]).t() @ sT
            grads["A"] = net.A.detach() @ (dM + dM.t())
        return (gx, None) + tuple(grads[name] for name, _ in net.named_parameters())


class _PhiGradFn(torch.autograd.Function):
    """net.getGrad(x) under autograd (single precision, first order through the gradient = second order in Phi): the backward is the
    vector-Jacobian product of grad Phi -- the per-tile adjoint's four products per layer on the packed images, nocf_phi_grad_bwd_f32 --
    which returns (d grad Phi / d x)' gbar and streams the rows of the parameter gradients, contracted here."""

    @staticmethod
    def forward(ctx, x, net, *params):
        xd = x.detach()
        if xd.dim() != 2 or xd.shape[1] != net.d + 1:
            raise ValueError(f"x must be n-by-{net.d + 1}")
        ctx.net, ctx.x_req = net, bool(x.requires_grad)
        ctx.save_for_backward(xd)
        ctx.versions = [p._version for p in net.parameters()]
        return net._grad_f32(xd)

    @staticmethod
    @once_differentiable
    def backward(ctx, gbar):
        (x,) = ctx.saved_tensors
        net = ctx.net
        if [p._version for p in net.parameters()] != ctx.versions:
            raise RuntimeError("Phi backward: a parameter was modified in place between this forward and its backward")
        dev = x.device
        n, D1, m, L = x.shape[0], net.d + 1, net.m, net.nTh - 1
        g = gbar.detach().to(torch.float32).contiguous()
        with torch.no_grad():
            st, keep, ws = net._c_struct(n)
            rows = 2 * n
            Y, Ob, Wb = (torch.empty(rows, m, device=dev) for _ in range(3))
            V, Ab, Qb, U0 = (torch.empty(L, rows, m, device=dev) for _ in range(4))
            Gb, Sx = torch.empty(rows, D1, device=dev), torch.empty(rows, D1, device=dev)
            sbar = torch.empty(n, D1, device=dev)
            with torch.cuda.device(dev):
                rc = _lib.lib().nocf_phi_grad_bwd_f32(
                    C.byref(st), _lib.ptr(x), n, _lib.ptr(g), _lib.ptr(sbar), _lib.ptr(Y), _lib.ptr(Ob), _lib.ptr(V), _lib.ptr(Ab), _lib.ptr(Qb),
                    _lib.ptr(U0), _lib.ptr(Wb), _lib.ptr(Gb), _lib.ptr(Sx), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
            _lib.check(rc, "nocf_phi_grad_bwd_f32")
            Y1, Ob1, Wb1, Gb1, Sx1 = Y[:n], Ob[:n], Wb[:n], Gb[:n], Sx[:n]
            grads = {"N.layers.0.weight": Ob1.t() @ Sx1 + Y1.t() @ Gb1, "N.layers.0.bias": Ob1.sum(0)}
            for i in range(1, L + 1):
                grads[f"N.layers.{i}.weight"] = Qb[i - 1, :n].t() @ U0[i - 1, :n] + V[i - 1, :n].t() @ Ab[i - 1, :n]
                grads[f"N.layers.{i}.bias"] = Qb[i - 1, :n].sum(0)
            grads["w.weight"] = Wb1.sum(0).reshape(1, -1)
            grads["c.weight"] = Gb1.sum(0).reshape(1, -1)
            grads["c.bias"] = torch.zeros(1, device=dev)
            dM = Gb1.t() @ Sx1
            grads["A"] = net.A.detach() @ (dM + dM.t())
        return (sbar if ctx.x_req else None, None) + tuple(grads[name] for name, _ in net.named_parameters())
