"""
CPU ORACLE for the OCflow rollout hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The shipped path (neuraloc_amd/) never imports it and has no CPU
fallback: it raises when the HIP library is missing.

What this is: a from-scratch eager-PyTorch restatement of the reference's
algorithm, written as pure functions over two plain records (PhiParams,
ProbSpec).  Every function issues the same aten ops in the same order as the
reference function it cites, so on the same CPU/torch build it reproduces the
reference bit-for-bit (checked by tests/golden/make_golden.py, which imports
/root/reference in the build container and asserts max|diff| == 0 before it
writes the fixtures).  Because it is the same eager workload as the reference,
bench.py times it as the CPU baseline ("kind": "port").

Parity status: PINNED.  tests/golden/*.npz hold outputs of the *reference
itself* (not of this file) on the five pretrained checkpoints; see
tests/test_oracle_golden.py.

Citations are file:line into donken/NeuralOC (mounted at /root/reference in the
build container only).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

KIND_CROSS2D = "cross2d"
KIND_SWARM = "swarmtraj"
KIND_QUAD = "quadcopter"


# ----------------------------------------------------------------------------
# records
# ----------------------------------------------------------------------------
@dataclass
class PhiParams:
    """Weights of the value network (src/Phi.py:57-87): ResNet body K_i, b_i,
    head w, low-rank quadratic A and linear term c."""
    K: List[torch.Tensor]          # K[0]: (m, d+1); K[i>=1]: (m, m)
    b: List[torch.Tensor]          # (m,)
    w: torch.Tensor                # (1, m)
    A: torch.Tensor                # (r, d+1)
    cw: torch.Tensor               # (1, d+1)
    cb: torch.Tensor               # (1,)

    @property
    def nTh(self) -> int:
        return len(self.K)

    @property
    def m(self) -> int:
        return self.K[0].shape[0]

    @property
    def d(self) -> int:
        return self.K[0].shape[1] - 1

    @staticmethod
    def from_state_dict(sd, dtype=None) -> "PhiParams":
        """Keys are the reference's state_dict names."""
        nTh = 0
        while f"N.layers.{nTh}.weight" in sd:
            nTh += 1
        cv = (lambda t: torch.as_tensor(t).to(dtype)) if dtype is not None else torch.as_tensor
        return PhiParams(
            K=[cv(sd[f"N.layers.{i}.weight"]) for i in range(nTh)],
            b=[cv(sd[f"N.layers.{i}.bias"]) for i in range(nTh)],
            w=cv(sd["w.weight"]), A=cv(sd["A"]), cw=cv(sd["c.weight"]), cb=cv(sd["c.bias"]))

    @staticmethod
    def from_module(net) -> "PhiParams":
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        return PhiParams.from_state_dict(sd)

    def to(self, dtype) -> "PhiParams":
        return PhiParams([k.to(dtype) for k in self.K], [b.to(dtype) for b in self.b],
                         self.w.to(dtype), self.A.to(dtype), self.cw.to(dtype), self.cb.to(dtype))


@dataclass
class ProbSpec:
    """Constants of one problem object (src/problem/*.py __init__)."""
    kind: str
    xtarget: torch.Tensor          # (d,)
    obstacle: Optional[str] = None
    alph_Q: float = 1.0
    alph_W: float = 1.0
    r: float = 0.5
    training: bool = True
    mass: float = 1.0
    grav: float = 9.81
    agent_dim: int = field(init=False)
    n_agents: int = field(init=False)

    def __post_init__(self):
        self.agent_dim = {KIND_CROSS2D: 2, KIND_SWARM: 3, KIND_QUAD: 12}[self.kind]
        self.xtarget = self.xtarget.reshape(-1)
        self.n_agents = self.xtarget.numel() // self.agent_dim

    @staticmethod
    def from_object(prob) -> "ProbSpec":
        """Read a duck-typed problem object (the reference's or the package's)."""
        name = type(prob).__name__.lower()
        kind = {"cross2d": KIND_CROSS2D, "swarmtraj": KIND_SWARM, "quadcopter": KIND_QUAD}[name]
        return ProbSpec(kind=kind, xtarget=prob.xtarget.detach().cpu().clone(), obstacle=prob.obstacle,
                        alph_Q=prob.alph_Q, alph_W=prob.alph_W, r=prob.r, training=bool(prob.training),
                        mass=getattr(prob, "mass", 1.0), grav=getattr(prob, "grav", 9.81))

    def to(self, dtype) -> "ProbSpec":
        s = ProbSpec(self.kind, self.xtarget.to(dtype), self.obstacle, self.alph_Q, self.alph_W,
                     self.r, self.training, self.mass, self.grav)
        return s


# ----------------------------------------------------------------------------
# value network  (src/Phi.py)
# ----------------------------------------------------------------------------
def sigma(x):
    """antiderivative of tanh, overflow-safe form (src/Phi.py:8-9)."""
    ax = torch.abs(x)
    return ax + torch.log(1 + torch.exp(-2.0 * ax))


def resnet(P: PhiParams, s):
    """N(s): opening layer then nTh-1 residual layers (src/Phi.py:40-52)."""
    hN = 1.0 / (P.nTh - 1)
    u = sigma(F.linear(s, P.K[0], P.b[0]))
    for i in range(1, P.nTh):
        u = u + hN * sigma(F.linear(u, P.K[i], P.b[i]))
    return u


def phi_value(P: PhiParams, s):
    """Phi(s) = w.N(s) + 1/2 s'(A'A)s + c.s + cb   (src/Phi.py:91-96)."""
    AtA = torch.matmul(torch.t(P.A), P.A)
    quad = 0.5 * torch.sum(torch.matmul(s, AtA) * s, dim=1, keepdims=True)
    return F.linear(resnet(P, s), P.w) + quad + F.linear(s, P.cw, P.cb)


def phi_grad(P: PhiParams, s):
    """Analytic gradient of Phi wrt s=(x,t), n-by-(d+1)   (src/Phi.py:99-138).
    Works feature-major inside, like the reference, so the GEMM shapes (and
    hence the CPU summation order) are identical."""
    hN = 1.0 / (P.nTh - 1)
    AtA = torch.matmul(P.A.t(), P.A)
    pre0 = F.linear(s, P.K[0], P.b[0])
    states = [sigma(pre0)]
    cur = states[0]
    for i in range(1, P.nTh):
        cur = cur + hN * sigma(F.linear(cur, P.K[i], P.b[i]))
        states.append(cur)
    back = 0.0
    for i in range(P.nTh - 1, 0, -1):
        seed = P.w.t() if i == P.nTh - 1 else back
        gate = torch.tanh(F.linear(states[i - 1], P.K[i], P.b[i])).t()
        back = seed + hN * torch.mm(P.K[i].t(), gate * seed)
    gate0 = torch.tanh(pre0)
    back = torch.mm(P.K[0].t(), gate0.t() * back)
    g = back + torch.mm(AtA, s.t()) + P.cw.t()
    return g.t()


# ----------------------------------------------------------------------------
# problem physics  (src/problem/Cross2D.py, SwarmTraj.py, Quadcopter.py, src/utils.py:70-86)
# ----------------------------------------------------------------------------
def gauss_pdf(x, mu, cov):
    """diagonal-covariance Gaussian density (src/utils.py:70-86)."""
    n, k = x.shape
    mu = mu.view(1, k)
    cov = cov.view(1, k)
    denom = (2 * math.pi) ** (0.5 * k) * torch.sqrt(torch.prod(cov))
    num = torch.exp(-0.5 * torch.sum((x - mu) ** 2 / cov, 1, keepdims=True))
    return num / denom


def _tt(vals, like):
    return torch.tensor(vals, dtype=like.dtype, device=like.device).view(1, -1)


def _agent_obstacle(S: ProbSpec, xa):
    """Per-agent obstacle value; xa is (n*nAgents, agentDim).
    Cross2D.py:90-116 ; SwarmTraj.py:90-122 ; Quadcopter.py:116-122."""
    ref = S.xtarget
    if S.kind == KIND_CROSS2D:
        if S.obstacle == "softcorridor":
            cov = _tt([0.2, 0.2], ref)
            q = gauss_pdf(xa, _tt([-2.5, 0.], ref), cov)
            q2 = gauss_pdf(xa, _tt([2.5, 0.], ref), cov)
            q3 = gauss_pdf(xa, _tt([-1.5, 0.], ref), cov)
            q4 = gauss_pdf(xa, _tt([1.5, 0.], ref), cov)
            return q + q2 + q3 + q4
        if S.obstacle == "hardcorridor":
            mu1, mu2, cov = _tt([0., 4.], ref), _tt([0., -3.5], ref), _tt([1., 1.], ref)
            q = gauss_pdf(xa, mu1, cov) + gauss_pdf(xa, mu2, cov)
            if S.training:
                inside = (torch.norm(xa - mu1, dim=1) < 2.0 + S.r) | (torch.norm(xa - mu2, dim=1) < 2.0 + S.r)
            else:
                inside = (torch.norm(xa - mu1, dim=1) < 2.0) | (torch.norm(xa - mu2, dim=1) < 2.0)
                return inside                      # eval mode: the boolean mask itself
            q[~inside] = 0.0
            return q
        return 0.0 * xa
    if S.kind == KIND_SWARM:
        if S.obstacle == "blocks":
            pos = xa[:, 0:3]
            mu1, mu2 = _tt([0., 0., 2.], ref), _tt([2.5, 0., 2.], ref)
            cov1, cov2 = 3. * _tt([3., 1., 3.], ref), 3. * _tt([3., 1., 1.], ref)
            q = gauss_pdf(pos, mu1, cov1) + gauss_pdf(pos, mu2, cov2) + 999.
            px, py, pz = pos[:, 0], pos[:, 1], pos[:, 2]
            if S.training:
                r = S.r
                inside = ((px < 2.0 + r) & (px > -2.0 - r) & (py < 0.5 + r) & (py > -0.5 - r) & (pz < 7.0 + r)) \
                    | ((px < 4.0 + r) & (px > 2.0 - r) & (py < 1.0 + r) & (py > -1.0 - r) & (pz < 4.0 + r))
            else:
                inside = ((px < 2.0) & (px > -2.0) & (py < 0.5) & (py > -0.5) & (pz < 7.0)) \
                    | ((px < 4.0) & (px > 2.0) & (py < 1.0) & (py > -1.0) & (pz < 4.0))
                return inside.unsqueeze(1)
            q[~inside] = 0.0
            return q
        return 0.0 * xa
    # quadcopter: only obstacle=None is implemented by the reference
    return 0.0 * xa


def prob_Q(S: ProbSpec, x):
    """sum over agents of the obstacle value (Cross2D.py:118-125 and twins)."""
    if S.obstacle is not None:
        q = _agent_obstacle(S, x.reshape(-1, S.agent_dim))
        return torch.sum(q.reshape(x.shape[0], -1), dim=1, keepdim=True)
    return 0.0 * x[:, 0].unsqueeze(1)


def prob_W(S: ProbSpec, x):
    """pairwise interaction cost (Cross2D.py:127-162 ; SwarmTraj.py:131-164)."""
    k = S.agent_dim
    if S.kind == KIND_SWARM:
        fac_train, fac_many = 2.2, 3.2
    else:
        fac_train, fac_many = 2.2, 2.2
    if S.n_agents == 1:
        return 0.0 * x[:, 0]
    if S.n_agents == 2:
        dist = torch.norm(x[:, 0:k] - x[:, k:2 * k], p=2, dim=1, keepdim=True)
        near = dist < (fac_train * S.r if S.training else 2 * S.r)
        return near * torch.exp(-dist ** 2 / (2 * S.r ** 2))
    n = x.size(0)
    xa = x.view(n, S.n_agents, k)
    dist = torch.norm(xa.reshape(n, S.n_agents, 1, k) - xa.reshape(n, 1, S.n_agents, k), p=2, dim=3)
    near = dist < (fac_many * S.r if S.training else 2 * S.r)
    e = torch.exp(-(near * dist) ** 2 / (2 * S.r ** 2))
    ones = e == 1.
    return ((e.sum(dim=[1, 2]) - ones.sum(dim=[1, 2])) / 2.).view(-1, 1)


def _quad_f(ang):
    """rotation helpers f7,f8,f9 (Quadcopter.py:176-197)."""
    sps, sth, sph = torch.sin(ang[:, 0]), torch.sin(ang[:, 1]), torch.sin(ang[:, 2])
    cps, cth, cph = torch.cos(ang[:, 0]), torch.cos(ang[:, 1]), torch.cos(ang[:, 2])
    f7 = sps * sph + cps * sth * cph
    f8 = - cps * sph + sps * sth * cph
    f9 = cth * cph
    return f7, f8, f9


def _quad_u(S: ProbSpec, xa, pa):
    """thrust (Quadcopter.py:160-163)."""
    f7, f8, f9 = _quad_f(xa[:, 3:6])
    u = -1 / (2 * S.mass) * (f7 * pa[:, 6] + f8 * pa[:, 7] + f9 * pa[:, 8]).view(-1, 1)
    return u, f7, f8, f9


def _quad_W(S: ProbSpec, x):
    """Quadcopter.py:133-158.  nAgents>2 is broken upstream (line 146 slices the
    batch, not the coordinates) and no shipped config reaches it."""
    if S.n_agents == 1:
        return (0.0 * x[:, 0]).view(-1, 1)
    if S.n_agents == 2:
        dist = torch.norm(x[:, 0:3] - x[:, 12:15], p=2, dim=1, keepdim=True)
        return (dist < 2 * S.r) * torch.exp(-dist ** 2 / (2 * S.r ** 2))
    raise NotImplementedError("Quadcopter interaction cost for nAgents>2 is unreachable/buggy upstream")


def prob_LHQW(S: ProbSpec, x, p):
    """Lagrangian, Hamiltonian, obstacle and interaction costs.
    Cross2D.py:73-87 ; SwarmTraj.py:71-87 ; Quadcopter.py:86-113."""
    if S.kind == KIND_CROSS2D:
        Q = S.alph_Q * prob_Q(S, x)
        L = 0.5 * torch.sum(p ** 2, dim=1, keepdims=True) + Q
        if S.alph_W != 0.0:
            W = prob_W(S, x)
            L = L + S.alph_W * W
        else:
            W = 0.0 * L
        H = -L + torch.sum(p ** 2, dim=1, keepdims=True)
        return L, H, Q, W
    if S.kind == KIND_SWARM:
        Q = prob_Q(S, x).view(-1, 1) if S.alph_Q > 0 else 0. * x[:, 0].view(-1, 1)
        L = 0.5 * torch.sum(p ** 2, dim=1, keepdims=True) + S.alph_Q * Q
        if S.alph_W != 0.0:
            W = prob_W(S, x)
            L = L + S.alph_W * W
        else:
            W = 0.0 * L
        H = -L + torch.sum(p ** 2, dim=1, keepdims=True)
        return L, H, Q, W
    # quadcopter
    H = 0.
    Q = prob_Q(S, x).view(-1, 1)
    L = S.alph_Q * Q
    if S.alph_W > 0.0:
        W = _quad_W(S, x).view(-1, 1)
        L = L + S.alph_W * W
    else:
        W = 0.0 * L
    for i in range(S.n_agents):
        xa, pa = x[:, 12 * i:12 * (i + 1)], p[:, 12 * i:12 * (i + 1)]
        sq = (pa[:, 9] ** 2 + pa[:, 10] ** 2 + pa[:, 11] ** 2).view(-1, 1)
        u, f7, f8, f9 = _quad_u(S, xa, pa)
        L = L + 2 + u ** 2 + 0.25 * sq
        H = H - L \
            - torch.sum(xa[:, 6:9] * pa[:, 0:3], dim=1, keepdims=True) \
            - torch.sum(xa[:, 9:12] * pa[:, 3:6], dim=1, keepdims=True) \
            - (u / S.mass) * (f7 * pa[:, 6] + f8 * pa[:, 7] + f9 * pa[:, 8]).unsqueeze(1) \
            + S.grav * pa[:, 8].unsqueeze(1) + 0.5 * sq
    return L, H, Q, W


def prob_gradpH(S: ProbSpec, x, p):
    """dH/dp; the state moves with its negative.
    Cross2D.py:69-70 ; SwarmTraj.py:68-69 ; Quadcopter.py:65-84."""
    if S.kind != KIND_QUAD:
        return p
    out = torch.empty(0, device=x.device, dtype=x.dtype)
    for j in range(S.n_agents):
        xa, pa = x[:, 12 * j:12 * (j + 1)], p[:, 12 * j:12 * (j + 1)]
        u, f7, f8, f9 = _quad_u(S, xa, pa)
        out = torch.cat((out,
                         - xa[:, 6:],
                         - (u / S.mass) * f7.view(-1, 1),
                         - (u / S.mass) * f8.view(-1, 1),
                         - (u / S.mass) * f9.view(-1, 1) + S.grav,
                         (1. / 2.) * pa[:, 9:12]), dim=1)
    return out


def prob_ctrls(S: ProbSpec, x, p):
    """controls along the path (Cross2D.py:164-165 ; SwarmTraj.py:166-167 ; Quadcopter.py:165-174)."""
    if S.kind != KIND_QUAD:
        return -p
    out = torch.empty(0, device=x.device, dtype=x.dtype)
    for j in range(S.n_agents):
        xa, pa = x[:, 12 * j:12 * (j + 1)], p[:, 12 * j:12 * (j + 1)]
        u, _, _, _ = _quad_u(S, xa, pa)
        out = torch.cat((out, u, -0.5 * pa[:, 9:12]), dim=1)
    return out


# ----------------------------------------------------------------------------
# rollout  (src/OCflow.py)
# ----------------------------------------------------------------------------
def rhs(P: PhiParams, S: ProbSpec, z, t):
    """right-hand side of the (d+4)-component ODE (src/OCflow.py:104-140)."""
    n, width = z.shape
    d = width - 4
    s = F.pad(z[:, :d], (0, 1, 0, 0), value=t)
    g = phi_grad(P, s)
    L, H, Q, W = prob_LHQW(S, s[:, :d], g[:, 0:d])
    out = torch.zeros(n, d + 4, dtype=z.dtype, device=z.device)
    out[:, 0:d] = - prob_gradpH(S, s[:, :d], g[:, 0:d])
    out[:, d] = L.squeeze()
    out[:, d + 1] = torch.abs(g[:, -1] - H.squeeze())
    out[:, d + 2] = Q.squeeze()
    out[:, d + 3] = W.squeeze()
    return out


def step_rk4(P, S, z, t0, t1):
    """classical RK4 (src/OCflow.py:157-184); h is re-derived as t1-t0 like the reference."""
    h = t1 - t0
    z0 = z
    k = h * rhs(P, S, z0, t0)
    z = z0 + (1.0 / 6.0) * k
    k = h * rhs(P, S, z0 + 0.5 * k, t0 + (h / 2))
    z += (2.0 / 6.0) * k
    k = h * rhs(P, S, z0 + 0.5 * k, t0 + (h / 2))
    z += (2.0 / 6.0) * k
    k = h * rhs(P, S, z0 + k, t0 + h)
    z += (1.0 / 6.0) * k
    return z


def step_rk1(P, S, z, t0, t1):
    """forward Euler, in place (src/OCflow.py:143-155)."""
    z += (t1 - t0) * rhs(P, S, z, t0)
    return z


def rollout(x, P: PhiParams, S: ProbSpec, tspan, nt, stepper="rk4", alph=(1.0,) * 6,
            intermediates=False, noMean=False):
    """OCflow (src/OCflow.py:7-95).  Returns (Jc, cs) or (zFull, ctrlFull)."""
    n, d = x.shape
    h = (tspan[1] - tspan[0]) / nt

    z = torch.cat((x, torch.zeros(n, 4, dtype=x.dtype, device=x.device)), 1)
    tk = tspan[0]

    if intermediates:
        # the reference evaluates grad Phi at t=0 here only to size ctrlFull (src/OCflow.py:27-43)
        p_init = phi_grad(P, F.pad(x, [0, 1, 0, 0], value=0))[:, 0:d]
        zFull = torch.zeros(*z.shape, nt + 1, device=x.device, dtype=x.dtype)
        zFull[:, :, 0] = z
        c0 = prob_ctrls(S, z[:, 0:d], p_init)
        ctrlFull = torch.zeros(*c0.shape, nt + 1, dtype=x.dtype, device=x.device)

    for k in range(nt):
        if stepper == "rk4":
            z = step_rk4(P, S, z, tk, tk + h)
        elif stepper == "rk1":
            z = step_rk1(P, S, z, tk, tk + h)
        tk += h
        if intermediates:
            zFull[:, :, k + 1] = z
            s = F.pad(z[:, 0:d], [0, 1, 0, 0], value=tk - h)
            ctrlFull[:, :, k + 1] = prob_ctrls(S, z[:, 0:d], phi_grad(P, s)[:, 0:d])

    resG = z[:, 0:d] - S.xtarget
    cG = 0.5 * torch.sum(resG ** 2, 1, keepdims=True)
    sT = F.pad(z[:, 0:d], [0, 1, 0, 0], value=tspan[1])
    phi1 = phi_value(P, sT)
    gphi1 = phi_grad(P, sT)[:, 0:d]

    if noMean:
        cL = z[:, -4].view(-1, 1)
        cGv = cG.view(-1, 1)
        cHJt = z[:, -3].view(-1, 1)
        cHJf = torch.sum(torch.abs(phi1 - alph[0] * cG), 1).view(-1, 1)
        cHJg = torch.sum(torch.abs(gphi1 - alph[0] * resG), 1).view(-1, 1)
        cQ = z[:, -2].view(-1, 1)
        cW = z[:, -1].view(-1, 1)
        cs = [cL, cGv, cHJt, cHJf, cHJg, cQ, cW]
        Jc = cL + alph[0] * cGv + alph[3] * cHJt + alph[4] * cHJf + alph[5] * cHJg
        return Jc, cs

    cL = torch.mean(z[:, -4])
    cGm = torch.mean(cG)
    cHJt = torch.mean(z[:, -3])
    cHJf = torch.mean(torch.sum(torch.abs(phi1 - alph[0] * cG), 1))
    cHJg = torch.mean(torch.sum(torch.abs(gphi1 - alph[0] * resG), 1))
    cQ = torch.mean(z[:, -2])
    cW = torch.mean(z[:, -1])
    cs = [cL, cGm, cHJt, cHJf, cHJg, cQ, cW]
    Jc = cL + alph[0] * cGm + alph[3] * cHJt + alph[4] * cHJf + alph[5] * cHJg

    if intermediates:
        return zFull, ctrlFull
    return Jc, cs


def persample_table(x, P, S, tspan, nt, stepper="rk4", alph=(1.0,) * 6):
    """(n,7) table [L,G,HJt,HJfin,HJgrad,Q,W] = the noMean outputs stacked; the
    layout the C-ABI's `persample` buffer uses."""
    _, cs = rollout(x, P, S, tspan, nt, stepper, alph, noMean=True)
    return torch.cat([c.reshape(-1, 1) for c in cs], dim=1)
