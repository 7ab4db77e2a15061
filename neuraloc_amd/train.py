"""Training path: `Jc = OCflow(x0, net, prob, ...); Jc.backward()` (trainOC.py:172-174) on the MI355X.

A torch.autograd.Function whose forward is the fused HIP rollout (recording the stage inputs) and whose backward is
the hand-written adjoint of the discrete RK scheme (nocf_rollout_bwd_f32, csrc/nocf_bwd.inc).  The kernel streams the
per-evaluation vectors whose outer products are the weight gradients; the contractions over all samples and
evaluations are plain library GEMMs (torch.matmul).  Any depth nTh >= 2 (LDS permitting), Cross2D / SwarmTraj / Quadcopter, rk4 / rk1, fp32.
Only Jc carries a gradient (the 7 logged costs are detached, like the values trainOC prints)."""
import ctypes as C
import os

import torch
from torch.autograd.function import once_differentiable

from . import _lib

_STEPPERS = {"rk4": _lib.NOCF_RK4, "rk1": _lib.NOCF_RK1}


def _step_sizes(tspan, nt):
    """fp32 step sizes exactly as the rollout kernels form them: (float)((tk+h)-tk), tk += h in double"""
    h = (float(tspan[1]) - float(tspan[0])) / nt
    tk = float(tspan[0])
    out = []
    for _ in range(nt):
        out.append((tk + h) - tk)
        tk += h
    return torch.tensor(out, dtype=torch.float32)


_SCRATCH = {}
# measurement hook (bench.py): with "on" set, every weight-gradient contraction that runs as a LIBRARY GEMM (torch.bmm / matmul = hipBLASLt) is
# bracketed by events, so that the bench line can say how much of an iteration a vendor library computes (`vendor_gemm_ms`)
VENDOR_GEMM = {"on": False, "events": []}


def _vendor(fn):
    if not VENDOR_GEMM["on"]:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    VENDOR_GEMM["events"].append((e0, e1))
    return r


def _contract(X, Y, out=None):
    """X' Y for row streams X [K, m], Y [K, n] (out += when given): the weight gradients are sums of outer products over all
    samples and evaluations, K = 10^5..10^6 rows.  One library GEMM with such an output is a handful of tiles -- a 512 x 512 output
    64 of them, a quarter of the chip; 128 x 128 four -- so the rows are cut into S slabs (a power of two that divides K, slabs of
    >= 2048 rows), one BATCHED GEMM (hipBLASLt through torch.bmm) forms the S partial products and a fixed-order sum adds them:
    swarm50 512 x 512: 2.15 -> 1.17 ms, singlequad 128 x 128: 0.86 (the library's own two-launch contraction) -> 0.25 ms
    (tools/gemm_splitk_probe.py).  Row counts without such a divisor: the library's two-launch contraction (small outputs) or
    one GEMM."""
    m, n = X.shape[1], Y.shape[1]
    K = X.shape[0]
    S = next((s_ for s_ in (256, 128, 64, 32, 16, 8, 4, 2) if K % s_ == 0 and K // s_ >= 2048 and s_ * m * n <= (1 << 25)), 1)
    if S > 1 and X.is_contiguous() and Y.is_contiguous():
        r = _vendor(lambda: torch.bmm(X.view(S, K // S, m).transpose(1, 2), Y.view(S, K // S, n)).sum(0))
        return r if out is None else out.add_(r)
    if m > 512 or n > 512 or m * n > 128 * 160 or X.dtype != torch.float32:
        r = _vendor(lambda: X.t() @ Y)
        return r if out is None else out.add_(r)
    dev = X.device
    sc = _SCRATCH.get(dev)
    if sc is None:
        sc = _SCRATCH[dev] = torch.empty(1024 * 4096, device=dev)
    acc = out is not None
    if out is None:
        out = torch.empty(m, n, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().nocf_contract_f32(_lib.ptr(X), _lib.ptr(Y), X.shape[0], m, n, _lib.ptr(out), int(acc),
                                          _lib.ptr(sc), sc.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "nocf_contract_f32")
    return out


def _colsum(X):
    """column sums of a row stream X [K, n] -> [n] (nocf_colsum_f32: two launches, fixed order, one pass over X)"""
    dev = X.device
    sc = _SCRATCH.get(dev)
    if sc is None:
        sc = _SCRATCH[dev] = torch.empty(1024 * 4096, device=dev)
    out = torch.empty(X.shape[1], device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().nocf_colsum_f32(_lib.ptr(X), X.shape[0], X.shape[1], _lib.ptr(out), 0, _lib.ptr(sc), sc.numel(),
                                        _lib.stream_ptr(dev))
    _lib.check(rc, "nocf_colsum_f32")
    return out


def _unpack_partials(gpart, m, D1, net):
    """partial gradient vectors [rows, nocf_small_grad_floats] (per sample: lane adjoint; per workgroup: one-CU adjoint) -> the gradients by
    parameter name; the rows are added in a fixed order"""
    gv = gpart.sum(0)
    o = 0
    grads = {}
    for name, shape in (("N.layers.0.weight", (m, D1)), ("N.layers.0.bias", (m,)), ("N.layers.1.weight", (m, m)),
                        ("N.layers.1.bias", (m,)), ("w.weight", (1, m)), ("c.weight", (1, D1)), ("c.bias", (1,)),
                        ("dM", (D1, D1))):
        cnt = 1
        for v_ in shape:
            cnt *= v_
        grads[name] = gv[o:o + cnt].reshape(shape)
        o += cnt
    dM = grads.pop("dM")
    grads["A"] = net.A.detach() @ (dM + dM.t())
    return grads


class _OCflowTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, prob, tspan, nt, stepper, alph, n_total, group, *params):
        out = _OCflowTrain._forward_once(ctx, x, net, prob, tspan, nt, stepper, alph, n_total, group)
        if out is None:                                  # (a timed-out launch during the split-role kernels' probation: _lib.duo_guard)
            out = _OCflowTrain._forward_once(ctx, x, net, prob, tspan, nt, stepper, alph, n_total, group)
        return out

    @staticmethod
    def _forward_once(ctx, x, net, prob, tspan, nt, stepper, alph, n_total, group):
        ctx.x_needs_grad = bool(x.requires_grad)
        x = _lib.require_device_f32(x.detach(), "x")
        # the kernels index with Phi's d: a mismatching x would read and write out of bounds (same checks as OCflow._launch)
        if x.dim() != 2:
            raise ValueError("x must be nex-by-d")
        n, d = x.shape
        if d != net.d:
            raise ValueError(f"x has d={d} but Phi was built for d={net.d}")
        if len(alph) < 6:
            raise ValueError("alph needs 6 entries")
        if int(nt) < 1:
            raise ValueError("nt must be >= 1")
        pd_ = getattr(prob, "d", None)
        if pd_ is not None and int(pd_) != d:
            raise ValueError(f"the problem object has d={pd_} but x has d={d}")
        dev = x.device
        phi_st, keep1, ws = net._c_struct(n)
        prob_st, keep2 = prob._c_struct(dev)
        nstage = 4 if stepper == "rk4" else 1
        persample = torch.empty(n, 7, device=dev)
        sums = torch.empty(8, device=dev)
        z_out = torch.empty(n, d + 4, device=dev)
        alph_c = (C.c_float * 6)(*[float(a) for a in alph[:6]])
        _lib.check_errors()
        ctx.tape = None
        ctx.mid_rows = 0
        with torch.cuda.device(dev):
            L = _lib.lib_for(net.d, net.m, net.nTh, phi_st.r, prob_st.n_agents, fwd=prob_st.kind != _lib.PROB_QUADCOPTER)      # (the recording forward; the adjoint below asks again)
            # Training tape (wide two-layer networks on the split-role kernel): the recording forward keeps what autograd would keep of the
            # unrolled graph -- u0, tanh(o), tanh(q), a, grad Phi and three scalars of every evaluation, the terminal one included, 2.9 GB for
            # swarm50 -- and the backward is the split-role adjoint (csrc/nocf_duo_bwd.inc).  NOCF_DUO_BWD=0: the per-tile adjoint below.
            # (nocf_tape_floats knows the network's shape only: the split-role ADJOINT also needs a point-agent problem and r <= 10 -- asked
            # here, so that a shape the tape cannot serve takes the activation-record path below instead of holding 2.9 GB for nothing)
            tape_ok = (hasattr(L, "nocf_tape_floats") and os.environ.get("NOCF_ACT_REC", "1") not in ("0", "")
                       and prob_st.kind != _lib.PROB_QUADCOPTER and phi_st.r <= 10)
            ntape = int(L.nocf_tape_floats(int(d), int(net.m), int(net.nTh), int(n), int(nt), _STEPPERS[stepper])) if tape_ok else 0
            tape = None
            if ntape:
                try:
                    tape = torch.empty(ntape, device=dev)
                except torch.OutOfMemoryError:
                    tape = None
            if tape is not None:
                s_all = torch.empty(nt * nstage + 1, n, d + 1, device=dev)
                recorded = C.c_int32(0)
                rc = L.nocf_rollout_tape_f32(C.byref(phi_st), C.byref(prob_st), _lib.ptr(x), n,
                                             float(tspan[0]), float(tspan[1]), int(nt), _STEPPERS[stepper], alph_c,
                                             _lib.ptr(z_out), _lib.ptr(persample), _lib.ptr(sums), _lib.ptr(s_all),
                                             _lib.ptr(tape), C.byref(recorded),
                                             _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
                _lib.check(rc, "nocf_rollout_tape_f32")
                if recorded.value:
                    ctx.tape = tape
                    ctx.tape_lib = L                              # the adjoint must read this tape with the library that wrote it
                else:                                             # another kernel ran (residency, a knob): no tape -- the forward is
                    tape = None                                   # repeated below WITH the activation record that kernel can write
        if tape is not None:
            return _OCflowTrain._forward_tail(ctx, L, dev, net, prob, tspan, nt, stepper, alph, n_total, group, n, sums, s_all, z_out)
        s_all = torch.empty(nt * nstage, n, d + 1, device=dev)
        with torch.cuda.device(dev):
            # activation record (wide two-layer networks on the split-role kernel): the forward keeps u0, tanh(o), tanh(q), a and grad Phi
            # of every evaluation -- autograd's saved tensors, 2.9 GB for swarm50 -- and the adjoint loads them instead of re-running
            # grad Phi's forward sweep (NOCF_ACT_REC=0: recompute, as for every other shape)
            ctx.mid_rows = int(L.nocf_mid_grad_rows(int(d), int(net.m), int(net.nTh), int(phi_st.r), int(prob_st.n_agents), int(n))) \
                if hasattr(L, "nocf_rollout_bwd_mid_f32") else 0
            nact = 0 if os.environ.get("NOCF_ACT_REC", "1") in ("0", "") else int(
                L.nocf_activation_record_floats(int(d), int(net.m), int(net.nTh), int(n), int(nt), _STEPPERS[stepper]))
            act = None
            if nact:
                try:
                    act = torch.empty(nact, device=dev)
                except torch.OutOfMemoryError:                     # the record is an optimisation: without it the adjoint recomputes
                    act = None
            recorded = C.c_int32(0)
            rc = L.nocf_rollout_record_act_f32(
                                                    C.byref(phi_st), C.byref(prob_st), _lib.ptr(x), n,
                                                    float(tspan[0]), float(tspan[1]), int(nt), _STEPPERS[stepper], alph_c,
                                                    _lib.ptr(z_out), _lib.ptr(persample), _lib.ptr(sums), _lib.ptr(s_all),
                                                    _lib.ptr(act), C.byref(recorded),
                                                    _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, "nocf_rollout_record_act_f32")
        ctx.act = act if recorded.value else None
        return _OCflowTrain._forward_tail(ctx, L, dev, net, prob, tspan, nt, stepper, alph, n_total, group, n, sums, s_all, z_out)

    @staticmethod
    def _forward_tail(ctx, L, dev, net, prob, tspan, nt, stepper, alph, n_total, group, n, sums, s_all, z_out):
        _lib.track_rollout_status(L, dev, "OCflow (training forward)")
        if _lib.duo_guard(L, "OCflow (training forward)"):
            return None
        ctx.net, ctx.prob, ctx.tspan, ctx.nt, ctx.stepper, ctx.alph = net, prob, tspan, nt, stepper, list(alph)
        ctx.group = group
        if group is not None:
            from .distributed import reduce_cost_sums
            sums = reduce_cost_sums(sums, None if group is True else group)   # one RCCL all-reduce of 8 floats
            if not n_total:
                n_total = int(round(float(sums[7].item())))
        ctx.n_total = n_total or n
        ctx.save_for_backward(s_all, z_out)
        # the adjoint re-reads the weights from the module: they must still be the ones this forward ran with
        ctx.param_versions = [p._version for p in net.parameters()]
        means = sums[:7] / sums[7]
        Jc = means[0] + alph[0] * means[1] + alph[3] * means[2] + alph[4] * means[3] + alph[5] * means[4]
        return Jc, means.detach()

    @staticmethod
    @once_differentiable
    def backward(ctx, gJ, _gmeans):
        s_all, z_out = ctx.saved_tensors
        net, prob, nt, alph = ctx.net, ctx.prob, ctx.nt, ctx.alph
        if [p._version for p in net.parameters()] != ctx.param_versions:
            raise RuntimeError("OCflow backward: a parameter of Phi was modified in place (optimizer step, load_state_dict) between "
                               "this forward and its backward; the adjoint would run with weights that do not match the recorded "
                               "stage inputs.  Call backward() before changing the parameters.")
        dev = s_all.device
        n, d = z_out.shape[0], z_out.shape[1] - 4
        m, D1 = net.m, d + 1
        nstage = 4 if ctx.stepper == "rk4" else 1
        if getattr(ctx, "tape", None) is not None:
            out = _OCflowTrain._backward_tape(ctx, gJ, s_all, z_out, n, d, m, D1, nstage)
            if out is not None:
                return out
            s_all = s_all[:nt * nstage]                            # (the split-role adjoint did not qualify after all: per-tile adjoint)
        rows = (nt * nstage + 2) * n
        phi_st, keep1, ws = net._c_struct(n)
        prob_st, keep2 = prob._c_struct(dev)
        hs = _step_sizes(ctx.tspan, nt).to(dev)
        alph_c = (C.c_float * 6)(*[float(a) for a in alph[:6]])
        lam0 = torch.empty(n, d, device=dev) if ctx.x_needs_grad else None          # dJc/dx0, like autograd's x.grad
        lib = _lib.lib_for(net.d, net.m, net.nTh, phi_st.r, prob_st.n_agents)
        # small networks: the lane-style adjoint accumulates every gradient row in registers (one vector per sample)
        P = int(lib.nocf_small_grad_floats(d, m))
        gpart = torch.empty(n, P, device=dev) if (net.nTh == 2 and m <= 32 and D1 <= 32) else None
        if gpart is not None:
            with torch.cuda.device(dev):
                rc = lib.nocf_rollout_bwd_small_f32(C.byref(phi_st), C.byref(prob_st), n, int(nt), _STEPPERS[ctx.stepper],
                                                    float(ctx.tspan[1]), alph_c, 1.0 / float(ctx.n_total), _lib.ptr(s_all),
                                                    _lib.ptr(z_out), _lib.ptr(hs), _lib.ptr(gpart), _lib.ptr(lam0),
                                                    _lib.stream_ptr(dev))
            if rc == 0:
                return _OCflowTrain._finish(ctx, gJ, _unpack_partials(gpart, m, D1, net), lam0, net)
            if rc != -2:                                               # NOCF_E_SHAPE: not a lane-kernel shape -> row streams below
                _lib.check(rc, "nocf_rollout_bwd_small_f32")
        # medium two-layer networks: the one-CU adjoint accumulates the weight gradients in the kernel (one partial vector per workgroup);
        # with the forward's activation record it does not re-run grad Phi's forward sweep (NOCF_ACT_REC=0: it does)
        mid_rows = int(getattr(ctx, "mid_rows", 0))
        if mid_rows:
            gmid = torch.empty(mid_rows, P, device=dev)
            with torch.cuda.device(dev):
                rc = lib.nocf_rollout_bwd_mid_f32(C.byref(phi_st), C.byref(prob_st), n, int(nt), _STEPPERS[ctx.stepper],
                                                  float(ctx.tspan[1]), alph_c, 1.0 / float(ctx.n_total), _lib.ptr(s_all),
                                                  _lib.ptr(z_out), _lib.ptr(hs), _lib.ptr(getattr(ctx, "act", None)), _lib.ptr(gmid), mid_rows,
                                                  _lib.ptr(lam0),
                                                  _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
            if rc == 0:
                ctx.act = None
                return _OCflowTrain._finish(ctx, gJ, _unpack_partials(gmid, m, D1, net), lam0, net)
            if rc != -2:
                _lib.check(rc, "nocf_rollout_bwd_mid_f32")
        L = net.nTh - 1
        # every row the kernel does not write must be zero: the value block (last n rows) of Y / V / Ab / Gb
        Y, Ob, Wb = (torch.empty(rows, m, device=dev) for _ in range(3))
        V, Ab, Qb, U0 = (torch.empty(L, rows, m, device=dev) for _ in range(4))     # per residual layer
        Gb, Sx = torch.empty(rows, D1, device=dev), torch.empty(rows, D1, device=dev)
        for t in (Y, Gb):
            t[rows - n:].zero_()
        for t in (V, Ab):
            t[:, rows - n:].zero_()
        PHIb = torch.zeros(n, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib_for(net.d, net.m, net.nTh, phi_st.r, prob_st.n_agents).nocf_rollout_bwd_act_f32(
                                                 C.byref(phi_st), C.byref(prob_st), n, int(nt), _STEPPERS[ctx.stepper],
                                                 float(ctx.tspan[1]), alph_c, 1.0 / float(ctx.n_total),
                                                 _lib.ptr(s_all), _lib.ptr(z_out), _lib.ptr(hs),
                                                 _lib.ptr(Y), _lib.ptr(Ob), _lib.ptr(V), _lib.ptr(Ab), _lib.ptr(Qb),
                                                 _lib.ptr(U0), _lib.ptr(Wb), _lib.ptr(Gb), _lib.ptr(Sx),
                                                 _lib.ptr(PHIb), _lib.ptr(lam0), _lib.ptr(getattr(ctx, "act", None)),
                                                 _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, "nocf_rollout_bwd_act_f32")
        ctx.act = None
        sT = Sx[(nt * nstage + 1) * n:]                                     # s at the final time (value rows)
        grads = {"N.layers.0.weight": _contract(Ob, Sx, _contract(Y, Gb)), "N.layers.0.bias": _colsum(Ob)}
        for i in range(1, L + 1):
            grads[f"N.layers.{i}.weight"] = _contract(Qb[i - 1], U0[i - 1], _contract(V[i - 1], Ab[i - 1]))
            grads[f"N.layers.{i}.bias"] = _colsum(Qb[i - 1])
        grads["w.weight"] = _colsum(Wb).reshape(1, -1)
        grads["c.weight"] = (_colsum(Gb) + PHIb @ sT).reshape(1, -1)
        grads["c.bias"] = PHIb.sum().reshape(1)
        dM = _contract(Gb, Sx) + 0.5 * (sT * PHIb[:, None]).t() @ sT
        grads["A"] = net.A.detach() @ (dM + dM.t())
        return _OCflowTrain._finish(ctx, gJ, grads, lam0, net)

    @staticmethod
    def _backward_tape(ctx, gJ, s_all, z_out, n, d, m, D1, nstage):
        """the split-role adjoint on the forward's tape (include/nocf.h: nocf_rollout_bwd_tape_f32); None when it does not qualify"""
        net, prob, nt, alph, tape = ctx.net, ctx.prob, ctx.nt, ctx.alph, ctx.tape
        dev = s_all.device
        E = nt * nstage + 1
        R = E * n
        phi_st, keep1, ws = net._c_struct(n)
        prob_st, keep2 = prob._c_struct(dev)
        hs = _step_sizes(ctx.tspan, nt).to(dev)
        alph_c = (C.c_float * 6)(*[float(a) for a in alph[:6]])
        lam0 = torch.empty(n, d, device=dev) if ctx.x_needs_grad else None
        Y, Ab, Qb, Ob = (torch.empty(R, m, device=dev) for _ in range(4))
        Gb = torch.empty(R, D1, device=dev)
        lib = getattr(ctx, "tape_lib", None) or _lib.lib_for(net.d, net.m, net.nTh, phi_st.r, prob_st.n_agents, fwd=True)
        # the column sums (dw rows, qbar, obar -> dw, db1, db0) come from the kernel's epilogues: no Wb stream, no pass over Qb / Ob for the biases
        # (NOCF_DUO_CSUM=0: the dw rows are streamed and all three are summed afterwards, as up to round 4)
        csum_on = os.environ.get("NOCF_DUO_CSUM", "1") != "0" and hasattr(lib, "nocf_rollout_bwd_tape_sums_f32")
        Wb = None if csum_on else torch.empty(R, m, device=dev)
        cs = torch.empty(int(lib.nocf_bwd_colsum_floats(n)), device=dev) if csum_on else None
        # NOCF_DUO_DW=1: the two large weight gradients are accumulated in the kernel (weight-gradient roles, nocf_duo_bwd.inc); measured
        # slower than contracting the streams afterwards, so the default contracts them below
        dK1, dK0 = torch.empty(m, m, device=dev), torch.empty(m, D1, device=dev)
        nsc = int(lib.nocf_dw_scratch_floats())
        sc_dw = _SCRATCH.get(("dw", dev))
        if sc_dw is None or sc_dw.numel() < nsc:
            sc_dw = _SCRATCH[("dw", dev)] = torch.empty(max(nsc, 1), device=dev)
        dw_done = C.c_int32(0)
        with torch.cuda.device(dev):
            if csum_on:
                rc = lib.nocf_rollout_bwd_tape_sums_f32(C.byref(phi_st), C.byref(prob_st), n, int(nt), _STEPPERS[ctx.stepper], alph_c,
                                                        1.0 / float(ctx.n_total), _lib.ptr(s_all), _lib.ptr(z_out), _lib.ptr(hs), _lib.ptr(tape),
                                                        _lib.ptr(Y), _lib.ptr(Ab), _lib.ptr(Qb), _lib.ptr(Ob), _lib.ptr(Gb),
                                                        _lib.ptr(lam0), _lib.ptr(dK1), _lib.ptr(dK0), _lib.ptr(sc_dw), sc_dw.numel(), C.byref(dw_done),
                                                        _lib.ptr(cs), cs.numel(), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
            else:
                rc = lib.nocf_rollout_bwd_tape_f32(C.byref(phi_st), C.byref(prob_st), n, int(nt), _STEPPERS[ctx.stepper], alph_c,
                                                   1.0 / float(ctx.n_total), _lib.ptr(s_all), _lib.ptr(z_out), _lib.ptr(hs), _lib.ptr(tape),
                                                   _lib.ptr(Y), _lib.ptr(Ab), _lib.ptr(Wb), _lib.ptr(Qb), _lib.ptr(Ob), _lib.ptr(Gb),
                                                   _lib.ptr(lam0), _lib.ptr(dK1), _lib.ptr(dK0), _lib.ptr(sc_dw), sc_dw.numel(), C.byref(dw_done),
                                                   _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        if rc == -2:
            return None
        _lib.check(rc, "nocf_rollout_bwd_tape_f32")
        _lib.track_rollout_status(lib, dev, "OCflow (backward)")
        U0, TH1 = tape[:R * m].view(R, m), tape[2 * R * m:3 * R * m].view(R, m)
        gpad = (R * D1 + 3) // 4 * 4
        sc = tape[4 * R * m + gpad + n * m:].view(R, 4)
        Sx = s_all.view(R, D1)
        sT = Sx[R - n:]
        phib = torch.sign(sc[R - n:, 0]) * (float(alph[4]) / float(ctx.n_total))      # cotangent of Phi(z(T), T) (HJfin, src/OCflow.py:70-76)
        w = net.w.weight.detach().reshape(-1, 1)
        # the value's rows (weight gradients only: the state's share of this cotangent is already in the terminal lambda) ride on the
        # terminal block: qbar += phib v, obar += phib y, dw row += phib u_1
        u1 = tape[4 * R * m + gpad:4 * R * m + gpad + n * m].view(n, m)
        vT = TH1[R - n:] * w.t()
        if csum_on:                                               # kernel sums (without the value's rows) + the value's rows of the terminal block
            S = cs.view(-1, 3, m).sum(0)
            gw, gb1, gb0 = S[0] + phib @ u1, S[1] + phib @ vT, S[2] + phib @ Y[R - n:]
        Qb[R - n:].addcmul_(vT, phib[:, None])
        Ob[R - n:].addcmul_(Y[R - n:], phib[:, None])
        if not csum_on:
            Wb[R - n:].addcmul_(u1, phib[:, None])
            gw, gb1, gb0 = _colsum(Wb), _colsum(Qb), _colsum(Ob)
        if dw_done.value:                                         # (the kernel's sums include the value's rows)
            gK0, gK1 = dK0, dK1
        else:
            gK0, gK1 = _contract(Ob, Sx, _contract(Y, Gb)), _contract(Qb, U0, w * _contract(TH1, Ab))
        grads = {"N.layers.0.weight": gK0, "N.layers.0.bias": gb0,
                 "N.layers.1.weight": gK1, "N.layers.1.bias": gb1,
                 "w.weight": gw.reshape(1, -1),
                 "c.weight": (_colsum(Gb) + phib @ sT).reshape(1, -1), "c.bias": phib.sum().reshape(1)}
        dM = _contract(Gb, Sx) + 0.5 * (sT * phib[:, None]).t() @ sT
        grads["A"] = net.A.detach() @ (dM + dM.t())
        ctx.tape = None
        # a timed-out exchange leaves garbage rows: turn every gradient into NaN on the stream (the host raises at its next check)
        flat = [grads[name] for name, _ in net.named_parameters()]
        with torch.cuda.device(dev):
            for g_ in flat + ([lam0] if lam0 is not None else []):
                lib.nocf_poison_if_failed_f32(_lib.ptr(g_), g_.numel(), _lib.stream_ptr(dev))
        return _OCflowTrain._finish(ctx, gJ, grads, lam0, net)

    @staticmethod
    def _finish(ctx, gJ, grads, lam0, net):
        out = [gJ * grads[name] for name, _ in net.named_parameters()]
        if ctx.group is not None:
            from .distributed import allreduce_flat
            out = allreduce_flat(out, None if ctx.group is True else ctx.group)   # one all-reduce of all gradients
        gx = gJ * lam0 if lam0 is not None else None
        return (gx,) + (None,) * 8 + tuple(out)


class _OCflowTrain64(torch.autograd.Function):
    """the same in double precision (trainOC.py --prec double): nocf_rollout_record_f64 + nocf_rollout_bwd_f64 (csrc/nocf_f64_bwd.inc), the
    streamed rows contracted with double GEMMs"""

    @staticmethod
    def forward(ctx, x, net, prob, tspan, nt, stepper, alph, n_total, group, *params):
        ctx.x_needs_grad = bool(x.requires_grad)
        x = _lib.require_device_f64(x.detach(), "x")
        if x.dim() != 2:
            raise ValueError("x must be nex-by-d")
        n, d = x.shape
        if d != net.d:
            raise ValueError(f"x has d={d} but Phi was built for d={net.d}")
        if len(alph) < 6:
            raise ValueError("alph needs 6 entries")
        dev = x.device
        phi_st, keep1, ws = net._c_struct64()
        prob_st, keep2 = prob._c_struct64(dev)
        nstage = 4 if stepper == "rk4" else 1
        persample = torch.empty(n, 7, dtype=torch.float64, device=dev)
        sums = torch.empty(8, dtype=torch.float64, device=dev)
        z_out = torch.empty(n, d + 4, dtype=torch.float64, device=dev)
        s_all = torch.empty(nt * nstage, n, d + 1, dtype=torch.float64, device=dev)
        alph_c = (C.c_double * 6)(*[float(a) for a in alph[:6]])
        with torch.cuda.device(dev):
            rc = _lib.lib().nocf_rollout_record_f64(C.byref(phi_st), C.byref(prob_st), _lib.ptr(x), n, float(tspan[0]), float(tspan[1]), int(nt),
                                                    _STEPPERS[stepper], alph_c, _lib.ptr(z_out), _lib.ptr(persample), _lib.ptr(sums), _lib.ptr(s_all),
                                                    _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, "nocf_rollout_record_f64")
        ctx.net, ctx.prob, ctx.tspan, ctx.nt, ctx.stepper, ctx.alph, ctx.group = net, prob, tspan, nt, stepper, list(alph), group
        if group is not None:
            from .distributed import reduce_cost_sums
            sums = reduce_cost_sums(sums, None if group is True else group)
            if not n_total:
                n_total = int(round(float(sums[7].item())))
        ctx.n_total = n_total or n
        ctx.save_for_backward(s_all, z_out)
        ctx.param_versions = [p._version for p in net.parameters()]
        means = sums[:7] / sums[7]
        Jc = means[0] + alph[0] * means[1] + alph[3] * means[2] + alph[4] * means[3] + alph[5] * means[4]
        return Jc, means.detach()

    @staticmethod
    @once_differentiable
    def backward(ctx, gJ, _gmeans):
        s_all, z_out = ctx.saved_tensors
        net, prob, nt, alph = ctx.net, ctx.prob, ctx.nt, ctx.alph
        if [p._version for p in net.parameters()] != ctx.param_versions:
            raise RuntimeError("OCflow backward: a parameter of Phi was modified in place between this forward and its backward")
        dev = s_all.device
        n, d = z_out.shape[0], z_out.shape[1] - 4
        m, D1, L = net.m, d + 1, net.nTh - 1
        nstage = 4 if ctx.stepper == "rk4" else 1
        rows = (nt * nstage + 2) * n
        phi_st, keep1, ws = net._c_struct64()
        prob_st, keep2 = prob._c_struct64(dev)
        h = (float(ctx.tspan[1]) - float(ctx.tspan[0])) / nt
        tk, hs_l = float(ctx.tspan[0]), []
        for _ in range(nt):
            hs_l.append((tk + h) - tk)
            tk += h
        hs = torch.tensor(hs_l, dtype=torch.float64, device=dev)
        alph_c = (C.c_double * 6)(*[float(a) for a in alph[:6]])
        f64 = dict(dtype=torch.float64, device=dev)
        lam0 = torch.empty(n, d, **f64) if ctx.x_needs_grad else None
        Y, Ob, Wb = (torch.empty(rows, m, **f64) for _ in range(3))
        V, Ab, Qb, U0 = (torch.empty(L, rows, m, **f64) for _ in range(4))
        Gb, Sx = torch.empty(rows, D1, **f64), torch.empty(rows, D1, **f64)
        for t in (Y, Gb):
            t[rows - n:].zero_()
        for t in (V, Ab):
            t[:, rows - n:].zero_()
        PHIb = torch.zeros(n, **f64)
        with torch.cuda.device(dev):
            rc = _lib.lib().nocf_rollout_bwd_f64(C.byref(phi_st), C.byref(prob_st), n, int(nt), _STEPPERS[ctx.stepper], float(ctx.tspan[1]), alph_c,
                                                 1.0 / float(ctx.n_total), _lib.ptr(s_all), _lib.ptr(z_out), _lib.ptr(hs),
                                                 _lib.ptr(Y), _lib.ptr(Ob), _lib.ptr(V), _lib.ptr(Ab), _lib.ptr(Qb), _lib.ptr(U0), _lib.ptr(Wb),
                                                 _lib.ptr(Gb), _lib.ptr(Sx), _lib.ptr(PHIb), _lib.ptr(lam0), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, "nocf_rollout_bwd_f64")
        sT = Sx[(nt * nstage + 1) * n:]
        # (the row-slab form of _contract: one GEMM with a small output and ~10^6 rows runs on a handful of workgroups)
        grads = {"N.layers.0.weight": _contract(Ob, Sx, _contract(Y, Gb)), "N.layers.0.bias": Ob.sum(0)}
        for i in range(1, L + 1):
            grads[f"N.layers.{i}.weight"] = _contract(Qb[i - 1], U0[i - 1], _contract(V[i - 1], Ab[i - 1]))
            grads[f"N.layers.{i}.bias"] = Qb[i - 1].sum(0)
        grads["w.weight"] = Wb.sum(0).reshape(1, -1)
        grads["c.weight"] = (Gb.sum(0) + PHIb @ sT).reshape(1, -1)
        grads["c.bias"] = PHIb.sum().reshape(1)
        dM = _contract(Gb, Sx) + 0.5 * (sT * PHIb[:, None]).t() @ sT
        grads["A"] = net.A.detach() @ (dM + dM.t())
        return _OCflowTrain._finish(ctx, gJ, grads, lam0, net)


def ocflow_train(x, net, prob, tspan, nt, stepper, alph, n_total=None, group=None):
    """(Jc, cs) with Jc differentiable w.r.t. the parameters of `net`.  n_total: global batch size when x is one
    shard of it (the means of src/OCflow.py:80-86 run over all samples).  group: a torch.distributed process group
    (or True for the default one): the 8 cost sums are all-reduced in the forward, and the backward all-reduces the
    parameter gradients as ONE flat buffer (<= 1.37 MB for swarm50), so Jc and .grad are the global-batch values on
    every rank."""
    if stepper not in _STEPPERS:
        raise ValueError(f"stepper must be 'rk4' or 'rk1', got {stepper!r}")
    params = [p for _, p in net.named_parameters()]
    fn = _OCflowTrain64 if x.dtype == torch.float64 else _OCflowTrain
    Jc, means = fn.apply(x, net, prob, list(tspan), int(nt), stepper, list(alph), n_total, group, *params)
    return Jc, [means[i] for i in range(7)]
