"""OCflow -- drop-in for src/OCflow.py:7-95 on an MI355X.

Same call signature and return values as the reference:

    Jc, cs = OCflow(x, Phi, prob, tspan, nt, stepper="rk4", alph=[...], intermediates=False, noMean=False)

but the whole rollout (nt RK steps x stages, each: grad Phi + problem physics + running
costs, then the terminal costs) is one HIP launch (nocf_rollout_f32).  This module only
marshals arguments and forms the 7 means / Jc from the 8 sums the kernel returns.
There is no CPU or eager-torch fallback.
"""
import ctypes as C

import torch

from . import _lib

_STEPPERS = {"rk4": _lib.NOCF_RK4, "rk1": _lib.NOCF_RK1}


def ocG(z, xtarget):
    """terminal residual z[:, :d] - xtarget (src/OCflow.py:97-101)"""
    d = xtarget.shape[0]
    return z[:, 0:d] - xtarget


def _launch64(x, Phi, prob, tspan, nt, stepper, alph, intermediates):
    """the double-precision rollout (nocf_rollout_f64): the reference's --prec double (evalOC.py:19,28-31)"""
    x = _lib.require_device_f64(x, "x")
    if x.dim() != 2:
        raise ValueError("x must be nex-by-d")
    n, d = x.shape
    if d != Phi.d:
        raise ValueError(f"x has d={d} but Phi was built for d={Phi.d}")
    if int(nt) < 1:
        raise ValueError("nt must be >= 1")
    if stepper not in _STEPPERS:
        raise ValueError(f"stepper must be 'rk4' or 'rk1', got {stepper!r}")
    if len(alph) < 6:
        raise ValueError("alph needs 6 entries")
    # (only the mean path is differentiable -- OCflow routes it to train._OCflowTrain64 before it gets here; noMean / intermediates under
    # autograd raise like the single-precision launch does, instead of returning detached tensors)
    Phi._guard_no_autograd(x, "OCflow")
    phi_st, keep1, ws = Phi._c_struct64()
    prob_st, keep2 = prob._c_struct64(x.device)
    dev = x.device
    persample = torch.empty(n, 7, dtype=torch.float64, device=dev)
    sums = torch.empty(8, dtype=torch.float64, device=dev)
    zFull = ctrlFull = None
    if intermediates:
        p32, _k = prob._c_struct(dev)
        cdim = _lib.lib().nocf_ctrl_dim(C.byref(p32), d)
        zFull = torch.empty(nt + 1, n, d + 4, dtype=torch.float64, device=dev)
        ctrlFull = torch.empty(nt + 1, n, cdim, dtype=torch.float64, device=dev)
    alph_c = (C.c_double * 6)(*[float(a) for a in alph[:6]])
    with torch.cuda.device(dev):
        rc = _lib.lib().nocf_rollout_f64(C.byref(phi_st), C.byref(prob_st), _lib.ptr(x), n,
                                         float(tspan[0]), float(tspan[1]), int(nt), _STEPPERS[stepper], alph_c,
                                         None, _lib.ptr(persample), _lib.ptr(sums), _lib.ptr(zFull), _lib.ptr(ctrlFull),
                                         _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "nocf_rollout_f64")
    return persample, sums, zFull, ctrlFull


def _launch(x, Phi, prob, tspan, nt, stepper, alph, intermediates, means=None):
    """run the HIP rollout; returns (persample [n,7], sums [8], zFull_tm, ctrlFull_tm).  means: an 8-float device tensor that the same
    launch sequence fills with the 7 batch means and Jc (nocf_rollout_means_f32: for callers without an all-reduce behind the sums)"""
    if isinstance(x, torch.Tensor) and x.dtype == torch.float64:
        return _launch64(x, Phi, prob, tspan, nt, stepper, alph, intermediates)
    x = _lib.require_device_f32(x, "x")
    if x.dim() != 2:
        raise ValueError("x must be nex-by-d")
    n, d = x.shape
    if d != Phi.d:
        raise ValueError(f"x has d={d} but Phi was built for d={Phi.d}")
    if int(nt) < 1:
        # the reference divides by zero here (src/OCflow.py:25); SURVEY 8a note 10
        raise ValueError("nt must be >= 1")
    if stepper not in _STEPPERS:
        # the reference silently integrates nothing for an unknown stepper (src/OCflow.py:46-49)
        raise ValueError(f"stepper must be 'rk4' or 'rk1', got {stepper!r}")
    if len(alph) < 6:
        raise ValueError("alph needs 6 entries")
    Phi._guard_no_autograd(x, "OCflow")
    _lib.check_errors()                                    # a failed earlier rollout whose status has arrived raises here
    phi_st, keep1, ws = Phi._c_struct(n)
    prob_st, keep2 = prob._c_struct(x.device)
    dev = x.device
    persample = torch.empty(n, 7, dtype=torch.float32, device=dev)
    sums = torch.empty(8, dtype=torch.float32, device=dev)
    zFull = ctrlFull = None
    if intermediates:
        cdim = _lib.lib().nocf_ctrl_dim(C.byref(prob_st), d)
        zFull = torch.empty(nt + 1, n, d + 4, dtype=torch.float32, device=dev)
        ctrlFull = torch.empty(nt + 1, n, cdim, dtype=torch.float32, device=dev)
    alph_c = (C.c_float * 6)(*[float(a) for a in alph[:6]])
    with torch.cuda.device(dev):
        L = _lib.lib_for(Phi.d, Phi.m, Phi.nTh, phi_st.r, prob_st.n_agents, fwd=prob_st.kind != _lib.PROB_QUADCOPTER)
        if means is not None and hasattr(L, "nocf_rollout_means_f32"):
            rc = L.nocf_rollout_means_f32(C.byref(phi_st), C.byref(prob_st), _lib.ptr(x), n,
                                          float(tspan[0]), float(tspan[1]), int(nt), _STEPPERS[stepper], alph_c,
                                          None, _lib.ptr(persample), _lib.ptr(sums), _lib.ptr(means),
                                          _lib.ptr(zFull), _lib.ptr(ctrlFull),
                                          _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        else:
            rc = L.nocf_rollout_f32(C.byref(phi_st), C.byref(prob_st), _lib.ptr(x), n,
                                    float(tspan[0]), float(tspan[1]), int(nt), _STEPPERS[stepper], alph_c,
                                    None, _lib.ptr(persample), _lib.ptr(sums),
                                    _lib.ptr(zFull), _lib.ptr(ctrlFull),
                                    _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
            if means is not None and rc == 0:
                # a per-shape library cached before nocf_rollout_means_f32 existed: the library that RAN fills `means` with its own
                # nocf_cost_means_f32 (every build has it), so the caller never sees an unwritten buffer
                rc = L.nocf_cost_means_f32(_lib.ptr(sums), alph_c, _lib.ptr(means), _lib.stream_ptr(dev))
    _lib.check(rc, "nocf_rollout_f32")
    _lib.track_rollout_status(L, dev, "OCflow")
    if _lib.duo_guard(L, "OCflow"):
        return _launch(x, Phi, prob, tspan, nt, stepper, alph, intermediates, means)
    return persample, sums, zFull, ctrlFull


MAX_SEGMENTS = 16


def segments_supported(x, Phi, prob):
    """whether nocf_rollout_segments_f32 has a kernel for this network / problem (host-side query: nothing is launched or allocated)"""
    phi_st, keep1, ws = Phi._c_struct(x.shape[0])
    prob_st, keep2 = prob._c_struct(x.device)
    L = _lib.lib_for(Phi.d, Phi.m, Phi.nTh, phi_st.r, prob_st.n_agents, fwd=prob_st.kind != _lib.PROB_QUADCOPTER)
    if not hasattr(L, "nocf_rollout_segments_f32"):
        return False
    if not hasattr(L, "nocf_segments_supported"):               # (a per-shape library cached by an older build: the launch itself answers)
        return True
    return bool(L.nocf_segments_supported(C.byref(phi_st), C.byref(prob_st)))


def _launch_segments(x, Phi, prob, t0s, t1, nts, rows_per_seg, stepper, alph, slot0s=None, zFull=None, ctrlFull=None):
    """Several rollouts that differ only in start time and step count in ONE launch (nocf_rollout_segments_f32): rows
    [k * rows_per_seg, (k + 1) * rows_per_seg) of x are integrated over [t0s[k], t1] with nts[k] steps, trajectories and controls kept:
    segment k writes the time slots slot0s[k] .. slot0s[k] + nts[k] of zFull / ctrlFull (time-major [slots, n, .]; allocated here when not
    given).  Returns (persample [n,7], sums [nseg,8], zFull_tm, ctrlFull_tm), or None when this network / problem has no kernel that takes
    segments (the caller then launches them one by one)."""
    x = _lib.require_device_f32(x, "x")
    n, d = x.shape
    nseg = len(t0s)
    if not (1 <= nseg <= MAX_SEGMENTS) or len(nts) != nseg or rows_per_seg % 16 or not ((nseg - 1) * rows_per_seg < n <= nseg * rows_per_seg):
        return None
    if stepper not in _STEPPERS:
        raise ValueError(f"stepper must be 'rk4' or 'rk1', got {stepper!r}")
    if min(int(v) for v in nts) < 1:
        raise ValueError("nt must be >= 1")
    slot0s = [0] * nseg if slot0s is None else [int(v) for v in slot0s]
    Phi._guard_no_autograd(x, "shock sweep")
    _lib.check_errors()
    phi_st, keep1, ws = Phi._c_struct(n)
    prob_st, keep2 = prob._c_struct(x.device)
    dev = x.device
    L = _lib.lib_for(Phi.d, Phi.m, Phi.nTh, phi_st.r, prob_st.n_agents, fwd=prob_st.kind != _lib.PROB_QUADCOPTER)
    if not hasattr(L, "nocf_rollout_segments_f32"):
        return None
    slots = max(s0 + int(v) for s0, v in zip(slot0s, nts)) + 1
    persample = torch.empty(n, 7, dtype=torch.float32, device=dev)
    sums = torch.empty(nseg, 8, dtype=torch.float32, device=dev)
    cdim = _lib.lib().nocf_ctrl_dim(C.byref(prob_st), d)
    if zFull is None:
        zFull = torch.empty(slots, n, d + 4, dtype=torch.float32, device=dev)
        ctrlFull = torch.empty(slots, n, cdim, dtype=torch.float32, device=dev)
    if (tuple(zFull.shape) != (zFull.shape[0], n, d + 4) or tuple(ctrlFull.shape) != (zFull.shape[0], n, cdim) or zFull.shape[0] < slots
            or not zFull.is_contiguous() or not ctrlFull.is_contiguous()):
        raise ValueError("zFull / ctrlFull must be contiguous [slots, n, d+4] / [slots, n, a] with enough slots")
    alph_c = (C.c_float * 6)(*[float(a) for a in alph[:6]])
    t0_c = (C.c_double * nseg)(*[float(v) for v in t0s])
    nt_c = (C.c_int32 * nseg)(*[int(v) for v in nts])
    s0_c = (C.c_int32 * nseg)(*slot0s)
    with torch.cuda.device(dev):
        rc = L.nocf_rollout_segments_f32(C.byref(phi_st), C.byref(prob_st), _lib.ptr(x), n, nseg, int(rows_per_seg), t0_c, float(t1), nt_c, s0_c,
                                         _STEPPERS[stepper], alph_c, None, _lib.ptr(persample), _lib.ptr(sums),
                                         _lib.ptr(zFull), _lib.ptr(ctrlFull), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    if rc == -2:                                           # NOCF_E_SHAPE: no kernel with segments for this shape
        return None
    _lib.check(rc, "nocf_rollout_segments_f32")
    _lib.track_rollout_status(L, dev, "shock sweep")
    return persample, sums, zFull, ctrlFull


def costs_from_sums(sums, alph):
    """means of the 7 cost terms and Jc from the kernel's 8 sums (7 column sums + count).
    cs order is [L, G, HJt, HJfin, HJgrad, Q, W]; G is un-weighted (src/OCflow.py:80-90).
    On the device this is one tiny launch (nocf_cost_means_f32); host tensors (the gloo tests' injected sums) take
    the same formula in torch."""
    if sums.is_cuda and sums.dtype == torch.float32:
        out = torch.empty(8, dtype=torch.float32, device=sums.device)
        alph_c = (C.c_float * 6)(*[float(a) for a in alph[:6]])
        with torch.cuda.device(sums.device):
            rc = _lib.lib().nocf_cost_means_f32(_lib.ptr(sums), alph_c, _lib.ptr(out), _lib.stream_ptr(sums.device))
        _lib.check(rc, "nocf_cost_means_f32")
        return out[7], [out[i] for i in range(7)]
    means = sums[:7] / sums[7]
    cs = [means[i] for i in range(7)]
    Jc = cs[0] + alph[0] * cs[1] + alph[3] * cs[2] + alph[4] * cs[3] + alph[5] * cs[4]
    return Jc, cs


def OCflow(x, Phi, prob, tspan, nt, stepper="rk4", alph=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0],
           intermediates=False, noMean=False):
    """
    :param x:       nex-by-d tensor on the MI355X, fp32 (or float64 together with Phi and prob: the reference's --prec double;
                    differentiable in both precisions)
    :param Phi:     neuraloc_amd.Phi
    :param prob:    neuraloc_amd problem object (Cross2D / SwarmTraj / Quadcopter)
    :param tspan:   [t0, t1]
    :param nt:      number of time steps
    :param stepper: "rk4" or "rk1"
    :param alph:    6 multipliers [G, Q, W, HJt, HJfin, HJgrad]; entries 0,3,4,5 are used here
    :param intermediates: return (zFull [nex,d+4,nt+1], ctrlFull [nex,a,nt+1]) instead
    :param noMean:  return per-sample nex-by-1 costs instead of means (wins over intermediates,
                    like the reference: src/OCflow.py:66-76)
    :return: (Jc, cs)  or  (zFull, ctrlFull)
    """
    if (torch.is_grad_enabled() and not intermediates and not noMean
            and (x.requires_grad or any(p.requires_grad for p in Phi.parameters()))):
        from .train import ocflow_train                 # trainOC.py:172-173: Jc.backward() -> hand-written adjoint
        if int(nt) < 1:
            raise ValueError("nt must be >= 1")
        return ocflow_train(x, Phi, prob, tspan, nt, stepper, alph)
    want_means = not (noMean or intermediates) and isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32
    means = torch.empty(8, dtype=torch.float32, device=x.device) if want_means else None
    if means is not None:
        persample, sums, zF, cF = _launch(x, Phi, prob, tspan, nt, stepper, alph, False, means)
        # (the means came with the launch that reduced the per-sample table -- _launch fills them whichever library ran)
        return means[7], [means[i] for i in range(7)]
    persample, sums, zF, cF = _launch(x, Phi, prob, tspan, nt, stepper, alph, intermediates and not noMean)
    if noMean or intermediates:
        # results that are consumed on the host (plots, files): a timed-out exchange must raise HERE, not at the next call
        _lib.check_errors(sync=True)
    if noMean:
        cs = [persample[:, i:i + 1] for i in range(7)]
        Jc = cs[0] + alph[0] * cs[1] + alph[3] * cs[2] + alph[4] * cs[3] + alph[5] * cs[4]
        return Jc, cs
    if intermediates:
        # kernel layout is time-major [nt+1, nex, .]; the reference's is [nex, ., nt+1]
        return zF.permute(1, 2, 0), cF.permute(1, 2, 0)
    return costs_from_sums(sums, alph)
