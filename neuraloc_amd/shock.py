"""Shocked rollouts: integrate to t_s, displace the state, integrate on.

Same two-segment construction as the reference's only shock code path (src/plotter.py:815-824, driven by
evalOC.py:113-122, softcorridor only there): nShock = int(t_s*nt) steps on [t0, t_s], then 1+nt-nShock steps on
[t_s, t1] from the shocked state.  Here it works for every problem class and for a whole batch of states and
shocks at once (the BASELINE "singlequad shock-eval sweep"); both segments are the fused HIP rollout."""
import torch

from importlib import import_module

_oc = import_module(__package__ + ".OCflow")           # the module, not the function the package re-exports under the same name
from .OCflow import costs_from_sums


def shock_rollout(x, Phi, prob, nt, t_s, shock, tspan=(0.0, 1.0), alph=None, stepper="rk4", group=None, gather=False):
    """
    :param x:     nex-by-d initial states on the MI355X
    :param t_s:   shock time, tspan[0] < t_s < tspan[1], with int(t_s*nt) >= 1
    :param shock: 1-by-d or nex-by-d displacement added to the state at t_s
    :return: dict with
        traj      nex-by-d-by-(nt+3): states of segment 1 (nShock+1 columns) then segment 2 (nt-nShock+2 columns),
                  the layout the reference concatenates (the shocked state follows the unshocked one at t_s)
        ctrl      the same concatenation for the controls
        costs1, costs2   (Jc, cs) of the two segments
        nShock
    Sharded sweeps (BASELINE config 5): with `group` (a torch.distributed process group, or True for the default
    one) x holds THIS RANK's rows of the batch (a nex-by-d `shock` too); the costs are the global-batch means (one all-reduce of
    8 floats per segment) and traj / ctrl stay sharded by rows unless gather=True all-gathers them onto every rank.
    """
    alph = list(Phi.alph if alph is None else alph)
    d = x.shape[1]
    nShock = int(t_s * nt)
    if nShock < 1 or nShock > nt:
        # the reference would divide by zero (nShock = 0) or integrate backwards; SURVEY 8a note 10
        raise ValueError(f"shock time {t_s} gives nShock={nShock}; need 1 <= int(t_s*nt) <= nt")
    shock = torch.as_tensor(shock, dtype=x.dtype, device=x.device)
    # one launch per segment gives the trajectory, the controls AND the cost sums (the reference runs OCflow twice)
    with torch.no_grad():
        grp = None if group is True else group
        if group is not None:
            from .distributed import gather_rows, reduce_cost_sums
        _, sums1, zF1, cF1 = _oc._launch(x, Phi, prob, [tspan[0], t_s], nShock, stepper, alph, True)
        z1, c1 = zF1.permute(1, 2, 0), cF1.permute(1, 2, 0)
        if group is not None:
            sums1 = reduce_cost_sums(sums1, grp)
        costs1 = costs_from_sums(sums1, alph)
        xs = (z1[:, :d, -1] + shock).contiguous()
        n2 = 1 + nt - nShock
        _, sums2, zF2, cF2 = _oc._launch(xs, Phi, prob, [t_s, tspan[1]], n2, stepper, alph, True)
        z2, c2 = zF2.permute(1, 2, 0), cF2.permute(1, 2, 0)
        if group is not None:
            sums2 = reduce_cost_sums(sums2, grp)
        costs2 = costs_from_sums(sums2, alph)
        traj, ctrl = torch.cat((z1[:, :d, :], z2[:, :d, :]), dim=2), torch.cat((c1, c2), dim=2)
        if group is not None and gather:
            traj, ctrl, xs = gather_rows(traj.contiguous(), grp), gather_rows(ctrl.contiguous(), grp), gather_rows(xs, grp)
    return {"traj": traj, "ctrl": ctrl, "costs1": costs1, "costs2": costs2, "nShock": nShock, "x_shocked": xs}


def shock_sweep(x, Phi, prob, nt, shock_times, shocks, **kw):
    """every (t_s, shock) combination; shocks: k-by-d.  Returns a list of shock_rollout dicts (batched over x)."""
    out = []
    for t_s in shock_times:
        for k in range(shocks.shape[0]):
            res = shock_rollout(x, Phi, prob, nt, float(t_s), shocks[k:k + 1], **kw)
            res["t_s"], res["shock_index"] = float(t_s), k
            out.append(res)
    return out
