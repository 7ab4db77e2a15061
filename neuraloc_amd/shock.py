"""Shocked rollouts: integrate to t_s, displace the state, integrate on.

Same two-segment construction as the reference's only shock code path (src/plotter.py:815-824, driven by
evalOC.py:113-122, softcorridor only there): nShock = int(t_s*nt) steps on [t0, t_s], then 1+nt-nShock steps on
[t_s, t1] from the shocked state.  Here it works for every problem class and for a whole batch of states and
shocks at once (the BASELINE "singlequad shock-eval sweep"); both segments are the fused HIP rollout."""
import torch

from importlib import import_module

_oc = import_module(__package__ + ".OCflow")           # the module, not the function the package re-exports under the same name
from .OCflow import costs_from_sums
from . import _lib


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


def _on_shared_grid(tspan, nt, times):
    """Which shock times lie on ONE step grid from tspan[0]?  Segment 1 of a shocked rollout takes nShock = int(t_s nt) steps of
    (t_s - t0) / nShock; for t_s = 0.1, 0.2, ... and nt = 50 that is the same h = 0.02 for every t_s to the last bit or two, so the unshocked
    trajectory up to the LARGEST t_s contains every shorter segment 1.  Returns (T, N, h, [nShock_k or None per time])."""
    t0 = float(tspan[0])
    ok = [(float(t), int(float(t) * nt)) for t in times]
    cand = [(t, k) for t, k in ok if 1 <= k <= nt]
    if not cand:
        return None, 0, 0.0, [None] * len(ok)
    T, N = max(cand)
    h = (T - t0) / N
    on = [k if (1 <= k <= nt and abs(t0 + k * h - t) <= 4e-16 * k * max(1.0, abs(t))) else None for t, k in ok]
    return T, N, h, on


def shock_sweep(x, Phi, prob, nt, shock_times, shocks, tspan=(0.0, 1.0), alph=None, stepper="rk4", group=None, gather=False, shared=True):
    """Every (t_s, shock) combination; shocks: k-by-d.  Returns a list of shock_rollout dicts (batched over x), t_s-major.

    BASELINE config 5 (the shock-eval sweep) as the reference's code path would run it (src/plotter.py:815-824 once per t_s) integrates the
    same unshocked trajectory from t0 again for every shock time.  Here (round 5, single precision on one rank):
      * the unshocked trajectory is integrated ONCE, to the largest shock time, with trajectories and controls kept; the segment 1 of every
        t_s that lies on that step grid (_on_shared_grid: all of 0.1 ... 0.9 at nt = 50) is its prefix -- equal to the per-t_s segment to
        the state tolerance (their step sizes differ in the last bit; tests/test_hip_parity.py compares) -- 459 -> 279 steps;
      * the second segments of ALL (t_s, shock) pairs run as ONE launch over (pair, row) tiles (nocf_rollout_segments_f32: per-segment start
        time, step count and first output slot), so 9 x 512 rows fill the chip where one 512-row segment occupies an eighth of it, and the
        kernel writes each pair's shocked trajectory straight behind its copy of the unshocked prefix: `traj` / `ctrl` are views, not cats;
      * the costs of every segment 1 (running integrals from the trajectory's cost columns, terminal terms from one batched Phi / grad Phi
        call at (z(t_s), t_s): src/OCflow.py:58-90) and of every segment 2 are formed for all pairs at once.
    Row-sharded sweeps (`group`: x holds this rank's rows): every rank runs ALL (t_s, shock) pairs of its rows this way -- at 512 rows per
    rank the second-segment launch still has 9 x 32 tiles -- and the cost sums of all first / all second segments are all-reduced as two
    small tensors (plus one flag: the ranks agree on the path before any of them reduces); trajectories stay sharded by rows.
    Shock times off the grid, double precision, gather=True and networks without a segment-taking kernel run shock_rollout per pair, as before
    (shared=False forces that)."""
    alph = list(Phi.alph if alph is None else alph)
    times = [float(t) for t in shock_times]
    K = int(shocks.shape[0])
    T, N, h, on = _on_shared_grid(tspan, nt, times)
    # (every term of `use` is the same on all ranks of a sharded sweep, except the row count: that one goes through _sweep_shared's flag)
    use = (shared and not (group is not None and gather) and x.dtype == torch.float32 and len(times) * K >= 2
           and any(k is not None for k in on))
    fast = {}
    if use:
        fast = _sweep_shared(x, Phi, prob, nt, [(i, t, on[i]) for i, t in enumerate(times) if on[i] is not None], shocks, T, N,
                             tspan, alph, stepper, group) or {}
    out = []
    for i, t_s in enumerate(times):
        for k in range(K):
            res = fast.get((i, k))
            if res is None:
                res = shock_rollout(x, Phi, prob, nt, t_s, shocks[k:k + 1], tspan=tspan, alph=alph, stepper=stepper, group=group, gather=gather)
            res["t_s"], res["shock_index"] = t_s, k
            out.append(res)
    return out


def _means(sums, alph):
    """[P, 8] sums -> list of (Jc, cs) per row: the formula of costs_from_sums for many segments at once (a handful of launches in all)"""
    means = sums[:, :7] / sums[:, 7:8]
    Jc = means[:, 0] + alph[0] * means[:, 1] + alph[3] * means[:, 2] + alph[4] * means[:, 3] + alph[5] * means[:, 4]
    return [(Jc[p], [means[p, j] for j in range(7)]) for p in range(sums.shape[0])]


def _sweep_shared(x, Phi, prob, nt, on_grid, shocks, T, N, tspan, alph, stepper, group=None):
    """the shared-prefix sweep of shock_sweep; returns {(time index, shock index): result dict} or None (no segment kernel for this shape, or --
    sharded -- some rank could not take this path: all ranks then return None together)"""
    n, d = x.shape
    K = int(shocks.shape[0])
    shocks = torch.as_tensor(shocks, dtype=x.dtype, device=x.device)
    pairs = [(i, t, ns, k) for (i, t, ns) in on_grid for k in range(K)]
    grp = None if group is True else group
    with torch.no_grad():
        ok = n % 16 == 0 and _oc.segments_supported(x, Phi, prob)     # asked BEFORE the unshocked rollout and the pairs' buffers exist
        raw = []
        if ok:
            # (1) the unshocked trajectory, once
            _, _, zF, cF = _oc._launch(x, Phi, prob, [tspan[0], T], N, stepper, alph, True)           # [N+1, n, d+4], [N+1, n, a]
            cdim = cF.shape[2]
            # (2) all second segments: MAX_SEGMENTS pairs per launch
            for c0 in range(0, len(pairs), _oc.MAX_SEGMENTS):
                chunk = pairs[c0:c0 + _oc.MAX_SEGMENTS]
                P = len(chunk)
                try:
                    xs = torch.stack([zF[ns, :, :d] + shocks[k:k + 1] for (_, _, ns, k) in chunk]).reshape(P * n, d).contiguous()
                    Z = torch.empty(nt + 3, P * n, d + 4, dtype=x.dtype, device=x.device)
                    Cc = torch.empty(nt + 3, P * n, cdim, dtype=x.dtype, device=x.device)
                except torch.OutOfMemoryError:                  # the pair-at-a-time path needs one pair's buffers only
                    ok = False
                    break
                # every pair's copy of the unshocked prefix (slots beyond its nShock are overwritten by its second segment)
                Z[:N + 1].view(N + 1, P, n, d + 4).copy_(zF.unsqueeze(1).expand(N + 1, P, n, d + 4))
                Cc[:N + 1].view(N + 1, P, n, cdim).copy_(cF.unsqueeze(1).expand(N + 1, P, n, cdim))
                got = _oc._launch_segments(xs, Phi, prob, [t for (_, t, _, _) in chunk], tspan[1], [1 + nt - ns for (_, _, ns, _) in chunk], n,
                                           stepper, alph, slot0s=[ns + 1 for (_, _, ns, _) in chunk], zFull=Z, ctrlFull=Cc)
                if got is None:
                    ok = False
                    break
                raw.append((chunk, xs, Z, Cc, got[1]))
        if group is not None:                               # the ranks take this path together or not at all (before anybody reduces anything)
            import torch.distributed as dist
            from .distributed import reduce_cost_sums
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(grp) > 1:
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=x.device if dist.get_backend(grp) != "gloo" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=grp)
                ok = bool(int(flag[0]))
        if not ok:
            return None
        res = {}
        # costs of every segment 1: running integrals at its end + the terminal terms at (z(t_s), t_s)  (src/OCflow.py:58-90)
        ends = torch.stack([zF[ns] for (_, _, ns) in on_grid])                                   # [S, n, d+4]
        S = ends.shape[0]
        tcol = torch.tensor([t for (_, t, _) in on_grid], dtype=x.dtype, device=x.device).view(S, 1, 1).expand(S, n, 1)
        s_in = torch.cat((ends[:, :, :d], tcol), dim=2).reshape(S * n, d + 1).contiguous()
        phi1 = Phi._value_f32(s_in).view(S, n)
        g1 = Phi._grad_f32(s_in).view(S, n, d + 1)
        resid = ends[:, :, :d] - prob.xtarget.to(x.dtype).view(1, 1, d)                          # ocG (src/OCflow.py:97-101)
        cG = 0.5 * (resid * resid).sum(dim=2)
        a0 = float(alph[0])
        sums1 = torch.stack((ends[:, :, d].sum(1), cG.sum(1), ends[:, :, d + 1].sum(1), (phi1 - a0 * cG).abs().sum(1),
                             (g1[:, :, :d] - a0 * resid).abs().sum(dim=(1, 2)), ends[:, :, d + 2].sum(1), ends[:, :, d + 3].sum(1),
                             torch.full((S,), float(n), dtype=x.dtype, device=x.device)), dim=1)  # [S, 8]
        sums2 = torch.cat([r[4] for r in raw], dim=0)                                             # [pairs, 8]
        if group is not None:
            sums1, sums2 = reduce_cost_sums(sums1.contiguous(), grp), reduce_cost_sums(sums2.contiguous(), grp)
        costs1, costs2 = _means(sums1, alph), _means(sums2, alph)
        si = {i: j for j, (i, _, _) in enumerate(on_grid)}
        p0 = 0
        for chunk, xs, Z, Cc, _ in raw:
            for p, (i, t, ns, k) in enumerate(chunk):
                res[(i, k)] = {"traj": Z[:, p * n:(p + 1) * n, :d].permute(1, 2, 0), "ctrl": Cc[:, p * n:(p + 1) * n, :].permute(1, 2, 0),
                               "costs1": costs1[si[i]], "costs2": costs2[p0 + p], "nShock": ns, "x_shocked": xs[p * n:(p + 1) * n]}
            p0 += len(chunk)
        _lib.check_errors(sync=True)        # results that are consumed on the host (plots, files): a failed launch must raise here
    return res
