"""Batch-sharded OCflow: one process per GPU, samples split by rows, one all-reduce.

Samples are independent until the final mean (src/OCflow.py:80-86), so each rank integrates
its contiguous slice of x and contributes 8 numbers (7 cost sums + its row count); a single
SUM all-reduce over RCCL/xGMI (backend "nccl" on ROCm) makes the means identical on all
ranks.  No other collective exists on this path.
"""
import torch
import torch.distributed as dist

from importlib import import_module

# the OCflow MODULE (the package re-exports the function under the same name); the HIP launch is looked up at call time
# (_oc._launch): the CPU/gloo tests swap in a checker there
_oc = import_module(__package__ + ".OCflow")
from .OCflow import costs_from_sums


def shard_rows(n, rank, world):
    """contiguous, balanced row range [lo, hi) of rank `rank` among `world` ranks"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _sum_all_reduce(t, group):
    """in-place SUM all-reduce; RCCL ("nccl") reduces device memory directly, the gloo backend (tests: several ranks
    sharing one GPU, or CPU tensors) goes through the host"""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def reduce_cost_sums(local_sums, group=None):
    """SUM all-reduce of the 8-vector [7 cost sums, count]; in place, returns it"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        _sum_all_reduce(local_sums, group)
    return local_sums


def allreduce_flat(tensors, group=None):
    """SUM all-reduce of a list of tensors as ONE flat buffer (the parameter gradients of Phi: 415 ... 342 654
    floats); returns new tensors of the original shapes.  One collective instead of one per parameter: over xGMI the
    cost is latency, not bytes, at these sizes."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return list(tensors)
    flat = torch.cat([t.reshape(-1) for t in tensors])
    _sum_all_reduce(flat, group)
    out, o = [], 0
    for t in tensors:
        out.append(flat[o:o + t.numel()].view(t.shape))
        o += t.numel()
    return out


def gather_rows(t_local, group=None):
    """all-gather of row shards (dim 0) whose sizes may differ by one (shard_rows): every rank gets the concatenation in rank order.
    RCCL gathers device memory directly; the gloo backend goes through the host."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return t_local
    world = dist.get_world_size(group)
    via_host = t_local.is_cuda and dist.get_backend(group) == "gloo"
    src = t_local.cpu() if via_host else t_local
    rows = torch.tensor([src.shape[0]], dtype=torch.int64, device=src.device)
    all_rows = [torch.zeros_like(rows) for _ in range(world)]
    dist.all_gather(all_rows, rows, group=group)
    counts = [int(r.item()) for r in all_rows]
    pad = max(counts)
    buf = src.new_zeros((pad,) + tuple(src.shape[1:]))
    buf[: src.shape[0]] = src
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf.contiguous(), group=group)
    out = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    return out.to(t_local.device) if via_host else out


def OCflow_sharded(x_local, Phi, prob, tspan, nt, stepper="rk4", alph=[1.0] * 6, group=None, intermediates=False, gather=False):
    """OCflow over a batch whose rows are spread over the ranks of `group`.

    x_local is this rank's slice.  Returns the same (Jc, cs) on every rank, equal (up to the summation order of 8 fp32
    numbers) to OCflow on the concatenated batch.  intermediates=True returns (zFull, ctrlFull) like the reference
    (src/OCflow.py:37-55, [rows, d+4, nt+1] and [rows, a, nt+1]): this rank's rows by default -- the trajectories stay
    sharded, SURVEY 8(e) -- or, with gather=True, all rows on every rank (one all-gather per array)."""
    if not intermediates and torch.is_grad_enabled() and Phi is not None and any(p.requires_grad for p in Phi.parameters()):
        from .train import ocflow_train                  # training: sums and gradients are all-reduced inside
        return ocflow_train(x_local, Phi, prob, tspan, nt, stepper, alph, group=True if group is None else group)
    _, sums, zF, cF = _oc._launch(x_local, Phi, prob, tspan, nt, stepper, alph, bool(intermediates))
    if intermediates:
        zF, cF = zF.permute(1, 2, 0), cF.permute(1, 2, 0)           # kernel layout is time-major
        if gather:
            zF, cF = gather_rows(zF.contiguous(), group), gather_rows(cF.contiguous(), group)
        return zF, cF
    sums = reduce_cost_sums(sums, group)
    return costs_from_sums(sums, alph)
