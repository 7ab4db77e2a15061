"""CPU: host-side logic of the training / sharding path that needs no GPU."""
import torch

from neuraloc_amd.OCflow import costs_from_sums
from neuraloc_amd.distributed import allreduce_flat, reduce_cost_sums, shard_rows
from neuraloc_amd.train import _step_sizes


def test_step_sizes_follow_the_reference_time_bookkeeping():
    """src/OCflow.py:35,50 and :170: tk += h in double, each step re-derives h = (tk + h) - tk, cast to fp32"""
    for t0, t1, nt in [(0.0, 1.0, 80), (0.2, 0.7, 37), (0.0, 1.0, 3)]:
        hs = _step_sizes([t0, t1], nt)
        h = (t1 - t0) / nt
        tk = t0
        for k in range(nt):
            assert hs[k].item() == torch.tensor((tk + h) - tk, dtype=torch.float32).item()
            tk += h
        assert hs.dtype == torch.float32 and hs.shape == (nt,)


def test_costs_from_sums_on_host_tensors_matches_the_reference_formula():
    sums = torch.tensor([10.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 4.0])
    alph = [100.0, 9.0, 9.0, 0.5, 0.25, 0.125]
    Jc, cs = costs_from_sums(sums, alph)
    means = [v / 4.0 for v in (10.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0)]
    assert [float(c) for c in cs] == means
    assert float(Jc) == means[0] + 100.0 * means[1] + 0.5 * means[2] + 0.25 * means[3] + 0.125 * means[4]   # src/OCflow.py:88-90


def test_single_process_reductions_are_identities():
    ts = [torch.arange(6.0).reshape(2, 3), torch.ones(4)]
    out = allreduce_flat(ts)
    assert all(torch.equal(a, b) for a, b in zip(out, ts))
    s = torch.arange(8.0)
    assert torch.equal(reduce_cost_sums(s.clone()), s)
    assert shard_rows(10, 0, 1) == (0, 10)
