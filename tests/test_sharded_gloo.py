"""CPU, world_size 2 (gloo): the batch-sharded path -- row split, one SUM all-reduce of the 8-vector
[7 cost sums, count], means and Jc -- gives the reference's full-batch answer on every rank.
The per-rank launch is the checker here (no GPU in this container: neuraloc_amd.OCflow._launch is swapped for an oracle-backed
stand-in inside the spawned workers); on the GPU box the same functions run the HIP launch (bench.py --gpus N,
tests/test_sharded_gpu.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _checker_launch(orc, P, S):
    """a stand-in with the signature and return layout of neuraloc_amd.OCflow._launch, computed by the oracle"""
    def launch(x, Phi, prob, tspan, nt, stepper, alph, intermediates):
        with torch.no_grad():
            tab = orc.persample_table(x, P, S, tspan, nt, stepper, alph)
            sums = torch.cat((tab.double().sum(0), torch.tensor([float(x.shape[0])], dtype=torch.float64))).float()
            zF = cF = None
            if intermediates:
                z, c = orc.rollout(x, P, S, tspan, nt, stepper, alph, intermediates=True)
                zF, cF = z.permute(2, 0, 1).contiguous(), c.permute(2, 0, 1).contiguous()      # time-major like the kernel
        return tab, sums, zF, cF
    return launch


def _worker(rank, world, port, name, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_golden
    from util_hip import make_oracle
    from oracle import ocflow_oracle as orc
    from neuraloc_amd.distributed import OCflow_sharded, shard_rows

    g = load_golden(name)
    P, S = make_oracle(g, training=False)
    m = g.meta
    x = g.t("x")
    lo, hi = shard_rows(x.shape[0], rank, world)

    import importlib
    ocmod = importlib.import_module("neuraloc_amd.OCflow")
    ocmod._launch = _checker_launch(orc, P, S)          # no GPU in this container: the checker stands in for the HIP launch
    Jc, cs = OCflow_sharded(x[lo:hi], None, None, [0.0, 1.0], m["nt"], "rk4", m["alph"])
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array([float(Jc)] + [float(c) for c in cs] + [lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["softcorridor", "swap12"])
def test_two_rank_sharded_costs_match_the_reference(name, tmp_path):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from conftest import load_golden
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    g = load_golden(name)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0[:8], r1[:8]), "ranks disagree after the all-reduce"
    assert (r0[8], r0[9], r1[8], r1[9]) == (0, 20, 20, 40)
    want = np.concatenate(([float(g["eval_rk4/Jc"])], g["eval_rk4/cs"].astype(np.float64)))
    assert np.all(np.abs(r0[:8] - want) <= 1e-5 * np.abs(want) + 1e-7)


def test_shard_rows_is_a_partition():
    from neuraloc_amd.distributed import shard_rows
    for n in (1, 7, 1024, 1027):
        for world in (1, 2, 4, 8):
            edges = [shard_rows(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def _flat_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neuraloc_amd.distributed import allreduce_flat
    shapes = [(10, 5), (1, 5), (1,), (1, 16), (16, 5), (16,), (16, 16), (16,)]        # Phi(2,16,4).named_parameters()
    ts = [torch.full(s, float(rank + 1)) * (i + 1) for i, s in enumerate(shapes)]
    out = allreduce_flat(ts)
    ok = all(o.shape == t.shape and torch.equal(o, torch.full(t.shape, 3.0 * (i + 1))) for i, (o, t) in enumerate(zip(out, ts)))
    ok = ok and all(torch.equal(t, torch.full(t.shape, float(rank + 1) * (i + 1))) for i, t in enumerate(ts))   # inputs untouched
    np.save(os.path.join(out_dir, f"flat{rank}.npy"), np.array([float(ok)]))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_all_reduce_is_one_flat_sum(tmp_path):
    """the training backward all-reduces every parameter gradient of Phi in one flat buffer"""
    world = 2
    mp.spawn(_flat_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert np.load(tmp_path / "flat0.npy")[0] == 1.0 and np.load(tmp_path / "flat1.npy")[0] == 1.0


def _shock_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_golden
    from util_hip import make_oracle
    from oracle import ocflow_oracle as orc
    import importlib
    ocmod = importlib.import_module("neuraloc_amd.OCflow")
    from neuraloc_amd.distributed import OCflow_sharded, shard_rows
    from neuraloc_amd.shock import shock_rollout

    g = load_golden("softcorridor")
    P, S = make_oracle(g, training=False)
    m = g.meta
    ocmod._launch = _checker_launch(orc, P, S)

    class _Net:                                          # shock_rollout only reads Phi.alph when alph is not given
        alph = m["alph"]
    x = g.t("x")[:9]                                     # 9 rows over 2 ranks: shards of 5 and 4
    lo, hi = shard_rows(x.shape[0], rank, world)
    shock = torch.tensor([[-0.2, -0.7, -0.0, -0.6]])
    res = shock_rollout(x[lo:hi], _Net, None, 10, 0.3, shock, alph=m["alph"], group=True, gather=True)
    loc = shock_rollout(x[lo:hi], _Net, None, 10, 0.3, shock, alph=m["alph"], group=True, gather=False)
    zF, cF = OCflow_sharded(x[lo:hi], None, None, [0.0, 1.0], 6, "rk4", m["alph"], intermediates=True, gather=True)
    np.savez(os.path.join(out_dir, f"shock{rank}.npz"), traj=res["traj"].numpy(), ctrl=res["ctrl"].numpy(),
             J1=float(res["costs1"][0]), J2=float(res["costs2"][0]), loc_rows=loc["traj"].shape[0], zF=zF.numpy(), cF=cF.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_shock_sweep_matches_single_rank(tmp_path):
    """shocked two-segment rollouts and intermediates over a row-sharded batch -- outputs stay sharded or are
    all-gathered (uneven shards), costs are the global means -- equal the single-rank result"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from conftest import load_golden
    from util_hip import make_oracle
    from oracle import ocflow_oracle as orc
    import importlib
    ocmod = importlib.import_module("neuraloc_amd.OCflow")
    from neuraloc_amd.shock import shock_rollout
    world = 2
    mp.spawn(_shock_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load_golden("softcorridor")
    P, S = make_oracle(g, training=False)
    m = g.meta
    saved = ocmod._launch
    try:
        ocmod._launch = _checker_launch(orc, P, S)

        class _Net:
            alph = m["alph"]
        x = g.t("x")[:9]
        want = shock_rollout(x, _Net, None, 10, 0.3, torch.tensor([[-0.2, -0.7, -0.0, -0.6]]), alph=m["alph"])
        with torch.no_grad():
            zW, cW = orc.rollout(x, P, S, [0.0, 1.0], 6, "rk4", m["alph"], intermediates=True)
    finally:
        ocmod._launch = saved
    r0, r1 = np.load(tmp_path / "shock0.npz"), np.load(tmp_path / "shock1.npz")
    for r in (r0, r1):
        # (the checker's eager GEMMs round differently for 4/5-row shards than for the 9-row batch: 1e-7 relative)
        assert np.allclose(r["traj"], want["traj"].numpy(), rtol=1e-5, atol=1e-5) and np.allclose(r["ctrl"], want["ctrl"].numpy(), rtol=1e-4, atol=1e-4)
        assert np.array_equal(r0["traj"], r1["traj"]) and np.array_equal(r0["ctrl"], r1["ctrl"]), "ranks disagree after the all-gather"
        assert abs(float(r["J1"]) - float(want["costs1"][0])) <= 1e-5 * abs(float(want["costs1"][0]))
        assert abs(float(r["J2"]) - float(want["costs2"][0])) <= 1e-5 * abs(float(want["costs2"][0]))
        assert np.allclose(r["zF"], zW.numpy(), rtol=1e-4, atol=1e-4) and np.allclose(r["cF"], cW.numpy(), rtol=1e-4, atol=1e-4)
    assert (int(r0["loc_rows"]), int(r1["loc_rows"])) == (5, 4)
