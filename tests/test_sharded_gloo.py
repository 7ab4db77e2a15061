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


def _checker_segments(orc, P, S):
    """a stand-in for neuraloc_amd.OCflow._launch_segments (several time segments in one launch), computed by the oracle segment by segment"""
    def launch(x, Phi, prob, t0s, t1, nts, rows, stepper, alph, slot0s=None, zFull=None, ctrlFull=None):
        n = x.shape[0]
        slot0s = [0] * len(t0s) if slot0s is None else slot0s
        sums = torch.zeros(len(t0s), 8)
        with torch.no_grad():
            for k, (t0, ntk, s0) in enumerate(zip(t0s, nts, slot0s)):
                xs = x[k * rows:min(n, (k + 1) * rows)]
                tab = orc.persample_table(xs, P, S, [t0, t1], ntk, stepper, alph)
                sums[k] = torch.cat((tab.double().sum(0), torch.tensor([float(xs.shape[0])], dtype=torch.float64))).float()
                z, c = orc.rollout(xs, P, S, [t0, t1], ntk, stepper, alph, intermediates=True)
                zFull[s0:s0 + ntk + 1, k * rows:k * rows + xs.shape[0]] = z.permute(2, 0, 1)
                ctrlFull[s0:s0 + ntk + 1, k * rows:k * rows + xs.shape[0]] = c.permute(2, 0, 1)
        return None, sums, zFull, ctrlFull
    return launch


def _sweep_objects(orc, P, S, m, xtarget):
    class _Net:                                          # what the shared-prefix sweep asks of the network object
        alph = m["alph"]
        _value_f32 = staticmethod(lambda s: orc.phi_value(P, s))
        _grad_f32 = staticmethod(lambda s: orc.phi_grad(P, s))

    class _Prob:
        pass
    _Prob.xtarget = xtarget
    return _Net, _Prob


def _sweep_worker(rank, world, port, out_dir):
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
    from neuraloc_amd.distributed import shard_rows
    from neuraloc_amd.shock import shock_sweep
    g = load_golden("softcorridor")
    P, S = make_oracle(g, training=False)
    m = g.meta
    ocmod._launch = _checker_launch(orc, P, S)
    calls = []
    seg = _checker_segments(orc, P, S)
    ocmod._launch_segments = lambda *a, **k: (calls.append(len(a[3])), seg(*a, **k))[1]
    ocmod.segments_supported = lambda *a: True          # (the capability query is a library call: the stand-in has a segment "kernel")
    _Net, _Prob = _sweep_objects(orc, P, S, m, g.t("xtarget"))
    x = g.t("x")[:32]                                    # two shards of 16 rows: whole tiles, the shared-prefix path
    lo, hi = shard_rows(x.shape[0], rank, world)
    shocks = torch.tensor([[-0.2, -0.7, -0.0, -0.6], [0.1, 0.0, 0.0, 0.2]])
    res = shock_sweep(x[lo:hi], _Net, _Prob, 10, [0.2, 0.5, 0.33], shocks, alph=m["alph"], group=True)
    assert calls == [4], calls                           # the four on-grid pairs in ONE segment launch; t_s = 0.33 took the per-pair path
    np.savez(os.path.join(out_dir, f"sweep{rank}.npz"), n=len(res),
             **{f"traj{i}": r["traj"].numpy() for i, r in enumerate(res)},
             **{f"J1_{i}": float(r["costs1"][0]) for i, r in enumerate(res)}, **{f"J2_{i}": float(r["costs2"][0]) for i, r in enumerate(res)})
    # a rank whose shard is not whole tiles makes EVERY rank take the per-pair path (the flag all-reduce): no hang, same answers
    x2 = g.t("x")[:24]
    lo2, hi2 = (0, 16) if rank == 0 else (16, 24)
    calls.clear()
    res2 = shock_sweep(x2[lo2:hi2], _Net, _Prob, 10, [0.2, 0.5], shocks[:1], alph=m["alph"], group=True)
    np.savez(os.path.join(out_dir, f"sweepb{rank}.npz"), J2=float(res2[1]["costs2"][0]), ncalls=len(calls))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_shock_sweep_with_the_shared_prefix(tmp_path):
    """BASELINE config 5 over a row-sharded batch: every rank runs all (t_s, shock) pairs of its rows with the shared unshocked prefix and ONE
    segment launch; the costs are global means (two small all-reduces + the agreement flag), trajectories stay sharded -- equal to the
    single-process per-pair result; a rank that cannot take the path takes every rank off it."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from conftest import load_golden
    from util_hip import make_oracle
    from oracle import ocflow_oracle as orc
    import importlib
    ocmod = importlib.import_module("neuraloc_amd.OCflow")
    from neuraloc_amd.shock import shock_rollout
    world = 2
    mp.spawn(_sweep_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load_golden("softcorridor")
    P, S = make_oracle(g, training=False)
    m = g.meta
    saved = ocmod._launch
    try:
        ocmod._launch = _checker_launch(orc, P, S)
        _Net, _Prob = _sweep_objects(orc, P, S, m, g.t("xtarget"))
        x = g.t("x")[:32]
        shocks = torch.tensor([[-0.2, -0.7, -0.0, -0.6], [0.1, 0.0, 0.0, 0.2]])
        want = [shock_rollout(x, _Net, _Prob, 10, t, shocks[k:k + 1], alph=m["alph"]) for t in (0.2, 0.5, 0.33) for k in range(2)]
        want_b = shock_rollout(g.t("x")[:24], _Net, _Prob, 10, 0.5, shocks[:1], alph=m["alph"])
    finally:
        ocmod._launch = saved
    r = [np.load(tmp_path / "sweep0.npz"), np.load(tmp_path / "sweep1.npz")]
    assert int(r[0]["n"]) == int(r[1]["n"]) == 6
    for i, w in enumerate(want):
        both = np.concatenate([r[0][f"traj{i}"], r[1][f"traj{i}"]], axis=0)      # the ranks' row shards, in rank order
        assert both.shape == tuple(w["traj"].shape)
        assert np.allclose(both, w["traj"].numpy(), rtol=1e-5, atol=1e-4), i
        for rr in r:
            assert abs(float(rr[f"J1_{i}"]) - float(w["costs1"][0])) <= 2e-5 * abs(float(w["costs1"][0])), i
            assert abs(float(rr[f"J2_{i}"]) - float(w["costs2"][0])) <= 2e-5 * abs(float(w["costs2"][0])), i
    for k in range(2):
        b = np.load(tmp_path / f"sweepb{k}.npz")
        # (rank 0's shard qualified: it had launched before the flag took both ranks to the per-pair path; rank 1 never launched segments)
        assert int(b["ncalls"]) == (1 if k == 0 else 0) and abs(float(b["J2"]) - float(want_b["costs2"][0])) <= 2e-5 * abs(float(want_b["costs2"][0]))
