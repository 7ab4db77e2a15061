"""GPU, two ranks sharing the one MI355X of the test box (gloo rendezvous, reductions staged through the host):
the batch-sharded TRAINING path -- each rank rolls out and back-propagates its rows through the HIP kernels, the cost
sums and the flat gradient buffer are all-reduced -- leaves the full-batch Jc and parameter gradients on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neuraloc_amd as na
    from conftest import load_golden
    from util_hip import make_net, make_prob
    dev = torch.device("cuda:0")
    g = load_golden(name)
    net = make_net(g, dev).train()
    prob = make_prob(g, dev, training=True)
    x = g.t("x")[:30].to(dev)
    lo, hi = na.shard_rows(x.shape[0], rank, world)
    Jc, cs = na.OCflow_sharded(x[lo:hi].contiguous(), net, prob, [0.0, 1.0], 8, "rk4", g.meta["alph"])
    Jc.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
    np.save(os.path.join(out_dir, f"g{rank}.npy"), np.concatenate(([Jc.item()], flat)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["softcorridor", "swap12"])
def test_two_rank_training_matches_full_batch(name, tmp_path):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import neuraloc_amd as na
    from conftest import load_golden
    from util_hip import make_net, make_prob
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert np.array_equal(r0, r1), "ranks disagree after the all-reduces"
    dev = torch.device("cuda:0")
    g = load_golden(name)
    net = make_net(g, dev).train()
    prob = make_prob(g, dev, training=True)
    Jc, _ = na.OCflow(g.t("x")[:30].to(dev), net, prob, [0.0, 1.0], 8, "rk4", g.meta["alph"])
    Jc.backward()
    full = np.concatenate(([Jc.item()], torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()))
    scale = np.abs(full[1:]).max()
    assert abs(r0[0] - full[0]) <= 1e-5 * abs(full[0])
    assert np.abs(r0[1:] - full[1:]).max() <= 2e-5 * scale


def _shock_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neuraloc_amd as na
    from neuraloc_amd.shock import shock_rollout
    from conftest import load_golden
    from util_hip import make_net, make_prob
    dev = torch.device("cuda:0")
    g = load_golden("singlequad")
    net, prob = make_net(g, dev), make_prob(g, dev, training=False)
    x = g.t("x")[:21].to(dev)                              # 21 rows over 2 ranks: shards of 11 and 10
    lo, hi = na.shard_rows(x.shape[0], rank, world)
    shock = torch.zeros(1, g.meta["d"], device=dev)
    shock[0, :3] = torch.tensor([0.5, -0.5, 0.25])
    with torch.no_grad():
        res = shock_rollout(x[lo:hi].contiguous(), net, prob, 20, 0.3, shock, alph=g.meta["alph"], group=True, gather=True)
        zF, cF = na.OCflow_sharded(x[lo:hi].contiguous(), net, prob, [0.0, 1.0], 12, "rk4", g.meta["alph"], intermediates=True, gather=True)
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), traj=res["traj"].cpu().numpy(), ctrl=res["ctrl"].cpu().numpy(),
             J1=float(res["costs1"][0]), J2=float(res["costs2"][0]), zF=zF.cpu().numpy(), cF=cF.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_shock_sweep_on_the_gpu(tmp_path):
    """SURVEY 8(e) / BASELINE config 5: shocked quadcopter rollouts and intermediates over a row-sharded batch through the
    HIP kernels (uneven shards, all-gathered) equal the single-process result row for row"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import neuraloc_amd as na
    from neuraloc_amd.shock import shock_rollout
    from conftest import load_golden
    from util_hip import make_net, make_prob
    world = 2
    mp.spawn(_shock_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    dev = torch.device("cuda:0")
    g = load_golden("singlequad")
    net, prob = make_net(g, dev), make_prob(g, dev, training=False)
    x = g.t("x")[:21].to(dev)
    shock = torch.zeros(1, g.meta["d"], device=dev)
    shock[0, :3] = torch.tensor([0.5, -0.5, 0.25])
    with torch.no_grad():
        want = shock_rollout(x, net, prob, 20, 0.3, shock, alph=g.meta["alph"])
        zW, cW = na.OCflow(x, net, prob, [0.0, 1.0], 12, "rk4", g.meta["alph"], intermediates=True)
    r0, r1 = np.load(tmp_path / "s0.npz"), np.load(tmp_path / "s1.npz")
    for r in (r0, r1):
        assert np.array_equal(r["traj"], want["traj"].cpu().numpy()), "per-sample trajectories do not depend on the sharding"
        assert np.array_equal(r["ctrl"], want["ctrl"].cpu().numpy())
        assert np.array_equal(r["zF"], zW.cpu().numpy()) and np.array_equal(r["cF"], cW.cpu().numpy())
        for k, w in (("J1", want["costs1"][0]), ("J2", want["costs2"][0])):
            assert abs(float(r[k]) - float(w)) <= 1e-5 * abs(float(w))


def _sweep_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neuraloc_amd as na
    from neuraloc_amd import _lib
    from neuraloc_amd.shock import shock_sweep
    from conftest import load_golden
    from util_hip import full_states, make_net, make_prob
    dev = torch.device("cuda:0")
    g = load_golden("singlequad")
    net, prob = make_net(g, dev), make_prob(g, dev, training=False)
    x = full_states(g, 5)[:64].to(dev)                     # 64 rows over 2 ranks: two whole tiles each -> the shared-prefix path
    lo, hi = na.shard_rows(x.shape[0], rank, world)
    shocks = torch.zeros(1, g.meta["d"], device=dev)
    shocks[0, :3] = torch.tensor([0.5, -0.5, 0.25])
    times = [0.1 * k for k in range(1, 10)]
    with torch.no_grad():
        res = shock_sweep(x[lo:hi].contiguous(), net, prob, 50, times, shocks, alph=g.meta["alph"], group=True)
    np.savez(os.path.join(out_dir, f"w{rank}.npz"), kernel=_lib.lib().nocf_last_rollout_kernel().decode(),
             **{f"traj{i}": r["traj"].cpu().numpy() for i, r in enumerate(res)},
             **{f"J1_{i}": float(r["costs1"][0]) for i, r in enumerate(res)}, **{f"J2_{i}": float(r["costs2"][0]) for i, r in enumerate(res)})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_shock_sweep_with_the_shared_prefix_on_the_gpu(tmp_path):
    """BASELINE config 5 sharded by rows: every rank runs all nine shock times of its rows with the shared unshocked prefix and one
    segment launch (nocf_rollout_segments_f32), the costs are global means: equal to the single-process sweep (trajectories row for row: the
    per-sample arithmetic does not depend on the sharding) and, through it, to the per-t_s rollouts (tests/test_hip_parity.py)"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import neuraloc_amd as na
    from neuraloc_amd.shock import shock_sweep
    from conftest import load_golden
    from util_hip import full_states, make_net, make_prob
    world = 2
    mp.spawn(_sweep_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    dev = torch.device("cuda:0")
    g = load_golden("singlequad")
    net, prob = make_net(g, dev), make_prob(g, dev, training=False)
    x = full_states(g, 5)[:64].to(dev)
    shocks = torch.zeros(1, g.meta["d"], device=dev)
    shocks[0, :3] = torch.tensor([0.5, -0.5, 0.25])
    with torch.no_grad():
        want = shock_sweep(x, net, prob, 50, [0.1 * k for k in range(1, 10)], shocks, alph=g.meta["alph"])
    r = [np.load(tmp_path / "w0.npz"), np.load(tmp_path / "w1.npz")]
    assert str(r[0]["kernel"]) == "rollout_mono_kernel"
    for i, w in enumerate(want):
        both = np.concatenate([r[0][f"traj{i}"], r[1][f"traj{i}"]], axis=0)
        assert np.array_equal(both, w["traj"].cpu().numpy()), f"pair {i}: trajectories depend on the sharding"
        for rr in r:
            assert abs(float(rr[f"J1_{i}"]) - float(w["costs1"][0])) <= 1e-5 * abs(float(w["costs1"][0]))
            assert abs(float(rr[f"J2_{i}"]) - float(w["costs2"][0])) <= 1e-5 * abs(float(w["costs2"][0]))
