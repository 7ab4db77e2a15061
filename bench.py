#!/usr/bin/env python3
"""bench.py -- OCflow rollout throughput on MI355X (the metric BASELINE.json names).

  python bench.py --gpus N --steps K --warmup W [--workload swarm50]

A "step" is one OCflow call (non-intermediates: returns Jc and the 7 costs) over one batch of
synthetic states already resident in HBM.  Default workload = the configuration north_star
quotes its target on: swarm50 (true d=150, m=512, nTh=2), nt=80, n=1024 per GPU, fp32, RK4,
prob.eval().  Weights are the reference's pretrained swarm50 network exported to
tests/golden/swarm50.npz; states are xInit + var0 * (closed-form pseudo-normal table).

N>1: one process per GPU (torch.distributed, backend nccl = RCCL); the batch shards by rows,
every rank integrates its own 1024 samples (weak scaling) and one SUM all-reduce of 8 floats
per call forms the global means.  value = total trajectories / max-over-ranks wall time.

One JSON line on stdout (rank 0), with `roofline` (rollout kernel, fp32-MFMA roof, duration
from HIP events recorded around the kernel on its launch stream) and `cpu_baseline` (the oracle
-- the op-for-op eager-PyTorch port of the reference -- timed on the host cores, rank 0, N=1).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import neuraloc_amd as na                      # noqa: E402
from neuraloc_amd import _lib                  # noqa: E402
from neuraloc_amd.OCflow import _launch, costs_from_sums   # noqa: E402
from neuraloc_amd.distributed import reduce_cost_sums      # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3                   # MI355X_MICROARCH.md: dense fp32 matrix peak (= vector peak)
PEAK_HBM_GBS = 8000.0


def closed_form_normal(n, d, seed):
    """RNG-free pseudo-normal table (same generator as tests/golden/make_golden.py)"""
    i = np.arange(n * d, dtype=np.float64) + 1.0 + 1000.0 * seed
    u1 = np.clip(np.mod(i * 0.6180339887498949, 1.0), 1e-9, 1.0)
    u2 = np.mod(i * 0.7548776662466927 + 0.31, 1.0)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(z.reshape(n, d).astype(np.float32))


def load_workload(name):
    z = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
    meta = json.loads(str(z["meta"]))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    return meta, sd, torch.from_numpy(z["xtarget"]), torch.from_numpy(z["xInit"])


def make_states(meta, xInit, n, seed):
    xi = closed_form_normal(n, meta["d"], seed)
    if meta["name"] == "singlequad":
        xi[:, 3:] = 0.0
    return (xInit + meta["var0"] * xi).contiguous()


def flops_per_state_step(meta):
    """4 RHS evaluations x [4m(d+1) + 4m^2(nTh-1) + 4r(d+1)] (problem terms excluded)"""
    d, m, nTh = meta["d"], meta["m"], meta["nTh"]
    r = min(10, d + 1)
    return 4 * (4 * m * (d + 1) + 4 * m * m * (nTh - 1) + 4 * r * (d + 1))


def cpu_baseline(meta, sd, xtarget, x, nt, budget_s=25.0):
    """the oracle (checker / CPU port of the reference) timed on the host cores of this box.
    Eager PyTorch on ~400 small ops per RHS evaluation does not scale to every core of a big host, so a
    short probe (nt=2) picks the best intra-op thread count first; the figure reported is the full
    workload at that count (`cores`), and the probe table is kept in `sample`."""
    from oracle import ocflow_oracle as orc
    kind = {"Cross2D": orc.KIND_CROSS2D, "SwarmTraj": orc.KIND_SWARM, "Quadcopter": orc.KIND_QUAD}[meta["prob_class"]]
    P = orc.PhiParams.from_state_dict(sd)
    S = orc.ProbSpec(kind, xtarget, meta["obstacle"], meta["alph_Q"], meta["alph_W"], meta["r"], training=False)
    ncpu = os.cpu_count() or 1
    # eager PyTorch collapses far beyond 64 threads on these shapes (256 threads: < 1 traj/s), so the probe stops at 64
    cands = sorted({c for c in (1, 4, 8, 16, 32, 64, min(ncpu, 64)) if c <= ncpu})
    probe = {}
    with torch.no_grad():
        for th in cands:
            torch.set_num_threads(th)
            orc.rollout(x, P, S, [0.0, 1.0], 1, "rk4", meta["alph"])
            t0 = time.perf_counter()
            orc.rollout(x, P, S, [0.0, 1.0], 2, "rk4", meta["alph"])
            probe[th] = time.perf_counter() - t0
        best = min(probe, key=probe.get)
        torch.set_num_threads(best)
        est = probe[best] * nt / 2.0
        reps = max(1, min(5, int(budget_s / max(est, 1e-3))))
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            Jc, _ = orc.rollout(x, P, S, [0.0, 1.0], nt, "rk4", meta["alph"])
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    table = ", ".join(f"{k}t:{x.shape[0] * 2 / nt / v:.0f}" for k, v in probe.items())
    return {"value": x.shape[0] / med, "unit": "trajectories/s", "cores": best, "kind": "port",
            "sample": f"full workload n={x.shape[0]} nt={nt}, median of {len(times)} call(s) at the best of the probed "
                      f"intra-op thread counts (probe nt=2, traj/s-equivalent per count: {table}); eager PyTorch "
                      f"{torch.__version__}, os.cpu_count()={ncpu}",
            "seconds_per_call": med, "Jc": float(Jc)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", type=str, default="swarm50",
                    choices=["swap2", "softcorridor", "swap12", "swarm50", "singlequad"])
    ap.add_argument("--n", type=int, default=0, help="samples per GPU (default: the BASELINE.json n for the workload)")
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("NOCF_BENCH_BACKEND", "nccl")      # "gloo": only to exercise this script where ranks share a GPU
        ndev = torch.cuda.device_count()
        local_rank = local_rank % max(1, ndev)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    else:
        dist = None
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    meta, sd, xtarget, xInit = load_workload(args.workload)
    n = args.n or meta["n_full"]
    nt = args.nt or meta["nt"]
    alph = meta["alph"]
    net = na.Phi(nTh=meta["nTh"], m=meta["m"], d=meta["d"], alph=alph)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    cls = {"Cross2D": na.Cross2D, "SwarmTraj": na.SwarmTraj, "Quadcopter": na.Quadcopter}[meta["prob_class"]]
    if meta["prob_class"] == "Quadcopter":
        prob = cls(xtarget.to(dev), obstacle=None, alph_Q=meta["alph_Q"], alph_W=meta["alph_W"])
    else:
        prob = cls(xtarget.to(dev), obstacle=meta["obstacle"], alph_Q=meta["alph_Q"], alph_W=meta["alph_W"], r=meta["r"])
    prob.eval()
    x_cpu = make_states(meta, xInit, n, seed=200 + rank)
    x = x_cpu.to(dev)                                   # inputs resident in HBM before the timed region

    host_reduce = world > 1 and os.environ.get("NOCF_BENCH_BACKEND", "nccl") != "nccl"

    def step():
        _, sums, _, _ = _launch(x, net, prob, [0.0, 1.0], nt, "rk4", alph, False)
        if host_reduce:                                  # gloo test mode: reduce on the host
            sums = reduce_cost_sums(sums.cpu()).to(dev)
        elif world > 1:
            reduce_cost_sums(sums)                       # one RCCL all-reduce of 8 floats
        return costs_from_sums(sums, alph)

    L = _lib.lib()
    with torch.no_grad():
        for _ in range(args.warmup):
            Jc, cs = step()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        L.nocf_profile_begin()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            Jc, cs = step()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kms, nl = C.c_double(0.0), C.c_int32(0)
        L.nocf_profile_end(C.byref(kms), C.byref(nl))
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_reduce else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        total_traj = world * n * args.steps
        value = total_traj / elapsed
        kernel_ms = kms.value / max(1, nl.value)
        fl_launch = flops_per_state_step(meta) * n * nt
        achieved = fl_launch / (kernel_ms * 1e-3) / 1e12
        alg_bytes = 8 * (meta["d"] + 4) * n * nt        # state in/out once per step
        traffic = None
        tp = os.path.join(REPO, "profiles", f"hbm_traffic_{args.workload}.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "trajectories/sec (n_train x nt states integrated)", "value": value, "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic states (xInit + var0 * closed-form normal table); pretrained reference weights exported to npz",
            "config": {"workload": f"{args.workload} d={meta['d']} m={meta['m']} nTh={meta['nTh']} nt={nt} "
                                   f"n={n}/GPU rk4 eval-mode fp32", "global_batch": world * n,
                       "state_steps_per_s": value * nt, "parallelism": f"batch-sharded x{world}",
                       "Jc": float(Jc)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "kernel": "rollout_kernel", "kernel_ms": kernel_ms, "launches_timed": nl.value,
                         "algorithmic_flops_per_launch": fl_launch,
                         "algorithmic_hbm_bytes_per_launch": alg_bytes,
                         "achieved_hbm_GBps_algorithmic": alg_bytes / (kernel_ms * 1e-3) / 1e9,
                         "hbm_peak_GBps": PEAK_HBM_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(meta, sd, xtarget, x_cpu, nt)
            out["config"]["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
