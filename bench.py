#!/usr/bin/env python3
"""bench.py -- OCflow rollout throughput on MI355X (the metric BASELINE.json names).

  python bench.py --gpus N --steps K --warmup W [--workload swarm50] [--scaling strong|weak] [--n ROWS]

A "step" is one public `OCflow(x, Phi, prob, [0,1], nt, "rk4", alph)` call (non-intermediates: returns Jc and the 7
costs) over a batch of synthetic states already resident in HBM.  Default workload = the configuration north_star
quotes its target on: swarm50 (true d=150, m=512, nTh=2), nt=80, GLOBAL n=1024, fp32, RK4, prob.eval().  Weights are
the reference's pretrained swarm50 network exported to tests/golden/swarm50.npz; states are xInit + var0 * (closed-form
pseudo-normal table).

N>1: one process per GPU (torch.distributed, backend nccl = RCCL).  Default = STRONG scaling, the partition north_star
asks for: the global batch of n_train rows is split contiguously over the ranks (`shard_rows`), every rank integrates
its n/N rows and ONE SUM all-reduce of 8 floats per call forms the global means (src/OCflow.py:80-86 takes them over
the whole batch).  `--scaling weak` gives every rank its own n rows instead.  value = global trajectories / max-over-
ranks wall time.  `--n` (N=1) with 512/256/128 rows is the single-GPU proxy for the per-rank work at 2/4/8 GPUs.

`--workload singlequad-shock` is BASELINE config 5: 9 shock times x 4096 quadcopter states, two-segment rollouts with
trajectories and controls kept (neuraloc_amd.shock); it prints its own JSON line.

One JSON line on stdout (rank 0), with `roofline` (rollout kernel, fp32-MFMA roof, duration from HIP events recorded
around the kernel on its launch stream) and `cpu_baseline` (the oracle -- the op-for-op eager-PyTorch port of the
reference -- timed on the host cores, rank 0, N=1).
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
from neuraloc_amd.distributed import OCflow_sharded, shard_rows   # noqa: E402

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


def build_objects(meta, sd, xtarget, dev):
    net = na.Phi(nTh=meta["nTh"], m=meta["m"], d=meta["d"], alph=meta["alph"])
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    cls = {"Cross2D": na.Cross2D, "SwarmTraj": na.SwarmTraj, "Quadcopter": na.Quadcopter}[meta["prob_class"]]
    if meta["prob_class"] == "Quadcopter":
        prob = cls(xtarget.to(dev), obstacle=None, alph_Q=meta["alph_Q"], alph_W=meta["alph_W"])
    else:
        prob = cls(xtarget.to(dev), obstacle=meta["obstacle"], alph_Q=meta["alph_Q"], alph_W=meta["alph_W"], r=meta["r"])
    prob.eval()
    return net, prob


def flops_per_state_step(meta):
    """4 RHS evaluations x [4m(d+1) + 4m^2(nTh-1) + 4r(d+1)] (problem terms excluded)"""
    d, m, nTh = meta["d"], meta["m"], meta["nTh"]
    r = min(10, d + 1)
    return 4 * (4 * m * (d + 1) + 4 * m * m * (nTh - 1) + 4 * r * (d + 1))


def cpu_baseline(meta, sd, xtarget, x, nt, budget_s=60.0):
    """The oracle (checker / CPU port of the reference) timed on the host cores of this box: FULL nt-step rollouts of the same batch
    (SURVEY 8(d) times whole OCflow calls).  Protocol: (1) a thread-count probe -- the second of two 1-step calls per count (eager PyTorch
    on ~460 small ops per RHS evaluation does not scale to every core of a big host: at 256 threads it is slower than at 1); (2) at the
    best probed count: one 2-step warm-up call, then the MEDIAN OF 3 FULL ROLLOUTS = `value` (fewer when one rollout alone would exceed a
    third of the time budget: the count is stated).  Nothing is extrapolated into `value`; the 1-thread / all-cores figures of the probe are
    labelled as probe figures."""
    from oracle import ocflow_oracle as orc
    kind = {"Cross2D": orc.KIND_CROSS2D, "SwarmTraj": orc.KIND_SWARM, "Quadcopter": orc.KIND_QUAD}[meta["prob_class"]]
    P = orc.PhiParams.from_state_dict(sd)
    S = orc.ProbSpec(kind, xtarget, meta["obstacle"], meta["alph_Q"], meta["alph_W"], meta["r"], training=False)
    ncpu = os.cpu_count() or 1
    n = x.shape[0]

    def call(threads, steps):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        orc.rollout(x, P, S, [0.0, steps / nt], steps, "rk4", meta["alph"])
        return time.perf_counter() - t0

    notes = []
    with torch.no_grad():
        step_s = {}                                      # seconds per RK4 step by thread count (probe: second of two 1-step calls)
        for th in sorted({c for c in (1, 4, 8, 16, 32, 64, ncpu) if c <= ncpu}):
            first = call(th, 1)
            step_s[th] = first if first > 8.0 else call(th, 1)
            if step_s[th] > 8.0:
                notes.append(f"{th} threads: one step took {step_s[th]:.1f} s, larger counts not probed")
                break
        best = min(step_s, key=step_s.get)
        est_full = step_s[best] * nt
        reps = 3 if est_full <= budget_s / 3.0 else 1
        call(best, 2)                                    # warm-up at the chosen count
        fulls = []
        for _ in range(reps):
            torch.set_num_threads(best)
            t0 = time.perf_counter()
            orc.rollout(x, P, S, [0.0, 1.0], nt, "rk4", meta["alph"])
            fulls.append(time.perf_counter() - t0)
    full_s = float(np.median(fulls))
    table = ", ".join(f"{k}t:{n / nt / v:.0f}" for k, v in step_s.items())
    return {"value": n / full_s, "unit": "trajectories/s", "cores": best, "kind": "port",
            "value_basis": f"median of {reps} full {nt}-step rollout(s) at the best probed thread count",
            "full_rollout_s": full_s, "full_rollouts_s": fulls, "cpu_model": _cpu_model(), "os_cpu_count": ncpu,
            "probe_one_thread": n / nt / step_s[1], "probe_all_cores": (n / nt / step_s[ncpu]) if ncpu in step_s else None, "all_cores_count": ncpu,
            "sample": f"n={n} rows, the whole {nt}-step RK4 rollout (the workload of `value`), {reps} time(s) at {best} intra-op threads, median; "
                      f"thread probe (second of two 1-step calls, traj/s-equivalent -- probe figures, not `value`): {table}"
                      + ("; " + "; ".join(notes) if notes else "") + f"; eager PyTorch {torch.__version__}"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def read_traffic(workload, kernel, n_local):
    """HBM bytes per launch from the PMC passes (tools/parse_pmc.py writes the file; rocprofv3 --pmc cannot run inside this
    process).  Used only when it was collected for the same kernel and per-GPU batch; its provenance travels with it."""
    tp = os.path.join(REPO, "profiles", f"hbm_traffic_{workload}.json")
    if not os.path.exists(tp):
        return None, None
    try:
        j = json.load(open(tp))
    except Exception:
        return None, None
    if j.get("kernel_family") != kernel or int(j.get("n", -1)) != int(n_local):
        return None, f"{os.path.relpath(tp, REPO)} is for kernel={j.get('kernel_family')} n={j.get('n')}: not this run"
    return j.get("hbm_bytes_per_launch"), f"{os.path.relpath(tp, REPO)} ({j.get('source', '?')}, {j.get('collected', '?')})"


def bench_shock(args, dev, world=1, rank=0, dist=None):
    """BASELINE config 5: singlequad, 9 shock times x 4096 states x nt = 50, trajectories and controls kept.  N GPUs: the states are sharded by
    rows (strong scaling of the global batch), every rank runs all shock times of its rows (shock_sweep(group=True): shared unshocked prefix,
    one segment launch), the costs are global means; trajectories stay sharded."""
    from neuraloc_amd.shock import shock_sweep
    meta, sd, xtarget, xInit = load_workload("singlequad")
    net, prob = build_objects(meta, sd, xtarget, dev)
    n = args.n or meta["n_full"]
    nt = args.nt or meta["nt"]
    lo, hi = shard_rows(n, rank, world)
    x = make_states(meta, xInit, n, seed=200)[lo:hi].contiguous().to(dev)
    times = [0.1 * k for k in range(1, 10)]
    shocks = torch.zeros(1, meta["d"], device=dev)
    shocks[0, 0:3] = torch.tensor([0.5, -0.5, 0.25])
    kw = {"group": True} if world > 1 else {}
    for _ in range(max(1, args.warmup)):
        shock_sweep(x, net, prob, nt, times, shocks, alph=meta["alph"], **kw)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = shock_sweep(x, net, prob, nt, times, shocks, alph=meta["alph"], **kw)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist:
        host_reduce = os.environ.get("NOCF_BENCH_BACKEND", "nccl") != "nccl"
        tt = torch.tensor([el], dtype=torch.float64, device="cpu" if host_reduce else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    if rank == 0:
        out = {"metric": "shocked trajectories/sec (two-segment rollouts with trajectories and controls kept)",
               "value": len(times) * n * args.steps / el, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic states; pretrained reference weights exported to npz",
               "config": {"workload": f"singlequad-shock d={meta['d']} m={meta['m']} nt={nt} n={n} x {len(times)} shock times, rk4, eval-mode, fp32, "
                                      "intermediates=True (zFull + ctrlFull written)", "rows_per_gpu": hi - lo,
                          "Jc_last_segment": float(res[-1]["costs2"][0])}}
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


def quick_measure(name, dev, steps=10, warmup=3, n_rows=0):
    """one of the other BASELINE.json configurations at its full size, a few calls (the headline stays swarm50): what the driver's line
    carries in config.other_workloads"""
    L = _lib.lib()
    if name == "singlequad-shock":
        from neuraloc_amd.shock import shock_sweep
        meta, sd, xtarget, xInit = load_workload("singlequad")
        net, prob = build_objects(meta, sd, xtarget, dev)
        n, nt = meta["n_full"], meta["nt"]
        x = make_states(meta, xInit, n, seed=200).to(dev)
        times = [0.1 * k for k in range(1, 10)]
        shocks = torch.zeros(1, meta["d"], device=dev)
        shocks[0, 0:3] = torch.tensor([0.5, -0.5, 0.25])
        for _ in range(3):
            shock_sweep(x, net, prob, nt, times, shocks, alph=meta["alph"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            shock_sweep(x, net, prob, nt, times, shocks, alph=meta["alph"])
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        return {"workload": f"singlequad-shock n={n} x {len(times)} shock times nt={nt} (trajectories and controls kept)",
                "traj_per_s": len(times) * n / el, "ms_per_sweep": 1e3 * el, "kernel": L.nocf_last_rollout_kernel().decode()}
    meta, sd, xtarget, xInit = load_workload(name)
    net, prob = build_objects(meta, sd, xtarget, dev)
    n, nt = n_rows or meta["n_full"], meta["nt"]
    x = make_states(meta, xInit, n, seed=200).to(dev)
    with torch.no_grad():
        for _ in range(warmup):
            na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", meta["alph"])
        torch.cuda.synchronize()
        # three rounds of `steps` calls, the MEDIAN round is reported (a secondary line: one host hiccup in a 10-call round of a 0.1 ... 1 ms
        # workload is a 5x error -- seen once in round 5; the headline keeps the contract's single timed region of exactly K steps)
        rounds = []
        for _ in range(3):
            L.nocf_profile_begin()
            t0 = time.perf_counter()
            for _ in range(steps):
                na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", meta["alph"])
            torch.cuda.synchronize()
            el_r = (time.perf_counter() - t0) / steps
            kms, nl = C.c_double(0.0), C.c_int32(0)
            L.nocf_profile_end(C.byref(kms), C.byref(nl))
            rounds.append((el_r, kms.value / max(1, nl.value)))
        rounds.sort()
        el, kernel_ms = rounds[1]
    achieved = flops_per_state_step(meta) * n * nt / (kernel_ms * 1e-3) / 1e12
    return {"workload": f"{name} d={meta['d']} m={meta['m']} nt={nt} n={n}", "traj_per_s": n / el, "ms_per_step": 1e3 * el,
            "kernel": L.nocf_last_rollout_kernel().decode(), "kernel_ms": kernel_ms,
            "protocol": f"median of 3 rounds of {steps} calls", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "frac_of": "fp32 MFMA / vector peak 157.3 TFLOP/s (the small networks are latency-bound VALU work: SURVEY 8d)"}


def train_measure(name, dev, reps=10):
    """One training iteration of trainOC.py:170-174 (zero_grad, Jc = OCflow(...), Jc.backward(), Adam step) at the workload's full size,
    prob.train(): wall time per iteration, the two rollout kernels' own times (HIP events recorded by the library around the recording
    forward and around the adjoint) and the fraction of the fp32 MFMA roof: an iteration is 3 x the forward's algorithmic FLOPs (forward,
    vector-Jacobian product of grad Phi, weight-gradient outer products)."""
    L = _lib.lib()
    meta, sd, xtarget, xInit = load_workload(name)
    net, prob = build_objects(meta, sd, xtarget, dev)
    net.train(); prob.train()
    n, nt, alph = meta["n_full"], meta["nt"], meta["alph"]
    x = make_states(meta, xInit, n, seed=200).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)

    def window(fn):
        L.nocf_profile_begin()
        r = fn()
        kms, nl = C.c_double(0.0), C.c_int32(0)
        L.nocf_profile_end(C.byref(kms), C.byref(nl))
        return r, kms.value, L.nocf_last_rollout_kernel().decode()

    for _ in range(3):
        opt.zero_grad()
        Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph)
        Jc.backward()
        opt.step()
    torch.cuda.synchronize()
    # wall time: the loop as a training run executes it (the host runs ahead of the GPU; one synchronisation at the end)
    t0 = time.perf_counter()
    for _ in range(reps):
        opt.zero_grad()
        Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph)
        Jc.backward()
        opt.step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    # the two rollout kernels' own times: separate iterations (reading a profile window synchronises, which would sit inside the wall time)
    fwd_ms = bwd_ms = 0.0
    kreps = 3
    from neuraloc_amd import train as _train
    _train.VENDOR_GEMM["events"].clear()
    _train.VENDOR_GEMM["on"] = True
    for _ in range(kreps):
        opt.zero_grad()
        (Jc, _), ms, fk = window(lambda: na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph))
        fwd_ms += ms
        _, ms, bk = window(lambda: Jc.backward())
        bwd_ms += ms
        opt.step()
    torch.cuda.synchronize()
    _train.VENDOR_GEMM["on"] = False
    vendor_ms = sum(e0.elapsed_time(e1) for e0, e1 in _train.VENDOR_GEMM["events"]) / kreps
    _train.VENDOR_GEMM["events"].clear()
    na.check_errors(sync=True)
    fl = 3.0 * flops_per_state_step(meta) * n * nt
    return {"workload": f"train {name} d={meta['d']} m={meta['m']} nt={nt} n={n} (Adam step, prob.train())", "train_iter_ms": 1e3 * el,
            "trained_traj_per_s": n / el, "forward_kernel": fk, "forward_kernel_ms": fwd_ms / kreps,
            "adjoint_kernel": bk, "adjoint_kernel_ms": bwd_ms / kreps, "flops_per_iteration": fl,
            "vendor_gemm_ms": vendor_ms,
            "vendor_gemm_note": "weight-gradient contractions X'Y over the adjoint's row streams that run as hipBLASLt GEMMs (torch.bmm + fixed-order slab "
                                "sum, neuraloc_amd/train.py:_contract); 0 = every gradient was accumulated inside the hand-written adjoint. Kept for "
                                "swarm50: the library runs them at ~0.9 of the fp32 MFMA roof, the in-kernel alternative (NOCF_DUO_DW=1) is slower",
            "frac": fl / el / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "frac_of": "3 x SURVEY 8(d) forward FLOPs per iteration over the WHOLE iteration's wall time, of the fp32 MFMA peak", "Jc": float(Jc.detach())}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a child process (this parent has not
    touched the GPU: nothing here calls into HIP before this point), relay their output, leave with their exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", type=str, default="swarm50",
                    choices=["swap2", "softcorridor", "swap12", "swarm50", "singlequad", "singlequad-shock"])
    ap.add_argument("--scaling", type=str, default="strong", choices=["strong", "weak"])
    ap.add_argument("--n", type=int, default=0, help="batch rows: the GLOBAL batch (strong) or rows per GPU (weak); default: BASELINE.json's n")
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the short runs of the other four BASELINE configs, the shock sweep and the training iterations")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
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
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: run `python bench.py --gpus N` (it starts its own ranks) or torch.distributed.run with N ranks"
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.workload == "singlequad-shock":
        return bench_shock(args, dev, world, rank, dist)

    meta, sd, xtarget, xInit = load_workload(args.workload)
    n_arg = args.n or meta["n_full"]
    nt = args.nt or meta["nt"]
    alph = meta["alph"]
    net, prob = build_objects(meta, sd, xtarget, dev)
    if args.scaling == "strong":
        n_global = n_arg
        lo, hi = shard_rows(n_global, rank, world)
        x_cpu = make_states(meta, xInit, n_global, seed=200)[lo:hi].contiguous()     # every rank: its rows of the SAME global batch
    else:
        n_global = n_arg * world
        x_cpu = make_states(meta, xInit, n_arg, seed=200 + rank)
    n_local = x_cpu.shape[0]
    x = x_cpu.to(dev)                                   # inputs resident in HBM before the timed region

    def step():
        if world > 1:
            return OCflow_sharded(x, net, prob, [0.0, 1.0], nt, "rk4", alph)     # local rollout + one all-reduce of 8 floats
        return na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", alph)

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
    na.check_errors(sync=True)                         # a rollout whose workgroups timed out on each other raises here (its numbers are NaN)
    if dist:
        host_reduce = os.environ.get("NOCF_BENCH_BACKEND", "nccl") != "nccl"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_reduce else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        total_traj = n_global * args.steps
        value = total_traj / elapsed
        kernel_ms = kms.value / max(1, nl.value)
        kernel = L.nocf_last_rollout_kernel().decode()
        fl_launch = flops_per_state_step(meta) * n_local * nt
        achieved = fl_launch / (kernel_ms * 1e-3) / 1e12
        alg_bytes = 8 * (meta["d"] + 4) * n_local * nt  # state in/out once per step
        traffic, traffic_src = read_traffic(args.workload, kernel, n_local)
        out = {
            "metric": "trajectories/sec (n_train x nt states integrated)", "value": value, "unit": "trajectories/s",
            "n_gpus": world, "ranks": (dist.get_world_size() if dist else 1),
            "backend": (os.environ.get("NOCF_BENCH_BACKEND", "nccl") + (" (RCCL)" if os.environ.get("NOCF_BENCH_BACKEND", "nccl") == "nccl" else "")) if dist else "none (single process)",
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic states (xInit + var0 * closed-form normal table); pretrained reference weights exported to npz",
            "config": {"workload": f"{args.workload} d={meta['d']} m={meta['m']} nTh={meta['nTh']} nt={nt} "
                                   f"n={n_global} global ({n_local}/GPU) rk4 eval-mode fp32", "global_batch": n_global,
                       "rows_per_gpu": n_local, "state_steps_per_s": value * nt,
                       "parallelism": f"batch rows sharded x{world}, one 8-float SUM all-reduce per call", "Jc": float(Jc)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel, "kernel_ms": kernel_ms, "launches_timed": nl.value,
                         "algorithmic_flops_per_launch": fl_launch,
                         "algorithmic_hbm_bytes_per_launch": alg_bytes,
                         "achieved_hbm_GBps_algorithmic": alg_bytes / (kernel_ms * 1e-3) / 1e9,
                         "hbm_peak_GBps": PEAK_HBM_GBS,
                         "note": "per launch on rank 0: its rows x nt state-steps x SURVEY 8(d) FLOPs, over the kernel's HIP-event time"},
        }
        if world == 1 and args.workload == "swarm50" and not args.n and not args.nt and not args.no_other_workloads:
            others = []
            for name in ("swap2", "softcorridor", "swap12", "singlequad", "singlequad-shock"):
                try:
                    others.append(quick_measure(name, dev))
                except Exception as ex:                          # the headline line must survive a failure here
                    others.append({"workload": name, "error": repr(ex)[:200]})
            # the per-rank batches of the strong-scaling partition of the headline config (n_train = 1024 over 2 / 4 / 8 GPUs) and of
            # config 5 at 8 GPUs, on this one GPU: what a rank's rollout costs when the node is not there to measure it
            for name, rows in (("swarm50", 512), ("swarm50", 256), ("swarm50", 128), ("singlequad", 512)):
                try:
                    r = quick_measure(name, dev, steps=50, warmup=10, n_rows=rows)
                    r["proxy_for"] = f"one rank's batch of {name} at {1024 // rows if name == 'swarm50' else 4096 // rows} GPUs (strong scaling)"
                    others.append(r)
                except Exception as ex:
                    others.append({"workload": f"{name} n={rows}", "error": repr(ex)[:200]})
            for name in ("swarm50", "singlequad"):               # training: the reference's main use (trainOC.py:160-176)
                try:
                    others.append(train_measure(name, dev))
                except Exception as ex:
                    others.append({"workload": "train " + name, "error": repr(ex)[:200]})
            out["config"]["other_workloads"] = others
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(meta, sd, xtarget, x_cpu, nt)
            out["config"]["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
