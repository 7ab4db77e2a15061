#!/usr/bin/env python3
"""GPU box: rollout time of swarm50-shaped networks of other widths (closed-form weights) on the swarm50 problem, n rows, nt = 80: the
split-role kernel (m = 256: four members per group) against the per-tile kernel (NOCF_DUO=0).
   python tools/time_width.py [m=256] [n=1024]"""
import ctypes as C
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ["NOCF_ENV_WATCH"] = "1"
os.environ.setdefault("NOCF_JIT", "0")
import torch                                   # noqa: E402
import bench                                   # noqa: E402
import neuraloc_amd as na                      # noqa: E402
from neuraloc_amd import _lib                  # noqa: E402
from util_hip import synth_state_dict          # noqa: E402


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    meta, sd0, xtarget, xInit = bench.load_workload("swarm50")
    dev = torch.device("cuda:0")
    _, prob = bench.build_objects(meta, sd0, xtarget, dev)
    net = na.Phi(nTh=2, m=m, d=meta["d"], alph=meta["alph"])
    net.load_state_dict(synth_state_dict(2, m, meta["d"], seed=2))
    net = net.to(dev).eval()
    x = bench.make_states(meta, xInit, n, 200).to(dev)
    nt = meta["nt"]
    mm = dict(meta); mm["m"] = m
    fl = bench.flops_per_state_step(mm) * n * nt
    L = _lib.lib()
    for duo in ("1", "0"):
        os.environ["NOCF_DUO"] = duo
        with torch.no_grad():
            for _ in range(3):
                na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", meta["alph"])
            torch.cuda.synchronize()
            L.nocf_profile_begin()
            t0 = time.perf_counter()
            for _ in range(20):
                Jc, _ = na.OCflow(x, net, prob, [0.0, 1.0], nt, "rk4", meta["alph"])
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / 20
            kms, nl = C.c_double(0.0), C.c_int32(0)
            L.nocf_profile_end(C.byref(kms), C.byref(nl))
        k = kms.value / max(1, nl.value)
        print(f"m={m} n={n} nt={nt} NOCF_DUO={duo}: kernel {L.nocf_last_rollout_kernel().decode()} {k:.3f} ms ({1e3 * el:.3f} ms per call), "
              f"{fl / (k * 1e-3) / 1e12:.1f} TFLOP/s = {fl / (k * 1e-3) / 1e12 / bench.PEAK_F32_MFMA_TFLOPS:.3f} of the fp32 MFMA roof, Jc {float(Jc):.6e}")


if __name__ == "__main__":
    main()
