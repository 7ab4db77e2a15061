#!/usr/bin/env python3
"""Diagnostic: the weight-gradient contractions of the swarm50 training step (X' Y over K = 322 x 1024 rows) as one library GEMM
against a split over K through a batched GEMM + sum.   python tools/gemm_splitk_probe.py"""
import time
import torch

dev = torch.device("cuda:0")
K = 322 * 1024
for m, n in ((512, 512), (512, 151)):
    X = torch.randn(K, m, device=dev)
    Y = torch.randn(K, n, device=dev)

    def t(f, reps=5):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, r
    base, r0 = t(lambda: X.t() @ Y)
    line = f"X'[{m} x {K}] Y[{K} x {n}]: one GEMM {base:.3f} ms ({2 * m * n * K / base / 1e9:.1f} TFLOP/s)"
    for S in (2, 4, 8, 16, 32):
        ms, r = t(lambda: torch.bmm(X.view(S, K // S, m).transpose(1, 2), Y.view(S, K // S, n)).sum(0))
        err = (r - r0).abs().max().item() / r0.abs().max().item()
        line += f" | split {S}: {ms:.3f} ms (rel diff {err:.1e})"
    print(line, flush=True)
