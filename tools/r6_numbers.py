#!/usr/bin/env python3
"""Markdown table of a bench.py line (and the batch-row table) for DESIGN.md section 6:  python tools/r6_numbers.py profiles/r6"""
import json
import os
import sys


def main():
    d = sys.argv[1]
    j = json.loads([l for l in open(os.path.join(d, "10_bench_default.json")) if l.startswith("{")][-1])
    r, cb = j["roofline"], j["cpu_baseline"]
    print(f"| what | number | source |")
    print(f"|---|---|---|")
    print(f"| headline: swarm50 nt = 80 n = 1024, one GPU | **{j['value'] / 1e3:.1f} k trajectories/s**, {j['ms_per_step']:.3f} ms per call, kernel `{r['kernel']}` {r['kernel_ms']:.3f} ms | `10_bench_default.json` |")
    print(f"| roofline (fp32 MFMA, 157.3 TFLOP/s) | achieved {r['achieved']:.1f} TFLOP/s = **{r['frac']:.3f}**; algorithmic HBM bytes {r['algorithmic_hbm_bytes_per_launch'] / 1e6:.1f} MB per launch = {r['achieved_hbm_GBps_algorithmic']:.1f} GB/s of 8 000 | same |")
    if r.get("traffic"):
        print(f"| HBM traffic from the PMC passes | {r['traffic'] / 1e9:.2f} GB per launch (write-backs of the exchange area: section 3.2) | `03_hbm_traffic_n1024_duo.json` |")
    print(f"| CPU baseline (the oracle, `kind: port`) | {cb['value']:.1f} trajectories/s at {cb['cores']} of {cb['os_cpu_count']} threads ({cb['cpu_model']}), median of {len(cb['full_rollouts_s'])} full rollouts ({cb['full_rollout_s']:.2f} s) -> {j['value'] / cb['value']:.0f}x | `10_bench_default.json` |")
    for w in j["config"].get("other_workloads", []):
        name = w["workload"]
        if "train_iter_ms" in w:
            print(f"| {name} | **{w['train_iter_ms']:.2f} ms** per iteration: recording forward {w['forward_kernel_ms']:.2f} + adjoint {w['adjoint_kernel_ms']:.2f} (`{w['adjoint_kernel']}`) + library GEMMs {w['vendor_gemm_ms']:.2f}; {w['frac']:.3f} of the roof over the iteration | same |")
        elif "kernel_ms" in w:
            print(f"| {name} | {w['ms_per_step']:.4f} ms per call, kernel `{w['kernel']}` {w['kernel_ms']:.4f} ms, frac {w['frac']:.3f} | same |")
        else:
            print(f"| {name} | {w['traj_per_s'] / 1e6:.2f} M shocked trajectories/s" + (f", {w['ms_per_step']:.2f} ms per sweep" if 'ms_per_step' in w else "") + " | same |")
    pt = os.path.join(d, "05_proxy_table.txt")
    if os.path.exists(pt):
        rows = [l.split() for l in open(pt) if l.startswith("rows/GPU")]
        cells = ", ".join(f"{l[0].split('=')[1] if '=' in l[0] and l[0].split('=')[1] else l[1]}: {[x for x in l if x.startswith('kernel_ms=')][0].split('=')[1]}" for l in rows)
        print(f"| swarm50 by batch rows (kernel ms) | {cells} | `05_proxy_table.txt` |")


if __name__ == "__main__":
    main()
