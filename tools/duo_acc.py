#!/usr/bin/env python3
"""Diagnostic (-DNOCF_ACC build of nocf_duo.hip): where does every wave of the split-role kernel spend the rollout?  Every timeline point of the
kernel is a LAP counter in this build: the shader clocks since the wave's previous point are added to the point's bucket over the whole
rollout.  Prints, per role, the mean over all waves of clocks per tile and evaluation in each bucket (and its share), split by owner /
non-owner waves for role A's tile-0 owners.
   build:  cd neuraloc_amd/csrc && hipcc ... -DNOCF_ACC=1 -c nocf_duo.hip -o /tmp/duo_acc.o; hipcc ... -DNOCF_STAMPS=1 -c nocf_kernels.hip -o /tmp/k.o (the buffer setter) \\
           && hipcc --offload-arch=gfx950 -shared -fPIC -o libnocf_acc.so obj/nocf_kernels.o /tmp/duo_acc.o
   run:    python tools/duo_acc.py [n]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("NOCF_LIB_PATH", os.path.join(REPO, "neuraloc_amd", "csrc", "libnocf_acc.so"))
os.environ["NOCF_JIT"] = "0"
import torch                                   # noqa: E402
import bench                                   # noqa: E402
import neuraloc_amd as na                      # noqa: E402
from neuraloc_amd import _lib                  # noqa: E402

A_PT = {0: "own: entry (from the previous point of this wave)", 1: "own: WAIT for the partial gradients", 2: "own: sum, RK update, S stored", 3: "own: cost part (incl. wait for role B's scalars)",
        4: "to the S gather", 13: "S gather: polls + staging (the two gatherer waves)", 5: "S gather: the workgroup barrier behind it", 6: "P1 product", 7: "P1 epilogue + U / TH stores",
        8: "to the U gather", 9: "owner: z / A^T z (+ cost part) done", 14: "U gather: nap + polls + staging", 10: "U gather: the barrier behind it", 11: "P2 product",
        12: "P2 epilogue: ack wait, tanh, V store", 31: "tail (terminal costs, outputs)"}
B_PT = {20: "tile entry (from the previous point)", 21: "own states: polls + scatter + barrier", 22: "pair sums + barrier", 23: "QW combine + store",
        24: "to the V gather", 15: "V gather: polls + staging", 25: "V gather: the barrier behind it", 26: "P3 product (resets, tanh(o) request inside)", 27: "tanh(o) wait, y written",
        28: "y barrier (+ P4 operand reads)", 29: "P4 products + G stores", 31: "tail"}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    meta, sd, xtarget, xInit = bench.load_workload("swarm50")
    dev = torch.device("cuda:0")
    net, prob = bench.build_objects(meta, sd, xtarget, dev)
    x = bench.make_states(meta, xInit, n, 200).to(dev)
    buf = torch.zeros(32 * 16 * 2 * 4 * 32, dtype=torch.int64, device=dev)
    assert _lib.lib().nocf_debug_set_stamp_buffer(buf.data_ptr()) == 0, "this is not a diagnostic build"
    with torch.no_grad():
        for _ in range(2):
            buf.zero_()
            na.OCflow(x, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
        torch.cuda.synchronize()
        L = _lib.lib()
        import ctypes as C
        L.nocf_profile_begin()
        na.OCflow(x, net, prob, [0.0, 1.0], meta["nt"], "rk4", meta["alph"])
        kms, nl = C.c_double(0.0), C.c_int32(0)
        L.nocf_profile_end(C.byref(kms), C.byref(nl))
    print("kernel:", L.nocf_last_rollout_kernel().decode(), " n =", n, " kernel_ms (this build) = %.3f" % (kms.value / max(1, nl.value)))
    t = buf.view(32, 16, 2, 4, 32).cpu().double()            # [group][member][role][wave][bucket]
    used = t.sum(dim=(2, 3, 4)) > 0                          # (group, member) pairs that ran
    ng = int(used.any(dim=1).sum())
    G = int(used[0].sum())
    ntiles = (n + 15) // 16
    NT = (ntiles + ng - 1) // ng
    evals = meta["nt"] * 4 + 1
    print(f"{ng} groups x {G} members, {NT} tile(s) per group, {evals} evaluations; clocks PER TILE AND EVALUATION, mean over waves (min .. max)")
    for role, names in ((0, A_PT), (1, B_PT)):
        sel = t[:ng, :G, role]                               # [g][m][wave][bucket]
        tot = sel.sum(-1)
        print(("role A" if role == 0 else "role B") + f": total per wave {tot.mean() / (evals * NT):8.0f}  ({tot.min() / (evals * NT):.0f} .. {tot.max() / (evals * NT):.0f})")
        for k in sorted(names):
            v = sel[..., k] / (evals * NT)
            if float(v.max()) == 0.0:
                continue
            line = f"  {k:2d} {names[k]:58s} {v.mean():8.0f}  ({v.min():6.0f} .. {v.max():6.0f})  {100.0 * v.mean() * evals * NT / tot.mean():5.1f} %"
            if role == 0 and NT == 2:
                line += "   waves 0,1: %6.0f   waves 2,3: %6.0f" % (v[..., :2].mean(), v[..., 2:].mean())
            print(line)


if __name__ == "__main__":
    main()
