#!/usr/bin/env python3
"""Register / scratch table of every kernel instantiation of the shipped library (and, with --jit d m nTh r agents, of one per-shape JIT
build): compiles the translation units with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one line per kernel.
   python tools/isa_table.py [--jit 8 48 2 9 4] > profiles/rNN/isa_table.txt"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "neuraloc_amd", "csrc")


def demangle(names):
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not filt:
        return names
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return out[:len(names)]


def usage(src, extra):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as td:
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-c", src,
               "-o", os.path.join(td, "x.o"), "-Rpass-analysis=kernel-resource-usage"] + extra
        txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for ln in txt.splitlines():
        m = re.search(r"remark:\s*Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return rows


def main():
    extra, tag = [], "shipped library"
    srcs = ["nocf_kernels.hip", "nocf_duo.hip"]
    if "--jit" in sys.argv:
        i = sys.argv.index("--jit")
        d, m, t, r, a = (int(v) for v in sys.argv[i + 1:i + 6])
        extra = ["-DNOCF_JIT_ONLY", f"-DNOCF_XS_D={d}", f"-DNOCF_XS_M={m}", f"-DNOCF_XS_T={t}", f"-DNOCF_XS_R={r}", f"-DNOCF_XS_A={a}"]
        tag, srcs = f"per-shape JIT build d={d} m={m} nTh={t} r={r} agents={a}", ["nocf_kernels.hip"]
    print(f"# {tag}: hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage (ROCm clang), one line per __global__ instantiation")
    print("# %-118s %5s %5s %5s %8s %7s %7s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratchB", "Vspill", "Sspill", "occ"))
    for s in srcs:
        rows = usage(os.path.join(CSRC, s), extra)
        names = demangle([r["name"] for r in rows])
        for r, nm in zip(rows, names):
            nm = re.sub(r"\(.*$", "", nm).replace("void ", "")
            print("%-120s %5d %5d %5d %8d %7d %7d %4d" % (nm[:120], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1),
                                                         r.get("ScratchSize [bytes/lane]", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1),
                                                         r.get("Occupancy [waves/SIMD]", -1)))


if __name__ == "__main__":
    main()
