#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/slab_stamps.txt
for fast in 1; do for n in 512 1024; do
  NOCF_SLAB=2 NOCF_SLAB_FAST=$fast NOCF_LIB_PATH=neuraloc_amd/csrc/libnocf_stamps.so timeout 300 python tools/slab_stamps.py $n 2>&1 | grep -v amdgpu.ids >> gpurun_out/slab_stamps.txt
done; done
cat gpurun_out/slab_stamps.txt
