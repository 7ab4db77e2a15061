#!/bin/bash
# GPU box: A/B of two builds of the library on the same box (NOCF_LIB_PATH): rollout time by batch rows
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp NOCF_JIT=0
for n in ${NS:-1024 512 256 128 2048}; do
  for lib in ${LIBS:-neuraloc_amd/csrc/libnocf_prev.so neuraloc_amd/csrc/libnocf.so}; do
    r=$(NOCF_LIB_PATH=$PWD/$lib timeout 300 python bench.py --n $n --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms kernel, %.3f ms per step, Jc-side value %.1f" % (j["roofline"]["kernel_ms"], j["ms_per_step"], j["value"]))')
    echo "n=$n $(basename $lib): $r"
  done
done
