#!/bin/bash
# GPU box (round 5): the nap in front of the first U poll (NOCF_DUO_UDELAY, in 64-clock quanta) by batch rows and geometry
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp NOCF_JIT=0
O=${1:-gpurun_out/r5_udelay}
mkdir -p $O
: > $O/udelay.txt
for cfg in "16 128" "16 256" "8 512" "8 1024"; do
  set -- $cfg
  for U in ${US:-0 2 4 6 8 12}; do
    r=$(NOCF_DUO_G=$1 NOCF_DUO_UDELAY=$U timeout 300 python bench.py --n $2 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f" % j["roofline"]["kernel_ms"])')
    echo "G=$1 n=$2 udelay=$U kernel_ms=$r" | tee -a $O/udelay.txt
  done
done
