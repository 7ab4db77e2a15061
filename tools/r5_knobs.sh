#!/bin/bash
# GPU box (round 5): how early the predictive nap ends (NOCF_DUO_PWSH: 0 = 3/4 of the interval, 1 = 7/8, 2 = 15/16) and the adjoint's nap in
# front of its first abar0 poll (NOCF_DUO_BWD_UDELAY)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp NOCF_JIT=0
O=${1:-gpurun_out/r5_knobs}
mkdir -p $O
: > $O/knobs.txt
for cfg in "16 128" "16 256" "8 512"; do
  set -- $cfg
  for S in 0 1 2; do
    r=$(NOCF_DUO_G=$1 NOCF_DUO_PWSH=$S timeout 300 python bench.py --n $2 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f" % j["roofline"]["kernel_ms"])')
    echo "G=$1 n=$2 pwsh=$S kernel_ms=$r" | tee -a $O/knobs.txt
  done
done
for U in 0 4 6 8; do
  r=$(NOCF_DUO_BWD_UDELAY=$U timeout 300 python tools/time_train.py swarm50 8 2>/dev/null | grep "^{" | tail -1)
  echo "bwd_udelay=$U $r" | tee -a $O/knobs.txt
done
