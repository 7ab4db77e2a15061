#!/bin/bash
# stress of the exchange protocols: long runs of the split-role kernel (both exchange forms, both role maps, 1 / 2 / 4 tiles per group) and
# repeated parity suites; every Jc of a configuration must be identical from call to call (bench.py raises if a rollout timed out)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for n in 2048 1024 512 100; do
  for knobs in "NOCF_DUO_FAST=1" "NOCF_DUO_FAST=0" "NOCF_DUO_MAP=1"; do
    env $knobs timeout 900 python bench.py --n $n --steps 1500 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('n=$n $knobs steps=1500 kernel_ms=%.3f Jc=%.9e' % (j['roofline']['kernel_ms'], j['config']['Jc']))"
  done
done
timeout 600 python bench.py --workload singlequad --steps 3000 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('singlequad steps=3000 kernel_ms=%.3f Jc=%.9e' % (j['roofline']['kernel_ms'], j['config']['Jc']))"
for i in 1 2 3; do timeout 900 python -m pytest tests/test_duo_gpu.py tests/test_mono_gpu.py -q 2>&1 | tail -1; done
