#!/bin/bash
# stress of the exchange protocols: long runs of the slab kernel (both forms) and repeated parity suites; every Jc must be identical
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for n in 1024 512 100; do
  for fast in 1 0; do
    NOCF_SLAB=2 NOCF_SLAB_FAST=$fast timeout 600 python bench.py --n $n --steps 1500 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('n=$n fast=$fast steps=1500 kernel_ms=%.3f Jc=%.9e' % (j['roofline']['kernel_ms'], j['config']['Jc']))"
  done
done
timeout 600 python bench.py --workload singlequad --steps 3000 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('singlequad steps=3000 kernel_ms=%.3f Jc=%.9e' % (j['roofline']['kernel_ms'], j['config']['Jc']))"
for i in 1 2 3; do timeout 900 python -m pytest tests/test_slab_gpu.py tests/test_mono_gpu.py -q 2>&1 | tail -1; done
