#!/bin/bash
# bench line (without the CPU leg) for each BASELINE.json config on one box
cd "$(dirname "$0")/.."
for w in swap2 softcorridor swap12 swarm50 singlequad; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-13s traj/s %10.0f  ms/step %8.3f  kernel_ms %8.3f  TFLOP/s %6.2f  Jc %.6f' % ('$w', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['config']['Jc']))"
done
