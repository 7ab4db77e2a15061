#!/bin/bash
# A/B geometry knobs (NOCF_SUBTILES, NOCF_NWAVES) of one library on one box
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('traj/s %.0f  ms/step %.3f  kernel_ms %.3f  TFLOP/s %.1f frac %.3f Jc %.6f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['config']['Jc']))"
done
