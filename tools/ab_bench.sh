#!/bin/bash
# A/B the production library against experimental builds on one box (bench without the CPU leg)
cd "$(dirname "$0")/.."
for lib in "$@"; do
  echo "== $lib"
  NOCF_LIB_PATH=$PWD/neuraloc_amd/csrc/$lib timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('traj/s %.0f  ms/step %.3f  kernel_ms %.3f  TFLOP/s %.1f frac %.3f Jc %.6f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['config']['Jc']))"
done
