#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mono_gpu.py -q -x 2>&1 | grep -v amdgpu | tail -12
for mono in 1 0; do
  echo "== NOCF_MONO=$mono"
  NOCF_MONO=$mono timeout 300 python bench.py --workload singlequad --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernel'], 'kernel_ms=%.3f' % j['roofline']['kernel_ms'], 'traj/s=%.0f' % j['value'], 'frac=%.3f' % j['roofline']['frac'], 'Jc=%.6f' % j['config']['Jc'])"
  NOCF_MONO=$mono timeout 300 python bench.py --workload singlequad-shock --steps 5 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('shock sweep ms=%.2f traj/s=%.0f' % (j['ms_per_step'], j['value']))"
done
