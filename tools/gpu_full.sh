#!/bin/bash
# GPU box: the whole -m gpu suite, then the default bench line and the strong-scaling proxy table
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_default.json
cat gpurun_out/bench_default.json
: > gpurun_out/proxy_table.txt
for n in 1024 512 256 128; do
  timeout 300 python bench.py --n $n --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/proxy_table.txt
done
NOCF_SLAB=2 timeout 300 python bench.py --n 1024 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/proxy_table.txt
NOCF_SLAB=0 timeout 300 python bench.py --n 128 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/proxy_table.txt
python - <<'PY'
import json
for line in open("gpurun_out/proxy_table.txt"):
    try:
        j = json.loads(line)
        print("n=%d kernel=%s kernel_ms=%.3f ms_per_step=%.3f traj/s=%.0f frac=%.3f" % (j["config"]["rows_per_gpu"], j["roofline"]["kernel"], j["roofline"]["kernel_ms"], j["ms_per_step"], j["value"], j["roofline"]["frac"]))
    except Exception as e:
        print("ERR", line[:300])
PY
