#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_slab_gpu.py -q -x 2>&1 | tail -6 > gpurun_out/slab_pytest.log
cat gpurun_out/slab_pytest.log
: > gpurun_out/slab_table.txt
for fast in 1 0; do
  for n in 128 512 1024; do
    echo "== NOCF_SLAB_FAST=$fast n=$n" >> gpurun_out/slab_table.txt
    NOCF_SLAB=2 NOCF_SLAB_FAST=$fast timeout 300 python bench.py --n $n --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 >> gpurun_out/slab_table.txt
  done
done
python - <<'PY'
import json
for line in open("gpurun_out/slab_table.txt"):
    if line.startswith("=="):
        print(line.strip(), end="  ")
    else:
        try:
            j = json.loads(line)
            print("kernel_ms=%.3f ms_per_step=%.3f frac=%.3f Jc=%.6e" % (j["roofline"]["kernel_ms"], j["ms_per_step"], j["roofline"]["frac"], j["config"]["Jc"]))
        except Exception as e:
            print("ERR", line[:200])
PY
