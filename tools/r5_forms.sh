#!/bin/bash
# GPU box (round 5): the two group geometries of the split-role forward (NOCF_DUO_G = 8 / 16) -- parity tests of the kernel file, then
# the rollout time by batch rows for both (the per-rank batches of the strong-scaling partition: 512 / 256 / 128 rows)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp NOCF_JIT=0
O=${1:-gpurun_out/r5_forms}
mkdir -p $O
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout ${TEST_TIMEOUT:-1500} python -m pytest tests/test_duo_gpu.py -x -q -m gpu ${PYTEST_ARGS:-} > $O/pytest_duo.log 2>&1
  echo "pytest rc $?" >> $O/pytest_duo.log
  tail -5 $O/pytest_duo.log
fi
for G in ${GS:-16 8}; do
  : > $O/proxy_g$G.jsonl
  for n in ${NS:-1024 512 256 128 2048 4096}; do
    NOCF_DUO_G=$G timeout 300 python bench.py --n $n --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/proxy_g$G.jsonl
  done
done
python - "$O" <<'PY' | tee $O/proxy_table.txt
import json, sys, glob, os
print("swarm50 nt=80 on ONE MI355X by batch rows (bench.py --n ROWS) and group geometry (NOCF_DUO_G): 512 / 256 / 128 = the per-rank batch of n=1024 at 2 / 4 / 8 GPUs")
for f in sorted(glob.glob(sys.argv[1] + "/proxy_g*.jsonl")):
    G = os.path.basename(f)[7:-6]
    for line in open(f):
        try:
            j = json.loads(line)
            print("G=%-2s rows/GPU=%4d  kernel=%-22s kernel_ms=%.3f  ms_per_step=%.3f  traj/s=%8.0f  roofline.frac=%.3f" % (G, j["config"]["rows_per_gpu"], j["roofline"]["kernel"], j["roofline"]["kernel_ms"], j["ms_per_step"], j["value"], j["roofline"]["frac"]))
        except Exception as e:
            print("ERR", line[:200])
PY
