#!/bin/bash
# GPU box: swarm50 rollout time by batch rows on one GPU (the per-rank batches of the strong-scaling partition) + the training iteration
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp NOCF_JIT=0
O=${1:-gpurun_out/proxy}
mkdir -p $O
: > $O/proxy_table.jsonl
for n in ${NS:-4096 2048 1024 512 256 128}; do timeout 300 python bench.py --n $n --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/proxy_table.jsonl; done
python - "$O" <<'PY' | tee $O/proxy_table.txt
import json, sys
print("swarm50 nt=80 on ONE MI355X by batch rows (bench.py --n ROWS): 512 / 256 / 128 = the per-rank batch of n=1024 at 2 / 4 / 8 GPUs")
for line in open(sys.argv[1] + "/proxy_table.jsonl"):
    try:
        j = json.loads(line)
        print("rows/GPU=%4d  kernel=%-22s kernel_ms=%.3f  ms_per_step=%.3f  traj/s=%8.0f  roofline.frac=%.3f" % (j["config"]["rows_per_gpu"], j["roofline"]["kernel"], j["roofline"]["kernel_ms"], j["ms_per_step"], j["value"], j["roofline"]["frac"]))
    except Exception as e:
        print("ERR", line[:200])
PY
