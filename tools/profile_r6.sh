#!/bin/bash
# GPU box: the round-6 evidence under gpurun_out/r6e (copied to profiles/r6 afterwards).  Every step under its own timeout.
#   bash tools/profile_r6.sh [all|quick]      quick = kernel stats + PMC + bench line + proxy rows (no test suite, no stress)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp NOCF_JIT=0
MODE=${1:-all}
O=gpurun_out/r6e
mkdir -p $O
T="timeout 600"
# 1. kernel-trace stats of the default bench command (swarm50 n = 1024: split-role kernel), of the 128-row shard (fine geometry), the shock sweep
#    and of one training iteration each
$T rocprofv3 --kernel-trace --stats -d $O/prof_n1024 -o n1024 --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/prof_n1024.log 2>&1
find $O/prof_n1024 -name "*kernel_stats.csv" -exec cp {} $O/02_n1024_duo_kernel_stats.csv \;
$T rocprofv3 --kernel-trace --stats -d $O/prof_n128 -o n128 --output-format csv -- python3 bench.py --n 128 --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/prof_n128.log 2>&1
find $O/prof_n128 -name "*kernel_stats.csv" -exec cp {} $O/02_n128_duo_kernel_stats.csv \;
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_train -o tr --output-format csv -- python3 tools/time_train.py swarm50 5 > $O/prof_train.log 2>&1
find $O/prof_train -name "*kernel_stats.csv" -exec cp {} $O/08_train_swarm50_kernel_stats.csv \;
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_train_sq -o tr --output-format csv -- python3 tools/time_train.py singlequad 5 > $O/prof_train_sq.log 2>&1
find $O/prof_train_sq -name "*kernel_stats.csv" -exec cp {} $O/08_train_singlequad_kernel_stats.csv \;
# 2. PMC passes (separate runs, counters only): HBM traffic of the n = 1024 launch
$T rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_1024 -o f --output-format csv -- python3 bench.py --n 1024 --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/pmc_fetch_1024.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_1024 -o w --output-format csv -- python3 bench.py --n 1024 --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/pmc_write_1024.log 2>&1
python tools/parse_pmc.py $O/pmc_fetch_1024 $O/pmc_write_1024 swarm50 $O/03_hbm_traffic_n1024_duo.json "rollout_duo_kernel" "rollout_duo_kernel" 1024 "profiles/r6/03_hbm_traffic_n1024_duo.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes" > $O/03_parse_1024.log 2>&1
cp $O/03_hbm_traffic_n1024_duo.json profiles/hbm_traffic_swarm50.json 2>/dev/null
# 3. the default bench line (CPU leg, the other workloads, the shard rows, the training iterations), and the batch-row table
$T python bench.py > $O/10_bench_default.json 2> $O/10_bench_default.err
: > $O/05_proxy_table.jsonl
for n in 4096 2048 1024 512 256 128; do timeout 300 python bench.py --n $n --steps 50 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 >> $O/05_proxy_table.jsonl; done
python - "$O" <<'PY' > $O/05_proxy_table.txt
import json, sys
print("swarm50 nt=80 on ONE MI355X by batch rows (bench.py --n ROWS, the library's own choice of geometry): 512 / 256 / 128 = the per-rank batch of n=1024 at 2 / 4 / 8 GPUs")
for line in open(sys.argv[1] + "/05_proxy_table.jsonl"):
    try:
        j = json.loads(line)
        print("rows/GPU=%4d  kernel=%-22s kernel_ms=%.3f  ms_per_step=%.3f  traj/s=%8.0f  roofline.frac=%.3f" % (j["config"]["rows_per_gpu"], j["roofline"]["kernel"], j["roofline"]["kernel_ms"], j["ms_per_step"], j["value"], j["roofline"]["frac"]))
    except Exception as e:
        print("ERR", line[:200])
PY
timeout 300 python tools/time_train.py swarm50 30 2>&1 | grep "^{" > $O/08_train_times.txt
timeout 300 python tools/time_train.py singlequad 50 2>&1 | grep "^{" >> $O/08_train_times.txt
timeout 300 python bench.py --workload singlequad-shock --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/17_shock_sweep.json
if [ "$MODE" = "all" ]; then
  # 4. the GPU suite and a stress run of the exchange protocols (every Jc of a configuration identical from call to call)
  timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 > $O/09_pytest_gpu.log
  {
    for n in 2048 1024 512 256 128 100; do
      for knobs in "NOCF_DUO_FAST=1" "NOCF_DUO_FAST=0" "NOCF_DUO_MAP=1" "NOCF_DUO_G=16" "NOCF_DUO_G=8" "NOCF_DUO_DBG=8"; do
        env $knobs timeout 600 python bench.py --n $n --steps 600 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('n=$n $knobs steps=600 kernel_ms=%.3f Jc=%.9e' % (j['roofline']['kernel_ms'], j['config']['Jc']))"
      done
    done
    timeout 600 python bench.py --workload singlequad --steps 2000 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('singlequad steps=2000 kernel_ms=%.3f Jc=%.9e' % (j['roofline']['kernel_ms'], j['config']['Jc']))"
    timeout 600 python tools/duo_determinism.py 10 2>&1 | grep -v amdgpu.ids | tail -12
  } > $O/11_stress.txt 2>&1
fi
rm -rf $O/prof_n1024 $O/prof_n128 $O/prof_train $O/prof_train_sq $O/pmc_fetch_* $O/pmc_write_*
ls -la $O
