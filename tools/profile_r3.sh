#!/bin/bash
# GPU box: the round-3 evidence under gpurun_out/r3 (copied to profiles/r3 afterwards).  Every step under its own timeout.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O
T="timeout 600"
# 1. kernel-trace stats of the default bench command (swarm50 n = 1024: split-role kernel) and of the 512-row proxy
$T rocprofv3 --kernel-trace --stats -d $O/prof_n1024 -o n1024 --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/prof_n1024.log 2>&1
$T rocprofv3 --kernel-trace --stats -d $O/prof_n512 -o n512 --output-format csv -- python3 bench.py --n 512 --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_n512.log 2>&1
find $O/prof_n1024 -name "*kernel_stats.csv" -exec cp {} $O/02_n1024_duo_kernel_stats.csv \;
find $O/prof_n512 -name "*kernel_stats.csv" -exec cp {} $O/02_n512_duo_kernel_stats.csv \;
# 2. PMC passes (separate runs, counters only): HBM traffic of the n = 1024 launch
$T rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_1024 -o f --output-format csv -- python3 bench.py --n 1024 --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_1024.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_1024 -o w --output-format csv -- python3 bench.py --n 1024 --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_write_1024.log 2>&1
python tools/parse_pmc.py $O/pmc_fetch_1024 $O/pmc_write_1024 swarm50 $O/03_hbm_traffic_n1024_duo.json "rollout_duo_kernel" "rollout_duo_kernel" 1024 "profiles/r3/03_hbm_traffic_n1024_duo.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes" > $O/03_parse_1024.log 2>&1
cp $O/03_hbm_traffic_n1024_duo.json profiles/hbm_traffic_swarm50.json 2>/dev/null
# 3. the default bench line (with the CPU legs and the other workloads) and the strong-scaling proxy table
$T python bench.py > $O/04_bench_default.json 2> $O/04_bench_default.err
: > $O/05_proxy_table.jsonl
for n in 4096 2048 1024 512 256 128; do timeout 300 python bench.py --n $n --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/05_proxy_table.jsonl; done
NOCF_DUO=0 timeout 300 python bench.py --n 1024 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/05_proxy_table.jsonl
NOCF_DUO=0 timeout 300 python bench.py --n 128 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/05_proxy_table.jsonl
python - <<'PY' > gpurun_out/r3/05_proxy_table.txt
import json
print("swarm50 nt=80 on ONE MI355X by batch rows (bench.py --n ROWS): 512 / 256 / 128 = the per-rank batch of n=1024 at 2 / 4 / 8 GPUs; last two lines: the per-tile kernel (NOCF_DUO=0)")
for line in open("gpurun_out/r3/05_proxy_table.jsonl"):
    try:
        j = json.loads(line)
        print("rows/GPU=%4d  kernel=%-34s kernel_ms=%.3f  ms_per_step=%.3f  traj/s=%8.0f  roofline.frac=%.3f" % (j["config"]["rows_per_gpu"], j["roofline"]["kernel"], j["roofline"]["kernel_ms"], j["ms_per_step"], j["value"], j["roofline"]["frac"]))
    except Exception as e:
        print("ERR", line[:200])
PY
cat $O/05_proxy_table.txt
# 4. timelines, counters, training, double precision, tests
for n in 512 1024; do NOCF_LIB_PATH=neuraloc_amd/csrc/libnocf_stamps.so timeout 300 python tools/duo_timeline.py $n 2>&1 | grep -v amdgpu.ids; done > $O/06_duo_timeline.txt
timeout 300 python tools/time_train.py 2>&1 | grep -v amdgpu.ids > $O/08_train_times.txt
timeout 300 python tools/time_train.py singlequad 2>&1 | grep -v amdgpu.ids >> $O/08_train_times.txt
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_train -o tr --output-format csv -- python3 tools/time_train.py swarm50 5 > $O/prof_train.log 2>&1
find $O/prof_train -name "*kernel_stats.csv" -exec cp {} $O/08_train_swarm50_kernel_stats.csv \;
timeout 600 python tools/f64_time.py 2>&1 | grep -v amdgpu.ids > $O/10_f64_times.txt
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6 > $O/09_pytest_gpu.log
rm -rf $O/prof_n1024 $O/prof_n512 $O/prof_train $O/pmc_fetch_* $O/pmc_write_*
ls -la $O
