#!/bin/bash
# GPU box: the round-4 evidence under gpurun_out/r4 (copied to profiles/r4 afterwards).  Every step under its own timeout.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp NOCF_JIT=0
O=gpurun_out/r4
mkdir -p $O
T="timeout 600"
# 1. kernel-trace stats of the default bench command (swarm50 n = 1024: split-role kernel) and of one training iteration
$T rocprofv3 --kernel-trace --stats -d $O/prof_n1024 -o n1024 --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads > $O/prof_n1024.log 2>&1
find $O/prof_n1024 -name "*kernel_stats.csv" -exec cp {} $O/02_n1024_duo_kernel_stats.csv \;
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_train -o tr --output-format csv -- python3 tools/time_train.py swarm50 5 > $O/prof_train.log 2>&1
find $O/prof_train -name "*kernel_stats.csv" -exec cp {} $O/08_train_swarm50_kernel_stats.csv \;
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_train_sq -o tr --output-format csv -- python3 tools/time_train.py singlequad 5 > $O/prof_train_sq.log 2>&1
find $O/prof_train_sq -name "*kernel_stats.csv" -exec cp {} $O/08_train_singlequad_kernel_stats.csv \;
# 2. PMC passes (separate runs, counters only): HBM traffic of the n = 1024 launch
$T rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_1024 -o f --output-format csv -- python3 bench.py --n 1024 --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_1024.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_1024 -o w --output-format csv -- python3 bench.py --n 1024 --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_write_1024.log 2>&1
python tools/parse_pmc.py $O/pmc_fetch_1024 $O/pmc_write_1024 swarm50 $O/03_hbm_traffic_n1024_duo.json "rollout_duo_kernel" "rollout_duo_kernel" 1024 "profiles/r4/03_hbm_traffic_n1024_duo.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes" > $O/03_parse_1024.log 2>&1
cp $O/03_hbm_traffic_n1024_duo.json profiles/hbm_traffic_swarm50.json 2>/dev/null
# 3. the default bench line (with the CPU legs, the other workloads and the training iterations) and the strong-scaling proxy table
$T python bench.py > $O/04_bench_default.json 2> $O/04_bench_default.err
bash tools/proxy_table.sh $O > /dev/null 2>&1
mv $O/proxy_table.txt $O/05_proxy_table.txt; mv $O/proxy_table.jsonl $O/05_proxy_table.jsonl
# 4. timelines (diagnostic build), training times, double precision, memory-side writes by batch size, tests
for n in 512 1024; do NOCF_LIB_PATH=neuraloc_amd/csrc/libnocf_stamps.so timeout 300 python tools/duo_timeline.py $n 2>&1 | grep -v amdgpu.ids; done > $O/06_duo_timeline.txt
for n in 512 1024; do NOCF_LIB_PATH=neuraloc_amd/csrc/libnocf_stamps.so timeout 300 python tools/duo_bwd_timeline.py $n 2>&1 | grep -v "amdgpu.ids\|Warn\|warn"; done > $O/06_duo_bwd_timeline.txt
NOCF_LIB_PATH=neuraloc_amd/csrc/libnocf_stamps.so timeout 300 python tools/timeline_bwd.py singlequad 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" > $O/12_mono_bwd_timeline.txt
timeout 300 python tools/time_train.py 2>&1 | grep "^{" > $O/08_train_times.txt
timeout 300 python tools/time_train.py singlequad 2>&1 | grep "^{" >> $O/08_train_times.txt
NOCF_DUO_BWD=0 timeout 300 python tools/time_train.py 2>&1 | grep "^{" | sed 's/^/NOCF_DUO_BWD=0 (per-tile adjoint + record): /' >> $O/08_train_times.txt
NOCF_DUO_DW=1 timeout 300 python tools/time_train.py 2>&1 | grep "^{" | sed 's/^/NOCF_DUO_DW=1 (weight-gradient roles): /' >> $O/08_train_times.txt
NOCF_MONO_BWD=0 timeout 300 python tools/time_train.py singlequad 2>&1 | grep "^{" | sed 's/^/NOCF_MONO_BWD=0 (per-tile adjoint + record + contractions): /' >> $O/08_train_times.txt
NOCF_ACT_REC=0 timeout 300 python tools/time_train.py singlequad 2>&1 | grep "^{" | sed 's/^/NOCF_ACT_REC=0 (one-CU adjoint re-running grad Phi): /' >> $O/08_train_times.txt
timeout 600 python tools/f64_time.py 2>&1 | grep -v amdgpu.ids > $O/10_f64_times.txt
bash tools/wr_pmc.sh 128 256 512 1024 > /dev/null 2>&1; cp gpurun_out/wr_pmc/summary.txt $O/07_memory_side_writes_by_batch.txt
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 > $O/09_pytest_gpu.log
rm -rf $O/prof_n1024 $O/prof_train $O/prof_train_sq $O/pmc_fetch_* $O/pmc_write_*
ls -la $O
