# kernel statistics of the training iteration: bash tools/kstat_train.sh [workload] [rows of the table]
export NOCF_JIT=0 TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/ks; timeout 240 rocprofv3 --kernel-trace --stats -d gpurun_out/ks -o tr --output-format csv -- python3 tools/time_train.py ${1:-swarm50} 8 > gpurun_out/ks.log 2>&1
find gpurun_out/ks -name "*kernel_stats.csv" -exec head -${2:-4} {} \; | python3 -c "
import csv, sys
for r in csv.reader(sys.stdin):
    print(r[0][:70].ljust(70), *r[1:4])
"
