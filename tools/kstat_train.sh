export NOCF_JIT=0 TMPDIR=/tmp
rm -rf gpurun_out/ks; timeout 240 rocprofv3 --kernel-trace --stats -d gpurun_out/ks -o tr --output-format csv -- python3 tools/time_train.py ${1:-swarm50} 8 > gpurun_out/ks.log 2>&1
find gpurun_out/ks -name "*kernel_stats.csv" -exec head -4 {} \; | cut -c1-140
