#!/bin/bash
# differential timing of the adjoint kernel: libraries built with -DBWD_CUT=k stop phi_vjp after phase k
# (5: before ybar, 1: after ybar, 2: after vbar, 3: after ubar, 4: after the closing phase / before the row streams)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for L in libnocf_cut5.so libnocf_cut1.so libnocf_cut2.so libnocf_cut3.so libnocf_cut4.so libnocf.so; do
  rm -rf gpurun_out/cutprof
  NOCF_LIB_PATH=$PWD/neuraloc_amd/csrc/$L timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/cutprof -o c --output-format csv -- python3 tools/time_train.py swarm50 3 > gpurun_out/cut.log 2>&1
  f=$(find gpurun_out/cutprof -name "*kernel_stats.csv" | head -1)
  python3 - "$L" "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[2])):
    if "rollout_bwd_kernel" in row["Name"]:
        print(sys.argv[1], "adjoint kernel avg ms %.3f (calls %s)" % (float(row["AverageNs"]) / 1e6, row["Calls"]))
PY
done
rm -rf gpurun_out/cutprof
