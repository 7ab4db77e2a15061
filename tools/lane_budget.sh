#!/bin/bash
# GPU box: issue counters of the one-wavefront-per-sample kernels (swap2, softcorridor, swap12 at their BASELINE sizes) -> instructions and
# cycles per wave and RK evaluation (counters only, one rocprofv3 --pmc pass per group).  Output: gpurun_out/lane_budget/summary.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp NOCF_JIT=0
O=gpurun_out/lane_budget
rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  for wl in swap2 softcorridor swap12; do
    timeout 300 rocprofv3 --pmc $grp -d $O/p${i}_$wl -o p --output-format csv -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/p${i}_$wl.log 2>&1 || echo "pass $i $wl failed"
  done
done
python3 - "$O" <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
O = sys.argv[1]
# (grid size identifies the workload: swap2 / softcorridor 1024 samples = 256 workgroups of 4 waves, swap12 2048 = 512)
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    wl = f.split("/p")[1].split("/")[0].split("_", 1)[1]
    for row in csv.DictReader(open(f)):
        if "rollout_lane_kernel" in row.get("Kernel_Name", ""):
            key = (wl, row["Kernel_Name"].split("(")[0])
            tot[key][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[key][row["Counter_Name"]] += 1
for key in sorted(tot):
    print(key)
    c = {k: tot[key][k] / max(cnt[key][k], 1) for k in tot[key]}
    for k in sorted(c):
        print("    %-24s per launch %16.1f" % (k, c[k]))
    if c.get("SQ_WAVES"):
        w = c["SQ_WAVES"]
        print("    per wave: VALU %.0f  SALU %.0f  SMEM %.0f  wave-cycles %.0f  waiting %.0f" % (c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w,
              c.get("SQ_INSTS_SMEM", 0) / w, c.get("SQ_WAVE_CYCLES", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w))
PY
