#!/bin/bash
# GPU box: L2 hit / miss and fabric read counters of the double-precision rollout (swarm50), counters only.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/pmc_f64
rm -rf $O; mkdir -p $O
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $O/p$i -o p --output-format csv -- python3 tools/f64_time.py swarm50 > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rollout_f64" in row.get("Kernel_Name", ""):
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
for k in sorted(tot):
    print("%-36s per launch %18.1f   (%d records)" % (k, tot[k] / max(cnt[k], 1), cnt[k]))
PY
