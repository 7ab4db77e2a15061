#!/bin/bash
# GPU box: memory-system and issue counters of the rollout kernel of `bench.py --n $1` (default 1024), one rocprofv3 --pmc pass per
# counter group (counters only: no trace domains).  Prints per-launch averages for the kernel whose name contains $2 (default rollout_duo).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
N=${1:-1024}; K=${2:-rollout_duo}; shift; shift
O=gpurun_out/pmc_$N
rm -rf $O; mkdir -p $O
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do     # (the TA_* group aborts rocprofv3 on this image: left out)
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $O/p$i -o p --output-format csv -- python3 bench.py --n $N --steps 4 --warmup 1 --no-cpu-baseline "$@" > $O/p$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -2 $O/p$i.log | tr '\n' ' ')"
done
python3 - "$O" "$K" <<'PY'
import csv, glob, sys, collections
O, K = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if K in row.get("Kernel_Name", ""):
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
for k in sorted(tot):
    print("%-40s per launch %16.1f   (%d dispatch records)" % (k, tot[k] / max(cnt[k], 1), cnt[k]))
PY
