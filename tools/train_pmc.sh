#!/bin/bash
# GPU box: issue counters of the kernels of one training iteration (tools/time_train.py WORKLOAD), one rocprofv3 --pmc pass per group, counters
# only.  Per-launch averages for every kernel whose name contains "rollout".  usage: bash tools/train_pmc.sh [swarm50|singlequad]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp NOCF_JIT=0
W=${1:-swarm50}
O=gpurun_out/train_pmc_$W
rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $O/p$i -o p --output-format csv -- python3 tools/time_train.py $W 3 > $O/p$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -2 $O/p$i.log | tr '\n' ' ')"
done
python3 - "$O" <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
O = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row.get("Kernel_Name", "")
        if "rollout" in kn:
            key = kn.split("(")[0]
            tot[key][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[key][row["Counter_Name"]] += 1
for key in sorted(tot):
    print(key)
    for k in sorted(tot[key]):
        print("    %-32s per launch %16.1f   (%d records)" % (k, tot[key][k] / max(cnt[key][k], 1), cnt[key][k]))
PY
