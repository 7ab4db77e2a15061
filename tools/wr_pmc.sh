#!/bin/bash
# GPU box: how many of the split-role kernel's stores leave L2 on its memory side, by batch size (n = 128: one group of one tile per XCD,
# 0.4 MB of exchange area per 4 MB L2; n = 1024: four groups of two tiles, 3 MB)?  Counters only, one pass per group.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp NOCF_JIT=0
O=gpurun_out/wr_pmc
rm -rf $O; mkdir -p $O
for N in "$@"; do
  i=0
  for grp in "TCC_EA0_WRREQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_WRITEBACK_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp -d $O/n$N/p$i -o p --output-format csv -- python3 bench.py --n $N --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/n${N}_p$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -2 $O/n${N}_p$i.log | tr '\n' ' ')"
  done
done
python3 - "$O" "$@" <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
O = sys.argv[1]
for N in sys.argv[2:]:
    tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
    for f in glob.glob(O + "/n%s/p*/**/*counter_collection.csv" % N, recursive=True):
        for row in csv.DictReader(open(f)):
            if "rollout_duo" in row.get("Kernel_Name", ""):
                tot[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
    print("n = %s, per launch of rollout_duo_kernel: " % N + "  ".join("%s=%.4g" % (k, tot[k] / max(cnt[k], 1)) for k in sorted(tot)))
PY
rm -rf $O/n*/
