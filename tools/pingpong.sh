#!/bin/bash
# GPU box: the exchange microbenchmark (tools/micro/xcd_pingpong.hip): hop latencies, then which stores reach L2's memory side
# (separate counters-only rocprofv3 passes; every variant is its own kernel instantiation).  Output under gpurun_out/pp/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/pp
rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/pp tools/micro/xcd_pingpong.hip 2> $O/build.log || { tail -5 $O/build.log; exit 1; }
timeout 300 /tmp/pp 20000 > $O/latency.txt 2>&1
cat $O/latency.txt
i=0
for grp in "TCC_EA0_WRREQ_sum TCP_TCC_WRITE_REQ_sum" "WRITE_SIZE" "TCC_EA0_WRREQ_64B_sum TCC_WRITEBACK_sum" "TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $O/p$i -o pp --output-format csv -- /tmp/pp 2000 > $O/p$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -2 $O/p$i.log | tr '\n' ' ')"
done
python3 - "$O" <<'PY' | tee $O/pmc_per_variant.txt
import csv, glob, sys, collections
O = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
print("# per kernel instantiation, summed over its launches (4 per pp<> variant: same-XCD / cross-XCD x nap 0 / 1; 2000 rounds each: 2 x 2008 payload stores of 64 lanes x 16 B per launch, twice that with reset)")
for k in sorted(tot):
    print(k[:60].ljust(60), "  ".join("%s=%.0f" % (c, tot[k][c]) for c in sorted(tot[k])))
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
