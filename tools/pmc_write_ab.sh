cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for fast in 1 0; do
  NOCF_SLAB_FAST=$fast timeout 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmcw_$fast -o w --output-format csv -- python3 bench.py --n 1024 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcw_$fast.log 2>&1
  python - <<PY
import csv,glob
vals=[]
for f in glob.glob("gpurun_out/pmcw_$fast/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rollout_slab_kernel" in row.get("Kernel_Name","") and row.get("Counter_Name")=="WRITE_SIZE": vals.append(float(row["Counter_Value"]))
print("fast=$fast WRITE_SIZE KiB per launch:", vals)
PY
done
rm -rf gpurun_out/pmcw_*
