#!/bin/bash
# A/B of whole trees on ONE lease, interleaved: bash tools/r6_ab.sh OUT tree1 tree2 ...   (each tree: a checkout with its own built
# neuraloc_amd/csrc/libnocf.so and bench.py; "." = HEAD).  Headline bench only (no CPU leg, no other workloads), REPS rounds.
# Round 6: ab/r3 (c01c1a4), ab/r4 (2704cb1) and HEAD -> profiles/r6/01_ab_r3_vs_head.txt
out=$(realpath -m "$1"); shift
export NOCF_JIT=0
root=$PWD
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in $(seq 1 ${REPS:-3}); do
  for t in "$@"; do
    cd "$root/$t"
    line=$(python bench.py --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | grep '^{' | tail -1)
    python - "$rep" "$t" "$line" >> "$out" <<'PY'
import json, sys
rep, t, line = sys.argv[1:4]
try:
    j = json.loads(line)
    r = j.get("roofline", {})
    print(f"rep {rep}  {t:8s}  ms_per_step {j['ms_per_step']:.4f}  kernel_ms {r.get('kernel_ms', float('nan')):.4f}  frac {r.get('frac', float('nan')):.4f}")
except Exception as ex:
    print(f"rep {rep}  {t:8s}  FAILED {ex!r} {line[:200]}")
PY
    cd "$root"
  done
done
cat "$out"
