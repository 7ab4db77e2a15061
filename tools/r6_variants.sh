#!/bin/bash
# A/B of library builds of HEAD's sources on one lease: bash tools/r6_variants.sh OUT "ROWS..." libA.so libB.so ...
# a library may carry environment knobs: libX.so:NOCF_DUO_UDELAY=3:NOCF_DUO_DBG=8
# (libraries in neuraloc_amd/csrc/, built from nocf_duo.hip with -D... experiment macros; evaluation forward, kernel time from HIP events)
out=$(realpath -m "$1"); rows=$2; shift 2
export NOCF_JIT=0
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in $(seq 1 ${REPS:-2}); do
  for l in "$@"; do
    for n in $rows; do
      lib=${l%%:*}; envs=""; [ "$lib" != "$l" ] && envs=$(echo "${l#*:}" | tr ':' ' ')
      line=$(env $envs NOCF_LIB_PATH=$PWD/neuraloc_amd/csrc/$lib python bench.py --n $n --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | grep '^{' | tail -1)
      python - "$rep" "$l" "$n" "$line" >> "$out" <<'PY'
import json, sys
rep, l, n, line = sys.argv[1:5]
try:
    j = json.loads(line)
    r = j["roofline"]
    print(f"rep {rep}  {l:22s} n={int(n):5d}  ms_per_step {j['ms_per_step']:.4f}  kernel_ms {r['kernel_ms']:.4f}  frac {r['frac']:.4f}  Jc {j['config'].get('Jc')}")
except Exception as ex:
    print(f"rep {rep}  {l:22s} n={n}  FAILED {ex!r} {line[:160]}")
PY
    done
  done
done
cat "$out"
