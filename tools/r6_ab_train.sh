#!/bin/bash
# A/B of library builds on one lease, TRAINING: bash tools/r6_ab_train.sh OUT libA.so libB.so ...  (recording forward / adjoint kernel times of a swarm50
# iteration from the library's HIP events, then the wall time of an Adam iteration)
out=$(realpath -m "$1"); shift
export NOCF_JIT=0
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in $(seq 1 ${REPS:-2}); do
  for l in "$@"; do
    a=$(NOCF_LIB_PATH=$PWD/neuraloc_amd/csrc/$l python tools/time_rec.py 2>&1 | tail -1)
    b=$(NOCF_LIB_PATH=$PWD/neuraloc_amd/csrc/$l python tools/time_train.py swarm50 20 2>&1 | grep "^{" | cut -c1-160)
    echo "rep $rep  $l  | $a | $b" >> "$out"
  done
done
cat "$out"
