# A/B of library builds on one box: bash tools/r5_ab.sh libA.so libB.so ...   (recording forward / adjoint kernel times, one training iteration)
export NOCF_JIT=0
for l in "$@"; do
  echo "== $l"
  NOCF_LIB_PATH=$PWD/neuraloc_amd/csrc/$l python tools/time_rec.py 2>&1 | tail -1
  NOCF_LIB_PATH=$PWD/neuraloc_amd/csrc/$l python tools/time_train.py swarm50 20 2>&1 | grep "^{" | cut -c1-130
done
