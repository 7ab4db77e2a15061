# A/B of library builds, evaluation forward by batch rows: bash tools/r5_ab_fwd.sh libA.so libB.so ...
export NOCF_JIT=0
for l in "$@"; do
  for n in 1024 512 128; do
    echo -n "$l n=$n "
    NOCF_LIB_PATH=$PWD/neuraloc_amd/csrc/$l python bench.py --n $n --steps 60 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 | python3 -c "import sys, json; j = json.loads(sys.stdin.read()); print('kernel_ms %.3f  Jc %s' % (j['roofline']['kernel_ms'], j['config'].get('Jc')))"
  done
done
