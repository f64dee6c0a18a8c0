#!/bin/bash
# A/B of library variants on the small-dimension workloads (several wavefronts per SIMD): small_ab.sh <reps> <variant> ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
R=$1; shift
run() {
  local tag=$1; shift
  python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V', '$tag', round(d['ms_per_step'],4), '%.4e' % d['value'])"
}
for r in $(seq $R); do
  for V in "$@"; do
    if [ "$V" = prod ]; then unset WALNUTS_AMD_LIB; else export WALNUTS_AMD_LIB=$ROOT/tests/gpu_probes/libwalnuts_$V.so; fi
    run cfg3 --model funnel --chains 16384 --dim 128 --adapt-iters 300
    run cfg3_warm --model funnel --chains 16384 --dim 128 --adapt-iters 300 --phase warmup
    run std128 --chains 65536 --dim 128
    for D in $SMALL_AB_DIMS; do run std$D --chains 65536 --dim $D; run funnel$D --model funnel --chains 16384 --dim $D --adapt-iters 150; done
  done
done
