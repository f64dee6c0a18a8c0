#!/bin/bash
# every bench configuration (except the streaming config #4) for each library variant named on the command line
# usage: allcfg.sh <variant> [<variant> ...]   (variants are tests/gpu_probes/libwalnuts_<variant>.so; "prod" = the product)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
run() {  # tag, args...
  local tag=$1; shift
  python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V', '$tag', round(d['ms_per_step'],4), '%.4e' % d['value'], d['config'].get('geometry'))"
}
for V in "$@"; do
  if [ "$V" = prod ]; then unset WALNUTS_AMD_LIB; else export WALNUTS_AMD_LIB=$ROOT/tests/gpu_probes/libwalnuts_$V.so; fi
  run headline
  run warmup --phase warmup
  run cfg2 --model ill_normal --chains 4096 --dim 1024 --adapt-iters 300
  run cfg3 --model funnel --chains 16384 --dim 128 --adapt-iters 300
  run funnel1024 --model funnel --chains 16384 --dim 1024 --adapt-iters 150
  run rw1 --model rw1 --chains 16384 --dim 1024 --adapt-iters 150
  run d256 --chains 65536 --dim 256
  run d64 --chains 262144 --dim 64
done
