#!/bin/bash
# A/B of library variants on the headline workload, interleaved repeats: headline_ab.sh <reps> <variant> [<variant> ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
R=$1; shift
for r in $(seq $R); do
  for V in "$@"; do
    if [ "$V" = prod ]; then unset WALNUTS_AMD_LIB; else export WALNUTS_AMD_LIB=$ROOT/tests/gpu_probes/libwalnuts_$V.so; fi
    python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate $AB_ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V', round(d['ms_per_step'],4), '%.4e' % d['value'])"
  done
done
