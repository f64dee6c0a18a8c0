#!/bin/bash
# general-gradient path at the headline dimension: funnel and rw1, 16 384 chains x 1 024
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/general_grad
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for m in funnel rw1; do
  for g in "0 0" "1 16" "2 8" "4 4"; do
    set -- $g
    python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate --model $m --chains 16384 --dim 1024 --adapt-iters 150 --steps 10 --warmup 3 --waves-per-chain $1 --elems-per-lane $2 > $OUT/${m}_$1x$2.json 2> $OUT/${m}_$1x$2.err
    python3 -c "import json; d=json.load(open('$OUT/${m}_$1x$2.json')); print('$m', '$g', round(d['ms_per_step'],3), '%.3e' % d['value'], d['config']['grad_evals_per_transition_per_chain'], d['config']['geometry'])" 2>&1 | tail -1
  done
done
