#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exp2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=$ROOT/tests/gpu_probes
make -C $ROOT/oracle -s > /dev/null 2>&1


B="python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 --adapt-iters 100"
run() { name=$1; lib=$2; shift 2; WALNUTS_AMD_LIB=$lib $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name: $(python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print(d['ms_per_step'], d['value'], d['config']['grad_evals_per_transition_per_chain'], d['config']['geometry'])" 2>&1 | tail -1)"; }
for v in c20 c22 c40 c42; do
  run $v $P/libwalnuts_$v.so
done
run c42_116 $P/libwalnuts_c42.so --waves-per-chain 1 --elems-per-lane 16
run c42_44 $P/libwalnuts_c42.so --waves-per-chain 4 --elems-per-lane 4
S="--no-cpu-baseline --steps 4 --warmup 1 --adapt-iters 100"
for v in c40 c42; do
export WALNUTS_AMD_LIB=$P/libwalnuts_$v.so
rocprofv3 --pmc FETCH_SIZE -d $OUT/${v}_fetch -o f -- python3 $ROOT/bench.py $S > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/${v}_write -o w -- python3 $ROOT/bench.py $S > /dev/null 2>&1
done
