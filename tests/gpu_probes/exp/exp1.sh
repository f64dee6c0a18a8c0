#!/bin/bash
# bounding experiment: what would the register kernels run at if the span pool never left the chip?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exp1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=$ROOT/tests/gpu_probes
B="python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 --adapt-iters 100"
run() { name=$1; lib=$2; shift 2; WALNUTS_AMD_LIB=$lib $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name: $(python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print(d['ms_per_step'], d['value'], d['config']['grad_evals_per_transition_per_chain'], d['config']['geometry'])" 2>&1 | tail -1)"; }
run base "" 
run wpe2_28_wg4 $P/libwalnuts_wpe2.so --waves-per-chain 2 --elems-per-lane 8 --workgroups-per-cu 4
run alias2_28_wg4 $P/libwalnuts_alias2.so --waves-per-chain 2 --elems-per-lane 8 --workgroups-per-cu 4
run alias3_28_wg6 $P/libwalnuts_alias3.so --waves-per-chain 2 --elems-per-lane 8 --workgroups-per-cu 6
run wpe2_116_wg4 $P/libwalnuts_wpe2.so --waves-per-chain 1 --elems-per-lane 16 --workgroups-per-cu 4
run alias2_116_wg4 $P/libwalnuts_alias2.so --waves-per-chain 1 --elems-per-lane 16 --workgroups-per-cu 4
run alias2_44_wg2 $P/libwalnuts_alias2.so --waves-per-chain 4 --elems-per-lane 4 --workgroups-per-cu 2
run alias2_44_wg4 $P/libwalnuts_alias2.so --waves-per-chain 4 --elems-per-lane 4 --workgroups-per-cu 4
S="--no-cpu-baseline --steps 4 --warmup 1 --adapt-iters 100"
export WALNUTS_AMD_LIB=$P/libwalnuts_alias2.so
rocprofv3 --pmc FETCH_SIZE -d $OUT/alias2_fetch -o f -- python3 $ROOT/bench.py $S --waves-per-chain 2 --elems-per-lane 8 --workgroups-per-cu 4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/alias2_write -o w -- python3 $ROOT/bench.py $S --waves-per-chain 2 --elems-per-lane 8 --workgroups-per-cu 4 > /dev/null 2>&1
export WALNUTS_AMD_LIB=$P/libwalnuts_wpe2.so
rocprofv3 --pmc FETCH_SIZE -d $OUT/wpe2_fetch -o f -- python3 $ROOT/bench.py $S --waves-per-chain 2 --elems-per-lane 8 --workgroups-per-cu 4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/wpe2_write -o w -- python3 $ROOT/bench.py $S --waves-per-chain 2 --elems-per-lane 8 --workgroups-per-cu 4 > /dev/null 2>&1
find $OUT -name "*.db" | xargs ls -la
