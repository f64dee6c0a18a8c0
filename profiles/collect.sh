#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): bench lines, rocprofv3 kernel trace and the two PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots") for the
# headline workload and BASELINE config #4.  Outputs land in gpurun_out/prof_<tag>/; summarise locally with
#   python profiles/summarize.py <title> <trace_db> <fetch_db> <write_db>
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HEAD="--steps 10 --warmup 3 --adapt-iters 100"
CFG4="--model diag_normal --chains 8192 --dim 16384 --steps 5 --warmup 2 --adapt-iters 100"
python3 $ROOT/bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err
python3 $ROOT/bench.py --phase warmup --no-cpu-baseline > $OUT/bench_headline_warmup.json 2>> $OUT/bench_headline.err
python3 $ROOT/bench.py --no-cpu-baseline --model ill_normal --chains 4096 --dim 1024 --adapt-iters 300 > $OUT/bench_cfg2.json 2>> $OUT/bench_headline.err
python3 $ROOT/bench.py --no-cpu-baseline --model funnel --chains 16384 --dim 128 --adapt-iters 300 > $OUT/bench_cfg3.json 2>> $OUT/bench_headline.err
python3 $ROOT/bench.py --no-cpu-baseline $CFG4 > $OUT/bench_cfg4.json 2>> $OUT/bench_headline.err
rocprofv3 --kernel-trace --stats -d $OUT/headline_trace -o t -- python3 $ROOT/bench.py --no-cpu-baseline $HEAD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/headline_fetch -o f -- python3 $ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 1 --adapt-iters 100 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/headline_write -o w -- python3 $ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 1 --adapt-iters 100 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/cfg4_trace -o t -- python3 $ROOT/bench.py --no-cpu-baseline $CFG4 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/cfg4_fetch -o f -- python3 $ROOT/bench.py --no-cpu-baseline --model diag_normal --chains 8192 --dim 16384 --steps 3 --warmup 1 --adapt-iters 100 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/cfg4_write -o w -- python3 $ROOT/bench.py --no-cpu-baseline --model diag_normal --chains 8192 --dim 16384 --steps 3 --warmup 1 --adapt-iters 100 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/summary_trace -o t -- python3 $ROOT/tests/gpu_probes/summary_bench.py --draws 32 --reps 1 > $OUT/summary_bench_profiled.json 2>/dev/null
python3 $ROOT/tests/gpu_probes/summary_bench.py --draws 32 > $OUT/summary_bench_32.json 2>/dev/null
python3 $ROOT/tests/gpu_probes/summary_bench.py --draws 200 --chains 32768 --phi 0.5 > $OUT/summary_bench_200.json 2>/dev/null
find $OUT -name "*.db" | xargs ls -la
