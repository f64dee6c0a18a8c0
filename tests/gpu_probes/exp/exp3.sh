#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exp3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=$ROOT/tests/gpu_probes
B="python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 --adapt-iters 100"
run() { name=$1; lib=$2; shift 2; WALNUTS_AMD_LIB=$lib $B "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name: $(python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print(d['ms_per_step'], d['value'], d['config']['grad_evals_per_transition_per_chain'], d['config']['geometry'])" 2>&1 | tail -1)"; }
for wg in 4 3 2; do
  run c00_wg$wg $P/libwalnuts_c00.so --workgroups-per-cu $wg
done
S="--no-cpu-baseline --steps 4 --warmup 1 --adapt-iters 100"
export WALNUTS_AMD_LIB=$P/libwalnuts_c00.so
for wg in 4 3 2; do
rocprofv3 --pmc FETCH_SIZE -d $OUT/wg${wg}_fetch -o f -- python3 $ROOT/bench.py $S --workgroups-per-cu $wg > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/wg${wg}_write -o w -- python3 $ROOT/bench.py $S --workgroups-per-cu $wg > /dev/null 2>&1
done
python3 - <<'PY'
import sqlite3,glob,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/exp3'
for wg in (4,3,2):
    v={}
    for kind,f in (('fetch','f'),('write','w')):
        db=glob.glob(f'{out}/wg{wg}_{kind}/**/*results.db',recursive=True)
        cur=sqlite3.connect(db[0]).cursor()
        rows=cur.execute("select value from counters_collection where kernel_name like '%transition_kernel%' order by start").fetchall()
        v[kind]=rows[-1][0]
    print('wg',wg,'fetch KB',v['fetch'],'write KB',v['write'],'traffic GB',(2*v['fetch']+v['write'])*1024/1e9)
PY
