#!/bin/bash
# transitions per launch (wn_engine_sample_steps): fused_ab.sh <variant|prod> <T> [<T> ...] on the headline, its shards, config #2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
V=$1; shift
if [ "$V" = prod ]; then unset WALNUTS_AMD_LIB; else export WALNUTS_AMD_LIB=$ROOT/tests/gpu_probes/libwalnuts_$V.so; fi
run() {
  local tag=$1; shift
  python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), '%.4e' % d['value'], 'kernel', round(d['roofline']['avg_launch_ms'],4), 'launches', d['config']['launches'])"
}
for T in "$@"; do
  S=$((T*6)); [ $S -lt 24 ] && S=24
  run "headline T=$T" --transitions-per-launch $T --steps $S --warmup $T
  run "warmup   T=$T" --transitions-per-launch $T --steps $S --warmup $T --phase warmup
  run "shard32k T=$T" --transitions-per-launch $T --steps $S --warmup $T --chains 32768
  run "shard8k  T=$T" --transitions-per-launch $T --steps $S --warmup $T --chains 8192
  run "cfg2     T=$T" --transitions-per-launch $T --steps $S --warmup $T --model ill_normal --chains 4096 --dim 1024 --adapt-iters 300
  run "cfg3     T=$T" --transitions-per-launch $T --steps $S --warmup $T --model funnel --chains 16384 --dim 128 --adapt-iters 300
done
