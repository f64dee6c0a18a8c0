#!/bin/bash
# what the shards of the north star's strong-scaling target cost on ONE GPU: 65 536 chains / {1, 2, 4, 8} GPUs
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/shards
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in 65536 32768 16384 8192 4096; do
  for r in 0 16; do
    python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate --chains $c --steps 20 --warmup 5 --reserved-cus $r > $OUT/c${c}_r$r.json 2> $OUT/c${c}_r$r.err
    python3 -c "import json; d=json.load(open('$OUT/c${c}_r$r.json')); print($c, 'reserved', $r, round(d['ms_per_step'],4), 'ms  %.3e evals/s  kernel %.4f ms' % (d['value'], d['roofline']['avg_launch_ms']), d['config']['geometry']['workgroups'])" 2>&1 | tail -1
  done
done
python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate --config 5 --steps 10 --warmup 3 > $OUT/cfg5_1gpu.json 2> $OUT/cfg5.err
python3 -c "import json; d=json.load(open('$OUT/cfg5_1gpu.json')); print('config 5 on one GPU (262144 chains)', round(d['ms_per_step'],4), '%.3e' % d['value'])"
