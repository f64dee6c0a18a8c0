cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -4
B="python bench.py --no-cpu-baseline --no-parity-gate"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), '%.4e' % d['value'])"; }
for r in 1 2; do
  $B 2>/dev/null | show headline
  WALNUTS_AMD_PREGEN=0 $B 2>/dev/null | show headline_inline
done
$B --phase warmup 2>/dev/null | show warmup
$B --model ill_normal --chains 4096 --adapt-iters 300 --steps 40 2>/dev/null | show cfg2
$B --chains 8192 --steps 40 2>/dev/null | show shard8192
