cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
B="python bench.py --no-cpu-baseline --no-parity-gate"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), '%.4e' % d['value'])"; }
for r in 1 2; do
  $B --fma 0 2>/dev/null | show fma0
  $B --fma 1 2>/dev/null | show fma1
done
$B --fma 0 --phase warmup 2>/dev/null | show warm_fma0
$B --fma 1 --phase warmup 2>/dev/null | show warm_fma1
python bench.py --no-cpu-baseline --gate-chains 256 --gate-transitions 16 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); g=d['parity_gate']; print(d['ms_per_step'], g['max_rel_diff'], g['max_rel_diff_logp'], g['tree_mismatches'], g['near_ties_1e-12'], g['within_1e-10'])"
