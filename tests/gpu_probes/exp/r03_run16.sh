cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --steps 10 --warmup 3"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); g=d.get('parity_gate',{}); print('$1', round(d['ms_per_step'],4), '%.4e' % d['value'], d['roofline']['bound'], round(d['roofline']['frac'],3), g.get('within_1e-10'), g.get('tree_mismatches'), g.get('max_rel_diff_logp'))"; }
$B --model funnel --dim 16384 --chains 8192 --adapt-iters 60 --gate-chains 16 --gate-transitions 4 2>/dev/null | show funnel16384
$B --model rw1 --dim 16384 --chains 8192 --adapt-iters 60 --gate-chains 16 --gate-transitions 4 2>/dev/null | show rw1_16384
$B --model diag_normal --dim 16384 --chains 8192 --adapt-iters 60 --no-parity-gate 2>/dev/null | show cfg4
