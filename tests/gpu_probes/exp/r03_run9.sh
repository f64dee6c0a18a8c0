cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-parity-gate --steps 40 --warmup 5"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), '%.4e' % d['value'], d['config']['geometry']['workgroups'])"; }
for G in "1 16" "2 8" "4 4"; do set -- $G
 for W in 0 1 2; do
  $B --model ill_normal --chains 4096 --adapt-iters 300 --waves-per-chain $1 --elems-per-lane $2 --workgroups-per-cu $W 2>/dev/null | show "cfg2 ${1}x${2} wg/cu=$W"
 done
done
for G in "1 16" "2 8"; do set -- $G
 for W in 0 2; do
  $B --chains 8192 --waves-per-chain $1 --elems-per-lane $2 --workgroups-per-cu $W 2>/dev/null | show "8192 ${1}x${2} wg/cu=$W"
  $B --chains 16384 --waves-per-chain $1 --elems-per-lane $2 --workgroups-per-cu $W 2>/dev/null | show "16384 ${1}x${2} wg/cu=$W"
 done
done
