#!/bin/bash
# the bench's same-run parity gate (device vs the oracle in the reference's arithmetic -- libm, every product rounded --
# under both reference-side summation orders: Eigen 3.4's SSE2 redux (restated) and sequential loops; identical trees
# required, near-tie audit) widened: 512 chains x 24 transitions for every bench configuration.  WIDE_GATE_ARGS adds
# bench arguments to every line (e.g. "--fma 0" for the device build with every product rounded).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
gate() {
  local tag=$1; shift
  python3 $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --gate-chains 512 --gate-transitions 24 $WIDE_GATE_ARGS "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['parity_gate']
for name, o in g['orders'].items():
    print('$tag', '%-10s' % name, 'chains', g['chains'], 'transitions', g['transitions'], 'tree_mismatches', o['tree_mismatches'], 'max_rel_diff_positions %.3e' % o['max_rel_diff'], 'max_rel_diff_logp %.3e' % o['max_rel_diff_logp'], 'near_ties', {k: v['near'] for k, v in o['near_ties_1e-12'].items()}, 'decisions', {k: v['decisions'] for k, v in o['near_ties_1e-12'].items()})"
}
gate headline
gate cfg2 --model ill_normal --chains 4096 --dim 1024 --adapt-iters 300
gate cfg3 --model funnel --chains 16384 --dim 128 --adapt-iters 300
gate funnel1024 --model funnel --chains 16384 --dim 1024 --adapt-iters 150
gate rw1 --model rw1 --chains 16384 --dim 1024 --adapt-iters 150
gate d256 --chains 65536 --dim 256
# the streaming kernels that hold the trajectory's moving end in registers (fewer chains: the oracle pays per dimension)
gate cfg4 --model diag_normal --chains 8192 --dim 16384 --adapt-iters 100 --gate-chains 96 --gate-transitions 16
gate diag6000 --model diag_normal --chains 8192 --dim 6000 --adapt-iters 100 --gate-chains 128 --gate-transitions 16
gate funnel16384 --model funnel --chains 8192 --dim 16384 --adapt-iters 60 --gate-chains 32 --gate-transitions 8
gate rw1_16384 --model rw1 --chains 8192 --dim 16384 --adapt-iters 60 --gate-chains 32 --gate-transitions 8
