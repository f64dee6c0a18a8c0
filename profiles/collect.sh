#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round's bench lines of every BASELINE configuration + the rocprofv3
# kernel trace of the headline.  Outputs land in gpurun_out/$PROFILE_ROUND/ (default r06); PMC passes are profiles/pmc.sh's.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=${PROFILE_ROUND:-r06}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the line the driver runs, as the driver runs it: headline + the legs for configs #2-#4 and the warmup phase
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2> $OUT/bench.err
B="python3 $ROOT/bench.py --legs none"
$B > $OUT/bench_headline.json 2>> $OUT/bench.err
$B --phase warmup --no-cpu-baseline --no-parity-gate > $OUT/bench_headline_warmup.json 2>> $OUT/bench.err
$B --no-cpu-baseline --model ill_normal --chains 4096 --dim 1024 --adapt-iters 300 > $OUT/bench_cfg2.json 2>> $OUT/bench.err
$B --no-cpu-baseline --model funnel --chains 16384 --dim 128 --adapt-iters 300 > $OUT/bench_cfg3.json 2>> $OUT/bench.err
$B --no-cpu-baseline --model diag_normal --chains 8192 --dim 16384 --steps 8 --warmup 8 --adapt-iters 100 > $OUT/bench_cfg4.json 2>> $OUT/bench.err
# config #4 on the kernels of the earlier rounds: sixteen wavefronts per chain streaming BOTH ends of every micro step
# (round 4's kernel: what still runs beyond 16 384 dimensions), and the same without its two round-4 savings (round 3's)
$B --no-cpu-baseline --no-parity-gate --waves-per-chain 16 --elems-per-lane -1 --model diag_normal --chains 8192 --dim 16384 --steps 8 --warmup 8 --adapt-iters 100 > $OUT/bench_cfg4_both_ends_streamed.json 2>> $OUT/bench.err
WALNUTS_AMD_NO_LDS_MASS=1 WALNUTS_AMD_NO_FAR_END_SUMS=1 $B --no-cpu-baseline --no-parity-gate --waves-per-chain 16 --elems-per-lane -1 --model diag_normal --chains 8192 --dim 16384 --steps 8 --warmup 8 --adapt-iters 100 > $OUT/bench_cfg4_round3_byte_budget.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --phase warmup --model diag_normal --chains 8192 --dim 16384 --steps 8 --warmup 8 --adapt-iters 100 > $OUT/bench_cfg4_warmup.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --model diag_normal --chains 8192 --dim 12000 --steps 8 --warmup 8 --adapt-iters 100 > $OUT/bench_diag_12000.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --model std_normal --chains 8192 --dim 16384 --steps 8 --warmup 8 --adapt-iters 100 > $OUT/bench_std_16384.json 2>> $OUT/bench.err
# 4 097-8 192 dimensions: the held streaming kernels (the default for one-pass gradients) beside the (16, 8) register kernels
$B --no-cpu-baseline --no-parity-gate --model diag_normal --chains 8192 --dim 8192 --steps 8 --warmup 8 --adapt-iters 100 > $OUT/bench_diag_8192.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --waves-per-chain 16 --elems-per-lane 8 --model diag_normal --chains 8192 --dim 8192 --steps 8 --warmup 8 --adapt-iters 100 > $OUT/bench_diag_8192_register_kernels.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --model diag_normal --chains 8192 --dim 6000 --steps 8 --warmup 8 --adapt-iters 100 > $OUT/bench_diag_6000.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --waves-per-chain 16 --elems-per-lane 8 --model diag_normal --chains 8192 --dim 6000 --steps 8 --warmup 8 --adapt-iters 100 > $OUT/bench_diag_6000_register_kernels.json 2>> $OUT/bench.err
# the funnel's two passes on sixteen wavefronts streaming both ends (round 4's geometry) beside the held default
$B --no-cpu-baseline --no-parity-gate --waves-per-chain 16 --elems-per-lane -1 --model funnel --chains 8192 --dim 16384 --steps 8 --warmup 8 --adapt-iters 60 > $OUT/bench_funnel_16384_both_ends_streamed.json 2>> $OUT/bench.err
$B --no-cpu-baseline --config 5 --steps 16 --warmup 8 > $OUT/bench_cfg5_one_gpu.json 2>> $OUT/bench.err
$B --no-cpu-baseline --model funnel --chains 16384 --dim 1024 --adapt-iters 150 > $OUT/bench_funnel_1024.json 2>> $OUT/bench.err
$B --no-cpu-baseline --model rw1 --chains 16384 --dim 1024 --adapt-iters 150 > $OUT/bench_rw1_1024.json 2>> $OUT/bench.err
$B --no-cpu-baseline --fma 0 > $OUT/bench_headline_every_product_rounded.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --transitions-per-launch 1 > $OUT/bench_headline_one_transition_per_launch.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --chain-groups 1 > $OUT/bench_headline_one_chain_group.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --chain-groups 1 --model ill_normal --chains 4096 --dim 1024 --adapt-iters 300 > $OUT/bench_cfg2_one_chain_group.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --chain-groups 1 --model funnel --chains 16384 --dim 128 --adapt-iters 300 > $OUT/bench_cfg3_one_chain_group.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --chains 8192 --steps 40 > $OUT/bench_shard_8192.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --chains 16384 --steps 40 > $OUT/bench_shard_16384.json 2>> $OUT/bench.err
$B --no-cpu-baseline --no-parity-gate --chains 32768 --steps 40 > $OUT/bench_shard_32768.json 2>> $OUT/bench.err
$B --no-cpu-baseline --model funnel --chains 8192 --dim 16384 --steps 8 --warmup 8 --adapt-iters 60 --gate-chains 16 --gate-transitions 4 > $OUT/bench_funnel_16384_streaming.json 2>> $OUT/bench.err
$B --no-cpu-baseline --model rw1 --chains 8192 --dim 16384 --steps 8 --warmup 8 --adapt-iters 60 --gate-chains 16 --gate-transitions 4 > $OUT/bench_rw1_16384_streaming.json 2>> $OUT/bench.err
$B --gpus 2 --backend gloo --no-cpu-baseline --steps 16 --warmup 8 > $OUT/bench_gloo2_one_gpu.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/headline_trace -o t -- python3 $ROOT/bench.py --legs none --no-cpu-baseline --no-parity-gate > $OUT/bench_headline_under_trace.json 2>/dev/null
python3 $ROOT/profiles/summarize.py $R $(find $OUT/headline_trace -name "*results.db" | head -1) > $OUT/kernel_trace_headline.txt 2>&1
rm -rf $OUT/headline_trace
# ... and of the DRIVER's command with its legs: per kernel, the launch durations bench.py's avg_launch_ms must agree with
rocprofv3 --kernel-trace --stats -d $OUT/driver_trace -o t -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_line_under_trace.json 2>/dev/null
python3 $ROOT/profiles/summarize.py $R $(find $OUT/driver_trace -name "*results.db" | head -1) --per-kernel > $OUT/kernel_trace_driver_line.txt 2>&1
rm -rf $OUT/driver_trace
# the whole drop-in call, phase by phase (WALNUTS_AMD_TIMING), and a longer run of the randomised parity campaign
(cd $ROOT && WALNUTS_AMD_TIMING=1 timeout 900 python3 tests/gpu_probes/sample_device_e2e.py > $OUT/sample_device_e2e.txt 2>&1)
(cd $ROOT && timeout 600 python3 tests/gpu_probes/fuzz_parity.py --seconds 240 --seed 506 > $OUT/fuzz_parity.txt 2>&1; tail -3 $OUT/fuzz_parity.txt)
# (set COLLECT_PARITY=1 for the wide reference-order gate and the campaigns on the held streaming kernels as well: ~12 min)
if [ "${COLLECT_PARITY:-0}" = 1 ]; then
  (cd $ROOT && { echo "# device (fused multiply-adds, the default) against the reference-order oracle"; bash tests/gpu_probes/exp/wide_gate.sh;
     echo "# device with every product rounded (--fma 0)"; WIDE_GATE_ARGS="--fma 0" bash tests/gpu_probes/exp/wide_gate.sh; } > $OUT/parity_gate_wide.txt 2>&1)
  (cd $ROOT && timeout 400 python3 tests/gpu_probes/fuzz_parity.py --held --seconds 240 --seed 514 > $OUT/fuzz_parity_held.txt 2>&1; tail -1 $OUT/fuzz_parity_held.txt)
  (cd $ROOT && timeout 400 python3 tests/gpu_probes/fuzz_parity.py --held-two-pass --seconds 300 --seed 513 > $OUT/fuzz_parity_held_two_pass.txt 2>&1; tail -1 $OUT/fuzz_parity_held_two_pass.txt)
fi
for f in $OUT/bench_*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']
print('$(basename $f)', round(d['ms_per_step'],4), 'ms', '%.3e' % d['value'], r['bound'], round(r['frac'],4), 'gate' if 'parity_gate' in d else '', d.get('parity_gate',{}).get('max_rel_diff_logp'), d.get('parity_gate',{}).get('tree_mismatches'))"; done
