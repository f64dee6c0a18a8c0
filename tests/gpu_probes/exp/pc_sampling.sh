#!/bin/bash
# PC-sampling histogram of the headline kernel (rocprofv3 beta feature): where one wavefront per SIMD spends its cycles,
# instruction by instruction.  pc_sampling.sh <method: stochastic|host_trap> <interval> [bench args ...]
# Writes gpurun_out/pcs_<method>/{header.txt,histogram.txt}; the raw CSV stays on the box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
METHOD=${1:-stochastic}; INTERVAL=${2:-65536}; shift 2
UNIT=cycles; [ "$METHOD" = host_trap ] && UNIT=time
OUT=$ROOT/gpurun_out/pcs_$METHOD
RAW=/tmp/pcs_raw_$METHOD
rm -rf "$RAW"; mkdir -p "$OUT" "$RAW"
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 420 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT \
  --pc-sampling-interval $INTERVAL --kernel-trace --output-format csv -d "$RAW" -- \
  python3 $ROOT/bench.py --no-cpu-baseline --no-parity-gate --steps 20 --warmup 3 "$@" > "$OUT/run.log" 2>&1
echo "rocprofv3 exit $?" >> "$OUT/run.log"
find "$RAW" -type f | head -50 > "$OUT/files.txt"
python3 $ROOT/tests/gpu_probes/exp/pc_histogram.py "$RAW" "$OUT"
