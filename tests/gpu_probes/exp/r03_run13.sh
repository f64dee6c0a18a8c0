cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
echo "# tests/gpu_probes/exp/wide_gate.sh: bench.py's same-run parity gate widened to 512 chains x 24 transitions per configuration; one MI355X."
echo "# device against the oracle in the reference's arithmetic (libm, every product rounded) under the two reference-side summation"
echo "# orders: eigen_sse2 = Eigen 3.4's vectorised redux with 2-lane packets (restated), sequential = left-to-right loops."
echo "## device build: fused multiply-adds (the default)"
bash tests/gpu_probes/exp/wide_gate.sh
echo "## device build: every product rounded (--fma 0)"
WIDE_GATE_ARGS="--fma 0" bash tests/gpu_probes/exp/wide_gate.sh
} 2>&1 | tee gpurun_out/r03/parity_gate_wide.txt
