cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q 2>&1 | tail -3
python tests/gpu_probes/fuzz_parity.py --seed 303 --seconds 420 2>&1 | grep -v amdgpu.ids | tail -80 > gpurun_out/r03/fuzz_parity.txt; tail -5 gpurun_out/r03/fuzz_parity.txt
