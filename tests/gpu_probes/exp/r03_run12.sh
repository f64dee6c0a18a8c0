cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q 2>&1 | tail -4
export WALNUTS_AMD_TIMING=1
python tests/gpu_probes/sample_device_e2e.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/sample_device_e2e.txt
