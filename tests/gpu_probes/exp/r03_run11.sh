cd $GRAFT_REPO_ROOT
export WALNUTS_AMD_TIMING=1
python tests/gpu_probes/sample_device_e2e.py 2>&1 | grep -v amdgpu.ids
