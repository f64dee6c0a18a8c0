cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python tests/gpu_probes/timeline.py 2>&1 | grep -v amdgpu.ids | head -30 | tee gpurun_out/r03/timeline_sampling_end.txt
python tests/gpu_probes/timeline.py --warmup 2>&1 | grep -v amdgpu.ids | head -34 | tee gpurun_out/r03/timeline_warmup_end.txt
