cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
echo "== timeline fma pregen"; python tests/gpu_probes/timeline.py 2>&1 | head -32 | tee gpurun_out/r03/timeline_sampling_fma.txt
export PROFILE_ROUND=r03
bash profiles/pmc.sh headline_fma1 --fma 1 > /dev/null 2>&1; cat gpurun_out/pmc_headline_fma1/summary.txt
bash profiles/pmc.sh headline_fma0 --fma 0 > /dev/null 2>&1; cat gpurun_out/pmc_headline_fma0/summary.txt
