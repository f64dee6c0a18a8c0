cd $GRAFT_REPO_ROOT
echo "== timeline preload"; python tests/gpu_probes/timeline.py 2>&1 | head -30
cp tests/gpu_probes/libwalnuts_tl_base.so tests/gpu_probes/libwalnuts_tl.so
echo "== timeline base"; python tests/gpu_probes/timeline.py 2>&1 | head -30
