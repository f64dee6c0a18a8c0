cd $GRAFT_REPO_ROOT
echo "== timeline inline"; WALNUTS_AMD_PREGEN=0 python tests/gpu_probes/timeline.py 2>&1 | head -32
echo "== timeline pregen"; WALNUTS_AMD_PREGEN=1 python tests/gpu_probes/timeline.py 2>&1 | head -32
AB_ARGS="" bash tests/gpu_probes/exp/headline_ab.sh 2 prod stale
