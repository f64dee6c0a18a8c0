cd $GRAFT_REPO_ROOT
WALNUTS_AMD_LIB=$GRAFT_REPO_ROOT/tests/gpu_probes/libwalnuts_pre.so python tests/gpu_probes/exp/quick_parity.py 2>&1 | tail -3
AB_ARGS="" bash tests/gpu_probes/exp/headline_ab.sh 3 prod pre
export WALNUTS_AMD_PREGEN=0
echo inline-momentum; AB_ARGS="" bash tests/gpu_probes/exp/headline_ab.sh 2 prod pre
