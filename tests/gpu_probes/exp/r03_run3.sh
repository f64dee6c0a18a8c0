cd $GRAFT_REPO_ROOT
AB_ARGS="" bash tests/gpu_probes/exp/headline_ab.sh 2 prio0 prio3
export WALNUTS_AMD_PREGEN=0
echo inline; AB_ARGS="" bash tests/gpu_probes/exp/headline_ab.sh 2 prio0 prio3
