cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-parity-gate"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), '%.4e' % d['value'], d['config']['geometry'])"; }
$B 2>/dev/null | show "1x16"
$B --waves-per-chain 2 --elems-per-lane 8 2>/dev/null | show "2x8"
$B --waves-per-chain 4 --elems-per-lane 4 2>/dev/null | show "4x4"
$B --waves-per-chain 2 --elems-per-lane 8 --fma 0 2>/dev/null | show "2x8 fma0"
WALNUTS_AMD_PREGEN=0 $B --waves-per-chain 2 --elems-per-lane 8 2>/dev/null | show "2x8 inline"
export PROFILE_ROUND=r03
bash profiles/pmc.sh headline_2x8 --waves-per-chain 2 --elems-per-lane 8 > /dev/null 2>&1; cat gpurun_out/pmc_headline_2x8/summary.txt
