set -x
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-parity-gate"
for r in 1 2; do
  WALNUTS_AMD_PREGEN=0 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('inline ', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), '%.4e' % d['value'])"
  WALNUTS_AMD_PREGEN=1 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pregen ', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), '%.4e' % d['value'])"
done
WALNUTS_AMD_PREGEN=1 $B --phase warmup 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pregen warmup', round(d['ms_per_step'],4), '%.4e' % d['value'])"
WALNUTS_AMD_PREGEN=0 $B --phase warmup 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('inline warmup', round(d['ms_per_step'],4), '%.4e' % d['value'])"
python bench.py --gpus 2 --backend gloo --chains 65536 --no-cpu-baseline > gpurun_out/r03/bench_gloo2_one_gpu.json 2> gpurun_out/r03/bench_gloo2_one_gpu.err; echo rc=$?; tail -c 600 gpurun_out/r03/bench_gloo2_one_gpu.json; tail -5 gpurun_out/r03/bench_gloo2_one_gpu.err
python bench.py > gpurun_out/r03/bench_headline.json 2>gpurun_out/r03/bench_headline.err; echo rc=$?; head -c 700 gpurun_out/r03/bench_headline.json
