#!/bin/bash
# Runs HERE: one gpurun call that collects the round's end-state evidence (bench lines of every configuration, the
# kernel trace, the PMC passes), then files it under profiles/ keyed to the current walnuts_amd/csrc hash.
#   usage: PROFILE_ROUND=r04 bash profiles/final_collect.sh
cd /root/repo
export PROFILE_ROUND=${PROFILE_ROUND:-r04}
R=$PROFILE_ROUND
/usr/local/graft/bin/gpurun --timeout 3000 -- "export PROFILE_ROUND=$R; bash profiles/collect.sh; bash profiles/pmc.sh headline > /dev/null 2>&1; bash profiles/pmc.sh headline_warmup --phase warmup > /dev/null 2>&1; bash profiles/pmc.sh funnel_1024 --model funnel --chains 16384 --dim 1024 --adapt-iters 150 > /dev/null 2>&1; bash profiles/pmc.sh rw1_1024 --model rw1 --chains 16384 --dim 1024 --adapt-iters 150 > /dev/null 2>&1; bash profiles/pmc.sh cfg4 --model diag_normal --chains 8192 --dim 16384 --steps 8 > /dev/null 2>&1; bash profiles/pmc.sh cfg2 --model ill_normal --chains 4096 --adapt-iters 300 > /dev/null 2>&1; bash profiles/pmc.sh cfg3 --model funnel --chains 16384 --dim 128 --adapt-iters 300 > /dev/null 2>&1; for t in headline headline_warmup funnel_1024 rw1_1024 cfg4 cfg2 cfg3; do echo \"== \$t\"; grep -E 'HBM|dispatch' gpurun_out/pmc_\$t/summary.txt | head -3; done" 2>&1 | tail -50
python profiles/record_pmc.py headline headline_warmup funnel_1024 rw1_1024 cfg4 cfg2 cfg3
mkdir -p profiles/bench_$R profiles/$R && cp gpurun_out/$R/bench_*.json profiles/bench_$R/
cp gpurun_out/$R/kernel_trace_headline.txt profiles/$R/kernel_trace_headline.txt   # (summarised on the GPU box by collect.sh)
cp gpurun_out/$R/sample_device_e2e.txt profiles/$R/sample_device_e2e.txt
cp gpurun_out/$R/fuzz_parity.txt profiles/$R/fuzz_parity.txt
head -30 profiles/$R/kernel_trace_headline.txt
