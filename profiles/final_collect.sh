#!/bin/bash
# Runs HERE: the round's end-state evidence in TWO gpurun calls on the current sources -- first the PMC passes, filed
# under profiles/ and keyed to the walnuts_amd/csrc hash (so that the bench lines taken afterwards carry their
# `roofline.traffic`), then the bench lines of every configuration, the kernel trace, the drop-in call's phases and a
# parity campaign.
#   usage: PROFILE_ROUND=r06 bash profiles/final_collect.sh
cd /root/repo
export PROFILE_ROUND=${PROFILE_ROUND:-r06}
R=$PROFILE_ROUND
/usr/local/graft/bin/gpurun --timeout 2400 -- "export PROFILE_ROUND=$R; bash profiles/pmc.sh headline > /dev/null 2>&1; bash profiles/pmc.sh headline_warmup --phase warmup > /dev/null 2>&1; bash profiles/pmc.sh funnel_1024 --model funnel --chains 16384 --dim 1024 --adapt-iters 150 > /dev/null 2>&1; bash profiles/pmc.sh rw1_1024 --model rw1 --chains 16384 --dim 1024 --adapt-iters 150 > /dev/null 2>&1; bash profiles/pmc.sh cfg4 --model diag_normal --chains 8192 --dim 16384 --steps 8 > /dev/null 2>&1; bash profiles/pmc.sh cfg2 --model ill_normal --chains 4096 --adapt-iters 300 > /dev/null 2>&1; bash profiles/pmc.sh cfg3 --model funnel --chains 16384 --dim 128 --adapt-iters 300 > /dev/null 2>&1; bash profiles/pmc.sh funnel_16384 --model funnel --chains 8192 --dim 16384 --steps 8 --adapt-iters 60 > /dev/null 2>&1; bash profiles/pmc.sh rw1_16384 --model rw1 --chains 8192 --dim 16384 --steps 8 --adapt-iters 60 > /dev/null 2>&1; bash profiles/pmc.sh diag_4096 --model diag_normal --chains 8192 --dim 4096 --adapt-iters 60 > /dev/null 2>&1; bash profiles/pmc.sh diag_2048 --model diag_normal --chains 8192 --dim 2048 --adapt-iters 60 > /dev/null 2>&1; for t in headline headline_warmup funnel_1024 rw1_1024 cfg4 cfg2 cfg3 funnel_16384 rw1_16384 diag_4096 diag_2048; do echo \"== \$t\"; grep -E 'HBM|dispatch' gpurun_out/pmc_\$t/summary.txt | head -3; done" 2>&1 | tail -30
mkdir -p profiles/bench_$R profiles/$R
python profiles/record_pmc.py headline headline_warmup funnel_1024 rw1_1024 cfg4 cfg2 cfg3 funnel_16384 rw1_16384 diag_4096 diag_2048
/usr/local/graft/bin/gpurun --timeout 3000 -- "export PROFILE_ROUND=$R; export COLLECT_PARITY=${COLLECT_PARITY:-0}; bash profiles/collect.sh" 2>&1 | tail -40
cp gpurun_out/$R/bench_*.json profiles/bench_$R/
cp gpurun_out/$R/kernel_trace_headline.txt profiles/$R/kernel_trace_headline.txt   # (summarised on the GPU box by collect.sh)
cp gpurun_out/$R/kernel_trace_driver_line.txt profiles/$R/kernel_trace_driver_line.txt
cp gpurun_out/$R/sample_device_e2e.txt profiles/$R/sample_device_e2e.txt
cp gpurun_out/$R/fuzz_parity.txt profiles/$R/fuzz_parity.txt
for f in parity_gate_wide fuzz_parity_held fuzz_parity_held_two_pass; do   # (COLLECT_PARITY=1, see collect.sh)
  [ -s gpurun_out/$R/$f.txt ] && cp gpurun_out/$R/$f.txt profiles/$R/$f.txt
done
tail -4 profiles/$R/kernel_trace_headline.txt
