#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 PMC passes of the headline bench (each counter group in its own
# run, never combined with tracing), summarised as per-launch values of the steady-state sampling dispatches.
#   usage: profiles/pmc.sh <tag> [bench.py args...]      -> gpurun_out/pmc_<tag>/summary.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WALNUTS_AMD_CHAIN_GROUPS=1   # (one dispatch per launch: the per-launch counter values below are one kernel's)
# (steps and warmup are multiples of the transitions per launch: every profiled dispatch is a full launch)
ARGS="--no-cpu-baseline --no-parity-gate --legs none --min-launches 2 --steps 16 --warmup 8 --adapt-iters 100 $*"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/p$i -o p -- python3 $ROOT/bench.py $ARGS > $OUT/p$i.log 2>&1
done
python3 - <<PY > $OUT/summary.txt
import sqlite3, glob
vals = {}
for db in sorted(glob.glob("$OUT/p*/**/*results.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select counter_name, value, duration from counters_collection where kernel_name like '%transition_kernel%' order by start").fetchall()
    names = sorted(set(r[0] for r in rows))
    for n in names:
        v = [r for r in rows if r[0] == n]
        # the last FULL launch = the steady state of the timed phase (a warmup run ends with the engine's short
        # observe-only launch -- wn_engine flush_pending_observation --, same kernel, a fraction of the duration)
        durs = sorted(r[2] for r in v)
        v = [r for r in v if r[2] >= 0.5 * durs[len(durs) // 2]]   # (half the median: the adaptation's first launches are long)
        vals[n] = (v[-1][1], v[-1][2] / 1e3)
print("# per launch of the transition kernel (last full-length dispatch = steady state; one launch = --transitions-per-launch transitions of every chain, default 8), args: $ARGS")
for n, (v, d) in vals.items():
    print(f"{n:28s} {v:18.1f}   (dispatch {d:.1f} us under this pass)")
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    f, w = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
    print(f"HBM traffic per launch = (2*FETCH_SIZE + WRITE_SIZE) KB = {(2*f+w)*1024/1e9:.3f} GB  (gfx950: FETCH_SIZE counts wide reads at 1/2)")
import argparse, hashlib, json, os
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="std_normal"); ap.add_argument("--chains", type=int, default=65536)
ap.add_argument("--dim", type=int, default=1024); ap.add_argument("--phase", default="sampling")
import sys
sys.path.insert(0, "$ROOT")
import bench
ap.add_argument("--transitions-per-launch", type=int, default=bench.DEFAULT_TRANSITIONS_PER_LAUNCH)
known, _ = ap.parse_known_args("$ARGS".split())
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    json.dump({"model": known.model, "chains": known.chains, "dim": known.dim, "phase": known.phase,
               "transitions_per_launch": max(1, known.transitions_per_launch), "csrc_sha": bench.csrc_sha(), "chain_groups": 1, "bytes_per_launch": (2 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024,
               "source": "profiles/${PROFILE_ROUND:-r06}/pmc_$TAG.txt: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, last "
                         "(steady-state) dispatch, (2*FETCH_SIZE + WRITE_SIZE)*1024 per the gfx950 note in MI355X_MICROARCH.md"},
              open("$OUT/traffic.json", "w"))
if "SQ_WAVE_CYCLES" in vals:
    wc = vals["SQ_WAVE_CYCLES"][0]
    for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if n in vals: print(f"{n}/SQ_WAVE_CYCLES = {vals[n][0]/wc:.3f}")
PY
cat $OUT/summary.txt
rm -rf $OUT/p[0-9] $OUT/p[0-9].log   # the rocpd databases stay on the box: summary.txt + traffic.json are what is filed
