#!/usr/bin/env python3
"""bench.py -- leapfrog grad-evals/sec of the GPU-resident Walnuts engine (BASELINE.json metric).

A "step" is one MCMC transition (walnuts.hpp:520-563) of EVERY chain.  One launch of the persistent transition
kernel runs `--transitions-per-launch` consecutive steps (default 8: the workgroup that fetched a chain runs that
many transitions of it back to back -- the chains are independent -- so the launch and its tail, the last chains
finishing while the chip drains, are paid once per launch; every transition's draw plane is written).  `--steps` and
`--warmup` are MINIMA: the timed region holds whole launches only and at least `--min-launches` (12) of them -- the
first dispatches after a join run 10-50 % off steady state --, so the driver's `--steps 20 --warmup 5` times 96
transitions after 8 untimed ones; the line's `steps` / `warmup` are the transitions actually run, `steps_requested` /
`warmup_requested` the command line's (`--exact-steps`: exactly K, the last launch short).  Default workload =
BASELINE.json's headline: 65 536 chains x 1 024-dim standard normal, default SamplingConfig, parameters adapted by
`--adapt-iters` on-device warmup transitions (untimed), then the untimed and the timed sampling transitions.  Inputs are
generated on the device (counter-based stream) and are resident in HBM when the timed region starts.

The default headline run on one GPU also measures, in the SAME line under `configs`, BASELINE configs #2, #3 and #4 and
the headline's adaptive-warmup phase (throughput for sampling and separately for warmup) as short
legs -- each with `value`, `ms_per_step`, `roofline` (PMC traffic included) and its own reference-order `parity_gate`
(`--legs none` / a comma list; the whole command takes ~20 s).

  python bench.py --gpus N --steps K --warmup W

N > 1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or started
plainly -- then this process, before it touches the GPU, starts the N ranks itself (`launch_ranks`) and forwards
rank 0's JSON line; it fails when the node has fewer than N GPUs.

Multi-GPU (`--scaling`): "strong" (default, the north star's target: the 65 536 chains are SHARDED over the N
GPUs) or "weak" (`--chains` per GPU).  `--config 5` = BASELINE config #5: 262 144 chains sharded over the GPUs.
The path's only exchange is the all-gather of each iteration's draws (RCCL over xGMI), overlapped with the next
transition.

Rank 0 prints ONE JSON line.  `value` = gradient evaluations of all chains on all ranks in the timed region
/ max-over-ranks wall time.

`roofline` describes the dominant kernel, timed with HIP events on the stream it is launched on:
  * register kernels (D <= 8192: the trajectory end never leaves the chip): bound "fp64-valu".  `achieved` = useful
    fp64 flops (10*D per gradient evaluation) / kernel time against the 78.6 TFLOP/s fp64 vector
    peak; `traffic` = HBM bytes per launch from the rocprofv3 PMC passes recorded in profiles/pmc_traffic.json FOR
    THIS BUILD of walnuts_amd/csrc (entries carry the source hash; a stale entry is refused), `traffic_frac` = that
    / kernel time / 8 TB/s.  The 56*D "algorithmic" byte figure of the metric definition is reported under
    `algorithmic` only: those bytes never reach HBM, so their rate is not a bandwidth.
  * streaming kernels (D > 8192): bound "hbm", `achieved` = algorithmic 56*D bytes per gradient evaluation / time.
`parity_gate`: 64 chains replayed transition by transition on the CPU oracle in the REFERENCE's
arithmetic (libm, every product rounded) under both summation orders the reference side can have -- Eigen 3.4's SSE2
redux (restated from its published algorithm) and sequential loops -- from the device's own states with the same random
streams: max relative difference of the selected position per transition, tree agreement, and the count of decisions
within 1e-12 of a threshold, per order.
`cpu_baseline` = that oracle (a CPU port of the reference algorithm) timed on the host cores for a bounded sample.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TF = 78.6   # fp64 vector peak (half the 157.3 TFLOP/s fp32 vector peak of the same guide)
FLOPS_PER_GRAD_EVAL_PER_DIM = 10.0   # leapfrog 7*D + model 3*D
HEADLINE_CHAINS = 65536
DEFAULT_TRANSITIONS_PER_LAUNCH = 8   # consecutive transitions of every chain per kernel launch (wn_engine_sample_steps)
CONFIG5_CHAINS = 262144


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--chains", type=int, default=HEADLINE_CHAINS,
                    help="total chains (strong scaling) or chains per GPU (weak scaling)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: --chains is the job's total, sharded over the GPUs (north star); weak: per GPU")
    ap.add_argument("--config", type=int, default=0, help="5 = BASELINE config #5 (262 144 chains, sharded)")
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--model", default="std_normal", choices=["std_normal", "diag_normal", "ill_normal", "funnel", "rw1"],
                    help="diag_normal: sigma_d = 1 + (d mod 16) (config #4); ill_normal: sigma_d = d + 1 (config #2, "
                         "examples/examples.cpp:20-31)")
    ap.add_argument("--adapt-iters", type=int, default=100, help="untimed adaptive warmup transitions")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--waves-per-chain", type=int, default=0)
    ap.add_argument("--elems-per-lane", type=int, default=0)
    ap.add_argument("--workgroups-per-cu", type=int, default=0)
    ap.add_argument("--lds-vectors", type=int, default=-1)
    ap.add_argument("--fma", type=int, default=-1, choices=[-1, 0, 1],
                    help="1: fused multiply-adds in the integrator (library default), 0: every product rounded (the "
                         "element-wise bits of the reference's x86-64 -O3 build); -1: the library default")
    ap.add_argument("--reserved-cus", type=int, default=-1,
                    help="CUs left free for the RCCL all-gather kernels (default: 0 on one GPU, 16 otherwise)")
    ap.add_argument("--transitions-per-launch", type=int, default=DEFAULT_TRANSITIONS_PER_LAUNCH,
                    help="transitions of every chain per kernel launch (wn_engine_sample_steps): the workgroup that "
                         "fetched a chain runs them back to back; a step stays ONE transition of all chains")
    ap.add_argument("--chain-groups", type=int, default=0,
                    help="wn_config::chain_groups: 0 = the engine's choice (2 when there are more chains than resident "
                         "workgroups), 1-4 = that many independently launched chain groups")
    ap.add_argument("--adopt-stream", action="store_true",
                    help="N > 1: launch the kernels on torch's current stream (wn_engine_set_stream, one chain group) "
                         "instead of ordering the engine's own streams against it with events")
    ap.add_argument("--order-streams", action="store_true",
                    help="exercise the N > 1 stream ordering (wn_engine_wait_stream / _release_stream) on one GPU too")
    ap.add_argument("--gather-method", choices=["collective", "p2p"], default="collective",
                    help="collective: all_gather_into_tensor (RCCL picks the algorithm); p2p: all-pairs, every block "
                         "as its own point-to-point transfer (grouped send/recv: one xGMI link per pair)")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="all-gather the draw block of every k-th launch (1 = every draw of every transition, the north star's "
                         "exchange: one collective per launch on the block of its draw planes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-parity-gate", action="store_true")
    ap.add_argument("--gate-chains", type=int, default=64)
    ap.add_argument("--gate-transitions", type=int, default=8)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend (gloo only to exercise the N>1 code path when ranks share one GPU)")
    ap.add_argument("--per-launch-events", action="store_true",
                    help="time every launch with its own pair of HIP events (two events between two kernels) instead of "
                         "one pair around the timed region")
    ap.add_argument("--phase", default="sampling", choices=["sampling", "warmup"],
                    help="which transition kind is timed")
    ap.add_argument("--min-launches", type=int, default=12,
                    help="the timed region holds at least this many whole launches (--steps is a minimum)")
    ap.add_argument("--exact-steps", action="store_true",
                    help="time exactly --steps transitions (the last launch short) instead of whole launches")
    ap.add_argument("--legs", default="auto",
                    help="auto: the default headline run on one GPU also measures BASELINE configs #2-#4 and the headline's "
                         "warmup phase (short legs under \"configs\" of the same JSON line); none; or a comma list of "
                         "cfg2,cfg3,cfg4,headline_warmup")
    return ap.parse_args()


def model_setup(name, D):
    import walnuts_amd as wa

    if name == "std_normal":
        return wa.MODEL_STD_NORMAL, None
    if name == "funnel":
        return wa.MODEL_FUNNEL, None
    if name == "rw1":
        return wa.MODEL_RW1, None   # examples/examples.cpp:34-49 through the public model interface
    if name == "ill_normal":
        return wa.MODEL_DIAG_NORMAL, np.array([(d + 1.0) ** 2 for d in range(D)])
    return wa.MODEL_DIAG_NORMAL, np.array([(1.0 + (d % 16)) ** 2 for d in range(D)])


def oracle_model(name):
    import wno

    return {"std_normal": wno.MODEL_STD_NORMAL, "diag_normal": wno.MODEL_DIAG_NORMAL,
            "ill_normal": wno.MODEL_DIAG_NORMAL, "funnel": wno.MODEL_FUNNEL, "rw1": wno.MODEL_RW1}[name]


def csrc_sha():
    """Hash of the kernel sources this library was built from (what a recorded PMC measurement is valid for): every
    source file under walnuts_amd/csrc, sub-directories (models/) included, plus out-of-tree model sources named in
    $MODELS.  profiles/pmc.sh calls this function -- one definition."""
    d = os.path.join(ROOT, "walnuts_amd", "csrc")
    files = []
    for base, _, names in os.walk(d):
        files += [os.path.join(base, f) for f in names if f.endswith((".h", ".hip", ".inc")) or f == "Makefile"]
    files = sorted(files, key=lambda p: os.path.relpath(p, d))
    files += [p for p in os.environ.get("MODELS", "").split() if os.path.isfile(p)]
    h = hashlib.sha256()
    for p in files:
        h.update(os.path.relpath(p, d).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(args, D, chains_local, transitions_per_average_launch):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes recorded under profiles/ (counters
    are collected in their own runs, never inside a timed bench run).  Only an entry recorded for THIS source hash of
    walnuts_amd/csrc and THIS number of transitions per launch counts; anything else is reported as stale and not
    used.  The profiled dispatch ran a full launch; when K is not a multiple of the transitions per launch the average
    launch of the timed region is shorter and the figure is scaled by transitions."""
    switched = [v for v in ("WALNUTS_AMD_NO_LDS_MASS", "WALNUTS_AMD_NO_FAR_END_SUMS", "WALNUTS_AMD_LIB") if os.environ.get(v)]
    if switched:   # (the recorded passes ran the library as built, with its default switches)
        return None, "not recorded for this run's switches (" + ", ".join(switched) + ")"
    # ... and the engine's own choice of kernel: a run that asks for a launch geometry runs another kernel
    asked = [f"--{k.replace('_', '-')} {getattr(args, k)}" for k, default in
             (("waves_per_chain", 0), ("elems_per_lane", 0), ("workgroups_per_cu", 0), ("lds_vectors", -1))
             if getattr(args, k) != default]
    if asked:
        return None, "not recorded for this run's launch geometry (" + ", ".join(asked) + ")"
    try:
        entries = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except OSError:
        return None, "no profiles/pmc_traffic.json"
    sha = csrc_sha()
    stale = None
    for e in entries:
        if (e["model"], e["chains"], e["dim"], e["phase"], e.get("transitions_per_launch", 1)) == (
                args.model, chains_local, D, args.phase, max(1, args.transitions_per_launch)):
            if e.get("csrc_sha") == sha:
                scale = transitions_per_average_launch / e.get("transitions_per_launch", 1)
                # (the counter passes run ONE chain group so that a launch is one dispatch; the same chains do the same
                # work in two concurrent kernels, so the bytes per launch carry over -- but say so)
                groups = (f"; recorded with {e['chain_groups']} chain group(s) per launch, this run: "
                          f"{os.environ.get('WALNUTS_AMD_CHAIN_GROUPS') or args.chain_groups or 'the engine default (2)'}"
                          if "chain_groups" in e else "")
                return e["bytes_per_launch"] * scale, ("RECORDED, not measured in this run (" + e["source"] +
                                                        "; same source hash, same workload" + groups + ")")
            stale = f"stale PMC entry refused (recorded for csrc {e.get('csrc_sha')}, this build is {sha})"
    return None, stale or "this workload was not profiled"


def host_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container on a
    256-thread host is often limited to a few cores; os.cpu_count() does not know)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, D):
    """The oracle timed on the host cores: reference arithmetic order (left-to-right sums, libm), persistent worker
    threads, chains dealt out dynamically.  Bounded: 8 chains per core for ~args.cpu_seconds of sampling transitions
    on all cores, then 8 chains on ONE core for a third of that."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wno

    cores = host_cores()
    _, params = model_setup(args.model, D)

    def run(chains, threads, seconds):
        cfg = wno.default_config(rng_mode=wno.RNG_PHILOX, math_mode=wno.MATH_LIBM, reduce_lanes=0)
        e = wno.Engine(oracle_model(args.model), D, chains, cfg, params=params)
        e.init_positions(args.seed, 0, 2.0)
        e.init_masses_from_grad(1e-5)
        e.set_step_sizes(1.0)
        e.adapt_step(args.seed, 0)
        e.seed_chains(args.seed + 1, 0)
        adapt = min(args.adapt_iters, 100)
        for _ in range(adapt):
            e.warmup_step(threads)
        if args.phase == "sampling":
            e.freeze()
            step = lambda: e.sample_step(threads)
        else:
            step = lambda: e.warmup_step(threads)
        step()
        g0 = int(e.grad_evals().sum())
        t0 = time.perf_counter()
        n = 0
        while True:
            step()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or n >= 100000:
                break
        return (int(e.grad_evals().sum()) - g0) / dt, adapt, n, dt

    v_all, adapt, n, dt = run(8 * cores, cores, args.cpu_seconds)
    v_one, _, n1, dt1 = run(8, 1, max(2.0, args.cpu_seconds / 3))
    return {"value": v_all, "unit": "grad-evals/s", "cores": cores, "kind": "port",
            "per_core": v_all / cores, "one_core": v_one,
            "sample": f"{8 * cores} chains x {D}-dim {args.model}, {adapt} adaptive warmup transitions then {n} "
                      f"{args.phase} transitions in {dt:.1f} s on {cores} persistent threads; one_core: 8 chains, "
                      f"{n1} transitions in {dt1:.1f} s on 1 thread (oracle: CPU port of the reference algorithm, "
                      "sequential sums + libm; the reference Eigen build is unavailable: Eigen is not vendored, "
                      "CMakeLists.txt:28-41)"}


def parity_gate(args, D, cfg_kwargs):
    return reference_order_gate(args.model, D, cfg_kwargs, chains=args.gate_chains, transitions=args.gate_transitions,
                                adapt_iters=args.adapt_iters, seed=args.seed, phase=args.phase)


def reference_order_gate(model, D, cfg_kwargs, *, chains=64, transitions=8, adapt_iters=100, seed=1234, phase="sampling",
                         lib_path=None, model_id=None, params=None, oracle_model_id=None, init_scale=2.0,
                         step_size=None):
    """SURVEY.md section 8d "parity gate in the same run": a subset of chains replayed on the CPU restatement in the
    REFERENCE's arithmetic (libm exp/log, every product rounded) with the identical random stream, one transition at a
    time from the device's own state -- under BOTH summation orders the reference side can have: "eigen_sse2", the
    order of Eigen 3.4's vectorised redux with 2-lane packets (what `.sum()` / `.dot()` execute in the reference's
    default x86-64 build; restated from Eigen's published algorithm, oracle/wn_oracle.cpp), and "sequential", plain
    left-to-right loops.  Per order: the max relative difference of the selected position per transition (north star:
    <= 1e-10), how many chains built the identical tree, and how many decisions sat within 1e-12 (relative) of their
    threshold -- |H0-H1| <= max_error (walnuts.hpp:339), the U-turn signs (:199-200), log u < delta (:379) -- i.e.
    where a different summation order could have flipped the tree.  The top-level fields are the worst over both.
    phase "warmup": the gated transitions are ADAPTIVE ones (adaptive_walnuts.hpp:234-251) -- the oracle is handed the
    device's Adam state, mass-estimator planes and min-micro-steps value before each, and Adam's six numbers and the
    estimator's planes after the transition are compared as well (`max_rel_diff_adapt`)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import walnuts_amd as wa
    import wno

    Cg, T = chains, transitions
    if model_id is None:
        model_id, params = model_setup(model, D)
    dcfg = wa.default_config(lib_path, **cfg_kwargs)
    dev = wa.DeviceEngine(model_id, D, Cg, dcfg, params=params, lib_path=lib_path)
    orders = {"eigen_sse2": wno.REDUCE_EIGEN_SSE2, "sequential": 0}
    orcs = {}
    # (SamplingConfig / WarmupConfig fields among cfg_kwargs go to both sides; launch geometry etc. to the device only)
    shared = {k: v for k, v in cfg_kwargs.items() if hasattr(wno.Config(), k) and k not in ("math_mode", "reduce_lanes", "rng_mode", "fma")}
    for name, lanes in orders.items():
        ocfg = wno.default_config(rng_mode=wno.RNG_PHILOX, math_mode=wno.MATH_LIBM, reduce_lanes=lanes, **shared)
        orcs[name] = wno.Engine(oracle_model(model) if oracle_model_id is None else oracle_model_id, D, Cg, ocfg, params=params)
        orcs[name].set_tie_tolerance(1e-12)
        orcs[name].seed_chains(seed + 1, 0)
    dev.init_positions(seed, 0, init_scale)
    dev.init_masses_from_grad(1e-5)
    dev.set_step_sizes(1.0 if step_size is None else step_size)
    if step_size is None:
        dev.adapt_step(seed, 0)   # (step_size: that step as it is, no search)
    dev.seed_chains(seed + 1, 0)
    adapt = min(adapt_iters, 40)
    warm = phase == "warmup"
    for _ in range(adapt):
        dev.warmup_step()
    if warm:
        for orc in orcs.values():   # (the adapters must exist before their state is handed in)
            orc.set_positions(dev.positions())
            orc.init_masses_from_grad(1e-5)
            orc.set_step_sizes(1.0)
    else:
        dev.freeze()
    dev.synchronize()
    inv_mass, steps, mm = dev.inv_mass(), dev.step_sizes(), dev.min_micro()
    rec = {name: {"rel": [], "rel_lp": [], "same": [], "rel_adapt": []} for name in orders}

    def plane_rel(x, y):
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        return float(np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-300))) if x.size else 0.0

    for t in range(T):
        pos = dev.positions()
        g_dev0 = dev.grad_evals().copy()
        g_orc0 = {}
        if warm:
            adam0, est0, mm0, it0 = dev.adam(), dev.estimator(), dev.min_micro(), dev.iteration
        for name, orc in orcs.items():
            orc.set_positions(pos)
            if warm:
                orc.set_adapt_state(adam0, est0, mm0, it0)
            else:
                orc.set_sampler_state(inv_mass, steps, mm)
            orc.set_transition_index(adapt + t)
            g_orc0[name] = orc.grad_evals().copy()
        dev.warmup_step() if warm else dev.sample_step()
        for orc in orcs.values():
            orc.warmup_step(host_cores()) if warm else orc.sample_step(host_cores())
        dev.synchronize()
        a, la = dev.positions(), dev.logp()
        for name, orc in orcs.items():
            b, lb = orc.positions(), orc.logp()
            same = (dev.depths() == orc.depths()) & ((dev.grad_evals() - g_dev0) == (orc.grad_evals() - g_orc0[name]))
            if warm:
                de, oe = dev.estimator(), orc.estimator()
                # Adam: theta, m, v, t, b1pow, b2pow (m passes through zero: relative to the row's largest magnitude)
                da, oa = dev.adam()[same], orc.adam()[same]
                ra = float(np.max(np.abs(da - oa) / np.maximum(np.max(np.abs(oa), axis=1, keepdims=True), 1e-300))) if same.any() else 0.0
                # the estimator's means pass through zero as well: relative to each chain's largest entry of the plane
                rp = max(float(np.max(np.max(np.abs(de[k][same] - oe[k][same]), axis=1)
                                      / np.maximum(np.max(np.abs(oe[k][same]), axis=1), 1e-300))) if same.any() else 0.0
                         for k in ("draw_mean", "draw_ssd", "score_mean", "score_ssd"))
                rec[name]["rel_adapt"].append(max(ra, rp, plane_rel(de["weights"][same], oe["weights"][same])))
            r = np.max(np.abs(a - b), axis=1) / np.maximum(np.max(np.abs(b), axis=1), 1e-300)
            # the selected position is a leaf of element-wise leapfrog arithmetic; its log density is a sum over D
            # and shows the summation orders
            rl = np.abs(la - lb) / np.maximum(np.abs(lb), 1e-300)
            rec[name]["rel"].append(float(r[same].max()) if same.any() else float("nan"))
            rec[name]["rel_lp"].append(float(rl[same].max()) if same.any() else float("nan"))
            rec[name]["same"].append(int(same.sum()))
    out_orders = {}
    for name, orc in orcs.items():
        ties = orc.near_ties()
        r = rec[name]
        out_orders[name] = {
            "max_rel_diff_per_transition": r["rel"], "max_rel_diff": float(np.nanmax(r["rel"])),
            "max_rel_diff_logp": float(np.nanmax(r["rel_lp"])),
            "chains_with_identical_tree_per_transition": r["same"],
            "tree_mismatches": int(Cg * T - sum(r["same"])),
            "near_ties_1e-12": {k: {"near": v[0], "decisions": v[1]} for k, v in ties.items()}}
        if warm:
            out_orders[name]["max_rel_diff_adapt"] = float(np.max(r["rel_adapt"]))
    worst = max(o["max_rel_diff"] for o in out_orders.values())
    worst_lp = max(o["max_rel_diff_logp"] for o in out_orders.values())
    lanes_per_chain, streaming = dev.lanes, bool(dev.streaming)
    held_tiles = dev.held_tiles if streaming else 0
    dev.close()
    return {"chains": Cg, "transitions": T, "phase": phase,
            **({"max_rel_diff_adapt": max(o["max_rel_diff_adapt"] for o in out_orders.values())} if warm else {}),
            "device_arithmetic": "fused multiply-adds" if dcfg.fused_multiply_add else "every product rounded",
            "kernel": {"lanes_per_chain": lanes_per_chain, "streaming": streaming, "held_tiles": held_tiles},
            "reference_side": "libm, every product rounded; summation orders: Eigen 3.4 SSE2 redux (restated) and sequential",
            "orders": out_orders,
            "max_rel_diff": worst, "max_rel_diff_logp": worst_lp,
            "within_1e-10": bool(worst <= 1e-10 and worst_lp <= 1e-10),
            "tree_mismatches": int(max(o["tree_mismatches"] for o in out_orders.values())),
            "near_ties_1e-12": out_orders["eigen_sse2"]["near_ties_1e-12"]}


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rccl_algorithm_seen():
    """Which algorithm / protocol RCCL used for the draw all-gather, from its own log (rank 0: NCCL_DEBUG=INFO with the
    TUNING and COLL subsystems written to NCCL_DEBUG_FILE, set up in main() before the process group exists).  Best
    effort: the wording of those lines is the library's; what is returned is the matching lines' own text."""
    path = os.environ.get("WN_RCCL_LOG")
    if not path or not os.path.exists(path):
        return "no RCCL log (NCCL_DEBUG_FILE not written)"
    seen = []
    try:
        for line in open(path, errors="replace"):
            low = line.lower()
            if "allgather" in low and ("algo" in low or "proto" in low):
                text = line.split("NCCL INFO", 1)[-1].strip()
                if text not in seen:
                    seen.append(text)
    except OSError as e:
        return f"RCCL log unreadable: {e}"
    return seen[:6] if seen else "RCCL log holds no AllGather algorithm line"


def count_devices_without_hip():
    """GPUs of this node counted from the KFD topology in sysfs (a node with simd_count > 0 is a GPU): no HIP call, no
    context left on the devices while the parent waits for its ranks.  torch.cuda.device_count() stays off HIP only
    while its amdsmi path works and falls back to hipGetDeviceCount otherwise.  None when the topology cannot be read:
    the pre-check is then skipped and a shortage is reported by the ranks themselves."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir("/sys/class/kfd"):
        return 0   # no KFD driver: no AMD GPU on this node
    try:
        n = 0
        for node in os.listdir(base):
            for line in open(os.path.join(base, node, "properties")):
                if line.startswith("simd_count ") and int(line.split()[1]) > 0:
                    n += 1
        visible = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if visible:
            n = min(n, len([v for v in visible.split(",") if v.strip() != ""]))
        return n
    except (OSError, ValueError):
        return None


def launch_ranks(args, argv, script=None, device_count=None):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: THIS process never touches the GPU (it
    counts devices from the KFD topology in sysfs, and nothing else); it starts the N ranks as fresh
    child processes through torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1), forwards rank 0's
    JSON line to stdout (everything else the children print goes to stderr) and returns their exit code.
    Fewer than N devices is an error -- except with `--backend gloo`, where the ranks share the devices there are
    (rank r -> device r mod device_count): that exercises the N > 1 path on a one-GPU box, it is not a scaling run."""
    import subprocess

    if device_count is None:
        device_count = count_devices_without_hip()
    if device_count is not None and device_count < args.gpus and args.backend != "gloo":
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this node shows {device_count} "
              "(use --backend gloo to run the ranks on the devices there are)", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = 0
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            rec = json.loads(line)
            if rec.get("n_gpus") != args.gpus:
                print(f"bench.py: the ranks report n_gpus={rec.get('n_gpus')}, --gpus was {args.gpus}", file=sys.stderr)
                proc.wait()
                return 3
            sys.stdout.write(line)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {lines}", file=sys.stderr)
        return 4
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl" and rank == 0 and "NCCL_DEBUG" not in os.environ:
            # (so that the line can say which all-gather algorithm ran: rccl_algorithm_seen)
            log = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"wn_rccl_{os.getpid()}.log")
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,COLL,TUNING", NCCL_DEBUG_FILE=log,
                              WN_RCCL_LOG=log)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    out = run_job(args, rank, local_rank, world, dist)
    if rank == 0:
        legs = config_legs(args, world)
        if legs:
            # the other single-GPU BASELINE configurations and the headline's warmup phase, measured in this same run
            # (throughput for the sampling phase and separately for warmup)
            out["configs"] = {}
            for name, spec in legs.items():
                t_leg = time.perf_counter()
                leg_args = argparse.Namespace(**{**vars(args), **spec})
                try:
                    rec = run_job(leg_args, rank, local_rank, world, dist, cpu=False)
                    out["configs"][name] = {k: rec[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "config",
                                                                "roofline", "parity_gate") if k in rec}
                except Exception as exc:   # (a leg that cannot run must not cost the line its headline)
                    import traceback

                    out["configs"][name] = {"error": f"{type(exc).__name__}: {exc}",
                                            "traceback": traceback.format_exc().splitlines()[-6:]}
                    torch.cuda.empty_cache()
                out["configs"][name]["leg_wall_s"] = time.perf_counter() - t_leg
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# BASELINE.json configs #2-#4 (definitions: SURVEY.md section 8d, examples/examples.cpp:20-31) and the headline's warmup
# phase: short legs of the default single-GPU run, each with its own roofline and reference-order parity gate
CONFIG_LEGS = {
    "cfg2": dict(model="ill_normal", chains=4096, dim=1024, adapt_iters=300, phase="sampling",
                 gate_chains=64, gate_transitions=4),
    "cfg3": dict(model="funnel", chains=16384, dim=128, adapt_iters=300, phase="sampling",
                 gate_chains=64, gate_transitions=4),
    "cfg4": dict(model="diag_normal", chains=8192, dim=16384, adapt_iters=100, phase="sampling",
                 gate_chains=64, gate_transitions=4),
    "headline_warmup": dict(model="std_normal", chains=HEADLINE_CHAINS, dim=1024, adapt_iters=100, phase="warmup",
                            gate_chains=64, gate_transitions=4),
}


def config_legs(args, world):
    """Which extra legs this invocation runs: `--legs auto` (default) = all of CONFIG_LEGS when the line is the default
    headline workload on one GPU (what the driver runs), none otherwise; `--legs none`; or a comma list."""
    if world != 1 or args.legs == "none":
        return {}
    if args.legs == "auto":
        headline = (args.model, args.chains, args.dim, args.phase, args.config, args.scaling) == (
            "std_normal", HEADLINE_CHAINS, 1024, "sampling", 0, "strong")
        geometry = (args.waves_per_chain, args.elems_per_lane, args.workgroups_per_cu, args.lds_vectors,
                    args.chain_groups) != (0, 0, 0, -1, 0)
        return dict(CONFIG_LEGS) if headline and not geometry and not args.no_parity_gate else {}
    return {k: CONFIG_LEGS[k] for k in args.legs.split(",")}


def run_job(args, rank, local_rank, world, dist, cpu=True):
    """One workload on this process group: engine set-up, untimed adaptation and warm-up launches, the timed region,
    roofline, and on one GPU the reference-order parity gate (and with `cpu` the CPU baseline).  Returns the record on
    rank 0."""
    import torch

    import walnuts_amd as wa
    from walnuts_amd.distributed import DrawGather, shard_chains

    D = args.dim
    if args.config == 5:
        args.scaling, args.chains = "strong", CONFIG5_CHAINS
    if args.scaling == "strong":
        total_chains = args.chains
        chain0, C = shard_chains(total_chains, rank, world)
    else:
        C = args.chains
        total_chains = C * world
        chain0 = rank * C
    model_id, params = model_setup(args.model, D)
    # the transition kernel is persistent and would hold every CU for the whole step; with more than one GPU a
    # few CUs stay free so that RCCL's all-gather of the previous draws really runs underneath it
    reserved = args.reserved_cus if args.reserved_cus >= 0 else (16 if world > 1 else 0)
    cfg_kwargs = dict(device=local_rank, waves_per_chain=args.waves_per_chain, elems_per_lane=args.elems_per_lane,
                      workgroups_per_cu=args.workgroups_per_cu, lds_vectors=args.lds_vectors,
                      chain_groups=args.chain_groups)
    if args.fma >= 0:
        cfg_kwargs["fused_multiply_add"] = args.fma
    cfg = wa.default_config(reserved_cus=reserved, **cfg_kwargs)
    eng = wa.DeviceEngine(model_id, D, C, cfg, params=params)
    # With more than one GPU the draws are gathered by collectives torch issues behind its current stream.  The engine
    # keeps its own streams (and its chain groups) and is ordered against that stream with events in both directions:
    # a launch waits for the collective that last read the buffer it overwrites, the collective waits for the launch.
    # (--adopt-stream: the kernels on torch's stream itself, one chain group -- the round-3 arrangement.)
    # The two directions use two streams: the collectives are issued behind torch's current stream (`torch_stream`),
    # their completion is waited for on `inbound` -- were it the same stream, a launch would also wait for the previous
    # launch of EVERY group (which that stream was told to wait for), and the groups would run in lock step again.
    torch_stream = torch.cuda.current_stream().cuda_stream if (world > 1 or args.order_streams) else None
    inbound = torch.cuda.Stream() if torch_stream is not None else None
    if torch_stream is not None and args.adopt_stream:
        eng.set_stream(torch_stream)
        torch_stream = inbound = None
    # InitConfigBuilder on the device: positions ~ N(0, 2^2) (init_radius, pyfunc.py:57), masses from the
    # gradient with smoothing 1e-5, step-size search from step_size_init = 1.0.  Streams are keyed by the GLOBAL
    # chain id, so the result does not depend on the sharding.
    eng.init_positions(args.seed, chain0, 2.0)
    eng.init_masses_from_grad(1e-5)
    eng.set_step_sizes(1.0)
    eng.adapt_step(args.seed, chain0)
    eng.seed_chains(args.seed + 1, chain0)

    T = max(1, args.transitions_per_launch)
    gather = DrawGather(dist, world, rank, total_chains, D, "cuda", torch.float64,
                        counts=None if args.scaling == "strong" else [C] * world, transitions=T,
                        method=args.gather_method)

    def run_steps(first, count, timed_phase, compute=True, exchange=True):
        """`count` steps (transitions of all chains) starting at step `first`, T per launch.  compute / exchange: the
        two halves of a step, switched off one at a time by the N > 1 run's separate legs."""
        launches = 0
        i = first
        while i < first + count:
            n = min(T, first + count - i)
            launch_id = i // T
            if inbound is not None:
                with torch.cuda.stream(inbound):
                    block = gather.buffer(launch_id)   # (`inbound` waits for the collective that last read the buffer)
            else:
                block = gather.buffer(launch_id)   # [rows, D], or [T, rows, D]: one draw plane per transition
            step_fn = eng.warmup_steps if timed_phase == "warmup" else eng.sample_steps
            if compute:
                if inbound is not None:
                    eng.wait_stream(inbound.cuda_stream)
                step_fn(n, block.data_ptr(), D, gather.rows * D)
                if torch_stream is not None:
                    eng.release_stream(torch_stream)   # the collective below is issued behind torch's stream
            # the path's only exchange: all-gather of the launch's draws over xGMI, overlapped with the next launch
            # (no-op on one GPU)
            if exchange and launch_id % args.gather_every == 0:
                gather.launch(launch_id)
            launches += 1
            i += n
        return launches

    def fence():
        gather.drain()
        if world > 1:
            dist.barrier()
        eng.synchronize()
        torch.cuda.synchronize()

    # --steps / --warmup are MINIMA: the timed region holds whole launches only (T transitions each), and at least
    # --min-launches of them -- the first dispatches after a join run 10-50 % off steady state, and a region of 8+8+4
    # transitions is a fragile sample.  `steps` / `warmup` in the line are the transitions actually run;
    # `steps_requested` / `warmup_requested` what the command line asked for.  (--exact-steps: exactly K, last launch short.)
    steps_requested, warmup_requested = args.steps, args.warmup
    if args.exact_steps:
        n_steps, n_warm = args.steps, args.warmup
    else:
        n_steps = max(-(-args.steps // T), args.min_launches) * T
        n_warm = -(-args.warmup // T) * T
    for i in range(0, args.adapt_iters, T):
        eng.warmup_steps(min(T, args.adapt_iters - i))
    if args.phase == "sampling":
        eng.freeze()
    run_steps(0, n_warm, args.phase)
    fence()
    g_before = eng.total_grad_evals()
    # the dominant kernel's average launch duration: HIP events on the stream it is launched on, over the timed region
    # -- one pair around the K launches (default) or one pair per launch
    per_launch = args.per_launch_events or world > 1   # (with collectives on the stream the region is not kernel time)
    if per_launch:
        eng.timing_reset()
    else:
        eng.region_begin()
    t0 = time.perf_counter()
    launches = run_steps(0, n_steps, args.phase)
    if not per_launch:
        region_total_ms, region_launches = eng.region_ms()   # (waits for the last launch: part of the fence anyway)
    fence()
    elapsed = time.perf_counter() - t0
    g_after = eng.total_grad_evals()
    ktimes = eng.kernel_times_ms() if per_launch else [region_total_ms / max(region_launches, 1)]
    groups = eng.chain_groups
    timing_method = ("one pair of HIP events per launch" if per_launch else
                     f"one pair of HIP events around the {region_launches} launches of the timed region (gaps included)")
    if groups > 1:
        # a launch is `groups` kernels running side by side on their own streams, one contiguous block of chains each
        # (wn_config::chain_groups): bytes and flops per launch count all of them, the duration is the launch's share
        # of the region -- which is also what each of the overlapping kernels lasts in a kernel trace
        timing_method += (f"; a launch = {groups} concurrent kernels (chain groups), each lasting about the whole launch"
                          if not per_launch else f"; around group 0's kernel of the launch's {groups} concurrent ones")

    grad_evals = g_after - g_before
    legs = None
    if world > 1:
        # Two more legs of the same K steps, so that the kernel's scaling and the exchange's cost can be told apart in
        # ONE run (the line's `value` stays the full step: compute + every gathered draw): compute only, then the
        # exchange only (the same blocks, nobody computing).
        def leg(**kw):
            fence()
            t1 = time.perf_counter()
            run_steps(0, n_steps, args.phase, **kw)
            fence()
            dt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            return float(dt.item())

        g0 = eng.total_grad_evals()
        compute_s = leg(exchange=False)
        gc = torch.tensor([eng.total_grad_evals() - g0], dtype=torch.int64, device="cuda")
        dist.all_reduce(gc, op=dist.ReduceOp.SUM)
        exchange_s = leg(compute=False)
        legs = {"compute_s": compute_s, "compute_grad_evals": int(gc.item()), "exchange_s": exchange_s}
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        g = torch.tensor([grad_evals], dtype=torch.int64, device="cuda")
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        total_grad_evals = int(g.item())
    else:
        total_grad_evals = grad_evals

    if rank == 0:
        avg_kernel_ms = float(np.mean(ktimes)) if len(ktimes) else float("nan")
        kernel_s = avg_kernel_ms * 1e-3
        evals_per_launch = grad_evals / max(launches, 1)   # this rank's launches
        algorithmic_bytes = 56.0 * D * evals_per_launch
        algorithmic_gbps = algorithmic_bytes / kernel_s / 1e9
        streaming = bool(eng.streaming)
        traffic, traffic_source = measured_traffic(args, D, C, n_steps / max(launches, 1))
        if streaming:
            design_b = 16.0 if eng.held_tiles > 0 else 40.0
            roofline = {"bound": "hbm", "achieved": algorithmic_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": algorithmic_gbps / HBM_PEAK_GBPS, "traffic": traffic,
                        "traffic_source": traffic_source,
                        "traffic_frac": (traffic / kernel_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                        "kernel": "wn::transition_kernel_mem", "avg_launch_ms": avg_kernel_ms, "avg_launch_ms_from": timing_method,
                        "algorithmic_bytes_per_launch": algorithmic_bytes,
                        # the same launch priced on what the kernel has to move by design.  Both ends streamed: theta
                        # and rho in and out and the inverse mass in, the element-wise gradient recomputed = 40*D.  The
                        # moving end held in registers, the inverse mass in LDS (engine.held_tiles > 0): only the new
                        # state goes out = 16*D (the U-turn tests' far ends come on top in either case)
                        "design_bytes": {"per_grad_eval_per_dim": design_b, "GBps": algorithmic_gbps * design_b / 56.0,
                                         "frac": algorithmic_gbps * design_b / 56.0 / HBM_PEAK_GBPS},
                        # a model whose gradient needs sums over all coordinates or its neighbours (funnel, rw1) takes
                        # TWO passes per micro step: theta, rho and the inverse mass in and theta, rho out in the
                        # first, theta, rho in and rho out in the second = 72*D by design
                        "two_pass_floor": ({"per_grad_eval_per_dim": 72, "GBps": algorithmic_gbps * 72.0 / 56.0,
                                            "frac": algorithmic_gbps * 72.0 / 56.0 / HBM_PEAK_GBPS}
                                           if args.model in ("funnel", "rw1") and eng.held_tiles == 0 else None),
                        "note": "achieved / frac price the launch at SURVEY.md section 8(d)'s 56*D bytes per grad-eval (theta, "
                                "rho, gradient and inverse mass read; theta, rho, gradient written).  The one-pass streaming "
                                "kernel never stores a gradient (element-wise: recomputed) and, when it fits, keeps the "
                                "inverse mass in LDS for the whole transition, so it moves 32-40*D plus the U-turn tests' "
                                "span ends -- 16*D plus those where the trajectory's moving end stays in registers "
                                "(geometry.held_tiles > 0) --: frac exceeds 1 on the 56*D definition.  The bandwidth actually drawn is "
                                "traffic_frac (rocprofv3 PMC; part of it served by the 256 MiB Infinity Cache); models "
                                "whose gradient needs two passes per micro step (funnel, rw1) move 72*D"}
        else:
            flops = FLOPS_PER_GRAD_EVAL_PER_DIM * D * evals_per_launch
            tf = flops / kernel_s / 1e12
            roofline = {"bound": "fp64-valu", "achieved": tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                        "frac": tf / FP64_VALU_PEAK_TF, "traffic": traffic, "traffic_source": traffic_source,
                        "traffic_frac": (traffic / kernel_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                        "kernel": "wn::transition_kernel_chip", "avg_launch_ms": avg_kernel_ms, "avg_launch_ms_from": timing_method,
                        "algorithmic": {"bytes_per_launch": algorithmic_bytes, "GBps_equivalent": algorithmic_gbps,
                                        "ratio_to_hbm_peak": algorithmic_gbps / HBM_PEAK_GBPS,
                                        "note": "56*D bytes per grad-eval by the metric's definition; the trajectory "
                                                "end lives in VGPRs and the span pool in LDS, so these bytes never "
                                                "reach HBM and this is not a bandwidth"},
                        "note": "useful fp64 flops = 10*D per grad-eval (SURVEY.md section 8d's count, whether or not a "
                                "multiply-add is issued as one instruction); the kernel is bound by dependent latency "
                                "and issue of the fp64 vector pipe -- one wavefront per SIMD -- not by HBM"}
            assert roofline["frac"] <= 1.0 and (roofline["traffic_frac"] or 0.0) <= 1.0
        out = {
            "metric": "leapfrog grad-evals/sec (all chains)",
            "value": total_grad_evals / elapsed,
            "unit": "grad-evals/s",
            "n_gpus": world,
            "steps": n_steps,
            "warmup": n_warm,
            "steps_requested": steps_requested,
            "warmup_requested": warmup_requested,
            "ms_per_step": elapsed / max(n_steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{total_chains} chains x {D}-dim {args.model} ({C} on this GPU), default SamplingConfig, "
                            f"{args.adapt_iters} on-device adaptive warmup transitions then timed {args.phase} transitions",
                "chains_per_gpu": C, "global_chains": total_chains, "dim": D, "model": args.model,
                "phase": args.phase, "parallelism": f"chains sharded over {world} GPU(s)"
                                                    + (f", {'RCCL' if args.backend == 'nccl' else 'gloo'} all-gather of draws every "
                                                       f"{args.gather_every} launch(es), one collective per block of {T} draw planes" if world > 1 else ""),
                "geometry": {"lanes_per_chain": eng.lanes, "dim_padded": eng.dim_padded,
                             "workgroups": eng.workgroups, "lds_pool_vectors": eng.lds_vectors,
                             "reserved_cus": reserved, "chain_groups": groups,
                             "held_tiles": eng.held_tiles if eng.streaming else 0},
                "grad_evals_per_transition_per_chain": grad_evals / max(n_steps, 1) / C,
                "transitions_per_launch": T, "launches": launches,
                "arithmetic": ("fused multiply-adds in the integrator (as an FMA-target build of the reference)"
                               if cfg.fused_multiply_add else
                               "every product rounded (the reference's x86-64 -O3 element-wise bits)"),
                "csrc_sha": csrc_sha(),
                "stream_version": wa.stream_version(),
            },
            "roofline": roofline,
        }
        if legs is not None:
            full_ms = elapsed / max(n_steps, 1) * 1e3
            comp_ms = legs["compute_s"] / max(n_steps, 1) * 1e3
            out["value_compute_only"] = legs["compute_grad_evals"] / legs["compute_s"]
            out["ms_per_step_compute_only"] = comp_ms
            out["exchange_ms_per_step"] = {
                "exposed": full_ms - comp_ms,                                     # what the gathers add to a step
                "alone": legs["exchange_s"] / max(n_steps, 1) * 1e3,           # the same collectives with nobody computing
                "bytes_per_step_per_rank_in": (world - 1) * gather.rows * D * 8 / max(args.gather_every, 1),
                "method": args.gather_method if args.backend == "nccl" else "gloo (host staging)",
                "library_algorithm": rccl_algorithm_seen() if args.backend == "nccl" else None,
                "note": "three legs of the same K steps in one run: full (value), compute only (value_compute_only), "
                        "exchange only; DESIGN.md section 6 holds the prediction these are to confirm or refute"}
        # (the checker's legs must not cost the line its measurement: a failure is reported in place)
        if world == 1 and not args.no_parity_gate:
            try:
                out["parity_gate"] = parity_gate(args, D, cfg_kwargs)
            except Exception as exc:
                out["parity_gate"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and cpu and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, D)
            except Exception as exc:
                out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
    eng.close()
    del gather
    torch.cuda.empty_cache()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
