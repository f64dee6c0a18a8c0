#!/usr/bin/env python3
"""bench.py -- leapfrog grad-evals/sec of the GPU-resident Walnuts engine (BASELINE.json metric).

A "step" is one MCMC transition (walnuts.hpp:520-563) of EVERY chain: one launch of the persistent
transition kernel.  Default workload = BASELINE.json's headline: 65 536 chains x 1 024-dim standard normal
per GPU, default SamplingConfig, parameters adapted by `--adapt-iters` on-device warmup transitions
(untimed), then W untimed + K timed sampling transitions.  Inputs are generated on the device (counter-based
stream) and are resident in HBM when the timed region starts.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Rank 0 prints ONE JSON line.  `value` = gradient evaluations of all chains on all ranks in the timed region
/ max-over-ranks wall time.  `roofline.achieved` = 56*D bytes per gradient evaluation (read
theta, rho, grad, inverse mass, write theta, rho, grad) x gradient evaluations per launch / average HIP-event
duration of the transition kernel on the stream it is launched on.  `cpu_baseline` = the oracle (a CPU port
of the reference algorithm, reference arithmetic order, libm) on the host cores for a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--chains", type=int, default=65536, help="chains PER GPU (weak scaling)")
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--model", default="std_normal", choices=["std_normal", "diag_normal", "ill_normal", "funnel"],
                    help="diag_normal: sigma_d = 1 + (d mod 16) (config #4); ill_normal: sigma_d = d + 1 (config #2, "
                         "examples/examples.cpp:20-31)")
    ap.add_argument("--adapt-iters", type=int, default=100, help="untimed adaptive warmup transitions")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--waves-per-chain", type=int, default=0)
    ap.add_argument("--elems-per-lane", type=int, default=0)
    ap.add_argument("--workgroups-per-cu", type=int, default=0)
    ap.add_argument("--lds-vectors", type=int, default=-1)
    ap.add_argument("--reg-vectors", type=int, default=-1)
    ap.add_argument("--reserved-cus", type=int, default=-1,
                    help="CUs left free for the RCCL all-gather kernels (default: 0 on one GPU, 16 otherwise)")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="all-gather the draws of every k-th transition (1 = every draw, the north star's exchange)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend (gloo only to exercise the N>1 code path when ranks share one GPU)")
    ap.add_argument("--phase", default="sampling", choices=["sampling", "warmup"],
                    help="which transition kind is timed")
    return ap.parse_args()


def model_setup(name, D):
    import walnuts_amd as wa

    if name == "std_normal":
        return wa.MODEL_STD_NORMAL, None
    if name == "funnel":
        return wa.MODEL_FUNNEL, None
    if name == "ill_normal":
        return wa.MODEL_DIAG_NORMAL, np.array([(d + 1.0) ** 2 for d in range(D)])
    return wa.MODEL_DIAG_NORMAL, np.array([(1.0 + (d % 16)) ** 2 for d in range(D)])


def measured_traffic(args, D):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes recorded under profiles/ (counters
    are collected in their own runs, never inside a timed bench run); None when this workload was not profiled."""
    try:
        entries = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except OSError:
        return None, None
    for e in entries:
        if (e["model"], e["chains"], e["dim"], e["phase"]) == (args.model, args.chains, D, args.phase):
            return e["bytes_per_launch"], e["source"]
    return None, None


def cpu_baseline(args, D):
    """The oracle timed on the host cores: reference arithmetic order (left-to-right sums, libm), one thread per
    chain block.  Bounded: 8 chains per core, ~args.cpu_seconds of sampling transitions."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wno

    cores = os.cpu_count() or 1
    chains = 8 * cores
    om = {"std_normal": wno.MODEL_STD_NORMAL, "diag_normal": wno.MODEL_DIAG_NORMAL, "ill_normal": wno.MODEL_DIAG_NORMAL,
          "funnel": wno.MODEL_FUNNEL}[args.model]
    _, params = model_setup(args.model, D)
    cfg = wno.default_config(rng_mode=wno.RNG_PHILOX, math_mode=wno.MATH_LIBM, reduce_lanes=0)
    e = wno.Engine(om, D, chains, cfg, params=params)
    e.init_positions(args.seed, 0, 2.0)
    e.init_masses_from_grad(1e-5)
    e.set_step_sizes(1.0)
    e.adapt_step(args.seed, 0)
    e.seed_chains(args.seed + 1, 0)
    adapt = min(args.adapt_iters, 100)
    for _ in range(adapt):
        e.warmup_step(cores)
    if args.phase == "sampling":
        e.freeze()
        step = lambda: e.sample_step(cores)
    else:
        step = lambda: e.warmup_step(cores)
    step()
    g0 = int(e.grad_evals().sum())
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds or n >= 100000:
            break
    g1 = int(e.grad_evals().sum())
    return {"value": (g1 - g0) / dt, "unit": "grad-evals/s", "cores": cores, "kind": "port",
            "per_core": (g1 - g0) / dt / cores,
            "sample": f"{chains} chains x {D}-dim {args.model}, {adapt} adaptive warmup transitions then {n} "
                      f"{args.phase} transitions in {dt:.1f} s on {cores} threads (oracle: CPU port of the reference "
                      "algorithm, sequential sums + libm; the reference Eigen build is unavailable: Eigen is not "
                      "vendored, CMakeLists.txt:28-41)"}


def main():
    args = parse()
    import torch

    import walnuts_amd as wa

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    D, C = args.dim, args.chains
    model_id, params = model_setup(args.model, D)
    # the transition kernel is persistent and would hold every CU for the whole step; with more than one GPU a
    # few CUs stay free so that RCCL's all-gather of the previous draws really runs underneath it
    reserved = args.reserved_cus if args.reserved_cus >= 0 else (16 if world > 1 else 0)
    cfg = wa.default_config(device=local_rank, waves_per_chain=args.waves_per_chain, elems_per_lane=args.elems_per_lane,
                            workgroups_per_cu=args.workgroups_per_cu, lds_vectors=args.lds_vectors,
                            reserved_cus=reserved, reg_vectors=args.reg_vectors)
    eng = wa.DeviceEngine(model_id, D, C, cfg, params=params)
    if world > 1:
        # kernels on torch's current stream: RCCL collectives on the draws are then ordered after them by torch
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
    chain0 = rank * C
    # InitConfigBuilder on the device: positions ~ N(0, 2^2) (init_radius, pyfunc.py:57), masses from the
    # gradient with smoothing 1e-5, step-size search from step_size_init = 1.0
    eng.init_positions(args.seed, chain0, 2.0)
    eng.init_masses_from_grad(1e-5)
    eng.set_step_sizes(1.0)
    eng.adapt_step(args.seed, chain0)
    eng.seed_chains(args.seed + 1, chain0)

    from walnuts_amd.distributed import DrawGather

    gather = DrawGather(dist, world, C, D, "cuda", torch.float64)

    def one_step(i, timed_phase):
        plane = gather.buffer(i)
        if timed_phase == "warmup":
            eng.warmup_step(plane.data_ptr(), D)
        else:
            eng.sample_step(plane.data_ptr(), D)
        # the path's only exchange: all-gather of this iteration's draws over xGMI, overlapped with the next
        # transition (no-op on one GPU)
        if i % args.gather_every == 0:
            gather.launch(i)

    def fence():
        gather.drain()
        if world > 1:
            dist.barrier()
        eng.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.adapt_iters):
        eng.warmup_step()
    if args.phase == "sampling":
        eng.freeze()
    for i in range(args.warmup):
        one_step(i, args.phase)
    fence()
    g_before = eng.total_grad_evals()
    eng.timing_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i, args.phase)
    fence()
    elapsed = time.perf_counter() - t0
    g_after = eng.total_grad_evals()
    ktimes = eng.kernel_times_ms()

    grad_evals = g_after - g_before
    if world > 1:
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
        bytes_per_launch = 56.0 * D * grad_evals / max(args.steps, 1)  # this rank's launches
        achieved = bytes_per_launch / (avg_kernel_ms * 1e-3) / 1e9
        traffic, traffic_source = measured_traffic(args, D)
        streaming = bool(eng.streaming)
        out = {
            "metric": "leapfrog grad-evals/sec (all chains)",
            "value": total_grad_evals / elapsed,
            "unit": "grad-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{C} chains x {D}-dim {args.model} per GPU, default SamplingConfig, "
                            f"{args.adapt_iters} on-device adaptive warmup transitions then timed {args.phase} transitions",
                "chains_per_gpu": C, "global_chains": C * world, "dim": D, "model": args.model,
                "phase": args.phase, "parallelism": f"chains sharded over {world} GPU(s)"
                                                    + (f", {'RCCL' if args.backend == 'nccl' else 'gloo'} all-gather of draws every "
                                                       f"{args.gather_every} step(s)" if world > 1 else ""),
                "geometry": {"lanes_per_chain": eng.lanes, "dim_padded": eng.dim_padded,
                             "workgroups": eng.workgroups, "lds_pool_vectors": eng.lds_vectors,
                             "reserved_cus": reserved},
                "grad_evals_per_transition_per_chain": grad_evals / max(args.steps, 1) / C,
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "wn::transition_kernel_mem" if streaming else "wn::transition_kernel",
                         "avg_launch_ms": avg_kernel_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": ("algorithmic bytes = 56*D per grad-eval; the streaming kernels move theta, rho and the "
                                  "inverse mass per micro step and recompute the element-wise gradient"
                                  if streaming else
                                  "algorithmic bytes = 56*D per grad-eval; the trajectory end lives in VGPRs and the "
                                  "span pool in LDS, so measured HBM traffic is far below the algorithmic figure")},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, D)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
