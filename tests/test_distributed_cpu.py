"""CPU tier, world_size 2 over gloo: the N>1 driver logic (chain sharding, double-buffered all-gather of draws,
the controllers' all-reduced statistics) run on the test-only workgroup emulation gives exactly the chains of a single-rank run."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "cpusim")]
import build as simbuild
import walnuts_amd as wa
from walnuts_amd.distributed import DrawGather, shard_chains, global_rhat, global_warmup_spread

SIM = simbuild.build()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
D, TOTAL, ITERS = 6, int(os.environ.get("WN_TEST_TOTAL", "4")), 3
first, count = shard_chains(TOTAL, rank, world)
pos = np.random.default_rng(3).normal(size=(TOTAL, D))

def make(first, count):
    e = wa.DeviceEngine(wa.MODEL_STD_NORMAL, D, count, wa.default_config(SIM), lib_path=SIM)
    e.set_positions(pos[first:first + count]); e.set_step_sizes(0.6); e.seed_chains(99, first)
    return e

eng = make(first, count)
gather = DrawGather(dist, world, rank, TOTAL, D, "cpu", torch.float64)
seen = []
spread = None
for it in range(ITERS):
    plane = gather.buffer(it)
    if it == 2:
        spread = global_warmup_spread(dist, eng, TOTAL)   # the warmup controller's statistic over both ranks
        eng.freeze()
    eng.warmup_step(plane.data_ptr(), D) if it < 2 else eng.sample_step(plane.data_ptr(), D)
    eng.synchronize()
    gather.launch(it)
    seen.append(gather.result(it).clone().numpy())
gather.drain()
eng.sample_step(); eng.sample_step()
rhat = global_rhat(dist, eng)                             # the sampling controller's statistic over both ranks
# one launch of three transitions per chain (wn_engine_sample_steps): ONE collective moves the block of three planes
blocks = DrawGather(dist, world, rank, TOTAL, D, "cpu", torch.float64, transitions=3)
block = blocks.buffer(0)
assert tuple(block.shape) == (3, blocks.rows, D)
eng.sample_steps(3, block.data_ptr(), D, blocks.rows * D)
eng.synchronize()
blocks.launch(0)
fused = blocks.result(0).clone().numpy()
blocks.drain()
# the same block exchanged all-pairs (grouped point-to-point transfers instead of the library's all-gather)
direct = DrawGather(dist, world, rank, TOTAL, D, "cpu", torch.float64, transitions=3, method="p2p")
direct.buffer(0).copy_(block)
direct.launch(0)
assert np.array_equal(direct.result(0).numpy(), fused), "p2p gather differs from the collective"
direct.drain()
if rank == 0:
    ref = make(0, TOTAL)
    for it in range(ITERS):
        buf = np.empty((TOTAL, D))
        ptr = buf.ctypes.data
        if it == 2:
            ref_spread = ref.warmup_spread()
            ref.freeze()
        ref.warmup_step(ptr, D) if it < 2 else ref.sample_step(ptr, D)
        ref.synchronize()
        assert np.array_equal(buf, seen[it]), (it, buf, seen[it])
    ref.sample_step(); ref.sample_step()
    assert np.allclose(spread, ref_spread, rtol=1e-12), (spread, ref_spread)
    assert abs(rhat - ref.rhat()) <= 1e-12 * rhat, (rhat, ref.rhat())
    assert fused.shape == (3, TOTAL, D)
    for k in range(3):
        buf = np.empty((TOTAL, D))
        ref.sample_step(buf.ctypes.data, D)
        ref.synchronize()
        assert np.array_equal(buf, fused[k]), (k, buf, fused[k])
    print("DISTRIBUTED_OK")
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_chains_partition():
    sys.path.insert(0, ROOT)
    from walnuts_amd.distributed import shard_chains

    for total, world in ((65536, 8), (10, 3), (7, 8), (262144, 8)):
        parts = [shard_chains(total, r, world) for r in range(world)]
        assert sum(c for _, c in parts) == total
        nxt = 0
        for first, count in parts:
            assert first == nxt
            nxt += count


def _run_world(tmp_path, world, total):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   WN_TEST_TOTAL=str(total), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=800)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "DISTRIBUTED_OK" in outs[0]


@pytest.mark.timeout(900)
def test_two_rank_gloo_run_matches_single_rank(tmp_path):
    _run_world(tmp_path, 2, 4)


@pytest.mark.timeout(900)
def test_four_rank_gloo_run_with_uneven_shards_matches_single_rank(tmp_path):
    # 6 chains over 4 ranks: shards of 2, 2, 1, 1 -- the gathered block is still the single-rank run's, in order
    _run_world(tmp_path, 4, 6)
