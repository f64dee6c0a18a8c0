"""CPU tier: `python bench.py --gpus N` starts N ranks itself when no launcher did."""
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "helpers", "fake_rank.py")
sys.path.insert(0, ROOT)


def _args(gpus, backend="nccl"):
    return types.SimpleNamespace(gpus=gpus, backend=backend)


@pytest.mark.timeout(300)
def test_launcher_starts_n_ranks_and_forwards_rank0_json(capfd):
    import bench

    rc = bench.launch_ranks(_args(2), ["--gpus", "2", "--steps", "3"], script=FAKE, device_count=2)
    out, err = capfd.readouterr()
    assert rc == 0
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out                      # ONE JSON line on stdout, the ranks' chatter on stderr
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["argv"] == ["--gpus", "2", "--steps", "3"] and rec["master"] == "127.0.0.1"
    assert "rank 1 of 2 up" in err


@pytest.mark.timeout(300)
def test_launcher_refuses_too_few_devices_unless_gloo(capfd):
    import bench

    assert bench.launch_ranks(_args(4), [], script=FAKE, device_count=1) == 2
    assert "needs 4 GPUs" in capfd.readouterr().err
    assert bench.launch_ranks(_args(2, "gloo"), ["--gpus", "2", "--backend", "gloo"], script=FAKE, device_count=1) == 0


@pytest.mark.timeout(300)
def test_launcher_propagates_failure_and_checks_n_gpus(capfd, monkeypatch):
    import bench

    monkeypatch.setenv("FAKE_RC", "7")
    assert bench.launch_ranks(_args(2), [], script=FAKE, device_count=2) != 0
    monkeypatch.delenv("FAKE_RC")
    capfd.readouterr()
    monkeypatch.setenv("FAKE_N_GPUS", "1")             # ranks that report another world size: refused, nothing forwarded
    assert bench.launch_ranks(_args(2), [], script=FAKE, device_count=2) == 3
    assert '"metric"' not in capfd.readouterr().out


@pytest.mark.timeout(300)
def test_plain_invocation_without_gpus_fails_loudly():
    # this container has no GPU: --gpus 2 must not silently become a one-rank run
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True)
    import torch

    if torch.cuda.device_count() < 2:
        assert p.returncode == 2 and "needs 2 GPUs" in p.stderr and p.stdout.strip() == ""
