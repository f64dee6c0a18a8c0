"""Stand-in rank for the launcher test: prints what a rank of bench.py would (noise + rank 0's JSON line)."""
import json
import os
import sys

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
print(f"rank {rank} of {world} up; argv {sys.argv[1:]}", flush=True)
n = int(os.environ.get("FAKE_N_GPUS", world))
if rank == 0:
    print(json.dumps({"metric": "fake", "n_gpus": n, "argv": sys.argv[1:],
                      "master": os.environ.get("MASTER_ADDR")}), flush=True)
sys.exit(int(os.environ.get("FAKE_RC", "0")) if rank == world - 1 else 0)
