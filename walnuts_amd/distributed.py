"""Multi-GPU driver logic: chains shard across ranks, the only exchange is an all-gather of each iteration's
draws (RCCL over xGMI when the process group is `nccl`; `gloo` in the CPU tests).

The reference has no distributed layer (thread-per-chain on one host, adapt.hpp:249-254, sampler.hpp:182-187);
chains never exchange state inside a transition and per-chain tuning is never pooled (adapt.hpp:257-258), so the
partition is embarrassing: rank r owns global chains [r*C, (r+1)*C) and keys its random streams by the GLOBAL
chain id, which makes results independent of the number of ranks.
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def shard_chains(total_chains: int, rank: int, world: int) -> Tuple[int, int]:
    """-> (first global chain id, chain count) of `rank`; contiguous blocks, remainder to the low ranks."""
    base, rem = divmod(total_chains, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


class DrawGather:
    """Double-buffered asynchronous all-gather of the per-iteration draw plane [C_local, D]."""

    def __init__(self, dist, world: int, chains_local: int, dim: int, device, dtype):
        import torch

        self.dist, self.world = dist, world
        self.local = [torch.empty((chains_local, dim), dtype=dtype, device=device) for _ in range(2)]
        self.gathered = ([torch.empty((world * chains_local, dim), dtype=dtype, device=device) for _ in range(2)]
                         if world > 1 else None)
        self.pending: List[Optional[object]] = [None, None]

    def buffer(self, it: int):
        """The draw plane iteration `it` writes into (waits for the collective that last read it)."""
        b = it & 1
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        return self.local[b]

    def launch(self, it: int):
        """Start gathering iteration `it`'s draws; overlaps the next transition."""
        if self.world == 1:
            return None
        b = it & 1
        self.pending[b] = self.dist.all_gather_into_tensor(self.gathered[b], self.local[b], async_op=True)
        return self.gathered[b]

    def result(self, it: int):
        """Gathered draws of iteration `it` ([world*C_local, D]; the local plane when world == 1)."""
        b = it & 1
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        return self.local[b] if self.world == 1 else self.gathered[b]

    def drain(self):
        for b in (0, 1):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None
