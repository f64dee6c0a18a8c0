"""Multi-GPU driver logic: chains shard across ranks; the exchanges are an all-gather of each iteration's draws
and the controllers' few-double all-reduces (RCCL over xGMI when the process group is `nccl`; `gloo` in the CPU
tests).

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
    """Double-buffered asynchronous all-gather of the per-iteration draw plane.

    Rank r writes its shard's draws into `buffer(it)` ([rows, D]; rows = the largest shard, so that every rank
    contributes an equally sized block -- `all_gather_into_tensor` needs that -- and uneven shards cost only padding
    rows).  `result(it)` is the gathered [total_chains, D] block in global chain order.

    transitions > 1: one launch produces that many consecutive draw planes (wn_engine_sample_steps); `buffer(it)` is
    then [transitions, rows, D], ONE collective moves the whole block (fewer, larger exchanges) and `result(it)` is
    [transitions, total_chains, D]."""

    def __init__(self, dist, world: int, rank: int, total_chains: int, dim: int, device, dtype, counts=None,
                 transitions: int = 1, method: str = "collective"):
        import torch

        if method not in ("collective", "p2p"):
            raise ValueError("gather method must be 'collective' or 'p2p'")
        self.method = method
        self.dist, self.world, self.rank = dist, world, rank
        self.counts = list(counts) if counts is not None else [shard_chains(total_chains, r, world)[1]
                                                                for r in range(world)]
        self.rows = max(self.counts)
        self.even = all(c == self.rows for c in self.counts)
        self.transitions = int(transitions)
        if self.transitions < 1:
            raise ValueError("transitions per launch must be at least 1")
        block = (self.rows, dim) if self.transitions == 1 else (self.transitions, self.rows, dim)
        self.local = [torch.empty(block, dtype=dtype, device=device) for _ in range(2)]
        # (every rank's block concatenated along the first axis: [world * T, rows, D] or [world * rows, D])
        self.gathered = ([torch.empty((world * block[0],) + block[1:], dtype=dtype, device=device) for _ in range(2)]
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
        """Start gathering iteration `it`'s draws (the whole block of a multi-transition launch); overlaps the next
        launch."""
        if self.world == 1:
            return None
        b = it & 1
        if self.method == "collective":   # whatever algorithm the library picks (RCCL: ring / tree / direct)
            self.pending[b] = self.dist.all_gather_into_tensor(self.gathered[b], self.local[b], async_op=True)
            return self.gathered[b]
        # all-pairs, direct: this rank's block goes to every peer as its own point-to-point transfer (and the peers'
        # blocks arrive the same way), one grouped launch.  On a fully connected xGMI node every pair has a link of its
        # own, so the W - 1 inbound blocks arrive on W - 1 links at once -- the time of ONE block over one link, where a
        # ring all-gather forwards W - 1 blocks over each link one after the other.
        g = self.gathered[b].view((self.world,) + tuple(self.local[b].shape))
        g[self.rank].copy_(self.local[b])
        ops = []
        for r in range(self.world):
            if r != self.rank:
                ops.append(self.dist.P2POp(self.dist.isend, self.local[b], r))
                ops.append(self.dist.P2POp(self.dist.irecv, g[r], r))
        self.pending[b] = _Requests(self.dist.batch_isend_irecv(ops))
        return self.gathered[b]

    def result(self, it: int):
        """Gathered draws of iteration `it` in global chain order ([total_chains, D])."""
        import torch

        b = it & 1
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        if self.transitions > 1:   # [world * T, rows, D] -> [T, total_chains, D]
            if self.world == 1:
                return self.local[b][:, : self.counts[0]]
            g = self.gathered[b].view(self.world, self.transitions, self.rows, -1)
            return torch.cat([g[r, :, :c] for r, c in enumerate(self.counts)], dim=1)
        if self.world == 1:
            return self.local[b][: self.counts[0]]
        if self.even:
            return self.gathered[b]
        return torch.cat([self.gathered[b][r * self.rows: r * self.rows + c] for r, c in enumerate(self.counts)])

    def drain(self):
        for b in (0, 1):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None


class _Requests:
    """The requests of one grouped point-to-point exchange behind the `.wait()` of a collective's work handle."""

    def __init__(self, reqs):
        self.reqs = list(reqs)

    def wait(self):
        for r in self.reqs:
            r.wait()


def _all_reduce(dist, values, op, device):
    import torch

    t = torch.as_tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=op)
    return t.cpu().numpy()


def global_rhat(dist, engine, device="cpu") -> float:
    """R-hat of the log density over the chains of ALL ranks (sampler.hpp:132-145): each rank reduces its chains'
    Welford statistics to three doubles, one all-reduce, one more double for the variance of the chain means."""
    s = _all_reduce(dist, engine.lp_sums(), dist.ReduceOp.SUM, device)          # sum means, sum variances, chains
    q = _all_reduce(dist, [engine.lp_sq_dev(s[0] / s[2])], dist.ReduceOp.SUM, device)[0]
    variance_of_means = q / (s[2] - 1)
    mean_of_variances = s[1] / s[2]
    return float((1 + variance_of_means / mean_of_variances) ** 0.5)


def global_warmup_spread(dist, engine, total_chains: int, device="cpu"):
    """(max rel. step distance, max rel. mass distance) from the geometric means over the chains of ALL ranks
    (adapt.hpp:193-221): D+1 doubles all-reduced (SUM), then 2 doubles (MAX)."""
    import numpy as np

    sum_log_step, colsum = engine.warmup_sums()
    tot = _all_reduce(dist, np.concatenate([[sum_log_step], colsum]), dist.ReduceOp.SUM, device)
    rel_step, rel_mass = engine.warmup_max_rel(float(tot[0]), tot[1:], total_chains)
    m = _all_reduce(dist, [rel_step, rel_mass], dist.ReduceOp.MAX, device)
    return float(m[0]), float(m[1])
